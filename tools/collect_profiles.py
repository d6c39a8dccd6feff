"""Copies the judged summaries of a profiling run from gpurun_out/ (scratch) into profiles/ (tracked).

usage: python tools/collect_profiles.py <stats_dir> <pmc_dir> <bench_json> [round_tag]
  stats_dir : rocprofv3 --kernel-trace --stats output (contains */*_kernel_stats.csv and bench.json)
  pmc_dir   : contains fetch/ and write/ outputs of the two --pmc passes
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(stats_dir, pmc_dir, bench_json, tag="r01"):
    prof = os.path.join(ROOT, "profiles")
    shutil.copy(bench_json, os.path.join(prof, f"bench_{tag}.json"))
    shutil.copy(glob.glob(os.path.join(stats_dir, "*", "*_kernel_stats.csv"))[0],
                os.path.join(prof, f"{tag}_kernel_stats_cfg2.csv"))
    shutil.copy(os.path.join(stats_dir, "bench.json"), os.path.join(prof, f"{tag}_bench_under_rocprof.json"))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for which in ("fetch", "write"):
        f = glob.glob(os.path.join(pmc_dir, which, "*", "*_counter_collection.csv"))[0]
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    with open(os.path.join(prof, f"{tag}_pmc_fetch_write_per_kernel.csv"), "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["kernel", "dispatches", "FETCH_SIZE_KB_avg", "WRITE_SIZE_KB_avg", "hbm_bytes_avg(2*FETCH+WRITE)"])
        for k in sorted(agg, key=lambda k: -sum(agg[k].get("FETCH_SIZE", [0]))):
            fs, ws = agg[k].get("FETCH_SIZE", []), agg[k].get("WRITE_SIZE", [])
            fa = sum(fs) / max(len(fs), 1)
            wa = sum(ws) / max(len(ws), 1)
            w.writerow([k[:90], max(len(fs), len(ws)), f"{fa:.1f}", f"{wa:.1f}", f"{(2 * fa + wa) * 1024:.0f}"])
    key = [k for k in agg if k.startswith("void k_pna_agg_fwd")]
    if key:
        fs, ws = agg[key[0]]["FETCH_SIZE"], agg[key[0]]["WRITE_SIZE"]
        fa, wa = sum(fs) / len(fs), sum(ws) / len(ws)
        old = json.load(open(os.path.join(prof, "pmc_scatter.json")))
        old.update({"FETCH_SIZE_KB_avg": fa, "WRITE_SIZE_KB_avg": wa, "traffic_bytes_per_launch": (2 * fa + wa) * 1024,
                    "workload": f"bench.py cfg-2 (N=81920, E=163840, H=128), {len(fs)} dispatches per counter pass"})
        json.dump(old, open(os.path.join(prof, "pmc_scatter.json"), "w"), indent=1)
        print("scatter traffic/launch", (2 * fa + wa) * 1024, "algorithmic", old["algorithmic_bytes_per_launch"])


if __name__ == "__main__":
    main(*sys.argv[1:])
