"""Copies the judged summaries of a profiling run (tools/profile_round.sh) from gpurun_out/<tag>/ (scratch) into
profiles/ (tracked).

usage: python tools/collect_profiles.py [tag]        (default r03; reads gpurun_out/<tag>/)
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_stats(src_dir, dst):
    f = sorted(glob.glob(os.path.join(src_dir, "*", "*_kernel_stats.csv")), key=os.path.getmtime)
    if f:  # gpurun merges every run's files into the same directory: the newest is this run's
        shutil.copy(f[-1], dst)
        return True
    return False


def main(tag="r03"):
    src = os.path.join(ROOT, "gpurun_out", tag)
    prof = os.path.join(ROOT, "profiles")
    shutil.copy(os.path.join(src, "bench.json"), os.path.join(prof, f"bench_{tag}.json"))
    kernel_stats(os.path.join(src, "stats"), os.path.join(prof, f"{tag}_kernel_stats_cfg2.csv"))
    kernel_stats(os.path.join(src, "stats_ns"), os.path.join(prof, f"{tag}_kernel_stats_cfg2_single_stream.csv"))
    for c in (3, 5):
        kernel_stats(os.path.join(src, f"stats_cfg{c}"), os.path.join(prof, f"{tag}_kernel_stats_cfg{c}_single_stream.csv"))
    for name in ("stats/bench.json", "stats_ns/bench.json"):
        if os.path.exists(os.path.join(src, name)):
            shutil.copy(os.path.join(src, name), os.path.join(prof, f"{tag}_bench_under_rocprof{'_single_stream' if 'ns' in name else ''}.json"))
    for c in (1, 3, 4, 5):
        f = os.path.join(src, f"bench_cfg{c}.json")
        if os.path.exists(f) and os.path.getsize(f) > 0:
            shutil.copy(f, os.path.join(prof, f"{tag}_bench_cfg{c}.json"))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for which in ("fetch", "write"):
        fs = sorted(glob.glob(os.path.join(src, "pmc", which, "*", "*_counter_collection.csv")), key=os.path.getmtime)
        if not fs:
            continue
        for r in csv.DictReader(open(fs[-1])):
            # bench.py also times the scatter kernel on a 16 384-graph batch (roofline.large): keep the cfg-2 launches
            # (81 920 atoms x 32 threads) apart from those
            name = r["Kernel_Name"]
            if name.startswith("void k_pna_agg_fwd") and int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0) != 81920 * 32:
                name += " [16384-graph batch]"
            agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    # matrix-pipe utilisation per kernel: SQ_VALU_MFMA_BUSY_CYCLES (sum over the 1024 SIMDs of 32 cycles per
    # v_mfma_f32_32x32x16_bf16) over GRBM_GUI_ACTIVE / 8 (cycles per XCD; rocprofv3 reports the sum over the 8 XCDs)
    mf = collections.defaultdict(lambda: collections.defaultdict(list))
    for which, counter in (("mfma", "SQ_VALU_MFMA_BUSY_CYCLES"), ("gui", "GRBM_GUI_ACTIVE")):
        fs = sorted(glob.glob(os.path.join(src, "pmc", which, "*", "*_counter_collection.csv")), key=os.path.getmtime)
        if not fs:
            continue
        for r in csv.DictReader(open(fs[-1])):
            if r["Counter_Name"] == counter:
                mf[r["Kernel_Name"]][counter].append(float(r["Counter_Value"]))
                mf[r["Kernel_Name"]][counter + ".ns"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    rows = []
    for k, d in mf.items():
        b, gu = d.get("SQ_VALU_MFMA_BUSY_CYCLES", []), d.get("GRBM_GUI_ACTIVE", [])
        if not b or not gu or sum(b) == 0:
            continue
        ba, ga = sum(b) / len(b), sum(gu) / len(gu)
        us = sum(d["GRBM_GUI_ACTIVE.ns"]) / len(gu) / 1e3
        rows.append((k, len(b), ba, ga, us, ga / 8 / us / 1e3, ba / (1024 * ga / 8)))
    if rows:
        with open(os.path.join(prof, f"{tag}_pmc_mfma_busy_per_kernel.csv"), "w", newline="") as fh:
            w = csv.writer(fh)
            w.writerow(["kernel", "dispatches", "SQ_VALU_MFMA_BUSY_CYCLES_avg", "GRBM_GUI_ACTIVE_avg", "duration_us_avg",
                        "clock_GHz(GUI_ACTIVE/8/duration)", "mfma_utilisation(BUSY/(1024*GUI_ACTIVE/8))"])
            for k, n, ba, ga, us, ghz, util in sorted(rows, key=lambda r: -r[2] * r[1]):
                # GRBM_GUI_ACTIVE also counts the idle cycles around a dispatch: for kernels shorter than ~50 us the
                # derived clock reads 2.8-4.7 GHz, so neither it nor the utilisation is reported
                short = us < 50.0
                w.writerow([k[:90], n, f"{ba:.0f}", f"{ga:.0f}", f"{us:.1f}", "n/a (< 50 us)" if short else f"{ghz:.2f}",
                            "n/a (< 50 us)" if short else f"{util:.3f}"])
    if agg:
        with open(os.path.join(prof, f"{tag}_pmc_fetch_write_per_kernel.csv"), "w", newline="") as fh:
            w = csv.writer(fh)
            w.writerow(["kernel", "dispatches", "FETCH_SIZE_KB_avg", "WRITE_SIZE_KB_avg", "hbm_bytes_avg(2*FETCH+WRITE)"])
            for k in sorted(agg, key=lambda k: -sum(agg[k].get("FETCH_SIZE", [0]))):
                fs_, ws_ = agg[k].get("FETCH_SIZE", []), agg[k].get("WRITE_SIZE", [])
                fa = sum(fs_) / max(len(fs_), 1)
                wa = sum(ws_) / max(len(ws_), 1)
                w.writerow([k[:90], max(len(fs_), len(ws_)), f"{fa:.1f}", f"{wa:.1f}", f"{(2 * fa + wa) * 1024:.0f}"])
        # the kernel that performs the scatter-aggregate inside the step: the fused edge kernel when it ran, else k_pna_agg_fwd
        key = [k for k in agg if k.startswith("k_pna_edge_fwd")] or \
              [k for k in agg if k.startswith("void k_pna_agg_fwd") and not k.endswith("batch]")]
        if key:
            fs_, ws_ = agg[key[0]]["FETCH_SIZE"], agg[key[0]]["WRITE_SIZE"]
            fa, wa = sum(fs_) / len(fs_), sum(ws_) / len(ws_)
            fused = key[0].startswith("k_pna_edge_fwd")
            N, E, H = 81920, 163840, 128
            old = {"kernel": "k_pna_edge_fwd" if fused else "k_pna_agg_fwd<4>",
                   "FETCH_SIZE_KB_avg": fa, "WRITE_SIZE_KB_avg": wa, "traffic_bytes_per_launch": (2 * fa + wa) * 1024,
                   "algorithmic_bytes_per_launch": 4 * E * H + 4 * E + 16 * N * H,
                   "own_algorithmic_bytes_per_launch": (24 * N * H + 16 * E + 4 * N + 8 * E * H) if fused else None,
                   "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and, in a separate pass, --pmc WRITE_SIZE; counters "
                             "are KB; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half the bytes of wide "
                             "coalesced streaming reads); WRITE_SIZE exact for 16-B/lane stores",
                   "conv": "PNA", "N": N, "E": E, "H": H, "round": tag,
                   "workload": f"bench.py cfg-2 (N={N}, E={E}, H={H}), {len(fs_)} dispatches per counter pass"}
            json.dump(old, open(os.path.join(prof, "pmc_scatter.json"), "w"), indent=1)
            print("scatter traffic/launch", (2 * fa + wa) * 1024, "algorithmic", old["algorithmic_bytes_per_launch"])


if __name__ == "__main__":
    main(*sys.argv[1:])
