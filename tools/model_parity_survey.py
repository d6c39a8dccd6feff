"""Prints whole-model parity numbers (HIP vs fp64 oracle, CPU fp32 oracle vs fp64, the recorded reference envelope,
per-layer intermediates) for the cases of tests/model_cases.py, and which cases the HIP path already holds inside the
PERMUTATION-ONLY envelope (conditioning.json "max_perm") with a 0.7 margin -> gpurun_out/tight_cases.json (copied to
tests/golden/tight_cases.json, which tests/parity_util.py reads).
Diagnostic, GPU box:  python tools/model_parity_survey.py [case ...]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.model_cases import MODEL_CASES, build_case  # noqa: E402
from tests.parity_util import compare_with_oracle, parity_bounds  # noqa: E402

MARGIN = 0.7

if __name__ == "__main__":
    tight = []
    for name in sys.argv[1:] or list(MODEL_CASES):
        t0 = time.time()
        cfg, batch, target = build_case(name)
        r = compare_with_oracle(cfg, batch, device="cuda:0", target=target)
        hip = {"pred": r["pred_hip64"], "loss": r["loss_hip64"], "grad_l2": r["grad_l2_hip64"], "grad_max": r["grad_max_hip64"],
               "inter": max(r["inter_hip64"].values()), "dinter": max(r["dinter_hip64"].values())}
        b = parity_bounds(r, name, kind="max")
        bp = parity_bounds(r, name, kind="max_perm")
        flag = " ".join(f"{k}={hip[k]:.1e}/{b[k]:.1e}{'!' if hip[k] > b[k] else ''}" for k in hip)
        ok_perm = all(hip[k] <= MARGIN * bp[k] for k in hip)
        if ok_perm:
            tight.append(name)
        print(f"{name:28s} {time.time() - t0:5.1f}s loss_rel {r['loss_rel']:.1e} pred_rel {r['pred_rel']:.1e} | hip/bound {flag}"
              f" | inside 0.7 x permutation-only bound: {ok_perm} (" +
              " ".join(f"{k}={hip[k] / bp[k]:.2f}" for k in hip) + ")", flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump({"what": "cases the HIP path holds inside 0.7 x the permutation-only envelope bound (tools/model_parity_survey.py)",
               "cases": tight}, open(os.path.join(ROOT, "gpurun_out", "tight_cases.json"), "w"), indent=1)
    print("tight cases:", len(tight), "of", len(sys.argv[1:] or list(MODEL_CASES)))
