"""Prints whole-model parity numbers (HIP vs fp64 oracle, CPU fp32 oracle vs fp64, the recorded reference envelope,
per-layer intermediates) for the cases of tests/model_cases.py.  Diagnostic, GPU box:  python tools/model_parity_survey.py [case ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.model_cases import MODEL_CASES, build_case  # noqa: E402
from tests.parity_util import compare_with_oracle, parity_bounds  # noqa: E402

if __name__ == "__main__":
    for name in sys.argv[1:] or list(MODEL_CASES):
        t0 = time.time()
        cfg, batch, target = build_case(name)
        r = compare_with_oracle(cfg, batch, device="cuda:0", target=target)
        b = parity_bounds(r, name)
        hip = {"pred": r["pred_hip64"], "loss": r["loss_hip64"], "grad_l2": r["grad_l2_hip64"], "grad_max": r["grad_max_hip64"],
               "inter": max(r["inter_hip64"].values()), "dinter": max(r["dinter_hip64"].values())}
        flag = " ".join(f"{k}={hip[k]:.1e}/{b[k]:.1e}{'!' if hip[k] > b[k] else ''}" for k in hip)
        print(f"{name:28s} {time.time() - t0:5.1f}s loss_rel {r['loss_rel']:.1e} pred_rel {r['pred_rel']:.1e} | hip/bound {flag}",
              flush=True)
