"""Prints the single-layer parity numbers (HIP vs fp64 oracle, CPU fp32 oracle vs fp64) for the cases of
tests/test_conv_gpu.py.  Diagnostic: `python tools/conv_parity_survey.py [case ...]` on a GPU box."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from tests.test_conv_gpu import CASES, build_case  # noqa: E402
from tests.conv_parity import run_conv_case  # noqa: E402

if __name__ == "__main__":
    names = sys.argv[1:] or list(CASES)
    for name in names:
        t0 = time.time()
        kw, batch = build_case(name)
        r = run_conv_case(batch=batch, **kw)
        print(name, f"{time.time() - t0:.1f}s", json.dumps({k: (float(f"{v:.3g}") if isinstance(v, float) else v)
                                                               for k, v in r.items()}), flush=True)
