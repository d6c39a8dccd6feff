"""One-off: the full cfg-2 model (PNA H=128, L=6) on a batch large enough to select every large-batch kernel, three-way
against the CPU oracle in fp32 and fp64 (tests/parity_util.compare_with_oracle).  Too slow for the suite."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnnepcsaft_amd.data import default_config, synthetic_batch
from tests.parity_util import compare_with_oracle, assert_as_close_as_cpu_fp32
cfg = default_config(2)
graphs = int(os.environ.get("GRAPHS", "1024"))
res = compare_with_oracle(cfg, synthetic_batch(graphs, 2), device="cuda:0")
print(json.dumps({k: (float(v) if not isinstance(v, str) else v) for k, v in res.items()}, indent=1))
assert res["loss_rel"] <= 1e-5
assert_as_close_as_cpu_fp32(res)
print("ok")
