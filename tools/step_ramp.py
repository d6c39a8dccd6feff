"""Per-step wall time of the bench step right after start-up (is the timed region of `bench.py --steps 20 --warmup 5`
in steady state?).  Prints ms per step for the first 60 steps, synchronising after each."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnnepcsaft_amd import dp, functional as Fn, ops
from gnnepcsaft_amd.data import calc_deg, default_config, synthetic_batch
from gnnepcsaft_amd.train.models import create_model
dev = torch.device("cuda:0")
s = torch.cuda.Stream(device=dev); torch.cuda.set_stream(s)
cfg = default_config(2)
b_cpu = synthetic_batch(4096, 2)
deg = calc_deg(b_cpu)
torch.manual_seed(0)
model = create_model(cfg, deg).to(dev).train()
model.model.validate_inputs = False
model.model.max_degree_hint = len(deg) - 1
flat = dp.FlatGradAllReduce(model)
Fn.set_grad_in_place(True); ops.set_wgrad_side_stream(True)
b = b_cpu.to(dev)
ts = []
for i in range(60):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    flat.zero_grad(); b._gnx_pack = None
    model.training_step(b, 0).backward()
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print("per-step ms:", " ".join(f"{t:.2f}" for t in ts))
# unsynchronised blocks of 20 like the bench
for rep in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        flat.zero_grad(); b._gnx_pack = None
        model.training_step(b, 0).backward()
    torch.cuda.synchronize(); print(f"block {rep}: {(time.perf_counter()-t0)/20*1e3:.3f} ms/step")
