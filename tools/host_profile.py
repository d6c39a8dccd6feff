"""cProfile of the Python side of one training step (host-bound regime: 512 graphs).  Prints the top functions by
cumulative and by own time.  usage: python tools/host_profile.py [graphs]"""
import cProfile, os, pstats, sys, time, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gnnepcsaft_amd import dp, functional as Fn, ops
from gnnepcsaft_amd.data import calc_deg, default_config, synthetic_batch
from gnnepcsaft_amd.train.models import create_model
graphs = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device("cuda:0")
st = torch.cuda.Stream(device=dev); torch.cuda.set_stream(st)
cfg = default_config(2)
b_cpu = synthetic_batch(graphs, 2); deg = calc_deg(b_cpu)
torch.manual_seed(0)
model = create_model(cfg, deg).to(dev).train()
model.model.validate_inputs = False; model.model.max_degree_hint = len(deg) - 1
flat = dp.FlatGradAllReduce(model); Fn.set_grad_in_place(True); ops.set_wgrad_side_stream(True)
b = b_cpu.to(dev)
def step():
    flat.zero_grad(); b._gnx_pack = None
    model.training_step(b, 0).backward()
for _ in range(10): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50): step()
torch.cuda.synchronize(); print("ms/step", (time.perf_counter() - t0) / 50 * 1e3)
# forward-only host time and backward host time
t0 = time.perf_counter()
for _ in range(50):
    flat.zero_grad(); b._gnx_pack = None
    loss = model.training_step(b, 0)
t1 = time.perf_counter(); torch.cuda.synchronize()
print("host ms for zero_grad+pack+forward (async):", (t1 - t0) / 50 * 1e3)
pr = cProfile.Profile(); pr.enable()
for _ in range(30): step()
pr.disable(); torch.cuda.synchronize()
for key in ("cumulative", "tottime"):
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats(key).print_stats(28); print(s.getvalue()[:6000])
