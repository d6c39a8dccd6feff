"""cProfile of the host side of eager steps (launch path), cfg-2 sizes: where the CPU time per launch goes."""
import cProfile, os, pstats, sys, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gnnepcsaft_amd import functional as Fn, ops
from gnnepcsaft_amd.data import default_config, synthetic_batch
from gnnepcsaft_amd.data.batching import calc_deg
from gnnepcsaft_amd.train.models import create_model
from gnnepcsaft_amd import dp

dev = torch.device("cuda:0")
cfg = default_config(2)
batch = synthetic_batch(int(os.environ.get("GRAPHS", "4096")), 2)
deg = calc_deg([batch])
torch.manual_seed(0)
model = create_model(cfg, deg).to(dev)
model.train()
model.model.validate_inputs = False
flat = dp.FlatGradAllReduce(model)
Fn.set_grad_in_place(True)
ops.set_wgrad_side_stream(True)
b = batch.to(dev)
s = torch.cuda.Stream()
torch.cuda.set_stream(s)
def step():
    flat.zero_grad()
    b._gnx_pack = None
    loss = model.training_step(b, 0)
    loss.backward()
for _ in range(5): step()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(20): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host issue time/step {(t1-t0)/20*1e3:.2f} ms, incl. drain {(t2-t0)/20*1e3:.2f} ms")
pr = cProfile.Profile()
with torch.autograd.set_multithreading_enabled(False):
    pr.enable()
    for _ in range(20): step()
    pr.disable()
torch.cuda.synchronize()
st = io.StringIO()
pstats.Stats(pr, stream=st).sort_stats("tottime").print_stats(45)
print(st.getvalue()[:9000])
