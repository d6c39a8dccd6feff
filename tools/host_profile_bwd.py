"""Host time of every autograd Function's forward / backward (Python side, launches are asynchronous) per step."""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gnnepcsaft_amd import dp, functional as Fn, ops
from gnnepcsaft_amd.data import calc_deg, default_config, synthetic_batch
from gnnepcsaft_amd.train.models import create_model
acc = collections.defaultdict(float); cnt = collections.Counter()
def wrap(cls, name):
    f = getattr(cls, name)
    def timed(*a, **k):
        t0 = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            acc[f"{cls.__name__}.{name}"] += time.perf_counter() - t0; cnt[f"{cls.__name__}.{name}"] += 1
    setattr(cls, name, staticmethod(timed))
for c in (Fn.EmbedSumFn, Fn.LinearFn, Fn.BatchNormFn, Fn.SegmentPoolFn, Fn.HuberAPEFn, Fn.PNAConvFn):
    wrap(c, "forward"); wrap(c, "backward")
graphs = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device("cuda:0"); st = torch.cuda.Stream(device=dev); torch.cuda.set_stream(st)
cfg = default_config(2); b_cpu = synthetic_batch(graphs, 2); deg = calc_deg(b_cpu)
torch.manual_seed(0)
model = create_model(cfg, deg).to(dev).train()
model.model.validate_inputs = False; model.model.max_degree_hint = len(deg) - 1
flat = dp.FlatGradAllReduce(model); Fn.set_grad_in_place(True); ops.set_wgrad_side_stream(True)
b = b_cpu.to(dev)
def step():
    flat.zero_grad(); b._gnx_pack = None
    model.training_step(b, 0).backward()
for _ in range(10): step()
torch.cuda.synchronize(); acc.clear(); cnt.clear()
n = 50; t0 = time.perf_counter()
for _ in range(n): step()
torch.cuda.synchronize(); print("ms/step", (time.perf_counter() - t0) / n * 1e3)
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print(f"{k:28s} {v / n * 1e3:7.3f} ms/step  ({cnt[k] / n:.0f} calls, {v / cnt[k] * 1e6:6.1f} us each)")
print("sum", sum(acc.values()) / n * 1e3)
