"""One molecule through InferenceEngine (eager launches, so that the kernel trace names every launch): used under
rocprofv3 --kernel-trace --stats to see what a single-molecule call is made of.  usage: python tools/infer_single_trace.py [calls]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnnepcsaft_amd.data import default_config, synthetic_batch
from gnnepcsaft_amd.data.batching import calc_deg
from gnnepcsaft_amd.inference import InferenceEngine
from gnnepcsaft_amd.train.models import create_model

dev = torch.device("cuda:0")
calls = int(sys.argv[1]) if len(sys.argv) > 1 else 40
cfg = default_config(2)
b = synthetic_batch(1, 2).to(dev)
deg = calc_deg([synthetic_batch(4096, 2)])
torch.manual_seed(0)
model = create_model(cfg, deg).to(dev).eval()
model.model.validate_inputs = False
model.model.max_degree_hint = len(deg) - 1
eng = InferenceEngine(model)
for _ in range(calls):
    eng(b.x, b.edge_index, b.edge_attr, None, validate=False)
torch.cuda.synchronize()
print("atoms", b.x.size(0), "calls", calls)
