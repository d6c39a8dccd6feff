"""Forward-only throughput: InferenceEngine (BatchNorm folded, weight-only tables precomputed) vs model.eval()."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnnepcsaft_amd import ops
from gnnepcsaft_amd.data import default_config, synthetic_batch
from gnnepcsaft_amd.data.batching import calc_deg
from gnnepcsaft_amd.inference import InferenceEngine
from gnnepcsaft_amd.train.models import create_model

dev = torch.device("cuda:0")
for cfg_i, graphs in ((2, 4096), (2, 1), (3, 16384)):
    cfg = default_config(cfg_i)
    b = synthetic_batch(graphs, cfg_i).to(dev)
    deg = calc_deg([synthetic_batch(min(graphs, 4096), cfg_i)])
    torch.manual_seed(0)
    model = create_model(cfg, deg).to(dev).eval()
    model.model.validate_inputs = False
    model.model.max_degree_hint = len(deg) - 1
    eng = InferenceEngine(model)
    batch_arg = b.batch if graphs > 1 else None

    def run_model():
        b._gnx_pack = None
        with torch.no_grad():
            return model.model(b.x, b.edge_index, b.edge_attr, batch_arg)

    def run_engine():
        return eng(b.x, b.edge_index, b.edge_attr, batch_arg, validate=False)

    runs = [("model.eval()", run_model), ("InferenceEngine", run_engine)]
    if graphs == 1:
        runs.append(("engine.single (HIP graph)", lambda: eng.single(b.x, b.edge_index, b.edge_attr)))
    for name, fn in runs:
        for _ in range(5): fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 30
        for _ in range(n): fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print(f"cfg-{cfg_i} {graphs:6d} graphs  {name:26s} {dt*1e3:8.3f} ms/call  {graphs/dt:12.0f} graphs/s", flush=True)
    ops.check_range(dev)
