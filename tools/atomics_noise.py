"""Run-to-run spread of the gradients of one training step (fp32 atomics in the weight / table gradients arrive in a
different order every run): the metric of tests/test_fused_gpu.py::test_model_with_and_without_the_fused_edge_kernel
evaluated between REPEATS of the same configuration.  usage: python tools/atomics_noise.py [case] [repeats]"""
import copy, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnnepcsaft_amd import functional as Fn
from gnnepcsaft_amd.train.models import GNNePCSAFT
from tests.model_cases import build_case
from tests.parity_util import rel_err

name = sys.argv[1] if len(sys.argv) > 1 else "pna_cfg2_full_1024"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
cfg, batch, target = build_case(name)
torch.manual_seed(0)
model = GNNePCSAFT(cfg).to("cuda:0").train()
state = copy.deepcopy(model.state_dict())
b = batch.to("cuda:0")
runs = []
for i in range(reps):
    model.load_state_dict(state)
    model.zero_grad(set_to_none=True)
    model.max_degree_hint = len(cfg["deg"]) - 1
    if hasattr(b, "_gnx_pack"):
        del b._gnx_pack
    pred = model(b.x, b.edge_index, b.edge_attr, b.batch)
    loss, _ = Fn.HuberAPEFn.apply(pred, getattr(b, target), 0.01)
    loss.backward()
    torch.cuda.synchronize()
    runs.append({n: p.grad.detach().clone() for n, p in model.named_parameters()})
g0 = runs[0]
G = max(float(v.abs().max()) for v in g0.values())
worst = {}
for r in runs[1:]:
    for n in g0:
        e = float(rel_err(r[n], g0[n], floor=1e-2 * G))
        worst[n] = max(worst.get(n, 0.0), e)
top = sorted(worst.items(), key=lambda kv: -kv[1])[:6]
print(name, "repeats", reps, "largest run-to-run rel_err (floor 1e-2 G):", [(n, f"{e:.2e}") for n, e in top])
