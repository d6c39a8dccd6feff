"""Soak: many training steps (fused optimizer, both streams); checks that device memory stays flat and the loss falls."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnnepcsaft_amd import functional as Fn, ops, dp
from gnnepcsaft_amd.data import default_config, synthetic_batch
from gnnepcsaft_amd.data.batching import calc_deg
from gnnepcsaft_amd.optim import configure_fused_optimizers
from gnnepcsaft_amd.train.models import create_model

dev = torch.device("cuda:0")
gen = int(os.environ.get("CONFIG", "2"))           # 2: 20-atom molecules, 5: skewed 5..80 atoms
cfg = default_config(gen)
steps = int(os.environ.get("STEPS", "300"))
# four batches of different sizes: every step changes the row counts (ragged tiles, different kernel dispatches)
sizes = [int(v) for v in os.environ.get("BATCHES", "2048,4096,1536,3000").split(",")]
batches = [synthetic_batch(n, gen, seed=100 + i).to(dev) for i, n in enumerate(sizes)]
deg = calc_deg([synthetic_batch(2048, gen)])
torch.manual_seed(0)
model = create_model(cfg, deg).to(dev)
model.train()
model.model.validate_inputs = False
model.model.max_degree_hint = len(deg) - 1
flat = dp.FlatGradAllReduce(model)
opt = configure_fused_optimizers(model, flat)["optimizer"]
Fn.set_grad_in_place(True)
ops.set_wgrad_side_stream(True)
losses, mem = [], []
for i in range(steps):
    b = batches[i % 4]
    b._gnx_pack = None
    opt.zero_grad()
    loss = model.training_step(b, i)
    loss.backward()
    opt.step()
    if i % 50 == 0 or i == steps - 1:
        torch.cuda.synchronize()
        ops.check_range(dev)
        losses.append(float(loss)); mem.append(torch.cuda.memory_allocated() / 2**20)
        free, total = torch.cuda.mem_get_info()
        print(f"step {i}: loss {losses[-1]:.5f}  torch allocated {mem[-1]:.1f} MiB  device used {(total-free)/2**20:.0f} MiB", flush=True)
assert losses[-1] < losses[0], (losses[0], losses[-1])
assert abs(mem[-1] - mem[2]) < 64.0, mem  # batches of four sizes: the caching allocator settles after one round
print("soak ok")
