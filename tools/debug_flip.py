"""Diagnostic: where does the HIP backward leave the fp64 oracle in a given strict case, and is it a near-boundary decision?"""
import copy, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnnepcsaft_amd import functional as Fn
from gnnepcsaft_amd.data import calc_deg
from oracle import pyg_restatement as O
from tests.parity_util import capture_intermediates, make_models
from tests.model_cases import build_case

name, layer = sys.argv[1], int(sys.argv[2])
cfg, batch, _ = build_case(name)
oracle, native = make_models(cfg, 0)
oracle.train(); native.train()
o64 = copy.deepcopy(oracle).double()
caps = {}
for tag, m, dt, dev in (("c64", o64, torch.float64, "cpu"), ("c32", oracle, torch.float32, "cpu"), ("hip", native.to("cuda:0"), torch.float32, "cuda:0")):
    b = batch.to(dev)
    cap = capture_intermediates(m)
    with cap as inter:
        pred = m(b.x, b.edge_index, b.edge_attr, b.batch)
    if tag == "hip":
        loss, _ = Fn.HuberAPEFn.apply(pred, b.para, 0.01)
    else:
        loss = O.ape_huber_loss(pred, b.para.to(dt))
    loss.backward()
    caps[tag] = (inter, cap.grads(), cap)
k = f"d_conv{layer}"
g64, g32, gh = caps["c64"][1][k], caps["c32"][1][k], caps["hip"][1][k]
scale = g64.abs().max()
eh, ec = (gh - g64).abs() / scale, (g32 - g64).abs() / scale
print(k, "max err hip", float(eh.max()), "cpu", float(ec.max()), "entries hip>1e-4:", int((eh > 1e-4).sum()), "of", eh.numel())
idx = torch.nonzero(eh > 1e-4)
cols = sorted(set(int(c) for c in idx[:, 1]))
print("columns affected:", cols[:20], "rows affected:", len(set(int(r) for r in idx[:, 0])))
conv64 = caps["c64"][0][f"conv{layer}"]; convh = caps["hip"][0][f"conv{layer}"]; conv32 = caps["c32"][0][f"conv{layer}"]
act64 = caps["c64"][0][f"act{layer}"]; acth = caps["hip"][0][f"act{layer}"]; act32 = caps["c32"][0][f"act{layer}"]
for c in cols[:4]:
    x = conv64[:, c]
    mean, std = x.mean(), x.std(unbiased=False)
    print(f" col {c}: conv mean {float(mean):.4g} std {float(std):.4g} |mean|/std {float(mean.abs()/std):.3g}")
    # decisions: relu mask of act
    m64, mh, m32 = act64[:, c] > 0, acth[:, c] > 0, act32[:, c] > 0
    print("   relu mask differs hip/64:", int((m64 != mh).sum()), " cpu32/64:", int((m64 != m32).sum()))
    d = torch.nonzero(m64 != mh).flatten()
    for r in d[:5]:
        bn = o64.batch_norms[layer].module
        yy = (x[r] - mean) / (std * std + 1e-5).sqrt() * bn.weight[c] + bn.bias[c]
        print(f"    row {int(r)}: y64 {float(yy):.3e} act64 {float(act64[r,c]):.3e} acth {float(acth[r,c]):.3e} act32 {float(act32[r,c]):.3e} conv64 {float(conv64[r,c]):.9g} convh {float(convh[r,c]):.9g} conv32 {float(conv32[r,c]):.9g}")
    print("   conv col err hip", float((convh[:, c]-x).abs().max()), "cpu", float((conv32[:, c]-x).abs().max()))
