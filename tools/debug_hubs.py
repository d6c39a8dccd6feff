import copy, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnnepcsaft_amd import functional as Fn
from oracle import pyg_restatement as O
from tests.parity_util import capture_intermediates, make_models
from tests.model_cases import build_case
from tests.conv_parity import pna_event_rows
cfg, batch, _ = build_case("pna_hubs")
oracle, native = make_models(cfg, 0)
oracle.train(); native.train()
o64 = copy.deepcopy(oracle).double()
res = {}
for tag, m, dev in (("c64", o64, "cpu"), ("c32", oracle, "cpu"), ("hip", native.to("cuda:0"), "cuda:0")):
    b = batch.to(dev)
    with capture_intermediates(m) as inter:
        m(b.x, b.edge_index, b.edge_attr, b.batch)
    res[tag] = dict(inter)
deg = torch.bincount(batch.edge_index[1], minlength=batch.x.size(0))
c64, ch, c32 = res["c64"]["conv0"], res["hip"]["conv0"], res["c32"]["conv0"]
scale = c64.abs().max()
eh = (ch - c64).abs().max(dim=1).values / scale
ec = (c32 - c64).abs().max(dim=1).values / scale
top = torch.argsort(eh, descending=True)[:8]
print("embed err hip", float((res['hip']['embed']-res['c64']['embed']).abs().max()))
for r in top:
    print(f"row {int(r)} deg {int(deg[r])} err hip {float(eh[r]):.2e} cpu {float(ec[r]):.2e}")
print("rows with err>1e-5:", int((eh > 1e-5).sum()), "their degrees:", sorted(set(int(d) for d in deg[eh > 1e-5])))
