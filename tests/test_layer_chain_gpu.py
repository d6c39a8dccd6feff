"""Teacher-forced layer-chain parity: every layer of a DEEP model on its REAL input.

tests/test_conv_gpu.py holds one conv layer to 1e-5 on synthetic ``randn().relu()`` inputs; tests/test_model_gpu.py holds
whole models to the reference's own fp32 reproducibility envelope (up to 0.5 on an intermediate's gradient).  Between the
two: the fp64 oracle (``oracle/pyg_restatement.py``, restating the wiring of
/root/reference/gnnepcsaft/train/models.py:196-227) is run once through the whole model and the loss; then, for EVERY
layer l, the native ``conv_l -> BatchNorm_l -> ReLU`` is fed the oracle's own input of layer l (cast to fp32 -- the
activations a deep model really produces: post-BatchNorm-ReLU rows, max|conv output| / sigma ~ 20-50) and the oracle's own
upstream gradient, and its output and ALL gradients (input, bond table, every parameter of the conv and of the BatchNorm)
are compared with the fp64 oracle ON THE SAME fp32-cast input.  The CPU fp32 oracle runs on the same tensors and is
printed beside the HIP path (the reference's own distance to fp64).

No error propagates from layer to layer.  Node rows holding a discrete decision inside its rounding band (std mask,
near-tied extremum, hidden ReLU at 0: tests/conv_parity.py) are dropped from the forward comparison and their upstream
gradient is zeroed; BatchNorm outputs closer to the following ReLU's kink than the forward tolerance, propagated through
the BatchNorm scale |gamma| / sigma, get a zero upstream gradient element-wise.  Both exclusions are COUNTED and asserted
small (<= 2 % of the rows, <= 0.2 % of the entries).  Bars (and what was measured): see the comment above ``bars`` below --
forward 1e-5 (max norm) on every layer; gradients 5e-5 in the L2 norm and 1e-3 in the max norm, because on real
activations single events outside any rigorous band decide the max norm for the CPU fp32 oracle just as for the HIP path.
"""
import copy

import pytest
import torch

from oracle import pyg_restatement as O
from tests import conv_parity
from tests.conv_parity import bond_codes, gine_event_rows, pna_event_rows
from tests.model_cases import build_case
from tests.parity_util import rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-5
GRAD_L2 = 5e-5


def _oracle_chain(o64, batch, target):
    """One fp64 forward + loss + backward of the oracle, keeping every layer's input and the gradient at its output."""
    x = o64.node_embed(batch.x)
    ea = o64.edge_embed(batch.edge_attr)
    acts = [x]
    for conv, bn in zip(o64.convs, o64.batch_norms):
        a = torch.relu(bn(conv(x=acts[-1], edge_index=batch.edge_index, edge_attr=ea)))
        a.retain_grad()
        acts.append(a)
    pred = o64.mlp(o64.global_pool(acts[-1], batch.batch))
    loss = O.ape_huber_loss(pred, getattr(batch, target).double())
    loss.backward()
    return [a.detach() for a in acts[:-1]], [a.grad.detach() for a in acts[1:]]


@pytest.mark.parametrize("name", ["pna_cfg2_full_1024", "pna_cfg5_shaped", "gine_cfg3_full_1024"])
def test_every_layer_on_the_deep_models_real_activations(gpu_device, name):
    from gnnepcsaft_amd import ops
    from gnnepcsaft_amd.train.models import GNNePCSAFT
    cfg, batch, target = build_case(name)
    torch.manual_seed(0)
    o32 = O.GNNePCSAFT(cfg).train()
    state = copy.deepcopy(o32.state_dict())
    o64 = copy.deepcopy(o32).double().train()
    inputs, upstream = _oracle_chain(o64, batch, target)
    o64.load_state_dict({k: v.double() if v.dtype.is_floating_point else v for k, v in state.items()})  # running stats back
    o64.zero_grad()

    native = GNNePCSAFT(cfg)
    native.load_state_dict(state, strict=True)
    native = native.to(gpu_device).train()
    b = batch.to(gpu_device)
    N = batch.x.size(0)
    pack = ops.pack_graph(b.edge_index, b.edge_attr, b.batch, N, int(batch.num_graphs))
    code = bond_codes(batch.edge_attr)
    pna = cfg["conv"] == "PNA"
    worst, failures = {}, []
    for l in range(cfg["propagation_depth"]):
        x32 = inputs[l].float()                      # teacher forcing: the oracle's input of layer l, as fp32 holds it
        g32 = upstream[l].float()
        conv64, bn64 = copy.deepcopy(o64.convs[l]), copy.deepcopy(o64.batch_norms[l])
        # ---- fp64 arbiter on the fp32-cast input
        with torch.no_grad():
            table64 = o64.edge_embed(native.edge_embed.combos.cpu())          # the 60 bond-feature combinations
            # hidden-ReLU band of GINE: 12 instead of 4 roundings of the Linear's error scale -- the INPUT of its hidden Linear
            # (the sum aggregate) carries its own fp32 error, on top of the product's
            saved_band = conv_parity.LIN_BAND_ULPS
            conv_parity.LIN_BAND_ULPS = saved_band if pna else 12.0
            try:
                ev = (pna_event_rows if pna else gine_event_rows)(conv64, x32.double(), batch.edge_index,
                                                                  table64.index_select(0, code))
            finally:
                conv_parity.LIN_BAND_ULPS = saved_band
        rows = ev["rows"]
        xx = x32.double().requires_grad_(True)
        be = table64.detach().clone().requires_grad_(True)
        y = conv64(x=xx, edge_index=batch.edge_index, edge_attr=be.index_select(0, code))
        z = bn64(y)
        bnm = bn64.module
        # A BatchNorm output closer to the following ReLU's kink than what the forward tolerance maps to cannot be decided at
        # that tolerance: the conv output is held to TOL x max|y| (norm-wise), BatchNorm multiplies an error of the conv
        # output by |gamma| / sigma -- and max|y| / sigma is ~ 20-50 on these activations, which is why "one layer on randn"
        # said nothing about this.  Those entries get a zero upstream gradient on both sides (element-wise, counted).
        with torch.no_grad():
            yd = y.detach()
            sigma = (yd.var(dim=0, unbiased=False) + bnm.eps).sqrt()
            band = TOL * float(yd.abs().max()) * bnm.weight.detach().abs() / sigma
            near_kink = z.detach().abs() <= band
        g = g32.double() * (~rows).unsqueeze(1) * (~near_kink)
        torch.relu(z).backward(g)
        ref_out = torch.relu(z).detach()
        ref_grads = {"conv." + n: p.grad for n, p in conv64.named_parameters()}
        ref_grads.update({"bn." + n: p.grad for n, p in bn64.named_parameters()})
        # ---- native conv + BatchNorm(+ReLU) on the same fp32 tensors
        nconv, nbn = native.convs[l], native.batch_norms[l]
        native.zero_grad(set_to_none=True)
        xn = x32.to(gpu_device).requires_grad_(True)
        ben = table64.float().to(gpu_device).requires_grad_(True)
        out = nbn(nconv(x=xn, edge_index=pack, edge_attr=ben), relu=True)
        out.backward(g.float().to(gpu_device))
        torch.cuda.synchronize()
        got = {"conv." + n: p.grad for n, p in nconv.named_parameters()}
        got.update({"bn." + n: p.grad for n, p in nbn.named_parameters()})
        # ---- the CPU fp32 oracle on the same tensors: its own distance to fp64 is the yardstick where 1e-5 is not attainable
        c32, b32 = copy.deepcopy(o32.convs[l]), copy.deepcopy(o32.batch_norms[l])
        c32.zero_grad()
        b32.zero_grad()
        x3 = x32.clone().requires_grad_(True)
        be3 = table64.float().requires_grad_(True)
        o3 = torch.relu(b32(c32(x=x3, edge_index=batch.edge_index, edge_attr=be3.index_select(0, code))))
        o3.backward(g.float())
        cpu_grads = {"conv." + n: p.grad for n, p in c32.named_parameters()}
        cpu_grads.update({"bn." + n: p.grad for n, p in b32.named_parameters()})
        # ---- bookkeeping + comparison
        keep = ~rows
        n_excl, n_kink = int(rows.sum()), int(near_kink.sum())
        assert n_excl <= max(2, N // 50), (name, l, "event rows", n_excl, N)           # <= 2 % of the rows
        assert n_kink <= max(8, z.numel() // 500), (name, l, "BatchNorm outputs at the ReLU kink", n_kink)  # <= 0.2 %
        def l2(a_, ref_):
            a_, ref_ = a_.detach().double().cpu(), ref_.detach().double().cpu()
            d_ = float(ref_.norm())
            return float((a_ - ref_).norm()) / d_ if d_ > 0 else 0.0

        hip = {"out": rel_err(out.detach().cpu()[keep], ref_out[keep]), "dx": rel_err(xn.grad.cpu(), xx.grad),
               "dbe": rel_err(ben.grad.cpu(), be.grad), "dx_l2": l2(xn.grad, xx.grad), "dbe_l2": l2(ben.grad, be.grad)}
        cpu = {"out": rel_err(o3.detach()[keep], ref_out[keep]), "dx": rel_err(x3.grad, xx.grad), "dbe": rel_err(be3.grad, be.grad),
               "dx_l2": l2(x3.grad, xx.grad), "dbe_l2": l2(be3.grad, be.grad)}
        G = max(float(v.abs().max()) for v in ref_grads.values())
        # a bias that only shifts the BatchNorm's input by a constant has an analytically ZERO gradient (BatchNorm removes
        # the mean): what any fp32 path computes for it is the rounding noise of a cancelling sum over all rows
        post_last = 2 * (cfg["post_layers"] - 1)
        zero_grad = {"conv.lin.bias", "conv.nn.2.bias"} | {f"conv.post_nns.{t}.{post_last}.bias" for t in range(cfg["towers"])}
        noise = {}
        num_h = num_c = den = 0.0
        for n_, ref in ref_grads.items():
            if n_ in zero_grad:
                noise[n_] = (float((got[n_].cpu() - ref).abs().max()) / G, float((cpu_grads[n_] - ref).abs().max()) / G)
                continue
            hip["d" + n_] = rel_err(got[n_].cpu(), ref, floor=1e-3 * G)
            cpu["d" + n_] = rel_err(cpu_grads[n_], ref, floor=1e-3 * G)
            num_h += float(((got[n_].cpu().double() - ref) ** 2).sum())
            num_c += float(((cpu_grads[n_].double() - ref) ** 2).sum())
            den += float((ref ** 2).sum())
        hip["dparam_l2"], cpu["dparam_l2"] = (num_h / den) ** 0.5, (num_c / den) ** 0.5
        pk = [k for k in hip if k.startswith("dconv") or k.startswith("dbn")]
        wp = max(pk, key=lambda k: hip[k])
        print(f"{name} layer {l}: hip out {hip['out']:.1e} | dx l2 {hip['dx_l2']:.1e} max {hip['dx']:.1e} | dbe l2 {hip['dbe_l2']:.1e} "
              f"max {hip['dbe']:.1e} | dparam l2 {hip['dparam_l2']:.1e} worst {hip[wp]:.1e} ({wp[1:]}) || cpu fp32 out {cpu['out']:.1e} | "
              f"dx l2 {cpu['dx_l2']:.1e} max {cpu['dx']:.1e} | dbe l2 {cpu['dbe_l2']:.1e} max {cpu['dbe']:.1e} | dparam l2 "
              f"{cpu['dparam_l2']:.1e} worst {max(cpu[k] for k in pk):.1e} || zero-gradient biases / max grad: hip "
              f"{max(v[0] for v in noise.values()):.1e} cpu {max(v[1] for v in noise.values()):.1e} | excluded rows {n_excl}/{N}, "
              f"kink entries {n_kink}", flush=True)
        # Bars.  Forward: 1e-5 in the max norm (measured: <= 9.0e-6 on every layer of the three models, the CPU fp32 oracle
        # 3.4e-6 .. 9.6e-6 on the PNA ones).  Gradients: GRAD_L2 = 5e-5 in the L2 norm (input, bond table, all parameters
        # jointly; measured 4e-7 .. 3.9e-5, at or below 1.1e-5 on 14 of the 14 cfg-2 / cfg-3 layers); in the max norm a
        # gradient is decided by single discrete events that the exclusion bands do not catch on real activations -- the CPU
        # fp32 oracle shows the same (max-norm dx 8.1e-5 / 2.9e-5 / 2.6e-3 and L2 1.9e-4 on layers 1 / 3 / 5 of the cfg-2
        # model, where the HIP path has 1.5e-4 / 9.7e-5 / 1.7e-5 and L2 1.5e-6) -- so there the bar is 1e-3 = "a handful of
        # entries moved by one event"; an analytically zero bias gradient: noise below 1e-4 of the largest gradient entry.
        bars = {"out": TOL, "dx_l2": GRAD_L2, "dbe_l2": GRAD_L2, "dparam_l2": GRAD_L2, "dx": 1e-3, "dbe": 1e-3}
        for k, v in hip.items():
            bound = bars.get(k, 1e-3)
            if v > bound:
                failures.append((l, k, v, bound))
        for n_, (vh, _vc) in noise.items():
            if vh > 1e-4:
                failures.append((l, "noise " + n_, vh, 1e-4))
        worst[l] = max(hip[k] for k in ("out", "dx_l2", "dbe_l2", "dparam_l2"))
    print(name, "worst error per layer:", {l: f"{v:.1e}" for l, v in worst.items()})
    assert not failures, (name, failures)
