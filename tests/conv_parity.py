"""Single-layer parity harness: ONE PNAConv / GINEConv forward + every gradient on identical inputs and weights, HIP
path vs the oracle's module (``oracle/pyg_restatement.py`` PNAConv / GINEConv, which restate the layers the reference
builds at ``/root/reference/gnnepcsaft/train/models.py:445-457, 529-538``), with the oracle evaluated in fp64 as the
arbiter.  The oracle is the checker here, never the thing measured.

Why rows are excluded (and counted).  The reference algorithm has discrete events whose outcome ANY fp32 evaluation
(the reference's own CPU path included) decides by its rounding errors:
  * [3P] StdAggregation hard-masks ``std`` to 0 where ``var <= 1e-5`` and evaluates ``var = mean(x^2) - mean(x)^2``
    with an absolute error ~ eps32 * mean(x^2): an entry whose exact variance lies within that error of the
    threshold jumps by sqrt(1e-5) = 3.2e-3 either way;
  * min / max aggregation sends the whole gradient to the winning message: a runner-up within rounding error of the
    extremum (but not exactly tied: exact ties split evenly on both paths) may win instead;
  * a ReLU pre-activation within rounding error of 0 flips its gradient mask.
Such entries are identified from the fp64 evaluation alone (``VAR_BAND_ULPS`` / ``LIN_BAND_ULPS`` fp32 roundings of the
quantity's error scale: mean(x^2) for the variance, sum |a||w| + |b| for a Linear output).  Node rows that hold one (for an edge-level
event: the edge's target row) are dropped from the forward comparison, and their upstream gradient is set to zero so
that no gradient (input, bond table or parameter) depends on which way the event went.  Everything else is held to
the north-star tolerance.  The counts are returned and asserted small by the tests.
"""
from __future__ import annotations

import copy
import math
from typing import Dict, Optional

import torch

from oracle import pyg_restatement as O
from tests.parity_util import rel_err

# half-widths of the excluded bands, in fp32 unit roundoffs (2^-24) of the quantity's error scale:
VAR_BAND_ULPS = 16.0      # var = mean(x^2) - mean(x)^2 over n <= 4 messages: worst case (3n + 2) roundings of mean(x^2)
LIN_BAND_ULPS = 4.0       # a K-term Linear output against sum |a||w| + |b| (typical error ~ sum / sqrt(K): >= 40 sigma)
STD_THRESHOLD = 1e-5


def bond_codes(edge_attr: torch.Tensor) -> torch.Tensor:
    """Mixed-radix code of the 3 bond features, row index into the 60-row table (cartesian_prod order)."""
    from gnnepcsaft_amd.data import BOND_FEATURE_DIMS
    code = torch.zeros(edge_attr.size(0), dtype=torch.long)
    for k, d in enumerate(BOND_FEATURE_DIMS):
        code = code * d + edge_attr[:, k]
    return code


def _l2(a: torch.Tensor, ref: torch.Tensor) -> float:
    a, ref = a.detach().double().cpu(), ref.detach().double().cpu()
    den = float(ref.norm())
    num = float((a - ref).norm())
    return 0.0 if num == 0.0 else (num / den if den > 0 else float("inf"))


def _lin_with_scale(lin, a):
    """(a W^T + b, |a| |W|^T + |b|): a Linear's output and the magnitude its fp32 rounding errors scale with."""
    z = torch.nn.functional.linear(a, lin.weight, lin.bias)
    s = torch.nn.functional.linear(a.abs(), lin.weight.abs(), None if lin.bias is None else lin.bias.abs())
    return z, s


def _near_zero(z, s):
    return z.abs() <= LIN_BAND_ULPS * 2.0 ** -24 * s


def _mlp_events(seq, a):
    """Run a Sequential(Linear, (ReLU, Linear)*) in fp64; returns (output, its error scale, per-row flag of a ReLU
    pre-activation inside the band around 0)."""
    lins = [m for m in seq if isinstance(m, torch.nn.Linear)]
    flag = torch.zeros(a.size(0), dtype=torch.bool)
    z = s = None
    for k, lin in enumerate(lins):
        z, s = _lin_with_scale(lin, a)
        if k < len(lins) - 1:
            flag |= _near_zero(z, s).any(dim=1)
            a = z.relu()
    return z, s, flag


def _runner_up_gap(m, s, index, N, reduce):
    """Per (node, channel): gap between the extremum and the nearest DIFFERENT message value, and the error scale."""
    ext = O.scatter(m, index, 0, N, reduce)
    is_ext = m == ext.index_select(0, index)
    big = torch.finfo(m.dtype).max
    rest = torch.where(is_ext, torch.full_like(m, -big if reduce == "max" else big), m)
    filled = torch.full_like(ext, -big if reduce == "max" else big)
    second = filled.scatter_reduce(0, index.view(-1, *[1] * (m.dim() - 1)).expand_as(m), rest,
                                   reduce="amax" if reduce == "max" else "amin", include_self=True)
    gap = (ext - second).abs()
    gap = torch.where(second.abs() >= big, torch.full_like(gap, float("inf")), gap)
    smax = torch.zeros_like(ext).scatter_reduce(0, index.view(-1, *[1] * (m.dim() - 1)).expand_as(m), s, reduce="amax",
                                                include_self=True)
    return gap, smax


def pna_event_rows(conv64: O.PNAConv, x64, edge_index, ea64) -> Dict[str, torch.Tensor]:
    """fp64 evaluation of one PNAConv; returns the boolean mask of node rows that hold a DISCRETE EVENT of the
    reference algorithm inside the fp32 rounding band (so that any fp32 evaluation, the reference's CPU path
    included, may land on either side of it):
      * std: exact variance within the band of StdAggregation's hard 1e-5 mask;
      * min / max: the runner-up message within the band of the extremum without being exactly tied (the gradient
        goes to whichever wins);
      * ReLU: a hidden pre-activation of the row's post-MLP, or of an incoming edge's pre-MLP, within the band of 0
        (the gradient mask flips).
    """
    N = x64.size(0)
    T, F = conv64.towers, conv64.F_in
    xt = x64.view(-1, T, F)
    j, i = edge_index[0], edge_index[1]
    e = conv64.edge_encoder(ea64).view(-1, 1, F).repeat(1, T, 1)
    h = torch.cat([xt.index_select(0, i), xt.index_select(0, j), e], dim=-1)
    ms, ss = [], []
    relu_edge = torch.zeros(h.size(0), dtype=torch.bool)
    for t, nn in enumerate(conv64.pre_nns):
        z, s, fl = _mlp_events(nn, h[:, t])
        ms.append(z)
        ss.append(s)
        relu_edge |= fl
    m, s = torch.stack(ms, dim=1), torch.stack(ss, dim=1)
    mean = O.scatter(m, i, 0, N, "mean")
    msq = O.scatter(m * m, i, 0, N, "mean")
    var = msq - mean * mean
    near_std = (var - STD_THRESHOLD).abs() <= VAR_BAND_ULPS * 2.0 ** -24 * msq
    near_ext = torch.zeros_like(near_std)
    if m.size(0) > 0:
        for reduce in ("min", "max"):
            gap, smax = _runner_up_gap(m, s, i, N, reduce)
            near_ext |= gap <= 2 * LIN_BAND_ULPS * 2.0 ** -24 * smax
    rows = near_std.flatten(1).any(dim=1) | near_ext.flatten(1).any(dim=1)
    relu_rows = torch.zeros(N, dtype=torch.bool)
    relu_rows[i[relu_edge]] = True
    A = conv64.aggr_module(m, i, dim_size=N, dim=0)
    out = torch.cat([xt, A], dim=-1)
    for t, nn in enumerate(conv64.post_nns):
        relu_rows |= _mlp_events(nn, out[:, t])[2]
    return {"rows": rows | relu_rows, "std": int(near_std.sum()), "ext": int(near_ext.sum()),
            "relu": int(relu_rows.sum()), "var": var}


def gine_event_rows(conv64: O.GINEConv, x64, edge_index, ea64) -> Dict[str, torch.Tensor]:
    """Same for GINEConv: the message ReLU (edge -> its target row) and the hidden ReLU of ``nn``."""
    N = x64.size(0)
    j, i = edge_index[0], edge_index[1]
    e, se = _lin_with_scale(conv64.lin, ea64)
    xj = x64.index_select(0, j)
    pre = xj + e
    edge_flag = _near_zero(pre, xj.abs() + se).any(dim=1)
    rows = torch.zeros(N, dtype=torch.bool)
    rows[i[edge_flag]] = True
    agg = O.scatter(pre.relu(), i, 0, N, "sum") + (1 + conv64.eps) * x64
    rows |= _mlp_events(conv64.nn, agg)[2]
    return {"rows": rows, "std": 0, "ext": 0, "relu": int(rows.sum())}


def run_conv_case(kind: str, H: int, batch, *, towers: int = 1, pre_layers: int = 2, post_layers: int = 4,
                  seed: int = 0, device: str = "cuda:0", x_rows: Optional[torch.Tensor] = None,
                  deg=None) -> Dict[str, float]:
    """Build oracle (fp32 + fp64) and native conv with identical weights; run fwd + bwd of all three on the same
    x / bond table / upstream gradient.  Returns norm-wise relative errors (max norm ``*_max`` and Frobenius ``*_l2``)
    of the HIP path and of the CPU fp32 oracle against fp64, plus the excluded-row bookkeeping."""
    from gnnepcsaft_amd import nn as gnn
    from gnnepcsaft_amd import ops
    from gnnepcsaft_amd.data import calc_deg
    from gnnepcsaft_amd.nn import Linear, ReLU

    g = torch.Generator().manual_seed(1000 + seed)
    N = batch.x.size(0)
    code = bond_codes(batch.edge_attr)
    # layer inputs shaped like the model's: post-BatchNorm-ReLU activations, xavier-sized bond embeddings
    if x_rows is None:
        x = torch.randn(N, H, generator=g).relu_()
    else:
        x = x_rows.clone()
    BE = (torch.rand(60, H, generator=g) * 2 - 1) * (3.0 * math.sqrt(6.0 / (5 + H)))
    dout = torch.randn(N, H, generator=g)
    torch.manual_seed(seed)
    if kind == "PNA":
        deg = calc_deg(batch) if deg is None else deg
        o32 = O.PNAConv(H, H, ["mean", "min", "max", "std"], ["identity", "amplification", "attenuation"],
                        torch.tensor(deg, dtype=torch.long), H, towers, pre_layers, post_layers, divide_input=True)
        nat = gnn.PNAConv(H, H, ["mean", "min", "max", "std"], ["identity", "amplification", "attenuation"],
                          torch.tensor(deg, dtype=torch.long), H, towers, pre_layers, post_layers, divide_input=True)
    else:
        o32 = O.GINEConv(torch.nn.Sequential(torch.nn.Linear(H, H), torch.nn.ReLU(), torch.nn.Linear(H, H)),
                         train_eps=False, edge_dim=H)
        nat = gnn.GINEConv(nn=torch.nn.Sequential(Linear(H, H), ReLU(), Linear(H, H)), train_eps=False, edge_dim=H)
    res = nat.load_state_dict(o32.state_dict(), strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    o64 = copy.deepcopy(o32).double()

    with torch.no_grad():
        finder = pna_event_rows if kind == "PNA" else gine_event_rows
        info = finder(o64, x.double(), batch.edge_index, BE.double().index_select(0, code))
    excl_rows = info["rows"]
    dout = dout * (~excl_rows).unsqueeze(1).to(dout.dtype)
    keep = ~excl_rows

    def run_oracle(conv, dtype):
        xx = x.detach().clone().to(dtype).requires_grad_(True)
        be = BE.detach().clone().to(dtype).requires_grad_(True)
        out = conv(xx, batch.edge_index, be.index_select(0, code))
        out.backward(dout.to(dtype))
        grads = {n: p.grad for n, p in conv.named_parameters()}
        return out.detach(), xx.grad, be.grad, grads

    out64, dx64, dbe64, g64 = run_oracle(o64, torch.float64)
    out32, dx32, dbe32, g32 = run_oracle(o32, torch.float32)

    nat = nat.to(device)
    b = batch.to(device)
    pack = ops.pack_graph(b.edge_index, b.edge_attr, b.batch, N, batch.num_graphs)
    xx = x.detach().to(device).requires_grad_(True)
    be = BE.detach().to(device).requires_grad_(True)
    out = nat(xx, pack, be)
    out.backward(dout.to(device))
    torch.cuda.synchronize()
    gn = {n: p.grad for n, p in nat.named_parameters()}

    r: Dict[str, float] = {"rows_excluded": int(excl_rows.sum()), "std_entries_in_band": info["std"],
                           "extremum_entries_in_band": info["ext"], "relu_rows_in_band": info["relu"], "rows": N,
                           "edges": int(batch.edge_index.size(1))}
    for tag, (o, dx, dbe, gg) in {"hip": (out.detach().cpu(), xx.grad.cpu(), be.grad.cpu(), gn),
                                  "cpu": (out32, dx32, dbe32, g32)}.items():
        r[f"out_max_{tag}"] = rel_err(o[keep], out64[keep])
        r[f"out_l2_{tag}"] = _l2(o[keep], out64[keep])
        r[f"dx_max_{tag}"] = rel_err(dx, dx64)
        r[f"dx_l2_{tag}"] = _l2(dx, dx64)
        r[f"dbe_max_{tag}"] = rel_err(dbe, dbe64)
        worst_max, worst_l2, name = 0.0, 0.0, ""
        G = max(float(v.abs().max()) for v in g64.values())
        for n, ref in g64.items():
            # biases' gradients can be tiny against the weights': floor the denominator at 1e-3 of the largest entry
            e_max = rel_err(gg[n], ref, floor=1e-3 * G)
            if e_max > worst_max:
                worst_max, name = e_max, n
            worst_l2 = max(worst_l2, _l2(gg[n], ref) if float(ref.norm()) > 1e-3 * G else 0.0)
        r[f"dparam_max_{tag}"], r[f"dparam_l2_{tag}"], r[f"dparam_argmax_{tag}"] = worst_max, worst_l2, name
        r[f"dparam_each_{tag}"] = {n: float(f"{rel_err(gg[n], ref, floor=1e-3 * G):.2e}") for n, ref in g64.items()}
    return r


def model_event_report(model64: "O.GNNePCSAFT", batch) -> Dict[str, object]:
    """Runs the fp64 oracle model once (train mode) and counts, layer by layer, the discrete events inside the fp32
    rounding band (``pna_event_rows`` / ``gine_event_rows`` on every conv's actual input, plus BatchNorm -> ReLU
    outputs within the band of 0 and near-tied maxima of a max pool).  A case whose total is 0 is one where every fp32
    evaluation must agree with fp64 to rounding level; a case with events can differ by the size of an event whichever
    fp32 path (the reference's CPU one included) evaluates it."""
    captured = {}

    def grab(name):
        def hook(mod, args, kwargs, out):
            captured[name] = (args, kwargs, out)
        return hook

    handles = []
    for l, (conv, bn) in enumerate(zip(model64.convs, model64.batch_norms)):
        handles.append(conv.register_forward_hook(grab(f"conv{l}"), with_kwargs=True))
        handles.append(bn.register_forward_hook(grab(f"bn{l}"), with_kwargs=True))
    for k in (1, 4):
        handles.append(model64.mlp[k].register_forward_hook(grab(f"mlp_bn{k}"), with_kwargs=True))
    was_training = model64.training
    saved = {n: b.clone() for n, b in model64.named_buffers()}
    model64.train()
    with torch.no_grad():
        model64(batch.x, batch.edge_index, batch.edge_attr, batch.batch)
    for h in handles:
        h.remove()
    with torch.no_grad():  # the extra forward must not move the running statistics
        for n, b in model64.named_buffers():
            b.copy_(saved[n])
    model64.train(was_training)
    per_layer, total = [], 0
    u = 2.0 ** -24
    with torch.no_grad():
        for l, conv in enumerate(model64.convs):
            _, kw, _ = captured[f"conv{l}"]
            finder = pna_event_rows if isinstance(conv, O.PNAConv) else gine_event_rows
            info = finder(conv, kw["x"], kw["edge_index"], kw["edge_attr"])
            (xin,), _, y = captured[f"bn{l}"]
            bn = model64.batch_norms[l].module
            scale = (y - bn.bias).abs() + bn.bias.abs()
            relu_bn = int((y.abs() <= LIN_BAND_ULPS * u * scale).sum())
            n_ev = int(info["rows"].sum()) + relu_bn
            per_layer.append({"rows": int(info["rows"].sum()), "std": info["std"], "ext": info["ext"],
                              "relu": info["relu"], "relu_after_bn": relu_bn})
            total += n_ev
        for k in (1, 4):
            (xin,), _, y = captured[f"mlp_bn{k}"]
            bn = model64.mlp[k]
            scale = (y - bn.bias).abs() + bn.bias.abs()
            total += int((y.abs() <= LIN_BAND_ULPS * u * scale).sum())
    return {"total": total, "layers": per_layer}
