"""Inference form of the GNN: an eval-only engine built from a trained ``GNNePCSAFT``.

What the reference does at inference (``/root/reference/gnnepcsaft/demo/utils.py:899,950``, ``train/models.py:229-254``):
``model.eval()``, ``pred_with_bounds(graph)`` with ``batch=None`` for one molecule or a PyG batch for many.  In eval
mode BatchNorm is an affine map with constant statistics, so everything that depends on the weights only is evaluated
ONCE here instead of once per call:

* every BatchNorm is folded into the Linear that feeds it (``W' = diag(g/s) W``, ``b' = g/s (b - mean) + beta``,
  ``s = sqrt(running_var + eps)``), and the following ReLU rides in that product's epilogue;
* the 60-row bond tables of every layer (``BondEmb -> edge_encoder -> pre-layer-0 slice`` for PNA, ``lin`` for GINE);
* PNA's per-degree effective post-layer-0 weights ``W_eff(d)`` (built once per degree-class count, then reused);
* the concatenated atom-embedding table.

A call is then: pack -> embedding gather -> per layer {2 node products, gather-combine, edge product, scatter-aggregate,
grouped product, node products} -> pool -> 3 small products, with no autograd tape and no BatchNorm launches.
The result equals ``model.eval()(…)`` up to fp32 re-association of the folded affine maps (``tests/test_inference_gpu.py``).
"""
from __future__ import annotations

from typing import List, Optional

import torch

from . import nn as gnn
from . import ops
from .ops import GraphPack


def _fold(weight: torch.Tensor, bias: Optional[torch.Tensor], bn: torch.nn.BatchNorm1d):
    """(W', b') of BatchNorm_eval(x W^T + b)."""
    s = torch.sqrt(bn.running_var.double() + bn.eps)
    g = (bn.weight.double() if bn.weight is not None else torch.ones_like(s)) / s
    b0 = bias.double() if bias is not None else torch.zeros_like(s)
    beta = bn.bias.double() if bn.bias is not None else torch.zeros_like(s)
    w = (weight.double() * g[:, None]).float().contiguous()
    b = (g * (b0 - bn.running_mean.double()) + beta).float().contiguous()
    return w, b


class InferenceEngine:
    """Forward-only evaluator with the weight-only work precomputed.  Build it again after the model's weights change."""

    def __init__(self, model, max_degree: Optional[int] = None):
        """``model``: a ``GNNePCSAFT`` (or the ``GNNePCSAFTL`` wrapper) on a HIP device.  ``max_degree`` (default: the
        model's ``max_degree_hint``): an upper bound of the in-degree makes packing sync-free; a batch above it trips
        the range flag (``ops.check_range``).  ``None`` reads each batch's maximum back (one sync per call)."""
        model = getattr(model, "model", model)
        p0 = next(model.parameters())
        if not p0.is_cuda:
            raise ValueError("InferenceEngine needs the model on a HIP device (there is no CPU fallback)")
        self.device = p0.device
        self.num_para = model.num_para
        self.pool_type = model.global_pool_type
        self.lower = (model.lower_bounds[:3] if self.num_para == 3 else model.lower_bounds[3:]).to(self.device)
        self.upper = (model.upper_bounds[:3] if self.num_para == 3 else model.upper_bounds[3:]).to(self.device)
        with torch.no_grad():
            self.atom_offsets = model.node_embed.offsets
            self.atom_table = torch.cat([e.weight for e in model.node_embed.atom_embedding_list], 0).contiguous()
            BE = model.edge_embed.table().detach().contiguous()  # [60, H]
            self.R = BE.size(0)
            self.layers: List[dict] = []
            self.is_pna = isinstance(model.convs[0], gnn.PNAConv)
            for conv, bnw in zip(model.convs, model.batch_norms):
                bn = bnw.module
                if self.is_pna:
                    self.layers.append(self._prep_pna(conv, bn, BE))
                else:
                    self.layers.append(self._prep_gine(conv, bn, BE))
            m = model.mlp
            w0, b0 = _fold(m[0].weight, m[0].bias, m[1])
            w3, b3 = _fold(m[3].weight, m[3].bias, m[4])
            self.mlp = (w0, b0, w3, b3, m[6].weight.detach().contiguous(), m[6].bias.detach().contiguous())
        self.max_degree = max_degree if max_degree is not None else getattr(model, "max_degree_hint", None)
        self._graphs: dict = {}

    # -------------------------------------------------------------------------------------------------------------
    def _prep_pna(self, conv: "gnn.PNAConv", bn, BE):
        T, F = conv.towers, conv.F_in
        H = T * F
        avg = conv.aggr_module.avg_log()
        lay = {"T": T, "F": F, "avg": avg, "pre": [], "post": [], "weff": {}}
        EE = ops.gemm([(BE, None, conv.edge_encoder.weight)], torch.empty(self.R, F, device=BE.device),
                      bias=conv.edge_encoder.bias)
        Te = torch.empty(self.R, H, device=BE.device)
        Wi, Wj, weff, wx, bp = [], [], [], [], []
        for t in range(T):
            pre = conv.pre_nns[t].linears()
            post = conv.post_nns[t].linears()
            W0, b0 = pre[0].weight.detach(), pre[0].bias.detach()
            ops.gemm([(EE, None, W0[:, 2 * F:3 * F])], Te[:, t * F:(t + 1) * F], bias=b0)
            Wi.append(W0[:, 0:F].contiguous())
            Wj.append(W0[:, F:2 * F].contiguous())
            lay["pre"].append([(l.weight.detach().contiguous(), l.bias.detach().contiguous()) for l in pre[1:]])
            Wp = post[0].weight.detach().contiguous()
            weff.append(Wp)  # W_eff(d) is built per degree-class count on first use (lay["weff"][D])
            wx.append(Wp[:, 0:F].contiguous())
            bp.append(post[0].bias.detach().contiguous())
            lay["post"].append([(l.weight.detach().contiguous(), l.bias.detach().contiguous()) for l in post[1:]])
        lin_w, lin_b = _fold(conv.lin.weight.detach(), conv.lin.bias.detach(), bn)
        lay.update(Te=Te, Wi=Wi, Wj=Wj, Wp=weff, wx=wx, bp=bp, lin=(lin_w, lin_b),
                   pre_layers=conv.pre_layers, post_layers=conv.post_layers)
        return lay

    def _prep_gine(self, conv: "gnn.GINEConv", bn, BE):
        l0, l2 = conv.nn[0], conv.nn[2]
        Le = ops.gemm([(BE, None, conv.lin.weight)], torch.empty(self.R, conv.lin.weight.size(0), device=BE.device),
                      bias=conv.lin.bias)
        w2, b2 = _fold(l2.weight.detach(), l2.bias.detach(), bn)
        return {"Le": Le, "eps": float(conv.initial_eps), "w0": l0.weight.detach().contiguous(),
                "b0": l0.bias.detach().contiguous(), "w2": w2, "b2": b2}

    # -------------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def __call__(self, x: torch.Tensor, edge_index: torch.Tensor, edge_attr: torch.Tensor,
                 batch: Optional[torch.Tensor] = None, pack: Optional[GraphPack] = None,
                 validate: bool = True) -> torch.Tensor:
        """fp32[B, P] (B = 1 with keepdim semantics when ``batch`` is None), same values as ``model.eval()(…)``."""
        if pack is None:
            pack = ops.pack_graph(edge_index, edge_attr, batch, x.size(0), None, validate=validate)
            pack.max_degree_hint = self.max_degree
        h = ops.embed_sum_fwd(x, self.atom_table, self.atom_offsets)
        for lay in self.layers:
            h = self._pna_layer(h, pack, lay) if self.is_pna else self._gine_layer(h, pack, lay)
        if batch is not None or pack.has_batch:
            g = ops.segment_pool_fwd(h, pack.graph_ptr, pack.B, self.pool_type)
        else:
            g = ops.segment_pool_fwd(h, pack.graph_ptr, 1, self.pool_type)
        w0, b0, w3, b3, w6, b6 = self.mlp
        dev = g.device
        a = ops.gemm([(g, None, w0)], torch.empty(g.size(0), w0.size(0), device=dev), bias=b0, relu=True)
        a = ops.gemm([(a, None, w3)], torch.empty(a.size(0), w3.size(0), device=dev), bias=b3, relu=True)
        return ops.gemm([(a, None, w6)], torch.empty(a.size(0), w6.size(0), device=dev), bias=b6)

    def single(self, x: torch.Tensor, edge_index: torch.Tensor, edge_attr: torch.Tensor) -> torch.Tensor:
        """One molecule (``batch=None``), replayed from a HIP graph captured per (atoms, directed bonds) shape: a call is
        three small copies into the graph's static inputs plus one graph launch instead of ~60 kernel launches.
        Needs ``max_degree`` (sync-free packing).  Returns a fp32[1, P] tensor owned by the graph (valid until the
        next call with the same shape); integer inputs are range-checked lazily (``ops.check_range``)."""
        if self.max_degree is None:
            raise ValueError("InferenceEngine.single needs max_degree (graph capture cannot synchronise)")
        key = (int(x.size(0)), int(edge_index.size(1)))
        slot = self._graphs.get(key)
        if slot is None:
            sx, se, sa = x.clone(), edge_index.clone(), edge_attr.clone()
            stream = torch.cuda.Stream(device=self.device)
            stream.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(stream):
                for _ in range(2):  # warm-up outside the capture (lazy library state, allocator pools)
                    self(sx, se, sa, None, validate=False)
            torch.cuda.current_stream(self.device).wait_stream(stream)
            graph = torch.cuda.CUDAGraph()
            # thread_local: another thread's HIP call (a process group's watchdog) must not abort the capture
            with torch.cuda.graph(graph, stream=stream, capture_error_mode="thread_local"):
                out = self(sx, se, sa, None, validate=False)
            slot = self._graphs[key] = (graph, sx, se, sa, out)
            if len(self._graphs) > 256:  # bounded cache: drop the oldest shape
                self._graphs.pop(next(iter(self._graphs)))
        graph, sx, se, sa, out = slot
        sx.copy_(x)
        se.copy_(edge_index)
        sa.copy_(edge_attr)
        graph.replay()
        return out

    def pred_with_bounds(self, data) -> torch.Tensor:
        """``GNNePCSAFT.pred_with_bounds`` (models.py:229-254) on the folded engine."""
        x, ei, ea = data.x, data.edge_index, data.edge_attr
        if not (isinstance(x, torch.Tensor) and isinstance(ei, torch.Tensor) and isinstance(ea, torch.Tensor)):
            raise ValueError("Invalid input data")
        out = self(x, ei, ea, getattr(data, "batch", None))
        return torch.minimum(torch.maximum(out, self.lower), self.upper)

    # -------------------------------------------------------------------------------------------------------------
    def _pna_layer(self, x: torch.Tensor, pack: GraphPack, lay: dict) -> torch.Tensor:
        T, F = lay["T"], lay["F"]
        N, H = x.shape
        dev = x.device
        P, Q = torch.empty(N, H, device=dev), torch.empty(N, H, device=dev)
        for t in range(T):
            xt = x[:, t * F:(t + 1) * F]
            ops.gemm([(xt, None, lay["Wi"][t])], P[:, t * F:(t + 1) * F])
            ops.gemm([(xt, None, lay["Wj"][t])], Q[:, t * F:(t + 1) * F])
        dc = pack.degree_classes(self.max_degree)
        tiles = pack.edge_tiles(dc.D - 1) if (dc is not None and lay["pre_layers"] == 2 and F % 4 == 0 and F <= 128 and
                                              all(w.is_contiguous() for w, _ in (lay["pre"][t][0] for t in range(T)))) else None
        if tiles is not None:
            # message assembly -> pre-layer 1 -> aggregate in ONE launch; nothing is kept for a backward here, so neither h1
            # nor the messages are written at all (gnx_pna_edge_fwd with NULL h1 / m)
            _, _, A = ops.pna_edge_fwd(P, Q, lay["Te"], pack, T, F, [lay["pre"][t][0][0] for t in range(T)],
                                       [lay["pre"][t][0][1] for t in range(T)], dc.D - 1, keep=False)
        else:
            h = ops.edge_combine_fwd(P, Q, lay["Te"], pack, relu=lay["pre_layers"] > 1)
            for i in range(lay["pre_layers"] - 1):
                hn = torch.empty(pack.E, H, device=dev)
                for t in range(T):
                    w, b = lay["pre"][t][i]
                    ops.gemm([(h[:, t * F:(t + 1) * F], None, w)], hn[:, t * F:(t + 1) * F], bias=b,
                             relu=i < lay["pre_layers"] - 2)
                h = hn
            A = ops.pna_aggregate_fwd(h, pack, T, F)
        z = torch.empty(N, H, device=dev)
        relu0 = lay["post_layers"] > 1
        if dc is not None:
            weff = lay["weff"].get(dc.D)
            if weff is None:  # weight-only: built once per degree-class count, reused by every later call
                weff = lay["weff"][dc.D] = [ops.pna_weff(lay["Wp"][t], F, dc.D, lay["avg"]) for t in range(T)]
        else:
            amp, att = pack.degree_scalers(lay["avg"])
        for t in range(T):
            At = A[:, t * 4 * F:(t + 1) * 4 * F]
            xt = x[:, t * F:(t + 1) * F]
            if dc is not None:
                ops.gemm_grouped([(xt, None, lay["wx"][t], 0), (At, None, weff[t][0], 4 * F * F)],
                                 z[:, t * F:(t + 1) * F], dc, bias=lay["bp"][t], relu=relu0)
            else:  # more than 64 distinct in-degrees: the ungrouped 4-segment product
                Wp = lay["Wp"][t]
                ops.gemm([(xt, None, Wp[:, 0:F]), (At, None, Wp[:, F:5 * F]), (At, amp, Wp[:, 5 * F:9 * F]),
                          (At, att, Wp[:, 9 * F:13 * F])], z[:, t * F:(t + 1) * F], bias=lay["bp"][t], relu=relu0)
        for i in range(lay["post_layers"] - 1):
            zn = torch.empty(N, H, device=dev)
            for t in range(T):
                w, b = lay["post"][t][i]
                ops.gemm([(z[:, t * F:(t + 1) * F], None, w)], zn[:, t * F:(t + 1) * F], bias=b,
                         relu=i < lay["post_layers"] - 2)
            z = zn
        lin_w, lin_b = lay["lin"]
        # lin + folded BatchNorm + the model's ReLU in one product
        return ops.gemm([(z, None, lin_w)], torch.empty(N, H, device=dev), bias=lin_b, relu=True)

    def _gine_layer(self, x: torch.Tensor, pack: GraphPack, lay: dict) -> torch.Tensor:
        N = x.size(0)
        dev = x.device
        agg = ops.gine_aggregate_fwd(x, lay["Le"], pack, lay["eps"])
        a1 = ops.gemm([(agg, None, lay["w0"])], torch.empty(N, lay["w0"].size(0), device=dev), bias=lay["b0"], relu=True)
        return ops.gemm([(a1, None, lay["w2"])], torch.empty(N, lay["w2"].size(0), device=dev), bias=lay["b2"],
                        relu=True)
