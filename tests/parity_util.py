"""Shared helpers for parity tests: build the oracle model and the native model with identical weights, run both on
the same batch, report norm-wise relative errors.  The oracle is the checker here, never the thing measured."""
from __future__ import annotations

import copy
from typing import Dict

import torch

from oracle import pyg_restatement as O


def rel_err(a: torch.Tensor, ref: torch.Tensor) -> float:
    """max|a - ref| / max|ref|  (norm-wise relative error in the max norm; 0/0 -> 0)."""
    a = a.detach().double().cpu()
    ref = ref.detach().double().cpu()
    den = float(ref.abs().max()) if ref.numel() else 0.0
    num = float((a - ref).abs().max()) if ref.numel() else 0.0
    if den == 0.0:
        return 0.0 if num == 0.0 else float("inf")
    return num / den


def make_models(cfg: dict, seed: int = 0):
    """(oracle model on CPU, native model on CPU with the same state dict)."""
    from gnnepcsaft_amd.train.models import GNNePCSAFT
    torch.manual_seed(seed)
    oracle = O.GNNePCSAFT(cfg)
    native = GNNePCSAFT(cfg)
    missing = native.load_state_dict(oracle.state_dict(), strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return oracle, native


def compare_with_oracle(cfg: dict, batch, device: str = "cuda:0", seed: int = 0, target: str = "para",
                        dtype64_ref: bool = False) -> Dict[str, float]:
    """One training-mode forward + APE-Huber loss + backward on both paths; returns relative errors."""
    from gnnepcsaft_amd import functional as Fn
    from gnnepcsaft_amd.data import calc_deg
    cfg = copy.deepcopy(cfg)
    cfg["deg"] = calc_deg(batch)
    oracle, native = make_models(cfg, seed)
    oracle.train()
    native.train()
    tgt = getattr(batch, target)
    if dtype64_ref:
        oracle = oracle.double()
    pred_o = oracle(batch.x, batch.edge_index, batch.edge_attr, batch.batch)
    loss_o = O.ape_huber_loss(pred_o, tgt.to(pred_o.dtype))
    loss_o.backward()

    native = native.to(device)
    b = batch.to(device)
    pred_n = native(b.x, b.edge_index, b.edge_attr, b.batch)
    loss_n, both = Fn.HuberAPEFn.apply(pred_n, getattr(b, target), 0.01)
    loss_n.backward()
    torch.cuda.synchronize()

    out = {"pred_rel": rel_err(pred_n, pred_o), "loss_rel": rel_err(loss_n, loss_o),
           "mape_rel": rel_err(both[1], O.mape(pred_o.detach(), tgt.to(pred_o.dtype)))}
    worst, worst_name = 0.0, ""
    po = dict(oracle.named_parameters())
    for name, p in native.named_parameters():
        g_ref = po[name].grad
        assert p.grad is not None, f"no grad for {name}"
        e = rel_err(p.grad, g_ref)
        if e > worst:
            worst, worst_name = e, name
    out["grad_rel_max"] = worst
    out["grad_rel_argmax"] = worst_name
    bo = dict(oracle.named_buffers())
    wb = 0.0
    for name, bf in native.named_buffers():
        if name in bo and bf.dtype.is_floating_point:
            wb = max(wb, rel_err(bf, bo[name]))
    out["buffer_rel_max"] = wb
    return out
