"""Shared helpers for parity tests: build the oracle model and the native model with identical weights, run both on
the same batch, report norm-wise relative errors.  The oracle is the checker here, never the thing measured.

Why a three-way comparison: the reference algorithm is ill-conditioned in fp32 at random initialisation — PyG's
StdAggregation computes mean(x^2) - mean(x)^2 (cancellation amplified by |mean|/std ~ 1e2) and hard-masks std at
var <= 1e-5 (a discontinuity), so the CPU fp32 path itself only agrees with its own fp64 evaluation to 1e-4..1e-3 on
predictions and worse on gradients (measured: tests/golden/conditioning.json).  Any re-association of the fp32 GEMM
sums (MKL with another thread count, or MFMA tiles) moves the result by that much.  So end-to-end we assert that the
HIP path is as close to the exact (fp64) answer as the reference's fp32 path is, and assert the north-star 1e-5 where
the path is well conditioned (every op on identical inputs: tests/test_ops_gpu.py; GINE models end-to-end).
"""
from __future__ import annotations

import copy
from typing import Dict, Optional

import torch

from oracle import pyg_restatement as O


def rel_err(a: torch.Tensor, ref: torch.Tensor, floor: float = 0.0) -> float:
    """max|a - ref| / max(max|ref|, floor)  (norm-wise relative error in the max norm; 0/0 -> 0)."""
    a = a.detach().double().cpu()
    ref = ref.detach().double().cpu()
    den = max(float(ref.abs().max()) if ref.numel() else 0.0, floor)
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


def _grads(model) -> Dict[str, torch.Tensor]:
    return {n: p.grad.detach().double().cpu() for n, p in model.named_parameters()}


def grad_errors(g: Dict[str, torch.Tensor], ref: Dict[str, torch.Tensor]) -> Dict[str, float]:
    """Global relative L2 error of the whole gradient vector, and the worst per-parameter max-norm error with the
    denominator floored at 1 % of the largest gradient entry (biases that feed a BatchNorm have an analytically zero
    gradient: a plain relative error on them is noise over noise)."""
    num = sum(float(((g[n] - ref[n]) ** 2).sum()) for n in ref)
    den = sum(float((ref[n] ** 2).sum()) for n in ref)
    G = max(float(ref[n].abs().max()) for n in ref)
    worst, name = 0.0, ""
    for n in ref:
        e = rel_err(g[n], ref[n], floor=1e-2 * G)
        if e > worst:
            worst, name = e, n
    return {"l2": (num / den) ** 0.5 if den > 0 else 0.0, "max": worst, "argmax": name}


def compare_with_oracle(cfg: dict, batch, device: str = "cuda:0", seed: int = 0, target: str = "para",
                        with_fp64: bool = True) -> Dict[str, float]:
    """One training-mode forward + APE-Huber loss + backward on the HIP path, the CPU fp32 oracle and (optionally) the
    oracle in fp64; returns errors hip-vs-cpu32 (``*_rel``), hip-vs-fp64 (``*_hip64``) and cpu32-vs-fp64 (``*_cpu64``)."""
    from gnnepcsaft_amd import functional as Fn
    from gnnepcsaft_amd.data import calc_deg
    cfg = copy.deepcopy(cfg)
    cfg["deg"] = calc_deg(batch)
    oracle, native = make_models(cfg, seed)
    oracle.train()
    native.train()
    tgt = getattr(batch, target)
    o64 = copy.deepcopy(oracle).double() if with_fp64 else None

    pred_o = oracle(batch.x, batch.edge_index, batch.edge_attr, batch.batch)
    loss_o = O.ape_huber_loss(pred_o, tgt)
    loss_o.backward()

    native = native.to(device)
    b = batch.to(device)
    pred_n = native(b.x, b.edge_index, b.edge_attr, b.batch)
    loss_n, both = Fn.HuberAPEFn.apply(pred_n, getattr(b, target), 0.01)
    loss_n.backward()
    torch.cuda.synchronize()

    g_o, g_n = _grads(oracle), _grads(native)
    ge = grad_errors(g_n, g_o)
    out = {"pred_rel": rel_err(pred_n, pred_o), "loss_rel": rel_err(loss_n, loss_o),
           "mape_rel": rel_err(both[1], O.mape(pred_o.detach(), tgt)),
           "grad_rel_l2": ge["l2"], "grad_rel_max": ge["max"], "grad_rel_argmax": ge["argmax"]}
    bo = dict(oracle.named_buffers())
    wb = 0.0
    for name, bf in native.named_buffers():
        if name in bo and bf.dtype.is_floating_point:
            wb = max(wb, rel_err(bf, bo[name]))
    out["buffer_rel_max"] = wb
    if with_fp64:
        pred_64 = o64(batch.x, batch.edge_index, batch.edge_attr, batch.batch)
        loss_64 = O.ape_huber_loss(pred_64, tgt.double())
        loss_64.backward()
        g_64 = _grads(o64)
        e_h, e_c = grad_errors(g_n, g_64), grad_errors(g_o, g_64)
        out.update(pred_hip64=rel_err(pred_n, pred_64), pred_cpu64=rel_err(pred_o, pred_64),
                   loss_hip64=rel_err(loss_n, loss_64), loss_cpu64=rel_err(loss_o, loss_64),
                   grad_l2_hip64=e_h["l2"], grad_l2_cpu64=e_c["l2"], grad_max_hip64=e_h["max"],
                   grad_max_cpu64=e_c["max"])
    return out


def assert_as_close_as_cpu_fp32(res: Dict[str, float], slack: float = 3.0) -> None:
    """The HIP path may sit no further from the fp64 answer than ``slack`` x the reference's CPU fp32 path, plus the
    spread of that CPU-fp32-vs-fp64 distance itself across this suite's PNA cases (predictions 1e-5..9e-4, gradient
    L2 2e-4..4e-3, worst per-parameter gradient 3e-3..2e-2: discrete events of the reference algorithm — a
    StdAggregation mask flip at var ~ 1e-5, a ReLU / arg-extremum flip — hit either fp32 path at random, so a single
    case can favour either side by an order of magnitude).  The loss (a mean over all outputs) is held to 1e-5."""
    eps = {"pred": 1e-3, "loss": 1e-5, "grad_l2": 5e-3, "grad_max": 2e-2}
    for k in ("pred", "loss", "grad_l2", "grad_max"):
        hip, cpu = res[f"{k}_hip64"], res[f"{k}_cpu64"]
        assert hip <= slack * cpu + eps[k], (k, hip, cpu, res)
