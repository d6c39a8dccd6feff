"""Device-side optimizer step for the hot path's parameters.

``FusedAdamW`` / ``FusedSGD`` are ``torch.optim.Optimizer`` subclasses (so the reference's
``CosineAnnealingWarmRestarts`` scheduler drives them unchanged, /root/reference/gnnepcsaft/train/models.py:66-75) whose
``step()`` is ONE kernel over flat fp32 buffers: parameters are re-pointed to views of one flat buffer, gradients live in
``dp.FlatGradAllReduce``'s flat buffer (the all-reduce payload), Adam moments are flat too."""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib
from ._lib import check, handle
from .dp import FlatGradAllReduce


def flatten_parameters(module: torch.nn.Module) -> torch.Tensor:
    """Move every trainable parameter's storage into one flat buffer (``p.data`` become views); returns the buffer."""
    params = [p for p in module.parameters() if p.requires_grad]
    total = sum(p.numel() for p in params)
    flat = torch.empty(total, dtype=params[0].dtype, device=params[0].device)
    off = 0
    for p in params:
        n = p.numel()
        flat[off:off + n].copy_(p.data.reshape(-1))
        p.data = flat[off:off + n].view_as(p)
        off += n
    return flat


class _FlatOptimizer(torch.optim.Optimizer):
    def __init__(self, module: torch.nn.Module, defaults: dict, grads: Optional[FlatGradAllReduce] = None):
        params = [p for p in module.parameters() if p.requires_grad]
        super().__init__(params, defaults)
        self.flat_param = flatten_parameters(module)
        self.grads = grads if grads is not None else FlatGradAllReduce(module)
        if self.grads.flat.numel() != self.flat_param.numel():
            raise ValueError("gradient and parameter flat buffers differ in size")
        self._step = 0

    def zero_grad(self, set_to_none: bool = False):  # pylint: disable=arguments-differ
        self.grads.zero_grad()  # keeps p.grad as views of the flat gradient buffer

    # ---- checkpointing: plain tensors and numbers only, so ``torch.load(weights_only=True)`` reads it back -----------
    _FLAT_STATE = ()

    def flat_state_dict(self) -> dict:
        out = {"kind": type(self).__name__, "step": int(self._step), "numel": int(self.flat_param.numel()),
               "param_groups": [{k: (list(v) if isinstance(v, tuple) else v) for k, v in g.items() if k != "params"}
                                for g in self.param_groups]}
        for name in self._FLAT_STATE:
            out[name] = getattr(self, name).detach().cpu().clone()
        return out

    def load_flat_state_dict(self, state: dict) -> None:
        if state.get("kind") != type(self).__name__ or int(state.get("numel", -1)) != self.flat_param.numel():
            raise ValueError(f"optimizer state of {state.get('kind')} with {state.get('numel')} elements does not fit "
                             f"{type(self).__name__} with {self.flat_param.numel()}")
        self._step = int(state["step"])
        for g, sg in zip(self.param_groups, state["param_groups"]):
            for k, v in sg.items():
                g[k] = tuple(v) if k == "betas" else v
        for name in self._FLAT_STATE:
            getattr(self, name).copy_(state[name].to(self.flat_param.device))


class FusedAdamW(_FlatOptimizer):
    """torch.optim.AdamW(lr, weight_decay, amsgrad=True, eps=1e-5) semantics, one kernel per step."""
    _FLAT_STATE = ("exp_avg", "exp_avg_sq", "max_exp_avg_sq")

    def __init__(self, module, lr=1e-3, betas=(0.9, 0.999), eps=1e-5, weight_decay=1e-2, grads=None):
        super().__init__(module, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=True), grads)
        z = torch.zeros_like(self.flat_param)
        self.exp_avg, self.exp_avg_sq, self.max_exp_avg_sq = z, z.clone(), z.clone()

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        g = self.param_groups[0]
        self._step += 1
        check(_lib.load().gnx_adamw_amsgrad(handle(self.flat_param.device), self.flat_param.data_ptr(),
                                            self.grads.flat.data_ptr(), self.exp_avg.data_ptr(),
                                            self.exp_avg_sq.data_ptr(), self.max_exp_avg_sq.data_ptr(),
                                            self.flat_param.numel(), float(g["lr"]), float(g["betas"][0]),
                                            float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"]), self._step))
        return loss


class FusedSGD(_FlatOptimizer):
    """torch.optim.SGD(lr, momentum=0, weight_decay=0, nesterov=False) semantics."""

    def __init__(self, module, lr=1e-3, grads=None):
        super().__init__(module, dict(lr=lr), grads)

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        check(_lib.load().gnx_sgd(handle(self.flat_param.device), self.flat_param.data_ptr(), self.grads.flat.data_ptr(),
                                  self.flat_param.numel(), float(self.param_groups[0]["lr"])))
        return loss


def configure_fused_optimizers(lit_module, grads: Optional[FlatGradAllReduce] = None) -> dict:
    """Same dict shape as ``GNNePCSAFTL.configure_optimizers`` (models.py:47-75) with the fused optimizers."""
    from torch.optim.lr_scheduler import CosineAnnealingWarmRestarts
    cfg = lit_module.config
    if cfg["optimizer"] == "adam":
        opt = FusedAdamW(lit_module, lr=cfg["learning_rate"], weight_decay=cfg["weight_decay"], eps=1e-5, grads=grads)
    elif cfg["optimizer"] == "sgd":
        opt = FusedSGD(lit_module, lr=cfg["learning_rate"], grads=grads)
    else:
        raise ValueError(f"Unsupported optimizer: {cfg['optimizer']}.")
    return {"optimizer": opt,
            "lr_scheduler": {"scheduler": CosineAnnealingWarmRestarts(opt, cfg["warmup_steps"], T_mult=2, eta_min=1e-6),
                             "interval": "epoch", "frequency": 10}}
