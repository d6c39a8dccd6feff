"""A LightningModule-shaped base class (``lightning`` is not installable here): just the surface the reference's
``GNNePCSAFTL`` touches — ``save_hyperparameters``, ``hparams``, ``log``, ``log_dict``, ``device``
(/root/reference/gnnepcsaft/train/models.py:26-34, 94-107, 150-152)."""
from __future__ import annotations

import inspect
from typing import Any, Dict

import torch


class _HParams(dict):
    __getattr__ = dict.get


class LightningModuleLite(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self._hparams = _HParams()
        self.logged_metrics: Dict[str, Any] = {}
        self.trainer = None
        self.global_step = 0

    def save_hyperparameters(self, *args, **kwargs):
        """Collects the ``__init__`` arguments of the calling frame, like Lightning's ``save_hyperparameters()``."""
        frame = inspect.currentframe().f_back
        info = inspect.getargvalues(frame)
        for name in info.args:
            if name != "self":
                self._hparams[name] = info.locals[name]

    @property
    def hparams(self):
        return self._hparams

    @property
    def device(self) -> torch.device:
        try:
            return next(self.parameters()).device
        except StopIteration:
            return torch.device("cpu")

    def log(self, name: str, value, **kwargs):  # pylint: disable=unused-argument
        """Keeps the latest value (tensors stay on device — no sync in the hot loop); ``sync_dist`` is honoured by the
        data-parallel driver (``gnnepcsaft_amd.dp.reduce_logged``)."""
        self.logged_metrics[name] = value.detach() if isinstance(value, torch.Tensor) else value

    def log_dict(self, metrics: Dict[str, Any], **kwargs):
        for k, v in metrics.items():
            self.log(k, v, **kwargs)
