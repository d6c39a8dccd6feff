"""Minimal trainer with ``lightning.Trainer.fit``'s call shape for the GNN path — what
``ltrain_and_evaluate`` needs from Lightning at /root/reference/gnnepcsaft/train/train.py:85-115 (``max_steps``,
``log_every_n_steps``, ``val_check_interval``, several validation loaders, ``ckpt_path``) — plus a ``DataLoader`` that
collates like PyG's.  Orchestration only (Ray / wandb / absl stay out of scope); every step runs the HIP path.

One process per GPU: if ``torch.distributed`` is initialised, gradients are averaged with one flat all-reduce per step
(``dp.FlatGradAllReduce``) and logged scalars are averaged for ``sync_dist``.
"""
from __future__ import annotations

import os
from typing import Iterable, List, Optional, Sequence

import numpy as np
import torch

from .. import dp, functional as Fn, ops
from ..data.batching import Batch, Data
from ..optim import configure_fused_optimizers


class DataLoader:
    """Batches a list of ``Data`` with ``Batch.from_data_list`` (PyG DataLoader semantics: optional shuffle per epoch,
    last batch kept).

    Data parallelism: with ``torch.distributed`` initialised and world size W > 1 (or explicit ``rank`` / ``world``) every
    rank draws the SAME epoch order (same seed on every rank) and takes its strided share ``order[rank::W]`` of it, the
    order being padded by wrapping to a multiple of W -- what Lightning injects into the reference's loaders under DDP
    (``torch.utils.data.DistributedSampler``; /root/reference/gnnepcsaft/train/train.py:85-88): ranks see disjoint
    graphs, every rank iterates the same number of batches, and the effective batch is W x ``batch_size``."""

    def __init__(self, dataset: Sequence[Data], batch_size: int = 1, shuffle: bool = False, seed: int = 0,
                 rank: Optional[int] = None, world: Optional[int] = None, **_ignored):
        self.dataset, self.batch_size, self.shuffle = list(dataset), int(batch_size), bool(shuffle)
        self._rng = np.random.Generator(np.random.PCG64(seed))
        if (rank is None) != (world is None):
            raise ValueError("DataLoader: pass rank and world together (or neither: taken from torch.distributed)")
        self._rank, self._world = rank, world

    def _shard(self):
        if self._world is not None:
            return int(self._rank), int(self._world)
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            return dist.get_rank(), dist.get_world_size()
        return 0, 1

    def _per_rank(self) -> int:
        _, world = self._shard()
        return (len(self.dataset) + world - 1) // world

    def __len__(self) -> int:
        return (self._per_rank() + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        order = np.arange(len(self.dataset))
        if self.shuffle:
            self._rng.shuffle(order)
        rank, world = self._shard()
        if world > 1 and len(order):
            total = self._per_rank() * world
            order = np.resize(order, total)[rank::world]  # np.resize pads by repeating the order from its start
        self.last_order = order
        for i in range(0, len(order), self.batch_size):
            yield Batch.from_data_list([self.dataset[j] for j in order[i:i + self.batch_size]])

    def rng_state(self) -> dict:
        """The shuffle generator's state as plain ints (PCG64: 128-bit state and increment)."""
        st = self._rng.bit_generator.state
        return {"state": int(st["state"]["state"]), "inc": int(st["state"]["inc"]), "has_uint32": int(st["has_uint32"]),
                "uinteger": int(st["uinteger"])}

    def set_rng_state(self, st: dict) -> None:
        self._rng.bit_generator.state = {"bit_generator": "PCG64", "state": {"state": int(st["state"]), "inc": int(st["inc"])},
                                         "has_uint32": int(st["has_uint32"]), "uinteger": int(st["uinteger"])}


# Classes a Lightning ``.ckpt`` of the reference pickles next to its ``state_dict``: ``save_hyperparameters(config)``
# (/root/reference/gnnepcsaft/train/models.py:31) stores an ``ml_collections.ConfigDict``.  ml_collections is not
# installed here, and nothing from a checkpoint may be executed: the safe loader is given inert stand-ins under those
# import paths, which only keep the pickled state.
_INERT_CLASSES = (
    ("ml_collections.config_dict.config_dict", "ConfigDict"),
    ("ml_collections.config_dict.config_dict", "FrozenConfigDict"),
    ("ml_collections.config_dict.config_dict", "FieldReference"),
)


class _Inert:
    """Holds whatever state the pickle carries; runs no code from the file."""

    def __setstate__(self, state):
        self.__dict__["_state"] = state

    def plain(self):
        return _plain(self.__dict__.get("_state", self.__dict__))


def _plain(obj):
    """Recursively turns inert stand-ins / containers into plain dicts and lists (ConfigDict keeps its entries under
    ``_fields``; a FieldReference its value under ``_value``)."""
    if isinstance(obj, _Inert):
        return obj.plain()
    if isinstance(obj, dict):
        if "_fields" in obj and isinstance(obj["_fields"], dict):
            return _plain(obj["_fields"])
        if "_value" in obj and len(obj) <= 4:
            return _plain(obj["_value"])
        return {k: _plain(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_plain(v) for v in obj)
    return obj


def read_checkpoint(path: str) -> dict:
    """Reads this trainer's checkpoints and Lightning ``.ckpt`` files (train.py:121-151 of the reference) with
    ``torch.load(weights_only=True)``; hyper-parameter containers of classes that are not importable come back as
    plain dicts.  Returns the checkpoint dict (``state_dict``, ``global_step``, ``hyper_parameters`` …)."""
    stubs = [type(name, (_Inert,), {"__module__": module}) for module, name in _INERT_CLASSES]
    with torch.serialization.safe_globals(stubs):
        ckpt = torch.load(path, map_location="cpu", weights_only=True)
    return {k: (_plain(v) if k != "state_dict" else v) for k, v in ckpt.items()}


class Trainer:
    def __init__(self, max_steps: int = -1, log_every_n_steps: int = 50, val_check_interval: Optional[int] = None,
                 default_root_dir: Optional[str] = None, enable_checkpointing: bool = True, fused_optimizer: bool = True,
                 device: Optional[str] = None, **_ignored):
        self.max_steps, self.log_every_n_steps = int(max_steps), int(log_every_n_steps)
        self.val_check_interval = val_check_interval
        self.default_root_dir = default_root_dir
        self.enable_checkpointing = enable_checkpointing
        self.fused_optimizer = fused_optimizer
        self.device = torch.device(device) if device else torch.device("cuda", torch.cuda.current_device())
        self.global_step = 0
        self.current_epoch = 0
        self.logged: List[dict] = []
        self.validation_results: List[dict] = []
        self._opt = self._sched = self._loader = None
        self._batch_in_epoch = 0
        self._epoch_rng = None

    # ------------------------------------------------------------------------------------------------------------
    def save_checkpoint(self, model, path: str) -> None:
        """Weights AND everything a resumed run needs to continue exactly where this one stopped (Lightning's
        ``ckpt_path`` resume, reference train.py:105-115): optimizer moments / step count / learning rate, LR-scheduler
        state, epoch, position inside the epoch with the shuffle generator's state at the epoch's start, and the
        dropout counter.  Tensors, numbers, lists and dicts only: ``read_checkpoint`` loads it with
        ``weights_only=True``.  Under data parallelism rank 0 writes and everybody waits for it."""
        import torch.distributed as dist
        ddp = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        if not ddp or dist.get_rank() == 0:
            os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
            ckpt = {"state_dict": {k: v.detach().cpu() for k, v in model.state_dict().items()},
                    "global_step": self.global_step, "epoch": self.current_epoch,
                    "hyper_parameters": {"config": dict(model.config)}, "loops": {
                        "batch_in_epoch": self._batch_in_epoch, "loader_rng_at_epoch_start": self._epoch_rng,
                        "loader_rng_now": self._loader.rng_state() if hasattr(self._loader, "rng_state") else None}}
            if self._opt is not None:
                ckpt["optimizer_states"] = [self._opt.flat_state_dict() if hasattr(self._opt, "flat_state_dict")
                                            else self._opt.state_dict()]
                ckpt["lr_schedulers"] = [self._sched.state_dict()]
            drop = getattr(model.model, "dropout", None)
            if drop is not None and hasattr(drop, "calls"):
                ckpt["dropout"] = {"seed": int(drop.seed), "calls": int(drop.calls)}
            tmp = path + ".tmp"
            torch.save(ckpt, tmp)
            os.replace(tmp, path)
        if ddp:
            dist.barrier()

    @staticmethod
    def load_state_dict(model, path: str) -> dict:
        """Reads a checkpoint's ``state_dict`` (this trainer's, or a Lightning ``.ckpt``) with ``weights_only=True`` —
        nothing from the file is executed (``read_checkpoint``)."""
        ckpt = read_checkpoint(path)
        model.load_state_dict(ckpt["state_dict"])
        return ckpt

    # ------------------------------------------------------------------------------------------------------------
    def fit(self, model, train_dataloaders: Iterable, val_dataloaders: Optional[Sequence[Iterable]] = None,
            ckpt_path: Optional[str] = None) -> None:
        model.to(self.device)
        model.train()
        model.trainer = self
        ckpt = None
        if ckpt_path:
            ckpt = self.load_state_dict(model, ckpt_path)
            self.global_step = int(ckpt.get("global_step", 0))
            self.current_epoch = int(ckpt.get("epoch", 0))
        dp.broadcast_parameters(model)
        grads = dp.FlatGradAllReduce(model)
        oc = configure_fused_optimizers(model, grads) if self.fused_optimizer else model.configure_optimizers()
        opt = oc["optimizer"]
        sched, freq = oc["lr_scheduler"]["scheduler"], int(oc["lr_scheduler"].get("frequency", 1))
        self._opt, self._sched, self._loader = opt, sched, train_dataloaders
        skip = 0
        if ckpt is not None:  # the rest of the training state (absent in weights-only / foreign checkpoints)
            if ckpt.get("optimizer_states"):
                st = ckpt["optimizer_states"][0]
                opt.load_flat_state_dict(st) if hasattr(opt, "load_flat_state_dict") and "kind" in st else opt.load_state_dict(st)
            if ckpt.get("lr_schedulers"):
                sched.load_state_dict(ckpt["lr_schedulers"][0])
                for g, lr in zip(opt.param_groups, sched.get_last_lr()):
                    g["lr"] = lr
            loops = ckpt.get("loops") or {}
            skip = int(loops.get("batch_in_epoch", 0))
            if hasattr(train_dataloaders, "set_rng_state"):
                # mid-epoch: replay the interrupted epoch's shuffle and skip what was consumed; else continue the stream
                st = loops.get("loader_rng_at_epoch_start") if skip > 0 else loops.get("loader_rng_now")
                if st is not None:
                    train_dataloaders.set_rng_state(st)
            # a checkpoint written again before any further step (e.g. global_step >= max_steps) must keep the position
            self._batch_in_epoch = skip
            self._epoch_rng = loops.get("loader_rng_at_epoch_start") if skip > 0 else None
            drop = getattr(model.model, "dropout", None)
            if ckpt.get("dropout") and drop is not None and hasattr(drop, "calls"):
                drop.seed, drop.calls = int(ckpt["dropout"]["seed"]), int(ckpt["dropout"]["calls"])
        prev_in_place = Fn._GRAD_IN_PLACE  # pylint: disable=protected-access
        prev_validate = model.model.validate_inputs
        Fn.set_grad_in_place(self.fused_optimizer)
        model.model.validate_inputs = False  # one range check per logging interval instead of one sync per batch
        try:
            while self.max_steps < 0 or self.global_step < self.max_steps:
                self._epoch_rng = train_dataloaders.rng_state() if hasattr(train_dataloaders, "rng_state") else None
                self._batch_in_epoch = 0
                for batch in train_dataloaders:
                    self._batch_in_epoch += 1
                    if skip >= self._batch_in_epoch:
                        continue  # batches of the interrupted epoch that the checkpointed run had already consumed
                    b = batch.to(self.device, non_blocking=True)
                    if self.fused_optimizer:
                        opt.zero_grad()
                    else:
                        grads.zero_grad()
                    loss = model.training_step(b, self.global_step)
                    loss.backward()
                    grads.all_reduce()
                    opt.step()
                    self.global_step += 1
                    model.global_step = self.global_step
                    if self.global_step % self.log_every_n_steps == 0:
                        ops.check_range(self.device)
                        rec = dp.reduce_logged(model.logged_metrics)
                        rec.update(step=self.global_step, epoch=self.current_epoch, lr=opt.param_groups[0]["lr"])
                        self.logged.append(rec)
                    if self.val_check_interval and val_dataloaders and \
                            self.global_step % int(self.val_check_interval) == 0:
                        self._validate(model, val_dataloaders)
                    if 0 <= self.max_steps <= self.global_step:
                        break
                else:
                    # the epoch ran to its end (no break): epoch bookkeeping + the scheduler's cadence
                    skip = 0
                    self._batch_in_epoch = 0
                    self._epoch_rng = None
                    self.current_epoch += 1
                    if self.current_epoch % freq == 0:  # "interval": "epoch", "frequency": 10 (models.py:72-73)
                        sched.step()
                    if self.max_steps < 0:
                        break  # one epoch when no step budget is given
                    continue
                # stopped inside an epoch by max_steps
                try:
                    n_batches = len(train_dataloaders)
                except TypeError:  # an iterable without __len__: the epoch's end is only known once it is exhausted
                    n_batches = None
                if n_batches is not None and self._batch_in_epoch >= n_batches:  # exactly at its last batch: the epoch is complete
                    self._batch_in_epoch = 0
                    self._epoch_rng = None
                    self.current_epoch += 1
                    if self.current_epoch % freq == 0:
                        sched.step()
                break
        finally:
            Fn.set_grad_in_place(prev_in_place)
            model.model.validate_inputs = prev_validate
        ops.check_range(self.device)
        if self.enable_checkpointing and self.default_root_dir:
            self.save_checkpoint(model, os.path.join(self.default_root_dir, "last.ckpt"))

    def _validate(self, model, val_dataloaders) -> None:
        if model.rho_batch is None or model.vp_batch is None:
            return  # the CPU PC-SAFT oracle is not wired in: nothing to evaluate (see GNNePCSAFTL.validation_step)
        was_training = model.training
        model.eval()
        with torch.no_grad():
            for idx, loader in enumerate(val_dataloaders):
                for i, batch in enumerate(loader):
                    out = model.validation_step(batch.to(self.device), i, idx)
                    self.validation_results.append({"step": self.global_step, "dataloader_idx": idx, **out})
        model.train(was_training)
