"""Drop-in for the GNN part of ``/root/reference/gnnepcsaft/train/models.py``: same class / function / attribute /
state-dict names and config keys, with the forward/backward running on hand-written gfx950 kernels.

Mirrors: ``GNNePCSAFTL`` (:23-156), ``GNNePCSAFT`` (:159-254), ``get_conv`` (:441-584), ``get_global_pool`` (:587-595),
``create_model`` (:598-606).  Out of scope: the HabitchNN MLP baseline (:257-438) and every conv other
than PNA / GINE — the dispatch is kept and raises for them.
"""
from __future__ import annotations

import math
from typing import Any, Optional, Union

import torch
from torch.nn import Dropout, ModuleList, Sequential
from torch.optim.lr_scheduler import CosineAnnealingWarmRestarts

from .. import functional as Fn
from .. import nn as gnn
from .. import ops
from ..nn import BatchNorm, BatchNorm1d, Linear, ReLU
from ..ops import GraphPack
from .lightning_lite import LightningModuleLite

_OUT_OF_SCOPE_CONVS = ("GCN", "GAT", "GATv2", "Transformer", "SAGE", "GIN", "Edge", "GatedGraph", "Graph", "ARMA", "SG")


def _pack_of(graphs, validate: bool, max_degree_hint: Optional[int] = None) -> GraphPack:
    """GraphPack of a Batch-like object, cached on it (the packer runs once per batch, like PyG's collate)."""
    pack = getattr(graphs, "_gnx_pack", None)
    if pack is None or pack.device != graphs.x.device:
        batch = getattr(graphs, "batch", None)
        num_graphs = getattr(graphs, "num_graphs", None) if batch is not None else None
        pack = ops.pack_graph(graphs.edge_index, graphs.edge_attr, batch, graphs.x.size(0), num_graphs,
                              validate=validate)
        pack.max_degree_hint = max_degree_hint
        try:
            graphs._gnx_pack = pack
        except AttributeError:
            pass
    return pack


class GNNePCSAFTL(LightningModuleLite):
    """Graph neural network to predict PCSAFT parameters (Lightning-style wrapper; reference :23-156)."""

    def __init__(self, config: dict[str, Any]):
        super().__init__()
        self.save_hyperparameters()
        self.config = config
        self.model = GNNePCSAFT(config)
        # hooks for the CPU PC-SAFT label oracle (feos), which stays outside this package (north star)
        self.rho_batch = None
        self.vp_batch = None

    def forward(self, x: torch.Tensor, edge_index: torch.Tensor, edge_attr: torch.Tensor,
                batch: torch.Tensor, pack: Optional[GraphPack] = None) -> torch.Tensor:
        """Forward pass of the model"""
        return self.model(x, edge_index, edge_attr, batch, pack=pack)

    def configure_optimizers(self):
        if self.config["optimizer"] == "adam":
            opt = torch.optim.AdamW(self.parameters(), lr=self.config["learning_rate"],
                                    weight_decay=self.config["weight_decay"], amsgrad=True, eps=1e-5)
        elif self.config["optimizer"] == "sgd":
            opt = torch.optim.SGD(self.parameters(), lr=self.config["learning_rate"], momentum=0.0, weight_decay=0.0,
                                  nesterov=False)
        else:
            raise ValueError(f"Unsupported optimizer: {self.config['optimizer']}.")
        return {
            "optimizer": opt,
            "lr_scheduler": {
                "scheduler": CosineAnnealingWarmRestarts(opt, self.config["warmup_steps"], T_mult=2, eta_min=1e-6),
                "interval": "epoch",
                "frequency": 10,
            },
        }

    def training_step(self, graphs, batch_idx):  # pylint: disable=W0613
        if self.config["dataset"] in ("esper_assoc", "esper_assoc_only"):
            target: torch.Tensor = graphs.assoc
        else:
            target: torch.Tensor = graphs.para
        x, edge_index, edge_attr, batch = graphs.x, graphs.edge_index, graphs.edge_attr, graphs.batch
        pred: torch.Tensor = self(x, edge_index, edge_attr, batch,
                                  pack=_pack_of(graphs, self.model.validate_inputs, self.model.max_degree_hint))
        # ape = (pred - target) / target ; huber(ape, 0, delta=0.01) ; mape(pred, target) -- one kernel
        loss, both = Fn.HuberAPEFn.apply(pred, target, 0.01)
        self.log("train_huber", loss, on_step=True, batch_size=target.shape[0], sync_dist=True)
        self.log("train_mape", both[1], on_step=True, batch_size=target.shape[0], sync_dist=True)
        return loss

    def validation_step(self, graphs, batch_idx, dataloader_idx: int = 0):  # pylint: disable=W0613
        """Same contract as the reference (:110-153).  The density / vapour-pressure evaluation needs the CPU PC-SAFT
        solver (``rho_batch`` / ``vp_batch`` of the reference's train/utils.py:252-300, feos), which is out of this
        package's scope: assign callables to ``self.rho_batch`` / ``self.vp_batch`` to enable it."""
        import numpy as np

        if self.rho_batch is None or self.vp_batch is None:
            raise RuntimeError("validation_step needs the CPU PC-SAFT oracle: set .rho_batch and .vp_batch "
                               "(reference gnnepcsaft/train/utils.py:252-300); it is out of scope here")
        metrics_dict = {}
        pred_para = self.model.pred_with_bounds(graphs).squeeze().detach()
        if self.config["num_para"] == 2:
            para_assoc = 10 ** (pred_para * torch.tensor([-1.0, 1.0], device=pred_para.device))
            para_msigmae = graphs.para
        else:
            para_assoc = 10 ** (graphs.assoc * torch.tensor([-1.0, 1.0], device=pred_para.device))
            para_msigmae = pred_para
        all_pred_para = (torch.hstack([para_msigmae, para_assoc, graphs.munanb, graphs.mw]).cpu().to(torch.float64)
                         .tolist())
        pred_rho = self.rho_batch(all_pred_para, graphs.rho)
        pred_vp = self.vp_batch(all_pred_para, graphs.vp)
        rho = [rho[:, -1] for rho in graphs.rho if rho.shape[0] > 0]
        vp = [vp[:, -1] for vp in graphs.vp if vp.shape[0] > 0]
        mape_den = [np.mean(np.abs(pred - exp) / exp).item() for pred, exp in zip(pred_rho, rho)]
        mape_vp = [np.mean(np.abs(pred - exp) / exp).item() for pred, exp in zip(pred_vp, vp)]
        metrics_dict.update({"mape_den": np.asarray(mape_den).mean().item(), "mape_vp": np.asarray(mape_vp).mean().item()})
        self.log_dict(metrics_dict, on_step=False, on_epoch=True, batch_size=1, sync_dist=True)
        return metrics_dict

    def test_step(self, graphs, batch_idx, dataloader_idx=0):
        return self.validation_step(graphs, batch_idx, dataloader_idx)


class GNNePCSAFT(torch.nn.Module):  # pylint: disable=R0902
    """Graph neural network to predict PCSAFT parameters (reference :159-254)."""

    def __init__(self, config: dict):
        super().__init__()
        self.convs = ModuleList()
        self.batch_norms = ModuleList()
        self.lower_bounds = torch.tensor([1.0, 1.9, 50.0, -1 * math.log10(0.9), math.log10(200.0)])
        self.upper_bounds = torch.tensor([25.0, 4.5, 550.0, -1 * math.log10(0.0001), math.log10(5000.0)])
        self.num_para = config["num_para"]

        self.node_embed = gnn.AtomEncoder(config["hidden_dim"])
        self.edge_embed = gnn.BondEncoder(config["hidden_dim"])
        self.dropout = Dropout(p=config["dropout"])
        self.global_pool = get_global_pool(config)
        self.global_pool_type = config["global_pool"]

        for _ in range(config["propagation_depth"]):
            self.convs.append(get_conv(config))
            self.batch_norms.append(BatchNorm(config["hidden_dim"]))

        self.mlp = Sequential(
            Linear(config["hidden_dim"], config["hidden_dim"] // 2),
            BatchNorm1d(config["hidden_dim"] // 2),
            ReLU(),
            Linear(config["hidden_dim"] // 2, config["hidden_dim"] // 4),
            BatchNorm1d(config["hidden_dim"] // 4),
            ReLU(),
            Linear(config["hidden_dim"] // 4, config["num_para"]),
        )
        # integer inputs are range-checked on device; True = read the flag back (one sync) when a batch is packed
        self.validate_inputs = True
        # sync-free packing (HIP-graph capture): upper bound of the in-degree, e.g. len(config["deg"]) - 1; a batch that
        # exceeds it trips the range flag (ops.check_range).  None = read the batch's maximum back (one sync).
        self.max_degree_hint = None
        self._bounds_cache = {}

    def forward(self, x: torch.Tensor, edge_index: torch.Tensor, edge_attr: torch.Tensor,
                batch: Union[torch.Tensor, None], pack: Optional[GraphPack] = None) -> torch.Tensor:
        """Forward pass of the model.  ``pack`` (optional) is the pre-built GraphPack of (edge_index, edge_attr, batch)."""
        if pack is None:
            pack = ops.pack_graph(edge_index, edge_attr, batch, x.size(0), None, validate=self.validate_inputs)
        x = self.node_embed(x)
        edge_attr = self.edge_embed.table()  # 60 encoded bond-feature combinations; edges index it by pack.code

        for conv, batch_norm in zip(self.convs, self.batch_norms):
            x = self.dropout(x)
            # PNA and GINE both take edge_attr (reference :211-214); relu fused into the BatchNorm kernel
            x = batch_norm(conv(x=x, edge_index=pack, edge_attr=edge_attr), relu=True)

        if batch is not None or pack.has_batch:
            x = self.global_pool(x, pack)
        else:  # batch None: reduce over all rows, keepdim (reference :220-225)
            x = Fn.SegmentPoolFn.apply(x, pack.graph_ptr, 1, self.global_pool_type)
        # readout mlp (reference :186-194, :226): Linear -> BN -> ReLU -> Linear -> BN -> ReLU -> Linear
        x = self.mlp[1](self.mlp[0](x), relu=True)
        x = self.mlp[4](self.mlp[3](x), relu=True)
        return self.mlp[6](x)

    def pred_with_bounds(self, data):
        """Forward pass of the model with bounds."""
        x, edge_index, edge_attr, batch = data.x, data.edge_index, data.edge_attr, data.batch
        if isinstance(x, torch.Tensor) and isinstance(edge_index, torch.Tensor) and isinstance(edge_attr, torch.Tensor):
            params = self.forward(x, edge_index, edge_attr, batch, pack=_pack_of(data, self.validate_inputs))
            key = (x.device, self.num_para)
            if key not in self._bounds_cache:
                upper = (self.upper_bounds[:3] if self.num_para == 3 else self.upper_bounds[3:]).to(device=x.device)
                lower = (self.lower_bounds[:3] if self.num_para == 3 else self.lower_bounds[3:]).to(device=x.device)
                self._bounds_cache[key] = (lower.contiguous(), upper.contiguous())
            lower, upper = self._bounds_cache[key]
            return ops.clip_rows(params, lower, upper)
        raise ValueError("Invalid input data")


def get_conv(config: dict):
    """Returns the convolution layer."""
    aggregators = ["mean", "min", "max", "std"]
    scalers = ["identity", "amplification", "attenuation"]
    if config["conv"] == "PNA":
        return gnn.PNAConv(
            in_channels=config["hidden_dim"],
            out_channels=config["hidden_dim"],
            aggregators=aggregators,
            scalers=scalers,
            deg=torch.tensor(config["deg"], dtype=torch.long),
            edge_dim=config["hidden_dim"],
            towers=config["towers"],
            pre_layers=config["pre_layers"],
            post_layers=config["post_layers"],
            divide_input=True,
        )
    if config["conv"] == "GINE":
        return gnn.GINEConv(
            nn=Sequential(
                Linear(config["hidden_dim"], config["hidden_dim"]),
                ReLU(),
                Linear(config["hidden_dim"], config["hidden_dim"]),
            ),
            train_eps=False,
            edge_dim=config["hidden_dim"],
        )
    if config["conv"] in _OUT_OF_SCOPE_CONVS:
        raise NotImplementedError(
            f"conv={config['conv']!r} is outside the MI355X hot-path scope (PNA, GINE); see SURVEY.md §2 row 2")
    raise ValueError(f"Unsupported convolution: {config['conv']}.")


def get_global_pool(config: dict):
    """Returns the global pooling layer."""
    if config["global_pool"] == "mean":
        return gnn.MeanAggregation()
    if config["global_pool"] == "max":
        return gnn.MaxAggregation()
    if config["global_pool"] == "add":
        return gnn.SumAggregation()
    raise ValueError(f"Unsupported global pooling: {config['global_pool']}.")


def create_model(config: dict[str, Any], deg: list[int]):
    """Creates a model, as specified by the config."""
    config["deg"] = deg

    if config["model"].lower() == "gnn":
        return GNNePCSAFTL(config)
    if config["model"].lower() == "habitch":
        raise NotImplementedError("HabitchNN (dense MLP baseline, reference :257-438) is out of the hot-path scope")
    raise ValueError(f"Unsupported model: {config['model']}.")
