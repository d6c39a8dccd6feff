"""world_size 2 and 4 ``gloo`` tests (CPU) of the data-parallel logic: shard by graph, per-rank forward/backward (per-rank
BatchNorm statistics and loss mean, as Lightning DDP without SyncBatchNorm), ONE exchange step = flat-buffer gradient
all-reduce (average), ``sync_dist`` metric mean.  The oracle stands in for the compute here (there is no GPU in this
container and no CPU fallback in the product); what is under test is ``gnnepcsaft_amd.dp`` + ``shard_by_graph``.

Parity for W ranks = W single-process runs on the W shards with averaged gradients — NOT the oracle on the full batch
(BN statistics differ)."""
import os
import socket
import sys
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _setup():
    from gnnepcsaft_amd.data import calc_deg, default_config, synthetic_batch
    from oracle import pyg_restatement as O
    cfg = default_config(2)
    cfg.update(hidden_dim=16, propagation_depth=2)
    batch = synthetic_batch(24, 5)
    cfg["deg"] = calc_deg(batch)
    torch.manual_seed(0)
    model = O.GNNePCSAFT(cfg)
    return cfg, batch, model


def _shard_grads(model, shard):
    from oracle import pyg_restatement as O
    model.zero_grad()
    model.train()
    pred = model(shard.x, shard.edge_index, shard.edge_attr, shard.batch)
    loss = O.ape_huber_loss(pred, shard.para)
    loss.backward()
    return loss.detach()


def _worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gnnepcsaft_amd import dp
    from gnnepcsaft_amd.data import shard_by_graph
    _, batch, model = _setup()
    if rank != 0:  # ranks start from different weights; the initial broadcast must fix that
        with torch.no_grad():
            for p in model.parameters():
                p.add_(1.0)
    dp.broadcast_parameters(model)
    flat = dp.FlatGradAllReduce(model)
    flat.zero_grad()
    shard = shard_by_graph(batch, world, rank)
    loss = _shard_grads_keep(model, shard)
    if world == 2:
        # the overlapped form: every conv layer's slice on its own (in backward order, as the hook issues them), then
        # the remainder, then ONE scale -- must equal the single-call form the world-4 run uses
        assert len(flat.layer_slices) == 2 and all(hi > lo for lo, hi in flat.layer_slices)
        for lo, hi in reversed(flat.layer_slices):
            flat.reduce_slice(lo, hi)
        works = flat.all_reduce(async_op=True)
        assert len(works) == 3  # 2 layer slices + everything after them (the oracle registers its convs first)
        flat.finish()
        flat.zero_grad()
        # a second step through the same object: zero_grad reset the bookkeeping
        loss = _shard_grads_keep(model, shard)
        flat.all_reduce()
    else:
        flat.all_reduce()
    logged = dp.reduce_logged({"train_huber": loss})
    torch.save({"flat": flat.flat.clone(), "logged": logged, "nbytes": flat.nbytes,
                "w0": next(model.parameters()).detach().clone()}, os.path.join(outdir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def _shard_grads_keep(model, shard):
    """like _shard_grads but accumulates into the existing (flat-buffer) .grad views"""
    from oracle import pyg_restatement as O
    model.train()
    pred = model(shard.x, shard.edge_index, shard.edge_attr, shard.batch)
    loss = O.ape_huber_loss(pred, shard.para)
    loss.backward()
    return loss.detach()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 4])
def test_dp_gloo_matches_averaged_shard_gradients(world):
    from gnnepcsaft_amd.data import shard_by_graph
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(world, _free_port(), d), nprocs=world, join=True)
        outs = [torch.load(os.path.join(d, f"r{r}.pt"), weights_only=True) for r in range(world)]
    # expected: per-shard gradients from single-process runs, averaged
    _, batch, model = _setup()
    grads, losses = [], []
    threads = torch.get_num_threads()
    torch.set_num_threads(1)  # same fp32 summation order as the workers
    try:
        for r in range(world):
            losses.append(_shard_grads(model, shard_by_graph(batch, world, r)))
            grads.append(torch.cat([p.grad.reshape(-1) for p in model.parameters()]))
    finally:
        torch.set_num_threads(threads)
    want = sum(grads) / world
    for r in range(world):
        assert float((outs[r]["flat"] - want).abs().max()) <= 1e-5 * float(want.abs().max()), r
        assert torch.equal(outs[r]["flat"], outs[0]["flat"]), "all ranks hold the same averaged gradient"
        assert abs(outs[r]["logged"]["train_huber"] - float(sum(losses) / world)) < 1e-7
        assert outs[r]["nbytes"] == want.numel() * 4
        assert torch.equal(outs[r]["w0"], outs[0]["w0"]), "broadcast_parameters made the ranks identical"
