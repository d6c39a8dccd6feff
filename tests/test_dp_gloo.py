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


def _loader_worker(rank, world, port, outdir):
    """Each rank iterates two epochs of the trainer's DataLoader under an initialised process group and records the
    graph ids of every batch (the dataset items carry their id in ``para``)."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gnnepcsaft_amd.train.trainer import DataLoader
    loader = DataLoader(_id_dataset(23), batch_size=4, shuffle=True, seed=11)
    epochs = [[b.para[:, 0].long().tolist() for b in loader] for _ in range(2)]
    torch.save({"epochs": epochs, "len": len(loader)}, os.path.join(outdir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def _id_dataset(n):
    from gnnepcsaft_amd.data import synthetic_batch
    graphs = synthetic_batch(n, 1).to_data_list()
    for i, g in enumerate(graphs):
        g.para = torch.tensor([[float(i), 3.0, 200.0]])
    return graphs


@pytest.mark.timeout(300)
def test_trainer_dataloader_shards_the_epoch_across_ranks():
    """under data parallelism every rank must train on ITS share of the shuffled epoch (the
    DistributedSampler Lightning injects for the reference, train/train.py:85-88), not on identical batches: same
    shuffle on every rank, disjoint strided shares padded by wrapping, equal batch counts, effective batch W x batch_size."""
    import numpy as np
    from gnnepcsaft_amd.train.trainer import DataLoader
    world, n = 2, 23
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_loader_worker, args=(world, _free_port(), d), nprocs=world, join=True)
        outs = [torch.load(os.path.join(d, f"r{r}.pt"), weights_only=True) for r in range(world)]
    single = DataLoader(_id_dataset(n), batch_size=4, shuffle=True, seed=11)  # no process group: the whole order
    for ep in range(2):
        full = [i for b in single for i in b.para[:, 0].long().tolist()]
        assert sorted(full) == list(range(n))
        per_rank = [[i for batch in outs[r]["epochs"][ep] for i in batch] for r in range(world)]
        assert len(per_rank[0]) == len(per_rank[1]) == (n + world - 1) // world
        # rank r holds positions r, r + W, ... of the same shuffled order (padded with its first entries)
        padded = list(np.resize(np.asarray(full), len(per_rank[0]) * world))
        for r in range(world):
            assert per_rank[r] == padded[r::world]
        assert set(per_rank[0]) | set(per_rank[1]) == set(range(n))
        assert len(set(per_rank[0]) & set(per_rank[1])) <= world - 1  # only the wrap-around padding may repeat
        assert [len(b) for b in outs[0]["epochs"][ep]] == [len(b) for b in outs[1]["epochs"][ep]]
    assert outs[0]["len"] == outs[1]["len"] == 3 and len(single) == 6
    # explicit rank / world without a process group (what a launcher that shards by hand passes)
    a = DataLoader(_id_dataset(n), batch_size=4, shuffle=False, rank=1, world=4)
    assert [i for b in a for i in b.para[:, 0].long().tolist()] == [1, 5, 9, 13, 17, 21]
    with pytest.raises(ValueError):
        DataLoader(_id_dataset(3), rank=0)


def test_flat_grad_all_reduce_second_exchange_without_zero_grad():
    """the per-slice bookkeeping is reset by finish(), so a second backward + exchange without
    zero_grad() (gradient accumulation) exchanges the conv slices again instead of silently skipping them."""
    from gnnepcsaft_amd import dp
    _, _, model = _setup()
    flat = dp.FlatGradAllReduce(model)
    flat.collective = True  # exercise the bookkeeping without a process group: stand-in collective below
    calls = []

    class _Work:
        def wait(self):
            return None

    real = dist.all_reduce
    dist.all_reduce = lambda t, **kw: (calls.append(t.numel()), _Work())[1]
    try:
        for _ in range(2):
            for lo, hi in flat.layer_slices:
                flat.reduce_slice(lo, hi)
            flat.all_reduce()
    finally:
        dist.all_reduce = real
    per_step = len(flat.layer_slices) + 1
    assert len(calls) == 2 * per_step and sum(calls) == 2 * flat.flat.numel()
