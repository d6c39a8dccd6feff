"""2-rank data-parallel step on the real HIP path: both ranks drive cuda:0 (the test box has one GPU), the exchange runs
over ``gloo`` (RCCL needs one device per rank).  What is under test: shard_by_graph -> per-rank pack/forward/backward with
weight-gradient kernels accumulating into the flat buffer on two streams -> ONE flat all-reduce -> same averaged gradient
on every rank, equal to the average of single-process per-shard gradients (per-rank BatchNorm statistics, as Lightning
DDP without SyncBatchNorm)."""
import os
import socket
import sys
import tempfile

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _setup():
    from gnnepcsaft_amd.data import calc_deg, default_config, synthetic_batch
    from gnnepcsaft_amd.train.models import create_model
    cfg = default_config(2)
    cfg.update(hidden_dim=32, towers=2, propagation_depth=2)
    batch = synthetic_batch(24, 5)
    deg = calc_deg(batch)
    torch.manual_seed(0)
    model = create_model(cfg, deg)
    return batch, model


def _worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gnnepcsaft_amd import dp, functional as Fn, ops
    from gnnepcsaft_amd.data import shard_by_graph
    dev = torch.device("cuda:0")
    batch, model = _setup()
    model.to(dev).train()
    if rank != 0:
        with torch.no_grad():
            for p in model.parameters():
                p.add_(0.5)
    dp.broadcast_parameters(model)
    flat = dp.FlatGradAllReduce(model)
    Fn.set_grad_in_place(True)
    ops.set_wgrad_side_stream(True)
    shard = shard_by_graph(batch, world, rank).to(dev)
    flat.enable_overlap(True)  # conv-layer slices leave during backward, on the weight-gradient stream
    for _ in range(2):  # second round: zero_grad resets the slice bookkeeping
        flat.zero_grad()
        shard._gnx_pack = None
        loss = model.training_step(shard, 0)
        loss.backward()
        assert len(flat._works) == len(flat.layer_slices) == 2, "one slice per conv layer must be in flight after backward"
        flat.all_reduce(async_op=True)
        flat.finish()
    torch.cuda.synchronize()
    flat.enable_overlap(False)
    logged = dp.reduce_logged(model.logged_metrics)
    torch.save({"flat": flat.flat.cpu(), "logged": logged}, os.path.join(outdir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.timeout(600)
def test_dp_two_ranks_on_hip_path(gpu_device):
    import torch.multiprocessing as mp
    from gnnepcsaft_amd.data import shard_by_graph
    world = 2
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(world, _free_port(), d), nprocs=world, join=True)
        outs = [torch.load(os.path.join(d, f"r{r}.pt"), weights_only=True) for r in range(world)]
    batch, model = _setup()
    model.to(gpu_device).train()
    grads, losses = [], []
    for r in range(world):
        model.zero_grad()
        loss = model.training_step(shard_by_graph(batch, world, r).to(gpu_device), 0)
        loss.backward()
        losses.append(float(loss))
        grads.append(torch.cat([p.grad.reshape(-1) for p in model.parameters()]).cpu())
    want = sum(grads) / world
    for r in range(world):
        assert torch.equal(outs[r]["flat"], outs[0]["flat"]), "ranks must hold the same averaged gradient"
        err = float((outs[r]["flat"] - want).abs().max()) / float(want.abs().max())
        assert err <= 1e-5, err
        assert abs(outs[r]["logged"]["train_huber"] - sum(losses) / world) <= 1e-6


@pytest.mark.timeout(600)
def test_rccl_exchange_path_with_one_rank(gpu_device):
    """The nccl (= RCCL) backend on the one GPU of this box: `bench.py --gpus 1 --force-dp` initialises the process
    group, broadcasts, runs the per-layer overlapped slices and the remainder through RCCL and scales -- every call
    the N-GPU run makes, with world size 1 (two ranks cannot share a device under RCCL).  Run as a child process: it
    owns its process group."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    for launch, expect in (("eager", "per-layer slices"), ("graph", "one call after backward")):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dp", "--steps", "3",
                            "--warmup", "2", "--batch", "512", "--no-cpu-baseline", "--launch", launch, "--overlap"], env=env,
                           capture_output=True, text=True, timeout=500)
        assert r.returncode == 0, r.stderr[-3000:]
        out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
        assert out["n_gpus"] == 1 and out["config"]["grad_allreduce_bytes"] == 2204931 * 4
        assert expect in out["config"]["grad_exchange"] and out["loss"] == out["loss"], out["config"]
        assert out["config"]["hip_graph"] == (launch == "graph")
