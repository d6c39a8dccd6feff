#!/usr/bin/env python
"""bench.py — graphs/s of the GNN forward+backward hot path on N MI355X (BASELINE.json metric).

One step = zero_grad -> pack (CSR of the resident int64 batch) -> forward -> APE-Huber loss -> backward
(-> gradient all-reduce when N>1) over one batch of synthetic molecular graphs already resident in HBM
(optimizer step, data generation and H2D copies excluded).

Workload at N=1: BASELINE.json configs[1] — PNA, hidden=128, L=6, pre=2, post=4, T=1, batch=4096 synthetic molecules of
20 atoms / 40 directed bonds.  N>1: the same per-GPU batch on every rank (weak scaling), graphs independent per rank,
one exchange step (flat-buffer gradient all-reduce over RCCL).

Launch: ``python bench.py --gpus N`` starts N rank processes itself (one per GPU, RCCL = torch.distributed "nccl") when
it was not started under torchrun (no WORLD_SIZE in the environment); under
``python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`` it is one of the ranks.  Either way N must
equal the number of ranks, N devices must be visible, and ``n_gpus`` in the JSON is the number of ranks that joined the
timing barrier.  ``--dry-run`` rehearses the launch / barrier / exchange protocol on the CPU (gloo, no kernels).

Prints ONE JSON line on rank 0 (see the contract in the task statement) with two extra objects:
  roofline     — the scatter-aggregate: algorithmic bytes per launch (4EH + 4E + 16NH) / the average
                 duration of the kernel that performs it in the step, timed live with HIP events attached to the kernel's
                 dispatch on the launch stream (hipExtLaunchKernelGGL start / stop events: the kernel's own execution
                 time, the quantity rocprofv3's kernel trace reports) during the instrumented steps, vs 8 TB/s HBM.  Since
                 round 3 that kernel is the FUSED edge kernel (gnx_pna_edge_fwd: gather + pre-layer-1 product + aggregate);
                 ``roofline.fused`` adds its own algorithmic bytes, ``roofline.standalone_scatter_kernel`` the stand-alone
                 gnx_pna_aggregate_fwd on the same CSR;
  cpu_baseline — the oracle (pure-torch restatement of the reference's PyG CPU path, kind "port") timed on this
                 box's host cores on a bounded sample of the same workload (rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TF = 157.3     # MI355X_MICROARCH.md: fp32 matrix peak
MFMA_BF16_PEAK_TF = 2500.0   # dense bf16 matrix peak (256 CUs x 4 SIMDs x 1024 FLOP/clk x 2.4 GHz)


def ref_flops_per_graph(cfg, n=20, e=40):
    """Reference-algorithm dense FLOPs per graph, forward; fwd+bwd = 3x."""
    H, T, L, P = cfg["hidden_dim"], cfg["towers"], cfg["propagation_depth"], cfg["num_para"]
    F = H // T
    if cfg["conv"] == "PNA":
        f_edge = 2 * H * F + T * (6 * F * F + 2 * (cfg["pre_layers"] - 1) * F * F)
        f_node = T * (26 * F * F + 2 * (cfg["post_layers"] - 1) * F * F) + 2 * H * H
    else:
        f_edge, f_node = 2 * H * H, 4 * H * H
    f_read = 2 * (H * H // 2 + (H // 2) * (H // 4) + (H // 4) * P)
    return L * (e * f_edge + n * f_node) + f_read


def _free_port() -> int:
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(args) -> int:
    """Parent of a ``--gpus N`` run that was not started under torchrun: starts N fresh rank processes (this process
    never touches the GPU: ``torch.cuda.device_count()`` does not initialise it, and nothing is exec'ed from a process
    that did) with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, forwards rank 0's stdout (the JSON line), and
    returns non-zero if any rank fails."""
    import subprocess
    n = args.gpus
    if not args.dry_run:
        have = torch.cuda.device_count()
        if have < n:
            print(f"[bench] --gpus {n} but only {have} HIP device(s) visible", file=sys.stderr, flush=True)
            return 2
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    try:
        pending = dict(enumerate(procs))
        while pending:
            for r, p in list(pending.items()):
                code = p.poll()
                if code is None:
                    continue
                del pending[r]
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 1
                    print(f"[bench] rank {r} exited with {code}; stopping the other ranks", file=sys.stderr, flush=True)
                    for q in pending.values():
                        q.terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


def dry_run(args, world: int, rank: int) -> None:
    """The launch / barrier / timing / exchange protocol without a GPU: gloo ranks, a flat buffer of cfg-2's gradient
    size standing in for the step.  Checks that N ranks really joined (tests/test_host_cpu.py runs it at N = 2)."""
    import torch.distributed as dist
    if world > 1 or args.force_dp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    flat = torch.full((2204931,), float(rank + 1))
    joined = torch.ones(1)

    def step():
        if world > 1:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
            flat.mul_(1.0 / world)

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if world > 1:
        dist.barrier()
        dist.all_reduce(joined, op=dist.ReduceOp.SUM)
    elapsed = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({"metric": "molecular graphs/sec (fwd+bwd)", "value": 0.0, "unit": "graphs/s",
                          "n_gpus": int(joined.item()), "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": float(elapsed) / max(args.steps, 1) * 1e3, "higher_is_better": True,
                          "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "none",
                          "config": {"workload": "dry run: launch + barrier + gradient exchange protocol only (gloo, CPU)",
                                     "parallelism": f"dp{world}"}, "dry_run": True}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", type=int, default=2, help="BASELINE.json config index (1-5)")
    ap.add_argument("--batch", type=int, default=0, help="graphs per GPU (default: the config's batch / gpus rule)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=10, help="(kept for compatibility; the baseline protocol is fixed)")
    ap.add_argument("--no-side-stream", action="store_true", help="keep weight-gradient kernels on the main stream")
    ap.add_argument("--no-degree-classes", action="store_true", help="PNA post-layer 0 as the 13F-wide 4-segment product")
    ap.add_argument("--launch", choices=["auto", "eager", "graph"], default="auto",
                    help="eager: ~360 kernel launches per step from Python; graph: the step captured once into a HIP graph "
                         "and replayed (immune to a slow or contended host CPU, ~7 %% slower than eager when the host "
                         "keeps up: replay serialises the streams); auto (default): both are timed on a few untimed "
                         "probe steps after the warm-up and the faster one runs the timed region")
    ap.add_argument("--graph", action="store_true", help="same as --launch graph")
    ap.add_argument("--dry-run", action="store_true",
                    help="rehearse launch + barrier + exchange on the CPU (gloo); no kernels, value 0")
    ap.add_argument("--force-dp", action="store_true",
                    help="run the RCCL exchange path even with one rank (rehearsal on a one-GPU box)")
    ap.add_argument("--overlap", action="store_true",
                    help="hand every conv layer's slice of the gradient buffer to RCCL during backward (opt-in: the overlapped "
                         "exchange has never run with more than one RCCL rank -- no multi-GPU box so far; default = ONE "
                         "all-reduce after backward)")
    ap.add_argument("--no-overlap", action="store_true", help="(default since round 3; kept for compatibility)")
    args = ap.parse_args()
    if args.graph:
        args.launch = "graph"

    if args.gpus < 1:
        print("[bench] --gpus must be >= 1", file=sys.stderr)
        sys.exit(2)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))  # this process stays GPU-free; the ranks are fresh children
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"[bench] --gpus {args.gpus} but WORLD_SIZE={world}: start one rank per GPU "
              f"(python bench.py --gpus N, or torchrun --nproc-per-node N bench.py --gpus N)", file=sys.stderr, flush=True)
        sys.exit(2)
    if args.dry_run:
        dry_run(args, world, rank)
        return
    if world > 1 or args.force_dp:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if torch.cuda.device_count() <= local_rank:
            print(f"[bench] rank {rank}: LOCAL_RANK {local_rank} but {torch.cuda.device_count()} device(s) visible",
                  file=sys.stderr, flush=True)
            sys.exit(2)
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    else:
        dist = None
    assert torch.cuda.is_available(), "bench.py needs a HIP device (there is no CPU fallback for the product path)"
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    # everything (warm-up, capture, replay) runs on ONE non-default stream: autograd's AccumulateGrad nodes remember the
    # stream they were created on, and a node bound to the legacy default stream cannot take part in a graph capture
    main_stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(main_stream)

    from gnnepcsaft_amd import _lib, dp, functional as Fn, ops
    from gnnepcsaft_amd.data import calc_deg, default_config, synthetic_batch
    from gnnepcsaft_amd.train.models import create_model

    cfg = default_config(args.config)
    per_gpu = args.batch or {1: 32, 2: 4096, 3: 16384, 4: 131072 // 8, 5: 65536 // 8}[args.config]
    gen_cfg = 5 if args.config == 5 else args.config
    # weak scaling: every rank gets its own batch of the same size (different seed), graphs are independent
    from gnnepcsaft_amd.data.synthetic import BASE_SEED
    batch_cpu = synthetic_batch(per_gpu, gen_cfg, seed=BASE_SEED + args.config + 1000 * rank)
    deg = calc_deg(synthetic_batch(min(per_gpu, 4096), gen_cfg))  # same histogram source on every rank
    torch.manual_seed(0)
    model = create_model(cfg, deg).to(dev)
    model.train()
    model.model.validate_inputs = False  # range flag is read back once per step below (no sync inside the step)
    dp.broadcast_parameters(model)
    flat = dp.FlatGradAllReduce(model, force_collective=args.force_dp)
    Fn.set_grad_in_place(True)  # weight-gradient kernels accumulate straight into the flat all-reduce buffer
    ops.set_wgrad_side_stream(not args.no_side_stream)  # wgrad kernels overlap the dgrad chain on a second stream
    Fn.set_degree_classes(not args.no_degree_classes)
    ops.set_wgrad_batching(os.environ.get("GNX_WGRAD_BATCH", "1") != "0")
    b = batch_cpu.to(dev)
    N_nodes, E_edges = b.x.size(0), b.edge_index.size(1)
    H, T = cfg["hidden_dim"], cfg["towers"]

    # sync-free packing: PNA's own in-degree histogram bounds the degree (a batch above it trips the range flag)
    model.model.max_degree_hint = len(deg) - 1

    def step_body():
        flat.zero_grad()
        b._gnx_pack = None  # a new batch arrives every step in training: the packer is part of the step
        loss = model.training_step(b, 0)
        loss.backward()
        return loss

    # gradient exchange: every conv layer's slice of the flat buffer is handed to RCCL as soon as that layer's
    # weight-gradient launches are issued (overlaps the rest of backward); the remainder follows after backward;
    # finish() waits stream-wise and turns the sum into the average with one gnx_scale launch
    flat.enable_overlap(args.overlap and not args.no_overlap and args.launch != "graph")

    def eager_step():
        loss = step_body()
        flat.all_reduce(async_op=True)
        flat.finish()
        return loss

    for _ in range(args.warmup):
        loss = eager_step()
    ops.check_range(dev)  # validates the integer inputs of the warm-up steps (sync)
    torch.cuda.synchronize()

    # Launch mode.  Eager = every kernel launched from Python (the host must stay ahead of the GPU: ~6 ms of host time
    # per step against ~8 ms of GPU time at cfg-2); graph = the whole step (pack + forward + loss + backward, all
    # streams) captured once into a HIP graph and replayed, the batch living in static device buffers and the gradient
    # exchange staying outside.  On a fast host eager wins by ~7 % (replay serialises the streams); on a slow or
    # contended host the graph wins.  "auto" measures both on untimed probe steps and keeps the faster one -- every
    # rank takes the same decision (max over ranks), because the exchange pattern differs between the modes.
    step = eager_step
    graphed = False
    autotune = None
    if args.launch in ("graph", "auto"):
        overlap_was = flat._overlap
        graph_step = None
        try:
            flat.enable_overlap(False)  # collectives cannot be captured: one exchange after the replay instead
            flat.finish()
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            # thread_local: with a process group alive, ProcessGroupNCCL's watchdog thread polls its events at any time; in
            # the default (global) mode such a call from ANOTHER thread during the capture aborts the process
            # (hipErrorStreamCaptureUnsupported -- seen once in tests/test_dp_gpu.py)
            with torch.cuda.graph(graph, stream=main_stream, capture_error_mode="thread_local"):
                static_loss = step_body()

            def graph_step():
                graph.replay()
                flat.all_reduce()
                return static_loss
        except Exception as e:  # pylint: disable=broad-except
            print(f"[bench] HIP-graph capture failed ({type(e).__name__}: {e}); running eagerly", file=sys.stderr, flush=True)
            graph_step = None
        torch.cuda.synchronize()
        captured = graph_step is not None
        if dist is not None:  # the exchange pattern differs between the modes: a rank that could not capture takes all with it
            ok = torch.tensor([1.0 if captured else 0.0], dtype=torch.float64, device=dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            captured = bool(ok.item() > 0.5)
        use_graph = captured
        if captured:
            for _ in range(2):
                graph_step()
            torch.cuda.synchronize()
            if args.launch == "auto":
                def probe(fn, n=8):
                    for _ in range(2):  # untimed: after the capture the allocator serves eager steps from fresh blocks
                        fn()
                    torch.cuda.synchronize()
                    t_ = time.perf_counter()
                    for _ in range(n):
                        fn()
                    torch.cuda.synchronize()
                    return (time.perf_counter() - t_) / n * 1e3

                t_graph = probe(graph_step)
                flat.enable_overlap(overlap_was)
                t_eager = probe(eager_step)
                if dist is not None:
                    tt = torch.tensor([t_eager, t_graph], dtype=torch.float64, device=dev)
                    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                    t_eager, t_graph = float(tt[0]), float(tt[1])
                use_graph = t_graph < t_eager
                autotune = {"probe_steps": 8, "eager_ms": t_eager, "graph_ms": t_graph}
        if use_graph:
            flat.enable_overlap(False)
            step, graphed = graph_step, True
        else:
            flat.enable_overlap(overlap_was)
            step = eager_step

    agg_k = _lib.K_PNA_AGG_FWD if cfg["conv"] == "PNA" else _lib.K_GINE_AGG_FWD
    agg_bk = _lib.K_PNA_AGG_BWD if cfg["conv"] == "PNA" else _lib.K_GINE_AGG_BWD
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    ops.check_range(dev)
    # Per-kernel durations: HIP event pairs around every launch group of every kernel group, on the stream the launches
    # go to (gnx_prof_*), over `steps` further eager steps of the same workload (event records cannot live inside a
    # replayed graph).  Two passes: (1) AS RUN -- weight-gradient kernels on their own stream, overlapping the
    # input-gradient chain; (2) ISOLATED -- one stream, every kernel alone on the device.  The difference is contention
    # between the two streams (HBM, matrix cores), not the kernel itself; both are reported.
    all_groups = list(_lib.KERNEL_GROUPS)

    def instrumented(side_stream: bool):
        ops.set_wgrad_side_stream(side_stream)
        ops.prof_begin(dev, all_groups)
        for _ in range(args.steps):
            eager_step()
        torch.cuda.synchronize()
        out_ = {k: ops.prof_read_work(dev, k) for k in all_groups}
        ops.prof_end(dev)
        return out_

    prof_run = instrumented(not args.no_side_stream)
    prof_iso = instrumented(False) if not args.no_side_stream else prof_run
    ops.set_wgrad_side_stream(not args.no_side_stream)
    large = scatter_past_l3(dev, cfg) if (rank == 0 and cfg["conv"] == "PNA") else None
    # the step's PNA layers run the scatter-aggregate inside the fused edge kernel (gnx_pna_edge_fwd) when eligible; the
    # stand-alone kernel (gnx_pna_aggregate_fwd: hub-heavy batches, pre_layers != 2) is then timed on this batch's CSR
    fused_fwd = cfg["conv"] == "PNA" and prof_run[_lib.K_PNA_EDGE_FWD]["launches"] > 0
    standalone = scatter_standalone(dev, cfg, b, model) if (rank == 0 and fused_fwd) else None
    joined = 1
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
        j = torch.ones(1, dtype=torch.float64, device=dev)
        dist.all_reduce(j, op=dist.ReduceOp.SUM)
        joined = int(j.item())  # ranks that took part in the timed region
    loss_val = float(loss.detach())
    if rank == 0:
        print(f"[bench] gpu: {args.steps} steps in {elapsed:.3f}s, loss {loss_val:.6f}", file=sys.stderr, flush=True)
    assert loss_val == loss_val, "loss is NaN"

    if rank == 0:
        assert joined == world, f"{joined} ranks joined the barrier, expected {world}"
        graphs = per_gpu * world * args.steps
        value = graphs / elapsed
        def per_launch(p, k):
            n = max(p[k]["launches"], 1)
            return {"launches": p[k]["launches"], "avg_us": p[k]["ms"] / n * 1e3, "bytes": p[k]["bytes"] / n,
                    "gbs": (p[k]["bytes"] / (p[k]["ms"] * 1e-3) / 1e9) if p[k]["ms"] > 0 else 0.0}

        f_run, f_iso = per_launch(prof_run, agg_k), per_launch(prof_iso, agg_k)
        roof_kernel = "k_pna_agg_fwd" if cfg["conv"] == "PNA" else "k_gine_fwd"
        fused_info = None
        if fused_fwd:
            # a fused gather -> MLP -> aggregate kernel is reported against the SAME algorithmic figure as
            # the scatter-aggregate alone (4EH + 4E + 16NH per launch) and, additionally, against its own bytes
            fr, fi = per_launch(prof_run, _lib.K_PNA_EDGE_FWD), per_launch(prof_iso, _lib.K_PNA_EDGE_FWD)
            scatter_bytes = 4.0 * E_edges * H + 4.0 * E_edges + 16.0 * N_nodes * H
            fused_info = {"own_alg_bytes_per_launch": fr["bytes"], "own_achieved": fr["gbs"],
                          "own_frac": fr["gbs"] / HBM_PEAK_GBS,
                          "what": "message assembly (gather of P, Q, Te rows) + pre-layer-1 split-bf16 product + "
                                  "mean|min|max|std in one launch; writes h1, the messages and the aggregate once"}
            for d_ in (fr, fi):
                d_["bytes"] = scatter_bytes
                d_["gbs"] = scatter_bytes / (d_["avg_us"] * 1e-6) / 1e9 if d_["avg_us"] > 0 else 0.0
            f_run, f_iso = fr, fi
            roof_kernel = "k_pna_edge_fwd (fused: scatter-aggregate inside the edge kernel)"
        b_run, b_iso = per_launch(prof_run, agg_bk), per_launch(prof_iso, agg_bk)
        # HBM traffic of the scatter kernel from the PMC counters: only if the committed counter file was collected on
        # THIS workload (same conv, atoms, bonds, width); otherwise null
        traffic, traffic_src = None, None
        pmc = os.path.join(ROOT, "profiles", "pmc_scatter.json")
        if os.path.exists(pmc):
            try:
                pj = json.load(open(pmc))
                if (pj.get("conv", "PNA") == cfg["conv"] and pj.get("N") == N_nodes and pj.get("E") == E_edges
                        and pj.get("H") == H and pj.get("kernel", "k_pna_agg_fwd").startswith(roof_kernel.split(" ")[0])):
                    traffic = pj.get("traffic_bytes_per_launch")
                    # PMC counters need rocprofv3: the figure is NOT measured in this run, it is read from the committed file
                    traffic_src = f"profiles/pmc_scatter.json ({pj.get('round', 'r?')}, rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes)"
            except Exception:  # pylint: disable=broad-except
                traffic = None
        # the step's products: algorithmic FLOPs executed (the restructured layers do fewer than the reference
        # formulation), and the bf16-MFMA FLOPs the split-operand kernels issue for them (6 per algorithmic one)
        gemm_groups = (_lib.K_GEMM_WS, _lib.K_GEMM_TILED, _lib.K_GEMM_SMALL, _lib.K_GEMM_WGRAD, _lib.K_GEMM_WGRAD_BATCHED)
        exec_flops = sum(prof_run[k]["flops"] for k in gemm_groups) / args.steps
        mfma_flops = sum(prof_run[k]["mfma_bf16_flops"] for k in gemm_groups) / args.steps
        gemm_ms = sum(prof_iso[k]["ms"] for k in gemm_groups) / args.steps
        step_s = elapsed / args.steps
        ref_flops = 3 * ref_flops_per_graph(cfg) if args.config != 5 else None
        kernels = []
        for k, name in _lib.KERNEL_GROUPS.items():
            r_, i_ = prof_run[k], prof_iso[k]
            if r_["launches"] == 0:
                continue
            t_ = i_["ms"] * 1e-3
            e = {"group": name, "launches_per_step": r_["launches"] / args.steps, "ms_per_step": r_["ms"] / args.steps,
                 "ms_per_step_isolated": i_["ms"] / args.steps, "avg_us_isolated": i_["ms"] / max(i_["launches"], 1) * 1e3,
                 "alg_gbs_isolated": i_["bytes"] / t_ / 1e9 if t_ > 0 else 0.0,
                 "frac_hbm_isolated": i_["bytes"] / t_ / 1e9 / HBM_PEAK_GBS if t_ > 0 else 0.0}
            if i_["mfma_bf16_flops"] > 0:
                e["bf16_mfma_tflops_isolated"] = i_["mfma_bf16_flops"] / t_ / 1e12
                e["frac_bf16_mfma_peak_isolated"] = e["bf16_mfma_tflops_isolated"] / MFMA_BF16_PEAK_TF
            kernels.append(e)
        kernels.sort(key=lambda e: -e["ms_per_step"])
        out = {
            "metric": "molecular graphs/sec (fwd+bwd)", "value": value, "unit": "graphs/s", "n_gpus": joined,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"BASELINE configs[{args.config - 1}]: {cfg['conv']} hidden={H} L={cfg['propagation_depth']} "
                                   f"towers={T} pre={cfg['pre_layers']} post={cfg['post_layers']}, {per_gpu} graphs/GPU "
                                   f"({N_nodes} atoms, {E_edges} directed bonds), fwd+loss+bwd incl. CSR packing",
                       "graphs_per_gpu": per_gpu, "parallelism": f"dp{world}", "hip_graph": graphed,
                       "launch": args.launch, "launch_autotune": autotune,
                       "grad_allreduce_bytes": flat.nbytes if flat.collective else 0,
                       "grad_exchange": ("rccl, %d per-layer slices overlapped with backward + 1" % len(flat.layer_slices)
                                         if flat._overlap else ("rccl, one call after backward" if flat.collective
                                                                else "none (1 rank)"))},
            "loss": loss_val,
            # the scatter-aggregate kernel: algorithmic bytes per launch / average launch duration (HIP events attached
            # to the dispatch on the launch stream, as run); "bwd" carries both the as-run (two streams) and the isolated duration; "large"
            # is the same forward kernel on cfg-4's per-GPU batch, whose 1.0 GB per launch does not fit the Infinity Cache
            "roofline": {"bound": "hbm", "kernel": roof_kernel, "fused": fused_info, "standalone_scatter_kernel": standalone,
                         "achieved": f_run["gbs"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": f_run["gbs"] / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_src, "alg_bytes_per_launch": f_run["bytes"],
                         "avg_us": f_run["avg_us"],
                         "timing": "HIP events attached to the dispatch (hipExtLaunchKernelGGL start/stop)",
                         "launches": f_run["launches"], "avg_us_isolated": f_iso["avg_us"],
                         "bwd": {"alg_bytes_per_launch": b_run["bytes"], "avg_us": b_run["avg_us"], "achieved": b_run["gbs"],
                                 "frac": b_run["gbs"] / HBM_PEAK_GBS, "avg_us_isolated": b_iso["avg_us"],
                                 "achieved_isolated": b_iso["gbs"], "frac_isolated": b_iso["gbs"] / HBM_PEAK_GBS},
                         "large": large},
            "mfma": {"arithmetic": "fp32 operands as three bf16 pieces, six bf16 MFMAs per product, fp32 accumulation "
                                   "(products and weight gradients with >= 4096 rows; GNX_GEMM_SPLIT=0 selects the "
                                   "exact-fp32 MFMA kernels)",
                     "ref_formulation_flops_per_graph_fwd_bwd": ref_flops,
                     "executed_flops_per_graph_fwd_bwd": exec_flops / per_gpu,
                     "executed_bf16_mfma_flops_per_graph": mfma_flops / per_gpu,
                     "bf16_mfma_tflops_over_step": mfma_flops / step_s / 1e12,
                     "frac_bf16_mfma_peak_over_step": mfma_flops / step_s / 1e12 / MFMA_BF16_PEAK_TF,
                     "frac_bf16_mfma_peak_inside_product_kernels": (mfma_flops / (gemm_ms * 1e-3) / 1e12 / MFMA_BF16_PEAK_TF)
                     if gemm_ms > 0 else None,
                     "peak_bf16_mfma_tflops": MFMA_BF16_PEAK_TF, "product_kernels_ms_per_step_isolated": gemm_ms},
            "kernels": kernels[:6],
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, deg, batch_cpu, args.cpu_steps)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def scatter_standalone(dev, cfg, b, model, launches: int = 12):
    """The stand-alone scatter-aggregate kernel (gnx_pna_aggregate_fwd) on the bench batch's own CSR with random
    messages, timed with the same dispatch-attached events: what the kernel does when a layer is not eligible for the
    fused edge kernel (and the figure earlier rounds reported as ``roofline``)."""
    from gnnepcsaft_amd import _lib, ops
    from gnnepcsaft_amd.train.models import _pack_of
    H, T = cfg["hidden_dim"], cfg["towers"]
    pack = _pack_of(b, False, model.model.max_degree_hint)
    m = torch.randn(pack.E, H, device=dev)
    for _ in range(2):
        ops.pna_aggregate_fwd(m, pack, T, H // T)
    ops.prof_begin(dev, [_lib.K_PNA_AGG_FWD])
    for _ in range(launches):
        ops.pna_aggregate_fwd(m, pack, T, H // T)
    r = ops.prof_read_work(dev, _lib.K_PNA_AGG_FWD)
    ops.prof_end(dev)
    n = max(r["launches"], 1)
    gbs = r["bytes"] / (r["ms"] * 1e-3) / 1e9 if r["ms"] > 0 else 0.0
    return {"kernel": "k_pna_agg_fwd", "launches": r["launches"], "alg_bytes_per_launch": r["bytes"] / n,
            "avg_us": r["ms"] / n * 1e3, "achieved": gbs, "frac": gbs / HBM_PEAK_GBS}


def scatter_past_l3(dev, cfg, launches: int = 12):
    """The scatter-aggregate forward kernel alone on BASELINE configs[3]'s per-GPU batch (16 384 graphs: 327 680 atoms,
    655 360 directed bonds, H = 128): 1.0 GB of algorithmic bytes per launch, four times the 256 MiB Infinity Cache, so
    the figure cannot be flattered by cache residency (cfg-2's 252 MB per launch can).  Random messages, the real CSR
    of synthetic molecules; timed with the same per-launch HIP events."""
    from gnnepcsaft_amd import _lib, ops
    from gnnepcsaft_amd.data import synthetic_batch
    H, T = cfg["hidden_dim"], cfg["towers"]
    graphs = 16384 if H <= 128 else 4096
    b = synthetic_batch(graphs, 2, seed=424242).to(dev)
    pack = ops.pack_graph(b.edge_index, b.edge_attr, b.batch, b.x.size(0), graphs)
    m = torch.randn(pack.E, H, device=dev)
    for _ in range(2):
        ops.pna_aggregate_fwd(m, pack, T, H // T)
    ops.prof_begin(dev, [_lib.K_PNA_AGG_FWD])
    for _ in range(launches):
        ops.pna_aggregate_fwd(m, pack, T, H // T)
    r = ops.prof_read_work(dev, _lib.K_PNA_AGG_FWD)
    ops.prof_end(dev)
    n = max(r["launches"], 1)
    gbs = r["bytes"] / (r["ms"] * 1e-3) / 1e9 if r["ms"] > 0 else 0.0
    return {"workload": f"{graphs} graphs ({pack.N} atoms, {pack.E} directed bonds), H={H}", "launches": r["launches"],
            "alg_bytes_per_launch": r["bytes"] / n, "avg_us": r["ms"] / n * 1e3, "achieved": gbs,
            "frac": gbs / HBM_PEAK_GBS}


def host_cores() -> int:
    """Threads this process may really use: min(affinity, cgroup CPU quota); a GPU box shares its host between
    8 GPUs (16 cores per GPU), and oversubscribing torch threads beyond the quota makes the CPU run far slower."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    quota = None
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = int(q) / int(p)
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / p
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = max(1, min(n, int(quota + 0.5)))
    elif n > 32:
        n = 16  # no quota visible on a many-core host: stay within one GPU's CPU share
    return n


def cpu_baseline(cfg, deg, batch_cpu, steps):  # pylint: disable=unused-argument
    """The oracle (pure-torch restatement of the reference's PyG CPU op sequence; PyG itself is not installable here)
    timed on the host cores: zero_grad -> forward -> APE-Huber -> backward, fp32, all cores.  BASELINE.md §3: at batch
    32 (configs[0] as written) AND at the CPU's best batch, so the GPU/CPU ratio is against the CPU's best case;
    10 timed steps at 32 and 512, 2 at 4096 (12 s each) to keep the whole baseline near 45 s."""
    import copy
    from gnnepcsaft_amd.data import synthetic_batch
    from oracle import pyg_restatement as O
    cores = host_cores()
    torch.set_num_threads(cores)
    print(f"[bench] cpu_baseline on {cores} host threads ...", file=sys.stderr, flush=True)
    c = copy.deepcopy(cfg)
    c["deg"] = deg
    gen = 5 if c["towers"] > 1 and c["hidden_dim"] >= 512 else (3 if c["conv"] == "GINE" else 2)
    results = {}
    for graphs, timed in ((32, 10), (512, 10), (4096, 2)):
        batch = batch_cpu if graphs == int(batch_cpu.num_graphs) else synthetic_batch(graphs, gen)
        torch.manual_seed(0)
        model = O.GNNePCSAFT(c)
        model.train()
        times = []
        for i in range(timed + 1):
            t0 = time.perf_counter()
            model.zero_grad()
            pred = model(batch.x, batch.edge_index, batch.edge_attr, batch.batch)
            loss = O.ape_huber_loss(pred, batch.para)
            loss.backward()
            if i > 0:
                times.append(time.perf_counter() - t0)
        times.sort()
        med = times[len(times) // 2]
        results[graphs] = {"graphs_per_s": graphs / med, "timed_steps": len(times), "median_s_per_step": med}
        print(f"[bench] cpu batch {graphs}: {graphs / med:.1f} graphs/s ({med:.3f} s/step)", file=sys.stderr, flush=True)
    best = max(results, key=lambda g: results[g]["graphs_per_s"])
    return {"value": results[best]["graphs_per_s"], "unit": "graphs/s", "cores": cores, "kind": "port",
            "best_batch": best, "by_batch": {str(g): r for g, r in results.items()},
            "sample": "median of 10 timed fwd+loss+bwd steps (1 warm-up) at batch 32 and 512, of 2 at batch 4096; value = the "
                      "best of the three; oracle = pure-torch restatement of the PyG CPU op sequence, torch threads = cores"}


if __name__ == "__main__":
    main()
