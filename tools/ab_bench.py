"""Same-box A/B of the bench step under feature switches (box-to-box spread is ~5 %, so every optimisation is judged
inside one process): alternates the variants REPS times and prints the median ms/step of each.
usage: python tools/ab_bench.py [--config 2] [--batch 4096] [--steps 30] [--reps 4] variant ...
variants: base | nomerge | noahead | noside | nocentered ... (see VARIANTS)"""
import argparse, os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gnnepcsaft_amd import _lib, dp, functional as Fn, ops
from gnnepcsaft_amd.data import calc_deg, default_config, synthetic_batch
from gnnepcsaft_amd.train.models import create_model

VARIANTS = {
    "base": lambda dev: None,
    "nomerge": lambda dev: Fn.set_merge_last_post(False),
    "noahead": lambda dev: Fn.set_prepare_ahead(False),
    "nonative": lambda dev: Fn.set_native_layer_backward(False),
    "nobondaside": lambda dev: Fn.set_bond_chain_aside(False),
    "noside": lambda dev: ops.set_wgrad_side_stream(False),
    "nocentered": lambda dev: ops.set_option(dev, _lib.OPT_STD_BWD_CENTERED, 0),
    "nopipe": lambda dev: ops.set_option(dev, _lib.OPT_GEMM_PIPE, 0),
    "nowgpipe": lambda dev: ops.set_option(dev, _lib.OPT_WGRAD_PIPE, 0),
    "wgs1024": lambda dev: ops.set_option(dev, _lib.OPT_WGRAD_WGS, 1024),
    "wgs96": lambda dev: ops.set_option(dev, _lib.OPT_WGRAD_WGS, 96),
    "wgs128": lambda dev: ops.set_option(dev, _lib.OPT_WGRAD_WGS, 128),
    "wgs64": lambda dev: ops.set_option(dev, _lib.OPT_WGRAD_WGS, 64),
    "wgs160": lambda dev: ops.set_option(dev, _lib.OPT_WGRAD_WGS, 160),
    "wgs128rows2048": lambda dev: (ops.set_option(dev, _lib.OPT_WGRAD_WGS, 128), setattr(ops.DegreeClasses, "WGRAD_ROWS", 2048)),
    "wgs128rows512": lambda dev: (ops.set_option(dev, _lib.OPT_WGRAD_WGS, 128), setattr(ops.DegreeClasses, "WGRAD_ROWS", 512)),
    "wgs192": lambda dev: ops.set_option(dev, _lib.OPT_WGRAD_WGS, 192),
    "wgs384": lambda dev: ops.set_option(dev, _lib.OPT_WGRAD_WGS, 384),
    "rows4096": lambda dev: setattr(ops.DegreeClasses, "WGRAD_ROWS", 4096),
    "wgs256": lambda dev: ops.set_option(dev, _lib.OPT_WGRAD_WGS, 256),
    "wgs512": lambda dev: ops.set_option(dev, _lib.OPT_WGRAD_WGS, 512),
    "wgs2048": lambda dev: ops.set_option(dev, _lib.OPT_WGRAD_WGS, 2048),
    "rows256": lambda dev: setattr(ops.DegreeClasses, "WGRAD_ROWS", 256),
    "rows1024": lambda dev: setattr(ops.DegreeClasses, "WGRAD_ROWS", 1024),
    "rows2048": lambda dev: setattr(ops.DegreeClasses, "WGRAD_ROWS", 2048),
    "nobatch": lambda dev: ops.set_wgrad_batching(False),
    "nofused": lambda dev: Fn.set_fused_edge(False),
    "nobatchwo": lambda dev: Fn.set_batch_weight_only(False),
    "nofusedbwd": lambda dev: ops.set_option(dev, _lib.OPT_EDGE_FUSED, 2),
    "noas3": lambda dev: ops.set_option(dev, _lib.OPT_GEMM_AS, 0),
    "nowsfast": lambda dev: ops.set_option(dev, _lib.OPT_GEMM_WS_FAST, 0),
    "nosplitahead": lambda dev: ops.set_option(dev, _lib.OPT_SPLIT_AHEAD, 0),
    "embfp32": lambda dev: ops.set_option(dev, _lib.OPT_EMBED_BWD_MFMA, 2),
    "classearly": lambda dev: Fn.set_class_wgrad_after_agg(False),
    "flushlate": lambda dev: Fn.set_wgrad_flush_before_dx(False),
    "notail": lambda dev: Fn.set_tail_wgrad_all_cus(False),
    "tailearly": lambda dev: Fn.set_tail_wgrad_all_cus(True, True),
    "ws8waves": lambda dev: ops.set_option(dev, _lib.OPT_GEMM_WS_FAST, 1),
    "rcloop": lambda dev: ops.set_option(dev, _lib.OPT_AGG_BWD_RECOMPUTE, 2),
    "rows128": lambda dev: ops.set_option(dev, _lib.OPT_GEMM_TILE_ROWS, 128),
    "rows96": lambda dev: ops.set_option(dev, _lib.OPT_GEMM_TILE_ROWS, 96),
}


def reset(dev):
    Fn.set_merge_last_post(True); Fn.set_prepare_ahead(True); Fn.set_bond_chain_aside(True); Fn.set_native_layer_backward(True); ops.set_wgrad_side_stream(True); Fn.set_fused_edge(True); Fn.set_batch_weight_only(True); Fn.set_tail_wgrad_all_cus(True); Fn.set_class_wgrad_after_agg(True); Fn.set_wgrad_flush_before_dx(True)
    ops.set_option(dev, _lib.OPT_STD_BWD_CENTERED, 1)
    ops.set_option(dev, _lib.OPT_GEMM_PIPE, 1)
    ops.set_option(dev, _lib.OPT_WGRAD_PIPE, 1)
    ops.set_option(dev, _lib.OPT_EDGE_FUSED, 1)
    ops.set_option(dev, _lib.OPT_GEMM_AS, 1)
    ops.set_option(dev, _lib.OPT_EMBED_BWD_MFMA, 1)
    ops.set_option(dev, _lib.OPT_SPLIT_AHEAD, 1)
    ops.set_option(dev, _lib.OPT_AGG_BWD_RECOMPUTE, 1)
    ops.set_option(dev, _lib.OPT_GEMM_WS_FAST, 2)
    ops.set_option(dev, _lib.OPT_GEMM_TILE_ROWS, 0)
    ops.set_option(dev, _lib.OPT_WGRAD_WGS, 0); ops.DegreeClasses.WGRAD_ROWS = 1024; ops.set_wgrad_batching(True)
    for k, v in EXTRA_RESET.items():
        v(dev)


EXTRA_RESET = {}

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=2); ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--steps", type=int, default=30); ap.add_argument("--reps", type=int, default=4)
    ap.add_argument("variants", nargs="*", default=["base"])
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    st = torch.cuda.Stream(device=dev); torch.cuda.set_stream(st)
    cfg = default_config(a.config)
    per = a.batch or {1: 32, 2: 4096, 3: 16384, 4: 16384, 5: 8192}[a.config]
    b_cpu = synthetic_batch(per, 5 if a.config == 5 else a.config)
    deg = calc_deg(synthetic_batch(min(per, 4096), 5 if a.config == 5 else a.config))
    torch.manual_seed(0)
    model = create_model(cfg, deg).to(dev).train()
    model.model.validate_inputs = False
    model.model.max_degree_hint = len(deg) - 1
    flat = dp.FlatGradAllReduce(model)
    Fn.set_grad_in_place(True)
    b = b_cpu.to(dev)

    def run(n):
        for _ in range(n):
            flat.zero_grad(); b._gnx_pack = None
            model.training_step(b, 0).backward()

    res = {v: [] for v in a.variants}
    for rep in range(a.reps + 1):
        for v in a.variants:
            reset(dev)
            for part in v.split("+"):
                VARIANTS[part](dev)
            run(3); torch.cuda.synchronize()
            t0 = time.perf_counter(); run(a.steps); torch.cuda.synchronize()
            if rep > 0:
                res[v].append((time.perf_counter() - t0) / a.steps * 1e3)
    for v in a.variants:
        print(f"{v:24s} median {statistics.median(res[v]):.3f} ms/step   all {' '.join(f'{x:.3f}' for x in res[v])}", flush=True)
