"""Builds ``libgnnepcsaft_hip.so`` (the C-ABI of include/gnx.h) in-tree with hipcc for gfx950.

Usage: ``python -m gnnepcsaft_amd.build [--force]``.  hipcc cross-compiles without a GPU.
"""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
INCLUDE = PKG.parent / "include"
LIB = PKG / "libgnnepcsaft_hip.so"
OBJ_DIR = PKG / "csrc" / "build"
SOURCES = ["gnx_api.hip", "gnx_pack.hip", "gnx_embed.hip", "gnx_gemm.hip", "gnx_aggregate.hip", "gnx_norm.hip", "gnx_optim.hip", "gnx_layer.hip", "gnx_fused.hip"]
# -fno-slp-vectorize: the SLP vectoriser turns the fp32 -> 3 x bf16 split into packed-f32 VALU (v_pk_add_f32), which issues
# badly beside MFMAs (tools/ubench/wave_specialised_overlap.hip: 1.41 -> 1.24 us per step without it); cfg-2 step 7.89 -> 7.77 ms
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-slp-vectorize", "-Wall",
         "-Wno-unused-function",
         f"-I{INCLUDE}", f"-I{CSRC}"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def _digest() -> str:
    h = hashlib.sha256()
    for f in sorted(list(CSRC.glob("*.hip")) + list(CSRC.glob("*.hpp")) + [INCLUDE / "gnx.h", Path(__file__)]):
        h.update(f.name.encode())
        h.update(f.read_bytes())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = False) -> Path:
    stamp = OBJ_DIR / "stamp.txt"
    dig = _digest()
    if not force and LIB.exists() and stamp.exists() and stamp.read_text() == dig:
        return LIB
    OBJ_DIR.mkdir(parents=True, exist_ok=True)
    hipcc = _hipcc()

    def compile_one(src: str) -> str:
        obj = OBJ_DIR / (src.replace(".hip", ".o"))
        cmd = [hipcc, *FLAGS, "-c", str(CSRC / src), "-o", str(obj)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        return str(obj)

    with ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(LIB), *objs], capture_output=True,
                       text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stderr}")
    stamp.write_text(dig)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
