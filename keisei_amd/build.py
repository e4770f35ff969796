"""Builds libkeisei_amd.so (hand-written HIP kernels for gfx950 behind a C ABI) in-tree.

    python -m keisei_amd.build [--force]

hipcc cross-compiles without a GPU; objects are cached by source mtime under
keisei_amd/csrc/_build/.  The .so is git-ignored but travels with gpurun snapshots.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

CSRC = Path(__file__).resolve().parent / "csrc"
DIAG = bool(os.environ.get("KA_DIAG"))      # diagnostic build: ablation switches compiled in (common.h ka_diag_env), separate .so
OUT = Path(__file__).resolve().parent / ("libkeisei_amd_diag.so" if DIAG else "libkeisei_amd.so")
SOURCES = ["capi.hip", "conv3x3.hip", "wgrad.hip", "board.hip", "gemm.hip", "loss.hip", "optim.hip", "gae.hip", "rollout.hip", "transformer.hip", "tower.hip", "shogi_env.hip"]
ARCH = "gfx950"
FLAGS = ["-O3", f"--offload-arch={ARCH}", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function",
         "-Wno-unknown-pragmas", "-Wno-sometimes-uninitialized"]
# conv3x3.hip: no SLP vectorisation -- v_pk_fma_f32 / v_pk_add_f32 beside MFMAs cost more than the scalar pair they replace
# (MI355X_MICROARCH.md, filler prices) and the packed temporaries spilled the masked epilogue (profiles/NOTES_r04.md)
PER_FILE = {"gae.hip": ["-ffp-contract=off"], "conv3x3.hip": ["-fno-slp-vectorize"]}


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (Path(cand).exists() or cand == "hipcc"):
            return cand
    raise RuntimeError("hipcc not found")


def _stale(obj: Path, deps) -> bool:
    if not obj.exists():
        return True
    t = obj.stat().st_mtime
    return any(Path(d).stat().st_mtime > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> Path:
    bdir = CSRC / ("_build_diag" if DIAG else "_build")
    bdir.mkdir(exist_ok=True)
    hipcc = _hipcc()
    headers = list(CSRC.glob("*.h"))
    jobs = []
    for src in SOURCES:
        obj = bdir / (src + ".o")
        if force or _stale(obj, [CSRC / src, *headers, Path(__file__)]):
            jobs.append((src, obj))

    def compile_one(job):
        src, obj = job
        cmd = [hipcc, *FLAGS, *(["-DKA_DIAG"] if DIAG else []), *PER_FILE.get(src, []), "-c", str(CSRC / src), "-o", str(obj)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        return src

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            for done in ex.map(compile_one, jobs):
                if verbose:
                    print(f"[keisei_amd.build] compiled {done}")
    objs = [str(bdir / (s + ".o")) for s in SOURCES]
    if jobs or force or _stale(OUT, objs):
        cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", str(OUT), *objs]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        if verbose:
            print(f"[keisei_amd.build] linked {OUT}")
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
