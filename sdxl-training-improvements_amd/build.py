"""Build libsdxlstep.so (all HIP kernels + plan engine + C ABI) for gfx950, in-tree.

    python sdxl-training-improvements_amd/build.py [--force] [--diag]

--diag builds libsdxlstep_diag.so instead: the same sources + csrc/gemm_sk.hip + csrc/attention_pl.hip with -DSDXL_DIAG (experiment knobs,
stream-K GEMM, software-pipelined attention forward, phase-plane stride-2 convolution, W = 32 three-tap weight gradient: include/sdxlstep_diag.h).  The product library has none of them.

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels to the GPU box with the snapshot.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

HERE = Path(__file__).resolve().parent
CSRC = HERE / "csrc"
OBJ = HERE / "build"
LIB = HERE / "libsdxlstep.so"
OBJ_DIAG = HERE / "build_diag"
LIB_DIAG = HERE / "libsdxlstep_diag.so"
DIAG_SOURCES = ["gemm_sk.hip", "attention_pl.hip"]
SOURCES = ["gemm.hip", "gemm256.hip", "conv_wgrad3.hip", "wgrad256.hip", "gemm_cr256.hip", "gemm_pl.hip", "attention.hip", "attention_bwd_pl.hip", "norm.hip", "elementwise.hip", "loss.hip", "optimizer.hip", "engine.hip", "capi.hip"]
HEADERS = ["common.h", "kernels.h", "gemm_tiles.h", "attn_tiles.h", "engine.h", "../../include/sdxlstep.h", "../../include/sdxlstep_diag.h"]
# -fvisibility=hidden: the dynamic symbol table holds the SDXL_API entry points of include/sdxlstep.h / sdxlstep_diag.h and nothing else
# (no C++ internals, no __device_stub__s); tests/test_host_boundary.py checks `nm -D`
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-fvisibility-inlines-hidden", "-Wall", "-Wno-unused-function", "-Wno-pass-failed"]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stale(target: Path, deps) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(Path(d).stat().st_mtime > t for d in deps)


def build(force: bool = False, verbose: bool = True, diag: bool = False) -> Path:
    objdir, libpath = (OBJ_DIAG, LIB_DIAG) if diag else (OBJ, LIB)
    sources = SOURCES + (DIAG_SOURCES if diag else [])
    flags = FLAGS + (["-DSDXL_DIAG"] if diag else [])
    objdir.mkdir(exist_ok=True)
    hipcc = _hipcc()
    hdrs = [CSRC / h for h in HEADERS]
    jobs = []
    for s in sources:
        src, obj = CSRC / s, objdir / (s + ".o")
        if force or _stale(obj, [src] + hdrs):
            jobs.append((src, obj))

    def cc(job):
        src, obj = job
        cmd = [hipcc] + flags + ["-c", str(src), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src.name}:\n{r.stdout}\n{r.stderr}")
        if r.stderr.strip() and verbose:
            print(r.stderr, file=sys.stderr)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(cc, jobs))
    objs = [objdir / (s + ".o") for s in sources]
    if force or jobs or _stale(libpath, objs + [CSRC / "exports.map"]):
        # the version script keeps everything but sdxl_* local (kernel handle objects, libstdc++ instantiations, device-stub functions)
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", f"-Wl,--version-script={CSRC / 'exports.map'}", "-o", str(libpath)] + [str(o) for o in objs]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return libpath


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, diag="--diag" in sys.argv))
