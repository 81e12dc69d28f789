"""In-tree build of the gfx950 HIP library (libvvtts_hip.so).  hipcc cross-compiles without a GPU.

The .so is git-ignored but travels with the repo snapshot to the GPU box; nothing is
pip-installed, so the loader always sees the in-tree file."""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libvvtts_hip.so")
SOURCES = ["vv_gemm", "vv_attention", "vv_elementwise", "vv_posconv", "vv_vocoder", "vv_vocoder_x3", "vv_mel", "vv_ingest", "vv_api"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-fvisibility=hidden", "-std=c++17", "-Wno-unused-value", "-Wno-unused-result",
         "-Werror=extra-tokens"]      # `#endif code;` silently drops the code (round 4: an uninitialised epilogue flag)
EXTRA_FLAGS = {}      # per-file extras (none needed at present)
LINK_FLAGS = ["-Wl,--version-script=" + os.path.join(CSRC, "vvtts.map")]      # exports = the vv_* entry points of include/vvtts.h


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _newer(target: str, deps) -> bool:
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(d) <= t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    return _build(force, verbose)


def _build(force: bool, verbose: bool) -> str:
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "vvtts.h"))
    hipcc = _hipcc()

    def one(name: str):
        src, obj = os.path.join(CSRC, name + ".hip"), os.path.join(OBJ, name + ".o")
        if not force and _newer(obj, [src] + headers):
            return obj, False
        cmd = [hipcc] + FLAGS + EXTRA_FLAGS.get(name, []) + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {name}.hip:\n{r.stdout}\n{r.stderr}")
        return obj, True

    with ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        res = list(ex.map(one, SOURCES))
    objs = [o for o, _ in res]
    if force or any(ch for _, ch in res) or not _newer(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + LINK_FLAGS + ["-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
