"""Builds ste_gan_amd/libstegan_hip.so and libstegan_hip_f16.so (gfx950 code objects) in-tree (the oracle is Python: nothing
else to compile).  Same sources, same C ABI; the second is compiled with -DSTEGAN_OPERAND_F16 -- the 2-byte operand type of
the matrix-core kernels is fp16 instead of bf16 (csrc/common.h, BASELINE config 5).

hipcc cross-compiles without a GPU; the .so files travel to the GPU box with the repo snapshot.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libstegan_hip.so")
LIB_F16 = os.path.join(HERE, "libstegan_hip_f16.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function"]


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(os.path.dirname(HERE), "include", "stegan_hip.h"))
    variants = [(LIB, ".o", []), (LIB_F16, ".f16.o", ["-DSTEGAN_OPERAND_F16"])]
    jobs, links = [], []
    for lib, ext, defs in variants:
        objs = []
        for s in srcs:
            src = os.path.join(CSRC, s)
            obj = os.path.join(CSRC, s[:-4] + ext)
            objs.append(obj)
            if force or _newer(obj, [src] + hdrs):
                jobs.append([HIPCC, *FLAGS, *defs, "-c", src, "-o", obj])
        links.append((lib, objs))
    jobs.sort(key=lambda c: -os.path.getsize(c[-3]))          # the big translation units first

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed:\n{' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    for lib, objs in links:
        if force or _newer(lib, objs):
            run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs])
    return LIB


def source_hash() -> str:
    """sha256 over the kernel sources (csrc/*.hip, *.h, the C-ABI header): what a committed profile was measured on.  bench.py
    prints counter-derived fields only when the profile's hash equals the tree's."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h")))
    files.append(os.path.join(os.path.dirname(HERE), "include", "stegan_hip.h"))
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
