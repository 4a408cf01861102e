"""Build libpdekan.so (HIP kernels + C ABI) in-tree for gfx950 with hipcc.

    python -m pde_amd.build            # build if sources are newer than the .so
    python -m pde_amd.build --force

hipcc cross-compiles without a GPU, so this runs in the build container; the
resulting shared object travels to the GPU box with the source tree.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libpdekan.so")
SOURCES = ["kan_kernels.hip", "kan_mfma.hip", "kan_head.hip", "kan_sparse.hip", "kan_f16.hip", "kan_small.hip", "kan_deeponet.hip",
           "pdekan_api.hip"]
# kan_f16.hip: without the SLP vectorizer.  With it, hipcc 7.2 (-O3, gfx950) builds the 12-jet weight-gradient kernel
# into code whose expanded columns come out wrong in lanes 48..63 (first element of each packed pair), whatever the
# source-level form of the expansion; the explicit vector types in the kernels are packed by legalization regardless.
EXTRA_FLAGS = {"kan_f16.hip": ["-fno-slp-vectorize"]}
INCLUDE = os.path.join(HERE, "..", "include")
ARCH = "gfx950"


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libpdekan.so cannot be built here")
    return exe


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    # every header next to the sources and every public header: a new one cannot be forgotten
    headers = [os.path.join(d, f) for d in (CSRC, INCLUDE) for f in sorted(os.listdir(d)) if f.endswith(".h")]
    deps = [os.path.join(CSRC, s) for s in SOURCES] + headers + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB
    objs = []
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    procs = []
    for s in SOURCES:
        obj = os.path.join(HERE, "build", s.replace(".hip", ".o"))
        objs.append(obj)
        cmd = [_hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-c",
               os.path.join(CSRC, s), "-o", obj, "-I", CSRC] + EXTRA_FLAGS.get(s, [])
        if verbose:
            cmd.append("-Rpass-analysis=kernel-resource-usage")
            print(" ".join(cmd))
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {s}:\n{out}")
        if verbose:
            print(out)
    cmd = [_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB] + objs
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}")
    return LIB


if __name__ == "__main__":
    path = build(force="--force" in sys.argv, verbose="-v" in sys.argv)
    print(path)
