#!/usr/bin/env python
"""Build lib/libw2v2.so (the C-ABI HIP library) for gfx950 with hipcc.

No torch, no cmake: one `hipcc -c` per .hip file (in parallel), one link.  The
shared object is written in-tree (git-ignored, but it travels to the GPU box
with the repo snapshot).  Rebuilds only when a source or header is newer than
the library.
"""

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB_DIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIB_DIR, "libw2v2.so")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")

ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function",
         "-ffp-contract=on"]


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found (need ROCm's hipcc to build libw2v2.so)")
    return exe


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    deps = sources() + headers + [os.path.join(INCLUDE, "w2v2.h"), __file__]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True, tuning=False):
    """tuning=True builds lib/libw2v2_tuning.so with -DW2V2_TUNING: the same sources with the tuning knobs and timing
    ablations readable from the environment (csrc/common.h: tune_int).  Only tools/ load it (W2V2_NATIVE_LIB); the
    shipping library reads no environment variable."""
    if tuning:
        return _build(os.path.join(HERE, "build_tuning"), os.path.join(LIB_DIR, "libw2v2_tuning.so"), FLAGS + ["-DW2V2_TUNING"], True, verbose)
    return _build(OBJ, LIB, FLAGS, force, verbose)


def _build(OBJ, LIB, FLAGS, force, verbose):
    if not force and LIB == globals()["LIB"] and not needs_build():
        return LIB
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIB_DIR, exist_ok=True)
    cc = hipcc()

    def compile_one(src):
        obj = os.path.join(OBJ, os.path.basename(src)[:-4] + ".o")
        cmd = [cc] + FLAGS + ["-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, sources()))
    cmd = [cc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB] + objs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    if verbose:
        print(f"built {LIB}")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, tuning="--tuning" in sys.argv)
