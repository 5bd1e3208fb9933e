"""Build libvnext_hip.so (hipcc, gfx950) in-tree: vnext_amd/lib/libvnext_hip.so.

`python -m vnext_amd.build` or `vnext_amd.build.build_hip()`.  hipcc
cross-compiles without a GPU; the built library travels to the GPU box with the
source snapshot (it is git-ignored, not gpurun-ignored).
"""
from __future__ import annotations

import glob
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libvnext_hip.so")

HIPCC_FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
    "-munsafe-fp-atomics",  # global_atomic_add_f32/f64 instead of CAS loops
    "-Wall", "-Wno-unused-function",
]


def sources() -> list[str]:
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + \
        glob.glob(os.path.join(HERE, "..", "include", "*.h")) + [os.path.abspath(__file__)]
    return any(os.path.getmtime(p) > t for p in deps)


def build_hip(force: bool = False, verbose: bool = False, out: str | None = None, defines=()) -> str:
    """`out` / `defines` build an experiment copy (e.g. -DVNX_FWD_WPE=4) beside the product library;
    `VNX_HIP_LIB=<path>` makes vnext_amd._lib load it (development aid, tools/wpe_sweep.py)."""
    target = out or LIB_PATH
    if out is None and not force and not _stale():
        return LIB_PATH
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    os.makedirs(os.path.dirname(target), exist_ok=True)
    tmp = target + ".tmp"
    cmd = [hipcc] + HIPCC_FLAGS + [f"-D{d}" for d in defines] + ["-o", tmp] + sources()
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    os.replace(tmp, target)
    return target


if __name__ == "__main__":
    print(build_hip(force="--force" in sys.argv, verbose=True))
