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


OBJ_DIR = os.path.join(LIB_DIR, "obj")
COMPILE_FLAGS = [f for f in HIPCC_FLAGS if f != "-shared"]


def _headers() -> list[str]:
    return glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(HERE, "..", "include", "*.h")) + \
        [os.path.abspath(__file__)]


def build_hip(force: bool = False, verbose: bool = False, out: str | None = None, defines=()) -> str:
    """One object per source file, compiled in parallel and cached by mtime (a one-file edit rebuilds
    in seconds), then linked.  `out` / `defines` build an experiment copy (e.g. -DVNX_FWD_WPE=4) beside
    the product library; `VNX_HIP_LIB=<path>` makes vnext_amd._lib load it (development aid)."""
    from concurrent.futures import ThreadPoolExecutor
    target = out or LIB_PATH
    if out is None and not force and not _stale():
        return LIB_PATH
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    obj_dir = OBJ_DIR if not defines else OBJ_DIR + "_" + "_".join(d.replace("=", "-") for d in defines)
    os.makedirs(obj_dir, exist_ok=True)
    hdr_time = max(os.path.getmtime(h) for h in _headers())

    def compile_one(src):
        obj = os.path.join(obj_dir, os.path.basename(src) + ".o")
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(src), hdr_time):
            return obj
        cmd = [hipcc] + COMPILE_FLAGS + [f"-D{d}" for d in defines] + ["-c", "-o", obj, src]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
        return obj
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, sources()))
    tmp = target + ".tmp"
    os.makedirs(os.path.dirname(os.path.abspath(target)), exist_ok=True)
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + objs)
    os.replace(tmp, target)
    return target


def build_kbench(verbose: bool = False) -> str:
    """tools/kbench.bin: the stand-alone timing / cross-check harness (development tool)."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    src = os.path.join(HERE, "..", "tools", "kbench.hip")
    out = os.path.join(HERE, "..", "tools", "kbench.bin")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-o", out, src, "-L" + LIB_DIR, "-lvnext_hip",
           "-Wl,-rpath,$ORIGIN/../vnext_amd/lib"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    print(build_hip(force="--force" in sys.argv, verbose=True))
    if "--kbench" in sys.argv:
        print(build_kbench(verbose=True))
