"""Build the HIP libraries (hipcc, gfx950) in-tree.

* `vnext_amd/lib/libvnext_hip.so` -- the PRODUCT: the drop-in C ABI of include/vnext_hip.h (+ the measurement aids
  of vnext_hip_debug.h).  One kernel choice per call, made from the call's sizes; no process-wide knob.
* `vnext_amd/lib/libvnext_hip_dev.so` -- the DEVELOPMENT build: the same sources compiled with -DVNX_DEV_VARIANTS,
  plus the archived kernels under tools/experiments/msda_tile/.  It also exports include/vnext_hip_dev.h
  (`vnx_set_kernel_variant`: forced kernel configurations for A/B timing, the parity tests of every kernel form, and
  timing ablations that return wrong results by construction).  tests/, tools/kbench and tools/*.py load it when
  they ask for a variant; nothing on the product path does.

`python -m vnext_amd.build` or `vnext_amd.build.build_hip()` / `build_dev()`.  hipcc cross-compiles without a GPU;
the built libraries travel to the GPU box with the source snapshot (git-ignored, not gpurun-ignored).
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
DEV_LIB_PATH = os.path.join(LIB_DIR, "libvnext_hip_dev.so")
DEV_DEFINE = "VNX_DEV_VARIANTS"
EXPERIMENT_SRC = os.path.join(HERE, "..", "tools", "experiments", "msda_tile")

HIPCC_FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
    "-munsafe-fp-atomics",  # global_atomic_add_f32/f64 instead of CAS loops
    "-Wall", "-Wno-unused-function",
]


def sources(dev: bool = False) -> list[str]:
    src = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    if dev:      # the LDS-staged forwards: measured, retired from the product build
        src += sorted(glob.glob(os.path.join(EXPERIMENT_SRC, "*.hip")))
    return src


def _stale(target: str = LIB_PATH, dev: bool = False) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    deps = sources(dev) + glob.glob(os.path.join(CSRC, "*.h")) + \
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
    dev = DEV_DEFINE in defines
    if out is None and not force and not _stale():
        return LIB_PATH
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    obj_dir = OBJ_DIR if not defines else OBJ_DIR + "_" + "_".join(d.replace("=", "-") for d in defines)
    if out is not None and os.path.dirname(os.path.abspath(out)) != os.path.abspath(LIB_DIR):
        obj_dir = os.path.join(os.path.dirname(os.path.abspath(out)), "obj")      # experiment builds keep their objects beside them (tools/ab/...)
    os.makedirs(obj_dir, exist_ok=True)
    hdr_time = max(os.path.getmtime(h) for h in _headers())

    def compile_one(src):
        obj = os.path.join(obj_dir, os.path.basename(src) + ".o")
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(src), hdr_time):
            return obj
        cmd = [hipcc] + COMPILE_FLAGS + [f"-D{d}" for d in defines] + ["-I" + CSRC, "-c", "-o", obj, src]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
        return obj
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, sources(dev)))
    tmp = target + ".tmp"
    os.makedirs(os.path.dirname(os.path.abspath(target)), exist_ok=True)
    # -Bsymbolic: the product and the development library define the same C++ symbols and kernel stubs and may be
    # loaded into one process (tests); each must bind its references to its OWN definitions, not to whichever was
    # loaded first
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,-Bsymbolic", "-o", tmp] + objs)
    os.replace(tmp, target)
    return target


def build_dev(force: bool = False, verbose: bool = False, extra_defines=(), out: str | None = None) -> str:
    """libvnext_hip_dev.so: -DVNX_DEV_VARIANTS (+ extra A/B defines), with the archived experiment kernels."""
    target = out or DEV_LIB_PATH
    if out is None and not extra_defines and not force and not _stale(DEV_LIB_PATH, dev=True):
        return DEV_LIB_PATH
    return build_hip(force=force, verbose=verbose, out=target, defines=(DEV_DEFINE,) + tuple(extra_defines))


def build_ab(name: str, defines, files, verbose: bool = False) -> str:
    """tools/ab/<name>/libvnext_hip_dev.so: the development library with `files` (base names under csrc/) recompiled with
    extra -D `defines`; every other object is the development build's own (build_dev() first).  An A/B library costs one
    or two compilations instead of a full rebuild; `LD_LIBRARY_PATH=tools/ab/<name> tools/kbench.bin ...` runs it."""
    build_dev(verbose=verbose)
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    dev_obj = OBJ_DIR + "_" + DEV_DEFINE
    out_dir = os.path.join(HERE, "..", "tools", "ab", name)
    os.makedirs(out_dir, exist_ok=True)
    objs = []
    for src in sources(dev=True):
        base = os.path.basename(src)
        if base in files:
            obj = os.path.join(out_dir, base + ".o")
            cmd = [hipcc] + COMPILE_FLAGS + [f"-D{DEV_DEFINE}"] + [f"-D{d}" for d in defines] + ["-I" + CSRC, "-c", "-o", obj, src]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            subprocess.check_call(cmd)
        else:
            obj = os.path.join(dev_obj, base + ".o")
        objs.append(obj)
    target = os.path.join(out_dir, "libvnext_hip_dev.so")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,-Bsymbolic", "-o", target] + objs)
    return target


def build_kbench(verbose: bool = False) -> str:
    """tools/kbench.bin: the stand-alone timing / cross-check harness (development tool)."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    src = os.path.join(HERE, "..", "tools", "kbench.hip")
    out = os.path.join(HERE, "..", "tools", "kbench.bin")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-o", out, src, "-L" + LIB_DIR, "-lvnext_hip_dev",
           "-Wl,-rpath,$ORIGIN/../vnext_amd/lib"]     # the development library: kbench forces kernel variants
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    print(build_hip(force="--force" in sys.argv, verbose=True))
    print(build_dev(force="--force" in sys.argv, verbose=True))
    if "--kbench" in sys.argv:
        print(build_kbench(verbose=True))
