"""Sweep the amdgpu_waves_per_eu hint of the three tuned MSDA kernels.

  python tools/wpe_sweep.py build      # here (no GPU): experiment libraries under tools/experiments/libs/
  python tools/wpe_sweep.py run        # on the GPU box: time each with tools/time_variants.py
"""
import os
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
LIBS = os.path.join(ROOT, "tools", "experiments", "libs")
CONFIGS = [("base", []), ("fwd2", ["VNX_FWD_WPE=2"]), ("fwd4", ["VNX_FWD_WPE=4"]), ("fwd6", ["VNX_FWD_WPE=6"]),
           ("fwd8", ["VNX_FWD_WPE=8"]), ("k1_2", ["VNX_K1_WPE=2"]), ("k1_4", ["VNX_K1_WPE=4"]),
           ("k1_6", ["VNX_K1_WPE=6"]), ("k1_8", ["VNX_K1_WPE=8"])]
# Measured on MI355X (decoder 360p / encoder 360p): forward 10.5 / 62 us for no hint, 2 and 4; 14.3 / 113 us
# at 6; 23.8 / 248 us at 8.  grad_loc kernel: no effect at any value.  Neither kernel carries a hint.

if sys.argv[1] == "build":
    from vnext_amd.build import build_hip
    for name, defs in CONFIGS:
        print(build_hip(out=os.path.join(LIBS, f"libvnext_hip_{name}.so"), defines=defs))
else:
    for name, _ in CONFIGS:
        bwd = [] if name.startswith("fwd") else ["--bwd-only"]
        if len(sys.argv) > 2 and not any(name.startswith(p) for p in sys.argv[2].split(",")):
            continue
        env = dict(os.environ, VNX_HIP_LIB=os.path.join(LIBS, f"libvnext_hip_{name}.so"))
        for args in (["--shape", "dec360", "--dist", "U", "--variants", "0"],
                     ["--shape", "enc360", "--dist", "M", "--variants", "0", "--inner", "6"]):
            out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "time_variants.py")] + args + bwd,
                                 env=env, capture_output=True, text=True).stdout
            for line in out.splitlines():
                if "variant" in line:
                    print(f"{name:6s} {args[1]:7s} {line.strip()}", flush=True)
