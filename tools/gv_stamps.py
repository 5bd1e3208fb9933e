"""Dump per-workgroup phase timestamps of the owner-computes grad_value kernel (dev tool)."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import MultiScaleDeformableAttention as MSDA
from vnext_amd import _lib
from tools.time_variants import make_inputs
sh, lsi, val, loc, attn, go = make_inputs("360", 300, 5, "U", torch.float32, 1)
_lib.set_kernel_variant(408)
for _ in range(3):
    MSDA.ms_deform_attn_backward(val, sh, lsi, loc, attn, go, 64, levels_packed=True)
torch.cuda.synchronize()
n = 4096 * 16
buf = (ctypes.c_ulonglong * n)()
print("rc", _lib.lib().vnx_debug_read_gv_stamps(buf, n))
a = np.array(buf[:], dtype=np.int64).reshape(-1, 16)
a = a[a[:, 0] > 0]
TICK = 100.0  # s_memtime: 100 MHz constant clock
t0 = a[:, 0].min()
real = a[:, 6] > 0
print("workgroups", len(a), "real", real.sum())
rel = (a - t0) / TICK  # s_memtime ticks at 100 MHz -> us
print("kernel span us:", (a[:, 7].max() - t0) / TICK)
d = a[real]
order = [0, 1, 2, 3, 8, 9, 10, 11, 12, 6, 7]
names = ["start->meta+barrier", "decode+slab zero+issue", "->chunk0 top", "stage+geometry+rank", "barrier1",
         "offsets+barrier2", "scatter+barrier3", "apply+barrier4", "reset->slabwrite", "slab write"]
for i, nm in enumerate(names):
    x = (d[:, order[i + 1]] - d[:, order[i]]) / TICK
    print(f"{nm:26s} median {np.median(x):9.1f}  p90 {np.percentile(x, 90):9.1f}  max {x.max():9.1f}   (x100 ticks)")
tot = (d[:, 7] - d[:, 0]) / TICK
print("per-WG total median", np.median(tot), "p90", np.percentile(tot, 90))
