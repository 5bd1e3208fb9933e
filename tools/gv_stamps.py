"""Per-workgroup phase timestamps of the records-fed grad_value kernel (development tool)."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import MultiScaleDeformableAttention as MSDA
from vnext_amd import _lib
from tools.time_variants import make_inputs
B = int(sys.argv[1]) if len(sys.argv) > 1 else 5
sh, lsi, val, loc, attn, go = make_inputs("360", 300, B, "U", torch.float32, 1)
_lib.set_kernel_variant(408)
for _ in range(3):
    MSDA.ms_deform_attn_backward(val, sh, lsi, loc, attn, go, 64, levels_packed=True)
torch.cuda.synchronize()
n = 4096 * 16
buf = (ctypes.c_ulonglong * n)()
print("rc", _lib.dev_lib().vnx_debug_read_rec_stamps(buf, n))
a = np.array(buf[:], dtype=np.int64).reshape(-1, 16)
a = a[a[:, 0] > 0]
real = a[:, 12] > 0
d = a[real]
print("workgroups", len(a), "real", len(d), " (ticks; 100 ticks ~ 0.05 us at 2.1 GHz)")
order = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12]
names = ["meta+zero cnt+barrier", "decode+slab zero+prefetch issue", "->chunk0 top", "stage+record decode+rank", "barrier1 wait",
         "offsets+barrier2", "scatter+barrier3", "apply", "barrier4 wait", "chunk1 (whole)", "chunk2.. (rest)", "slab write"]
for i, nm in enumerate(names):
    x = (d[:, order[i + 1]] - d[:, order[i]])
    print(f"{nm:34s} median {np.median(x):8.0f}  p90 {np.percentile(x, 90):8.0f}  max {x.max():8.0f}")
tot = d[:, 12] - d[:, 0]
print("per-WG total median", np.median(tot), "p90", np.percentile(tot, 90), " kernel span", a[:, [0, 12]].max() - a[:, 0].min())
