import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vnext_amd import _lib
lib = _lib.lib()
def run(H, W, n, reps=50):
    feats = torch.randn(1, 8, H, W, device="cuda")
    ref = torch.rand(n, 2, device="cuda") * 300
    params = 0.3 * torch.randn(n, 169, device="cuda")
    idx = torch.zeros(n, dtype=torch.int32, device="cuda")
    out = torch.empty(n, 2 * H, 2 * W, device="cuda")
    def call():
        st = torch.cuda.current_stream().cuda_stream
        rc = lib.vnx_dynamic_mask_head_forward(0, feats.data_ptr(), ref.data_ptr(), params.data_ptr(), idx.data_ptr(), out.data_ptr(), 1, 8, H, W, n, 169, 8, st)
        assert rc == 0, lib.vnx_last_error()
    for _ in range(5): call()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps): call()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); e1.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    print(f"H={H} W={W} n={n}: {us:.2f} us/launch, {n*H*W/us/1e3:.2f} Gpix/s low-res, out {n*4*H*W*4/us/1e6:.2f} TB/s")
cases = [(48, 80, 300), (48, 63, 300), (48, 126, 300), (50, 126, 300), (92, 160, 300), (48, 80, 30), (48, 80, 3000)]
if len(sys.argv) > 1 and sys.argv[1] == "--n-sweep":      # is there a tail of a second generation of waves?
    cases = [(48, 80, n) for n in (136, 200, 204, 205, 250, 272, 273, 274, 290, 300, 340, 409, 410, 600)]
for (H, W, n) in cases:
    run(H, W, n)
