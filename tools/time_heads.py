"""Timing of the fused dynamic mask head at the BASELINE frame sizes (development tool)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vnext_amd.heads import dynamic_mask_with_coords
from tools.time_variants import time_graph

for name, (H, W) in (("360p", (48, 80)), ("720p", (92, 160))):
    n = 300
    sets = []
    for i in range(8):
        g = torch.Generator(device="cuda").manual_seed(i)
        feats = torch.randn(1, 8, H, W, device="cuda", generator=g)
        ref = torch.rand(1, n, 2, device="cuda", generator=g) * torch.tensor([W * 8.0, H * 8.0], device="cuda")
        params = 0.3 * torch.randn(1, n, 169, device="cuda", generator=g)
        sets.append((feats, ref, params))
    with torch.no_grad():
        fns = [(lambda s=s: dynamic_mask_with_coords(s[0], s[1], s[2], [n], 8)) for s in sets] * 3
        med, mn = time_graph(fns)
    out_bytes = n * 4 * H * W * 4
    alg = out_bytes + 8 * H * W * 4 + n * 171 * 4
    print(f"mask head {name}: n={n} out {out_bytes/1e6:.1f} MB: {med:.2f} us (min {mn:.2f})  {alg/med/1e6:.2f} TB/s "
          f"({alg/med/1e6/8:.1%} of 8 TB/s)")
    # the reference op chain on the GPU through PyTorch (MIOpen grouped convs), for scale
    import torch.nn.functional as F
    def chain(feats, ref, params):
        ys, xs = torch.meshgrid(torch.arange(H, device="cuda") * 8.0 + 4, torch.arange(W, device="cuda") * 8.0 + 4, indexing="ij")
        loc = torch.stack([xs.reshape(-1), ys.reshape(-1)], 1)
        rel = (ref.reshape(1, n, 1, 2) - loc.reshape(1, 1, H * W, 2)).permute(0, 1, 3, 2)
        x = torch.cat([rel, feats.reshape(1, 1, 8, H * W).expand(1, n, 8, H * W)], 2).reshape(1, n * 10, H, W)
        p = params[0]
        w0, w1, w2, b0, b1, b2 = torch.split(p, [80, 64, 8, 8, 8, 1], 1)
        x = F.relu(F.conv2d(x, w0.reshape(n * 8, 10, 1, 1), b0.reshape(-1), groups=n))
        x = F.relu(F.conv2d(x, w1.reshape(n * 8, 8, 1, 1), b1.reshape(-1), groups=n))
        x = F.conv2d(x, w2.reshape(n, 8, 1, 1), b2.reshape(-1), groups=n).reshape(n, 1, H, W)
        x = F.pad(x, (0, 1, 0, 1), mode="replicate")
        x = F.interpolate(x, size=(2 * H + 1, 2 * W + 1), mode="bilinear", align_corners=True)
        x = F.pad(x, (1, 0, 1, 0), mode="replicate")
        return x[:, :, :2 * H, :2 * W]
    with torch.no_grad():
        a = chain(*sets[0]); b = dynamic_mask_with_coords(*sets[0], [n], 8)
        print("   max |fused - torch op chain| =", float((a.reshape(-1) - b.reshape(-1)).abs().max()), "scale", float(a.abs().max()))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3): chain(*sets[0])
        torch.cuda.synchronize(); e0.record()
        for i in range(10): chain(*sets[i % 8])
        e1.record(); e1.synchronize()
        print(f"   same chain as PyTorch ops on this GPU: {e0.elapsed_time(e1)*100:.1f} us per frame")
