import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vnext_amd.heads import dynamic_mask_with_coords
for (H, W) in ((48, 80), (92, 160)):
    n = 300
    sets = []
    for i in range(4):
        g = torch.Generator(device="cuda").manual_seed(i)
        feats = torch.randn(1, 8, H, W, device="cuda", generator=g)
        ref = torch.rand(1, n, 2, device="cuda", generator=g) * torch.tensor([W * 8.0, H * 8.0], device="cuda")
        params = 0.3 * torch.randn(1, n, 169, device="cuda", generator=g)
        sets.append((feats, ref, params))
    with torch.no_grad():
        for i in range(12):
            dynamic_mask_with_coords(*sets[i % 4], [n], 8)
    torch.cuda.synchronize()
