"""oracle/heads_oracle.py -- TEST INFRASTRUCTURE ONLY.

numpy restatements of the two per-instance heads on the hot path:

* dynamic mask head (CondInst-style): follows
  projects/SeqFormer/seqformer/models/segmentation_condInst.py:425-493
  (dynamic_mask_with_coords), :404-422 (mask_heads_forward), :614-637
  (parse_dynamic_params: split order [w0, w1, w2, b0, b1, b2]), :665-678
  (compute_locations: pixel centres x*stride + stride//2), :640-662 (aligned_bilinear).
* IDOL re-identification: cosine / dot similarity, bi-softmax association
  (projects/IDOL/idol/models/tracker.py:229-244) and the contrastive loss
  (projects/IDOL/idol/models/deformable_detr.py:418-454).

The reference has NO tests for these functions ("parity unpinned" by its own test-suite,
SURVEY.md section 8c); this file is pinned instead to outputs of the reference's function bodies
executed in the build container (oracle/make_golden_heads.py -> tests/golden/heads_*.npz).
"""
from __future__ import annotations

import numpy as np


# ------------------------------------------------------------------ dynamic mask head
def split_params(params, in_ch=8, ch=8, rel_coord=True):
    """[n, 169] -> (W0 [n,8,10], W1 [n,8,8], W2 [n,1,8], b0 [n,8], b1 [n,8], b2 [n,1])"""
    c0 = in_ch + (2 if rel_coord else 0)
    sizes = [c0 * ch, ch * ch, ch, ch, ch, 1]
    assert params.shape[1] == sum(sizes)
    parts = np.split(params, np.cumsum(sizes)[:-1], axis=1)
    n = params.shape[0]
    return (parts[0].reshape(n, ch, c0), parts[1].reshape(n, ch, ch), parts[2].reshape(n, 1, ch),
            parts[3], parts[4], parts[5])


def aligned_bilinear_x2(x):
    """[..., h, w] -> [..., 2h, 2w]: replicate-pad, align_corners bilinear to 2h+1, pad, crop."""
    def up1d(a, axis):
        a = np.moveaxis(a, axis, -1)
        n = a.shape[-1]
        out = np.empty(a.shape[:-1] + (2 * n,), dtype=a.dtype)
        out[..., 1::2] = a                                   # Y = 2k+1 -> in[k]
        out[..., 0] = a[..., 0]                              # Y = 0    -> in[0]
        out[..., 2::2] = 0.5 * (a[..., :-1] + a[..., 1:])    # Y = 2k   -> (in[k-1] + in[k]) / 2
        return np.moveaxis(out, -1, axis)
    return up1d(up1d(x, -1), -2)


def dynamic_mask_head(mask_feats, reference_points, params, num_insts, stride=8, upsample=True):
    """mask_feats [N, 8, H, W]; reference_points [sum n, 2] (x, y in image pixels);
    params [sum n, 169]; num_insts: instances per image -> logits [sum n, 2H, 2W]."""
    mask_feats = np.asarray(mask_feats)
    dt = mask_feats.dtype
    N, C, H, W = mask_feats.shape
    W0, W1, W2, b0, b1, b2 = split_params(np.asarray(params, dtype=dt), C)
    ref = np.asarray(reference_points, dtype=dt)
    xs = (np.arange(W, dtype=np.float32) * stride + stride // 2).astype(dt)
    ys = (np.arange(H, dtype=np.float32) * stride + stride // 2).astype(dt)
    outs = []
    j = 0
    for i, n in enumerate(num_insts):
        feats = mask_feats[i].reshape(C, H * W)
        for _ in range(n):
            # the reference rounds the relative coordinates to fp32 whatever the dtype
            # (`relative_coords.float()`, segmentation_condInst.py:447)
            relx = np.broadcast_to((ref[j, 0] - xs[None, :]).astype(np.float32).astype(dt), (H, W)).reshape(1, H * W)
            rely = np.broadcast_to((ref[j, 1] - ys[:, None]).astype(np.float32).astype(dt), (H, W)).reshape(1, H * W)
            x0 = np.concatenate([relx, rely, feats], 0)                      # [10, HW]
            x1 = np.maximum(W0[j] @ x0 + b0[j][:, None], 0)
            x2 = np.maximum(W1[j] @ x1 + b1[j][:, None], 0)
            y = W2[j] @ x2 + b2[j][:, None]
            outs.append(y.reshape(H, W))
            j += 1
    logits = np.stack(outs, 0) if outs else np.zeros((0, H, W), dt)
    return aligned_bilinear_x2(logits) if upsample else logits


def aligned_bilinear_x2_adjoint(g):
    """Transpose of aligned_bilinear_x2: [..., 2h, 2w] -> [..., h, w]."""
    def down1d(a, axis):
        a = np.moveaxis(a, axis, -1)
        out = a[..., 1::2].copy()                            # Y = 2k+1 <- in[k]
        out[..., 0] += a[..., 0]                             # Y = 0    <- in[0]
        out[..., :-1] += 0.5 * a[..., 2::2]                  # Y = 2k   <- (in[k-1] + in[k]) / 2
        out[..., 1:] += 0.5 * a[..., 2::2]
        return np.moveaxis(out, -1, axis)
    return down1d(down1d(g, -2), -1)


def dynamic_mask_head_backward(mask_feats, reference_points, params, num_insts, grad_out, stride=8):
    """Gradients of dynamic_mask_head(...) contracted with grad_out [sum n, 2H, 2W]:
    -> (grad_feats [N, 8, H, W], grad_ref [sum n, 2], grad_params [sum n, 169]).
    What autograd derives for the reference chain (grouped 1x1 convs + ReLU,
    segmentation_condInst.py:404-422; the `.float()` on the relative coordinates passes the
    gradient through); ReLU gradient is 0 at 0 like F.relu's."""
    mask_feats = np.asarray(mask_feats)
    dt = mask_feats.dtype
    N, C, H, W = mask_feats.shape
    params = np.asarray(params, dtype=dt)
    W0, W1, W2, b0, b1, b2 = split_params(params, C)
    ref = np.asarray(reference_points, dtype=dt)
    xs = (np.arange(W, dtype=np.float32) * stride + stride // 2).astype(dt)
    ys = (np.arange(H, dtype=np.float32) * stride + stride // 2).astype(dt)
    gl_all = aligned_bilinear_x2_adjoint(np.asarray(grad_out, dtype=dt)).reshape(-1, 1, H * W)
    gfeats = np.zeros_like(mask_feats)
    gref = np.zeros_like(ref)
    gparams = np.zeros_like(params)
    j = 0
    for i, n in enumerate(num_insts):
        feats = mask_feats[i].reshape(C, H * W)
        for _ in range(n):
            relx = np.broadcast_to((ref[j, 0] - xs[None, :]).astype(np.float32).astype(dt), (H, W)).reshape(1, H * W)
            rely = np.broadcast_to((ref[j, 1] - ys[:, None]).astype(np.float32).astype(dt), (H, W)).reshape(1, H * W)
            x0 = np.concatenate([relx, rely, feats], 0)
            x1 = np.maximum(W0[j] @ x0 + b0[j][:, None], 0)
            x2 = np.maximum(W1[j] @ x1 + b1[j][:, None], 0)
            gl = gl_all[j]                                    # [1, HW]
            g2 = (W2[j].T @ gl) * (x2 > 0)                    # [8, HW]
            g1 = (W1[j].T @ g2) * (x1 > 0)
            g0 = W0[j].T @ g1                                 # [10, HW]
            gparams[j] = np.concatenate([(g1 @ x0.T).ravel(), (g2 @ x1.T).ravel(), (gl @ x2.T).ravel(),
                                         g1.sum(1), g2.sum(1), gl.sum(1)])
            gref[j] = g0[:2].sum(1)
            gfeats[i] += g0[2:].reshape(C, H, W)
            j += 1
    return gfeats, gref, gparams


# ------------------------------------------------------------------------ reid head
def bisoftmax(sim):
    """(softmax over tracks + softmax over detections) / 2   (tracker.py:232-235)"""
    def sm(a, axis):
        e = np.exp(a - a.max(axis=axis, keepdims=True))
        return e / e.sum(axis=axis, keepdims=True)
    return 0.5 * (sm(sim, 1) + sm(sim, 0))


def similarity(a, b, cosine=False):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    if cosine:
        a = a / np.maximum(np.linalg.norm(a, axis=1, keepdims=True), 1e-12)
        b = b / np.maximum(np.linalg.norm(b, axis=1, keepdims=True), 1e-12)
    return a @ b.T
