"""IDOL's online tracker against the ids assigned by the reference's IDOL_Tracker on the same
synthetic videos (oracle/make_golden_tracker.py)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR
from vnext_amd.models import tracker as trk

ARGS = dict(init_score_thr=0.2, obj_score_thr=0.1, nms_thr_pre=0.5, nms_thr_post=0.05, addnew_score_thr=0.2,
            memo_tracklet_frames=10, memo_momentum=0.8, long_match=True, frame_weight=True,
            temporal_weight=True, memory_len=3)


def _torch_scores(embeds, memo, metric):
    """tests-only restatement of vnext_amd.heads.match_scores (tracker.py:228-244)"""
    if metric == "cosine":
        return torch.nn.functional.normalize(embeds, dim=1) @ torch.nn.functional.normalize(memo, dim=1).t()
    feats = embeds @ memo.t()
    if metric == "bisoftmax":
        return (feats.softmax(1) + feats.softmax(0)) / 2
    return feats.softmax(1)


def _run(g, v, device):
    tr = trk.IDOL_Tracker(**ARGS)
    for t in range(int(g[f"v{v}.frames"])):
        p = f"v{v}.f{t}."
        dev = lambda k: torch.from_numpy(g[p + k]).to(device)  # noqa: E731
        _, _, ids, kept = tr.match(bboxes=dev("bboxes"), labels=dev("labels"), masks=dev("masks"),
                                   track_feats=dev("embeds"), frame_id=t, indices=g[p + "indices"].tolist())
        np.testing.assert_array_equal(np.array(kept, dtype=np.int64), g[p + "kept"], err_msg=f"video {v} frame {t}")
        np.testing.assert_array_equal(ids.numpy(), g[p + "ids"], err_msg=f"video {v} frame {t}")


@pytest.mark.parametrize("v", [0, 1, 2])
def test_ids_equal_reference_cpu(v, monkeypatch):
    monkeypatch.setattr(trk, "_pairwise_dot", lambda a, b: a @ b.t())
    monkeypatch.setattr(trk, "_match_scores", _torch_scores)
    _run(dict(np.load(os.path.join(GOLDEN_DIR, "tracker_idol.npz"))), v, "cpu")


@pytest.mark.gpu
@pytest.mark.parametrize("v", [0, 1, 2])
def test_ids_equal_reference_on_the_hip_kernels(v):
    _run(dict(np.load(os.path.join(GOLDEN_DIR, "tracker_idol.npz"))), v, "cuda:0")


def test_greedy_nms_and_empty_frame():
    iou = np.array([[1, .6, .1], [.6, 1, .7], [.1, .7, 1]])
    assert trk.greedy_nms(iou, 0.5).tolist() == [True, False, True]   # 1 is suppressed, so it cannot suppress 2
    tr = trk.IDOL_Tracker(**ARGS)
    out = tr.match(torch.zeros(0, 5), torch.zeros(0, dtype=torch.long), torch.zeros(0, 1, 4, 4), torch.zeros(0, 8), 0, [])
    assert out[2].numel() == 0 and out[3] == []


# ---------------------------------------------------------------------------------------------------
# DeviceTracker: the memory bank on the device, one vnx_tracker_frame call per frame, no host copy
# (vnext_amd/csrc/tracker.hip; reference tracker.py:103-298)

def _run_device(g, v):
    tr = trk.DeviceTracker(**ARGS)
    for t in range(int(g[f"v{v}.frames"])):
        p = f"v{v}.f{t}."
        dev = lambda k: torch.from_numpy(g[p + k]).to("cuda:0")  # noqa: E731
        _, _, ids, kept = tr.match(bboxes=dev("bboxes"), labels=dev("labels"), masks=dev("masks"),
                                   track_feats=dev("embeds"), frame_id=t, indices=g[p + "indices"].tolist())
        np.testing.assert_array_equal(np.array(kept, dtype=np.int64), g[p + "kept"], err_msg=f"video {v} frame {t}")
        np.testing.assert_array_equal(ids.numpy(), g[p + "ids"], err_msg=f"video {v} frame {t}")
    return tr


@pytest.mark.gpu
@pytest.mark.parametrize("v", [0, 1, 2])
def test_device_tracker_ids_equal_reference(v):
    tr = _run_device(dict(np.load(os.path.join(GOLDEN_DIR, "tracker_idol.npz"))), v)
    created, unplaced, frames = tr.counters()
    assert unplaced == 0 and frames > 0 and created > 0


def _crowded_video(seed, frames, objects, C, h, w, clutter):
    """Many objects that come and go, duplicates, clutter; scores descending per frame (the caller's order)."""
    g = torch.Generator().manual_seed(seed)
    ident = 3.0 * torch.randn(objects, C, generator=g)
    pos = torch.rand(objects, 2, generator=g) * torch.tensor([w - 8.0, h - 6.0])
    vel = torch.randn(objects, 2, generator=g) * 0.5
    cls = torch.randint(0, 5, (objects,), generator=g)
    alive_from = torch.randint(0, max(1, frames // 2), (objects,), generator=g)
    alive_for = torch.randint(3, frames, (objects,), generator=g)
    ys, xs = torch.arange(h)[:, None].float(), torch.arange(w)[None, :].float()
    out = []
    for t in range(frames):
        rows = []
        for k in range(objects):
            if t < alive_from[k] or t >= alive_from[k] + alive_for[k] or torch.rand(1, generator=g).item() < 0.25:
                continue
            p = pos[k] + vel[k] * t

            def rect(dx=0.0):
                inside = (xs >= p[0] + dx) & (xs < p[0] + dx + 7) & (ys >= p[1]) & (ys < p[1] + 5)
                return torch.where(inside, 3.0, -3.0) + 0.5 * torch.randn(h, w, generator=g)
            score = 0.3 + 0.65 * torch.rand(1, generator=g).item()
            rows.append((score, int(cls[k]), rect(), ident[k] + 0.5 * torch.randn(C, generator=g)))
            if torch.rand(1, generator=g).item() < 0.4:
                rows.append((score * 0.85, int(cls[k]), rect(float(torch.randint(0, 3, (1,), generator=g))),
                             ident[k] + 0.7 * torch.randn(C, generator=g)))
        for _ in range(int(torch.randint(0, clutter + 1, (1,), generator=g))):
            m = -3.0 + 0.5 * torch.randn(h, w, generator=g)
            y0, x0 = int(torch.randint(0, h - 3, (1,), generator=g)), int(torch.randint(0, w - 3, (1,), generator=g))
            m[y0:y0 + 3, x0:x0 + 3] += 6.0
            rows.append((0.05 + 0.5 * torch.rand(1, generator=g).item(), 0, m, 2.0 * torch.randn(C, generator=g)))
        rows.sort(key=lambda r: -r[0])
        n = len(rows)
        if n == 0:
            out.append((torch.zeros(0, 5), torch.zeros(0, dtype=torch.long), torch.zeros(0, 1, h, w), torch.zeros(0, C)))
            continue
        bboxes = torch.cat([torch.rand(n, 4, generator=g), torch.tensor([r[0] for r in rows])[:, None]], 1)
        out.append((bboxes, torch.tensor([r[1] for r in rows]), torch.stack([r[2] for r in rows])[:, None],
                    torch.stack([r[3] for r in rows])))
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("opts", [
    dict(long_match=True, frame_weight=True, temporal_weight=True, memory_len=3),       # the model's inference setting
    dict(long_match=False, frame_weight=False, temporal_weight=False, memory_len=10),   # the class defaults
    dict(long_match=True, frame_weight=False, temporal_weight=False, memory_len=2, memo_tracklet_frames=3, capacity=64),
    dict(long_match=False, frame_weight=True, temporal_weight=False, memory_len=4, match_metric="softmax", match_score_thr=0.6),
    dict(long_match=True, frame_weight=True, temporal_weight=True, memory_len=5, match_metric="cosine", match_score_thr=0.7),
])
def test_device_tracker_makes_the_host_trackers_decisions(opts):
    """Differential: the host-side IDOL_Tracker (pinned to the reference above) and the device tracker on
    crowded videos -- 40 objects that enter and leave, duplicates, clutter, empty frames, expiring tracklets
    whose slots are reused -- must assign the same ids to every detection of every frame."""
    args = dict(ARGS, **opts)
    capacity = args.pop("capacity", 128)
    for seed in (1, 2):
        video = _crowded_video(seed, frames=30, objects=40, C=32, h=24, w=40, clutter=6)
        video[7] = (torch.zeros(0, 5), torch.zeros(0, dtype=torch.long), torch.zeros(0, 1, 24, 40), torch.zeros(0, 32))
        host, dev = trk.IDOL_Tracker(**args), trk.DeviceTracker(capacity=capacity, **args)
        most = 0
        for t, (bboxes, labels, masks, embeds) in enumerate(video):
            b, l, m, e = (x.to("cuda:0") for x in (bboxes, labels, masks, embeds))
            n = b.shape[0]
            most = max(most, n)
            _, _, ids_h, kept_h = host.match(b, l, m, e, t, list(range(n)))
            ids_d = dev.match_device(b, l, m, e, t).cpu().numpy()
            want = np.full(n, -3, dtype=np.int64)
            want[np.array(kept_h, dtype=np.int64)] = ids_h.numpy()
            np.testing.assert_array_equal(ids_d, want, err_msg=f"seed {seed} frame {t} opts {opts}")
        created, unplaced, frames = dev.counters()
        assert created == host.num_tracklets and unplaced == 0 and frames == len(video) - 1   # the empty frame is a no-op
        assert most > 30 and created > 40
        if capacity < 128:
            assert created > capacity                    # more tracklets than slots over the video: slots were reused


@pytest.mark.gpu
def test_device_tracker_frame_does_not_touch_the_host():
    """match_device enqueues and returns: any synchronising call inside it raises under sync-debug mode."""
    video = _crowded_video(3, frames=6, objects=10, C=32, h=16, w=24, clutter=2)
    dev = trk.DeviceTracker(**ARGS)
    frames = [tuple(x.to("cuda:0") for x in fr) for fr in video]
    dev.match_device(*frames[0], 0)            # allocates and resets the state
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("error")
    try:
        outs = [dev.match_device(*fr, t) for t, fr in enumerate(frames[1:], 1)]
    finally:
        torch.cuda.set_sync_debug_mode("default")
    assert all(o.is_cuda for o in outs)
    assert dev.counters()[2] == sum(1 for fr in frames if fr[0].shape[0] > 0)


@pytest.mark.gpu
def test_device_tracker_reports_a_full_memory_bank():
    """More simultaneous tracklets than slots: the frame still gets its ids, the counter says how many
    tracklets could not be remembered (the caller raises `capacity`)."""
    g = torch.Generator().manual_seed(0)
    n, C = 12, 16
    masks = torch.full((n, 1, 8, 16), -3.0)
    for i in range(n):
        masks[i, 0, i % 8, (i // 8) * 8:(i // 8) * 8 + 4] = 3.0      # disjoint masks: nothing is suppressed
    bboxes = torch.cat([torch.rand(n, 4, generator=g), torch.linspace(0.9, 0.6, n)[:, None]], 1)
    dev = trk.DeviceTracker(capacity=8, **ARGS)
    ids = dev.match_device(bboxes.cuda(), torch.zeros(n, dtype=torch.long).cuda(), masks.cuda(),
                           torch.randn(n, C, generator=g).cuda(), 0).cpu()
    assert ids.tolist() == list(range(n))
    assert dev.counters() == (n, n - 8, 1)


@pytest.mark.gpu
@pytest.mark.parametrize("n,pixels", [(1, 5), (7, 64), (33, 1000), (300, 14400)])
def test_mask_intersections_are_exact(n, pixels):
    g = torch.Generator().manual_seed(n)
    logits = torch.randn(n, pixels, generator=g)
    logits[:, ::7] = 0.0                                             # sigmoid(0) = 0.5 is not > 0.5
    b = (logits > 0).double()
    got = trk.mask_intersections(logits.cuda()).cpu().double()
    assert torch.equal(got, b @ b.t())


def test_tracker_config_is_validated_on_the_host(hip_lib):
    """vnx_tracker_state_bytes / _frame_workspace_bytes are pure host functions: sizes for a valid config, 0 and a
    message for an invalid one (no GPU needed, nothing is launched)."""
    import ctypes
    from vnext_amd import _lib
    good = dict(capacity=64, channels=256, memory_len=3, memo_tracklet_frames=10, match_metric=0, long_match=1,
                frame_weight=1, temporal_weight=1, nms_thr_pre=0.5, nms_thr_post=0.05, init_score_thr=0.2,
                addnew_score_thr=0.2, match_score_thr=0.5, memo_momentum=0.8)
    cfg = _lib.TrackerConfig(**good)
    size = hip_lib.vnx_tracker_state_bytes(ctypes.addressof(cfg))
    # ids / last_frame / exist / label / long_len + embed + memo + long_embed + long_score, each 16-byte aligned
    assert size == 64 + 5 * 64 * 4 + 2 * 64 * 256 * 4 + 64 * 3 * 256 * 4 + 64 * 3 * 4
    assert hip_lib.vnx_tracker_frame_workspace_bytes(ctypes.addressof(cfg), 20, 14400) > 20 * 14400 // 8
    assert hip_lib.vnx_mask_intersections_workspace_bytes(300, 14400) >= 300 * 225 * 8
    for bad in (dict(capacity=0), dict(capacity=4096), dict(channels=30), dict(memory_len=0), dict(memory_len=64),
                dict(match_metric=3), dict(memo_momentum=1.5), dict(memo_tracklet_frames=-1)):
        cfg = _lib.TrackerConfig(**{**good, **bad})
        assert hip_lib.vnx_tracker_state_bytes(ctypes.addressof(cfg)) == 0, bad
        assert b"config out of range" in hip_lib.vnx_last_error()


@pytest.mark.gpu
def test_tracker_frame_argument_checks():
    import ctypes
    from vnext_amd import _lib
    lib = _lib.lib()
    dev = trk.DeviceTracker(**ARGS)
    g = torch.Generator().manual_seed(0)
    n = 3
    frame = (torch.rand(n, 5, generator=g).cuda(), torch.zeros(n, dtype=torch.long).cuda(),
             torch.randn(n, 1, 8, 8, generator=g).cuda(), torch.randn(n, 16, generator=g).cuda())
    dev.match_device(*frame, 0)
    cfgp, st = dev._cfg_ptr, dev.state.data_ptr()
    ws = torch.empty(lib.vnx_tracker_frame_workspace_bytes(cfgp, 600, 64), dtype=torch.uint8, device="cuda")
    ids = torch.empty(600, dtype=torch.int64, device="cuda")
    args = lambda num, wsb: (cfgp, st, frame[2].data_ptr(), frame[3].data_ptr(), frame[0].data_ptr(),     # noqa: E731
                             frame[1].data_ptr(), num, 64, 1, ids.data_ptr(), ws.data_ptr(), wsb, 0)
    assert lib.vnx_tracker_frame(*args(513, ws.numel())) != 0 and b"up to 512 detections" in lib.vnx_last_error()
    assert lib.vnx_tracker_frame(*args(3, 16)) != 0 and b"workspace smaller" in lib.vnx_last_error()
    assert lib.vnx_tracker_frame(*args(0, 0)) == 0                                    # an empty frame is a no-op
    with pytest.raises(RuntimeError, match="no CPU implementation"):
        trk.DeviceTracker(**ARGS).match_device(*(t.cpu() for t in frame), 0)
