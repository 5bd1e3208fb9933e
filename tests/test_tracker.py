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
