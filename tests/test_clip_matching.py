"""Clip-level instance tracking against the reference's Videos / Clips."""
import os
import types

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR
from vnext_amd.models import tracker as trk
from vnext_amd.models.clip_matching import Clips, Videos


def _run(g, v, device):
    n_clips, L, clen, K, h, w = (int(x) for x in g[f"v{v}.cfg"])
    video = Videos(clen, L, K, (h, w), device)
    for c in range(n_clips):
        cls = torch.from_numpy(g[f"v{v}.c{c}.cls"]).to(device)
        logits = torch.from_numpy(g[f"v{v}.c{c}.logits"]).to(device)
        res = types.SimpleNamespace(pred_classes=cls.argmax(1), scores=cls.max(1)[0], cls_probs=cls, pred_masks=logits)
        video.update(Clips(g[f"v{v}.c{c}.frames"].tolist(), res))
    out_cls, out_logits = video.get_result()
    np.testing.assert_allclose(out_cls.cpu().numpy(), g[f"v{v}.out_cls"], rtol=1e-5, atol=1e-6)
    want = g[f"v{v}.out_logits"]
    got = out_logits.cpu().numpy()
    assert got.shape == want.shape
    np.testing.assert_array_equal(np.isnan(got), np.isnan(want))      # frames no clip of a track covers
    np.testing.assert_allclose(np.nan_to_num(got), np.nan_to_num(want), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("v", [0, 1])
def test_merged_tracks_equal_reference_cpu(v, monkeypatch):
    monkeypatch.setattr(trk, "_pairwise_dot", lambda a, b: a @ b.t())
    _run(dict(np.load(os.path.join(GOLDEN_DIR, "clip_matching.npz"))), v, "cpu")


@pytest.mark.gpu
@pytest.mark.parametrize("v", [0, 1])
def test_merged_tracks_equal_reference_on_the_hip_kernels(v):
    _run(dict(np.load(os.path.join(GOLDEN_DIR, "clip_matching.npz"))), v, "cuda:0")
