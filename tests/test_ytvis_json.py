"""COCO run-length encoding of the result masks (pycocotools' published algorithm restated) and the YTVIS record."""
import json

import numpy as np
import torch

from vnext_amd.utils.ytvis_json import (counts_to_string, instances_to_coco_json_video, rle_counts, rle_decode,
                                        rle_encode, string_to_counts)


def test_known_answers_of_the_coco_rle_format():
    # cocoapi's own documentation example: M = [0 0 1 1 1 0 1] -> counts [2 3 1 1]; [1 1 1 1 1 1 0] -> [0 6 1]
    assert rle_counts(np.array([[0, 0, 1, 1, 1, 0, 1]]).T).tolist() == [2, 3, 1, 1]
    assert rle_counts(np.array([[1, 1, 1, 1, 1, 1, 0]]).T).tolist() == [0, 6, 1]
    # strings worked out by hand from maskApi.c's rleToString (pycocotools is not installed here, so this
    # piece is NOT pinned to outputs of the library itself): 4x4 with the centre 2x2 set is, column-major,
    # 0000 0110 0110 0000 -> counts [5 2 2 2 5] -> '5' '2' '2', then differences 2-2=0 -> '0', 5-2=3 -> '3'
    m = np.zeros((4, 4), dtype=np.uint8); m[1:3, 1:3] = 1
    assert rle_encode(m) == {"size": [4, 4], "counts": "52203"}
    assert rle_encode(np.zeros((2, 3), dtype=np.uint8)) == {"size": [2, 3], "counts": "6"}
    assert rle_encode(np.ones((2, 3), dtype=np.uint8)) == {"size": [2, 3], "counts": "06"}


def test_counts_string_round_trip_including_negative_differences():
    rng = np.random.default_rng(0)
    for _ in range(50):
        counts = rng.integers(0, 5000, size=rng.integers(1, 40)).tolist()
        assert string_to_counts(counts_to_string(counts)) == counts


def test_masks_round_trip_and_video_record():
    rng = np.random.default_rng(1)
    masks = [torch.from_numpy(rng.random((37, 53)) > 0.7) for _ in range(3)] + [None]
    for m in masks[:3]:
        np.testing.assert_array_equal(rle_decode(rle_encode(m.numpy())), m.numpy())
    out = {"pred_scores": [0.9], "pred_labels": [4], "pred_masks": [masks]}
    rec = instances_to_coco_json_video([{"video_id": 7, "length": 4, "height": 37, "width": 53}], out)
    assert rec[0]["video_id"] == 7 and rec[0]["category_id"] == 4 and len(rec[0]["segmentations"]) == 4
    assert not rle_decode(rec[0]["segmentations"][3]).any()              # an absent frame is an empty mask
    json.dumps(rec)                                                        # serialisable as the evaluator writes it


def test_rle_writer_reproduces_the_pycocotools_strings_held_by_the_reference_tests():
    """tests/golden/rle_coco.json (oracle/make_golden_rle.py): the four RLE strings under /root/reference/tests
    (tests/data/test_coco_evaluation.py:24, tests/test_visualizer.py:54) are pycocotools outputs.  Decoding one and
    encoding the mask again must give the identical string -- that pins string_to_counts, rle_decode, rle_counts and
    counts_to_string (negative differences, multi-character counts, the column-major order) to the library."""
    import os
    from conftest import ROOT
    cases = json.load(open(os.path.join(ROOT, "tests", "golden", "rle_coco.json")))
    assert len(cases) == 4
    for c in cases:
        h, w = c["size"]
        counts = string_to_counts(c["counts"])
        assert sum(counts) == h * w and all(n >= 0 for n in counts)
        assert counts_to_string(counts) == c["counts"]
        mask = rle_decode({"size": c["size"], "counts": c["counts"]})
        assert mask.shape == (h, w) and mask.any() and not mask.all()
        assert rle_encode(mask) == {"size": [h, w], "counts": c["counts"]}
        # runs alternate 0 / 1 starting with 0: the set pixels are the odd runs
        assert int(mask.sum()) == sum(counts[1::2])
