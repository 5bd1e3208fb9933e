"""YTVIS result records from a model's video output (the on-disk format
step after the path).

`instances_to_coco_json_video(inputs, outputs)` builds what
projects/SeqFormer/seqformer/data/ytvis_eval.py:174-211 (IDOL: idol/data/ytvis_eval.py:174-214, which also
accepts `None` for frames a track is absent from) builds: one record per (track, class) with
COCO run-length-encoded masks.  The reference calls pycocotools' `mask_util.encode`
(pycocotools is a third-party dependency, not in the reference tree and not installed here, pinned
nowhere by VNext); its published algorithm (cocoapi common/maskApi.c: rleEncode, rleToString) is
restated below: runs over the column-major mask starting with a run of zeros, each count -- from the
third on as the difference to the count two places back -- written in 5-bit groups, low group first,
bit 0x20 = "more groups follow", + 48 to land in printable ASCII.  Parity: unpinned against the library
itself (it cannot run here); tests hold hand-derived strings and encode/decode round trips.
"""
from __future__ import annotations

import numpy as np


def rle_counts(mask):
    """[h, w] boolean / 0-1 array -> run lengths over the column-major order, first run = zeros."""
    flat = np.asarray(mask, dtype=np.uint8).reshape(-1, order="F")
    if flat.size == 0:
        return np.zeros(0, dtype=np.int64)
    change = np.flatnonzero(flat[1:] != flat[:-1]) + 1
    bounds = np.concatenate(([0], change, [flat.size]))
    runs = np.diff(bounds)
    return np.concatenate(([0], runs)) if flat[0] else runs


def counts_to_string(counts):
    out = []
    for i, x in enumerate(int(c) for c in counts):
        if i > 2:
            x -= int(counts[i - 2])
        more = True
        while more:
            c = x & 0x1f
            x >>= 5
            more = (x != -1) if (c & 0x10) else (x != 0)
            if more:
                c |= 0x20
            out.append(chr(c + 48))
    return "".join(out)


def string_to_counts(s):
    counts, p = [], 0
    while p < len(s):
        x, k, more = 0, 0, True
        while more:
            c = ord(s[p]) - 48
            x |= (c & 0x1f) << (5 * k)
            more = bool(c & 0x20)
            p += 1
            k += 1
            if not more and (c & 0x10):
                x |= -1 << (5 * k)
        if len(counts) > 2:
            x += counts[-2]
        counts.append(x)
    return counts


def rle_encode(mask):
    """-> {"size": [h, w], "counts": str}  (mask_util.encode(...)[0] with counts decoded to utf-8)"""
    h, w = np.asarray(mask).shape[-2:]
    return {"size": [int(h), int(w)], "counts": counts_to_string(rle_counts(mask))}


def rle_decode(rle):
    h, w = rle["size"]
    flat = np.zeros(h * w, dtype=np.uint8)
    pos, val = 0, 0
    for c in string_to_counts(rle["counts"]):
        flat[pos:pos + c] = val
        pos += c
        val ^= 1
    return flat.reshape((h, w), order="F").astype(bool)


def instances_to_coco_json_video(inputs, outputs):
    """inputs: [{"video_id", "length", "height", "width", ...}]; outputs: a model's
    {"pred_scores", "pred_labels", "pred_masks"} -> list of YTVIS result dicts."""
    assert len(inputs) == 1, "More than one inputs are loaded for inference!"
    video_id = inputs[0]["video_id"]
    results = []
    for score, label, masks in zip(outputs["pred_scores"], outputs["pred_labels"], outputs["pred_masks"]):
        segms = []
        for m in masks:
            if m is None:
                m = np.zeros((inputs[0]["height"], inputs[0]["width"]), dtype=np.uint8)
            segms.append(rle_encode(np.asarray(m.cpu() if hasattr(m, "cpu") else m)))
        results.append({"video_id": video_id, "score": score, "category_id": label, "segmentations": segms})
    return results
