"""IDOL's online tracker: mask NMS, embedding association, memory bank
(SURVEY.md section 8 row a7 `IDOL_Tracker.match`, and (f) rank 3).

Same constructor arguments, same `match(bboxes, labels, masks, track_feats, frame_id, indices)`
call and the same decisions as projects/IDOL/idol/models/tracker.py:50-298.  What moves:

  reference (per frame)                                    here (per frame)
  ------------------------------------------------------   ------------------------------------------
  mask NMS: O(n^2) Python loop, one mask_iou launch and    ONE [n, n] mask-IoU matrix: binarise, one
  one host sync per pair (:28-46)                          similarity launch on the matrix cores
                                                           (intersections of 0/1 rows are exact in
                                                           fp32), ONE copy to the host; the greedy
                                                           pass runs on the matrix
  second mask_iou of the unselected detections (:276,292)  rows of the same matrix
  torch.mm + 2 softmax + Python loop with `conf > thr`,    similarity + bi-softmax kernels, ONE copy;
  `id > -1` syncs per detection (:228-263)                 the greedy assignment runs on the host copy
  per-tracklet stack / weighted mean launches (:176-187)   one [tracks, rows] x [rows, C] product

Box scores, labels, ids, frame counters live on the host (they only steer control flow);
embeddings and masks never leave the device.

`DeviceTracker` (round 2) goes the rest of the way: the memory bank itself is a device blob and a frame
is one `vnx_tracker_frame` call -- mask bits + popcount intersections, the similarity against all slots,
and one workgroup for the greedy passes (vnext_amd/csrc/tracker.hip) -- with NO host copy: `match_device`
returns the frame's ids as a device tensor, and the model reads all frames' ids once per video.
`IDOL_Tracker` stays as the host-side statement of the same decisions (CPU tests, differential tests).
"""
from __future__ import annotations

import numpy as np
import torch


def _pairwise_dot(a, b):
    """[n, C] x [k, C] -> [n, k] on the HIP similarity kernel (tests swap in torch.mm on CPU)."""
    from ..heads import similarity
    with torch.no_grad():
        return similarity(a, b)


def _match_scores(embeds, memo_embeds, metric):
    from ..heads import match_scores
    return match_scores(embeds, memo_embeds, metric)


def mask_intersections(mask_logits):
    """[n, pixels] logits -> [n, n] fp32 |mask_i & mask_j| of the binarised masks (exact integers).
    On the GPU: bit words + popcount (`vnx_mask_intersections`); tests on the CPU patch `_pairwise_dot`."""
    n = mask_logits.shape[0]
    if mask_logits.is_cuda:
        from .. import _lib
        logits = mask_logits.float().contiguous()
        inter = torch.empty(n, n, dtype=torch.int32, device=logits.device)
        ws = torch.empty(_lib.lib().vnx_mask_intersections_workspace_bytes(n, logits.shape[1]), dtype=torch.uint8,
                         device=logits.device)
        with torch.cuda.device(logits.device):
            _lib.check(_lib.lib().vnx_mask_intersections(logits.data_ptr(), n, logits.shape[1], inter.data_ptr(),
                                                         ws.data_ptr(), ws.numel(), _lib.current_stream(logits)))
        return inter.float()
    b = (mask_logits > 0).float()
    pad = (-b.shape[1]) % 4
    if pad:
        b = torch.nn.functional.pad(b, (0, pad))
    return _pairwise_dot(b, b)


def mask_iou_matrix(mask_logits):
    """[n, 1, h, w] or [n, h, w] logits -> [n, n] IoU of the binarised masks (sigmoid > 0.5),
    with the reference's +1e-6 on both terms (tracker.py:17-25)."""
    n = mask_logits.shape[0]
    inter = mask_intersections(mask_logits.reshape(n, -1))
    area = inter.diagonal()
    return (inter + 1e-6) / (area[:, None] + area[None, :] - inter + 1e-6)


def greedy_nms(iou, thr):
    """keep[i] for detections already in score order of the caller (tracker.py:28-46)."""
    n = iou.shape[0]
    keep = np.ones(n, dtype=bool)
    for i in range(n - 1):
        if keep[i]:
            keep[i + 1:] &= ~(iou[i, i + 1:] > thr)
    return keep


class IDOL_Tracker(object):
    def __init__(self, nms_thr_pre=0.7, nms_thr_post=0.3, init_score_thr=0.2, addnew_score_thr=0.5,
                 obj_score_thr=0.1, match_score_thr=0.5, memo_tracklet_frames=10, memo_backdrop_frames=1,
                 memo_momentum=0.5, nms_conf_thr=0.5, nms_backdrop_iou_thr=0.5, nms_class_iou_thr=0.7,
                 with_cats=True, match_metric='bisoftmax', long_match=False, frame_weight=False,
                 temporal_weight=False, memory_len=10):
        assert 0 <= memo_momentum <= 1.0
        assert memo_tracklet_frames >= 0
        assert memo_backdrop_frames >= 0
        assert match_metric in ['bisoftmax', 'softmax', 'cosine']
        self.memory_len, self.temporal_weight, self.long_match, self.frame_weight = \
            memory_len, temporal_weight, long_match, frame_weight
        self.nms_thr_pre, self.nms_thr_post = nms_thr_pre, nms_thr_post
        self.init_score_thr, self.addnew_score_thr, self.obj_score_thr = init_score_thr, addnew_score_thr, obj_score_thr
        self.match_score_thr = match_score_thr
        self.memo_tracklet_frames, self.memo_backdrop_frames, self.memo_momentum = \
            memo_tracklet_frames, memo_backdrop_frames, memo_momentum
        self.match_metric = match_metric
        self.num_tracklets = 0
        self.tracklets = dict()     # id -> {embed, long_embed [device rows], long_score, label, last_frame, exist_frame}

    @property
    def empty(self):
        return not self.tracklets

    # ------------------------------------------------------------------ memory bank
    def _memo(self, like):
        """-> (memo_embeds [m, C] on the device, ids [m], exist_frame [m])  (tracker.py:165-205)"""
        ids = list(self.tracklets.keys())
        exist = np.array([self.tracklets[k]['exist_frame'] for k in ids], dtype=np.float64)
        if not self.long_match:
            return torch.stack([self.tracklets[k]['embed'] for k in ids]), np.array(ids), exist
        rows, weights = [], []
        for r, k in enumerate(ids):
            v = self.tracklets[k]
            w = np.array(v['long_score'], dtype=np.float32)
            if self.temporal_weight:   # torch.range(0, 1, 1/L)[1:] = 1/L, 2/L, ..., 1
                L = len(w)
                w = w + (np.arange(1, L + 1, dtype=np.float32) / np.float32(L))
            w = w / w.sum()
            for e, wi in zip(v['long_embed'], w):
                rows.append(e)
                weights.append((r, len(rows) - 1, wi))
        A = np.zeros((len(ids), len(rows)), dtype=np.float32)
        for r, c, wi in weights:
            A[r, c] = wi
        E = torch.stack(rows)
        return torch.from_numpy(A).to(E.device, E.dtype) @ E, np.array(ids), exist

    def update_memo(self, ids, scores, embeds, labels, frame_id):
        """ids [n] host int; scores [n] host float (the box score); embeds [n, C] device."""
        hit = [i for i in range(len(ids)) if ids[i] > -1 and int(ids[i]) in self.tracklets]
        if hit:   # momentum update of all continued tracklets in one pass (tracker.py:118-120)
            old = torch.stack([self.tracklets[int(ids[i])]['embed'] for i in hit])
            new = (1 - self.memo_momentum) * old + self.memo_momentum * embeds[hit]
            for r, i in enumerate(hit):
                v = self.tracklets[int(ids[i])]
                v['embed'] = new[r]
                v['long_score'].append(float(scores[i]))
                v['long_embed'].append(embeds[i])
                v['last_frame'], v['label'] = frame_id, int(labels[i])
                v['exist_frame'] += 1
        for i in range(len(ids)):
            k = int(ids[i])
            if k > -1 and k not in self.tracklets:
                self.tracklets[k] = dict(embed=embeds[i], long_embed=[embeds[i]], long_score=[float(scores[i])],
                                         label=int(labels[i]), last_frame=frame_id, exist_frame=1)
        for k in [k for k, v in self.tracklets.items() if frame_id - v['last_frame'] >= self.memo_tracklet_frames]:
            self.tracklets.pop(k)
        for v in self.tracklets.values():
            if len(v['long_embed']) > self.memory_len:
                v['long_embed'].pop(0)
            if len(v['long_score']) > self.memory_len:
                v['long_score'].pop(0)

    # ------------------------------------------------------------------------ match
    def _assign(self, scores, memo_ids, exist):
        """greedy detection -> tracklet assignment on the host copy (tracker.py:245-263)."""
        n = scores.shape[0]
        ids = np.full(n, -2, dtype=np.int64)
        for i in range(n):
            row = scores[i]
            if self.frame_weight:
                strong = (memo_ids > -1) & (row > 0.5)
                if strong.sum() > 1:   # several candidates: prefer the longer-lived tracklets
                    fw = exist[strong]
                    w = row.copy()
                    w[strong] = w[strong] * fw
                    w[~strong] = w[~strong] * fw.mean()
                    j = int(np.argmax(w))
                    conf = w[j]
                else:
                    j = int(np.argmax(row))
                    conf = row[j]
            else:
                j = int(np.argmax(row))
                conf = row[j]
            if conf > self.match_score_thr and memo_ids[j] > -1:
                ids[i] = memo_ids[j]
                scores[:i, j] = 0
                scores[i + 1:, j] = 0
        return ids

    def match(self, bboxes, labels, masks, track_feats, frame_id, indices):
        """bboxes [n, 5] (cxcywh + score), labels [n], masks [n, 1, h, w] logits, track_feats
        [n, C], indices: the query index of every detection.
        -> (bboxes, labels, ids [n'] host long tensor, indices) of the detections kept by the
        mask NMS; ids: tracklet id, -1 = backdrop, -2 = suppressed duplicate."""
        n = bboxes.shape[0]
        if n == 0:
            return bboxes, labels, torch.zeros(0, dtype=torch.long), []
        iou = mask_iou_matrix(masks)
        host = torch.cat([iou, bboxes[:, 4:5].to(iou.dtype), labels[:, None].to(iou.dtype)], 1).cpu().numpy()
        iou, score, label = host[:, :n].astype(np.float64), host[:, n], host[:, n + 1].astype(np.int64)
        keep = greedy_nms(iou, self.nms_thr_pre)
        kept = np.nonzero(keep)[0]
        kept_dev = torch.from_numpy(kept).to(bboxes.device)
        indices = [indices[i] for i in kept]
        bboxes, labels, embeds = bboxes[kept_dev], labels[kept_dev], track_feats[kept_dev]
        iou, score, label = iou[np.ix_(kept, kept)], score[kept], label[kept]
        n = len(kept)

        if not self.empty:
            memo_embeds, memo_ids, exist = self._memo(embeds)
            scores = _match_scores(embeds, memo_embeds, self.match_metric).cpu().numpy().astype(np.float64)
            ids = self._assign(scores, memo_ids, exist)
            new = (ids == -2) & (score > self.addnew_score_thr)
        else:
            ids = np.full(n, -2, dtype=np.int64)
            new = score > self.init_score_thr
        ids[new] = np.arange(self.num_tracklets, self.num_tracklets + int(new.sum()))
        self.num_tracklets += int(new.sum())
        # left-overs that overlap no earlier detection become backdrops (-1), duplicates stay -2
        for i in np.nonzero(ids == -2)[0]:
            if (iou[i, :i] < self.nms_thr_post).all():
                ids[i] = -1
        self.update_memo(ids, score, embeds, label, frame_id)
        return bboxes, labels, torch.from_numpy(ids), indices


METRICS = {"bisoftmax": 0, "softmax": 1, "cosine": 2}


class DeviceTracker(object):
    """IDOL_Tracker with the tracklets in device memory (vnext_amd/csrc/tracker.hip; reference
    projects/IDOL/idol/models/tracker.py:50-298).  Same constructor arguments plus `capacity`
    (tracklet slots alive at a time) and `channels` (embedding width, fixed by the first frame if None).

    match_device(...) -> ids [n] int64 ON THE DEVICE for every input detection: tracklet id, -1 back-drop,
                         -2 duplicate, -3 removed by the mask NMS.  No host synchronisation.
    match(...)        -> the reference's (bboxes, labels, ids, indices) of the detections the NMS kept;
                         costs the one device->host copy of `ids` that signature implies.
    """

    def __init__(self, nms_thr_pre=0.7, nms_thr_post=0.3, init_score_thr=0.2, addnew_score_thr=0.5,
                 obj_score_thr=0.1, match_score_thr=0.5, memo_tracklet_frames=10, memo_backdrop_frames=1,
                 memo_momentum=0.5, nms_conf_thr=0.5, nms_backdrop_iou_thr=0.5, nms_class_iou_thr=0.7,
                 with_cats=True, match_metric='bisoftmax', long_match=False, frame_weight=False,
                 temporal_weight=False, memory_len=10, capacity=1024, channels=None):
        assert 0 <= memo_momentum <= 1.0
        assert memo_tracklet_frames >= 0
        assert memo_backdrop_frames >= 0
        assert match_metric in METRICS
        self._args = dict(capacity=int(capacity), memory_len=int(memory_len),
                          memo_tracklet_frames=int(memo_tracklet_frames), match_metric=METRICS[match_metric],
                          long_match=int(bool(long_match)), frame_weight=int(bool(frame_weight)),
                          temporal_weight=int(bool(temporal_weight)), nms_thr_pre=nms_thr_pre,
                          nms_thr_post=nms_thr_post, init_score_thr=init_score_thr,
                          addnew_score_thr=addnew_score_thr, match_score_thr=match_score_thr,
                          memo_momentum=memo_momentum)
        self.channels = channels
        self.cfg = None
        self.state = None

    # limits of vnext_amd/csrc/tracker.hip (kTrkMaxDet, kTrkMaxMem, kTrkMaxCap)
    MAX_DETS, MAX_MEMORY_LEN, MAX_CAPACITY = 512, 16, 2048

    @classmethod
    def supports(cls, memory_len=10, max_dets=0, capacity=1024):
        """whether the device kernels take this configuration; callers fall back to `IDOL_Tracker` otherwise"""
        return 1 <= int(memory_len) <= cls.MAX_MEMORY_LEN and int(max_dets) <= cls.MAX_DETS and \
            1 <= int(capacity) <= cls.MAX_CAPACITY

    def _start(self, device, channels):
        import ctypes
        from .. import _lib
        self.channels = int(channels)
        self.cfg = _lib.TrackerConfig(channels=self.channels, **self._args)
        self._cfg_ptr = ctypes.addressof(self.cfg)
        size = _lib.lib().vnx_tracker_state_bytes(self._cfg_ptr)
        if size == 0:
            raise _lib.VnextHipError(_lib.lib().vnx_last_error().decode())
        self.state = torch.empty(size, dtype=torch.uint8, device=device)
        with torch.cuda.device(device):
            _lib.check(_lib.lib().vnx_tracker_reset(self._cfg_ptr, self.state.data_ptr(), _lib.current_stream(self.state)))

    def counters(self):
        """(tracklets created, tracklets that found no free slot, frames processed) -- one host copy."""
        if self.state is None:
            return 0, 0, 0
        return tuple(self.state[:12].view(torch.int32).tolist())

    @property
    def num_tracklets(self):
        return self.counters()[0]

    def match_device(self, bboxes, labels, masks, track_feats, frame_id):
        from .. import _lib
        n = bboxes.shape[0]
        if not bboxes.is_cuda:
            raise RuntimeError("DeviceTracker: no CPU implementation (HIP library only); IDOL_Tracker is the host form")
        if self.state is None:
            self._start(bboxes.device, track_feats.shape[1])
        ids = torch.empty(n, dtype=torch.int64, device=bboxes.device)
        if n == 0:
            return ids
        logits = masks.reshape(n, -1).float().contiguous()
        embeds = track_feats.float().contiguous()
        scores = bboxes[:, 4].float().contiguous()
        labels = labels.to(torch.int64).contiguous()
        lib = _lib.lib()
        ws = torch.empty(lib.vnx_tracker_frame_workspace_bytes(self._cfg_ptr, n, logits.shape[1]), dtype=torch.uint8,
                         device=bboxes.device)
        with torch.cuda.device(bboxes.device):
            _lib.check(lib.vnx_tracker_frame(self._cfg_ptr, self.state.data_ptr(), logits.data_ptr(), embeds.data_ptr(),
                                             scores.data_ptr(), labels.data_ptr(), n, logits.shape[1], int(frame_id),
                                             ids.data_ptr(), ws.data_ptr(), ws.numel(), _lib.current_stream(ids)))
        return ids

    def match(self, bboxes, labels, masks, track_feats, frame_id, indices):
        n = bboxes.shape[0]
        if n == 0:
            return bboxes, labels, torch.zeros(0, dtype=torch.long), []
        ids = self.match_device(bboxes, labels, masks, track_feats, frame_id).cpu()
        kept = torch.nonzero(ids > -3).squeeze(1)
        kept_dev = kept.to(bboxes.device)
        return bboxes[kept_dev], labels[kept_dev], ids[kept], [indices[i] for i in kept.tolist()]
