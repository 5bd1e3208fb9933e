"""Clip-level instance tracking for long SeqFormer videos.

Same decisions and results as `Videos` / `Clips` of
projects/SeqFormer/seqformer/models/clip_output.py:11-144 (overlapping clips are linked by the
space-time IoU of their masks on the shared frames, Hungarian matching at threshold 0.01, a
track's mask logits / class probabilities are the average over the clips that contain it).

Laid out for the machine differently.  The reference keeps two dense
[num_clips, 120, video_length, H/4, W/4] float tensors (19 GB for a 36-frame 360p video, hence its
MERGE_ON_CPU default) and evaluates the sIoU by broadcasting to [C, N_s, N_i, T, HW].  Here a clip
stores only its own frames ([n, clip_len, HW]), and the sIoU of a stored clip against the incoming one
is ONE matrix product on the matrix cores -- intersections = A . B^T over the shared frames' pixels,
unions from the two row sums -- so everything stays on the device; the only host step is the
[N_s, N_i] Hungarian assignment.
"""
from __future__ import annotations

from typing import List

import torch
from scipy.optimize import linear_sum_assignment

from . import tracker as _trk      # shares the similarity entry point (tests swap it for torch.mm on CPU)


class Clips:
    """One clip's kept instances (clip_output.py:131-144): cls_probs [n, K], mask_logits [n, T, h, w]."""

    def __init__(self, frame_idx: List[int], results):
        self.frame_idx = list(frame_idx)
        self.frame_set = set(frame_idx)
        self.classes, self.scores = results.pred_classes, results.scores
        self.cls_probs = results.cls_probs
        self.mask_logits = results.pred_masks
        self.mask_probs = results.pred_masks.sigmoid()
        self.num_instance = len(self.scores)


class Videos:
    def __init__(self, num_frames, video_length, num_classes, image_size, device):
        self.num_frames, self.video_length, self.num_classes = num_frames, video_length, num_classes
        self.image_size, self.device = tuple(image_size), device
        self.match_threshold = 0.01
        self.num_inst = 0
        self.saved_idx_set = set()
        self.clips = []     # per stored clip: (frame_idx, track ids [n], logits [n,T,h,w], probs [n,T,hw], cls [n,K])

    @property
    def num_clip(self):
        return len(self.clips)

    def get_siou(self, input_clip):
        """-> [num_inst, N_i]: mean over the recent clips that contain the track of the space-time
        IoU on the frames shared with the incoming clip (clip_output.py:36-63, 73-81)."""
        n_i = input_clip.num_instance
        siou = torch.zeros(self.num_inst, n_i, device=self.device)
        count = torch.zeros(self.num_inst, device=self.device)
        in_pos = {f: k for k, f in enumerate(input_clip.frame_idx)}
        b_all = input_clip.mask_probs.flatten(2).to(self.device, torch.float32)          # [N_i, T_in, HW]
        for frame_idx, ids, _, probs, _ in self.clips[max(self.num_clip - len(input_clip.frame_idx), 0):]:
            shared = [(k, in_pos[f]) for k, f in enumerate(frame_idx) if f in in_pos and f in self.saved_idx_set]
            if not shared or len(ids) == 0:
                continue
            a = probs[:, [k for k, _ in shared]].flatten(1)                                # [n_c, |shared| HW]
            b = b_all[:, [j for _, j in shared]].flatten(1)
            inter = _trk._pairwise_dot(a.contiguous(), b.contiguous())                     # one GEMM
            union = a.sum(1)[:, None] + b.sum(1)[None, :] - inter
            siou[ids] += inter / (union + 1e-6)
            count[ids] += 1
        return siou / (count[:, None] + 1e-6)

    def update(self, input_clip):
        n_i = input_clip.num_instance
        scores = self.get_siou(input_clip) if (self.num_inst and n_i) else torch.zeros(self.num_inst, n_i)
        above = scores > self.match_threshold
        host = (scores * above.float()).cpu().numpy()
        above = above.cpu().numpy()
        rows, cols = linear_sum_assignment(host, maximize=True) if host.size else ([], [])
        track_of = {}
        for r, c in zip(rows, cols):
            if above[r, c]:
                track_of[int(c)] = int(r)
        for c in range(n_i):                       # unmatched instances open new tracks, in input order
            if c not in track_of:
                track_of[c] = self.num_inst
                self.num_inst += 1
        ids = torch.tensor([track_of[c] for c in range(n_i)], dtype=torch.long, device=self.device)
        logits = input_clip.mask_logits.to(self.device, torch.float32)
        self.clips.append((list(input_clip.frame_idx), ids, logits, input_clip.mask_probs.flatten(2).to(self.device, torch.float32),
                           input_clip.cls_probs.to(self.device, torch.float32)))
        self.saved_idx_set.update(input_clip.frame_set)

    def get_result(self):
        """-> (class probabilities [N, K], mask logits [N, video_length, h, w]); frames no clip of a
        track covers are 0/0 = NaN as in the reference (they threshold to background)."""
        h, w = self.image_size
        total = torch.zeros(self.num_inst, self.video_length, h, w, device=self.device)
        seen = torch.zeros(self.num_inst, self.video_length, device=self.device)
        cls = torch.zeros(self.num_inst, self.num_classes, device=self.device)
        in_clips = torch.zeros(self.num_inst, device=self.device)
        for frame_idx, ids, logits, _, cls_probs in self.clips:
            if len(ids) == 0:
                continue
            f = torch.tensor(frame_idx, dtype=torch.long, device=self.device)
            total[ids[:, None], f[None, :]] += logits
            seen[ids[:, None], f[None, :]] += 1
            cls[ids] += cls_probs
            in_clips[ids] += 1
        return cls / in_clips[:, None], total / seen[..., None, None]
