"""IDOL meta-architecture (b): registry surface, training branch, and the
video-level inference post-processing against the reference's `IDOL.inference` + IDOL_Tracker
(oracle/make_golden_idol_inference.py)."""
import os

import numpy as np
import pytest
import torch

import vnext_amd.models  # noqa: F401
from conftest import GOLDEN_DIR
from vnext_amd import train as T
from vnext_amd.models import idol as idol_mod
from vnext_amd.models import tracker as trk
from vnext_amd.registry import META_ARCH_REGISTRY, build_model, get_idol_cfg

TINY = {"MODEL.IDOL.ENC_LAYERS": 1, "MODEL.IDOL.DEC_LAYERS": 2, "MODEL.IDOL.NUM_OBJECT_QUERIES": 110,
        "MODEL.IDOL.DIM_FEEDFORWARD": 64, "MODEL.IDOL.DROPOUT": 0.0}


def _loss_reid_torch(ref, key, pos, neg, aux):
    dot = ref @ key.t()
    cos = torch.nn.functional.normalize(ref, dim=1) @ torch.nn.functional.normalize(key, dim=1).t()
    lse_neg = torch.logsumexp(dot.masked_fill(~neg, float("-inf")), dim=0)
    lse_pos = torch.logsumexp((-dot).masked_fill(~pos, float("-inf")), dim=0)
    contrast = torch.nn.functional.softplus((lse_neg + lse_pos).clamp_min(torch.finfo(dot.dtype).min))
    return contrast.sum(), ((((cos - pos.to(cos.dtype)) ** 2) * aux).sum(0) / aux.sum(0).clamp_min(1)).sum()


def _torch_scores(embeds, memo, metric):
    feats = embeds @ memo.t()
    return (feats.softmax(1) + feats.softmax(0)) / 2 if metric == "bisoftmax" else feats.softmax(1)


@pytest.fixture
def cpu_stand_ins(monkeypatch):
    """PyTorch restatements (oracle/, tests only) for the HIP entry points so the model steps on CPU."""
    from oracle.heads_torch_fallback import dynamic_mask_head_torch
    from oracle.msda_torch_fallback import msda_grid_sample
    from vnext_amd.ops.modules import ms_deform_attn as mod

    class Fn:
        @staticmethod
        def apply(value, shapes, lsi, loc, attn, step):
            return msda_grid_sample(value, shapes, loc, attn)
    monkeypatch.setattr(mod, "MSDeformAttnFunction", Fn)
    monkeypatch.setattr(idol_mod, "dynamic_mask_head", dynamic_mask_head_torch)
    monkeypatch.setattr(idol_mod, "loss_reid", _loss_reid_torch)
    monkeypatch.setattr(trk, "_pairwise_dot", lambda a, b: a @ b.t())
    monkeypatch.setattr(trk, "_match_scores", _torch_scores)


def test_registry_and_state_dict_names():
    assert "IDOL" in META_ARCH_REGISTRY
    model = build_model(get_idol_cfg(**{"MODEL.DEVICE": "cpu", **TINY}))
    keys = set(model.state_dict())
    for k in ("detr.detr.transformer.encoder.layers.0.self_attn.sampling_offsets.weight",
              "detr.detr.transformer.decoder.layers.1.cross_attn.output_proj.weight",
              "detr.detr.transformer.decoder.layers.0.self_attn.in_proj_weight",
              "detr.detr.transformer.level_embed", "detr.detr.transformer.reference_points.weight",
              "detr.detr.class_embed.1.weight", "detr.detr.bbox_embed.0.layers.2.bias", "detr.detr.query_embed.weight",
              "detr.controller.layers.2.weight", "detr.mask_head.lay1.weight", "detr.reid_embed_head.layers.1.weight"):
        assert k in keys, k
    assert not any("output_proj_box" in k or "self_attn_box" in k for k in keys)      # SeqFormer-only modules


def test_training_branch_loss_names_and_gradients(cpu_stand_ins):
    torch.manual_seed(2)
    model = build_model(get_idol_cfg(**{"MODEL.DEVICE": "cpu", **TINY})).train()
    pairs = T.synthetic_clips(2, 2, 64, 96, "cpu", seed=9, num_instances=2)
    pairs[1]["instances"][1]["gt_ids"] = torch.tensor([0, -1])       # an object missing from a reference frame
    losses = model(pairs)
    names = {"loss_ce", "loss_bbox", "loss_giou", "loss_mask", "loss_dice"}
    assert set(losses) == names | {"loss_reid", "loss_reid_aux"} | {f"{k}_0" for k in names}
    assert all(torch.isfinite(v) for v in losses.values())
    sum(losses.values()).backward()
    missing = [n for n, p in model.named_parameters() if p.requires_grad and p.grad is None]
    assert not missing, missing


def test_inference_output_format(cpu_stand_ins):
    torch.manual_seed(3)
    model = build_model(get_idol_cfg(**{"MODEL.DEVICE": "cpu", "MODEL.IDOL.BATCH_INFER_LEN": 2, **TINY})).eval()
    g = torch.Generator().manual_seed(0)
    video = [{"image": [torch.rand(3, 64, 96, generator=g) * 255 for _ in range(3)], "height": 70, "width": 100}]
    res = model(video)
    assert set(res) == {"image_size", "pred_scores", "pred_labels", "pred_masks"}       # idol.py:466-471
    assert res["image_size"] == (70, 100) and len(res["pred_masks"]) == len(res["pred_scores"]) == len(res["pred_labels"])
    for track in res["pred_masks"]:
        assert len(track) == 3 and all(m is None or (tuple(m.shape) == (70, 100) and m.dtype == torch.bool) for m in track)


def _associate_from_golden(g, v, device, tracker_cls=None, capacity=None):
    model = build_model(get_idol_cfg(**{"MODEL.DEVICE": device, **TINY})).eval()
    logits = torch.from_numpy(g[f"v{v}.pred_logits"]).to(device)
    boxes = torch.from_numpy(g[f"v{v}.pred_boxes"]).to(device)
    masks = torch.from_numpy(g[f"v{v}.pred_masks"]).to(device)
    embeds = torch.from_numpy(g[f"v{v}.pred_inst_embed"]).to(device)
    picks = model.select_candidates(logits, boxes)
    per_frame = []
    for f, c in enumerate(picks):
        q = torch.from_numpy(c).to(device)
        per_frame.append({"indices": c.tolist(), "logits": logits[f, q], "boxes": boxes[f, q], "embeds": embeds[f, q],
                          "masks": masks[f, q]})
    oh, ow, ih, iw = (int(x) for x in g[f"v{v}.sizes"])
    args = dict(init_score_thr=0.2, obj_score_thr=0.1, nms_thr_pre=0.5, nms_thr_post=0.05, addnew_score_thr=0.2,
                memo_tracklet_frames=10, memo_momentum=0.8, long_match=True, frame_weight=True, temporal_weight=True,
                memory_len=3)
    if capacity is None:
        tracker = (tracker_cls or trk.IDOL_Tracker)(**args)
        res = model.associate(per_frame, tracker, (oh, ow), (ih, iw))
    else:       # a device tracker with too few slots: the video is associated again by the host tracker
        tracker = tracker_cls(capacity=capacity, **args)
        with pytest.raises(RuntimeError, match="more simultaneous tracklets"):
            model.associate(per_frame, tracker_cls(capacity=capacity, **args), (oh, ow), (ih, iw))
        res = model.associate(per_frame, tracker, (oh, ow), (ih, iw), host_factory=lambda: trk.IDOL_Tracker(**args))
        assert tracker.counters()[1] > 0
    np.testing.assert_array_equal(np.array(res["pred_labels"]), g[f"v{v}.labels"])
    np.testing.assert_allclose(np.array(res["pred_scores"]), g[f"v{v}.scores"], rtol=1e-5)
    present = g[f"v{v}.present"]
    want = np.unpackbits(g[f"v{v}.masks"], axis=-1)[..., :ow].astype(bool)
    assert len(res["pred_masks"]) == present.shape[0]
    for i, track in enumerate(res["pred_masks"]):
        assert [m is not None for m in track] == present[i].tolist()
        for t, m in enumerate(track):
            if m is not None:
                assert float((m.numpy() != want[i, t]).mean()) < 2e-3, (i, t)


@pytest.mark.parametrize("v", [0, 1])
def test_video_postprocessing_equals_reference_cpu(v, cpu_stand_ins):
    _associate_from_golden(dict(np.load(os.path.join(GOLDEN_DIR, "inference_idol.npz"))), v, "cpu")


@pytest.mark.gpu
@pytest.mark.parametrize("v", [0, 1])
def test_video_postprocessing_equals_reference_on_gpu(v):
    _associate_from_golden(dict(np.load(os.path.join(GOLDEN_DIR, "inference_idol.npz"))), v, "cuda:0")


@pytest.mark.gpu
@pytest.mark.parametrize("v", [0, 1])
def test_video_postprocessing_with_the_tracker_on_the_device(v):
    """The same fixtures through DeviceTracker: ids stay on the device, one copy per video (models/idol.py:associate)."""
    _associate_from_golden(dict(np.load(os.path.join(GOLDEN_DIR, "inference_idol.npz"))), v, "cuda:0", trk.DeviceTracker)


@pytest.mark.gpu
def test_a_video_that_outgrows_the_device_tracker_is_associated_again_on_the_host():
    """slot overflow used to raise after the whole video and lose it."""
    assert trk.DeviceTracker.supports(memory_len=3, max_dets=300) and not trk.DeviceTracker.supports(memory_len=17)
    assert not trk.DeviceTracker.supports(memory_len=3, max_dets=513)
    _associate_from_golden(dict(np.load(os.path.join(GOLDEN_DIR, "inference_idol.npz"))), 0, "cuda:0", trk.DeviceTracker,
                           capacity=1)


@pytest.mark.gpu
def test_idol_train_step_and_inference_on_gpu():
    cfg = get_idol_cfg(**{"MODEL.DEVICE": "cuda:0", "MODEL.IDOL.BATCH_INFER_LEN": 2, **TINY})
    model = build_model(cfg).train()
    opt = T.build_optimizer(model, base_lr=1e-4)
    pairs = T.synthetic_clips(2, 2, 96, 160, "cuda:0", seed=3, num_instances=3)
    before = model.detr.reid_embed_head.layers[0].weight.detach().clone()
    l0 = T.train_step(model, opt, pairs)
    l1 = T.train_step(model, opt, pairs)
    assert torch.isfinite(l0) and torch.isfinite(l1)
    assert not torch.equal(before, model.detr.reid_embed_head.layers[0].weight), "reid losses must reach the head"
    model.eval()
    video = [{"image": pairs[0]["image"] + pairs[1]["image"] + pairs[0]["image"][:1], "height": 96, "width": 160}]
    res = model(video)                      # chunks of 2, 2 and 1 frames: two graphs captured, one replayed
    assert set(res) == {"image_size", "pred_scores", "pred_labels", "pred_masks"}
    assert all(len(track) == 5 for track in res["pred_masks"]) and len(model._graphs) == 2
    model.graph_inference = False
    eager = model(video)
    assert res["pred_labels"] == eager["pred_labels"]
    np.testing.assert_allclose(res["pred_scores"], eager["pred_scores"], rtol=1e-5)


def test_config_c1_single_480x640_frame_on_cpu_with_the_oracle_op(monkeypatch):
    """one 480x640 frame, random-init R50 + the 4-level, 6-layer encoder on CPU
    with the op served by the C oracle -> memory [1, 6380, 256] (level shapes 60x80, 30x40, 15x20, 8x10)."""
    from oracle import msda_oracle as O
    from vnext_amd.ops.functions import ms_deform_attn_func as func_mod

    class OracleOp:
        @staticmethod
        def ms_deform_attn_forward(value, shapes, lsi, loc, attn, im2col_step):
            return torch.from_numpy(O.msda_forward(value.numpy(), shapes.numpy(), lsi.numpy(), loc.numpy(),
                                                   attn.numpy())).to(value.dtype)
    monkeypatch.setattr(func_mod, "MSDA", OracleOp)
    torch.manual_seed(0)
    model = build_model(get_idol_cfg(**{"MODEL.DEVICE": "cpu", "MODEL.IDOL.DEC_LAYERS": 1,
                                        "MODEL.IDOL.NUM_OBJECT_QUERIES": 20, "MODEL.IDOL.DROPOUT": 0.0})).eval()
    with torch.no_grad():
        x, mask = model._preprocess([torch.rand(3, 480, 640) * 255])
        srcs, hs, memory, refs, inter_refs, loop_boxes = model._encode_decode(x, mask)
        # the box head of the detector = the refinement loop's own prediction (idol_transformer.py)
        _, own = model._box_heads(hs, refs, [0])
    torch.testing.assert_close(loop_boxes[[0]], own, rtol=0, atol=1e-6)
    assert torch.equal(loop_boxes, inter_refs)
    assert [tuple(s.shape[-2:]) for s in srcs] == [(60, 80), (30, 40), (15, 20), (8, 10)]
    assert tuple(memory.shape) == (1, 6380, 256) and tuple(hs.shape) == (1, 1, 20, 256)
    assert torch.isfinite(memory).all() and torch.isfinite(hs).all()


@pytest.mark.gpu
def test_idol_train_step_under_bf16_autocast_on_gpu():
    """BASELINE config 3 (bf16): GEMMs and the op's value in bf16, fp32 reference points / matching /
    losses / reid kernels."""
    cfg = get_idol_cfg(**{"MODEL.DEVICE": "cuda:0", **TINY})
    model = build_model(cfg).train()
    pairs = T.synthetic_clips(1, 2, 96, 160, "cuda:0", seed=5, num_instances=3)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        losses = model(pairs)
    total = sum(losses.values())
    assert torch.isfinite(total)
    total.backward()
    g = model.detr.reid_embed_head.layers[0].weight.grad
    assert g is not None and torch.isfinite(g).all()


@pytest.mark.gpu
@pytest.mark.parametrize("amp", [False, True])
def test_graph_captured_training_trunk_matches_eager_gradients_on_gpu(amp):
    """IDOL.graph_training: the training trunk replayed from forward / backward hipGraphs (fp32, and under bf16 autocast: config
    3) gives the eager trunk's losses and gradients; the second step replays.  Dropout off so that both are deterministic."""
    import random

    def run(graph):
        torch.manual_seed(11)
        random.seed(11)       # (the contrastive negatives are drawn with random.sample, as the reference draws them)
        model = build_model(get_idol_cfg(**{"MODEL.DEVICE": "cuda:0", "MODEL.IDOL.DROPOUT": 0.0, **TINY})).train()
        for m in model.modules():
            if isinstance(m, torch.nn.MultiheadAttention):
                m.dropout = 0.0
        model.graph_training = graph
        pairs = T.synthetic_clips(1, 2, 96, 160, "cuda:0", seed=6, num_instances=3)
        out = []
        for _ in range(2):
            model.zero_grad(set_to_none=True)
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
                losses = model(pairs)
            sum(losses.values()).backward()
            out.append(({k: float(v) for k, v in losses.items()},
                        {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}))
        assert len(model._train_trunks) == (1 if graph else 0)
        return out
    eager, graphed = run(False), run(True)
    # fp32: MIOpen's weight-gradient kernels and the mask head's atomics sum in run-dependent order; bf16: BASELINE's tolerance
    rtol, gtol = (5e-2, 1e-1) if amp else (2e-4, 1e-2)
    for (le, ge), (lg, gg) in zip(eager, graphed):
        for k in le:
            np.testing.assert_allclose(lg[k], le[k], rtol=rtol, atol=1e-4, err_msg=k)
        assert set(ge) == set(gg)
        for n in ge:
            if amp:      # bf16: a tensor's gradient as a whole.  Two EAGER bf16 runs of this model differ by up to 0.20-0.25 of a
                #          tensor's norm (MIOpen's bf16 weight-gradient kernels sum in run-dependent order; measured, round 6;
                #          eager against graphed: 0.10-0.12; fp32: 0.0000) -- the bound is that noise, not bf16's tolerance
                assert float((gg[n] - ge[n]).norm()) <= 0.5 * float(ge[n].norm()) + 1e-6, n
                continue
            scale = float(ge[n].abs().max()) + 1e-12
            assert float((gg[n] - ge[n]).abs().max()) <= gtol * scale + 1e-7, n


def _idol_ddp_worker(rank, world, port, out):
    import sys
    import torch.distributed as dist
    from conftest import ROOT
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from oracle.heads_torch_fallback import dynamic_mask_head_torch
    from oracle.msda_torch_fallback import msda_grid_sample
    from vnext_amd.ops.modules import ms_deform_attn as mod

    class Fn:
        @staticmethod
        def apply(value, shapes, lsi, loc, attn, step):
            return msda_grid_sample(value, shapes, loc, attn)
    mod.MSDeformAttnFunction = Fn
    idol_mod.dynamic_mask_head = dynamic_mask_head_torch
    idol_mod.loss_reid = _loss_reid_torch
    T.init_distributed("gloo")
    torch.manual_seed(0)
    model = build_model(get_idol_cfg(**{"MODEL.DEVICE": "cpu", **TINY})).train()
    ddp = T.wrap_ddp(model)
    pairs = T.synthetic_clips(2, 2, 64, 96, "cpu", seed=21, num_instances=2)
    import random
    random.seed(3)                                    # the host-RNG negative sampling of select_pos_neg
    mine = [pairs[i] for i in T.shard_indices(len(pairs), rank, world)]
    sum(ddp(mine).values()).backward()
    flat = torch.cat([p.grad.flatten() for p in model.parameters() if p.grad is not None])
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    if rank == 0:
        torch.save(gathered, out)
    dist.barrier()
    dist.destroy_process_group()


def test_idol_ddp_world_size_2_gloo_ranks_agree(tmp_path):
    """Key/reference pairs sharded over two CPU processes (gloo): after backward both ranks hold the
    same (averaged) gradient, reid head included."""
    import torch.multiprocessing as mp
    out = str(tmp_path / "idol_ddp.pt")
    port = 29800 + (os.getpid() % 150)
    mp.spawn(_idol_ddp_worker, args=(2, port, out), nprocs=2, join=True)
    g0, g1 = torch.load(out)
    assert torch.equal(g0, g1) and float(g0.abs().sum()) > 0
