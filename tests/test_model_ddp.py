"""Model plumbing and clip-level data parallelism (b, e)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import vnext_amd.models  # noqa: F401  (registers the meta-archs)
from conftest import ROOT
from vnext_amd import train as T
from vnext_amd.registry import META_ARCH_REGISTRY, build_model, get_seqformer_cfg

TINY = {"MODEL.SeqFormer.ENC_LAYERS": 1, "MODEL.SeqFormer.DEC_LAYERS": 2, "MODEL.SeqFormer.NUM_OBJECT_QUERIES": 12,
        "MODEL.SeqFormer.DIM_FEEDFORWARD": 64, "MODEL.SeqFormer.DROPOUT": 0.0, "INPUT.SAMPLING_FRAME_NUM": 2}


def test_registry_surface_and_state_dict_names():
    assert "SeqFormer" in META_ARCH_REGISTRY
    cfg = get_seqformer_cfg(**{"MODEL.DEVICE": "cpu", **TINY})
    model = build_model(cfg)
    keys = set(model.state_dict())
    # the names checkpoints of the reference carry
    for k in ("detr.detr.transformer.encoder.layers.0.self_attn.sampling_offsets.weight",
              "detr.detr.transformer.decoder.layers.1.cross_attn.output_proj_box.weight",
              "detr.detr.transformer.decoder.layers.0.self_attn_box.in_proj_weight",
              "detr.detr.transformer.level_embed", "detr.detr.transformer.reference_points.weight",
              "detr.detr.class_embed.1.weight", "detr.detr.bbox_embed.0.layers.2.bias",
              "detr.detr.query_embed.weight", "detr.detr.input_proj.3.0.weight", "detr.controller.layers.2.weight",
              "detr.mask_head.lay1.weight", "detr.mask_head.dcn.weight"):
        assert k in keys, k


def test_shard_indices_cover_all_clips_once():
    for world in (1, 2, 3, 8):
        seen = sorted(i for r in range(world) for i in T.shard_indices(17, r, world))
        assert seen == list(range(17))


def _grid_sample_function():
    """Differentiable stand-in for the HIP op on CPU (tests only)."""
    from oracle.msda_torch_fallback import msda_grid_sample

    class Fn:
        @staticmethod
        def apply(value, shapes, lsi, loc, attn, step):
            return msda_grid_sample(value, shapes, loc, attn)
    return Fn


def _cpu_stand_ins():
    """Swap the two HIP entry points the model calls for differentiable PyTorch restatements
    (oracle/, tests only) so the model steps on CPU; returns the undo."""
    from oracle.heads_torch_fallback import dynamic_mask_head_torch
    from vnext_amd.models import seqformer as sf
    from vnext_amd.ops.modules import ms_deform_attn as mod
    old = (mod.MSDeformAttnFunction, sf.dynamic_mask_head)
    mod.MSDeformAttnFunction = _grid_sample_function()
    sf.dynamic_mask_head = dynamic_mask_head_torch

    def undo():
        mod.MSDeformAttnFunction, sf.dynamic_mask_head = old
    return undo


def test_training_branch_returns_the_reference_loss_names_and_reaches_every_parameter():
    undo = _cpu_stand_ins()
    try:
        torch.manual_seed(1)
        model = build_model(get_seqformer_cfg(**{"MODEL.DEVICE": "cpu", **TINY})).train()
        clips = T.synthetic_clips(2, 2, 64, 96, "cpu", seed=5, num_instances=2)
        clips[1].pop("instances")                       # a clip without annotations: empty target set
        losses = model(clips)
        names = {"loss_ce", "loss_bbox", "loss_giou", "loss_mask", "loss_dice"}
        assert set(losses) == names | {f"{k}_0" for k in names} | {"class_error"}   # DEC_LAYERS = 2
        assert all(torch.isfinite(v) for v in losses.values())
        sum(losses.values()).backward()
        # the encoder layers own an output_proj_box they never call (ms_deform_attn.py:77-80 creates
        # it for every MSDeformAttn; only decode_forward uses it) -- the reason the reference
        # needs FIND_UNUSED_PARAMETERS; a static set, so DDP's static graph handles it
        missing = [n for n, p in model.named_parameters()
                   if p.requires_grad and p.grad is None and not ("encoder" in n and "output_proj_box" in n)]
        assert not missing, missing
        # no annotations at all still gives a graph that reaches the mask branch (DDP static graph)
        model.zero_grad()
        for c in clips:
            c.pop("instances", None)
        sum(model(clips).values()).backward()
        assert model.detr.controller.layers[0].weight.grad is not None
    finally:
        undo()


def test_box_predictions_of_the_refinement_loop_are_the_detectors_box_head():
    """The decoder's refinement loop and the detector's box head evaluate the same expression with the same modules and
    references (deformable_transformer.py:366-380 / deformable_detr.py:195-213); `_heads` takes the loop's result.  Values
    and the gradient of every parameter must equal the head's own recomputation."""
    undo = _cpu_stand_ins()
    try:
        torch.manual_seed(2)
        cfg = {"MODEL.DEVICE": "cpu", **TINY, "MODEL.SeqFormer.DEC_LAYERS": 3}
        model = build_model(get_seqformer_cfg(**cfg)).train()
        clips = T.synthetic_clips(2, 2, 64, 96, "cpu", seed=7, num_instances=2)

        def run(reuse):
            model.zero_grad()
            x, mask = model._preprocess(clips)
            srcs, masks, poss = model._features(x, mask)
            d = model.detr.detr
            hs, hs_box, _, init_ref, inter_refs, inter_boxes, _, _ = d.transformer(srcs, masks, poss, d.query_embed.weight)
            assert inter_boxes is not None and inter_boxes.requires_grad and not inter_refs.requires_grad
            assert torch.equal(inter_boxes.detach(), inter_refs)
            logits, boxes = model._heads(hs, hs_box, init_ref, inter_refs, inter_boxes if reuse else None)
            g = torch.Generator().manual_seed(3)
            ((boxes * torch.randn(boxes.shape, generator=g)).sum() + (logits * torch.randn(logits.shape, generator=g)).sum()).backward()
            return boxes.detach(), {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}

        boxes_a, grads_a = run(True)
        boxes_b, grads_b = run(False)
        torch.testing.assert_close(boxes_a, boxes_b, rtol=0, atol=1e-6)
        assert set(grads_a) == set(grads_b) and any("bbox_embed" in n for n in grads_a)
        for n in grads_a:
            torch.testing.assert_close(grads_a[n], grads_b[n], rtol=0, atol=2e-5 * float(grads_b[n].abs().max()) + 1e-7, msg=n)
    finally:
        undo()


def _ddp_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    _cpu_stand_ins()
    T.init_distributed("gloo")
    torch.manual_seed(0)
    cfg = get_seqformer_cfg(**{"MODEL.DEVICE": "cpu", **TINY})
    model = build_model(cfg).train()
    ddp = T.wrap_ddp(model)
    clips = T.synthetic_clips(4, 2, 64, 96, "cpu", seed=7)
    mine = [clips[i] for i in T.shard_indices(len(clips), rank, world)]
    loss = sum(ddp(mine).values())
    loss.backward()
    flat = torch.cat([p.grad.flatten() for p in model.parameters() if p.grad is not None])
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    if rank == 0:
        torch.save({"grads": gathered, "loss": loss.detach()}, out)
    dist.barrier()
    dist.destroy_process_group()


def _ddp_timer_worker(rank, world, port, out):
    """A small dense model (no MSDA stand-ins needed): 4 ranks, 1 MB buckets, CommTimer hook."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    T.init_distributed("gloo")
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(256, 512), torch.nn.ReLU(), torch.nn.Linear(512, 512), torch.nn.ReLU(),
                                torch.nn.Linear(512, 256))
    timer = T.CommTimer()
    ddp = T.wrap_ddp(model, bucket_cap_mb=1, comm_timer=timer)
    x = torch.randn(8, 256, generator=torch.Generator().manual_seed(100 + rank))
    reports = []
    for _ in range(4):          # DDP rebuilds its buckets (to the requested cap) after the first iterations
        timer.begin_step()
        model.zero_grad()
        ddp(x).square().mean().backward()
        reports.append(timer.report())
    flat = torch.cat([p.grad.flatten() for p in model.parameters()])
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    if rank == 0:
        torch.save({"grads": gathered, "reports": reports}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_ddp_world_size_4_gloo_buckets_and_comm_timer(tmp_path):
    """Four CPU processes over gloo (the widest world this container's 8 cores run comfortably): explicit bucket
    cap, the timing hook in place of DDP's own all-reduce -- every rank still ends with the mean of the four shards'
    gradients, and the timer reports one span per bucket and the exposed tail."""
    out = str(tmp_path / "ddp4.pt")
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_ddp_timer_worker, args=(4, port, out), nprocs=4, join=True)
    res = torch.load(out)
    for g in res["grads"][1:]:
        assert torch.equal(g, res["grads"][0])
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(256, 512), torch.nn.ReLU(), torch.nn.Linear(512, 512), torch.nn.ReLU(),
                                torch.nn.Linear(512, 256))
    acc = None
    for r in range(4):
        model.zero_grad()
        x = torch.randn(8, 256, generator=torch.Generator().manual_seed(100 + r))
        model(x).square().mean().backward()
        flat = torch.cat([p.grad.flatten() for p in model.parameters()])
        acc = flat if acc is None else acc + flat
    torch.testing.assert_close(res["grads"][0], acc / 4, rtol=1e-5, atol=1e-7)
    rep = res["reports"][-1]
    # 0.53 M parameters = 2.1 MB in 1 MB buckets: at least two all-reduces per step, all of them timed
    assert rep["buckets"] >= 2 and rep["allreduce_ms_sum"] > 0 and rep["exposed_allreduce_ms"] is not None
    assert 0 <= rep["exposed_allreduce_ms"] <= rep["allreduce_ms_sum"] + 1e-6


def test_rank_affinity_partitions_the_allowed_cores():
    """set_rank_affinity without a GPU: the allowed cores divided evenly, restored afterwards."""
    if not hasattr(os, "sched_setaffinity"):
        pytest.skip("no sched_setaffinity on this platform")
    before = os.sched_getaffinity(0)
    threads = torch.get_num_threads()
    try:
        seen = []
        for r in range(2):
            os.sched_setaffinity(0, before)
            info = T.set_rank_affinity(r, 2)
            assert info is not None and info["cpus"] == max(1, len(before) // 2)
            seen.append(os.sched_getaffinity(0))
        if len(before) >= 2:
            assert not (seen[0] & seen[1])
    finally:
        os.sched_setaffinity(0, before)
        torch.set_num_threads(threads)


def test_ddp_world_size_2_gloo_averages_shard_gradients(tmp_path):
    """Two CPU processes over gloo: after backward every rank holds the same gradient, and it is
    the mean of the two shards' single-process gradients."""
    out = str(tmp_path / "ddp.pt")
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_ddp_worker, args=(2, port, out), nprocs=2, join=True)
    res = torch.load(out)
    g0, g1 = res["grads"]
    assert torch.equal(g0, g1)
    # single process reference: average of the per-shard gradients
    undo = _cpu_stand_ins()
    try:
        torch.manual_seed(0)
        model = build_model(get_seqformer_cfg(**{"MODEL.DEVICE": "cpu", **TINY})).train()
        clips = T.synthetic_clips(4, 2, 64, 96, "cpu", seed=7)
        acc = None
        for r in range(2):
            model.zero_grad()
            sum(model([clips[i] for i in T.shard_indices(4, r, 2)]).values()).backward()
            flat = torch.cat([p.grad.flatten() for p in model.parameters() if p.grad is not None])
            acc = flat if acc is None else acc + flat
    finally:
        undo()
    np.testing.assert_allclose(g0.numpy(), (acc / 2).numpy(), rtol=1e-4, atol=1e-7)


@pytest.mark.gpu
def test_train_step_and_inference_on_gpu():
    cfg = get_seqformer_cfg(**{"MODEL.DEVICE": "cuda:0", **TINY})
    model = build_model(cfg).train()
    opt = T.build_optimizer(model)
    clips = T.synthetic_clips(2, 2, 96, 160, "cuda:0", seed=3)
    before = model.detr.detr.transformer.encoder.layers[0].self_attn.value_proj.weight.detach().clone()
    l0 = T.train_step(model, opt, clips)
    l1 = T.train_step(model, opt, clips)
    assert torch.isfinite(l0) and torch.isfinite(l1)
    after = model.detr.detr.transformer.encoder.layers[0].self_attn.value_proj.weight
    assert not torch.equal(before, after), "the op's backward must reach the parameters"
    model.eval()
    model.multi_cls = False      # one result per kept query (with MULTI_CLS_ON the count depends on the scores)
    res = model(clips[:1])
    assert set(res) == {"image_size", "pred_scores", "pred_labels", "pred_masks"}   # seqformer.py:403-408
    assert len(res["pred_masks"]) == len(res["pred_scores"]) == len(res["pred_labels"]) == 10
    assert tuple(res["pred_masks"][0].shape) == (2, 96, 160) and res["pred_masks"][0].dtype == torch.bool


@pytest.mark.gpu
def test_train_step_under_bf16_autocast_on_gpu():
    """bf16 I/O: GEMMs and the op's `value` / output in bf16, locations and
    attention weights fp32 (softmax autocasts up) -- the mixed signature the C ABI accepts."""
    cfg = get_seqformer_cfg(**{"MODEL.DEVICE": "cuda:0", **TINY})
    model = build_model(cfg).train()
    clips = T.synthetic_clips(1, 2, 96, 160, "cuda:0", seed=4)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        losses = model(clips)
    total = sum(losses.values())
    assert torch.isfinite(total)
    total.backward()
    g = model.detr.detr.transformer.encoder.layers[0].self_attn.value_proj.weight.grad
    assert g is not None and torch.isfinite(g).all() and float(g.abs().sum()) > 0


@pytest.mark.gpu
def test_graph_replayed_inference_equals_eager_on_gpu():
    """The inference trunk replayed from a hipGraph (capture, then two replays with different
    clips) gives what the eager trunk gives."""
    torch.manual_seed(5)
    model = build_model(get_seqformer_cfg(**{"MODEL.DEVICE": "cuda:0", **TINY})).eval()
    clips = T.synthetic_clips(3, 2, 96, 160, "cuda:0", seed=8, num_instances=0)
    graphed = [model([c]) for c in clips]              # first call captures, the others replay
    assert len(model._graphs) == 1
    model.graph_inference = False
    eager = [model([c]) for c in clips]
    for g, e in zip(graphed, eager):
        assert g["pred_labels"] == e["pred_labels"]
        np.testing.assert_allclose(g["pred_scores"], e["pred_scores"], rtol=1e-5)
        for mg, me in zip(g["pred_masks"], e["pred_masks"]):
            assert float((mg != me).float().mean()) < 1e-3


@pytest.mark.gpu
def test_graph_captured_training_trunk_matches_eager_gradients_on_gpu():
    """Training trunk replayed from forward/backward hipGraphs: same losses and gradients as the
    eager trunk (dropout off so that both are deterministic)."""
    def run(graph):
        torch.manual_seed(11)
        model = build_model(get_seqformer_cfg(**{"MODEL.DEVICE": "cuda:0", **TINY})).train()
        model.graph_training = graph
        clips = T.synthetic_clips(1, 2, 96, 160, "cuda:0", seed=6, num_instances=2)
        out = []
        for _ in range(2):                     # the second step replays
            model.zero_grad(set_to_none=True)
            losses = model(clips)
            sum(losses.values()).backward()
            out.append(({k: float(v) for k, v in losses.items()},
                        {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}))
        return out
    eager, graphed = run(False), run(True)
    for (le, ge), (lg, gg) in zip(eager, graphed):
        for k in le:
            np.testing.assert_allclose(lg[k], le[k], rtol=2e-4, atol=1e-5, err_msg=k)
        assert set(ge) == set(gg)
        for n in ge:
            scale = float(ge[n].abs().max()) + 1e-12
            # MIOpen's weight-gradient kernels and the mask head's atomics sum in run-dependent order
            assert float((gg[n] - ge[n]).abs().max()) <= 1e-2 * scale + 1e-7, n


@pytest.mark.gpu
def test_clip_matching_inference_on_gpu():
    """CLIP_MATCHING: a 7-frame video as clips of 3 frames every 2 -- (query, class) results with a
    mask per frame; one graph per clip shape."""
    torch.manual_seed(6)
    cfg = get_seqformer_cfg(**{"MODEL.DEVICE": "cuda:0", "MODEL.SeqFormer.CLIP_MATCHING": True,
                               "MODEL.SeqFormer.CLIP_LENGTH": 3, "MODEL.SeqFormer.CLIP_STRIDE": 2, **TINY})
    model = build_model(cfg).eval()
    model.multi_cls = False
    video = [{"image": T.synthetic_clips(1, 7, 96, 160, "cuda:0", seed=2, num_instances=0)[0]["image"],
              "height": 100, "width": 170}]
    res = model(video)
    assert set(res) == {"image_size", "pred_scores", "pred_labels", "pred_masks"} and res["image_size"] == (100, 170)
    assert len(res["pred_masks"]) == len(res["pred_scores"]) >= 10            # at least the first clip's 10 tracks
    assert all(tuple(m.shape) == (7, 100, 170) and m.dtype == torch.bool for m in res["pred_masks"])
    assert len(model._graphs) == 1


def test_frozen_batchnorm_folded_into_the_convolution():
    from vnext_amd.models.seqformer import FrozenBatchNorm2d, conv_bn
    torch.manual_seed(0)
    conv = torch.nn.Conv2d(5, 7, 3, 2, 1, bias=False)
    bn = FrozenBatchNorm2d(7)
    bn.weight.copy_(torch.rand(7) + 0.5); bn.bias.copy_(torch.randn(7))
    bn.running_mean.copy_(torch.randn(7)); bn.running_var.copy_(torch.rand(7) + 0.2)
    x = torch.randn(2, 5, 9, 11)
    np.testing.assert_allclose(conv_bn(x, conv, bn).detach().numpy(), bn(conv(x)).detach().numpy(), rtol=1e-4, atol=1e-5)
    bn.running_var.mul_(2.0)                      # a changed buffer invalidates the cached constants
    np.testing.assert_allclose(conv_bn(x, conv, bn).detach().numpy(), bn(conv(x)).detach().numpy(), rtol=1e-4, atol=1e-5)
    conv_bn(x, conv, bn).sum().backward()
    assert conv.weight.grad is not None


@pytest.mark.gpu
@pytest.mark.parametrize("graph", [False, True])
@pytest.mark.parametrize("arch", ["SeqFormer", "IDOL"])
def test_ddp_wrapper_over_rccl_on_one_gpu(arch, graph):
    """The N > 1 code path with the real backend: a one-rank RCCL process group, the DDP wrapper
    bench.py / train.py use (static graph, gradients as bucket views), the criteria's own
    all-reduce of the box count, frozen stages and never-used parameters -- three steps.
    graph: with the training trunk replayed from hipGraphs (`train.capture_training_graphs`, called before the wrapper; the
    trunk's gradients reach DDP's buckets when the replayed backward returns) -- and the gradients equal the eager DDP
    step's."""
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel
    from vnext_amd.registry import get_idol_cfg
    port = 29600 + (os.getpid() % 300) + (0 if arch == "SeqFormer" else 1) + (2 if graph else 0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        torch.manual_seed(21)
        if arch == "SeqFormer":
            model = build_model(get_seqformer_cfg(**{"MODEL.DEVICE": "cuda:0", **TINY})).train()
        else:
            model = build_model(get_idol_cfg(**{"MODEL.DEVICE": "cuda:0", "MODEL.IDOL.ENC_LAYERS": 1,
                                                "MODEL.IDOL.DEC_LAYERS": 2, "MODEL.IDOL.NUM_OBJECT_QUERIES": 110,
                                                "MODEL.IDOL.DIM_FEEDFORWARD": 64, "MODEL.IDOL.DROPOUT": 0.0})).train()
        clips = T.synthetic_clips(2, 2, 96, 160, "cuda:0", seed=12, num_instances=2)
        if graph:      # BEFORE the wrapper (train.capture_training_graphs says why)
            assert T.capture_training_graphs(model, clips)["enabled"]
        ddp = DistributedDataParallel(model, device_ids=[0], broadcast_buffers=False, find_unused_parameters=False,
                                      static_graph=True, gradient_as_bucket_view=True)
        opt = T.build_optimizer(model)
        losses = [float(T.train_step(ddp, opt, clips)) for _ in range(3)]
        assert all(np.isfinite(losses))
        if not graph:
            # a second wrapper around the same model, stepped under bf16 autocast from its first iteration (BASELINE config 3; a
            # static-graph DDP instance cannot change precision mid-run): the transformer layers' shadow weights
            # (ops/shadow_weights.py) hand their gradients back to the fp32 parameters DDP's hooks sit on -- every parameter the fp32
            # step reaches is reached, and the swap leaves the modules with their fp32 Parameters
            reached = {n for n, p in model.named_parameters() if p.grad is not None}
            del ddp
            ddp16 = DistributedDataParallel(model, device_ids=[0], broadcast_buffers=False, find_unused_parameters=False,
                                            static_graph=True, gradient_as_bucket_view=True)
            for _ in range(2):
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    amp_loss = float(T.train_step(ddp16, opt, clips))
                assert np.isfinite(amp_loss)
            assert {n for n, p in model.named_parameters() if p.grad is not None} == reached
            assert all(isinstance(p, torch.nn.Parameter) and p.dtype == torch.float32 for p in model.parameters())
        if graph:
            assert len(model._train_trunks) == 1
            # one more backward through DDP, replayed, against the same model's eager backward through DDP
            import random
            grads = []
            for g in (True, False):
                model.graph_training = g
                for m in model.modules():
                    if isinstance(m, torch.nn.Dropout):
                        m.p = 0.0
                    if isinstance(m, torch.nn.MultiheadAttention):
                        m.dropout = 0.0
                random.seed(3)
                ddp.zero_grad(set_to_none=True)
                sum(ddp(clips).values()).backward()
                grads.append({n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None})
            assert set(grads[0]) == set(grads[1])
            for n in grads[0]:      # (MIOpen's weight-gradient kernels sum in run-dependent order: several per cent on single
                #                         elements of the trunk's filters -- a tensor's gradient as a whole is the stable quantity)
                assert float((grads[0][n] - grads[1][n]).norm()) <= 2e-2 * float(grads[1][n].norm()) + 1e-7, n
    finally:
        dist.destroy_process_group()


def test_folding_all_backbone_scales_at_once_changes_nothing():
    """`fold_all` (one multi-tensor multiply for all filters, one in the backward) against the per-convolution
    expression `conv.weight * scale` it replaces: same features, same filter gradients, frozen stages untouched."""
    from vnext_amd.models import seqformer as SF
    torch.manual_seed(0)
    net = SF.ResNet50Trunk().freeze(2)
    for m in net.modules():
        if isinstance(m, SF.FrozenBatchNorm2d):
            m.weight.copy_(torch.rand_like(m.weight) + 0.5)
            m.bias.copy_(torch.randn_like(m.bias) * 0.1)
            m.running_mean.copy_(torch.randn_like(m.running_mean) * 0.1)
            m.running_var.copy_(torch.rand_like(m.running_var) + 0.5)
    x = torch.randn(1, 3, 64, 96)
    outs = net(x)
    sum((o * o).sum() for o in outs).backward()
    got = {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}
    assert not any(n.startswith(("stem", "res2")) for n in got) and len(got) == 42
    for p in net.parameters():
        p.grad = None

    def plain(x):
        y = F.max_pool2d(F.relu(SF.conv_bn(x, net.stem[0], net.stem[1])), 3, 2, 1)
        feats = []
        for stage in (net.res2, net.res3, net.res4, net.res5):
            for b in stage:
                y = b(y)
            feats.append(y)
        return feats[1:]
    import torch.nn.functional as F
    want = plain(x)
    sum((o * o).sum() for o in want).backward()
    for a, b in zip(outs, want):
        assert torch.equal(a, b)
    for n, p in net.named_parameters():
        if p.grad is not None:
            assert torch.equal(got[n], p.grad), n


@pytest.mark.gpu
@pytest.mark.parametrize("amp", [False, True])
def test_config4_seqformer_720p_step_reaches_every_parameter(amp):
    """BASELINE config 4 at N = 1 (projects/SeqFormer/configs/large_model/swin_ytvis.yaml:29,45 -- 720p frames, one clip per
    GPU on 8 GPUs; R50 trunk for Swin-L): the full-size model on ONE T = 5 clip of 720 x 1280 frames -- the 19 560-pixel
    encoder, 6 decoder layers x 5 frames of 720p mask-head training, criterion and optimiser -- runs, the loss is finite and
    every parameter that takes part receives a finite gradient; fp32 and bf16 autocast.  (bench.py reports its
    ms / step as other_configs.seqformer_train_step_720p.)"""
    torch.manual_seed(0)
    model = build_model(get_seqformer_cfg(**{"MODEL.DEVICE": "cuda:0"})).train()
    opt = T.build_optimizer(model)
    clips = T.synthetic_clips(1, 5, 720, 1280, "cuda:0", seed=104, num_instances=4)
    if amp:
        with torch.autocast("cuda", dtype=torch.bfloat16):
            losses = model(clips)
    else:
        losses = model(clips)
    total = sum(losses.values())
    assert torch.isfinite(total), {k: float(v) for k, v in losses.items()}
    opt.zero_grad(set_to_none=True)
    total.backward()
    with_grad = [(n, p) for n, p in model.named_parameters() if p.requires_grad and p.grad is not None]
    # the encoder layers own an output_proj_box they never call (the reason the reference needs FIND_UNUSED_PARAMETERS): as in
    # test_training_branch_returns_the_reference_loss_names_and_reaches_every_parameter
    missing = [n for n, p in model.named_parameters()
               if p.requires_grad and p.grad is None and not ("encoder" in n and "output_proj_box" in n)]
    assert not missing, missing[:10]
    assert len(with_grad) > 300
    bad = [n for n, p in with_grad if not bool(torch.isfinite(p.grad).all())]
    assert not bad, bad[:10]
    assert sum(float(p.grad.abs().sum()) > 0 for _, p in with_grad) > 0.9 * len(with_grad)
    opt.step()                                         # the clipped AdamW step at this size
    assert all(bool(torch.isfinite(p).all()) for _, p in with_grad[:50])
    del model, opt
    torch.cuda.empty_cache()
