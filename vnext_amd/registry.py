"""META_ARCH_REGISTRY surface (model plug point).

The reference registers its models on Detectron2's registry
(`@META_ARCH_REGISTRY.register() class SeqFormer`, projects/SeqFormer/seqformer/seqformer.py:74;
`build_model(cfg)` = `META_ARCH_REGISTRY.get(cfg.MODEL.META_ARCHITECTURE)(cfg)`,
detectron2/modeling/meta_arch/build.py:16-25).  When detectron2 is importable the real registry
is used, so `projects/{SeqFormer,IDOL}` configs resolve to these classes; otherwise (this image
has no detectron2 / fvcore) a registry with the same `register` / `get` contract stands in.
"""
from __future__ import annotations

from types import SimpleNamespace

try:  # pragma: no cover - not installed in the build image
    from detectron2.modeling import META_ARCH_REGISTRY  # type: ignore
    HAVE_DETECTRON2 = True
except Exception:  # noqa: BLE001
    HAVE_DETECTRON2 = False

    class _Registry:
        def __init__(self, name):
            self._name = name
            self._obj_map = {}

        def register(self, obj=None):
            def deco(cls):
                name = cls.__name__
                assert name not in self._obj_map, f"{name} already registered in {self._name}"
                self._obj_map[name] = cls
                return cls
            return deco if obj is None else deco(obj)

        def get(self, name):
            if name not in self._obj_map:
                raise KeyError(f"No object named '{name}' found in '{self._name}' registry!")
            return self._obj_map[name]

        def __contains__(self, name):
            return name in self._obj_map

    META_ARCH_REGISTRY = _Registry("META_ARCH")


def build_model(cfg):
    """detectron2/modeling/meta_arch/build.py:16-25"""
    import torch
    model = META_ARCH_REGISTRY.get(cfg.MODEL.META_ARCHITECTURE)(cfg)
    model.to(torch.device(cfg.MODEL.DEVICE))
    return model


def _ns(**kw):
    return SimpleNamespace(**kw)


def get_seqformer_cfg(**overrides):
    """The key names and defaults of projects/SeqFormer/seqformer/config.py:5-85 (+ the D2 keys the
    meta-arch reads), as attribute namespaces -- yacs is not installed here."""
    cfg = _ns(
        MODEL=_ns(META_ARCHITECTURE="SeqFormer", DEVICE="cuda",
                  PIXEL_MEAN=[123.675, 116.280, 103.530], PIXEL_STD=[58.395, 57.120, 57.375],
                  MASK_ON=True,
                  SeqFormer=_ns(NUM_CLASSES=40, MASK_WEIGHT=2.0, DICE_WEIGHT=5.0, GIOU_WEIGHT=2.0, L1_WEIGHT=5.0,
                                CLASS_WEIGHT=2.0, DEEP_SUPERVISION=True, MASK_STRIDE=4, MATCH_STRIDE=4,
                                FOCAL_ALPHA=0.25, SET_COST_CLASS=2, SET_COST_BOX=5, SET_COST_GIOU=2,
                                NHEADS=8, DROPOUT=0.1, DIM_FEEDFORWARD=1024, ENC_LAYERS=6, DEC_LAYERS=6,
                                HIDDEN_DIM=256, NUM_OBJECT_QUERIES=300, DEC_N_POINTS=4, ENC_N_POINTS=4,
                                NUM_FEATURE_LEVELS=4, MERGE_ON_CPU=True, MULTI_CLS_ON=True,
                                APPLY_CLS_THRES=0.05, CLIP_MATCHING=False, CLIP_LENGTH=5, CLIP_STRIDE=1)),
        INPUT=_ns(SAMPLING_FRAME_NUM=5),
        SOLVER=_ns(OPTIMIZER="ADAMW", BACKBONE_MULTIPLIER=0.1, BASE_LR=2e-4, WEIGHT_DECAY=1e-4),
        FIND_UNUSED_PARAMETERS=True,
    )
    for dotted, v in overrides.items():
        node = cfg
        parts = dotted.split(".")
        for p in parts[:-1]:
            node = getattr(node, p)
        setattr(node, parts[-1], v)
    return cfg


def get_idol_cfg(**overrides):
    """Key names and defaults of projects/IDOL/idol/config.py:9-89 (+ the D2 keys the meta-arch reads)."""
    cfg = _ns(
        MODEL=_ns(META_ARCHITECTURE="IDOL", DEVICE="cuda",
                  PIXEL_MEAN=[123.675, 116.280, 103.530], PIXEL_STD=[58.395, 57.120, 57.375], MASK_ON=True,
                  IDOL=_ns(NUM_CLASSES=40, MASK_WEIGHT=2.0, DICE_WEIGHT=5.0, GIOU_WEIGHT=2.0, L1_WEIGHT=5.0,
                           CLASS_WEIGHT=2.0, REID_WEIGHT=2.0, DEEP_SUPERVISION=True, MASK_STRIDE=4, MATCH_STRIDE=4,
                           FOCAL_ALPHA=0.25, SET_COST_CLASS=2, SET_COST_BOX=5, SET_COST_GIOU=2,
                           NHEADS=8, DROPOUT=0.1, DIM_FEEDFORWARD=1024, ENC_LAYERS=6, DEC_LAYERS=6,
                           HIDDEN_DIM=256, NUM_OBJECT_QUERIES=300, DEC_N_POINTS=4, ENC_N_POINTS=4,
                           NUM_FEATURE_LEVELS=4, CLIP_STRIDE=1, MERGE_ON_CPU=True, MULTI_CLS_ON=True,
                           APPLY_CLS_THRES=0.05, TEMPORAL_SCORE_TYPE="mean", INFERENCE_SELECT_THRES=0.1,
                           NMS_PRE=0.5, ADD_NEW_SCORE=0.2, INFERENCE_FW=True, INFERENCE_TW=True, MEMORY_LEN=3,
                           BATCH_INFER_LEN=10)),
        INPUT=_ns(SAMPLING_FRAME_NUM=2, COCO_PRETRAIN=False),
        DATASETS=_ns(TEST=("ytvis_2019_val",)),
        SOLVER=_ns(OPTIMIZER="ADAMW", BACKBONE_MULTIPLIER=0.1, BASE_LR=1e-4, WEIGHT_DECAY=1e-4),
        FIND_UNUSED_PARAMETERS=True,
    )
    for dotted, v in overrides.items():
        node = cfg
        parts = dotted.split(".")
        for p in parts[:-1]:
            node = getattr(node, p)
        setattr(node, parts[-1], v)
    return cfg
