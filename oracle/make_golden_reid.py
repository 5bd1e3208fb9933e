"""oracle/make_golden_reid.py -- TEST INFRASTRUCTURE ONLY (fixture generator).

Executes the reference's own `SetCriterion.loss_reid` body
(projects/IDOL/idol/models/deformable_detr.py:418-454, cut out with ast -- the file imports
fvcore / torchvision at the top) and the tracker's similarity expressions
(projects/IDOL/idol/models/tracker.py:229-244, four `match_metric` branches, evaluated with the
same torch expressions) on seeded inputs; stores inputs and outputs in tests/golden/reid_*.npz.

    python -m oracle.make_golden_reid
"""
from __future__ import annotations

import os
import types

import numpy as np
import torch
import torch.nn.functional as F

from oracle.ref_extract import extract

REF = "/root/reference/projects/IDOL/idol/models/deformable_detr.py"
OUT_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def main():
    torch.set_default_dtype(torch.float64)
    loss_reid = extract(REF, ["loss_reid"])["loss_reid"]
    gen = torch.Generator().manual_seed(5)

    # ---- training: one image, R reference queries, I instances --------------------------------
    R, I, C = 40, 5, 256
    ref = torch.randn(R, C, generator=gen) * 0.3
    key = torch.randn(I, C, generator=gen) * 0.3
    pos = torch.rand(R, I, generator=gen) < 0.15
    neg = (~pos) & (torch.rand(R, I, generator=gen) < 0.7)
    pos[:, 3] = False            # an instance without positives (pos_neg_select.py:49-50)
    neg[:, 4] = False
    neg[:3, 4] = True
    aux = pos.clone()
    for i in range(I):           # the reference samples <= 10x negatives for the cosine loss
        cand = torch.nonzero(neg[:, i]).flatten()
        n_pos = int(pos[:, i].sum())
        take = 10 if n_pos == 0 else min(len(cand), n_pos * 10)
        perm = cand[torch.randperm(len(cand), generator=gen)[:take]]
        aux[perm, i] = True
    items = []
    for i in range(I):           # build qd_items exactly as select_pos_neg does (:40-62)
        pe, ne = ref[pos[:, i]], ref[neg[:, i]]
        emb = torch.cat([pe, ne], 0)
        lab = torch.cat([torch.ones(len(pe)), torch.zeros(len(ne))], 0)
        contrast = torch.einsum('nc,kc->nk', [emb, key[i:i + 1]])
        aux_neg = ref[aux[:, i] & ~pos[:, i]]
        aemb = F.normalize(torch.cat([pe, aux_neg], 0), dim=1)
        alab = torch.cat([torch.ones(len(pe)), torch.zeros(len(aux_neg))], 0)
        cosine = torch.einsum('nc,kc->nk', [aemb, F.normalize(key[i:i + 1], dim=1)])
        items.append({'contrast': contrast, 'label': lab, 'aux_consin': cosine, 'aux_label': alab})
    out = loss_reid(types.SimpleNamespace(), {'pred_qd': items, 'pred_logits': torch.zeros(1)}, None, None, None, None)
    np.savez_compressed(os.path.join(OUT_DIR, "reid_loss.npz"), ref=ref.numpy(), key=key.numpy(),
                        pos=pos.numpy(), neg=neg.numpy(), aux=aux.numpy(),
                        loss_reid=out['loss_reid'].numpy(), loss_reid_aux=out['loss_reid_aux'].numpy(),
                        n_items=np.array(len(items)))
    print("loss_reid", float(out['loss_reid']), "aux", float(out['loss_reid_aux']))

    # ---- inference: tracker association scores (tracker.py:229-244) ---------------------------
    for name, (n, k) in (("small", (7, 5)), ("frame", (43, 61)), ("ragged", (17, 300))):
        embeds = torch.randn(n, C, generator=gen) * 0.2
        memo = torch.randn(k, C, generator=gen) * 0.2
        feats = torch.mm(embeds, memo.t())
        d2t, t2d = feats.softmax(dim=1), feats.softmax(dim=0)
        np.savez_compressed(os.path.join(OUT_DIR, f"reid_match_{name}.npz"), embeds=embeds.numpy(),
                            memo=memo.numpy(), longrang=feats.numpy(), bisoftmax=((d2t + t2d) / 2).numpy(),
                            softmax=d2t.numpy(),
                            cosine=torch.mm(F.normalize(embeds, p=2, dim=1), F.normalize(memo, p=2, dim=1).t()).numpy())
        print(name, n, k)


if __name__ == "__main__":
    main()
