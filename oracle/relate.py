"""CPU restatement of `PoseRelationModule` (`PR/models/regressor_relate.py:9-139`) -- TEST INFRASTRUCTURE ONLY
(imported by tests/ and oracle/gen_golden.py; never by the product).

The reference's only cross-player operator (used at `PR/PyTorch3DTest.py:3526-3534,3758-3759`
between rot6d and the fit when `use_relate=True`): 27 relation units over the players of ONE frame, appearance
feature = the flattened 24x3x3 pose (216), geometry feature = an 8-d sin/cos embedding of pairwise box deltas.

Pin status: pinned.  `tests/golden/relate.npz` is produced by the reference classes themselves
(`PoseRelationModule` with hash-seeded weights; the one `.cuda()` call inside `__PositionalEmbedding`, :63, is
made a no-op for the CPU run by the generator).

Quirk kept as is (:133-138): `output = w_mn.view(N,N,1) * w_v.view(N,1,-1)` multiplies the attention row of
player m with the VALUE OF PLAYER m ITSELF (not of player n), so after `sum(-2)` a unit's output is
`(sum_n softmax_mn) * V_m`: the soft-max weights only enter through their (rounded) row sum."""
import math

import numpy as np
import torch

N_REL, DIM_K, DIM_G, DIM_A = 27, 8, 8, 216


def positional_embedding(boxes, dim_g=DIM_G, wave_len=1000.0):
    """`PoseRelationModule.__PositionalEmbedding` (:34-74): boxes [N,4] = (x_min, y_min, x_max, y_max) -> [N,N,8]."""
    f = torch.as_tensor(boxes, dtype=torch.float32)
    x_min, y_min, x_max, y_max = torch.chunk(f, 4, dim=1)
    cx, cy = (x_min + x_max) * 0.5, (y_min + y_max) * 0.5
    w, h = (x_max - x_min) + 1.0, (y_max - y_min) + 1.0
    dx = torch.log(torch.clamp(torch.abs((cx - cx.view(1, -1)) / w), min=1e-3))
    dy = torch.log(torch.clamp(torch.abs((cy - cy.view(1, -1)) / h), min=1e-3))
    dw = torch.log(w / w.view(1, -1))
    dh = torch.log(h / h.view(1, -1))
    n = dx.shape[0]
    pos = torch.stack((dx, dy, dw, dh), dim=-1)                         # [N,N,4]
    feat_range = torch.arange(dim_g / 8)
    dim_mat = 1.0 / torch.pow(torch.tensor(wave_len), feat_range / (dim_g / 8))
    mul = (100.0 * pos.view(n, n, 4, -1) * dim_mat.view(1, 1, 1, -1)).view(n, n, -1)
    return torch.cat((torch.sin(mul), torch.cos(mul)), dim=-1)


def relation_unit(sd, prefix, f_a, emb):
    """`RelationUnit.forward` (:109-139)."""
    n = f_a.shape[0]
    lin = lambda name, x: x @ torch.as_tensor(sd[f'{prefix}.{name}.weight']).t() + torch.as_tensor(sd[f'{prefix}.{name}.bias'])
    w_g = torch.relu(lin('WG', emb.view(-1, DIM_G))).view(n, n)
    w_k, w_q, w_v = lin('WK', f_a), lin('WQ', f_a), lin('WV', f_a)
    scaled = (w_k.view(n, 1, DIM_K) * w_q.view(1, n, DIM_K)).sum(-1) / np.sqrt(DIM_K)
    w_mn = torch.softmax(torch.log(torch.clamp(w_g, min=1e-6)) + scaled, dim=1)
    return (w_mn.view(n, n, 1) * w_v.view(n, 1, -1)).sum(-2)


def pose_relation_forward(sd, pose_rotmats, boxes):
    """`PoseRelationModule.forward` (:26-32) + `RelationModule.forward` (:86-97): [N,24,3,3], [N,4] -> [N,24,3,3]."""
    f_a = torch.as_tensor(pose_rotmats, dtype=torch.float32).reshape(-1, DIM_A)
    emb = positional_embedding(boxes)
    cat = torch.cat([relation_unit(sd, f'relation_module.relation.{r}', f_a, emb) for r in range(N_REL)], dim=-1)
    return (cat + f_a).view(-1, 24, 3, 3)
