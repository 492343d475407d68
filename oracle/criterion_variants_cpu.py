"""CPU oracle of the criterion variants of SURVEY 8(f) rank 2  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE (only tests/ import it).

Restates, in plain torch CPU ops + scipy, libs/utils/loss4OL.py:68-232 (`Criterion4OL` of trainOLV2.py / trainOLV3.py /
testOLV3.py, default stageMode=False), libs/utils/loss4OLV2.py:12-186 (`Criterion4OL`, the one-to-many variant) and
libs/utils/dynamic_assign.py:292-357 (`assignOne2Many`).  Shared terms (assignment cost of `assign`, focal vector) come from
oracle/phnet_cpu.py.  Pinned against tests/golden/criterion_variants_tiny.npz, produced by the reference's own classes
(tests/golden/make_goldens_criteria.py; tests/test_oracle_criteria.py)."""
from __future__ import annotations

from typing import List, Sequence, Tuple

import torch
import torch.nn.functional as F

from . import phnet_cpu as O

Tensor = torch.Tensor
INFINITY = 987654.0                       # dynamic_assign.py:3


def line_iou_aligned(pred_px: Tensor, tgt_px: Tensor, img_w: int, radius: float = 15.0) -> Tensor:
    """dynamic_assign.py:5-36 with aligned=True: [m,S],[m,S] -> [m]."""
    ovr = torch.min(pred_px + radius, tgt_px + radius) - torch.max(pred_px - radius, tgt_px - radius)
    uni = torch.max(pred_px + radius, tgt_px + radius) - torch.min(pred_px - radius, tgt_px - radius)
    bad = (tgt_px < 0) | (tgt_px >= img_w)
    ovr = ovr.masked_fill(bad, 0.0)
    uni = uni.masked_fill(bad, 0.0)
    return ovr.sum(-1) / (uni.sum(-1) + 1e-9)


def one2many_cost(pred: Tensor, tgt: Tensor, g: O.Geometry) -> Tuple[Tensor, Tensor]:
    """(C = cost - iou, pairwise iou) of assignOne2Many (dynamic_assign.py:292-337): the `assign` cost with focal alpha 0.5."""
    pred, tgt = pred.detach().clone(), tgt.detach().clone()
    w, h = g.img_w, g.img_h
    pxs = pred[:, 6:] * (w - 1)
    txs = tgt[:, 6:]
    bad = (txs < 0) | (txs >= w)
    d = (txs[None] - pxs[:, None]).abs().masked_fill(bad[None].expand(pxs.shape[0], -1, -1), 0.0)
    dist = d.sum(-1) / ((~bad).sum(1).float() + 1e-9)[None]
    dist = 1 - dist / (dist.max() + 1e-4)
    prob = pred[:, :2].sigmoid()
    eps, alpha, gamma = 1e-12, 0.5, 2
    neg = -(1 - prob + eps).log() * (1 - alpha) * prob.pow(gamma)
    posc = -(prob + eps).log() * alpha * (1 - prob).pow(gamma)
    lab = tgt[:, 1].long()
    cls = posc[:, lab] - neg[:, lab]
    scale = torch.tensor([h - 1.0, w - 1.0])
    start = torch.cdist(pred[:, 2:4] * scale, tgt[:, 2:4] * scale, p=2)
    start = 1 - start / (start.max() + 1e-4)
    theta = torch.cdist(pred[:, 4:5], tgt[:, 4:5], p=1) * 180
    theta = 1 - theta / (theta.max() + 1e-4)
    cost = -(dist * start * theta) ** 2 * 3.0 + cls * 1.0
    iou = O._pairwise_line_iou(pxs, txs, w)
    return cost - iou, iou


def assign_one2many(pred: Tensor, tgt: Tensor, g: O.Geometry) -> Tuple[Tensor, Tensor]:
    """dynamic_assign.py:338-357: k_j = clamp(int(sum of the 4 largest IoUs of column j), min 1) anchors per label; rounds of the
    Hungarian solver over ALL columns, keeping the pairs of the columns that still want anchors and retiring their rows."""
    from scipy.optimize import linear_sum_assignment
    C, iou = one2many_cost(pred, tgt, g)
    iou = iou.clamp(min=0.0)
    ks = torch.clamp(torch.topk(iou, 4, dim=0)[0].sum(0).int(), min=1)
    C = C.clone()
    rows_out, cols_out = [], []
    while int(ks.sum()) > 0:
        r, c = linear_sum_assignment(C.numpy(), maximize=False)
        r, c = torch.as_tensor(r), torch.as_tensor(c)
        keep = ks > 0                                    # as shipped: the PER-COLUMN mask indexes the row-sorted PAIR list (:351)
        r, c = r[keep], c[keep]
        rows_out.append(r); cols_out.append(c)
        ks[ks > 0] -= 1
        C[r, :] = INFINITY
    return torch.cat(rows_out), torch.cat(cols_out)


def _branch_v1(stage_preds: Sequence[Tensor], gt: Tensor, g: O.Geometry):
    """loss4OL.py:88-166, stageMode False, one branch: (matched rows per stage, cls [N], reg [m] or 0, iou [m] or 0)."""
    cls_sum, reg_sum, iou_sum, matched = 0.0, 0.0, 0.0, []
    scale = torch.tensor([g.n_strips, g.img_w - 1.0, 180.0, g.n_strips], dtype=torch.float32)
    for preds in stage_preds:
        pred = preds[0]
        tgt = gt[0][gt[0][:, 1] == 1]
        labels = torch.zeros(pred.shape[0], dtype=torch.long)
        if tgt.shape[0] == 0:
            cls_sum = cls_sum + O.focal_vector(pred[:, :2], labels)
            matched.append(torch.zeros(0, dtype=torch.long))
            continue
        with torch.no_grad():
            rows, cols = O.hungarian(O.assignment_cost(pred, tgt, g))
        matched.append(rows)
        labels[rows] = 1
        cls_sum = cls_sum + O.focal_vector(pred[:, :2], labels)
        m = rows.shape[0]
        reg_sum = reg_sum + F.smooth_l1_loss(pred[rows, 2:6] * scale, tgt[cols, 2:6] * scale, reduction="none").mean(-1) / m
        iou_sum = iou_sum + (1 - line_iou_aligned(pred[rows, 6:] * (g.img_w - 1), tgt[cols, 6:], g.img_w, 15.0)) / m
    k = 1 * g.refine_layers
    return matched, cls_sum / k, reg_sum / k, iou_sum / k


def _inst_loss(rows_last: Tensor, cls: Tensor, reg, iou, g: O.Geometry) -> Tensor:
    """loss4OL.py:168-175: per-anchor loss; the summed per-pair terms land on the anchors matched at the LAST stage."""
    inst = cls * g.cls_weight
    if rows_last.numel() == 0:
        return inst
    add = torch.zeros_like(inst)
    add[rows_last] = reg * g.reg_weight + iou * g.iou_weight
    return inst + add


def frame_loss_v1(fir: Sequence[Tensor], sec: Sequence[Tensor], gates: Sequence[Tensor], gt: Tensor, g: O.Geometry):
    """loss4OL.py:177-232 (stageMode False) -> (matched rows of branch B per stage, scalar loss)."""
    ma, cls_a, reg_a, iou_a = _branch_v1(fir, gt, g)
    mb, cls_b, reg_b, iou_b = _branch_v1(sec, gt, g)
    la = _inst_loss(ma[-1], cls_a, reg_a, iou_a, g)
    lb = _inst_loss(mb[-1], cls_b, reg_b, iou_b, g)
    d = torch.stack(list(gates), dim=0).squeeze().mean(dim=0)
    delta = torch.median(la - lb).detach()
    return mb, torch.sum((1 - d) * (la - delta / 2) + d * (lb + delta / 2))


def _branch_v2(stage_preds: Sequence[Tensor], gt: Tensor, g: O.Geometry):
    """loss4OLV2.py:28-93 (`line_loss_diff_A`): one-to-many matches, scalar regression / IoU terms."""
    cls_sum, reg_sum, iou_sum, matched = 0.0, 0.0, 0.0, []
    scale = torch.tensor([g.n_strips, g.img_w - 1.0, 180.0, g.n_strips], dtype=torch.float32)
    for preds in stage_preds:
        pred = preds[0]
        tgt = gt[0][gt[0][:, 1] == 1]
        labels = torch.zeros(pred.shape[0], dtype=torch.long)
        if tgt.shape[0] == 0:
            cls_sum = cls_sum + O.focal_vector(pred[:, :2], labels)
            matched.append(torch.zeros(0, dtype=torch.long))
            continue
        with torch.no_grad():
            rows, cols = assign_one2many(pred, tgt, g)
        matched.append(rows)
        labels[rows] = 1
        cls_sum = cls_sum + O.focal_vector(pred[:, :2], labels)
        reg_sum = reg_sum + F.smooth_l1_loss(pred[rows, 2:6] * scale, tgt[cols, 2:6] * scale, reduction="none").mean()
        iou_sum = iou_sum + (1 - line_iou_aligned(pred[rows, 6:] * (g.img_w - 1), tgt[cols, 6:], g.img_w, 15.0)).mean()
    k = 1 * len(stage_preds)
    return matched, cls_sum / k, reg_sum / k, iou_sum / k


def frame_loss_v2(fir: Sequence[Tensor], sec: Sequence[Tensor], gates: Sequence[Tensor], gt: Tensor, g: O.Geometry):
    """loss4OLV2.py:152-178 -> (matched rows of branch B per stage, scalar loss, last_priors [1,k,6+S])."""
    _, cls_a, reg_a, iou_a = _branch_v2(fir, gt, g)
    mb, cls_b, reg_b, iou_b = _branch_v2(sec, gt, g)
    d = torch.stack(list(gates), dim=0).squeeze().mean(dim=0)
    delta = torch.median(cls_a - cls_b).detach()
    cls = torch.sum((1 - d) * (cls_a - delta / 2) + d * (cls_b + delta / 2))
    total = (reg_a + reg_b) * g.reg_weight / 2 + (iou_a + iou_b) * g.iou_weight / 2 + cls * g.cls_weight
    return mb, total, sec[-1][:, mb[-1], :]
