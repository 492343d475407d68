"""Training criterion of the two-branch lane head behind the reference's API
(`from libs.utils.loss4OLV3 import Criterion4OL`; reference: libs/utils/loss4OLV3.py:12-123 with
dynamic_assign.py:128-190, focal_loss.py:78-136, dynamic_assignV2.py:55-98).

Round-1 state: the cost matrix / focal / smooth-L1 / LaneIoU arithmetic runs as device tensor ops and the
240 x <=4 assignment is solved on the host exactly as the reference does (scipy Hungarian on a .cpu() copy);
the on-device assignment + fused loss kernels are the "next" rows of SURVEY.md 8(f)."""
import torch
import torch.nn as nn
import torch.nn.functional as F
from scipy.optimize import linear_sum_assignment


def _masked(t, mask):
    return t.masked_fill(mask, 0.0)


class Criterion4OL(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.refine_layers = 3
        self.img_h, self.img_w = cfg.img_h, cfg.img_w
        self.n_strips, self.n_offsets = cfg.num_points - 1, cfg.num_points
        self.cls_weight, self.reg_weight, self.iou_weight = cfg.cls_weight, cfg.reg_weight, cfg.iou_weight
        self.focal_alpha, self.focal_gamma = (0.1, 0.9), 2.0
        # LaneIoULoss() is built with its class defaults, not cfg (dynamic_assignV2.py:56; loss4OLV3.py:28)
        self.liou_half_width, self.liou_img_h, self.liou_img_w = 7.5 / 768, 400, 960

    # ---- label assignment --------------------------------------------------------------------------------
    @torch.no_grad()
    def assignment_cost(self, pred, tgt):
        w, h = self.img_w, self.img_h
        pxs, txs = pred[:, 6:] * (w - 1), tgt[:, 6:]
        bad = (txs < 0) | (txs >= w)
        d = _masked((txs[None] - pxs[:, None]).abs(), bad[None].expand(pxs.shape[0], -1, -1))
        dist = d.sum(-1) / ((~bad).sum(1).float() + 1e-9)[None]
        dist = 1 - dist / (dist.max() + 1e-4)
        prob = pred[:, :2].sigmoid()
        neg = -(1 - prob + 1e-12).log() * 0.75 * prob.pow(2)
        posc = -(prob + 1e-12).log() * 0.25 * (1 - prob).pow(2)
        lab = tgt[:, 1].long()
        cls = posc[:, lab] - neg[:, lab]
        scale = pred.new_tensor([h - 1.0, w - 1.0])
        start = torch.cdist(pred[:, 2:4] * scale, tgt[:, 2:4] * scale, p=2)
        start = 1 - start / (start.max() + 1e-4)
        theta = torch.cdist(pred[:, 4:5], tgt[:, 4:5], p=1) * 180
        theta = 1 - theta / (theta.max() + 1e-4)
        cost = -(dist * start * theta) ** 2 * 3.0 + cls
        lo_p, hi_p = (pxs - 15.0)[:, None], (pxs + 15.0)[:, None]
        lo_t, hi_t = (txs - 15.0)[None], (txs + 15.0)[None]
        ovr = _masked(torch.min(hi_p, hi_t) - torch.max(lo_p, lo_t), bad[None].expand(pxs.shape[0], -1, -1))
        uni = _masked(torch.max(hi_p, hi_t) - torch.min(lo_p, lo_t), bad[None].expand(pxs.shape[0], -1, -1))
        return cost - ovr.sum(-1) / (uni.sum(-1) + 1e-9)

    @staticmethod
    def solve(cost):
        rows, cols = linear_sum_assignment(cost.detach().cpu().numpy(), maximize=False)
        return torch.as_tensor(rows), torch.as_tensor(cols)

    # ---- loss terms ------------------------------------------------------------------------------------------
    def focal(self, logits, labels):
        p = F.softmax(logits, dim=1) + 1e-6
        onehot = F.one_hot(labels, 2).to(logits.dtype) + 1e-6
        focal = -logits.new_tensor(self.focal_alpha) * torch.pow(1.0 - p, self.focal_gamma) * torch.log(p)
        return (onehot * focal).sum(dim=1)

    def lane_iou(self, pred, tgt):
        dy = self.liou_img_h / (pred.shape[1] - 1) * 2
        pd = (pred[:, 2:] - pred[:, :-2]).detach() * self.liou_img_w
        pw = self.liou_half_width * torch.sqrt(pd.pow(2) + dy ** 2) / dy
        pw = torch.cat([pw[:, :1], pw, pw[:, -1:]], dim=1)
        td = (tgt[:, 2:] - tgt[:, :-2]) * self.liou_img_w
        td = torch.where(td.abs() > 1e4, torch.zeros_like(td), td)
        tw = self.liou_half_width * torch.sqrt(td.pow(2) + dy ** 2) / dy
        tw = torch.cat([tw[:, :1], tw, tw[:, -1:]], dim=1)
        bad = (tgt < 0) | (tgt >= 1.0)
        ovr = _masked(torch.min(pred + pw, tgt + tw) - torch.max(pred - pw, tgt - tw), bad)
        uni = _masked(torch.max(pred + pw, tgt + tw) - torch.min(pred - pw, tgt - tw), bad)
        return (1 - ovr.sum(-1) / (uni.sum(-1) + 1e-9)).mean()

    def line_loss_diff(self, predictions_lists, targets):
        cls_sum, reg_sum, iou_sum, matched = 0.0, 0.0, 0.0, []
        scale = targets.new_tensor([self.n_strips, self.img_w - 1.0, 180.0, self.n_strips])
        for preds in predictions_lists:
            for pred, target in zip(preds, targets):
                tgt = target[target[:, 1] == 1]
                labels = torch.zeros(pred.shape[0], dtype=torch.long, device=pred.device)
                if tgt.shape[0] == 0:
                    cls_sum = cls_sum + self.focal(pred[:, :2], labels)
                    matched.append([])
                    continue
                rows, cols = self.solve(self.assignment_cost(pred.detach(), tgt))
                matched.append(rows)
                rows_d, cols_d = rows.to(pred.device), cols.to(pred.device)
                labels[rows_d] = 1
                cls_sum = cls_sum + self.focal(pred[:, :2], labels)
                reg_sum = reg_sum + F.smooth_l1_loss(pred[rows_d, 2:6] * scale, tgt[cols_d, 2:6] * scale, reduction="none").mean()
                iou_sum = iou_sum + self.lane_iou(pred[rows_d, 6:] * (self.img_w - 1) / self.img_w, tgt[cols_d, 6:] / self.img_w)
        k = len(targets) * self.refine_layers
        return matched, cls_sum / k, reg_sum / k, iou_sum / k

    def loss4OneStep(self, output, batch, diff=None):
        assert diff is not None
        targets = batch["lane_line"]
        _, cls_a, reg_a, iou_a = self.line_loss_diff(output["predictions_fir"], targets)
        matched_b, cls_b, reg_b, iou_b = self.line_loss_diff(output["predictions_sec"], targets)
        d = torch.stack(list(diff), dim=0).squeeze().mean(dim=0)
        delta = torch.median(cls_a - cls_b).detach()
        cls = torch.sum((1 - d) * (cls_a - delta / 2) + d * (cls_b + delta / 2))
        total = (reg_a + reg_b) * self.reg_weight + (iou_a + iou_b) * self.iou_weight + cls * self.cls_weight
        return matched_b, total

    def forward(self, output, gt_lane, diff=None):
        return self.loss4OneStep(output, {"lane_line": gt_lane}, diff)
