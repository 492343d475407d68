"""Training criterion of the two-branch lane head behind the reference's API
(`from libs.utils.loss4OLV3 import Criterion4OL`; reference: libs/utils/loss4OLV3.py:12-123 with
dynamic_assign.py:128-190, focal_loss.py:78-136, dynamic_assignV2.py:55-98).

Sync-free and shape-static: the label assignment (cost matrix + exact matching) is one HIP launch
(`phnet_lane_assign`), all label rows are carried with a validity mask instead of being filtered, and the matched
anchors come back as a fixed-size device vector padded with -1.  Nothing in the criterion reads a device value on the
host, so a whole training step can be captured in a hipGraph.  The focal / smooth-L1 / LaneIoU arithmetic is still
expressed as device tensor ops."""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from phnet_amd import hip_ops as K


class _FusedFrameLoss(torch.autograd.Function):
    """(6 predictions, 3 gates) -> frame loss: phnet_frame_loss computes the value and every input gradient in two launches."""

    @staticmethod
    def forward(ctx, crit, tgt, *tensors):
        preds = [t.reshape(-1, t.shape[-1]).contiguous() for t in tensors[:6]]
        gates = [t.reshape(-1).contiguous() for t in tensors[6:]]
        loss, dpred, dgate, _, rows_sorted = K.frame_loss(
            preds, gates, tgt.contiguous(), crit.img_w, crit.img_h, crit.cls_weight, crit.reg_weight, crit.iou_weight,
            crit.liou_half_width, crit.liou_img_h, crit.liou_img_w)
        ctx.save_for_backward(dpred, dgate)
        ctx.pshape, ctx.gshape = tensors[0].shape, tensors[6].shape
        ctx.mark_non_differentiable(rows_sorted)
        return loss.view(()), rows_sorted

    @staticmethod
    def backward(ctx, gloss, _rows):
        dpred, dgate = ctx.saved_tensors
        dp, dg = dpred * gloss, dgate * gloss
        return (None, None, *[dp[i].view(ctx.pshape) for i in range(6)], *[dg[i].view(ctx.gshape) for i in range(3)])


class _FusedClipLoss(torch.autograd.Function):
    """T frames x (6 predictions, 3 gates) -> (frame losses [T], matched rows [T,6,L]): phnet_clip_loss, two launches for the whole
    clip; the upstream gradients scale all of its gradients in two launches."""

    @staticmethod
    def forward(ctx, crit, tgt, T, *tensors):
        per = [tensors[9 * t:9 * t + 9] for t in range(T)]
        preds = [[x.reshape(-1, x.shape[-1]).contiguous() for x in fr[:6]] for fr in per]
        gates = [[x.reshape(-1).contiguous() for x in fr[6:]] for fr in per]
        loss, dpred, dgate, _, rows_sorted = K.clip_loss(
            preds, gates, tgt.contiguous(), crit.img_w, crit.img_h, crit.cls_weight, crit.reg_weight, crit.iou_weight,
            crit.liou_half_width, crit.liou_img_h, crit.liou_img_w)
        ctx.save_for_backward(dpred, dgate)
        ctx.T, ctx.pshape, ctx.gshape = T, tensors[0].shape, tensors[6].shape
        ctx.mark_non_differentiable(rows_sorted)
        return loss, rows_sorted

    @staticmethod
    def backward(ctx, gloss, _rows):
        dpred, dgate = ctx.saved_tensors
        dp, dg = dpred * gloss.view(-1, 1, 1, 1), dgate * gloss.view(-1, 1, 1)
        out = []
        for t in range(ctx.T):
            out += [dp[t, i].view(ctx.pshape) for i in range(6)] + [dg[t, i].view(ctx.gshape) for i in range(3)]
        return (None, None, None, *out)


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
        self._consts = {}
        self.fused = True          # phnet_frame_loss (2 launches / frame); False -> tensor-op spelling
        self.frame_observer = None # callable(output, gt_lane, diff, matched, loss) per frame of a clip_loss call
        self.clip_fused = os.environ.get("PHNET_CLIP_LOSS", "1") != "0"     # clip_loss: the frames of a clip in one pair of launches (A/B switch)

    def _const(self, key, values, like):
        k = (key, like.device)
        if k not in self._consts:
            self._consts[k] = torch.tensor(values, dtype=torch.float32, device=like.device)
        return self._consts[k]

    # ---- loss terms ------------------------------------------------------------------------------------------
    def focal(self, logits, labels):
        p = F.softmax(logits, dim=1) + 1e-6
        onehot = torch.stack([1.0 - labels, labels], dim=1) + 1e-6
        focal = -self._const("alpha", self.focal_alpha, logits) * torch.pow(1.0 - p, self.focal_gamma) * torch.log(p)
        return (onehot * focal).sum(dim=1)

    def lane_iou_rows(self, pred, tgt):
        """1 - LaneIoU per row ([L])."""
        dy = self.liou_img_h / (pred.shape[1] - 1) * 2
        pd = (pred[:, 2:] - pred[:, :-2]).detach() * self.liou_img_w
        pw = self.liou_half_width * torch.sqrt(pd.pow(2) + dy ** 2) / dy
        pw = torch.cat([pw[:, :1], pw, pw[:, -1:]], dim=1)
        td = (tgt[:, 2:] - tgt[:, :-2]) * self.liou_img_w
        td = torch.where(td.abs() > 1e4, torch.zeros_like(td), td)
        tw = self.liou_half_width * torch.sqrt(td.pow(2) + dy ** 2) / dy
        tw = torch.cat([tw[:, :1], tw, tw[:, -1:]], dim=1)
        ok = ~((tgt < 0) | (tgt >= 1.0))
        ovr = (torch.min(pred + pw, tgt + tw) - torch.max(pred - pw, tgt - tw)) * ok
        uni = (torch.max(pred + pw, tgt + tw) - torch.min(pred - pw, tgt - tw)) * ok
        return 1 - ovr.sum(-1) / (uni.sum(-1) + 1e-9)

    def line_loss_diff(self, predictions_lists, targets):
        """One branch, all stages.  Returns (matched [stages] x i64[L] ascending / -1 padded, cls[N], reg, iou)."""
        cls_sum, reg_sum, iou_sum, matched = 0.0, 0.0, 0.0, []
        scale = self._const("scale", [self.n_strips, self.img_w - 1.0, 180.0, self.n_strips], targets)
        for preds in predictions_lists:
            for pred, tgt in zip(preds, targets):
                pred_c = pred.contiguous()
                rows, rows_sorted, _ = K.lane_assign(pred_c.detach(), tgt.contiguous(), self.img_w, self.img_h)
                matched.append(rows_sorted)
                valid = (rows >= 0)
                vf = valid.to(pred.dtype)
                m = vf.sum()
                safe = rows.clamp(min=0)
                labels = torch.zeros(pred.shape[0], dtype=pred.dtype, device=pred.device).index_put_((safe,), vf, accumulate=True)
                cls_sum = cls_sum + self.focal(pred[:, :2], labels)
                sel = pred[safe]                                                     # [L, 6+S]
                # invalid label rows hold -1e5 everywhere: neutralise them BEFORE the arithmetic so no inf/nan appears
                tsel = torch.where(valid[:, None], tgt, sel.detach())
                reg = F.smooth_l1_loss(sel[:, 2:6] * scale, tsel[:, 2:6] * scale, reduction="none")
                reg_sum = reg_sum + (reg * vf[:, None]).sum() / (m.clamp(min=1.0) * 4)
                liou = self.lane_iou_rows(sel[:, 6:] * (self.img_w - 1) / self.img_w, tsel[:, 6:] / self.img_w)
                iou_sum = iou_sum + (liou * vf).sum() / m.clamp(min=1.0)
        k = len(targets) * self.refine_layers
        return matched, cls_sum / k, reg_sum / k, iou_sum / k

    def loss4OneStep(self, output, batch, diff=None):
        assert diff is not None
        targets = batch["lane_line"]
        fa, fb = output["predictions_fir"], output["predictions_sec"]
        if (self.fused and targets.shape[0] == 1 and len(fa) == 3 and len(fb) == 3 and targets.shape[1] <= 4
                and fa[0].shape[-2] <= 256 and fa[0].is_cuda):
            loss, rows_sorted = _FusedFrameLoss.apply(self, targets[0], *fa, *fb, *diff)
            return [rows_sorted[3], rows_sorted[4], rows_sorted[5]], loss
        return self.loss4OneStep_tensor_ops(output, batch, diff)

    def loss4OneStep_tensor_ops(self, output, batch, diff=None):
        """The same criterion spelled out in device tensor ops (used as the cross-check of the fused kernel)."""
        targets = batch["lane_line"]
        _, cls_a, reg_a, iou_a = self.line_loss_diff(output["predictions_fir"], targets)
        matched_b, cls_b, reg_b, iou_b = self.line_loss_diff(output["predictions_sec"], targets)
        d = torch.stack(list(diff), dim=0).squeeze().mean(dim=0)
        delta = torch.median(cls_a - cls_b).detach()
        cls = torch.sum((1 - d) * (cls_a - delta / 2) + d * (cls_b + delta / 2))
        total = (reg_a + reg_b) * self.reg_weight + (iou_a + iou_b) * self.iou_weight + cls * self.cls_weight
        return matched_b, total

    def clip_loss(self, outputs, gt_lanes, diffs):
        """Sum of `forward(outputs[t], gt_lanes[t:t+1], diffs[t])[1]` over the frames of a clip - the caller's loop of
        trainOLV3.py:150-171 - in two launches when the fused kernels apply (this class's own criterion, <= 8 frames), else frame by
        frame.  outputs: list over frames of {"predictions_fir": 3 x [1,N,6+S], "predictions_sec": 3 x ...}; gt_lanes [T,L,6+S];
        diffs: list over frames of the 3 gate tensors."""
        T = len(outputs)
        fa, fb = outputs[0]["predictions_fir"], outputs[0]["predictions_sec"]
        own = type(self).loss4OneStep is Criterion4OL.loss4OneStep
        if (own and self.fused and self.clip_fused and 1 <= T <= 8 and len(fa) == 3 and len(fb) == 3 and gt_lanes.shape[1] <= 4 and fa[0].shape[-2] <= 256
                and fa[0].is_cuda and fa[0].shape[0] == 1):
            flat = []
            for t in range(T):
                flat += [*outputs[t]["predictions_fir"], *outputs[t]["predictions_sec"], *diffs[t]]
            loss, rows = _FusedClipLoss.apply(self, gt_lanes, T, *flat)
            if self.frame_observer is not None:            # what a per-frame call would have returned (tests, logging)
                for t in range(T):
                    self.frame_observer(outputs[t], gt_lanes[t:t + 1], diffs[t], [rows[t, 3], rows[t, 4], rows[t, 5]], loss[t])
            return loss.sum()
        total = 0.0
        for t in range(T):
            total = total + self(outputs[t], gt_lanes[t:t + 1], diffs[t])[1]
        return total

    def forward(self, output, gt_lane, diff=None):
        """-> (matched anchors of branch B per stage: i64[L] ascending, padded with -1; scalar loss).
        The reference returns variable-length index tensors; `matched[s][matched[s] >= 0]` recovers them."""
        return self.loss4OneStep(output, {"lane_line": gt_lane}, diff)
