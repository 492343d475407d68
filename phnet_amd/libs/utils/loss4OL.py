"""`Criterion4OL` of trainOLV2.py / trainOLV3.py / testOLV3.py behind the reference's API
(`from libs.utils.loss4OL import Criterion4OL`; reference: libs/utils/loss4OL.py:68-232 with dynamic_assign.py:5-36,128-190 and
focal_loss.py:78-136).  Differences to the V3 criterion (libs/utils/loss4OLV3.py): the regression / IoU terms stay per matched
PAIR (smooth-L1 mean over the four start / angle / length values, 1 - line IoU with a fixed 15 px radius, each divided by the number
of pairs), are summed over the stages BY POSITION in the row-sorted pair list and land on the anchors matched at the LAST stage; the
branch balance (median shift, gate-weighted sum) then acts on that whole per-anchor loss vector.

Sync-free like the V3 criterion and fused like it: `phnet_frame_loss_variant` (csrc/loss_variants.hip) computes the assignment
(exact matching, no scipy, no device->host copy), every loss term and every input gradient of a frame in two launches; label
rows are carried with a validity mask, matched anchors come back as fixed-size vectors padded with -1.  `fused = False` spells
the same arithmetic in device tensor ops (the kernels' cross-check).  Only the default `stageMode=False` path of the reference
is built."""
import torch
import torch.nn.functional as F

from phnet_amd import hip_ops as K
from .loss4OLV3 import Criterion4OL as _CriterionV3


def line_iou_rows(pred_px, tgt_px, img_w: float, radius: float = 15.0):
    """dynamic_assign.py:5-36, aligned=True: [L,S] x [L,S] -> [L]; label columns outside the image do not count."""
    ok = ~((tgt_px < 0) | (tgt_px >= img_w))
    ovr = (torch.min(pred_px + radius, tgt_px + radius) - torch.max(pred_px - radius, tgt_px - radius)) * ok
    uni = (torch.max(pred_px + radius, tgt_px + radius) - torch.min(pred_px - radius, tgt_px - radius)) * ok
    return ovr.sum(-1) / (uni.sum(-1) + 1e-9)


class _FusedVariantLoss(torch.autograd.Function):
    """(6 predictions, 3 gates) -> frame loss of criterion variant 1 / 2: phnet_frame_loss_variant computes the value and every
    input gradient in two launches (csrc/loss_variants.hip)."""

    @staticmethod
    def forward(ctx, crit, variant, tgt, *tensors):
        preds = [t.reshape(-1, t.shape[-1]).contiguous() for t in tensors[:6]]
        gates = [t.reshape(-1).contiguous() for t in tensors[6:]]
        loss, dpred, dgate, prow, pcol, srt = K.frame_loss_variant(variant, preds, gates, tgt.contiguous(), crit.img_w, crit.img_h,
                                                                   crit.cls_weight, crit.reg_weight, crit.iou_weight)
        ctx.save_for_backward(dpred, dgate)
        ctx.pshape, ctx.gshape = tensors[0].shape, tensors[6].shape
        ctx.mark_non_differentiable(prow, pcol, srt)
        return loss.view(()), prow, pcol, srt

    @staticmethod
    def backward(ctx, gloss, *_):
        dpred, dgate = ctx.saved_tensors
        dp, dg = dpred * gloss, dgate * gloss
        return (None, None, None, *[dp[i].view(ctx.pshape) for i in range(6)], *[dg[i].view(ctx.gshape) for i in range(3)])


def fusable(targets, fa, fb) -> bool:
    return (targets.shape[0] == 1 and len(fa) == 3 and len(fb) == 3 and targets.shape[1] <= 4 and fa[0].shape[-2] <= 256
            and fa[0].is_cuda)


class Criterion4OL(_CriterionV3):
    def __init__(self, cfg):
        super().__init__(cfg)
        self.num_classes = cfg.max_lanes + 1
        self.stageWeight = [0.5, 1.0, 1.5]

    def line_loss_diff(self, predictions_lists, targets, stageMode=False):
        """One branch, all stages -> (matched [stages] x i64[L] ascending / -1 padded, cls [N], reg [L], iou [L]); reg / iou are
        indexed by POSITION in the row-sorted pair list (positions >= number of pairs hold zeros)."""
        if stageMode:
            raise NotImplementedError("only the default stageMode=False path of loss4OL.py is built (loss4OL.py:177-180)")
        cls_sum, matched = 0.0, []
        L = targets.shape[1]
        reg_pos = targets.new_zeros(L)
        iou_pos = targets.new_zeros(L)
        scale = self._const("scale", [self.n_strips, self.img_w - 1.0, 180.0, self.n_strips], targets)
        ar = torch.arange(L, device=targets.device)
        for preds in predictions_lists:
            for pred, tgt in zip(preds, targets):
                n = pred.shape[0]
                rows, rows_sorted, _ = K.lane_assign(pred.contiguous().detach(), tgt.contiguous(), self.img_w, self.img_h)
                matched.append(rows_sorted)
                valid = rows >= 0
                vf = valid.to(pred.dtype)
                m = vf.sum().clamp(min=1.0)
                safe = rows.clamp(min=0)
                labels = torch.zeros(n, dtype=pred.dtype, device=pred.device).index_put_((safe,), vf, accumulate=True)
                cls_sum = cls_sum + self.focal(pred[:, :2], labels)
                sel = pred[safe]                                                     # [L,6+S], row j = the anchor of label j
                tsel = torch.where(valid[:, None], tgt, sel.detach())                # invalid label rows neutralised (they hold -1e5)
                reg = F.smooth_l1_loss(sel[:, 2:6] * scale, tsel[:, 2:6] * scale, reduction="none").mean(-1) / m
                px = sel[:, 6:] * (self.img_w - 1)
                tpx = torch.where(valid[:, None], tgt[:, 6:], px.detach())
                iou = (1 - line_iou_rows(px, tpx, self.img_w, 15.0)) / m
                order = torch.argsort(torch.where(valid, rows, n + ar))              # pairs ascending by anchor, invalid labels last
                reg_pos = reg_pos + (reg * vf)[order]
                iou_pos = iou_pos + (iou * vf)[order]
        k = len(targets) * self.refine_layers
        return matched, cls_sum / k, reg_pos / k, iou_pos / k

    def CalculateInstLoss(self, matched_row, cls_loss, reg_yxtl_loss, iou_loss):
        """Per-anchor loss [N]: the summed pair terms are added at the anchors matched at the last stage (loss4OL.py:168-175)."""
        inst = cls_loss * self.cls_weight
        ok = (matched_row >= 0).to(inst.dtype)
        add = (reg_yxtl_loss * self.reg_weight + iou_loss * self.iou_weight) * ok
        return inst.index_put((matched_row.clamp(min=0),), add, accumulate=True)

    def loss4OneStep(self, output, batch, diff=None, stageMode=False):
        assert diff is not None
        targets = batch["lane_line"]
        fa, fb = output["predictions_fir"], output["predictions_sec"]
        if self.fused and not stageMode and fusable(targets, fa, fb):
            loss, _, _, srt = _FusedVariantLoss.apply(self, 1, targets[0], *fa, *fb, *diff)
            return [srt[3], srt[4], srt[5]], loss
        return self.loss4OneStep_tensor_ops(output, batch, diff, stageMode)

    def loss4OneStep_tensor_ops(self, output, batch, diff=None, stageMode=False):
        """The same criterion in device tensor ops (~450 launches per frame; the cross-check of the fused kernels)."""
        targets = batch["lane_line"]
        ma, cls_a, reg_a, iou_a = self.line_loss_diff(output["predictions_fir"], targets, stageMode)
        mb, cls_b, reg_b, iou_b = self.line_loss_diff(output["predictions_sec"], targets, stageMode)
        loss_a = self.CalculateInstLoss(ma[-1], cls_a, reg_a, iou_a)
        loss_b = self.CalculateInstLoss(mb[-1], cls_b, reg_b, iou_b)
        d = torch.stack(list(diff), dim=0).squeeze().mean(dim=0)
        delta = torch.median(loss_a - loss_b).detach()
        total = torch.sum((1 - d) * (loss_a - delta / 2) + d * (loss_b + delta / 2))
        return mb, total

    def forward(self, output, gt_lane, diff=None):
        """-> (matched anchors of branch B per stage: i64[L] ascending, -1 padded; scalar loss)."""
        return self.loss4OneStep(output, {"lane_line": gt_lane}, diff)
