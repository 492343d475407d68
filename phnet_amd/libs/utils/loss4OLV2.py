"""The one-to-many variant of the criterion behind the reference's API (`from libs.utils.loss4OLV2 import Criterion4OL`;
reference: libs/utils/loss4OLV2.py:12-186 with dynamic_assign.py:292-357 `assignOne2Many`).  Every label takes up to four anchors
(rounds of the exact matching on device: `phnet_lane_assign_one2many`, one launch, no scipy, no host copy); classification is a
per-anchor focal vector balanced between the branches as in V3, regression / IoU are means over all pairs.  Both branches go through
`line_loss_diff_A`, as in the reference's `loss4OneStep` (:155-156; its `line_loss_diff_B` / cross-frame terms are dead code there).
Returns (matched, loss, last_priors) like the reference; the pair lists are fixed-size (16) and padded with -1, `last_priors` is
[B,16,6+S] with ZERO rows where the list is padded.  Fused: `phnet_frame_loss_variant` (two launches per frame, csrc/loss_variants.hip)."""
import torch
import torch.nn.functional as F

from phnet_amd import hip_ops as K
from .loss4OL import _FusedVariantLoss, fusable, line_iou_rows
from .loss4OLV3 import Criterion4OL as _CriterionV3


class Criterion4OL(_CriterionV3):
    def __init__(self, cfg):
        super().__init__(cfg)
        self.num_classes = cfg.max_lanes + 1
        self.last_target = None

    def line_loss_diff_A(self, predictions_lists, targets):
        """-> (matched [stages] x i64[16] in the reference's pair order / -1 padded, cls [N], reg, iou)."""
        cls_sum, reg_sum, iou_sum, matched = 0.0, 0.0, 0.0, []
        scale = self._const("scale", [self.n_strips, self.img_w - 1.0, 180.0, self.n_strips], targets)
        for preds in predictions_lists:
            for pred, tgt in zip(preds, targets):
                n = pred.shape[0]
                rows, cols, _ = K.lane_assign_one2many(pred.contiguous().detach(), tgt.contiguous(), self.img_w, self.img_h)
                matched.append(rows)
                valid = rows >= 0
                vf = valid.to(pred.dtype)
                m = vf.sum().clamp(min=1.0)
                safe = rows.clamp(min=0)
                labels = torch.zeros(n, dtype=pred.dtype, device=pred.device).index_put_((safe,), vf, accumulate=True)
                cls_sum = cls_sum + self.focal(pred[:, :2], labels)
                sel = pred[safe]                                                     # [16,6+S]
                tlab = tgt[cols.clamp(min=0)]
                tsel = torch.where(valid[:, None], tlab, sel.detach())
                reg = F.smooth_l1_loss(sel[:, 2:6] * scale, tsel[:, 2:6] * scale, reduction="none")
                reg_sum = reg_sum + (reg * vf[:, None]).sum() / (m * 4)
                px = sel[:, 6:] * (self.img_w - 1)
                tpx = torch.where(valid[:, None], tlab[:, 6:], px.detach())
                iou_sum = iou_sum + ((1 - line_iou_rows(px, tpx, self.img_w, 15.0)) * vf).sum() / m
        k = len(targets) * len(predictions_lists)
        return matched, cls_sum / k, reg_sum / k, iou_sum / k

    @staticmethod
    def last_priors(pred_last, rows):
        """predictions_sec[-1] at the last stage's matched anchors as a FIXED-SIZE [B,16,6+S] tensor: row p belongs to pair p where
        rows[p] >= 0 and is ZERO where the pair list is padded (the reference returns only the matched rows: `last[:, rows >= 0]`)."""
        valid = (rows >= 0)
        return pred_last[:, rows.clamp(min=0), :] * valid[None, :, None].to(pred_last.dtype)

    def loss4OneStep(self, output, batch, diff=None):
        assert diff is not None
        targets = batch["lane_line"]
        fa, fb = output["predictions_fir"], output["predictions_sec"]
        if self.fused and fusable(targets, fa, fb):
            loss, prow, _, _ = _FusedVariantLoss.apply(self, 2, targets[0], *fa, *fb, *diff)
            mb = [prow[3], prow[4], prow[5]]
            return mb, loss, self.last_priors(fb[-1], mb[-1])
        return self.loss4OneStep_tensor_ops(output, batch, diff)

    def loss4OneStep_tensor_ops(self, output, batch, diff=None):
        """The same criterion in device tensor ops (the cross-check of the fused kernels)."""
        targets = batch["lane_line"]
        _, cls_a, reg_a, iou_a = self.line_loss_diff_A(output["predictions_fir"], targets)
        mb, cls_b, reg_b, iou_b = self.line_loss_diff_A(output["predictions_sec"], targets)
        d = torch.stack(list(diff), dim=0).squeeze().mean(dim=0)
        delta = torch.median(cls_a - cls_b).detach()
        cls = torch.sum((1 - d) * (cls_a - delta / 2) + d * (cls_b + delta / 2))
        total = (reg_a + reg_b) * self.reg_weight / 2 + (iou_a + iou_b) * self.iou_weight / 2 + cls * self.cls_weight
        return mb, total, self.last_priors(output["predictions_sec"][-1], mb[-1])

    def forward(self, output, gt_lane, diff=None):
        return self.loss4OneStep(output, {"lane_line": gt_lane}, diff)
