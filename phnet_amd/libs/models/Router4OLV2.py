"""Router4OLV2 model family on the MI355X HIP kernels - the model `testOLV3.py` imports.

Drop-in for `libs.models.Router4OLV2` (reference: libs/models/Router4OLV2.py): same class names (`Encoder`, `RouterV2`,
`RouterOL`), constructor signatures, state_dict keys / shapes / order (306 entries for ResNet-18, frozen against the reference
in tests/golden/state_keys_v2.json) and `forward(inputs: dict)` contract.  INFERENCE ONLY, like the reference in practice: its
training path cannot run as shipped (the model returns `predictions_lists`, Router4OLV2.py:283, the criterion reads
`predictions_fir`, libs/utils/loss4OL.py:177), so `model.train()` + forward raises here.

What differs from the V1 family (libs/models/Router4OL.py) and how it is run:
  * per-level feature widths 64 / 32 / 16 with 24 / 48 / 96 sample points, 72 x-offsets, hidden width 256: ROI pooling
    `phnet_roi_pool_fwd` with C < 64, per-anchor products `phnet_dyn_bmm_ln_relu_fwd_any` (csrc/v2head.hip);
  * gate = Conv1d + BatchNorm1d stack (`AdaptiveRouter4LaneV2`): one launch, `phnet_gate_v2_fwd`;
  * branch B input = content + sinusoidal table (not a concatenation), decoder of width 256 (8 heads x 32:
    `phnet_attention_fwd` with E = 32 H); frames without memory (the first `save_freq`) attend to their OWN tokens
    (Router4OLV2.py:320-325);
  * two towers per branch (cls, reg); `reg_layers` emits (3 start/angle deltas, length, 72 offsets);
  * eval output = HARD routing `torch.where(mean gate >= 0.5, branch B, branch A)` (:508-511): `phnet_route_lines`;
  * `saveMemory4Test` writes its positive mask into a temporary (`mask[keep_inds][keep] = True`, :574), so the memory of a
    frame is ONE token per stage - the mean over all 240 anchors.  Reproduced as is (`faithful_memory = True`); the
    evidently intended behaviour (kept lanes' tokens + mean of the rest, as in Router4OL.py:563-584) is one flag away.
The whole clip runs without a host synchronisation (fused device-side decode + NMS), one device->host copy at the end.
"""
import math

import torch
import torch.nn as nn

from phnet_amd import functional as PF
from phnet_amd import hip_ops as K
from phnet_amd.trunk import encoder_fwd_v2
from ..ops import nms
from ..utils.lane import Lane
from .fpnV2 import FPN
from .resnet import ResNetWrapper
from .Router import AdaptiveRouter4LaneV2
from .Router4OL import DetNetV2, LinearModule
from .utils.dynamic_head import DynamicConvV2
from .utils.transformer import TransformerDecoder, TransformerDecoderLayer

_INFERENCE_ONLY = ("the Router4OLV2 family is inference-only: the reference's training path cannot run as shipped "
                   "(Router4OLV2.py:283 returns `predictions_lists`, libs/utils/loss4OL.py:177 reads `predictions_fir`)")


class PositionalEncoding(nn.Module):
    """Sinusoidal table as a buffer `pos_table` [n_position, d_hid] (libs/models/SeqFormer/position_encoding.py:61-86,
    normalize=False): interleaved sin / cos of position / temperature^(2*(i//2)/d)."""

    def __init__(self, d_hid=64, n_position=240, temperature=10000, normalize=False, scale=None):
        super().__init__()
        if normalize or scale is not None:
            raise NotImplementedError("Router4OLV2.py:105-106 uses normalize=False")
        pos = torch.arange(n_position, dtype=torch.float32)
        dim_t = torch.arange(d_hid, dtype=torch.float32)
        dim_t = temperature ** (2 * (torch.div(dim_t, 2, rounding_mode="floor")) / d_hid)
        tab = pos[..., None] / dim_t
        tab[:, 0::2] = tab[:, 0::2].sin()
        tab[:, 1::2] = tab[:, 1::2].cos()
        self.register_buffer("pos_table", tab)

    def forward(self, x):                                            # [N,B,C] -> the table, repeated over the batch
        return self.pos_table.unsqueeze(1).repeat(1, x.shape[1], 1)


class Encoder(nn.Module):
    """ResNet trunk without its last stage + per-level-width FPN; returns three NHWC levels [T,h,w,C_l], fine -> coarse."""

    def __init__(self, cfg):
        super().__init__()
        self.backbone = ResNetWrapper(**cfg.backbone)
        self.neck = FPN(**cfg.neck) if cfg.haskey("neck") else None
        if self.neck is None:
            raise NotImplementedError("the Router4OLV2 family needs the fpnV2 neck (options4OLV3.py:59-64)")

    def forward(self, batch):
        if self.training:
            raise NotImplementedError(_INFERENCE_ONLY)
        frames = batch["img"] if isinstance(batch, dict) else batch
        with torch.no_grad():
            return encoder_fwd_v2(self, frames)


class RouterV2(nn.Module):
    """Lane head of the V2 family (reference RouterV2, Router4OLV2.py:34-286)."""

    def __init__(self, prior_feat_channels=(64, 32, 16), reg_hidden_dim=256, num_fc=2, refine_layers=3,
                 sample_points=(24, 48, 96), cfg=None):
        super().__init__()
        self.cfg = cfg
        self.img_w, self.img_h = cfg.img_w, cfg.img_h
        self.n_strips, self.n_offsets = cfg.num_points - 1, cfg.num_points
        self.num_priors = cfg.num_priors
        self.prior_feat_channels, self.sample_points = list(prior_feat_channels), list(sample_points)
        self.refine_layers, self.reg_hidden_dim = refine_layers, reg_hidden_dim
        for stage, sp in enumerate(self.sample_points):
            idx = (torch.linspace(0, 1, steps=sp, dtype=torch.float32) * self.n_strips).long()
            self.register_buffer(f"sample_x_indexs_{stage}", idx)
            self.register_buffer(f"prior_feat_ys_{stage}", torch.flip(1 - idx.float() / self.n_strips, dims=[-1]))
        self.register_buffer("prior_ys", torch.linspace(1, 0, steps=self.n_offsets, dtype=torch.float32))
        self.prior_embeddings = nn.Embedding(self.num_priors, 3)
        with torch.no_grad():
            self.prior_embeddings.weight.copy_(DetNetV2._initial_anchors(self))          # the same hand-placed anchors (:180-222)
            pri, on_map = self.generate_priors_from_embeddings()
        self.register_buffer("priors", pri)
        self.register_buffer("priors_on_featmap", on_map)

        def tower():
            mods = []
            for _ in range(num_fc):
                mods += [*LinearModule(reg_hidden_dim)]
            return nn.ModuleList(mods)
        e = reg_hidden_dim
        self.reg_modules, self.cls_modules = tower(), tower()
        self.reg_layers, self.cls_layers = nn.Linear(e, self.n_offsets + 4), nn.Linear(e, 2)
        self.reg_modules_sec, self.cls_modules_sec = tower(), tower()
        self.reg_layers_sec, self.cls_layers_sec = nn.Linear(e, self.n_offsets + 4), nn.Linear(e, 2)
        for lin in (self.cls_layers, self.reg_layers, self.cls_layers_sec, self.reg_layers_sec):
            for p in lin.parameters():
                nn.init.normal_(p, mean=0., std=1e-3)
        layer = TransformerDecoderLayer(d_model=e, nhead=8, dim_feedforward=512, dropout=0.1, activation="gelu", normalize_before=True)
        self.transformer_Dec = TransformerDecoder(layer, 2, nn.LayerNorm(e))
        self.PositionEmbedding = PositionalEncoding(d_hid=e, n_position=self.num_priors, temperature=64, normalize=False)
        self.DHead_series = nn.ModuleList(DynamicConvV2(feat_size=self.sample_points[s], inplanes=self.prior_feat_channels[s],
                                                        outplanes=e, early_return=False) for s in range(refine_layers))
        self.pro_embedding = nn.Embedding(self.num_priors, e)
        self.router = AdaptiveRouter4LaneV2(num_priors=self.num_priors, features_channels=self.prior_feat_channels,
                                            num_points=self.sample_points, out_channels=1, reduction=4, stages=refine_layers)
        self._branch_cache = {}

    # ---- anchors (the V1 formulas; stage-0 sample columns) -------------------------------------------------------
    def _line_xs(self, sy, sx, theta):
        return (sx * (self.img_w - 1) + ((1 - self.prior_ys - sy) * self.img_h / torch.tan(theta * math.pi + 1e-5))) / (self.img_w - 1)

    def generate_priors_from_embeddings(self):
        emb = self.prior_embeddings.weight
        xs = self._line_xs(emb[:, 0:1], emb[:, 1:2], emb[:, 2:3])
        z = emb.new_zeros(emb.shape[0], 1)
        pri = torch.cat([z, z, emb, z, xs], dim=1)
        return pri, pri[:, 6 + self.sample_x_indexs_0]

    # ---- branches -----------------------------------------------------------------------------------------------
    def _branch_weights(self, sec: bool):
        """The two towers of a branch as one 3-GEMM chain (layer 1 concatenated [2E,E], layer 2 block-diagonal [2E,2E], heads
        block-structured [(2 + 4 + S) -> pad 4, 2E]): the head output row is (cls 2 | start/angle deltas 3, length | S offsets),
        the layout `phnet_lane_update_fwd` takes.  Kept until a parameter changes."""
        s = "_sec" if sec else ""
        g = lambda name: getattr(self, name + s)                                         # noqa: E731
        tw, hd = [g("cls_modules"), g("reg_modules")], [g("cls_layers"), g("reg_layers")]
        params = [p for t in tw for p in (t[0].weight, t[0].bias, t[2].weight, t[2].bias)] + [p for h in hd for p in (h.weight, h.bias)]
        ver = tuple((p._version, p.data_ptr()) for p in params)
        hit = self._branch_cache.get(sec)
        if hit is None or hit[0] != ver:
            with torch.no_grad():
                w1 = torch.cat([t[0].weight for t in tw], dim=0)
                b1 = torch.cat([t[0].bias for t in tw], dim=0)
                w2 = torch.block_diag(*[t[2].weight for t in tw])
                b2 = torch.cat([t[2].bias for t in tw], dim=0)
                wh = torch.block_diag(*[h.weight for h in hd])
                bh = torch.cat([h.bias for h in hd], dim=0)
                pad = (-wh.shape[0]) % 4
                if pad:
                    wh = torch.cat([wh, wh.new_zeros(pad, wh.shape[1])], dim=0)
                    bh = torch.cat([bh, bh.new_zeros(pad)], dim=0)
            hit = (ver, tuple(t.contiguous() for t in (w1, b1, w2, b2, wh, bh)))
            self._branch_cache[sec] = hit
        return hit[1]

    def _branch(self, feat, priors, sec: bool):
        """feat [B,N,E], priors [B,N,6+S] -> (predictions, prediction_lines) [B,N,6+S]  (Router4OLV2.py:288-361)."""
        w1, b1, w2, b2, wh, bh = self._branch_weights(sec)
        h = PF.linear(feat, w1, b1, relu=True)
        h = PF.linear(h, w2, b2, relu=True)
        out = PF.linear(h, wh, bh)
        b = priors.shape[0]
        preds, lines = K.lane_update_fwd(priors.reshape(b * self.num_priors, -1).contiguous(),
                                         out.reshape(b * self.num_priors, -1).contiguous(), self.prior_ys, self.img_w, self.img_h)
        return preds.view_as(priors), lines.view_as(priors)

    def forward_first(self, decode_feat_l, priors):
        return self._branch(decode_feat_l, priors, False)

    def forward_second(self, last_cut, attn_feat, stage, priors):
        """attn_feat [N,1,E]; last_cut: None / empty (-> the frame's own tokens), [M,1,E], or ([M,1,E], valid bool[M])."""
        mask = None
        if isinstance(last_cut, tuple):
            last_cut, mask = last_cut
        if last_cut is None or last_cut.shape[0] == 0:
            last_cut, mask = attn_feat, None
        feat = self.transformer_Dec(tgt=attn_feat, memory=last_cut, memory_key_valid=mask)
        return self._branch(feat.reshape(1, self.num_priors, -1), priors, True)

    # ---- one frame ------------------------------------------------------------------------------------------------
    def stage_front(self, fmap, stage, priors, on_map, pro_feat, gate_out=None):
        """ROI pooling, gate, dynamic head, branch A of one stage for B frames: fmap [B,h,w,C_s]; priors [B,N,6+S];
        on_map [B,N,P_s]; pro_feat [B,N,E]."""
        roi, roi_cp = K.roi_pool_fwd(fmap, on_map.contiguous(), getattr(self, f"prior_feat_ys_{stage}"), with_cp=True)
        gate = self.router(roi_cp, stage, out=gate_out)                              # [B,N,1]
        local = self.DHead_series[stage](pro_feat, roi)                              # [B,N,E]
        pred_a, lines_a = self.forward_first(local, priors)
        return dict(gate=gate, local=local, pred_a=pred_a, lines_a=lines_a)

    def forward(self, x, last_cuts=None, stage0=None, gate_rows=None):
        """x = the three pyramid levels (fine -> coarse, NHWC [1,h,w,C_l]) of ONE frame; last_cuts = None (no memory yet: the
        decoder attends to the frame's own tokens) or a list over remembered frames of per-stage (tokens, valid).
        stage0 (optional) = this frame's stage-0 `stage_front` results; gate_rows (optional) = [S,N] buffer the gate scores are
        written into.  Returns (output dict with `predictions_lists` / `predictions_sec`, per-stage tokens [N,1,E], gates)."""
        if self.training:
            raise NotImplementedError(_INFERENCE_ONLY)
        levels = list(x)[::-1]
        priors, on_map = self.priors.unsqueeze(0), self.priors_on_featmap.unsqueeze(0)
        pro_feat = self.pro_embedding.weight.detach().unsqueeze(0)
        pos = self.PositionEmbedding.pos_table
        out_a, out_b, attn_feats, gates = [], [], [], []
        for stage in range(self.refine_layers):
            if stage == 0 and stage0 is not None:
                fr = stage0
            else:
                fr = self.stage_front(levels[stage], stage, priors, on_map, pro_feat, None if gate_rows is None else gate_rows[stage])
            attn = (fr["local"][0] + pos).unsqueeze(1)                               # content + table [N,1,E] (:266-269)
            mem = None
            if last_cuts:
                mem = (torch.cat([c[stage][0] for c in last_cuts], dim=0), torch.cat([c[stage][1] for c in last_cuts], dim=0))
            pred_b, lines_b = self.forward_second(mem, attn, stage, priors)
            pro_feat = fr["local"]
            out_a.append(fr["pred_a"]); out_b.append(pred_b); attn_feats.append(attn); gates.append(fr["gate"])
            if stage != self.refine_layers - 1:
                priors, on_map = K.blend_priors(fr["gate"].contiguous(), fr["lines_a"].contiguous(), lines_b.contiguous(),
                                                getattr(self, f"sample_x_indexs_{stage + 1}"))
        return {"predictions_lists": out_a, "seg": None, "flow": None, "predictions_sec": out_b}, attn_feats, gates

    # ---- decode (the V1 code: Router4OLV2.py:363-448 repeats Router4OL.py:394-479) -----------------------------------
    predictions_to_pred = DetNetV2.predictions_to_pred
    decode_device = DetNetV2.decode_device
    get_lanes = DetNetV2.get_lanes


class RouterOL(nn.Module):
    def __init__(self, cfg, criterion=None):
        super().__init__()
        if cfg.backbone == "revcol":
            raise NotImplementedError("the RevCol backbone is a dead branch of the reference (Router4OLV2.py:474-475)")
        self.backbone = Encoder(cfg=cfg)
        self.router = RouterV2(cfg=cfg)
        self.criterion = criterion
        self.save_freq = cfg.save_freq
        self.save_freq_max = cfg.save_freq_max
        self.crop_size = cfg.dscfg.crop_size
        self.org_size = (cfg.dscfg.org_height, cfg.dscfg.org_width)
        self.faithful_memory = True     # True: saveMemory4Test as shipped (memory = mean token only); False: kept lanes + mean of the rest
        self.batch_stage0 = True        # stage-0 pooling / gate / dynamic head / branch A of all frames in one batch

    def _memory(self, attn_feats, anchors_sorted):
        """Per stage (tokens [L+1,1,E], valid [L+1]): the positives' tokens (none when faithful_memory) + the mean of the rest."""
        rows = anchors_sorted
        if self.faithful_memory:
            rows = torch.full_like(anchors_sorted, -1)
        return [K.memory_tokens(a.contiguous(), rows.contiguous()) for a in attn_feats]

    def infer_device(self, frame: torch.Tensor):
        """Eval forward of one clip without host synchronisation (hipGraph-capturable).  Returns (kept_rows [T,max_lanes,6+S],
        num [T], anchors [T,max_lanes], aux) on the device; aux = per-frame lines / gates / stage outputs for the tests."""
        det = self.router
        T = frame.shape[0]
        feats = self.backbone(frame)
        N = det.num_priors
        gate_rows = torch.empty((T, det.refine_layers, N), dtype=torch.float32, device=frame.device)
        stage0 = None
        if self.batch_stage0:
            front = det.stage_front(feats[-1], 0, det.priors.unsqueeze(0).expand(T, -1, -1).contiguous(),
                                    det.priors_on_featmap.unsqueeze(0).expand(T, -1, -1).contiguous(),
                                    det.pro_embedding.weight.detach().unsqueeze(0).expand(T, -1, -1))
            gate_rows[:, 0] = front["gate"].view(T, N)
            stage0 = [{k: v[t:t + 1] for k, v in front.items()} for t in range(T)]
        last_cuts, rows, nums, anchors, aux = [], [], [], [], []
        for t in range(T):
            cur = tuple(f[t:t + 1] for f in feats)
            mem = None if t < self.save_freq else last_cuts
            outputs, cur_cut, gates = det(cur, mem, None if stage0 is None else stage0[t], gate_rows[t])
            lines = K.route_lines(gate_rows[t], outputs["predictions_lists"][-1][0].contiguous(),
                                  outputs["predictions_sec"][-1][0].contiguous(), hard=True)          # [N,6+S]
            dec = det.decode_device(lines)
            rows.append(dec["kept_rows"]); nums.append(dec["num"]); anchors.append(dec["anchors"])
            aux.append(dict(lines=lines, outputs=outputs, keep_mask=dec["keep_mask"], keep_c=dec["keep_c"]))
            last_cuts.append(self._memory([c.detach() for c in cur_cut], dec["anchors_sorted"]))
            if t >= self.save_freq_max:
                last_cuts.pop(0)
        return torch.stack(rows), torch.stack(nums), torch.stack(anchors), dict(frames=aux, gates=gate_rows)

    def lanes_from_device(self, kept_rows: torch.Tensor, nums: torch.Tensor):
        rows, n = kept_rows.cpu(), nums.cpu().tolist()
        return {"lane_lines": [self.router.predictions_to_pred(rows[t, :n[t]]) if n[t] else [] for t in range(len(n))]}

    def forward(self, inputs: dict):
        frame, lanes = inputs.values()
        if self.training:
            raise NotImplementedError(_INFERENCE_ONLY)
        if not frame.is_cuda:
            raise RuntimeError("phnet_amd runs on the GPU only: move the model and the clip to cuda")
        with torch.no_grad():
            rows, nums, _, _ = self.infer_device(frame)
            return self.lanes_from_device(rows, nums)
