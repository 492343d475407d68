"""PHNet clip model on the MI355X HIP kernels, behind the reference's module API.

Drop-in for `libs.models.Router4OL` (reference: libs/models/Router4OL.py): same class names, constructor signatures,
state_dict keys/shapes, `forward(inputs: dict)` contract (train: summed clip loss; eval: {'lane_lines': [...]}).
Parameters live in ordinary nn.Module containers; every contraction / normalisation / pooling / NMS on the path runs
through phnet_amd.hip_ops (C-ABI -> gfx950 kernels).  There is no CPU path: tensors must be on the GPU.
"""
import copy
import math

import torch
import torch.nn as nn

from phnet_amd import functional as PF
from phnet_amd.arena import SinkPool, grad_sink
from phnet_amd.trunk import encoder_forward
from ..ops import nms
from ..utils.lane import Lane
from .fpn import FPN
from .resnet import ResNetWrapper
from .Router import AdaptiveRouter4Lane
from .utils.dynamic_head import DynamicConv
from .utils.transformer import TransformerDecoder, TransformerDecoderLayer


def LinearModule(hidden_dim):
    return nn.ModuleList([nn.Linear(hidden_dim, hidden_dim), nn.ReLU(inplace=False)])


class PositionalEncodingLearned(nn.Module):
    def __init__(self, num_embeddings, num_pos_feats=256):
        super().__init__()
        self.embed = nn.Embedding(num_embeddings, num_pos_feats)
        nn.init.uniform_(self.embed.weight)


class Encoder(nn.Module):
    """ResNet trunk + FPN; returns (P3, P4, P5) as NHWC tensors [T,h,w,64] (internal layout of the head)."""

    def __init__(self, cfg):
        super().__init__()
        self.backbone = ResNetWrapper(**cfg.backbone)
        self.neck = FPN(**cfg.neck) if cfg.haskey("neck") else None
        if self.neck is None:
            raise NotImplementedError("the hot path needs the FPN neck (options4OL.py:58-61)")
        self.staged = False                     # True: forward outside autograd + finish_backward() (phnet_amd/trunk.py, graphed.py)
        self._staged = None

    def finish_backward(self, stage_done=None):
        """Staged mode only: the trunk's backward, to be called after the lane head's `loss.backward()`."""
        from phnet_amd.trunk import encoder_backward_staged
        encoder_backward_staged(self, stage_done)

    def forward(self, batch):
        frames = batch["img"] if isinstance(batch, dict) else batch
        return encoder_forward(self, frames)


class DetNetV2(nn.Module):
    """Anchor-based lane head: ROI pooling along anchors, routing gate, dynamic head, branch A (per-frame MLPs),
    branch B (cross-frame transformer), three coarse-to-fine refinement stages."""

    def __init__(self, prior_feat_channels=64, fc_hidden_dim=64, num_fc=2, refine_layers=3, sample_points=36, cfg=None):
        super().__init__()
        self.cfg = cfg
        self.img_w, self.img_h = cfg.img_w, cfg.img_h
        self.n_strips, self.n_offsets = cfg.num_points - 1, cfg.num_points
        self.num_priors = cfg.num_priors
        self.sample_points, self.refine_layers = sample_points, refine_layers
        self.fc_hidden_dim = self.reg_hidden_dim = fc_hidden_dim
        self.prior_feat_channels = prior_feat_channels
        idx = (torch.linspace(0, 1, steps=sample_points, dtype=torch.float32) * self.n_strips).long()
        self.register_buffer("sample_x_indexs", idx)
        self.register_buffer("prior_feat_ys", torch.flip(1 - idx.float() / self.n_strips, dims=[-1]))
        self.register_buffer("prior_ys", torch.linspace(1, 0, steps=self.n_offsets, dtype=torch.float32))
        self.prior_embeddings = nn.Embedding(self.num_priors, 3)
        with torch.no_grad():
            self.prior_embeddings.weight.copy_(self._initial_anchors())
            pri, on_map = self._expand_anchors(self.prior_embeddings.weight)
        self.register_buffer("priors", pri)
        self.register_buffer("priors_on_featmap", on_map)

        def tower(width):
            mods = []
            for _ in range(num_fc):
                mods += [*LinearModule(width)]
            return nn.ModuleList(mods)
        c = fc_hidden_dim
        self.reg_modules, self.cls_modules, self.iou_modules = tower(c), tower(c), tower(c)
        self.reg_layers, self.cls_layers, self.iou_layers = nn.Linear(c, 4), nn.Linear(c, 2), nn.Linear(c, self.n_offsets)
        self.reg_modules_sec, self.cls_modules_sec, self.iou_modules_sec = tower(2 * c), tower(2 * c), tower(2 * c)
        self.reg_layers_sec, self.cls_layers_sec = nn.Linear(2 * c, 4), nn.Linear(2 * c, 2)
        self.iou_layers_sec = nn.Linear(2 * c, self.n_offsets)
        for lin in (self.cls_layers, self.reg_layers, self.cls_layers_sec, self.reg_layers_sec):
            for p in lin.parameters():
                nn.init.normal_(p, mean=0., std=1e-3)
        layer = TransformerDecoderLayer(d_model=2 * c, nhead=8, dim_feedforward=256, dropout=0.1, activation="gelu",
                                        normalize_before=True)
        self.transformer_Dec = TransformerDecoder(layer, 2, nn.LayerNorm(2 * c))
        self.PositionEmbedding = PositionalEncodingLearned(num_embeddings=self.num_priors, num_pos_feats=c)
        head = DynamicConv(feat_size=sample_points, inplanes=c, early_return=False)
        self.DHead_series = nn.ModuleList(copy.deepcopy(head) for _ in range(refine_layers))
        self.pro_embedding = nn.Embedding(self.num_priors, prior_feat_channels)
        self.router = AdaptiveRouter4Lane(num_priors=self.num_priors, features_channels=prior_feat_channels,
                                          num_points=sample_points, out_channels=1, reduction=4, stages=refine_layers)
        self._branch_cache = None              # per-clip cache of the assembled tower weights (see _branch_weights)
        self._sink_pool = None                 # per-clip zero buffers of the gradient sinks (phnet_amd.arena.SinkPool)

    # ---- anchors ---------------------------------------------------------------------------------------------
    def _initial_anchors(self) -> torch.Tensor:
        n = self.num_priors
        quarter, half = n // 4, n // 2
        side, bottom = 0.8 / (quarter // 2 - 1), 0.5 / (quarter // 2 + 1)
        rows = []
        for i in range(n):
            even = i % 2 == 0
            if i < quarter:
                rows.append(((i // 2) * side, 0., 0.16 if even else 0.32))
            elif i < half:
                rows.append((0., ((i - quarter) // 2 + 1) * bottom, 0.2 if even else 0.4))
            elif i < half + quarter:
                rows.append((0., ((i - half) // 2 + 1) * bottom + 0.5, 0.6 if even else 0.8))
            else:
                rows.append((((i - half - quarter) // 2) * side, 1., 0.68 if even else 0.84))
        return torch.tensor(rows, dtype=torch.float32)

    def _line_xs(self, sy, sx, theta):
        return (sx * (self.img_w - 1) + ((1 - self.prior_ys - sy) * self.img_h / torch.tan(theta * math.pi + 1e-5))) / (self.img_w - 1)

    def _expand_anchors(self, emb: torch.Tensor):
        if emb.is_cuda:
            # the prior update with a zero head IS the anchor expansion (start / angle + tanh(0), length 0, xs of the straight
            # line): one launch forward, one backward (phnet_lane_update_fwd / _bwd) instead of ~25 elementwise ATen launches
            S = self.n_offsets
            zero_head = getattr(self, "_zero_head", None)
            if zero_head is None or zero_head.device != emb.device:
                zero_head = self._zero_head = torch.zeros((1, self.num_priors, 6 + S), dtype=torch.float32, device=emb.device)
            pri0 = torch.nn.functional.pad(emb, (2, 1 + S)).unsqueeze(0)                 # [1,N,6+S]: (0, 0, sy, sx, theta, 0, 0...)
            pri = PF.lane_update(pri0, zero_head, self.prior_ys, self.img_w, self.img_h)[1].squeeze(0)
            return pri, pri[:, 6 + self.sample_x_indexs]
        sy, sx, theta = emb.split(1, dim=1)                              # (module construction on the host: plain tensor ops)
        xs = self._line_xs(sy, sx, theta)
        z = emb.new_zeros(emb.shape[0], 1)
        pri = torch.cat([z, z, emb, z, xs], dim=1)
        return pri, pri[:, 6 + self.sample_x_indexs]

    def generate_priors_from_embeddings(self):
        return self._expand_anchors(self.prior_embeddings.weight)

    # ---- the two branches --------------------------------------------------------------------------------------
    def _branch_weights(self, sec: bool):
        """The three towers (cls / reg / offsets) of one branch as ONE 3-GEMM chain: layer 1 weights concatenated
        ([3C,C]: the towers share their input), layer 2 block-diagonal ([3C,3C]), output heads block-structured
        ([44,3C]: 2 cls + 4 reg + S offsets rows, zero-padded to a multiple of 4).  Same arithmetic as nine separate
        Linear layers (the extra terms are exact zeros); 3x fewer launches forward, ~3x fewer backward.  Assembled once
        per clip (the weights do not change inside a clip) and cached."""
        key = "sec" if sec else "fir"
        if self._branch_cache is None:
            self._branch_cache = {}
        if key not in self._branch_cache:
            s = "_sec" if sec else ""
            g = lambda name: getattr(self, name + s)                                 # noqa: E731
            tw = [g("cls_modules"), g("reg_modules"), g("iou_modules")]
            hd = [g("cls_layers"), g("reg_layers"), g("iou_layers")]
            params = [x for t, h in zip(tw, hd) for x in (t[0].weight, t[0].bias, t[2].weight, t[2].bias, h.weight, h.bias)]
            if all(p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() for p in params):
                # one launch assembles the six operands, one launch scatters their gradients (csrc/towers.hip)
                self._branch_cache[key] = PF.assemble_towers(tw[0][0].in_features, [h.out_features for h in hd], self._sink_pool, params)
                return self._branch_cache[key]
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
            self._branch_cache[key] = tuple(grad_sink(t, self._sink_pool) for t in (w1, b1, w2, b2, wh, bh))
        return self._branch_cache[key]

    def _tower_params(self, sec: bool):
        s = "_sec" if sec else ""
        g = lambda name: getattr(self, name + s)                                     # noqa: E731
        tw = [g("cls_modules"), g("reg_modules"), g("iou_modules")]
        hd = [g("cls_layers"), g("reg_layers"), g("iou_layers")]
        return [x for t, h in zip(tw, hd) for x in (t[0].weight, t[0].bias, t[2].weight, t[2].bias, h.weight, h.bias)], [h.out_features for h in hd]

    def _branch(self, feat, priors, sec: bool):
        if not torch.is_grad_enabled() and feat.is_cuda and feat.shape[-1] in (64, 128):
            # no autograd graph wanted (inference, the deferred branch-B passes): towers + heads + prior update in ONE launch
            from phnet_amd import hip_ops as K
            params, head_out = self._tower_params(sec)
            preds, lines = K.tower_chain_fwd(feat.reshape(-1, feat.shape[-1]).contiguous(), params, head_out,
                                             priors.reshape(-1, priors.shape[-1]).contiguous(), self.prior_ys, self.img_w, self.img_h)
            return preds.view(priors.shape), lines.view(priors.shape)
        w1, b1, w2, b2, wh, bh = self._branch_weights(sec)
        h = PF.linear(feat, w1, b1, relu=True)
        h = PF.linear(h, w2, b2, relu=True)
        out = PF.linear(h, wh, bh)
        return PF.lane_update(priors, out.reshape(-1, self.num_priors, out.shape[-1]), self.prior_ys, self.img_w, self.img_h)

    def forward_first(self, decode_feat_l, priors):
        return self._branch(decode_feat_l, priors, False)

    def forward_second(self, last_cut, attn_feat, stage, priors):
        """attn_feat [N,1,128]; last_cut: None, a [M,1,128] tensor, or (tokens [M,1,128], valid bool[M]) with fixed M."""
        mask = None
        if isinstance(last_cut, tuple):
            last_cut, mask = last_cut
        if last_cut is not None and last_cut.shape[0] != 0:
            feat = self.transformer_Dec(tgt=attn_feat, memory=last_cut, memory_key_valid=mask)
        else:
            feat = attn_feat
        return self._branch(feat.reshape(1, self.num_priors, -1), priors, True)

    # ---- one refinement stage / one frame ---------------------------------------------------------------------
    def stage_front(self, fmap, stage, priors, on_map, pro_feat):
        """The part of a stage that needs no other frame: ROI pooling, routing gate, dynamic head, branch A.  Batched over B frames:
        fmap [B,h,w,C]; priors [B,N,6+S]; on_map [B,N,P]; pro_feat [B,N,C].  Returns dict(gate, local, pred_a, lines_a)."""
        roi, roi_cp = PF.roi_pool(fmap, on_map, self.prior_feat_ys)                  # [B,N,P,C], [B,N,C,P]
        gate = self.router(roi_cp, stage)                                            # [B,N,1]
        local = self.DHead_series[stage](pro_feat, roi)                              # [B,N,C]
        pred_a, lines_a = self.forward_first(local, priors)
        return dict(gate=gate, local=local, pred_a=pred_a, lines_a=lines_a)

    def stage_back(self, front, stage, priors, memory):
        """Branch B of ONE frame on top of its stage_front results (all [1,...])."""
        gate, local = front["gate"], front["local"]
        attn = front.get("attn")
        if attn is None:
            pos = self.PositionEmbedding.embed.weight.unsqueeze(1)                   # [N,1,C]
            attn = torch.cat([local.transpose(0, 1), pos], dim=-1)                   # [N,1,2C]
        pred_b, lines_b = self.forward_second(memory, attn, stage, priors)
        return dict(pred_a=front["pred_a"], lines_a=front["lines_a"], pred_b=pred_b, lines_b=lines_b, attn=attn, gate=gate, local=local)

    def stage_forward(self, fmap, stage, priors, on_map, pro_feat, memory):
        """fmap [1,h,w,C] NHWC level of this stage; priors [1,N,6+S]; on_map [1,N,P]; pro_feat [1,N,C];
        memory None | [M,1,2C] | ([M,1,2C], valid bool[M]).  Returns dict(pred_a, lines_a, pred_b, lines_b, attn, gate, local)."""
        return self.stage_back(self.stage_front(fmap, stage, priors, on_map, pro_feat), stage, priors, memory)

    def stage0_all_frames(self, fmaps0):
        """Stage 0 of every frame starts from the same learned anchors and embeddings, and only its branch B looks at
        earlier frames - so ROI pooling, routing gate, dynamic head and branch A of stage 0 run ONCE for the whole clip (GEMM rows
        T*N instead of N, a fifth of the launches).  fmaps0 [T,h,w,C] = the stage-0 pyramid level of all frames.
        Returns a list over frames of stage_front dicts ([1,...] views; their gradients meet in one cat)."""
        T = fmaps0.shape[0]
        if self.training:
            self.priors, self.priors_on_featmap = self.generate_priors_from_embeddings()
        priors = self.priors.unsqueeze(0).expand(T, -1, -1)
        on_map = self.priors_on_featmap.unsqueeze(0).expand(T, -1, -1).contiguous()
        pro = self.pro_embedding.weight.unsqueeze(0).expand(T, -1, -1)
        front = self.stage_front(fmaps0, 0, priors, on_map, pro)
        parts = {k: v.split(1, dim=0) for k, v in front.items()}
        return [{k: parts[k][t] for k in parts} for t in range(T)]

    def forward(self, x, last_cuts=None, stage0=None):
        """x = (P3, P4, P5) NHWC [1,h,w,C] for ONE frame; last_cuts = list over remembered frames of per-stage tokens;
        stage0 (optional) = this frame's entry of stage0_all_frames (self.priors must then be current)."""
        levels = list(x)[::-1]
        last_cuts = last_cuts or []
        if self.training and stage0 is None:
            self.priors, self.priors_on_featmap = self.generate_priors_from_embeddings()
        priors, on_map = self.priors.unsqueeze(0), self.priors_on_featmap.unsqueeze(0)
        pro_feat = self.pro_embedding.weight.unsqueeze(0)
        out_a, out_b, attn_feats, gates = [], [], [], []
        for stage in range(self.refine_layers):
            mem = None
            if len(last_cuts):
                mem = (torch.cat([fr[stage][0] for fr in last_cuts], dim=0), torch.cat([fr[stage][1] for fr in last_cuts], dim=0))
            if stage == 0 and stage0 is not None:
                r = self.stage_back(stage0, 0, priors, mem)
            else:
                r = self.stage_forward(levels[stage], stage, priors, on_map, pro_feat, mem)
            pro_feat = r["local"].detach()
            out_a.append(r["pred_a"]); out_b.append(r["pred_b"]); attn_feats.append(r["attn"]); gates.append(r["gate"])
            if stage != self.refine_layers - 1:
                from phnet_amd import hip_ops as K
                priors, on_map = K.blend_priors(r["gate"].detach().contiguous(), r["lines_a"].detach().contiguous(),
                                                r["lines_b"].detach().contiguous(), self.sample_x_indexs)
        return {"predictions_fir": out_a, "predictions_sec": out_b}, attn_feats, gates

    def forward_clips(self, x, last_cuts=None, stage0=None):
        """Eval-time twin of forward() for the SAME frame index of B independent clips: x = (P3, P4, P5) [B,h,w,C];
        last_cuts = list over remembered frames of per-stage (tokens [B,L+1,E], valid [B,L+1]); stage0 = stage_front results
        [B,...] of stage 0.  Every head kernel sees B*N rows; attention and the memory tokens stay inside each clip.
        Returns ({"predictions_fir": [...], "predictions_sec": [...]} with [B,N,6+S] entries, attn feats [B,N,2C], gates [B,N,1])."""
        levels = list(x)[::-1]
        last_cuts = last_cuts or []
        B, N = levels[0].shape[0], self.num_priors
        priors = self.priors.unsqueeze(0).expand(B, -1, -1)
        on_map = self.priors_on_featmap.unsqueeze(0).expand(B, -1, -1).contiguous()
        pro_feat = self.pro_embedding.weight.unsqueeze(0).expand(B, -1, -1)
        pos = self.PositionEmbedding.embed.weight.unsqueeze(0).expand(B, -1, -1)
        out_a, out_b, attn_feats, gates = [], [], [], []
        from phnet_amd import hip_ops as K
        for stage in range(self.refine_layers):
            front = stage0 if (stage == 0 and stage0 is not None) else self.stage_front(levels[stage], stage, priors, on_map, pro_feat)
            local = front["local"]
            attn = torch.cat([local, pos], dim=-1)                                   # [B,N,2C]
            if len(last_cuts):
                mem = torch.cat([fr[stage][0] for fr in last_cuts], dim=1)           # [B,M,2C]
                valid = torch.cat([fr[stage][1] for fr in last_cuts], dim=1)         # [B,M]
                feat = self.transformer_Dec(tgt=attn.reshape(B * N, -1), memory=mem.reshape(-1, mem.shape[-1]),
                                            memory_key_valid=valid.reshape(-1), batch=B)
            else:
                feat = attn
            pred_b, lines_b = self._branch(feat.reshape(B, N, -1), priors, True)
            pro_feat = local.detach()
            out_a.append(front["pred_a"]); out_b.append(pred_b); attn_feats.append(attn); gates.append(front["gate"])
            if stage != self.refine_layers - 1:
                priors, on_map = K.blend_priors(front["gate"].detach().contiguous(), front["lines_a"].detach().contiguous(),
                                                lines_b.detach().contiguous(), self.sample_x_indexs)
        return {"predictions_fir": out_a, "predictions_sec": out_b}, attn_feats, gates

    # ---- decode ---------------------------------------------------------------------------------------------------
    def predictions_to_pred(self, predictions, ori_img_h=None, cut_height=0):
        """Kept lanes [k,6+S] (length already in strips) -> list of Lane.  Host-side, float64 like the reference
        (Router4OL.py:394-435) but without mutating the prior_ys buffer."""
        rows = predictions.detach().cpu()
        ys_all = self.prior_ys.detach().cpu().double()
        lanes = []
        for row in rows:
            xs = row[6:].clone()
            start = min(max(0, int(round(row[2].item() * self.n_strips))), self.n_strips)
            end = min(start + int(round(row[5].item())) - 1, self.n_offsets - 1)
            inside = ((xs[:start] >= 0.) & (xs[:start] <= 1.)).numpy()
            outside = ~(inside[::-1].cumprod()[::-1].astype(bool))
            xs[end + 1:] = -2
            xs[:start][torch.from_numpy(outside.copy())] = -2
            sel = xs >= 0
            lx, ly = xs[sel].flip(0).double(), ys_all[sel].flip(0)
            if lx.numel() <= 1:
                continue
            lanes.append(Lane(points=torch.stack([lx, ly], dim=1).numpy(),
                              metadata={"start_x": row[3], "start_y": row[2], "conf": row[1]}))
        return lanes

    def decode_device(self, lines: torch.Tensor):
        """lines [N,6+S] (blended predictions of one frame) -> device-resident decode (hip_ops.lane_decode): the
        sync-free form of get_lanes - confidence mask, NMS and the gathered kept rows in one launch."""
        from phnet_amd import hip_ops as K
        tp = self.cfg.test_parameters
        return K.lane_decode(lines.contiguous(), tp.conf_threshold, tp.nms_thres, self.cfg.max_lanes, self.img_w)

    def get_lanes(self, output, org_size=None, crop_size=0, as_lanes=True):
        """output [B,N,6+S] blended lines -> (decoded per batch item, keep_inds, keep) as in Router4OL.py:437-479."""
        decoded, keep_inds, keep = [], None, []
        for predictions in output:
            scores = torch.softmax(predictions[:, :2], dim=1)[:, 1]
            keep_inds = scores >= self.cfg.test_parameters.conf_threshold
            cand = predictions[keep_inds]
            keep = []
            if cand.shape[0] == 0:
                decoded.append([])
                continue
            rows = torch.cat([cand[:, :4], cand[:, 5:]], dim=-1).detach().clone()
            rows[:, 3] = rows[:, 3] * (self.img_w - 1)
            rows[:, 4] = rows[:, 4] * self.n_strips
            rows[:, 5:] = rows[:, 5:] * (self.img_w - 1)
            keep, num_to_keep, _ = nms(rows.contiguous(), scores[keep_inds].contiguous(),
                                       overlap=self.cfg.test_parameters.nms_thres, top_k=self.cfg.max_lanes)
            keep = keep[:int(num_to_keep)]
            kept = cand[keep].clone()
            if kept.shape[0] == 0:
                decoded.append([])
                continue
            kept[:, 5] = torch.round(kept[:, 5] * self.n_strips)
            decoded.append(self.predictions_to_pred(kept) if as_lanes else kept)
        return decoded, keep_inds, keep


BRANCH_B_SITES = 1 << 10      # dropout site numbering of branch B inside DropoutStream.items (functional.py)


class _Stacked(list):
    """The per-stage token rings (or their validity masks) as a list, with the one tensor they are views of in `.stacked`."""
    stacked = None


class _BranchBDeferred(torch.autograd.Function):
    """Backward of ALL branch-B passes of a clip as ONE batch.

    Forward dependencies tie the (frame t, stage s) passes of branch B into a chain (the memory of (t, s) is picked by the label
    assignment of (t' < t, s), its priors are the blend of (t, s-1)), so the forward runs pass by pass on 240 rows - but
    nothing flows BACK along those links: memory tokens and blended priors are detached (Router4OL.py:292-302, 563-584).  The
    gradient of a pass needs only d loss / d predictions of that pass, and every pass uses the same decoder / tower weights.
    So the passes run forward WITHOUT autograd (schedule loop in RouterOL.train_clip_stage_major), and this node recomputes
    them as one batch (S*T*240 rows per kernel; decoder batch = the S*(T-1) passes that have a memory) with autograd enabled
    when the gradients of their predictions arrive, and back-propagates through that batch: ~70 launches instead of
    ~40 per pass x 12-15 passes, at the price of one extra batched forward.  Dropout: both runs draw their masks per ITEM
    (functional.DropoutStream.items), so the batch sees exactly the masks the passes saw.
    Inputs that receive gradients: the tokens of each stage [T,N,E], the stage-0 priors [T,N,6+S] and branch B's parameters."""

    @staticmethod
    def forward(ctx, model, pre_pred, pri_all, rings, valids, n_tok, pri0, *rest):
        """rest = the n_tok token tensors, then branch B's parameters (inputs of this node so that the OUTER graph reaches them:
        stock DistributedDataParallel marks parameters it cannot reach from the loss as unused and would see them a second
        time when the inner backward ran - their gradients are handed back through this node instead)."""
        ctx.model, ctx.rings, ctx.valids, ctx.n_tok = model, rings, valids, n_tok
        ctx.params = rest[n_tok:]
        ctx.save_for_backward(pri_all, *[t.detach() for t in rest[:n_tok]])
        return pre_pred.view_as(pre_pred)

    @staticmethod
    def backward(ctx, d_pred):
        pri_all, *tokens = ctx.saved_tensors
        det = ctx.model.detNet
        S, T = len(tokens), tokens[0].shape[0]
        need_pri = ctx.needs_input_grad[6]
        wanted = [i for i, p in enumerate(ctx.params) if ctx.needs_input_grad[7 + ctx.n_tok + i]]
        # the assembled tower weights of this recomputation get their OWN gradient sinks: the inner backward below consumes the
        # graph that hands a sink's buffer on, so a cache entry must never be shared between two recomputations (several
        # forward passes before one backward, trainOL.py:205-212) or with the forward that is being built right now
        outer_cache, det._branch_cache = det._branch_cache, None
        try:
            with torch.enable_grad():
                tok = torch.cat(tokens, dim=0).requires_grad_()                      # [S*T,N,E], item (s, t) at s*T + t
                pri = pri_all.detach().requires_grad_(need_pri)
                pred = ctx.model._branch_b_batch(tok, pri, ctx.rings, ctx.valids, S, T)
                # (arena-backed parameters: the HIP backward kernels accumulate straight into .grad and autograd reports None)
                grads = torch.autograd.grad([pred], [tok] + ([pri] if need_pri else []) + [ctx.params[i] for i in wanted],
                                            [d_pred.contiguous()], allow_unused=True)
        finally:
            det._branch_cache = outer_cache
        ctx.rings = ctx.valids = None
        d_tok = grads[0].split(T, dim=0)
        d_pri0 = grads[1][:T] if need_pri else None
        d_par = [None] * len(ctx.params)
        for j, i in enumerate(wanted):
            d_par[i] = grads[1 + int(need_pri) + j]
        return (None, None, None, None, None, None, d_pri0) + tuple(d_tok) + tuple(d_par)


class RouterOL(nn.Module):
    def __init__(self, cfg, criterion=None):
        super().__init__()
        self.backbone = Encoder(cfg=cfg)
        self.detNet = DetNetV2(cfg=cfg)
        self.criterion = criterion
        self.save_freq_max = cfg.save_freq_max
        self.crop_size = cfg.dscfg.crop_size
        self.org_size = (cfg.dscfg.org_height, cfg.dscfg.org_width)
        self.sync_free_eval = True      # eval: fused device-side decode, one D2H copy per clip (False: per-frame get_lanes)
        self.batch_stage0 = True        # stage-0 ROI pooling / dynamic head / branch A of all frames in one batch
        # training schedule of the (frame, stage) grid: "wavefront" (anti-diagonals: branch B of up to 3 pairs as one batch),
        # "stage" (every stage's frame-independent front batched over the frames, branch B frame by frame) or "frame" (the
        # reference's loop order)
        self.schedule = "stage"          # measured at T = 5, ResNet-34 320x800: frame 25.0, stage 22.3, wavefront 22.9 ms per step
        # stage-major schedule only: branch B's passes run forward without autograd and their backward runs as ONE batch
        # (_BranchBDeferred); False: every pass is its own autograd sub-graph (same arithmetic, same dropout masks)
        self.defer_branch_b = True

    @property
    def stage_major(self):
        return self.schedule != "frame"

    @stage_major.setter
    def stage_major(self, on: bool):
        self.schedule = "stage" if on else "frame"

    def _begin_clip(self):
        det = self.detNet
        det._branch_cache = None                                               # weights may have changed since the last clip
        det._sink_pool = SinkPool() if self.training else None
        for head in det.DHead_series:
            head.begin_clip(det._sink_pool)
        # the anchors expanded from the embeddings are kept on the module (as in the reference, Router4OL.py:258-259) - but
        # without their autograd history: a graph that outlives its step keeps AccumulateGrad nodes bound to that step's
        # stream, and a later hipGraph capture on another stream then faults (PyTorch: "AccumulateGrad node's stream does
        # not match")
        if det.priors.grad_fn is not None:
            det.priors = det.priors.detach()
            det.priors_on_featmap = det.priors_on_featmap.detach()

    def infer_device(self, frame: torch.Tensor):
        """Eval forward of one clip without any host synchronisation (hipGraph-capturable): returns
        (kept_rows [T,max_lanes,6+S], num [T], anchors [T,max_lanes]) on the device."""
        self._begin_clip()
        feats = self.backbone(frame)
        last_cuts, rows, nums, anchors = [], [], [], []
        stage0 = self.detNet.stage0_all_frames(feats[-1]) if self.batch_stage0 else None
        for t in range(frame.shape[0]):
            cur = tuple(f[t:t + 1] for f in feats)
            outputs, cur_cut, gates = self.detNet(cur, last_cuts, None if stage0 is None else stage0[t])
            d = torch.stack(gates, dim=0).mean(dim=0)
            lines = outputs["predictions_sec"][-1] * d + outputs["predictions_fir"][-1] * (1 - d)
            dec = self.detNet.decode_device(lines[0])
            rows.append(dec["kept_rows"]); nums.append(dec["num"]); anchors.append(dec["anchors"])
            last_cuts.append([self._tokens(feat.detach(), dec["anchors_sorted"]) for feat in cur_cut])
            if t >= self.save_freq_max:
                last_cuts.pop(0)
        self._begin_clip()
        return torch.stack(rows), torch.stack(nums), torch.stack(anchors)

    def infer_clips_device(self, frames: torch.Tensor):
        """Eval forward of B clips at once, frames [B,T,3,H,W]: the per-frame chain is serial only INSIDE a clip, so frame t of
        all clips runs through the lane head together (B*N rows per kernel instead of N - the head is launch-bound at N = 240).
        No host synchronisation (hipGraph-capturable).  Returns (kept_rows [B,T,max_lanes,6+S], num [B,T], anchors [B,T,max_lanes])."""
        from phnet_amd import hip_ops as K
        B, T = frames.shape[:2]
        self._begin_clip()
        feats = self.backbone(frames.transpose(0, 1).reshape(T * B, *frames.shape[2:]))      # frame-major: [t*B + b]
        det = self.detNet
        n0 = feats[-1].shape[0]
        front0 = det.stage_front(feats[-1], 0, det.priors.unsqueeze(0).expand(n0, -1, -1),
                                 det.priors_on_featmap.unsqueeze(0).expand(n0, -1, -1).contiguous(),
                                 det.pro_embedding.weight.unsqueeze(0).expand(n0, -1, -1))
        last_cuts, rows, nums, anchors = [], [], [], []
        for t in range(T):
            cur = tuple(f[t * B:(t + 1) * B] for f in feats)
            outputs, cur_cut, gates = det.forward_clips(cur, last_cuts, {k: v[t * B:(t + 1) * B] for k, v in front0.items()})
            d = torch.stack(gates, dim=0).mean(dim=0)
            lines = outputs["predictions_sec"][-1] * d + outputs["predictions_fir"][-1] * (1 - d)
            dec = det.decode_device(lines)                                                    # batched over the B clips
            rows.append(dec["kept_rows"]); nums.append(dec["num"]); anchors.append(dec["anchors"])
            last_cuts.append([K.memory_tokens(feat.detach().contiguous(), dec["anchors_sorted"].contiguous()) for feat in cur_cut])
            if t >= self.save_freq_max:
                last_cuts.pop(0)
        self._begin_clip()
        return torch.stack(rows, dim=1), torch.stack(nums, dim=1), torch.stack(anchors, dim=1)

    def forward_clips_train(self, frames: torch.Tensor, lanes: torch.Tensor):
        """Training forward of B clips in one pass: frames [B,T,3,H,W], lanes [B,T,max_lanes,6+S] -> summed loss of all
        clips.  Frame t of every clip goes through the lane head together (B*N rows per kernel); attention, memory tokens,
        label assignment and the criterion stay per clip.  BatchNorm statistics are taken over all B*T frames - exactly what
        the reference computes with B data-parallel ranks and SyncBatchNorm (trainOL.py:141): the clips of a step are
        "virtual ranks" on one GPU."""
        from phnet_amd import hip_ops as K
        B, T = frames.shape[:2]
        self._begin_clip()
        PF.DropoutStream.begin_step(frames.device)
        feats = self.backbone(frames.transpose(0, 1).reshape(T * B, *frames.shape[2:]))      # frame-major: [t*B + b]
        det = self.detNet
        det.priors, det.priors_on_featmap = det.generate_priors_from_embeddings()
        n0 = feats[-1].shape[0]
        front0 = det.stage_front(feats[-1], 0, det.priors.unsqueeze(0).expand(n0, -1, -1),
                                 det.priors_on_featmap.unsqueeze(0).expand(n0, -1, -1).contiguous(),
                                 det.pro_embedding.weight.unsqueeze(0).expand(n0, -1, -1))
        front0 = {k: v.split(B, dim=0) for k, v in front0.items()}                           # per frame index: [B,...] (one cat backward)
        levels = [f.split(B, dim=0) for f in feats]
        last_cuts, total_loss = [], 0.0
        for t in range(T):
            cur = tuple(lv[t] for lv in levels)
            outputs, cur_cut, gates = det.forward_clips(cur, last_cuts, {k: v[t] for k, v in front0.items()})
            matched_all = []
            per_clip = {k: [p.split(1, dim=0) for p in v] for k, v in outputs.items()}       # split: one cat in the backward
            gate_clip = [gt.split(1, dim=0) for gt in gates]
            for b in range(B):                                                               # criterion per clip (2 launches each)
                out_b = {k: [ps[b] for ps in v] for k, v in per_clip.items()}
                matched, loss_b = self.criterion(out_b, lanes[b, t:t + 1], [gs[b] for gs in gate_clip])
                total_loss = total_loss + loss_b
                matched_all.append(matched)
            with torch.no_grad():
                tokens = []
                for s_i, feat in enumerate(cur_cut):
                    rows = torch.stack([matched_all[b][s_i] for b in range(B)])              # [B,L] matched anchors, -1 padded
                    tokens.append(K.memory_tokens(feat.detach().contiguous(), rows))
                last_cuts.append(tokens)
                if t >= self.save_freq_max:
                    last_cuts.pop(0)
        self._begin_clip()
        return total_loss

    def _branch_b_batch(self, tok, pri, rings, valids, S: int, T: int):
        """Branch B of all (stage, frame) items at once: tok [S*T,N,E] tokens, pri [S*T,N,6+S] priors, rings[s] [W+T,L1,E] /
        valids[s] [W+T,L1] the token rings of each stage (frame t attends to slots t .. t+W-1).  Frames without memory (t = 0)
        skip the decoder (Router4OL.py:378-382).  Returns the predictions [S*T,N,6+S]."""
        det = self.detNet
        N, E, W = det.num_priors, tok.shape[-1], self.save_freq_max
        feat = tok
        if T > 1:
            L1 = rings[0].shape[1]
            tokv = tok.view(S, T, N, E)
            tgt = tokv[:, 1:].reshape(S * (T - 1) * N, E)
            ring = getattr(rings, "stacked", None)                                              # [S,W+T,L1,E]
            valid = getattr(valids, "stacked", None)
            if ring is None or valid is None or ring.shape[0] != S:
                ring = torch.stack(rings) if S > 1 else rings[0].unsqueeze(0)
                valid = torch.stack(valids) if S > 1 else valids[0].unsqueeze(0)
            st = ring.stride()
            mem = ring.as_strided((S, T - 1, W * L1, E), (st[0], st[1], E, 1), ring.storage_offset() + st[1]).reshape(-1, E)
            sv = valid.stride()
            key_valid = valid.as_strided((S, T - 1, W * L1), (sv[0], sv[1], 1), valid.storage_offset() + sv[1]).reshape(-1)
            with PF.DropoutStream.items(BRANCH_B_SITES, 0, N):
                dec = det.transformer_Dec(tgt=tgt, memory=mem, memory_key_valid=key_valid, batch=S * (T - 1))
            feat = torch.cat([tokv[:, :1], dec.view(S, T - 1, N, E)], dim=1).reshape(S * T, N, E)
        pred, _ = det._branch(feat, pri, True)
        return pred

    def _clip_loss(self, per_frame, lanes):
        """Criterion over the frames of a clip whose predictions are all there (stage-major / wavefront schedules): one call when
        the criterion offers `clip_loss` (loss4OLV3: two launches for the clip), else the reference's loop (trainOLV3.py:150-171)."""
        outs = [{"predictions_fir": fr["predictions_fir"], "predictions_sec": fr["predictions_sec"]} for fr in per_frame]
        if hasattr(self.criterion, "clip_loss"):
            return self.criterion.clip_loss(outs, lanes, [fr["gates"] for fr in per_frame])
        total_loss = 0.0
        for t, out in enumerate(outs):
            _, frame_loss = self.criterion(out, lanes[t:t + 1], per_frame[t]["gates"])
            total_loss = total_loss + frame_loss
        return total_loss

    def train_clip_stage_major(self, frame: torch.Tensor, lanes: torch.Tensor):
        """Training forward of one clip in STAGE-major order.  What ties the frames of a clip together is only branch B's
        memory: stage s of frame t attends to the stage-s tokens of the up to `save_freq_max` frames before it
        (Router4OL.py:515-560), and those tokens are picked by the label assignment of that frame's OWN stage-s
        predictions.  ROI pooling, routing gate, dynamic head and branch A of stage s depend on frame t alone (through the
        stage-(s-1) blend).  So instead of 15 serial (frame, stage) iterations of ~40 launches on 240 rows, every stage runs
        its frame-independent part ONCE for the whole clip (T*240 rows per kernel, as stage 0 already did) and only
        branch B + the assignment + the token gather walk the frames - forward; branch B's BACKWARD runs as one batch over all
        (frame, stage) passes (`defer_branch_b`, _BranchBDeferred).  Same arithmetic per (frame, stage); the matched
        anchors that feed the memory come from the stand-alone assignment kernel (the criterion recomputes the identical
        assignment later, tests/test_model_gpu.py)."""
        from phnet_amd import hip_ops as K
        det = self.detNet
        T, S, N = frame.shape[0], det.refine_layers, det.num_priors
        W, L1 = self.save_freq_max, lanes.shape[1] + 1
        dev = frame.device
        feats = self.backbone(frame)                                           # 3 x [T,h,w,C] NHWC
        levels = list(feats)[::-1]
        det.priors, det.priors_on_featmap = det.generate_priors_from_embeddings()
        priors = det.priors.unsqueeze(0).expand(T, -1, -1)
        on_map = det.priors_on_featmap.unsqueeze(0).expand(T, -1, -1).contiguous()
        pro = det.pro_embedding.weight.unsqueeze(0).expand(T, -1, -1)
        E = 2 * det.fc_hidden_dim
        defer = self.defer_branch_b
        det._branch_weights(True)                                              # assembled WITH autograd, before any no_grad use
        per_frame = [{"predictions_fir": [], "predictions_sec": [], "gates": []} for _ in range(T)]
        pos = det.PositionEmbedding.embed.weight.unsqueeze(0).expand(T, -1, -1)
        tokens, rings, valids, pre_pred, pri_used = [], _Stacked(), _Stacked(), [], []
        pri0 = priors
        # the token rings of all stages in ONE zero-filled allocation each (2 fills per clip instead of 2 per stage, and the batched
        # backward finds them stacked already)
        rings.stacked = torch.zeros((S, W + T, L1, E), dtype=torch.float32, device=dev)
        valids.stacked = torch.zeros((S, W + T, L1), dtype=torch.bool, device=dev)
        for stage in range(S):
            front = det.stage_front(levels[stage], stage, priors, on_map, pro)                 # everything [T,...]
            attn_all = torch.cat([front["local"], pos], dim=-1)                               # branch B's input of all frames [T,N,E]
            # memory tokens of this stage: a ring with W leading slots that are never valid - frame t's window is ALWAYS
            # ring[t : t + W] (fixed key positions: the batched backward sees the keys where the forward saw them)
            ring, ring_valid = rings.stacked[stage], valids.stacked[stage]
            if defer:
                src, pri_src = attn_all.detach(), priors.detach()
            else:
                src = attn_all
                pri_src = priors
            tok_t = src.split(1, dim=0)
            pri_t = pri_src.split(1, dim=0) if pri_src.requires_grad else [pri_src[t:t + 1] for t in range(T)]
            lines_b, preds_b = [], []
            for t in range(T):
                mem = (ring[t:t + W].view(-1, 1, E), ring_valid[t:t + W].view(-1)) if t > 0 else None
                with torch.set_grad_enabled(not defer), PF.DropoutStream.items(BRANCH_B_SITES, stage * (T - 1) + t - 1 if t else 0, 0):
                    pred_b, line_b = det.forward_second(mem, tok_t[t].transpose(0, 1), stage, pri_t[t])
                with torch.no_grad():                                            # assignment + the tokens it selects: one launch
                    K.lane_assign_tokens(pred_b[0].detach().contiguous(), lanes[t].contiguous(), det.img_w, det.img_h,
                                         tok_t[t].detach().contiguous(), (ring[W + t], ring_valid[W + t]))
                preds_b.append(pred_b)
                lines_b.append(line_b.detach())
            pa = front["pred_a"].split(1, dim=0)                                               # one cat in the backward
            gt = front["gate"].split(1, dim=0)
            for t in range(T):
                per_frame[t]["predictions_fir"].append(pa[t])
                per_frame[t]["gates"].append(gt[t])
                if not defer:
                    per_frame[t]["predictions_sec"].append(preds_b[t])
            if defer:
                tokens.append(attn_all); rings.append(ring); valids.append(ring_valid)
                pre_pred.append(torch.cat(preds_b, dim=0)); pri_used.append(pri_src)
            if stage != S - 1:
                priors, on_map = K.blend_priors(front["gate"].detach().contiguous(), front["lines_a"].detach().contiguous(),
                                                torch.cat(lines_b, dim=0), det.sample_x_indexs)
                pro = front["local"].detach()
        if defer:
            params = list(det.transformer_Dec.parameters())
            for name in ("cls_modules_sec", "reg_modules_sec", "iou_modules_sec", "cls_layers_sec", "reg_layers_sec", "iou_layers_sec"):
                params += list(getattr(det, name).parameters())
            pred_all = _BranchBDeferred.apply(self, torch.cat(pre_pred, dim=0), torch.cat(pri_used, dim=0), rings, valids, len(tokens),
                                              pri0, *tokens, *params)
            pb = pred_all.split(1, dim=0)                                                      # one cat in the backward
            for s_ in range(S):
                for t in range(T):
                    per_frame[t]["predictions_sec"].append(pb[s_ * T + t])
        return self._clip_loss(per_frame, lanes)

    def train_clip_wavefront(self, frame: torch.Tensor, lanes: torch.Tensor):
        """Training forward of one clip as a WAVEFRONT over (frame t, stage s).  (t, s) depends on (t, s-1) - its priors are
        that stage's blend - and on (t' < t, s) - branch B attends to the stage-s tokens of earlier frames; so every pair on
        an anti-diagonal t + s = d is independent of the others.  Branch B is the launch-bound part of the step (two decoder
        layers + towers: ~40 forward and ~50 backward launches on 240 rows, 15 times in a row) and uses the SAME transformer
        and tower weights at every stage (Router4OL.py:86-103): the up to three pairs of a wavefront go through it as ONE
        batch (the attention kernels treat them as independent clips, their memory windows are fixed-length slices of the
        per-stage token rings with key masks), i.e. T + 2 serial branch-B passes instead of 3T.  The frame-independent front
        of stages 1 and 2 (ROI pooling, gate, dynamic head, branch A: per-stage weights) runs per pair; stage 0's for the
        whole clip at once.  Same arithmetic per (t, s) as the frame-major loop (tests/test_model_gpu.py)."""
        from phnet_amd import hip_ops as K
        det = self.detNet
        T, S, N = frame.shape[0], det.refine_layers, det.num_priors
        W, L1 = self.save_freq_max, lanes.shape[1] + 1
        dev = frame.device
        feats = self.backbone(frame)                                           # 3 x [T,h,w,C] NHWC
        levels = [lv.split(1, dim=0) for lv in list(feats)[::-1]]              # levels[s][t]: [1,h,w,C]
        det.priors, det.priors_on_featmap = det.generate_priors_from_embeddings()
        pos = det.PositionEmbedding.embed.weight.unsqueeze(0)                  # [1,N,C]
        front0 = det.stage_front(feats[-1], 0, det.priors.unsqueeze(0).expand(T, -1, -1),
                                 det.priors_on_featmap.unsqueeze(0).expand(T, -1, -1).contiguous(),
                                 det.pro_embedding.weight.unsqueeze(0).expand(T, -1, -1))
        front0["attn"] = torch.cat([front0["local"], pos.expand(T, -1, -1)], dim=-1)
        front0 = {k: v.split(1, dim=0) for k, v in front0.items()}
        pri0 = det.priors.unsqueeze(0)
        E = 2 * det.fc_hidden_dim
        # per-stage token rings with W leading slots that are never valid: frame t's window is ALWAYS ring[t : t + W]
        ring = [torch.zeros((W + T, L1, E), dtype=torch.float32, device=dev) for _ in range(S)]
        ring_valid = [torch.zeros((W + T, L1), dtype=torch.bool, device=dev) for _ in range(S)]
        nxt = {}                                                               # (t, s) -> (priors, on_map, pro) for its front
        per_frame = [{"predictions_fir": [None] * S, "predictions_sec": [None] * S, "gates": [None] * S} for _ in range(T)]
        for d in range(T + S - 1):
            pairs = [(t, d - t) for t in range(T) if 0 <= d - t < S]
            fronts, priors = [], []
            for t, s in pairs:
                if s == 0:
                    fr, pri = {k: v[t] for k, v in front0.items()}, pri0
                else:
                    pri, on_map, pro = nxt.pop((t, s))
                    fr = det.stage_front(levels[s][t], s, pri, on_map, pro)
                    fr["attn"] = torch.cat([fr["local"], pos], dim=-1)
                fronts.append(fr); priors.append(pri)
            # ---- branch B of the whole wavefront: pairs with earlier frames go through the decoder as one batch ----
            with_mem = [i for i, (t, s) in enumerate(pairs) if t > 0]
            feat = [fr["attn"][0] for fr in fronts]                            # [N,E] each
            if with_mem:
                tgt = torch.cat([feat[i] for i in with_mem], dim=0) if len(with_mem) > 1 else feat[with_mem[0]]
                mem = torch.cat([ring[pairs[i][1]][pairs[i][0]:pairs[i][0] + W].view(-1, E) for i in with_mem], dim=0)
                valid = torch.cat([ring_valid[pairs[i][1]][pairs[i][0]:pairs[i][0] + W].view(-1) for i in with_mem], dim=0)
                dec = det.transformer_Dec(tgt=tgt, memory=mem, memory_key_valid=valid, batch=len(with_mem))
                for j, i in enumerate(with_mem):
                    feat[i] = dec[j * N:(j + 1) * N]
            feat_all = torch.stack(feat, dim=0) if len(feat) > 1 else feat[0].unsqueeze(0)          # [k,N,E]
            pri_all = torch.cat(priors, dim=0) if len(priors) > 1 else priors[0]
            pred_b, lines_b = det._branch(feat_all, pri_all, True)                                   # [k,N,6+S]
            pb = pred_b.split(1, dim=0) if len(pairs) > 1 else [pred_b]
            with torch.no_grad():
                for i, (t, s) in enumerate(pairs):
                    _, rows_sorted, _ = K.lane_assign(pb[i][0].detach().contiguous(), lanes[t].contiguous(), det.img_w, det.img_h)
                    K.memory_tokens(fronts[i]["attn"][0].detach().contiguous(), rows_sorted.contiguous(),
                                    out=(ring[s][W + t], ring_valid[s][W + t]))
                todo = [i for i, (t, s) in enumerate(pairs) if s + 1 < S]
                if todo:
                    gate = torch.cat([fronts[i]["gate"].detach() for i in todo], dim=0)
                    la = torch.cat([fronts[i]["lines_a"].detach() for i in todo], dim=0)
                    lb = lines_b.detach() if len(todo) == len(pairs) else torch.cat([lines_b[i:i + 1].detach() for i in todo], dim=0)
                    pri_n, map_n = K.blend_priors(gate.contiguous(), la.contiguous(), lb.contiguous(), det.sample_x_indexs)
                    for j, i in enumerate(todo):
                        t, s = pairs[i]
                        nxt[(t, s + 1)] = (pri_n[j:j + 1], map_n[j:j + 1], fronts[i]["local"].detach())
            for i, (t, s) in enumerate(pairs):
                per_frame[t]["predictions_fir"][s] = fronts[i]["pred_a"]
                per_frame[t]["predictions_sec"][s] = pb[i]
                per_frame[t]["gates"][s] = fronts[i]["gate"]
        return self._clip_loss(per_frame, lanes)

    def lanes_from_device(self, kept_rows: torch.Tensor, nums: torch.Tensor):
        """One device->host copy per clip, then the host-side Lane construction (Router4OL.py:394-435)."""
        rows, n = kept_rows.cpu(), nums.cpu().tolist()
        return {"lane_lines": [self.detNet.predictions_to_pred(rows[t, :n[t]]) if n[t] else [] for t in range(len(n))]}

    def forward(self, inputs: dict):
        frame, lanes = inputs.values()
        if not frame.is_cuda:
            raise RuntimeError("phnet_amd runs on the GPU only: move the model and the clip to cuda")
        if frame.dim() == 5:                                   # [B,T,3,H,W]: several clips per step, head batched across clips
            if self.training:
                return self.forward_clips_train(frame, lanes)
            rows, nums, _ = self.infer_clips_device(frame)
            B = frame.shape[0]
            return [self.lanes_from_device(rows[b], nums[b]) for b in range(B)]
        if not self.training and self.sync_free_eval:
            rows, nums, _ = self.infer_device(frame)
            return self.lanes_from_device(rows, nums)
        T = frame.shape[0]
        self._begin_clip()
        if self.training:
            PF.DropoutStream.begin_step(frame.device)                          # fresh dropout masks for this clip's fwd + bwd
            if self.schedule != "frame" and self.batch_stage0:
                loss = (self.train_clip_wavefront if self.schedule == "wavefront" else self.train_clip_stage_major)(frame, lanes)
                self._begin_clip()
                return loss
        feats = self.backbone(frame)                                           # 3 x [T,h,w,C] NHWC
        last_cuts = []
        total_loss = 0.0
        clip_outputs = {"lane_lines": []}
        per_level = [f.split(1, dim=0) for f in feats]                         # SplitBackward = one cat per level, not T slice backwards
        stage0 = self.detNet.stage0_all_frames(feats[-1]) if self.batch_stage0 else None
        for t in range(T):
            cur = tuple(lv[t] for lv in per_level)
            outputs, cur_cut, gates = self.detNet(cur, last_cuts, None if stage0 is None else stage0[t])
            if self.training:
                matched, frame_loss = self.criterion(outputs, lanes[t:t + 1], gates)
                total_loss = total_loss + frame_loss
            else:
                d = torch.stack(gates, dim=0).mean(dim=0)
                lines = outputs["predictions_sec"][-1] * d + outputs["predictions_fir"][-1] * (1 - d)
                lane_lines, keep_inds, keep = self.detNet.get_lanes(lines, self.org_size, self.crop_size)
                clip_outputs["lane_lines"].append(lane_lines[0])
            with torch.no_grad():
                if self.training:
                    last_cuts.append(self.saveMemory(matched, cur_cut))
                else:
                    last_cuts.append(self.saveMemory4Test(keep_inds, keep, cur_cut))
                if t >= self.save_freq_max:
                    last_cuts.pop(0)
        self._begin_clip()
        return total_loss if self.training else clip_outputs

    def _tokens(self, feat, rows):
        """feat [N,1,E]; rows i64[L] positive anchors ascending, -1 padded  ->  (tokens [L+1,1,E], valid bool[L+1]):
        the positives in prior-index order, then the mean of all other anchors (Router4OL.py:563-584), fixed size."""
        from phnet_amd import hip_ops as K
        return K.memory_tokens(feat.contiguous(), rows.contiguous())

    def saveMemory(self, matched_indices, curr_cut):
        return [self._tokens(feat.detach(), rows) for rows, feat in zip(matched_indices, curr_cut)]

    def saveMemory4Test(self, keep_inds, keep, curr_cut):
        rows = torch.full((self.detNet.cfg.max_lanes,), -1, dtype=torch.int64, device=curr_cut[0].device)
        if len(keep):
            idx = torch.sort(torch.where(keep_inds)[0][keep])[0]
            rows[: idx.numel()] = idx
        return [self._tokens(feat.detach(), rows) for feat in curr_cut]
