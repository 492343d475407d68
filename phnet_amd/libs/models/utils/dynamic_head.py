"""Dynamic per-anchor feature enhancement (reference: libs/models/utils/dynamic_head.py:6-59 DynamicConv, :61-112 DynamicConvV2)."""
import torch
import torch.nn as nn

from phnet_amd import functional as PF
from phnet_amd.arena import grad_sink


class DynamicConv(nn.Module):
    def __init__(self, feat_size=36, inplanes=64, early_return=False):
        super().__init__()
        if early_return:
            raise NotImplementedError("early_return=True is not used on the hot path (Router4OL.py:110)")
        c = inplanes
        self.hidden_dim, self.dim_dynamic, self.num_params = c, 2 * c, 2 * c * c
        self.dynamic_layer_1 = nn.Sequential(nn.Linear(c, self.num_params // 8), nn.Linear(self.num_params // 8, self.num_params))
        self.dynamic_layer_2 = nn.Sequential(nn.Linear(2 * c * feat_size, self.num_params // 8),
                                             nn.Linear(self.num_params // 8, self.num_params))
        self.norm1 = nn.LayerNorm(2 * c)
        self.norm2 = nn.LayerNorm(c)
        self.activation = nn.ReLU()
        self.out_layer = nn.Sequential(nn.Linear(c * feat_size, 6 * c), nn.Linear(6 * c, c))
        self.norm3 = nn.LayerNorm(c)
        self._folded = None          # per-clip cache of folded Linear->Linear pairs (see _fold)
        self._sink_pool = None

    def begin_clip(self, sink_pool=None):
        self._folded = None
        self._sink_pool = sink_pool

    def _fold(self, name: str, seq: nn.Sequential):
        """A Linear->Linear pair with nothing in between (dynamic_head.py:16-17, 27-28) is one affine map:
        y = x (W2 W1)^T + (W2 b1 + b2).  Folding it once per clip (the weights are constant inside a clip, the pair is
        applied 5 times per clip) turns 2*M*K*H + 2*M*H*N FLOPs per call into 2*M*K*N - 16x fewer for
        dynamic_layer_1 (K=64, H=1024, N=8192) - and the fold itself is an ordinary GEMM on the HIP kernels, so
        autograd carries the gradients back to both original weight matrices.  Same result up to fp32 re-association."""
        if self._folded is None:
            self._folded = {}
        if name not in self._folded:
            l1, l2 = seq[0], seq[1]
            w_eff_t = PF.linear(l1.weight.t().contiguous(), l2.weight)              # [K, N] = (W2 W1)^T
            b_eff = PF.linear(l1.bias.unsqueeze(0), l2.weight, l2.bias).view(-1)    # W2 b1 + b2 (a view: `[0]` costs a zero-fill + copy in the backward)
            self._folded[name] = (grad_sink(w_eff_t.t().contiguous(), self._sink_pool), grad_sink(b_eff.contiguous(), self._sink_pool))
        return self._folded[name]

    def forward(self, pro_feature: torch.Tensor, roi_feature: torch.Tensor) -> torch.Tensor:
        """pro_feature [B,N,C], roi_feature [B,N,P,C] -> [B,N,C]."""
        b, n, p, c = roi_feature.shape
        roi = roi_feature.reshape(b * n, p, c)
        pro = pro_feature.reshape(b * n, c)
        l1, l2, lo = self.dynamic_layer_1, self.dynamic_layer_2, self.out_layer
        we, be = self._fold("dynamic_layer_1", l1)
        w1 = PF.linear(pro, we, be).view(b * n, c, 2 * c)
        f = PF.dyn_bmm_ln_relu(roi, w1, self.norm1.weight, self.norm1.bias, self.norm1.eps)
        w2 = PF.linear(PF.linear(f.detach().reshape(b * n, -1), l2[0].weight, l2[0].bias), l2[1].weight, l2[1].bias)
        f = PF.dyn_bmm_ln_relu(f, w2.view(b * n, 2 * c, c), self.norm2.weight, self.norm2.bias, self.norm2.eps)
        wo, bo = self._fold("out_layer", lo)
        f = PF.linear(f.reshape(b * n, -1), wo, bo)
        f = PF.layer_norm(f, self.norm3.weight, self.norm3.bias, eps=self.norm3.eps)
        return f.view(b, n, c)


class DynamicConvV2(nn.Module):
    """Per-level dynamic head of the Router4OLV2 family (dynamic_head.py:61-112): inplanes C in {64,32,16}, feat_size P in
    {24,48,96} (C*P = 1536), proposal / output width `outplanes` = 256.  Inference only: the Linear->Linear pairs are folded
    once per clip, the two per-anchor products run on the run-time-shape kernel of csrc/v2head.hip."""

    def __init__(self, feat_size=24, inplanes=32, outplanes=256, early_return=False):
        super().__init__()
        if early_return:
            raise NotImplementedError("early_return=True is not used (Router4OLV2.py:111-114)")
        c = inplanes
        self.inplanes, self.feat_size, self.dim_dynamic, self.outplanes = c, feat_size, 2 * c, outplanes
        self.num_params = c * 2 * c
        q = self.num_params // 4
        self.dynamic_layer_1 = nn.Sequential(nn.Linear(outplanes, q), nn.Linear(q, self.num_params))
        self.dynamic_layer_2 = nn.Sequential(nn.Linear(2 * c * feat_size, q), nn.Linear(q, self.num_params))
        self.norm1 = nn.LayerNorm(2 * c)
        self.norm2 = nn.LayerNorm(c)
        self.activation = nn.ReLU()
        self.out_layer = nn.Sequential(nn.Linear(c * feat_size, 2 * outplanes), nn.Linear(2 * outplanes, outplanes))
        self.norm3 = nn.LayerNorm(outplanes)
        self._folded = None

    def begin_clip(self, sink_pool=None):
        pass

    def _fold(self, name: str, seq: nn.Sequential):
        """Linear -> Linear with nothing in between = one affine map (as in DynamicConv._fold).  Only `dynamic_layer_1` is
        folded (256 -> C*C/2 -> 2*C*C: 8x fewer FLOPs per call at C = 64); the weights are constant at inference, so the fold
        is kept until a parameter changes."""
        l1, l2 = seq[0], seq[1]
        ver = (l1.weight._version, l1.bias._version, l2.weight._version, l2.bias._version, l1.weight.data_ptr(), l2.weight.data_ptr())
        if self._folded is None:
            self._folded = {}
        hit = self._folded.get(name)
        if hit is None or hit[0] != ver:
            w_eff_t = PF.linear(l1.weight.detach().t().contiguous(), l2.weight.detach())
            b_eff = PF.linear(l1.bias.detach().unsqueeze(0), l2.weight.detach(), l2.bias.detach())[0]
            hit = (ver, w_eff_t.t().contiguous(), b_eff.contiguous())
            self._folded[name] = hit
        return hit[1], hit[2]

    def forward(self, pro_feature: torch.Tensor, roi_feature: torch.Tensor) -> torch.Tensor:
        """pro_feature [B,N,256], roi_feature [B,N,P,C] -> [B,N,256]."""
        if self.training:
            raise NotImplementedError("the Router4OLV2 family is inference-only")
        from phnet_amd import hip_ops as K
        b, n, p, c = roi_feature.shape
        roi = roi_feature.reshape(b * n, p, c).contiguous()
        pro = pro_feature.reshape(b * n, -1)
        l2, lo = self.dynamic_layer_2, self.out_layer
        we, be = self._fold("dynamic_layer_1", self.dynamic_layer_1)
        w1 = PF.linear(pro, we, be).view(b * n, c, 2 * c)
        f = K.dyn_bmm_ln_relu_fwd_any(roi, w1, self.norm1.weight.detach(), self.norm1.bias.detach(), self.norm1.eps)
        w2 = PF.linear(PF.linear(f.reshape(b * n, -1), l2[0].weight, l2[0].bias), l2[1].weight, l2[1].bias).view(b * n, 2 * c, c)
        f = K.dyn_bmm_ln_relu_fwd_any(f, w2, self.norm2.weight.detach(), self.norm2.bias.detach(), self.norm2.eps)
        f = PF.linear(PF.linear(f.reshape(b * n, -1), lo[0].weight, lo[0].bias), lo[1].weight, lo[1].bias)
        f = PF.layer_norm(f, self.norm3.weight, self.norm3.bias, eps=self.norm3.eps)
        return f.view(b, n, -1)
