"""Adaptive routing gates (reference: libs/models/Router.py:39-81 AdaptiveRouter4Lane, :83-132 AdaptiveRouter4LaneV2) on the HIP kernels."""
import copy

import torch
import torch.nn as nn

from phnet_amd import functional as PF


class AdaptiveRouter4Lane(nn.Module):
    def __init__(self, num_priors=240, features_channels=64, num_points=36, out_channels=1, reduction=4, stages=3):
        super().__init__()
        width = features_channels * num_points
        self.inp = width // reduction
        mlp = nn.Sequential(nn.Linear(width, self.inp), nn.ReLU(), nn.Linear(self.inp, out_channels), nn.ReLU())
        gain = nn.init.calculate_gain("tanh")
        for m in mlp:
            if isinstance(m, nn.Linear):
                nn.init.xavier_uniform_(m.weight, gain=gain)
        self.layers = nn.ModuleList(copy.deepcopy(mlp) for _ in range(stages))
        norm = nn.LayerNorm([features_channels, num_points])
        self.pre_norm = nn.ModuleList(copy.deepcopy(norm) for _ in range(stages))

        def dw():
            return nn.Conv2d(num_priors, num_priors, 3, padding=1, groups=num_priors)
        block = nn.Sequential(dw(), copy.deepcopy(norm), nn.ReLU(), dw(), copy.deepcopy(norm))
        net = nn.ModuleList(copy.deepcopy(block) for _ in range(4))
        self.DWNets = nn.ModuleList(copy.deepcopy(net) for _ in range(stages))

    def forward(self, xs: torch.Tensor, stage: int, thres: float = 0.5) -> torch.Tensor:
        """xs [B,N,C,P] (detached by the caller; B frames share the N per-anchor filters) -> [B,N,1] in [0.5, 1)."""
        b, n, c, p = xs.shape
        pn = self.pre_norm[stage]
        params = [pn.weight, pn.bias]
        for blk in self.DWNets[stage]:
            params += [blk[0].weight, blk[0].bias, blk[1].weight, blk[1].bias, blk[3].weight, blk[3].bias, blk[4].weight, blk[4].bias]
        x = PF.gate_stack(xs.reshape(b * n, c, p), params, eps=pn.eps, anchors=n)      # one fused launch (csrc/gate.hip)
        mlp = self.layers[stage]
        h = PF.linear(x.reshape(b * n, c * p), mlp[0].weight, mlp[0].bias, relu=True)
        if mlp[2].out_features == 1:                                     # ReLU before the sigmoid (Router.py:76-80)
            return PF.gate_tail(h, mlp[2].weight, mlp[2].bias).view(b, n, 1)
        h = PF.linear(h, mlp[2].weight, mlp[2].bias, relu=True)
        return torch.sigmoid(h).view(b, n, -1)


class _ConvBN1d(nn.Module):
    """mmcv ConvModule(conv_cfg=Conv1d, norm_cfg=BN1d) as a parameter container: `.conv` (no bias: bias='auto' with a norm
    layer) and `.bn`; the activation is ReLU (mmcv's default act_cfg).  Router.py:93-106."""

    def __init__(self, cin, cout, k, padding=0):
        super().__init__()
        self.conv = nn.Conv1d(cin, cout, k, padding=padding, bias=False)
        self.bn = nn.BatchNorm1d(cout)

    def folded(self):
        """Eval-mode BatchNorm as y = conv * s + t."""
        bn = self.bn
        s = bn.weight.detach() * torch.rsqrt(bn.running_var + bn.eps)
        return self.conv.weight.detach().contiguous(), s.contiguous(), (bn.bias.detach() - bn.running_mean * s).contiguous()


class AdaptiveRouter4LaneV2(nn.Module):
    """Gate of the Router4OLV2 family: per stage Conv1d(C -> C/r, k3) + BN1d + ReLU, Conv1d(C/r -> C/C_last, k1) + BN1d + ReLU,
    Flatten, Linear(C*P/C_last -> P); score = sigmoid(mean of the P outputs).  One launch per call (csrc/v2head.hip), inference
    only.  The reference's constructor does not take `num_priors` / `out_channels` although its only caller passes them
    (Router4OLV2.py:120-124 vs Router.py:84: TypeError as shipped); they are accepted and ignored here."""

    def __init__(self, num_priors=None, features_channels=(32, 16, 8), num_points=(24, 48, 96), out_channels=None, reduction=2, stages=3):
        super().__init__()
        features_channels, num_points = list(features_channels), list(num_points)
        self.inplane = features_channels[0] * num_points[0]
        assert all(c * p == self.inplane for c, p in zip(features_channels, num_points))
        last = features_channels[-1]
        self.layers = nn.ModuleList()
        gain = nn.init.calculate_gain("tanh")
        for s in range(stages):
            c, p = features_channels[s], num_points[s]
            lin = nn.Linear(c * p // last, p)
            nn.init.xavier_uniform_(lin.weight, gain=gain)
            self.layers.append(nn.Sequential(_ConvBN1d(c, c // reduction, 3, padding=1), _ConvBN1d(c // reduction, c // last, 1),
                                             nn.Flatten(1), lin))

    def forward(self, xs: torch.Tensor, stage: int, thres: float = 0.5, out=None) -> torch.Tensor:
        """xs [B,N,C,P] -> [B,N,1] in (0,1).  out (optional): a [B*N] slice to write the scores into."""
        if self.training:
            raise NotImplementedError("the Router4OLV2 family is inference-only (BatchNorm1d in eval form; the reference's training "
                                      "path cannot run as shipped)")
        from phnet_amd import hip_ops as K
        b, n, c, p = xs.shape
        seq = self.layers[stage]
        w1, s1, t1 = seq[0].folded()
        w2, s2, t2 = seq[1].folded()
        score = K.gate_v2_fwd(xs.reshape(b * n, c, p).contiguous(), w1, s1, t1, w2.reshape(w2.shape[0], -1), s2, t2,
                              seq[3].weight.detach().contiguous(), seq[3].bias.detach().contiguous(), out)
        return score.view(b, n, 1)
