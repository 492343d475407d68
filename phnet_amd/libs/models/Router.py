"""Adaptive routing gate (reference: libs/models/Router.py:39-81) on the HIP LayerNorm / depth-wise / GEMM kernels."""
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
