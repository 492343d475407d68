"""Dynamic per-anchor feature enhancement (reference: libs/models/utils/dynamic_head.py:6-59)."""
import torch
import torch.nn as nn

from phnet_amd import functional as PF


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

    def forward(self, pro_feature: torch.Tensor, roi_feature: torch.Tensor) -> torch.Tensor:
        """pro_feature [B,N,C], roi_feature [B,N,P,C] -> [B,N,C]."""
        b, n, p, c = roi_feature.shape
        roi = roi_feature.reshape(b * n, p, c)
        pro = pro_feature.reshape(b * n, c)
        l1, l2, lo = self.dynamic_layer_1, self.dynamic_layer_2, self.out_layer
        w1 = PF.linear(PF.linear(pro, l1[0].weight, l1[0].bias), l1[1].weight, l1[1].bias).view(b * n, c, 2 * c)
        f = PF.bmm(roi, w1)
        f = PF.layer_norm(f, self.norm1.weight, self.norm1.bias, relu=True, eps=self.norm1.eps)
        w2 = PF.linear(PF.linear(f.detach().reshape(b * n, -1), l2[0].weight, l2[0].bias), l2[1].weight, l2[1].bias)
        f = PF.bmm(f, w2.view(b * n, 2 * c, c))
        f = PF.layer_norm(f, self.norm2.weight, self.norm2.bias, relu=True, eps=self.norm2.eps)
        f = PF.linear(PF.linear(f.reshape(b * n, -1), lo[0].weight, lo[0].bias), lo[1].weight, lo[1].bias)
        f = PF.layer_norm(f, self.norm3.weight, self.norm3.bias, eps=self.norm3.eps)
        return f.view(b, n, c)
