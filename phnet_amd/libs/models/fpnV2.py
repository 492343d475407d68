"""Parameter container of the per-level-width FPN neck of the Router4OLV2 family (state_dict layout of
libs/models/fpnV2.py:70-100 with mmcv ConvModule = bare Conv2d(bias=True), sub-key `.conv`).  Compute: phnet_amd/trunk.py
(`encoder_fwd_v2`)."""
import torch.nn as nn

from .fpn import ConvModule


class FPN(nn.Module):
    def __init__(self, in_channels, out_channels, num_outs, start_level=0, end_level=-1, attention=False, **unused):
        super().__init__()
        if not isinstance(out_channels, (list, tuple)) or len(out_channels) != len(in_channels):
            raise TypeError("fpnV2.FPN takes one output width per level (options/options4OLV3.py:59-64)")
        if num_outs != len(in_channels) or start_level != 0 or end_level != -1 or attention:
            raise NotImplementedError("only the configuration of options/options4OLV3.py:59-64 is supported")
        self.in_channels, self.out_channels, self.num_outs = list(in_channels), list(out_channels), num_outs
        self.num_ins = len(in_channels)
        self.lateral_convs = nn.ModuleList(ConvModule(ci, co, 1) for ci, co in zip(in_channels, out_channels))
        self.fpn_convs = nn.ModuleList(ConvModule(co, co, 3, padding=1) for co in out_channels)
        # fpnV2.py:91-100: the coarser level is first projected to the finer level's width (1x1), then resized and added
        self.upsample_convs = nn.ModuleList(ConvModule(out_channels[i + 1], out_channels[i], 1) for i in range(self.num_ins - 1))
