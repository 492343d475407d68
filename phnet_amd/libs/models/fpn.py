"""Parameter container of the FPN neck (state_dict layout of libs/models/fpn.py:70-106 with mmcv ConvModule =
bare Conv2d(bias=True), sub-key `.conv`).  Compute: phnet_amd/trunk.py."""
import torch
import torch.nn as nn


class ConvModule(nn.Module):
    def __init__(self, cin, cout, k, padding=0):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, padding=padding, bias=True)
        self.conv.weight.data = self.conv.weight.data.contiguous(memory_format=torch.channels_last)


class FPN(nn.Module):
    def __init__(self, in_channels, out_channels, num_outs, attention=False, **unused):
        super().__init__()
        if num_outs != len(in_channels) or attention:
            raise NotImplementedError("only the configuration of options/options4OL.py:58-61 is on the hot path")
        self.in_channels, self.out_channels, self.num_outs = list(in_channels), out_channels, num_outs
        self.lateral_convs = nn.ModuleList(ConvModule(c, out_channels, 1) for c in in_channels)
        self.fpn_convs = nn.ModuleList(ConvModule(out_channels, out_channels, 3, padding=1) for _ in in_channels)
