"""Parameter containers of the ResNet trunk (state_dict layout of the reference's libs/models/resnet.py:49-95,
148-361).  Compute lives in phnet_amd/trunk.py (HIP kernels); these classes only own parameters/buffers, laid out so
that the kernels read them in place: conv weights are channels_last (= OHWI in memory)."""
import torch
import torch.nn as nn

_DEPTHS = {"resnet18": (2, 2, 2, 2), "resnet34": (3, 4, 6, 3)}


def _conv(cin, cout, k, stride=1, pad=0):
    m = nn.Conv2d(cin, cout, k, stride=stride, padding=pad, bias=False)
    nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
    m.weight.data = m.weight.data.contiguous(memory_format=torch.channels_last)
    return m


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, cin, cout, stride):
        super().__init__()
        self.conv1 = _conv(cin, cout, 3, stride, 1)
        self.bn1 = nn.BatchNorm2d(cout)
        self.conv2 = _conv(cout, cout, 3, 1, 1)
        self.bn2 = nn.BatchNorm2d(cout)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(_conv(cin, cout, 1, stride, 0), nn.BatchNorm2d(cout))


class ResNet(nn.Module):
    def __init__(self, depths, widths=(64, 128, 256, 512)):
        super().__init__()
        self.conv1 = _conv(3, 64, 7, 2, 3)
        self.bn1 = nn.BatchNorm2d(64)
        cin = 64
        for i, (d, w) in enumerate(zip(depths, widths)):
            blocks = []
            for b in range(d):
                blocks.append(BasicBlock(cin, w, 2 if (b == 0 and i > 0) else 1))
                cin = w
            setattr(self, f"layer{i + 1}", nn.Sequential(*blocks))
        self.expansion = 1


class ResNetWrapper(nn.Module):
    """Same constructor keywords as the reference wrapper (resnet.py:148-160); `pretrained=True` would need a
    network download (resnet.py:315) and is refused."""

    def __init__(self, resnet="resnet18", pretrained=False, replace_stride_with_dilation=(False, False, False),
                 out_conv=False, fea_stride=8, out_channel=128, in_channels=(64, 128, 256, 512), cfg=None):
        super().__init__()
        if resnet not in _DEPTHS:
            raise ValueError(f"phnet_amd supports {sorted(_DEPTHS)} trunks, got {resnet!r}")
        if pretrained:
            raise RuntimeError("pretrained=True fetches ImageNet weights over the network; load a checkpoint instead")
        if any(replace_stride_with_dilation) or out_conv:
            raise NotImplementedError("dilated / out_conv trunks are not on the hot path")
        self.in_channels = list(in_channels)
        self.model = ResNet(_DEPTHS[resnet], tuple(in_channels))
