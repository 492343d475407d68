"""Duck-typed configuration object (attribute access + .haskey), equivalent for the hot path to what
libs/utils/config.py:Config gives the reference model (Router4OL.py:25,42-46,446,463-464,511-513)."""
from types import SimpleNamespace


class Cfg(dict):
    def __getattr__(self, k):
        try:
            v = self[k]
        except KeyError:
            raise AttributeError(k)
        return Cfg(v) if isinstance(v, dict) and not isinstance(v, Cfg) else v

    def haskey(self, k):
        return k in self


def make_cfg(img_h=320, img_w=800, arch="resnet34", num_points=36, num_priors=240, max_lanes=4, save_freq_max=8,
             conf_threshold=0.5, nms_thres=50, cls_weight=8.0, reg_weight=0.5, iou_weight=1.5) -> Cfg:
    """Keys/values of options/options4OL.py with the BASELINE geometry."""
    return Cfg(img_h=img_h, img_w=img_w, num_points=num_points, num_priors=num_priors, max_lanes=max_lanes,
               save_freq_max=save_freq_max,
               backbone=dict(resnet=arch, pretrained=False, replace_stride_with_dilation=[False, False, False], out_conv=False),
               neck=dict(in_channels=[128, 256, 512], out_channels=64, num_outs=3, attention=False),
               cls_weight=cls_weight, reg_weight=reg_weight, iou_weight=iou_weight,
               test_parameters=dict(conf_threshold=conf_threshold, nms_thres=nms_thres, nms_topk=max_lanes),
               dscfg=SimpleNamespace(crop_size=480, org_height=1280, org_width=1920))


def make_cfg_v2(img_h=320, img_w=800, arch="resnet18", num_points=72, num_priors=240, max_lanes=4, save_freq=1, save_freq_max=5,
                conf_threshold=0.5, nms_thres=50) -> Cfg:
    """Keys/values of options/options4OLV3.py (the Router4OLV2 family run by testOLV3.py) at the BASELINE frame size."""
    return Cfg(img_h=img_h, img_w=img_w, num_points=num_points, num_priors=num_priors, max_lanes=max_lanes,
               save_freq=save_freq, save_freq_max=save_freq_max,
               backbone=dict(resnet=arch, pretrained=False, replace_stride_with_dilation=[False, False, False], out_conv=False),
               neck=dict(in_channels=[64, 128, 256], out_channels=[16, 32, 64], num_outs=3, start_level=0, end_level=-1, attention=False),
               test_parameters=dict(conf_threshold=conf_threshold, nms_thres=nms_thres, nms_topk=max_lanes),
               dscfg=SimpleNamespace(crop_size=480, org_height=1280, org_width=1920))
