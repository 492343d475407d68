from .nms import nms

__all__ = ["nms"]
