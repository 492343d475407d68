"""Drop-in for the reference's `libs.ops.nms` (libs/ops/nms.py:32-33 -> pybind11 nms_impl.nms_forward).
Same signature and 3-tuple return; the work is one launch of the HIP lane-NMS kernel on the current stream."""
import torch

from phnet_amd import hip_ops as K


def nms(boxes: torch.Tensor, scores: torch.Tensor, overlap: float, top_k: int):
    """boxes [K, 5+n_offsets] float32 CUDA contiguous, scores [K] -> (keep [K] i64, num_to_keep 0-d i64, parent [K] i64)."""
    if not boxes.is_cuda:
        raise RuntimeError("boxes must be a CUDA tensor")                # csrc/nms.cpp:40 CHECK_CUDA
    if not boxes.is_contiguous():
        raise RuntimeError("boxes must be contiguous")                   # csrc/nms.cpp:41 CHECK_CONTIGUOUS
    if boxes.dim() != 2 or boxes.shape[1] < 6:
        raise RuntimeError("Wrong number of offsets")                    # nms_kernel.cu:154 (PROP_SIZE check)
    return K.lane_nms(boxes.float(), scores.float().contiguous(), overlap, top_k)
