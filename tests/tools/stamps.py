"""Diagnostic: per-phase cycle stamps of the implicit-GEMM loop.  Needs a library built with -DPHNET_STAMPS
(hipcc ... -DPHNET_STAMPS -c csrc/conv.hip; link as phnet_amd/lib/libphnet_stamps.so) and PHNET_LIB pointing at it."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from phnet_amd import hip_ops as K
from phnet_amd._lib import lib
from bench_conv import SHAPES

h = ctypes.CDLL(os.environ["PHNET_LIB"])
buf = (ctypes.c_ulonglong * 8)()
for name, N, Hi, Wi, Ci, Co, R, st, pad in SHAPES[:4] + SHAPES[5:7]:
    x = torch.randn(N, Hi, Wi, Ci, device="cuda"); w = torch.randn(Co, R, R, Ci, device="cuda") * 0.05
    for what in ("fwd", "dgrad"):
        ho, wo = K.conv_out_hw(Hi, Wi, R, R, st, pad)
        gy = torch.randn(N, ho, wo, Co, device="cuda")
        fn = (lambda: K.conv2d_fwd(x, w, None, st, pad)) if what == "fwd" else (lambda: K.conv2d_dgrad(gy, w, (Hi, Wi), st, pad))
        fn(); torch.cuda.synchronize()
        h.phnet_debug_stamps(None, 1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        h.phnet_debug_stamps(buf, 0)
        ld, mma, fill, bar, nt = [buf[i] for i in range(5)]
        tot = ld + mma + fill + bar
        print(f"{name:18s} {what:5s} {e0.elapsed_time(e1)*1e3:7.1f} us | wave-iterations {nt:8d} | cycles per wave-iteration: load-issue {ld/nt:6.0f}  "
              f"frag-read+MFMA {mma/nt:6.0f}  wait+split+LDS-fill {fill/nt:6.0f}  barrier {bar/nt:6.0f}  total {tot/nt:6.0f}")
