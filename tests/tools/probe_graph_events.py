"""Probe: can timing events be recorded inside a captured hipGraph (hipEventRecordWithFlags external)?"""
import ctypes, torch
hip = ctypes.CDLL("libamdhip64.so")
hip.hipEventCreate.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
hip.hipEventRecordWithFlags.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint]
hip.hipEventRecord.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
hip.hipEventElapsedTime.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_void_p, ctypes.c_void_p]
hip.hipEventSynchronize.argtypes = [ctypes.c_void_p]
evs = []
for _ in range(4):
    e = ctypes.c_void_p(); assert hip.hipEventCreate(ctypes.byref(e)) == 0; evs.append(e)
x = torch.randn(4096, 4096, device="cuda")
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    y = x @ x
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
for flag in (1, 0):
    try:
        with torch.cuda.graph(g):
            st = torch.cuda.current_stream().cuda_stream
            r0 = hip.hipEventRecordWithFlags(evs[0], st, flag)
            y = x @ x
            r1 = hip.hipEventRecordWithFlags(evs[1], st, flag)
            z = y + 1
            r2 = hip.hipEventRecordWithFlags(evs[2], st, flag)
        print("flag", flag, "record rc", r0, r1, r2)
        g.replay(); torch.cuda.synchronize()
        t = ctypes.c_float()
        rc = hip.hipEventElapsedTime(ctypes.byref(t), evs[0], evs[1]); print("elapsed rc", rc, "matmul ms", t.value)
        rc = hip.hipEventElapsedTime(ctypes.byref(t), evs[1], evs[2]); print("elapsed rc", rc, "add ms", t.value)
        break
    except Exception as e:
        print("flag", flag, "failed:", type(e).__name__, str(e)[:200])
        g = torch.cuda.CUDAGraph()
