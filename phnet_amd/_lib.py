"""ctypes binding of libphnet_hip.so.  argtypes come from include/phnet_hip.h (single source of truth).

The product path has no fallback: if the library is missing (not built) every op raises.
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("PHNET_LIB") or os.path.join(_HERE, "lib", "libphnet_hip.so")      # PHNET_LIB: kernel experiments only
HEADER = os.path.join(os.path.dirname(_HERE), "include", "phnet_hip.h")
TUNING_HEADER = os.path.join(os.path.dirname(_HERE), "include", "phnet_hip_tuning.h")    # benchmark / A-B switches, not the boundary

_CTYPES = {"int32_t": ctypes.c_int32, "int64_t": ctypes.c_int64, "uint64_t": ctypes.c_uint64,
           "float": ctypes.c_float, "int": ctypes.c_int, "void": None}

ERRORS = {-1: "PHNET_ERR_ARG (bad shape / null pointer / unsupported size)",
          -2: "PHNET_ERR_WORKSPACE (workspace too small)",
          -3: "PHNET_ERR_LAUNCH (HIP launch error)"}


def declared_functions(header: str = None):
    """[(name, restype, [argtypes])] for every prototype in the public header (+ the tuning header when none is named)."""
    if header is None:
        return declared_functions(HEADER) + declared_functions(TUNING_HEADER)
    text = open(header).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    out = []
    for m in re.finditer(r"\b(int|uint64_t)\s+(phnet_\w+)\s*\(([^)]*)\)\s*;", text):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        argtypes = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "*" in a:
                    argtypes.append(ctypes.c_void_p)
                else:
                    argtypes.append(_CTYPES[a.replace("const", "").split()[0]])
        out.append((name, _CTYPES[ret], argtypes))
    return out


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise RuntimeError(
                f"{SO_PATH} is missing: the HIP extension has not been built (python -m phnet_amd.build). "
                "phnet_amd has no CPU / eager fallback.")
        # PyTorch ships its own copy of the HIP runtime: it must be in the process BEFORE this library is loaded, so that the
        # loader binds libphnet_hip.so to that same runtime (two runtimes = our kernels registered with one, torch's streams
        # and device pointers owned by the other: every launch fails)
        import torch  # noqa: F401
        handle = ctypes.CDLL(SO_PATH)
        for name, restype, argtypes in declared_functions():
            fn = getattr(handle, name)          # AttributeError if the .so lacks a declared symbol
            fn.restype = restype
            fn.argtypes = argtypes
        if handle.phnet_abi_version() != 1:
            raise RuntimeError("libphnet_hip.so ABI version mismatch")
        _lib = handle
    return _lib


def check(rc: int, what: str):
    if rc != 0:
        raise RuntimeError(f"{what} failed: {ERRORS.get(rc, rc)}")
