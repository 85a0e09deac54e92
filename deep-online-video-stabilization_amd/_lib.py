"""ctypes binding of libstabnet_hip.so (include/stabnet_hip.h).  There is no CPU fallback: if the library is
missing or a call fails, this raises."""
from __future__ import annotations

import ctypes
import os
import re

# torch bundles its own HIP runtime (libamdhip64.so.7, same SONAME as /opt/rocm's).  Import it BEFORE loading the
# library so the process has exactly one runtime -- the one that owns torch's device memory and streams.
import torch  # noqa: F401  (plumbing: device memory, streams)

_HERE = os.path.dirname(os.path.abspath(__file__))
# STABNET_LIB: debug switch for A/B runs against another build of the library (symbols it lacks are skipped)
LIB_PATH = os.environ.get("STABNET_LIB") or os.path.join(_HERE, "libstabnet_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "stabnet_hip.h")

_lib = None


class StabnetError(RuntimeError):
    pass


_CTYPE = {
    "int": ctypes.c_int, "float": ctypes.c_float, "size_t": ctypes.c_size_t, "long": ctypes.c_long,
    "int64_t": ctypes.c_int64, "double": ctypes.c_double,
}


def declared_symbols(header_path: str = HEADER_PATH):
    """[(name, restype, [argtype...])] parsed from the C header (the single source of the ABI)."""
    txt = open(header_path).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    out = []
    for m in re.finditer(r"^\s*(const char\s*\*|int|size_t|void|double)\s+(stabnet_\w+)\s*\(([^;]*?)\)\s*;", txt, flags=re.S | re.M):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        argtypes = []
        args = " ".join(args.split())
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "*" in a:
                    argtypes.append(ctypes.c_void_p)
                else:
                    base = a.replace("const ", "").split()[0]
                    argtypes.append(_CTYPE[base])
        restype = {"int": ctypes.c_int, "size_t": ctypes.c_size_t, "void": None, "double": ctypes.c_double}.get(ret, ctypes.c_char_p)
        out.append((name, restype, argtypes))
    return out


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise StabnetError(
                "libstabnet_hip.so not built (%s missing): run `python -m stabnet_amd.build` "
                "(or __graft_entry__.build()); there is no CPU fallback" % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        for name, restype, argtypes in declared_symbols():
            if os.environ.get("STABNET_LIB") and not hasattr(L, name):
                continue
            fn = getattr(L, name)          # AttributeError if the header and the library disagree
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = L
    return _lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = lib().stabnet_last_error()
        raise StabnetError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else "?"))


def call(name: str, *args, device=None):
    """Call an int-returning entry point and raise on a non-zero status.  `device` (torch.device of the tensors whose
    pointers are passed) makes that GPU current for the call: kernels launch on the CURRENT device, so a pointer of
    cuda:1 handed over while cuda:0 is current would fault."""
    fn = getattr(lib(), name)
    if device is not None and device.type == "cuda":
        with torch.cuda.device(device):
            check(fn(*args), name)
    else:
        check(fn(*args), name)
