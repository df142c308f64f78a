"""ctypes binding of libyogo_hip.so (the C ABI declared in include/yogo_hip.h).

There is deliberately NO fallback: if the shared library is missing or a call fails, a RuntimeError is raised.
PyTorch is used by the callers only for device memory, streams and autograd plumbing.
"""
from __future__ import annotations

import ctypes
import os
import re
from typing import Dict, List, Tuple

import torch

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "lib", "libyogo_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_PKG), "include", "yogo_hip.h")

_CTYPES = {
    "int": ctypes.c_int,
    "float": ctypes.c_float,
    "double": ctypes.c_double,
    "long long": ctypes.c_longlong,
    "size_t": ctypes.c_size_t,
    "yogo_stream_t": ctypes.c_void_p,
}


def parse_header(path: str = HEADER_PATH) -> Dict[str, Tuple[str, List[str]]]:
    """{function name: (return type, [argument types])} for every prototype in the header."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    out: Dict[str, Tuple[str, List[str]]] = {}
    for m in re.finditer(r"(?:^|;|\{)\s*((?:const\s+)?[A-Za-z_][\w ]*?[\w\*])\s+(\w+)\s*\(([^()]*)\)\s*(?=;)", src, flags=re.S):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        if not name.startswith("yogo_"):
            continue
        argt: List[str] = []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                if "*" in a:
                    argt.append("ptr")
                else:
                    argt.append(" ".join(a.split(" ")[:-1]))
        out[name] = (ret, argt)
    return out


_lib = None
_protos: Dict[str, Tuple[str, List[str]]] = {}


def lib() -> ctypes.CDLL:
    global _lib, _protos
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"yogo_amd: HIP extension not built ({LIB_PATH} missing). Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `bash yogo_amd/csrc/build.sh`. There is no CPU fallback."
        )
    _protos = parse_header()
    _lib = bind(LIB_PATH, _protos)
    return _lib


def bind(path: str, protos: Dict[str, Tuple[str, List[str]]]) -> ctypes.CDLL:
    """dlopen `path` and give every prototype of the header its argument types (AttributeError if the library does not export a
    declared symbol)"""
    L = ctypes.CDLL(path)
    for name, (ret, args) in protos.items():
        fn = getattr(L, name)
        fn.restype = ctypes.c_char_p if "char" in ret else ctypes.c_int
        fn.argtypes = [ctypes.c_void_p if a == "ptr" else _CTYPES[a] for a in args]
    return L


def prototypes() -> Dict[str, Tuple[str, List[str]]]:
    lib()
    return dict(_protos)


def _ptr(t):
    if t is None:
        return None
    if isinstance(t, torch.Tensor):
        # every tensor argument of the C ABI is a DEVICE pointer (host buffers travel as ctypes addresses): a host tensor here
        # would make a kernel dereference host memory -- a GPU memory fault that takes the process down instead of an error
        if not t.is_cuda:
            raise RuntimeError(f"yogo_amd: a host tensor {tuple(t.shape)} {t.dtype} was handed to a HIP entry point; the hot path "
                               "takes device tensors only (was a buffer left on the CPU?)")
        return t.data_ptr()
    return t


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def call(name: str, *args):
    """Call a C-ABI entry point; tensors are passed as raw device pointers. Raises RuntimeError on failure."""
    L = lib()
    fn = getattr(L, name)
    rc = fn(*[_ptr(a) for a in args])
    if rc != 0:
        msg = L.yogo_hip_last_error()
        raise RuntimeError(f"{name} failed (code {rc}): {msg.decode() if msg else '?'}")


def query_size(name: str, *args) -> int:
    out = ctypes.c_size_t(0)
    call(name, *args, ctypes.addressof(out))
    return int(out.value)


def query_ints(name: str, n: int, *args) -> List[int]:
    outs = [ctypes.c_int(0) for _ in range(n)]
    call(name, *args, *[ctypes.addressof(o) for o in outs])
    return [int(o.value) for o in outs]


def require_cuda(t: torch.Tensor, what: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(
            f"yogo_amd: {what} must live on an MI355X device (got {t.device}); the hot path is HIP-only, there is no CPU fallback"
        )


def launch_log(enable: bool) -> None:
    """start (and clear) / stop the library's launch log (include/yogo_hip.h: yogo_hip_launch_log)"""
    call("yogo_hip_launch_log", 1 if enable else 0)


def read_launch_log() -> List[str]:
    """the recorded lines "<kernel instantiation> | <planner parameters>", one per kernel launch"""
    need = ctypes.c_size_t(0)
    call("yogo_hip_launch_log_read", None, 0, ctypes.addressof(need))
    buf = ctypes.create_string_buffer(int(need.value) + 1)
    call("yogo_hip_launch_log_read", ctypes.addressof(buf), len(buf), ctypes.addressof(need))
    return [ln for ln in buf.value.decode().split("\n") if ln]
