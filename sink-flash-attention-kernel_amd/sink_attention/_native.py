"""ctypes binding of libsfa.so (C ABI in include/sfa.h).

The shared library holds the hand-written gfx950 HIP kernels.  There is NO
fallback: if the library is missing or a call fails, this module raises.
Build it with ``make -C sink-flash-attention-kernel_amd`` (or
``python -c "import __graft_entry__ as g; g.build()"`` from the repo root).
"""
from __future__ import annotations

import ctypes
import os
import struct
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SFA_LIB_PATH") or os.path.join(_HERE, "libsfa.so")   # env: A/B builds in development

SFA_DTYPE = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}
FLAG_FORCE_GENERIC = 0x1
FLAG_DECODE_ONE_PASS = 0x4
FLAG_BWD_OVERLAP = 0x8        # dQ kernel of a small grid on the library's side stream (include/sfa.h)
FLAG_BWD_DKDV_ASM = 0x10      # dispatch override: hand-placed dK/dV kernel
FLAG_BWD_DKDV_WS = 0x20       # dispatch override: wave-specialised compiled dK/dV kernel
ABI_VERSION = 2

# flags every backward of the ops passes to sfa_bwd / sfa_bwd_varlen (and to the workspace query): the overlap is ON by
# default here - the C entry points do nothing of the kind unless asked
_bwd_flags = FLAG_BWD_OVERLAP


def set_backward_options(overlap=None, dkdv=None):
    """Process-wide options of the ops' backward pass.  overlap: True / False - dQ kernel of small grids on the library's
    side stream (default True).  dkdv: None (the library's rule) / "asm" (hand-placed kernel) / "ws" (wave-specialised
    compiled kernel): the dispatch override the parity tests use.  Returns the previous (overlap, dkdv)."""
    global _bwd_flags
    prev = (bool(_bwd_flags & FLAG_BWD_OVERLAP),
            "asm" if _bwd_flags & FLAG_BWD_DKDV_ASM else ("ws" if _bwd_flags & FLAG_BWD_DKDV_WS else None))
    f = _bwd_flags
    if overlap is not None:
        f = (f | FLAG_BWD_OVERLAP) if overlap else (f & ~FLAG_BWD_OVERLAP)
    if dkdv is not None or overlap is None:
        f &= ~(FLAG_BWD_DKDV_ASM | FLAG_BWD_DKDV_WS)
        if dkdv == "asm":
            f |= FLAG_BWD_DKDV_ASM
        elif dkdv == "ws":
            f |= FLAG_BWD_DKDV_WS
        elif dkdv not in (None, "rule"):
            raise ValueError("dkdv must be None, 'rule', 'asm' or 'ws'")
    _bwd_flags = f
    return prev


def bwd_flags(flags=0) -> int:
    return flags | _bwd_flags


class SfaTensor(ctypes.Structure):
    _fields_ = [
        ("ptr", ctypes.c_void_p),
        ("shape", ctypes.c_int64 * 4),
        ("stride", ctypes.c_int64 * 4),
        ("dtype", ctypes.c_int32),
        ("reserved", ctypes.c_int32),
    ]


_lib = None
_lock = threading.Lock()


def _bind(lib):
    P = ctypes.POINTER(SfaTensor)
    vp, i32, u32, f32, sz, i64 = (ctypes.c_void_p, ctypes.c_int, ctypes.c_uint, ctypes.c_float, ctypes.c_size_t,
                                   ctypes.c_int64)
    lib.sfa_abi_version.restype = i32
    lib.sfa_abi_version.argtypes = []
    lib.sfa_last_error.restype = ctypes.c_char_p
    lib.sfa_last_error.argtypes = []
    lib.sfa_last_path.restype = ctypes.c_char_p
    lib.sfa_last_path.argtypes = []
    lib.sfa_debug_set_ptr.restype = i32
    lib.sfa_debug_set_ptr.argtypes = [ctypes.c_void_p]
    lib.sfa_debug_set_variant.restype = i32
    lib.sfa_debug_set_variant.argtypes = [i32, i32]
    lib.sfa_debug_set_stage_events.restype = i32
    lib.sfa_debug_set_stage_events.argtypes = [ctypes.POINTER(ctypes.c_void_p), i32]
    lib.sfa_fwd.restype = i32
    lib.sfa_fwd.argtypes = [P, P, P, P, vp, vp, i32, i32, f32, u32, vp]
    lib.sfa_bwd_workspace_bytes.restype = sz
    lib.sfa_bwd_workspace_bytes.argtypes = [i64, i64, i64, i64, i64, i32, i32, i32, u32]
    lib.sfa_bwd.restype = i32
    lib.sfa_bwd.argtypes = [P, P, P, P, P, vp, vp, P, P, P, vp, vp, sz, i32, i32, f32, u32, vp]
    lib.sfa_varlen_supported.restype = i32
    lib.sfa_varlen_supported.argtypes = [i32, i64]
    lib.sfa_fwd_varlen.restype = i32
    lib.sfa_fwd_varlen.argtypes = [P, P, P, P, vp, vp, vp, i32, i32, i32, i32, f32, u32, vp]
    lib.sfa_bwd_varlen.restype = i32
    lib.sfa_bwd_varlen.argtypes = [P, P, P, P, P, vp, vp, P, P, P, vp, vp, i32, i32, vp, sz, i32, i32, f32, u32, vp]
    lib.sfa_decode_workspace_bytes.restype = sz
    lib.sfa_decode_workspace_bytes.argtypes = [i64, i64, i64, i64, i64, i32]
    lib.sfa_decode.restype = i32
    lib.sfa_decode.argtypes = [P, P, P, P, vp, vp, sz, f32, u32, vp]
    lib.sfa_decode_ring.restype = i32
    lib.sfa_decode_ring.argtypes = [P, P, P, i64, P, P, i64, P, vp, vp, sz, f32, u32, vp]
    lib.sfa_decode_ring_step.restype = i32
    lib.sfa_decode_ring_step.argtypes = [P, P, P, i64, P, P, i64, i64, P, P, P, vp, vp, sz, f32, u32, vp]
    lib.sfa_decode_ring_step_dyn.restype = i32
    lib.sfa_decode_ring_step_dyn.argtypes = [P, P, P, P, P, P, P, P, vp, vp, vp, sz, f32, u32, vp]


def lib():
    """Load (once) and return the ctypes handle; raises if libsfa.so is absent."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise RuntimeError(
                        f"sink_attention: {LIB_PATH} not found. The HIP extension is required (there is no "
                        "Python/CPU fallback). Build it with `make -C sink-flash-attention-kernel_amd`.")
                handle = ctypes.CDLL(LIB_PATH)
                _bind(handle)
                v = handle.sfa_abi_version()
                if v != ABI_VERSION:
                    raise RuntimeError(f"libsfa.so ABI version {v}, expected {ABI_VERSION}: rebuild the extension")
                _lib = handle
    return _lib


def last_path() -> str:
    return lib().sfa_last_path().decode()


_DESC = struct.Struct("P4q4qii")        # SfaTensor's layout: one pack instead of ten ctypes field stores (1 us against 5 per
assert _DESC.size == ctypes.sizeof(SfaTensor)   # descriptor; a backward call builds ten of them)


def desc(t: torch.Tensor) -> SfaTensor:
    """Describe a 4-D [B,H,N,D] tensor (any B/H/N strides, unit D stride)."""
    return SfaTensor.from_buffer_copy(_DESC.pack(t.data_ptr(), *t.shape, *t.stride(), SFA_DTYPE[t.dtype], 0))


def check(status: int, what: str):
    if status != 0:
        msg = lib().sfa_last_error().decode()
        kind = "argument/support error" if status < 0 else "HIP error"
        raise RuntimeError(f"{what} failed ({kind} {status}): {msg}")


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError(
                "sink_attention runs only on MI355X (HIP) tensors; got a CPU tensor and there is no CPU fallback")


def unit_inner(t: torch.Tensor) -> torch.Tensor:
    """Kernels take arbitrary B/H/N strides but need the head dim contiguous."""
    if t.stride(-1) != 1 and t.shape[-1] > 1:
        return t.contiguous()
    return t


def stream_ptr(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream
