"""ctypes binding of libcat_lstm.so (include/cat_lstm.h): the LSTM recurrence of the self-play learner as one launch
per direction.  No CPU fallback inside: callers on a CUDA/HIP device in bf16 get these kernels or an exception."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

PKG = Path(__file__).resolve().parent
ROOT = PKG.parent
LIB_PATH = PKG / "libcat_lstm.so"
SRC = PKG / "csrc" / "cat_lstm.hip"
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-shared"]
HIDDEN = 128
EXPORTED_SYMBOLS = ("cat_lstm_abi_version", "cat_lstm_last_error", "cat_lstm_saved_acts_bytes", "cat_lstm_saved_cell_bytes",
                    "cat_lstm_seq_forward", "cat_lstm_seq_backward")


class Dims(C.Structure):
    _fields_ = [("G", C.c_int32), ("T", C.c_int32), ("B", C.c_int32), ("pad", C.c_int32)]


class FwdArgs(C.Structure):
    _fields_ = [("d", Dims), ("xproj", C.c_void_p), ("sx_g", C.c_int64), ("sx_t", C.c_int64), ("sx_b", C.c_int64),
                ("w_hh", C.c_void_p), ("sw_g", C.c_int64), ("h0", C.c_void_p), ("c0", C.c_void_p), ("keep", C.c_void_p),
                ("out", C.c_void_p), ("so_g", C.c_int64), ("so_t", C.c_int64), ("so_b", C.c_int64),
                ("h_last", C.c_void_p), ("c_last", C.c_void_p), ("h_in", C.c_void_p), ("saved_acts", C.c_void_p),
                ("saved_cell", C.c_void_p)]


class BwdArgs(C.Structure):
    _fields_ = [("d", Dims), ("d_out", C.c_void_p), ("so_g", C.c_int64), ("so_t", C.c_int64), ("so_b", C.c_int64),
                ("d_h_last", C.c_void_p), ("d_c_last", C.c_void_p), ("w_hh", C.c_void_p), ("sw_g", C.c_int64),
                ("keep", C.c_void_p), ("saved_acts", C.c_void_p), ("saved_cell", C.c_void_p),
                ("d_xproj", C.c_void_p), ("sx_g", C.c_int64), ("sx_t", C.c_int64), ("sx_b", C.c_int64),
                ("d_h0", C.c_void_p), ("d_c0", C.c_void_p)]


class NativeLibraryMissing(RuntimeError):
    pass


def build(force: bool = False, verbose: bool = False) -> Path:
    """Compile the kernels in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    hdr = ROOT / "include" / "cat_lstm.h"
    stale = (not LIB_PATH.exists() or LIB_PATH.stat().st_mtime < SRC.stat().st_mtime
             or LIB_PATH.stat().st_mtime < hdr.stat().st_mtime)
    if force or stale:
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        cmd = [hipcc, *HIPCC_FLAGS, f"-I{ROOT / 'include'}", "-o", str(LIB_PATH), str(SRC)]
        res = subprocess.run(cmd, capture_output=True, text=True)
        if verbose or res.returncode != 0:
            print(" ".join(cmd))
            print(res.stdout, res.stderr)
        if res.returncode != 0:
            raise RuntimeError("hipcc failed building libcat_lstm.so")
    return LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise NativeLibraryMissing(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'`. "
                "The bf16 learner on a GPU has no other LSTM path.")
        import torch  # noqa: F401  (torch's HIP runtime first, as in _native.lib)
        L = C.CDLL(str(LIB_PATH))
        L.cat_lstm_abi_version.restype = C.c_int
        L.cat_lstm_last_error.restype = C.c_char_p
        for n in ("cat_lstm_saved_acts_bytes", "cat_lstm_saved_cell_bytes"):
            getattr(L, n).restype = C.c_size_t
            getattr(L, n).argtypes = [C.c_void_p]
        for n in ("cat_lstm_seq_forward", "cat_lstm_seq_backward"):
            getattr(L, n).restype = C.c_int
            getattr(L, n).argtypes = [C.c_void_p, C.c_void_p]
        assert L.cat_lstm_abi_version() == 1
        _lib = L
    return _lib


def _check(rc: int, what: str) -> None:
    if rc != 0:
        raise RuntimeError(f"{what} failed ({rc}): {lib().cat_lstm_last_error().decode()}")


def _ptr(t) -> int:
    return 0 if t is None else t.data_ptr()


def _stream():
    import torch
    return torch.cuda.current_stream().cuda_stream


def saved_sizes(G: int, T: int, B: int):
    d = Dims(G, T, B, 0)
    return lib().cat_lstm_saved_acts_bytes(C.byref(d)), lib().cat_lstm_saved_cell_bytes(C.byref(d))


def seq_forward(xproj, w_hh, h0, c0, keep, save: bool):
    """xproj bf16 [G, T, B, 4H] (any outer strides), w_hh bf16 [G, 4H, H] (rows contiguous), h0/c0 bf16 [G, B, H],
    keep fp32 [T, B] or None.  Returns out [G, T, B, H], h_T, c_T and, with ``save``, (h_in, acts, cell) for backward."""
    import torch
    G, T, B, H4 = xproj.shape
    assert H4 == 4 * HIDDEN and xproj.dtype == torch.bfloat16 and xproj.stride(3) == 1
    assert w_hh.shape == (G, H4, HIDDEN) and w_hh.dtype == torch.bfloat16 and w_hh.stride(1) == HIDDEN and w_hh.stride(2) == 1
    h0, c0 = h0.contiguous(), c0.contiguous()
    assert h0.shape == (G, B, HIDDEN) and c0.shape == h0.shape and h0.dtype == c0.dtype == torch.bfloat16
    if keep is not None:
        assert keep.shape == (T, B) and keep.dtype == torch.float32 and keep.is_contiguous()
    dev = xproj.device
    out = torch.empty(G, T, B, HIDDEN, dtype=torch.bfloat16, device=dev)
    hT, cT = torch.empty_like(h0), torch.empty_like(c0)
    h_in = acts = cell = None
    if save:
        na, nc = saved_sizes(G, T, B)
        h_in = torch.empty(G, T, B, HIDDEN, dtype=torch.bfloat16, device=dev)
        acts = torch.empty(na, dtype=torch.uint8, device=dev)
        cell = torch.empty(nc, dtype=torch.uint8, device=dev)
    a = FwdArgs(Dims(G, T, B, 0), xproj.data_ptr(), xproj.stride(0), xproj.stride(1), xproj.stride(2),
                w_hh.data_ptr(), w_hh.stride(0), h0.data_ptr(), c0.data_ptr(), _ptr(keep),
                out.data_ptr(), out.stride(0), out.stride(1), out.stride(2), hT.data_ptr(), cT.data_ptr(),
                _ptr(h_in), _ptr(acts), _ptr(cell))
    _check(lib().cat_lstm_seq_forward(C.byref(a), _stream()), "cat_lstm_seq_forward")
    return out, hT, cT, (h_in, acts, cell)


def seq_backward(d_out, d_hT, d_cT, w_hh, keep, acts, cell, dims, want_state_grads: bool):
    """Gradients of seq_forward: d_xproj [G, T, B, 4H] and, if wanted, d_h0 / d_c0."""
    import torch
    G, T, B = dims
    dev = w_hh.device
    if d_out is not None:
        assert d_out.shape == (G, T, B, HIDDEN) and d_out.dtype == torch.bfloat16
        if d_out.stride(3) != 1:
            d_out = d_out.contiguous()
    d_hT = None if d_hT is None else d_hT.contiguous()
    d_cT = None if d_cT is None else d_cT.contiguous()
    d_x = torch.empty(G, T, B, 4 * HIDDEN, dtype=torch.bfloat16, device=dev)
    d_h0 = torch.empty(G, B, HIDDEN, dtype=torch.bfloat16, device=dev) if want_state_grads else None
    d_c0 = torch.empty(G, B, HIDDEN, dtype=torch.bfloat16, device=dev) if want_state_grads else None
    so = (0, 0, 0) if d_out is None else d_out.stride()[:3]
    a = BwdArgs(Dims(G, T, B, 0), _ptr(d_out), so[0], so[1], so[2], _ptr(d_hT), _ptr(d_cT), w_hh.data_ptr(), w_hh.stride(0),
                _ptr(keep), acts.data_ptr(), cell.data_ptr(), d_x.data_ptr(), d_x.stride(0), d_x.stride(1), d_x.stride(2),
                _ptr(d_h0), _ptr(d_c0))
    _check(lib().cat_lstm_seq_backward(C.byref(a), _stream()), "cat_lstm_seq_backward")
    return d_x, d_h0, d_c0
