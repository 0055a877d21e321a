"""ctypes binding of libcat_learn.so (include/cat_lstm.h, include/cat_trunk.h): the LSTM recurrence and the
convolutional trunk of the self-play learner's networks, one launch per direction each.  No CPU fallback inside: callers
on a CUDA/HIP device in bf16 get these kernels or an exception."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

PKG = Path(__file__).resolve().parent
ROOT = PKG.parent
LIB_PATH = PKG / "libcat_learn.so"
if os.environ.get("CAT_LEARN_LIB"):         # diagnostic builds (A/B of kernel variants): another build of the same sources
    LIB_PATH = Path(os.environ["CAT_LEARN_LIB"]).resolve()
SOURCES = tuple(PKG / "csrc" / f"cat_{n}.hip" for n in ("lstm", "trunk", "ppo", "dense", "rollout"))
HEADERS = tuple(ROOT / "include" / f"cat_{n}.h" for n in ("lstm", "trunk", "ppo", "dense", "rollout"))
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-shared"]
HIDDEN = 128
EXPORTED_SYMBOLS = ("cat_lstm_abi_version", "cat_lstm_last_error", "cat_lstm_blocks", "cat_lstm_saved_acts_bytes", "cat_lstm_saved_cell_bytes",
                    "cat_lstm_seq_forward", "cat_lstm_seq_backward")
TRUNK_SYMBOLS = ("cat_trunk_abi_version", "cat_trunk_last_error", "cat_trunk_out_positions", "cat_trunk_supported",
                 "cat_trunk_backward_blocks", "cat_trunk_forward", "cat_trunk_backward", "cat_trunk_grad_finish")


class Dims(C.Structure):
    _fields_ = [("G", C.c_int32), ("T", C.c_int32), ("B", C.c_int32), ("pad", C.c_int32)]


class FwdArgs(C.Structure):
    _fields_ = [("d", Dims), ("xproj", C.c_void_p), ("sx_g", C.c_int64), ("sx_t", C.c_int64), ("sx_b", C.c_int64),
                ("bias", C.c_void_p), ("sb_g", C.c_int64), ("bias2", C.c_void_p), ("sb2_g", C.c_int64), ("w_hh", C.c_void_p), ("sw_g", C.c_int64), ("h0", C.c_void_p), ("c0", C.c_void_p), ("keep", C.c_void_p),
                ("out", C.c_void_p), ("so_g", C.c_int64), ("so_t", C.c_int64), ("so_b", C.c_int64),
                ("h_last", C.c_void_p), ("c_last", C.c_void_p), ("h_in", C.c_void_p), ("saved_acts", C.c_void_p),
                ("saved_cell", C.c_void_p)]


class BwdArgs(C.Structure):
    _fields_ = [("d", Dims), ("d_out", C.c_void_p), ("so_g", C.c_int64), ("so_t", C.c_int64), ("so_b", C.c_int64),
                ("d_h_last", C.c_void_p), ("d_c_last", C.c_void_p), ("w_hh", C.c_void_p), ("sw_g", C.c_int64),
                ("keep", C.c_void_p), ("saved_acts", C.c_void_p), ("saved_cell", C.c_void_p),
                ("d_xproj", C.c_void_p), ("sx_g", C.c_int64), ("sx_t", C.c_int64), ("sx_b", C.c_int64),
                ("d_h0", C.c_void_p), ("d_c0", C.c_void_p), ("part_dbias", C.c_void_p)]


class TrunkDims(C.Structure):
    _fields_ = [("G", C.c_int32), ("N", C.c_int32), ("C", C.c_int32), ("R", C.c_int32)]


class TrunkParams(C.Structure):
    _fields_ = [("w1", C.c_void_p), ("b1", C.c_void_p), ("w2", C.c_void_p), ("b2", C.c_void_p),
                ("sw1_g", C.c_int64), ("sb1_g", C.c_int64), ("sw2_g", C.c_int64), ("sb2_g", C.c_int64)]


class TrunkRows(C.Structure):
    _fields_ = [("rows", C.c_void_p), ("sel", C.c_int32), ("block", C.c_int32)]


class TrunkFwd(C.Structure):
    _fields_ = [("d", TrunkDims), ("p", TrunkParams), ("x", C.c_void_p), ("sx_g", C.c_int64), ("sx_n", C.c_int64), ("x_rows", TrunkRows),
                ("out", C.c_void_p), ("so_g", C.c_int64), ("so_n", C.c_int64)]


class TrunkBwd(C.Structure):
    _fields_ = [("d", TrunkDims), ("p", TrunkParams), ("x", C.c_void_p), ("sx_g", C.c_int64), ("sx_n", C.c_int64), ("x_rows", TrunkRows),
                ("out", C.c_void_p), ("d_out", C.c_void_p), ("so_g", C.c_int64), ("so_n", C.c_int64),
                ("part_dw1", C.c_void_p), ("part_db1", C.c_void_p), ("part_dw2", C.c_void_p), ("part_db2", C.c_void_p)]


ROLLOUT_SYMBOLS = ("cat_rollout_abi_version", "cat_rollout_last_error", "cat_rollout_pack", "cat_rollout_sample", "cat_rollout_post")
DENSE_SYMBOLS = ("cat_dense_abi_version", "cat_dense_last_error", "cat_dense_bias_act", "cat_dense_act_grad", "cat_dense_sum_chunks",
                 "cat_dense_wgrad_splits", "cat_dense_wgrad", "cat_dense_forward", "cat_dense_dgrad", "cat_dense_sum_chunks2")
PPO_SYMBOLS = ("cat_ppo_abi_version", "cat_ppo_last_error", "cat_ppo_loss_grad", "cat_ppo_adam_step", "cat_ppo_gae_scan")


class PpoLoss(C.Structure):
    _fields_ = [("G", C.c_int32), ("M", C.c_int32), ("chunks", C.c_int32), ("pad", C.c_int32),
                ("logits", C.c_void_p), ("values", C.c_void_p), ("actions", C.c_void_p),
                ("old_logp", C.c_void_p), ("adv", C.c_void_p), ("ret", C.c_void_p),
                ("ratio_clip", C.c_float), ("value_scale", C.c_float), ("entropy_scale", C.c_float), ("pad2", C.c_float),
                ("d_logits", C.c_void_p), ("d_values", C.c_void_p), ("partial", C.c_void_p)]


class PpoAdam(C.Structure):
    _fields_ = [("G", C.c_int32), ("P", C.c_int32), ("chunks", C.c_int32), ("pad", C.c_int32),
                ("ar", C.c_void_p), ("col_train", C.c_void_p), ("epoch_active", C.c_void_p),
                ("m", C.c_void_p), ("v", C.c_void_p), ("steps", C.c_void_p), ("master", C.c_void_p), ("lp", C.c_void_p),
                ("kl_out", C.c_void_p), ("norm_partial", C.c_void_p),
                ("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
                ("grad_norm_clip", C.c_float), ("kl_threshold", C.c_float)]


class TrunkFinish(C.Structure):
    _fields_ = [("d", TrunkDims), ("blocks", C.c_int32), ("accumulate", C.c_int32),
                ("part_dw1", C.c_void_p), ("part_db1", C.c_void_p), ("part_dw2", C.c_void_p), ("part_db2", C.c_void_p),
                ("dw1", C.c_void_p), ("db1", C.c_void_p), ("dw2", C.c_void_p), ("db2", C.c_void_p),
                ("sw1_g", C.c_int64), ("sb1_g", C.c_int64), ("sw2_g", C.c_int64), ("sb2_g", C.c_int64)]


class NativeLibraryMissing(RuntimeError):
    pass


def build(force: bool = False, verbose: bool = False) -> Path:
    """Compile the kernels in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    newest = max(f.stat().st_mtime for f in SOURCES + HEADERS)
    stale = not LIB_PATH.exists() or LIB_PATH.stat().st_mtime < newest
    if force or stale:
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        cmd = [hipcc, *HIPCC_FLAGS, f"-I{ROOT / 'include'}", "-o", str(LIB_PATH), *map(str, SOURCES)]
        res = subprocess.run(cmd, capture_output=True, text=True)
        if verbose or res.returncode != 0:
            print(" ".join(cmd))
            print(res.stdout, res.stderr)
        if res.returncode != 0:
            raise RuntimeError("hipcc failed building libcat_learn.so")
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
        L.cat_lstm_blocks.restype = C.c_int
        L.cat_lstm_blocks.argtypes = [C.c_void_p]
        for n in ("cat_lstm_saved_acts_bytes", "cat_lstm_saved_cell_bytes"):
            getattr(L, n).restype = C.c_size_t
            getattr(L, n).argtypes = [C.c_void_p]
        for n in ("cat_lstm_seq_forward", "cat_lstm_seq_backward"):
            getattr(L, n).restype = C.c_int
            getattr(L, n).argtypes = [C.c_void_p, C.c_void_p]
        assert L.cat_lstm_abi_version() == 1
        L.cat_trunk_abi_version.restype = C.c_int
        L.cat_trunk_last_error.restype = C.c_char_p
        for n in ("cat_trunk_out_positions", "cat_trunk_supported", "cat_trunk_backward_blocks"):
            getattr(L, n).restype = C.c_int
            getattr(L, n).argtypes = [C.c_void_p]
        for n in ("cat_trunk_forward", "cat_trunk_backward", "cat_trunk_grad_finish"):
            getattr(L, n).restype = C.c_int
            getattr(L, n).argtypes = [C.c_void_p, C.c_void_p]
        assert L.cat_trunk_abi_version() == 2
        L.cat_ppo_abi_version.restype = C.c_int
        L.cat_ppo_last_error.restype = C.c_char_p
        for n in ("cat_ppo_loss_grad", "cat_ppo_adam_step", "cat_ppo_gae_scan"):
            getattr(L, n).restype = C.c_int
            getattr(L, n).argtypes = [C.c_void_p, C.c_void_p]
        assert L.cat_ppo_abi_version() == 2
        L.cat_dense_abi_version.restype = C.c_int
        L.cat_dense_last_error.restype = C.c_char_p
        L.cat_dense_bias_act.restype = C.c_int
        L.cat_dense_bias_act.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        L.cat_dense_act_grad.restype = C.c_int
        L.cat_dense_act_grad.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
        L.cat_dense_sum_chunks.restype = C.c_int
        L.cat_dense_sum_chunks.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
                                           C.c_int32, C.c_void_p]
        L.cat_dense_wgrad_splits.restype = C.c_int
        L.cat_dense_wgrad_splits.argtypes = [C.c_int32] * 4
        L.cat_dense_wgrad.restype = C.c_int
        L.cat_dense_wgrad.argtypes = [C.c_void_p, C.c_void_p]
        L.cat_dense_sum_chunks2.restype = C.c_int
        L.cat_dense_sum_chunks2.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
        for n in ("cat_dense_forward", "cat_dense_dgrad"):
            getattr(L, n).restype = C.c_int
            getattr(L, n).argtypes = [C.c_void_p, C.c_void_p]
        assert L.cat_dense_abi_version() == 2
        L.cat_rollout_abi_version.restype = C.c_int
        L.cat_rollout_last_error.restype = C.c_char_p
        for n in ("cat_rollout_pack", "cat_rollout_sample", "cat_rollout_post"):
            getattr(L, n).restype = C.c_int
            getattr(L, n).argtypes = [C.c_void_p, C.c_void_p]
        assert L.cat_rollout_abi_version() == 1
        _lib = L
    return _lib


def _check(rc: int, what: str) -> None:
    if rc != 0:
        err = (lib().cat_trunk_last_error() if "trunk" in what else lib().cat_ppo_last_error() if "ppo" in what
               else lib().cat_dense_last_error() if "dense" in what else lib().cat_rollout_last_error() if "rollout" in what
               else lib().cat_lstm_last_error())
        raise RuntimeError(f"{what} failed ({rc}): {err.decode()}")


def _ptr(t) -> int:
    return 0 if t is None else t.data_ptr()


def _stream():
    import torch
    return torch.cuda.current_stream().cuda_stream


def saved_sizes(G: int, T: int, B: int):
    d = Dims(G, T, B, 0)
    return lib().cat_lstm_saved_acts_bytes(C.byref(d)), lib().cat_lstm_saved_cell_bytes(C.byref(d))


def seq_forward(xproj, w_hh, bias, h0, c0, keep, save: bool, state_out=None, bias2=None):
    """xproj bf16 [G, T, B, 4H] (any outer strides), w_hh bf16 [G, 4H, H] (rows contiguous), bias bf16 [G, 4H] or None,
    h0/c0 bf16 [G, B, H], keep fp32 [T, B] or None.  Returns out [G, T, B, H], h_T, c_T and, with ``save``, (h_in, acts, cell) for backward."""
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
    if state_out is not None:   # written where the caller keeps the state (may be h0 / c0 themselves: a workgroup reads its
        hT, cT = state_out      # rows of the initial state before it writes the final one)
        assert hT.shape == h0.shape and cT.shape == h0.shape and hT.dtype == cT.dtype == torch.bfloat16 and hT.is_contiguous() and cT.is_contiguous()
    else:
        hT, cT = torch.empty_like(h0), torch.empty_like(c0)
    h_in = acts = cell = None
    if save:
        na, nc = saved_sizes(G, T, B)
        h_in = torch.empty(G, T, B, HIDDEN, dtype=torch.bfloat16, device=dev)
        acts = torch.empty(na, dtype=torch.uint8, device=dev)
        cell = torch.empty(nc, dtype=torch.uint8, device=dev)
    for bv in (bias, bias2):
        assert bv is None or (bv.shape == (G, H4) and bv.dtype == torch.bfloat16 and bv.stride(1) == 1)
    assert bias2 is None or bias is not None
    a = FwdArgs(Dims(G, T, B, 0), xproj.data_ptr(), xproj.stride(0), xproj.stride(1), xproj.stride(2),
                _ptr(bias), 0 if bias is None else bias.stride(0), _ptr(bias2), 0 if bias2 is None else bias2.stride(0), w_hh.data_ptr(), w_hh.stride(0), h0.data_ptr(), c0.data_ptr(), _ptr(keep),
                out.data_ptr(), out.stride(0), out.stride(1), out.stride(2), hT.data_ptr(), cT.data_ptr(),
                _ptr(h_in), _ptr(acts), _ptr(cell))
    _check(lib().cat_lstm_seq_forward(C.byref(a), _stream()), "cat_lstm_seq_forward")
    return out, hT, cT, (h_in, acts, cell)


def seq_backward(d_out, d_hT, d_cT, w_hh, keep, acts, cell, dims, want_state_grads: bool, want_bias_grad: bool = False):
    """Gradients of seq_forward: d_xproj [G, T, B, 4H], if wanted d_h0 / d_c0, and if wanted the per-workgroup partial sums
    [G, blocks, 4H] of the bias gradient."""
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
    part = None
    if want_bias_grad:
        dd = Dims(G, T, B, 0)
        part = torch.empty(G, lib().cat_lstm_blocks(C.byref(dd)), 4 * HIDDEN, dtype=torch.float32, device=dev)
    so = (0, 0, 0) if d_out is None else d_out.stride()[:3]
    a = BwdArgs(Dims(G, T, B, 0), _ptr(d_out), so[0], so[1], so[2], _ptr(d_hT), _ptr(d_cT), w_hh.data_ptr(), w_hh.stride(0),
                _ptr(keep), acts.data_ptr(), cell.data_ptr(), d_x.data_ptr(), d_x.stride(0), d_x.stride(1), d_x.stride(2),
                _ptr(d_h0), _ptr(d_c0), _ptr(part))
    _check(lib().cat_lstm_seq_backward(C.byref(a), _stream()), "cat_lstm_seq_backward")
    return d_x, d_h0, d_c0, part


# ---------------------------------------------------------------------------------------------- convolutional trunk
def trunk_supported(G: int, N: int, C_in: int, R: int) -> bool:
    d = TrunkDims(G, N, C_in, R)
    return bool(lib().cat_trunk_supported(C.byref(d)))


def _trunk_params(w1, b1, w2, b2, G: int, C_in: int) -> TrunkParams:
    import torch
    assert w1.shape == (G, 64, C_in, 5) and w1[0].is_contiguous() and b1.shape == (G, 64) and b1.stride(1) == 1
    assert w2.shape == (G, 32, 64, 5) and w2[0].is_contiguous() and b2.shape == (G, 32) and b2.stride(1) == 1
    assert all(t.dtype == torch.bfloat16 for t in (w1, b1, w2, b2))
    return TrunkParams(w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), w1.stride(0), b1.stride(0), w2.stride(0), b2.stride(0))


def _trunk_rows(x, rows, block: int):
    """(samples N, cat_trunk_rows) of an input [G, rows of x, C * R]: all of them, or -- ``rows`` int64 [sel] on the device,
    ``block`` -- the minibatch of ``sel`` sequences out of ``block`` per step, read in place (include/cat_trunk.h)."""
    import torch
    if rows is None:
        return x.shape[1], TrunkRows(None, 0, 0)
    assert rows.dtype == torch.int64 and rows.is_contiguous() and rows.device == x.device and rows.dim() == 1
    sel = rows.shape[0]
    assert 0 < sel <= block and x.shape[1] % block == 0
    return (x.shape[1] // block) * sel, TrunkRows(rows.data_ptr(), sel, block)


def trunk_forward(x, w1, b1, w2, b2, R: int, rows=None, block: int = 0):
    """x bf16 [G, N, C * R] in (channel, ray) order -> bf16 [G, N, L2 * 32] in (position, channel) order.  With ``rows``: x is
    the whole [G, steps * block, C * R] buffer and N = steps * len(rows) (``_trunk_rows``)."""
    import torch
    G, _, CR = x.shape
    C_in = CR // R
    assert x.dtype == torch.bfloat16 and x.stride(2) == 1 and C_in * R == CR
    N, xr = _trunk_rows(x, rows, block)
    d = TrunkDims(G, N, C_in, R)
    L2 = lib().cat_trunk_out_positions(C.byref(d))
    out = torch.empty(G, N, L2 * 32, dtype=torch.bfloat16, device=x.device)
    a = TrunkFwd(d, _trunk_params(w1, b1, w2, b2, G, C_in), x.data_ptr(), x.stride(0), x.stride(1), xr, out.data_ptr(), out.stride(0), out.stride(1))
    _check(lib().cat_trunk_forward(C.byref(a), _stream()), "cat_trunk_forward")
    return out


def trunk_backward(x, w1, b1, w2, b2, out, d_out, R: int, rows=None, block: int = 0):
    """fp32 per-workgroup partial sums [G, B, 64, 32], [G, B, 64], [G, B, 32, 320], [G, B, 32] of the parameter gradients
    (columns kk * C + c and kk * 64 + c_in): the caller adds the B slabs up.  ``rows``, ``block``: as in ``trunk_forward``."""
    import torch
    G, _, CR = x.shape
    C_in = CR // R
    N, xr = _trunk_rows(x, rows, block)
    assert out.shape[1] == N
    d = TrunkDims(G, N, C_in, R)
    d_out = d_out.contiguous()
    assert out.is_contiguous() and d_out.shape == out.shape and d_out.dtype == torch.bfloat16
    nb = lib().cat_trunk_backward_blocks(C.byref(d))
    dev = x.device
    pw1 = torch.empty(G, nb, 64, 32, dtype=torch.float32, device=dev)
    pb1 = torch.empty(G, nb, 64, dtype=torch.float32, device=dev)
    pw2 = torch.empty(G, nb, 32, 320, dtype=torch.float32, device=dev)
    pb2 = torch.empty(G, nb, 32, dtype=torch.float32, device=dev)
    a = TrunkBwd(d, _trunk_params(w1, b1, w2, b2, G, C_in), x.data_ptr(), x.stride(0), x.stride(1), xr, out.data_ptr(), d_out.data_ptr(),
                 out.stride(0), out.stride(1), pw1.data_ptr(), pb1.data_ptr(), pw2.data_ptr(), pb2.data_ptr())
    _check(lib().cat_trunk_backward(C.byref(a), _stream()), "cat_trunk_backward")
    return pw1, pb1, pw2, pb2


def trunk_grad_finish(parts, C_in: int, R: int, slots=None):
    """The four slab buffers of ``trunk_backward`` -> the parameter gradients in the parameters' own shapes (bf16): ADDED into
    ``slots`` = (dw1, db1, dw2, db2) views (returns None) or returned as new tensors."""
    import torch
    pw1, pb1, pw2, pb2 = parts
    G, nb = pw2.shape[:2]
    acc = slots is not None
    if not acc:
        dev = pw1.device
        slots = (torch.empty(G, 64, C_in, 5, dtype=torch.bfloat16, device=dev), torch.empty(G, 64, dtype=torch.bfloat16, device=dev),
                 torch.empty(G, 32, 64, 5, dtype=torch.bfloat16, device=dev), torch.empty(G, 32, dtype=torch.bfloat16, device=dev))
    for t in slots:
        assert t.dtype == torch.bfloat16 and t[0].is_contiguous()
    a = TrunkFinish(TrunkDims(G, 16, C_in, R), nb, 1 if acc else 0, pw1.data_ptr(), pb1.data_ptr(), pw2.data_ptr(), pb2.data_ptr(),
                    slots[0].data_ptr(), slots[1].data_ptr(), slots[2].data_ptr(), slots[3].data_ptr(),
                    slots[0].stride(0), slots[1].stride(0), slots[2].stride(0), slots[3].stride(0))
    _check(lib().cat_trunk_grad_finish(C.byref(a), _stream()), "cat_trunk_grad_finish")
    return None if acc else slots


# ---------------------------------------------------------------------------------------------- PPO loss / optimiser step
PPO_CHUNKS = 64


def ppo_loss_grad(logits, values, actions, old_logp, adv, ret, ratio_clip: float, value_scale: float, entropy_scale: float):
    """logits bf16 [G, ..., 4], values bf16 [G, ...(, 1)], actions int64 / old_logp / adv / ret fp32 [G, ...] (M samples per
    agent each, contiguous).  Returns (sums fp32 [G, 4] = surrogate, squared value error, entropy, KL; d_logits; d_values)."""
    import torch
    G = logits.shape[0]
    M = logits[0].numel() // 4
    for t, dt in ((logits, torch.bfloat16), (values, torch.bfloat16), (actions, torch.int64), (old_logp, torch.float32),
                  (adv, torch.float32), (ret, torch.float32)):
        assert t.dtype == dt and t.is_contiguous() and t.shape[0] == G
    assert values[0].numel() == M and actions[0].numel() == M and old_logp[0].numel() == M and adv[0].numel() == M and ret[0].numel() == M
    d_logits, d_values = torch.empty_like(logits), torch.empty_like(values)
    partial = torch.empty(G, PPO_CHUNKS, 4, dtype=torch.float32, device=logits.device)
    a = PpoLoss(G, M, PPO_CHUNKS, 0, logits.data_ptr(), values.data_ptr(), actions.data_ptr(), old_logp.data_ptr(), adv.data_ptr(),
                ret.data_ptr(), ratio_clip, value_scale, entropy_scale, 0.0, d_logits.data_ptr(), d_values.data_ptr(), partial.data_ptr())
    _check(lib().cat_ppo_loss_grad(C.byref(a), _stream()), "cat_ppo_loss_grad")
    return torch.bmm(ones_row(G, PPO_CHUNKS, logits.device), partial).squeeze(1), d_logits, d_values


def ppo_adam_step(ar, col_train, epoch_active, m, v, steps, master, lp, kl_out, scratch, lr, beta1, beta2, eps, grad_norm_clip,
                  kl_threshold) -> None:
    """In-place optimiser step on the flat [G, P] buffers (see include/cat_ppo.h); ``lp`` = bf16 copy or None;
    ``scratch`` fp32 [G, 256]."""
    import torch
    G, P = master.shape
    for t in (ar, col_train, epoch_active, m, v, steps, master, kl_out, scratch):
        assert t.dtype == torch.float32 and t.is_contiguous()
    assert ar.shape == (G, P + 1) and col_train.shape == (G, P) and scratch.shape == (G, 256) and kl_out.shape == (G,)
    assert lp is None or (lp.dtype == torch.bfloat16 and lp.is_contiguous() and lp.shape == (G, P))
    a = PpoAdam(G, P, 256, 0, ar.data_ptr(), col_train.data_ptr(), epoch_active.data_ptr(), m.data_ptr(), v.data_ptr(), steps.data_ptr(),
                master.data_ptr(), _ptr(lp), kl_out.data_ptr(), scratch.data_ptr(), lr, beta1, beta2, eps, grad_norm_clip, kl_threshold or 0.0)
    _check(lib().cat_ppo_adam_step(C.byref(a), _stream()), "cat_ppo_adam_step")


class PpoGae(C.Structure):
    _fields_ = [("G", C.c_int32), ("T", C.c_int32), ("N", C.c_int32), ("pad", C.c_int32), ("rewards", C.c_void_p), ("values", C.c_void_p),
                ("dones", C.c_void_p), ("last_values", C.c_void_p), ("gamma", C.c_float), ("lambda_", C.c_float),
                ("adv", C.c_void_p), ("ret", C.c_void_p)]


def ppo_gae(rewards, values, dones, last_values, gamma: float, lam: float, adv_out, ret_out) -> None:
    """rewards, values, adv_out, ret_out fp32 [G, T, N] contiguous; dones bool / uint8 [T, N]; last_values fp32 [G, N]: the whole
    reverse scan in one launch (include/cat_ppo.h)."""
    import torch
    G, T, N = rewards.shape
    for t in (rewards, values, adv_out, ret_out):
        assert t.dtype == torch.float32 and t.is_contiguous() and t.shape == (G, T, N)
    last_values = last_values.contiguous()
    assert last_values.dtype == torch.float32 and last_values.shape == (G, N)
    d8 = dones.contiguous().view(torch.uint8) if dones.dtype == torch.bool else dones.contiguous()
    assert d8.dtype == torch.uint8 and d8.shape == (T, N)
    a = PpoGae(G, T, N, 0, rewards.data_ptr(), values.data_ptr(), d8.data_ptr(), last_values.data_ptr(), gamma, lam,
               adv_out.data_ptr(), ret_out.data_ptr())
    _check(lib().cat_ppo_gae_scan(C.byref(a), _stream()), "cat_ppo_gae_scan")


# ---------------------------------------------------------------------------------------------- dense-layer epilogues
ACT_NONE, ACT_RELU, ACT_TANH = 0, 1, 2
DENSE_CHUNKS = 64


class DenseDims(C.Structure):
    _fields_ = [("G", C.c_int32), ("M", C.c_int32), ("out", C.c_int32), ("act", C.c_int32)]


def dense_bias_act_(y, bias, act: int):
    """y bf16 [G, M, out] contiguous <- act(y + bias[g]) in place; bias bf16 [G, out] (row stride free)."""
    import torch
    G, M, out = y.shape
    assert y.dtype == torch.bfloat16 and y.is_contiguous() and bias.shape == (G, out) and bias.dtype == torch.bfloat16 and bias.stride(1) == 1
    d = DenseDims(G, M, out, act)
    _check(lib().cat_dense_bias_act(C.byref(d), y.data_ptr(), bias.data_ptr(), bias.stride(0), _stream()), "cat_dense_bias_act")
    return y


def dense_act_grad(d_y, y, act: int):
    """(d_y * act'(y), fp32 per-chunk column sums [G, chunks, out] of it: ``sum_chunks`` adds them up).  With ACT_NONE the
    first result is d_y itself."""
    import torch
    G, M, out = d_y.shape
    d_y = d_y.contiguous()
    assert d_y.dtype == torch.bfloat16 and (act == ACT_NONE or (y.shape == d_y.shape and y.is_contiguous()))
    g_out = torch.empty_like(d_y) if act != ACT_NONE else None
    chunks = max(1, min(256, M // 32))
    partial = torch.empty(G, chunks, out, dtype=torch.float32, device=d_y.device)
    d = DenseDims(G, M, out, act)
    _check(lib().cat_dense_act_grad(C.byref(d), d_y.data_ptr(), _ptr(y) if act != ACT_NONE else 0, _ptr(g_out), partial.data_ptr(), chunks,
                                    _stream()), "cat_dense_act_grad")
    return (g_out if g_out is not None else d_y), partial


def sum_chunks(partial, dst=None, dst2=None, accumulate: bool = False):
    """partial fp32 [G, chunks, ...] -> bf16 [G, ...] sums over the chunks: into ``dst`` (and ``dst2``), each [G, n] with
    contiguous rows and any row stride, optionally added to what they hold; a new tensor when ``dst`` is None."""
    import torch
    G, chunks = partial.shape[:2]
    n = partial[0, 0].numel()
    assert partial.dtype == torch.float32 and partial.is_contiguous()
    if dst is None:
        dst = torch.empty((G,) + tuple(partial.shape[2:]), dtype=torch.bfloat16, device=partial.device)
    for t in (dst, dst2):
        assert t is None or (t.dtype == torch.bfloat16 and t.shape[0] == G and t[0].numel() == n and t[0].is_contiguous())
    _check(lib().cat_dense_sum_chunks(partial.data_ptr(), G, chunks, n, dst.data_ptr(), dst.stride(0), _ptr(dst2),
                                      0 if dst2 is None else dst2.stride(0), 1 if accumulate else 0, _stream()), "cat_dense_sum_chunks")
    return dst


class WgradArgs(C.Structure):
    _fields_ = [("G", C.c_int32), ("K", C.c_int32), ("M", C.c_int32), ("N", C.c_int32), ("a", C.c_void_p), ("b", C.c_void_p),
                ("partial", C.c_void_p), ("splits", C.c_int32), ("pad", C.c_int32),
                ("b1", C.c_void_p), ("partial1", C.c_void_p), ("N1", C.c_int32), ("pad1", C.c_int32)]


def wgrad_supported(g, x) -> bool:
    return g.shape[1] >= 512


class SumJob(C.Structure):
    _fields_ = [("partial", C.c_void_p), ("chunks", C.c_int32), ("n", C.c_int32), ("dst0", C.c_void_p), ("sd0_g", C.c_int64),
                ("dst1", C.c_void_p), ("sd1_g", C.c_int64), ("accumulate", C.c_int32), ("pad", C.c_int32)]


def _sum_job(partial, dst, accumulate: bool) -> SumJob:
    G, chunks = partial.shape[:2]
    n = partial[0, 0].numel()
    assert partial.is_contiguous() and dst.shape[0] == G and dst[0].numel() == n and dst[0].is_contiguous()
    return SumJob(partial.data_ptr(), chunks, n, dst.data_ptr(), dst.stride(0), 0, 0, 1 if accumulate else 0, 0)


def dense_wgrad(g, x, slot=None, bias_job=None):
    """g bf16 [G, K, M] (gradient of a layer's pre-activations), x bf16 [G, K, N] (its input) -> sum_k g[k, m] x[k, n] as
    bf16 [G, M, N]: ADDED into ``slot`` (a [G, M, N] view whose [M, N] blocks are contiguous) when given, else returned.
    ``bias_job`` = (partial fp32 [G, chunks, M], bias slot [G, M]): that chunk sum rides in the same launch."""
    import torch
    G, K, M = g.shape
    N = x.shape[2]
    g, x = g.contiguous(), x.contiguous()
    assert g.dtype == x.dtype == torch.bfloat16 and x.shape[:2] == (G, K)
    S = lib().cat_dense_wgrad_splits(G, K, M, N)
    partial = torch.empty(G, S, M * N, dtype=torch.float32, device=g.device)
    a = WgradArgs(G, K, M, N, g.data_ptr(), x.data_ptr(), partial.data_ptr(), S, 0, None, None, 0, 0)
    _check(lib().cat_dense_wgrad(C.byref(a), _stream()), "cat_dense_wgrad")
    out = None
    if slot is not None:
        assert slot.shape == (G, M, N) and slot[0].is_contiguous()
        dst = slot.view(G, M * N) if slot.is_contiguous() else slot.as_strided((G, M * N), (slot.stride(0), 1))
    else:
        out = torch.empty(G, M * N, dtype=torch.bfloat16, device=g.device)
        dst = out
    ja = _sum_job(partial, dst, slot is not None)
    jb = None if bias_job is None else _sum_job(bias_job[0], bias_job[1], True)
    _check(lib().cat_dense_sum_chunks2(C.byref(ja), None if jb is None else C.byref(jb), G, _stream()), "cat_dense_sum_chunks2")
    return None if slot is not None else out.view(G, M, N)


def dense_wgrad2(g, x0, x1, slot0, slot1) -> None:
    """The weight gradients of TWO layers that share the gradient g [G, K, M] of their summed pre-activations (an LSTM layer:
    W_ih with its input x0 [G, K, N0], W_hh with h_in x1 [G, K, N1]): one launch reads g once, one launch adds the slabs of
    both up into ``slot0`` [G, M, N0] and ``slot1`` [G, M, N1]."""
    import torch
    G, K, M = g.shape
    N0, N1 = x0.shape[2], x1.shape[2]
    g, x0, x1 = g.contiguous(), x0.contiguous(), x1.contiguous()
    assert g.dtype == x0.dtype == x1.dtype == torch.bfloat16 and x0.shape[:2] == (G, K) and x1.shape[:2] == (G, K)
    S = lib().cat_dense_wgrad_splits(G, K, M, 128 * (-(-N0 // 128) - (-N1 // 128)))
    p0 = torch.empty(G, S, M * N0, dtype=torch.float32, device=g.device)
    p1 = torch.empty(G, S, M * N1, dtype=torch.float32, device=g.device)
    a = WgradArgs(G, K, M, N0, g.data_ptr(), x0.data_ptr(), p0.data_ptr(), S, 0, x1.data_ptr(), p1.data_ptr(), N1, 0)
    _check(lib().cat_dense_wgrad(C.byref(a), _stream()), "cat_dense_wgrad")
    jobs = []
    for part, slot, N in ((p0, slot0, N0), (p1, slot1, N1)):
        assert slot.shape == (G, M, N) and slot[0].is_contiguous()
        dst = slot.view(G, M * N) if slot.is_contiguous() else slot.as_strided((G, M * N), (slot.stride(0), 1))
        jobs.append(_sum_job(part, dst, True))
    _check(lib().cat_dense_sum_chunks2(C.byref(jobs[0]), C.byref(jobs[1]), G, _stream()), "cat_dense_sum_chunks2")


class GemmArgs(C.Structure):
    _fields_ = [("G", C.c_int32), ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("x_or_gr", C.c_void_p), ("w", C.c_void_p),
                ("sw_g", C.c_int64), ("bias", C.c_void_p), ("sb_g", C.c_int64), ("act", C.c_int32), ("pad", C.c_int32), ("out", C.c_void_p)]


def gemm_supported(x, w) -> bool:
    """The layer kernels take rows of 16-byte runs: in-features a multiple of 8, weight rows contiguous and aligned."""
    return (x.shape[2] % 8 == 0 and w.stride(2) == 1 and w.stride(1) == w.shape[2] and w.stride(0) % 8 == 0 and w.data_ptr() % 16 == 0)


def dense_forward(x, w, bias, act: int):
    """act(x [G, M, K] @ w [G, N, K]^T + bias [G, N]) -> bf16 [G, M, N]; bias may be None (act must then be ACT_NONE)."""
    import torch
    G, M, K = x.shape
    N = w.shape[1]
    x = x.contiguous()
    assert x.dtype == w.dtype == torch.bfloat16 and w.shape == (G, N, K) and gemm_supported(x, w)
    assert bias is None or (bias.shape == (G, N) and bias.dtype == torch.bfloat16 and bias.stride(1) == 1)
    y = torch.empty(G, M, N, dtype=torch.bfloat16, device=x.device)
    a = GemmArgs(G, M, N, K, x.data_ptr(), w.data_ptr(), w.stride(0), _ptr(bias), 0 if bias is None else bias.stride(0), act, 0, y.data_ptr())
    _check(lib().cat_dense_forward(C.byref(a), _stream()), "cat_dense_forward")
    return y


def dense_dgrad(gr, w):
    """gr [G, M, N] @ w [G, N, K] -> bf16 [G, M, K]: the gradient w.r.t. the layer's input."""
    import torch
    G, M, N = gr.shape
    K = w.shape[2]
    gr = gr.contiguous()
    assert gr.dtype == w.dtype == torch.bfloat16 and w.shape == (G, N, K) and w.stride(2) == 1 and w.stride(1) == K and K % 8 == 0
    dx = torch.empty(G, M, K, dtype=torch.bfloat16, device=gr.device)
    a = GemmArgs(G, M, N, K, gr.data_ptr(), w.data_ptr(), w.stride(0), 0, 0, 0, 0, dx.data_ptr())
    _check(lib().cat_dense_dgrad(C.byref(a), _stream()), "cat_dense_dgrad")
    return dx


_ONES = {}


def ones_row(G: int, n: int, device):
    """cached fp32 [G, 1, n] of ones (the left operand of the "add the slabs up" GEMMs)"""
    import torch
    key = (G, n, str(device))
    if key not in _ONES:
        _ONES[key] = torch.ones(G, 1, n, dtype=torch.float32, device=device)
    return _ONES[key]


# ---------------------------------------------------------------------------------------------- rollout tick glue
class PackArgs(C.Structure):
    _fields_ = [("N", C.c_int32), ("A", C.c_int32), ("R", C.c_int32), ("G", C.c_int32), ("agent", C.c_int32 * 8),
                ("n_cops", C.c_int32), ("first_agent_state", C.c_int32), ("distance_scale", C.c_float), ("type_scale", C.c_float),
                ("obs_distance", C.c_void_p), ("obs_type", C.c_void_p), ("shared_distance", C.c_void_p), ("shared_type", C.c_void_p),
                ("policy_in", C.c_void_p), ("sp_g", C.c_int64), ("sp_n", C.c_int64),
                ("value_in", C.c_void_p), ("sv_g", C.c_int64), ("sv_n", C.c_int64)]


class SampleArgs(C.Structure):
    _fields_ = [("N", C.c_int32), ("A", C.c_int32), ("G", C.c_int32), ("pad", C.c_int32), ("agent", C.c_int32 * 8),
                ("logits", C.c_void_p), ("uniform", C.c_void_p), ("values", C.c_void_p),
                ("act_out", C.c_void_p), ("logp_out", C.c_void_p), ("value_out", C.c_void_p), ("sa_g", C.c_int64), ("sl_g", C.c_int64),
                ("actions", C.c_void_p)]


def rollout_pack(raw, agent_indices, n_cops: int, first_agent_state: bool, distance_scale: float, type_scale: float, policy_in, value_in):
    """raw = the env core's output buffers (``VecCopsEnv.raw_outputs()``) -> policy_in bf16 [G, N, 2R], value_in bf16
    [G, N, 4R] (views with contiguous rows and any outer strides)."""
    import torch
    od, ot, sd, st = raw["obs_distance"], raw["obs_type"], raw["shared_distance"], raw["shared_type"]
    N, A, R = od.shape
    G = len(agent_indices)
    assert od.dtype == torch.float16 and ot.dtype == torch.uint8 and all(t.is_contiguous() for t in (od, ot, sd, st))
    for t, w in ((policy_in, 2 * R), (value_in, 4 * R)):
        assert t.dtype == torch.bfloat16 and t.shape == (G, N, w) and t.stride(2) == 1
    a = PackArgs(N, A, R, G, (C.c_int32 * 8)(*agent_indices), n_cops, 1 if first_agent_state else 0, distance_scale, type_scale,
                 od.data_ptr(), ot.data_ptr(), sd.data_ptr(), st.data_ptr(), policy_in.data_ptr(), policy_in.stride(0), policy_in.stride(1),
                 value_in.data_ptr(), value_in.stride(0), value_in.stride(1))
    _check(lib().cat_rollout_pack(C.byref(a), _stream()), "cat_rollout_pack")


def rollout_sample(logits, uniform, values, act_out, logp_out, value_out, actions, agent_indices) -> None:
    """logits bf16 [G, N, 4]; uniform fp32 [G, N]; values bf16 [G, N] or None; act_out int64 / logp_out, value_out fp32
    [G, N] row-strided views; actions int32 [N, A]: column ``agent_indices[g]`` receives agent g's draw."""
    import torch
    G, N, _ = logits.shape
    assert logits.dtype == torch.bfloat16 and logits.is_contiguous() and uniform.dtype == torch.float32 and uniform.is_contiguous()
    assert act_out.dtype == torch.int64 and act_out.stride(1) == 1 and logp_out.dtype == torch.float32 and logp_out.stride(1) == 1
    assert actions.dtype == torch.int32 and actions.is_contiguous() and actions.shape[0] == N
    if value_out is not None:
        assert values.dtype == torch.bfloat16 and values.is_contiguous() and value_out.stride(0) == logp_out.stride(0) and value_out.stride(1) == 1
    a = SampleArgs(N, actions.shape[1], G, 0, (C.c_int32 * 8)(*agent_indices), logits.data_ptr(), uniform.data_ptr(), _ptr(values),
                   act_out.data_ptr(), logp_out.data_ptr(), _ptr(value_out), act_out.stride(0), logp_out.stride(0), actions.data_ptr())
    _check(lib().cat_rollout_sample(C.byref(a), _stream()), "cat_rollout_sample")


class PostArgs(C.Structure):
    _fields_ = [("N", C.c_int32), ("A", C.c_int32), ("G", C.c_int32), ("pad", C.c_int32), ("agent", C.c_int32 * 8),
                ("reward", C.c_void_p), ("terminated", C.c_void_p), ("reward_out", C.c_void_p), ("sr_g", C.c_int64),
                ("done_out", C.c_void_p), ("start_out", C.c_void_p), ("keep_out", C.c_void_p)]


def rollout_post(raw, agent_indices, reward_out, done_out=None, start_out=None, keep_out=None) -> None:
    """raw["reward"] fp32 [N, A], raw["terminated"] u8 [N] -> reward_out fp32 [G, N] (row-strided view); done_out / start_out
    bool [N]; keep_out fp32 [N] = 1 - terminated."""
    import torch
    rew, term = raw["reward"], raw["terminated"]
    N, A = rew.shape
    assert rew.dtype == torch.float32 and rew.is_contiguous() and term.dtype == torch.uint8 and term.shape == (N,)
    assert reward_out.dtype == torch.float32 and reward_out.shape == (len(agent_indices), N) and reward_out.stride(1) == 1
    for t, dt in ((done_out, torch.bool), (start_out, torch.bool), (keep_out, torch.float32)):
        assert t is None or (t.dtype == dt and t.numel() == N and t.is_contiguous())
    a = PostArgs(N, A, len(agent_indices), 0, (C.c_int32 * 8)(*agent_indices), rew.data_ptr(), term.data_ptr(), reward_out.data_ptr(),
                 reward_out.stride(0), _ptr(done_out), _ptr(start_out), _ptr(keep_out))
    _check(lib().cat_rollout_post(C.byref(a), _stream()), "cat_rollout_post")
