"""Role-stacked networks: the G policy (or value) networks of one role evaluated as ONE batched network.

Same architectures as ``models.LSTMPolicy`` / ``models.LSTMValue`` (reference
``src/models/lstm_policy_net.py:28-53``, ``lstm_value_net.py:46-75``) and the same parameter names per
agent (``agent_state_dict`` / ``load_agent_state_dict`` round-trip through the per-agent modules), but every
weight is a ``[G, ...]`` tensor and every layer one ``baddbmm`` over the G networks: an agent more costs no
extra kernel launch.  The G agents' parameters are rows of ONE flat fp32 buffer (``FlatParams``): gradients
accumulate into views of one flat gradient buffer, so the gradient all-reduce, the per-agent norm clip and
Adam are a handful of kernels per optimiser step whatever the number of layers.

The LSTM recurrence over a BPTT window is one autograd node (``_LSTMSeq``): in bf16 on the GPU one launch of
``libcat_learn.so`` forward and one backward for the whole window of a layer (``csrc/cat_lstm.hip``), the weight
gradient ONE GEMM over all steps (fp32 accumulation); in fp32 / on the CPU the same recurrence step by step.
"""
from __future__ import annotations

import math
import os
import warnings
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn as nn

from .. import _learn_native
from .models import LSTMPolicy, LSTMValue, Policy, Value, conv_out_len

HIDDEN = 128
# the stacked parameters are rows of one flat buffer (row stride = all parameters of an agent), and their gradients are
# views of the flat gradient buffer with the same strides: autograd's "layout contract" note about that is expected
warnings.filterwarnings("ignore", message="grad and param do not obey the gradient layout contract")


# ---------------------------------------------------------------------------------------------- LSTM cell
def _cell_fwd(ig: torch.Tensor, hg: torch.Tensor, c: torch.Tensor):
    """One LSTM cell step from the input-side and hidden-side gate pre-activations (gate order i, f, g, o, as
    ``nn.LSTM``).  ig, hg: [G, B, 4H]; c: [G, B, H].  Returns h', c', and the activated gates [G, B, 4H]."""
    if ig.is_cuda:   # the fused pointwise kernel nn.LSTMCell uses
        G, B, H4 = ig.shape
        hy, cy, ws = torch.ops.aten._thnn_fused_lstm_cell(ig.reshape(G * B, H4), hg.reshape(G * B, H4), c.reshape(G * B, H4 // 4))
        return hy.view(G, B, -1), cy.view(G, B, -1), ws.view(G, B, H4)
    pre = ig + hg
    i, f, g, o = pre.chunk(4, dim=-1)
    i, f, g, o = torch.sigmoid(i), torch.sigmoid(f), torch.tanh(g), torch.sigmoid(o)
    cy = f * c + i * g
    return o * torch.tanh(cy), cy, torch.cat([i, f, g, o], dim=-1)


def _cell_bwd(dh: torch.Tensor, dc: Optional[torch.Tensor], c: torch.Tensor, cy: torch.Tensor, ws: torch.Tensor):
    """Gradient of ``_cell_fwd`` w.r.t. the summed gate pre-activations and the incoming cell state."""
    if dh.is_cuda:
        G, B, H = c.shape
        dg, dcx, _ = torch.ops.aten._thnn_fused_lstm_cell_backward_impl(
            dh.reshape(G * B, H), None if dc is None else dc.reshape(G * B, H), c.reshape(G * B, H), cy.reshape(G * B, H),
            ws.reshape(G * B, 4 * H), False)
        return dg.view(G, B, 4 * H), dcx.view(G, B, H)
    i, f, g, o = ws.chunk(4, dim=-1)
    tc = torch.tanh(cy)
    dct = dh * o * (1 - tc * tc)
    if dc is not None:
        dct = dct + dc
    dg = torch.cat([dct * g * i * (1 - i), dct * c * f * (1 - f), dct * i * (1 - g * g), dh * tc * o * (1 - o)], dim=-1)
    return dg, dct * f


def _grad_slot(p: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    """The pre-assigned gradient view of a FlatParams leaf (``FlatParams.views``), else None."""
    if p is None or not (p.is_leaf and p.requires_grad) or p.grad is None:
        return None
    return p.grad


def _weight_grad(g: torch.Tensor, x: torch.Tensor, slot: Optional[torch.Tensor], bias_job=None) -> Optional[torch.Tensor]:
    """sum over the rows of g[:, k, :]^T x[:, k, :] for the G stacked layers (bf16 on the GPU): added into ``slot`` (returns
    None) or returned.  The reduction runs over tens of thousands of rows into a small matrix: ``csrc/cat_dense.hip``'s
    split-K kernel where its tiling applies, the library GEMM otherwise (tiny heads)."""
    if _learn_native.wgrad_supported(g, x):
        return _learn_native.dense_wgrad(g, x, slot, bias_job)
    assert bias_job is None
    if slot is not None:
        slot.baddbmm_(g.transpose(1, 2), x)
        return None
    return torch.bmm(g.transpose(1, 2), x)


class _LSTMSeq(torch.autograd.Function):
    """h_t, c_t = cell(xproj[:, t] + h_{t-1} W_hh^T, c_{t-1}) for t = 0..T-1, with the state zeroed where ``keep[t]`` is 0
    (an episode starts at step t).  xproj [G, T, B, 4H]; w_hh [G, 4H, H]; b_ih, b_hh [G, 4H]: their sum is added to every
    step's gate pre-activations (kernel path only; None = already inside xproj); h0, c0 [G, B, H]; keep fp32 [T, B] or
    None.  Returns out [G, T, B, H], h_T, c_T.  Parameters that are FlatParams leaves get their gradient ADDED straight
    into their slot of the flat gradient buffer by backward (which then returns None for them): no AccumulateGrad pass.

    bf16 on a GPU: the whole window is ONE launch of ``libcat_learn.so`` per direction (``csrc/cat_lstm.hip``: W_hh
    resident in registers, the products on the matrix cores, the cell in fp32) -- no other path exists there.  fp32 /
    CPU (the parity tests, ``compute_bf16=False``): the same recurrence step by step in torch."""

    @staticmethod
    def forward(ctx, xproj, w_hh, b_ih, b_hh, h0, c0, keep, state_out=None, grad_mode=True, shared=None):
        G, T, B, _ = xproj.shape
        ctx.shared = shared        # _SharedWgrad of the projection that produced xproj: its weight gradient is taken here
        # ``grad_mode`` = torch.is_grad_enabled() at the CALL (inside a Function's forward it always reads False, and
        # needs_input_grad stays True for parameters under no_grad): a rollout tick keeps nothing for a backward pass
        train = grad_mode and any(ctx.needs_input_grad)
        ctx.has_keep = keep is not None
        ctx.set_materialize_grads(False)           # unused final-state gradients arrive as None, not as zero tensors
        ctx.native = xproj.is_cuda and xproj.dtype == torch.bfloat16
        if ctx.native:
            out, hT, cT, (h_in, acts, cell) = _learn_native.seq_forward(xproj, w_hh, b_ih, h0, c0, keep, save=train, state_out=state_out,
                                                                        bias2=b_hh if b_ih is not None else None)
            ctx.has_bias = b_ih is not None
            ctx.slots = (_grad_slot(w_hh), _grad_slot(b_ih), _grad_slot(b_hh))
            if train:
                ctx.save_for_backward(w_hh, h_in, acts, cell, keep if keep is not None else torch.empty(0))
                ctx.dims = (G, T, B)
            return out, hT, cT
        assert b_ih is None and b_hh is None, "the step-by-step path takes the bias inside xproj"
        w_t = w_hh.transpose(1, 2)
        h, c = h0, c0
        hs, cs, cys, wss, outs = [], [], [], [], []
        kf = None if keep is None else keep.to(xproj.dtype).view(T, 1, B, 1)
        for t in range(T):
            if kf is not None:
                h, c = h * kf[t], c * kf[t]
            hg = torch.bmm(h, w_t)
            hy, cy, ws = _cell_fwd(xproj[:, t], hg, c)
            if train:
                hs.append(h); cs.append(c); cys.append(cy); wss.append(ws)
            outs.append(hy)
            h, c = hy, cy
        out = torch.stack(outs, 1)
        if train:
            ctx.save_for_backward(w_hh, torch.stack(hs, 1), torch.stack(cs, 0), torch.stack(cys, 0), torch.stack(wss, 0),
                                  kf if kf is not None else torch.empty(0))
        return out, h, c

    @staticmethod
    def backward(ctx, d_out, d_hT, d_cT):
        d_bih = d_bhh = None
        w_slot = None
        if ctx.native:
            w_hh, h_in, acts, cell, keep = ctx.saved_tensors
            G, T, B = ctx.dims
            want_state = ctx.needs_input_grad[4] or ctx.needs_input_grad[5]
            dg_all, dh, dc, part = _learn_native.seq_backward(d_out, d_hT, d_cT, w_hh, keep if ctx.has_keep else None, acts, cell,
                                                             ctx.dims, want_state, want_bias_grad=ctx.has_bias)
            w_slot, ih_slot, hh_slot = ctx.slots
            if part is not None:   # add the workgroups' partial sums up
                if ih_slot is not None and (hh_slot is not None or not ctx.needs_input_grad[3]):
                    _learn_native.sum_chunks(part, ih_slot, hh_slot, accumulate=True)
                else:
                    d_bih = _learn_native.sum_chunks(part)
                    d_bhh = d_bih if ctx.needs_input_grad[3] else None
        else:
            w_hh, h_in, cs, cys, wss, keep = ctx.saved_tensors
            G, T, B = h_in.shape[:3]
            dh, dc = d_hT, d_cT
            dgs = [None] * T
            for t in range(T - 1, -1, -1):
                if d_out is None:
                    dh_t = dh if dh is not None else torch.zeros_like(cs[t])
                else:
                    dh_t = d_out[:, t] if dh is None else d_out[:, t] + dh
                dg, dc = _cell_bwd(dh_t, dc, cs[t], cys[t], wss[t])
                dh = torch.bmm(dg, w_hh)
                if ctx.has_keep:
                    dh, dc = dh * keep[t], dc * keep[t]
                dgs[t] = dg
            dg_all = torch.stack(dgs, 1)                                          # [G, T, B, 4H] = d xproj
        a = dg_all.reshape(G, T * B, dg_all.shape[3])
        hin = h_in.reshape(G, T * B, h_in.shape[3])
        sh = ctx.shared
        if (ctx.native and sh is not None and sh.slot is not None and w_slot is not None and _learn_native.wgrad_supported(a, hin)):
            # W_ih's and W_hh's gradients share d xproj: one launch reads it once (the projection's backward then skips its own)
            _learn_native.dense_wgrad2(a, sh.x, hin, sh.slot, w_slot)
            sh.taken, d_w = True, None
        else:
            d_w = _weight_grad(a, hin, w_slot) if ctx.native else torch.bmm(a.transpose(1, 2), hin)   # one product over all steps
        return dg_all, d_w, d_bih, d_bhh, dh, dc, None, None, None, None


# ---------------------------------------------------------------------------------------------- flat parameters
class FlatParams:
    """All parameters of the G agents of a role as rows of one ``[G, P]`` fp32 buffer.  ``views[name]`` is the
    stacked ``[G, ...]`` view of one parameter in the compute dtype with ``.grad`` pre-assigned to the matching view
    of the flat gradient buffer, so a backward pass accumulates straight into ``grad``."""

    def __init__(self, shapes: Dict[str, Tuple[int, ...]], G: int, device, compute_dtype: torch.dtype):
        self.G, self.names = G, list(shapes)
        self.offsets, off = {}, 0
        for n, shp in shapes.items():
            k = int(math.prod(shp))
            self.offsets[n] = (off, k, tuple(shp))
            off += (k + 7) // 8 * 8                                              # 16-byte aligned starts in bf16
        self.P = off
        self.master = torch.zeros(G, self.P, dtype=torch.float32, device=device)
        self.compute_dtype = compute_dtype
        self.lp = self.master if compute_dtype == torch.float32 else torch.zeros(G, self.P, dtype=compute_dtype, device=device)
        self.grad = torch.zeros(G, self.P, dtype=compute_dtype, device=device)
        self.views: Dict[str, torch.Tensor] = {}
        for n, (o, k, shp) in self.offsets.items():
            v = self.lp[:, o:o + k].view(G, *shp).detach().requires_grad_(True)
            v.grad = self.grad[:, o:o + k].view(G, *shp)
            self.views[n] = v

    def master_view(self, name: str) -> torch.Tensor:
        o, k, shp = self.offsets[name]
        return self.master[:, o:o + k].view(self.G, *shp)

    def column_mask(self, prefix: str) -> torch.Tensor:
        """[P] fp32, 1 on the columns of the parameters whose name starts with ``prefix``."""
        m = torch.zeros(self.P, dtype=torch.float32, device=self.master.device)
        for n, (o, k, _) in self.offsets.items():
            if n.startswith(prefix):
                m[o:o + k] = 1.0
        return m

    @torch.no_grad()
    def refresh(self) -> None:
        """compute-dtype copy of the master weights (no-op in fp32)."""
        if self.lp is not self.master:
            self.lp.copy_(self.master)


# ---------------------------------------------------------------------------------------------- the stacked net
def _net_shapes(kind: str, R: int, arch: str = "lstm", vwidth: Optional[int] = None) -> Dict[str, Tuple[int, ...]]:
    """Parameter names / shapes of one network.  ``arch`` "lstm": the reference's recurrent pair (lstm_policy_net.py:28-53,
    lstm_value_net.py:46-75); "mlp": its non-recurrent pair -- ``Policy`` = the conv trunk + ``net`` 256-128-64-4
    (policy_net.py:17-33), ``Value`` = ``net`` over the whole flattened shared state of ``vwidth`` entries, 512-256-128-64-1
    (value_net.py:18-28)."""
    L2 = conv_out_len(R)
    if arch == "mlp":
        if kind == "policy":
            s = {"features_extractor.0.weight": (64, 2, 5), "features_extractor.0.bias": (64,),
                 "features_extractor.2.weight": (32, 64, 5), "features_extractor.2.bias": (32,),
                 "features_extractor.5.weight": (256, 32 * L2), "features_extractor.5.bias": (256,)}
            dims = [256, 128, 64, 4]
        else:
            assert vwidth, "the non-recurrent critic needs the width of the flattened shared state"
            s, dims = {}, [vwidth, 512, 256, 128, 64, 1]
        for j in range(len(dims) - 1):
            s[f"net.{2 * j}.weight"] = (dims[j + 1], dims[j])
            s[f"net.{2 * j}.bias"] = (dims[j + 1],)
        return s
    C, layers = (2, 1) if kind == "policy" else (4, 2)
    s = {"features_extractor.0.weight": (64, C, 5), "features_extractor.0.bias": (64,),
         "features_extractor.2.weight": (32, 64, 5), "features_extractor.2.bias": (32,),
         "features_extractor.5.weight": (256, 32 * L2), "features_extractor.5.bias": (256,)}
    for l in range(layers):
        s[f"lstm.weight_ih_l{l}"] = (4 * HIDDEN, 256 if l == 0 else HIDDEN)
        s[f"lstm.weight_hh_l{l}"] = (4 * HIDDEN, HIDDEN)
        s[f"lstm.bias_ih_l{l}"] = (4 * HIDDEN,)
        s[f"lstm.bias_hh_l{l}"] = (4 * HIDDEN,)
    dims = [HIDDEN, 128, 64, 4] if kind == "policy" else [HIDDEN, 256, 128, 64, 1]
    for j in range(len(dims) - 1):
        s[f"{kind}_head.{2 * j}.weight"] = (dims[j + 1], dims[j])
        s[f"{kind}_head.{2 * j}.bias"] = (dims[j + 1],)
    return s


def role_param_shapes(R: int, arch: str = "lstm", vwidth: Optional[int] = None) -> Dict[str, Tuple[int, ...]]:
    """Names/shapes of one agent's parameters: ``policy.*`` then ``value.*`` (names below the prefix as in models.py)."""
    out = {}
    for kind in ("policy", "value"):
        for n, shp in _net_shapes(kind, R, arch, vwidth).items():
            out[f"{kind}.{n}"] = shp
    return out


class _Lin(torch.autograd.Function):
    """y = x @ w^T + b for G stacked layers: x [G, M, in], w [G, out, in], b [G, out].  The bias gradient is a GEMM with a
    row of ones, not a column reduction: ``at::sum`` over hundreds of thousands of rows is a multi-block reduction with
    a semaphore buffer, and inside a replayed HIP graph of the backward pass that reduction returned garbage for
    exactly these bias gradients (ROCm 7.2 / torch 2.10; weights and activations were right).  A GEMM has no such state
    and accumulates in fp32."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        return torch.baddbmm(b.unsqueeze(1), x, w.transpose(1, 2))

    @staticmethod
    def backward(ctx, go):
        x, w = ctx.saved_tensors
        go = go.contiguous()
        dx = torch.bmm(go, w) if ctx.needs_input_grad[0] else None
        dw = torch.bmm(go.transpose(1, 2), x)
        ones = torch.ones(go.shape[0], 1, go.shape[1], dtype=go.dtype, device=go.device)
        db = torch.bmm(ones, go).squeeze(1)
        return dx, dw, db


_OWN_GEMMS = os.environ.get("CAT_LIB_GEMM", "0") != "1"   # the layers' products through csrc/cat_dense.hip (else the BLAS library)


class _SharedWgrad:
    """Hand-over between a bare projection (``_LinAct`` with ``shared``) and the ``_LSTMSeq`` that consumes its output: the
    projection's forward leaves its input and its weight's gradient slot here; the recurrence's backward -- which runs first
    and holds the gradient both weight gradients are products of -- takes both in one launch and says so."""

    def __init__(self):
        self.x = self.slot = None
        self.taken = False


class _LinAct(torch.autograd.Function):
    """act(x @ w^T + b) for G stacked layers in bf16 on the GPU (``csrc/cat_dense.hip``): product + bias + activation in
    one MFMA kernel, the activation's derivative and the bias gradient (column sums) in one pass over the incoming
    gradient, the input gradient and the weight gradient as MFMA kernels of their own (CAT_LIB_GEMM=1: the two
    products that are not reductions over the rows through the BLAS library instead, with the bias / activation as an
    in-place pass).  act: 0 none, 1 ReLU, 2 tanh; b None = a bare product.
    Gradients of FlatParams leaves are added straight into their slots of the flat gradient buffer (see _LSTMSeq)."""

    @staticmethod
    def forward(ctx, x, w, b, act, shared=None):
        ctx.own = _OWN_GEMMS and _learn_native.gemm_supported(x, w)
        ctx.shared = shared
        if shared is not None:
            shared.x, shared.slot, shared.taken = x, _grad_slot(w), False
        if ctx.own:   # product, bias and activation in one kernel (csrc/cat_dense.hip)
            y = _learn_native.dense_forward(x, w, b, act)
        else:
            y = torch.bmm(x, w.transpose(1, 2))
            if b is not None:
                _learn_native.dense_bias_act_(y, b, act)
        ctx.save_for_backward(x, w, y if act != 0 else None)
        ctx.act, ctx.has_bias = act, b is not None
        ctx.slots = (_grad_slot(w), _grad_slot(b))
        return y

    @staticmethod
    def backward(ctx, go):
        x, w, y = ctx.saved_tensors
        w_slot, b_slot = ctx.slots
        g, db, bias_job = go.contiguous(), None, None
        if ctx.has_bias:
            g, part = _learn_native.dense_act_grad(g, y, ctx.act)
            if b_slot is not None and w_slot is not None and _learn_native.wgrad_supported(g, x):
                bias_job = (part, b_slot)                      # its chunk sum rides in the weight gradient's launch
            elif b_slot is not None:
                _learn_native.sum_chunks(part, b_slot, accumulate=True)
            else:
                db = _learn_native.sum_chunks(part)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = _learn_native.dense_dgrad(g, w) if ctx.own else torch.bmm(g, w)
        if ctx.shared is not None and ctx.shared.taken:            # the recurrence's backward has added this weight gradient already
            assert bias_job is None
            return dx, None, db, None, None
        return dx, _weight_grad(g, x, w_slot, bias_job), db, None, None


def _lin_act(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor], act: int) -> torch.Tensor:
    """act(x [G, M, in] @ w[G, out, in]^T + b[G, out]); act 0 none, 1 ReLU, 2 tanh."""
    if x.is_cuda and x.dtype == torch.bfloat16:
        return _LinAct.apply(x, w, b, act)
    y = _Lin.apply(x, w, b)
    return torch.relu(y) if act == 1 else torch.tanh(y) if act == 2 else y


def _lin(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """x [G, M, in] @ w[G, out, in]^T + b[G, out]"""
    return _Lin.apply(x, w, b)


class _Expand(torch.autograd.Function):
    """out[g, j] = src[g, idx[j]] (idx == n selects 0): lays a small convolution weight out as the dense ("Toeplitz")
    matrix of the equivalent fully connected layer.  Backward: every source element gathers the gradients of the
    positions it was copied to (``back`` [n, L], entries == number of outputs select 0) and adds them up -- a gather plus
    a short per-element sum, no atomics and no multi-block reduction."""

    @staticmethod
    def forward(ctx, src, idx, back):
        ctx.save_for_backward(back)
        ctx.n = src.shape[1]
        ext = torch.cat([src, src.new_zeros(src.shape[0], 1)], dim=1)
        return ext.index_select(1, idx)

    @staticmethod
    def backward(ctx, go):
        (back,) = ctx.saved_tensors
        ext = torch.cat([go, go.new_zeros(go.shape[0], 1)], dim=1)
        return ext.index_select(1, back.reshape(-1)).view(go.shape[0], ctx.n, -1).sum(-1), None, None


class _ConvTrunk(torch.autograd.Function):
    """Conv1d(C, 64, 5, stride 2) -> ReLU -> Conv1d(64, 32, 5, stride 3) -> ReLU of the G stacked networks, bf16 on the
    GPU: one launch of ``libcat_learn.so`` forward and one backward (``csrc/cat_trunk.hip``); the 64-channel intermediate
    never leaves the LDS.  x [G, N, C * R] (channel, ray); w1 [G, 64, C, 5]; b1 [G, 64]; w2 [G, 32, 64, 5]; b2 [G, 32].
    Returns [G, N, L2 * 32] in (position, channel) order, like the dense path below.  ``rows`` (int64 [sel]), ``block``: x is
    the whole rollout buffer [G, steps * block, C * R] and the kernels read the minibatch of ``sel`` sequences per step out of
    it in place (N = steps * sel) -- no gathered copy of the observations."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, R, rows=None, block=0):
        out = _learn_native.trunk_forward(x, w1, b1, w2, b2, R, rows, block)
        ctx.save_for_backward(x, w1, b1, w2, b2, out)
        ctx.R, ctx.rows, ctx.block = R, rows, block
        ctx.slots = tuple(_grad_slot(p) for p in (w1, b1, w2, b2))
        return out

    @staticmethod
    def backward(ctx, go):
        x, w1, b1, w2, b2, out = ctx.saved_tensors
        parts = _learn_native.trunk_backward(x, w1, b1, w2, b2, out, go, ctx.R, ctx.rows, ctx.block)
        if all(s is not None for s in ctx.slots):   # slabs added up straight into the flat gradient buffer
            _learn_native.trunk_grad_finish(parts, w1.shape[2], ctx.R, ctx.slots)
            return None, None, None, None, None, None, None, None
        dw1, db1, dw2, db2 = _learn_native.trunk_grad_finish(parts, w1.shape[2], ctx.R)
        return None, dw1, db1, dw2, db2, None, None, None


DENSE_PAD = 1     # widths of the dense-trunk layers can be zero-padded to multiples of this (padding did not avoid the fault below)
DENSE_ROWS = 4096  # and their products run over at most this many rows at a time


def _conv_as_dense_indices(c_out: int, c_in: int, k: int, stride: int, l_in: int, in_layout: str, device,
                           in_width: Optional[int] = None, out_multiple: int = 1):
    """Index tensors for ``_Expand``: the [rows, cols] matrix of Conv1d(c_in, c_out, k, stride) on an input flattened as
    (channel, position) [``in_layout`` "cl", the observation vector] or (position, channel) ["lc", the previous stacked
    layer's output]; outputs are flattened (position, channel).  cols = ``in_width`` >= l_in * c_in and rows = l_out * c_out
    rounded up to ``out_multiple``: the extra rows / columns are zero weights (and zero biases).  Also the tiling of the
    bias.  Returns (idx, back, b_idx, b_back, l_out, rows)."""
    l_out = (l_in - k) // stride + 1
    n_w = c_out * c_in * k
    cols = l_in * c_in if in_width is None else in_width
    rows = -(-(l_out * c_out) // out_multiple) * out_multiple
    assert cols >= l_in * c_in
    idx = torch.full((rows, cols), n_w, dtype=torch.long)
    back = torch.full((n_w, l_out), rows * cols, dtype=torch.long)
    lo, co, ci, kk = torch.meshgrid(torch.arange(l_out), torch.arange(c_out), torch.arange(c_in), torch.arange(k), indexing="ij")
    pos = stride * lo + kk
    col = (ci * l_in + pos) if in_layout == "cl" else (pos * c_in + ci)
    row = lo * c_out + co
    w_id = (co * c_in + ci) * k + kk                                        # flat index into weight [c_out, c_in, k]
    idx[row.reshape(-1), col.reshape(-1)] = w_id.reshape(-1)
    back[w_id.reshape(-1), lo.reshape(-1)] = (row * cols + col).reshape(-1)
    b_idx = torch.full((rows,), c_out, dtype=torch.long)                     # c_out selects the appended zero
    b_idx[:l_out * c_out] = torch.arange(c_out).repeat(l_out)                # bias of output (l, c) = bias[c]
    b_back = (torch.arange(l_out).unsqueeze(0) * c_out + torch.arange(c_out).unsqueeze(1))   # [c_out, l_out]
    return idx.reshape(-1).to(device), back.to(device), b_idx.to(device), b_back.to(device), l_out, rows


class StackedNet:
    """Functional forward of the G stacked policy or value networks of a role over a FlatParams."""

    def __init__(self, kind: str, R: int, fp: FlatParams, arch: str = "lstm", vwidth: Optional[int] = None):
        assert kind in ("policy", "value") and arch in ("lstm", "mlp")
        self.kind, self.R, self.fp, self.arch, self.vwidth = kind, R, fp, arch, vwidth
        self.C, self.layers = (2, 1) if kind == "policy" else (4, 2)
        self.L1, self.L2 = (R - 5) // 2 + 1, conv_out_len(R)
        self.n_head = 3 if kind == "policy" else 4
        self.head = f"{kind}_head"
        if arch == "mlp":   # the reference's non-recurrent pair: no LSTM; the critic has no trunk either (an MLP over the whole state)
            self.layers, self.head = 0, "net"
            self.n_head = 3 if kind == "policy" else 5
        self.in_width = self.C * R if (arch == "lstm" or kind == "policy") else int(vwidth)
        dev = fp.master.device
        # Both convolutions run as ONE dense GEMM each, on the Toeplitz expansion of their (tiny) weights.  It multiplies
        # the flops by ~10 -- irrelevant on the matrix cores -- and removes what the update was actually spending its
        # time on: the im2col copies of a [samples * positions, channels * taps] matrix (hundreds of MB per minibatch),
        # their unfold backward, and GEMMs with a 10..320-wide inner dimension over half a million rows.
        # (bf16 on the GPU with R <= 64 the two convolutions are one kernel instead, ``_ConvTrunk``.)
        self.fused_trunk = os.environ.get("CAT_DENSE_TRUNK", "0") != "1"
        # With R = 90 the BLAS library's batched GEMM of the second layer faulted at training size on this stack (a memory
        # access fault inside its kernel for [3 x 16384 x 2752] x [2752 x 416], also zero-padded to 2816 / 512; the same call
        # in isolation passes: it depends on where its operands lie).  Rollout-tick sizes (4096 rows) and the R = 64 widths
        # have run for hours.  So: at most DENSE_ROWS rows per product (zero-padding the widths to multiples of DENSE_PAD is
        # available but did not help).
        up = lambda n: -(-n // DENSE_PAD) * DENSE_PAD
        self.in1 = up(self.C * R)
        self.t1 = _conv_as_dense_indices(64, self.C, 5, 2, R, "cl", dev, in_width=self.in1, out_multiple=DENSE_PAD)
        self.t2 = _conv_as_dense_indices(32, 64, 5, 3, self.L1, "lc", dev, in_width=self.t1[5], out_multiple=DENSE_PAD)
        assert self.t1[4] == self.L1 and self.t2[4] == self.L2

    def w(self, name: str) -> torch.Tensor:
        return self.fp.views[f"{self.kind}.{name}"]

    def initial_state(self, B: int):
        z = torch.zeros(self.layers, self.fp.G, B, HIDDEN, device=self.fp.master.device, dtype=self.fp.compute_dtype)
        return z, z.clone()

    def forward(self, x: torch.Tensor, state, keep: Optional[torch.Tensor], update_state: bool = False,
                select: Optional[torch.Tensor] = None):
        """x: [G, T, B, C*R] (time-major); state: (h, c) each [layers, G, B, H]; keep: [T, B] (1 = carry the state
        into step t, 0 = an episode starts there) or None.  Returns (out [G, T, B, n_out], new state).  ``update_state``
        (rollouts, no_grad, kernel path): the recurrence kernels write the new state INTO ``state`` and that is returned.
        ``select`` (int64 [B'], a PPO minibatch): the network runs on x[:, :, select] -- the sequences ``select`` of the
        rollout buffer x -- which the fused trunk kernels read in place; state and keep are those of the selection."""
        G, T, B_all, _ = x.shape
        B = B_all if select is None else select.shape[0]
        dt = self.fp.compute_dtype
        N = T * B
        if self.arch == "mlp" and self.kind == "value":   # value_net.py:18-28: no trunk, no recurrence
            if select is not None:
                x = x.index_select(2, select)
            y = x.reshape(G, N, self.in_width).to(dt)
            for j in range(self.n_head):
                y = _lin_act(y, self.w(f"net.{2 * j}.weight"), self.w(f"net.{2 * j}.bias"), 1 if j < self.n_head - 1 else 0)
            return y.view(G, T, B, -1), state
        fused = (self.fused_trunk and x.is_cuda and dt == torch.bfloat16 and _learn_native.trunk_supported(G, N, self.C, self.R))
        if select is not None and not (fused and x.dtype == dt and x.is_contiguous()):
            x, select = x.index_select(2, select), None
        z = x.reshape(G, T * x.shape[2], self.C * self.R).to(dt)                                 # (channel, ray) order
        if fused:
            z = _ConvTrunk.apply(z, self.w("features_extractor.0.weight"), self.w("features_extractor.0.bias"),
                                 self.w("features_extractor.2.weight"), self.w("features_extractor.2.bias"), self.R, select, B_all)
        else:   # fp32 / CPU, or a ray count whose intermediate does not fit the LDS (R = 90): dense GEMMs
            z = torch.nn.functional.pad(z, (0, self.in1 - self.C * self.R))
            layers = []
            for name, (idx, back, b_idx, b_back, l_out, rows), width in (("features_extractor.0", self.t1, self.in1),
                                                                         ("features_extractor.2", self.t2, self.t1[5])):
                layers.append((_Expand.apply(self.w(name + ".weight").reshape(G, -1), idx, back).view(G, rows, width),
                               _Expand.apply(self.w(name + ".bias"), b_idx, b_back)))
            # in row chunks of DENSE_ROWS: the shapes of a rollout tick, which the BLAS library handles (see __init__)
            outs = []
            for r0 in range(0, N, DENSE_ROWS):
                zc = z[:, r0:r0 + DENSE_ROWS]
                for w, b in layers:
                    zc = torch.relu(_lin(zc, w, b))                                              # [G, rows, l_out * c_out (+ zeros)], (l, c)
                outs.append(zc)
            z = outs[0] if len(outs) == 1 else torch.cat(outs, dim=1)
        wfc = self.w("features_extractor.5.weight").view(G, 256, 32, self.L2).transpose(2, 3).reshape(G, 256, self.L2 * 32)
        if z.shape[2] != self.L2 * 32:                                                           # the dense path's zero padding
            wfc = torch.nn.functional.pad(wfc, (0, z.shape[2] - self.L2 * 32))
        f = _lin_act(z, wfc, self.w("features_extractor.5.bias"), 2)                                 # [G, T*B, 256], tanh
        h0, c0 = state
        kp = None if keep is None else keep.to(torch.float32).contiguous()
        native = f.is_cuda and dt == torch.bfloat16
        inp = f
        hs, cs = [], []
        for l in range(self.layers):
            w_ih, b_ih, b_hh = (self.w(f"lstm.{n}_l{l}") for n in ("weight_ih", "bias_ih", "bias_hh"))
            shared = None
            if native:   # the recurrence kernel adds the biases and returns their gradient: the projection is a bare GEMM
                shared = _SharedWgrad() if torch.is_grad_enabled() else None
                xp = _LinAct.apply(inp, w_ih, None, 0, shared).view(G, T, B, 4 * HIDDEN)
            else:
                xp, b_ih, b_hh = _lin(inp, w_ih, b_ih + b_hh).view(G, T, B, 4 * HIDDEN), None, None
            in_place = update_state and native and not torch.is_grad_enabled() and h0.dtype == dt
            out, hT, cT = _LSTMSeq.apply(xp, self.w(f"lstm.weight_hh_l{l}"), b_ih, b_hh, h0[l].to(dt), c0[l].to(dt), kp,
                                         (h0[l], c0[l]) if in_place else None, torch.is_grad_enabled(), shared)
            hs.append(hT); cs.append(cT)
            inp = out.reshape(G, N, HIDDEN)
        y = inp
        for j in range(self.n_head):
            y = _lin_act(y, self.w(f"{self.head}.{2 * j}.weight"), self.w(f"{self.head}.{2 * j}.bias"), 1 if j < self.n_head - 1 else 0)
        if self.layers == 0 or (update_state and native and not torch.is_grad_enabled() and h0.dtype == dt):
            return y.view(G, T, B, -1), state
        return y.view(G, T, B, -1), (torch.stack(hs, 0), torch.stack(cs, 0))


# ---------------------------------------------------------------------------------------------- per-agent checkpoints
def _module_for(kind: str, R: int, arch: str = "lstm", vwidth: Optional[int] = None) -> nn.Module:
    if arch == "mlp":
        return Policy(R) if kind == "policy" else Value(int(vwidth))
    return LSTMPolicy(R) if kind == "policy" else LSTMValue(R)


@torch.no_grad()
def init_from_modules(fp: FlatParams, R: int, seeds: Sequence[int], arch: str = "lstm", vwidth: Optional[int] = None) -> None:
    """Initialise row g of the flat buffer exactly as freshly constructed per-agent modules would be (PyTorch's
    default initialisers, seeded per agent), so a stacked run and a per-module run start from the same weights."""
    for g, seed in enumerate(seeds):
        gen_state = torch.random.get_rng_state()
        torch.manual_seed(seed)
        for kind in ("policy", "value"):
            sd = _module_for(kind, R, arch, vwidth).state_dict()
            for n, v in sd.items():
                fp.master_view(f"{kind}.{n}")[g].copy_(v)
        torch.random.set_rng_state(gen_state)
    fp.refresh()


_OLD_NAMES = (("trunk.features.", "features_extractor."), ("trunk.lstm.", "lstm."))   # "cat-mappo-2" checkpoints (round 2)


def _reference_key(kind: str, name: str) -> str:
    """A round-2 checkpoint's key -> the reference modules' key (identity for keys that already are)."""
    for old, new in _OLD_NAMES:
        if name.startswith(old):
            return new + name[len(old):]
    if name.startswith("head."):
        return f"{kind}_head." + name[len("head."):]
    return name


@torch.no_grad()
def agent_state_dict(fp: FlatParams, g: int) -> Dict[str, Dict[str, torch.Tensor]]:
    """{"policy": state_dict, "value": state_dict} of agent g with the REFERENCE modules' keys
    (``features_extractor.N.*``, ``lstm.*``, ``policy_head.N.*`` / ``value_head.N.*``: lstm_policy_net.py:28-53,
    lstm_value_net.py:46-75), loadable by them and by models.LSTMPolicy / LSTMValue alike."""
    out = {"policy": {}, "value": {}}
    for n in fp.names:
        kind, name = n.split(".", 1)
        out[kind][name] = fp.master_view(n)[g].clone()
    return out


@torch.no_grad()
def load_agent_state_dict(fp: FlatParams, g: int, sd: Dict[str, Dict[str, torch.Tensor]], kinds=("policy", "value")) -> None:
    """Inverse of ``agent_state_dict``; also takes state dicts saved by the reference's own modules (same keys; a
    ``torch.compile`` wrapper's ``_orig_mod.`` prefix is dropped) and round-2 checkpoints (old key names)."""
    for kind in kinds:
        seen = set()
        for name, v in sd[kind].items():
            name = _reference_key(kind, name[len("_orig_mod."):] if name.startswith("_orig_mod.") else name)
            dst = fp.master_view(f"{kind}.{name}")[g]
            if tuple(v.shape) != tuple(dst.shape):
                raise ValueError(f"{kind}.{name}: checkpoint shape {tuple(v.shape)} != {tuple(dst.shape)} "
                                 f"(a different ray count? the reference's modules are built for 90 rays)")
            dst.copy_(v.to(fp.master.device, torch.float32))
            seen.add(f"{kind}.{name}")
        missing = [n for n in fp.names if n.startswith(kind + ".") and n not in seen]
        if missing:
            raise KeyError(f"checkpoint lacks {missing[:3]}{'...' if len(missing) > 3 else ''}")
    fp.refresh()


@torch.no_grad()
def adam_state_dict(fp: FlatParams, g: int, m: torch.Tensor, v: torch.Tensor, steps: torch.Tensor, lr: float,
                    betas=(0.9, 0.999), eps: float = 1e-8) -> dict:
    """Agent g's Adam state in ``torch.optim.Adam.state_dict()`` form over ``chain(policy.parameters(), value.parameters())``
    -- the optimiser skrl's MAPPO builds per agent and writes under "optimizer" -- so the reference can resume from it.
    Parameter order = registration order of the modules = ``fp.names``."""
    state = {}
    for i, n in enumerate(fp.names):
        o, k, shp = fp.offsets[n]
        state[i] = {"step": steps[g, o:o + k].max().to(torch.float32).cpu() if k else torch.tensor(0.0),
                    "exp_avg": m[g, o:o + k].view(*shp).clone(), "exp_avg_sq": v[g, o:o + k].view(*shp).clone()}
    group = {"lr": lr, "betas": tuple(betas), "eps": eps, "weight_decay": 0, "amsgrad": False, "maximize": False,
             "foreach": None, "capturable": False, "differentiable": False, "fused": None,
             "params": list(range(len(fp.names)))}
    return {"state": state, "param_groups": [group]}


@torch.no_grad()
def load_adam_state_dict(fp: FlatParams, g: int, m: torch.Tensor, v: torch.Tensor, steps: torch.Tensor, sd: dict) -> None:
    if "state" not in sd:                      # round-2 layout: flat rows
        m[g].copy_(sd["m"]); v[g].copy_(sd["v"]); steps[g].copy_(sd["steps"])
        return
    m[g].zero_(); v[g].zero_(); steps[g].zero_()
    for i, n in enumerate(fp.names):
        st = sd["state"].get(i)
        if st is None:                         # a parameter that never received a gradient has no entry in torch's Adam
            continue
        o, k, _ = fp.offsets[n]
        m[g, o:o + k].copy_(st["exp_avg"].reshape(-1)); v[g, o:o + k].copy_(st["exp_avg_sq"].reshape(-1))
        steps[g, o:o + k].fill_(float(st["step"]))
