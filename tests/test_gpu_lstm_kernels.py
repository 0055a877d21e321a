"""GPU: the LSTM-window kernels of libcat_learn.so (include/cat_lstm.h, csrc/cat_lstm.hip) against a plain PyTorch fp32
reference of the same recurrence (nn.LSTM semantics, gate order i f g o, states zeroed where keep == 0).

Tolerances: operands and stored results are bf16 (8 significant bits), accumulation and the cell are fp32; the
reference runs in fp32 from the SAME bf16-rounded inputs.  Forward values are O(1): |err| <= 2e-2.  Gradients are
compared per tensor by relative L2 error (<= 2 %) and direction (cosine >= 0.999)."""
import pytest

pytestmark = pytest.mark.gpu

H = 128


def _reference(xproj, w_hh, h0, c0, keep):
    """fp32, step by step, autograd-differentiable.  xproj [G,T,B,4H]; keep [T,B] or None."""
    import torch
    G, T, B, _ = xproj.shape
    h, c = h0, c0
    outs = []
    for t in range(T):
        if keep is not None:
            k = keep[t].view(1, B, 1)
            h, c = h * k, c * k
        pre = xproj[:, t] + torch.bmm(h, w_hh.transpose(1, 2))
        i, f, g, o = pre.chunk(4, dim=-1)
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
        h = torch.sigmoid(o) * torch.tanh(c)
        outs.append(h)
    return torch.stack(outs, 1), h, c


def _case(G, T, B, with_keep, seed, strided=False):
    import torch
    gen = torch.Generator(device="cuda").manual_seed(seed)
    dev = "cuda"
    if strided:                                    # xproj as a view with padded outer strides (what a caller's slice gives)
        big = torch.randn(G, T, B + 3, 4 * H + 8, generator=gen, device=dev).to(torch.bfloat16)
        xproj = big[:, :, 1:B + 1, 8:]
    else:
        xproj = torch.randn(G, T, B, 4 * H, generator=gen, device=dev).to(torch.bfloat16)
    flat = (0.15 * torch.randn(G, 4 * H * H + 24, generator=gen, device=dev)).to(torch.bfloat16)
    w_hh = flat[:, 16:16 + 4 * H * H].view(G, 4 * H, H)             # rows of a wider flat parameter buffer, as FlatParams
    h0 = (0.5 * torch.randn(G, B, H, generator=gen, device=dev)).to(torch.bfloat16)
    c0 = torch.randn(G, B, H, generator=gen, device=dev).to(torch.bfloat16)
    keep = (torch.rand(T, B, generator=gen, device=dev) > 0.2).float() if with_keep else None
    return xproj, w_hh, h0, c0, keep


def _rel(a, b):
    import torch
    a, b = a.float().flatten(), b.float().flatten()
    return float((a - b).norm() / b.norm().clamp_min(1e-12)), float(torch.dot(a, b) / (a.norm() * b.norm()).clamp_min(1e-12))


@pytest.mark.parametrize("G,T,B,with_keep,strided", [(3, 16, 100, True, False), (1, 1, 16, False, False), (5, 16, 1024, True, False),
                                                     (2, 7, 33, True, True), (3, 16, 48, False, False)])
def test_forward_and_backward_match_the_fp32_reference(G, T, B, with_keep, strided):
    import torch
    from as_cops_and_thieves_amd.selfplay.stacked import _LSTMSeq
    xproj, w_hh, h0, c0, keep = _case(G, T, B, with_keep, seed=G * 1000 + T * 10 + B, strided=strided)
    gen = torch.Generator(device="cuda").manual_seed(7)
    r_out = torch.randn(G, T, B, H, generator=gen, device="cuda")
    r_h, r_c = torch.randn(G, B, H, generator=gen, device="cuda"), torch.randn(G, B, H, generator=gen, device="cuda")

    bias = (0.3 * torch.randn(G, 4 * H, generator=gen, device="cuda")).to(torch.bfloat16) if B != 48 else None   # one case without
    leaves = [t.detach().clone().requires_grad_(True) for t in (xproj, w_hh, h0, c0)]
    b_leaf = None if bias is None else bias.detach().clone().requires_grad_(True)
    out, hT, cT = _LSTMSeq.apply(leaves[0], leaves[1], b_leaf, None, leaves[2], leaves[3], keep)
    assert out.dtype == torch.bfloat16 and out.shape == (G, T, B, H)
    loss = (out.float() * r_out).sum() + (hT.float() * r_h).sum() + (cT.float() * r_c).sum()
    loss.backward()

    ref_leaves = [t.detach().float().clone().requires_grad_(True) for t in (xproj, w_hh, h0, c0)]
    rb = None if bias is None else bias.detach().float().clone().requires_grad_(True)
    ro, rh, rc = _reference(ref_leaves[0] if rb is None else ref_leaves[0] + rb.view(G, 1, 1, -1), *ref_leaves[1:], keep)
    ((ro * r_out).sum() + (rh * r_h).sum() + (rc * r_c).sum()).backward()
    torch.cuda.synchronize()

    assert float((out.detach().float() - ro.detach()).abs().max()) <= 2e-2
    assert float((hT.detach().float() - rh.detach()).abs().max()) <= 2e-2 and float((cT.detach().float() - rc.detach()).abs().max()) <= 4e-2
    pairs = list(zip(("d_xproj", "d_w_hh", "d_h0", "d_c0"), leaves, ref_leaves)) + ([] if bias is None else [("d_bias", b_leaf, rb)])
    for name, got, want in pairs:
        err, cos = _rel(got.grad, want.grad)
        assert err <= 2e-2 and cos >= 0.999, (name, err, cos)


def test_inference_call_keeps_nothing_and_equals_the_training_forward():
    """no_grad (a rollout tick, T = 1) goes through the same kernel without the saved buffers."""
    import torch
    from as_cops_and_thieves_amd.selfplay.stacked import _LSTMSeq
    xproj, w_hh, h0, c0, keep = _case(3, 1, 4096, True, seed=3)
    with torch.no_grad():
        o1, h1, c1 = _LSTMSeq.apply(xproj, w_hh, None, None, h0, c0, keep, None, torch.is_grad_enabled())   # as StackedNet calls it
    leaves = [t.detach().clone().requires_grad_(True) for t in (xproj, w_hh, h0, c0)]
    o2, h2, c2 = _LSTMSeq.apply(leaves[0], leaves[1], None, None, leaves[2], leaves[3], keep)
    torch.cuda.synchronize()
    assert torch.equal(o1, o2) and torch.equal(h1, h2) and torch.equal(c1, c2) and torch.equal(o1[:, 0], h1)


def test_one_launch_per_direction_inside_a_hip_graph():
    """The kernels take the caller's stream: captured into a HIP graph with the rest of the PPO step and replayed."""
    import torch
    from as_cops_and_thieves_amd.selfplay.stacked import _LSTMSeq
    xproj, w_hh, h0, c0, keep = _case(3, 16, 256, True, seed=11)
    xs = xproj.clone().requires_grad_(True)
    ws = w_hh.detach().clone().requires_grad_(True)

    def step():
        out, hT, cT = _LSTMSeq.apply(xs, ws, None, None, h0, c0, keep)
        gx, gw = torch.autograd.grad(out.float().square().sum(), (xs, ws))
        return out, gx, gw
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            eager = [t.clone() for t in step()]
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        got = step()
    with torch.no_grad():
        xs.copy_(xproj)                                   # same inputs: the replay must reproduce the eager result bit for bit
    g.replay(); g.replay()
    torch.cuda.synchronize()
    for a, b in zip(got, eager):
        assert torch.equal(a, b)


def test_two_bias_vectors_are_added_by_the_kernel():
    """b_ih and b_hh handed over separately (as StackedNet does) = their bf16 sum handed over as one vector."""
    import torch
    from as_cops_and_thieves_amd import _learn_native
    xproj, w_hh, h0, c0, keep = _case(3, 5, 70, True, seed=21)
    gen = torch.Generator(device="cuda").manual_seed(9)
    b1 = (0.3 * torch.randn(3, 4 * H, generator=gen, device="cuda")).to(torch.bfloat16)
    b2 = (0.3 * torch.randn(3, 4 * H, generator=gen, device="cuda")).to(torch.bfloat16)
    for save in (False, True):
        two = _learn_native.seq_forward(xproj, w_hh, b1, h0, c0, keep, save=save, bias2=b2)
        one = _learn_native.seq_forward(xproj, w_hh, b1 + b2, h0, c0, keep, save=save)
        torch.cuda.synchronize()
        assert all(torch.equal(a, b) for a, b in zip(two[:3], one[:3]))
