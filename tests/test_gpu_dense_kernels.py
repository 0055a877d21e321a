"""GPU: the dense-layer epilogue kernels of libcat_learn.so (include/cat_dense.h) against plain PyTorch fp32: act(y + b)
in place, and d_y * act'(y) with its column sums (the bias gradient).  Tolerances: bf16 storage (2^-8 relative) on the
elementwise results, bf16 rounding again on the column sums (they are stored as bf16 gradients)."""
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("G,M,out,act", [(3, 16384, 512, 0), (3, 16384, 256, 2), (3, 5000, 128, 1), (2, 777, 64, 1), (3, 4096, 4, 0),
                                         (3, 4096, 1, 0), (1, 100, 12, 2)])
def test_bias_act_and_its_gradient_match_torch(G, M, out, act):
    import torch
    from as_cops_and_thieves_amd import _learn_native as ln
    gen = torch.Generator(device="cuda").manual_seed(out * 10 + act)
    pre = torch.randn(G, M, out, generator=gen, device="cuda").to(torch.bfloat16)
    flat = torch.randn(G, out + 24, generator=gen, device="cuda").to(torch.bfloat16)
    bias = flat[:, 8:8 + out]                                             # a row-strided view, as a FlatParams parameter
    f = {0: lambda t: t, 1: torch.relu, 2: torch.tanh}[act]
    want = f(pre.float() + bias.float().unsqueeze(1))
    y = ln.dense_bias_act_(pre.clone(), bias, act)
    assert torch.allclose(y.float(), want, rtol=2 ** -7, atol=2e-3)
    d_y = torch.randn(G, M, out, generator=gen, device="cuda").to(torch.bfloat16)
    g, part = ln.dense_act_grad(d_y, y, act)
    db = ln.sum_chunks(part).float()
    slot = torch.ones(G, out + 8, device="cuda", dtype=torch.bfloat16)          # accumulate into a row-strided gradient slot
    ln.sum_chunks(part, slot[:, 4:4 + out], accumulate=True)
    der = {0: torch.ones_like(want), 1: (y.float() > 0).float(), 2: 1 - y.float() ** 2}[act]
    g_want = d_y.float() * der
    torch.cuda.synchronize()
    assert torch.allclose(g.float(), g_want, rtol=2 ** -7, atol=1e-6)
    assert torch.allclose(db, g.float().sum(1), rtol=2 ** -7, atol=2e-2)
    assert torch.allclose(slot[:, 4:4 + out].float(), 1 + g.float().sum(1), rtol=2 ** -7, atol=2e-2) and bool((slot[:, :4] == 1).all())


@pytest.mark.parametrize("G,K,M,N", [(3, 16384, 512, 256), (3, 16384, 256, 288), (2, 5000, 64, 128), (1, 777, 128, 64), (3, 16384, 512, 128),
                                     (2, 1000, 8, 8), (3, 4096, 136, 72), (3, 16384, 4, 64), (3, 16384, 1, 64), (2, 3000, 5, 12)])
def test_weight_gradient_kernel_matches_a_gemm_in_fp32(G, K, M, N):
    """sum_k g[k, m] x[k, n] through the split-K MFMA kernel (transposed LDS reads) against torch.bmm in fp32 on the same
    bf16 inputs; asymmetric random data, ragged tiles (M, N not multiples of 128; K not a multiple of 32).  The result is
    stored as bf16: 2^-8 relative + the fp32 accumulation-order difference."""
    import torch
    from as_cops_and_thieves_amd import _learn_native as ln
    gen = torch.Generator(device="cuda").manual_seed(K + M)
    g = torch.randn(G, K, M, generator=gen, device="cuda").to(torch.bfloat16)
    x = (torch.randn(G, K, N, generator=gen, device="cuda") + 0.5).to(torch.bfloat16)
    want = torch.bmm(g.float().transpose(1, 2), x.float())
    got = ln.dense_wgrad(g, x)
    scale = float(want.abs().max())
    assert got.shape == (G, M, N) and float((got.float() - want).abs().max()) <= 2 ** -7 * scale
    flat = torch.ones(G, M * N + 16, device="cuda", dtype=torch.bfloat16)          # accumulate into a row-strided slot
    slot = flat[:, 8:8 + M * N].view(G, M, N)
    assert ln.dense_wgrad(g, x, slot) is None
    torch.cuda.synchronize()
    assert float((slot.float() - (1 + want)).abs().max()) <= 2 ** -6 * scale and bool((flat[:, :8] == 1).all())


@pytest.mark.parametrize("G,M,N,K,act", [(3, 16384, 256, 288, 2), (3, 16384, 512, 256, 0), (3, 4096, 128, 128, 1), (2, 1000, 4, 64, 0),
                                          (3, 777, 1, 64, 0), (1, 130, 136, 72, 1), (3, 16384, 256, 416, 2), (5, 300, 64, 128, 1)])
def test_layer_forward_and_input_gradient_kernels_match_fp32_gemms(G, M, N, K, act):
    """cat_dense_forward / cat_dense_dgrad against torch in fp32 on the same bf16 operands (w as a row-strided view of a
    flat buffer, ragged tiles, the 4- and 1-wide heads).  Stored as bf16: 2^-7 relative to the largest value."""
    import torch
    from as_cops_and_thieves_amd import _learn_native as ln
    gen = torch.Generator(device="cuda").manual_seed(M + N + K)
    x = torch.randn(G, M, K, generator=gen, device="cuda").to(torch.bfloat16)
    flat = (torch.randn(G, N * K + 40, generator=gen, device="cuda") / K ** 0.5).to(torch.bfloat16)
    w = flat[:, 16:16 + N * K].view(G, N, K)
    b = torch.randn(G, N, generator=gen, device="cuda").to(torch.bfloat16)
    f = {0: lambda t: t, 1: torch.relu, 2: torch.tanh}[act]
    want = f(torch.bmm(x.float(), w.float().transpose(1, 2)) + b.float().unsqueeze(1))
    got = ln.dense_forward(x, w, b, act)
    assert got.shape == (G, M, N) and float((got.float() - want).abs().max()) <= 2 ** -7 * max(1.0, float(want.abs().max()))
    bare = ln.dense_forward(x, w, None, 0)
    assert float((bare.float() - torch.bmm(x.float(), w.float().transpose(1, 2))).abs().max()) <= 2 ** -7 * 8
    gr = torch.randn(G, M, N, generator=gen, device="cuda").to(torch.bfloat16)
    dx_want = torch.bmm(gr.float(), w.float())
    dx = ln.dense_dgrad(gr, w)
    torch.cuda.synchronize()
    assert dx.shape == (G, M, K) and float((dx.float() - dx_want).abs().max()) <= 2 ** -7 * max(1.0, float(dx_want.abs().max()))


@pytest.mark.parametrize("G,K,M,N0,N1", [(3, 16384, 512, 256, 128), (2, 4099, 512, 128, 128), (1, 1024, 130, 72, 8)])
def test_two_inputs_sharing_one_gradient_get_their_weight_gradients_in_one_launch(G, K, M, N0, N1):
    """cat_dense_wgrad with a second input (an LSTM layer's W_ih and W_hh share d xproj): each gradient equals the fp32
    product of the same bf16 operands to the accuracy of a bf16 result, added INTO the slots."""
    import torch
    from as_cops_and_thieves_amd import _learn_native as ln
    gen = torch.Generator(device="cuda").manual_seed(K + N1)
    g = torch.randn(G, K, M, generator=gen, device="cuda").to(torch.bfloat16)
    x0 = torch.randn(G, K, N0, generator=gen, device="cuda").to(torch.bfloat16)
    x1 = torch.randn(G, K, N1, generator=gen, device="cuda").to(torch.bfloat16)
    flat = torch.zeros(G, M * (N0 + N1) + 16, dtype=torch.bfloat16, device="cuda")      # slots inside a wider flat buffer
    s0 = flat[:, 8:8 + M * N0].view(G, M, N0)
    s1 = flat[:, 8 + M * N0:8 + M * (N0 + N1)].view(G, M, N1)
    s0.fill_(1.0)
    ln.dense_wgrad2(g, x0, x1, s0, s1)
    torch.cuda.synchronize()
    for slot, x, base in ((s0, x0, 1.0), (s1, x1, 0.0)):
        want = torch.bmm(g.float().transpose(1, 2), x.float()) + base
        err = float((slot.float() - want).norm() / want.norm())
        assert err <= 6e-3, err
    assert float(flat[:, :8].abs().max()) == 0.0 and float(flat[:, 8 + M * (N0 + N1):].abs().max()) == 0.0
