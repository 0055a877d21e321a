"""GPU: the convolutional-trunk kernels of libcat_learn.so (include/cat_trunk.h, csrc/cat_trunk.hip) against a plain
PyTorch fp32 reference of the same four layers (Conv1d written out as unfold + einsum and checked against
torch.nn.functional.conv1d: the reference's nn.Conv1d modules).

Tolerances: operands, the LDS-resident intermediate and the stored result are bf16, accumulation is fp32; the reference
runs in fp32 from the SAME bf16-rounded inputs and weights and rounds its intermediate to bf16 too (otherwise the ReLU
masks of the second layer differ on the ~0.1 % of elements whose pre-activation is within a bf16 ulp of zero, which a
sum of random-sign terms shows as sqrt(0.1 %) = 3 % relative error).  Forward: |err| <= 2e-2 on O(1) values.  Parameter gradients
(sums over thousands of samples): relative L2 error <= 2 %, cosine >= 0.999."""
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def _conv1d(x, w, b, stride):
    """torch.nn.Conv1d's arithmetic written out (windows by unfold, one einsum): the same sums as F.conv1d, in fp32,
    without MIOpen's kernel search (tens of seconds per new shape on a fresh box); checked against F.conv1d below."""
    win = x.unfold(2, w.shape[2], stride)                                    # [N, C_in, L_out, K]
    return torch.einsum("nclk,ock->nol", win, w) + b.view(1, -1, 1)


def _reference(x, w1, b1, w2, b2, C, R):
    """x [G, N, C*R] (channel, ray) fp32 -> [G, N, L2*32] (position, channel)."""
    outs = []
    for g in range(x.shape[0]):
        z = torch.relu(_conv1d(x[g].view(-1, C, R), w1[g], b1[g], 2))
        z = z.to(torch.bfloat16).float()        # the kernels keep the intermediate in bf16 (as a bf16 torch model would)
        z = torch.relu(_conv1d(z, w2[g], b2[g], 3))                          # [N, 32, L2]
        outs.append(z.transpose(1, 2).reshape(z.shape[0], -1))
    return torch.stack(outs, 0)


def test_the_reference_formulation_is_conv1d():
    import torch.nn.functional as F
    gen = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(50, 4, 64, generator=gen, device="cuda")
    w, b = torch.randn(64, 4, 5, generator=gen, device="cuda"), torch.randn(64, generator=gen, device="cuda")
    assert torch.allclose(_conv1d(x, w, b, 2), F.conv1d(x, w, b, stride=2), atol=1e-4, rtol=1e-4)
    z = torch.randn(50, 64, 30, generator=gen, device="cuda")
    w2, b2 = torch.randn(32, 64, 5, generator=gen, device="cuda"), torch.randn(32, generator=gen, device="cuda")
    assert torch.allclose(_conv1d(z, w2, b2, 3), F.conv1d(z, w2, b2, stride=3), atol=1e-3, rtol=1e-4)


def _case(G, N, C, R, seed):
    import torch
    gen = torch.Generator(device="cuda").manual_seed(seed)
    bf = torch.bfloat16
    x = torch.rand(G, N, C * R, generator=gen, device="cuda").to(bf)
    flat = torch.randn(G, 64 * C * 5 + 64 + 32 * 64 * 5 + 32 + 40, generator=gen, device="cuda")
    o = 8
    w1 = (0.4 * flat[:, o:o + 64 * C * 5]).to(bf).view(G, 64, C, 5); o += 64 * C * 5
    b1 = (0.2 * flat[:, o:o + 64]).to(bf); o += 64
    w2 = (0.08 * flat[:, o:o + 32 * 64 * 5]).to(bf).view(G, 32, 64, 5); o += 32 * 64 * 5
    b2 = (0.2 * flat[:, o:o + 32]).to(bf)
    return x, w1, b1, w2, b2


def _rel(a, b):
    import torch
    a, b = a.float().flatten(), b.float().flatten()
    return float((a - b).norm() / b.norm().clamp_min(1e-12)), float(torch.dot(a, b) / (a.norm() * b.norm()).clamp_min(1e-12))


@pytest.mark.parametrize("G,N,C,R", [(3, 1000, 2, 64), (3, 4099, 4, 64), (1, 16, 2, 64), (5, 300, 4, 32), (2, 20000, 2, 64),
                                     (3, 2000, 4, 90), (2, 1500, 2, 90), (1, 100, 4, 102), (2, 333, 2, 22)])
def test_forward_and_parameter_gradients_match_conv1d_in_fp32(G, N, C, R):
    import torch
    from as_cops_and_thieves_amd.selfplay.stacked import _ConvTrunk
    x, w1, b1, w2, b2 = _case(G, N, C, R, seed=G * 100 + C * 10 + R)
    leaves = [t.detach().clone().requires_grad_(True) for t in (w1, b1, w2, b2)]
    out = _ConvTrunk.apply(x, *leaves, R)
    ref_leaves = [t.detach().float().clone().requires_grad_(True) for t in (w1, b1, w2, b2)]
    ref = _reference(x.float(), *ref_leaves, C, R)
    assert out.shape == ref.shape and out.dtype == torch.bfloat16
    assert float((out.detach().float() - ref.detach()).abs().max()) <= 2e-2 * max(1.0, float(ref.detach().abs().max()))
    gen = torch.Generator(device="cuda").manual_seed(5)
    r = torch.randn(ref.shape, generator=gen, device="cuda").to(torch.bfloat16)
    (out.float() * r.float()).sum().backward()
    (ref * r.float()).sum().backward()
    torch.cuda.synchronize()
    errs = {name: _rel(got.grad, want.grad) for name, got, want in zip(("d_w1", "d_b1", "d_w2", "d_b2"), leaves, ref_leaves)}
    print(errs)
    for name, (err, cos) in errs.items():
        assert err <= 2e-2 and cos >= 0.999, (name, errs)


def test_stacked_network_with_the_fused_trunk_equals_the_dense_path():
    """StackedNet.forward through _ConvTrunk against the same network through the Toeplitz-dense GEMMs (bf16 both)."""
    import torch
    from as_cops_and_thieves_amd.selfplay.stacked import FlatParams, StackedNet, init_from_modules, role_param_shapes
    G, T, B, R = 3, 4, 64, 64
    fp = FlatParams(role_param_shapes(R), G, torch.device("cuda"), torch.bfloat16)
    init_from_modules(fp, R, seeds=[1, 2, 3])
    fp.refresh()
    outs = []
    for fused in (True, False):
        net = StackedNet("value", R, fp)
        net.fused_trunk = fused
        x = torch.rand(G, T, B, 4 * R, generator=torch.Generator(device="cuda").manual_seed(0), device="cuda")
        fp.grad.zero_()
        y, _ = net.forward(x, net.initial_state(B), None)
        y.float().square().sum().backward()
        outs.append((y.detach().float().clone(), fp.grad.float().clone()))
    torch.cuda.synchronize()
    assert float((outs[0][0] - outs[1][0]).abs().max()) <= 2e-2
    err, cos = _rel(outs[0][1], outs[1][1])
    assert err <= 5e-2 and cos >= 0.995, (err, cos)


@pytest.mark.parametrize("C,R,steps,block,sel", [(2, 64, 16, 300, 75), (4, 64, 5, 64, 64), (4, 90, 3, 200, 37)])
def test_minibatch_rows_read_in_place_equal_the_gathered_copy(C, R, steps, block, sel):
    """cat_trunk_rows: the kernels read sample n = (step, j) from row step * block + rows[j] of the rollout buffer -- the same
    bits, forward and in every parameter gradient, as running them on torch's gathered copy of those rows."""
    import torch
    from as_cops_and_thieves_amd.selfplay.stacked import _ConvTrunk
    G = 3
    x, w1, b1, w2, b2 = _case(G, steps * block, C, R, seed=C * R + sel)
    gen = torch.Generator(device="cuda").manual_seed(3)
    rows = torch.randperm(block, generator=gen, device="cuda")[:sel].contiguous()
    gathered = x.view(G, steps, block, C * R).index_select(2, rows).reshape(G, steps * sel, C * R).contiguous()
    results = []
    for args in ((x, rows, block), (gathered, None, 0)):
        leaves = [t.detach().clone().requires_grad_(True) for t in (w1, b1, w2, b2)]
        out = _ConvTrunk.apply(args[0], *leaves, R, args[1], args[2])
        r = torch.randn(out.shape, generator=torch.Generator(device="cuda").manual_seed(5), device="cuda").to(torch.bfloat16)
        (out.float() * r.float()).sum().backward()
        results.append((out.detach(), [t.grad for t in leaves]))
    torch.cuda.synchronize()
    assert results[0][0].shape == (G, steps * sel, results[1][0].shape[2]) and torch.equal(results[0][0], results[1][0])
    for got, want in zip(results[0][1], results[1][1]):
        assert torch.equal(got, want)
