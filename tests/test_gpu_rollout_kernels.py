"""GPU: the rollout-tick glue kernels of libcat_learn.so (include/cat_rollout.h) against the torch formulations they
replace: ``packing.py``'s rows (skrl's sorted-key flattening of the reference's Dict spaces) and inverse-CDF sampling."""
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("q11", [True, False])
def test_packed_rows_equal_packing_py(q11):
    import torch
    from as_cops_and_thieves_amd import VecCopsEnv, load_preset, packing
    from as_cops_and_thieves_amd import _learn_native as ln
    env = VecCopsEnv(load_preset("grandbyrinth", 3, 2), 300, num_rays=64, max_step_count=50, seed=4)
    env.reset()
    for t in range(5):
        env.step(env.random_actions(t))
    obs, state = env._obs(), env.state()
    agents = list(env.possible_agents)
    idx = [4, 0, 2]                                                   # a thief first: both teams' shared channels are used
    N, R, d, ty = 300, 64, 1.0 / 400.0, 0.25
    scale_p = torch.tensor([d] * R + [ty] * R, device="cuda")
    scale_v = torch.tensor(([d] * R + [ty] * R) * 2, device="cuda")
    want_p = torch.stack([packing.pack_policy_input(obs[agents[i]]) for i in idx]) * scale_p
    first = sorted(state)[0]
    want_v = torch.stack([packing.pack_agent_state(state[first if q11 else agents[i]])[:, :4 * R] for i in idx]) * scale_v
    big_p = torch.zeros(3, 2, N, 2 * R, device="cuda", dtype=torch.bfloat16)       # a time slice of a rollout buffer: strided rows
    big_v = torch.zeros(3, 2, N, 4 * R, device="cuda", dtype=torch.bfloat16)
    ln.rollout_pack(env.raw_outputs(), idx, 3, q11, d, ty, big_p[:, 1], big_v[:, 1])
    torch.cuda.synchronize()
    assert torch.equal(big_p[:, 1], want_p.to(torch.bfloat16)) and torch.equal(big_v[:, 1], want_v.to(torch.bfloat16))
    assert not bool(big_p[:, 0].any()) and float(want_v.abs().sum()) > 0
    env.close()


def test_sampled_actions_follow_the_inverse_cdf_and_land_in_the_action_matrix():
    import torch
    from as_cops_and_thieves_amd import _learn_native as ln
    G, N, A, T = 3, 5000, 5, 4
    gen = torch.Generator(device="cuda").manual_seed(9)
    logits = (2 * torch.randn(G, N, 4, generator=gen, device="cuda")).to(torch.bfloat16)
    values = torch.randn(G, N, generator=gen, device="cuda").to(torch.bfloat16)
    u = torch.rand(G, N, generator=gen, device="cuda")
    act_buf = torch.full((G, T, N), -1, dtype=torch.long, device="cuda")
    logp_buf, val_buf = torch.zeros(G, T, N, device="cuda"), torch.zeros(G, T, N, device="cuda")
    actions = torch.full((N, A), 7, dtype=torch.int32, device="cuda")
    cols = [1, 4, 2]
    ln.rollout_sample(logits, u, values, act_buf[:, 2], logp_buf[:, 2], val_buf[:, 2], actions, cols)
    torch.cuda.synchronize()
    logp_all = torch.log_softmax(logits.float(), dim=-1)
    cdf = torch.cumsum(logp_all.exp(), dim=-1)[..., :-1]
    want = (u.unsqueeze(-1) >= cdf).sum(-1)
    got = act_buf[:, 2]
    agree = float((got == want).float().mean())
    assert agree > 0.999                                              # a draw within float rounding of a CDF step may land on either side
    assert torch.allclose(logp_buf[:, 2], logp_all.gather(-1, got.unsqueeze(-1)).squeeze(-1), atol=1e-5)
    assert torch.equal(val_buf[:, 2], values.float()) and bool((act_buf[:, [0, 1, 3]] == -1).all())
    for g, c in enumerate(cols):
        assert torch.equal(actions[:, c].long(), got[g])
    assert bool((actions[:, [0, 3]] == 7).all())
    freq = torch.stack([(got == k).float().mean() for k in range(4)])
    assert float((freq - logp_all.exp().mean((0, 1))).abs().max()) < 0.02        # empirical frequencies follow the mean probabilities


def test_post_step_rewards_and_flags():
    import torch
    from as_cops_and_thieves_amd import _learn_native as ln
    N, A = 3000, 5
    gen = torch.Generator(device="cuda").manual_seed(3)
    raw = {"reward": torch.randn(N, A, generator=gen, device="cuda"), "terminated": (torch.rand(N, generator=gen, device="cuda") < 0.3).to(torch.uint8)}
    buf = torch.zeros(3, 4, N, device="cuda")
    done, start = torch.zeros(N, dtype=torch.bool, device="cuda"), torch.ones(N, dtype=torch.bool, device="cuda")
    keep = torch.full((1, N), 7.0, device="cuda")
    ln.rollout_post(raw, [4, 0, 2], buf[:, 1], done, start, keep)
    torch.cuda.synchronize()
    assert torch.equal(buf[:, 1], raw["reward"].t()[[4, 0, 2]]) and not bool(buf[:, [0, 2, 3]].any())
    want = raw["terminated"].bool()
    assert torch.equal(done, want) and torch.equal(start, want) and torch.equal(keep[0], (~want).float())
