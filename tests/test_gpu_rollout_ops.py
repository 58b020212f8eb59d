"""GPU: the rollout bookkeeping kernels (uhc_amd/csrc/uhc_rollout.hip, through the C-ABI) against the framework expressions they replace
and against the reference's row-by-row Welford filter; then a whole sampling pass with and without them."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _welford(rows, n=0, M=None, S=None):
    """uhc/khrylib/utils/zfilter.py:17-27, one row at a time."""
    d = rows.shape[1]
    M = np.zeros(d) if M is None else M.copy()
    S = np.zeros(d) if S is None else S.copy()
    for x in rows:
        n += 1
        if n == 1:
            M[...] = x
        else:
            old = M.copy()
            M[...] = old + (x - old) / n
            S[...] = S + (x - old) * (x - M)
    return n, M, S


@pytest.mark.parametrize("n_rows,dim", [(1024, 657), (100, 70), (1, 5), (4096, 33)])
def test_filter_push_equals_row_by_row_welford(n_rows, dim):
    import torch
    from uhc_amd.khrylib.utils.zfilter import RunningStat
    rng = np.random.default_rng(n_rows + dim)
    x1 = rng.normal(loc=rng.normal(scale=3.0, size=dim), scale=rng.uniform(0.01, 2.0, size=dim), size=(n_rows, dim))
    x2 = rng.normal(loc=1.0, scale=0.5, size=(n_rows, dim))
    w = (rng.uniform(size=n_rows) < 0.3).astype(np.int32)
    rs = RunningStat((dim,))
    rs.push_batch(torch.from_numpy(x1).cuda())
    n, M, S = _welford(x1)
    assert rs.n == n
    np.testing.assert_allclose(rs.mean, M, rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(rs._S, S, rtol=1e-11, atol=1e-12)
    rs.push_batch(torch.from_numpy(x2).cuda(), weights=torch.from_numpy(w).cuda())  # 0/1 weights: only those rows count
    n, M, S = _welford(x2[w != 0], n, M, S)
    assert rs.n == n
    np.testing.assert_allclose(rs.mean, M, rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(rs._S, S, rtol=1e-11, atol=1e-12)
    rs.push_batch(torch.from_numpy(x2).cuda(), weights=torch.zeros(n_rows, dtype=torch.int32, device="cuda"))  # nothing selected: unchanged
    assert rs.n == n
    np.testing.assert_allclose(rs.mean, M, rtol=1e-12, atol=1e-13)
    # bit-reproducible
    a, b = RunningStat((dim,)), RunningStat((dim,))
    for r in (a, b):
        r.push_batch(torch.from_numpy(x1).cuda())
        r.push_batch(torch.from_numpy(x2).cuda(), weights=torch.from_numpy(w).cuda())
    assert np.array_equal(a.mean, b.mean) and np.array_equal(a._S, b._S)


def test_filter_apply_is_the_reference_expression_bit_for_bit():
    import torch
    from uhc_amd.khrylib.utils.zfilter import ZFilter
    rng = np.random.default_rng(3)
    x = torch.from_numpy(rng.normal(scale=4.0, size=(257, 41))).cuda()
    for demean, destd, clip in [(True, True, 5.0), (True, False, 0.0), (False, True, 10.0)]:
        f = ZFilter((41,), demean=demean, destd=destd, clip=clip)
        f.rs.push_batch(x)
        y = f(x, update=False)  # library kernel
        ref = f._call_torch(x)  # the same filter through framework ops
        assert torch.equal(y, ref)
        t = torch.zeros(1, dtype=torch.long, device="cuda")
        out = torch.empty_like(x)
        assert f(x, update=False, out=out, step_counter=t) is out and torch.equal(out, ref) and int(t.item()) == 1
    g = ZFilter((41,))  # n = 1: std = |mean| (zfilter.py:45)
    g.rs.push_batch(x[:1])
    assert torch.equal(g(x, update=False), g._call_torch(x))


def test_act_and_record_equal_the_framework_expressions():
    import torch
    from uhc_amd import rollout_ops as ops
    torch.manual_seed(0)
    n_env, T, od, ad = 96, 5, 37, 11
    dev = "cuda"
    state = torch.randn(n_env, od, dtype=torch.float64, device=dev)
    mean = torch.randn(n_env, ad, dtype=torch.float64, device=dev)
    log_std = torch.randn(1, ad, dtype=torch.float64, device=dev) * 0.5 - 2.0
    noise = torch.randn(n_env, ad, dtype=torch.float64, device=dev)
    flags = (torch.rand(T, n_env, device=dev) < 0.4).double()
    states = torch.zeros(n_env, T, od, dtype=torch.float64, device=dev)
    actions = torch.zeros(n_env, T, ad, dtype=torch.float64, device=dev)
    action = torch.zeros(n_env, ad, dtype=torch.float64, device=dev)
    t = torch.full((1,), 3, dtype=torch.long, device=dev)
    ops.act(t, state, mean, log_std, noise, flags, states, actions, action)
    want = torch.where(flags[3].reshape(-1, 1).bool(), mean, mean + torch.exp(log_std.expand_as(mean)) * noise)
    # (exp of the library kernel and of the framework may differ in the last bit; everything else is the same arithmetic)
    assert float((action - want).abs().max()) < 1e-15 and torch.equal(actions[:, 3], action) and torch.equal(states[:, 3], state)
    assert torch.equal(action[flags[3] != 0], mean[flags[3] != 0])
    assert float(actions[:, :3].abs().max()) == 0.0 and float(states[:, 4].abs().max()) == 0.0
    # record
    reward = torch.rand(n_env, dtype=torch.float64, device=dev)
    done = (torch.rand(n_env, device=dev) < 0.2).int()
    end = (torch.rand(n_env, device=dev) < 0.1).int()
    parts = torch.rand(n_env, 6, dtype=torch.float64, device=dev)
    er = torch.full((), 2.5, dtype=torch.float64, device=dev)
    rewards = torch.zeros(n_env, T, dtype=torch.float64, device=dev)
    dones = torch.zeros(n_env, T, dtype=torch.float64, device=dev)
    crs = torch.full((), 1.0, dtype=torch.float64, device=dev)
    cis = torch.ones(5, dtype=torch.float64, device=dev)
    redo = torch.tensor([0, 1, 3, 0x70b, 0x80, 0xc1, 0x40000041, 0x60000041] * (n_env // 8) + [1] * (n_env % 8), dtype=torch.int32, device=dev)
    rc = torch.tensor([10, 20, 30, 40, 50, 60, 70], dtype=torch.long, device=dev)
    ops.record(t, reward, done, end, er, parts, 5, rewards, dones, crs, cis, redo, rc)
    assert rc.tolist() == [10 + int(((redo & 1) != 0).sum()), 20 + int(((redo & 2) != 0).sum()), 30 + int(((redo & 0x80) != 0).sum()), 40 + int(((redo & 0x40) != 0).sum()), 50 + int(((redo & 8) != 0).sum()),
                           60 + int(((redo & (1 << 30)) != 0).sum()), 70 + int(((redo & (1 << 29)) != 0).sum())]  # (tier 4: solved by Newton on the primal; its iteration cap)
    assert torch.equal(rewards[:, 3], reward + end.double() * 2.5) and torch.equal(dones[:, 3], done.double())
    np.testing.assert_allclose(float(crs), 1.0 + float(reward.sum()), rtol=1e-14)
    np.testing.assert_allclose(cis.cpu().numpy(), 1.0 + parts[:, :5].sum(0).cpu().numpy(), rtol=1e-14)


def test_sampling_pass_with_and_without_the_library_bookkeeping(tmp_path, monkeypatch):
    """One sampling pass (graphs off, same seeds) through the library's bookkeeping kernels and through the framework ops: same batch.
    The two filters round differently in the last bits, the physics amplifies that a little over the pass."""
    import torch
    from tests.test_gpu_agent import _cfg, _loader
    from uhc_amd import rollout_ops
    from uhc_amd.agents import agent_dict
    torch.set_default_dtype(torch.float64)
    out = []
    for fused in (True, False):
        if not fused:
            monkeypatch.setattr(rollout_ops, "usable", lambda *a: False)
        torch.manual_seed(1)
        np.random.seed(1)
        cfg = _cfg(tmp_path, n_env=32, batch=32 * 6)
        agent = agent_dict[cfg.agent_name](cfg, torch.float64, torch.device("cuda", 0), data_loader=_loader(cfg))
        agent.use_graph = False
        agent.seed(5)
        batch, log = agent.sample(cfg.min_batch_size)
        out.append((batch.states.clone(), batch.actions.clone(), batch.rewards.clone(), batch.masks.clone(), agent.running_state.rs.n, log.avg_c_reward))
    a, b = out
    assert a[4] == b[4] and torch.equal(a[3], b[3])
    assert float((a[0] - b[0]).abs().max()) < 1e-7 and float((a[1] - b[1]).abs().max()) < 1e-7 and float((a[2] - b[2]).abs().max()) < 1e-7
    assert abs(a[5] - b[5]) < 1e-9
