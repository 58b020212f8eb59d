"""CPU tests of the learner: GAE / PPO pieces against reference golden vectors, the observation filter,
the clip sampler, config, and the 2-rank gloo gradient all-reduce path."""
import os
import sys
import types

import numpy as np
import pytest
import torch

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(autouse=True)
def f64():
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    yield
    torch.set_default_dtype(old)


def _nets(sd=23, ad=9, h=(32, 16)):
    from uhc_amd.khrylib.models.mlp import MLP
    from uhc_amd.khrylib.rl.core import PolicyGaussian, Value
    pol = PolicyGaussian(types.SimpleNamespace(policy_hsize=h, policy_htype="gelu", fix_std=True, log_std=-2.3), action_dim=ad, state_dim=sd)
    return pol, Value(MLP(sd, h, "gelu"))


def test_gae_matches_reference_flat_and_segmented():
    from uhc_amd.khrylib.rl.core import estimate_advantages
    g = np.load(os.path.join(G, "g8_gae.npz"))
    r, m, v = (torch.from_numpy(g[k]) for k in ("rewards", "masks", "values"))
    a, ret = estimate_advantages(r, m, v, 0.95, 0.95)
    np.testing.assert_allclose(a.numpy(), g["advantages"], atol=1e-13)
    np.testing.assert_allclose(ret.numpy(), g["returns"], atol=1e-13)
    # segment-parallel scan == flat scan when every segment ends with mask 0
    m2 = m.clone()
    m2[49::50] = 0
    a1, r1 = estimate_advantages(r, m2, v, 0.95, 0.95)
    a2, r2 = estimate_advantages(r, m2, v, 0.95, 0.95, seg_len=50)
    np.testing.assert_allclose(a1.numpy(), a2.numpy(), atol=1e-13)
    np.testing.assert_allclose(r1.numpy(), r2.numpy(), atol=1e-13)


def test_policy_value_ppo_loss_and_grads_match_reference():
    g = np.load(os.path.join(G, "g8_ppo_small.npz"))
    pol, val = _nets()
    pol.load_state_dict({k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("pol_")})
    val.load_state_dict({k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("val_")})
    x, a = torch.from_numpy(g["x"]), torch.from_numpy(g["a"])
    np.testing.assert_allclose(pol(x).loc.detach().numpy(), g["mean"], atol=1e-14)
    np.testing.assert_allclose(pol.get_log_prob(x, a).detach().numpy(), g["log_prob"], atol=1e-12)
    np.testing.assert_allclose(val(x).detach().numpy(), g["value"], atol=1e-14)
    from uhc_amd.khrylib.rl.agents import AgentPPO
    ag = AgentPPO(env=None, policy_net=pol, value_net=val, dtype=torch.float64, device=torch.device("cpu"), gamma=0.95, data_loader=None)
    ind = torch.arange(64)
    loss = ag.ppo_loss(x, a, torch.from_numpy(g["advs"]), torch.from_numpy(g["fixed_lp"]), ind)
    assert loss.item() == pytest.approx(float(g["ppo_loss"]), abs=1e-13)
    loss.backward()
    for n, p in pol.named_parameters():
        if p.grad is not None:
            np.testing.assert_allclose(p.grad.numpy(), g["polgrad_" + n], atol=1e-13)
    vl = (val(x) - torch.from_numpy(g["rets"])).pow(2).mean()
    assert vl.item() == pytest.approx(float(g["value_loss"]), abs=1e-13)
    n_pol = sum(p.numel() for p in _nets(657, 105, (2048, 1024, 512))[0].parameters())
    assert n_pol == 4024530  # policy parameter count at 657 -> 105


@pytest.mark.parametrize("branch", ["full", "mini"])
def test_ppo_update_policy_matches_the_reference_update(branch):
    """Fixture G8b: the reference's own AgentPPO.update_policy (uhc/khrylib/rl/agents/agent_ppo.py:16-51) run end to end -- three epochs of the full-batch
    branch and of the mini-batch branch (:23-43; numpy's global generator shuffles, the permutations compose, the ragged tail is dropped): parameters of both
    nets after the update."""
    g = np.load(os.path.join(G, "g8b_ppo_update.npz"))
    pol, val = _nets(17, 6, (24, 12))
    pol.load_state_dict({k[5:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("pol0_")})
    val.load_state_dict({k[5:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("val0_")})
    from uhc_amd.khrylib.rl.agents import AgentPPO
    opt_p = torch.optim.Adam([p for p in pol.parameters() if p.requires_grad], lr=float(g["policy_lr"]))
    opt_v = torch.optim.Adam(val.parameters(), lr=float(g["value_lr"]))
    ag = AgentPPO(env=None, policy_net=pol, value_net=val, dtype=torch.float64, device=torch.device("cpu"), gamma=0.95, data_loader=None,
                  optimizer_policy=opt_p, optimizer_value=opt_v, opt_num_epochs=int(g["epochs"]), value_opt_niter=1, clip_epsilon=float(g["clip_epsilon"]),
                  mini_batch_size=int(g["mini_batch_size"]), use_mini_batch=branch == "mini", policy_grad_clip=[(list(pol.parameters()), float(g["grad_clip"]))])
    np.random.seed(int(g["np_seed"]))
    ag.update_policy(*(torch.from_numpy(g[k]).clone() for k in ("x", "a", "rets", "advs", "exps")))
    moved = 0.0
    for n, p in pol.named_parameters():
        np.testing.assert_allclose(p.detach().numpy(), g[f"pol_{branch}_" + n], atol=1e-12, err_msg=n)
        moved = max(moved, float(np.abs(g[f"pol_{branch}_" + n] - g["pol0_" + n]).max()))
    for n, p in val.named_parameters():
        np.testing.assert_allclose(p.detach().numpy(), g[f"val_{branch}_" + n], atol=1e-12, err_msg=n)
    assert moved > 1e-3  # (the update did something)
    assert len(ag.last_losses) == (3 if branch == "full" else 3 * (70 // 16))


def test_policy_mcp_matches_reference():
    from uhc_amd.models.policy_mcp import PolicyMCP

    class C(dict):
        __getattr__ = dict.__getitem__

    g = np.load(os.path.join(G, "g12_policy_mcp.npz"))
    cfg = C(policy_hsize=[24, 16, 12], policy_htype="gelu", fix_std=True, log_std=-2.3, num_primitive=4, composer_dim=[20, 10])
    pol = PolicyMCP(cfg, action_dim=7, state_dim=19)
    sd = {k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("pol_")}
    assert set(sd) == set(pol.state_dict())  # the reference's parameter names: released checkpoints load unchanged
    pol.load_state_dict(sd)
    x, a = torch.from_numpy(g["x"]), torch.from_numpy(g["a"])
    np.testing.assert_allclose(pol.composer(x).detach().numpy(), g["weight"], atol=1e-14)
    np.testing.assert_allclose(pol(x).loc.detach().numpy(), g["mean"], atol=1e-14)
    lp = pol.get_log_prob(x, a)
    np.testing.assert_allclose(lp.detach().numpy(), g["log_prob"], atol=1e-11)
    (-lp.mean()).backward()
    for n, p in pol.named_parameters():
        if p.grad is not None:
            np.testing.assert_allclose(p.grad.numpy(), g["polgrad_" + n], atol=1e-11, err_msg=n)
    assert pol.select_action(x, True).shape == (32, 7)


def test_zfilter_sequential_and_batched():
    from uhc_amd.khrylib.utils.zfilter import ZFilter
    g = np.load(os.path.join(G, "g7_zfilter.npz"))
    z = ZFilter((12,), clip=5)
    ys = np.array([z(x) for x in g["xs"]])
    np.testing.assert_allclose(ys, g["ys"], atol=1e-14)
    np.testing.assert_allclose(z(g["xs"][0], update=False), g["y_noupdate"], atol=1e-14)
    zb = ZFilter((12,), clip=5)
    zb(torch.from_numpy(g["xs"][:70]))
    zb(torch.from_numpy(g["xs"][70:]))
    assert zb.rs.n == int(g["n"])
    np.testing.assert_allclose(zb.rs.mean, g["mean"], atol=1e-13)
    np.testing.assert_allclose(zb.rs.std, g["std"], atol=1e-13)
    z1 = ZFilter((3,))
    z1(np.array([1.0, -2.0, 3.0]))
    np.testing.assert_allclose(z1.rs.var, [1.0, 4.0, 9.0])  # n == 1: var = mean^2 (zfilter.py:34-35)


def test_zfilter_weighted_batch_push_equals_row_pushes():
    """push_batch(weights=done) counts exactly the flagged rows (the last observation of a finished episode), with no host-side count."""
    import torch
    from uhc_amd.khrylib.utils.zfilter import ZFilter
    rng = np.random.default_rng(3)
    x = rng.normal(size=(40, 5)) * 3 + 1
    w = rng.integers(0, 2, size=40)
    a, b = ZFilter((5,), clip=5), ZFilter((5,), clip=5)
    a(torch.from_numpy(x[:10]))
    for r in x[:10]:
        b(r)
    a.rs.push_batch(torch.from_numpy(x[10:]), weights=torch.from_numpy(w[10:]))
    a.rs.push_batch(torch.from_numpy(x[10:]), weights=torch.zeros(30))  # nothing flagged: a no-op
    for r, wi in zip(x[10:], w[10:]):
        if wi:
            b(r)
    assert a.rs.n == b.rs.n == 10 + int(w[10:].sum())
    np.testing.assert_allclose(a.rs.mean, b.rs.mean, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(a.rs.var, b.rs.var, rtol=1e-10)


def test_zfilter_out_and_step_counter_host_path():
    """ZFilter(x, out=, step_counter=) -- the rollout's last op of a step -- on host tensors (the framework path): same numbers as the
    plain call, written in place, counter advanced by one."""
    import torch
    from uhc_amd.khrylib.utils.zfilter import ZFilter
    rng = np.random.default_rng(8)
    x = torch.from_numpy(rng.normal(size=(12, 7)) * 2 - 1)
    a, b = ZFilter((7,), clip=5), ZFilter((7,), clip=5)
    want = a(x)
    out, t = torch.zeros_like(x), torch.zeros(1, dtype=torch.long)
    got = b(x, out=out, step_counter=t)
    assert got is out and torch.equal(out, want) and int(t) == 1 and a.rs.n == b.rs.n == 12
    assert torch.equal(b(x, update=False, out=out, step_counter=t), a(x, update=False)) and int(t) == 2 and b.rs.n == 12


def test_dataset_sampling_interface():
    import random
    from uhc_amd.data_loaders.dataset_amass_single import DatasetAMASSSingle
    from uhc_amd.data_loaders.synthetic import make_synthetic_amass
    data = make_synthetic_amass(6, seed=3, t_range=(20, 60))
    data["too_short"] = {k: (v[:4] if isinstance(v, np.ndarray) and v.ndim == 2 else v) for k, v in data["0-synth_0000"].items()}
    specs = dict(file_path="synthetic", t_min=5, t_max=30, mode="all")
    dl = DatasetAMASSSingle(specs, "train", pickle_data=data)
    assert dl.get_len() == 6 and "too_short" not in dl.data_keys
    assert len(dl.sample_keys) == sum(data[k]["pose_aa"].shape[0] // 30 + 1 for k in dl.data_keys)
    random.seed(5)
    np.random.seed(5)
    s = dl.sample_seq()
    T = data[dl.curr_key]["pose_aa"].shape[0]
    assert s["seq_name"] == dl.curr_key and 0 <= dl.fr_start < T - 5 and dl.fr_end == min(dl.fr_start + 30, T)
    assert s["pose_aa"].shape == (dl.fr_end - dl.fr_start, 72) and s["beta"].shape[1] == 16 and s["gender"][0] == 0
    assert s["has_obj"] is False and s["num_obj"] == 0
    # same seeds -> same windows (the loader draws from the global generators like the reference)
    random.seed(5)
    np.random.seed(5)
    s2 = dl.sample_seq()
    assert s2["seq_name"] == s["seq_name"] and np.array_equal(s2["pose_aa"], s["pose_aa"])
    # success-weighted sampling: clips that always succeed are picked less often
    fd = {k: [[1.0, 0]] * 10 for k in dl.data_keys}
    fd[dl.data_keys[0]] = [[0.2, 0]] * 10
    np.random.seed(0)
    picks = [dl.sample_seq(freq_dict=fd, sampling_temp=0.1, sampling_freq=1.0)["seq_name"] for _ in range(200)]
    assert picks.count(dl.data_keys[0]) > 150
    full = dl.iter_seq()
    assert full["pose_aa"].shape[0] == data[dl.data_keys[0]]["pose_aa"].shape[0]


def test_config_defaults_and_schedules(tmp_path):
    from uhc_amd.utils.config_utils.copycat_config import Config
    c = Config(cfg_id="copycat_mi355x", base_dir=str(tmp_path))
    assert (c.gamma, c.tau, c.clip_epsilon, c.num_optim_epoch, c.min_batch_size) == (0.95, 0.95, 0.2, 10, 50000)
    assert c.policy_hsize == [2048, 1024, 512] and c.policy_htype == "gelu" and c.fix_std and c.log_std == -2.3
    assert c.obs_v == 2 and c.meta_pd and c.residual_force and c.residual_force_scale == 100 and c.residual_force_lim == 100.0
    assert c.reward_id == "world_rfc_implicit" and c.reward_weights["k_e"] == 5.0
    assert os.path.isdir(c.model_dir) and c.model_dir.endswith("results/motion_im/copycat_mi355x/models")
    c2 = Config(cfg_id="x", base_dir=str(tmp_path), cfg_dict=dict(adp_iter_cp=[0, 100, 200], adp_noise_rate_cp=[1.0, 0.5], adp_policy_lr_cp=[1e-4, 5e-5, 1e-5]))
    c2.update_adaptive_params(50)
    assert c2.adp_noise_rate == pytest.approx(0.75) and c2.adp_policy_lr == pytest.approx(7.5e-5)
    c2.update_adaptive_params(150)
    assert c2.adp_noise_rate == pytest.approx(0.5) and c2.adp_policy_lr == pytest.approx(3e-5)
    c2.update_adaptive_params(500)
    assert c2.adp_policy_lr == pytest.approx(1e-5) and c2.adp_log_std == pytest.approx(-2.3)


# ---- data-parallel learner: 2 gloo ranks with half the batch each == 1 process with the whole batch ----------
def _update(rank, world, port, out_path, wire=None, fused=True, epochs=3, overlap=False):
    import torch.distributed as dist
    torch.set_default_dtype(torch.float64)
    sys.path.insert(0, ROOT)
    from uhc_amd.khrylib.rl.agents import AgentPPO, RolloutBatch
    if world > 1:
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    torch.manual_seed(3)
    pol, val = _nets()
    opt_p = torch.optim.Adam(pol.parameters(), lr=5e-3)
    opt_v = torch.optim.Adam(val.parameters(), lr=3e-3)
    ag = AgentPPO(env=None, policy_net=pol, value_net=val, dtype=torch.float64, device=torch.device("cpu"), gamma=0.95, data_loader=None,
                  tau=0.95, optimizer_policy=opt_p, optimizer_value=opt_v, opt_num_epochs=epochs, clip_epsilon=0.2,
                  policy_grad_clip=[(pol.parameters(), 0.5)])
    ag.grad_wire_dtype, ag.fuse_grad_exchange, ag.overlap_grad_exchange = wire, fused, overlap
    calls = []
    if world > 1:  # count the collectives of the update: one flat gradient exchange per optimisation epoch + the advantage statistics
        real = dist.all_reduce
        dist.all_reduce = lambda t, *a, **k: (calls.append(int(t.numel())), real(t, *a, **k))[1]
    rng = np.random.default_rng(0)
    n_env, T = 8, 12
    st = torch.from_numpy(rng.normal(size=(n_env, T, 23)))
    ac = torch.from_numpy(rng.normal(scale=0.2, size=(n_env, T, 9)))
    rw = torch.from_numpy(rng.uniform(size=(n_env, T)))
    mk = torch.from_numpy((rng.uniform(size=(n_env, T)) > 0.1).astype(np.float64))
    mk[:, -1] = 0
    ex = torch.from_numpy((rng.uniform(size=(n_env, T)) > 0.2).astype(np.float64))
    lo, hi = (0, n_env) if world == 1 else ((0, 3) if rank == 0 else (3, n_env))  # uneven shards on purpose
    sl = slice(lo, hi)
    n = (hi - lo) * T
    batch = RolloutBatch(st[sl].reshape(n, -1), ac[sl].reshape(n, -1), rw[sl].reshape(n, 1), mk[sl].reshape(n, 1), ex[sl].reshape(n), T)
    ag.update_params(batch)
    if rank == 0:
        torch.save({"pol": pol.state_dict(), "val": val.state_dict(), "calls": calls}, out_path)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_two_rank_gloo_update_equals_single_process(tmp_path):
    import torch.multiprocessing as mp
    single, multi = str(tmp_path / "single.pt"), str(tmp_path / "multi.pt")
    _update(0, 1, 0, single)
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_update, args=(2, port, multi), nprocs=2, join=True)
    a, b = torch.load(single), torch.load(multi)
    for net in ("pol", "val"):
        for k in a[net]:
            np.testing.assert_allclose(a[net][k].numpy(), b[net][k].numpy(), atol=1e-10, err_msg=f"{net}.{k}")
    # ONE gradient exchange per optimisation epoch -- policy and value gradients in one flat buffer (+ their two sample counts)
    npar = sum(v.numel() for k, v in a["pol"].items() if k != "action_log_std") + sum(v.numel() for v in a["val"].values())
    big = [c for c in b["calls"] if c > 100]
    assert big == [npar + 2] * 3, (big, npar)


def test_two_rank_gloo_float32_wire_and_separate_exchanges(tmp_path):
    """`grad_allreduce_dtype: float32` (half the bytes per exchange): the 2-rank update stays within float32 rounding of the float64 one;
    two exchanges per epoch (value, then policy: the reference's order of the two steps) give the same update as the fused exchange."""
    import torch.multiprocessing as mp
    single, w32, sep = str(tmp_path / "single.pt"), str(tmp_path / "w32.pt"), str(tmp_path / "sep.pt")
    _update(0, 1, 0, single)
    port = 27500 + (os.getpid() % 2000)
    mp.spawn(_update, args=(2, port, w32, torch.float32, True), nprocs=2, join=True)
    mp.spawn(_update, args=(2, port + 1, sep, None, False), nprocs=2, join=True)
    a, b, c = torch.load(single), torch.load(w32), torch.load(sep)
    for net in ("pol", "val"):
        for k in a[net]:
            np.testing.assert_allclose(a[net][k].numpy(), c[net][k].numpy(), atol=1e-10, err_msg=f"{net}.{k}")
            np.testing.assert_allclose(a[net][k].numpy(), b[net][k].numpy(), atol=2e-5, rtol=1e-4, err_msg=f"float32 wire {net}.{k}")
    assert len([x for x in c["calls"] if x > 100]) == 6  # value and policy apart: two exchanges per epoch


def test_float32_wire_is_the_default_and_its_drift_over_a_full_update_is_bounded(tmp_path):
    """`grad_allreduce_dtype` defaults to float32 (SURVEY 8e's 32 MB per exchange).  What that costs: over a full 10-epoch
    update (the release configs' num_optim_epoch) on 2 gloo ranks with uneven shards, with learning rates 100 x the release ones, the weights stay
    within 1e-4 relative / 5e-5 absolute of the float64 single-process update; `grad_allreduce_dtype: float64` keeps the 1e-10 equality."""
    import torch.multiprocessing as mp
    from uhc_amd.utils.config_utils.copycat_config import Config
    assert Config(cfg_id="copycat_mi355x", base_dir=str(tmp_path)).grad_allreduce_dtype == "float32"
    assert Config(cfg_id="x", base_dir=str(tmp_path), cfg_dict=dict(grad_allreduce_dtype="float64")).grad_allreduce_dtype == "float64"
    single, w32 = str(tmp_path / "single10.pt"), str(tmp_path / "w32_10.pt")
    _update(0, 1, 0, single, None, True, 10)
    port = 25500 + (os.getpid() % 2000)
    mp.spawn(_update, args=(2, port, w32, torch.float32, True, 10), nprocs=2, join=True)
    a, b = torch.load(single), torch.load(w32)
    worst = 0.0
    for net in ("pol", "val"):
        for k in a[net]:
            np.testing.assert_allclose(a[net][k].numpy(), b[net][k].numpy(), atol=5e-5, rtol=1e-4, err_msg=f"float32 wire, 10 epochs: {net}.{k}")
            worst = max(worst, float((a[net][k] - b[net][k]).abs().max()))
    assert 0.0 < worst  # (the float32 wire was really used)
    assert len([c for c in b["calls"] if c > 100]) == 10


def test_overlapped_exchange_is_the_same_update(tmp_path):
    """the value gradient's half of the exchange is started (async) before the surrogate's backward pass and waited for
    after it -- two collectives of half the size per epoch, the same sums: the 2-rank update is bit-identical to the single-buffer one."""
    import torch.multiprocessing as mp
    from uhc_amd.utils.config_utils.copycat_config import Config
    assert Config(cfg_id="copycat_mi355x", base_dir=str(tmp_path)).overlap_grad_exchange is True
    one, two = str(tmp_path / "one.pt"), str(tmp_path / "two.pt")
    port = 23500 + (os.getpid() % 2000)
    mp.spawn(_update, args=(2, port, one, torch.float32, True, 3, False), nprocs=2, join=True)
    mp.spawn(_update, args=(2, port + 1, two, torch.float32, True, 3, True), nprocs=2, join=True)
    a, b = torch.load(one), torch.load(two)
    for net in ("pol", "val"):
        for k in a[net]:
            np.testing.assert_array_equal(a[net][k].numpy(), b[net][k].numpy(), err_msg=f"{net}.{k}")
    nval = sum(v.numel() for v in a["val"].values())
    npol = sum(v.numel() for k, v in a["pol"].items() if k != "action_log_std")
    assert [c for c in b["calls"] if c > 100] == [nval + 1, npol + 1] * 3


def test_process_amass_raw_collects_action_files(tmp_path):
    """uhc_amd/data_process/process_amass_raw.py (reference :83-131): <data set>/<subject>/<action>.npz -> "<data set>_<subject>_<action>",
    shape.npz and other files skipped, arrays unchanged (checked identical to the imported reference on the same tree when written)."""
    from uhc_amd.data_process.process_amass_raw import ALL_SEQUENCES, read_data
    rng = np.random.default_rng(0)
    want = {}
    for ds, subs in (("CMU", {"01": ["01_01_poses.npz", "shape.npz", "notes.txt"], "02": ["02_03_poses.npz"]}), ("KIT", {"3": ["walk_poses.npz", "shape.npz"]})):
        for sub, files in subs.items():
            os.makedirs(tmp_path / ds / sub)
            for f in files:
                if f.endswith(".npz"):
                    arrs = dict(poses=rng.normal(size=(5, 156)), trans=rng.normal(size=(5, 3)), betas=rng.normal(size=16), mocap_framerate=np.float64(120.0))
                    np.savez(tmp_path / ds / sub / f, **arrs)
                    if f != "shape.npz":
                        want[f"{ds}_{sub}_{f[:-4]}"] = arrs
                else:
                    (tmp_path / ds / sub / f).write_text("x")
    db = read_data(str(tmp_path), ["CMU", "KIT"], log=lambda *a: None)
    assert sorted(db) == sorted(want) == ["CMU_01_01_01_poses", "CMU_02_02_03_poses", "KIT_3_walk_poses"]
    for k, arrs in want.items():
        for f, v in arrs.items():
            np.testing.assert_array_equal(db[k][f], v)
    assert len(ALL_SEQUENCES) == 19 and "DanceDB" in ALL_SEQUENCES


# ---- data-parallel bookkeeping: the clip-success history and the end-reward mean are the same on every rank -----------
def _freq_ranks(rank, world, port, out_dir):
    import pickle
    import types
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from uhc_amd.agents.agent_copycat import AgentCopycat
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    ag = types.SimpleNamespace(freq_dict={"a": [[1.0, 0]], "b": []}, max_freq=4, device=torch.device("cpu"))  # shared history: one row of "a"
    for name in ("_freq_add", "sync_freq_dict", "_global_mean"):
        setattr(ag, name, types.MethodType(getattr(AgentCopycat, name), ag))
    ag._freq_add("a", [[0.5 + rank, 10 * rank]])          # each rank's own rollout results
    if rank == 0:
        ag._freq_add("b", [[True, 0]] * 3)                 # rank 0's evaluation entries
    ag.sync_freq_dict()
    first = {k: list(v) for k, v in ag.freq_dict.items()}
    ag._freq_add("a", [[0.25, 7 + rank]])
    ag.sync_freq_dict()                                    # a second merge must not duplicate what is already common
    mean = ag._global_mean(1.0 + rank, 10 * (rank + 1))    # weights 10, 20 -> (10 + 40) / 30
    pickle.dump((first, ag.freq_dict, mean), open(os.path.join(out_dir, f"r{rank}.p"), "wb"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_freq_dict_merge_and_reward_mean(tmp_path):
    """only rank 0 evaluates and every rank appended its own rollout results, so the sampling distributions
    drifted apart.  After sync_freq_dict every rank holds the same history: common base + every rank's new rows in rank order,
    truncated to max_freq as the reference truncates (agent_copycat.py:598-604)."""
    import pickle
    import torch.multiprocessing as mp
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_freq_ranks, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (pickle.load(open(tmp_path / f"r{r}.p", "rb")) for r in (0, 1))
    assert r0[0] == r1[0] and r0[1] == r1[1]
    assert r0[0]["a"] == [[1.0, 0], [0.5, 0], [1.5, 10]] and r0[0]["b"] == [[True, 0]] * 3
    assert r0[1]["a"] == [[0.5, 0], [1.5, 10], [0.25, 7], [0.25, 8]]  # last max_freq = 4 rows
    assert r0[2] == pytest.approx(50.0 / 30.0) and r1[2] == pytest.approx(50.0 / 30.0)
