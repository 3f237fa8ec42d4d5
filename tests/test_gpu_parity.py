"""GPU parity tests: the HIP path (through the C ABI, via the simgan_amd shim) against
(a) the fixtures captured from the reference (tests/golden) and (b) the CPU oracle on seeded
inputs.  Tolerance: BASELINE.json north_star -- 1e-4 relative fp32 (helpers.RTOL/ATOL)."""
import numpy as np
import pytest

from helpers import assert_close, load

pytestmark = pytest.mark.gpu


class Box:  # duck-typed gym.spaces.Box (a2c/model.py:55-57 reads __class__.__name__ and .shape)
    def __init__(self, shape):
        self.shape = tuple(shape)


@pytest.fixture(scope="module")
def sg():
    import simgan_amd
    return simgan_amd


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle
    return oracle


def make_policy(sg, m):
    if m["kind"] == "mlp":
        return sg.Policy((m["O"],), Box((m["A"],)), base_kwargs={"recurrent": False, "hidden_size": m["H"]})
    return sg.SplitPolicy((m["O"],), Box((m["A"],)), base_kwargs={"hidden_size": m["H"], "num_feet": m["num_feet"]})


# ------------------------------------------------------------------ tile engine
@pytest.mark.parametrize("mode,M,N,K", [(0, 16, 16, 16), (0, 32, 112, 96), (0, 64, 64, 48), (1, 16, 96, 112),
                                        (1, 32, 112, 112), (1, 64, 64, 16), (2, 112, 96, 16), (2, 64, 48, 64),
                                        (2, 16, 112, 32), (2, 112, 112, 32)])
def test_gemm_engine(sg, mode, M, N, K):
    """NT/NN/TN LDS-tile MFMA GEMMs vs numpy float64, with asymmetric operands (catches transposes)."""
    import ctypes as C
    from simgan_amd import _lib
    ctx = _lib.Context.default()
    rng = np.random.default_rng(mode * 1000 + M + N + K)
    if mode == 0:
        A, B = rng.standard_normal((M, K)), rng.standard_normal((N, K)); ref = A @ B.T
    elif mode == 1:
        A, B = rng.standard_normal((M, K)), rng.standard_normal((K, N)); ref = A @ B
    else:
        A, B = rng.standard_normal((K, M)), rng.standard_normal((K, N)); ref = 2.0 * (A.T @ B)
    A, B = A.astype(np.float32), B.astype(np.float32)
    Cm = np.zeros((M, N), np.float32)
    _lib.check(ctx.lib.sg_test_gemm(ctx.h, mode, M, N, K, _lib.fptr(A), _lib.fptr(B), _lib.fptr(Cm)))
    assert_close(Cm, ref, rtol=1e-5, atol=1e-4, what=f"gemm mode {mode}")


# ----------------------------------------------------------------------- policy
POLICY_CASES = ["policy_mlp_tiny", "policy_mlp_northstar", "policy_mlp_hopper",
                "policy_split_hopper", "policy_split_laikago", "policy_split_tiny"]


@pytest.mark.parametrize("name", POLICY_CASES)
def test_policy_golden(sg, name):
    g = load(name)
    p = make_policy(sg, g["meta"])
    assert p.num_params == g["params"].size
    p.set_flat_params(g["params"])
    assert np.array_equal(p.get_flat_params(), g["params"])  # pad/unpad round trip is exact
    n = g["obs"].shape[0]
    v, a, lp, _ = p.act(g["obs"], None, None, noise=g["noise"])
    assert_close(v, g["act_value"], what="act value")
    assert_close(a, g["act_action"], what="act action")
    assert_close(lp, g["act_logp"], what="act logp")
    v, a, lp, _ = p.act(g["obs"], None, None, deterministic=True)
    assert_close(a, g["det_action"], what="det action")
    assert_close(lp, g["det_logp"], what="det logp")
    assert_close(p.get_value(g["obs"], None, None), g["get_value"], what="get_value")
    v, lp, ent, _ = p.evaluate_actions(g["obs"], None, None, g["eval_action"])
    assert_close(v, g["eval_value"], what="eval value")
    assert_close(lp, g["eval_logp"], what="eval logp")
    assert_close(float(ent), g["eval_entropy"], what="entropy")
    assert tuple(v.shape) == (n, 1) and tuple(lp.shape) == (n, 1)


def test_policy_library_rng(sg):
    """Without injected noise the library's generator samples: actions differ call to call,
    log-probs stay consistent with evaluate_actions."""
    g = load("policy_mlp_northstar")
    p = make_policy(sg, g["meta"])
    p.set_flat_params(g["params"])
    v1, a1, lp1, _ = p.act(g["obs"], None, None)
    v2, a2, lp2, _ = p.act(g["obs"], None, None)
    assert not np.allclose(a1.numpy(), a2.numpy())
    _, lp_e, _, _ = p.evaluate_actions(g["obs"], None, None, a1)
    assert_close(lp_e, lp1, what="logp(sampled action)")
    z = (a1.numpy() - g["mean"]) / g["std"]
    assert abs(z.mean()) < 0.3 and 0.7 < z.std() < 1.3


# -------------------------------------------------------------------------- GAE
@pytest.mark.parametrize("use_gae", [1, 0])
@pytest.mark.parametrize("proper", [1, 0])
def test_compute_returns_golden(sg, use_gae, proper):
    g = load("gae")
    T, N = g["rewards"].shape[:2]
    ro = sg.RolloutStorage(T, N, (3,), Box((2,)), 1, 0)
    ro.rewards.copy_(ro.rewards.new_tensor(g["rewards"]))
    ro.value_preds.copy_(ro.value_preds.new_tensor(g["value_preds"]))
    ro.masks.copy_(ro.masks.new_tensor(g["masks"]))
    ro.bad_masks.copy_(ro.bad_masks.new_tensor(g["bad_masks"]))
    ro.compute_returns(g["next_value"], bool(use_gae), 0.99, 0.95, bool(proper))
    upto = T if use_gae else T + 1
    assert_close(ro.returns.numpy()[:upto], g[f"returns_gae{use_gae}_proper{proper}"][:upto], rtol=1e-5, what="returns")
    assert_close(ro.value_preds.numpy(), g[f"value_preds_gae{use_gae}_proper{proper}"], what="value_preds")


# -------------------------------------------------------------------------- PPO
PPO_CASES = ["ppo_mlp_tiny", "ppo_mlp_northstar", "ppo_mlp_onestep", "ppo_split_hopper", "ppo_split_laikago"]


def fill_rollout(ro, g):
    for name in ("obs", "obs_feat", "actions", "rewards", "value_preds", "returns", "action_log_probs", "masks", "bad_masks"):
        if name in g and getattr(ro, name).numel():
            getattr(ro, name).copy_(getattr(ro, name).new_tensor(g[name]))


def _ppo_problem(sg, g):
    m = g["meta"]
    p = make_policy(sg, m)
    p.set_flat_params(g["params0"])
    ro = sg.RolloutStorage(m["T"], m["N"], (m["O"],), Box((m["A"],)), 1, g["obs_feat"].shape[-1])
    fill_rollout(ro, g)
    agent = sg.algo.PPO(p, m["clip_param"], m["ppo_epoch"], m["num_mini_batch"], m["value_loss_coef"],
                        m["entropy_coef"], lr=m["lr"], eps=m["eps"], max_grad_norm=m["max_grad_norm"])
    return p, agent, ro


@pytest.mark.parametrize("name", PPO_CASES)
def test_ppo_update_golden(sg, name):
    g = load(name)
    m = g["meta"]
    p, agent, ro = _ppo_problem(sg, g)
    losses = agent.update(ro, perms=g["perms"])
    assert_close(ro.device_advantages(), g["advantages"], rtol=1e-5, what="advantages")
    assert_close(losses, g["losses"], what="ppo losses")
    mm, vv, step = agent.get_adam()
    assert step == m["ppo_epoch"] * m["num_mini_batch"]
    assert_close(mm, g["adam_m"], rtol=1e-3, atol=1e-7, what="adam m")
    assert_close(vv, g["adam_v"], rtol=1e-3, atol=1e-10, what="adam v")
    assert_close(p.get_flat_params(), g["params1"], what="params after update")


def test_ppo_mlp_separate_forward_kernel_matches_golden(sg, monkeypatch):
    """Policy normally runs the fused forward+backward kernel; the two-kernel path (what SplitPolicy uses) must give
    the same trajectory on it."""
    monkeypatch.setenv("SG_PPO_FUSED", "0")
    g = load("ppo_mlp_northstar")
    p, agent, ro = _ppo_problem(sg, g)
    losses = agent.update(ro, perms=g["perms"])
    assert_close(losses, g["losses"], what="ppo losses")
    assert_close(p.get_flat_params(), g["params1"], what="params after update")


@pytest.mark.parametrize("kind,O,A,H,f,T,N,M,E", [("mlp", 47, 12, 64, 1, 5, 7, 3, 2), ("mlp", 11, 3, 64, 1, 9, 13, 4, 2), ("mlp", 5, 2, 8, 1, 3, 1, 1, 3),
                                                   ("split", 14, 7, 100, 1, 7, 9, 5, 2), ("split", 64, 28, 100, 4, 3, 11, 2, 1)])
def test_ppo_ragged_minibatches_vs_oracle(sg, kind, O, A, H, f, T, N, M, E):
    """T*N not divisible by num_mini_batch (the sampler drops the remainder) and minibatches that are not a multiple of
    the 16-row groups: fused (Policy) and two-kernel (SplitPolicy) paths against the CPU oracle, two updates."""
    from oracle import oracle as orc
    rng = np.random.default_rng(O * 100 + T)
    bk = {"recurrent": False, "hidden_size": H} if kind == "mlp" else {"hidden_size": H, "num_feet": f}
    pol = (sg.Policy if kind == "mlp" else sg.SplitPolicy)((O,), Box((A,)), base_kwargs=bk, seed=T)
    ro = sg.RolloutStorage(T, N, (O,), Box((A,)), 1, 1)
    obs = rng.standard_normal((T + 1, N, O)).astype(np.float32)
    ro.obs.copy_(ro.obs.new_tensor(obs))
    v, a, lp, _ = pol.act(obs[:-1].reshape(-1, O), None, None, noise=rng.standard_normal((T * N, A)).astype(np.float32))
    npv = lambda x: x.numpy() if hasattr(x, "numpy") else np.asarray(x)  # noqa: E731
    act, logp = npv(a).reshape(T, N, A), npv(lp).reshape(T, N, 1)
    vp = np.concatenate([npv(v).reshape(T, N, 1), np.zeros((1, N, 1), np.float32)])
    ret = (vp + rng.standard_normal(vp.shape) * 0.5).astype(np.float32)
    ro.actions.copy_(ro.actions.new_tensor(act)); ro.action_log_probs.copy_(ro.action_log_probs.new_tensor(logp))
    ro.value_preds.copy_(ro.value_preds.new_tensor(vp)); ro.returns.copy_(ro.returns.new_tensor(ret))
    p0 = (pol.get_flat_params() + 0.02 * rng.standard_normal(pol.num_params)).astype(np.float32)   # leave ratio == 1
    pol.set_flat_params(p0)
    agent = sg.algo.PPO(pol, 0.2, E, M, 0.5, 0.01, lr=3e-4, eps=1e-5, max_grad_norm=0.5)
    d = orc.dims(orc.KIND_MLP if kind == "mlp" else orc.KIND_SPLIT, O, A, H, f)
    par, adam = p0.copy(), orc.AdamState(p0.size)
    cfg = orc.ppo_cfg(0.2, E, M, 0.5, 0.01, 3e-4, 1e-5, 0.5, True)
    for _ in range(2):
        perms = np.stack([rng.permutation(T * N) for _ in range(E)]).astype(np.int64)
        losses = agent.update(ro, perms=perms)
        olosses = orc.ppo_update(d, par, adam, cfg, obs, act, vp[..., 0], ret[..., 0], logp[..., 0], perms)
        assert_close(losses, olosses, what="PPO losses")
        assert_close(pol.get_flat_params(), par, what="policy params")


def test_ppo_graph_replay_is_bit_exact(sg, monkeypatch):
    """Two updates through the captured hipGraph equal two updates launched kernel by kernel."""
    g = load("ppo_mlp_northstar")
    m = g["meta"]

    def run():
        pol, agent, ro = _ppo_problem(sg, g)
        out = [agent.update(ro, perms=g["perms"]) for _ in range(2)]
        return out, pol.get_flat_params(), agent.get_adam()

    a = run()
    monkeypatch.setenv("SG_PPO_GRAPH", "0")
    b = run()
    assert a[0] == b[0] and np.array_equal(a[1], b[1])
    assert np.array_equal(a[2][0], b[2][0]) and np.array_equal(a[2][1], b[2][1]) and a[2][2] == b[2][2] == 2 * m["ppo_epoch"] * m["num_mini_batch"]


def test_ppo_lr_schedule_and_errors(sg):
    g = load("ppo_mlp_tiny")
    m = g["meta"]
    p = make_policy(sg, m)
    p.set_flat_params(g["params0"])
    ro = sg.RolloutStorage(m["T"], m["N"], (m["O"],), Box((m["A"],)), 1, 1)
    fill_rollout(ro, g)
    agent = sg.algo.PPO(p, 0.2, 1, 1, 0.5, 0.0, lr=3e-4, eps=1e-5, max_grad_norm=0.5)
    sg.update_linear_schedule(agent.optimizer, 1, 2, 3e-4)  # a2c/utils.py:68-72 -> lr = 1.5e-4
    assert abs(agent.optimizer.param_groups[0]['lr'] - 1.5e-4) < 1e-12
    p0 = p.get_flat_params()
    agent.update(ro, perms=g["perms"][:1])
    d_half = np.abs(p.get_flat_params() - p0).max()
    p.set_flat_params(g["params0"])
    agent2 = sg.algo.PPO(p, 0.2, 1, 1, 0.5, 0.0, lr=3e-4, eps=1e-5, max_grad_norm=0.5)
    agent2.update(ro, perms=g["perms"][:1])
    d_full = np.abs(p.get_flat_params() - p0).max()
    assert 0.45 < d_half / d_full < 0.55  # first Adam step is ~lr*sign(g)
    # a2c/storage.py:152-157: more minibatches than rows is an error
    too_many = sg.algo.PPO(p, 0.2, 1, m["T"] * m["N"] + 1, 0.5, 0.0, lr=3e-4, eps=1e-5, max_grad_norm=0.5)
    with pytest.raises(Exception, match="PPO requires"):
        too_many.update(ro)


# ---------------------------------------------------------------- discriminator
class Loader:  # stands in for torch DataLoader(TensorDataset(expert), batch_size=B, shuffle=True)
    def __init__(self, expert, batch_size):
        self.expert, self.batch_size = expert, batch_size


DISC_CASES = ["disc_tiny", "disc_northstar", "disc_hopper", "disc_single_batch"]


@pytest.mark.parametrize("chain", ["thin", "wide"])   # 4-row (v_mfma 4x4x1) and 16-row (16x16x4) chain kernels
@pytest.mark.parametrize("name", DISC_CASES)
def test_disc_update_golden(sg, name, chain, monkeypatch):
    monkeypatch.setenv("SG_DISC_CHAIN", chain)
    g = load(name)
    m = g["meta"]
    D = sg.algo.gail.Discriminator(m["F"], m["Hd"], None)
    D.set_flat_params(g["params0"])
    ro = sg.RolloutStorage(m["T"], m["N"], (3,), Box((2,)), 1, m["F"])
    ro.obs_feat.copy_(ro.obs_feat.new_tensor(g["obs_feat"]))
    loader = Loader(g["expert"], m["B"])
    for ep in range(m["epochs"]):
        losses = D.update_gail_dyn(loader, ro, expert_perm=g[f"expert_perm{ep}"], policy_perm=g[f"policy_perm{ep}"],
                                   alpha=g[f"alpha{ep}"])
        assert D.last_n_steps == int(g[f"n_steps{ep}"])
        assert_close(losses, g[f"losses{ep}"], what=f"disc losses ep{ep}")
        assert_close(D.get_flat_params(), g[f"params_after{ep}"], what=f"disc params ep{ep}")


class PairDataset:
    def __init__(self, s, a):
        self.tensors = (s, a)


class PairLoader:
    """Shape of the reference's DataLoader(TensorDataset(states, actions)) as Discriminator.update reads it."""
    def __init__(self, s, a, batch_size):
        self.dataset, self.batch_size = PairDataset(s, a), batch_size


@pytest.mark.parametrize("name", ["disc_classic_sa", "disc_classic_dyn"])
def test_disc_update_classic_golden(sg, name):
    """Discriminator.update (a2c/algo/gail.py:91-152), both row assemblies, against the reference's output."""
    g = load(name)
    m = g["meta"]
    in_dim = g["e_state"].shape[1] + g["e_action"].shape[1]
    D = sg.algo.gail.Discriminator(in_dim, m["Hd"], None)
    D.set_flat_params(g["params0"])
    ro = sg.RolloutStorage(m["T"], m["N"], (m["O"],), Box((m["A"],)), 1, m["F"])
    ro.obs.copy_(ro.obs.new_tensor(g["obs"]))
    ro.actions.copy_(ro.actions.new_tensor(g["actions"]))
    ro.obs_feat.copy_(ro.obs_feat.new_tensor(g["obs_feat"]))
    filt = None
    if m["use_filt"]:
        filt = lambda x, update=False: np.clip((x - g["filt_mean"]) / g["filt_std"], -5.0, 5.0)  # noqa: E731
    losses = D.update(PairLoader(g["e_state"], g["e_action"], m["B"]), ro, obsfilt=filt, is_gail_dyn=bool(m["dyn"]),
                      a_dim=m["a_dim"] or None, expert_perm=g["expert_perm"], policy_perm=g["policy_perm"], alpha=g["alpha"])
    assert D.last_n_steps == int(g["n_steps"])
    assert_close(losses, g["losses"], what="classic D losses")
    assert_close(D.get_flat_params(), g["params_after"], what="classic D params")


@pytest.mark.parametrize("name", ["disc_tiny", "disc_northstar"])
def test_disc_resume_and_graph_replay_are_bit_exact(sg, name, monkeypatch):
    """Checkpoint / resume (weights + Adam moments + step count into a fresh object) continues the trajectory
    bit for bit, and the hipGraph replay of an epoch equals launching the same kernels one by one."""
    g = load(name)
    m = g["meta"]
    ro = sg.RolloutStorage(m["T"], m["N"], (3,), Box((2,)), 1, m["F"])
    ro.obs_feat.copy_(ro.obs_feat.new_tensor(g["obs_feat"]))
    loader = Loader(g["expert"], m["B"])
    kw0 = dict(expert_perm=g["expert_perm0"], policy_perm=g["policy_perm0"], alpha=g["alpha0"])
    kw1 = dict(expert_perm=g["expert_perm1"], policy_perm=g["policy_perm1"], alpha=g["alpha1"])

    def fresh():
        D = sg.algo.gail.Discriminator(m["F"], m["Hd"], None)
        D.set_flat_params(g["params0"])
        return D

    A = fresh()
    A.update_gail_dyn(loader, ro, **kw0)
    mm, vv, step = A.get_adam()
    assert step == int(g["n_steps0"])
    Bd = fresh()
    Bd.set_flat_params(A.get_flat_params())
    Bd.set_adam(mm, vv, step)
    la = A.update_gail_dyn(loader, ro, **kw1)       # second epoch: graph replay on A
    lb = Bd.update_gail_dyn(loader, ro, **kw1)      # first epoch of a resumed object
    assert la == lb and np.array_equal(A.get_flat_params(), Bd.get_flat_params())
    assert_close(A.get_flat_params(), g["params_after1"], what="resumed trajectory vs reference")
    monkeypatch.setenv("SG_DISC_GRAPH", "0")
    Cd = fresh()
    Cd.update_gail_dyn(loader, ro, **kw0)
    lc = Cd.update_gail_dyn(loader, ro, **kw1)
    assert lc == la and np.array_equal(Cd.get_flat_params(), A.get_flat_params())


@pytest.mark.parametrize("chain", ["thin", "wide"])
@pytest.mark.parametrize("F,Hd,B,Ne,T,N", [(7, 16, 1, 5, 3, 2), (7, 16, 3, 11, 4, 5), (7, 16, 10, 35, 6, 7), (25, 100, 17, 60, 5, 11),
                                          (86, 100, 33, 70, 3, 40), (86, 100, 130, 300, 9, 30)])
def test_disc_ragged_batches_vs_oracle(sg, chain, F, Hd, B, Ne, T, N, monkeypatch):
    """Batch sizes that are not multiples of the 4-row / 16-row blocks, expert sets that do not divide into batches
    (drop_last), rollouts that do not either: both chain kernels against the CPU oracle."""
    from oracle import oracle as orc
    monkeypatch.setenv("SG_DISC_CHAIN", chain)
    rng = np.random.default_rng(F * 1000 + B)
    D = sg.algo.gail.Discriminator(F, Hd, None, seed=B)
    p0 = D.get_flat_params()
    ro = sg.RolloutStorage(T, N, (3,), Box((2,)), 1, F)
    feat = rng.standard_normal((T + 1, N, F)).astype(np.float32)
    ro.obs_feat.copy_(ro.obs_feat.new_tensor(feat))
    expert = (rng.standard_normal((Ne, F)) * 0.7 + 0.2).astype(np.float32)
    n_d = min(Ne // B, (T * N) // B)
    eperm = rng.permutation(Ne).astype(np.int64)
    pperm = rng.permutation(T * N).astype(np.int64)
    alpha = rng.random(n_d * B).astype(np.float32)
    par, adam = p0.copy(), orc.AdamState(p0.size)
    for _ in range(2):   # two epochs: the second one continues the Adam state (and replays the captured graph)
        losses = D.update_gail_dyn(Loader(expert, B), ro, expert_perm=eperm, policy_perm=pperm, alpha=alpha)
        assert D.last_n_steps == n_d
        olosses, on = orc.disc_update(F, Hd, par, adam, expert, feat, B, eperm, pperm, alpha)
        assert on == n_d
        assert_close(losses, olosses, what=f"D losses B={B}")
        assert_close(D.get_flat_params(), par, what=f"D params B={B}")


@pytest.mark.parametrize("chain", ["thin", "wide"])
def test_disc_saturated_activations_vs_oracle(sg, chain, monkeypatch):
    """Weights x4 and inputs x3: tanh units saturate, logits reach +-20 (sigmoid / log-sigmoid tails), some
    gradient-penalty rows have |g| far from 1 -- the step must still track the CPU oracle."""
    from oracle import oracle as orc
    monkeypatch.setenv("SG_DISC_CHAIN", chain)
    F, Hd, B, Ne, T, N = 86, 100, 128, 512, 8, 64
    rng = np.random.default_rng(17)
    D = sg.algo.gail.Discriminator(F, Hd, None, seed=2)
    p0 = (D.get_flat_params() * 4.0).astype(np.float32)
    D.set_flat_params(p0)
    ro = sg.RolloutStorage(T, N, (3,), Box((2,)), 1, F)
    feat = (3.0 * rng.standard_normal((T + 1, N, F))).astype(np.float32)
    ro.obs_feat.copy_(ro.obs_feat.new_tensor(feat))
    expert = (3.0 * rng.standard_normal((Ne, F)) + 1.0).astype(np.float32)
    eperm, pperm = rng.permutation(Ne).astype(np.int64), rng.permutation(T * N).astype(np.int64)
    alpha = rng.random(4 * B).astype(np.float32)
    losses = D.update_gail_dyn(Loader(expert, B), ro, expert_perm=eperm, policy_perm=pperm, alpha=alpha)
    par, adam = p0.copy(), orc.AdamState(p0.size)
    olosses, n_d = orc.disc_update(F, Hd, par, adam, expert, feat, B, eperm, pperm, alpha)
    assert n_d == 4 and D.last_n_steps == 4
    assert all(np.isfinite(x) for x in losses) and olosses[0] > 5.0      # the penalty term dominates: far from the init regime
    assert_close(losses, olosses, what="saturated D losses")
    assert_close(D.get_flat_params(), par, what="saturated D params")


def test_disc_short_expert_is_an_error(sg):
    """Ne < batch: the reference raises on the alpha*expert + (1-alpha)*policy size mismatch."""
    D = sg.algo.gail.Discriminator(7, 16, None)
    ro = sg.RolloutStorage(4, 8, (3,), Box((2,)), 1, 7)
    with pytest.raises(Exception, match="must match the size"):
        D.update_gail_dyn(Loader(np.zeros((5, 7), np.float32), 8), ro)


@pytest.mark.parametrize("name", ["ckpt_policy_mlp", "ckpt_policy_split"])
def test_checkpoint_interop_policy(sg, name, tmp_path):
    """A reference-written `[actor_critic, ob_rms]` file -> device policy -> the reference's own outputs; and back out
    to a file with the reference's layout (a2c/main.py:78-88,260-269)."""
    import os

    from helpers import GOLDEN
    from simgan_amd import checkpoint as ck
    g = load(name)
    pol, ob_rms = ck.load_policy(os.path.join(GOLDEN, name + ".pt"))
    assert type(pol).__name__ == ("Policy" if g["meta"]["kind"] == "mlp" else "SplitPolicy")
    v, a, lp, _ = pol.act(g["obs"], None, None, deterministic=True)
    assert_close(v, g["value"], what="value")
    assert_close(a, g["action"], what="action")
    assert_close(lp, g["logp"], what="log-prob")
    assert (ob_rms is None) == (name == "ckpt_policy_split")
    out = str(tmp_path / "saved.pt")
    ck.save_policy(out, pol, ob_rms)
    back = ck.read_reference_checkpoint(out)
    assert np.array_equal(np.concatenate([x.reshape(-1) for x in back["state_dict"].values()]), g["flat"])


def test_policy_ensemble_batched_inference(sg):
    """N rows, each drawing one of K policies (hopper_env_combined_policy.py:211-216): the grouped batched forwards
    equal row-by-row batch-1 `act` calls with the same sampling noise."""
    from simgan_amd.ensemble import PolicyEnsemble
    rng = np.random.default_rng(5)
    pols = [sg.SplitPolicy((14,), Box((7,)), base_kwargs={"hidden_size": 100, "num_feet": 1}, seed=70 + k) for k in range(5)]
    ens = PolicyEnsemble(pols)
    obs = rng.standard_normal((37, 14)).astype(np.float32)
    noise = rng.standard_normal((37, 7)).astype(np.float32)
    act, ind = ens.act(obs, noise=noise, rng=np.random.default_rng(9))
    assert act.shape == (37, 7) and set(np.unique(ind)) <= set(range(5)) and len(np.unique(ind)) > 1
    for r in range(37):
        _, a1, _, _ = pols[ind[r]].act(obs[r:r + 1], None, None, noise=noise[r:r + 1])
        assert_close(act[r], np.asarray(a1)[0], rtol=1e-6, what=f"row {r}")
    det, _ = ens.act(obs, ind=ind, deterministic=True)
    assert not np.allclose(det, act)


def test_checkpoint_interop_discriminator(sg):
    import os

    from helpers import GOLDEN
    from simgan_amd import checkpoint as ck
    g = load("ckpt_disc")
    D, ret_rms = ck.load_discriminator(os.path.join(GOLDEN, "ckpt_disc.pt"))
    assert np.array_equal(D.get_flat_params(), g["flat"])
    mm, vv, step = D.get_adam()
    assert np.array_equal(mm, g["adam_m"]) and np.array_equal(vv, g["adam_v"]) and step == int(g["step"])
    assert_close(D.returns.numpy() if hasattr(D.returns, "numpy") else D.returns, g["returns"], rtol=0, atol=0, what="returns")


@pytest.mark.parametrize("name", ["relabel_tiny", "relabel_northstar"])
def test_relabel_golden(sg, name):
    g = load(name)
    m = g["meta"]
    T, N, F = m["T"], m["N"], m["F"]
    # (1) per-step API, exactly the reference loop a2c/main_gail_dyn_ppo.py:275-292
    D = sg.algo.gail.Discriminator(F, m["Hd"], None)
    D.set_flat_params(g["params"])
    assert D.returns is None
    rms = sg.RunningMeanStd(shape=())
    for call in range(2):
        feat, masks, off = g[f"obs_feat{call}"], g[f"masks{call}"], float(g[f"offset{call}"])
        rewards = np.zeros((T, N, 1), np.float32)
        for step in range(T):
            rew, ret = D.predict_reward_combined(feat[step + 1], m["gamma"], masks[step], offset=off)
            if call == 0 and step == 0:
                assert_close(rew, g["raw_reward0"], what="raw reward")
            rms.update(ret.view(-1).numpy())
            rewards[step, :, 0] = np.clip(rew.view(-1).numpy() / np.sqrt(rms.var + 1e-7), -10.0, 10.0)
        assert_close(rewards, g[f"rewards{call}"], what="rewards (per-step API)")
        assert_close(D.returns, g[f"d_returns{call}"], what="D.returns")
        assert_close(rms.get_state(), g[f"rms{call}"], rtol=1e-5, what="ret_rms")
    # (2) fused on-device relabel
    D2 = sg.algo.gail.Discriminator(F, m["Hd"], None)
    D2.set_flat_params(g["params"])
    rms2 = sg.RunningMeanStd(shape=())
    for call in range(2):
        ro = sg.RolloutStorage(T, N, (3,), Box((2,)), 1, F)
        ro.obs_feat.copy_(ro.obs_feat.new_tensor(g[f"obs_feat{call}"]))
        ro.masks.copy_(ro.masks.new_tensor(g[f"masks{call}"]))
        D2.relabel_rewards(ro, m["gamma"], float(g[f"offset{call}"]), rms2)
        assert_close(ro.rewards.numpy(), g[f"rewards{call}"], what="rewards (fused)")
        assert_close(D2.returns, g[f"d_returns{call}"], what="D.returns (fused)")
        assert_close(rms2.get_state(), g[f"rms{call}"], rtol=1e-5, what="ret_rms (fused)")


# ------------------------------------------------------- full outer iterations
@pytest.mark.parametrize("name", ["iter_mlp", "iter_split"])
def test_full_iteration_golden(sg, name):
    """a2c/main_gail_dyn_ppo.py:209-304 through the drop-in classes, 2 outer iterations, with the
    reference's RNG artefacts injected; every intermediate the reference logs is compared."""
    g = load(name)
    m = g["meta"]
    T, N, F, B = m["T"], m["N"], m["F"], m["B"]
    p = make_policy(sg, m)
    p.set_flat_params(g["pi_params0"])
    D = sg.algo.gail.Discriminator(F, m["Hd"], None)
    D.set_flat_params(g["d_params0"])
    agent = sg.algo.PPO(p, 0.2, m["ppo_epoch"], m["num_mini_batch"], 0.5, 0.0, lr=3e-4, eps=1e-5, max_grad_norm=0.5)
    loader = Loader(g["expert"], B)
    ro = sg.RolloutStorage(T, N, (m["O"],), Box((m["A"],)), 1, F)
    ro.obs[0].copy_(ro.obs.new_tensor(g["obs0"]))
    rms = sg.RunningMeanStd(shape=())
    t = ro.obs.new_tensor
    for j in range(m["iters"]):
        for step in range(T):
            value, action, logp, hxs = p.act(ro.obs[step], ro.recurrent_hidden_states[step], ro.masks[step],
                                             noise=g[f"it{j}_noise"][step])
            ro.insert(t(g[f"it{j}_env_obs"][step]), hxs, action, logp, value, t(g[f"it{j}_env_reward"][step]),
                      t(g[f"it{j}_env_masks"][step]), t(g[f"it{j}_env_bad"][step]), t(g[f"it{j}_env_feat"][step]))
        assert_close(ro.actions.numpy(), g[f"it{j}_actions"], what="rollout actions")
        assert_close(ro.action_log_probs.numpy(), g[f"it{j}_action_log_probs"], what="rollout logp")
        next_value = p.get_value(ro.obs[-1], ro.recurrent_hidden_states[-1], ro.masks[-1])
        assert_close(next_value, g[f"it{j}_next_value"], what="next_value")
        for ep in range(m["gail_epoch"]):
            losses = D.update_gail_dyn(loader, ro, expert_perm=g[f"it{j}_d{ep}_expert_perm"],
                                       policy_perm=g[f"it{j}_d{ep}_policy_perm"], alpha=g[f"it{j}_d{ep}_alpha"])
            assert_close(losses, g[f"it{j}_d_losses"][ep], what="D losses")
        assert_close(D.get_flat_params(), g[f"it{j}_d_params"], what="D params")
        num_of_dones = (1.0 - ro.masks).sum().cpu().numpy() + N / 2
        num_of_expert_dones = (T * N) / m["gail_tar_length"]
        d_sa = 1 - num_of_dones / (num_of_dones + num_of_expert_dones)
        r_sa = np.log(d_sa) - np.log(1 - d_sa)
        assert_close(r_sa, g[f"it{j}_r_sa"], rtol=1e-6, what="r_sa")
        D.relabel_rewards(ro, m["gamma"], -r_sa, rms)
        assert_close(ro.rewards.numpy(), g[f"it{j}_rewards"], what="rewards")
        assert_close(rms.get_state(), g[f"it{j}_rms"], rtol=1e-5, what="ret_rms")
        ro.compute_returns(next_value, True, m["gamma"], m["gae_lambda"], True)
        assert_close(ro.returns.numpy()[:T], g[f"it{j}_returns"][:T], what="returns")
        losses = agent.update(ro, perms=g[f"it{j}_ppo_perms"])
        assert_close(losses, g[f"it{j}_ppo_losses"], what="ppo losses")
        assert_close(p.get_flat_params(), g[f"it{j}_pi_params"], what="pi params")
        ro.after_update()
