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
    _lib.check_test(_lib.load_test().sg_test_gemm(ctx.h, mode, M, N, K, _lib.fptr(A), _lib.fptr(B), _lib.fptr(Cm)))
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


@pytest.mark.parametrize("kind,O,A,H,f,n", [("mlp", 14, 7, 64, 1, 37), ("split", 14, 7, 100, 1, 301), ("split", 64, 28, 100, 4, 64),
                                           ("mlp", 111, 12, 64, 1, 2048)])
def test_policy_ensemble_one_launch_vs_oracle(sg, orc, kind, O, A, H, f, n):
    """sg_policy_act_ensemble: N rows, each drawing one of five saved policies per step
    (hopper_env_combined_policy.py:113-140,211-216), ONE device launch -- every row against the oracle's policy_act with
    the weights of the member it drew."""
    from simgan_amd.ensemble import PolicyEnsemble
    rng = np.random.default_rng(5 + n)
    mk = (lambda k: sg.Policy((O,), Box((A,)), base_kwargs={"recurrent": False, "hidden_size": H}, seed=70 + k)) if kind == "mlp" else \
         (lambda k: sg.SplitPolicy((O,), Box((A,)), base_kwargs={"hidden_size": H, "num_feet": f}, seed=70 + k))
    pols = [mk(k) for k in range(5)]
    for p_ in pols:   # widen the heads: the production init makes every member's mean nearly zero
        p_.set_flat_params(p_.get_flat_params() + 0.05 * rng.standard_normal(p_.num_params).astype(np.float32))
    ens = PolicyEnsemble(pols)
    obs = rng.standard_normal((n, O)).astype(np.float32)
    noise = rng.standard_normal((n, A)).astype(np.float32)
    ind = rng.integers(0, 5, size=n)
    ind[:3] = [4, 4, 0]
    value, act, logp, ind_out = ens.act(obs, ind=ind, noise=noise, full=True)
    assert np.array_equal(ind_out, ind)
    d = orc.dims(orc.KIND_MLP if kind == "mlp" else orc.KIND_SPLIT, O, A, H, f)
    flats = [p_.get_flat_params() for p_ in pols]
    for k in range(5):
        rows = np.nonzero(ind == k)[0]
        ov, oa, olp = orc.policy_act(d, flats[k], obs[rows], noise[rows])
        assert_close(act.numpy()[rows], oa, what=f"member {k} actions")
        assert_close(value.numpy()[rows], ov, what=f"member {k} values")
        assert_close(logp.numpy()[rows], olp, what=f"member {k} log-probs")
    det, _ = ens.act(obs, ind=ind, deterministic=True)
    for k in range(5):
        rows = np.nonzero(ind == k)[0]
        assert_close(det[rows], orc.policy_act(d, flats[k], obs[rows])[1], what=f"member {k} mode")
    # one member only / library noise: rows of one policy reproduce Policy.act's own launch bit for bit
    a1, _ = ens.act(obs[:20], ind=np.full(20, 2), noise=noise[:20])
    _, a2, _, _ = pols[2].act(obs[:20], None, None, noise=noise[:20])
    assert_close(a1, a2.numpy(), rtol=1e-6, what="ensemble vs Policy.act")
    z1, _ = ens.act(obs, ind=ind)
    z2, _ = ens.act(obs, ind=ind)
    assert np.isfinite(z1).all() and not np.array_equal(z1, z2)      # fresh draws per call


def test_ensemble_rejects_bad_members(sg):
    from simgan_amd.ensemble import PolicyEnsemble
    a = sg.Policy((5,), Box((2,)), base_kwargs={"hidden_size": 8})
    b = sg.Policy((6,), Box((2,)), base_kwargs={"hidden_size": 8})
    with pytest.raises(AssertionError):
        PolicyEnsemble([a, b])
    ens = PolicyEnsemble([a, a])
    with pytest.raises(AssertionError):
        ens.act(np.zeros((3, 5), np.float32), ind=[0, 1, 2])


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


# ------------------------------------------------------- the plain-PPO caller (a2c/main.py), BASELINE.json configs[4]
@pytest.mark.parametrize("fixture", ["iter_refine", "iter_refine_h100"])
def test_refine_iteration_golden(sg, fixture):
    """a2c/main.py:78-88,199-257 through driver.PpoLearner: warm start from a reference-written checkpoint, reset_critic,
    reset_variance, linear LR decay, two outer iterations at the Laikago refinement shape (obs 111, act 12, h64,
    num_mini_batch 8, clip 0.1, lr 1.5e-4) with the reference's noise / permutations injected -- and the same from a
    100-unit behaviour policy, where reset_critic leaves a 64-unit critic beside the 100-unit actor (a2c/model.py:80-87)."""
    import os

    from helpers import GOLDEN
    from simgan_amd.driver import PpoLearner
    g = load(fixture)
    m = g["meta"]
    T, N, O, A, H = m["T"], m["N"], m["O"], m["A"], m["H"]
    Hc = 64
    p = PpoLearner.warm_start(os.path.join(GOLDEN, fixture + "_warm.pt"), (O,), Box((A,)), warm_start_logstd=m["warm_start_logstd"])
    assert p.hidden_size == H and p.critic_hidden == Hc
    flat = p.get_flat_params()
    assert flat.size == g["pi_params0"].size
    na = H * O + H + H * H + H
    nc, nc_beh = Hc * O + Hc + Hc * Hc + Hc + Hc + 1, H * O + H + H * H + H + H + 1
    assert np.array_equal(flat[:na], g["behaviour_params"][:na])                       # actor kept
    assert np.array_equal(flat[na + nc:-A], g["behaviour_params"][na + nc_beh:-A])     # mean head kept
    assert np.all(flat[-A:] == np.float32(m["warm_start_logstd"]))                     # reset_variance
    W1c = flat[na:na + Hc * O].reshape(Hc, O)                                          # reset_critic: orthogonal, gain sqrt2
    assert_close(W1c @ W1c.T if Hc <= O else W1c.T @ W1c, 2.0 * np.eye(min(Hc, O)), rtol=0, atol=1e-5, what="critic.0 orthogonal init")
    assert np.all(flat[na + Hc * O:na + Hc * O + Hc] == 0)
    p.set_flat_params(g["pi_params0"])      # the critic draw itself is torch-RNG specific: continue from the reference's
    agent = sg.algo.PPO(p, m["clip_param"], m["ppo_epoch"], m["num_mini_batch"], 0.5, 0.0, lr=m["lr"], eps=1e-5, max_grad_norm=0.5)
    ro = sg.RolloutStorage(T, N, (O,), Box((A,)), 1, O)
    ro.obs[0].copy_(ro.obs.new_tensor(g["obs0"]))
    ro.obs_feat[0].copy_(ro.obs.new_tensor(g["obs0"]))
    learner = PpoLearner(p, agent, ro, gamma=m["gamma"], gae_lambda=m["gae_lambda"], use_linear_lr_decay=True, lr=m["lr"],
                         num_updates=m["num_updates"])

    class ScriptedEnvs:                     # plays back the transitions the reference saw
        def __init__(self, j):
            self.j, self.t = j, 0

        def step(self, action):
            t_, j = self.t, self.j
            self.t += 1
            done = g[f"it{j}_env_masks"][t_][:, 0] == 0
            infos = [({"bad_transition": True} if b == 0 else {}) for b in g[f"it{j}_env_bad"][t_][:, 0]]
            return ro.obs.new_tensor(g[f"it{j}_env_obs"][t_]), ro.obs.new_tensor(g[f"it{j}_env_reward"][t_]), done, infos

    for j in range(m["iters"]):
        noises = iter(g[f"it{j}_noise"])
        real_act = p.act
        p.act = lambda *a_, **k_: real_act(*a_, noise=next(noises), **k_)
        learner.collect(ScriptedEnvs(j))
        p.act = real_act
        assert_close(ro.actions.numpy(), g[f"it{j}_actions"], what="rollout actions")
        assert_close(ro.action_log_probs.numpy(), g[f"it{j}_action_log_probs"], what="rollout logp")
        assert_close(ro.value_preds.numpy()[:T], g[f"it{j}_value_preds_rollout"][:T], what="rollout values")
        assert np.array_equal(ro.obs_feat.numpy(), ro.obs.numpy())                      # identity feature copy
        perms = iter([g[f"it{j}_ppo_perms"]])
        real_update = agent.update
        agent.update = lambda r_: real_update(r_, perms=next(perms))
        out = learner.update()
        agent.update = real_update
        assert agent.optimizer.param_groups[0]["lr"] == pytest.approx(float(g["lrs"][j]), rel=1e-12)
        assert_close(ro.returns.numpy()[:T], g[f"it{j}_returns"][:T], what="returns")
        assert_close([out["value_loss"], out["action_loss"], out["dist_entropy"]], g[f"it{j}_ppo_losses"], what="ppo losses")
        assert_close(p.get_flat_params(), g[f"it{j}_pi_params"], what="pi params")
    # the reference's reset_critic builds a 64-unit critic whatever the actor is: so does the shim, on a fresh device handle
    q = sg.Policy((5,), Box((2,)), base_kwargs={"hidden_size": 32})
    n0 = q.num_params
    q.reset_critic((5,))
    assert q.critic_hidden == 64 and q.hidden_size == 32 and q.num_params == n0 - (32 * 5 + 32 + 32 * 32 + 32 + 32 + 1) + (64 * 5 + 64 + 64 * 64 + 64 + 64 + 1)
    import pickle
    q2 = pickle.loads(pickle.dumps(q))          # survives the package's own pickle with its critic width
    assert q2.critic_hidden == 64 and np.array_equal(q2.get_flat_params(), q.get_flat_params())
    # ... and is refused while an agent built on the policy holds the handle it would replace (the main's order is :85 then :149);
    # the action-noise stream goes on where it was
    q3 = sg.Policy((5,), Box((2,)), base_kwargs={"hidden_size": 32})
    seed3 = q3.seed
    agent3 = sg.algo.PPO(q3, 0.2, 1, 1, 0.5, 0.0, lr=3e-4, eps=1e-5, max_grad_norm=0.5)
    with pytest.raises(RuntimeError, match="hold its device handle"):
        q3.reset_critic((5,))
    del agent3
    import gc
    gc.collect()
    q3.reset_critic((5,))
    assert q3.critic_hidden == 64 and q3.seed == seed3


def test_collect_then_update_device_resident_equals_drop_in(sg):
    """GailDynLearner.collect() + update() twice: the device-resident fast path must train on the same rollout as the
    drop-in (upload-per-call) mode -- in particular slot 0 after after_update (the host mirrors roll over too)."""
    from simgan_amd.driver import GailDynLearner
    T, N, O, A, F = 6, 8, 14, 7, 25
    rng = np.random.default_rng(3)
    expert = rng.standard_normal((64, F)).astype(np.float32)
    script = [[(rng.standard_normal((N, O)).astype(np.float32), rng.standard_normal((N, 1)).astype(np.float32),
                rng.random(N) < 0.15, rng.standard_normal((N, F)).astype(np.float32)) for _ in range(T)] for _ in range(2)]
    noise = rng.standard_normal((2, T, N, A)).astype(np.float32)
    obs0 = rng.standard_normal((N, O)).astype(np.float32)

    class Envs:
        def __init__(self, j):
            self.j, self.t = j, 0

        def step(self, action):
            o, r, d_, f_ = script[self.j][self.t]
            self.t += 1
            return o, r, d_, [{"sas_feat": f_[i], **({"bad_transition": True} if (d_[i] and i % 2) else {})} for i in range(N)]

    results = []
    for resident in (False, True):
        pol = sg.SplitPolicy((O,), Box((A,)), base_kwargs={"hidden_size": 100, "num_feet": 1}, seed=1)
        disc = sg.algo.gail.Discriminator(F, 100, None, seed=2)
        agent = sg.algo.PPO(pol, 0.2, 2, 2, 0.5, 0.0, lr=3e-4, eps=1e-5, max_grad_norm=0.5)
        disc.seed, agent.seed = 11, 12          # same library draws in both modes
        ro = sg.RolloutStorage(T, N, (O,), Box((A,)), 1, F)
        ro.device_resident = resident
        ro.obs[0].copy_(ro.obs.new_tensor(obs0))
        if resident:
            ro.sync_to_device()
        learner = GailDynLearner(pol, agent, disc, ro, expert, gail_batch_size=16, gail_epoch=2, gail_tar_length=5.0)
        outs = []
        for j in range(2):
            it = iter(noise[j])
            real_act = pol.act
            pol.act = lambda *a_, **k_: real_act(*a_, noise=next(it), **k_)
            learner.collect(Envs(j), lambda infos: np.stack([i["sas_feat"] for i in infos]))
            pol.act = real_act
            outs.append(learner.update())
        results.append((outs, pol.get_flat_params(), disc.get_flat_params(), ro.obs.numpy()[0].copy(), ro.masks.numpy()[0].copy()))
    (o0, p0, d0, s0, m0), (o1, p1, d1, s1, m1) = results
    for a_, b_ in zip(o0, o1):
        # same kernels on the same data: bit-identical.  (The one host-computed scalar, r_sa, comes from numpy's float64 log in
        # drop-in mode and from the device's in resident mode: equal to the last bit or two.)
        a_, b_ = dict(a_), dict(b_)
        assert a_.keys() == b_.keys() and a_.pop("r_sa") == pytest.approx(b_.pop("r_sa"), rel=1e-14)
        assert a_ == b_, (a_, b_)
    assert np.array_equal(p0, p1) and np.array_equal(d0, d1) and np.array_equal(s0, s1) and np.array_equal(m0, m1)
    assert np.array_equal(s1, script[1][T - 1][0])


def test_predict_prob_single_step(sg, orc):
    """a2c/algo/gail.py:212-217"""
    rng = np.random.default_rng(8)
    F, Hd = 25, 100
    D = sg.algo.gail.Discriminator(F, Hd, None, seed=4)
    x = rng.standard_normal((9, F)).astype(np.float32) * 2.0
    par = D.get_flat_params()
    rew, _ = orc.disc_predict_reward(F, Hd, par, x, 0.99, np.ones(9, np.float32), 0.0)
    # reward = log(s + 1e-7) - log(1 - s + 1e-7)  =>  s = sigmoid(reward) up to the 1e-7 terms
    s_ref = 1.0 / (1.0 + np.exp(-rew[:, 0].astype(np.float64)))
    assert_close(D.predict_prob(x).numpy()[:, 0], s_ref, rtol=1e-4, atol=1e-6, what="sigmoid(D(x))")
    s0 = D.predict_prob_single_step(x[0, :11], x[0, 11:14], x[0, 14:])
    # the reference's own two-argument form on row batches: an [n, 1] tensor (a2c/algo/gail.py:212-217)
    two = D.predict_prob_single_step(x[:, :11], x[:, 11:])
    assert tuple(two.shape) == (x.shape[0], 1) and abs(float(two[0, 0]) - s0) < 1e-7
    assert isinstance(s0, float) and s0 == pytest.approx(float(s_ref[0]), rel=1e-4)


@pytest.mark.parametrize("tag", ["small", "northstar"])
def test_grad_pen_value_golden(sg, orc, tag):
    """Discriminator.compute_grad_pen_combined / compute_grad_pen (a2c/algo/gail.py:53-89): the penalty's value on the reference's
    rows with its torch.rand draw injected, against the reference's scalar; with the library's own draw, against the oracle on
    the alphas the library reports nothing about -- so only its range is checked there."""
    g = load("grad_pen")
    m = g["meta"][tag]
    F, n, sp = m["F"], m["n"], m["split"]
    D = sg.algo.gail.Discriminator(F, m["Hd"], None, seed=9)
    D.set_flat_params(g[f"{tag}_params"])
    e, p = g[f"{tag}_expert"], g[f"{tag}_policy"]
    v = D.compute_grad_pen_combined(e, p, 10.0, alpha=g[f"{tag}_alpha"])
    assert np.asarray(v).shape == () and np.asarray(v).dtype == np.float32
    assert_close(v, g[f"{tag}_value"], what="compute_grad_pen_combined")
    assert_close(v, orc.disc_grad_pen(F, m["Hd"], g[f"{tag}_params"], e, p, g[f"{tag}_alpha"], 10.0), what="... vs oracle")
    v2 = D.compute_grad_pen(e[:, :sp], e[:, sp:], p[:, :sp], p[:, sp:], lambda_=4.0, alpha=g[f"{tag}_alpha2"])
    assert_close(v2, g[f"{tag}_value2"], what="compute_grad_pen")
    # the call changes no state, and the library's own draw gives a value between the per-row extremes over alpha in [0, 1]
    assert np.array_equal(D.get_flat_params(), g[f"{tag}_params"])
    v3 = float(D.compute_grad_pen_combined(e, p, 10.0))
    assert np.isfinite(v3) and v3 >= 0.0 and v3 != float(v)
    with pytest.raises(Exception):
        D.compute_grad_pen_combined(e, p, 10.0, alpha=np.full(n, 1.5, np.float32))


def test_predict_reward_state_action_golden(sg):
    """Discriminator.predict_reward (a2c/algo/gail.py:195-199) against the reference's own output, two calls with
    Discriminator.returns carried between them."""
    g = load("predict_reward")
    m = g["meta"]
    D = sg.algo.gail.Discriminator(m["S"] + m["A"], m["Hd"], None)
    D.set_flat_params(g["params"])
    for c in range(2):
        rew, ret = D.predict_reward(g[f"state{c}"], g[f"action{c}"], m["gamma"], g[f"masks{c}"], offset=0.25 * c)
        assert tuple(rew.shape) == (m["n"], 1) and tuple(ret.shape) == (m["n"], 1)
        assert_close(rew.numpy(), g[f"reward{c}"], what=f"reward, call {c}")
        assert_close(ret.numpy(), g[f"returns{c}"], what=f"returns, call {c}")
        assert_close(D.returns.numpy(), g[f"returns{c}"], what="Discriminator.returns")


def test_feed_forward_generator_on_a_device_backed_rollout(sg):
    """The method form of simgan_amd.storage.feed_forward_batches (tests/test_host_logic.py checks the batches against the
    reference fixture): same tuples from a real RolloutStorage, permutation injected."""
    g = dict(np.load(__import__("os").path.join(__import__("helpers").GOLDEN, "ffgen.npz")))
    T, N = g["rewards"].shape[:2]
    ro = sg.RolloutStorage(T, N, (g["obs"].shape[-1],), Box((g["actions"].shape[-1],)), 1, g["obs_feat"].shape[-1])
    for name in ("obs", "obs_feat", "actions", "rewards", "value_preds", "returns", "action_log_probs", "masks", "bad_masks"):
        getattr(ro, name).copy_(getattr(ro, name).new_tensor(g[name]))
    batches = list(ro.feed_forward_generator(None, mini_batch_size=8, perm=g["disc_perm"]))
    assert len(batches) == int(g["disc_n_batches"]) == 2
    for b, tup in enumerate(batches):
        assert np.array_equal(tup[9].numpy(), g[f"disc_b{b}_next_obs_feat"]) and np.array_equal(tup[0].numpy(), g[f"disc_b{b}_obs"])
        assert tup[7] is None


def test_bare_torch_load_through_alias_modules_gives_the_shim(sg):
    """a2c/main.py:81-83 / my_pybullet_envs/utils.py:43-46: `torch.load(path)` of a reference whole-module checkpoint,
    with `third_party.a2c_ppo_acktr` resolving to this repository's alias package, unpickles straight into the
    device-backed Policy / SplitPolicy."""
    import os
    import subprocess
    import sys

    from helpers import GOLDEN
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, numpy as np, torch\n"
        "import simgan_amd as sg\n"
        "sys.path.insert(0, %r)\n"
        "from helpers import load, assert_close\n"
        "for name in ('ckpt_policy_mlp', 'ckpt_policy_split'):\n"
        "    g = load(name)\n"
        "    actor_critic, ob_rms = torch.load(%r + '/' + name + '.pt', map_location='cpu', weights_only=False)\n"
        "    assert isinstance(actor_critic, (sg.Policy, sg.SplitPolicy)), type(actor_critic)\n"
        "    assert np.array_equal(actor_critic.get_flat_params(), g['flat'])\n"
        "    v, a, lp, _ = actor_critic.act(torch.from_numpy(g['obs']), None, None, deterministic=True)\n"
        "    assert_close(a, g['action'], what='action'); assert_close(v, g['value'], what='value')\n"
        "    if name == 'ckpt_policy_mlp':\n"
        "        assert type(ob_rms) is sg.RunningMeanStd and np.array_equal(ob_rms.mean, g['rms_mean'])\n"
        "print('ok')\n") % (os.path.join(root, "tests"), GOLDEN)
    env = dict(os.environ, PYTHONPATH=root)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env, cwd="/tmp")
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-3000:]


def test_torch_save_by_the_unchanged_main_writes_the_reference_layout(sg, tmp_path):
    """a2c/main_gail_dyn_ppo.py:307-316: `torch.save([actor_critic, ob_rms], path)` on policies built through the
    reference's import path.  The file names only the reference's classes (tests/test_interop.py loads such a file with the
    reference alone, in the dev container); reloaded with a bare torch.load it is a device-backed policy with the same
    weights and outputs; a policy built from `simgan_amd` directly keeps the package's compact pickle."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, numpy as np, torch, pickle\n"
        "import simgan_amd as sg\n"
        "from simgan_amd import checkpoint as ck\n"
        "from third_party.a2c_ppo_acktr.model import Policy\n"
        "from third_party.a2c_ppo_acktr.model_split import SplitPolicy\n"
        "class Box:\n"
        "    def __init__(self, s): self.shape = s\n"
        "for pol in (Policy((11,), Box((3,)), base_kwargs={'recurrent': False, 'hidden_size': 64}),\n"
        "            SplitPolicy((14,), Box((7,)), base_kwargs={'hidden_size': 100, 'num_feet': 1}),\n"
        "            Policy((9,), Box((4,)), base_kwargs={'recurrent': False, 'hidden_size': 32})):\n"
        "    if pol.hidden_size == 32: pol.reset_critic((9,))      # 64-unit critic beside a 32-unit actor\n"
        "    path = sys.argv[1] + '/p.pt'\n"
        "    torch.save([pol, None], path)                         # exactly the main's call\n"
        "    c = ck.read_reference_checkpoint(path)                # parsed with inert stand-ins: the reference's layout\n"
        "    assert np.array_equal(np.concatenate([v.reshape(-1) for v in c['state_dict'].values()]), pol.get_flat_params())\n"
        "    assert c['class_name'] == type(pol).__name__ and c['hidden'] == pol.hidden_size\n"
        "    back, rms = torch.load(path, map_location='cpu', weights_only=False)\n"
        "    assert type(back) is type(pol) and rms is None and back.critic_hidden == pol.critic_hidden\n"
        "    obs = np.random.default_rng(0).standard_normal((5, pol.obs_dim)).astype(np.float32)\n"
        "    a0 = pol.act(obs, None, None, deterministic=True); a1 = back.act(obs, None, None, deterministic=True)\n"
        "    assert all(np.array_equal(np.asarray(x), np.asarray(y)) for x, y in zip(a0[:3], a1[:3]))\n"
        "native = sg.Policy((11,), Box((3,)), base_kwargs={'recurrent': False, 'hidden_size': 64})\n"
        "n2 = pickle.loads(pickle.dumps(native))\n"
        "assert type(n2) is sg.Policy and np.array_equal(n2.get_flat_params(), native.get_flat_params())\n"
        "print('ok')\n")
    env = dict(os.environ, PYTHONPATH=root)
    r = subprocess.run([sys.executable, "-c", code, str(tmp_path)], capture_output=True, text=True, timeout=600, env=env, cwd="/tmp")
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-3000:]


def test_injected_draws_are_validated(sg):
    T, N, F = 4, 8, 7
    D = sg.algo.gail.Discriminator(F, 16, None)
    ro = sg.RolloutStorage(T, N, (3,), Box((2,)), 1, F)
    loader = Loader(np.zeros((40, F), np.float32), 8)
    from simgan_amd._lib import SimganHipError   # lengths and index ranges are checked behind the C ABI (include/simgan_hip.h)
    with pytest.raises(SimganHipError, match="expert_perm holds 39"):
        D.update_gail_dyn(loader, ro, expert_perm=np.arange(39))
    with pytest.raises(SimganHipError, match=r"policy_perm\[31\] = 32 is outside"):
        D.update_gail_dyn(loader, ro, policy_perm=np.arange(1, 33))
    with pytest.raises(SimganHipError, match="alpha holds 8"):
        D.update_gail_dyn(loader, ro, alpha=np.zeros(8, np.float32))
    losses = D.update_gail_dyn(loader, ro)
    ep, pp, al = D.last_draws()
    assert ep.size == 40 and pp.size == 32 and al.size == 32 and np.array_equal(np.sort(pp), np.arange(32))
    D2 = sg.algo.gail.Discriminator(F, 16, None)
    D2.set_flat_params(np.zeros(D2.num_params, np.float32) + 0.01)
    D.set_flat_params(np.zeros(D.num_params, np.float32) + 0.01)
    D.set_adam(np.zeros(D.num_params), np.zeros(D.num_params), 0)
    a_ = D.update_gail_dyn(loader, ro)
    b_ = D2.update_gail_dyn(loader, ro, *D.last_draws())      # replaying exported draws reproduces the epoch bit for bit
    assert a_ == b_ and np.array_equal(D.get_flat_params(), D2.get_flat_params())


KNOBS = [{"SG_PPO_FUSED": "0"}, {"SG_PPO_ROWS": "16"}, {"SG_PPO_ROWS": "32"}, {"SG_PPO_WAVES": "4"}, {"SG_DISC_CHAIN": "wide"},
         {"SG_PPO_GRAPH": "0", "SG_DISC_GRAPH": "0"}, {"SG_WGRAD_XCD": "0"}]


@pytest.mark.parametrize("knob", KNOBS, ids=[",".join(f"{k}={v}" for k, v in kn.items()) for kn in KNOBS])
def test_every_launch_variant_keeps_parity(knob):
    """The library's environment knobs select other kernels / launch geometries for the same math (unfused PPO forward,
    16- or 32-row PPO groups, 4-wave PPO workgroups, the 16-row discriminator chain kernel, direct launches instead of graph
    replay, the linear weight-gradient tile order): the reference trajectories must hold under each of them."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, **knob)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_parity.py"), "-q", "-m", "gpu", "-x",
                        "-p", "no:cacheprovider", "-k", "ppo_update_golden or disc_update_golden or full_iteration_golden or refine_iteration_golden"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and " passed" in r.stdout, (r.stdout[-1500:], r.stderr[-1500:])
