"""GPU, BASELINE.json's full north-star shape (T=128, N=512, obs 47, act 12, D-in 86, Policy h64,
D h100, 100k expert rows): the HIP path against the CPU oracle on the same seeded inputs, plus
size-independent properties of the library's own random machinery.

The oracle runs ~7 ms per discriminator step and ~50 ms per 4096-row PPO step on one core, so a
whole discriminator epoch prefix and a whole PPO epoch are compared here in seconds."""
import ctypes as C

import numpy as np
import pytest

from helpers import assert_close, assert_close_adam

pytestmark = pytest.mark.gpu

T, N, O, A, F, H, HD, B, NE = 128, 512, 47, 12, 86, 64, 100, 128, 100000


class Box:
    def __init__(self, shape):
        self.shape = tuple(shape)


class Loader:
    def __init__(self, expert, batch_size):
        self.expert, self.batch_size = expert, batch_size


@pytest.fixture(scope="module")
def world():
    import simgan_amd as sg
    from simgan_amd import _lib
    rng = np.random.default_rng(2024)
    pol = sg.Policy((O,), Box((A,)), base_kwargs={"recurrent": False, "hidden_size": H}, seed=3)
    disc = sg.algo.gail.Discriminator(F, HD, None, seed=4)
    ro = sg.RolloutStorage(T, N, (O,), Box((A,)), 1, F)
    ro.device_resident = True
    lib = _lib.load()
    _lib.check(lib.sg_rollout_fill_synthetic(ro.h, pol.h, 99, 0.01))
    _lib.check(lib.sg_rollout_compute_returns_policy(ro.h, pol.h, 1, 0.99, 0.95, 1))
    ro.sync_from_device()
    expert = rng.standard_normal((NE, F)).astype(np.float32)
    return dict(sg=sg, lib=lib, _lib=_lib, rng=rng, pol=pol, disc=disc, ro=ro, expert=expert)


def test_synthetic_fill_properties(world):
    """sg_rollout_fill_synthetic: N(0,1) obs, Bernoulli masks, and actions/log-probs/values produced by
    the policy itself (PPO ratios start at exactly 1)."""
    ro, pol = world["ro"], world["pol"]
    obs = ro.obs.numpy()
    assert abs(obs.mean()) < 0.01 and abs(obs.std() - 1.0) < 0.01
    m = ro.masks.numpy()
    assert set(np.unique(m)) <= {0.0, 1.0} and 0.005 < 1.0 - m.mean() < 0.02
    sel = np.arange(0, T * N, 997)
    o = obs[:-1].reshape(-1, O)[sel]
    a = ro.actions.numpy().reshape(-1, A)[sel]
    v, lp, _, _ = pol.evaluate_actions(o, None, None, a)
    assert_close(lp, ro.action_log_probs.numpy().reshape(-1, 1)[sel], what="stored log-probs")
    assert_close(v, ro.value_preds.numpy()[:-1].reshape(-1, 1)[sel], what="stored values")


def test_gae_full_size_vs_oracle(world):
    from oracle import oracle as orc
    ro = world["ro"]
    nv = ro.returns.numpy()[T, :, 0]          # compute_returns_policy parks get_value(obs[T]) in returns[T]
    ret, _ = orc.compute_returns(ro.rewards.numpy()[..., 0], ro.value_preds.numpy()[..., 0], ro.masks.numpy()[..., 0],
                                 ro.bad_masks.numpy()[..., 0], nv, 1, 0.99, 0.95, 1)
    assert_close(ro.returns.numpy()[:T, :, 0], ret[:T], rtol=1e-5, what="GAE returns, 65,536 rows")
    assert_close(ro.value_preds.numpy()[T, :, 0], nv, rtol=0, atol=0, what="value_preds[T] = next_value")


def test_disc_epoch_prefix_full_size_vs_oracle(world):
    """64 consecutive Adam steps of update_gail_dyn at batch 128 on the full-size problem."""
    from oracle import oracle as orc
    sg, rng, ro = world["sg"], world["rng"], world["ro"]
    steps = 64
    expert = world["expert"][:steps * B]
    D = sg.algo.gail.Discriminator(F, HD, None, seed=11)
    p0 = D.get_flat_params()
    eperm = rng.permutation(steps * B).astype(np.int64)
    pperm = rng.permutation(T * N).astype(np.int64)
    alpha = rng.random(steps * B).astype(np.float32)
    losses = D.update_gail_dyn(Loader(expert, B), ro, expert_perm=eperm, policy_perm=pperm, alpha=alpha)
    assert D.last_n_steps == steps
    par, adam = p0.copy(), orc.AdamState(p0.size)
    olosses, n_d = orc.disc_update(F, HD, par, adam, expert, ro.obs_feat.numpy(), B, eperm, pperm, alpha)
    assert n_d == steps
    assert_close(losses, olosses, what="D losses after 64 steps")
    assert_close(D.get_flat_params(), par, what="D params after 64 steps")
    assert np.abs(par - p0).max() > 1e-2   # the trajectory moved far more than the tolerance


def test_ppo_epoch_full_size_vs_oracle(world):
    """One full PPO epoch: 16 optimizer steps on 4096-row minibatches with gradient-norm clipping."""
    from oracle import oracle as orc
    sg, rng, ro = world["sg"], world["rng"], world["ro"]
    pol = sg.Policy((O,), Box((A,)), base_kwargs={"recurrent": False, "hidden_size": H}, seed=21)
    # move off the behaviour policy so ratios leave 1 and the clip branches are exercised
    p0 = (pol.get_flat_params() + 0.01 * rng.standard_normal(pol.num_params)).astype(np.float32)
    pol.set_flat_params(p0)
    agent = sg.algo.PPO(pol, 0.2, 1, 16, 0.5, 0.01, lr=3e-4, eps=1e-5, max_grad_norm=0.5)
    perms = rng.permutation(T * N).astype(np.int64)[None, :]
    losses = agent.update(ro, perms=perms)
    d = orc.dims(orc.KIND_MLP, O, A, H, 1)
    par, adam = p0.copy(), orc.AdamState(p0.size)
    olosses = orc.ppo_update(d, par, adam, orc.ppo_cfg(0.2, 1, 16, 0.5, 0.01, 3e-4, 1e-5, 0.5, True), ro.obs.numpy(),
                             ro.actions.numpy(), ro.value_preds.numpy()[..., 0], ro.returns.numpy()[..., 0],
                             ro.action_log_probs.numpy()[..., 0], perms)
    assert_close(ro.device_advantages().numpy().reshape(-1), orc.advantages(ro.returns.numpy()[:-1], ro.value_preds.numpy()[:-1]),
                 rtol=1e-5, what="normalised advantages")
    assert_close(losses, olosses, what="PPO losses over the epoch")
    assert_close(pol.get_flat_params(), par, what="policy params after 16 steps")


def test_relabel_full_size_vs_oracle(world):
    from oracle import oracle as orc
    sg, ro, disc = world["sg"], world["ro"], world["disc"]
    rms = sg.RunningMeanStd(shape=())
    disc.returns = None
    disc.relabel_rewards(ro, 0.99, -0.25, rms)
    ro.sync_from_device([3])
    orew, oret, orms = orc.relabel(F, HD, disc.get_flat_params(), ro.obs_feat.numpy(), ro.masks.numpy()[..., 0], 0.99,
                                   -0.25, None, [0.0, 1.0, 1e-4])
    assert_close(ro.rewards.numpy()[..., 0], orew, what="relabelled rewards, 65,536 rows")
    assert_close(rms.get_state(), orms, rtol=1e-5, what="ret_rms after 128 merges")
    assert_close(disc.returns.numpy()[:, 0], oret, what="Discriminator.returns")


def test_library_rng_and_determinism(world):
    """Without injected artefacts the library draws its own permutations / alpha: every minibatch row
    is visited exactly once per epoch (the cycle-walking Feistel map is a bijection), and a run is a pure
    function of (weights, data, seed): two identical contexts give bit-identical results."""
    sg, ro, expert = world["sg"], world["ro"], world["expert"]

    def run():
        D = sg.algo.gail.Discriminator(F, HD, None, seed=5)
        D.seed = 1234
        l1 = D.update_gail_dyn(Loader(expert[:4096], B), ro)
        pol = sg.Policy((O,), Box((A,)), base_kwargs={"recurrent": False, "hidden_size": H}, seed=6)
        agent = sg.algo.PPO(pol, 0.2, 1, 16, 0.5, 0.0, lr=3e-4, eps=1e-5, max_grad_norm=0.5)
        agent.seed = 77
        l2 = agent.update(ro)
        return l1, D.get_flat_params(), l2, pol.get_flat_params()

    a, b = run(), run()
    assert a[0] == b[0] and a[2] == b[2]
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[3], b[3])
    assert all(np.isfinite(x) for x in a[0] + a[2])
    # permutation bijectivity: fill one through the test of PPO's generator via a tiny rollout
    from simgan_amd import _lib
    lib = world["lib"]
    small = sg.RolloutStorage(5, 7, (O,), Box((A,)), 1, F)      # 35 rows: not a power of two
    small.device_resident = True
    polS = sg.Policy((O,), Box((A,)), base_kwargs={"recurrent": False, "hidden_size": H}, seed=8)
    _lib.check(lib.sg_rollout_fill_synthetic(small.h, polS.h, 5, 0.1))
    _lib.check(lib.sg_rollout_compute_returns_policy(small.h, polS.h, 1, 0.99, 0.95, 1))
    agentS = sg.algo.PPO(polS, 0.2, 3, 5, 0.5, 0.0, lr=3e-4, eps=1e-5, max_grad_norm=0.5)
    out = agentS.update(small)
    assert all(np.isfinite(x) for x in out)


@pytest.mark.parametrize("n", [1, 2, 35, 4096, 65536, 100000])
def test_device_permutation_is_a_bijection(world, n):
    _lib, lib = world["_lib"], world["lib"]
    ctx = _lib.Context.default()
    fn = _lib.load_test().sg_test_rng
    perm = np.empty(n, np.int64)
    _lib.check_test(fn(ctx.h, 0, n, 12345, perm.ctypes.data_as(C.c_void_p)))
    assert np.array_equal(np.sort(perm), np.arange(n))
    if n >= 4096:   # not the identity, not a rotation: displacement statistics of a random permutation
        disp = np.abs(perm - np.arange(n)) / n
        assert 0.25 < disp.mean() < 0.42
        perm2 = np.empty(n, np.int64)
        _lib.check_test(fn(ctx.h, 0, n, 12346, perm2.ctypes.data_as(C.c_void_p)))
        assert (perm != perm2).mean() > 0.99


def test_device_uniform_and_normal(world):
    _lib, lib = world["_lib"], world["lib"]
    ctx = _lib.Context.default()
    fn = _lib.load_test().sg_test_rng
    n = 1 << 20
    u = np.empty(n, np.float32)
    _lib.check_test(fn(ctx.h, 1, n, 5, u.ctypes.data_as(C.c_void_p)))
    assert u.min() >= 0.0 and u.max() < 1.0
    assert abs(u.mean() - 0.5) < 2e-3 and abs(u.var() - 1 / 12) < 1e-3
    z = np.empty(n, np.float32)
    _lib.check_test(fn(ctx.h, 2, n, 5, z.ctypes.data_as(C.c_void_p)))
    assert abs(z.mean()) < 5e-3 and abs(z.std() - 1.0) < 5e-3 and np.isfinite(z).all()
    assert abs((np.abs(z) > 2).mean() - 0.0455) < 2e-3


# ------------------------------------------------------------------ the other BASELINE.json configurations at full size
def _filled_rollout(sg, lib, _lib, pol, T_, N_, O_, A_, F_, seed):
    ro = sg.RolloutStorage(T_, N_, (O_,), Box((A_,)), 1, F_)
    ro.device_resident = True
    _lib.check(lib.sg_rollout_fill_synthetic(ro.h, pol.h, seed, 0.01))
    _lib.check(lib.sg_rollout_compute_returns_policy(ro.h, pol.h, 1, 0.99, 0.95, 1))
    ro.sync_from_device()
    return ro


FULL_PPO_CASES = [
    # BASELINE.json configs[4]: Laikago policy refinement, 2048 envs -> 256 per GPU (train_laika_power.sh:7)
    dict(id="refine", kind="mlp", O=111, A=12, H=64, f=1, N=256, M=8, clip=0.1, lr=1.5e-4),
    # configs[1]: HopperCombinedEnv-v1 GAIL-dyn, SplitPolicy h100 as shipped (train_hopper_deform.sh:5), 256 envs
    dict(id="hopper-split", kind="split", O=14, A=7, H=100, f=1, N=256, M=16, clip=0.2, lr=3e-4),
    # configs[2]: LaikagoCombinedEnv-v1 GAIL-dyn, SplitPolicy h100 nf=4 (train_laika_heavy.sh:5), 512 envs
    dict(id="laikago-split", kind="split", O=64, A=28, H=100, f=4, N=512, M=16, clip=0.2, lr=3e-4),
]


@pytest.mark.parametrize("c", FULL_PPO_CASES, ids=[c["id"] for c in FULL_PPO_CASES])
def test_ppo_epoch_full_size_other_configs_vs_oracle(world, c):
    """One full PPO epoch (num_mini_batch optimizer steps over all T*N rows) at each remaining configuration's real
    shape -- the shape-specialised kernel instances bench.py's hopper / laikago / refine workloads launch."""
    from oracle import oracle as orc
    sg, lib, _lib = world["sg"], world["lib"], world["_lib"]
    rng = np.random.default_rng(77)
    if c["kind"] == "mlp":
        pol = sg.Policy((c["O"],), Box((c["A"],)), base_kwargs={"recurrent": False, "hidden_size": c["H"]}, seed=31)
    else:
        pol = sg.SplitPolicy((c["O"],), Box((c["A"],)), base_kwargs={"hidden_size": c["H"], "num_feet": c["f"]}, seed=31)
    ro = _filled_rollout(sg, lib, _lib, pol, T, c["N"], c["O"], c["A"], 4, 5)
    p0 = (pol.get_flat_params() + 0.01 * rng.standard_normal(pol.num_params)).astype(np.float32)
    pol.set_flat_params(p0)
    agent = sg.algo.PPO(pol, c["clip"], 1, c["M"], 0.5, 0.01, lr=c["lr"], eps=1e-5, max_grad_norm=0.5)
    perms = rng.permutation(T * c["N"]).astype(np.int64)[None, :]
    losses = agent.update(ro, perms=perms)
    d = orc.dims(orc.KIND_MLP if c["kind"] == "mlp" else orc.KIND_SPLIT, c["O"], c["A"], c["H"], c["f"])
    par, adam = p0.copy(), orc.AdamState(p0.size)
    olosses = orc.ppo_update(d, par, adam, orc.ppo_cfg(c["clip"], 1, c["M"], 0.5, 0.01, c["lr"], 1e-5, 0.5, True), ro.obs.numpy(),
                             ro.actions.numpy(), ro.value_preds.numpy()[..., 0], ro.returns.numpy()[..., 0],
                             ro.action_log_probs.numpy()[..., 0], perms)
    assert_close(losses, olosses, what=f"{c['id']}: PPO losses over the epoch")
    assert_close(pol.get_flat_params(), par, what=f"{c['id']}: policy params after {c['M']} steps")
    assert np.abs(par - p0).max() > 10 * c["lr"] * 0.5


def test_disc_epoch_prefix_hopper_shape_vs_oracle(world):
    """64 Adam steps of update_gail_dyn at the HopperCombinedEnv-v1 shape (D-in 25 = 11+3+11, hidden 100, 256 envs)."""
    from oracle import oracle as orc
    sg, lib, _lib, rng = world["sg"], world["lib"], world["_lib"], np.random.default_rng(12)
    Fh, Nh, steps = 25, 256, 64
    pol = sg.SplitPolicy((14,), Box((7,)), base_kwargs={"hidden_size": 100, "num_feet": 1}, seed=2)
    ro = _filled_rollout(sg, lib, _lib, pol, T, Nh, 14, 7, Fh, 6)
    expert = rng.standard_normal((steps * B, Fh)).astype(np.float32)
    D = sg.algo.gail.Discriminator(Fh, HD, None, seed=13)
    p0 = D.get_flat_params()
    eperm = rng.permutation(steps * B).astype(np.int64)
    pperm = rng.permutation(T * Nh).astype(np.int64)
    alpha = rng.random(steps * B).astype(np.float32)
    losses = D.update_gail_dyn(Loader(expert, B), ro, expert_perm=eperm, policy_perm=pperm, alpha=alpha)
    par, adam = p0.copy(), orc.AdamState(p0.size)
    olosses, n_d = orc.disc_update(Fh, HD, par, adam, expert, ro.obs_feat.numpy(), B, eperm, pperm, alpha)
    assert n_d == steps == D.last_n_steps
    assert_close(losses, olosses, what="D losses after 64 steps (Hopper shape)")
    assert_close(D.get_flat_params(), par, what="D params after 64 steps (Hopper shape)")


def _disc_epoch(sg, ro, expert, eperm, pperm, alpha, p0, fused, ctx=None, Fin=F):
    import os
    old = os.environ.get("SG_DISC_FUSED")
    os.environ["SG_DISC_FUSED"] = "1" if fused else "0"
    try:
        D = sg.algo.gail.Discriminator(Fin, HD, None, seed=11) if ctx is None else sg.algo.gail.Discriminator(Fin, HD, None, ctx=ctx, seed=11)
        D.set_flat_params(p0)
        losses = D.update_gail_dyn(Loader(expert, B), ro, expert_perm=eperm, policy_perm=pperm, alpha=alpha)
        return np.asarray(losses, dtype=np.float64), D.get_flat_params(), D.get_adam()
    finally:
        if old is None:
            os.environ.pop("SG_DISC_FUSED", None)
        else:
            os.environ["SG_DISC_FUSED"] = old


def test_one_launch_step_is_bit_identical_to_the_two_launch_step(world):
    """k_disc_step4 (chain and weight-gradient workgroups in one launch, joined by a flag hand-off inside it) against
    k_disc_chain4 + k_disc_wgrad on the same 256 steps: same arithmetic in the same order, so every weight, both Adam
    moments and the loss sums must be EQUAL -- a single stale word crossing the hand-off would show."""
    sg, ro = world["sg"], world["ro"]
    rng = np.random.default_rng(77)
    steps = 256
    expert = world["expert"][:steps * B]
    eperm = rng.permutation(steps * B).astype(np.int64)
    pperm = rng.permutation(T * N).astype(np.int64)
    alpha = rng.random(steps * B).astype(np.float32)
    p0 = sg.algo.gail.Discriminator(F, HD, None, seed=5).get_flat_params()
    l2, p2, a2 = _disc_epoch(sg, ro, expert, eperm, pperm, alpha, p0, fused=False)
    l1, p1, a1 = _disc_epoch(sg, ro, expert, eperm, pperm, alpha, p0, fused=True)
    assert np.array_equal(l1, l2), (l1, l2)
    assert np.array_equal(p1, p2), f"{(p1 != p2).sum()} of {p1.size} weights differ, worst {np.abs(p1 - p2).max():.3g}"
    for x, y in zip(a1[:2], a2[:2]):
        assert np.array_equal(x, y)
    assert np.abs(p1 - p0).max() > 1e-2


def _run_under_load(world, subject, n_noise=5):
    """subject(ctx) on a context of its own while `n_noise` other contexts keep the chip and the fabric busy with kernels that
    never wait for anything (policy evaluation on 16k rows, host buffers up and down every call).  A launch that waits inside
    itself must not share the device with ANOTHER such launch (two of them can hold the slots each other's workgroups need
    until their spins time out: csrc/sg_common.h, sg_ctx_exclusive -- the library then uses the multi-launch forms on its own),
    so the load here is of the kind that can only delay, and the one-launch form is forced on for the subject."""
    import threading
    sg, _lib = world["sg"], world["_lib"]
    stop, errs, out = threading.Event(), [], [None]
    rng = np.random.default_rng(5)
    obs = rng.standard_normal((16384, O)).astype(np.float32)
    act = rng.standard_normal((16384, A)).astype(np.float32)

    def noise(i):
        try:
            ctx = _lib.Context(0)
            pol = sg.Policy((O,), Box((A,)), base_kwargs={"recurrent": False, "hidden_size": H}, seed=40 + i, ctx=ctx)
            while not stop.is_set():
                pol.evaluate_actions(obs, None, None, act)
        except Exception as e:  # noqa: BLE001
            errs.append("noise: " + repr(e))

    def subj():
        try:
            out[0] = subject(_lib.Context(0))
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))

    th = [threading.Thread(target=noise, args=(i,)) for i in range(n_noise)]
    [t.start() for t in th]
    ts = threading.Thread(target=subj)
    ts.start()
    ts.join(900)
    stop.set()
    [t.join(60) for t in th]
    assert not errs, errs
    assert out[0] is not None, "the subject did not finish"
    return out[0]


def test_one_launch_step_hand_off_under_load(world):
    """The same comparison while the GPU is shared unevenly: five other contexts keep it busy while one context runs the
    one-launch epoch three times; each run must reproduce the two-launch result of an idle GPU bit for bit.  (A hand-off whose
    flag can overtake its data passes on an idle chip and fails under load: the store drain in front of the flag was first
    missing, and this is how it showed.)"""
    import os
    sg, ro = world["sg"], world["ro"]
    rng = np.random.default_rng(78)
    steps = 192
    expert = world["expert"][:steps * B]
    eperm = rng.permutation(steps * B).astype(np.int64)
    pperm = rng.permutation(T * N).astype(np.int64)
    alpha = rng.random(steps * B).astype(np.float32)
    p0 = sg.algo.gail.Discriminator(F, HD, None, seed=6).get_flat_params()
    want_l, want_p, _ = _disc_epoch(sg, ro, expert, eperm, pperm, alpha, p0, fused=False)
    feat = ro.obs_feat.numpy().copy()

    def subject(ctx):
        r = sg.RolloutStorage(T, N, (O,), Box((A,)), 1, F, ctx=ctx)
        r.obs_feat.copy_(r.obs_feat.new_tensor(feat))
        res = []
        for _ in range(3):
            D = sg.algo.gail.Discriminator(F, HD, None, ctx=ctx, seed=11)
            D.set_flat_params(p0)
            ls = D.update_gail_dyn(Loader(expert, B), r, expert_perm=eperm, policy_perm=pperm, alpha=alpha)
            res.append((np.asarray(ls, dtype=np.float64), D.get_flat_params()))
        return res

    os.environ["SG_DISC_FUSED"] = "1"
    try:
        res = _run_under_load(world, subject)
    finally:
        os.environ.pop("SG_DISC_FUSED", None)
    for ls, p in res:
        assert np.array_equal(p, want_p), f"{(p != want_p).sum()} weights differ, worst {np.abs(p - want_p).max():.3g}"
        assert np.array_equal(ls, want_l)


def test_one_launch_forms_are_left_to_contexts_that_have_the_device_to_themselves(world):
    """Two contexts of one process that both own learner objects on one device: neither may pick a launch that waits inside itself
    (two such launches can block each other: csrc/sg_common.h, sg_ctx_exclusive).  With the switches unset the library takes the
    two-launch step on its own for the second context (the module's fixtures live on the default one), and goes back to the
    one-launch step for the default context once the other's objects are gone; the profiling slots say which ran."""
    sg, _lib, lib = world["sg"], world["_lib"], world["lib"]
    import os
    assert "SG_DISC_FUSED" not in os.environ and "SG_PPO_PAIR" not in os.environ
    rng = np.random.default_rng(79)
    steps = 8
    expert = world["expert"][:steps * B]
    feat = world["ro"].obs_feat.numpy().copy()

    def step_launches(ctx):
        r = sg.RolloutStorage(T, N, (O,), Box((A,)), 1, F, ctx=ctx)
        r.obs_feat.copy_(r.obs_feat.new_tensor(feat))
        D = sg.algo.gail.Discriminator(F, HD, None, ctx=ctx, seed=11)
        ctx.profile_reset()
        ctx.profile(True)
        D.update_gail_dyn(Loader(expert, B), r, expert_perm=rng.permutation(steps * B).astype(np.int64),
                          policy_perm=rng.permutation(T * N).astype(np.int64), alpha=rng.random(steps * B).astype(np.float32))
        ctx.profile(False)
        return ctx.profile_read(7)[1], ctx.profile_read(0)[1]   # launches of k_disc_step4, of k_disc_chain4

    import gc
    other = _lib.Context(0)          # a second context with a discriminator of its own: nobody is alone any more
    one, chain = step_launches(other)
    assert (one, chain) == (0, steps), (one, chain)
    gc.collect()                     # (its rollout and discriminator were locals of step_launches: gone)
    solo = _lib.Context.default()
    one, chain = step_launches(solo)
    assert (one, chain) == (steps, 0), (one, chain)


# ------------------------------------------------------------------ SplitPolicy: one launch per PPO step (k_ppo_pair)
PAIR_CASES = [
    # configs[1] as shipped: 64 row groups of 32 rows x 3 trunks = 192 workgroups, the shape-specialised instance
    dict(id="hopper-split", O=14, A=7, H=100, f=1, N=256, M=16, E=2),
    # LaikagoCombined trunks at a 2048-row minibatch (256 envs): the (2, 4, 7) instance
    dict(id="laikago-split-n256", O=64, A=28, H=100, f=4, N=256, M=16, E=1),
    # a width no instance is specialised for, ragged last row group: the run-time-shape instance
    dict(id="split-h72-f2-ragged", O=21, A=14, H=72, f=2, N=37, M=4, E=2),
]


def _ppo_update_split(sg, lib, _lib, c, p0, perms, pair, ctx=None, ro=None):
    import os
    old = os.environ.get("SG_PPO_PAIR")
    os.environ["SG_PPO_PAIR"] = "1" if pair else "0"
    try:
        kw = {} if ctx is None else {"ctx": ctx}
        pol = sg.SplitPolicy((c["O"],), Box((c["A"],)), base_kwargs={"hidden_size": c["H"], "num_feet": c["f"]}, seed=31, **kw)
        pol.set_flat_params(p0)
        agent = sg.algo.PPO(pol, 0.2, c["E"], c["M"], 0.5, 0.01, lr=3e-4, eps=1e-5, max_grad_norm=0.5)
        losses = agent.update(ro, perms=perms)
        return np.asarray(losses, dtype=np.float64), pol.get_flat_params(), agent.get_adam()
    finally:
        if old is None:
            os.environ.pop("SG_PPO_PAIR", None)
        else:
            os.environ["SG_PPO_PAIR"] = old


def _pair_problem(world, c, seed):
    sg, lib, _lib = world["sg"], world["lib"], world["_lib"]
    rng = np.random.default_rng(seed)
    pol = sg.SplitPolicy((c["O"],), Box((c["A"],)), base_kwargs={"hidden_size": c["H"], "num_feet": c["f"]}, seed=31)
    ro = _filled_rollout(sg, lib, _lib, pol, T, c["N"], c["O"], c["A"], 4, 5)
    p0 = (pol.get_flat_params() + 0.01 * rng.standard_normal(pol.num_params)).astype(np.float32)
    perms = np.stack([rng.permutation(T * c["N"]) for _ in range(c["E"])]).astype(np.int64)
    return ro, p0, perms


@pytest.mark.parametrize("c", PAIR_CASES, ids=[c["id"] for c in PAIR_CASES])
def test_one_launch_split_ppo_step_is_bit_identical_to_the_two_launch_step(world, c):
    """k_ppo_pair (every trunk's fused forward + backward in one launch, the two actor workgroups of a row group swapping
    their head outputs through a flag hand-off) against k_ppo_fwd + k_ppo_bwd on the same steps: the swapped values are the ones
    the forward launch leaves in the OUT stacks and the loss code is shared, so losses, weights and both Adam moments must be
    EQUAL -- one stale word crossing the hand-off would show."""
    sg, lib, _lib = world["sg"], world["lib"], world["_lib"]
    ro, p0, perms = _pair_problem(world, c, 91)
    l2, p2, a2 = _ppo_update_split(sg, lib, _lib, c, p0, perms, pair=False, ro=ro)
    l1, p1, a1 = _ppo_update_split(sg, lib, _lib, c, p0, perms, pair=True, ro=ro)
    assert np.array_equal(l1, l2), (l1, l2)
    assert np.array_equal(p1, p2), f"{(p1 != p2).sum()} of {p1.size} weights differ, worst {np.abs(p1 - p2).max():.3g}"
    for x, y in zip(a1[:2], a2[:2]):
        assert np.array_equal(x, y)
    assert np.abs(p1 - p0).max() > 1e-3


def test_one_launch_split_ppo_step_under_load(world):
    """The same comparison while five other contexts keep the GPU busy: the pair launch, forced on for one context, must
    reproduce the two-launch result of an idle GPU bit for bit, three updates in a row (no flag overtaking its data, no torn
    word)."""
    import os
    sg, lib, _lib = world["sg"], world["lib"], world["_lib"]
    c = PAIR_CASES[0]
    ro, p0, perms = _pair_problem(world, c, 92)
    want_l, want_p, _ = _ppo_update_split(sg, lib, _lib, c, p0, perms, pair=False, ro=ro)
    fields = {k: getattr(ro, k).numpy().copy() for k in ("obs", "actions", "value_preds", "returns", "action_log_probs", "masks")}

    def subject(ctx):
        r = sg.RolloutStorage(T, c["N"], (c["O"],), Box((c["A"],)), 1, 4, ctx=ctx)
        for k, v in fields.items():
            getattr(r, k).copy_(getattr(r, k).new_tensor(v))
        res = []
        for _ in range(3):
            pol = sg.SplitPolicy((c["O"],), Box((c["A"],)), base_kwargs={"hidden_size": c["H"], "num_feet": c["f"]}, seed=31, ctx=ctx)
            pol.set_flat_params(p0)
            agent = sg.algo.PPO(pol, 0.2, c["E"], c["M"], 0.5, 0.01, lr=3e-4, eps=1e-5, max_grad_norm=0.5)
            ls = agent.update(r, perms=perms)
            res.append((np.asarray(ls, dtype=np.float64), pol.get_flat_params()))
        return res

    os.environ["SG_PPO_PAIR"] = "1"
    try:
        res = _run_under_load(world, subject)
    finally:
        os.environ.pop("SG_PPO_PAIR", None)
    for ls, p in res:
        assert np.array_equal(p, want_p), f"{(p != want_p).sum()} weights differ, worst {np.abs(p - want_p).max():.3g}"
        assert np.array_equal(ls, want_l)


def test_one_launch_split_ppo_step_after_two_launch_updates_of_the_same_agent(world):
    """The actor pairs' tagged words live in the H1 row stacks, where the two-launch step keeps activations: an agent that
    changes mode between updates (SG_PPO_PAIR toggled) must still reproduce the all-two-launch trajectory bit for bit."""
    import os
    sg, lib, _lib = world["sg"], world["lib"], world["_lib"]
    c = PAIR_CASES[0]
    ro, p0, perms = _pair_problem(world, c, 96)

    def run(modes):
        pol = sg.SplitPolicy((c["O"],), Box((c["A"],)), base_kwargs={"hidden_size": c["H"], "num_feet": c["f"]}, seed=31)
        pol.set_flat_params(p0)
        agent = sg.algo.PPO(pol, 0.2, c["E"], c["M"], 0.5, 0.01, lr=3e-4, eps=1e-5, max_grad_norm=0.5)
        out = []
        for m in modes:
            os.environ["SG_PPO_PAIR"] = m
            try:
                out.append((np.asarray(agent.update(ro, perms=perms), dtype=np.float64), pol.get_flat_params()))
            finally:
                os.environ.pop("SG_PPO_PAIR", None)
        return out

    want = run(["0", "0", "0", "0"])
    got = run(["0", "1", "0", "1"])
    for (wl, wp), (gl, gp) in zip(want, got):
        assert np.array_equal(gl, wl), (gl, wl)
        assert np.array_equal(gp, wp), f"{(gp != wp).sum()} weights differ"


# -------------------------------------------------------- one-row-group minibatches: the smallest grid the PPO step runs on
ONE_GROUP_CASES = [
    # BASELINE.json configs[0]'s exact geometry (a2c/main.py + a2c/arguments.py defaults): 8 envs x 128 steps, obs 11 / act 3,
    # h64, 32 minibatches of 32 rows, entropy coefficient 0.01
    dict(id="hopper_ppo", O=11, A=3, H=64, N=8, M=32, E=3, clip=0.2, lr=3e-4, ecoef=0.01),
    # 16-row steps (one row tile), a trunk boundary that is not a multiple of 64 parameters, no entropy term
    dict(id="rows16", O=20, A=5, H=48, N=4, M=32, E=2, clip=0.1, lr=1e-3, ecoef=0.0),
    # a wider observation on 32-row steps
    dict(id="obs30", O=30, A=6, H=64, N=8, M=32, E=2, clip=0.2, lr=3e-4, ecoef=0.0),
    # the north-star policy on 32-row steps
    dict(id="northstar_policy", O=47, A=12, H=64, N=8, M=32, E=2, clip=0.2, lr=3e-4, ecoef=0.0),
]


@pytest.mark.parametrize("c", ONE_GROUP_CASES, ids=[c["id"] for c in ONE_GROUP_CASES])
def test_one_row_group_minibatches_vs_oracle(world, c):
    """a2c/algo/ppo.py:74-149 at the reference's own CPU-runnable geometry (a2c/arguments.py:97-100: 32 minibatches of a 1024-row
    rollout -> 32-ROW optimizer steps): one workgroup per trunk per step, E x M steps of k_ppo_bwd / k_ppo_reduce / k_ppo_adam,
    against the oracle -- losses of the update and the parameters after E*M Adam steps.  (Round 5's one-launch-per-epoch form of
    this case, k_ppo_small, measured slower -- profiles/r05_ppo_small_negative.txt -- and was removed in round 6.)"""
    from oracle import oracle as orc
    sg, lib, _lib = world["sg"], world["lib"], world["_lib"]
    rng = np.random.default_rng(5)
    pol = sg.Policy((c["O"],), Box((c["A"],)), base_kwargs={"recurrent": False, "hidden_size": c["H"]}, seed=31)
    ro = _filled_rollout(sg, lib, _lib, pol, T, c["N"], c["O"], c["A"], 4, 9)
    p0 = (pol.get_flat_params() + 0.01 * rng.standard_normal(pol.num_params)).astype(np.float32)
    pol.set_flat_params(p0)
    perms = np.stack([rng.permutation(T * c["N"]) for _ in range(c["E"])]).astype(np.int64)
    assert T * c["N"] // c["M"] <= 32, "the cases are one-row-group minibatches"
    agent = sg.algo.PPO(pol, c["clip"], c["E"], c["M"], 0.5, c["ecoef"], lr=c["lr"], eps=1e-5, max_grad_norm=0.5)
    losses = agent.update(ro, perms=perms)
    assert agent.get_adam()[2] == c["E"] * c["M"]
    d = orc.dims(orc.KIND_MLP, c["O"], c["A"], c["H"], 1)
    par, adam = p0.copy(), orc.AdamState(p0.size)
    olosses = orc.ppo_update(d, par, adam, orc.ppo_cfg(c["clip"], c["E"], c["M"], 0.5, c["ecoef"], c["lr"], 1e-5, 0.5, True), ro.obs.numpy(),
                             ro.actions.numpy(), ro.value_preds.numpy()[..., 0], ro.returns.numpy()[..., 0],
                             ro.action_log_probs.numpy()[..., 0], perms)
    assert_close(losses, olosses, what=f"{c['id']}: PPO losses of one update")
    assert_close_adam(pol.get_flat_params(), par, c["lr"], c["E"] * c["M"], what=f"{c['id']}: parameters after {c['E'] * c['M']} steps")
    assert np.abs(pol.get_flat_params() - p0).max() > 5 * c["lr"] * 0.5
