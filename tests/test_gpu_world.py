"""GPU: the library's world > 1 code on ONE device, against the single-process oracle.

`world` contexts (one host thread each; ctypes releases the GIL around every library call) share device 0 and a
LOOPBACK communicator (csrc/sg_comm.cpp: shared-memory transport behind the same sg_comm_* calls the RCCL path uses), so
every data-parallel branch of the HIP code executes for real: env-column sharding, the global advantage / return
statistics, `count_dones`, 1/B over the GLOBAL minibatch, the replicated discriminator's all-gather and row numbering,
the sharded discriminator's rank slicing and per-step gradient all-reduce.

Reference semantics: the reference is single-process, so "world ranks of N columns" must equal
the reference at num_processes = world * N.  Injected draws are therefore GLOBAL -- the permutation the reference's
sampler would draw over T * N * world rows in its numbering t * (N * world) + n (a2c/storage.py:159-185) -- the same
arrays on every rank, and the result must equal the oracle run once on the concatenated rollout.
"""
import ctypes as C
import os
import sys
import threading

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

pytestmark = pytest.mark.gpu

GAMMA, LAM = 0.99, 0.95


class Box:
    def __init__(self, shape):
        self.shape = tuple(shape)


class Loader:
    def __init__(self, expert, batch_size):
        self.expert, self.batch_size = expert, batch_size


def run_ranks(world, fn, timeout_s=600):
    """fn(rank, ctx) on `world` threads, each with its own context on device 0 joined by one loopback communicator.
    Returns the list of results; the first exception of any rank is re-raised."""
    from simgan_amd import _lib
    os.environ.setdefault("SG_LOOPBACK_TIMEOUT_S", "120")
    uid = _lib.comm_loopback_id()
    out, err = [None] * world, [None] * world

    def body(rank):
        try:
            ctx = _lib.Context(0)
            ctx.comm_init(uid, rank, world)
            assert ctx.comm_kind() == "loopback" and ctx.comm_info() == (rank, world)
            out[rank] = fn(rank, ctx)
            ctx.synchronize()
        except BaseException as exc:   # noqa: BLE001 -- reported to the test below
            err[rank] = exc

    threads = [threading.Thread(target=body, args=(r,), daemon=True) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout_s)
    assert not any(t.is_alive() for t in threads), "a rank is still running (loopback collective stuck?)"
    bad = [(r, e) for r, e in enumerate(err) if e is not None]
    if bad:   # a rank that fails leaves its peers to time out in their next collective: report the root cause first
        bad.sort(key=lambda re_: ("peer rank failed" in str(re_[1])) + 2 * ("timed out" in str(re_[1])))
        raise AssertionError("; ".join(f"rank {r}: {type(e).__name__}: {str(e)[:300]}" for r, e in bad[:3])) from bad[0][1]
    return out


def make_global(world, T, N_loc, O, A, F, kind, H, feet, Ne, seed):
    """The world = 1 problem the ranks must reproduce: a rollout of world * N_loc columns."""
    from oracle import oracle as orc
    rng = np.random.default_rng(seed)
    Ng = world * N_loc
    f32 = lambda *s: rng.standard_normal(s).astype(np.float32)  # noqa: E731
    d = orc.dims(orc.KIND_MLP if kind == "mlp" else orc.KIND_SPLIT, O, A, H, feet)
    g = dict(world=world, T=T, N_loc=N_loc, Ng=Ng, O=O, A=A, F=F, kind=kind, H=H, feet=feet, d=d,
             obs=f32(T + 1, Ng, O), obs_feat=f32(T + 1, Ng, F), masks=(rng.random((T + 1, Ng)) > 0.08).astype(np.float32),
             bad_masks=(rng.random((T + 1, Ng)) > 0.03).astype(np.float32), rewards=f32(T, Ng),
             expert=f32(Ne, F) if Ne else None)
    g["pi"] = (0.3 * f32(orc.policy_num_params(d))).astype(np.float32)
    # a consistent "old policy": actions / log-probs / values from a perturbed copy, so ratios and clipping are exercised
    old = (g["pi"] + 0.02 * f32(g["pi"].size)).astype(np.float32)
    acts, logp, vals = [], [], []
    for t in range(T):
        v, a, lp = orc.policy_act(d, old, g["obs"][t], f32(Ng, A))
        acts.append(a); logp.append(lp[:, 0]); vals.append(v[:, 0])
    g["actions"], g["logp"] = np.stack(acts), np.stack(logp)
    g["value_preds"] = np.concatenate([np.stack(vals), f32(1, Ng)]).astype(np.float32)
    return g


def shard(a, rank, n_loc):
    return np.ascontiguousarray(a[:, rank * n_loc:(rank + 1) * n_loc])


def build_rank(sg, g, rank, ctx, hp, disc_seed=11):
    """Policy / PPO / Discriminator / RolloutStorage of one rank on its column shard (drop-in mode: host tensors are the
    source, every device call uploads what it reads)."""
    T, N, O, A, F = g["T"], g["N_loc"], g["O"], g["A"], g["F"]
    if g["kind"] == "mlp":
        pol = sg.Policy((O,), Box((A,)), base_kwargs={"recurrent": False, "hidden_size": g["H"]}, ctx=ctx)
    else:
        pol = sg.SplitPolicy((O,), Box((A,)), base_kwargs={"hidden_size": g["H"], "num_feet": g["feet"]}, ctx=ctx)
    pol.set_flat_params(g["pi"])
    agent = sg.algo.PPO(pol, hp["clip"], hp["E"], hp["M"], 0.5, hp["ecoef"], lr=3e-4, eps=1e-5, max_grad_norm=0.5)
    ro = sg.RolloutStorage(T, N, (O,), Box((A,)), 1, F, ctx=ctx)
    put = lambda dst, src: dst.copy_(dst.new_tensor(np.ascontiguousarray(src).reshape(tuple(dst.shape))))  # noqa: E731
    put(ro.obs, shard(g["obs"], rank, N)); put(ro.obs_feat, shard(g["obs_feat"], rank, N))
    put(ro.actions, shard(g["actions"], rank, N)); put(ro.action_log_probs, shard(g["logp"], rank, N))
    put(ro.value_preds, shard(g["value_preds"], rank, N)); put(ro.masks, shard(g["masks"], rank, N))
    put(ro.bad_masks, shard(g["bad_masks"], rank, N)); put(ro.rewards, shard(g["rewards"], rank, N))
    disc = None
    if g["expert"] is not None:
        disc = sg.algo.gail.Discriminator(F, hp["Hd"], None, ctx=ctx, seed=disc_seed)
        disc.set_flat_params(g["dpar"])
    return pol, agent, disc, ro


def oracle_returns(g, next_value):
    from oracle import oracle as orc
    return orc.compute_returns(g["rewards"], g["value_preds"], g["masks"], g["bad_masks"], next_value, 1, GAMMA, LAM, 1)


SHAPES = {
    # tiny, ragged on purpose: T*N_glob not a multiple of M, batch not a multiple of 16
    "mlp_small": dict(T=7, N_loc=6, O=11, A=3, F=9, kind="mlp", H=16, feet=1, Hd=16, B=24, Ne=200, E=2, M=5, clip=0.2, ecoef=0.01),
    # the north-star policy / discriminator shapes (specialised kernels: k_ppo_bwd<.,3,4,true>, k_disc_chain4<6,7>)
    "northstar": dict(T=16, N_loc=16, O=47, A=12, F=86, kind="mlp", H=64, feet=1, Hd=100, B=128, Ne=1024, E=2, M=4, clip=0.2, ecoef=0.0),
    # SplitPolicy (separate forward / backward kernels, state-dependent log-std)
    "split": dict(T=8, N_loc=8, O=14, A=7, F=25, kind="split", H=100, feet=1, Hd=100, B=32, Ne=320, E=2, M=2, clip=0.2, ecoef=0.01),
}


def prepare(world, name, seed=0):
    from oracle import oracle as orc
    hp = SHAPES[name]
    g = make_global(world, hp["T"], hp["N_loc"], hp["O"], hp["A"], hp["F"], hp["kind"], hp["H"], hp["feet"], hp["Ne"], seed)
    rng = np.random.default_rng(seed + 1)
    g["dpar"] = (0.2 * rng.standard_normal(orc.disc_num_params(hp["F"], hp["Hd"]))).astype(np.float32)
    TNg = hp["T"] * g["Ng"]
    g["ppo_perms"] = np.stack([rng.permutation(TNg) for _ in range(hp["E"])]).astype(np.int64)
    g["d_eperm"] = rng.permutation(hp["Ne"]).astype(np.int64)
    g["d_pperm"] = rng.permutation(TNg).astype(np.int64)
    g["alpha"] = rng.random((hp["Ne"] // hp["B"]) * hp["B"]).astype(np.float32)
    return hp, g


@pytest.mark.parametrize("world,name", [(2, "mlp_small"), (4, "mlp_small"), (8, "mlp_small"), (2, "northstar"), (4, "northstar"),
                                        (8, "northstar"), (2, "split"), (4, "split")])
def test_ppo_update_global_perms_equal_world1_oracle(world, name):
    """(i) PPO: the reference's permutations over the GLOBAL rollout, given to every rank, reproduce the single-process
    losses and weights; (ii) the advantage statistics are the global ones; every replica ends bit-identical."""
    import simgan_amd as sg
    from oracle import oracle as orc
    from helpers import assert_close
    hp, g = prepare(world, name)
    next_value = np.random.default_rng(5).standard_normal(g["Ng"]).astype(np.float32)
    ret, vp = oracle_returns(g, next_value)
    pi, adam = g["pi"].copy(), orc.AdamState(g["pi"].size)
    cfg = orc.ppo_cfg(hp["clip"], hp["E"], hp["M"], 0.5, hp["ecoef"], 3e-4, 1e-5, 0.5, True)
    want = orc.ppo_update(g["d"], pi, adam, cfg, g["obs"], g["actions"], vp, ret, g["logp"], g["ppo_perms"])
    adv_n = orc.advantages(ret[:-1], vp[:-1]).reshape(hp["T"], g["Ng"])   # normalised over the GLOBAL rollout (a2c/algo/ppo.py:66-68)

    def rank_fn(rank, ctx):
        pol, agent, _, ro = build_rank(sg, g, rank, ctx, hp)
        ro.compute_returns(next_value[rank * hp["N_loc"]:(rank + 1) * hp["N_loc"]], True, GAMMA, LAM, True)
        losses = agent.update(ro, perms=g["ppo_perms"])
        return dict(losses=losses, pi=pol.get_flat_params(), adv=ro.device_advantages().numpy()[..., 0],
                    ret=ro.returns.numpy()[..., 0].copy())

    res = run_ranks(world, rank_fn)
    for r in res[1:]:
        assert np.array_equal(res[0]["pi"], r["pi"]), "replicas diverged"
        assert r["losses"] == res[0]["losses"]
    assert_close(np.concatenate([r["ret"][:-1] for r in res], axis=1), ret[:-1], what="GAE returns (sharded columns)")
    assert_close(np.concatenate([r["adv"] for r in res], axis=1), adv_n, rtol=1e-4, atol=1e-5, what="globally normalised advantages")
    assert_close(res[0]["losses"], want, what=f"PPO losses, world {world}")
    assert_close(res[0]["pi"], pi, what=f"policy after the update, world {world}")
    assert np.max(np.abs(pi - g["pi"])) > 1e-4


@pytest.mark.parametrize("mode", ["replicated", "sharded"])
@pytest.mark.parametrize("world,name", [(2, "mlp_small"), (4, "mlp_small"), (8, "mlp_small"), (2, "northstar"), (8, "northstar"), (4, "split")])
def test_disc_update_global_perms_equal_world1_oracle(world, name, mode):
    """(i) D, both data-parallel modes: the reference's draws over the GLOBAL rollout (DataLoader shuffle, policy-row
    permutation in (t, n_global) numbering, mixup alpha) reproduce the single-process losses / weights over two epochs.
    Replicated mode: this pins the numbering of the all-gathered union.  Sharded mode: rank slicing by row ownership."""
    import simgan_amd as sg
    from oracle import oracle as orc
    from helpers import assert_close
    hp, g = prepare(world, name)
    rng = np.random.default_rng(9)
    draws = [(g["d_eperm"], g["d_pperm"], g["alpha"]),
             (rng.permutation(hp["Ne"]).astype(np.int64), rng.permutation(hp["T"] * g["Ng"]).astype(np.int64),
              rng.random(g["alpha"].size).astype(np.float32))]
    dpar, dadam = g["dpar"].copy(), orc.AdamState(g["dpar"].size)
    want = [orc.disc_update(hp["F"], hp["Hd"], dpar, dadam, g["expert"], g["obs_feat"], hp["B"], *dr) for dr in draws]

    def rank_fn(rank, ctx):
        ctx.set_disc_dp(mode == "sharded")
        _, _, disc, ro = build_rank(sg, g, rank, ctx, hp)
        out = [disc.update_gail_dyn(Loader(g["expert"], hp["B"]), ro, expert_perm=dr[0], policy_perm=dr[1], alpha=dr[2]) for dr in draws]
        return dict(losses=out, n=disc.last_n_steps, dpar=disc.get_flat_params(), adam=disc.get_adam())

    res = run_ranks(world, rank_fn)
    for r in res[1:]:
        assert np.array_equal(res[0]["dpar"], r["dpar"]), "replicas diverged"
    assert res[0]["n"] == want[-1][1] == min(hp["Ne"] // hp["B"], hp["T"] * g["Ng"] // hp["B"])
    for e in range(2):
        assert_close(res[0]["losses"][e], want[e][0], what=f"D losses epoch {e}, world {world}, {mode}")
    assert_close(res[0]["dpar"], dpar, what=f"D weights, world {world}, {mode}")
    assert res[0]["adam"][2] == dadam.t.value
    assert_close(res[0]["adam"][0], dadam.m, rtol=1e-3, atol=1e-6, what="D Adam m")


@pytest.mark.parametrize("world", [2, 4])
def test_replicated_disc_reuses_the_gathered_rows_until_the_rollout_changes(world):
    """Replicated mode gathers every rank's policy rows once per LEARNER UPDATE, not once per epoch: the gail_epoch calls of an
    update read the same device-resident rollout.  Three epochs on rollout A (one all-gather), then the rollout's obs_feat is
    replaced on the device (another all-gather must happen, the stale union must not be used), two epochs on rollout B -- all
    five epochs against the single-process oracle, which sees A, A, A, B, B."""
    import simgan_amd as sg
    from simgan_amd import _lib
    from oracle import oracle as orc
    from helpers import assert_close
    hp, g = prepare(world, "northstar", seed=5)
    rng = np.random.default_rng(10)
    featB = rng.standard_normal(g["obs_feat"].shape).astype(np.float32)
    draws = [(rng.permutation(hp["Ne"]).astype(np.int64), rng.permutation(hp["T"] * g["Ng"]).astype(np.int64),
              rng.random(g["alpha"].size).astype(np.float32)) for _ in range(5)]
    dpar, dadam = g["dpar"].copy(), orc.AdamState(g["dpar"].size)
    want = [orc.disc_update(hp["F"], hp["Hd"], dpar, dadam, g["expert"], g["obs_feat"] if e < 3 else featB, hp["B"], *draws[e]) for e in range(5)]

    def rank_fn(rank, ctx):
        ctx.set_disc_dp(False)
        _, _, disc, ro = build_rank(sg, g, rank, ctx, hp)
        ro.sync_to_device()
        ro.device_resident = True
        loader = Loader(g["expert"], hp["B"])
        gathers = C.c_longlong(0)
        hook = _lib.load_test().sg_test_disc_gathers
        out, counts = [], []
        for e in range(5):
            if e == 3:   # a new rollout: the host mirror is rewritten and pushed (what collect() does step by step)
                ro.obs_feat.copy_(ro.obs_feat.new_tensor(shard(featB, rank, hp["N_loc"]).reshape(tuple(ro.obs_feat.shape))))
                ro.sync_to_device([_lib.F_OBS_FEAT])
            out.append(disc.update_gail_dyn(loader, ro, expert_perm=draws[e][0], policy_perm=draws[e][1], alpha=draws[e][2]))
            _lib.check_test(hook(disc.h, C.byref(gathers)))
            counts.append(gathers.value)
        return dict(losses=out, counts=counts, dpar=disc.get_flat_params())

    res = run_ranks(world, rank_fn)
    for r in res:
        assert r["counts"] == [1, 1, 1, 2, 2], r["counts"]
        assert np.array_equal(res[0]["dpar"], r["dpar"]), "replicas diverged"
    for e in range(5):
        assert_close(res[0]["losses"][e], want[e][0], what=f"D losses epoch {e}, world {world}")
    assert_close(res[0]["dpar"], dpar, what=f"D weights after 5 epochs, world {world}")


@pytest.mark.parametrize("world", [2, 4, 8])
def test_relabel_statistics_and_count_dones_equal_world1_oracle(world):
    """(ii) the per-step return statistics of the reward relabel (float64 RunningMeanStd merge over the GLOBAL batch),
    Discriminator.returns carried across two calls, and count_dones, against the oracle on the concatenated rollout."""
    import simgan_amd as sg
    from simgan_amd import _lib
    from oracle import oracle as orc
    from helpers import assert_close
    hp, g = prepare(world, "northstar")
    g2 = make_global(world, hp["T"], hp["N_loc"], hp["O"], hp["A"], hp["F"], "mlp", hp["H"], 1, 0, seed=3)
    rms0 = [0.0, 1.0, 1e-4]
    w1, dret, rms1 = orc.relabel(hp["F"], hp["Hd"], g["dpar"], g["obs_feat"], g["masks"], GAMMA, -0.3, None, rms0)
    w2, dret2, rms2 = orc.relabel(hp["F"], hp["Hd"], g["dpar"], g2["obs_feat"], g2["masks"], GAMMA, 0.1, dret, rms1)

    def rank_fn(rank, ctx):
        _, _, disc, ro = build_rank(sg, g, rank, ctx, hp)
        rms = sg.RunningMeanStd(shape=())
        disc.relabel_rewards(ro, GAMMA, -0.3, rms)
        r1 = ro.rewards.numpy()[..., 0].copy()
        ro.sync_to_device()
        dones = C.c_double(0)
        _lib.check(ctx.lib.sg_rollout_count_dones(ro.h, C.byref(dones)))
        put = lambda dst, src: dst.copy_(dst.new_tensor(np.ascontiguousarray(src).reshape(tuple(dst.shape))))  # noqa: E731
        put(ro.obs_feat, shard(g2["obs_feat"], rank, hp["N_loc"])); put(ro.masks, shard(g2["masks"], rank, hp["N_loc"]))
        disc.relabel_rewards(ro, GAMMA, 0.1, rms)
        return dict(r1=r1, r2=ro.rewards.numpy()[..., 0].copy(), rms=rms.get_state(), dones=dones.value,
                    dret=disc.returns.numpy()[:, 0].copy())

    res = run_ranks(world, rank_fn)
    assert_close(np.concatenate([r["r1"] for r in res], axis=1), w1, what="relabelled rewards, first call")
    assert_close(np.concatenate([r["r2"] for r in res], axis=1), w2, what="relabelled rewards, second call (returns and ret_rms carried)")
    assert_close(np.concatenate([r["dret"] for r in res]), dret2, what="Discriminator.returns")
    for r in res:
        assert_close(r["rms"], rms2, rtol=1e-6, atol=1e-9, what="ret_rms state")
        assert r["dones"] == float((1.0 - g["masks"]).sum()), (r["dones"], float((1.0 - g["masks"]).sum()))


def test_config4_per_rank_shape_prefix_world8():
    """(iii) BASELINE.json configs[3]: LaikagoCombinedEnv-v1, 4096 envs sharded 8 ways -> 512 columns per rank, 100k expert
    rows: n_d = min(100000 / 128, 128 * 4096 / 128) = 781 discriminator steps per epoch over the global rollout (replicated
    mode, the default), then a 64-step prefix in sharded mode, then one PPO epoch of the SplitPolicy (h100, 4 feet) on
    32,768-row global minibatches.  All against the oracle on the concatenated 4096-column rollout."""
    import simgan_amd as sg
    from oracle import oracle as orc
    from helpers import assert_close
    world, T, N_loc, O, A, F, H, Hd, B, Ne = 8, 128, 512, 64, 28, 86, 100, 100, 128, 100000
    hp = dict(T=T, N_loc=N_loc, O=O, A=A, F=F, kind="split", H=H, feet=4, Hd=Hd, B=B, Ne=Ne, E=1, M=16, clip=0.2, ecoef=0.0)
    g = make_global(world, T, N_loc, O, A, F, "split", H, 4, Ne, seed=2)
    rng = np.random.default_rng(4)
    g["dpar"] = (0.1 * rng.standard_normal(orc.disc_num_params(F, Hd))).astype(np.float32)
    TNg = T * g["Ng"]
    eperm, pperm = rng.permutation(Ne).astype(np.int64), rng.permutation(TNg).astype(np.int64)
    alpha = rng.random((Ne // B) * B).astype(np.float32)
    ppo_perms = rng.permutation(TNg).astype(np.int64)[None]
    n_prefix = 64
    next_value = rng.standard_normal(g["Ng"]).astype(np.float32)

    dpar, dadam = g["dpar"].copy(), orc.AdamState(g["dpar"].size)
    want_rep, n_d = orc.disc_update(F, Hd, dpar, dadam, g["expert"], g["obs_feat"], B, eperm, pperm, alpha)
    assert n_d == 781
    dpar_s, dadam_s = g["dpar"].copy(), orc.AdamState(g["dpar"].size)
    want_sh, _ = orc.disc_update(F, Hd, dpar_s, dadam_s, g["expert"][eperm[:n_prefix * B]], g["obs_feat"], B,
                                 np.arange(n_prefix * B), pperm, alpha)
    ret, vp = oracle_returns(g, next_value)
    pi, adam = g["pi"].copy(), orc.AdamState(g["pi"].size)
    cfg = orc.ppo_cfg(0.2, 1, 16, 0.5, 0.0, 3e-4, 1e-5, 0.5, True)
    want_ppo = orc.ppo_update(g["d"], pi, adam, cfg, g["obs"], g["actions"], vp, ret, g["logp"], ppo_perms)

    def rank_fn(rank, ctx):
        pol, agent, disc, ro = build_rank(sg, g, rank, ctx, hp)
        rep = disc.update_gail_dyn(Loader(g["expert"], B), ro, expert_perm=eperm, policy_perm=pperm, alpha=alpha)
        out = dict(rep=rep, n=disc.last_n_steps, dpar_rep=disc.get_flat_params())
        # sharded prefix: a 64-batch expert set ends the epoch after 64 steps (n_d = min(n_expert / B, ...))
        ctx.set_disc_dp(True)
        disc2 = sg.algo.gail.Discriminator(F, Hd, None, ctx=ctx, seed=11)
        disc2.set_flat_params(g["dpar"])
        sh = disc2.update_gail_dyn(Loader(g["expert"][eperm[:n_prefix * B]], B), ro, expert_perm=np.arange(n_prefix * B),
                                   policy_perm=pperm, alpha=alpha)
        out.update(sh=sh, n_sh=disc2.last_n_steps, dpar_sh=disc2.get_flat_params())
        ro.compute_returns(next_value[rank * N_loc:(rank + 1) * N_loc], True, GAMMA, LAM, True)
        out.update(ppo=agent.update(ro, perms=ppo_perms), pi=pol.get_flat_params())
        return out

    res = run_ranks(world, rank_fn, timeout_s=1500)
    for r in res[1:]:
        for k in ("dpar_rep", "dpar_sh", "pi"):
            assert np.array_equal(res[0][k], r[k]), f"replicas diverged ({k})"
    assert res[0]["n"] == 781 and res[0]["n_sh"] == n_prefix
    assert_close(res[0]["rep"], want_rep, what="D losses, 781 replicated steps")
    assert_close(res[0]["dpar_rep"], dpar, what="D weights after 781 replicated steps")
    assert_close(res[0]["sh"], want_sh, what="D losses, 64 sharded steps")
    assert_close(res[0]["dpar_sh"], dpar_s, what="D weights after 64 sharded steps")
    assert_close(res[0]["ppo"], want_ppo, what="PPO losses, 16 steps on 32,768-row global minibatches")
    assert_close(res[0]["pi"], pi, what="SplitPolicy (Laikago) after the PPO epoch")


def test_config5_per_rank_shape_world8():
    """BASELINE.json configs[4]: Laikago policy refinement (a2c/main.py caller), 2048 envs sharded 8 ways -> 256 columns per
    rank, obs 111 / act 12 / Policy h64, 8 minibatches, clip 0.1, lr 1.5e-4: one PPO epoch (8 steps on 32,768-row global
    minibatches, the reference's permutation over the 262,144 global rows) against the oracle on the concatenated rollout."""
    import simgan_amd as sg
    from oracle import oracle as orc
    from helpers import assert_close
    world, T, N_loc, O, A, H = 8, 128, 256, 111, 12, 64
    hp = dict(T=T, N_loc=N_loc, O=O, A=A, F=4, kind="mlp", H=H, feet=1, Hd=16, B=8, Ne=0, E=1, M=8, clip=0.1, ecoef=0.0)
    g = make_global(world, T, N_loc, O, A, 4, "mlp", H, 1, 0, seed=6)
    rng = np.random.default_rng(8)
    perms = rng.permutation(T * g["Ng"]).astype(np.int64)[None]
    next_value = rng.standard_normal(g["Ng"]).astype(np.float32)
    ret, vp = oracle_returns(g, next_value)
    pi, adam = g["pi"].copy(), orc.AdamState(g["pi"].size)
    cfg = orc.ppo_cfg(0.1, 1, 8, 0.5, 0.0, 1.5e-4, 1e-5, 0.5, True)
    want = orc.ppo_update(g["d"], pi, adam, cfg, g["obs"], g["actions"], vp, ret, g["logp"], perms)

    def rank_fn(rank, ctx):
        pol = sg.Policy((O,), Box((A,)), base_kwargs={"recurrent": False, "hidden_size": H}, ctx=ctx)
        pol.set_flat_params(g["pi"])
        agent = sg.algo.PPO(pol, 0.1, 1, 8, 0.5, 0.0, lr=1.5e-4, eps=1e-5, max_grad_norm=0.5)
        ro = sg.RolloutStorage(T, N_loc, (O,), Box((A,)), 1, 4, ctx=ctx)
        put = lambda dst, src: dst.copy_(dst.new_tensor(np.ascontiguousarray(src).reshape(tuple(dst.shape))))  # noqa: E731
        for name, key in (("obs", "obs"), ("actions", "actions"), ("action_log_probs", "logp"), ("value_preds", "value_preds"),
                          ("masks", "masks"), ("bad_masks", "bad_masks"), ("rewards", "rewards")):
            put(getattr(ro, name), shard(g[key], rank, N_loc))
        ro.compute_returns(next_value[rank * N_loc:(rank + 1) * N_loc], True, GAMMA, LAM, True)
        return dict(losses=agent.update(ro, perms=perms), pi=pol.get_flat_params())

    res = run_ranks(world, rank_fn, timeout_s=900)
    for r in res[1:]:
        assert np.array_equal(res[0]["pi"], r["pi"]), "replicas diverged"
    assert_close(res[0]["losses"], want, what="PPO losses, refinement shape, world 8")
    assert_close(res[0]["pi"], pi, what="policy after the epoch, refinement shape, world 8")
    assert np.max(np.abs(pi - g["pi"])) > 1e-4


@pytest.mark.parametrize("mode", ["replicated", "sharded"])
def test_library_rng_update_world2_replays_through_the_oracle(mode):
    """The production path for N > 1 (no injected draws: the library's generator, device-resident rollouts, the drivers'
    update()): two ranks, one full GailDynLearner.update() each.  The draws every phase consumed are read back and the
    same update is replayed through the oracle on the concatenated rollout."""
    import simgan_amd as sg
    from simgan_amd import _lib
    from simgan_amd.driver import ExpertLoader, GailDynLearner
    from oracle import oracle as orc
    from helpers import assert_close, assert_close_adam
    world, name = 2, "northstar"
    hp, g = prepare(world, name)
    T, N, Ng, F, Hd, B = hp["T"], hp["N_loc"], g["Ng"], hp["F"], hp["Hd"], hp["B"]
    E_d = 2

    def rank_fn(rank, ctx):
        ctx.set_disc_dp(mode == "sharded")
        pol, agent, disc, ro = build_rank(sg, g, rank, ctx, hp)
        agent.seed, disc.seed = 1234, 77          # the streams that must agree across ranks are seeded explicitly
        ro.sync_to_device()
        ro.device_resident = True
        learner = GailDynLearner(pol, agent, disc, ro, ExpertLoader(g["expert"], B), gail_batch_size=B, gail_epoch=E_d,
                                 gamma=GAMMA, gae_lambda=LAM, gail_tar_length=50.0)
        draws = []
        orig = disc.update_gail_dyn

        def spy(loader, rollouts, **k):
            out = orig(loader, rollouts, **k)
            draws.append(disc.last_draws())
            return out
        disc.update_gail_dyn = spy
        info = dict(learner.update())           # read the results ring inside the rank's own thread
        ro.sync_from_device()
        return dict(info=info, draws=draws, perms=agent.last_perms(), pi=pol.get_flat_params(), dpar=disc.get_flat_params(),
                    rewards=ro.rewards.numpy()[..., 0].copy(), returns=ro.returns.numpy()[..., 0].copy(), rms=learner.ret_rms.get_state())

    res = run_ranks(world, rank_fn)
    for k in ("pi", "dpar"):
        assert np.array_equal(res[0][k], res[1][k]), f"replicas diverged ({k})"
    tn_loc = T * N

    def to_global(local_rows, rank):
        t, n = local_rows // N, local_rows % N
        return t * Ng + rank * N + n

    # ---- discriminator epochs
    dpar, dadam = g["dpar"].copy(), orc.AdamState(g["dpar"].size)
    d_losses = None
    for e in range(E_d):
        ep, pp0, al = res[0]["draws"][e]
        assert np.array_equal(ep, res[1]["draws"][e][0]) and np.array_equal(al, res[1]["draws"][e][2]), "expert / alpha streams must be global"
        if mode == "replicated":     # one global permutation, identical on both ranks, in the reference's numbering
            assert np.array_equal(pp0, res[1]["draws"][e][1]) and sorted(pp0) == list(range(T * Ng))
            pperm = pp0
        else:                        # per-rank local permutations, B/world rows per rank per step
            b_loc = B // world
            n_d = min(hp["Ne"] // B, tn_loc // b_loc)
            pperm = np.zeros(T * Ng, np.int64)
            used = []
            for k in range(n_d):
                for r in range(world):
                    rows = to_global(res[r]["draws"][e][1][k * b_loc:(k + 1) * b_loc], r)
                    pperm[k * B + r * b_loc:k * B + (r + 1) * b_loc] = rows
                    used.append(rows)
            rest = np.setdiff1d(np.arange(T * Ng), np.concatenate(used))
            pperm[n_d * B:] = rest[:T * Ng - n_d * B]
        d_losses, _ = orc.disc_update(F, Hd, dpar, dadam, g["expert"], g["obs_feat"], B, ep, pperm, al)
    assert_close([res[0]["info"][k] for k in ("gail_loss", "gail_loss_e", "gail_loss_p")], d_losses, what=f"D losses ({mode}, library draws)")
    assert_close_adam(res[0]["dpar"], dpar, 1e-3, dadam.t.value, what=f"D weights ({mode}, library draws)")
    # ---- offset, relabel, returns
    r_sa = orc.alive_bonus(g["masks"], T, Ng, 50.0)
    assert abs(res[0]["info"]["r_sa"] - r_sa) <= 1e-6 * max(1.0, abs(r_sa))
    rewards, _, rms = orc.relabel(F, Hd, dpar, g["obs_feat"], g["masks"], GAMMA, -r_sa, None, [0.0, 1.0, 1e-4])
    assert_close(np.concatenate([r["rewards"] for r in res], axis=1), rewards, what="relabelled rewards")
    assert_close(res[0]["rms"], rms, rtol=1e-6, atol=1e-9, what="ret_rms")
    nv = orc.policy_forward(g["d"], g["pi"], g["obs"][T])[0][:, 0]
    ret, vp = orc.compute_returns(rewards, g["value_preds"], g["masks"], g["bad_masks"], nv, 1, GAMMA, LAM, 1)
    assert_close(np.concatenate([r["returns"][:-1] for r in res], axis=1), ret[:-1], what="GAE returns")
    # ---- PPO: the single-process permutation that draws the same minibatches (mb_loc rows of each rank per step)
    M, E = hp["M"], hp["E"]
    mb_loc = tn_loc // M
    perms = np.zeros((E, T * Ng), np.int64)
    for e in range(E):
        used = [to_global(res[r]["perms"][e, k * mb_loc:(k + 1) * mb_loc], r) for k in range(M) for r in range(world)]
        used = np.concatenate(used)
        perms[e] = np.concatenate([used, np.setdiff1d(np.arange(T * Ng), used)])
    pi, adam = g["pi"].copy(), orc.AdamState(g["pi"].size)
    cfg = orc.ppo_cfg(hp["clip"], E, M, 0.5, hp["ecoef"], 3e-4, 1e-5, 0.5, True)
    want = orc.ppo_update(g["d"], pi, adam, cfg, g["obs"], g["actions"], vp, ret, g["logp"], perms)
    assert_close([res[0]["info"][k] for k in ("value_loss", "action_loss", "dist_entropy")], want, what="PPO losses (library draws)")
    assert_close(res[0]["pi"], pi, what="policy (library draws)")


def test_injected_draws_are_validated_behind_the_c_abi():
    """Element counts and index ranges of every injected array are checked by the library itself (include/simgan_hip.h),
    not only by the Python shim: call the C entry points directly with short / out-of-range arrays."""
    import simgan_amd as sg
    from simgan_amd import _lib
    hp, g = prepare(1, "mlp_small")
    ctx = _lib.Context.default()
    pol, agent, disc, ro = build_rank(sg, g, 0, ctx, hp)
    disc.set_expert(g["expert"])
    ro.sync_to_device()
    out, nst = (C.c_float * 3)(), C.c_int(0)
    TN = hp["T"] * hp["N_loc"]
    good_e, good_p = np.arange(hp["Ne"], dtype=np.int64), np.arange(TN, dtype=np.int64)
    al = np.zeros(hp["Ne"], np.float32)

    def d_call(ep, n_ep, pp, n_pp, a, n_a):
        return ctx.lib.sg_disc_update_gail_dyn(disc.h, ro.h, hp["B"], _lib.i64ptr(ep), n_ep, _lib.i64ptr(pp), n_pp, _lib.fptr(a), n_a,
                                               1, out, C.byref(nst))
    assert d_call(good_e, good_e.size - 1, good_p, good_p.size, al, al.size) != 0 and b"expert_perm holds" in ctx.lib.sg_last_error()
    assert d_call(good_e, good_e.size, good_p, good_p.size + 5, al, al.size) != 0 and b"policy_perm holds" in ctx.lib.sg_last_error()
    bad = good_p.copy(); bad[3] = TN
    assert d_call(good_e, good_e.size, bad, bad.size, al, al.size) != 0 and b"policy_perm[3]" in ctx.lib.sg_last_error()
    assert d_call(good_e, good_e.size, good_p, good_p.size, al, 3) != 0 and b"alpha holds 3" in ctx.lib.sg_last_error()
    assert d_call(good_e, good_e.size, good_p, good_p.size, al, al.size) == 0
    perms = np.stack([np.arange(TN, dtype=np.int64)] * hp["E"])
    assert ctx.lib.sg_ppo_update(agent.h, ro.h, _lib.i64ptr(perms), perms.size - 1, 1, out) != 0 and b"perms holds" in ctx.lib.sg_last_error()
    bad = perms.copy(); bad[1, 2] = -1
    assert ctx.lib.sg_ppo_update(agent.h, ro.h, _lib.i64ptr(bad), bad.size, 1, out) != 0 and b"is outside" in ctx.lib.sg_last_error()
    assert ctx.lib.sg_ppo_update(agent.h, ro.h, _lib.i64ptr(perms), perms.size, 1, out) == 0


@pytest.mark.parametrize("workload,gpus", [("northstar", 2), ("refine", 2)])
def test_bench_launches_its_own_ranks_and_reports_the_n_gpu_line(workload, gpus):
    """`python bench.py --gpus N` with NO launcher around it (how the driver ran N = 1 in round 2) must start the N ranks
    itself and print exactly one JSON line.  On this one-GPU box the ranks share the device over the loopback communicator
    (--loopback); everything else -- torch.distributed.run, the gloo control plane, per-rank timing, max over ranks, both
    discriminator modes, the comm block of the line -- is the path the 8-GPU run takes."""
    import json
    import subprocess
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", SG_LOOPBACK_TIMEOUT_S="120")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "bench.py", "--gpus", str(gpus), "--loopback", "--steps", "2", "--warmup", "1",
                        "--workload", workload, "--no-cpu-baseline"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-6000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    assert out["n_gpus"] == gpus and out["steps"] == 2 and out["scaling"] == "weak" and out["value"] > 0
    assert len(out["per_rank_ms_per_step"]) == gpus
    assert out["comm"]["kind"] == "loopback" and out["comm"]["nranks_reported_by_rccl"] == gpus
    assert all(np.isfinite(v) for v in out["last_losses"].values())
    if workload == "northstar":
        assert out["comm"]["disc_mode"] == "replicated"
        alt = out["comm"]["disc_other_mode"]
        assert alt["mode"] == "sharded" and "error" not in alt and alt["value"] > 0, alt
        assert out["config"]["optimizer_steps_per_update"] == 5 * 781 + 160 or gpus != 8   # n_d follows the GLOBAL rollout


def test_bench_self_launch_propagates_a_failing_rank():
    """A rank that dies must fail the whole command (non-zero exit), not leave a hung or silently truncated run."""
    import subprocess
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", SG_LOOPBACK_TIMEOUT_S="20", SG_BENCH_FAIL_RANK="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--loopback", "--steps", "1", "--warmup", "0", "--no-cpu-baseline",
                        "--no-other-disc-mode", "--no-other-allreduce"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    # ... and the launcher's SIGTERM to the surviving rank 0 must not lose the line: it says what happened and where
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) <= 1, lines
    if lines:
        import json
        out = json.loads(lines[0])
        assert out["value"] is None and "error" in out and "stage" in out and set(out["ranks"]) == {"0", "1"}, out


def test_bench_watchdog_reports_a_rank_that_never_arrives():
    """First-run insurance for the 8-GPU launch (round-3 verdict #6): one of 8 ranks hangs before it joins the communicator
    (the hook stands in for a stuck ncclCommInitRank) -- every other rank then blocks in communicator set-up.  The watchdog
    must end the run inside its budget with ONE JSON line that names the error, the stage, and each rank's last stage."""
    import json
    import subprocess
    import time
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", SG_LOOPBACK_TIMEOUT_S="600", SG_BENCH_HANG_RANK="5")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    t0 = time.time()
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "8", "--loopback", "--steps", "1", "--warmup", "1", "--no-cpu-baseline",
                        "--no-other-disc-mode", "--no-other-allreduce", "--init-timeout", "30"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    took = time.time() - t0
    assert r.returncode != 0 and took < 240, (r.returncode, took, r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, (lines, r.stderr[-3000:])
    out = json.loads(lines[0])
    assert out["value"] is None and out["n_gpus"] == 8 and "watchdog" in out["error"], out
    assert "TEST HOOK" in out["ranks"]["5"], out["ranks"]
    assert sum("communicator" in v or "process group" in v for v in out["ranks"].values()) >= 6, out["ranks"]


def test_bench_replica_check_is_reported(tmp_path):
    """N > 1: after the warm-up every rank hashes its policy and discriminator weights; the line carries the verdict."""
    import json
    import subprocess
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", SG_LOOPBACK_TIMEOUT_S="120")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--loopback", "--steps", "1", "--warmup", "1", "--no-cpu-baseline",
                        "--no-other-disc-mode", "--no-other-allreduce"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert out["replica_check"]["ok"] is True and out["replica_check"]["ranks"] == 2, out["replica_check"]


def test_bench_line_survives_the_death_of_the_bench_process_after_the_headline():
    """After the headline measurement the line is in the keeping of a helper process (bench.LineKeeper): the bench process is
    killed outright (SIGKILL, the hook stands in for a GPU fault or an out-of-memory kill in a later leg) and ONE line still comes
    out, with the measured value and `error_after_headline`."""
    import json
    import subprocess
    env = dict(os.environ, SG_BENCH_DIE_AFTER_HEADLINE="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "bench.py", "--workload", "refine", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode != 0, "the hook kills the process"
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, (lines, r.stderr[-2000:])
    out = json.loads(lines[0])
    assert out["value"] > 0 and out["ms_per_step"] > 0 and "error_after_headline" in out, out
    # ... and without the hook the same command prints the same one line, without the mark
    env.pop("SG_BENCH_DIE_AFTER_HEADLINE")
    r = subprocess.run([sys.executable, "bench.py", "--workload", "refine", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=600)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1 and "error_after_headline" not in json.loads(lines[0]), (r.returncode, lines, r.stderr[-2000:])


def test_bench_line_survives_a_peer_rank_dying_after_the_headline():
    """N > 1: rank 1 is killed after the headline (in what would be the other-mode / peer-mesh legs); the launcher ends rank 0,
    whose watchdog prints the line it had -- the measured value with `error_after_headline`, not an empty line."""
    import json
    import subprocess
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", SG_LOOPBACK_TIMEOUT_S="60", SG_BENCH_DIE_AFTER_HEADLINE="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--loopback", "--workload", "refine", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, (lines, r.stderr[-3000:])
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["value"] > 0 and "error_after_headline" in out, out
