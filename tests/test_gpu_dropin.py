"""Drop-in mode (host tensors are the rollout, simgan_amd/storage.py): a field crosses PCIe when its host tensor changed
since the device copy last matched it -- once per change, not once per call -- and an edit made by slicing between two
calls, as the reference main makes them (a2c/main_gail_dyn_ppo.py:276-292), is picked up by the next device call."""
import numpy as np
import pytest

from helpers import assert_close

pytestmark = pytest.mark.gpu


class Box:
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


def _filled(sg, T=16, N=8, O=5, A=3, F=7, seed=3):
    rng = np.random.default_rng(seed)
    ro = sg.RolloutStorage(T, N, (O,), Box((A,)), 1, F)
    ro.rewards.copy_(ro.rewards.new_tensor(rng.standard_normal((T, N, 1)).astype(np.float32)))
    ro.value_preds.copy_(ro.value_preds.new_tensor(rng.standard_normal((T + 1, N, 1)).astype(np.float32)))
    ro.masks.copy_(ro.masks.new_tensor((rng.random((T + 1, N, 1)) > 0.1).astype(np.float32)))
    ro.bad_masks.copy_(ro.bad_masks.new_tensor((rng.random((T + 1, N, 1)) > 0.05).astype(np.float32)))
    ro.obs_feat.copy_(ro.obs_feat.new_tensor(rng.standard_normal((T + 1, N, F)).astype(np.float32)))
    return ro, rng


def _oracle_returns(orc, ro, nv):
    vp = ro.value_preds.numpy()[..., 0].copy()
    ret, _ = orc.compute_returns(ro.rewards.numpy()[..., 0], vp, ro.masks.numpy()[..., 0], ro.bad_masks.numpy()[..., 0], nv, 1, 0.99, 0.95, 1)
    return ret


def test_slice_edit_between_two_calls_reaches_the_device(sg, orc):
    ro, rng = _filled(sg)
    T, N = ro.num_steps, ro.num_processes
    nv = rng.standard_normal(N).astype(np.float32)
    want = _oracle_returns(orc, ro, nv)
    ro.compute_returns(nv, True, 0.99, 0.95, True)
    assert_close(ro.returns.numpy()[:T, :, 0], want[:T], rtol=1e-5, what="returns, first call")
    first = ro.bytes_uploaded
    assert first == 4 * (T * N + 4 * (T + 1) * N), "the first call uploads the five fields it reads, once"
    # nothing changed: nothing crosses PCIe, same result
    ro.compute_returns(nv, True, 0.99, 0.95, True)
    assert ro.bytes_uploaded == first, "an unchanged rollout must not be uploaded again"
    assert_close(ro.returns.numpy()[:T, :, 0], want[:T], rtol=1e-5, what="returns, unchanged rollout")
    # the main's own kind of edit: assignment to a slice (a2c/main_gail_dyn_ppo.py:291)
    ro.rewards[3] = ro.rewards[3] + 1.5
    want2 = _oracle_returns(orc, ro, nv)
    assert np.abs(want2[:4] - want[:4]).max() > 0.5
    ro.compute_returns(nv, True, 0.99, 0.95, True)
    assert ro.bytes_uploaded == first + 4 * T * N, "exactly the edited field is uploaded"
    assert_close(ro.returns.numpy()[:T, :, 0], want2[:T], rtol=1e-5, what="returns after rollouts.rewards[3] was edited by slicing")
    # .copy_ into a slice, and a re-bound attribute
    ro.masks[5].copy_(ro.masks[5] * 0.0)
    ro.value_preds = ro.value_preds.clone() * 0.5
    want3 = _oracle_returns(orc, ro, nv)
    ro.compute_returns(nv, True, 0.99, 0.95, True)
    assert_close(ro.returns.numpy()[:T, :, 0], want3[:T], rtol=1e-5, what="returns after masks[5].copy_ and a re-bound value_preds")


def test_write_through_a_numpy_view_needs_mark_host_written(sg, orc):
    ro, rng = _filled(sg, seed=4)
    T, N = ro.num_steps, ro.num_processes
    nv = rng.standard_normal(N).astype(np.float32)
    ro.compute_returns(nv, True, 0.99, 0.95, True)
    ro.rewards.numpy()[2] += 2.0            # torch's version counter does not see this
    want = _oracle_returns(orc, ro, nv)
    ro.mark_host_written()
    ro.compute_returns(nv, True, 0.99, 0.95, True)
    assert_close(ro.returns.numpy()[:T, :, 0], want[:T], rtol=1e-5, what="returns after mark_host_written()")


def test_discriminator_epochs_upload_obs_feat_once(sg):
    """gail_epoch x update_gail_dyn + the relabel on one rollout (a2c/main_gail_dyn_ppo.py:255-292): obs_feat crosses once."""
    T, N, F, Hd, B = 16, 8, 7, 16, 8
    ro, rng = _filled(sg, T=T, N=N, F=F, seed=5)
    expert = rng.standard_normal((64, F)).astype(np.float32)

    class Loader:
        def __init__(self, e, b):
            self.expert, self.batch_size = e, b

    def run(always):
        import os
        os.environ["SG_ROLLOUT_ALWAYS_UPLOAD"] = "1" if always else "0"
        try:
            D = sg.algo.gail.Discriminator(F, Hd, None, seed=2)
            ro.mark_host_written()
            b0 = ro.bytes_uploaded
            losses = [D.update_gail_dyn(Loader(expert, B), ro) for _ in range(3)]
            rms = sg.RunningMeanStd(shape=())
            D.relabel_rewards(ro, 0.99, 0.25, rms)
            return losses, ro.rewards.numpy().copy(), D.get_flat_params(), ro.bytes_uploaded - b0
        finally:
            os.environ.pop("SG_ROLLOUT_ALWAYS_UPLOAD", None)

    l1, r1, p1, up1 = run(False)
    l2, r2, p2, up2 = run(True)
    feat_bytes, mask_bytes = 4 * (T + 1) * N * F, 4 * (T + 1) * N
    assert up1 == feat_bytes + mask_bytes, (up1, feat_bytes, mask_bytes)
    assert up2 == 4 * feat_bytes + mask_bytes
    assert l1 == l2 and np.array_equal(r1, r2) and np.array_equal(p1, p2), "tracking what changed must not change any result"


def test_literal_main_sequence_equals_the_resident_learner(sg):
    """The unchanged main's own call sequence on HOST tensors (a2c/main_gail_dyn_ppo.py:239-304 -- what bench.py's `dropin.literal_main`
    times: gail_epoch x update_gail_dyn, T x [predict_reward_combined + host ret_rms.update + clip], compute_returns, agent.update,
    after_update) against GailDynLearner.update() on a device-resident twin (fused on-device relabel, nothing uploaded): same
    seeds, same rollout, two updates.  The discriminator sees identical rows in both, so its weights must be EQUAL; rewards,
    returns and the policy agree at the float32 tolerance (the relabel's running statistics are float64 on the host in one and
    on the device in the other)."""
    import bench
    import torch
    from simgan_amd import _lib
    from simgan_amd.driver import alive_bonus_offset
    from simgan_amd.utils import RunningMeanStd
    w = dict(kind="split", T=16, N=32, O=14, A=7, F=25, H=100, feet=1, Hd=100, E_p=2, M=4, E_d=2, B=128, Ne=512, clip=0.2)
    T, N = w["T"], w["N"]
    ctx = _lib.Context.default()

    def build():
        pol, disc, agent, ro, loader, expert, learner = bench.build_problem(sg, w, seed=5)
        _lib.check(ctx.lib.sg_rollout_fill_synthetic(ro.h, pol.h, 77, 0.03))
        return pol, disc, agent, ro, loader, learner

    # (a) resident learner
    pol_r, disc_r, agent_r, ro_r, loader_r, learner_r = build()
    outs_r = [dict(learner_r.update().resolve()) for _ in range(2)]
    ro_r.sync_from_device()
    # (b) the literal sequence on host tensors
    pol, disc, agent, ro, loader, _ = build()
    ro.sync_from_device()
    ro.device_resident = False
    ret_rms = RunningMeanStd(shape=())
    outs = []
    for _ in range(2):
        with torch.no_grad():
            next_value = pol.get_value(ro.obs[-1], ro.recurrent_hidden_states[-1], ro.masks[-1]).detach()
        for _e in range(w["E_d"]):
            gail_loss, gail_loss_e, gail_loss_p = disc.update_gail_dyn(loader, ro)
        num_of_dones = float((1.0 - ro.masks).sum().cpu().numpy())
        r_sa = alive_bonus_offset(num_of_dones, T, N, 500.0)
        for step in range(T):
            ro.rewards[step], returns = disc.predict_reward_combined(ro.obs_feat[step + 1], bench.GAMMA, ro.masks[step], offset=-r_sa)
            ret_rms.update(returns.view(-1).cpu().numpy())
            rews = ro.rewards[step].view(-1).cpu().numpy()
            rews = np.clip(rews / np.sqrt(ret_rms.var + 1e-7), -10.0, 10.0)
            ro.rewards[step] = torch.Tensor(rews).view(-1, 1)
        ro.compute_returns(next_value, True, bench.GAMMA, bench.LAM, True)
        value_loss, action_loss, dist_entropy = agent.update(ro)
        rewards_after, returns_after = ro.rewards.numpy().copy(), ro.returns.numpy().copy()
        ro.after_update()
        outs.append(dict(gail_loss=gail_loss, gail_loss_e=gail_loss_e, gail_loss_p=gail_loss_p, value_loss=value_loss,
                         action_loss=action_loss, dist_entropy=dist_entropy, r_sa=r_sa))
    assert np.array_equal(disc.get_flat_params(), disc_r.get_flat_params()), "the discriminator saw the same rows in both modes"
    for a_, b_ in zip(outs, outs_r):
        for k in ("gail_loss", "gail_loss_e", "gail_loss_p"):
            assert a_[k] == b_[k], (k, a_[k], b_[k])
        assert a_["r_sa"] == pytest.approx(b_["r_sa"], rel=1e-12)
        assert_close([a_["value_loss"], a_["action_loss"], a_["dist_entropy"]], [b_["value_loss"], b_["action_loss"], b_["dist_entropy"]],
                     rtol=1e-4, atol=2e-5, what="PPO losses, literal main vs resident learner")
    assert_close(rewards_after, ro_r.rewards.numpy(), rtol=1e-4, atol=1e-5, what="relabelled rewards of the second update")
    assert_close(returns_after[:T], ro_r.returns.numpy()[:T], rtol=1e-4, atol=1e-5, what="GAE returns of the second update")
    assert_close(ret_rms.get_state(), learner_r.ret_rms.get_state(), rtol=1e-5, what="ret_rms after two updates")
    p_a, p_b = pol.get_flat_params(), pol_r.get_flat_params()
    move = np.linalg.norm(p_a.astype(np.float64) - build()[0].get_flat_params())
    assert np.linalg.norm(p_a.astype(np.float64) - p_b) <= 2e-2 * move, "policies of the two modes drifted apart"


def test_verify_mode_catches_a_write_behind_torchs_back(sg, orc, monkeypatch):
    """SG_ROLLOUT_VERIFY=1 (simgan_amd/storage.py): a write through a numpy view of a host tensor does not move torch's version
    counter, so the dirty tracker would train on the stale device copy -- in verify mode the next device call raises instead;
    mark_host_written() is the remedy, and writes torch can see never trip it."""
    from simgan_amd import storage
    monkeypatch.setattr(storage, "_VERIFY", True)
    ro, rng = _filled(sg)
    T, N = ro.num_steps, ro.num_processes
    nv = rng.standard_normal(N).astype(np.float32)
    ro.compute_returns(nv, True, 0.99, 0.95, True)
    ro.rewards[2] = ro.rewards[2] * 0.5                      # visible to torch: fine
    ro.compute_returns(nv, True, 0.99, 0.95, True)
    assert_close(ro.returns.numpy()[:T, :, 0], _oracle_returns(orc, ro, nv)[:T], rtol=1e-5, what="returns after a visible edit")
    ro.rewards.numpy()[5] += 2.0                            # behind torch's back
    with pytest.raises(RuntimeError, match="mark_host_written"):
        ro.compute_returns(nv, True, 0.99, 0.95, True)
    ro.mark_host_written()
    ro.compute_returns(nv, True, 0.99, 0.95, True)
    assert_close(ro.returns.numpy()[:T, :, 0], _oracle_returns(orc, ro, nv)[:T], rtol=1e-5, what="returns after mark_host_written")


def _relabel_problem(sg, seed=11, T=12, N=16, F=9, Hd=16):
    rng = np.random.default_rng(seed)
    D = sg.algo.gail.Discriminator(F, Hd, None, seed=3)
    ro = sg.RolloutStorage(T, N, (4,), Box((2,)), 1, F)
    ro.obs_feat.copy_(ro.obs_feat.new_tensor(rng.standard_normal((T + 1, N, F)).astype(np.float32)))
    ro.masks.copy_(ro.masks.new_tensor((rng.random((T + 1, N, 1)) > 0.2).astype(np.float32)))
    return D, ro, rng


def _loop(D, ro, steps, gamma=0.97, offset=0.3):
    """The main's relabel loop (a2c/main_gail_dyn_ppo.py:275-280), returning every call's (reward, returns)."""
    out = []
    for step in steps:
        rew, ret = D.predict_reward_combined(ro.obs_feat[step + 1], gamma, ro.masks[step], offset=offset)
        ro.rewards[step] = rew
        out.append((rew.numpy().copy(), ret.numpy().copy()))
    return out


def _calls(D):
    """(per-call launches, fused launches) so far, read off the ctypes entry points through a counting shim."""
    return D._n_single, D._n_steps


def _counting(D):
    lib = D.lib
    D._n_single = D._n_steps = 0

    class Counting(object):
        def __getattr__(self, name):
            f = getattr(lib, name)
            if name == "sg_disc_predict_reward":
                def g(*a):
                    D._n_single += 1
                    return f(*a)
                return g
            if name == "sg_disc_predict_reward_steps":
                def g(*a):
                    D._n_steps += 1
                    return f(*a)
                return g
            return f

    D.lib = Counting()
    return D


def test_the_mains_relabel_loop_is_one_launch_and_bit_identical_to_T_calls(sg, monkeypatch):
    """a2c/main_gail_dyn_ppo.py:275-280 through the unchanged call: the call for step 0 computes all T steps
    (sg_disc_predict_reward_steps), steps 1 .. T-1 are served from it -- every (reward, returns) pair, Discriminator.returns
    after every call and at the end EQUAL what T separate launches give; over two consecutive loops (the second continues
    Discriminator.returns), with a discriminator update in between."""
    from simgan_amd.algo import gail
    D, ro, rng = _relabel_problem(sg)
    T, N = ro.num_steps, ro.num_processes
    p0 = D.get_flat_params()
    expert = rng.standard_normal((64, 9)).astype(np.float32)

    class Loader:
        def __init__(self, e, b):
            self.expert, self.batch_size = e, b

    def run(prefetch):
        monkeypatch.setattr(gail, "_PREFETCH", prefetch)
        D.set_flat_params(p0)
        D.set_adam(np.zeros_like(p0), np.zeros_like(p0), 0)
        D.returns = None
        _counting(D)
        res, mid = [], []
        for it in range(2):
            for step in range(T):
                res += _loop(D, ro, [step])
                mid.append(D.returns.numpy().copy())            # read mid-loop: the state after the calls made so far
            D.update_gail_dyn(Loader(expert, 16), ro, expert_perm=np.arange(64), policy_perm=np.arange(T * N), alpha=np.full(64, 0.5, np.float32))
        counts = _calls(D)
        D.lib = D.ctx.lib
        return res, mid, counts, D.returns.numpy().copy()

    fast, fast_mid, fast_counts, fast_end = run(True)
    slow, slow_mid, slow_counts, slow_end = run(False)
    assert fast_counts == (0, 2) and slow_counts == (2 * T, 0), (fast_counts, slow_counts)
    for k, ((ra, ta), (rb, tb)) in enumerate(zip(fast, slow)):
        assert np.array_equal(ra, rb) and np.array_equal(ta, tb), f"call {k}: served from the fused launch != its own launch"
    for a_, b_ in zip(fast_mid, slow_mid):
        assert np.array_equal(a_, b_)
    assert np.array_equal(fast_end, slow_end)


@pytest.mark.parametrize("case", ["foreign_rows", "mask_edit", "feat_edit", "gamma_change", "offset_change", "out_of_order", "weights_change",
                                  "returns_assigned", "starts_late", "resident"])
def test_relabel_loop_cache_misses_fall_back_to_the_per_call_path(sg, monkeypatch, case):
    """Everything that is not the main's pattern must give exactly what per-call launches give: the served-from-cache state
    is committed (Discriminator.returns after the calls served so far) and the call takes its own launch."""
    from simgan_amd.algo import gail
    T = 12

    def scenario(D, ro, rng_):
        out = _loop(D, ro, range(4))
        if case == "foreign_rows":
            x = rng_.standard_normal((ro.num_processes, 9)).astype(np.float32)
            rew, ret = D.predict_reward_combined(x, 0.97, ro.masks[4], offset=0.3)
            out.append((rew.numpy().copy(), ret.numpy().copy()))
            out += _loop(D, ro, range(5, T))
        elif case == "mask_edit":
            ro.masks[6] = ro.masks[6] * 0.0
            out += _loop(D, ro, range(4, T))
        elif case == "feat_edit":
            ro.obs_feat[8] = ro.obs_feat[8] + 1.0
            out += _loop(D, ro, range(4, T))
        elif case == "gamma_change":
            out += _loop(D, ro, range(4, T), gamma=0.9)
        elif case == "offset_change":
            out += _loop(D, ro, range(4, T), offset=-0.1)
        elif case == "out_of_order":
            out += _loop(D, ro, [6, 5, 4, 7, 8])
        elif case == "weights_change":
            D.set_flat_params(D.get_flat_params() * np.float32(1.01))
            out += _loop(D, ro, range(4, T))
        elif case == "returns_assigned":
            D.returns = np.full((ro.num_processes, 1), 0.25, np.float32)
            out += _loop(D, ro, range(4, T))
        elif case == "starts_late":
            out += _loop(D, ro, range(4, T))
            D.returns = None
            out += _loop(D, ro, range(3, T))                     # a loop that does not start at step 0: never prefetched
        elif case == "resident":
            out += _loop(D, ro, range(4, T))
            ro.sync_to_device()
            ro.device_resident = True                            # host slices are then not what the device holds: per-call path
            out += _loop(D, ro, range(0, T))
        out.append((D.returns.numpy().copy(), D.returns.numpy().copy()))
        return out

    results = []
    for prefetch in (True, False):
        monkeypatch.setattr(gail, "_PREFETCH", prefetch)
        D, ro, rng = _relabel_problem(sg)
        _counting(D)
        results.append((scenario(D, ro, np.random.default_rng(5)), _calls(D)))
    (fast, fc), (slow, sc) = results
    assert fc[1] >= 1 and sc[1] == 0, (fc, sc)
    assert len(fast) == len(slow)
    for k, ((ra, ta), (rb, tb)) in enumerate(zip(fast, slow)):
        assert np.array_equal(ra, rb) and np.array_equal(ta, tb), f"{case}: call {k} differs between the cached and the per-call path"
