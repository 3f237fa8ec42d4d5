"""SURVEY.md section 8(f) row N4 on the GPU: the expert wire format and the environment pool with VecNormalize's return
scaling FEED the HIP path, and what the HIP path then computes is checked against the CPU oracle.

  * tests/golden/expert_trajs.pkl (the collector's pickle, a2c/collect_tarsim_traj.py:218-265) -> simgan_amd.expert
    (my_pybullet_envs/utils.py:170-263) -> the matrix the reference's own helpers produced (expert_trajs.npz) ->
    sg_disc_set_expert -> discriminator epochs with injected draws == oracle.disc_update on the fixture's matrix;
  * the scripted rewards / dones the reference's VecNormalize was run on (vecnormalize.npz, vec_normalize.py:50-58) played
    through simgan_amd.envs.make_vec_envs (EnvPool + ReturnNormalizer) and driver.PpoLearner.collect(): the rewards that
    reach the DEVICE rollout are the reference's scaled rewards, bit for bit, and the GAE returns and the PPO update that
    follow equal the oracle's on those rewards -- in drop-in and in device-resident mode.
"""
import os

import numpy as np
import pytest
from helpers import GOLDEN, assert_close, load

pytestmark = pytest.mark.gpu


class Box:  # duck-typed gym.spaces.Box (a2c/model.py:55-57 reads __class__.__name__ and .shape)
    def __init__(self, shape):
        self.shape = shape


@pytest.fixture(scope="module")
def sg():
    import simgan_amd
    return simgan_amd


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle
    return oracle


@pytest.mark.parametrize("through_learner", [False, True])
def test_expert_pickle_feeds_the_discriminator_kernels(sg, orc, through_learner):
    from simgan_amd import expert as ex
    from simgan_amd.driver import ExpertLoader, GailDynLearner
    g = load("expert_trajs")
    path = os.path.join(GOLDEN, "expert_trajs.pkl")
    mat, n_e = ex.expert_matrix(path, s_idx=(0, 2), a_idx=(0, 1), downsample_freq=2, start_idx=g["start_idx"])
    ref = g["merged"].astype(np.float32)                    # what the reference's helpers + torch.Tensor(...) hand the DataLoader
    assert mat.dtype == np.float32 and n_e == ref.shape[0] and np.array_equal(mat, ref)
    F, Hd, B, T, N = ref.shape[1], 16, 3, 4, 5              # 8 expert rows, batches of 3: drop_last leaves 2 steps per epoch
    rng = np.random.default_rng(17)
    D = sg.algo.gail.Discriminator(F, Hd, None, seed=5)
    p0 = D.get_flat_params()
    ro = sg.RolloutStorage(T, N, (3,), Box((2,)), 1, F)
    feat = (rng.standard_normal((T + 1, N, F)) * 0.5).astype(np.float32)
    ro.obs_feat.copy_(ro.obs_feat.new_tensor(feat))
    loader = ExpertLoader(mat, B)                           # driver's stand-in for DataLoader(TensorDataset(expert), B, shuffle, drop_last)
    if through_learner:                                     # ... or handed to the learner as the bare matrix (main_gail_dyn_ppo.py:165-174)
        pol = sg.Policy((3,), Box((2,)), base_kwargs={"recurrent": False, "hidden_size": 8}, seed=1)
        agent = sg.algo.PPO(pol, 0.2, 1, 1, 0.5, 0.0, lr=3e-4, eps=1e-5, max_grad_norm=0.5)
        loader = GailDynLearner(pol, agent, D, ro, mat, gail_batch_size=B, gail_epoch=1,
                                gail_tar_length=ex.gail_tar_length(n_e, 3, 2)).loader
        assert loader.batch_size == B and np.array_equal(loader.expert, ref)
    n_d = min(n_e // B, (T * N) // B)
    assert n_d == 2
    par, adam = p0.copy(), orc.AdamState(p0.size)
    for epoch in range(3):
        eperm = rng.permutation(n_e).astype(np.int64)
        pperm = rng.permutation(T * N).astype(np.int64)
        alpha = rng.random(n_d * B).astype(np.float32)
        losses = D.update_gail_dyn(loader, ro, expert_perm=eperm, policy_perm=pperm, alpha=alpha)
        olosses, on = orc.disc_update(F, Hd, par, adam, ref, feat, B, eperm, pperm, alpha)
        assert D.last_n_steps == on == n_d
        assert_close(losses, olosses, what=f"D losses, epoch {epoch}")
        assert_close(D.get_flat_params(), par, what=f"D params, epoch {epoch}")


class ScriptedEnv:
    """One column of the scripted run the reference's VecNormalize saw: raw reward and done flag per step from the fixture;
    observations a deterministic function of (column, step, episode) so that the policy's actions matter to nothing here."""

    def __init__(self, gid, seed, raw, news, obs_dim):
        self.gid, self.seed, self.raw, self.news, self.obs_dim, self.t, self.episodes = gid, seed, raw, news, obs_dim, 0, 0

    def _obs(self):
        return np.sin(np.arange(self.obs_dim) * 0.37 + 0.9 * self.gid + 0.05 * self.t + 0.5 * self.episodes).astype(np.float32)

    def reset(self):
        self.episodes += 1
        return self._obs()

    def step(self, action):
        t = self.t
        self.t += 1
        done = bool(self.news[t, self.gid])
        return self._obs(), self.raw[t, self.gid], done, ({"bad_transition": True} if done and self.gid % 2 else {})


@pytest.mark.parametrize("resident", [False, True])
def test_env_pool_rewards_reach_the_device_and_the_ppo_update_matches_the_oracle(sg, orc, resident):
    from simgan_amd import _lib
    from simgan_amd.driver import PpoLearner
    from simgan_amd.envs import make_vec_envs
    g = load("vecnormalize")
    raw, news, gamma = g["raw"], g["news"], float(g["gamma"])
    T, N = raw.shape
    O, A, H, M, E = 6, 2, 16, 3, 2
    envs = make_vec_envs(lambda gid, seed: ScriptedEnv(gid, seed, raw, news, O), seed=7, num_processes=N, gamma=gamma)
    assert [e.seed for e in envs.venv.envs] == [7 + i for i in range(N)]          # a2c/envs.py:68
    pol = sg.Policy((O,), Box((A,)), base_kwargs={"recurrent": False, "hidden_size": H}, seed=3)
    agent = sg.algo.PPO(pol, 0.2, E, M, 0.5, 0.01, lr=3e-4, eps=1e-5, max_grad_norm=0.5)
    ro = sg.RolloutStorage(T, N, (O,), Box((A,)), 1, O)
    ro.device_resident = resident
    obs0 = envs.reset()
    ro.obs[0].copy_(obs0)
    ro.obs_feat[0].copy_(obs0)
    if resident:
        ro.sync_to_device()
    learner = PpoLearner(pol, agent, ro, gamma=gamma, gae_lambda=0.95)
    rng = np.random.default_rng(23)
    noises = iter(rng.standard_normal((T, N, A)).astype(np.float32))
    real_act = pol.act
    pol.act = lambda *a_, **k_: real_act(*a_, noise=next(noises), **k_)
    learner.collect(envs)
    pol.act = real_act
    # (1) the pool's return scaling is the reference's, on the host ...
    scaled32 = g["scaled"].astype(np.float32)               # VecPyTorch: torch.from_numpy(reward).unsqueeze(1).float(), a2c/envs.py:208
    assert np.array_equal(ro.rewards.numpy()[:, :, 0], scaled32)
    assert_close([envs.ret_rms.mean, envs.ret_rms.var, envs.ret_rms.count], g["rms"][-1], rtol=1e-12, atol=0, what="ret_rms")
    assert np.array_equal(ro.masks.numpy()[1:, :, 0], 1.0 - news.astype(np.float32))
    # ... (2) and those are the rewards in HBM: read the device copy back
    if not resident:
        ro.sync_to_device()
    dev = np.empty((T, N, 1), np.float32)
    _lib.check(ro.lib.sg_rollout_download(ro.h, _lib.F_REWARDS, _lib.fptr(dev), dev.size))
    assert np.array_equal(dev[:, :, 0], scaled32)
    # (3) the update the HIP path computes from them == the oracle's on the fixture's scaled rewards
    obs, act = ro.obs.numpy().copy(), ro.actions.numpy().copy()
    vp, logp = ro.value_preds.numpy().copy(), ro.action_log_probs.numpy().copy()
    masks, bad = ro.masks.numpy().copy(), ro.bad_masks.numpy().copy()
    p0 = pol.get_flat_params()
    d = orc.dims(orc.KIND_MLP, O, A, H, 1)
    nv = pol.get_value(ro.obs[-1], ro.recurrent_hidden_states[-1], ro.masks[-1])
    nv = nv.numpy() if hasattr(nv, "numpy") else np.asarray(nv)
    oval, _, _ = orc.policy_act(d, p0, obs[-1], noise=np.zeros((N, A), np.float32))
    assert_close(nv, oval, what="next_value")
    oret, ovp = orc.compute_returns(scaled32, vp[..., 0], masks[..., 0], bad[..., 0], oval.reshape(-1), True, gamma, 0.95, True)
    perms = np.stack([rng.permutation(T * N) for _ in range(E)]).astype(np.int64)
    real_update = agent.update
    agent.update = lambda r_, **k_: real_update(r_, perms=perms, **k_)
    out = learner.update()
    agent.update = real_update
    out = dict(out)
    if resident:
        ro.sync_from_device([_lib.F_RETURNS])
    assert_close(ro.returns.numpy()[:T, :, 0], oret[:T], what="GAE returns from the pool's rewards")
    par, adam = p0.copy(), orc.AdamState(p0.size)
    cfg = orc.ppo_cfg(0.2, E, M, 0.5, 0.01, 3e-4, 1e-5, 0.5, True)
    olosses = orc.ppo_update(d, par, adam, cfg, obs, act, ovp, oret, logp[..., 0], perms)
    assert_close([out["value_loss"], out["action_loss"], out["dist_entropy"]], olosses, what="PPO losses")
    assert_close(pol.get_flat_params(), par, what="policy params after the update")


class ScriptedDynEnv(ScriptedEnv):
    """The GAIL-dyn environments hand the learner a (s, a, s') feature row per step in info (a2c/main_gail_dyn_ppo.py:220-226 builds it
    from info["sas_window"]); here a deterministic function of (column, step, action)."""

    def __init__(self, gid, seed, raw, news, obs_dim, feat_dim):
        ScriptedEnv.__init__(self, gid, seed, raw, news, obs_dim)
        self.feat_dim = feat_dim

    def step(self, action):
        o, r, done, info = ScriptedEnv.step(self, action)
        a = np.asarray(action, np.float32).reshape(-1)
        info = dict(info, sas_feat=(np.cos(np.arange(self.feat_dim) * 0.21 + 0.7 * self.gid + 0.03 * self.t) + 0.1 * a.sum()).astype(np.float32))
        return o, r, done, info


def test_gail_dyn_iteration_through_the_env_pool_matches_the_oracle(sg, orc):
    """a2c/main_gail_dyn_ppo.py:209-304 with the rollout filled by GailDynLearner.collect() from an EnvPool shard (env ids, seeds,
    auto-reset, VecNormalize on the environment's own rewards -- which the relabel then overwrites, as in the reference), the expert
    matrix from the pickle fixture, and the update's draws injected: D epochs, r_sa, relabel + float64 statistics, returns and the
    PPO update of the HIP path against the oracle on the rollout the pool produced."""
    from simgan_amd import expert as ex
    from simgan_amd.driver import GailDynLearner, alive_bonus_offset
    from simgan_amd.envs import make_vec_envs
    g, ge = load("vecnormalize"), load("expert_trajs")
    raw, news, gamma = g["raw"], g["news"], float(g["gamma"])
    T, N = raw.shape
    mat, n_e = ex.expert_matrix(os.path.join(GOLDEN, "expert_trajs.pkl"), s_idx=(0, 2), a_idx=(0, 1), downsample_freq=2, start_idx=ge["start_idx"])
    F, Hd, B, O, A, H, M, E, Ed = mat.shape[1], 16, 4, 6, 2, 16, 3, 2, 2
    envs = make_vec_envs(lambda gid, seed: ScriptedDynEnv(gid, seed, raw, news, O, F), seed=11, num_processes=N, gamma=gamma)
    pol = sg.Policy((O,), Box((A,)), base_kwargs={"recurrent": False, "hidden_size": H}, seed=3)
    disc = sg.algo.gail.Discriminator(F, Hd, None, seed=4)
    agent = sg.algo.PPO(pol, 0.2, E, M, 0.5, 0.0, lr=3e-4, eps=1e-5, max_grad_norm=0.5)
    ro = sg.RolloutStorage(T, N, (O,), Box((A,)), 1, F)
    ro.obs[0].copy_(envs.reset())
    tar_len = ex.gail_tar_length(n_e, 3, 2)
    learner = GailDynLearner(pol, agent, disc, ro, mat, gail_batch_size=B, gail_epoch=Ed, gamma=gamma, gail_tar_length=tar_len)
    rng = np.random.default_rng(29)
    noises = iter(rng.standard_normal((T, N, A)).astype(np.float32))
    real_act = pol.act
    pol.act = lambda *a_, **k_: real_act(*a_, noise=next(noises), **k_)
    learner.collect(envs, lambda infos: np.stack([i["sas_feat"] for i in infos]))
    pol.act = real_act
    assert np.array_equal(ro.rewards.numpy()[:, :, 0], g["scaled"].astype(np.float32))       # the pool's return scaling, before the relabel
    obs, feat, act = ro.obs.numpy().copy(), ro.obs_feat.numpy().copy(), ro.actions.numpy().copy()
    vp, logp = ro.value_preds.numpy()[..., 0].copy(), ro.action_log_probs.numpy()[..., 0].copy()
    masks, bad = ro.masks.numpy()[..., 0].copy(), ro.bad_masks.numpy()[..., 0].copy()
    assert np.abs(feat[1:]).max() > 0.5 and np.array_equal(masks[1:], 1.0 - news.astype(np.float32))
    n_d = min(n_e // B, (T * N) // B)
    draws = [(rng.permutation(n_e).astype(np.int64), rng.permutation(T * N).astype(np.int64), rng.random(n_d * B).astype(np.float32)) for _ in range(Ed)]
    perms = np.stack([rng.permutation(T * N) for _ in range(E)]).astype(np.int64)
    p0, dp0 = pol.get_flat_params(), disc.get_flat_params()
    it = iter(draws)
    real_d, real_p = disc.update_gail_dyn, agent.update
    disc.update_gail_dyn = lambda loader, r_, **k_: real_d(loader, r_, **dict(k_, **dict(zip(("expert_perm", "policy_perm", "alpha"), next(it)))))
    agent.update = lambda r_, **k_: real_p(r_, perms=perms, **k_)
    out = learner.update()                                    # drop-in mode: host tensors are the rollout
    disc.update_gail_dyn, agent.update = real_d, real_p
    # ---- the same iteration through the oracle
    dp, d_adam = dp0.copy(), orc.AdamState(dp0.size)
    for ep, pp, al in draws:
        dl, nd = orc.disc_update(F, Hd, dp, d_adam, mat, feat, B, ep, pp, al)
        assert nd == n_d
    r_sa = alive_bonus_offset(float((1.0 - masks).sum()), T, N, tar_len)
    rewards, d_ret, rms = orc.relabel(F, Hd, dp, feat, masks, gamma, -r_sa, None, [0.0, 1.0, 1e-4])
    d = orc.dims(orc.KIND_MLP, O, A, H, 1)
    nv = orc.policy_forward(d, p0, obs[T])[0][:, 0]
    oret, ovp = orc.compute_returns(rewards, vp, masks, bad, nv, True, gamma, 0.95, True)
    par, adam = p0.copy(), orc.AdamState(p0.size)
    ol = orc.ppo_update(d, par, adam, orc.ppo_cfg(0.2, E, M, 0.5, 0.0, 3e-4, 1e-5, 0.5, True), obs, act, ovp, oret, logp, perms)
    assert_close([out["gail_loss"], out["gail_loss_e"], out["gail_loss_p"]], dl, what="D losses of the last epoch")
    assert_close(disc.get_flat_params(), dp, what="D params")
    assert out["r_sa"] == pytest.approx(r_sa, rel=1e-12)
    assert_close(learner.ret_rms.get_state(), rms, rtol=1e-6, what="ret_rms")
    assert_close([out["value_loss"], out["action_loss"], out["dist_entropy"]], ol, what="PPO losses")
    assert_close(pol.get_flat_params(), par, what="policy params")
