"""CPU: pin oracle/sg_oracle.c against the fixtures captured from the reference
(tools/gen_golden.py).  The reference has no tests of its own for this path
(SURVEY.md section 4), so these captures are the pin."""
import numpy as np
import pytest

from oracle import oracle as orc
from helpers import assert_close, load

KIND = {"mlp": orc.KIND_MLP, "split": orc.KIND_SPLIT}


def dims_of(m):
    return orc.dims(KIND[m["kind"]], m["O"], m["A"], m["H"], m["num_feet"])


POLICY_CASES = ["policy_mlp_tiny", "policy_mlp_northstar", "policy_mlp_hopper",
                "policy_split_hopper", "policy_split_laikago", "policy_split_tiny"]


@pytest.mark.parametrize("name", POLICY_CASES)
def test_policy_forward_act_evaluate(name):
    g = load(name)
    d = dims_of(g["meta"])
    assert orc.policy_num_params(d) == g["params"].size
    value, mean, logstd = orc.policy_forward(d, g["params"], g["obs"])
    assert_close(mean, g["mean"], what="mean")
    assert_close(np.exp(logstd), g["std"], what="std")
    assert_close(value, g["get_value"], what="value")
    v, a, lp = orc.policy_act(d, g["params"], g["obs"], g["noise"])
    assert_close(v, g["act_value"], what="act value")
    assert_close(a, g["act_action"], what="act action")
    assert_close(lp, g["act_logp"], what="act logp")
    v, a, lp = orc.policy_act(d, g["params"], g["obs"], None)
    assert_close(a, g["det_action"], what="det action")
    assert_close(lp, g["det_logp"], what="det logp")
    v, lp, ent = orc.policy_evaluate(d, g["params"], g["obs"], g["eval_action"])
    assert_close(v, g["eval_value"], what="eval value")
    assert_close(lp, g["eval_logp"], what="eval logp")
    assert_close(ent, g["eval_entropy"], what="entropy")


@pytest.mark.parametrize("use_gae", [1, 0])
@pytest.mark.parametrize("proper", [1, 0])
def test_compute_returns(use_gae, proper):
    g = load("gae")
    ret, vp = orc.compute_returns(g["rewards"][..., 0], g["value_preds"][..., 0], g["masks"][..., 0],
                                  g["bad_masks"][..., 0], g["next_value"][:, 0], use_gae, 0.99, 0.95, proper)
    T = g["rewards"].shape[0]
    want = g[f"returns_gae{use_gae}_proper{proper}"][..., 0]
    # the GAE branch never writes returns[T] (a2c/storage.py:110-120); compare the slots it defines
    upto = T if use_gae else T + 1
    assert_close(ret[:upto], want[:upto], rtol=1e-5, what="returns")
    assert_close(vp, g[f"value_preds_gae{use_gae}_proper{proper}"][..., 0], what="value_preds")


def test_running_mean_std():
    g = load("rms")
    st = [0.0, 1.0, 1e-4]
    for x, want in zip(g["xs"], g["states"]):
        st = orc.rms_update(st, x)
        assert_close(st, want, rtol=1e-6, what="rms state")


PPO_CASES = ["ppo_mlp_tiny", "ppo_mlp_northstar", "ppo_mlp_onestep", "ppo_split_hopper",
             "ppo_split_laikago"]


def ppo_cfg_of(m):
    return orc.ppo_cfg(m["clip_param"], m["ppo_epoch"], m["num_mini_batch"], m["value_loss_coef"],
                       m["entropy_coef"], m["lr"], m["eps"], m["max_grad_norm"], True)


@pytest.mark.parametrize("name", PPO_CASES)
def test_ppo_update(name):
    g = load(name)
    m = g["meta"]
    d = dims_of(m)
    adv = orc.advantages(g["returns"][:-1], g["value_preds"][:-1])
    assert_close(adv, g["advantages"], rtol=1e-5, what="advantages")
    params = g["params0"].copy()
    adam = orc.AdamState(params.size)
    losses = orc.ppo_update(d, params, adam, ppo_cfg_of(m), g["obs"], g["actions"],
                            g["value_preds"][..., 0], g["returns"][..., 0],
                            g["action_log_probs"][..., 0], g["perms"])
    assert_close(losses, g["losses"], what="ppo losses")
    assert_close(adam.m, g["adam_m"], rtol=1e-3, atol=1e-7, what="adam m")
    assert_close(adam.v, g["adam_v"], rtol=1e-3, atol=1e-10, what="adam v")
    assert_close(params, g["params1"], what="params after update")
    # the update must actually have moved the parameters by more than the tolerance
    assert np.max(np.abs(g["params1"] - g["params0"])) > 1e-4


DISC_CASES = ["disc_tiny", "disc_northstar", "disc_hopper", "disc_single_batch"]


@pytest.mark.parametrize("name", DISC_CASES)
def test_disc_update(name):
    g = load(name)
    m = g["meta"]
    params = g["params0"].copy()
    adam = orc.AdamState(params.size)
    for ep in range(m["epochs"]):
        losses, n_d = orc.disc_update(m["F"], m["Hd"], params, adam, g["expert"], g["obs_feat"], m["B"],
                                      g[f"expert_perm{ep}"], g[f"policy_perm{ep}"], g[f"alpha{ep}"])
        assert n_d == int(g[f"n_steps{ep}"])
        assert_close(losses, g[f"losses{ep}"], what=f"disc losses ep{ep}")
        assert_close(params, g[f"params_after{ep}"], what=f"disc params ep{ep}")


def _classic_rows(g):
    """Row assembly of Discriminator.update (a2c/algo/gail.py:102-109, 117-122) from a fixture."""
    m = g["meta"]
    flat = lambda a: a.reshape(-1, a.shape[-1])  # noqa: E731
    es = g["e_state"]
    if m["use_filt"]:
        es = np.clip((es - g["filt_mean"]) / g["filt_std"], -5.0, 5.0).astype(np.float32)
    expert = np.concatenate([es, g["e_action"]], axis=1)
    if m["dyn"]:
        rows = np.concatenate([flat(g["obs_feat"][:-1]), flat(g["obs"][:-1])[:, -m["a_dim"]:], flat(g["obs_feat"][1:])], axis=1)
    else:
        rows = np.concatenate([flat(g["obs"][:-1]), flat(g["actions"])], axis=1)
    return expert, np.ascontiguousarray(rows, np.float32)


@pytest.mark.parametrize("name", ["disc_classic_sa", "disc_classic_dyn"])
def test_disc_update_classic(name):
    """Discriminator.update: same step as update_gail_dyn on (state | action) rows; the oracle takes the
    rows as the [1:] slots of a feature tensor."""
    g = load(name)
    m = g["meta"]
    expert, rows = _classic_rows(g)
    in_dim = rows.shape[1]
    feat = np.concatenate([np.zeros((1, m["N"], in_dim), np.float32), rows.reshape(m["T"], m["N"], in_dim)])
    params = g["params0"].copy()
    adam = orc.AdamState(params.size)
    losses, n_d = orc.disc_update(in_dim, m["Hd"], params, adam, expert, feat, m["B"], g["expert_perm"],
                                  g["policy_perm"], g["alpha"])
    assert n_d == int(g["n_steps"])
    assert_close(losses, g["losses"], what="classic D losses")
    assert_close(params, g["params_after"], what="classic D params")


@pytest.mark.parametrize("name", ["relabel_tiny", "relabel_northstar"])
def test_relabel(name):
    g = load(name)
    m = g["meta"]
    rew, ret = orc.disc_predict_reward(m["F"], m["Hd"], g["params"], g["obs_feat0"][1], m["gamma"],
                                       g["masks0"][0], float(g["offset0"]))
    assert_close(rew, g["raw_reward0"], what="raw reward")
    d_ret, rms = None, [0.0, 1.0, 1e-4]
    for call in range(2):
        rewards, d_ret, rms = orc.relabel(m["F"], m["Hd"], g["params"], g[f"obs_feat{call}"],
                                          g[f"masks{call}"][..., 0], m["gamma"], float(g[f"offset{call}"]),
                                          d_ret, rms)
        assert_close(rewards, g[f"rewards{call}"][..., 0], what="relabelled rewards")
        assert_close(d_ret, g[f"d_returns{call}"][:, 0], what="D.returns")
        assert_close(rms, g[f"rms{call}"], rtol=1e-5, what="ret_rms")


def test_predict_reward_state_action():
    """Discriminator.predict_reward(state, action, gamma, masks, offset) a2c/algo/gail.py:195-199: the combined form on
    cat(state, action); two calls, the second continues Discriminator.returns."""
    g = load("predict_reward")
    m = g["meta"]
    ret = None
    for c in range(2):
        x = np.concatenate([g[f"state{c}"], g[f"action{c}"]], axis=1)
        rew, ret = orc.disc_predict_reward(m["S"] + m["A"], m["Hd"], g["params"], x, m["gamma"], g[f"masks{c}"][:, 0], 0.25 * c, ret)
        assert_close(rew, g[f"reward{c}"], what=f"reward, call {c}")
        assert_close(ret, g[f"returns{c}"], what=f"returns, call {c}")


@pytest.mark.parametrize("tag", ["small", "northstar"])
def test_grad_pen_value(tag):
    """Discriminator.compute_grad_pen_combined / compute_grad_pen (a2c/algo/gail.py:53-89) on the reference's rows and draws."""
    g = load("grad_pen")
    m = g["meta"][tag]
    for suffix, lam in (("", 10.0), ("2", 4.0)):
        v = orc.disc_grad_pen(m["F"], m["Hd"], g[f"{tag}_params"], g[f"{tag}_expert"], g[f"{tag}_policy"], g[f"{tag}_alpha{suffix}"], lam)
        assert_close(v, g[f"{tag}_value{suffix}"], what=f"grad_pen {tag}{suffix}")


@pytest.mark.parametrize("name", ["iter_mlp", "iter_split"])
def test_full_iteration(name):
    """a2c/main_gail_dyn_ppo.py:239-304 restated with oracle calls, 2 outer iterations."""
    g = load(name)
    m = g["meta"]
    d = dims_of(m)
    T, N, F, Hd, B = m["T"], m["N"], m["F"], m["Hd"], m["B"]
    pi = g["pi_params0"].copy()
    dp = g["d_params0"].copy()
    pi_adam, d_adam = orc.AdamState(pi.size), orc.AdamState(dp.size)
    cfg = orc.ppo_cfg(0.2, m["ppo_epoch"], m["num_mini_batch"], 0.5, 0.0, 3e-4, 1e-5, 0.5, True)
    obs = np.zeros((T + 1, N, m["O"]), np.float32)
    obs_feat = np.zeros((T + 1, N, F), np.float32)
    masks = np.ones((T + 1, N), np.float32)
    bad = np.ones((T + 1, N), np.float32)
    obs[0] = g["obs0"]
    d_ret, rms = None, [0.0, 1.0, 1e-4]
    for j in range(m["iters"]):
        actions = np.zeros((T, N, m["A"]), np.float32)
        logp = np.zeros((T, N), np.float32)
        vp = np.zeros((T + 1, N), np.float32)
        for t in range(T):
            v, a, lp = orc.policy_act(d, pi, obs[t], g[f"it{j}_noise"][t])
            actions[t], logp[t], vp[t] = a, lp[:, 0], v[:, 0]
            obs[t + 1] = g[f"it{j}_env_obs"][t]
            obs_feat[t + 1] = g[f"it{j}_env_feat"][t]
            masks[t + 1] = g[f"it{j}_env_masks"][t][:, 0]
            bad[t + 1] = g[f"it{j}_env_bad"][t][:, 0]
        assert_close(actions, g[f"it{j}_actions"], what="rollout actions")
        assert_close(logp, g[f"it{j}_action_log_probs"][..., 0], what="rollout logp")
        nv = orc.policy_forward(d, pi, obs[T])[0]
        assert_close(nv, g[f"it{j}_next_value"], what="next_value")
        for ep in range(m["gail_epoch"]):
            losses, _ = orc.disc_update(F, Hd, dp, d_adam, g["expert"], obs_feat, B,
                                        g[f"it{j}_d{ep}_expert_perm"], g[f"it{j}_d{ep}_policy_perm"],
                                        g[f"it{j}_d{ep}_alpha"])
            assert_close(losses, g[f"it{j}_d_losses"][ep], what="D losses")
        assert_close(dp, g[f"it{j}_d_params"], what="D params")
        r_sa = orc.alive_bonus(masks, T, N, m["gail_tar_length"])
        assert_close(r_sa, g[f"it{j}_r_sa"], rtol=1e-9, what="r_sa")
        rewards, d_ret, rms = orc.relabel(F, Hd, dp, obs_feat, masks, m["gamma"], -r_sa, d_ret, rms)
        assert_close(rewards, g[f"it{j}_rewards"][..., 0], what="rewards")
        assert_close(rms, g[f"it{j}_rms"], rtol=1e-5, what="rms")
        ret, vp = orc.compute_returns(rewards, vp, masks, bad, nv[:, 0], 1, m["gamma"], m["gae_lambda"], 1)
        assert_close(ret[:T], g[f"it{j}_returns"][:T, :, 0], what="returns")
        losses = orc.ppo_update(d, pi, pi_adam, cfg, obs, actions, vp, ret, logp, g[f"it{j}_ppo_perms"])
        assert_close(losses, g[f"it{j}_ppo_losses"], what="ppo losses")
        assert_close(pi, g[f"it{j}_pi_params"], what="pi params")
        # after_update a2c/storage.py:96-101
        obs[0], obs_feat[0], masks[0], bad[0] = obs[T], obs_feat[T], masks[T], bad[T]


@pytest.mark.parametrize("fixture", ["iter_refine", "iter_refine_h100"])
def test_refine_iteration(fixture):
    """a2c/main.py:199-257 (policy refinement: warm start, reset critic / variance, linear LR decay, no D) restated
    with oracle calls: 2 outer iterations at the Laikago refinement shape (obs 111, act 12, h64, 8 minibatches, clip 0.1), and
    from a 100-unit behaviour policy, where reset_critic leaves a 64-unit critic beside the 100-unit actor
    (a2c/model.py:80-87 hard-codes 64)."""
    g = load(fixture)
    m = g["meta"]
    T, N, O, A, H = m["T"], m["N"], m["O"], m["A"], m["H"]
    Hc = 64                                         # the reference's reset_critic
    d = orc.dims(KIND[m["kind"]], O, A, H, m["num_feet"], Hc)
    pi = g["pi_params0"].copy()
    assert pi.size == orc.policy_num_params(d)
    # warm start keeps the actor and the mean head, re-draws the critic (zero biases), resets logstd (a2c/main.py:85-87)
    beh = g["behaviour_params"]
    na = H * O + H + H * H + H                      # base.actor.*
    nc = Hc * O + Hc + Hc * Hc + Hc + Hc + 1        # base.critic.* + critic_linear, 64 wide after the reset
    nc_beh = H * O + H + H * H + H + H + 1          # ... and as wide as the actor in the behaviour policy
    assert np.array_equal(pi[:na], beh[:na])
    assert np.array_equal(pi[na + nc:na + nc + A * H + A], beh[na + nc_beh:na + nc_beh + A * H + A])
    assert np.all(pi[-A:] == np.float32(m["warm_start_logstd"]))
    assert np.all(pi[na + Hc * O:na + Hc * O + Hc] == 0) and pi[na + nc - 1] == 0
    W1c = pi[na:na + Hc * O].reshape(Hc, O)
    assert_close((W1c @ W1c.T if Hc <= O else W1c.T @ W1c), 2.0 * np.eye(min(Hc, O)), rtol=0, atol=1e-5, what="critic.0 orthogonal, gain sqrt 2")
    adam = orc.AdamState(pi.size)
    obs = np.zeros((T + 1, N, O), np.float32)
    masks, bad = np.ones((T + 1, N), np.float32), np.ones((T + 1, N), np.float32)
    obs[0] = g["obs0"]
    for j in range(m["iters"]):
        lr = m["lr"] - m["lr"] * (j / float(m["num_updates"]))        # a2c/utils.py:68-72
        assert lr == float(g["lrs"][j])
        cfg = orc.ppo_cfg(m["clip_param"], m["ppo_epoch"], m["num_mini_batch"], 0.5, 0.0, lr, 1e-5, 0.5, True)
        actions, logp = np.zeros((T, N, A), np.float32), np.zeros((T, N), np.float32)
        vp, rewards = np.zeros((T + 1, N), np.float32), np.zeros((T, N), np.float32)
        for t in range(T):
            v, a, lp = orc.policy_act(d, pi, obs[t], g[f"it{j}_noise"][t])
            actions[t], logp[t], vp[t] = a, lp[:, 0], v[:, 0]
            obs[t + 1] = g[f"it{j}_env_obs"][t]
            rewards[t] = g[f"it{j}_env_reward"][t][:, 0]
            masks[t + 1] = g[f"it{j}_env_masks"][t][:, 0]
            bad[t + 1] = g[f"it{j}_env_bad"][t][:, 0]
        assert_close(actions, g[f"it{j}_actions"], what="rollout actions")
        assert_close(logp, g[f"it{j}_action_log_probs"][..., 0], what="rollout logp")
        nv = orc.policy_forward(d, pi, obs[T])[0]
        assert_close(nv, g[f"it{j}_next_value"], what="next_value")
        ret, vp = orc.compute_returns(rewards, vp, masks, bad, nv[:, 0], 1, m["gamma"], m["gae_lambda"], 1)
        assert_close(ret[:T], g[f"it{j}_returns"][:T, :, 0], what="returns")
        losses = orc.ppo_update(d, pi, adam, cfg, obs, actions, vp, ret, logp, g[f"it{j}_ppo_perms"])
        assert_close(losses, g[f"it{j}_ppo_losses"], what="ppo losses")
        assert_close(pi, g[f"it{j}_pi_params"], what="pi params")
        obs[0], masks[0], bad[0] = obs[T], masks[T], bad[T]


# ---------------------------------------------------------------------------------------------------------------------
# oracle/sg_cpu_fast.c -- the batched, vectorised CPU implementation bench.py times as `cpu_baseline` -- against the
# oracle: same gradients and loss sums at the suite's tolerance, for both policy kinds and the discriminator.
@pytest.mark.parametrize("kind,O,A,H,f,Hc", [("mlp", 47, 12, 64, 1, 0), ("mlp", 11, 3, 64, 1, 0), ("split", 14, 7, 100, 1, 0),
                                             ("split", 64, 28, 100, 4, 0), ("mlp", 5, 2, 7, 1, 0), ("mlp", 20, 5, 100, 1, 64), ("mlp", 9, 4, 24, 1, 64)])
def test_fast_cpu_ppo_gradient_equals_the_oracle(kind, O, A, H, f, Hc):
    rng = np.random.default_rng(O * 100 + H)
    d = orc.dims(orc.KIND_MLP if kind == "mlp" else orc.KIND_SPLIT, O, A, H, f, Hc)
    n = 203                                      # not a multiple of the 4-row panels
    # (a state-dependent log-std head on N(0, 0.15) weights gives log-probs of -1000 and sums that cancel to 1e-4 of their terms)
    par = (rng.standard_normal(orc.policy_num_params(d)) * (0.15 if kind == "mlp" else 0.05)).astype(np.float32)
    obs = rng.standard_normal((n, O)).astype(np.float32)
    act = rng.standard_normal((n, A)).astype(np.float32)
    v_now, lp, _ = orc.policy_evaluate(d, par, obs, act)
    old_logp = (lp[:, 0] + 0.3 * rng.standard_normal(n)).astype(np.float32)      # ratios on both sides of the clip
    vpred = (v_now[:, 0] + 0.3 * rng.standard_normal(n)).astype(np.float32)
    ret = (v_now[:, 0] + rng.standard_normal(n)).astype(np.float32)
    adv = rng.standard_normal(n).astype(np.float32)
    rows = rng.permutation(n)[:150]
    for ecoef, clipped in ((0.0, True), (0.01, False)):
        cfg = orc.ppo_cfg(0.2, 1, 1, 0.5, ecoef, 3e-4, 1e-5, 0.5, clipped)
        G0, s0 = orc.ppo_grad_rows(d, par, cfg, obs, act, vpred, ret, old_logp, adv, rows, 1.0 / rows.size)
        G1, s1 = orc.ppo_grad_rows_fast(d, par, cfg, obs, act, vpred, ret, old_logp, adv, rows, 1.0 / rows.size)
        assert_close(s1, s0, what="loss sums")
        assert np.abs(G1 - G0).max() <= 1e-4 * np.abs(G0).max() + 1e-7, (np.abs(G1 - G0).max(), np.abs(G0).max())


@pytest.mark.parametrize("F,Hd,nb", [(86, 100, 128), (25, 100, 128), (7, 5, 3), (86, 100, 37)])
def test_fast_cpu_discriminator_gradient_equals_the_oracle(F, Hd, nb):
    rng = np.random.default_rng(F + Hd + nb)
    par = (rng.standard_normal(orc.disc_num_params(F, Hd)) * 0.2).astype(np.float32)
    e = rng.standard_normal((nb, F)).astype(np.float32)
    p = (rng.standard_normal((nb, F)) * 0.8 + 0.3).astype(np.float32)
    al = rng.random(nb).astype(np.float32)
    G0, s0 = orc.disc_grad_rows(F, Hd, par, e, p, al, 1.0 / nb)
    G1, s1 = orc.disc_grad_rows_fast(F, Hd, par, e, p, al, 1.0 / nb)
    assert_close(s1, s0, what="loss sums")
    assert np.abs(G1 - G0).max() <= 1e-4 * np.abs(G0).max() + 1e-7, (np.abs(G1 - G0).max(), np.abs(G0).max())


# ------------------------------------------------------------------------------------------ the float64 arbiter
# oracle/sg_oracle_f64.c is the oracle's source with float := double.  It is NOT a parity oracle (the reference computes in
# float32); these tests only establish that it is the same algorithm -- it reproduces the reference's captures to float32
# round-off -- so that tools/parity_f64.py may use it as the point both float32 evaluations are measured from.
@pytest.mark.parametrize("name", ["ppo_mlp_northstar", "ppo_split_hopper"])
def test_float64_arbiter_runs_the_same_ppo_update(name):
    from oracle import oracle64 as o64
    g = load(name)
    m = g["meta"]
    d = o64.dims(KIND[m["kind"]], m["O"], m["A"], m["H"], m["num_feet"])
    cfg = o64.ppo_cfg(m["clip_param"], m["ppo_epoch"], m["num_mini_batch"], m["value_loss_coef"], m["entropy_coef"], m["lr"], m["eps"],
                      m["max_grad_norm"], True)
    params = g["params0"].astype(np.float64)
    adam = o64.AdamState(params.size)
    assert adam.m.dtype == np.float64
    losses = o64.ppo_update(d, params, adam, cfg, g["obs"], g["actions"], g["value_preds"][..., 0], g["returns"][..., 0],
                            g["action_log_probs"][..., 0], g["perms"])
    assert_close(losses, g["losses"], what="ppo losses (float64)")
    assert_close(params, g["params1"], what="params after update (float64)")


@pytest.mark.parametrize("name", ["disc_northstar", "disc_hopper"])
def test_float64_arbiter_runs_the_same_discriminator_epochs(name):
    from oracle import oracle64 as o64
    g = load(name)
    m = g["meta"]
    params = g["params0"].astype(np.float64)
    adam = o64.AdamState(params.size)
    for ep in range(m["epochs"]):
        losses, n_d = o64.disc_update(m["F"], m["Hd"], params, adam, g["expert"], g["obs_feat"], m["B"],
                                      g[f"expert_perm{ep}"], g[f"policy_perm{ep}"], g[f"alpha{ep}"])
        assert n_d == int(g[f"n_steps{ep}"])
        assert_close(losses, g[f"losses{ep}"], what=f"disc losses ep{ep} (float64)")
        assert_close(params, g[f"params_after{ep}"], what=f"disc params ep{ep} (float64)")
