"""Generate tests/golden/*.npz by RUNNING THE REFERENCE in the dev container.

Dev-only: imports /root/reference (via tools/ref_import.py stand-in modules for
gym/pybullet), drives the reference's own Policy / SplitPolicy / RolloutStorage /
PPO / Discriminator / RunningMeanStd on seeded inputs, records every RNG-derived
artefact the hot path draws (torch.randperm / torch.rand / sampling noise) and
dumps inputs + outputs as small fixtures.  The fixtures (data only) are
committed; this script documents how they were made.  Re-run:
    python tools/gen_golden.py
The call sequences mirror a2c/main_gail_dyn_ppo.py:239-304 (a2c/ =
third_party/a2c_ppo_acktr/).
"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from ref_import import import_reference  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
ns = import_reference()
torch.set_num_threads(1)  # a2c/main_gail_dyn_ppo.py:64

# ---------------------------------------------------------------- RNG capture
_REC = []
_randperm, _rand = torch.randperm, torch.rand


def _rp(n, *a, **k):
    r = _randperm(n, *a, **k)
    _REC.append(("randperm", r.clone().numpy()))
    return r


def _rd(*a, **k):
    r = _rand(*a, **k)
    _REC.append(("rand", r.clone().numpy()))
    return r


torch.randperm, torch.rand = _rp, _rd


def flat_params(module):
    return np.concatenate([v.detach().numpy().reshape(-1) for v in module.state_dict().values()]).astype(np.float32)


def make_policy(kind, O, A, H, f, seed):
    torch.manual_seed(seed)
    if kind == "mlp":
        p = ns.Policy((O,), ns.Box(shape=(A,)), base_kwargs={"recurrent": False, "hidden_size": H})
    else:
        p = ns.SplitPolicy((O,), ns.Box(shape=(A,)), base_kwargs={"hidden_size": H, "num_feet": f})
    # the init makes the mean head tiny (weights/50 or gain 0.02); bump everything a little so
    # fixtures exercise non-trivial means / state-dependent logstd and non-zero biases
    with torch.no_grad():
        for q in p.parameters():
            q.add_(0.05 * torch.randn_like(q))
    return p


def save(name, **arrs):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


def meta(**kw):
    return np.frombuffer(json.dumps(kw).encode(), dtype=np.uint8)


# ------------------------------------------------------------------ A. policy
def gen_policy(name, kind, O, A, H, f, n, seed):
    p = make_policy(kind, O, A, H, f, seed)
    params = flat_params(p)
    torch.manual_seed(seed + 1)
    obs = torch.randn(n, O)
    hxs, masks = torch.zeros(n, 1), torch.ones(n, 1)
    with torch.no_grad():
        torch.manual_seed(seed + 2)
        value, action, logp, _ = p.act(obs, hxs, masks)
        torch.manual_seed(seed + 2)
        noise = torch.randn(n, A)  # same stream torch.normal(mean,std) consumed
        v_det, a_det, lp_det, _ = p.act(obs, hxs, masks, deterministic=True)
        # check noise reproduces the sampled action: action = mean + std*noise
        value2, feat, _ = p.base(obs, hxs, masks)
        dist = p.dist(feat)
        recon = dist.mean + dist.stddev * noise
        if not torch.allclose(recon, action, rtol=0, atol=1e-6):
            noise = (action - dist.mean) / dist.stddev
        torch.manual_seed(seed + 3)
        act_eval = action + 0.3 * torch.randn(n, A)
        ev_value, ev_logp, ev_ent, _ = p.evaluate_actions(obs, hxs, masks, act_eval)
        gv = p.get_value(obs, hxs, masks)
    save(name, meta=meta(kind=kind, O=O, A=A, H=H, num_feet=f), params=params, obs=obs.numpy(),
         noise=noise.numpy(), act_value=value.numpy(), act_action=action.numpy(),
         act_logp=logp.numpy(), det_action=a_det.numpy(), det_logp=lp_det.numpy(),
         mean=dist.mean.numpy(), std=dist.stddev.numpy(), eval_action=act_eval.numpy(),
         eval_value=ev_value.numpy(), eval_logp=ev_logp.numpy(), eval_entropy=np.float32(ev_ent.item()),
         get_value=gv.numpy())


# --------------------------------------------------------------------- B. GAE
def fill_rollout(ro, T, N, O, A, F, seed, p_done=0.1, p_bad=0.05):
    g = torch.Generator().manual_seed(seed)
    ro.obs.copy_(torch.randn(T + 1, N, O, generator=g))
    if F:
        ro.obs_feat.copy_(torch.randn(T + 1, N, F, generator=g))
    ro.actions.copy_(torch.randn(T, N, A, generator=g))
    ro.rewards.copy_(torch.randn(T, N, 1, generator=g))
    ro.value_preds.copy_(torch.randn(T + 1, N, 1, generator=g))
    ro.action_log_probs.copy_(torch.randn(T, N, 1, generator=g))
    ro.masks.copy_((torch.rand(T + 1, N, 1, generator=g) > p_done).float())
    ro.bad_masks.copy_((torch.rand(T + 1, N, 1, generator=g) > p_bad).float())


def gen_gae():
    T, N, O, A = 12, 6, 3, 2
    out = {}
    ro = ns.RolloutStorage(T, N, (O,), ns.Box(shape=(A,)), 1, 0)
    fill_rollout(ro, T, N, O, A, 0, 11)
    nv = torch.randn(N, 1, generator=torch.Generator().manual_seed(12))
    out.update(rewards=ro.rewards.numpy().copy(), value_preds=ro.value_preds.numpy().copy(),
               masks=ro.masks.numpy().copy(), bad_masks=ro.bad_masks.numpy().copy(),
               next_value=nv.numpy(), gamma=np.float32(0.99), lam=np.float32(0.95))
    vp0, ret0 = ro.value_preds.clone(), ro.returns.clone()
    for use_gae in (1, 0):
        for proper in (1, 0):
            ro.value_preds.copy_(vp0)
            ro.returns.copy_(ret0)
            ro.compute_returns(nv, bool(use_gae), 0.99, 0.95, bool(proper))
            out[f"returns_gae{use_gae}_proper{proper}"] = ro.returns.numpy().copy()
            out[f"value_preds_gae{use_gae}_proper{proper}"] = ro.value_preds.numpy().copy()
    save("gae", **out)


# --------------------------------------------------------------------- C. PPO
def rollout_from_policy(p, T, N, O, A, F, seed, kind):
    """Fill a RolloutStorage the way the main loop would (act -> insert), synthetic env."""
    ro = ns.RolloutStorage(T, N, (O,), ns.Box(shape=(A,)), 1, F)
    g = torch.Generator().manual_seed(seed)
    ro.obs[0].copy_(torch.randn(N, O, generator=g))
    for step in range(T):
        with torch.no_grad():
            value, action, logp, hxs = p.act(ro.obs[step], ro.recurrent_hidden_states[step], ro.masks[step])
        obs = torch.randn(N, O, generator=g)
        reward = torch.randn(N, 1, generator=g)
        masks = (torch.rand(N, 1, generator=g) > 0.08).float()
        bad = (torch.rand(N, 1, generator=g) > 0.03).float()
        feat = torch.randn(N, F, generator=g) if F else None
        ro.insert(obs, hxs, action, logp, value, reward, masks, bad, feat)
    return ro


def perturb(p, scale, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for q in p.parameters():
            q.add_(scale * torch.randn(q.shape, generator=g))


def rollout_arrays(ro):
    return dict(obs=ro.obs.numpy().copy(), obs_feat=ro.obs_feat.numpy().copy(),
                actions=ro.actions.numpy().copy(), rewards=ro.rewards.numpy().copy(),
                value_preds=ro.value_preds.numpy().copy(), returns=ro.returns.numpy().copy(),
                action_log_probs=ro.action_log_probs.numpy().copy(), masks=ro.masks.numpy().copy(),
                bad_masks=ro.bad_masks.numpy().copy())


def gen_ppo(name, kind, O, A, H, f, T, N, E, M, clip, ecoef, lr, seed, pert=0.02):
    p = make_policy(kind, O, A, H, f, seed)
    torch.manual_seed(seed + 5)
    ro = rollout_from_policy(p, T, N, O, A, 1, seed + 6, kind)  # feat_len 0 breaks a2c/storage.py:171
    with torch.no_grad():
        nv = p.get_value(ro.obs[-1], ro.recurrent_hidden_states[-1], ro.masks[-1]).detach()
    ro.compute_returns(nv, True, 0.99, 0.95, True)
    perturb(p, pert, seed + 7)  # move the policy off the behaviour policy so clipping is exercised
    params0 = flat_params(p)
    agent = ns.PPO(p, clip, E, M, 0.5, ecoef, lr=lr, eps=1e-5, max_grad_norm=0.5)
    arrs = rollout_arrays(ro)
    _REC.clear()
    torch.manual_seed(seed + 8)
    vl, al, de = agent.update(ro)
    perms = np.stack([r for k, r in _REC if k == "randperm"]).astype(np.int64)
    assert perms.shape == (E, T * N)
    adv = ro.returns[:-1] - ro.value_preds[:-1]
    adv = (adv - adv.mean()) / (adv.std() + 1e-5)
    st = agent.optimizer.state_dict()["state"]
    exp_avg = np.concatenate([st[i]["exp_avg"].numpy().reshape(-1) for i in range(len(st))])
    exp_avg_sq = np.concatenate([st[i]["exp_avg_sq"].numpy().reshape(-1) for i in range(len(st))])
    # optimizer param order == module.parameters() order == state_dict order here
    save(name, meta=meta(kind=kind, O=O, A=A, H=H, num_feet=f, T=T, N=N, ppo_epoch=E,
                         num_mini_batch=M, clip_param=clip, entropy_coef=ecoef, lr=lr, eps=1e-5,
                         value_loss_coef=0.5, max_grad_norm=0.5),
         params0=params0, params1=flat_params(p), perms=perms, advantages=adv.numpy(),
         losses=np.array([vl, al, de], np.float64), adam_m=exp_avg, adam_v=exp_avg_sq, **arrs)


# ----------------------------------------------------------------------- D. D
def gen_disc(name, F, Hd, B, Ne, T, N, epochs, seed):
    from torch.utils.data import DataLoader, TensorDataset
    torch.manual_seed(seed)
    D = ns.Discriminator(F, Hd, "cpu")
    params0 = flat_params(D.trunk)
    g = torch.Generator().manual_seed(seed + 1)
    expert = torch.randn(Ne, F, generator=g) * 0.8 + 0.3
    ro = ns.RolloutStorage(T, N, (3,), ns.Box(shape=(2,)), 1, F)
    fill_rollout(ro, T, N, 3, 2, F, seed + 2)
    loader = DataLoader(TensorDataset(expert), batch_size=B, shuffle=True, drop_last=Ne > B)
    out = {}
    for ep in range(epochs):
        _REC.clear()
        torch.manual_seed(seed + 10 + ep)
        loss = D.update_gail_dyn(loader, ro)
        rps = [r for k, r in _REC if k == "randperm"]
        alphas = [r for k, r in _REC if k == "rand"]
        out[f"expert_perm{ep}"] = rps[0].astype(np.int64)   # DataLoader sampler draws first
        out[f"policy_perm{ep}"] = rps[1].astype(np.int64)
        assert len(rps[0]) == Ne and len(rps[1]) == T * N
        out[f"alpha{ep}"] = np.concatenate([a.reshape(-1) for a in alphas]).astype(np.float32)
        out[f"losses{ep}"] = np.array(loss, np.float64)
        out[f"params_after{ep}"] = flat_params(D.trunk)
        out[f"n_steps{ep}"] = np.int64(len(alphas))
    save(name, meta=meta(F=F, Hd=Hd, B=B, Ne=Ne, T=T, N=N, epochs=epochs), params0=params0,
         expert=expert.numpy(), obs_feat=ro.obs_feat.numpy().copy(), **out)


def gen_disc_classic(name, O, A, F, Hd, B, Ne, T, N, dyn, a_dim, use_filt, seed):
    """Discriminator.update (a2c/algo/gail.py:91-152): the state/action variant and its is_gail_dyn row
    assembly; the expert loader yields (state, action) pairs, obsfilt normalises the expert states."""
    from torch.utils.data import DataLoader, TensorDataset
    torch.manual_seed(seed)
    in_dim = (F + a_dim + F) if dyn else (O + A)
    D = ns.Discriminator(in_dim, Hd, "cpu")
    params0 = flat_params(D.trunk)
    g = torch.Generator().manual_seed(seed + 1)
    s_dim = (F + a_dim) if dyn else O
    e_state = torch.randn(Ne, s_dim, generator=g) * 0.8 + 0.3
    e_action = torch.randn(Ne, in_dim - s_dim, generator=g) * 0.5
    ro = ns.RolloutStorage(T, N, (O,), ns.Box(shape=(A,)), 1, F)
    fill_rollout(ro, T, N, O, A, F, seed + 2)
    filt_mean = np.linspace(-0.2, 0.3, s_dim).astype(np.float32)
    filt_std = np.linspace(0.7, 1.4, s_dim).astype(np.float32)
    obsfilt = (lambda x, update=False: np.clip((x - filt_mean) / filt_std, -5.0, 5.0)) if use_filt else None
    loader = DataLoader(TensorDataset(e_state, e_action), batch_size=B, shuffle=True, drop_last=Ne > B)
    _REC.clear()
    torch.manual_seed(seed + 10)
    loss = D.update(loader, ro, obsfilt=obsfilt, is_gail_dyn=dyn, a_dim=a_dim if dyn else None)
    rps = [r for k, r in _REC if k == "randperm"]
    alphas = [r for k, r in _REC if k == "rand"]
    assert len(rps[0]) == Ne and len(rps[1]) == T * N
    save(name, meta=meta(O=O, A=A, F=F, Hd=Hd, B=B, Ne=Ne, T=T, N=N, dyn=int(dyn), a_dim=a_dim, use_filt=int(use_filt)),
         params0=params0, e_state=e_state.numpy(), e_action=e_action.numpy(), filt_mean=filt_mean, filt_std=filt_std,
         obs=ro.obs.numpy().copy(), actions=ro.actions.numpy().copy(), obs_feat=ro.obs_feat.numpy().copy(),
         expert_perm=rps[0].astype(np.int64), policy_perm=rps[1].astype(np.int64),
         alpha=np.concatenate([a.reshape(-1) for a in alphas]).astype(np.float32),
         losses=np.array(loss, np.float64), params_after=flat_params(D.trunk), n_steps=np.int64(len(alphas)))


# ----------------------------------------------------------------- E. relabel
def gen_relabel(name, F, Hd, T, N, seed):
    torch.manual_seed(seed)
    D = ns.Discriminator(F, Hd, "cpu")
    with torch.no_grad():
        for q in D.trunk.parameters():
            q.mul_(3.0)  # widen logits so sigmoid saturation / the 1e-7 terms matter
    params = flat_params(D.trunk)
    rms = ns.RunningMeanStd(shape=())
    out = {}
    for call in range(2):
        ro = ns.RolloutStorage(T, N, (3,), ns.Box(shape=(2,)), 1, F)
        fill_rollout(ro, T, N, 3, 2, F, seed + 3 + call)
        ro.obs_feat.mul_(2.0)
        offset = -0.37 if call == 0 else 0.21
        rets = []
        for step in range(T):  # a2c/main_gail_dyn_ppo.py:275-292
            ro.rewards[step], returns = D.predict_reward_combined(ro.obs_feat[step + 1], 0.99, ro.masks[step], offset=offset)
            if step == 0 and call == 0:
                out["raw_reward0"] = ro.rewards[0].numpy().copy()
            rms.update(returns.view(-1).cpu().numpy())
            rews = ro.rewards[step].view(-1).cpu().numpy()
            rews = np.clip(rews / np.sqrt(rms.var + 1e-7), -10.0, 10.0)
            ro.rewards[step] = torch.Tensor(rews).view(-1, 1)
            rets.append(returns.numpy().copy())
        out[f"obs_feat{call}"] = ro.obs_feat.numpy().copy()
        out[f"masks{call}"] = ro.masks.numpy().copy()
        out[f"offset{call}"] = np.float64(offset)
        out[f"rewards{call}"] = ro.rewards.numpy().copy()
        out[f"d_returns{call}"] = D.returns.numpy().copy()
        out[f"rms{call}"] = np.array([rms.mean, rms.var, rms.count], np.float64)
    save(name, meta=meta(F=F, Hd=Hd, T=T, N=N, gamma=0.99), params=params, **out)


def gen_rms():
    rms = ns.RunningMeanStd(shape=())
    g = np.random.RandomState(5)
    xs, states = [], []
    for i in range(5):
        x = (g.randn(17) * (1 + i) + i).astype(np.float32)
        rms.update(x)
        xs.append(x)
        states.append([rms.mean, rms.var, rms.count])
    save("rms", xs=np.stack(xs), states=np.array(states, np.float64))


# ----------------------------------------------------- F. full outer iteration
def gen_iteration(name, kind, O, A, H, f, F, Hd, T, N, B, Ne, E, M, Ed, iters, seed):
    """a2c/main_gail_dyn_ppo.py:239-304 on a synthetic rollout source, `iters` outer iterations."""
    from torch.utils.data import DataLoader, TensorDataset
    p = make_policy(kind, O, A, H, f, seed)
    torch.manual_seed(seed + 1)
    D = ns.Discriminator(F, Hd, "cpu")
    agent = ns.PPO(p, 0.2, E, M, 0.5, 0.0, lr=3e-4, eps=1e-5, max_grad_norm=0.5)
    g = torch.Generator().manual_seed(seed + 2)
    expert = torch.randn(Ne, F, generator=g)
    loader = DataLoader(TensorDataset(expert), batch_size=B, shuffle=True, drop_last=Ne > B)
    gail_tar_length = 37.0
    rms = ns.RunningMeanStd(shape=())
    ro = ns.RolloutStorage(T, N, (O,), ns.Box(shape=(A,)), 1, F)
    ro.obs[0].copy_(torch.randn(N, O, generator=g))
    out = dict(pi_params0=flat_params(p), d_params0=flat_params(D.trunk), expert=expert.numpy(),
               obs0=ro.obs[0].numpy().copy())
    for j in range(iters):
        # synthetic rollout fill: act -> fake env -> insert   (:209-236)
        noises, envs = [], []
        for step in range(T):
            with torch.no_grad():
                torch.manual_seed(seed + 100 + 1000 * j + step)
                noises.append(torch.randn(N, A).numpy())
                torch.manual_seed(seed + 100 + 1000 * j + step)
                value, action, logp, hxs = p.act(ro.obs[step], ro.recurrent_hidden_states[step], ro.masks[step])
            obs = torch.randn(N, O, generator=g)
            reward = torch.randn(N, 1, generator=g)
            masks = (torch.rand(N, 1, generator=g) > 0.1).float()
            bad = (torch.rand(N, 1, generator=g) > 0.05).float()
            feat = torch.randn(N, F, generator=g)
            envs.append((obs.numpy().copy(), reward.numpy().copy(), masks.numpy().copy(),
                         bad.numpy().copy(), feat.numpy().copy()))
            ro.insert(obs, hxs, action, logp, value, reward, masks, bad, feat)
        out[f"it{j}_noise"] = np.stack(noises)
        for i, nm in enumerate(("env_obs", "env_reward", "env_masks", "env_bad", "env_feat")):
            out[f"it{j}_{nm}"] = np.stack([e[i] for e in envs])
        out[f"it{j}_actions"] = ro.actions.numpy().copy()
        out[f"it{j}_action_log_probs"] = ro.action_log_probs.numpy().copy()
        out[f"it{j}_value_preds_rollout"] = ro.value_preds.numpy().copy()
        with torch.no_grad():
            nv = p.get_value(ro.obs[-1], ro.recurrent_hidden_states[-1], ro.masks[-1]).detach()
        out[f"it{j}_next_value"] = nv.numpy().copy()
        dl = []
        for ep in range(Ed):  # :255-256
            _REC.clear()
            torch.manual_seed(seed + 500 + 10 * j + ep)
            dl.append(D.update_gail_dyn(loader, ro))
            rps = [r for k, r in _REC if k == "randperm"]
            out[f"it{j}_d{ep}_expert_perm"] = rps[0].astype(np.int64)
            out[f"it{j}_d{ep}_policy_perm"] = rps[1].astype(np.int64)
            out[f"it{j}_d{ep}_alpha"] = np.concatenate([r.reshape(-1) for k, r in _REC if k == "rand"]).astype(np.float32)
        out[f"it{j}_d_losses"] = np.array(dl, np.float64)
        out[f"it{j}_d_params"] = flat_params(D.trunk)
        # alive bonus :258-271
        num_of_dones = (1.0 - ro.masks).sum().cpu().numpy() + N / 2
        num_of_expert_dones = (T * N) / gail_tar_length
        d_sa = 1 - num_of_dones / (num_of_dones + num_of_expert_dones)
        r_sa = np.log(d_sa) - np.log(1 - d_sa)
        out[f"it{j}_r_sa"] = np.float64(r_sa)
        for step in range(T):  # :275-292
            ro.rewards[step], returns = D.predict_reward_combined(ro.obs_feat[step + 1], 0.99, ro.masks[step], offset=-r_sa)
            rms.update(returns.view(-1).cpu().numpy())
            rews = ro.rewards[step].view(-1).cpu().numpy()
            rews = np.clip(rews / np.sqrt(rms.var + 1e-7), -10.0, 10.0)
            ro.rewards[step] = torch.Tensor(rews).view(-1, 1)
        out[f"it{j}_rewards"] = ro.rewards.numpy().copy()
        out[f"it{j}_rms"] = np.array([rms.mean, rms.var, rms.count], np.float64)
        out[f"it{j}_d_returns"] = D.returns.numpy().copy()
        ro.compute_returns(nv, True, 0.99, 0.95, True)  # :299
        out[f"it{j}_returns"] = ro.returns.numpy().copy()
        _REC.clear()
        torch.manual_seed(seed + 900 + j)
        losses = agent.update(ro)  # :302
        out[f"it{j}_ppo_perms"] = np.stack([r for k, r in _REC if k == "randperm"]).astype(np.int64)
        out[f"it{j}_ppo_losses"] = np.array(losses, np.float64)
        out[f"it{j}_pi_params"] = flat_params(p)
        ro.after_update()  # :304
    save(name, meta=meta(kind=kind, O=O, A=A, H=H, num_feet=f, F=F, Hd=Hd, T=T, N=N, B=B, Ne=Ne,
                         ppo_epoch=E, num_mini_batch=M, gail_epoch=Ed, iters=iters,
                         gail_tar_length=gail_tar_length, gamma=0.99, gae_lambda=0.95), **out)


def save_legacy_without_source(obj, path):
    """torch's legacy (non-zip) container records the SOURCE TEXT of every pickled nn.Module class next to the object
    (torch.serialization: persistent_id -> inspect.getsource) so a later load can warn about code drift.  A fixture
    is data, not reference source: with getsourcefile failing, torch stores None for the source instead (its own
    documented fallback, "Couldn't retrieve source code for container") and the container is otherwise unchanged."""
    import inspect
    import warnings
    real = inspect.getsourcefile

    def no_source(_obj):
        raise OSError("source capture disabled for fixtures")

    inspect.getsourcefile = no_source
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            torch.save(obj, path, _use_new_zipfile_serialization=False)
    finally:
        inspect.getsourcefile = real
    blob = open(path, "rb").read()
    for needle in (b"class Policy(nn.Module)", b"class MLPBase", b"class DiagGaussian", b"class AddBias", b"def forward("):
        assert needle not in blob, f"{path} still embeds reference source ({needle!r})"


# ------------------------------------------- G. checkpoint files and the expert wire format
def gen_checkpoints():
    """Whole-module checkpoints exactly as the reference writes them (a2c/main.py:260-269,
    a2c/main_gail_dyn_ppo.py:319): the .pt files are data (pickled objects by reference to class names)."""
    import pickle
    from torch.utils.data import DataLoader, TensorDataset
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
    g = torch.Generator().manual_seed(77)
    # MLP policy + observation statistics, legacy container (what the shipped trained_models_*/ppo/*.pt use)
    p = make_policy("mlp", 11, 3, 64, 1, 700)
    rms = ns.RunningMeanStd(shape=(11,))
    rms.update(np.random.RandomState(3).randn(50, 11) * 2.0 + 0.5)
    save_legacy_without_source([p, rms], os.path.join(out_dir, "ckpt_policy_mlp.pt"))
    obs = torch.randn(6, 11, generator=g)
    with torch.no_grad():
        v, a, lp, _ = p.act(obs, None, None, deterministic=True)
    save("ckpt_policy_mlp", meta=meta(kind="mlp", O=11, A=3, H=64, f=1), flat=flat_params(p), obs=obs.numpy(), value=v.numpy(),
         action=a.numpy(), logp=lp.numpy(), rms_mean=rms.mean, rms_var=rms.var, rms_count=np.float64(rms.count))
    # split policy, zip container (torch >= 1.6 default)
    p = make_policy("split", 14, 7, 32, 1, 710)
    torch.save([p, None], os.path.join(out_dir, "ckpt_policy_split.pt"))
    obs = torch.randn(5, 14, generator=g)
    with torch.no_grad():
        v, a, lp, _ = p.act(obs, None, None, deterministic=True)
    save("ckpt_policy_split", meta=meta(kind="split", O=14, A=7, H=32, f=1), flat=flat_params(p), obs=obs.numpy(),
         value=v.numpy(), action=a.numpy(), logp=lp.numpy())
    # discriminator after one epoch (Adam state, running returns)
    torch.manual_seed(720)
    D = ns.Discriminator(7, 16, "cpu")
    expert = torch.randn(24, 7, generator=g)
    ro = ns.RolloutStorage(4, 8, (3,), ns.Box(shape=(2,)), 1, 7)
    fill_rollout(ro, 4, 8, 3, 2, 7, 721)
    D.update_gail_dyn(DataLoader(TensorDataset(expert), batch_size=8, shuffle=True, drop_last=True), ro)
    D.predict_reward_combined(ro.obs_feat[1], 0.99, ro.masks[0], offset=0.1)
    torch.save(D, os.path.join(out_dir, "ckpt_disc.pt"))
    ps = list(D.trunk.parameters())
    st = D.optimizer.state
    save("ckpt_disc", meta=meta(F=7, Hd=16), flat=flat_params(D.trunk),
         adam_m=np.concatenate([st[q]["exp_avg"].numpy().reshape(-1) for q in ps]),
         adam_v=np.concatenate([st[q]["exp_avg_sq"].numpy().reshape(-1) for q in ps]),
         step=np.int64(int(st[ps[0]]["step"])), returns=D.returns.numpy())
    # expert trajectories in the collector's wire format (a2c/collect_tarsim_traj.py:218-265): L = 3 past steps
    r = np.random.RandomState(11)
    trajs = {}
    for ti in range(3):
        rows = []
        for _ in range(5 + ti):
            rows.append([list(r.randn(4)) for _ in range(3)] + [list(r.randn(4)) for _ in range(3)] + [list(r.randn(4))])  # equal widths: the reference stacks the tuples with np.array, which modern numpy only accepts for a regular shape
        trajs[ti] = rows
    blob = pickle.dumps(trajs, protocol=2)
    with open(os.path.join(out_dir, "expert_trajs.pkl"), "wb") as f:
        f.write(blob)
    gu = ns.gan_utils
    _REC.clear()
    torch.manual_seed(5)
    ri = torch.randint
    drawn = []
    torch.randint = lambda *a_, **k_: (drawn.append(ri(*a_, **k_)) or drawn[-1])
    try:
        sas = gu.load_sas_wpast_from_pickle(os.path.join(out_dir, "expert_trajs.pkl"), downsample_freq=2)
    finally:
        torch.randint = ri
    merged = gu.select_and_merge_sas(sas, s_idx=np.array([0, 2]), a_idx=np.array([0, 1]))
    one = gu.select_and_merge_sas([np.asarray(x[0]) for x in sas], s_idx=np.array([0]), a_idx=np.array([0]))
    save("expert_trajs", start_idx=drawn[0].numpy().astype(np.int64), merged=merged, one=one,
         n_items=np.int64(len(sas)), first_item=np.asarray(sas[0]))


# ------------------------------------------- H. the plain-PPO caller: a2c/main.py:78-88,199-257 (BASELINE.json configs[4])
def gen_refine(name, O, A, H, T, N, E, M, clip, lr, num_updates, iters, logstd, seed):
    """Policy refinement as train_laika_power.sh:7 runs it: warm start from a behaviour checkpoint, reset_critic,
    reset_variance(--warm-start-logstd), then per outer iteration linear LR decay -> rollout -> get_value ->
    compute_returns -> PPO.update -> after_update.  The rollout's feature slot is the observation itself
    (replace_obs_with_feat identity, a2c/main.py:168-169,218)."""
    out_dir = OUT
    behaviour = make_policy("mlp", O, A, H, 1, seed)
    warm_path = os.path.join(out_dir, name + "_warm.pt")
    torch.save([behaviour, None], warm_path)                      # zip container: pickled objects + tensors, no source text
    blob = open(warm_path, "rb").read()
    assert b"class Policy(nn.Module)" not in blob and b"def forward(" not in blob
    out = dict(behaviour_params=flat_params(behaviour))
    actor_critic, _ = torch.load(warm_path, map_location="cpu", weights_only=False)   # a2c/main.py:81-83
    torch.manual_seed(seed + 1)
    actor_critic.reset_critic((O,))                               # :85
    actor_critic.reset_variance(ns.Box(shape=(A,)), logstd)       # :86-87
    out["pi_params0"] = flat_params(actor_critic)
    agent = ns.PPO(actor_critic, clip, E, M, 0.5, 0.0, lr=lr, eps=1e-5, max_grad_norm=0.5)
    g = torch.Generator().manual_seed(seed + 2)
    ro = ns.RolloutStorage(T, N, (O,), ns.Box(shape=(A,)), 1, O)
    obs0 = torch.randn(N, O, generator=g)
    ro.obs[0].copy_(obs0)
    ro.obs_feat[0].copy_(obs0)
    out["obs0"] = obs0.numpy().copy()
    lrs = []
    for j in range(iters):
        ns.a2c_utils.update_linear_schedule(agent.optimizer, j, num_updates, lr)      # :201-205
        lrs.append(agent.optimizer.param_groups[0]["lr"])
        noises, envs = [], []
        for step in range(T):                                     # :207-244
            with torch.no_grad():
                torch.manual_seed(seed + 100 + 1000 * j + step)
                noises.append(torch.randn(N, A).numpy())
                torch.manual_seed(seed + 100 + 1000 * j + step)
                value, action, logp, hxs = actor_critic.act(ro.obs[step], ro.recurrent_hidden_states[step], ro.masks[step])
            obs = torch.randn(N, O, generator=g)
            reward = torch.randn(N, 1, generator=g)
            masks = (torch.rand(N, 1, generator=g) > 0.1).float()
            bad = (torch.rand(N, 1, generator=g) > 0.05).float()
            envs.append((obs.numpy().copy(), reward.numpy().copy(), masks.numpy().copy(), bad.numpy().copy()))
            ro.insert(obs, hxs, action, logp, value, reward, masks, bad, obs.clone())
        out[f"it{j}_noise"] = np.stack(noises)
        for i, nm in enumerate(("env_obs", "env_reward", "env_masks", "env_bad")):
            out[f"it{j}_{nm}"] = np.stack([e[i] for e in envs])
        out[f"it{j}_actions"] = ro.actions.numpy().copy()
        out[f"it{j}_action_log_probs"] = ro.action_log_probs.numpy().copy()
        out[f"it{j}_value_preds_rollout"] = ro.value_preds.numpy().copy()
        with torch.no_grad():
            nv = actor_critic.get_value(ro.obs[-1], ro.recurrent_hidden_states[-1], ro.masks[-1]).detach()   # :246-249
        out[f"it{j}_next_value"] = nv.numpy().copy()
        ro.compute_returns(nv, True, 0.99, 0.95, True)            # :251-252
        out[f"it{j}_returns"] = ro.returns.numpy().copy()
        _REC.clear()
        torch.manual_seed(seed + 900 + j)
        losses = agent.update(ro)                                 # :254
        out[f"it{j}_ppo_perms"] = np.stack([r for k, r in _REC if k == "randperm"]).astype(np.int64)
        out[f"it{j}_ppo_losses"] = np.array(losses, np.float64)
        out[f"it{j}_pi_params"] = flat_params(actor_critic)
        ro.after_update()                                         # :256
    save(name, meta=meta(kind="mlp", O=O, A=A, H=H, num_feet=1, T=T, N=N, ppo_epoch=E, num_mini_batch=M, clip_param=clip,
                         lr=lr, num_updates=num_updates, iters=iters, warm_start_logstd=logstd, gamma=0.99, gae_lambda=0.95),
         lrs=np.array(lrs, np.float64), **out)


def gen_vecnormalize():
    """VecNormalize(ob=False, ret=True).step_wait reward scaling (a2c/envs.py:120-125, vec_normalize.py:50-58) on a
    scripted vectorised environment: inputs (raw rewards, dones) and outputs (scaled rewards, ret, ret_rms) per step."""
    from third_party.a2c_ppo_acktr.envs import VecNormalize
    n, steps = 6, 40
    r = np.random.RandomState(17)
    script = [((r.randn(n) * (1.0 + 0.2 * t)).astype(np.float32), r.rand(n) < 0.15) for t in range(steps)]

    class Scripted:
        num_envs = n
        observation_space = ns.Box(shape=(3,))
        action_space = ns.Box(shape=(2,))
        t = 0

        def step_wait(self):
            rews, news = script[self.t]
            self.t += 1
            return np.zeros((n, 3), np.float32), rews.copy(), news.copy(), [{} for _ in range(n)]

        def reset(self):
            return np.zeros((n, 3), np.float32)

    vn = VecNormalize(Scripted(), gamma=0.99, ob=False)
    vn.reset()
    outs, rets, states = [], [], []
    for t in range(steps):
        _, rews, _, _ = vn.step_wait()
        outs.append(np.asarray(rews).copy())
        rets.append(vn.ret.copy())
        states.append([vn.ret_rms.mean, vn.ret_rms.var, vn.ret_rms.count])
    save("vecnormalize", raw=np.stack([s_[0] for s_ in script]), news=np.stack([s_[1] for s_ in script]),
         scaled=np.stack(outs), ret=np.stack(rets), rms=np.array(states, np.float64), gamma=np.float64(0.99))


def gen_ffgen():
    """RolloutStorage.feed_forward_generator (a2c/storage.py:144-192): the 10-tuples the reference yields for PPO
    (num_mini_batch, with advantages) and for the discriminator (mini_batch_size, advantages None, ragged tail dropped),
    with the permutation each call drew."""
    global _REC
    T, N, O, A, F = 5, 4, 3, 2, 4
    ro = ns.RolloutStorage(T, N, (O,), ns.Box(shape=(A,)), 1, F)
    fill_rollout(ro, T, N, O, A, F, 700)
    ro.returns.copy_(torch.randn(T + 1, N, 1, generator=torch.Generator().manual_seed(701)))
    adv = torch.randn(T, N, 1, generator=torch.Generator().manual_seed(702))
    out = dict(rollout_arrays(ro), advantages=adv.numpy().copy())
    out["recurrent_hidden_states"] = ro.recurrent_hidden_states.numpy().copy()
    names = ("obs", "hxs", "actions", "value_preds", "returns", "masks", "old_logp", "adv", "obs_feat", "next_obs_feat")
    for tag, kw, a in (("ppo", dict(num_mini_batch=3), adv), ("disc", dict(mini_batch_size=8), None)):
        torch.manual_seed(710)
        _REC = []
        batches = list(ro.feed_forward_generator(a, **kw))
        out[f"{tag}_perm"] = np.concatenate([r for k, r in _REC if k == "randperm"]).astype(np.int64)
        out[f"{tag}_n_batches"] = np.int64(len(batches))
        for b, tup in enumerate(batches):
            for nm, t in zip(names, tup):
                if t is not None:
                    out[f"{tag}_b{b}_{nm}"] = t.numpy().copy()
    save("ffgen", meta=meta(T=T, N=N, O=O, A=A, F=F, num_mini_batch=3, mini_batch_size=8), **out)


def gen_predict_reward():
    """Discriminator.predict_reward(state, action, gamma, masks, offset) (a2c/algo/gail.py:195-199), two consecutive calls
    (the second continues self.returns)."""
    torch.manual_seed(720)
    S, Ad, Hd, n = 6, 3, 16, 9
    D = ns.Discriminator(S + Ad, Hd, "cpu")
    g = torch.Generator().manual_seed(721)
    out = dict(params=flat_params(D.trunk))
    for c in range(2):
        st, ac = torch.randn(n, S, generator=g), torch.randn(n, Ad, generator=g)
        mk = (torch.rand(n, 1, generator=g) > 0.3).float()
        rew, ret = D.predict_reward(st, ac, 0.97, mk, offset=0.25 * c)
        out.update({f"state{c}": st.numpy(), f"action{c}": ac.numpy(), f"masks{c}": mk.numpy(), f"reward{c}": rew.numpy().copy(),
                    f"returns{c}": ret.numpy().copy()})
    save("predict_reward", meta=meta(S=S, A=Ad, Hd=Hd, n=n, gamma=0.97), **out)


def gen_grad_pen():
    """Discriminator.compute_grad_pen_combined / compute_grad_pen (a2c/algo/gail.py:53-89): the penalty's value for given
    expert / policy rows, with the torch.rand(n, 1) draw recorded."""
    out, m = {}, {}
    for tag, F, Hd, n, split, seed in (("small", 9, 16, 11, 6, 730), ("northstar", 86, 100, 128, 74, 740)):
        torch.manual_seed(seed)
        D = ns.Discriminator(F, Hd, "cpu")
        with torch.no_grad():
            for q in D.trunk.parameters():      # default init keeps ||dD/dx|| small: spread the norms around 1
                q.mul_(2.5)
        g = torch.Generator().manual_seed(seed + 1)
        e, p = torch.randn(n, F, generator=g) * 0.8 + 0.1, torch.randn(n, F, generator=g)
        _REC.clear()
        torch.manual_seed(seed + 2)
        v = D.compute_grad_pen_combined(e, p, 10.0)
        alpha = [r for k, r in _REC if k == "rand"][0].reshape(-1).astype(np.float32)
        _REC.clear()
        torch.manual_seed(seed + 3)
        v2 = D.compute_grad_pen(e[:, :split], e[:, split:], p[:, :split], p[:, split:], lambda_=4.0)
        alpha2 = [r for k, r in _REC if k == "rand"][0].reshape(-1).astype(np.float32)
        out.update({f"{tag}_params": flat_params(D.trunk), f"{tag}_expert": e.numpy(), f"{tag}_policy": p.numpy(), f"{tag}_alpha": alpha,
                    f"{tag}_value": np.float32(v.item()), f"{tag}_alpha2": alpha2, f"{tag}_value2": np.float32(v2.item())})
        m[tag] = dict(F=F, Hd=Hd, n=n, split=split)
    save("grad_pen", meta=meta(**m), **out)


if __name__ == "__main__":
    if len(sys.argv) > 1:   # regenerate selected fixtures only: python tools/gen_golden.py refine vecnormalize checkpoints
        for what in sys.argv[1:]:
            if what == "refine":
                gen_refine("iter_refine", O=111, A=12, H=64, T=8, N=16, E=2, M=8, clip=0.1, lr=1.5e-4, num_updates=4, iters=2,
                           logstd=-1.3, seed=600)
            elif what == "refine_h100":
                # the same loop from a 100-unit behaviour policy: reset_critic then leaves a 64-unit critic beside the
                # 100-unit actor (a2c/model.py:80-87 hard-codes 64)
                gen_refine("iter_refine_h100", O=20, A=5, H=100, T=8, N=16, E=2, M=4, clip=0.1, lr=1.5e-4, num_updates=4, iters=2,
                           logstd=-1.3, seed=650)
            elif what == "vecnormalize":
                gen_vecnormalize()
            elif what == "checkpoints":
                gen_checkpoints()
            elif what == "closures":
                gen_ffgen()
                gen_predict_reward()
            elif what == "grad_pen":
                gen_grad_pen()
            else:
                raise SystemExit(f"unknown fixture group {what}")
        sys.exit(0)
    gen_policy("policy_mlp_tiny", "mlp", 5, 2, 8, 1, 16, 100)
    gen_policy("policy_mlp_northstar", "mlp", 47, 12, 64, 1, 24, 110)
    gen_policy("policy_mlp_hopper", "mlp", 11, 3, 64, 1, 8, 120)
    gen_policy("policy_split_hopper", "split", 14, 7, 100, 1, 20, 130)
    gen_policy("policy_split_laikago", "split", 64, 28, 100, 4, 12, 140)
    gen_policy("policy_split_tiny", "split", 6, 7, 12, 1, 9, 150)
    gen_gae()
    gen_rms()
    gen_ppo("ppo_mlp_tiny", "mlp", 5, 2, 8, 1, T=8, N=4, E=2, M=2, clip=0.2, ecoef=0.01, lr=3e-4, seed=200, pert=0.05)
    gen_ppo("ppo_mlp_northstar", "mlp", 47, 12, 64, 1, T=16, N=8, E=3, M=4, clip=0.2, ecoef=0.0, lr=3e-4, seed=210)
    gen_ppo("ppo_mlp_onestep", "mlp", 11, 3, 64, 1, T=10, N=6, E=1, M=1, clip=0.1, ecoef=0.01, lr=1.5e-4, seed=220, pert=0.05)
    gen_ppo("ppo_split_hopper", "split", 14, 7, 100, 1, T=12, N=8, E=2, M=3, clip=0.2, ecoef=0.0, lr=3e-4, seed=230)
    gen_ppo("ppo_split_laikago", "split", 64, 28, 100, 4, T=8, N=8, E=2, M=2, clip=0.2, ecoef=0.01, lr=3e-4, seed=240)
    gen_disc("disc_tiny", F=7, Hd=16, B=8, Ne=40, T=4, N=8, epochs=2, seed=300)
    gen_disc("disc_northstar", F=86, Hd=100, B=128, Ne=400, T=8, N=64, epochs=2, seed=310)
    gen_disc("disc_hopper", F=25, Hd=100, B=128, Ne=300, T=16, N=16, epochs=1, seed=320)
    gen_disc("disc_single_batch", F=7, Hd=16, B=8, Ne=8, T=4, N=8, epochs=1, seed=330)  # Ne == B: drop_last False, one batch (Ne < B raises in the reference)
    gen_disc_classic("disc_classic_sa", O=11, A=3, F=5, Hd=32, B=16, Ne=70, T=6, N=8, dyn=False, a_dim=0, use_filt=True, seed=340)
    gen_disc_classic("disc_classic_dyn", O=9, A=4, F=6, Hd=32, B=16, Ne=48, T=6, N=8, dyn=True, a_dim=3, use_filt=False, seed=350)
    gen_relabel("relabel_tiny", F=7, Hd=16, T=6, N=5, seed=400)
    gen_relabel("relabel_northstar", F=86, Hd=100, T=8, N=32, seed=410)
    gen_iteration("iter_mlp", "mlp", 47, 12, 64, 1, F=86, Hd=100, T=8, N=16, B=32, Ne=200, E=2, M=2, Ed=2, iters=2, seed=500)
    gen_checkpoints()
    gen_iteration("iter_split", "split", 14, 7, 100, 1, F=25, Hd=100, T=8, N=16, B=32, Ne=100, E=2, M=2, Ed=2, iters=2, seed=510)
    gen_refine("iter_refine", O=111, A=12, H=64, T=8, N=16, E=2, M=8, clip=0.1, lr=1.5e-4, num_updates=4, iters=2, logstd=-1.3, seed=600)
    gen_refine("iter_refine_h100", O=20, A=5, H=100, T=8, N=16, E=2, M=4, clip=0.1, lr=1.5e-4, num_updates=4, iters=2, logstd=-1.3, seed=650)
    gen_vecnormalize()
    gen_ffgen()
    gen_predict_reward()
    gen_grad_pen()
