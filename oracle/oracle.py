"""ctypes binding of oracle/libsg_oracle.so (the CPU restatement in sg_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never by the product package simgan_amd/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
# oracle/oracle64.py executes this file a second time with REAL_BITS = 64: the float64 ARBITER (sg_oracle_f64.c), same functions
# on float64 arrays.  32 (the default, `from oracle import oracle`) is the parity oracle.
REAL_BITS = globals().get("REAL_BITS", 32)
_R = np.float32 if REAL_BITS == 32 else np.float64
_CR = C.c_float if REAL_BITS == 32 else C.c_double
_SO, _SRC = ("libsg_oracle.so", "sg_oracle.c") if REAL_BITS == 32 else ("libsg_oracle64.so", "sg_oracle_f64.c")

KIND_MLP, KIND_SPLIT = 0, 1


class PolicyDims(C.Structure):
    _fields_ = [("kind", C.c_int), ("O", C.c_int), ("A", C.c_int), ("H", C.c_int),
                ("num_feet", C.c_int), ("Hc", C.c_int)]


class PPOCfg(C.Structure):
    _fields_ = [("clip_param", _CR), ("ppo_epoch", C.c_int), ("num_mini_batch", C.c_int),
                ("value_loss_coef", _CR), ("entropy_coef", _CR), ("lr", _CR),
                ("eps", _CR), ("max_grad_norm", _CR),
                ("use_clipped_value_loss", C.c_int)]


def build(force=False):
    so = os.path.join(_HERE, _SO)
    srcs = [os.path.join(_HERE, "sg_oracle.c"), os.path.join(_HERE, _SRC)]
    if force or not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(x) for x in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B", _SO],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.orc_policy_num_params.restype = C.c_int64
        _LIB.orc_disc_num_params.restype = C.c_int64
        _LIB.orc_clip_grad_norm.restype = _CR
        _LIB.orc_alive_bonus.restype = C.c_double
        _LIB.orc_disc_update.restype = C.c_int
    return _LIB


_FAST = {}


def build_fast_native():
    """bench.py's cpu_baseline leg: compile sg_cpu_fast.c for THE HOST IT IS TIMED ON (-march=native) into a scratch
    directory.  Returns the path, or None when that fails (the in-tree x86-64-v3 build is used then)."""
    import tempfile
    out = os.path.join(tempfile.mkdtemp(prefix="sg_cpu_fast_"), "libsg_cpu_fast_native.so")
    cmd = ["gcc", "-O3", "-fPIC", "-std=gnu99", "-ffast-math", "-march=native", "-fopenmp", "-shared", "-o", out,
           os.path.join(_HERE, "sg_cpu_fast.c"), "-lm"]
    try:
        subprocess.check_call(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        return out
    except (OSError, subprocess.CalledProcessError):
        return None


def fast_lib(native=False):
    """oracle/sg_cpu_fast.c: the batched, vectorised CPU implementation bench.py times as `cpu_baseline` (parity-checked
    against the oracle in tests/test_oracle_golden.py; never the checker itself)."""
    key = "native" if native else "v3"
    if key not in _FAST:
        so = build_fast_native() if native else None
        if so is None:
            so = os.path.join(_HERE, "libsg_cpu_fast.so")
            src = os.path.join(_HERE, "sg_cpu_fast.c")
            if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
                subprocess.check_call(["make", "-C", _HERE, "-B", "libsg_cpu_fast.so"], stdout=subprocess.DEVNULL)
            key2 = "v3"
        else:
            key2 = "native"
        _FAST[key] = (C.CDLL(so), key2)
    return _FAST[key]


def _f(a):
    a = np.ascontiguousarray(a, dtype=_R)
    return a, a.ctypes.data_as(C.POINTER(_CR))


def _f32(a):   # the vectorised CPU baseline (sg_cpu_fast.c) is float32 whatever REAL_BITS says
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(C.POINTER(C.c_float))


def _fp32(a):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _fp(a):  # in-place float32 array -> pointer (must already be contiguous float32)
    assert a.dtype == _R and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(_CR))


def _i64(a):
    a = np.ascontiguousarray(a, dtype=np.int64)
    return a, a.ctypes.data_as(C.POINTER(C.c_int64))


def dims(kind, O, A, H, num_feet=1, Hc=0):
    """Hc: hidden size of the critic trunk when it differs from H (Policy.reset_critic, a2c/model.py:80-87); 0 = H."""
    return PolicyDims(kind, O, A, H, num_feet, 0 if Hc == H else Hc)


def policy_num_params(d):
    return int(lib().orc_policy_num_params(C.byref(d)))


def disc_num_params(F, Hd):
    return int(lib().orc_disc_num_params(F, Hd))


def policy_forward(d, params, obs):
    obs, po = _f(obs)
    n = obs.shape[0]
    params, pp = _f(params)
    value = np.empty((n, 1), _R)
    mean = np.empty((n, d.A), _R)
    logstd = np.empty((n, d.A), _R)
    lib().orc_policy_forward(C.byref(d), pp, po, n, _fp(value), _fp(mean), _fp(logstd))
    return value, mean, logstd


def policy_act(d, params, obs, noise=None):
    obs, po = _f(obs)
    n = obs.shape[0]
    params, pp = _f(params)
    value = np.empty((n, 1), _R)
    action = np.empty((n, d.A), _R)
    logp = np.empty((n, 1), _R)
    if noise is None:
        pn = None
    else:
        noise, pn = _f(noise)
    lib().orc_policy_act(C.byref(d), pp, po, n, pn, _fp(value), _fp(action), _fp(logp))
    return value, action, logp


def policy_evaluate(d, params, obs, action):
    obs, po = _f(obs)
    action, pa = _f(action)
    n = obs.shape[0]
    params, pp = _f(params)
    value = np.empty((n, 1), _R)
    logp = np.empty((n, 1), _R)
    ent = _CR(0)
    lib().orc_policy_evaluate(C.byref(d), pp, po, pa, n, _fp(value), _fp(logp), C.byref(ent))
    return value, logp, float(ent.value)


def compute_returns(rewards, value_preds, masks, bad_masks, next_value, use_gae, gamma, lam,
                    proper_time_limits):
    """rewards [T,N]; value_preds/masks/bad_masks [T+1,N]; returns (returns[T+1,N], value_preds')."""
    rewards, pr = _f(rewards)
    T, N = rewards.shape[:2]
    vp = np.array(value_preds, dtype=_R, copy=True).reshape(T + 1, N)
    ret = np.zeros((T + 1, N), _R)
    masks, pm = _f(masks)
    bad_masks, pb = _f(bad_masks)
    next_value, pn = _f(next_value)
    lib().orc_compute_returns(T, N, pr, _fp(vp), _fp(ret), pm, pb, pn, int(use_gae),
                              _CR(gamma), _CR(lam), int(proper_time_limits))
    return ret, vp


def advantages(returns, value_preds):
    returns, pr = _f(returns)
    value_preds, pv = _f(value_preds)
    n = returns.size
    adv = np.empty(n, _R)
    lib().orc_advantages(pr, pv, C.c_int64(n), _fp(adv))
    return adv


def ppo_cfg(clip_param=0.2, ppo_epoch=10, num_mini_batch=16, value_loss_coef=0.5,
            entropy_coef=0.0, lr=3e-4, eps=1e-5, max_grad_norm=0.5, use_clipped_value_loss=True):
    return PPOCfg(clip_param, ppo_epoch, num_mini_batch, value_loss_coef, entropy_coef, lr, eps,
                  max_grad_norm, int(use_clipped_value_loss))


class AdamState:
    def __init__(self, n):
        self.m = np.zeros(n, _R)
        self.v = np.zeros(n, _R)
        self.t = C.c_int64(0)


def ppo_grad_rows(d, params, cfg, obs, actions, value_preds, returns, old_logp, adv, rows, inv_B):
    """Gradient sum + loss sums over `rows` (flattened t*N+n ids). Returns (G, sums[3])."""
    params, pp = _f(params)
    obs, po = _f(obs)
    actions, pa = _f(actions)
    value_preds, pv = _f(value_preds)
    returns, pr = _f(returns)
    old_logp, pl = _f(old_logp)
    adv, pad = _f(adv)
    rows, prow = _i64(rows)
    G = np.zeros(params.size, _R)
    sums = (C.c_double * 3)(0, 0, 0)
    lib().orc_ppo_grad_rows(C.byref(d), pp, C.byref(cfg), po, pa, pv, pr, pl, pad, prow,
                            int(rows.size), _CR(inv_B), _fp(G), sums)
    return G, np.array(list(sums))


def ppo_grad_rows_fast(d, params, cfg, obs, actions, value_preds, returns, old_logp, adv, rows, inv_B, native=False, n_threads=1):
    """oracle/sg_cpu_fast.c:fast_ppo_grad_rows -- same contract as ppo_grad_rows, batched GEMMs (n_threads > 1: rows split
    over an OpenMP team)."""
    params, pp = _f32(params)
    obs, po = _f32(obs)
    actions, pa = _f32(actions)
    value_preds, pv = _f32(value_preds)
    returns, pr = _f32(returns)
    old_logp, pl = _f32(old_logp)
    adv, pad = _f32(adv)
    rows, prow = _i64(rows)
    G = np.zeros(params.size, np.float32)
    sums = (C.c_double * 3)(0, 0, 0)
    if n_threads > 1:
        fast_lib(native)[0].fast_ppo_grad_rows_mt(C.byref(d), pp, C.byref(cfg), po, pa, pv, pr, pl, pad, prow, int(rows.size),
                                                  C.c_float(inv_B), _fp32(G), sums, int(n_threads), C.c_int64(params.size))
    else:
        fast_lib(native)[0].fast_ppo_grad_rows(C.byref(d), pp, C.byref(cfg), po, pa, pv, pr, pl, pad, prow,
                                               int(rows.size), C.c_float(inv_B), _fp32(G), sums)
    return G, np.array(list(sums))


def disc_grad_rows_fast(F, Hd, params, expert_rows, policy_rows, alpha, inv_B, lambda_=10.0, native=False, n_threads=1):
    """oracle/sg_cpu_fast.c:fast_disc_grad_rows -- same contract as disc_grad_rows, batched GEMMs."""
    params, pp = _f32(params)
    expert_rows, pe = _f32(expert_rows)
    policy_rows, ppol = _f32(policy_rows)
    alpha, pal = _f32(alpha)
    nb = expert_rows.shape[0]
    G = np.zeros(params.size, np.float32)
    sums = (C.c_double * 3)(0, 0, 0)
    if n_threads > 1:
        fast_lib(native)[0].fast_disc_grad_rows_mt(F, Hd, pp, pe, ppol, pal, nb, C.c_float(inv_B), C.c_float(lambda_), _fp32(G), sums, int(n_threads))
    else:
        fast_lib(native)[0].fast_disc_grad_rows(F, Hd, pp, pe, ppol, pal, nb, C.c_float(inv_B), C.c_float(lambda_), _fp32(G), sums)
    return G, np.array(list(sums))


def ppo_grad_rows_mt(d, params, cfg, obs, actions, value_preds, returns, old_logp, adv, rows, inv_B, n_threads):
    """All-cores form of ppo_grad_rows (bench.py's cpu_baseline leg only)."""
    params, pp = _f(params)
    obs, po = _f(obs)
    actions, pa = _f(actions)
    value_preds, pv = _f(value_preds)
    returns, pr = _f(returns)
    old_logp, pl = _f(old_logp)
    adv, pad = _f(adv)
    rows, prow = _i64(rows)
    G = np.zeros(params.size, _R)
    sums = (C.c_double * 3)(0, 0, 0)
    lib().orc_ppo_grad_rows_mt(C.byref(d), pp, C.byref(cfg), po, pa, pv, pr, pl, pad, prow,
                               int(rows.size), _CR(inv_B), _fp(G), sums, int(n_threads))
    return G, np.array(list(sums))


def disc_grad_rows_mt(F, Hd, params, expert_rows, policy_rows, alpha, inv_B, n_threads, lambda_=10.0):
    params, pp = _f(params)
    expert_rows, pe = _f(expert_rows)
    policy_rows, ppol = _f(policy_rows)
    alpha, pal = _f(alpha)
    nb = expert_rows.shape[0]
    G = np.zeros(params.size, _R)
    sums = (C.c_double * 3)(0, 0, 0)
    lib().orc_disc_grad_rows_mt(F, Hd, pp, pe, ppol, pal, nb, _CR(inv_B), _CR(lambda_),
                                _fp(G), sums, int(n_threads))
    return G, np.array(list(sums))


def ppo_apply(params, G, adam, cfg):
    """In-place clip + Adam on params (float32 contiguous)."""
    lib().orc_ppo_apply(_fp(params), _fp(G), _fp(adam.m), _fp(adam.v), C.byref(adam.t),
                        C.c_int64(params.size), C.byref(cfg))


def ppo_update(d, params, adam, cfg, obs, actions, value_preds, returns, old_logp, perms):
    """params/adam updated in place. obs [T+1,N,O] actions [T,N,A] value_preds/returns [T+1,N]
    old_logp [T,N] perms [E,T*N] -> (value_loss, action_loss, entropy)."""
    obs, po = _f(obs)
    T, N = obs.shape[0] - 1, obs.shape[1]
    actions, pa = _f(actions)
    value_preds, pv = _f(value_preds)
    returns, pr = _f(returns)
    old_logp, pl = _f(old_logp)
    perms, pperm = _i64(perms)
    out = (_CR * 3)()
    lib().orc_ppo_update(C.byref(d), _fp(params), _fp(adam.m), _fp(adam.v), C.byref(adam.t),
                         C.byref(cfg), T, N, po, pa, pv, pr, pl, pperm, out)
    return tuple(float(x) for x in out)


def disc_grad_rows(F, Hd, params, expert_rows, policy_rows, alpha, inv_B, lambda_=10.0):
    params, pp = _f(params)
    expert_rows, pe = _f(expert_rows)
    policy_rows, ppol = _f(policy_rows)
    alpha, pal = _f(alpha)
    nb = expert_rows.shape[0]
    G = np.zeros(params.size, _R)
    sums = (C.c_double * 3)(0, 0, 0)
    lib().orc_disc_grad_rows(F, Hd, pp, pe, ppol, pal, nb, _CR(inv_B), _CR(lambda_),
                             _fp(G), sums)
    return G, np.array(list(sums))


def disc_grad_pen(F, Hd, params, expert_rows, policy_rows, alpha, lambda_=10.0):
    """Discriminator.compute_grad_pen_combined a2c/algo/gail.py:67-89, the value: lambda * mean((||dD/dx(mix)|| - 1)^2)
    (orc_disc_grad_rows' third sum; its gradient output is discarded)."""
    n = np.asarray(expert_rows).shape[0]
    _, sums = disc_grad_rows(F, Hd, params, expert_rows, policy_rows, alpha, 1.0 / n, lambda_)
    return _R(lambda_) * _R(sums[2] / n)


def adam_step(params, G, adam, lr, eps):
    lib().orc_adam_step(_fp(params), _fp(G), _fp(adam.m), _fp(adam.v), C.byref(adam.t),
                        C.c_int64(params.size), _CR(lr), _CR(eps))


def disc_update(F, Hd, params, adam, expert, obs_feat, B, expert_perm, policy_perm, alpha,
                lr=1e-3, eps=1e-8):
    """params/adam in place. expert [Ne,F]; obs_feat [T+1,N,F]. -> ((loss, expert, policy), n_d)."""
    expert, pe = _f(expert)
    obs_feat, pf = _f(obs_feat)
    T, N = obs_feat.shape[0] - 1, obs_feat.shape[1]
    expert_perm, pep = _i64(expert_perm)
    policy_perm, ppp = _i64(policy_perm)
    alpha, pal = _f(alpha)
    out = (_CR * 3)()
    n_d = lib().orc_disc_update(F, Hd, _fp(params), _fp(adam.m), _fp(adam.v), C.byref(adam.t),
                                _CR(lr), _CR(eps), pe, C.c_int64(expert.shape[0]), pf,
                                T, N, B, pep, ppp, pal, out)
    return tuple(float(x) for x in out), n_d


def disc_predict_reward(F, Hd, params, x, gamma, masks, offset, returns=None):
    """returns None <=> Discriminator.returns is None. -> (reward [n,1], returns [n,1])."""
    params, pp = _f(params)
    x, px = _f(x)
    n = x.shape[0]
    masks, pm = _f(masks)
    first = returns is None
    ret = np.zeros(n, _R) if first else np.array(returns, _R).reshape(n).copy()
    reward = np.empty(n, _R)
    lib().orc_disc_predict_reward(F, Hd, pp, px, n, _CR(gamma), pm, _CR(offset),
                                  _fp(ret), int(first), _fp(reward))
    return reward.reshape(n, 1), ret.reshape(n, 1)


def rms_update(state, x):
    st = (C.c_double * 3)(*state)
    x, px = _f(x)
    lib().orc_rms_update(st, px, int(x.size))
    return [st[0], st[1], st[2]]


def relabel(F, Hd, params, obs_feat, masks, gamma, offset, d_returns, rms_state):
    """-> (rewards [T,N], d_returns [N], rms_state[3]).  d_returns None <=> first call."""
    params, pp = _f(params)
    obs_feat, pf = _f(obs_feat)
    T, N = obs_feat.shape[0] - 1, obs_feat.shape[1]
    masks, pm = _f(masks)
    first = C.c_int(1 if d_returns is None else 0)
    ret = np.zeros(N, _R) if d_returns is None else np.array(d_returns, _R).reshape(N).copy()
    st = (C.c_double * 3)(*rms_state)
    rewards = np.empty((T, N), _R)
    lib().orc_relabel(F, Hd, pp, T, N, pf, pm, _CR(gamma), _CR(offset), _fp(ret),
                      C.byref(first), st, _fp(rewards))
    return rewards, ret, [st[0], st[1], st[2]]


def alive_bonus(masks, T, N, gail_tar_length):
    masks, pm = _f(masks)
    return float(lib().orc_alive_bonus(pm, T, N, C.c_double(gail_tar_length)))
