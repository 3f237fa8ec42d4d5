"""Discriminator -- host mirror of a2c/algo/gail.py:34-217 (GAIL-dyn variant)."""
import ctypes as C
import os
import weakref

import numpy as np

from .. import _lib
from ..utils import RunningMeanStd, derive_seed, to_host_tensor


_PREFETCH = os.environ.get("SG_RELABEL_PREFETCH", "1") != "0"


class _Trunk(object):
    """`discr.trunk.state_dict()` view (trunk.0 / trunk.2 / trunk.4 Linear layers)."""

    def __init__(self, d):
        self._d = d

    def state_dict(self):
        d = self._d
        flat, out, off = d.get_flat_params(), {}, 0
        for name, shape in d.param_shapes():
            n = int(np.prod(shape))
            out[name] = to_host_tensor(flat[off:off + n].reshape(shape).copy())
            off += n
        return out

    def load_state_dict(self, sd):
        self._d.set_flat_params(np.concatenate([_lib.as_f32(sd[n]).reshape(-1) for n, _ in self._d.param_shapes()]))


class Discriminator(object):
    def __init__(self, input_dim, hidden_dim, device=None, ctx=None, seed=0):
        self.device = device
        self.input_dim, self.hidden_dim = int(input_dim), int(hidden_dim)
        self.ctx = ctx or _lib.Context.default()
        self.lib = self.ctx.lib
        h = _lib.H()
        _lib.check(self.lib.sg_disc_create(self.ctx.h, self.input_dim, self.hidden_dim, C.byref(h)))
        self.h = h
        n = C.c_int64(0)
        _lib.check(self.lib.sg_disc_num_params(self.h, C.byref(n)))
        self.num_params = n.value
        self.trunk = _Trunk(self)
        self.ret_rms = RunningMeanStd(shape=())  # unused field kept from a2c/algo/gail.py:51
        self._expert_id = None
        self._calls = 0
        self.seed = derive_seed(seed, 0xD15C)   # DataLoader shuffle / feed_forward_generator / mixup-alpha streams
        self._ret_n = None
        self._steps = None        # predict_reward_combined: the relabel loop's T steps computed by one launch, being served call by call
        self._w_version = 0       # bumped by everything that changes the weights
        self._init_params(np.random.default_rng(seed))

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.sg_disc_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def param_shapes(self):
        F, Hd = self.input_dim, self.hidden_dim
        return [("0.weight", (Hd, F)), ("0.bias", (Hd,)), ("2.weight", (Hd, Hd)), ("2.bias", (Hd,)),
                ("4.weight", (1, Hd)), ("4.bias", (1,))]

    def _init_params(self, rng):
        """nn.Linear default init (kaiming_uniform a=sqrt5 => U(-1/sqrt(fan_in), 1/sqrt(fan_in))
        for weight and bias), a2c/algo/gail.py:40-43."""
        parts = []
        for name, shape in self.param_shapes():
            fan_in = shape[1] if len(shape) == 2 else {"0.bias": self.input_dim}.get(name, self.hidden_dim)
            b = 1.0 / np.sqrt(fan_in)
            parts.append(rng.uniform(-b, b, size=int(np.prod(shape))).astype(np.float32))
        self.set_flat_params(np.concatenate(parts))

    def get_flat_params(self):
        out = np.empty(self.num_params, np.float32)
        _lib.check(self.lib.sg_disc_get_params(self.h, _lib.fptr(out), out.size))
        return out

    def set_flat_params(self, flat):
        self._weights_changing()
        flat = _lib.as_f32(flat).reshape(-1)
        _lib.check(self.lib.sg_disc_set_params(self.h, _lib.fptr(flat), flat.size))

    def get_adam(self):
        n = self.num_params
        m, v = np.empty(n, np.float32), np.empty(n, np.float32)
        step = C.c_int64(0)
        _lib.check(self.lib.sg_disc_get_adam(self.h, _lib.fptr(m), _lib.fptr(v), n, C.byref(step)))
        return m, v, step.value

    def set_adam(self, m, v, step):
        """Restore the optimizer state (torch.optim.Adam exp_avg / exp_avg_sq / step, a2c/algo/gail.py:48)."""
        m, v = _lib.as_f32(m).reshape(-1), _lib.as_f32(v).reshape(-1)
        _lib.check(self.lib.sg_disc_set_adam(self.h, _lib.fptr(m), _lib.fptr(v), m.size, int(step)))

    def _weights_changing(self):
        self._steps_retire()
        self._w_version += 1

    def train(self, mode=True):
        return self

    def eval(self):
        return self

    def to(self, device):
        return self

    # ------------------------------------------------------------------ expert data
    def set_expert(self, expert):
        """Make the expert matrix [N_e, F] (a2c/main_gail_dyn_ppo.py:163-165) resident in HBM."""
        e = _lib.as_f32(expert).reshape(-1, self.input_dim)
        _lib.check(self.lib.sg_disc_set_expert(self.h, _lib.fptr(e), e.shape[0]))
        self.n_expert = e.shape[0]

    def _bind_loader(self, expert_loader):
        """The reference passes a torch DataLoader over TensorDataset(expert); only its
        .batch_size and the underlying matrix are needed (a2c/algo/gail.py:157-165)."""
        ds = getattr(expert_loader, "dataset", None)
        key = id(ds) if ds is not None else id(expert_loader)
        if key != self._expert_id:
            if ds is not None and hasattr(ds, "tensors"):
                mat = ds.tensors[0]
            elif ds is not None:
                mat = np.stack([_lib.as_f32(ds[i][0]) for i in range(len(ds))])
            else:
                mat = expert_loader.expert
            self.set_expert(mat)
            self._expert_id = key
        return int(expert_loader.batch_size)

    @staticmethod
    def _draw_args(ep, pp, al):
        """(pointer, count) triples of the injected draws for the C ABI, which checks lengths and index ranges itself
        (include/simgan_hip.h: sg_disc_update_gail_dyn)."""
        return (None if ep is None else _lib.i64ptr(ep), 0 if ep is None else ep.size,
                None if pp is None else _lib.i64ptr(pp), 0 if pp is None else pp.size,
                None if al is None else _lib.fptr(al), 0 if al is None else al.size)

    def last_draws(self):
        """(expert_perm, policy_perm, alpha) the last update epoch consumed (injected or library-drawn)."""
        cnt = (C.c_int64 * 3)()
        _lib.check(self.lib.sg_disc_last_draws(self.h, None, None, None, cnt))
        ep, pp = np.empty(cnt[0], np.int64), np.empty(cnt[1], np.int64)
        al = np.empty(cnt[2], np.float32)
        _lib.check(self.lib.sg_disc_last_draws(self.h, _lib.i64ptr(ep), _lib.i64ptr(pp), _lib.fptr(al), None))
        return ep, pp, al

    # ---------------------------------------------------------------------- updates
    def update_gail_dyn(self, expert_loader, rollouts, expert_perm=None, policy_perm=None, alpha=None, fetch_losses=True):
        """a2c/algo/gail.py:154-193, one epoch -> (loss, expert_loss, policy_loss) means.
        expert_perm / policy_perm / alpha inject the reference's RNG artefacts (parity tests).
        fetch_losses=False: return None as soon as the epoch is queued on the device (a driver that keeps only the last
        epoch's losses, like a2c/main_gail_dyn_ppo.py:255-256, need not wait for the earlier ones)."""
        self._weights_changing()
        B = self._bind_loader(expert_loader)
        rollouts._push([_lib.F_OBS_FEAT])
        out = (C.c_float * 3)()
        nst = C.c_int(0)
        ep = None if expert_perm is None else _lib.as_i64(expert_perm).reshape(-1)
        pp = None if policy_perm is None else _lib.as_i64(policy_perm).reshape(-1)
        al = None if alpha is None else _lib.as_f32(alpha).reshape(-1)
        # world > 1: injected draws are the reference's at num_processes = world * N -- policy_perm ranges over the
        # union of every rank's rows in the reference's numbering, in both data-parallel modes
        self._calls += 1
        _lib.check(self.lib.sg_disc_update_gail_dyn(
            self.h, rollouts.h, B, *self._draw_args(ep, pp, al), (self.seed + self._calls) & (2 ** 64 - 1),
            out if fetch_losses else None, C.byref(nst)))
        self.last_n_steps = nst.value
        return (float(out[0]), float(out[1]), float(out[2])) if fetch_losses else None

    def update(self, expert_loader, rollouts, obsfilt=None, is_gail_dyn=False, a_dim=None,
               expert_perm=None, policy_perm=None, alpha=None):
        """a2c/algo/gail.py:91-152 (state/action GAIL and its is_gail_dyn feature assembly), one epoch.
        Not called by any shipped driver (a2c/main_gail_dyn_ppo.py:256 uses update_gail_dyn); same device
        step on rows assembled here: policy rows = (state | action), or for is_gail_dyn
        (obs_feat | obs[:, -a_dim:] | next_obs_feat) (a2c/algo/gail.py:102-109)."""
        self._weights_changing()
        ds = expert_loader.dataset
        # the reference re-applies obsfilt (VecNormalize._obfilt with the CURRENT ob_rms) to every expert batch of every
        # call (a2c/algo/gail.py:118-120), so a filtered expert matrix is rebuilt per call; only the unfiltered one is cached
        key = ("update", id(ds)) if obsfilt is None else None
        if key is None or key != self._expert_id:
            es, ea = _lib.as_f32(ds.tensors[0]), _lib.as_f32(ds.tensors[1])
            if obsfilt is not None:
                es = np.asarray(obsfilt(es, update=False), np.float32)
            self.set_expert(np.concatenate([es, ea], axis=1))
            self._expert_id = key
        B = int(expert_loader.batch_size)
        npv = lambda t: (t.numpy() if hasattr(t, "numpy") else np.asarray(t))  # noqa: E731
        obs, acts, feat = npv(rollouts.obs), npv(rollouts.actions), npv(rollouts.obs_feat)
        flat = lambda a: a.reshape(-1, a.shape[-1])  # noqa: E731
        if not is_gail_dyn:
            rows = np.concatenate([flat(obs[:-1]), flat(acts)], axis=1)
        else:
            rows = np.concatenate([flat(feat[:-1]), flat(obs[:-1])[:, -a_dim:], flat(feat[1:])], axis=1)
        rows = np.ascontiguousarray(rows, np.float32)
        assert rows.shape[1] == self.input_dim, (rows.shape, self.input_dim)
        out = (C.c_float * 3)()
        nst = C.c_int(0)
        ep = None if expert_perm is None else _lib.as_i64(expert_perm).reshape(-1)
        pp = None if policy_perm is None else _lib.as_i64(policy_perm).reshape(-1)
        al = None if alpha is None else _lib.as_f32(alpha).reshape(-1)
        self._calls += 1
        _lib.check(self.lib.sg_disc_update_rows(
            self.h, _lib.fptr(rows), rows.shape[0], int(rollouts.num_processes), B, *self._draw_args(ep, pp, al),
            (self.seed + self._calls) & (2 ** 64 - 1), out, C.byref(nst)))
        self.last_n_steps = nst.value
        return float(out[0]), float(out[1]), float(out[2])

    # ---------------------------------------------------------------------- rewards
    def predict_reward_combined(self, d_in, gamma, masks, offset=0.0):
        """a2c/algo/gail.py:201-210 -> (reward [n,1], returns [n,1]); self.returns persists.

        The unchanged main calls this T times per update with `rollouts.obs_feat[step + 1]`, `rollouts.masks[step]`
        (a2c/main_gail_dyn_ppo.py:275-280).  When the arguments ARE those slices of a drop-in rollout, the call for step 0 runs
        all T steps in one launch (sg_disc_predict_reward_steps: same kernels, same rows, same recurrence) and the calls for
        steps 1 .. T-1 are served from its result -- bit-identical to T separate calls, one upload and one read-back instead of
        T of each.  Anything that is not that pattern (other rows, another order, a changed mask / obs_feat / weight / gamma /
        offset / Discriminator.returns in between) takes the per-call path from the state the served calls imply.
        SG_RELABEL_PREFETCH=0 turns the recognition off."""
        hit = self._steps_lookup(d_in, gamma, masks, offset)
        if hit is not None:
            return hit
        self._steps_retire()
        x = _lib.as_f32(d_in).reshape(-1, self.input_dim)
        n = x.shape[0]
        m = _lib.as_f32(masks).reshape(-1)
        assert m.size == n, f"masks: {m.size} entries for {n} rows"
        if self._ret_n and self._ret_n != n and self.returns is not None:   # the reference's broadcast would fail here
            raise ValueError(f"predict_reward_combined: Discriminator.returns holds {self._ret_n} rows, got {n}")
        reward = np.empty((n, 1), np.float32)
        returns = np.empty((n, 1), np.float32)
        _lib.check(self.lib.sg_disc_predict_reward(self.h, _lib.fptr(x), n, float(gamma), _lib.fptr(m),
                                                   float(offset), _lib.fptr(reward), _lib.fptr(returns)))
        self._ret_n = n
        return to_host_tensor(reward), to_host_tensor(returns)

    # -- the relabel loop's T calls from one launch ------------------------------------------------------------------------
    def _steps_slice(self, d_in, masks):
        """(rollout, step) when (d_in, masks) are exactly rollouts.obs_feat[step + 1] / rollouts.masks[step] of a drop-in rollout."""
        from ..storage import rollout_of_feat_slice
        ro = rollout_of_feat_slice(d_in)
        if ro is None or ro.device_resident or ro.feat_len != self.input_dim or getattr(masks, "_base", None) is not ro.masks:
            return None
        N, F = ro.num_processes, ro.feat_len
        off = d_in.storage_offset()
        if tuple(d_in.shape) != (N, F) or not d_in.is_contiguous() or off % (N * F) or tuple(masks.shape) != (N, 1) or not masks.is_contiguous():
            return None
        step = off // (N * F) - 1
        if step < 0 or step >= ro.num_steps or masks.storage_offset() != step * N:
            return None
        return ro, step

    def _steps_lookup(self, d_in, gamma, masks, offset):
        if not _PREFETCH or not hasattr(d_in, "storage_offset") or not hasattr(masks, "storage_offset"):
            return None
        hit = self._steps_slice(d_in, masks)
        if hit is None:
            return None
        ro, step = hit
        c = self._steps
        key = (float(gamma), float(offset), self._w_version, ro.obs_feat._version, ro.masks._version)
        if step == 0:
            self._steps_retire()
            if self._ret_n and self._ret_n != ro.num_processes and self.returns is not None:
                return None                                   # (the per-call path raises the reference's size error)
            T, N = ro.num_steps, ro.num_processes
            ro._push([_lib.F_OBS_FEAT, _lib.F_MASKS])
            reward, returns = np.empty((T, N, 1), np.float32), np.empty((T, N, 1), np.float32)
            _lib.check(self.lib.sg_disc_predict_reward_steps(self.h, ro.h, float(gamma), float(offset), _lib.fptr(reward), _lib.fptr(returns)))
            self._ret_n = N
            c = self._steps = {"ro": weakref.ref(ro), "key": key, "reward": reward, "returns": returns, "pos": -1, "T": T}
        elif c is None or c["ro"]() is not ro or c["key"] != key or step != c["pos"] + 1:
            return None
        c["pos"] = step
        # (no copies: each row is handed out once; `returns` aliases the member in the reference too -- a2c/algo/gail.py:210 returns self.returns)
        out = to_host_tensor(c["reward"][step]), to_host_tensor(c["returns"][step])
        if step == c["T"] - 1:
            self._steps = None                                # complete: the device's returns ARE the state after the T-th call
        return out

    def _steps_retire(self):
        """Leave the served-from-cache mode: Discriminator.returns on the device becomes what the calls served so far imply."""
        c, self._steps = self._steps, None
        if c is not None and 0 <= c["pos"] < c["T"] - 1:
            v = np.ascontiguousarray(c["returns"][c["pos"]].reshape(-1))
            _lib.check(self.lib.sg_disc_set_returns(self.h, _lib.fptr(v), v.size))

    def predict_reward(self, state, action, gamma, masks, offset=0.0):
        """a2c/algo/gail.py:195-199"""
        d_in = np.concatenate([_lib.as_f32(state), _lib.as_f32(action)], axis=1)
        return self.predict_reward_combined(d_in, gamma, masks, offset)

    def predict_prob_single_step(self, state, action, s_n=None):
        """a2c/algo/gail.py:212-217: sigmoid(D(cat(state, action))).  The reference's two-argument form takes row batches
        [n, .] and returns the probabilities as an [n, 1] tensor; with a third block `s_n` (a GAIL-dyn transition s, a, s')
        and ONE transition it returns a Python float, as the environments' discriminator-in-the-loop code consumes it."""
        if s_n is None:
            st, ac = _lib.as_f32(state), _lib.as_f32(action)
            if st.ndim == 1:
                st, ac = st.reshape(1, -1), ac.reshape(1, -1)
            x = np.ascontiguousarray(np.concatenate([st, ac], axis=1))
            assert x.shape[1] == self.input_dim, (x.shape, self.input_dim)
            return self.predict_prob(x)
        x = np.concatenate([_lib.as_f32(state).reshape(-1), _lib.as_f32(action).reshape(-1), _lib.as_f32(s_n).reshape(-1)])
        assert x.size == self.input_dim, (x.size, self.input_dim)
        x = np.ascontiguousarray(x.reshape(1, -1))
        out = np.empty(1, np.float32)
        _lib.check(self.lib.sg_disc_predict_prob(self.h, _lib.fptr(x), 1, _lib.fptr(out)))
        return float(out[0])

    def compute_grad_pen(self, expert_state, expert_action, policy_state, policy_action, lambda_=10., alpha=None):
        """a2c/algo/gail.py:53-65: the penalty on (state | action) rows."""
        cat = lambda x, y: np.concatenate([_lib.as_f32(x).reshape(len(x), -1), _lib.as_f32(y).reshape(len(y), -1)], axis=1)  # noqa: E731
        return self.compute_grad_pen_combined(cat(expert_state, expert_action), cat(policy_state, policy_action), lambda_, alpha=alpha)

    def compute_grad_pen_combined(self, expert_combined, policy_combined, lambda_=10., alpha=None):
        """a2c/algo/gail.py:67-89 -> lambda_ * mean((||dD/dx(alpha e + (1 - alpha) p)||_2 - 1)^2) as a 0-dim host tensor: the
        VALUE of the term.  In the reference it is a differentiable tensor that `update` / `update_gail_dyn` add to their loss
        before `.backward()` (:133, :179, its only callers); here the penalty and its double backward are formed inside the
        update step's kernels, so the returned value carries no graph.  `alpha` injects the reference's torch.rand(n, 1)
        draw (:72); default: the library's generator."""
        e = np.ascontiguousarray(_lib.as_f32(expert_combined).reshape(-1, self.input_dim))
        p = np.ascontiguousarray(_lib.as_f32(policy_combined).reshape(-1, self.input_dim))
        assert e.shape == p.shape, (e.shape, p.shape)        # alpha.expand_as / the elementwise mix would raise in the reference
        al = None if alpha is None else np.ascontiguousarray(_lib.as_f32(alpha).reshape(-1))
        assert al is None or al.size == e.shape[0], "one alpha per row pair"
        pen = np.empty(e.shape[0], np.float32)
        self._calls += 1
        _lib.check(self.lib.sg_disc_grad_pen(self.h, _lib.fptr(e), _lib.fptr(p), None if al is None else _lib.fptr(al), e.shape[0],
                                             (self.seed + self._calls) & (2 ** 64 - 1), _lib.fptr(pen)))
        return to_host_tensor(np.asarray(np.float32(lambda_) * pen.mean(dtype=np.float32), np.float32))

    def predict_prob(self, d_in):
        """Batched form of predict_prob_single_step: sigmoid(D(x)) for rows x [n, F] -> [n, 1]."""
        x = _lib.as_f32(d_in).reshape(-1, self.input_dim)
        out = np.empty((x.shape[0], 1), np.float32)
        _lib.check(self.lib.sg_disc_predict_prob(self.h, _lib.fptr(x), x.shape[0], _lib.fptr(out)))
        return to_host_tensor(out)

    @property
    def returns(self):
        c = self._steps
        if c is not None and 0 <= c["pos"] < c["T"] - 1:      # mid-loop: the state after the calls served so far
            return to_host_tensor(c["returns"][c["pos"]].copy())
        none = C.c_int(0)
        _lib.check(self.lib.sg_disc_get_returns(self.h, None, 0, C.byref(none)))
        if none.value:
            return None
        n = self._ret_n
        if not n:
            raise _lib.SimganHipError("Discriminator.returns: the handle holds returns this object never sized "
                                      "(set through another object?); assign .returns or call predict_reward* first")
        out = np.empty((n, 1), np.float32)
        _lib.check(self.lib.sg_disc_get_returns(self.h, _lib.fptr(out), n, C.byref(none)))
        return to_host_tensor(out)

    @returns.setter
    def returns(self, value):
        self._steps = None        # whatever was being served is void: the caller sets the state
        if value is None:
            _lib.check(self.lib.sg_disc_reset_returns(self.h))
            self._ret_n = None
        else:
            v = _lib.as_f32(value).reshape(-1)
            _lib.check(self.lib.sg_disc_set_returns(self.h, _lib.fptr(v), v.size))
            self._ret_n = v.size

    def relabel_rewards_auto(self, rollouts, gamma, gail_tar_length, no_alive_bonus=False):
        """a2c/main_gail_dyn_ppo.py:258-297 entirely on the device, nothing read back: the alive-bonus offset from the
        device's own done count, the fused relabel, ret_rms kept inside the library (`set_rms` / `scalars`).  Device-resident
        rollouts only (the rewards stay in HBM)."""
        assert rollouts.device_resident, "relabel_rewards_auto leaves the rewards on the device"
        self._steps_retire()
        _lib.check(self.lib.sg_disc_relabel_rewards_auto(self.h, rollouts.h, float(gamma), float(gail_tar_length),
                                                         1 if no_alive_bonus else 0))
        self._ret_n = rollouts.num_processes

    def set_rms(self, state):
        st = (C.c_double * 3)(*[float(x) for x in state])
        _lib.check(self.lib.sg_disc_set_rms(self.h, st))

    def scalars(self):
        """{mean, var, count} of the device-resident ret_rms, sum(1 - masks) and r_sa of the last relabel_rewards_auto."""
        out = (C.c_double * 5)()
        _lib.check(self.lib.sg_disc_get_scalars(self.h, out))
        return list(out)

    def relabel_rewards(self, rollouts, gamma, offset, ret_rms):
        """Fused a2c/main_gail_dyn_ppo.py:275-292 over all T steps on device; ret_rms is the caller's
        RunningMeanStd (float64 state updated in place); rollouts.rewards is rewritten."""
        self._steps_retire()
        rollouts._push([_lib.F_OBS_FEAT, _lib.F_MASKS])
        st = (C.c_double * 3)(*ret_rms.get_state())
        _lib.check(self.lib.sg_disc_relabel_rewards(self.h, rollouts.h, float(gamma), float(offset), st))
        ret_rms.set_state([st[0], st[1], st[2]])
        self._ret_n = rollouts.num_processes
        rollouts._pull([_lib.F_REWARDS])
