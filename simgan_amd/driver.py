"""The reference's two outer iterations (row A12 of SURVEY.md section 8) as reusable drivers over the drop-in
classes: `GailDynLearner` = a2c/main_gail_dyn_ppo.py:201-343 (learn the hybrid simulator) and `PpoLearner` =
a2c/main.py:199-290 (behaviour training / policy refinement: warm start, critic reset, linear LR decay, no D).

The reference mains cannot run unmodified in a current environment (np.infty, whole-module
torch.load, gym/pybullet), so this module reproduces their per-iteration call sequence:

    [rollout fill]  ->  get_value(obs[-1])  ->  gail_epoch x update_gail_dyn  ->  alive-bonus offset
    ->  reward relabel + return normalisation  ->  compute_returns  ->  PPO.update  ->  after_update

`GailDynLearner.update()` is the part the metric times (everything after the rollout fill);
`collect()` fills the rollout from any vectorised environment exposing
`step(action) -> (obs, reward, done, infos)` with `info["sas_feat"]` rows (the reference builds them
from info["sas_window"], a2c/main_gail_dyn_ppo.py:220-226).
"""
import collections.abc
import ctypes as C
import os

import numpy as np

from . import _lib
from .utils import RunningMeanStd, to_host_tensor, update_linear_schedule

_RESULT_SLOTS = 8


class PendingLosses(collections.abc.Mapping):
    """What `update()` returns on a device-resident rollout: the update's scalars (the reference main's log line,
    a2c/main_gail_dyn_ppo.py:322-338) as a read-only mapping that is filled from the library's results ring on first access.
    The update itself never makes the host wait, so the next update is queued while this one runs; a caller that logs every
    `log_interval` updates only pays for the ones it reads.  `resolve()` returns a plain dict."""

    def __init__(self, ctx, slot, keys):
        self._ctx, self._slot, self._keys, self._vals, self._error = ctx, slot, keys, None, None

    def resolve(self):
        """Raises SimganHipError when a launch of this update gave up waiting inside itself (sg_results_fetch); the slot is
        read once, a later access raises the same error again."""
        if self._error is not None:
            raise self._error
        if self._vals is None:
            out = (C.c_double * 13)()
            try:
                _lib.check(self._ctx.lib.sg_results_fetch(self._ctx.h, self._slot, out))
            except _lib.SimganHipError as exc:
                self._error = exc
                raise
            f32 = lambda x: float(np.float32(x))  # noqa: E731  (loss.item() values are float32 in the reference)
            vals = {"gail_loss": f32(out[0] / out[11]), "gail_loss_e": f32(out[1] / out[11]), "gail_loss_p": f32(out[2] / out[11]),
                    "value_loss": f32(out[8] / out[12]), "action_loss": f32(out[9] / out[12]), "dist_entropy": f32(out[10] / out[12]),
                    "r_sa": float(out[7]), "ret_rms": [float(out[3]), float(out[4]), float(out[5])], "dones": float(out[6])}
            self._vals = {k: vals[k] for k in self._keys}
        return self._vals

    def __getitem__(self, k):
        return self.resolve()[k]

    def __iter__(self):
        return iter(self._keys)

    def __len__(self):
        return len(self._keys)

    def __repr__(self):
        return f"PendingLosses({self._vals if self._vals is not None else 'not read yet'})"


class _ResultRing(object):
    """Slots of the library's results ring, handed out round-robin; a slot is read (so its mapping keeps its numbers)
    before it is published into again."""

    def __init__(self, ctx):
        self.ctx, self.n, self.live = ctx, 0, [None] * _RESULT_SLOTS

    def publish(self, disc, agent, keys):
        slot = self.n % _RESULT_SLOTS
        self.n += 1
        if self.live[slot] is not None and self.live[slot]._error is None:
            self.live[slot].resolve()       # (an update nobody read: its error, if any, surfaces here, eight updates later at most)
        _lib.check(self.ctx.lib.sg_results_publish(self.ctx.h, disc.h if disc is not None else None, agent.h, slot))
        self.live[slot] = PendingLosses(self.ctx, slot, keys)
        if os.environ.get("SG_UPDATE_SYNC") == "1":   # read the slot now: one host wait per update (profilers that serialise
            self.live[slot].resolve()                 # every dispatch cope badly with a host that runs eight updates ahead)
        return self.live[slot]


def alive_bonus_offset(num_of_dones, num_steps, num_processes, gail_tar_length, no_alive_bonus=False):
    """a2c/main_gail_dyn_ppo.py:258-271: r_sa = log d - log(1-d), d = 1 - dones/(dones + T*N/len_e),
    dones = sum(1-masks) + N/2."""
    if no_alive_bonus:
        return 0.0
    dones = num_of_dones + num_processes / 2
    num_of_expert_dones = (num_steps * num_processes) / gail_tar_length
    d_sa = 1 - dones / (dones + num_of_expert_dones)
    return float(np.log(d_sa) - np.log(1 - d_sa))


class ExpertLoader(object):
    """What the library needs from DataLoader(TensorDataset(expert), batch_size, shuffle=True,
    drop_last=len>batch) (a2c/main_gail_dyn_ppo.py:165-174): the matrix and the batch size."""

    def __init__(self, expert, batch_size):
        self.expert = _lib.as_f32(expert)
        self.batch_size = int(batch_size)


class _ResidentRms(RunningMeanStd):
    """`learner.ret_rms` while the state lives in the library (device-resident updates): a snapshot read from the device
    when the attribute is accessed (one stream synchronisation per access), whose every mutation -- `set_state`, `update`,
    `update_from_moments`, i.e. also a `relabel_rewards(..., learner.ret_rms)` call -- writes through to the device, so
    nothing done to it in place is lost on the next access or update."""

    def __init__(self, discr, state):
        RunningMeanStd.__init__(self, shape=())
        RunningMeanStd.set_state(self, state)
        self._discr = discr

    def set_state(self, st):
        RunningMeanStd.set_state(self, st)
        self._discr.set_rms(self.get_state())

    def update_from_moments(self, batch_mean, batch_var, batch_count):
        RunningMeanStd.update_from_moments(self, batch_mean, batch_var, batch_count)
        self._discr.set_rms(self.get_state())


class GailDynLearner(object):
    def __init__(self, actor_critic, agent, discr, rollouts, expert, gail_batch_size=128, gail_epoch=5,
                 gamma=0.99, gae_lambda=0.95, use_gae=True, use_proper_time_limits=True,
                 gail_tar_length=500.0, no_alive_bonus=False, use_linear_lr_decay=False, lr=None,
                 num_updates=None):
        self.actor_critic, self.agent, self.discr, self.rollouts = actor_critic, agent, discr, rollouts
        self.loader = expert if hasattr(expert, "batch_size") else ExpertLoader(expert, gail_batch_size)
        self.gail_epoch, self.gamma, self.gae_lambda = gail_epoch, gamma, gae_lambda
        self.use_gae, self.use_proper_time_limits = use_gae, use_proper_time_limits
        self.gail_tar_length, self.no_alive_bonus = gail_tar_length, no_alive_bonus
        self.use_linear_lr_decay, self.lr, self.num_updates = use_linear_lr_decay, lr, num_updates
        self._ret_rms = RunningMeanStd(shape=())       # a2c/main_gail_dyn_ppo.py:198-199
        self._rms_on_device = False                    # the state lives in the library while updates run device-resident
        self.j = 0
        self.world = getattr(rollouts.ctx, "world", 1)
        self._ring = _ResultRing(rollouts.ctx)

    @property
    def ret_rms(self):
        if self._rms_on_device:
            st = self.discr.scalars()[:3]
            self._ret_rms.set_state(st)
            return _ResidentRms(self.discr, st)     # mutations write through to the device
        return self._ret_rms

    @ret_rms.setter
    def ret_rms(self, value):
        self._ret_rms = value
        if self._rms_on_device:
            self.discr.set_rms(value.get_state())

    # ---------------------------------------------------------------- rollout fill (:209-236)
    def collect(self, envs, sas_feat_of_infos):
        ro, pol = self.rollouts, self.actor_critic
        for step in range(ro.num_steps):
            value, action, logp, hxs = pol.act(ro.obs[step], ro.recurrent_hidden_states[step], ro.masks[step])
            obs, reward, done, infos = envs.step(action)
            masks = np.array([[0.0] if d else [1.0] for d in done], np.float32)
            bad = np.array([[0.0] if 'bad_transition' in info.keys() else [1.0] for info in infos], np.float32)
            ro.insert(obs, hxs, action, logp, value, reward, to_host_tensor(masks), to_host_tensor(bad),
                      to_host_tensor(np.asarray(sas_feat_of_infos(infos), np.float32)))
        if ro.device_resident:
            ro.sync_to_device()

    # ------------------------------------------------------------- the timed part (:239-304)
    def update(self):
        ro, lib = self.rollouts, self.rollouts.lib
        if self.use_linear_lr_decay:            # :203-207
            update_linear_schedule(self.agent.optimizer, self.j, self.num_updates, self.lr)
        if ro.device_resident:
            return self._update_resident()
        if self._rms_on_device:                 # a drop-in update after device-resident ones: bring the state back
            self._ret_rms.set_state(self.discr.scalars()[:3])
            self._rms_on_device = False
        gail = None
        for e in range(self.gail_epoch):        # :255-256 -- the reference keeps the last epoch's losses only
            last = e == self.gail_epoch - 1
            gail = self.discr.update_gail_dyn(self.loader, ro, **({} if last else {"fetch_losses": False}))
        # drop-in mode: the host tensors are the rollout; every call below uploads what it reads and returns its result
        ro._push([_lib.F_MASKS])
        dones = C.c_double(0)
        _lib.check(lib.sg_rollout_count_dones(ro.h, C.byref(dones)))   # all ranks when world > 1
        r_sa = alive_bonus_offset(dones.value, ro.num_steps, ro.num_processes * self.world, self.gail_tar_length,
                                  self.no_alive_bonus)
        self.discr.relabel_rewards(ro, self.gamma, -r_sa, self._ret_rms)    # :275-297 fused on device
        next_value = self.actor_critic.get_value(ro.obs[-1], ro.recurrent_hidden_states[-1], ro.masks[-1])
        ro.compute_returns(next_value, self.use_gae, self.gamma, self.gae_lambda, self.use_proper_time_limits)
        ppo = self.agent.update(ro)             # :302
        ro.after_update()                       # :304
        self.j += 1
        return {"gail_loss": gail[0], "gail_loss_e": gail[1], "gail_loss_p": gail[2], "value_loss": ppo[0],
                "action_loss": ppo[1], "dist_entropy": ppo[2], "r_sa": r_sa}


def _gail_update_resident(self):
    """GailDynLearner.update() on a device-resident rollout: the same call sequence, every call only QUEUES work on the
    library's stream -- discriminator epochs without their loss read-back, the alive-bonus offset from the device's done
    count, ret_rms resident in the library, GAE with get_value(obs[T]) on the device, PPO, after_update -- and the scalars
    are published into the results ring at the end.  The host never waits inside an update."""
    ro, lib = self.rollouts, self.rollouts.lib
    if not self._rms_on_device:
        self.discr.set_rms(self._ret_rms.get_state())
        self._rms_on_device = True
    for _ in range(self.gail_epoch):            # :255-256
        self.discr.update_gail_dyn(self.loader, ro, fetch_losses=False)
    self.discr.relabel_rewards_auto(ro, self.gamma, self.gail_tar_length, self.no_alive_bonus)     # :258-297
    _lib.check(lib.sg_rollout_compute_returns_policy(ro.h, self.actor_critic.h, 1 if self.use_gae else 0, float(self.gamma),
                                                     float(self.gae_lambda), 1 if self.use_proper_time_limits else 0))
    self.agent.update(ro, fetch_losses=False)   # :302
    ro.after_update()                           # :304
    self.j += 1
    return self._ring.publish(self.discr, self.agent, ("gail_loss", "gail_loss_e", "gail_loss_p", "value_loss", "action_loss",
                                                       "dist_entropy", "r_sa"))


GailDynLearner._update_resident = _gail_update_resident


class PpoLearner(object):
    """a2c/main.py:199-257, the plain-PPO caller (BASELINE.json configs[0] and configs[4]):

        [lr decay :201-205] -> [rollout fill :207-244] -> get_value(obs[-1]) :246-249 -> compute_returns :251-252
        -> PPO.update :254 -> after_update :256

    Rewards are the environment's own (return-scaled by VecNormalize, simgan_amd/envs.py); the rollout's feature
    slot carries the observation itself (`replace_obs_with_feat` with no selector is an identity copy,
    a2c/main.py:168-169,218, my_pybullet_envs/utils.py:310-331)."""

    def __init__(self, actor_critic, agent, rollouts, gamma=0.99, gae_lambda=0.95, use_gae=True,
                 use_proper_time_limits=True, use_linear_lr_decay=False, lr=None, num_updates=None):
        self.actor_critic, self.agent, self.rollouts = actor_critic, agent, rollouts
        self.gamma, self.gae_lambda, self.use_gae = gamma, gae_lambda, use_gae
        self.use_proper_time_limits = use_proper_time_limits
        self.use_linear_lr_decay, self.lr, self.num_updates = use_linear_lr_decay, lr, num_updates
        if use_linear_lr_decay:
            assert lr is not None and num_updates, "linear LR decay needs the initial lr and num_updates (a2c/main.py:196-205)"
        self.j = 0
        self._ring = _ResultRing(rollouts.ctx)

    @staticmethod
    def warm_start(path, obs_shape, action_space, warm_start_logstd=None, ctx=None, critic_seed=1):
        """a2c/main.py:78-88: load the behaviour policy from a reference checkpoint (its ob_rms is assumed None, :79),
        re-initialise the critic, optionally reset the action log-std."""
        from .checkpoint import load_policy
        actor_critic, _ = load_policy(path, ctx=ctx)
        actor_critic.reset_critic(obs_shape, seed=critic_seed)
        if warm_start_logstd is not None:
            actor_critic.reset_variance(action_space, warm_start_logstd)
        return actor_critic

    # ---------------------------------------------------------------- rollout fill (:207-244)
    def collect(self, envs, feat_select_func=None):
        ro, pol = self.rollouts, self.actor_critic
        for step in range(ro.num_steps):
            value, action, logp, hxs = pol.act(ro.obs[step], ro.recurrent_hidden_states[step], ro.masks[step])
            obs, reward, done, infos = envs.step(action)
            feat = obs if feat_select_func is None else feat_select_func(obs)
            masks = np.array([[0.0] if d else [1.0] for d in done], np.float32)
            bad = np.array([[0.0] if 'bad_transition' in info.keys() else [1.0] for info in infos], np.float32)
            ro.insert(obs, hxs, action, logp, value, reward, to_host_tensor(masks), to_host_tensor(bad), feat)
        if ro.device_resident:
            ro.sync_to_device()

    # ------------------------------------------------------------- the timed part (:201-205, :246-256)
    def update(self):
        ro, lib = self.rollouts, self.rollouts.lib
        if self.use_linear_lr_decay:
            update_linear_schedule(self.agent.optimizer, self.j, self.num_updates, self.lr)
        if ro.device_resident:   # nothing below waits for the device: the losses are read from the results ring on demand
            _lib.check(lib.sg_rollout_compute_returns_policy(ro.h, self.actor_critic.h, 1 if self.use_gae else 0,
                                                             float(self.gamma), float(self.gae_lambda),
                                                             1 if self.use_proper_time_limits else 0))
            self.agent.update(ro, fetch_losses=False)
            ro.after_update()
            self.j += 1
            return self._ring.publish(None, self.agent, ("value_loss", "action_loss", "dist_entropy"))
        next_value = self.actor_critic.get_value(ro.obs[-1], ro.recurrent_hidden_states[-1], ro.masks[-1])
        ro.compute_returns(next_value, self.use_gae, self.gamma, self.gae_lambda, self.use_proper_time_limits)
        value_loss, action_loss, dist_entropy = self.agent.update(ro)
        ro.after_update()
        self.j += 1
        return {"value_loss": value_loss, "action_loss": action_loss, "dist_entropy": dist_entropy}
