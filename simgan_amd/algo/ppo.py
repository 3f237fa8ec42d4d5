"""PPO -- host mirror of a2c/algo/ppo.py:29-157.  update() runs entirely on the GPU
(advantage normalisation, E_p x M clipped-surrogate steps with global-norm clipping and Adam)."""
import ctypes as C

import numpy as np

from .. import _lib
from ..utils import derive_seed


class _ParamGroups(list):
    """`optimizer.param_groups[i]['lr'] = x` (a2c/utils.py:68-72) must reach the device."""


class _Group(dict):
    def __init__(self, owner, **kw):
        super().__init__(**kw)
        self._owner = owner

    def __setitem__(self, k, v):
        super().__setitem__(k, v)
        if k == 'lr':
            self._owner._set_lr(float(v))


class _Optimizer(object):
    def __init__(self, agent, lr, eps):
        self._agent = agent
        self.param_groups = _ParamGroups([_Group(self, lr=lr, eps=eps, betas=(0.9, 0.999))])

    def _set_lr(self, lr):
        _lib.check(self._agent.lib.sg_ppo_set_lr(self._agent.h, lr))

    def state(self):
        return self._agent.get_adam()


class PPO():
    def __init__(self,
                 actor_critic,
                 clip_param,
                 ppo_epoch,
                 num_mini_batch,
                 value_loss_coef,
                 entropy_coef,
                 symmetry_coef=0,
                 lr=None,
                 eps=None,
                 max_grad_norm=None,
                 use_clipped_value_loss=True,
                 mirror_obs=None,
                 mirror_act=None,
                 seed=0):
        if mirror_obs and symmetry_coef > 0:
            # a2c/algo/ppo.py:110-136: never enabled by any shipped script
            raise NotImplementedError("mirror-symmetry loss is out of scope")
        self.actor_critic = actor_critic
        self.clip_param = clip_param
        self.ppo_epoch = ppo_epoch
        self.num_mini_batch = num_mini_batch
        self.value_loss_coef = value_loss_coef
        self.entropy_coef = entropy_coef
        self.max_grad_norm = max_grad_norm
        self.use_clipped_value_loss = use_clipped_value_loss
        self.symmetry_coef = symmetry_coef
        self.mirror_obs = mirror_obs
        self.mirror_act = mirror_act
        self.is_cuda = True   # the rollout the update reads lives in HBM whatever the host tensors are

        self.ctx = actor_critic.ctx
        self.lib = self.ctx.lib
        cfg = _lib.PPOConfig(float(clip_param), int(ppo_epoch), int(num_mini_batch), float(value_loss_coef),
                             float(entropy_coef), float(lr), float(eps), float(max_grad_norm),
                             1 if use_clipped_value_loss else 0)
        h = _lib.H()
        _lib.check(self.lib.sg_ppo_create(self.ctx.h, actor_critic.h, C.byref(cfg), C.byref(h)))
        self.h = h
        if hasattr(actor_critic, "_register_handle_user"):
            actor_critic._register_handle_user(self)
        self.optimizer = _Optimizer(self, lr, eps)
        self._calls = 0
        self.seed = derive_seed(seed, 0xBADC0FFEE)   # minibatch-permutation stream (a2c/storage.py:159-162)

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.sg_ppo_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def update(self, rollouts, perms=None, fetch_losses=True):
        """a2c/algo/ppo.py:65-157 -> (value_loss_epoch, action_loss_epoch, dist_entropy_epoch).
        `perms` ([ppo_epoch, T*N] int64) injects the samplers' permutations (parity tests);
        default = the library's counter-based generator.  With a communicator of world > 1 the injected
        permutations are the reference's at num_processes = world * N: [ppo_epoch, T*N*world], the same on every rank
        (include/simgan_hip.h: sg_ppo_update).  fetch_losses=False: queue the update and return None without waiting for
        it (the losses are read later through the results ring, simgan_amd/driver.py)."""
        rollouts._push([_lib.F_OBS, _lib.F_ACTIONS, _lib.F_VALUE_PREDS, _lib.F_RETURNS, _lib.F_LOGP])
        out = (C.c_float * 3)()
        self._calls += 1
        if perms is not None:
            perms = _lib.as_i64(perms).reshape(self.ppo_epoch, -1)   # lengths and index ranges are checked behind the C ABI
        _lib.check(self.lib.sg_ppo_update(self.h, rollouts.h, None if perms is None else _lib.i64ptr(perms),
                                          0 if perms is None else perms.size, (self.seed + self._calls) & (2 ** 64 - 1),
                                          out if fetch_losses else None))
        self._last_perm_shape = (self.ppo_epoch, rollouts.num_steps * rollouts.num_processes)
        return (float(out[0]), float(out[1]), float(out[2])) if fetch_losses else None

    def last_perms(self):
        """[ppo_epoch, T*N] permutations the last update() consumed (injected or library-drawn)."""
        perms = np.empty(self._last_perm_shape, np.int64)
        _lib.check(self.lib.sg_ppo_last_perms(self.h, _lib.i64ptr(perms), perms.size))
        return perms

    def get_adam(self):
        n = self.actor_critic.num_params
        m, v = np.empty(n, np.float32), np.empty(n, np.float32)
        step = C.c_int64(0)
        _lib.check(self.lib.sg_ppo_get_adam(self.h, _lib.fptr(m), _lib.fptr(v), n, C.byref(step)))
        return m, v, step.value

    def set_adam(self, m, v, step):
        m, v = _lib.as_f32(m).reshape(-1), _lib.as_f32(v).reshape(-1)
        _lib.check(self.lib.sg_ppo_set_adam(self.h, _lib.fptr(m), _lib.fptr(v), m.size, int(step)))
