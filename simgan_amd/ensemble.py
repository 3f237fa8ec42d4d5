"""Batched device inference for the policies that live INSIDE the reference's environments.

The hybrid-sim environments call a batch-1 `actor_critic.act` in every worker every step: the behaviour policy
(hopper_env_combined_policy.py:313-317, laikago_env_combined_policy.py:425-429) and, in refinement mode, one of
five saved dynamics policies picked at random per step (`np_random.choice`, hopper...:113-140,211-216).  With the
pool's observations gathered into one [N, O] array per step the same work is ONE device launch
(`sg_policy_act_ensemble`, csrc/sg_policy.hip): row i runs through the resident weights of policy ind[i].
Environment stepping itself stays on the host (out of scope)."""
import ctypes as C
import os

import numpy as np

from . import _lib
from .utils import derive_seed, to_host_tensor


class PolicyEnsemble(object):
    def __init__(self, policies, seed=0):
        assert 0 < len(policies) <= 8, "1..8 ensemble members (SG_ENSEMBLE_MAX)"
        self.policies = list(policies)
        shapes = {(type(p), p.obs_dim, p.act_dim, p.hidden_size, p.critic_hidden, p.num_feet) for p in self.policies}
        assert len(shapes) == 1, "ensemble members must share kind, observation, action and hidden sizes"
        self.obs_dim, self.act_dim = self.policies[0].obs_dim, self.policies[0].act_dim
        self.lib = self.policies[0].lib
        self._handles = (_lib.H * len(self.policies))(*[p.h for p in self.policies])
        for p in self.policies:
            p._register_handle_user(self)
        self.seed = derive_seed(seed, 0xE5E, per_instance=True)
        self._calls = 0

    @classmethod
    def load(cls, policy_dir, env_name, iters=(80, 100, 120, 140, 160), ctx=None):
        """The reference's fixed ensemble: `<env>_<iter>.pt` for five iterations (hopper_env_combined_policy.py:113-140)."""
        from .checkpoint import load_policy
        return cls([load_policy(os.path.join(policy_dir, f"{env_name}_{int(i)}.pt"), ctx=ctx)[0] for i in iters])

    def __len__(self):
        return len(self.policies)

    def act(self, obs, ind=None, noise=None, deterministic=False, rng=None, full=False):
        """obs [N, O]; ind [N] = the member each row uses (default: one `rng.choice(K)` per row, as each environment
        draws per step); noise [N, A] standard normal draws (default: the library's generator) -> (actions [N, A], ind);
        full=True -> (value [N,1], action [N,A], action_log_probs [N,1], ind) like Policy.act."""
        obs = _lib.as_f32(obs).reshape(-1, self.obs_dim)
        n = obs.shape[0]
        if ind is None:
            ind = (rng or np.random.default_rng()).integers(0, len(self.policies), size=n)
        ind = np.ascontiguousarray(ind, np.int32).reshape(-1)
        assert ind.shape == (n,) and ind.min() >= 0 and ind.max() < len(self.policies)
        nz = None if noise is None else _lib.as_f32(noise).reshape(n, self.act_dim)
        value, action = np.empty((n, 1), np.float32), np.empty((n, self.act_dim), np.float32)
        logp = np.empty((n, 1), np.float32)
        self._calls += 1
        _lib.check(self.lib.sg_policy_act_ensemble(
            self._handles, len(self.policies), ind.ctypes.data_as(C.POINTER(C.c_int32)), _lib.fptr(obs), n,
            None if nz is None else _lib.fptr(nz), (self.seed + self._calls) & (2 ** 64 - 1), 1 if deterministic else 0,
            _lib.fptr(value), _lib.fptr(action), _lib.fptr(logp)))
        if full:
            return to_host_tensor(value), to_host_tensor(action), to_host_tensor(logp), ind
        return action, ind
