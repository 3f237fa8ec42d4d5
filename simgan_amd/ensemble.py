"""Batched device inference for the policies that live INSIDE the reference's environments.

The hybrid-sim environments call a batch-1 `actor_critic.act` in every worker every step: the behaviour policy
(hopper_env_combined_policy.py:313-317, laikago_env_combined_policy.py:425-429) and, in refinement mode, one of
five saved dynamics policies picked at random per step (`np_random.choice`, hopper...:113-140,211-216).  With the
environments' observations gathered into one [N, O] array per step the same work is a handful of N-row device
forwards: this class groups the rows by the policy they drew, runs one batched `act` per policy and scatters the
actions back.  Environment stepping itself stays on the host (out of scope)."""
import os

import numpy as np

from . import _lib


class PolicyEnsemble(object):
    def __init__(self, policies):
        assert len(policies) > 0
        self.policies = list(policies)
        a = {p.act_dim for p in self.policies}
        o = {p.obs_dim for p in self.policies}
        assert len(a) == 1 and len(o) == 1, "ensemble members must share observation and action sizes"
        self.obs_dim, self.act_dim = o.pop(), a.pop()

    @classmethod
    def load(cls, policy_dir, env_name, iters=(80, 100, 120, 140, 160), ctx=None):
        """The reference's fixed ensemble: `<env>_<iter>.pt` for five iterations (hopper_env_combined_policy.py:113-140)."""
        from .checkpoint import load_policy
        return cls([load_policy(os.path.join(policy_dir, f"{env_name}_{int(i)}.pt"), ctx=ctx)[0] for i in iters])

    def __len__(self):
        return len(self.policies)

    def act(self, obs, ind=None, noise=None, deterministic=False, rng=None):
        """obs [N, O]; ind [N] = the member each row uses (default: one `rng.choice(K)` per row, as each environment
        draws per step); noise [N, A] standard normal draws (default: the library's generator) -> actions [N, A]."""
        obs = _lib.as_f32(obs).reshape(-1, self.obs_dim)
        n = obs.shape[0]
        if ind is None:
            ind = (rng or np.random.default_rng()).integers(0, len(self.policies), size=n)
        ind = np.asarray(ind, np.int64).reshape(-1)
        assert ind.shape == (n,) and ind.min() >= 0 and ind.max() < len(self.policies)
        if noise is not None:
            noise = _lib.as_f32(noise).reshape(n, self.act_dim)
        out = np.empty((n, self.act_dim), np.float32)
        for k, pol in enumerate(self.policies):
            rows = np.nonzero(ind == k)[0]
            if rows.size == 0:
                continue
            _, a, _, _ = pol.act(obs[rows], None, None, deterministic=deterministic,
                                 noise=None if noise is None else noise[rows])
            out[rows] = a.numpy() if hasattr(a, "numpy") else np.asarray(a)
        return out, ind
