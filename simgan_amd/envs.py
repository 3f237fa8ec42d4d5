"""One environment pool per GPU: the part of a2c/envs.py:101-134 the learner depends on.

PyBullet environments stay on the host and are out of scope; what the update path needs from the reference's
`make_vec_envs` stack is (a) WHICH of the `num_processes` environments a rank owns, (b) the vectorised
`step(action) -> (obs, reward, done, infos)` contract with auto-reset (DummyVecEnv / ShmemVecEnv,
a2c/baselines/common/vec_env/dummy_vec_env.py:71-83), (c) `VecNormalize(ob=False)` return scaling of the rewards
(a2c/envs.py:120-125, a2c/baselines/common/vec_env/vec_normalize.py:50-58) and (d) `VecPyTorch`'s float32 / [N,1]
conversions (a2c/envs.py:199-210).  `make_vec_envs` below builds that stack for one rank's shard from environment
constructors; any gym-style object with reset() / step(a) works (the tests use a deterministic fake).
"""
import numpy as np

from .utils import RunningMeanStd, to_host_tensor


def shard_env_indices(num_processes, rank=0, world=1):
    """Global environment ids owned by `rank`: contiguous blocks of num_processes/world columns, so rank r's rollout
    columns are global columns [r*N/world, (r+1)*N/world)."""
    assert num_processes % world == 0, f"num_processes {num_processes} must divide by the world size {world}"
    n_loc = num_processes // world
    return list(range(rank * n_loc, (rank + 1) * n_loc))


class SerialVecEnv(object):
    """DummyVecEnv semantics (dummy_vec_env.py:71-83): step every environment in turn, reset the ones that finished
    and return the reset observation in their slot."""

    def __init__(self, env_fns):
        self.envs = [fn() for fn in env_fns]
        self.num_envs = len(self.envs)
        e0 = self.envs[0]
        self.observation_space = getattr(e0, "observation_space", None)
        self.action_space = getattr(e0, "action_space", None)

    def reset(self):
        return np.stack([np.asarray(e.reset()) for e in self.envs])

    def step(self, actions):
        assert len(actions) == self.num_envs
        obs, rews, dones, infos = [], np.zeros(self.num_envs, np.float32), np.zeros(self.num_envs, bool), []
        for i, e in enumerate(self.envs):
            o, rews[i], dones[i], info = e.step(actions[i])
            if dones[i]:
                o = e.reset()
            obs.append(np.asarray(o))
            infos.append(info)
        return np.stack(obs), rews, dones, infos

    def close(self):
        for e in self.envs:
            if hasattr(e, "close"):
                e.close()


class ReturnNormalizer(object):
    """VecNormalize(ob=False, ret=True).step_wait's reward path (vec_normalize.py:50-58):
        ret = ret*gamma + rews; ret_rms.update(ret); rews = clip(rews / sqrt(ret_rms.var + eps), +-cliprew); ret[news] = 0
    The running statistics are float64 (`self.ret = np.zeros(n)`); rewards keep the dtype numpy gives float32 / float64."""

    def __init__(self, num_envs, gamma=0.99, cliprew=10.0, epsilon=1e-8, ret=True):
        self.ret_rms = RunningMeanStd(shape=()) if ret else None
        self.ob_rms = None   # ob=False in every shipped configuration (a2c/envs.py:125)
        self.ret = np.zeros(num_envs)
        self.gamma, self.cliprew, self.epsilon = gamma, cliprew, epsilon

    def __call__(self, rews, news):
        self.ret = self.ret * self.gamma + rews
        if self.ret_rms:
            self.ret_rms.update(self.ret)
            rews = np.clip(rews / np.sqrt(self.ret_rms.var + self.epsilon), -self.cliprew, self.cliprew)
        self.ret[news] = 0.
        return rews

    def reset(self):
        self.ret = np.zeros_like(self.ret)


class EnvPool(object):
    """A rank's shard of the vectorised environments with the reference's wrapper stack: return-scaled rewards and
    host tensors shaped as VecPyTorch hands them to the main loop (obs float32 [N,O], reward float32 [N,1])."""

    def __init__(self, venv, gamma, global_ids):
        self.venv, self.global_ids = venv, list(global_ids)
        self.num_envs = venv.num_envs
        self.observation_space, self.action_space = venv.observation_space, venv.action_space
        self.normalizer = ReturnNormalizer(self.num_envs, gamma=0.99 if gamma is None else gamma, ret=gamma is not None)

    @property
    def ret_rms(self):
        return self.normalizer.ret_rms

    @property
    def ob_rms(self):
        return None

    def reset(self):
        self.normalizer.reset()
        return to_host_tensor(np.ascontiguousarray(self.venv.reset(), np.float32))

    def step(self, actions):
        a = actions.cpu().numpy() if hasattr(actions, "cpu") else np.asarray(actions)
        obs, rews, news, infos = self.venv.step(a)
        rews = self.normalizer(rews, news)
        return (to_host_tensor(np.ascontiguousarray(obs, np.float32)),
                to_host_tensor(np.ascontiguousarray(rews, np.float32).reshape(-1, 1)), news, infos)

    def close(self):
        self.venv.close()


def make_vec_envs(env_fn, seed, num_processes, gamma, rank=0, world=1, vec_cls=SerialVecEnv):
    """a2c/envs.py:101-134 for ONE rank: `env_fn(global_id, seed + global_id)` builds environment `global_id`
    (the reference seeds env i with seed + i, a2c/envs.py:68); the rank instantiates only the ids it owns."""
    ids = shard_env_indices(num_processes, rank, world)
    fns = [(lambda g=g: env_fn(g, seed + g)) for g in ids]
    return EnvPool(vec_cls(fns), gamma, ids)
