"""Expert trajectory wire format -> the [N_e, F] float32 matrix the discriminator trains on.

Mirrors the reference's host-side helpers (same names, arguments and results):
  load_sas_wpast_from_pickle   my_pybullet_envs/utils.py:170-199
  select_and_merge_sas         my_pybullet_envs/utils.py:232-263
and the way a2c/main_gail_dyn_ppo.py:141-175 turns their output into the GAIL data loader.

The pickle is `{traj_idx: [tuple, ...]}`; every tuple is a list of 2L+1 vectors
`[s_t, s_t-1, .., s_t-L+1, a_t, .., a_t-L+1, s_t+1]` (collect_tarsim_traj.py:218-265).  Pure numpy: nothing here
touches the GPU; the resulting matrix goes to `Discriminator.set_expert` / `driver.ExpertLoader`.
"""
import pickle

import numpy as np


def load_sas_wpast_from_pickle(pathname, downsample_freq=1, load_num_trajs=None, start_idx=None, rng=None):
    """-> list of 2L+1 arrays, element i = [N, len_i] (all rows of all loaded trajectories, downsampled).

    The reference draws each trajectory's first row with `torch.randint(0, downsample_freq, (n_trajs,))`
    (utils.py:178-179).  Pass `start_idx` (one int per trajectory) to reproduce a particular draw, or `rng`
    (numpy Generator); default is offset 0 for downsample_freq == 1 and a fresh numpy draw otherwise."""
    if isinstance(pathname, (bytes, bytearray)):
        saved = pickle.loads(pathname)
    elif isinstance(pathname, dict):
        saved = pathname
    else:
        with open(pathname, "rb") as handle:
            saved = pickle.load(handle)
    n_trajs = len(saved)
    downsample_freq = int(downsample_freq)
    if start_idx is None:
        if downsample_freq <= 1:
            start_idx = np.zeros(n_trajs, np.int64)
        else:
            start_idx = (rng or np.random.default_rng()).integers(0, downsample_freq, size=n_trajs)
    start_idx = np.asarray(start_idx, np.int64)
    assert start_idx.shape == (n_trajs,), "one start offset per trajectory"
    sas = []
    for traj_idx, traj_tuples in saved.items():
        sas.extend(traj_tuples[int(start_idx[traj_idx])::downsample_freq])
        if load_num_trajs and traj_idx >= load_num_trajs - 1:
            break
    n_items = len(sas[0])
    return [np.array([np.asarray(row[item]) for row in sas]) for item in range(n_items)]


def select_and_merge_sas(sas, s_idx=np.array([0, ]), a_idx=np.array([0, ])):
    """[s_{t-i} for i in s_idx] | [a_{t-j} for j in a_idx] | s_{t+1}, per row (or for a single tuple)."""
    one_dim = np.asarray(sas[0]).ndim == 1
    parts = [np.asarray(x)[None, :] if one_dim else np.asarray(x) for x in sas]
    assert parts[-1].ndim == 2
    len_time_win = (len(sas) - 1) // 2      # half old s, half old a, next s
    cols = [parts[i] for i in s_idx] + [parts[len_time_win + j] for j in a_idx] + [parts[-1]]
    merged = np.concatenate([c.astype(np.float64) for c in cols], axis=1)
    return merged[0, :] if one_dim else merged


def expert_matrix(pathname, s_idx=(0,), a_idx=(0,), downsample_freq=1, load_num_trajs=None, start_idx=None, rng=None):
    """a2c/main_gail_dyn_ppo.py:141-167 in one call -> (float32 [N_e, F] matrix, n_rows)."""
    sas = load_sas_wpast_from_pickle(pathname, downsample_freq, load_num_trajs, start_idx, rng)
    merged = select_and_merge_sas(sas, s_idx=np.asarray(s_idx), a_idx=np.asarray(a_idx))
    return np.ascontiguousarray(merged, np.float32), merged.shape[0]


def gail_tar_length(n_expert_rows, gail_traj_num, downsample_freq):
    """Average expert episode length the alive-bonus offset is scaled by (a2c/main_gail_dyn_ppo.py:167)."""
    return n_expert_rows * 1.0 / gail_traj_num * downsample_freq
