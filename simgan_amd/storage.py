"""RolloutStorage -- host mirror of a2c/storage.py:31-192 with a device-resident twin.

The reference mains read and write the buffers by slicing (`rollouts.obs[0].copy_(obs)`,
`rollouts.rewards[step] = ...`), so the attributes stay ordinary host tensors with the
reference's shapes.  The HBM copy inside libsimgan_hip.so is refreshed per operation:
  * default (drop-in) mode: every device operation first uploads the fields it reads -- those whose host tensor
    changed since the device copy last matched it -- and downloads the fields it writes: PCIe-inclusive, coherent with
    host-side edits.  "Changed" is read off the tensor itself: every in-place torch operation on the attribute or on
    a slice of it (`rollouts.rewards[step] = ...`, `rollouts.obs[0].copy_(obs)`, `insert`) bumps its `_version`; an
    attribute that is re-bound is a different tensor.  So `obs_feat` crosses PCIe once per rollout, not once per
    discriminator epoch.  Writes torch cannot see (through a `.numpy()` view of the attribute) need
    `mark_host_written()`; without torch (plain numpy mirrors) and with SG_ROLLOUT_ALWAYS_UPLOAD=1 every call uploads;
    SG_ROLLOUT_VERIFY=1 (debugging a port) checksums every field the tracker takes for unchanged and raises when it was
    written behind torch's back;
  * `device_resident = True` (driver / bench fast path): uploads are skipped, the device copy is
    the source of truth and `sync_from_device()` refreshes the host view on demand.
"""
import ctypes as C
import os
import weakref

import numpy as np

from . import _lib
from .utils import to_host_tensor, torch

_VERIFY = os.environ.get("SG_ROLLOUT_VERIFY") == "1"

# data pointer of a rollout's obs_feat host tensor -> the rollout: lets Discriminator.predict_reward_combined recognise the unchanged
# main's `rollouts.obs_feat[step + 1]` / `rollouts.masks[step]` slices (a2c/main_gail_dyn_ppo.py:276-280) and serve the loop's T
# calls from one fused launch (simgan_amd/algo/gail.py)
_BY_FEAT_PTR = weakref.WeakValueDictionary()


def rollout_of_feat_slice(t):
    """The RolloutStorage whose obs_feat tensor `t` is a slice of (torch view), or None."""
    base = getattr(t, "_base", None)
    if base is None or not hasattr(base, "data_ptr"):
        return None
    ro = _BY_FEAT_PTR.get(base.data_ptr())
    return ro if ro is not None and ro.obs_feat is base else None

_FIELD_ATTR = {
    _lib.F_OBS: "obs", _lib.F_OBS_FEAT: "obs_feat", _lib.F_ACTIONS: "actions",
    _lib.F_REWARDS: "rewards", _lib.F_VALUE_PREDS: "value_preds", _lib.F_RETURNS: "returns",
    _lib.F_LOGP: "action_log_probs", _lib.F_MASKS: "masks", _lib.F_BAD_MASKS: "bad_masks",
}


class RolloutStorage(object):
    def __init__(self, num_steps, num_processes, obs_shape, action_space,
                 recurrent_hidden_state_size, feat_len=0, ctx=None):
        T, N = int(num_steps), int(num_processes)
        if action_space.__class__.__name__ == 'Discrete':
            raise NotImplementedError("discrete action spaces never occur in SimGAN configs")
        if len(obs_shape) != 1:
            raise NotImplementedError("1-D observations only")
        O, A, F = int(obs_shape[0]), int(action_space.shape[0]), int(feat_len)
        self.ctx = ctx or _lib.Context.default()   # (first: creating the context is what initialises the HIP runtime, with its settings)
        self.lib = self.ctx.lib
        # the host tensors live in page-locked memory (sg_host_alloc): ordinary CPU tensors to whoever slices them, one DMA
        # per field at the link's speed to sg_rollout_upload / _download
        z = lambda *s: to_host_tensor(_lib.pinned_array(s, 0.0))  # noqa: E731
        o = lambda *s: to_host_tensor(_lib.pinned_array(s, 1.0))  # noqa: E731
        self.obs = z(T + 1, N, O)
        self.obs_feat = z(T + 1, N, F)
        self.recurrent_hidden_states = z(T + 1, N, recurrent_hidden_state_size)
        self.rewards = z(T, N, 1)
        self.value_preds = z(T + 1, N, 1)
        self.returns = z(T + 1, N, 1)
        self.action_log_probs = z(T, N, 1)
        self.actions = z(T, N, A)
        self.masks = o(T + 1, N, 1)
        # Masks that indicate whether it's a true terminal state or time limit end state
        self.bad_masks = o(T + 1, N, 1)
        if hasattr(self.obs_feat, "data_ptr") and self.obs_feat.numel():
            _BY_FEAT_PTR[self.obs_feat.data_ptr()] = self
        self.num_steps = T
        self.num_processes = N
        self.step = 0
        self.obs_dim, self.act_dim, self.feat_len = O, A, F
        self._device_resident = False
        self._host_written = False   # insert() wrote host slots since the last after_update()
        self._synced = {}            # field -> (weakref of the host tensor, its _version) when the device copy last matched it
        self._crcs = {}              # SG_ROLLOUT_VERIFY=1: field -> crc32 of the host content at that moment
        self.bytes_uploaded = 0      # host -> device traffic of this rollout so far (bench.py's `dropin` leg reads it)

        h = _lib.H()
        _lib.check(self.lib.sg_rollout_create(self.ctx.h, T, N, O, A, F, C.byref(h)))
        self.h = h

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.sg_rollout_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def to(self, device):
        return None  # host tensors stay on the host; the HBM twin lives in the library

    # ----------------------------------------------------------------- host <-> device
    def _host_np(self, field):
        t = getattr(self, _FIELD_ATTR[field])
        a = t.numpy() if hasattr(t, "numpy") else t
        if a.dtype != np.float32 or not a.flags["C_CONTIGUOUS"]:
            raise TypeError(f"rollouts.{_FIELD_ATTR[field]} must stay a contiguous float32 host tensor")
        return a

    @property
    def device_resident(self):
        return self._device_resident

    @device_resident.setter
    def device_resident(self, value):
        # resident operations change the device copy without telling the host mirrors: nothing is known to match afterwards
        self._device_resident = bool(value)
        self._synced = {}

    def _stamp(self, field):
        """What identifies the present content of a host field: (the tensor, its version counter), or None when writes to
        it cannot be seen (numpy mirrors)."""
        t = getattr(self, _FIELD_ATTR[field])
        v = getattr(t, "_version", None)
        return None if v is None else (weakref.ref(t), v)

    def _matches(self, field):
        was, now = self._synced.get(field), self._stamp(field)
        return was is not None and now is not None and was[0]() is now[0]() and was[1] == now[1]

    def mark_host_written(self, fields=None):
        """Tell the rollout that host fields were written behind torch's back (through a numpy view)."""
        for f in (fields if fields is not None else list(self._synced)):
            self._synced.pop(f, None)

    def _crc(self, field):
        import zlib
        return zlib.crc32(self._host_np(field))

    def sync_to_device(self, fields=None):
        for f in (fields if fields is not None else _FIELD_ATTR):
            a = self._host_np(f)
            if a.size:
                _lib.check(self.lib.sg_rollout_upload(self.h, f, _lib.fptr(a), a.size))
                self.bytes_uploaded += a.nbytes
            self._synced[f] = self._stamp(f)
            if _VERIFY:
                self._crcs[f] = self._crc(f)

    def sync_from_device(self, fields=None):
        for f in (fields if fields is not None else _FIELD_ATTR):
            a = self._host_np(f)
            if a.size:
                _lib.check(self.lib.sg_rollout_download(self.h, f, _lib.fptr(a), a.size))
            self._synced[f] = self._stamp(f)   # (written through the numpy view: the version counter did not move)
            if _VERIFY:
                self._crcs[f] = self._crc(f)

    def _push(self, fields):
        if self.device_resident:
            return
        if os.environ.get("SG_ROLLOUT_ALWAYS_UPLOAD") == "1":
            return self.sync_to_device(fields)
        stale = [f for f in fields if not self._matches(f)]
        if _VERIFY:
            # SG_ROLLOUT_VERIFY=1: a field the tracker takes for unchanged is checksummed against what was uploaded -- a write that
            # did not go through torch (rollouts.x.numpy()[...] = ..., a .data view, an array shared through torch.from_numpy, C or
            # environment code filling the buffer) is then an error here instead of a silently stale device copy
            for f in fields:
                if f not in stale and f in self._crcs and self._crc(f) != self._crcs[f]:
                    raise RuntimeError(f"rollouts.{_FIELD_ATTR[f]} was written behind torch's version counter since its last upload "
                                       "(a numpy view / .data / shared buffer): call rollouts.mark_host_written() after such writes, "
                                       "or set SG_ROLLOUT_ALWAYS_UPLOAD=1")
        if stale:
            self.sync_to_device(stale)

    def _pull(self, fields):
        if not self.device_resident:
            self.sync_from_device(fields)

    def device_advantages(self):
        out = np.empty((self.num_steps, self.num_processes, 1), np.float32)
        _lib.check(self.lib.sg_rollout_download(self.h, _lib.F_ADVANTAGES, _lib.fptr(out), out.size))
        return to_host_tensor(out)

    # ------------------------------------------------------------- reference methods
    def insert(self, obs, recurrent_hidden_states, actions, action_log_probs,
               value_preds, rewards, masks, bad_masks, obs_feat=None):
        """a2c/storage.py:70-84"""
        s = self.step

        def put(dst, src):
            if torch is not None and hasattr(dst, "copy_"):
                dst.copy_(src if hasattr(src, "dim") else torch.as_tensor(np.asarray(src, np.float32)))
            else:
                dst[...] = np.asarray(src, np.float32)

        put(self.obs[s + 1], obs)
        if obs_feat is not None:
            put(self.obs_feat[s + 1], obs_feat)
        put(self.recurrent_hidden_states[s + 1], recurrent_hidden_states)
        put(self.actions[s], actions)
        put(self.action_log_probs[s], action_log_probs)
        put(self.value_preds[s], value_preds)
        put(self.rewards[s], rewards)
        put(self.masks[s + 1], masks)
        put(self.bad_masks[s + 1], bad_masks)
        self.step = (self.step + 1) % self.num_steps
        self._host_written = True

    def mod_reward(self, offset, reverse_l):
        """a2c/storage.py:86-94 (called by no shipped script): add the per-environment `offset` [N] to the rewards of the
        `reverse_l` most recently inserted steps, walking back from the insert cursor with wrap-around.  Host tensors; the
        write is seen by the next device call like any other in-place edit."""
        off = offset.reshape(-1, 1) if hasattr(offset, "reshape") else np.asarray(offset, np.float32).reshape(-1, 1)
        assert off.shape[0] == self.rewards.shape[1], (off.shape, self.rewards.shape)
        for back in range(1, int(reverse_l) + 1):
            self.rewards[(self.step - back) % self.num_steps] += off

    def after_update(self):
        """a2c/storage.py:96-101"""
        if self.device_resident:   # the device copy is the rollout; the host mirrors are refreshed by sync_from_device()
            _lib.check(self.lib.sg_rollout_after_update(self.h))
            if not self._host_written:
                return
            # a host-side collector (driver.collect) filled this rollout through insert(): its next act() reads host
            # slot 0 and its next sync_to_device() uploads it, so the host mirrors must roll over as well
        for name in ("obs", "obs_feat", "recurrent_hidden_states", "masks", "bad_masks"):
            t = getattr(self, name)
            t[0] = t[-1]
        self._host_written = False

    def compute_returns(self, next_value, use_gae, gamma, gae_lambda, use_proper_time_limits=True):
        """a2c/storage.py:103-142, on device (one thread per env column, reverse scan over T)."""
        self._push([_lib.F_REWARDS, _lib.F_VALUE_PREDS, _lib.F_RETURNS, _lib.F_MASKS, _lib.F_BAD_MASKS])
        nv = _lib.as_f32(next_value).reshape(-1)
        assert nv.size == self.num_processes
        _lib.check(self.lib.sg_rollout_compute_returns(self.h, _lib.fptr(nv), 1 if use_gae else 0,
                                                       float(gamma), float(gae_lambda),
                                                       1 if use_proper_time_limits else 0))
        self._pull([_lib.F_RETURNS, _lib.F_VALUE_PREDS])

    def feed_forward_generator(self, advantages, num_mini_batch=None, mini_batch_size=None, perm=None):
        """a2c/storage.py:144-192 -- host-side generator kept for API completeness (the device PPO / discriminator
        updates gather rows themselves and do not use it).  `perm` injects the sampler's permutation of the T*N row ids
        (the reference draws torch.randperm through SubsetRandomSampler, a2c/storage.py:159-162); default: numpy's
        global generator."""
        return feed_forward_batches(self, advantages, num_mini_batch, mini_batch_size, perm)


def feed_forward_batches(ro, advantages, num_mini_batch=None, mini_batch_size=None, perm=None):
    """The generator body on any object carrying the rollout's host buffers (needs no device): yields the reference's
    10-tuple (obs, hxs, actions, value_preds, returns, masks, old_logp, adv | None, obs_feat, next_obs_feat) per minibatch,
    rows in flattened (t, n) order t*N+n, "cur" fields from slots [:-1], next_obs_feat from obs_feat[1:], the ragged tail
    dropped (BatchSampler(drop_last=True))."""
    num_steps, num_processes = ro.rewards.shape[0:2]
    batch_size = num_processes * num_steps
    if mini_batch_size is None:
        assert batch_size >= num_mini_batch, (
            "PPO requires the number of processes ({}) "
            "* number of steps ({}) = {} "
            "to be greater than or equal to the number of PPO mini batches ({})."
            "".format(num_processes, num_steps, num_processes * num_steps, num_mini_batch))
        mini_batch_size = batch_size // num_mini_batch
    if perm is None:
        perm = np.random.permutation(batch_size)
    else:
        perm = _lib.as_i64(perm).reshape(-1)
        assert perm.size == batch_size and np.array_equal(np.sort(perm), np.arange(batch_size)), "perm must be a permutation of the T*N row ids"

    def flat(t, sl):
        a = t.numpy() if hasattr(t, "numpy") else t
        a = a[sl]
        return a.reshape(-1, a.shape[-1])

    cur, nxt, al = slice(None, -1), slice(1, None), slice(None)
    for k in range(batch_size // mini_batch_size):  # drop_last
        idx = perm[k * mini_batch_size:(k + 1) * mini_batch_size]
        g = lambda t, sl: to_host_tensor(np.ascontiguousarray(flat(t, sl)[idx]))  # noqa: E731
        adv = None if advantages is None else to_host_tensor(
            np.ascontiguousarray(_lib.as_f32(advantages).reshape(-1, 1)[idx]))
        yield (g(ro.obs, cur), g(ro.recurrent_hidden_states, cur), g(ro.actions, al),
               g(ro.value_preds, cur), g(ro.returns, cur), g(ro.masks, cur),
               g(ro.action_log_probs, al), adv, g(ro.obs_feat, cur), g(ro.obs_feat, nxt))
