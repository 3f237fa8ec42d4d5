"""Host-side helpers mirroring a2c/utils.py and a2c/baselines/common/running_mean_std.py
(a2c/ = third_party/a2c_ppo_acktr/ in the reference)."""
import numpy as np

try:  # torch is the host tensor carrier when present (the reference mains pass torch tensors)
    import torch
except Exception:  # pragma: no cover
    torch = None


_INSTANCES = [0]


def derive_seed(seed, salt, per_instance=False):
    """RNG seed of one stochastic object: a splitmix64 hash of (constructor seed, salt).  The streams that must agree
    across the ranks of a data-parallel run -- the discriminator's DataLoader shuffle and mixup alpha are GLOBAL draws,
    csrc/sg_disc.hip -- depend on nothing else, so they are the same on every rank whatever else a rank has built.
    `per_instance=True` (policies: action noise) also mixes in a per-process instance counter, so two policies built
    with the same constructor seed do not share a noise stream.  Override with `obj.seed = x`."""
    inst = 0
    if per_instance:
        _INSTANCES[0] += 1
        inst = _INSTANCES[0]
    x = (int(seed) * 0x9E3779B97F4A7C15 + int(salt) * 0xD6E8FEB86659FD93 + inst * 0xBF58476D1CE4E5B9) & (2 ** 64 - 1)
    x ^= x >> 30
    x = (x * 0xBF58476D1CE4E5B9) & (2 ** 64 - 1)
    x ^= x >> 27
    x = (x * 0x94D049BB133111EB) & (2 ** 64 - 1)
    x ^= x >> 31
    return x >> 2   # headroom: callers add a per-call counter


def to_host_tensor(a):
    """numpy -> torch CPU tensor sharing memory (or the array itself without torch)."""
    return torch.from_numpy(a) if torch is not None else a


def update_linear_schedule(optimizer, epoch, total_num_epochs, initial_lr):
    """a2c/utils.py:68-72 -- works on any object exposing param_groups with a mutable 'lr'."""
    lr = initial_lr - (initial_lr * (epoch / float(total_num_epochs)))
    for param_group in optimizer.param_groups:
        param_group['lr'] = lr


class RunningMeanStd(object):
    """a2c/baselines/common/running_mean_std.py:27-58 (float64 Chan merge); state layout
    (mean, var, count) matches sg_disc_relabel_rewards' rms_state."""

    def __init__(self, epsilon=1e-4, shape=()):
        self.mean = np.zeros(shape, 'float64')
        self.var = np.ones(shape, 'float64')
        self.count = epsilon

    def update(self, x):
        x = np.asarray(x)
        self.update_from_moments(np.mean(x, axis=0), np.var(x, axis=0), x.shape[0])

    def update_from_moments(self, batch_mean, batch_var, batch_count):
        delta = batch_mean - self.mean
        tot_count = self.count + batch_count
        new_mean = self.mean + delta * batch_count / tot_count
        m2 = self.var * self.count + batch_var * batch_count + np.square(delta) * self.count * batch_count / tot_count
        self.mean, self.var, self.count = new_mean, m2 / tot_count, tot_count

    def get_state(self):
        return [float(self.mean), float(self.var), float(self.count)]

    def set_state(self, st):
        self.mean, self.var, self.count = np.float64(st[0]), np.float64(st[1]), float(st[2])


def orthogonal(rng, rows, cols, gain=1.0):
    """nn.init.orthogonal_ semantics (QR of a Gaussian, sign-fixed), numpy RNG: the production
    initialiser.  Not bit-matching torch's generator; parity tests load reference weights."""
    flat = rng.standard_normal((rows, cols))
    if rows < cols:
        flat = flat.T
    q, r = np.linalg.qr(flat)
    q = q * np.sign(np.diag(r))
    if rows < cols:
        q = q.T
    return (gain * q).astype(np.float32)
