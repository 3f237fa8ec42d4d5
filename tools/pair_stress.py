"""Stress of k_ppo_pair's tagged-word swap: many short-lived agents with DIFFERENT weights at one shape, on several contexts at
once, each compared bit for bit with the two-launch result for its weights.  Freed row stacks are recycled by the allocator, so
every round's words land on memory that holds another agent's words for the SAME step tags: a reader that could be served a stale
line would accept another agent's value.  Run on the GPU box:  python tools/pair_stress.py [rounds] [threads]"""
import os
import sys
import threading

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import simgan_amd as sg  # noqa: E402
from simgan_amd import _lib  # noqa: E402


class Box:
    def __init__(self, shape):
        self.shape = tuple(shape)


rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
n_threads = int(sys.argv[2]) if len(sys.argv) > 2 else 4
T, N, O, A, H, f, M, E = 128, 256, 14, 7, 100, 1, 16, 1
n_seeds = 6
lib = _lib.load()


def make(ctx=None):
    kw = {} if ctx is None else {"ctx": ctx}
    pol = sg.SplitPolicy((O,), Box((A,)), base_kwargs={"hidden_size": H, "num_feet": f}, seed=31, **kw)
    return pol


pol0 = make()
ro = sg.RolloutStorage(T, N, (O,), Box((A,)), 1, 4)
ro.device_resident = True
_lib.check(lib.sg_rollout_fill_synthetic(ro.h, pol0.h, 5, 0.01))
_lib.check(lib.sg_rollout_compute_returns_policy(ro.h, pol0.h, 1, 0.99, 0.95, 1))
ro.sync_from_device()
fields = {k: getattr(ro, k).numpy().copy() for k in ("obs", "actions", "value_preds", "returns", "action_log_probs", "masks")}
base = pol0.get_flat_params()
rng = np.random.default_rng(0)
p0s = [(base + 0.02 * rng.standard_normal(base.size)).astype(np.float32) for _ in range(n_seeds)]
perms = [np.stack([rng.permutation(T * N) for _ in range(E)]).astype(np.int64) for _ in range(n_seeds)]


def update(ctx, r, s):
    pol = make(ctx)
    pol.set_flat_params(p0s[s])
    agent = sg.algo.PPO(pol, 0.2, E, M, 0.5, 0.01, lr=3e-4, eps=1e-5, max_grad_norm=0.5)
    ls = agent.update(r, perms=perms[s])
    return np.asarray(ls, np.float64), pol.get_flat_params()


os.environ["SG_PPO_PAIR"] = "0"
want = [update(None, ro, s) for s in range(n_seeds)]
os.environ["SG_PPO_PAIR"] = "1"
bad, errs = [], []


def work(i):
    try:
        ctx = _lib.Context(0)
        r = sg.RolloutStorage(T, N, (O,), Box((A,)), 1, 4, ctx=ctx)
        for k, v in fields.items():
            getattr(r, k).copy_(getattr(r, k).new_tensor(v))
        for it in range(rounds):
            s = (it * 5 + i) % n_seeds
            ls, p = update(ctx, r, s)
            if not (np.array_equal(p, want[s][1]) and np.array_equal(ls, want[s][0])):
                bad.append((i, it, s, int((p != want[s][1]).sum()), float(np.abs(p - want[s][1]).max()), ls.tolist(), want[s][0].tolist()))
    except Exception as e:  # noqa: BLE001
        errs.append(repr(e))


th = [threading.Thread(target=work, args=(i,)) for i in range(n_threads)]
[t.start() for t in th]
[t.join() for t in th]
print(f"{n_threads} threads x {rounds} rounds: {len(bad)} mismatches, {len(errs)} errors")
for b in bad[:10]:
    print("  mismatch (thread, round, seed, #weights, worst, losses, want):", b)
for e in errs[:5]:
    print("  error:", e)
sys.exit(1 if bad or errs else 0)
