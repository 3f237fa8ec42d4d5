"""How far do two float32 evaluations of the SAME PPO update drift apart?  (CPU, oracle against itself.)

A PPO update is E x M clipped-surrogate Adam steps.  Perturbing the rewards by 1e-6 relative (a few ulp: what any change
of summation order upstream produces) and replaying the identical update through oracle/sg_oracle.c moves the final policy
weights by up to ~3e-4 absolute (~1 % of the update's length in L2) at the north-star shape: the clip / min / max decisions of
rows sitting on a branch boundary flip, and Adam turns a flipped near-zero gradient into a full lr-sized step.  The three
losses the update reports still agree to <1e-4.  tests/test_gpu_benchpath.py sizes its trajectory-level gates from these
numbers (about twice the floor per workload); the measurements are kept in profiles/r03_parity_floor.json.

    python tools/trajectory_sensitivity.py [northstar|hopper|laikago|refine ...] [--json out.json]
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as orc  # noqa: E402

SHAPES = {   # bench.py WORKLOADS: (T, N, O, A, H, kind, feet, E, M, clip, lr)
    "northstar": (128, 512, 47, 12, 64, "mlp", 1, 10, 16, 0.2, 3e-4),
    "hopper": (128, 256, 14, 7, 100, "split", 1, 10, 16, 0.2, 3e-4),
    "laikago": (128, 512, 64, 28, 100, "split", 4, 10, 16, 0.2, 3e-4),
    "refine": (128, 256, 111, 12, 64, "mlp", 1, 10, 8, 0.1, 1.5e-4),
}


def measure(name):
    T, N, O, A, H, kind, feet, E, M, clip, lr = SHAPES[name]
    rng = np.random.default_rng(0)
    d = orc.dims(orc.KIND_SPLIT if kind == "split" else orc.KIND_MLP, O, A, H, feet)
    n = orc.policy_num_params(d)
    par0 = (rng.standard_normal(n) * 0.1).astype(np.float32)   # plausible init: small weights
    if kind == "mlp":
        par0[-A:] = -0.5
    obs = rng.standard_normal((T + 1, N, O)).astype(np.float32)
    noise = rng.standard_normal((T * N, A)).astype(np.float32)
    v, act, lp = orc.policy_act(d, par0, obs[:-1].reshape(-1, O), noise)
    actions, logp = act.reshape(T, N, A), lp.reshape(T, N)
    vp = np.zeros((T + 1, N), np.float32)
    vp[:T] = v.reshape(T, N)
    masks = (rng.random((T + 1, N)) > 0.01).astype(np.float32)
    bad = np.ones((T + 1, N), np.float32)
    rewards = np.clip(rng.standard_normal((T, N)), -10, 10).astype(np.float32)
    nv = orc.policy_forward(d, par0, obs[T])[0][:, 0]
    perms = np.stack([rng.permutation(T * N) for _ in range(E)]).astype(np.int64)
    cfg = orc.ppo_cfg(clip, E, M, 0.5, 0.0, lr, 1e-5, 0.5, True)
    outs, losses = [], []
    for eps in (0.0, 1e-6, 1e-5):
        rw = (rewards * (1 + eps * rng.standard_normal(rewards.shape))).astype(np.float32)
        ret, vp2 = orc.compute_returns(rw, vp, masks, bad, nv, 1, 0.99, 0.95, 1)
        p, ad = par0.copy(), orc.AdamState(n)
        t = time.time()
        ls = orc.ppo_update(d, p, ad, cfg, obs, actions, vp2, ret, logp, perms)
        print(name, "perturbation", eps, "losses", ls, f"{time.time() - t:.1f} s", flush=True)
        outs.append(p)
        losses.append(ls)
    # the same update once more, unperturbed, with every step's gradient summed in a different ORDER (the rows of a minibatch
    # split over 7 threads, partial sums added in chunk order: orc_ppo_grad_rows_mt) -- the perturbation any second float32
    # implementation applies at EVERY step, not once
    ret, vp2 = orc.compute_returns(rewards, vp, masks, bad, nv, 1, 0.99, 0.95, 1)
    adv = orc.advantages(ret[:-1], vp2[:-1])
    p, ad = par0.copy(), orc.AdamState(n)
    mb = T * N // M
    acc = np.zeros(3)
    t = time.time()
    for e in range(E):
        for k in range(M):
            G, sums = orc.ppo_grad_rows_mt(d, p, cfg, obs, actions, vp2, ret, logp, adv, perms[e, k * mb:(k + 1) * mb], 1.0 / mb, 7)
            acc += sums / mb
            orc.ppo_apply(p, G, ad, cfg)
    ls = tuple(acc / (E * M))
    print(name, "reordered sums, losses", ls, f"{time.time() - t:.1f} s", flush=True)
    outs.append(p)
    losses.append(ls)
    rec = {"shape": dict(T=T, N=N, O=O, A=A, H=H, kind=kind, feet=feet, E=E, M=M, clip=clip, lr=lr), "steps": E * M}
    for i, key in ((1, "perturb_1e-06"), (2, "perturb_1e-05"), (3, "reordered_sums")):
        e = np.abs(outs[i] - outs[0])
        tol = 5e-5 + 1e-4 * np.abs(outs[0])
        rec[key] = dict(
            rel_l2_of_update=float(np.linalg.norm(outs[i] - outs[0]) / np.linalg.norm(outs[0] - par0)), worst_entry=float(e.max()),
            frac_beyond_tol=float(np.mean(e > tol)), max_move=float(np.abs(outs[0] - par0).max()),
            loss_abs_diff=[float(abs(a - b)) for a, b in zip(losses[i], losses[0])])
        print(name, key, rec[key], flush=True)
    return rec


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    out = None
    if "--json" in sys.argv:
        out = sys.argv[sys.argv.index("--json") + 1]
        args = [a for a in args if a != out]
    res = {name: measure(name) for name in (args or ["northstar"])}
    if out:
        with open(out, "w") as f:
            json.dump({"what": "float32 self-sensitivity of one PPO update (oracle vs itself under a relative reward perturbation)",
                       "tool": "tools/trajectory_sensitivity.py", "workloads": res}, f, indent=1)
