"""A PPO fuzz seed the HIP path missed at the strict tolerance (tools/fuzz_sweep.py): is the miss float32's own?  The case is
rebuilt exactly as tests/test_gpu_fuzz._ppo_case builds it, both updates are run on the device and through the oracle, and for the
update that missed the ORACLE IS RUN AGAINST ITSELF from the same start with the stored log-probs perturbed by one or two ulps
(relative 2e-7) -- the evidence tests/test_gpu_steplock.py uses for its branch-flip epochs: if noise of that size moves the oracle's
own parameters as far as the HIP path sits from it, a row on a clip / min / max boundary took the other branch.
    python tools/fuzz_ppo_floor.py <seed> [<seed> ...]   (GPU box; SG_FLOOR_OUT=<file> writes the records)"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import simgan_amd as sg  # noqa: E402
import test_gpu_fuzz as fz  # noqa: E402
from oracle import oracle as orc  # noqa: E402

out = []
for seed in [int(x) for x in sys.argv[1:]]:
    rng = np.random.default_rng(7000 + seed)
    kind = "mlp" if seed % 2 == 0 else "split"
    if kind == "mlp":
        O, A, H, f = int(rng.integers(1, 120)), int(rng.integers(1, 20)), int(rng.choice([8, 24, 64, 100])), 1
    else:
        f = int(rng.integers(1, 5))
        O, A, H = int(rng.integers(2, 70)), 7 * f, int(rng.choice([16, 64, 100]))
    T, N = int(rng.integers(2, 12)), int(rng.integers(1, 40))
    M = int(rng.integers(1, min(6, T * N) + 1))
    E = int(rng.integers(1, 4))
    clip, vcoef, ecoef = float(rng.choice([0.1, 0.2])), 0.5, float(rng.choice([0.0, 0.01]))
    bk = {"recurrent": False, "hidden_size": H} if kind == "mlp" else {"hidden_size": H, "num_feet": f}
    pol = (sg.Policy if kind == "mlp" else sg.SplitPolicy)((O,), fz.Box((A,)), base_kwargs=bk, seed=seed)
    ro = sg.RolloutStorage(T, N, (O,), fz.Box((A,)), 1, 1)
    obs = rng.standard_normal((T + 1, N, O)).astype(np.float32)
    ro.obs.copy_(ro.obs.new_tensor(obs))
    v, a, lp, _ = pol.act(obs[:-1].reshape(-1, O), None, None, noise=rng.standard_normal((T * N, A)).astype(np.float32))
    act, logp = fz._npv(a).reshape(T, N, A), fz._npv(lp).reshape(T, N, 1)
    vp = np.concatenate([fz._npv(v).reshape(T, N, 1), np.zeros((1, N, 1), np.float32)])
    ret = (vp + rng.standard_normal(vp.shape) * 0.5).astype(np.float32)
    ro.actions.copy_(ro.actions.new_tensor(act)); ro.action_log_probs.copy_(ro.action_log_probs.new_tensor(logp))
    ro.value_preds.copy_(ro.value_preds.new_tensor(vp)); ro.returns.copy_(ro.returns.new_tensor(ret))
    p0 = (pol.get_flat_params() + 0.02 * rng.standard_normal(pol.num_params)).astype(np.float32)
    pol.set_flat_params(p0)
    agent = sg.algo.PPO(pol, clip, E, M, vcoef, ecoef, lr=3e-4, eps=1e-5, max_grad_norm=0.5)
    d = orc.dims(orc.KIND_MLP if kind == "mlp" else orc.KIND_SPLIT, O, A, H, f)
    cfg = orc.ppo_cfg(clip, E, M, vcoef, ecoef, 3e-4, 1e-5, 0.5, True)
    rec = {"seed": seed, "shape": f"{kind} O={O} A={A} H={H} f={f} T={T} N={N} M={M} E={E} clip={clip}", "updates": []}
    for u in range(2):
        perms = np.stack([rng.permutation(T * N) for _ in range(E)]).astype(np.int64)
        start = pol.get_flat_params()
        m0, v0, t0 = agent.get_adam()
        agent.update(ro, perms=perms)
        p_hip = pol.get_flat_params()

        def oracle(lp_):
            ad = orc.AdamState(start.size)
            ad.m[:], ad.v[:] = m0, v0
            ad.t.value = t0
            p = start.copy()
            orc.ppo_update(d, p, ad, cfg, obs, act, vp[..., 0], ret[..., 0], lp_, perms)
            return p

        p_orc = oracle(logp[..., 0])
        err = np.abs(p_hip.astype(np.float64) - p_orc)
        bad = int((err > 1e-5 + 1e-4 * np.abs(p_orc)).sum())
        move = float(np.linalg.norm(p_orc.astype(np.float64) - start))
        prng = np.random.default_rng(seed)
        self_l2, self_max, self_bad = 0.0, 0.0, 0
        for _ in range(5):
            lp2 = (logp[..., 0].astype(np.float64) * (1.0 + 2e-7 * prng.standard_normal(logp[..., 0].shape))).astype(np.float32)
            dev = oracle(lp2).astype(np.float64) - p_orc
            self_l2 = max(self_l2, float(np.linalg.norm(dev) / (move + 1e-30)))
            self_max = max(self_max, float(np.abs(dev).max()))
            self_bad = max(self_bad, int((np.abs(dev) > 1e-5 + 1e-4 * np.abs(p_orc)).sum()))
        rec["updates"].append({"update": u, "hip_vs_oracle_out_of_tol": bad, "of": int(err.size), "hip_vs_oracle_max_abs": float(err.max()),
                               "hip_vs_oracle_rel_l2_of_move": float(np.linalg.norm(p_hip.astype(np.float64) - p_orc) / (move + 1e-30)),
                               "oracle_vs_itself_under_ulp_noise": {"out_of_tol_max": self_bad, "max_abs": self_max, "rel_l2_of_move": self_l2}})
    out.append(rec)
    print(json.dumps(rec))
if os.environ.get("SG_FLOOR_OUT"):
    json.dump(out, open(os.environ["SG_FLOOR_OUT"], "w"), indent=1)
