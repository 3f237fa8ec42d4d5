"""Which float32 evaluation of a whole update is closer to the exact one?  (GPU box; ~1 min per workload.)

Over the E x M = 80-320 clipped-surrogate Adam steps of one PPO update the HIP path and the float32 oracle drift apart by more
than the 1e-4 that holds per step and per epoch (rows on a clip / min / max boundary flip branch; Adam turns a flipped near-zero
gradient into an lr-sized step).  This tool runs ONE update of every bench.py workload, from an identical start and with identical
injected draws, three times -- the HIP library, oracle/sg_oracle.c (float32) and oracle/sg_oracle_f64.c (the same source with
float := double, the arbiter) -- with the PPO part taken epoch by epoch (a PPO object of ppo_epoch = 1 called E times: the
advantages are those of the whole update, Adam's state carries over, so the E calls ARE the update), and records per quantity
    |HIP - f64|,  |oracle32 - f64|,  |HIP - oracle32|.
The reference sequence is a2c/main_gail_dyn_ppo.py:255-304 (a2c/main.py:246-256 for the plain-PPO workloads).
Usage: python tools/parity_f64.py [out.json] [workload ...]      -> profiles/r06_parity_f64.json is a run of this."""
import concurrent.futures
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class Box:
    def __init__(self, shape):
        self.shape = tuple(shape)


def dev(a, b):
    a, b = np.asarray(a, np.float64).reshape(-1), np.asarray(b, np.float64).reshape(-1)
    e = np.abs(a - b)
    return {"max_abs": float(e.max()), "max_rel": float(np.max(e / (np.abs(b) + 1e-30))), "rel_l2": float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-300))}


def three(hip, o32, o64):
    return {"hip_vs_f64": dev(hip, o64), "oracle32_vs_f64": dev(o32, o64), "hip_vs_oracle32": dev(hip, o32)}


def traj(p, ref_end, start):
    """Distance of a policy `p` from the float64 trajectory's `ref_end`, in units of that update's own length."""
    p, ref_end, start = (np.asarray(x, np.float64) for x in (p, ref_end, start))
    move = np.linalg.norm(ref_end - start)
    return {"rel_l2_of_update": float(np.linalg.norm(p - ref_end) / move), "worst_entry": float(np.abs(p - ref_end).max()),
            "frac_beyond_1e-4": float(np.mean(np.abs(p - ref_end) > 1e-5 + 1e-4 * np.abs(ref_end)))}


def run_workload(name):
    import bench
    import simgan_amd as sg
    from oracle import oracle as o32
    from oracle import oracle64 as o64
    from simgan_amd import _lib
    w = bench.WORKLOADS[name]
    T, N, O, A, F, H, Hd, B, E, M = w["T"], w["N"], w["O"], w["A"], w["F"], w["H"], w["Hd"], w["B"], w["E_p"], w["M"]
    has_d = bool(w["E_d"])
    lr = w.get("lr", 3e-4)
    pol, disc, agent, ro, loader, expert, learner = bench.build_problem(sg, w, seed=0)
    lib = _lib.load()
    _lib.check(lib.sg_rollout_fill_synthetic(ro.h, pol.h, 1234, 0.01))
    ro.sync_from_device()
    f2 = lambda t: t.numpy()[..., 0].copy()  # noqa: E731
    obs, obs_feat, actions = ro.obs.numpy().copy(), ro.obs_feat.numpy().copy(), ro.actions.numpy().copy()
    logp, vp, masks, bad, rewards_in = f2(ro.action_log_probs), f2(ro.value_preds), f2(ro.masks), f2(ro.bad_masks), f2(ro.rewards)
    p0 = pol.get_flat_params()
    dp0 = disc.get_flat_params() if has_d else None
    rng = np.random.default_rng(2026)
    n_d = min(w["Ne"] // B, T * N // B) if has_d else 0
    draws = [(rng.permutation(w["Ne"]).astype(np.int64), rng.permutation(T * N).astype(np.int64), rng.random(n_d * B).astype(np.float32))
             for _ in range(w["E_d"])]
    perms = np.stack([rng.permutation(T * N) for _ in range(E)]).astype(np.int64)

    # ---------------------------------------------------------------- the HIP library
    hip = {"epochs": []}
    if has_d:
        for ep, pp, al in draws:
            hip["d_losses"] = disc.update_gail_dyn(loader, ro, expert_perm=ep, policy_perm=pp, alpha=al)
        hip["d_params"] = disc.get_flat_params()
        disc.relabel_rewards_auto(ro, bench.GAMMA, 500.0, False)
        hip["rms"] = disc.scalars()[:3]
    _lib.check(lib.sg_rollout_compute_returns_policy(ro.h, pol.h, 1, bench.GAMMA, bench.LAM, 1))
    agent1 = sg.algo.PPO(pol, w["clip"], 1, M, 0.5, w.get("ecoef", 0.0), lr=lr, eps=1e-5, max_grad_norm=0.5)
    for e in range(E):
        losses = agent1.update(ro, perms=perms[e:e + 1])
        hip["epochs"].append({"losses": list(losses), "params": pol.get_flat_params()})
    ro.sync_from_device()
    hip["rewards"], hip["returns"], hip["nv"] = f2(ro.rewards), f2(ro.returns)[:T], f2(ro.value_preds)[T]
    if has_d:
        hip["d_returns"] = disc.returns.numpy()[:, 0].copy()

    # ---------------------------------------------------------------- the two oracles, side by side
    def oracle_run(o):
        R = o._R
        out = {"epochs": []}
        d = o.dims(o.KIND_MLP if w["kind"] == "mlp" else o.KIND_SPLIT, O, A, H, w["feet"])
        if has_d:
            dp, d_adam = dp0.astype(R), o.AdamState(dp0.size)
            for ep, pp, al in draws:
                out["d_losses"], nd = o.disc_update(F, Hd, dp, d_adam, expert, obs_feat, B, ep, pp, al)
                assert nd == n_d
            out["d_params"] = dp
            r_sa = o.alive_bonus(masks, T, N, 500.0)
            rewards, out["d_returns"], out["rms"] = o.relabel(F, Hd, dp, obs_feat, masks, bench.GAMMA, -r_sa, None, [0.0, 1.0, 1e-4])
        else:
            rewards = rewards_in.astype(R)
        out["rewards"] = rewards
        par, adam = p0.astype(R), o.AdamState(p0.size)
        out["nv"] = o.policy_forward(d, par, obs[T])[0][:, 0]
        ret, vp2 = o.compute_returns(rewards, vp, masks, bad, out["nv"], 1, bench.GAMMA, bench.LAM, 1)
        out["returns"] = ret[:T]
        cfg = o.ppo_cfg(w["clip"], 1, M, 0.5, w.get("ecoef", 0.0), lr, 1e-5, 0.5, True)
        for e in range(E):
            losses = o.ppo_update(d, par, adam, cfg, obs, actions, vp2, ret, logp, perms[e:e + 1])
            out["epochs"].append({"losses": list(losses), "params": par.copy()})
        return out

    t0 = time.perf_counter()
    with concurrent.futures.ThreadPoolExecutor(max_workers=2) as pool:
        f32, f64 = pool.submit(oracle_run, o32), pool.submit(oracle_run, o64)
        r32, r64 = f32.result(), f64.result()
    rec = {"shape": {k: w[k] for k in ("T", "N", "O", "A", "F", "H", "Hd", "E_p", "M", "E_d", "B", "kind")}, "oracle_seconds": round(time.perf_counter() - t0, 1)}
    if has_d:
        for k in ("d_losses", "d_params", "rewards", "d_returns", "rms"):
            rec[k] = three(hip[k], r32[k], r64[k])
    for k in ("nv", "returns"):
        rec[k] = three(hip[k], r32[k], r64[k])
    rec["ppo_epochs"] = []
    for e in range(E):
        h, a, b = hip["epochs"][e], r32["epochs"][e], r64["epochs"][e]
        rec["ppo_epochs"].append({"epoch": e + 1, "steps": (e + 1) * M,
                                  "policy": {"hip_vs_f64": traj(h["params"], b["params"], p0), "oracle32_vs_f64": traj(a["params"], b["params"], p0),
                                             "hip_vs_oracle32": traj(h["params"], a["params"], p0)},
                                  "losses_abs": {"hip_vs_f64": [abs(x - y) for x, y in zip(h["losses"], b["losses"])],
                                                 "oracle32_vs_f64": [abs(x - y) for x, y in zip(a["losses"], b["losses"])]},
                                  "losses_f64": b["losses"]})
    mean_l = lambda r: np.mean([ep["losses"] for ep in r["epochs"]], axis=0)  # noqa: E731  (PPO.update's return value: the mean over all E*M steps)
    lh, la, lb = mean_l(hip), mean_l(r32), mean_l(r64)
    rec["ppo_update_losses"] = {"f64": [float(x) for x in lb], "hip_minus_f64": [float(x) for x in lh - lb], "oracle32_minus_f64": [float(x) for x in la - lb],
                                "hip_rel": [float(abs(x - y) / (abs(y) + 1e-30)) for x, y in zip(lh, lb)],
                                "oracle32_rel": [float(abs(x - y) / (abs(y) + 1e-30)) for x, y in zip(la, lb)]}
    last = rec["ppo_epochs"][-1]["policy"]
    rec["verdict"] = {"hip_rel_l2": last["hip_vs_f64"]["rel_l2_of_update"], "oracle32_rel_l2": last["oracle32_vs_f64"]["rel_l2_of_update"],
                      "hip_no_farther_than_oracle32": bool(last["hip_vs_f64"]["rel_l2_of_update"] <= 1.25 * last["oracle32_vs_f64"]["rel_l2_of_update"])}
    return rec


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "parity_f64.json")
    names = sys.argv[2:] or ["northstar", "hopper", "laikago", "refine", "hopper_ppo"]
    doc = {"what": "one update from an identical start with identical injected draws through the HIP library, the float32 oracle and the float64 "
                   "arbiter (oracle/sg_oracle_f64.c); PPO epoch by epoch; written by tools/parity_f64.py", "workloads": {}}
    for n in names:
        doc["workloads"][n] = run_workload(n)
        v = doc["workloads"][n]["verdict"]
        u = doc["workloads"][n]["ppo_update_losses"]
        print(f"{n}: policy after the update, rel L2 of the update from the float64 trajectory: HIP {v['hip_rel_l2']:.3e}, oracle32 {v['oracle32_rel_l2']:.3e}; "
              f"losses HIP-f64 {['%.1e' % x for x in u['hip_minus_f64']]}, oracle32-f64 {['%.1e' % x for x in u['oracle32_minus_f64']]}", flush=True)
        with open(out, "w") as f:
            json.dump(doc, f, indent=1)


if __name__ == "__main__":
    main()
