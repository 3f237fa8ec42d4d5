"""GPU: the PPO trajectory STEP-LOCKED to the oracle, epoch by epoch, at every BASELINE.json shape.

tests/test_gpu_benchpath.py compares a whole update (E x M = 80-160 clipped-surrogate Adam steps) end to end, where float32
branch flips make the policy comparable only as a trajectory (relative L2 of the update).  That leaves epochs 2-10 -- where
the ratios have left 1 and the clip / min / max branches are live (a2c/algo/ppo.py:88-106) -- pinned only through that loose
gate.  Here every epoch is pinned on its own: before each epoch the oracle is re-seeded from the DEVICE's weights and Adam
moments, both run that one epoch (M steps) on the same permutation, and the epoch's three loss means and the post-epoch
parameters are compared at rtol 1e-4 / atol 1e-5 (parameters with the Adam-outlier rule of tests/helpers.py: an element
whose gradient is of the order of eps moves by O(lr) on summation-order noise in any float32 implementation).

Two phases per shape, 10 epochs each:
  A  from the data-collecting policy (ratios start at exactly 1 and drift away over the epochs);
  B  after one full learner update (bench.py's step) has moved the policy: the rollout's stored log-probs are then those of
     an OLD policy, so ratios are off 1 and rows sit on both sides of the clip from the first step on.
The PPO object runs with ppo_epoch = 1 on the same rollout (advantages are recomputed identically each call,
a2c/algo/ppo.py:66-68; Adam's step count and moments carry over), the library draws the permutations and exports them
(sg_ppo_last_perms).  Every deviation goes to the JSON file named by SG_STEPLOCK_RECORD (committed: profiles/r04_parity.json).
"""
import json
import os
import time

import numpy as np
import pytest

from helpers import ATOL, RTOL

pytestmark = pytest.mark.gpu


def _flat2(a):
    return a.numpy()[..., 0] if hasattr(a, "numpy") else np.asarray(a)[..., 0]


def _record(workload, rec):
    path = os.environ.get("SG_STEPLOCK_RECORD")
    if not path:
        return
    try:
        with open(path) as f:
            doc = json.load(f)
    except (OSError, ValueError):
        doc = {"what": "PPO step-locked to the oracle: per epoch (oracle re-seeded from the device's weights and Adam moments "
                       "before it), deviation of the epoch's loss means and of the post-epoch parameters; written by "
                       "tests/test_gpu_steplock.py under SG_STEPLOCK_RECORD", "tolerance": {"rtol": RTOL, "atol": ATOL},
               "workloads": {}}
    doc["workloads"][workload] = rec
    with open(path, "w") as f:
        json.dump(doc, f, indent=1)


@pytest.mark.parametrize("workload", ["northstar", "hopper", "laikago", "refine", "hopper_ppo"])
def test_ppo_epochs_step_locked_to_the_oracle(workload):
    import bench
    import simgan_amd as sg
    from oracle import oracle as orc
    from simgan_amd import _lib

    w = bench.WORKLOADS[workload]
    T, N, O, A, H, M, E = w["T"], w["N"], w["O"], w["A"], w["H"], w["M"], w["E_p"]
    lr = w.get("lr", 3e-4)
    pol, disc, agent, ro, loader, expert, learner = bench.build_problem(sg, w, seed=0)
    lib = _lib.load()
    _lib.check(lib.sg_rollout_fill_synthetic(ro.h, pol.h, 4321, 0.01))
    d = orc.dims(orc.KIND_MLP if w["kind"] == "mlp" else orc.KIND_SPLIT, O, A, H, w["feet"])
    ecoef = w.get("ecoef", 0.0)
    cfg = orc.ppo_cfg(w["clip"], 1, M, 0.5, ecoef, lr, 1e-5, 0.5, True)
    # the one-epoch agent on the same policy (its own Adam state, seeded from the learner's where a phase needs it)
    step_agent = sg.algo.PPO(pol, w["clip"], 1, M, 0.5, ecoef, lr=lr, eps=1e-5, max_grad_norm=0.5, seed=7)
    record = {"shape": {k: w[k] for k in ("T", "N", "O", "A", "H", "E_p", "M", "kind", "clip")}, "lr": lr, "phases": {}}
    failures = []
    t_orc = 0.0
    for phase in ("A_from_collecting_policy", "B_after_one_full_update"):
        if phase.startswith("B"):
            learner.update().resolve()                      # bench.py's step: D epochs, relabel, GAE, E x M PPO steps, after_update
            m_, v_, t_ = agent.get_adam()
            step_agent.set_adam(m_, v_, t_)                # the trajectory continues from the learner's optimizer state
        _lib.check(lib.sg_rollout_compute_returns_policy(ro.h, pol.h, 1, bench.GAMMA, bench.LAM, 1))
        ro.sync_from_device()
        obs, actions = ro.obs.numpy().copy(), ro.actions.numpy().copy()
        logp, vp, ret = _flat2(ro.action_log_probs).copy(), _flat2(ro.value_preds).copy(), _flat2(ro.returns).copy()
        epochs = []
        record["phases"][phase] = epochs
        # The E epochs on the device first (each starts where the previous one ended), every epoch's start and end state kept;
        # then the same E epochs through the scalar oracle, each FROM THE DEVICE'S STATE at its start -- independent of one
        # another, so they run side by side on the host's cores (the C calls release the GIL): the same 10 epochs per phase in a
        # tenth of the wall time.
        dev_epochs = []
        for e in range(E):
            p0 = pol.get_flat_params()
            m0, v0, t0 = step_agent.get_adam()
            losses_hip = step_agent.update(ro)
            perms = step_agent.last_perms()
            assert perms.shape == (1, T * N) and np.array_equal(np.sort(perms[0]), np.arange(T * N))
            p_hip = pol.get_flat_params()
            m1, v1, t1 = step_agent.get_adam()
            dev_epochs.append((p0, m0, v0, t0, losses_hip, perms, p_hip, m1, v1, t1))

        def oracle_epoch(st):
            p0, m0, v0, t0, _, perms = st[:6]
            ad = orc.AdamState(p0.size)
            ad.m[:], ad.v[:] = m0, v0
            ad.t.value = t0
            p_orc = p0.copy()
            losses_orc = orc.ppo_update(d, p_orc, ad, cfg, obs, actions, vp, ret, logp, perms)
            return p_orc, losses_orc, ad

        import concurrent.futures
        tic = time.perf_counter()
        with concurrent.futures.ThreadPoolExecutor(max_workers=max(1, min(E, (os.cpu_count() or 2) - 1))) as pool:
            orc_out = list(pool.map(oracle_epoch, dev_epochs))
        t_orc += time.perf_counter() - tic
        for e in range(E):
            p0, m0, v0, t0, losses_hip, perms, p_hip, m1, v1, t1 = dev_epochs[e]
            p_orc, losses_orc, ad = orc_out[e]
            assert t1 == t0 + M == ad.t.value
            # ---- compare
            lh, lo = np.asarray(losses_hip, np.float64), np.asarray(losses_orc, np.float64)
            loss_err = np.abs(lh - lo)
            loss_ok = bool(np.all(loss_err <= ATOL + RTOL * np.abs(lo)))
            err = np.abs(p_hip.astype(np.float64) - p_orc)
            bad = err > ATOL + RTOL * np.abs(p_orc)
            move = float(np.linalg.norm(p_orc.astype(np.float64) - p0))
            # how live the clip is in this epoch (ratio of the epoch's START policy against the stored log-probs)
            sel = np.arange(0, T * N, 61)
            _, lp_now, _ = orc.policy_evaluate(d, p0, obs[:-1].reshape(-1, O)[sel], actions.reshape(-1, A)[sel])
            ratio = np.exp(lp_now[:, 0].astype(np.float64) - logp.reshape(-1)[sel])
            clipped = float(np.mean((ratio < 1 - w["clip"]) | (ratio > 1 + w["clip"])))
            rec = dict(epoch=e, losses_hip=[float(x) for x in lh], losses_oracle=[float(x) for x in lo],
                       loss_abs_err=[float(x) for x in loss_err], loss_rel_err=[float(x) for x in loss_err / (np.abs(lo) + 1e-30)],
                       params_max_abs_err=float(err.max()), params_beyond_tol=int(bad.sum()), params_total=int(bad.size),
                       params_rel_l2_of_epoch_move=float(np.linalg.norm(p_hip.astype(np.float64) - p_orc) / (move + 1e-30)),
                       epoch_move_l2=move, ratio_min=float(ratio.min()), ratio_max=float(ratio.max()), frac_rows_outside_clip=clipped,
                       adam_m_max_abs_err=float(np.abs(m1 - ad.m).max()), adam_v_max_rel_err=float(np.max(np.abs(v1 - ad.v) / (np.abs(ad.v) + 1e-12))))
            epochs.append(rec)
            print(f"{workload} {phase[0]} epoch {e}: loss abs err {['%.1e' % x for x in loss_err]}, params max abs {err.max():.2e}, "
                  f"{int(bad.sum())}/{bad.size} beyond tol, ratio [{ratio.min():.3f}, {ratio.max():.3f}], {100 * clipped:.1f} % rows outside the clip")
            if not loss_ok:
                failures.append(f"{phase} epoch {e}: losses {lh.tolist()} vs oracle {lo.tolist()}")
            # parameters: everything within (rtol, atol) but at most 5e-4 of the elements (Adam's eps-sized gradients), and
            # those within the bound lr * steps Adam itself guarantees -- tests/helpers.py:assert_close_adam
            strict = bad.sum() <= max(1, int(5e-4 * bad.size)) and err.max() <= lr * M
            # A BRANCH-FLIP epoch: in one of the epoch's M steps a row sat within an ulp of a clip / min / max boundary
            # (a2c/algo/ppo.py:92-106) and took the other branch -- the step's gradient then differs by that ONE row's term
            # (1/B of the minibatch), which Adam's m / sqrt(v) spreads over a trunk's parameters.  Seen on MI355X in 1 of 20
            # epochs at the north-star shape (39 of 15,321 entries beyond tolerance, worst 3.3e-5 = 0.1 lr, 0.4 % of the
            # epoch's move) and 1 of 20 at Laikago (phase B, 47 % of the rows outside the clip, ratios 0 ... 2.9: 22 % of
            # the entries, worst 1.1e-4, 3 % of the move); the 98 other epochs of the five shapes agree to ONE ulp.  Such an
            # epoch must still match in its losses, stay within ONE Adam step (lr) on every entry, and within 2 % of the
            # epoch's move or within 4x what ulp-sized noise does to the oracle itself; and there may be few of them.
            flip = False
            if not strict:
                # Evidence, not a guess: the ORACLE AGAINST ITSELF on this very epoch, from the same start, with the stored
                # log-probs perturbed by one or two ulps (relative 2e-7: the size of float32 round-off in a log-prob that
                # sums 12-28 terms through three GEMMs in another order).  If float32 noise of that size moves the oracle's
                # own parameters as far as the HIP path sits from it, the deviation is the arithmetic's, not a kernel's.
                self_l2, self_max = 0.0, 0.0
                prng = np.random.default_rng(1000 * e + len(epochs))
                for _ in range(3):
                    ad2 = orc.AdamState(p0.size)
                    ad2.m[:], ad2.v[:] = m0, v0
                    ad2.t.value = t0
                    p2 = p0.copy()
                    logp2 = (logp.astype(np.float64) * (1.0 + 2e-7 * prng.standard_normal(logp.shape))).astype(np.float32)
                    orc.ppo_update(d, p2, ad2, cfg, obs, actions, vp, ret, logp2, perms)
                    dev = p2.astype(np.float64) - p_orc
                    self_l2 = max(self_l2, float(np.linalg.norm(dev) / (move + 1e-30)))
                    self_max = max(self_max, float(np.abs(dev).max()))
                rec["oracle_self_rel_l2_under_ulp_noise"] = self_l2
                rec["oracle_self_max_abs_under_ulp_noise"] = self_max
                within_floor = rec["params_rel_l2_of_epoch_move"] <= 4.0 * self_l2 and err.max() <= 4.0 * self_max
                flip = loss_ok and err.max() <= lr and (rec["params_rel_l2_of_epoch_move"] <= 2e-2 or within_floor)
            rec["verdict"] = "exact" if strict else ("branch_flip" if flip else "FAIL")
            if not strict and not flip:
                failures.append(f"{phase} epoch {e}: {int(bad.sum())}/{bad.size} parameters beyond tol, max abs err {err.max():.3e}, "
                                f"{rec['params_rel_l2_of_epoch_move']:.2e} of the epoch's move")
            assert move > 1e-3, "an epoch must move the policy far more than the tolerance"
        if phase.startswith("B"):
            assert max(r["frac_rows_outside_clip"] for r in epochs) > 0.01, "phase B is meant to run with a live clip"
    record["oracle_seconds"] = round(t_orc, 1)
    flips = [(ph, r["epoch"]) for ph, eps in record["phases"].items() for r in eps if r["verdict"] == "branch_flip"]
    record["branch_flip_epochs"] = [f"{ph[0]}{e}" for ph, e in flips]
    record["exact_epochs"] = sum(r["verdict"] == "exact" for eps in record["phases"].values() for r in eps)
    _record(workload, record)
    assert not failures, failures
    assert len(flips) <= 4, f"{len(flips)} of {2 * E} epochs deviate like a branch flip: too many to be ties -- {flips}"
