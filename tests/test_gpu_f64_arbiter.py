"""GPU: whole-update parity judged by a float64 arbiter (round-5 verdict: "nobody has shown WHICH side is closer to the exact answer").

Per step and per epoch the HIP path holds 1e-4 against the float32 oracle (tests/test_gpu_parity.py, test_gpu_steplock.py).  Over a
whole update -- E x M clipped-surrogate Adam steps -- two float32 evaluations of the reference's algorithm part ways: a row on a
clip / min / max boundary flips branch in one and not in the other, Adam turns the flipped near-zero gradient into an lr-sized step,
and from that step on the trajectories diverge (profiles/r06_parity_f64.json: they sit at ~1e-5 of the update's length for 2-7
epochs and then jump to 1e-3 .. 1e-1 within an epoch or two).  Neither float32 result is "the" answer there.  This test runs one
update of every bench.py workload three times from an identical start with identical draws (tools/parity_f64.py): the HIP library,
oracle/sg_oracle.c, and oracle/sg_oracle_f64.c -- the same source compiled with float := double -- and requires that the HIP
path is NO FARTHER from the float64 trajectory than the float32 oracle is, epoch by epoch and for the update: a factor on the
oracle's own distance plus a small floor (two independent draws from one noise distribution), never a free-standing tolerance.
The quantities that are smooth in float32 (discriminator trajectory, rewards, returns, values) are required to sit at the same
distance from the arbiter as the oracle's to within float32 round-off.
The reference sequence: a2c/main_gail_dyn_ppo.py:255-304, a2c/algo/ppo.py:65-157."""
import json
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

# policy trajectory, in units of the float64 update's own length: HIP's distance <= EPOCH_FACTOR x the float32 oracle's + FLOOR after
# every epoch, <= UPDATE_FACTOR x + FLOOR after the last.  Measured (profiles/r06_parity_f64.json), HIP / oracle32 after the update:
# northstar 1.89e-2 / 1.91e-2, hopper 3.55e-2 / 3.51e-2, laikago 7.09e-2 / 9.31e-2, refine 5.2e-3 / 5.1e-3, hopper_ppo 1.2e-5 / 1.2e-5;
# the largest per-epoch ratio is refine's 2.0 at 3.7e-4 (epochs 4-5, just after its first branch flip).
EPOCH_FACTOR, UPDATE_FACTOR, FLOOR = 3.0, 2.0, 1e-3


def _record(workload, rec):
    path = os.environ.get("SG_PARITY_F64_RECORD")
    if not path:
        return
    try:
        with open(path) as f:
            doc = json.load(f)
    except (OSError, ValueError):
        doc = {"what": "tests/test_gpu_f64_arbiter.py under SG_PARITY_F64_RECORD (tools/parity_f64.py's record per workload)", "workloads": {}}
    doc["workloads"][workload] = rec
    with open(path, "w") as f:
        json.dump(doc, f, indent=1)


@pytest.mark.parametrize("workload", ["northstar", "hopper", "laikago", "refine", "hopper_ppo"])
def test_hip_is_no_farther_from_the_float64_trajectory_than_the_float32_oracle(workload):
    import parity_f64
    rec = parity_f64.run_workload(workload)
    _record(workload, rec)
    fails = []
    # smooth quantities: the same distance from the arbiter as the oracle's, to float32 round-off of the quantity's scale
    for k in ("d_losses", "d_params", "rewards", "d_returns", "nv", "returns"):
        if k not in rec:
            continue
        h, o = rec[k]["hip_vs_f64"], rec[k]["oracle32_vs_f64"]
        if not (h["rel_l2"] <= 2.0 * o["rel_l2"] + 2e-6 and h["max_abs"] <= 2.0 * o["max_abs"] + 2e-5):
            fails.append(f"{k}: HIP {h} vs oracle32 {o}")
    for e in rec["ppo_epochs"]:
        h, o = e["policy"]["hip_vs_f64"]["rel_l2_of_update"], e["policy"]["oracle32_vs_f64"]["rel_l2_of_update"]
        last = e is rec["ppo_epochs"][-1]
        lim = (UPDATE_FACTOR if last else EPOCH_FACTOR) * o + FLOOR
        if not h <= lim:
            fails.append(f"policy after epoch {e['epoch']}: HIP {h:.3e} from the float64 trajectory, oracle32 {o:.3e} (limit {lim:.3e})")
        for name, lh, lo, ref in zip(("value_loss", "action_loss", "dist_entropy"), e["losses_abs"]["hip_vs_f64"], e["losses_abs"]["oracle32_vs_f64"], e["losses_f64"]):
            if not lh <= 3.0 * lo + 2e-5 + 1e-4 * abs(ref):
                fails.append(f"{name} of epoch {e['epoch']}: |HIP - f64| {lh:.2e}, |oracle32 - f64| {lo:.2e}, value {ref:.4g}")
    u = rec["ppo_update_losses"]
    for name, dh, do, ref in zip(("value_loss", "action_loss", "dist_entropy"), u["hip_minus_f64"], u["oracle32_minus_f64"], u["f64"]):
        if not abs(dh) <= 3.0 * abs(do) + 2e-5 + 1e-4 * abs(ref):
            fails.append(f"{name} of the update: HIP - f64 {dh:.2e}, oracle32 - f64 {do:.2e}, value {ref:.4g}")
    v = rec["verdict"]
    print(f"{workload}: after the update HIP sits {v['hip_rel_l2']:.3e} of the update's length from the float64 trajectory, the float32 oracle {v['oracle32_rel_l2']:.3e}")
    assert not fails, fails
    assert np.isfinite(v["hip_rel_l2"]) and rec["ppo_epochs"][0]["policy"]["hip_vs_f64"]["rel_l2_of_update"] < 1e-3, "the first epoch must still be at round-off level"
