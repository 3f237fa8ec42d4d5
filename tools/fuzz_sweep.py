"""Extended randomised-shape parity sweep on the GPU box: the cases of tests/test_gpu_fuzz.py for seeds beyond the ones the
suite runs, every case two updates against the CPU oracle at the suite's tolerance (1e-4 relative fp32).
    python tools/fuzz_sweep.py [first_seed] [n_cases] [out.json]      -> summary JSON (cases, failures with their message)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import pytest  # noqa: E402
import simgan_amd as sg  # noqa: E402
import test_gpu_fuzz as fz  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100
out = sys.argv[3] if len(sys.argv) > 3 else None
mp = pytest.MonkeyPatch()
res = {k: {"cases": 0, "failures": []} for k in ("ppo", "disc_thin", "disc_wide", "ppo_widths_to_256", "disc_widths_to_256")}
t0 = time.time()
for seed in range(first, first + n):
    for name, fn in (("ppo", lambda: fz._ppo_case(sg, seed)),
                     ("disc_thin", lambda: fz.test_disc_random_shapes_vs_oracle(sg, seed, "thin", mp)),
                     ("disc_wide", lambda: fz.test_disc_random_shapes_vs_oracle(sg, seed, "wide", mp)),
                     # round 4: the whole range the reference's constructors accept (hidden / input widths up to 256);
                     # a refusal at creation counts as a failure
                     ("ppo_widths_to_256", lambda: fz._ppo_case(sg, seed, wide=True)),
                     ("disc_widths_to_256", lambda: fz._disc_case(sg, seed, "thin", mp, wide=True))):
        res[name]["cases"] += 1
        try:
            fn()
        except AssertionError as exc:
            res[name]["failures"].append({"seed": seed, "message": str(exc)[:300]})
        except Exception as exc:   # noqa: BLE001 -- a library error on an odd shape is a finding too
            res[name]["failures"].append({"seed": seed, "message": f"{type(exc).__name__}: {str(exc)[:300]}"})
mp.undo()
res["seconds"] = round(time.time() - t0, 1)
res["seeds"] = [first, first + n - 1]
print(json.dumps(res, indent=1))
if out:
    json.dump(res, open(out, "w"), indent=1)
