"""GPU diagnostic: a discriminator epoch as ONE persistent launch -- the real dataflow and byte counts of a step
(csrc/sg_test_pstep.hpp), GEMMs replaced by timed spins.  Prints us per step and where a step's time goes."""
import ctypes as C
import sys

import numpy as np

sys.path.insert(0, ".")
from simgan_amd import _lib  # noqa: E402

ctx = _lib.Context.default()
fn = _lib.load_test().sg_test_pstep_probe
S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
NAMES = {0: "fresh slot per step, plain weight loads", 1: "2-slot ring, sc1 weight loads"}
for cyc_phase, cyc_w in ((900, 600), (0, 0)):
    for mode in (0, 1, 2, 3, 4, 6, 7):
        st = (C.c_longlong * (3 * S * 4))()
        err = (C.c_int * 4)()
        _lib.check_test(fn(ctx.h, mode, S, 105, cyc_phase, cyc_w, st, err))
        s = np.array(st, dtype=np.int64).reshape(3, S, 4) * 10  # ns
        cm, cb, w = s
        k = slice(S // 4, S)  # steady state
        step = np.diff(cm[:, 0])[S // 4:]
        def med(x):
            return float(np.median(x)) / 1000.0
        print(f"phase {cyc_phase} cyc, W {cyc_w} cyc | mode {mode} ({NAMES[mode & 1]}{', early BCE operands' if mode & 2 else ''}{', W1 tiles first' if mode & 4 else ''}): "
              f"{med(step):.2f} us/step (p10 {np.percentile(step, 10) / 1000:.2f}, p90 {np.percentile(step, 90) / 1000:.2f}) | stale C {err[0]} W {err[1]} timeouts {err[2]} | "
              f"mixup C: weights visible -> phase 1 done {med(cm[k, 1] - cm[k, 0]):.2f}, -> last phase {med(cm[k, 2] - cm[k, 0]):.2f}, drain+publish {med(cm[k, 3] - cm[k, 2]):.2f}; "
              f"BCE C total {med(cb[k, 3] - cb[k, 0]):.2f}; "
              f"C published -> W saw all flags {med(w[k, 0] - cm[k, 3]):.2f}, W operands in {med(w[k, 1] - w[k, 0]):.2f}, spin+store {med(w[k, 2] - w[k, 1]):.2f}, "
              f"drain+publish {med(w[k, 3] - w[k, 2]):.2f}; W published -> C saw weights {med(cm[1:, 0][S // 4 - 1:] - w[:-1, 3][S // 4 - 1:]):.2f}", flush=True)
