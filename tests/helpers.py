"""Shared helpers for the parity tests: fixture loading and tolerances."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# north_star: "within 1e-4 rel fp32".  Single-op outputs are far tighter than this; multi-step
# trajectories (Adam amplifies round-off where v ~ 0) are checked at RTOL with a small ATOL floor.
RTOL = 1e-4
ATOL = 1e-5


def load(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    d = {k: z[k] for k in z.files}
    d["meta"] = json.loads(bytes(d["meta"]).decode()) if "meta" in d else {}
    return d


def relerr(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-30))


def assert_close(a, b, rtol=RTOL, atol=ATOL, what=""):
    a = np.asarray(a, np.float64).reshape(-1)
    b = np.asarray(b, np.float64).reshape(-1)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = np.abs(a - b)
    tol = atol + rtol * np.abs(b)
    bad = err > tol
    assert not bad.any(), (f"{what}: {bad.sum()}/{bad.size} out of tol; max abs err {err.max():.3e}, "
                           f"worst rel {np.max(err / (np.abs(b) + 1e-30)):.3e}")
