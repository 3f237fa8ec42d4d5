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


def assert_close_adam(a, b, lr, steps, rtol=RTOL, atol=ATOL, what="", max_outlier_frac=5e-4):
    """Post-Adam parameters against the oracle's.  Adam's step is lr * m / (sqrt(v) + eps): an element whose gradient is
    of the order of eps (1e-8 for the discriminator) turns float32 summation-order noise in g into an O(lr) difference
    per step, whatever the implementation.  So: everything within (rtol, atol) except at most `max_outlier_frac` of the
    elements, and those within the bound lr * steps that Adam itself guarantees."""
    a = np.asarray(a, np.float64).reshape(-1)
    b = np.asarray(b, np.float64).reshape(-1)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = np.abs(a - b)
    bad = err > atol + rtol * np.abs(b)
    assert bad.sum() <= max(1, int(max_outlier_frac * bad.size)), f"{what}: {bad.sum()}/{bad.size} out of tol; max abs err {err.max():.3e}"
    assert err.max() <= lr * steps, f"{what}: max abs err {err.max():.3e} exceeds lr * steps = {lr * steps:.3e}"
