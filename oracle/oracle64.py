"""The float64 ARBITER: oracle/oracle.py's functions over libsg_oracle64.so (sg_oracle_f64.c = sg_oracle.c with float := double).

TEST INFRASTRUCTURE ONLY, and not a parity oracle: the reference computes in float32, so these results differ from the
reference's by float32 round-off.  It answers one question -- when the float32 oracle and the HIP path drift apart over a whole
update, which of the two is closer to the same algorithm carried out exactly (tools/parity_f64.py)."""
import importlib.util
import os
import sys

_spec = importlib.util.spec_from_file_location("oracle._oracle64_impl", os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle.py"))
_mod = importlib.util.module_from_spec(_spec)
_mod.REAL_BITS = 64
_spec.loader.exec_module(_mod)
assert _mod._R.__name__ == "float64"
_this = sys.modules[__name__]
for _k, _v in vars(_mod).items():
    if not _k.startswith("__"):
        setattr(_this, _k, _v)
