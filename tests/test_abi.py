"""CPU: the C-ABI library loads (no GPU needed for dlopen) and exports every symbol that
include/simgan_hip.h declares, and the ctypes prototypes in simgan_amd/_lib.py cover them all.
No compute entry point is called here."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "simgan_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sg_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_surface():
    syms = declared_symbols()
    for must in ("sg_ctx_create", "sg_policy_act", "sg_policy_evaluate", "sg_rollout_compute_returns",
                 "sg_ppo_update", "sg_disc_update_gail_dyn", "sg_disc_predict_reward", "sg_disc_relabel_rewards",
                 "sg_ctx_comm_init"):
        assert must in syms
    assert len(syms) >= 45


def test_library_exports_every_declared_symbol():
    from simgan_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "libsimgan_hip.so missing: run __graft_entry__.build()"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, f"declared in include/simgan_hip.h but not exported: {missing}"


def test_ctypes_prototypes_cover_the_header():
    from simgan_amd import _lib
    declared = set(declared_symbols())
    bound = set(_lib.PROTOTYPES)
    assert declared == bound, (sorted(declared - bound), sorted(bound - declared))
    lib = _lib.load()   # binds restype/argtypes for every prototype; raises if a symbol is absent
    assert lib.sg_version().startswith(b"simgan_hip")


def test_no_cpu_fallback_when_library_is_missing(monkeypatch):
    """The product path must fail loudly without the HIP extension (no oracle / CPU fallback)."""
    import pytest
    from simgan_amd import _lib
    monkeypatch.setattr(_lib, "_LIB", None)
    monkeypatch.setattr(_lib, "LIB_PATH", os.path.join(ROOT, "simgan_amd", "does_not_exist.so"))
    with pytest.raises(_lib.SimganHipError, match="not built"):
        _lib.load()


def test_product_package_never_imports_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use oracle/."""
    pkg = os.path.join(ROOT, "simgan_amd")
    pat = re.compile(r"import\s+oracle|from\s+oracle|libsg_oracle|orc_[a-z]+\s*\(|oracle\.oracle")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h", ".hpp")):
                src = open(os.path.join(dirpath, f), errors="replace").read()
                assert not pat.search(src), f"{os.path.join(dirpath, f)} uses the oracle"
