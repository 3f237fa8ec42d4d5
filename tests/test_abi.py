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


def exported_functions(path):
    """Function symbols in the dynamic symbol table (`nm -D --defined-only`, type T)."""
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", path], check=True, capture_output=True, text=True).stdout
    return sorted(line.split()[2] for line in out.splitlines() if len(line.split()) == 3 and line.split()[1] == "T")


def test_library_exports_every_declared_symbol():
    from simgan_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "libsimgan_hip.so missing: run __graft_entry__.build()"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, f"declared in include/simgan_hip.h but not exported: {missing}"


def test_library_exports_nothing_but_the_header():
    """-fvisibility=hidden + SG_API: the exported functions are exactly the header's declarations -- no internal helper, no
    C++-mangled symbol, no test hook (those live in libsimgan_hip_test.so)."""
    from simgan_amd import _lib
    exported = exported_functions(_lib.LIB_PATH)
    assert exported == declared_symbols(), (sorted(set(exported) - set(declared_symbols())), sorted(set(declared_symbols()) - set(exported)))
    assert not [s for s in exported if s.startswith("sg_test_")]


def test_test_library_exports_only_hooks_and_is_not_used_by_the_product():
    from simgan_amd import _lib
    assert os.path.exists(_lib.TEST_LIB_PATH), "libsimgan_hip_test.so missing: run __graft_entry__.build()"
    exported = exported_functions(_lib.TEST_LIB_PATH)
    assert exported == sorted(_lib.TEST_PROTOTYPES), (exported, sorted(_lib.TEST_PROTOTYPES))
    pkg = os.path.join(ROOT, "simgan_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py") and f != "_lib.py":
                src = open(os.path.join(dirpath, f), errors="replace").read()
                assert "load_test" not in src and "sg_test_" not in src, f"{f} uses the test hooks"
    assert "load_test" not in open(os.path.join(ROOT, "bench.py")).read()


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


def test_no_kernel_of_the_product_library_spills_to_scratch():
    """Every kernel of libsimgan_hip.so keeps its state in registers and LDS: no VGPR spills, no private-segment (scratch) bytes
    (round 5 shipped one that round-tripped through scratch; tools/check_codeobj.py reads the code objects' metadata notes)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_codeobj
    ks = check_codeobj.kernels(os.path.join(ROOT, "simgan_amd", "libsimgan_hip.so"))
    assert len(ks) > 50, len(ks)
    bad = [(k["name"], k["vgpr_spill"], k["scratch"]) for k in ks if k["vgpr_spill"] or k["scratch"]]
    assert not bad, bad
