"""ctypes binding of libsimgan_hip.so (the C ABI declared in include/simgan_hip.h).

The product path has NO CPU fallback: if the shared library is missing or fails to load, importing
anything that needs it raises.  Build it with `python -c "import __graft_entry__ as g; g.build()"`
or `make -C simgan_amd/csrc`.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsimgan_hip.so")

c_float_p = C.POINTER(C.c_float)
c_i64_p = C.POINTER(C.c_int64)
c_double_p = C.POINTER(C.c_double)
c_u8_p = C.POINTER(C.c_uint8)
c_int_p = C.POINTER(C.c_int)
H = C.c_void_p  # opaque handle

# field ids (include/simgan_hip.h)
F_OBS, F_OBS_FEAT, F_ACTIONS, F_REWARDS, F_VALUE_PREDS, F_RETURNS, F_LOGP, F_MASKS, F_BAD_MASKS, \
    F_ADVANTAGES = range(10)
POLICY_MLP, POLICY_SPLIT = 0, 1
PROF_DISC_CHAIN, PROF_DISC_WGRAD, PROF_PPO_FWD, PROF_PPO_BWD, PROF_PPO_REDUCE, PROF_RELABEL, PROF_PPO_ADAM = range(7)


class PPOConfig(C.Structure):
    _fields_ = [("clip_param", C.c_float), ("ppo_epoch", C.c_int), ("num_mini_batch", C.c_int),
                ("value_loss_coef", C.c_float), ("entropy_coef", C.c_float), ("lr", C.c_float),
                ("eps", C.c_float), ("max_grad_norm", C.c_float),
                ("use_clipped_value_loss", C.c_int)]


# name -> (restype, argtypes); every symbol of include/simgan_hip.h
PROTOTYPES = {
    "sg_last_error": (C.c_char_p, []),
    "sg_version": (C.c_char_p, []),
    "sg_ctx_create": (C.c_int, [C.c_int, C.POINTER(H)]),
    "sg_ctx_destroy": (C.c_int, [H]),
    "sg_ctx_synchronize": (C.c_int, [H]),
    "sg_ctx_device_info": (C.c_int, [H, C.c_char_p, C.c_int, c_int_p, c_i64_p]),
    "sg_comm_unique_id": (C.c_int, [c_u8_p]),
    "sg_comm_loopback_id": (C.c_int, [c_u8_p]),
    "sg_ctx_comm_init": (C.c_int, [H, c_u8_p, C.c_int, C.c_int]),
    "sg_ctx_comm_kind": (C.c_int, [H, c_int_p]),
    "sg_ctx_comm_set_peer": (C.c_int, [H, C.c_int]),
    "sg_ctx_comm_info": (C.c_int, [H, c_int_p, c_int_p]),
    "sg_ctx_set_disc_dp": (C.c_int, [H, C.c_int]),
    "sg_policy_create": (C.c_int, [H, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(H)]),
    "sg_policy_create2": (C.c_int, [H, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(H)]),
    "sg_policy_destroy": (C.c_int, [H]),
    "sg_policy_num_params": (C.c_int, [H, c_i64_p]),
    "sg_policy_set_params": (C.c_int, [H, c_float_p, C.c_int64]),
    "sg_policy_get_params": (C.c_int, [H, c_float_p, C.c_int64]),
    "sg_policy_act": (C.c_int, [H, c_float_p, C.c_int, c_float_p, C.c_uint64, C.c_int, c_float_p, c_float_p, c_float_p]),
    "sg_policy_get_value": (C.c_int, [H, c_float_p, C.c_int, c_float_p]),
    "sg_policy_evaluate": (C.c_int, [H, c_float_p, c_float_p, C.c_int, c_float_p, c_float_p, c_float_p]),
    "sg_policy_act_ensemble": (C.c_int, [C.POINTER(H), C.c_int, C.POINTER(C.c_int32), c_float_p, C.c_int, c_float_p, C.c_uint64,
                                         C.c_int, c_float_p, c_float_p, c_float_p]),
    "sg_rollout_create": (C.c_int, [H, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(H)]),
    "sg_rollout_destroy": (C.c_int, [H]),
    "sg_rollout_upload": (C.c_int, [H, C.c_int, c_float_p, C.c_int64]),
    "sg_rollout_download": (C.c_int, [H, C.c_int, c_float_p, C.c_int64]),
    "sg_rollout_upload_step": (C.c_int, [H, C.c_int, C.c_int, c_float_p, C.c_int64]),
    "sg_rollout_download_step": (C.c_int, [H, C.c_int, C.c_int, c_float_p, C.c_int64]),
    "sg_rollout_after_update": (C.c_int, [H]),
    "sg_rollout_compute_returns": (C.c_int, [H, c_float_p, C.c_int, C.c_float, C.c_float, C.c_int]),
    "sg_rollout_compute_returns_policy": (C.c_int, [H, H, C.c_int, C.c_float, C.c_float, C.c_int]),
    "sg_rollout_fill_synthetic": (C.c_int, [H, H, C.c_uint64, C.c_float]),
    "sg_rollout_count_dones": (C.c_int, [H, c_double_p]),
    "sg_ppo_create": (C.c_int, [H, H, C.POINTER(PPOConfig), C.POINTER(H)]),
    "sg_ppo_destroy": (C.c_int, [H]),
    "sg_ppo_set_lr": (C.c_int, [H, C.c_float]),
    "sg_ppo_update": (C.c_int, [H, H, c_i64_p, C.c_int64, C.c_uint64, c_float_p]),
    "sg_ppo_last_perms": (C.c_int, [H, c_i64_p, C.c_int64]),
    "sg_ppo_get_adam": (C.c_int, [H, c_float_p, c_float_p, C.c_int64, c_i64_p]),
    "sg_ppo_set_adam": (C.c_int, [H, c_float_p, c_float_p, C.c_int64, C.c_int64]),
    "sg_disc_create": (C.c_int, [H, C.c_int, C.c_int, C.POINTER(H)]),
    "sg_disc_destroy": (C.c_int, [H]),
    "sg_disc_num_params": (C.c_int, [H, c_i64_p]),
    "sg_disc_set_params": (C.c_int, [H, c_float_p, C.c_int64]),
    "sg_disc_get_params": (C.c_int, [H, c_float_p, C.c_int64]),
    "sg_disc_get_adam": (C.c_int, [H, c_float_p, c_float_p, C.c_int64, c_i64_p]),
    "sg_disc_set_adam": (C.c_int, [H, c_float_p, c_float_p, C.c_int64, C.c_int64]),
    "sg_disc_set_expert": (C.c_int, [H, c_float_p, C.c_int64]),
    "sg_disc_update_gail_dyn": (C.c_int, [H, H, C.c_int, c_i64_p, C.c_int64, c_i64_p, C.c_int64, c_float_p, C.c_int64, C.c_uint64,
                                          c_float_p, c_int_p]),
    "sg_disc_update_rows": (C.c_int, [H, c_float_p, C.c_int64, C.c_int, C.c_int, c_i64_p, C.c_int64, c_i64_p, C.c_int64, c_float_p,
                                      C.c_int64, C.c_uint64, c_float_p, c_int_p]),
    "sg_disc_predict_reward": (C.c_int, [H, c_float_p, C.c_int, C.c_float, c_float_p, C.c_float, c_float_p, c_float_p]),
    "sg_disc_predict_prob": (C.c_int, [H, c_float_p, C.c_int, c_float_p]),
    "sg_disc_predict_reward_steps": (C.c_int, [H, H, C.c_float, C.c_float, c_float_p, c_float_p]),
    "sg_disc_grad_pen": (C.c_int, [H, c_float_p, c_float_p, c_float_p, C.c_int, C.c_uint64, c_float_p]),
    "sg_disc_last_draws": (C.c_int, [H, c_i64_p, c_i64_p, c_float_p, c_i64_p]),
    "sg_disc_reset_returns": (C.c_int, [H]),
    "sg_disc_get_returns": (C.c_int, [H, c_float_p, C.c_int, c_int_p]),
    "sg_disc_set_returns": (C.c_int, [H, c_float_p, C.c_int]),
    "sg_disc_relabel_rewards": (C.c_int, [H, H, C.c_float, C.c_float, c_double_p]),
    "sg_disc_relabel_rewards_auto": (C.c_int, [H, H, C.c_float, C.c_double, C.c_int]),
    "sg_disc_set_rms": (C.c_int, [H, c_double_p]),
    "sg_disc_get_scalars": (C.c_int, [H, c_double_p]),
    "sg_results_publish": (C.c_int, [H, H, H, C.c_int]),
    "sg_results_fetch": (C.c_int, [H, C.c_int, c_double_p]),
    "sg_ctx_profile": (C.c_int, [H, C.c_int]),
    "sg_ctx_profile_read": (C.c_int, [H, C.c_int, c_double_p, c_i64_p]),
    "sg_ctx_profile_reset": (C.c_int, [H]),
    "sg_host_alloc": (C.c_int, [C.c_int64, C.POINTER(C.c_void_p)]),
    "sg_host_free": (C.c_int, [C.c_void_p]),
    "sg_ctx_mark": (C.c_int, [H, c_int_p]),
    "sg_ctx_mark_elapsed": (C.c_int, [H, C.c_int, C.c_int, c_double_p]),
}

# libsimgan_hip_test.so (csrc/sg_test_api.h): test hooks and probes, for tests/ and tools/ only
TEST_LIB_PATH = os.path.join(_HERE, "libsimgan_hip_test.so")
c_ll_p = C.POINTER(C.c_longlong)
TEST_PROTOTYPES = {
    "sg_test_last_error": (C.c_char_p, []),
    "sg_test_gemm": (C.c_int, [H, C.c_int, C.c_int, C.c_int, C.c_int, c_float_p, c_float_p, c_float_p]),
    "sg_test_gemm_bench": (C.c_int, [H, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_ll_p]),
    "sg_test_mfma_probe": (C.c_int, [H, C.c_int, c_float_p, c_float_p, c_float_p]),
    "sg_test_flag_probe": (C.c_int, [H, C.c_int, C.c_int, C.c_int, C.c_int, c_ll_p, c_float_p]),
    "sg_test_fetch_probe": (C.c_int, [H, C.c_int, C.c_int, C.c_int, c_ll_p]),
    "sg_test_pmc_calibrate": (C.c_int, [H, C.c_int64]),
    "sg_test_pstep_probe": (C.c_int, [H, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_ll_p, c_int_p]),
    "sg_test_graph_state": (C.c_int, [H, H, c_int_p]),
    "sg_test_disc_phase_times": (C.c_int, [H, C.c_int, c_ll_p, C.c_int]),
    "sg_test_disc_step4_times": (C.c_int, [H, C.c_int, c_ll_p, C.c_int]),
    "sg_test_disc_gathers": (C.c_int, [H, c_ll_p]),
    "sg_test_ppo_phase_times": (C.c_int, [H, C.c_int, c_ll_p, C.c_int]),
    "sg_test_rng": (C.c_int, [H, C.c_int, C.c_int64, C.c_uint64, C.c_void_p]),
    "sg_test_tear_probe": (C.c_int, [H, C.c_int, C.c_int, C.c_int, c_ll_p]),
    "sg_test_raise_handoff_error": (C.c_int, [H, H]),
}
_TEST_LIB = None

_LIB = None


class SimganHipError(RuntimeError):
    pass


def load():
    """Load libsimgan_hip.so and bind every prototype.  Raises if the library is absent."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise SimganHipError(
            f"{LIB_PATH} not found: the HIP extension is not built. There is no CPU fallback; "
            "run `make -C simgan_amd/csrc` (hipcc, gfx950).")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _LIB = lib
    return lib


def check(rc):
    if rc != 0:
        raise SimganHipError(load().sg_last_error().decode(errors="replace"))


def load_test():
    """The test-hook library (never loaded by the product path).  The product library is loaded first: both share the
    process's HIP runtime and the hooks operate on its handles."""
    global _TEST_LIB
    if _TEST_LIB is not None:
        return _TEST_LIB
    load()
    if not os.path.exists(TEST_LIB_PATH):
        raise SimganHipError(f"{TEST_LIB_PATH} not found: run `make -C simgan_amd/csrc`")
    lib = C.CDLL(TEST_LIB_PATH)
    for name, (res, args) in TEST_PROTOTYPES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _TEST_LIB = lib
    return lib


def check_test(rc):
    if rc != 0:
        raise SimganHipError(load_test().sg_test_last_error().decode(errors="replace"))


def fptr(a):
    """float32 C-contiguous numpy array -> float* (no copy; caller keeps `a` alive)."""
    assert isinstance(a, np.ndarray) and a.dtype == np.float32 and a.flags["C_CONTIGUOUS"], (type(a), getattr(a, "dtype", None))
    return a.ctypes.data_as(c_float_p)


def i64ptr(a):
    assert isinstance(a, np.ndarray) and a.dtype == np.int64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(c_i64_p)


def as_f32(x):
    """torch tensor / array-like -> contiguous float32 numpy (shares memory when possible)."""
    if hasattr(x, "detach"):
        x = x.detach().cpu().numpy()
    return np.ascontiguousarray(x, dtype=np.float32)


def as_i64(x):
    if hasattr(x, "detach"):
        x = x.detach().cpu().numpy()
    return np.ascontiguousarray(x, dtype=np.int64)


class _PinnedBlock(object):
    """One sg_host_alloc allocation, freed when the last array viewing it is gone."""

    def __init__(self, nbytes):
        self.lib = load()
        p = C.c_void_p()
        check(self.lib.sg_host_alloc(int(nbytes), C.byref(p)))
        self.ptr = p.value

    def __del__(self):
        try:
            if self.ptr:
                self.lib.sg_host_free(C.c_void_p(self.ptr))
                self.ptr = None
        except Exception:
            pass


_PINNED_WARNED = False


def pinned_array(shape, fill=0.0):
    """float32 numpy array of `shape` in page-locked host memory (include/simgan_hip.h: sg_host_alloc).  The allocation
    lives as long as any view of the array (numpy / torch.from_numpy keep the owner alive through `.base`)."""
    n = int(np.prod(shape)) * 4
    if n == 0:
        return np.full(shape, fill, np.float32)
    try:
        blk = _PinnedBlock(n)
    except SimganHipError as exc:
        # RLIMIT_MEMLOCK / the container's pinned-memory limit, or a rollout too large to pin: pageable memory works everywhere
        # the pinned buffer does (the upload path stages it), only slower
        global _PINNED_WARNED
        if not _PINNED_WARNED:
            _PINNED_WARNED = True
            import warnings
            warnings.warn(f"simgan_amd: page-locked host allocation of {n} bytes failed ({exc}); the rollout's host tensors fall back to "
                          "pageable memory (uploads are staged and slower)", RuntimeWarning, stacklevel=3)
        return np.full(shape, fill, np.float32)
    buf = (C.c_char * n).from_address(blk.ptr)
    buf._owner = blk
    a = np.frombuffer(buf, dtype=np.float32).reshape(shape)
    a[...] = fill
    return a


_CTX = {}


class Context:
    """One per process/GPU.  `default()` creates it on LOCAL_RANK (or device 0)."""

    def __init__(self, device=0):
        self.lib = load()
        h = H()
        check(self.lib.sg_ctx_create(int(device), C.byref(h)))
        self.h = h
        self.device = int(device)
        self.rank, self.world = 0, 1
        self.disc_sharded = False

    @staticmethod
    def default():
        dev = int(os.environ.get("LOCAL_RANK", "0"))
        if dev not in _CTX:
            try:
                _CTX[dev] = Context(dev)
            except SimganHipError as exc:
                # a launcher that narrows the visible devices per rank (HIP_/ROCR_VISIBLE_DEVICES) leaves each rank one
                # device, numbered 0; anything else about the device is a real error
                if dev == 0 or "out of range (1 visible)" not in str(exc):
                    raise
                _CTX[dev] = Context(0)
        return _CTX[dev]

    def make_default(self):
        """Make this context the one `Context.default()` returns in this process (objects built without ctx= land on it)."""
        _CTX[int(os.environ.get("LOCAL_RANK", "0"))] = self
        return self

    def synchronize(self):
        check(self.lib.sg_ctx_synchronize(self.h))

    def device_info(self):
        name = C.create_string_buffer(256)
        cu = C.c_int(0)
        mem = C.c_int64(0)
        check(self.lib.sg_ctx_device_info(self.h, name, 256, C.byref(cu), C.byref(mem)))
        return name.value.decode(), cu.value, mem.value

    def comm_init(self, unique_id: bytes, rank: int, world: int):
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        check(self.lib.sg_ctx_comm_init(self.h, buf, rank, world))
        self.rank, self.world = rank, world
        self.disc_sharded = os.environ.get("SG_DISC_DP", "") == "sharded"   # read by sg_ctx_comm_init as well

    def comm_kind(self):
        """'none' | 'rccl' | 'loopback'"""
        k = C.c_int(0)
        check(self.lib.sg_ctx_comm_kind(self.h, C.byref(k)))
        return ("none", "rccl", "loopback")[k.value & 3]

    def comm_peer(self):
        """True when the small float32 all-reduces run over the peer mesh (SG_COMM_PEER=1 at comm_init)."""
        k = C.c_int(0)
        check(self.lib.sg_ctx_comm_kind(self.h, C.byref(k)))
        return bool(k.value & 4)

    def comm_set_peer(self, enable):
        """Collective: every rank calls it with the same value (include/simgan_hip.h: sg_ctx_comm_set_peer)."""
        check(self.lib.sg_ctx_comm_set_peer(self.h, 1 if enable else 0))

    def comm_info(self):
        """(rank, world) as the communicator itself reports them."""
        r, w = C.c_int(0), C.c_int(0)
        check(self.lib.sg_ctx_comm_info(self.h, C.byref(r), C.byref(w)))
        return r.value, w.value

    def set_disc_dp(self, sharded):
        check(self.lib.sg_ctx_set_disc_dp(self.h, 1 if sharded else 0))
        self.disc_sharded = bool(sharded)

    def mark(self):
        """A timestamp on the library's stream (the host does not wait) -> its number."""
        i = C.c_int(0)
        check(self.lib.sg_ctx_mark(self.h, C.byref(i)))
        return i.value

    def mark_elapsed(self, a, b):
        ms = C.c_double(0)
        check(self.lib.sg_ctx_mark_elapsed(self.h, a, b, C.byref(ms)))
        return ms.value

    def marks_reset(self):
        check(self.lib.sg_ctx_mark(self.h, None))

    def profile(self, enable):
        check(self.lib.sg_ctx_profile(self.h, 1 if enable else 0))

    def profile_reset(self):
        check(self.lib.sg_ctx_profile_reset(self.h))

    def profile_read(self, which):
        ms = C.c_double(0)
        n = C.c_int64(0)
        check(self.lib.sg_ctx_profile_read(self.h, which, C.byref(ms), C.byref(n)))
        return ms.value, n.value


def comm_unique_id() -> bytes:
    buf = (C.c_uint8 * 128)()
    check(load().sg_comm_unique_id(buf))
    return bytes(buf)


def comm_loopback_id() -> bytes:
    """Id of a loopback communicator (shared-memory transport between contexts of one host, any devices)."""
    buf = (C.c_uint8 * 128)()
    check(load().sg_comm_loopback_id(buf))
    return bytes(buf)
