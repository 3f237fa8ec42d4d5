"""CPU: host-side logic around the hot path -- the drop-in import surface (alias modules), the environment-pool shard
with VecNormalize's return scaling against a reference-generated fixture, and checkpoint-reading safety.  No GPU, no
compute entry point of libsimgan_hip.so is called."""
import json
import os
import pickle
import subprocess
import sys

import numpy as np
import pytest

from helpers import GOLDEN, assert_close, load

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = "/root/reference"


# ------------------------------------------------------------------------------------------ drop-in imports
def _run(code, pythonpath):
    env = dict(os.environ, PYTHONPATH=os.pathsep.join(pythonpath))
    return subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env, cwd="/tmp")


def test_reference_style_imports_resolve_to_the_shim():
    """The import block of a2c/main_gail_dyn_ppo.py:30-38 and a2c/main.py:30-35 (minus the CLI / environment modules,
    which stay the SimGAN checkout's own) with this repository ahead on PYTHONPATH."""
    code = (
        "from third_party.a2c_ppo_acktr import algo, utils\n"
        "from third_party.a2c_ppo_acktr.algo import gail\n"
        "from third_party.a2c_ppo_acktr.model import Policy\n"
        "from third_party.a2c_ppo_acktr.model_split import SplitPolicy\n"
        "from third_party.a2c_ppo_acktr.storage import RolloutStorage\n"
        "import simgan_amd as sg\n"
        "assert issubclass(Policy, sg.Policy) and issubclass(SplitPolicy, sg.SplitPolicy) and RolloutStorage is sg.RolloutStorage\n"
        "assert Policy.__module__ == 'third_party.a2c_ppo_acktr.model' and SplitPolicy.__module__ == 'third_party.a2c_ppo_acktr.model_split'\n"
        "assert algo.PPO is sg.algo.PPO and gail.Discriminator is sg.algo.gail.Discriminator\n"
        "assert utils.update_linear_schedule is sg.update_linear_schedule\n"
        "assert callable(utils.get_vec_normalize) and callable(utils.cleanup_log_dir)\n"
        "from third_party.a2c_ppo_acktr.baselines.common.running_mean_std import RunningMeanStd\n"
        "assert RunningMeanStd is sg.RunningMeanStd\n"
        "print('ok')\n")
    r = _run(code, [ROOT])
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stderr[-2000:]


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="needs the SimGAN checkout (development container only)")
def test_alias_package_overlays_a_simgan_checkout():
    """With the checkout BEHIND this repository on PYTHONPATH the modules this repository does not ship (arguments,
    the vendored baselines) still come from SimGAN, the hot-path ones from the shim."""
    code = (
        "import third_party.a2c_ppo_acktr as pkg\n"
        "from third_party.a2c_ppo_acktr.arguments import get_args\n"
        "from third_party.a2c_ppo_acktr.baselines.common import running_mean_std as r\n"
        "from third_party.a2c_ppo_acktr.baselines import logger\n"
        "from third_party.a2c_ppo_acktr.model import Policy\n"
        "import simgan_amd as sg\n"
        f"assert get_args.__code__.co_filename.startswith({REFERENCE!r}), get_args.__code__.co_filename\n"
        f"assert logger.__file__.startswith({REFERENCE!r})\n"
        f"assert r.__file__.startswith({ROOT!r}) and issubclass(Policy, sg.Policy)\n"
        "print('ok')\n")
    r = _run(code, [ROOT, REFERENCE])
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stderr[-2000:]


# --------------------------------------------------------------------------------------- environment pools
def test_return_normalizer_matches_vecnormalize_fixture():
    """a2c/envs.py:120-125 / vec_normalize.py:50-58 on the scripted rewards and dones the reference was run on."""
    from simgan_amd.envs import ReturnNormalizer
    g = load("vecnormalize")
    rn = ReturnNormalizer(g["raw"].shape[1], gamma=float(g["gamma"]))
    for t in range(g["raw"].shape[0]):
        out = rn(g["raw"][t], g["news"][t])
        assert np.array_equal(out, g["scaled"][t]), t            # same numpy expression order: bit-exact
        assert np.array_equal(rn.ret, g["ret"][t])
        assert_close([rn.ret_rms.mean, rn.ret_rms.var, rn.ret_rms.count], g["rms"][t], rtol=1e-12, atol=0, what="ret_rms")


class FakeEnv:
    """Deterministic stand-in for a PyBullet environment: obs = f(id, step), fixed-length episodes."""

    def __init__(self, gid, seed, obs_dim=5, ep_len=7):
        self.gid, self.seed, self.obs_dim, self.ep_len, self.t, self.episodes = gid, seed, obs_dim, ep_len + gid % 3, 0, 0

    def _obs(self):
        return np.cos(np.arange(self.obs_dim) * 0.3 + self.gid + 0.1 * self.t + self.episodes).astype(np.float32)

    def reset(self):
        self.t = 0
        self.episodes += 1
        return self._obs()

    def step(self, action):
        self.t += 1
        done = self.t >= self.ep_len
        info = {"bad_transition": True} if done and self.gid % 2 == 0 else {}
        return self._obs(), float(np.sum(action)) * 0.1 + self.gid, done, info


def test_env_pool_shards_cover_the_global_pool():
    from simgan_amd.envs import make_vec_envs, shard_env_indices
    assert sum((shard_env_indices(16, r, 4) for r in range(4)), []) == list(range(16))
    with pytest.raises(AssertionError):
        shard_env_indices(10, 0, 4)
    whole = make_vec_envs(lambda g, s: FakeEnv(g, s), seed=100, num_processes=8, gamma=0.99)
    shards = [make_vec_envs(lambda g, s: FakeEnv(g, s), seed=100, num_processes=8, gamma=0.99, rank=r, world=2) for r in range(2)]
    assert [e.seed for e in shards[1].venv.envs] == [104, 105, 106, 107]      # env i is seeded seed + i (a2c/envs.py:68)
    o_all = whole.reset().numpy()
    o_sh = np.concatenate([s.reset().numpy() for s in shards])
    assert np.array_equal(o_all, o_sh)
    rng = np.random.default_rng(0)
    for _ in range(12):
        a = rng.standard_normal((8, 3)).astype(np.float32)
        obs, rew, done, infos = whole.step(a)
        parts = [s.step(a[4 * r:4 * r + 4]) for r, s in enumerate(shards)]
        assert np.array_equal(obs.numpy(), np.concatenate([p[0].numpy() for p in parts]))     # auto-reset included
        assert np.array_equal(done, np.concatenate([p[2] for p in parts]))
        assert rew.shape == (8, 1) and rew.dtype.__str__() == "torch.float32"
        assert [("bad_transition" in i) for i in infos] == [("bad_transition" in i) for p in parts for i in p[3]]
    # return scaling uses per-pool statistics (each rank normalises with its own shard's running variance)
    assert whole.ret_rms.count == pytest.approx(1e-4 + 12 * 8) and shards[0].ret_rms.count == pytest.approx(1e-4 + 12 * 4)


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="needs the SimGAN checkout (development container only)")
def test_checkouts_shmem_vec_env_plugs_into_make_vec_envs():
    """SURVEY.md 8(f) N4 asks for a ShmemVecEnv-like pool per GPU; the multi-process pool itself stays the checkout's
    (a2c/envs.py:115-118, a2c/baselines/common/vec_env/shmem_vec_env.py:43-165: one process per environment, observations
    through shared arrays).  This is the adapter test: the checkout's class, unmodified, as `make_vec_envs(vec_cls=...)` for
    one rank's shard, stepping real worker processes, must give the same observations / rewards / dones as the in-process
    SerialVecEnv on the same deterministic environments (gym itself is absent here: the pool only needs the two spaces'
    shape / dtype, supplied by duck-typed stand-ins)."""
    code = r'''
import sys, types
import numpy as np
sys.path.insert(0, "tools")
from ref_import import install_stubs
install_stubs()
import gym.spaces                                             # the stand-in module: give the two classes the pool type-checks
gym.spaces.Dict, gym.spaces.Tuple = type("Dict", (), {}), type("Tuple", (), {})
gym.spaces = sys.modules["gym.spaces"]
from third_party.a2c_ppo_acktr.baselines.common.vec_env.shmem_vec_env import ShmemVecEnv
from simgan_amd.envs import make_vec_envs, SerialVecEnv

class Space:
    def __init__(self, shape): self.shape, self.dtype = tuple(shape), np.dtype(np.float32)

class FakeEnv:
    observation_space, action_space = Space((3,)), Space((2,))
    def __init__(self, gid, seed): self.gid, self.rng, self.t = gid, np.random.default_rng(seed), 0
    def reset(self):
        self.t = 0
        return self.rng.standard_normal(3).astype(np.float32) + self.gid
    def step(self, a):
        self.t += 1
        done = self.t >= 3 + self.gid % 2
        return (self.rng.standard_normal(3).astype(np.float32) + float(np.sum(a)), float(self.gid + 0.1 * self.t), done,
                {"bad_transition": True} if done and self.gid == 5 else {})
    def close(self): pass

shmem = lambda fns: ShmemVecEnv(fns, context="fork")          # fork: the workers inherit the gym stand-ins
pools = [make_vec_envs(FakeEnv, seed=40, num_processes=8, gamma=0.99, rank=1, world=2, vec_cls=c) for c in (shmem, SerialVecEnv)]
assert pools[0].global_ids == [4, 5, 6, 7] and len(pools[0].venv.procs) == 4
o = [p.reset().numpy() for p in pools]
assert np.array_equal(o[0], o[1]) and o[0].shape == (4, 3) and o[0].dtype == np.float32
rng = np.random.default_rng(1)
for _ in range(9):
    a = rng.standard_normal((4, 2)).astype(np.float32)
    r = [p.step(a) for p in pools]
    assert np.array_equal(r[0][0].numpy(), r[1][0].numpy())            # observations, auto-reset included
    # return-scaled rewards: ShmemVecEnv hands float64 rewards to VecNormalize (np.array of Python floats, shmem_vec_env.py:97),
    # DummyVecEnv / SerialVecEnv float32 ones (dummy_vec_env.py:41) -- the reference's own two pools differ by that rounding
    assert np.allclose(r[0][1].numpy(), r[1][1].numpy(), rtol=1e-6, atol=0) and r[0][1].shape == (4, 1) and r[0][1].dtype == r[1][1].dtype
    assert np.array_equal(np.asarray(r[0][2]), np.asarray(r[1][2]))
    assert [("bad_transition" in i) for i in r[0][3]] == [("bad_transition" in i) for i in r[1][3]]
pools[0].close()
print("ok")
'''
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd=ROOT,
                       env=dict(os.environ, PYTHONPATH=os.pathsep.join([ROOT, REFERENCE])))
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.stdout[-1000:], r.stderr[-3000:])


def test_feed_forward_generator_yields_the_reference_tuples():
    """RolloutStorage.feed_forward_generator against the reference's own output (tests/golden/ffgen.npz, written by
    tools/gen_golden.py from a2c/storage.py:144-192 with the sampler's permutation recorded): PPO form (num_mini_batch,
    advantages) and discriminator form (mini_batch_size 8 of 20 rows: two batches, ragged tail dropped, advantages None)."""
    import types
    from simgan_amd.storage import feed_forward_batches
    g = dict(np.load(os.path.join(GOLDEN, "ffgen.npz")))
    ro = types.SimpleNamespace(obs=g["obs"], obs_feat=g["obs_feat"], actions=g["actions"], rewards=g["rewards"],
                               value_preds=g["value_preds"], returns=g["returns"], action_log_probs=g["action_log_probs"],
                               masks=g["masks"], bad_masks=g["bad_masks"], recurrent_hidden_states=g["recurrent_hidden_states"])
    names = ("obs", "hxs", "actions", "value_preds", "returns", "masks", "old_logp", "adv", "obs_feat", "next_obs_feat")
    for tag, kw, adv in (("ppo", dict(num_mini_batch=3), g["advantages"]), ("disc", dict(mini_batch_size=8), None)):
        batches = list(feed_forward_batches(ro, adv, perm=g[f"{tag}_perm"], **kw))
        assert len(batches) == int(g[f"{tag}_n_batches"])
        for b, tup in enumerate(batches):
            assert len(tup) == 10
            for nm, t in zip(names, tup):
                key = f"{tag}_b{b}_{nm}"
                if key in g:
                    assert np.array_equal(np.asarray(t), g[key]), key
                else:
                    assert t is None and nm == "adv"
    with pytest.raises(AssertionError, match="permutation"):
        next(feed_forward_batches(ro, None, mini_batch_size=8, perm=np.zeros(20, np.int64)))
    with pytest.raises(AssertionError, match="PPO requires the number of processes"):
        next(feed_forward_batches(ro, None, num_mini_batch=21))
    # default draw: still a permutation, every row at most once
    rows = np.concatenate([np.asarray(t[3])[:, 0] for t in feed_forward_batches(ro, None, mini_batch_size=4)])
    assert sorted(rows.tolist()) == sorted(g["value_preds"][:-1].reshape(-1).tolist())


# -------------------------------------------------------------------------------------- checkpoint safety
class _Evil:
    def __reduce__(self):
        return (eval, ("__import__('os').environ.__setitem__('SG_PWNED', '1')",))


@pytest.mark.parametrize("payload", [_Evil(), [1, {"k": _Evil()}]])
def test_checkpoint_reader_refuses_globals_outside_the_allowlist(tmp_path, payload):
    """A crafted `.pt` whose pickle reduces on builtins.eval (or anything else a SimGAN checkpoint has no use for) is
    rejected before anything runs."""
    import torch
    from simgan_amd import checkpoint as ck
    os.environ.pop("SG_PWNED", None)
    for legacy in (True, False):
        path = str(tmp_path / f"evil_{int(legacy)}.pt")
        torch.save(payload, path, _use_new_zipfile_serialization=not legacy)
        with pytest.raises(pickle.UnpicklingError, match="refusing to resolve"):
            ck.read_reference_checkpoint(path)
        assert "SG_PWNED" not in os.environ


def test_checkpoint_reader_refuses_torch_and_os_callables(tmp_path):
    import torch
    from simgan_amd import checkpoint as ck

    class Hub:
        def __reduce__(self):
            return (torch.hub.load, ("x/y", "z"))

    class Sys:
        def __reduce__(self):
            return (os.system, ("true",))

    for i, obj in enumerate((Hub(), Sys())):
        path = str(tmp_path / f"e{i}.pt")
        torch.save([obj, None], path)
        with pytest.raises(pickle.UnpicklingError, match="refusing to resolve"):
            ck.read_reference_checkpoint(path)


def test_checkpoint_reader_refuses_nested_unrestricted_unpickle(tmp_path):
    """torch.storage._load_from_bytes is torch.load(BytesIO(b), weights_only=False): an allowlisted name that re-enters an
    UNRESTRICTED unpickler on bytes taken from the file (round-2 advisor finding, with a working proof of concept).  It is
    not on the allowlist -- legacy checkpoints rebuild storages through persistent_load and never need it -- so a file
    that smuggles a payload through it is refused and nothing runs."""
    import io
    import torch
    from simgan_amd import checkpoint as ck
    os.environ.pop("SG_PWNED", None)
    inner = io.BytesIO()
    torch.save(_Evil(), inner)

    class Nested:
        def __reduce__(self):
            return (torch.storage._load_from_bytes, (inner.getvalue(),))

    for legacy in (True, False):
        path = str(tmp_path / f"nested_{int(legacy)}.pt")
        torch.save([Nested(), None], path, _use_new_zipfile_serialization=not legacy)
        with pytest.raises(pickle.UnpicklingError, match="refusing to resolve"):
            ck.read_reference_checkpoint(path)
        assert "SG_PWNED" not in os.environ
    assert ("torch.storage", "_load_from_bytes") not in ck._ALLOWED


def test_lazy_policy_keeps_its_pickled_state_when_the_device_twin_cannot_be_built():
    """A Policy that torch.load is still assembling builds its device twin on first use (simgan_amd/model.py:_materialise).
    If that fails -- here: a module state that is not a policy's -- the pickled state must survive and every later access must
    raise the real cause again, not a bare AttributeError with the state gone (round-2 advisor finding)."""
    from simgan_amd.model import Policy
    p = Policy.__new__(Policy)
    p.__setstate__({"_modules": {"nothing": None}, "_parameters": {}})
    errs = []
    for _ in range(2):
        with pytest.raises(Exception) as ei:
            p.num_params
        errs.append((type(ei.value), str(ei.value)))
        assert p.__dict__.get("_pending") is not None
    assert errs[0] == errs[1] and not (errs[0][0] is AttributeError and errs[0][1] == "num_params")


def test_no_fixture_carries_reference_source():
    """A fixture is data: torch's legacy container would embed the source text of every pickled nn.Module class
    (tools/gen_golden.py:save_legacy_without_source disables that)."""
    needles = (b"class Policy(nn.Module)", b"class MLPBase", b"class DiagGaussian", b"class AddBias", b"class SplitPolicy",
               b"class Discriminator", b"def forward(", b"import torch")
    import zipfile
    for f in sorted(os.listdir(GOLDEN)):
        path = os.path.join(GOLDEN, f)
        blobs = [open(path, "rb").read()]
        if zipfile.is_zipfile(path):
            with zipfile.ZipFile(path) as z:
                blobs += [z.read(n) for n in z.namelist()]
        for b in blobs:
            for n in needles:
                assert n not in b, f"{f} contains {n!r}"


def test_legacy_container_fixture_is_still_legacy():
    """The source-free regeneration keeps the coverage it was there for: the non-zip container the shipped
    trained_models_*/ppo/*.pt use."""
    import zipfile
    assert not zipfile.is_zipfile(os.path.join(GOLDEN, "ckpt_policy_mlp.pt"))
    assert zipfile.is_zipfile(os.path.join(GOLDEN, "ckpt_policy_split.pt"))


def test_mod_reward_walks_back_from_the_cursor_with_wrap_around():
    """a2c/storage.py:86-94: rewards[(step - k) % T] += offset for k = 1..reverse_l (applied twice where the walk laps)."""
    import types

    from simgan_amd.storage import RolloutStorage
    from simgan_amd.utils import to_host_tensor
    T, N = 5, 3
    base = np.arange(T * N, dtype=np.float32).reshape(T, N, 1)
    off = np.array([0.5, -1.0, 2.0], np.float32)
    for step, back in ((2, 1), (2, 4), (0, 2), (3, 7)):
        ro = types.SimpleNamespace(rewards=to_host_tensor(base.copy()), step=step, num_steps=T)
        RolloutStorage.mod_reward(ro, to_host_tensor(off.copy()), back)
        want = base.copy()
        t = step
        for _ in range(back):
            t = (t - 1) % T
            want[t, :, 0] += off
        got = ro.rewards.numpy() if hasattr(ro.rewards, "numpy") else ro.rewards
        assert np.array_equal(got, want), (step, back)


# --------------------------------------------------------------------------------------- bench.py's stdout line
def test_bench_compact_line_keeps_the_contract_and_fits_a_2kb_tail():
    """The driver keeps 2,000 characters of stdout and the contract's keys: the compact line (bench.compact_line) must carry every
    contract key, the per-configuration numbers and the roofline / CPU-baseline objects inside that budget, for the N = 1 record
    of the driver's own command and for an N = 8 record (comm block, per-rank list, replica check)."""
    import bench
    full = json.load(open(os.path.join(ROOT, "profiles", "r06_v27_northstar_bench_full.json")))
    c = bench.compact_line(full, "gpurun_out/bench_full_northstar_n1.json")
    line = json.dumps(c)
    assert len(line) < 2000, len(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline",
              "cpu_baseline", "full_record"):
        assert k in c, k
    assert c["value"] == full["value"] and c["ms_per_step"] == full["ms_per_step"] and c["vs_baseline"] is None
    assert c["config"]["workload"].startswith("northstar") and len(c["config"]["workload"]) <= 120
    r = c["roofline"]
    assert r["bound"] == "mfma" and r["kernel"] == "k_disc_step4" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    assert 0.0 < r["mfma_busy"] < 1.0 and r["traffic"] > 1e6 and r["traffic_stale"] is False and r["counters_stale"] is False
    assert set(c["other_workloads_ms_envsteps_frac_busy"]) == {"hopper", "laikago", "refine", "hopper_ppo"}
    assert all(len(v) == 4 and v[0] > 0 for v in c["other_workloads_ms_envsteps_frac_busy"].values())
    assert c["dropin_ms"]["literal_main"] == full["dropin"]["literal_main"]["ms_per_step"]
    cb = c["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0 and cb["all_cores_value"] > cb["value"] and cb["unit"] == "env-steps/s"
    # an N = 8 record: the driver's scaling run
    full8 = dict(full, n_gpus=8, per_rank_ms_per_step=[26.3] * 8, replica_check={"ok": True, "weights_sha256_rank0": "0" * 16, "ranks": 8},
                 comm={"kind": "rccl", "nranks_reported_by_rccl": 8, "disc_mode": "replicated", "peer_allreduce": False, "allreduce_us": 12.3,
                       "allreduce_count_per_update": 160, "allreduce_form": "rccl", "backend": "x" * 300,
                       "disc_other_mode": {"mode": "sharded", "steps": 10, "ms_per_step": 80.0, "value": 6.5e6, "per_rank_ms_per_step": [80.0] * 8},
                       "peer": {"form": "peer mesh", "steps": 10, "ms_per_step": 40.0, "value": 1.3e7, "allreduce_us": 9.0, "per_rank_ms_per_step": [40.0] * 8}})
    full8.pop("other_workloads", None); full8.pop("dropin", None); full8.pop("cpu_baseline", None)
    c8 = bench.compact_line(full8, None)
    assert len(json.dumps(c8)) < 2000
    assert c8["comm"]["disc_mode"] == "replicated" and c8["comm"]["disc_other_mode"] == {"mode": "sharded", "value": 6.5e6, "ms_per_step": 80.0}
    assert c8["comm"]["peer"]["allreduce_us"] == 9.0 and len(c8["per_rank_ms_per_step"]) == 8 and c8["replica_check"]["ok"] is True
    assert "last_losses" in c8 and "full_record" not in c8
    # an error line (watchdog before the headline): value null, the error and the ranks' stages survive
    e = bench.compact_line({"metric": "m", "value": None, "unit": "env-steps/s", "n_gpus": 8, "steps": 1, "warmup": 1, "higher_is_better": True, "scaling": "weak",
                            "vs_baseline": None, "dtype": "f32", "data": "synthetic", "config": {"workload": "northstar"}, "ms_per_step": None,
                            "error": "watchdog: no progress", "stage": "init: communicator", "ranks": {"0": "x"}, "elapsed_s": 31.0})
    assert e["value"] is None and e["error"].startswith("watchdog") and e["stage"] and e["ranks"] == {"0": "x"}


def test_counter_summary_is_reproducible_from_the_committed_passes(tmp_path):
    """profiles/counters.json (what bench.py reads for roofline.mfma_busy) is tools/make_counters.py run on the committed
    per-pass summaries: re-derive the dominant kernels' MFMA-busy fractions from the raw counter lines."""
    import re
    doc = json.load(open(os.path.join(ROOT, "profiles", "counters.json")))
    tag = doc["tag"]
    assert set(doc["workloads"]) == {"northstar", "hopper", "laikago", "refine", "hopper_ppo"}
    for wl, kern in (("northstar", "k_disc_step4"), ("refine", "k_ppo_bwd")):
        vals = {}
        for line in open(os.path.join(ROOT, "profiles", f"{tag}_{wl}_pmc_sq_mfma.txt")):
            m = re.match(r"(?:void )?(\w+)(?:<[^>]*>)?\s+(\w+)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$", line)
            if m and m.group(1) == kern:
                vals[m.group(2)] = (float(m.group(4)), float(m.group(6)))
        busy = vals["SQ_VALU_MFMA_BUSY_CYCLES"][0] / (vals["SQ_BUSY_CYCLES"][1] * 1024)
        d = doc["workloads"][wl]["kernels"][kern]["derived"]
        assert abs(d["mfma_busy"] - busy) < 1e-4, (wl, d["mfma_busy"], busy)
        assert 0.5 < d["wave_wait"] + d["wave_issue_stall"] + d["wave_active"] <= 1.01
        assert d["mfma_flops"] > 0 and doc["workloads"][wl]["lib_sha256"]
