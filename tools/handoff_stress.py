"""Where does a one-launch discriminator step (k_disc_step4) go wrong under load?  Short epochs (default ONE step) of the
one-launch form on a context of its own, thousands of times, while other contexts keep the chip busy; after every epoch the
parameters are compared with the two-launch result of an idle GPU.  With one step per epoch a mismatch shows the footprint
of the corrupted gradient: one 16 x 16 weight tile, or entries of the bias / w3 vectors.
Usage (GPU box): python tools/handoff_stress.py [epochs] [steps_per_epoch] [noise_threads]"""
import os
import sys
import threading

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import simgan_amd as sg  # noqa: E402
from simgan_amd import _lib  # noqa: E402

T, N, O, A, F, H, HD, B = 128, 512, 47, 12, 86, 64, 100, 128
epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
n_noise = int(sys.argv[3]) if len(sys.argv) > 3 else 5


class Box:
    def __init__(self, shape):
        self.shape = tuple(shape)


class Loader:
    def __init__(self, expert, batch_size):
        self.expert, self.batch_size = expert, batch_size


rng = np.random.default_rng(7)
feat = rng.standard_normal((T + 1, N, F)).astype(np.float32)
expert = rng.standard_normal((steps * B, F)).astype(np.float32)
eperm = rng.permutation(steps * B).astype(np.int64)
pperm = rng.permutation(T * N).astype(np.int64)
alpha = rng.random(steps * B).astype(np.float32)
p0 = sg.algo.gail.Discriminator(F, HD, None, seed=6).get_flat_params()


def epoch(ctx, r, fused, D=None):
    os.environ["SG_DISC_FUSED"] = "1" if fused else "0"
    if D is None:
        D = sg.algo.gail.Discriminator(F, HD, None, ctx=ctx, seed=11)
    D.set_flat_params(p0)
    D.set_adam(np.zeros_like(p0), np.zeros_like(p0), 0)
    ls = D.update_gail_dyn(Loader(expert, B), r, expert_perm=eperm, policy_perm=pperm, alpha=alpha)
    return np.asarray(ls, dtype=np.float64), D.get_flat_params(), D


ctx0 = _lib.Context.default()
r0 = sg.RolloutStorage(T, N, (O,), Box((A,)), 1, F)
r0.obs_feat.copy_(r0.obs_feat.new_tensor(feat))
want_l, want_p, _ = epoch(ctx0, r0, False)
# parameter layout of the flat vector (state_dict order): W1 [HD, F], b1, W2 [HD, HD], b2, w3 [1, HD], b3
names = [("W1", HD * F), ("b1", HD), ("W2", HD * HD), ("b2", HD), ("w3", HD), ("b3", 1)]
stop, errs = threading.Event(), []
obs = rng.standard_normal((16384, O)).astype(np.float32)
act = rng.standard_normal((16384, A)).astype(np.float32)


def noise(i):
    try:
        ctx = _lib.Context(0)
        pol = sg.Policy((O,), Box((A,)), base_kwargs={"recurrent": False, "hidden_size": H}, seed=40 + i, ctx=ctx)
        while not stop.is_set():
            pol.evaluate_actions(obs, None, None, act)
    except Exception as e:  # noqa: BLE001
        errs.append("noise: " + repr(e))


th = [threading.Thread(target=noise, args=(i,)) for i in range(n_noise)]
[t.start() for t in th]
bad = 0
lib = _lib.load()
has_log = hasattr(lib, "sg_debug_step4_log")   # a library built with -DSG_STEP4_VERIFY=1
NLOG = 91 * 8 * 32 * 64
ref_log = None


def fetch_log(D):
    import ctypes as C
    out = np.empty(NLOG, np.float32)
    lib.sg_debug_step4_log.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong]
    assert lib.sg_debug_step4_log(D.h, out.ctypes.data, NLOG) == 0
    return out.view(np.uint32).reshape(91, 8, 32, 64)


def fetch_chainlog(D):
    import ctypes as C
    out = np.empty(96 * 8 * 8 * 64, np.uint32)
    lib.sg_debug_step4_chainlog.argtypes = [C.c_void_p, C.c_void_p]
    assert lib.sg_debug_step4_chainlog(D.h, out.ctypes.data) == 0
    return out.reshape(96, 8, 8, 64)


ref_chain = None
try:
    ctx = _lib.Context(0)
    r = sg.RolloutStorage(T, N, (O,), Box((A,)), 1, F, ctx=ctx)
    r.obs_feat.copy_(r.obs_feat.new_tensor(feat))
    D = None
    for e in range(epochs):
        ls, p, D = epoch(ctx, r, True, D)
        if has_log and steps == 1 and ref_log is None and np.array_equal(p, want_p):
            ref_log = fetch_log(D)
            ref_chain = fetch_chainlog(D)
        if not np.array_equal(p, want_p) or not np.array_equal(ls, want_l):
            if has_log and steps == 1 and ref_log is not None:
                import ctypes as C
                st = np.zeros(512 * 4, np.int64)
                lib.sg_debug_step4_stamps.argtypes = [C.c_void_p, C.c_void_p]
                assert lib.sg_debug_step4_stamps(D.h, st.ctypes.data) == 0
                st = st.reshape(512, 4) * 10
                ch, tl = st[:96], st[96:96 + 104]
                tl = tl[tl[:, 0] > 0]
                t0 = ch[:, 0].min()
                print(f"epoch {e}: chain blocks start {ch[:, 0].min() - t0}..{ch[:, 0].max() - t0} ns, end {ch[:, 1].min() - t0}..{ch[:, 1].max() - t0}; "
                      f"tile blocks start {tl[:, 0].min() - t0}..{tl[:, 0].max() - t0}, contracted {tl[:, 1].min() - t0}..{tl[:, 1].max() - t0}, stored {tl[:, 2].min() - t0}..{tl[:, 2].max() - t0}", flush=True)
                late = np.argsort(ch[:, 0])[-4:]
                print("   latest chain starts: " + ", ".join(f"block {b}: {ch[b, 0] - t0}..{ch[b, 1] - t0}" for b in late), flush=True)
                cl = fetch_chainlog(D)
                wc = np.argwhere(cl != ref_chain)
                names_c = ["xor(w1)", "xor(w2)", "xor(w2t)", "h1", "dz1|z1b", "xor(w1t)|DZ2 sample", "h2", "u1"]
                seen = {}
                for blk, wave, j, lane in wc:
                    seen.setdefault((int(blk), int(wave), int(j)), []).append(int(lane))
                print(f"epoch {e}: chain-side log differs in {len(seen)} (block, wave, item) entries:", flush=True)
                for (blk, wave, j), lanes in list(seen.items())[:20]:
                    print(f"   chain block {blk} ({'mixup' if blk < 32 else 'BCE'}) wave {wave} {names_c[j]}: {len(lanes)} lanes, e.g. lane {lanes[0]}: "
                          f"{cl[blk, wave, j, lanes[0]]:08x} vs {ref_chain[blk, wave, j, lanes[0]]:08x}", flush=True)
                lg = fetch_log(D)
                w = np.argwhere(lg != ref_log)
                print(f"epoch {e}: {len(w)} consumed operand words differ from the reference epoch's", flush=True)
                for wg, wave, j, lane in w[:4]:
                    half, cc, s_, side = j >> 4, (j >> 3) & 1, (j >> 1) & 3, j & 1
                    c = (16 if half else 0) + wave + 8 * cc
                    print(f"   tile wg {wg} (xcd {wg // 13}, slot {wg % 13}) wave {wave} lane {lane} (col {lane & 15}, k {lane >> 4}): stacked row {16 * c + 4 * s_ + (lane >> 4)} "
                          f"side {'R' if side else 'L'}: consumed {lg[wg, wave, j, lane]:08x} ({lg[wg, wave, j, lane:lane + 1].view(np.float32)[0]:.6g}) "
                          f"expected {ref_log[wg, wave, j, lane]:08x} ({ref_log[wg, wave, j, lane:lane + 1].view(np.float32)[0]:.6g})", flush=True)
            bad += 1
            d = p != want_p
            off, parts = 0, []
            for nm, n in names:
                k = int(d[off:off + n].sum())
                if k:
                    idx = np.nonzero(d[off:off + n])[0]
                    if nm in ("W1", "W2"):
                        cols = F if nm == "W1" else HD
                        tiles = sorted({(int(i // cols) // 16, int(i % cols) // 16) for i in idx})
                        parts.append(f"{nm}: {k} entries in tiles {tiles[:12]}{'...' if len(tiles) > 12 else ''}")
                    else:
                        parts.append(f"{nm}: {k} entries {idx[:8].tolist()}")
                off += n
            if bad <= 12:
                print(f"epoch {e}: {int(d.sum())} parameters differ (worst {np.abs(p - want_p).max():.3g}; losses equal: {np.array_equal(ls, want_l)}; "
                      f"loss {ls.tolist()} vs {want_l.tolist()}): " + "; ".join(parts), flush=True)
finally:
    stop.set()
    [t.join(60) for t in th]
print(f"{bad} of {epochs} epochs of {steps} step(s) differed from the two-launch result; errors: {errs}")
