"""Oracle (scalar, left-to-right) vs the vectorised CPU implementation (another summation order) on the fuzz seeds the HIP path
missed by ONE parameter: is the miss float32's own?"""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as orc

def case(seed):
    rng = np.random.default_rng(9000 + seed)
    F, Hd = [(86, 100), (25, 100), (int(rng.integers(1, 17)), int(rng.integers(1, 17))),
             (int(rng.integers(17, 130)), int(rng.integers(17, 130)))][seed % 4]
    B = int(rng.integers(1, 140)); T, N = int(rng.integers(2, 10)), int(rng.integers(1, 50))
    Ne = int(rng.integers(B, 4 * B + 20))
    if T * N < B: N = (B + T - 1) // T
    prng = np.random.default_rng(seed)
    parts = []
    for name, shape in [("0.weight", (Hd, F)), ("0.bias", (Hd,)), ("2.weight", (Hd, Hd)), ("2.bias", (Hd,)), ("4.weight", (1, Hd)), ("4.bias", (1,))]:
        fan_in = shape[1] if len(shape) == 2 else {"0.bias": F}.get(name, Hd)
        b = 1.0 / np.sqrt(fan_in)
        parts.append(prng.uniform(-b, b, size=int(np.prod(shape))).astype(np.float32))
    p0 = np.concatenate(parts)
    feat = rng.standard_normal((T + 1, N, F)).astype(np.float32)
    expert = (rng.standard_normal((Ne, F)) * 0.7 + 0.2).astype(np.float32)
    n_d = min(Ne // B, (T * N) // B)
    eperm = rng.permutation(Ne).astype(np.int64); pperm = rng.permutation(T * N).astype(np.int64)
    alpha = rng.random(n_d * B).astype(np.float32)
    return F, Hd, B, T, N, Ne, n_d, p0, feat, expert, eperm, pperm, alpha

out = []
for seed in [int(s) for s in sys.argv[1:]]:
    F, Hd, B, T, N, Ne, n_d, p0, feat, expert, eperm, pperm, alpha = case(seed)
    pa, aa = p0.copy(), orc.AdamState(p0.size)
    pb, ab = p0.copy(), orc.AdamState(p0.size)
    nf = feat[1:].reshape(T * N, F)
    rec = {"seed": seed, "shape": f"F={F} Hd={Hd} B={B} Ne={Ne} T={T} N={N}", "n_d": n_d}
    for u in range(1):
        for k in range(n_d):
            eb = expert[eperm[k * B:(k + 1) * B]]; pr = nf[pperm[k * B:(k + 1) * B]]; al = alpha[k * B:(k + 1) * B]
            Ga, _ = orc.disc_grad_rows(F, Hd, pa, eb, pr, al, 1.0 / B)
            Gb, _ = orc.disc_grad_rows_fast(F, Hd, pb, eb, pr, al, 1.0 / B)
            if k == 0:
                i = int(np.argmin(np.abs(Ga) + (Ga == 0) * 1e9))
                rec["step0_smallest_nonzero_|g|"] = float(abs(Ga[i])); rec["step0_|g|_median"] = float(np.median(np.abs(Ga)))
                rec["step0_max_|g_oracle - g_fast|"] = float(np.max(np.abs(Ga - Gb)))
            orc.adam_step(pa, Ga, aa, 1e-3, 1e-8); orc.adam_step(pb, Gb, ab, 1e-3, 1e-8)
    err = np.abs(pa.astype(np.float64) - pb); bad = err > 1e-5 + 1e-4 * np.abs(pa)
    rec["oracle_vs_fast_cpu_out_of_tol"] = int(bad.sum()); rec["oracle_vs_fast_cpu_max_abs"] = float(err.max()); rec["index_of_worst"] = int(err.argmax())
    out.append(rec); print(json.dumps(rec))
if os.environ.get("SG_FLOOR_OUT"): json.dump(out, open(os.environ["SG_FLOOR_OUT"], "w"), indent=1)
