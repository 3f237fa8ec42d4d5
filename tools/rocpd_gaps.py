"""Inter-kernel gaps from a rocprofv3 (rocpd sqlite) kernel trace: for every ordered pair (previous kernel ->
next kernel) on the stream, the idle time between the end of one dispatch and the start of the next.
Usage: python tools/rocpd_gaps.py gpurun_out/prof/<host>/<pid>_results.db"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    scols = [r[1] for r in cur.execute(f"pragma table_info({ks})")]
    name_col = "display_name" if "display_name" in scols else "kernel_name"
    rows = cur.execute(f"select s.{name_col}, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
    gaps = {}
    for (n0, s0, e0), (n1, s1, e1) in zip(rows[:-1], rows[1:]):
        key = (n0.split("(")[0].split("<")[0][:28], n1.split("(")[0].split("<")[0][:28])
        gaps.setdefault(key, []).append(s1 - e0)
    print(f"# inter-kernel gaps (ns) of {path}: end of previous dispatch -> start of next")
    print(f"{'previous':30s} {'next':30s} {'count':>7s} {'avg_ns':>9s} {'p50_ns':>9s} {'min_ns':>9s} {'total_ms':>9s}")
    for key, v in sorted(gaps.items(), key=lambda kv: -sum(kv[1])):
        if len(v) < 3:
            continue
        v = sorted(v)
        print(f"{key[0]:30s} {key[1]:30s} {len(v):7d} {sum(v) / len(v):9.0f} {v[len(v) // 2]:9d} {v[0]:9d} {sum(v) / 1e6:9.3f}")
    span = rows[-1][2] - rows[0][1]
    busy = sum(e - s for _, s, e in rows)
    print(f"# span {span / 1e6:.3f} ms, busy {busy / 1e6:.3f} ms, idle {(span - busy) / 1e6:.3f} ms")


if __name__ == "__main__":
    main(sys.argv[1])
