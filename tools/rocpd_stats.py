"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel launch count and duration stats.
Usage: python tools/rocpd_stats.py gpurun_out/prof/r1_results.db > profiles/<name>.txt"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in cur.execute(f"pragma table_info({kd})")]
    scols = [r[1] for r in cur.execute(f"pragma table_info({ks})")]
    name_col = "display_name" if "display_name" in scols else "kernel_name"
    rows = cur.execute(f"select s.{name_col}, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id").fetchall()
    stats = {}
    for name, st, en in rows:
        stats.setdefault(name, []).append(en - st)
    total = sum(sum(v) for v in stats.values())
    print(f"# rocprofv3 --kernel-trace summary of {path}")
    print(f"# {len(rows)} dispatches, {total / 1e6:.3f} ms total kernel time; columns: {cols}")
    print(f"{'kernel':60s} {'calls':>8s} {'total_ms':>10s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s}")
    for name, v in sorted(stats.items(), key=lambda kv: -sum(kv[1])):
        short = name.split("(")[0][:60]
        print(f"{short:60s} {len(v):8d} {sum(v) / 1e6:10.3f} {sum(v) / len(v) / 1e3:9.2f} {min(v) / 1e3:9.2f} "
              f"{max(v) / 1e3:9.2f} {100.0 * sum(v) / total:6.2f}")


if __name__ == "__main__":
    main(sys.argv[1])
