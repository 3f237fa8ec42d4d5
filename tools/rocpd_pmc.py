"""Per-kernel PMC counter summary of a rocprofv3 (rocpd sqlite) run collected with --pmc.
Usage: python tools/rocpd_pmc.py <results.db>   -> kernel, calls, counter, mean value per dispatch"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    t = lambda p: [x for x in tabs if x.startswith(p)][0]  # noqa: E731
    pe, ip, kd, ks = t("rocpd_pmc_event"), t("rocpd_info_pmc"), t("rocpd_kernel_dispatch"), t("rocpd_info_kernel_symbol")
    scols = [r[1] for r in cur.execute(f"pragma table_info({ks})")]
    name_col = "display_name" if "display_name" in scols else "kernel_name"
    rows = cur.execute(
        f"select s.{name_col}, p.name, e.value from {pe} e join {ip} p on e.pmc_id = p.id "
        f"join {kd} d on e.event_id = d.event_id join {ks} s on d.kernel_id = s.id").fetchall()
    agg = {}
    for k, c, v in rows:
        a = agg.setdefault((k.split("(")[0][:60], c), [0, 0.0])
        a[0] += 1
        a[1] += v
    print(f"# rocprofv3 --pmc summary of {path}")
    print(f"{'kernel':60s} {'counter':14s} {'samples':>8s} {'mean/dispatch':>16s}")
    for (k, c), (n, s) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{k:60s} {c:14s} {n:8d} {s / n:16.2f}")


if __name__ == "__main__":
    main(sys.argv[1])
