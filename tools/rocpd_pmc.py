"""Per-kernel PMC counter summary of a rocprofv3 (rocpd sqlite) run collected with --pmc.
Usage: python tools/rocpd_pmc.py <results.db>
  -> kernel, counter, dispatches, SUM over the counter's instances per dispatch (mean over dispatches), instances per
     dispatch, mean per instance, per-dispatch maximum over instances (mean over dispatches).
A counter has one sample per hardware instance and dispatch (SQ_*: 8 XCDs x 4 shader engines = 32; TCC_*: per channel, _sum
already folded; GRBM_*: one per XCD): chip totals are the `sum/dispatch` column, per-instance clocks (SQ_BUSY_CYCLES,
GRBM_GUI_ACTIVE) the `mean/instance` column."""
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
        f"select k, c, count(*), sum(n), sum(s), sum(m) from (select s.{name_col} as k, p.name as c, count(*) as n, sum(e.value) as s, "
        f"max(e.value) as m from {pe} e join {ip} p on e.pmc_id = p.id join {kd} d on e.event_id = d.event_id "
        f"join {ks} s on d.kernel_id = s.id group by e.event_id, p.name) group by k, c").fetchall()
    agg = {}
    for k, c, nd, ns, s, m in rows:
        a = agg.setdefault((k.split("(")[0][:60], c), [0, 0, 0.0, 0.0])
        a[0] += nd
        a[1] += ns
        a[2] += s
        a[3] += m
    print(f"# rocprofv3 --pmc summary of {path}")
    print(f"{'kernel':60s} {'counter':32s} {'dispatches':>10s} {'sum/dispatch':>16s} {'inst/disp':>9s} {'mean/instance':>16s} {'max/instance':>16s}")
    weight = {}
    for (k, c), (nd, ns, s, m) in agg.items():
        weight[k] = max(weight.get(k, 0.0), s)
    for (k, c), (nd, ns, s, m) in sorted(agg.items(), key=lambda kv: (-weight[kv[0][0]], kv[0][0], kv[0][1])):
        print(f"{k:60s} {c:32s} {nd:10d} {s / nd:16.2f} {ns / nd:9.1f} {s / ns:16.2f} {m / nd:16.2f}")


if __name__ == "__main__":
    main(sys.argv[1])
