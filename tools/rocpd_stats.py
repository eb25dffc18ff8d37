"""Per-kernel statistics out of a rocprofv3 rocpd SQLite file (what `--stats` prints as CSV in older formats).
usage: python tools/rocpd_stats.py <results.db> [--tail N]   (--tail: only the last N dispatches, and the gaps between them)"""
import json
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    ks = [t for t in tabs if "info_kernel_symbol" in t][0]
    rows = list(cur.execute(f"select s.kernel_name, d.start, d.end, d.grid_size_x, d.workgroup_size_x from {kd} d join {ks} s on d.kernel_id = s.id order by d.start"))
    if "--tail" in sys.argv:
        n = int(sys.argv[sys.argv.index("--tail") + 1])
        rows = rows[-n:]
        prev = None
        for name, st, en, g, w in rows:
            gap = (st - prev) / 1e3 if prev else 0.0
            print(f"{name[:60]:60s} grid {g // max(1, w):6d}  dur {(en - st) / 1e3:9.1f} us  gap {gap:8.1f} us")
            prev = en
        return
    agg = {}
    for name, st, en, g, w in rows:
        a = agg.setdefault(name, [0, 0.0, 1e30, 0.0])
        d = (en - st) / 1e3
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values())
    out = [{"kernel": k[:110], "calls": a[0], "total_ms": round(a[1] / 1e3, 3), "avg_us": round(a[1] / a[0], 2), "min_us": round(a[2], 2),
            "max_us": round(a[3], 2), "pct": round(100 * a[1] / tot, 2)} for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])]
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
