"""Per-kernel summary (calls, total, avg / min / max duration) of a rocprofv3 --kernel-trace run written in the rocpd
SQLite format: `python scripts/rocpd_stats.py <results.db> [out.csv]`."""
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    sym = [t for t in tabs if "info_kernel_symbol" in t][0]
    rows = db.execute(f"select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), "
                      f"max(d.end-d.start) from {kd} d join {sym} s on d.kernel_id=s.id group by s.kernel_name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    out = ["name,calls,total_ms,avg_us,min_us,max_us,pct"]
    for n, c, t, a, mn, mx in rows:
        out.append(f'"{n[:120]}",{c},{t / 1e6:.3f},{a / 1e3:.1f},{mn / 1e3:.1f},{mx / 1e3:.1f},{100 * t / tot:.2f}')
    text = "\n".join(out) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text)
    print("\n".join(out[:16]))


if __name__ == "__main__":
    main()
