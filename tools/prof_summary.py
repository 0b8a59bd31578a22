"""Summarise a rocprofv3 --kernel-trace --stats results .db into a per-kernel table (text).
usage: python tools/prof_summary.py <results.db> [out.txt]"""
import sqlite3
import sys


def main():
    con = sqlite3.connect(sys.argv[1])
    cur = con.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kt = [t for t in tabs if "kernel_dispatch" in t][0]
    sym = [t for t in tabs if "kernel_symbol" in t][0]
    q = ("select s.kernel_name, count(*), sum(k.end-k.start)/1e6, avg(k.end-k.start)/1e3, min(k.end-k.start)/1e3, "
         "max(k.end-k.start)/1e3 from %s k join %s s on k.kernel_id = s.id group by s.kernel_name order by 3 desc" % (kt, sym))
    rows = list(cur.execute(q))
    tot = sum(r[2] for r in rows)
    lines = ["rocprofv3 kernel-trace summary: %d kernels, %d launches, %.2f ms total GPU kernel time" % (len(rows), sum(r[1] for r in rows), tot),
             "%-96s %7s %11s %11s %10s %10s %6s" % ("kernel", "calls", "total_ms", "avg_us", "min_us", "max_us", "pct")]
    for r in rows[:40]:
        lines.append("%-96s %7d %11.3f %11.1f %10.1f %10.1f %5.1f%%" % (r[0][:96], r[1], r[2], r[3], r[4], r[5], 100 * r[2] / tot))
    txt = "\n".join(lines)
    print(txt)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(txt + "\n")


if __name__ == "__main__":
    main()
