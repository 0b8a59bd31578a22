"""Kernels around every launch whose name contains a marker, from a rocprofv3 --kernel-trace results .db: start offset, duration and the
idle gap before each kernel (us).  usage: python tools/prof_window.py <results.db> <marker> [n_before] [n_after] [max_windows]"""
import sqlite3
import sys


def main():
    con = sqlite3.connect(sys.argv[1])
    cur = con.cursor()
    marker = sys.argv[2]
    nb = int(sys.argv[3]) if len(sys.argv) > 3 else 6
    na = int(sys.argv[4]) if len(sys.argv) > 4 else 4
    mw = int(sys.argv[5]) if len(sys.argv) > 5 else 6
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kt = [t for t in tabs if "kernel_dispatch" in t][0]
    sym = [t for t in tabs if "kernel_symbol" in t][0]
    rows = list(cur.execute("select k.start, k.end, s.kernel_name from %s k join %s s on k.kernel_id = s.id order by k.start" % (kt, sym)))
    hits = [i for i, r in enumerate(rows) if marker in r[2]]
    for h in hits[-mw:]:
        print("---- window around launch %d (%s)" % (h, rows[h][2][:60]))
        t0 = rows[h][0]
        for i in range(max(1, h - nb), min(len(rows), h + na + 1)):
            s, e, n = rows[i]
            print("%10.1f us  dur %8.1f  gap %8.1f  %s%s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - rows[i - 1][1]) / 1e3, "* " if i == h else "  ", n[:90]))


if __name__ == "__main__":
    main()
