"""GPU idle gaps in a rocprofv3 --kernel-trace results .db: the idle time before each kernel, grouped by the kernel that
FOLLOWS the gap (what the device was waiting for) -- shows where host work / synchronisation stalls the queue.
usage: python tools/prof_gaps.py <results.db> [min_gap_us] [marker_substring skip_count]
With a marker, only the trace from the (skip_count+1)-th launch of a kernel whose name contains the marker is analysed
(e.g. the timed region of a bench run that launches one such kernel per step)."""
import collections
import sqlite3
import sys


def main():
    con = sqlite3.connect(sys.argv[1])
    cur = con.cursor()
    thr = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kt = [t for t in tabs if "kernel_dispatch" in t][0]
    sym = [t for t in tabs if "kernel_symbol" in t][0]
    rows = list(cur.execute("select k.start, k.end, s.kernel_name from %s k join %s s on k.kernel_id = s.id order by k.start" % (kt, sym)))
    if len(sys.argv) > 4:
        marks = [i for i, r in enumerate(rows) if sys.argv[3] in r[2]]
        rows = rows[marks[int(sys.argv[4])]:]
    busy = sum(e - s for s, e, _ in rows) / 1e6
    span = (max(e for _, e, _ in rows) - rows[0][0]) / 1e6
    print("kernels %d, busy %.1f ms, span %.1f ms" % (len(rows), busy, span))
    gaps, small, cnt = collections.Counter(), 0.0, collections.Counter()
    last_end, prev = rows[0][1], rows[0][2]
    for s, e, n in rows[1:]:
        g = s - last_end
        if g > 0:
            if g / 1e3 >= thr:
                gaps[(prev[:60], n[:60])] += g / 1e6
                cnt[(prev[:60], n[:60])] += 1
            else:
                small += g / 1e6
        if e > last_end:
            last_end, prev = e, n
    print("gaps < %.0f us: %.1f ms total; larger gaps by (kernel before -> kernel after):" % (thr, small))
    for k, v in gaps.most_common(25):
        print("%9.2f ms %5d x  %s  ->  %s" % (v, cnt[k], k[0], k[1]))


if __name__ == "__main__":
    main()
