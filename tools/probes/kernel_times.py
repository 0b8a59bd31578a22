"""Median per-launch time of the kernels whose name contains a substring, from a rocprofv3 --kernel-trace results .db,
grouped by grid size.  usage: python tools/probes/kernel_times.py <results.db> <substring>"""
import collections, sqlite3, sys
con = sqlite3.connect(sys.argv[1]); cur = con.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kt = [t for t in tabs if "kernel_dispatch" in t][0]; sym = [t for t in tabs if "kernel_symbol" in t][0]
rows = cur.execute("select k.end-k.start, s.kernel_name, k.grid_size_x, k.grid_size_y, k.grid_size_z from %s k join %s s on k.kernel_id = s.id where s.kernel_name like ? order by k.start" % (kt, sym), ("%" + sys.argv[2] + "%",))
agg = collections.defaultdict(list)
for d, n, gx, gy, gz in rows:
    agg[(n[:40], gx, gy, gz)].append(d / 1e3)
for k, v in sorted(agg.items()):
    print("%-40s grid (%d,%d,%d)  n=%d  median %.1f us" % (k[0], k[1], k[2], k[3], len(v), sorted(v)[len(v) // 2]))
