import sqlite3, sys, collections
con = sqlite3.connect(sys.argv[1]); cur = con.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kt = [t for t in tabs if "kernel_dispatch" in t][0]; sym = [t for t in tabs if "kernel_symbol" in t][0]
rows = list(cur.execute("select k.end-k.start, s.kernel_name, k.grid_size_x from %s k join %s s on k.kernel_id = s.id where s.kernel_name like '%%cross_attn_bwd%%' order by k.start" % (kt, sym)))
agg = collections.defaultdict(list)
for d, n, gx in rows:
    agg[(n[3:26], "split" if gx == 256 else "unsplit")].append(d / 1e3)
for k, v in sorted(agg.items()):
    print("%-26s %-8s n=%d  median %.1f us" % (k[0], k[1], len(v), sorted(v)[len(v) // 2]))
