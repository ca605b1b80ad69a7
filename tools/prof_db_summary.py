"""Per-kernel table from a rocprofv3 rocpd database (--kernel-trace): python tools/prof_db_summary.py <results.db> <steps> [csv_out]"""
import sqlite3, collections, sys, re
db = sqlite3.connect(sys.argv[1]); NS = int(sys.argv[2])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
cols = [r[1] for r in cur.execute(f"pragma table_info({ks})")]
namecol = 'display_name' if 'display_name' in cols else 'kernel_name'
names = dict(cur.execute(f"select id, {namecol} from {ks}"))
byk = collections.defaultdict(lambda: [0, 0]); tot = 0
rows = []
for kid, s, e, gx, wx in cur.execute(f"select kernel_id, start, end, grid_size_x, workgroup_size_x from {kd}"):
    n = names[kid].replace('void ', '').replace('skimi::', '').split('(')[0]
    byk[n][0] += e - s; byk[n][1] += 1; tot += e - s
    rows.append((n, gx, wx, e - s))
print(f"total kernel time {tot/1e6:.1f} ms = {tot/NS/1e6:.1f} ms per step")
print("| kernel | ms / step | % | calls / step | avg us |\n|---|---:|---:|---:|---:|")
for n, v in sorted(byk.items(), key=lambda kv: -kv[1][0])[:26]:
    print(f"| `{n[:90]}` | {v[0]/NS/1e6:.2f} | {100*v[0]/tot:.1f} | {v[1]/NS:.0f} | {v[0]/v[1]/1e3:.1f} |")
if len(sys.argv) > 3:
    import csv
    with open(sys.argv[3], 'w', newline='') as f:
        w = csv.writer(f); w.writerow(["kernel", "grid_x", "wg_x", "calls", "total_ns", "avg_ns"])
        g = collections.defaultdict(lambda: [0, 0])
        for n, gx, wx, d in rows: g[(n, gx, wx)][0] += d; g[(n, gx, wx)][1] += 1
        for k, v in sorted(g.items(), key=lambda kv: -kv[1][0]): w.writerow([k[0], k[1], k[2], v[1], v[0], v[0] // v[1]])
