"""Summarise a rocprofv3 --kernel-trace csv: per (kernel, grid) time per step.  usage: prof_summary.py trace.csv nsteps"""
import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
nsteps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
g = collections.defaultdict(lambda: [0, 0])
tot = 0
for r in rows:
    n = r['Kernel_Name']
    n = n.replace('void ', '').replace('skimi::', '')
    n = n.split('(')[0][:60]
    key = (n, r['Grid_Size_X'], r['Workgroup_Size_X'])
    d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    g[key][0] += d; g[key][1] += 1; tot += d
print(f"total {tot/nsteps/1e6:.2f} ms/step")
for k, v in sorted(g.items(), key=lambda kv: -kv[1][0])[:int(sys.argv[3]) if len(sys.argv) > 3 else 28]:
    print(f"{v[0]/nsteps/1e6:8.2f} ms/step {v[1]/nsteps:7.1f} calls  avg {v[0]/v[1]/1e3:9.1f} us  {k[0]} grid={k[1]} wg={k[2]}")
