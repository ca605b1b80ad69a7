"""Per-kernel totals of a rocprofv3 kernel-trace CSV.  usage: trace_top.py <kernel_trace.csv> <forwards in trace> [rows]"""
import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
nf = int(sys.argv[2]); top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
by = collections.defaultdict(lambda: [0, 0])
for r in rows:
    n = r['Kernel_Name'].replace('void ', '').replace('skimi::', '').split('(')[0]
    d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    by[n][0] += d; by[n][1] += 1
tot = sum(v[0] for v in by.values())
print(f"total kernel time {tot/1e6:.1f} ms = {tot/nf/1e6:.1f} ms per forward")
for n, v in sorted(by.items(), key=lambda kv: -kv[1][0])[:top]:
    print(f"{n[:72]:72s} {v[0]/nf/1e6:8.2f} ms/fwd {100*v[0]/tot:5.1f}% {v[1]/nf:7.1f} calls {v[0]/v[1]/1e3:9.1f} us")
