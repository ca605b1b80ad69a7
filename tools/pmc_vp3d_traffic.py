"""profiles/r02_vp3d_traffic.json: L2 <-> fabric bytes per TemporalModel call at B = 1 (the weight-streaming path) from two
separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over tools/prof_vp3d.py 1, corrected as MI355X_MICROARCH.md
prescribes for gfx950 (FETCH_SIZE tallies the 128-byte requests of 16-byte-per-lane loads at 64 bytes: x2; both in KB).
usage: pmc_vp3d_traffic.py <FETCH counter_collection.csv> <WRITE counter_collection.csv> <forwards> <out.json>"""
import csv, collections, hashlib, json, sys
from pathlib import Path
fetch, write, nf, outp = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]


def per_kernel(path, counter):
    by = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter or "vp3d_" not in r["Kernel_Name"]:
            continue
        n = r["Kernel_Name"].replace("void ", "").replace("skimi::", "").split("(")[0]
        by.setdefault(n, []).append(float(r["Counter_Value"]))
    return {k: (sum(v) / nf, len(v) / nf) for k, v in by.items()}   # KB per forward, launches per forward


f, w = per_kernel(fetch, "FETCH_SIZE"), per_kernel(write, "WRITE_SIZE")
rows = []
for k in f:
    rows.append({"kernel": k, "launches_per_call": f[k][1], "FETCH_SIZE_KB": round(f[k][0], 1), "WRITE_SIZE_KB": round(w[k][0], 1),
                 "fabric_bytes": int(2 * f[k][0] * 1024 + w[k][0] * 1024)})
tot = sum(r["fabric_bytes"] for r in rows)
src = Path(__file__).resolve().parent.parent / "skiing_analysis_pytorch_amd" / "csrc" / "vp3d_stream.hip"
out = {"kernel_sha": hashlib.sha256(src.read_bytes()).hexdigest()[:16],
       "model": "TemporalModel RF 27 (filter widths 3,3,3), 1024 channels, bf16x3, B = 1 clip 269 -> 243 frames",
       "command": "rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -- python3 tools/prof_vp3d.py 1 ; the same with --pmc WRITE_SIZE (separate passes, no other trace domains)",
       "correction": "gfx950: FETCH_SIZE x2 for 16-B-per-lane loads (MI355X_MICROARCH.md, HBM); WRITE_SIZE exact",
       "per_kernel": rows,
       "fabric_bytes_per_call": tot,
       "algorithmic_bytes_per_call": 34267276,
       "reading": "every layer's weights cross the fabric once (the workgroups that share a channel tile sit on one XCD); its "
                  "activation records (243-269 rows x 1024 x 4 B = 1 MB, written by the previous launch and still in the "
                  "Infinity Cache) are fetched once per XCD, i.e. 8 times: 12.6 + 8 MB for a dilated conv, 4.2 + 8 (+ residual) "
                  "for a 1x1 conv.  The re-fetches are L2 misses served by the on-package cache, not wasted HBM reads; the "
                  "path is bound by per-launch fixed costs (profiles/r02_vp3d_summary.md), not by this traffic."}
json.dump(out, open(outp, "w"), indent=1)
print(json.dumps(out, indent=1))
