"""profiles/r02_attn_traffic.json: HBM bytes per global-attention launch from the separate rocprofv3 --pmc passes over
tools/pmc_attn.py (FETCH_SIZE, WRITE_SIZE; counter_collection.csv of `--output-format csv`), corrected as
MI355X_MICROARCH.md prescribes for gfx950 (FETCH_SIZE counts the 128-byte requests of 16-byte-per-lane loads at 64 bytes:
x2; WRITE_SIZE is exact; both in KB), with the sha256 of the kernel source so that bench.py only quotes the figure for
the kernel it was measured on.
usage: pmc_traffic.py <FETCH_SIZE csv> <WRITE_SIZE csv> <out.json>"""
import csv, hashlib, json, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
fetch, write, outp = sys.argv[1:4]


def avg(path, counter):
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(path)) if "attn_q64" in r["Kernel_Name"] and r["Counter_Name"] == counter]
    assert v, (path, counter)
    return sum(v) / len(v)


f_kb, w_kb = avg(fetch, "FETCH_SIZE"), avg(write, "WRITE_SIZE")
src = ROOT / "skiing_analysis_pytorch_amd" / "csrc" / "attention_q64.hip"
out = {"kernel": "attn_q64_kernel",
       "kernel_sha": hashlib.sha256(src.read_bytes()).hexdigest()[:16],
       "shape": "batch 4 x seq 10992 x 16 heads x 64 (one global-attention launch of the default bench step)",
       "command": "rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -- python3 tools/pmc_attn.py ; the same with --pmc WRITE_SIZE (separate passes, no other trace domains)",
       "FETCH_SIZE_KB_per_launch": f_kb, "WRITE_SIZE_KB_per_launch": w_kb,
       "correction": "gfx950: FETCH_SIZE tallies 128-B requests at 64 B for 16-B-per-lane loads -> x2 (MI355X_MICROARCH.md, HBM); WRITE_SIZE exact",
       "hbm_bytes_per_launch": int(2 * f_kb * 1024 + w_kb * 1024),
       "algorithmic_bytes_per_launch": 4 * 10992 * 1024 * 2 * 4,
       "time_steps": 4}
json.dump(out, open(outp, "w"), indent=1)
print(json.dumps(out, indent=1))
