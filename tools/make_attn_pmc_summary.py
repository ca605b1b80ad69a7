"""profiles/r01_attn_mfma_util.json from rocprofv3 --pmc passes over tools/pmc_attn.py (CSV output).
usage: make_attn_pmc_summary.py <out.json> <counter_collection.csv> [<counter_collection.csv> ...]"""
import csv, json, sys
outp, files = sys.argv[1], sys.argv[2:]
out = {"kernel": "attn_q64_kernel",
       "shape": "batch 4 x seq 10992 x 16 heads x 64 (one global-attention launch of the default bench batch)",
       "command": "rocprofv3 --kernel-trace --output-format csv --pmc <counters> -- python tools/pmc_attn.py (one pass per counter group; no other trace domains)",
       "counters": {}}
dur = []
for f in files:
    rs = [r for r in csv.DictReader(open(f)) if "attn_q64" in r["Kernel_Name"]]
    for n in sorted(set(r["Counter_Name"] for r in rs)):
        v = [float(r["Counter_Value"]) for r in rs if r["Counter_Name"] == n]
        out["counters"][n] = sum(v) / len(v)
    dur += [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs]
ns = sum(dur) / len(dur)
c = out["counters"]
n_mfma = 4.0 * 4 * 16 * 10992 * 10992 * 64 / 32768 * (34.0 / 32.0)   # + the 2 reference MFMAs per 64-key tile and wave
out["avg_launch_us_under_pmc"] = ns / 1e3
out["mfma_instructions_per_launch"] = n_mfma
out["note_units"] = ("SQ_VALU_MFMA_BUSY_CYCLES counts MFMA-pipe cycles summed over all SIMDs (= 32 x N_mfma for "
                     "v_mfma_f32_32x32x16_bf16: %.3e expected; MI355X_MICROARCH.md cycle constants); SQ_WAVE_CYCLES / "
                     "SQ_ACTIVE_INST_* / SQ_WAIT_* count quad-cycles" % (32 * n_mfma))
# MFMA utilisation = busy MFMA-pipe cycles / (launch duration x clock x 1024 SIMDs); the shader clock during the
# launch is not in these counters: bracketed by the 2.4 GHz peak clock and a throttled 2.1 GHz
for ghz in (2.1, 2.4):
    out[f"mfma_utilisation_at_{ghz}GHz"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (ns * ghz * 1024)
out["mfma_utilisation_vs_peak_flops"] = (n_mfma * 32768 / (ns * 1e-9)) / 2.5e15
if "SQ_VALU_MFMA_COEXEC_CYCLES" in c:
    out["valu_mfma_coexec_share_of_mfma_busy"] = c["SQ_VALU_MFMA_COEXEC_CYCLES"] / c["SQ_VALU_MFMA_BUSY_CYCLES"]
json.dump(out, open(outp, "w"), indent=1)
print(json.dumps(out, indent=1))
