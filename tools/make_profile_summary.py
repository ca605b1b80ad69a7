"""profiles/rNN_summary.md from a rocprofv3 kernel trace of bench.py and the bench line.
usage: make_profile_summary.py <kernel_trace.csv> <bench_line.json> <batch forwards in trace> <out.md> [round tag, e.g. r03]"""
import csv, json, collections, sys
trace, linef, NS, outp = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
TAG = sys.argv[5] if len(sys.argv) > 5 else "r03"
def load_rows(path):
    """[(kernel name, grid_x, workgroup_x, duration ns)] from a rocprofv3 kernel trace: the CSV of
    --output-format csv, or the rocpd sqlite database that rocprofv3 writes by default."""
    if path.endswith(".db"):
        import sqlite3
        cur = sqlite3.connect(path).cursor()
        tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
        kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
        ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
        cols = [r[1] for r in cur.execute(f"pragma table_info({ks})")]
        names = dict(cur.execute(f"select id, {'display_name' if 'display_name' in cols else 'kernel_name'} from {ks}"))
        return [(names[k], str(gx), str(wx), e - s0)
                for k, s0, e, gx, wx in cur.execute(f"select kernel_id, start, end, grid_size_x, workgroup_size_x from {kd}")]
    return [(r['Kernel_Name'], r['Grid_Size_X'], r['Workgroup_Size_X'], int(r['End_Timestamp']) - int(r['Start_Timestamp']))
            for r in csv.DictReader(open(path))]


g = collections.defaultdict(lambda: [0, 0]); byk = collections.defaultdict(lambda: [0, 0]); tot = 0
for name, gx, wx, d in load_rows(trace):
    n = name.replace('void ', '').replace('skimi::', '').split('(')[0]
    k = (n, gx, wx)
    g[k][0] += d; g[k][1] += 1; byk[n][0] += d; byk[n][1] += 1; tot += d
line = json.load(open(linef))
out = []
out.append(f"# Round {int(TAG[1:])} — rocprofv3 --kernel-trace --stats of `python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-vp3d --no-fp8 --no-parity-mode --no-other-prec` (timed mode: {line['dtype']})\n")
out.append(f"MI355X (gfx950); {NS} batch forwards in the trace (1 one-stream preparation, then 1 warm-up and 2 timed steps of 2 concurrent batches each, `--streams 2`); one batch = 4 time steps x 8 views x 518x518 (`--batch 4`), all four heads (17 query points per step).")
out.append(f"Total kernel time {tot/1e6:.1f} ms = {tot/NS/1e6:.1f} ms per batch (summed kernel durations: with two batches in flight kernels of the two streams share the chip, so this is more than the wall time; includes the one-off weight upload / synthetic-weight kernels of the first forward).\n")
out.append("| kernel | ms / batch | % | calls / batch | avg us |\n|---|---:|---:|---:|---:|")
for n, v in sorted(byk.items(), key=lambda kv: -kv[1][0])[:24]:
    out.append(f"| `{n[:90]}` | {v[0]/NS/1e6:.2f} | {100*v[0]/tot:.1f} | {v[1]/NS:.0f} | {v[0]/v[1]/1e3:.1f} |")
rf = line['roofline']
out.append("\nUn-profiled bench line of the same build (`profiles/" + TAG + "_bench_line.json`, full default run incl. the CPU legs):")
out.append(f"- value {line['value']:.2f} frames/s, {line['ms_per_step']:.1f} ms/step (one step = {line['config'].get('streams', 1)} concurrent batches of 4 time steps), whole path {line['whole_path_tflops']:.0f} TFLOP/s")
one = rf.get('one_stream')
if one:
    out.append(f"- the same kernel with the chip to itself (one batch, one stream; measured by bench.py right after the timed region): {one['achieved']:.0f} TFLOP/s = {100*one['frac']:.1f}% (avg launch {one['avg_launch_us']:.0f} us over {one['launches']} launches)")
out.append(f"- roofline: {rf['kernel']}: {rf['achieved']:.0f} TFLOP/s = {100*rf['frac']:.1f}% of {rf['peak']:.0f} (avg launch {rf['avg_launch_us']:.0f} us over {rf['launches']} launches, HIP events on the launch stream inside bench.py); HBM traffic {(rf['traffic'] or 0)/1e6:.0f} MB per launch (`profiles/{TAG}_attn_traffic.json`: FETCH_SIZE x2 + WRITE_SIZE, separate PMC passes) against {4*10992*1024*2*4/1e6:.0f} MB algorithmic (q, k, v read once, o written once).  Run-to-run: fresh boxes differ by several % (clocks, thermal state).")
cb = line['cpu_baseline']
out.append(f"- cpu_baseline: {cb['value']:.4f} frames/s on {cb['cores']} cores ({cb['sample']})")
vp = line.get('vp3d')
if vp:
    out.append(f"- vp3d leg: {vp['clips_1']['us_per_call']:.0f} us per 243-frame clip at B=1 ({vp['clips_1']['achieved_GBps']:.0f} GB/s on SURVEY §8(d)'s algorithmic bytes = {100*vp['clips_1']['frac_of_hbm_peak']:.1f} % of 8 TB/s), {vp['clips_64']['us_per_call']/64:.1f} us per clip at B=64; CPU oracle {vp['cpu_oracle']['s_per_clip_with_flip_tta']*1e3:.1f} ms per clip on {vp['cpu_oracle']['cores']} cores (B = 2, the flip-TTA call: {vp.get('clips_2', {}).get('us_per_call', float('nan')):.0f} us per call)")
mp = line.get('mpjpe_vs_cpu_oracle')
if mp:
    out.append(f"- within_bar (top level, the timed mode on the ring rig): {line.get('within_bar')}; value_within_bar {line.get('value_within_bar')} ({line.get('mode_within_bar')})")
    for k in ("f16_mode", "bf16_mode", "fp8_mode", "bf16x3_parity_mode"):
        if k in mp:
            r = mp[k]
            out.append(f"- joints vs the CPU oracle, {k}: pose_enc max abs err {r['pose_enc_max_abs_err']:.2e}, MPJPE ring rig {r['mpjpe_ring_rig']:.2e} (within 1e-3: {r['within_bar']}), native scene {r['mpjpe_native_scene']:.2e}, depth rel err median {r['depth_rel_err_median']:.1e}")
for k, label in (("bf16_mode", "bf16 mode"), ("f16_mode", "f16 mode"), ("parity_mode", "bf16x3 everywhere"), ("fp8", "fp8 mode")):
    if k in line:
        out.append(f"- {label}: {line[k]['value']:.2f} frames/s ({line[k]['ms_per_step']:.0f} ms per step, {line[k]['steps']} steps)")
out.append("\nPer-shape split of the attention kernel from the same trace (grouped by grid size):\n")
out.append("| launches / batch | grid (threads) | shape | avg us |\n|---:|---:|---|---:|")
ga = None
for k, v in sorted(g.items(), key=lambda kv: -kv[1][0]):
    if 'attn' in k[0]:
        wg = int(k[1]) // int(k[2])
        if 'f32' in k[0]:
            shape = ("`attn_f32_kernel<64>`: the tracker's time / space attention (48-wide heads, fp32 kernel; grid x only: the launches are 3-D, q blocks x heads x sequences)"
                     if '<64>' in k[0] else
                     "`attn_f32_kernel<128>`: camera-head trunk attention over the 8 camera tokens (fp32 kernel)")
        elif wg == 2752:
            shape = "global attention: batch 4, seq 8 x 1374 = 10992, 16 heads x 64"; ga = v
        else:
            shape = "frame / DINOv2 attention: batch 32, seq 1374, 16 heads x 64"
        out.append(f"| {v[1]/NS:.0f} | {k[1]} = {wg} WG x {k[2]} | {shape} | {v[0]/v[1]/1e3:.1f} |")
if ga:
    out.append(f"\nThe {ga[0]/ga[1]/1e3:.0f} us of the global-attention launches under rocprofv3 against the {rf['avg_launch_us']:.0f} us that bench.py measures un-profiled (a different box; profiled passes also clock a few % lower, MI355X_MICROARCH.md cycle-constants note 2); {rf['flops_per_launch']/1e9:.1f} GFLOP per launch -> {rf['flops_per_launch']/(ga[0]/ga[1])/1e3:.0f} (profiled) / {rf['achieved']:.0f} (un-profiled) TFLOP/s.")
out.append("\nTrack head rows (at the aggregator's 16-bit precision since round 3): `gemm_kernel<64, 64, 64, 1, float, unsigned short, 4, true>` + `layernorm_kernel<8>` + `attn_f32_kernel<64>` are the tracker's small launches (4 iterations x 6 rounds of time / space attention over the whole batch; fp32 activations rounded to fp16 while staged), `conv_win128_kernel<F16>` the 3 x 3 convolutions of its DPT feature extractor (halo-window kernel, round 3; the small maps stay on `gemm_kernel<128, 128, 64, 1, unsigned short, unsigned short, 1, true>`).")
out.append("\nGEMM shapes behind the `gemm256*` rows (M = 4 x 10992 = 43968 token rows; template argument = epilogue): `gemm256w4_kernel<1, F16>` qkv 1024->3072 (bias -> bf16 rows for the attention) and `gemm256w4_kernel<2, F16>` fc2 4096->1024 (bias, LayerScale, fp32 residual) on the single-stream 4-wave loop; `gemm256pp_kernel<3, 1, F16>` fc1 1024->4096 (bias, GELU -> fp16) and `gemm256pp_kernel<2, 1, F16>` proj 1024->1024 (residual epilogue) on the 8-wave ping-pong loop (F16 = true: fp16 operands on v_mfma_f32_32x32x16_f16).  `gemm_x3w4_kernel<a_mode>` / `gemm_x3w4n_kernel` (256- / 128-column tiles), `gemm_kernel<...,3,float,float>` and `conv_direct_n32_kernel` are the fp32-accurate (bf16x3) convolutions of the depth and point DPT heads; `bilinear_ac_planes_kernel` is the upsample that writes their bf16 hi|lo operands directly.")
open(outp, 'w').write("\n".join(out) + "\n")
print("\n".join(out)[:2500])
