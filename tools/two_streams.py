# Experiment: one 4-time-step forward vs two concurrent 2-time-step forwards on two HIP streams (two host
# threads, two model instances): do the HBM-bound phases of one stream (GEMM epilogue bursts, LayerNorm,
# upsamples) overlap the MFMA-bound phases of the other?
import sys, time, threading, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from skiing_analysis_pytorch_amd import vggt, weights as Wt
from skiing_analysis_pytorch_amd._lib import PREC_BF16, PREC_BF16X3

cfg = Wt.VGGTConfig(enable_track=False)
sd = Wt.make_vggt_state_dict(cfg, seed=3, device="cuda")
want = {"camera", "depth", "point"}
def make():
    m = vggt.VGGT(config=cfg, prec=PREC_BF16, head_prec=PREC_BF16X3)
    m.load_state_dict(sd)
    return m
m4, ma, mb = make(), make(), make()
img4 = torch.rand(4, 8, 3, 518, 518, device="cuda")
ia, ib = img4[:2].contiguous(), img4[2:].contiguous()
def run_single(n):
    for _ in range(n): m4(img4, want=want)
def run_pair(n):
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    def work(m, x, s):
        with torch.cuda.stream(s):
            for _ in range(n): m(x, want=want)
    ta = threading.Thread(target=work, args=(ma, ia, sa)); tb = threading.Thread(target=work, args=(mb, ib, sb))
    ta.start(); tb.start(); ta.join(); tb.join()
def run_pair4(n):
    global ia, ib
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    def work(m, x, s):
        with torch.cuda.stream(s):
            for _ in range(n): m(x, want=want)
    ta = threading.Thread(target=work, args=(ma, img4, sa)); tb = threading.Thread(target=work, args=(mb, img4b, sb))
    ta.start(); tb.start(); ta.join(); tb.join()
img4b = img4.clone()
for name, fn, steps in (("single B=4", run_single, 4), ("two streams 2 x B=2", run_pair, 4), ("two streams 2 x B=4", run_pair4, 8),
                        ("single B=4", run_single, 4), ("two streams 2 x B=2", run_pair, 4), ("two streams 2 x B=4", run_pair4, 8)):
    fn(1); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(4); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 4
    print(f"{name}: {dt*1e3:.1f} ms per {steps} time steps = {steps/dt:.2f} frames/s", flush=True)
