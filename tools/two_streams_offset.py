# Does it matter how the two concurrent batches are phased?  Stream B's host thread starts `delay` ms after
# stream A's (both then loop freely); frames/s over 6 forwards per stream.
import sys, time, threading, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from skiing_analysis_pytorch_amd import vggt, weights as Wt
from skiing_analysis_pytorch_amd._lib import PREC_BF16, PREC_BF16X3
cfg = Wt.VGGTConfig(enable_track=False)
m = vggt.VGGT(config=cfg, prec=PREC_BF16, head_prec=PREC_BF16X3)
m.load_state_dict(Wt.make_vggt_state_dict(cfg, seed=3, device="cuda"))
want = {"camera", "depth", "point"}
ia = torch.rand(4, 8, 3, 518, 518, device="cuda"); ib = ia.clone()
m(ia, want=want); torch.cuda.synchronize()
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
def run(n, delay_ms):
    def work(x, s, d):
        time.sleep(d * 1e-3)
        with torch.cuda.stream(s):
            for _ in range(n): m(x, want=want)
    ta = threading.Thread(target=work, args=(ia, sa, 0)); tb = threading.Thread(target=work, args=(ib, sb, delay_ms))
    ta.start(); tb.start(); ta.join(); tb.join()
run(1, 0); torch.cuda.synchronize()
for delay in (0, 60, 110, 170, 0, 110):
    torch.cuda.synchronize(); t0 = time.perf_counter(); run(6, delay); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"delay {delay:3d} ms: {dt*1e3:.0f} ms for 2 x 6 forwards = {48/dt:.2f} frames/s (incl. the unoverlapped start)", flush=True)
