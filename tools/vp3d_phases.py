"""Phase timeline of the VP3D streaming kernel (SKIMI_ABLATIONS=1 build only): s_memtime stamps of wave 0 of every
workgroup of the LAST vp3d_mm launch of a forward; run with SKIMI_VP3D_LAST=<layer index 0..4> to stop the chain
after that layer (0 = block-1 dilated conv, 1 = its 1x1, 2, 3, 4 = shrink)."""
import ctypes as C, os, sys, numpy as np, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from skiing_analysis_pytorch_amd import vp3d, weights as W, _lib
from skiing_analysis_pytorch_amd._lib import PREC_BF16X3
m = vp3d.TemporalModel(17, 2, 17, [3, 3, 3], prec=PREC_BF16X3)
m.load_state_dict(W.make_vp3d_state_dict(seed=0, filter_widths=[3, 3, 3]))
x = torch.randn(1, 269, 17, 2, device="cuda")
out = torch.empty(1, 243, 17, 3, device="cuda")
h = C.CDLL(str(_lib.LIB_PATH))
for last in range(5):
    os.environ["SKIMI_VP3D_LAST"] = str(last)
    for _ in range(5):
        m(x, out=out)
    torch.cuda.synchronize()
    buf = (C.c_longlong * (1024 * 8))()
    assert h.skimi_debug_vp3d_ts(buf, 1024 * 8) == 0
    t = np.array(buf, dtype=np.int64).reshape(1024, 8)[:256]
    # s_memtime counts shader cycles, per XCD (the eight counters are not synchronised): deltas inside a workgroup only
    d = np.diff(t[:, :7], axis=1).astype(np.float64)
    names = ["prologue (kernargs, addresses) -> first loads issued", "main loop", "partials to LDS + barrier", "reduce", "epilogue math + stores issued", "stores drained"]
    tot = (t[:, 6] - t[:, 0]).astype(np.float64)
    print(f"layer {last}: total cycles med {np.median(tot):.0f} | " + " | ".join(f"{n}: {np.median(d[:, i]):.0f}" for i, n in enumerate(names)))
