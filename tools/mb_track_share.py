"""What the track head costs inside the bench step (2 concurrent batches of 4 time steps x 8 views x 518 x 518, PREC_F16):
frames/s with the full head (4 tracker iterations), with 1 iteration (the slope = one iteration of the tracker; what is
left at 0 iterations = its DPT feature extractor + correlation pyramid) and without the head.  One process, interleaved
rounds (boxes differ by several percent)."""
import sys, threading, time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from skiing_analysis_pytorch_amd import vggt, weights as W
from skiing_analysis_pytorch_amd._lib import PREC_BF16X3, PREC_F16
from skiing_analysis_pytorch_amd.infer import _side_streams

dev = torch.device("cuda", 0)
B, S, IMG, NS = 4, 8, 518, 2
g = torch.Generator(device=dev).manual_seed(1234)
batches = [(torch.rand((B, S, 3, IMG, IMG), generator=g, device=dev),
            torch.rand((B, 17, 2), generator=g, device=dev) * (IMG - 40) + 20) for _ in range(NS)]
variants = {}
for name, kw in (("iters4", dict(track_iters=4)), ("iters1", dict(track_iters=1)), ("notrack", None)):
    cfg = W.VGGTConfig(**(kw or {}))
    m = vggt.VGGT(config=cfg, prec=PREC_F16, head_prec=PREC_BF16X3)
    m.load_state_dict(W.make_vggt_state_dict(cfg, seed=0, device=dev))
    variants[name] = (m, kw is not None)
torch.cuda.empty_cache()


def step(name):
    m, track = variants[name]
    want = {"camera", "depth", "point"} | ({"track"} if track else set())
    main = torch.cuda.current_stream(dev)
    side = _side_streams(dev, NS)

    def worker(k):
        with torch.cuda.device(dev), torch.cuda.stream(side[k]):
            m(batches[k][0], query_points=batches[k][1] if track else None, want=want)
    for s_ in side:
        s_.wait_stream(main)
    th = [threading.Thread(target=worker, args=(k,)) for k in range(NS)]
    for t_ in th:
        t_.start()
    for t_ in th:
        t_.join()
    for s_ in side:
        main.wait_stream(s_)


for name in variants:
    step(name); step(name)
torch.cuda.synchronize()
acc = {n: [] for n in variants}
for rnd in range(3):
    for name in variants:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            step(name)
        torch.cuda.synchronize()
        acc[name].append((time.perf_counter() - t0) / 3)
    print(" | ".join(f"{n} {acc[n][-1] * 1e3:.1f} ms" for n in variants), flush=True)
ms = {n: 1e3 * min(v) for n, v in acc.items()}
it = (ms["iters4"] - ms["iters1"]) / 3
print(f"per step of {NS * B} time steps: full {ms['iters4']:.1f} ms, no track head {ms['notrack']:.1f} ms -> head {ms['iters4'] - ms['notrack']:.1f} ms = "
      f"{(ms['iters4'] - ms['notrack']) / (NS * B):.2f} ms per time step; one tracker iteration {it:.1f} ms per step "
      f"({it / (NS * B):.3f} ms per time step), extractor + pyramid + first-iteration setup {ms['iters1'] - it - ms['notrack']:.1f} ms")
