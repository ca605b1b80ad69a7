#!/bin/bash
# frames/s of the bench step against (time steps per call, concurrent streams); one line per setting
for cfg in "4 2" "4 3" "2 4" "8 1" "8 2" "6 2" "4 1" "3 3"; do
  set -- $cfg
  python bench.py --batch $1 --streams $2 --steps 3 --warmup 1 --no-cpu-baseline --no-vp3d --no-fp8 --no-parity-mode 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('batch $1 streams $2:', round(d['value'],2), 'frames/s', round(d['ms_per_step'],1), 'ms/step, attn frac', round(d['roofline']['frac'],3))"
done
