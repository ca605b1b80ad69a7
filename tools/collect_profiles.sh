#!/bin/bash
# rocprofv3 passes behind profiles/rNN_* (run on the GPU box from the repository root; raw output under gpurun_out/<tag>/):
#   tools/collect_profiles.sh r03
# Kernel trace + stats of the bench step, of the VideoPose3D chain, and SEPARATE --pmc passes (no other trace domains) for the
# global-attention launch (HBM traffic, MFMA busy / co-execution / LDS conflicts) and the B = 1 lifter (fabric traffic).
set -e
tag=${1:-r03}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $out/bench -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-vp3d --no-fp8 --no-parity-mode --no-other-prec > $out/bench.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/vp3d -- python3 tools/prof_vp3d.py 1 2 64 > $out/vp3d.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --output-format csv --pmc $c -d $out/attn_$c -- python3 tools/pmc_attn.py > $out/attn_$c.log 2>&1
  rocprofv3 --kernel-trace --output-format csv --pmc $c -d $out/vp3d_$c -- python3 tools/prof_vp3d.py 1 > $out/vp3d_$c.log 2>&1
done
rocprofv3 --kernel-trace --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES -d $out/attn_mfma -- python3 tools/pmc_attn.py > $out/attn_mfma.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_BANK_CONFLICT -d $out/attn_coexec -- python3 tools/pmc_attn.py > $out/attn_coexec.log 2>&1
find $out -name "*.csv" | sort
