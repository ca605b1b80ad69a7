"""CPU oracle of the hot path — TEST INFRASTRUCTURE ONLY.

Plain-PyTorch (CPU, fp32) functional restatements of the reference's algorithms, each
function citing the reference file:line it follows.  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import this package; the product
(skiing_analysis_pytorch_amd) never does.  The restatements are pinned against outputs of
the reference itself (tools/make_goldens.py imports /root/reference in the build container
and writes tests/golden/*.npz; tests/test_oracle_golden.py checks the oracle against them).
"""
