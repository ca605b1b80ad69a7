"""In-tree build of libskimi.so (HIP, gfx950 only) and of the oracle's C pieces.

`python -m skiing_analysis_pytorch_amd.build` or `__graft_entry__.build()`.
hipcc cross-compiles gfx950 without a GPU; objects are cached per source under
csrc/_obj and relinked only when a source or header changed.
"""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
OBJ = CSRC / "_obj"
LIB = PKG / "lib" / "libskimi.so"
ARCH = "gfx950"

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
CXXFLAGS = [
    f"--offload-arch={ARCH}",
    "-O3",
    "-std=c++17",
    "-fPIC",
    "-fno-gpu-rdc",
    "-Wall",
    "-Wno-unused-function",
    "-ffp-contract=off",  # keep a*b+c unfused on the host side and in epilogues: parity with torch CPU
]
# Timing ablations of the hot kernels (SKIMI_ATTN_ABL / SKIMI_GEMM256_ABL / SKIMI_X3_ABL: variants with parts of
# a kernel removed -- their results are WRONG) are compiled only into a profiling build: SKIMI_ABLATIONS=1.
if os.environ.get("SKIMI_ABLATIONS") == "1":
    CXXFLAGS.append("-DSKIMI_ABLATIONS")


# Per-file extras.  The attention kernels must not get packed-fp32 VALU ops (v_pk_fma_f32,
# v_pk_add_f32, v_pk_mul_f32 -- what the SLP vectorizer makes of adjacent scalar float ops): on
# gfx950 VOP3P instructions do not co-issue with an MFMA in flight (tools/coissue.hip: +5 cycles
# each on top of the MFMA, where v_fma_f32 / v_add_f32 / v_exp_f32 / v_max3_f32 hide completely),
# and the softmax VALU stream is what the attention MFMAs have to overlap with.
EXTRA_FLAGS = {
    "attention_bf16.hip": ["-fno-slp-vectorize"],
    "attention_q64.hip": ["-fno-slp-vectorize"],
}


def _digest(paths) -> str:
    h = hashlib.sha256()
    for p in sorted(paths):
        h.update(str(p).encode())
        h.update(Path(p).read_bytes())
    h.update(" ".join(CXXFLAGS).encode())
    return h.hexdigest()[:16]


def sources():
    return sorted(CSRC.glob("*.hip"))


def headers():
    return sorted(CSRC.glob("*.h")) + sorted((PKG.parent / "include").glob("*.h"))


def build(verbose: bool = False, jobs: int = 4) -> Path:
    OBJ.mkdir(parents=True, exist_ok=True)
    LIB.parent.mkdir(parents=True, exist_ok=True)
    hdr_digest = _digest(headers())
    objs, todo = [], []
    for src in sources():
        tag = _digest([src]) + "_" + hdr_digest
        if src.name in EXTRA_FLAGS:
            tag += "_" + hashlib.sha256(" ".join(EXTRA_FLAGS[src.name]).encode()).hexdigest()[:6]
        obj = OBJ / f"{src.stem}.{tag}.o"
        objs.append(obj)
        if not obj.exists():
            for stale in OBJ.glob(f"{src.stem}.*.o"):
                stale.unlink()
            todo.append((src, obj))

    def compile_one(job):
        src, obj = job
        cmd = [HIPCC, *CXXFLAGS, *EXTRA_FLAGS.get(src.name, []), "-c", str(src), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src.name}:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)

    if todo:
        with ThreadPoolExecutor(max_workers=jobs) as ex:
            list(ex.map(compile_one, todo))
    if todo or not LIB.exists():
        cmd = [HIPCC, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", str(LIB), *map(str, objs)]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    print(build(verbose=True))
