"""CPU: libskimi.so loads and exports every symbol include/skimi.h declares (no compute)."""
import ctypes
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def _declared_symbols():
    text = (ROOT / "include" / "skimi.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(skimi_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported():
    from skiing_analysis_pytorch_amd import _lib

    names = _declared_symbols()
    assert "skimi_gemm" in names and "skimi_vp3d_forward" in names
    handle = ctypes.CDLL(str(_lib.LIB_PATH))
    missing = [n for n in names if not hasattr(handle, n)]
    assert not missing, f"declared in skimi.h but not exported: {missing}"
    # and the ctypes table binds exactly the declared set
    assert sorted(_lib.exported_symbols()) == names


def test_version_and_error_channel():
    from skiing_analysis_pytorch_amd import _lib

    lib = _lib.lib()
    assert lib.skimi_version() >= 100
    # bad arguments are reported through the return code + skimi_last_error, never a crash
    assert lib.skimi_gemm(None, None) != 0
    assert b"null" in lib.skimi_last_error()
    assert lib.skimi_vp3d_create(17, 2, 17, (ctypes.c_int32 * 2)(3, 4), 2, 1024, 0) is None
    assert b"odd filter widths" in lib.skimi_last_error()


def test_gemm_desc_layout_matches_header():
    """sizeof(GemmDesc) in ctypes must equal the C struct (compiled probe)."""
    import subprocess
    import tempfile

    from skiing_analysis_pytorch_amd._lib import GemmDesc

    with tempfile.TemporaryDirectory() as td:
        src = Path(td) / "p.c"
        src.write_text('#include "skimi.h"\n#include <stdio.h>\nint main(){printf("%zu %zu %zu",sizeof(skimi_gemm_desc),'
                       '__builtin_offsetof(skimi_gemm_desc,out),__builtin_offsetof(skimi_gemm_desc,force_splitk));return 0;}')
        exe = Path(td) / "p"
        subprocess.run(["gcc", "-I", str(ROOT / "include"), str(src), "-o", str(exe)], check=True)
        size, off_out, off_fs = map(int, subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split())
    assert ctypes.sizeof(GemmDesc) == size
    assert GemmDesc.out.offset == off_out
    assert GemmDesc.force_splitk.offset == off_fs


def test_product_does_not_import_oracle():
    pkg = ROOT / "skiing_analysis_pytorch_amd"
    for f in pkg.rglob("*.py"):
        txt = f.read_text()
        assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f"{f} imports the oracle"


def test_cpu_tensor_rejected():
    import torch

    from skiing_analysis_pytorch_amd import _lib, vp3d, weights

    m = vp3d.TemporalModel(17, 2, 17, [3, 3, 3])
    assert m.receptive_field() == 27
    with pytest.raises(_lib.SkimiError):
        m(torch.zeros(1, 27, 17, 2))


@pytest.mark.parametrize("src,kernel", [("attention_q64.hip", "attn_q64_kernel"), ("attention_bf16.hip", "attn_bf16_kernel")])
def test_lds_dma_barriers_wait_for_vmcnt0(src, kernel):
    """The LDS-DMA double buffer of the attention kernels is only correct if every wave has waited
    `vmcnt(0)` on its own global_load_lds before it enters the workgroup barrier (a gfx950 barrier does not
    wait for memory counters by itself).  The source writes the wait out; this checks the EMITTED ISA of
    the build's own flags: every s_barrier of the kernel is directly preceded by an s_waitcnt vmcnt(0)
    with no LDS-DMA issued in between."""
    import shutil
    import subprocess

    from skiing_analysis_pytorch_amd import build as b

    if shutil.which(b.HIPCC) is None and not Path(b.HIPCC).exists():
        pytest.skip("hipcc not available")
    flags = [f for f in b.CXXFLAGS if f != "-fPIC"] + b.EXTRA_FLAGS.get(src, [])
    r = subprocess.run([b.HIPCC, *flags, "-S", "--offload-device-only", str(b.CSRC / src), "-o", "-"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    body, inside = [], False
    for ln in r.stdout.splitlines():
        if re.match(rf"^_ZN5skimi\d+{kernel}ILi0EE.*:", ln):
            inside = True
            continue
        if inside:
            if "s_endpgm" in ln:
                break
            ins = ln.strip()
            if ins and not ins.startswith((";", ".")):
                body.append(ins)
    barriers = [i for i, ins in enumerate(body) if ins.startswith("s_barrier")]
    assert len(barriers) >= 2, "expected the prologue barrier and the loop barrier"
    assert any("global_load_lds" in ins for ins in body)
    for i in barriers:
        j = i - 1
        while j >= 0 and not body[j].startswith("s_waitcnt"):
            assert "global_load_lds" not in body[j], "LDS-DMA issued between the last wait and the barrier"
            assert i - j <= 4, f"no s_waitcnt directly ahead of the s_barrier: {body[max(0, i - 5):i + 1]}"
            j -= 1
        assert j >= 0 and "vmcnt(0)" in body[j], body[max(0, i - 5):i + 1]
