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
