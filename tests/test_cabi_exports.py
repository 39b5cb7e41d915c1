"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol include/lvf.h declares,
and refuses to run without a GPU (no silent CPU fallback)."""
import ctypes as C
import os

import pytest

from lvio_fusion_amd import _lib


@pytest.fixture(scope="module")
def so():
    _lib.build()
    return C.CDLL(_lib.SO_PATH)


def test_every_declared_symbol_is_exported(so):
    names = _lib.declared_symbols()
    assert len(names) >= 40
    missing = [n for n in names if not hasattr(so, n)]
    assert not missing, f"declared in include/lvf.h but not exported: {missing}"
    # and the ctypes signature table covers the header
    assert sorted(_lib._SIGS) == names


def test_struct_layouts_match_header():
    assert C.sizeof(_lib.Camera) == 11 * 8
    assert C.sizeof(_lib.SolverOptions) == 8 * 8
    assert C.sizeof(_lib.IcpOptions) == 40


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from lvio_fusion_amd import api
    with pytest.raises(api.LvfError) as e:
        api.Context(0)
    assert "no usable HIP device" in str(e.value) or "HIP error" in str(e.value)


def test_product_path_does_not_touch_oracle():
    """Nothing under lvio_fusion_amd/ or include/ may import, include or link the oracle."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for sub in ("lvio_fusion_amd", "include"):
        for dp, _, fs in os.walk(os.path.join(root, sub)):
            for f in fs:
                if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp", ".sh")):
                    txt = open(os.path.join(dp, f), errors="ignore").read()
                    for needle in ("pyoracle", "liblvf_oracle", "oracle/", "import oracle", "from oracle"):
                        hits = [ln for ln in txt.splitlines() if needle in ln and not ln.lstrip().startswith(("//", "#", "*", '"""'))]
                        # comments may cite the oracle; code may not use it
                        code_hits = [ln for ln in hits if "oracle/se3_ops.h" not in ln and "oracle/imu.h" not in ln and "oracle's" not in ln]
                        assert not code_hits, f"{sub}/{f} references the oracle: {code_hits[:2]}"
