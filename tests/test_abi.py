"""The C-ABI library loads without a GPU and exports every symbol include/ehr.h declares (no compute calls)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "ehr.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ehr_[a-z_0-9]+)\s*\(", text)))


def test_header_declares_the_boundary():
    syms = declared_symbols()
    for name in ["ehr_ctx_create", "ehr_rasterize_fwd", "ehr_rasterize_grad", "ehr_interpolate_fwd",
                 "ehr_interpolate_grad", "ehr_antialias_topology", "ehr_antialias_fwd", "ehr_antialias_grad",
                 "ehr_render_mask_loss", "ehr_fused_plan"]:
        assert name in syms


def test_library_exports_every_declared_symbol():
    from easyhec_amd import _lib, build
    if not os.path.exists(_lib.LIB_PATH):
        build.build()
    so = ctypes.CDLL(_lib.LIB_PATH)
    missing = [s for s in declared_symbols() if not hasattr(so, s)]
    assert not missing, f"libehr_hip.so lacks {missing}"
    # and the ctypes table binds exactly the declared set
    assert sorted(_lib.SIGNATURES) == declared_symbols()


def test_library_reports_version_and_no_device_gracefully():
    from easyhec_amd import _lib
    lib = _lib.lib()
    assert lib.ehr_version() == 8
    n = lib.ehr_device_count()
    assert n >= 0
    if n == 0:
        h = ctypes.c_void_p()
        rc = lib.ehr_ctx_create(0, ctypes.byref(h))
        assert rc != 0 and b"out of range" in lib.ehr_last_error()


def test_product_fails_loudly_without_gpu_tensors():
    import torch
    from easyhec_amd import dr
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        dr.RasterizeCudaContext()
    with pytest.raises(RuntimeError):
        dr.interpolate(torch.ones(1, 3, 1), torch.zeros(1, 4, 4, 4), torch.zeros(1, 3, dtype=torch.int32))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "easyhec_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "libehr_oracle" not in src, f
