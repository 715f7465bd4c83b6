"""CPU tier: the C-ABI library builds, loads, and exports exactly the symbols
include/tfc_hip.h declares (no compute calls — there is no GPU here)."""
import ctypes
import os
import re

from conftest import ROOT


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "tfc_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return set(re.findall(r"\b(tfc_[a-z0-9_]+)\s*\(", text))


def test_library_exports_every_declared_symbol():
    from compression_amd import _lib
    lib = _lib.lib()
    declared = _header_symbols()
    assert declared, "no symbols parsed from include/tfc_hip.h"
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in tfc_hip.h but not exported"
    assert set(_lib.SIGNATURES) == declared
    assert lib.tfc_abi_version() == 2


def test_every_entry_point_cites_reference():
    text = open(os.path.join(ROOT, "include", "tfc_hip.h")).read()
    for needle in ("range_coder_ops.cc", "range_coder_kernels.cc", "range_coding_kernels.cc",
                   "pmf_to_cdf_kernels.cc", "gdn.py", "signal_conv.py", "continuous_batched.py"):
        assert needle in text


def test_table_validation_runs_without_gpu_errors_are_textual():
    # tfc_tables_create validates on the host before touching the device.
    from compression_amd import _lib
    import numpy as np
    lib = _lib.lib()
    bad = np.array([12, 1, 4096], np.int32)
    out = ctypes.c_void_p()
    rc = lib.tfc_tables_create(bad.ctypes.data, 1, 1, bad.size, None, ctypes.byref(out))
    assert rc != 0 and "CDF must start with 0." in _lib.last_error()
    rc = lib.tfc_tables_create(bad.ctypes.data, 3, 1, bad.size, None, ctypes.byref(out))
    assert rc != 0 and "`lookup` must be rank 1 or 2" in _lib.last_error()


def test_ops_fail_loudly_without_device():
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip("a device is present")
    import compression_amd as tfc
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        tfc.create_range_encoder([2], torch.tensor([12, 0, 4096], dtype=torch.int32))


def test_argument_validation_of_the_transform_and_bits_calls_is_host_side():
    """Bad scalar arguments are rejected before anything is launched (null tensors, no device needed)."""
    from compression_amd import _lib
    lib = _lib.lib()
    for m in (0.0, 1.0, -0.5, float("nan")):
        rc = lib.tfc_factorized_bits_forward_tail(None, None, None, 0, 1, 3, 3, None, 3, 3, m, None, None, None)
        assert rc != 0 and "laplace_tail_mass must be in (0, 1)" in _lib.last_error()
        rc = lib.tfc_noisy_normal_bits_backward_tail(None, None, None, 0, 1, 3, m, None, None, None, None)
        assert rc != 0 and "laplace_tail_mass must be in (0, 1)" in _lib.last_error()
    rc = lib.tfc_gdn_forward_general(None, None, 0, 32, 32, None, None, 0, 0, -1.0, 1.0, None)
    assert rc != 0 and "alpha and epsilon must be positive" in _lib.last_error()
    rc = lib.tfc_gdn_forward(None, None, 0, 32, 48, None, None, 0, 0, 1, 0, None)
    assert rc != 0 and "multiple of 32" in _lib.last_error()
    rc = lib.tfc_gdn_forward(None, None, 7, 32, 32, None, None, 0, 0, 1, 0, None)
    assert rc != 0 and "dtype" in _lib.last_error()
