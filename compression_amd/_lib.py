"""ctypes binding of the C ABI in include/tfc_hip.h (libtfc_hip.so).

There is no CPU fallback: if the library is missing this module raises on
import of the symbol table, and every op raises if no HIP device is present.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libtfc_hip.so")

_vp = C.c_void_p
_i64 = C.c_int64
_int = C.c_int

# name -> (restype, argtypes); must list every symbol include/tfc_hip.h declares.
SIGNATURES = {
    "tfc_abi_version": (_int, []),
    "tfc_last_error": (C.c_char_p, []),
    "tfc_profile_enable": (None, [_int]),
    "tfc_set_default_mode": (_int, [_int]),
    "tfc_get_default_mode": (_int, []),
    "tfc_set_chip_shared": (_int, [_int]),
    "tfc_image_to_unit": (_int, [_vp, _vp, _int, C.c_int64, _vp]),
    "tfc_unit_to_image": (_int, [_vp, _int, _vp, C.c_int64, _vp]),
    "tfc_index_prepare": (_int, [_vp, _int, _vp, C.c_int64, _int, _vp]),
    "tfc_build_tables": (_int, [_vp, C.c_int64, C.c_int64, _vp, _vp, C.c_int64, _int, _vp, _vp]),
    "tfc_build_tables_overflow": (_int, [_vp, C.c_int64, C.c_int64, _vp, _vp, C.c_int64, _int, _vp, _vp, _vp]),
    "tfc_deep_factorized_tails": (_int, [_vp, C.c_int64, C.c_int64, _int, _int, _vp, _int, _vp, _vp, _vp]),
    "tfc_pad2d": (_int, [_vp, _vp, _int, C.c_int64, C.c_int64, C.c_int64, C.c_int64, _int, _int, _int, _int, _int, _vp]),
    "tfc_cache_bytes": (_int, [C.POINTER(C.c_longlong)]),
    "tfc_cache_trim": (_int, [C.POINTER(C.c_longlong)]),
    "tfc_encoder_capacity": (_int, [_vp, C.POINTER(_i64)]),
    "tfc_profile_query": (_int, [C.c_char_p, C.POINTER(C.c_double), C.POINTER(_i64)]),
    "tfc_pipe_counters": (_int, [C.POINTER(_i64), C.POINTER(_i64)]),
    "tfc_set_pipe_format": (_int, [_int, _int]),
    "tfc_tables_create": (_int, [_vp, _int, _i64, _i64, _vp, C.POINTER(_vp)]),
    "tfc_tables_count": (_i64, [_vp]),
    "tfc_tables_destroy": (None, [_vp]),
    "tfc_encoder_create": (_int, [_vp, _i64, _vp, C.POINTER(_vp)]),
    "tfc_encoder_create_many": (_int, [_vp, _i64, _int, _vp, _vp]),
    "tfc_encoder_set_mode": (_int, [_vp, _int]),
    "tfc_encoder_set_deferred_errors": (_int, [_vp, _int]),
    "tfc_encoder_encode": (_int, [_vp, _vp, _vp, _i64, _vp]),
    "tfc_encoder_encode_many": (_int, [_int, _vp, _vp, _vp, _i64, _vp]),
    "tfc_encoder_encode_quantized": (_int, [_vp, _vp, _int, _vp, _vp, _i64, _i64, _vp]),
    "tfc_encoder_encode_quantized_indexed": (_int, [_vp, _vp, _int, _vp, _vp, _i64, _vp]),
    "tfc_encoder_encode_quantized_many": (_int, [_int, _vp, _vp, _int, _vp, _vp, _i64, _i64, _vp]),
    "tfc_decoder_decode_dequantized_many": (_int, [_int, _vp, _vp, _int, _vp, _vp, _i64, _i64, _vp]),
    "tfc_encoder_encode_quantized_indexed_many": (_int, [_int, _vp, _vp, _int, _vp, _vp, _i64, _vp]),
    "tfc_decoder_decode_dequantized_indexed_many": (_int, [_int, _vp, _vp, _vp, _int, _vp, _i64, _vp]),
    "tfc_encoder_finalize": (_int, [_vp, _vp, C.POINTER(_i64)]),
    "tfc_encoder_finalize_device": (_int, [_vp, _vp]),
    "tfc_encoder_finalize_device_many": (_int, [_int, _vp, _vp]),
    "tfc_encoder_status": (_int, [_vp, _vp, C.POINTER(_i64)]),
    "tfc_encoder_result": (_int, [_vp, C.POINTER(_vp), C.POINTER(_vp)]),
    "tfc_encoder_read": (_int, [_vp, _vp, _vp, _int, _vp]),
    "tfc_encoder_destroy": (None, [_vp]),
    "tfc_decoder_create": (_int, [_vp, _vp, _vp, _i64, _int, _vp, C.POINTER(_vp)]),
    "tfc_decoder_create_many": (_int, [_vp, _int, _vp, _vp, _vp]),
    "tfc_decoder_set_mode": (_int, [_vp, _int]),
    "tfc_decoder_decode": (_int, [_vp, _vp, _vp, _i64, _vp]),
    "tfc_decoder_decode_many": (_int, [_int, _vp, _vp, _vp, _i64, _vp]),
    "tfc_decoder_decode_dequantized": (_int, [_vp, _vp, _vp, _int, _vp, _vp, _i64, _i64, _vp]),
    "tfc_decoder_finalize": (_int, [_vp, _vp, _vp]),
    "tfc_decoder_finalize_device": (_int, [_vp, _vp, _vp]),
    "tfc_decoder_finalize_device_many": (_int, [_int, _vp, _vp, _vp]),
    "tfc_decoder_status": (_int, [_vp, _vp]),
    "tfc_decoder_destroy": (None, [_vp]),
    "tfc_range_encode": (_int, [_vp, _vp, _int, _vp, _vp, _int, _int, _int, _vp,
                                C.POINTER(_vp), C.POINTER(_i64)]),
    "tfc_range_decode": (_int, [_vp, _i64, _vp, _int, _vp, _vp, _int, _int, _int, _vp, _vp]),
    "tfc_unbounded_index_range_encode": (_int, [_vp, _vp, _i64, _vp, _i64, _i64, _vp, _vp, _int, _int, _int,
                                                _vp, C.POINTER(_vp), C.POINTER(_i64)]),
    "tfc_unbounded_index_range_decode": (_int, [_vp, _i64, _vp, _i64, _vp, _i64, _i64, _vp, _vp, _int, _int,
                                                _int, _vp, _vp]),
    "tfc_stochastic_round": (_int, [_vp, _int, _i64, C.c_float, _vp, _i64, _vp, _vp]),
    "tfc_free": (None, [_vp]),
    "tfc_pmf_to_quantized_cdf": (_int, [_vp, _i64, _i64, _int, _vp, _vp]),
    "tfc_gdn_forward": (_int, [_vp, _vp, _int, _i64, _i64, _vp, _vp, _int, _int, _int, _int, _vp]),
    "tfc_gdn_params_create": (_int, [_vp, _vp, _i64, _int, _vp, C.POINTER(_vp)]),
    "tfc_gdn_params_destroy": (None, [_vp]),
    "tfc_gdn_forward_prepared": (_int, [_vp, _vp, _vp, _i64, _int, _int, _int, _int, _vp]),
    "tfc_gdn_forward_general": (_int, [_vp, _vp, _int, _i64, _i64, _vp, _vp, _int, _int, C.c_float, C.c_float, _vp]),
    "tfc_gdn_backward": (_int, [_vp, _vp, _vp, _int, _i64, _i64, _vp, _vp, _int, _int, _int,
                                _int, _vp, _vp, _vp]),
    "tfc_conv2d_down": (_int, [_vp, _vp, _vp, _vp, _int, _i64, _i64, _i64, _i64, _i64, _int,
                               _int, _int, _int, _vp]),
    "tfc_conv2d_up": (_int, [_vp, _vp, _vp, _vp, _int, _i64, _i64, _i64, _i64, _i64, _int,
                             _int, _int, _int, _vp]),
    "tfc_conv2d_gdn": (_int, [_vp, _vp, _vp, _vp, _int, _i64, _i64, _i64, _i64, _i64, _int, _int, _int, _int, _vp, _int,
                              C.POINTER(_int), _vp]),
    "tfc_conv2d_weights_key": (None, [C.c_uint64]),
    "tfc_conv2d_drop_weights": (_int, [C.c_uint64]),
    "tfc_conv2d_wgrad": (_int, [_vp, _vp, _vp, _int, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _int,
                                _int, _int, _int, _vp]),
    "tfc_factorized_bits_forward": (_int, [_vp, _vp, _vp, _int, _i64, _i64, _i64, _vp, _int, _int,
                                           _vp, _vp, _vp]),
    "tfc_factorized_bits_backward": (_int, [_vp, _int, _i64, _i64, _i64, _vp, _int, _int, _vp, _vp,
                                            _vp, _vp]),
    "tfc_noisy_normal_bits_forward": (_int, [_vp, _vp, _vp, _vp, _int, _i64, _i64, _vp, _vp]),
    "tfc_noisy_normal_bits_backward": (_int, [_vp, _vp, _vp, _int, _i64, _i64, _vp, _vp, _vp, _vp]),
    "tfc_factorized_bits_backward_expected": (_int, [_vp, _vp, _int, _i64, _i64, _i64, _vp, _int, _int, _vp, _vp,
                                                     _vp, _vp]),
    "tfc_factorized_bits_forward_tail": (_int, [_vp, _vp, _vp, _int, _i64, _i64, _i64, _vp, _int, _int, C.c_float,
                                                _vp, _vp, _vp]),
    "tfc_factorized_bits_backward_tail": (_int, [_vp, _vp, _int, _i64, _i64, _i64, _vp, _int, _int, C.c_float, _vp,
                                                 _vp, _vp, _vp]),
    "tfc_noisy_normal_bits_forward_tail": (_int, [_vp, _vp, _vp, _vp, _int, _i64, _i64, C.c_float, _vp, _vp]),
    "tfc_noisy_normal_bits_backward_tail": (_int, [_vp, _vp, _vp, _int, _i64, _i64, C.c_float, _vp, _vp, _vp, _vp]),
}

ABI_VERSION = 2          # include/tfc_hip.h TFC_ABI_VERSION this binding was written against

_lib = None


class HipLibraryMissing(RuntimeError):
    pass


def lib():
    """Loads libtfc_hip.so once; raises HipLibraryMissing if it was not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipLibraryMissing(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback.")
        handle = C.CDLL(LIB_PATH)
        try:
            handle.tfc_abi_version.restype = _int
            version = handle.tfc_abi_version()
        except AttributeError:
            version = None
        if version != ABI_VERSION:
            raise HipLibraryMissing(
                f"{LIB_PATH} has ABI version {version}, this package needs {ABI_VERSION}: rebuild it with "
                "`python -c 'import __graft_entry__ as g; g.build(force=True)'`")
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def last_error() -> str:
    return lib().tfc_last_error().decode()


def check(rc: int) -> None:
    """Maps a non-zero status to the exception type the reference ops raise
    (tf.errors.InvalidArgumentError is a ValueError-like; we use ValueError)."""
    if rc:
        raise ValueError(last_error())


def require_device():
    import torch

    if not torch.cuda.is_available():
        raise RuntimeError(
            "compression_amd needs a HIP device (MI355X): torch.cuda.is_available() is False "
            "and there is no CPU fallback for the HIP kernels.")
    return torch.device("cuda", torch.cuda.current_device())


def stream_ptr() -> int:
    import torch

    return torch.cuda.current_stream().cuda_stream
