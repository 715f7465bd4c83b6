"""CPU tier: the hot kernels of libtfc_hip.so keep their state in registers.  A register spill inside a K loop
or a coding step passes every parity test and costs a factor (found twice in round 2: the fused GDN backward
kernel, and the 6-tile convolution kernel after an innocent-looking epilogue change)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

HOT = ["conv_bf16_kernel", "conv3_bf16_kernel", "conv_image_kernel", "conv_image_direct_kernel", "conv_up_fused_kernel",
       "conv_up_phase_kernel",
       "image_to_unit_kernel", "unit_to_image_kernel", "index_prepare_kernel",
       "enc_lanes_kernel", "dec_lanes_kernel", "enc_fast_kernel", "dec_fast_kernel",
       "enc_expand_kernel", "enc_chain_kernel", "enc_chain_direct_kernel", "dec_chain_kernel", "dec_parse_kernel",
       "gdn_fwd_bf16_kernelILi6E", "gdn_bwd_fused_bf16_kernelILi6E", "gdn_param_grad_kernelItLi6E",
       "noisy_normal_forward_kernel", "noisy_normal_backward_kernel", "factorized_forward_kernel",
       "ELi256EEEvNS_10BitsParamsE"]        # factorized_backward_kernel<..., MAXT = 256>: every MLP shape


def test_hot_kernels_do_not_spill():
    lib = os.path.join(ROOT, "compression_amd", "libtfc_hip.so")
    if not os.path.exists(lib):
        pytest.skip("libtfc_hip.so is not built")
    import check_scratch
    table = check_scratch.scan(lib)
    assert len(table) > 100
    for key in HOT:
        hits = {n: r for n, r in table.items() if key in n}
        assert hits, key
        spilled = {n: r["scratch"] for n, r in hits.items() if r["scratch"]}
        assert not spilled, spilled
