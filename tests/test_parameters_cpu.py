"""CPU tier: layer parameter objects and initializers (python/layers/parameters.py, initializers.py) — the
reference's parameters_test.py and initializers_test.py cases — and the layers taking them."""
import pytest
import torch

import compression_amd as tfc


@pytest.mark.parametrize("cls,shape", [("RDFTParameter", (3, 3, 1, 2)), ("RDFTParameter", (7, 3, 2)),
                                       ("RDFTParameter", (2, 3, 4, 2, 3)), ("GDNParameter", (2, 1, 3))])
def test_initial_value_is_reproduced_and_weights_round_trip(cls, shape):
    torch.manual_seed(0)
    initial = torch.rand(shape)
    parameter = getattr(tfc, cls)(initial)
    assert torch.allclose(parameter(), initial, atol=1e-6, rtol=0)
    assert parameter().dtype == torch.float32
    # a fresh object of the same configuration takes the stored variables and gives the same value
    cfg = parameter.get_config()
    clone = getattr(tfc, cls)(None, **{k: v for k, v in cfg.items() if k in ("shape", "minimum", "offset")})
    clone.set_weights(parameter.get_weights())
    assert torch.equal(clone(), parameter())
    with pytest.raises(ValueError):
        clone.set_weights(parameter.get_weights()[:-1] if len(parameter.get_weights()) > 1 else [])
    with pytest.raises(ValueError):
        getattr(tfc, cls)(None)


@pytest.mark.parametrize("shape", [(7, 3, 2), (5, 3, 1, 2)])
def test_rdft_gradients_propagate(shape):
    torch.manual_seed(1)
    parameter = tfc.RDFTParameter(torch.rand(shape))
    loss = (torch.rand(shape) * parameter()).sum()
    g_real, g_imag = torch.autograd.grad(loss, [parameter.real, parameter.imag])
    assert g_real.abs().max() > 0.1 and g_imag.abs().max() > 0.1
    assert parameter(torch.bfloat16).dtype == torch.bfloat16


def test_gdn_parameter_minimum():
    torch.manual_seed(2)
    initial = torch.rand(2, 1, 3)
    parameter = tfc.GDNParameter(initial, minimum=0.5)
    assert torch.allclose(parameter(), torch.clamp(initial, min=0.5), atol=1e-6, rtol=0)
    assert parameter.minimum == 0.5 and parameter.offset == 2 ** -18


def test_identity_initializer():
    k = tfc.IdentityInitializer(gain=3)((3, 4, 3), dtype=torch.int32)
    want = torch.tensor([[[0, 3, 0], [0, 0, 0], [0, 0, 0]],
                         [[0, 0, 0], [0, 3, 0], [0, 0, 0]],
                         [[0, 0, 0], [0, 0, 0], [0, 3, 0]],
                         [[0, 0, 0], [0, 0, 0], [0, 0, 0]]], dtype=torch.int32).permute(2, 0, 1)
    assert torch.equal(k, want)
    k = tfc.IdentityInitializer()((4, 5, 1, 1))
    want = torch.zeros(4, 5)
    want[2, 2] = 1
    assert torch.equal(k, want[:, :, None, None])
    with pytest.raises(ValueError):
        tfc.IdentityInitializer()((2, 3))


def test_layers_take_parameter_objects():
    """signal_conv.py:222-236, gdn.py:127-139: a tensor, a callable or a Parameter in place of the layer's own
    variables; the layer's default weight names are unchanged."""
    plain = tfc.SignalConv2D(8, 3, padding="same_zeros", use_bias=True, in_channels=4)
    assert sorted(plain.state_dict()) == ["bias", "kernel_imag", "kernel_real"]
    ident = tfc.IdentityInitializer()((3, 3, 4, 8))
    layer = tfc.SignalConv2D(8, 3, padding="same_zeros", use_bias=True, kernel_parameter=tfc.RDFTParameter(ident),
                             bias_parameter=torch.arange(8.0), in_channels=4)
    assert torch.allclose(layer.kernel, ident, atol=1e-6) and torch.equal(layer._bias_value(), torch.arange(8.0))
    assert sorted(layer.state_dict()) == ["_kernel_given.imag", "_kernel_given.real"]
    assert torch.equal(tfc.SignalConv2D(8, 3, padding="same_zeros", kernel_parameter=ident, in_channels=4).kernel, ident)
    with pytest.raises(ValueError):
        tfc.SignalConv2D(8, 3, padding="same_zeros", kernel_parameter="dct")
    beta = tfc.GDNParameter(torch.full((6,), 2.0), minimum=1e-6)
    gdn = tfc.GDN(beta_parameter=beta, gamma_parameter=lambda: 0.05 * torch.eye(6), num_channels=6)
    assert torch.allclose(gdn.beta, torch.full((6,), 2.0), atol=1e-6) and torch.equal(gdn.gamma, 0.05 * torch.eye(6))
    assert [n for n, _ in gdn.named_parameters()] == ["_beta_fixed.variable"]
