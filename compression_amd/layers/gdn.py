"""`GDN` layer (python/layers/gdn.py:41-470) on the fused HIP kernel."""
from __future__ import annotations

import torch

from . import functional, parameters

__all__ = ["GDN"]


class _GDNFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, beta, gamma, inverse, rectify, alpha, epsilon):
        ctx.save_for_backward(x, beta, gamma)
        ctx.cfg = (inverse, rectify, alpha, epsilon)
        return functional.gdn_forward(x, beta, gamma, inverse, rectify, alpha, epsilon)

    @staticmethod
    def backward(ctx, grad):
        x, beta, gamma = ctx.saved_tensors
        inverse, rectify, alpha, epsilon = ctx.cfg
        dx, dbeta, dgamma = functional.gdn_backward(x, grad.contiguous(), beta, gamma, inverse,
                                                    rectify, alpha, epsilon)
        return dx, dbeta, dgamma, None, None, None, None


class GDN(torch.nn.Module):
    """y_i = x_i / (beta_i + sum_j gamma[j, i] |x_j|^alpha)^epsilon  (inverse: multiply).

    Same constructor arguments as the reference layer (gdn.py:127-139).  `alpha_parameter` /
    `epsilon_parameter`: a number, a callable returning one, or None for a learned scalar
    (`reparam_alpha`, minimum 1 / `reparam_epsilon`, minimum 1e-6; gdn.py:345-369).  Fixed alpha in
    {1, 2} with epsilon in {1, .5} are the fused forward AND backward kernels (the fast paths of
    gdn.py:377-416); any other value runs the forward kernel's general-exponent variant, and its
    gradients (which include d/dalpha, d/depsilon) as device tensor ops (`functional.gdn_general_composite`).
    Weights: `reparam_beta` [C], `reparam_gamma` [C, C] (gdn_test.py:97-100), created
    on first call like a Keras `build`."""

    def __init__(self, inverse=False, rectify=False, data_format="channels_last",
                 alpha_parameter=1, beta_parameter=None, gamma_parameter=None,
                 epsilon_parameter=1, alpha_initializer=None, beta_initializer=None, gamma_initializer=None,
                 epsilon_initializer=None, num_channels=None):
        super().__init__()
        if data_format not in ("channels_first", "channels_last"):
            raise ValueError(f"Unknown data format: '{data_format}'.")
        self.inverse, self.rectify = bool(inverse), bool(rectify)
        self.data_format = data_format
        self._alpha_fixed, self._epsilon_fixed = alpha_parameter, epsilon_parameter
        self._beta_fixed, self._gamma_fixed = beta_parameter, gamma_parameter
        self._alpha_init = alpha_initializer or (lambda: torch.ones(()))
        self._epsilon_init = epsilon_initializer or (lambda: torch.ones(()))
        self._beta_init = beta_initializer or (lambda c: torch.ones(c))
        self._gamma_init = gamma_initializer or (lambda c: 0.1 * torch.eye(c))
        self.reparam_beta = self.reparam_gamma = None
        self.reparam_alpha = self.reparam_epsilon = None
        if alpha_parameter is None:
            self.reparam_alpha = torch.nn.Parameter(parameters.gdn_reparam_init(self._alpha_init().float()))
        if epsilon_parameter is None:
            self.reparam_epsilon = torch.nn.Parameter(parameters.gdn_reparam_init(self._epsilon_init().float()))
        if num_channels is not None:
            self.build(int(num_channels))

    @property
    def alpha(self):
        """A Python number when fixed, a 0-d tensor when learned (gdn.py:318-321)."""
        if self.reparam_alpha is not None:
            return parameters.gdn_reparam_value(self.reparam_alpha, minimum=1.0)
        v = self._alpha_fixed() if callable(self._alpha_fixed) else self._alpha_fixed
        return v

    @property
    def epsilon(self):
        if self.reparam_epsilon is not None:
            return parameters.gdn_reparam_value(self.reparam_epsilon, minimum=1e-6)
        v = self._epsilon_fixed() if callable(self._epsilon_fixed) else self._epsilon_fixed
        return v

    def build(self, c, device=None):
        if self._beta_fixed is None and self.reparam_beta is None:
            self.reparam_beta = torch.nn.Parameter(
                parameters.gdn_reparam_init(self._beta_init(c).float()).to(device))
        if self._gamma_fixed is None and self.reparam_gamma is None:
            self.reparam_gamma = torch.nn.Parameter(
                parameters.gdn_reparam_init(self._gamma_init(c).float()).to(device))

    def _cached_value(self, name, variable, minimum):
        """Under no_grad (compress / decompress) the reparameterised value is computed once per version
        of its variable instead of once per call (three small kernels per parameter and call otherwise).
        Keyed on storage and version counter like SignalConv2D's kernel cache: `invalidate_kernel_cache()`
        after a write through `.data`."""
        try:
            key = (variable.data_ptr(), variable._version, str(variable.device))
        except RuntimeError:                      # inference tensor: no version counter, no cache
            return parameters.gdn_reparam_value(variable, minimum=minimum)
        cache = self.__dict__.setdefault("_value_cache", {})
        hit = cache.get(name)
        if hit is None or hit[0] != key:
            v = parameters.gdn_reparam_value(variable, minimum=minimum).contiguous()
            if v.is_cuda:
                torch.cuda.current_stream().synchronize()      # complete before another stream reads it
            cache[name] = hit = (key, v)
        return hit[1]

    def _prepared_params(self, beta, gamma, dtype):
        """functional.GDNPrepared of the cached beta / gamma values (rebuilt when they are)."""
        cache = self.__dict__.setdefault("_value_cache", {})
        key = (beta.data_ptr(), gamma.data_ptr(), str(dtype), str(beta.device))
        hit = cache.get("prepared")
        if hit is None or hit[0] != key:
            prepared = functional.GDNPrepared(beta, gamma, dtype)
            # the image is built by a kernel on the CURRENT stream and then read by whatever stream runs a later
            # call (every pipeline lane has its own): complete before it is cached, like _cached_value
            torch.cuda.current_stream().synchronize()
            cache["prepared"] = hit = (key, prepared, beta, gamma)
        return hit[1]

    # The cache holds native handles (functional.GDNPrepared): it is not copied or pickled with the module, the
    # copy rebuilds its own on first use (copy.deepcopy for an EMA model, torch.save of the module object).
    def __getstate__(self):
        state = self.__dict__.copy()
        state.pop("_value_cache", None)
        return state

    def __deepcopy__(self, memo):
        import copy
        cls = self.__class__
        new = cls.__new__(cls)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            if k != "_value_cache":
                new.__dict__[k] = copy.deepcopy(v, memo)
        return new

    def invalidate_kernel_cache(self):
        self.__dict__["_value_cache"] = {}

    def _load_from_state_dict(self, *args, **kwargs):
        self.invalidate_kernel_cache()
        return super()._load_from_state_dict(*args, **kwargs)

    def _apply(self, fn, *args, **kwargs):
        self.invalidate_kernel_cache()
        return super()._apply(fn, *args, **kwargs)

    def train(self, mode=True):
        self.invalidate_kernel_cache()
        return super().train(mode)

    @property
    def beta(self):
        if self._beta_fixed is not None:
            return torch.as_tensor(self._beta_fixed() if callable(self._beta_fixed) else self._beta_fixed)
        if not torch.is_grad_enabled():
            return self._cached_value("beta", self.reparam_beta, 1e-6)
        return parameters.gdn_reparam_value(self.reparam_beta, minimum=1e-6)

    @property
    def gamma(self):
        if self._gamma_fixed is not None:
            return torch.as_tensor(self._gamma_fixed() if callable(self._gamma_fixed) else self._gamma_fixed)
        if not torch.is_grad_enabled():
            return self._cached_value("gamma", self.reparam_gamma, 0.0)
        return parameters.gdn_reparam_value(self.reparam_gamma, minimum=0.0)

    def forward(self, inputs):
        if inputs.dim() < 2:
            raise ValueError(f"Input tensor must have at least rank 2, received shape {tuple(inputs.shape)}.")
        x = inputs
        if self.data_format == "channels_first" and x.dim() > 2:
            x = x.movedim(1, -1)
        self.build(x.shape[-1], x.device)
        beta, gamma = self.beta.to(x.device), self.gamma.to(x.device)
        alpha, epsilon = self.alpha, self.epsilon
        fast = (not torch.is_tensor(alpha) and not torch.is_tensor(epsilon)
                and float(alpha) in (1.0, 2.0) and float(epsilon) in (1.0, 0.5))
        needs_grad = torch.is_grad_enabled() and any(
            torch.is_tensor(t) and t.requires_grad for t in (x, beta, gamma, alpha, epsilon))
        if fast and not needs_grad and self._beta_fixed is None and self._gamma_fixed is None \
                and x.is_cuda and x.dtype in functional._DTYPE_CODE \
                and (x.dtype != torch.float32 or x.shape[-1] <= 192):
            # inference on the layer's own variables: the kernels' parameter image is prepared once per
            # parameter version (same cache key as the reparameterised values)
            y = functional.gdn_forward(x.contiguous(), beta, gamma, self.inverse, self.rectify, float(alpha),
                                       float(epsilon), prepared=self._prepared_params(beta, gamma, x.dtype))
        elif fast:
            y = _GDNFunction.apply(x.contiguous(), beta, gamma, self.inverse, self.rectify,
                                   int(alpha) if float(alpha) in (1.0, 2.0) else alpha,
                                   1 if float(epsilon) == 1.0 else 0.5)
        elif needs_grad:
            a = alpha.to(x.device) if torch.is_tensor(alpha) else float(alpha)
            e = epsilon.to(x.device) if torch.is_tensor(epsilon) else float(epsilon)
            y = functional.gdn_general_composite(x, beta, gamma, a, e, self.inverse, self.rectify)
        else:
            y = functional.gdn_forward(x.contiguous(), beta, gamma, self.inverse, self.rectify,
                                       float(alpha), float(epsilon))
        if self.data_format == "channels_first" and y.dim() > 2:
            y = y.movedim(-1, 1)
        return y
