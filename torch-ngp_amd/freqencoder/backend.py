"""`_backend` of the frequency encoder: the two callables of freqencoder/src/bindings.cpp:5-8 over libngp_hip.so (fp32 only, as the
reference, which reads data_ptr<float>() on every tensor)."""
import types

import torch

import _ngp_capi as capi


def _f32(t, name):
    capi.dense(t, name)
    if t.dtype != torch.float32:
        raise RuntimeError(f"expected scalar type Float for {name} but found {t.dtype}")
    return t


def freq_encode_forward(inputs, B, D, deg, C, outputs):
    _f32(inputs, 'inputs'); _f32(outputs, 'outputs')
    capi.check(capi.lib.ngp_freq_encode_forward(capi.ptr(inputs), B, D, deg, C, capi.ptr(outputs), capi.stream()))


def freq_encode_backward(grad, outputs, B, D, deg, C, grad_inputs):
    _f32(grad, 'grad'); _f32(outputs, 'outputs'); _f32(grad_inputs, 'grad_inputs')
    capi.check(capi.lib.ngp_freq_encode_backward(capi.ptr(grad), capi.ptr(outputs), B, D, deg, C, capi.ptr(grad_inputs), capi.stream()))


_backend = types.SimpleNamespace(freq_encode_forward=freq_encode_forward, freq_encode_backward=freq_encode_backward)

__all__ = ['_backend']
