"""`_backend` of the SH encoder: the two callables of shencoder/src/bindings.cpp:5-8 over libngp_hip.so."""
import types

import _ngp_capi as capi


def sh_encode_forward(inputs, outputs, B, D, C, dy_dx):
    capi.dense(inputs, 'inputs')
    capi.dense(outputs, 'outputs')
    code = capi.float_code(inputs, 'inputs')
    capi.check(capi.lib.ngp_sh_encode_forward(capi.ptr(inputs), capi.ptr(outputs), B, D, C, capi.ptr(dy_dx), code, capi.stream()))


def sh_encode_backward(grad, inputs, B, D, C, dy_dx, grad_inputs):
    for t, name in ((grad, 'grad'), (inputs, 'inputs'), (dy_dx, 'dy_dx'), (grad_inputs, 'grad_inputs')):
        capi.dense(t, name)
    code = capi.float_code(grad, 'grad')
    capi.check(capi.lib.ngp_sh_encode_backward(capi.ptr(grad), capi.ptr(inputs), B, D, C, capi.ptr(dy_dx), capi.ptr(grad_inputs),
                                               code, capi.stream()))


_backend = types.SimpleNamespace(sh_encode_forward=sh_encode_forward, sh_encode_backward=sh_encode_backward)

__all__ = ['_backend']
