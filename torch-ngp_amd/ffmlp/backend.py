"""`_backend` of the fully fused MLP: the five callables of ffmlp/src/bindings.cpp:5-11 over libngp_hip.so.
All tensors must be fp16 CUDA tensors (the reference's CHECK_IS_HALF, ffmlp.cu:636-642)."""
import types

import torch

import _ngp_capi as capi


def _half(t, name):
    capi.dense(t, name)
    if t.dtype != torch.float16:
        raise RuntimeError(f"{name} must be a Half tensor")
    return t


def ffmlp_forward(inputs, weights, B, input_dim, output_dim, hidden_dim, num_layers, activation, output_activation,
                  forward_buffer, outputs):
    for t, n in ((inputs, 'inputs'), (weights, 'weights'), (forward_buffer, 'forward_buffer'), (outputs, 'outputs')):
        _half(t, n)
    capi.check(capi.lib.ngp_ffmlp_forward(capi.ptr(inputs), capi.ptr(weights), B, input_dim, output_dim, hidden_dim, num_layers,
                                          activation, output_activation, capi.ptr(forward_buffer), capi.ptr(outputs), capi.stream()))


def ffmlp_inference(inputs, weights, B, input_dim, output_dim, hidden_dim, num_layers, activation, output_activation,
                    inference_buffer, outputs):
    for t, n in ((inputs, 'inputs'), (weights, 'weights'), (outputs, 'outputs')):
        _half(t, n)
    capi.check(capi.lib.ngp_ffmlp_inference(capi.ptr(inputs), capi.ptr(weights), B, input_dim, output_dim, hidden_dim, num_layers,
                                            activation, output_activation, capi.ptr(inference_buffer), capi.ptr(outputs), capi.stream()))


def ffmlp_backward(grad, inputs, weights, forward_buffer, B, input_dim, output_dim, hidden_dim, num_layers, activation,
                   output_activation, calc_grad_inputs, backward_buffer, grad_inputs, grad_weights):
    for t, n in ((grad, 'grad'), (inputs, 'inputs'), (weights, 'weights'), (forward_buffer, 'forward_buffer'),
                 (backward_buffer, 'backward_buffer'), (grad_inputs, 'grad_inputs'), (grad_weights, 'grad_weights')):
        _half(t, n)
    # shapes outside the register-resident kernels (hidden 16/128/256, > 4 hidden layers, wide inputs) split the batch reduction of the
    # weight gradients over sample chunks: that needs scratch the reference signature has no argument for, so it is allocated here
    nbytes = int(capi.lib.ngp_ffmlp_backward_workspace_bytes(B, input_dim, hidden_dim, num_layers))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=grad.device) if nbytes else None
    capi.check(capi.lib.ngp_ffmlp_backward_ws(capi.ptr(grad), capi.ptr(inputs), capi.ptr(weights), capi.ptr(forward_buffer), B, input_dim,
                                              output_dim, hidden_dim, num_layers, activation, output_activation, int(bool(calc_grad_inputs)),
                                              capi.ptr(backward_buffer), capi.ptr(grad_inputs), capi.ptr(grad_weights), 0, capi.ptr(ws), nbytes,
                                              capi.stream()))


def allocate_splitk(size):
    capi.check(capi.lib.ngp_allocate_splitk(size))


def free_splitk():
    capi.check(capi.lib.ngp_free_splitk())


_backend = types.SimpleNamespace(ffmlp_forward=ffmlp_forward, ffmlp_inference=ffmlp_inference, ffmlp_backward=ffmlp_backward,
                                 allocate_splitk=allocate_splitk, free_splitk=free_splitk)

__all__ = ['_backend']
