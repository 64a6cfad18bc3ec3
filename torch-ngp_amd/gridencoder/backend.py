"""`_backend` of the grid encoder: the three callables the reference's pybind module exports
(gridencoder/src/bindings.cpp:5-9), same names, same positional arguments, tensors in place of
at::Tensor -- implemented by libngp_hip.so through its C ABI (include/ngp_hip.h).

A reference-style wrapper (`from .backend import _backend`, gridencoder/grid.py:9-12) works against this
object unchanged.
"""
import types

import _ngp_capi as capi


def _check_common(inputs, embeddings, offsets):
    capi.dense(inputs, 'inputs')
    capi.dense(embeddings, 'embeddings')
    capi.dense(offsets, 'offsets')
    capi.require_int32(offsets, 'offsets')
    if inputs.dtype != capi.torch.float32:
        # the reference reads inputs through data_ptr<float>() regardless of the dispatch type (gridencoder.cu:469)
        raise RuntimeError("expected scalar type Float for inputs but found " + str(inputs.dtype))


def grid_encode_forward(inputs, embeddings, offsets, outputs, B, D, C, L, S, H, dy_dx, gridtype, align_corners, interp):
    _check_common(inputs, embeddings, offsets)
    capi.dense(outputs, 'outputs')
    code = capi.float_code(embeddings, 'embeddings')
    capi.check(capi.lib.ngp_grid_encode_forward(
        capi.ptr(inputs), capi.ptr(embeddings), capi.ptr(offsets), capi.ptr(outputs), B, D, C, L, float(S), H,
        capi.ptr(dy_dx), gridtype, int(bool(align_corners)), interp, code, capi.stream()))


def grid_encode_backward(grad, inputs, embeddings, offsets, grad_embeddings, B, D, C, L, S, H, dy_dx, grad_inputs,
                         gridtype, align_corners, interp):
    _check_common(inputs, embeddings, offsets)
    capi.dense(grad, 'grad')
    capi.dense(grad_embeddings, 'grad_embeddings')
    code = capi.float_code(grad, 'grad')
    # large fp16 batches: the binned, atomic-free scatter needs scratch memory (include/ngp_hip.h, ngp_grid_encode_backward_ws); the
    # reference signature has no workspace argument, so it is allocated here
    arr, ws, nbytes = capi.grid_backward_workspace(offsets, B, D, C, L, S, H, gridtype, align_corners, code)
    capi.check(capi.lib.ngp_grid_encode_backward_ws(
        capi.ptr(grad), capi.ptr(inputs), capi.ptr(embeddings), capi.ptr(offsets), capi.ptr(grad_embeddings), B, D, C, L,
        float(S), H, capi.ptr(dy_dx), capi.ptr(grad_inputs), gridtype, int(bool(align_corners)), interp, code, 0.0,
        None if arr is None else capi.ctypes.cast(arr, capi.ctypes.c_void_p), capi.ptr(ws), nbytes, capi.stream()))


def grad_total_variation(inputs, embeddings, grad, offsets, weight, B, D, C, L, S, H, gridtype, align_corners):
    capi.dense(inputs, 'inputs')
    capi.dense(embeddings, 'embeddings')
    capi.dense(grad, 'grad')
    capi.dense(offsets, 'offsets')
    code = capi.float_code(embeddings, 'embeddings')
    if inputs.dtype != embeddings.dtype or grad.dtype != embeddings.dtype:
        raise RuntimeError("grad_total_variation: inputs, embeddings and grad must share one dtype")
    capi.check(capi.lib.ngp_grad_total_variation(
        capi.ptr(inputs), capi.ptr(embeddings), capi.ptr(grad), capi.ptr(offsets), float(weight), B, D, C, L, float(S), H,
        gridtype, int(bool(align_corners)), code, capi.stream()))


def grid_corner_indices(inputs, offsets, indices, B, D, L, S, H, gridtype, align_corners):
    """diagnostic extension, see include/ngp_hip.h"""
    capi.check(capi.lib.ngp_grid_corner_indices(capi.ptr(inputs), capi.ptr(offsets), capi.ptr(indices), B, D, L, float(S), H,
                                                gridtype, int(bool(align_corners)), capi.stream()))


_backend = types.SimpleNamespace(grid_encode_forward=grid_encode_forward, grid_encode_backward=grid_encode_backward,
                                 grad_total_variation=grad_total_variation, grid_corner_indices=grid_corner_indices)

__all__ = ['_backend']
