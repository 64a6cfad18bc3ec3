/* ref_glue_shencoder.cpp -- TEST INFRASTRUCTURE.  Appended by oracle/Makefile to the (piped, never stored) text of the reference's
 * shencoder/src/shencoder.cu (shencoder.h:9-10); fp32 (sphere_harmonics.py forces it). */
#define ORC_EXPORT extern "C" __attribute__((visibility("default")))
#define F32(p) at::Tensor((void*)(p), at::ScalarType::Float)
ORC_EXPORT int ref_sh_encode_forward(const float* inputs, float* outputs, uint32_t B, uint32_t D, uint32_t C, float* dy_dx) {
    try {
        sh_encode_forward(F32(inputs), F32(outputs), B, D, C, dy_dx ? at::optional<at::Tensor>(F32(dy_dx)) : at::optional<at::Tensor>());
    } catch (const std::exception& e) { fprintf(stderr, "sh_encode_forward: %s\n", e.what()); return 1; }
    return 0;
}
ORC_EXPORT int ref_sh_encode_backward(const float* grad, const float* inputs, uint32_t B, uint32_t D, uint32_t C, const float* dy_dx, float* grad_inputs) {
    try { sh_encode_backward(F32(grad), F32(inputs), B, D, C, F32(dy_dx), F32(grad_inputs)); }
    catch (const std::exception& e) { fprintf(stderr, "sh_encode_backward: %s\n", e.what()); return 1; }
    return 0;
}
