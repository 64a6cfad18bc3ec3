/* ref_glue_freqencoder.cpp -- TEST INFRASTRUCTURE.  Appended by oracle/Makefile to the (piped, never stored) text of the reference's
 * freqencoder/src/freqencoder.cu (freqencoder.h:6-10); fp32. */
#define ORC_EXPORT extern "C" __attribute__((visibility("default")))
#define F32(p) at::Tensor((void*)(p), at::ScalarType::Float)
ORC_EXPORT int ref_freq_encode_forward(const float* inputs, uint32_t B, uint32_t D, uint32_t deg, uint32_t C, float* outputs) {
    try { freq_encode_forward(F32(inputs), B, D, deg, C, F32(outputs)); }
    catch (const std::exception& e) { fprintf(stderr, "freq_encode_forward: %s\n", e.what()); return 1; }
    return 0;
}
ORC_EXPORT int ref_freq_encode_backward(const float* grad, const float* outputs, uint32_t B, uint32_t D, uint32_t deg, uint32_t C, float* grad_inputs) {
    try { freq_encode_backward(F32(grad), F32(outputs), B, D, deg, C, F32(grad_inputs)); }
    catch (const std::exception& e) { fprintf(stderr, "freq_encode_backward: %s\n", e.what()); return 1; }
    return 0;
}
