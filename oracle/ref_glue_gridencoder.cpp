/* ref_glue_gridencoder.cpp -- TEST INFRASTRUCTURE.  Appended by oracle/Makefile to the (piped, never stored) text of the reference's
 * gridencoder/src/gridencoder.cu: plain-pointer C entry points over the reference's own launchers (gridencoder.h:12-15).
 * dtype: 0 = float, 1 = half (c10::Half storage), as include/ngp_hip.h. */
#define ORC_EXPORT extern "C" __attribute__((visibility("default")))
static at::ScalarType orc_type(int dtype) { return dtype == 1 ? at::ScalarType::Half : at::ScalarType::Float; }
static at::optional<at::Tensor> orc_opt(void* p, at::ScalarType t) { return p ? at::optional<at::Tensor>(at::Tensor(p, t)) : at::optional<at::Tensor>(); }

ORC_EXPORT int ref_grid_encode_forward(const float* inputs, void* embeddings, const int32_t* offsets, void* outputs, uint32_t B, uint32_t D,
                                       uint32_t C, uint32_t L, float S, uint32_t H, void* dy_dx, uint32_t gridtype, int align_corners,
                                       uint32_t interp, int dtype) {
    const at::ScalarType t = orc_type(dtype);
    try {
        grid_encode_forward(at::Tensor((void*)inputs, at::ScalarType::Float), at::Tensor(embeddings, t), at::Tensor((void*)offsets, at::ScalarType::Int),
                            at::Tensor(outputs, t), B, D, C, L, S, H, orc_opt(dy_dx, t), gridtype, align_corners != 0, interp);
    } catch (const std::exception& e) { fprintf(stderr, "ref_grid_encode_forward: %s\n", e.what()); return 1; }
    return 0;
}

ORC_EXPORT int ref_grid_encode_backward(void* grad, const float* inputs, void* embeddings, const int32_t* offsets, void* grad_embeddings,
                                        uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, void* dy_dx, void* grad_inputs,
                                        uint32_t gridtype, int align_corners, uint32_t interp, int dtype) {
    const at::ScalarType t = orc_type(dtype);
    try {
        grid_encode_backward(at::Tensor(grad, t), at::Tensor((void*)inputs, at::ScalarType::Float), at::Tensor(embeddings, t),
                             at::Tensor((void*)offsets, at::ScalarType::Int), at::Tensor(grad_embeddings, t), B, D, C, L, S, H, orc_opt(dy_dx, t),
                             orc_opt(grad_inputs, t), gridtype, align_corners != 0, interp);
    } catch (const std::exception& e) { fprintf(stderr, "ref_grid_encode_backward: %s\n", e.what()); return 1; }
    return 0;
}

ORC_EXPORT int ref_grad_total_variation(void* inputs, void* embeddings, void* grad, const int32_t* offsets, float weight, uint32_t B, uint32_t D,
                                        uint32_t C, uint32_t L, float S, uint32_t H, uint32_t gridtype, int align_corners, int dtype) {
    const at::ScalarType t = orc_type(dtype);
    try {
        grad_total_variation(at::Tensor(inputs, t), at::Tensor(embeddings, t), at::Tensor(grad, t), at::Tensor((void*)offsets, at::ScalarType::Int),
                             weight, B, D, C, L, S, H, gridtype, align_corners != 0);
    } catch (const std::exception& e) { fprintf(stderr, "ref_grad_total_variation: %s\n", e.what()); return 1; }
    return 0;
}

/* the index helpers on their own (gridencoder.cu:50-84), D = 2 and 3: entry index BEFORE "* C + ch" */
ORC_EXPORT void ref_grid_index(uint32_t D, uint32_t gridtype, int align_corners, uint32_t hashmap_size, uint32_t resolution,
                               const uint32_t* pos_grid, uint32_t n, uint32_t* out) {
    for (uint32_t i = 0; i < n; i++) {
        if (D == 2) out[i] = get_grid_index<2, 1>(gridtype, align_corners != 0, 0, hashmap_size, resolution, pos_grid + 2 * (size_t)i);
        else out[i] = get_grid_index<3, 1>(gridtype, align_corners != 0, 0, hashmap_size, resolution, pos_grid + 3 * (size_t)i);
    }
}
ORC_EXPORT void ref_fast_hash3(const uint32_t* pos_grid, uint32_t n, uint32_t* out) {
    for (uint32_t i = 0; i < n; i++) out[i] = fast_hash<3>(pos_grid + 3 * (size_t)i);
}
