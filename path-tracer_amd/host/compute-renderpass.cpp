#include "compute-renderpass.hpp"

#include <cstring>

#include "hip-renderer.hpp"

void ComputeRenderpass::set_pipeline(const char* shader_name) { pipeline = shader_name; }

void ComputeRenderpass::set_ouput_texture(Texture* out_texture) {
    output_texture = out_texture;
    const uint32_t index = out_texture->bindless_index();
    std::memcpy(constants + 56, &index, sizeof index);  // the shader reads the output image index at byte 56
}

void ComputeRenderpass::set_dispatch_size(size_t count_x, size_t count_y, size_t count_z) {
    group_count_x = count_x; group_count_y = count_y; group_count_z = count_z;
}

void ComputeRenderpass::set_constant(off_t offset, uint64_t* constant) { std::memcpy(constants + offset, constant, sizeof(uint64_t)); }

void ComputeRenderpass::set_constant(off_t offset, Texture* texture) {
    const uint32_t index = texture->bindless_index();
    std::memcpy(constants + offset, &index, sizeof index);
}

void ComputeRenderpass::set_constant(off_t offset, Buffer* buffer) {
    const uint64_t address = buffer->device_address();
    std::memcpy(constants + offset, &address, sizeof address);
}

void ComputeRenderpass::execute(hiprenderer&, void* stream) {
    if (stream != nullptr) hiprenderer::check(ptr_ctx_set_stream(hiprenderer::api, stream), "ptr_ctx_set_stream");
    hiprenderer::check(ptr_dispatch(hiprenderer::api, pipeline.c_str(), constants, (uint32_t)group_count_x, (uint32_t)group_count_y, (uint32_t)group_count_z),
                       "ptr_dispatch");
}
