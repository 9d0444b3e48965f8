// Compute render pass with the reference's surface (include/compute-renderpass.hpp:9-41): a 64-byte
// constant block, a dispatch size, an output texture; execute() hands them to ptr_dispatch.
#pragma once

#include <cstddef>
#include <cstdint>
#include <string>

#include "renderpass.hpp"

class Texture;
class Buffer;

class ComputeRenderpass : public Renderpass {
public:
    // "compute" or "tonemapping": the reference's shader names select the HIP kernels.
    void set_pipeline(const char* shader_name);
    void set_dispatch_size(size_t count_x, size_t count_y, size_t count_z);
    void set_ouput_texture(Texture* out_texture);  // spelling kept from the reference

    void set_constant(off_t offset, uint64_t* constant);
    void set_constant(off_t offset, Texture* texture);
    void set_constant(off_t offset, Buffer* buffer);

    // include/compute-renderpass.hpp:26 execute(vkrenderer&, VkCommandBuffer): the command buffer becomes a hipStream_t.  A non-null
    // stream becomes the context's stream (ptr_ctx_set_stream) and stays it until another one is named.
    void execute(hiprenderer& renderer, void* stream = nullptr) final;

    Texture* output_texture = nullptr;

private:
    uint8_t constants[64] = {};
    size_t group_count_x = 0, group_count_y = 0, group_count_z = 0;
    std::string pipeline;
};
