// Base class of the render passes (include/renderpass.hpp:8-22 in the reference).  The reference's
// execute(vkrenderer&, VkCommandBuffer) records into a command buffer; here the second argument is the HIP stream the pass is
// enqueued on (a hipStream_t as void*), nullptr = the renderer's current stream.
#pragma once

class hiprenderer;

class Renderpass {
public:
    virtual ~Renderpass() = default;
    virtual void execute(hiprenderer& renderer, void* stream = nullptr) = 0;
};
