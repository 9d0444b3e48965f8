// Base class of the render passes (include/renderpass.hpp:8-22 in the reference), with the Vulkan
// command buffer replaced by nothing: work is stream-ordered on the renderer's HIP context.
#pragma once

class hiprenderer;

class Renderpass {
public:
    virtual ~Renderpass() = default;
    virtual void execute(hiprenderer& renderer) = 0;
};
