// The renderer the scene / renderpass code talks to: same static factory names and the same
// Buffer / Texture / Sampler shapes as the reference's vkrenderer (include/vk-renderer.hpp:23-228),
// implemented over the C ABI of libptr_hip.so instead of Vulkan.  Headless: no window, no swapchain;
// "presenting" is reading an image back.
#pragma once

#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "ptr_hip.h"

class Renderpass;
class ComputeRenderpass;
class Sampler;

class renderer_error : public std::runtime_error {
public:
    using std::runtime_error::runtime_error;
};

class Buffer {
public:
    explicit Buffer(size_t size);
    Buffer(ptr_handle adopted, size_t size) : device_buffer(adopted), buffer_size(size) {}  // a buffer the library created
    ~Buffer();
    void write(const void* data, off_t alloc_offset, size_t data_size) const;
    void read(void* dst, off_t offset, size_t size) const;
    uint64_t device_address() const;

    ptr_handle device_buffer = 0;
    size_t buffer_size = 0;
};

class Sampler {
public:
    ptr_handle device_sampler = 0;
};

class Texture {
public:
    Texture(size_t width, size_t height, size_t depth, int format = PTR_FORMAT_RGBA8_UNORM, Sampler* texture_sampler = nullptr);
    ~Texture();
    // Uploads width*height texels (the reference's Texture::update queues the same copy, include/vk-renderer.hpp:188-195).
    // Ownership of `new_data` stays with the caller.
    void update(const void* new_data);
    void read(void* dst) const;
    // width*height RGBA8 words: what the reference's present blit shows of an RGBA32F image (vkapi::blit_full, src/vk-api.cpp:1194-1224)
    void read_unorm8(void* dst) const;
    size_t size() const { return width * height * depth * pixel_size(format); }
    static size_t pixel_size(int format) { return format == PTR_FORMAT_RGBA8_UNORM ? 4 : (format == PTR_FORMAT_RGBA32_SFLOAT ? 16 : 0); }
    uint32_t bindless_index() const;

    Sampler* sampler = nullptr;
    size_t width = 0, height = 0, depth = 0;
    int format = 0;
    ptr_handle device_image = 0;
};

class hiprenderer {
public:
    explicit hiprenderer(int device = 0);
    ~hiprenderer();

    void begin_frame();   // nothing to wait for: the stream orders the frames
    void render();        // runs every render pass in creation order
    void finish_frame();  // advances the virtual frame index
    void wait_idle();

    static Buffer* create_buffer(size_t size);
    static Texture* create_2d_texture(size_t width, size_t height, int format, Sampler* sampler = nullptr);
    static Sampler* create_sampler(int filter, int address_mode);

    ComputeRenderpass* create_compute_renderpass();  // owned by the renderer, like the reference's

    uint32_t frame_index() const { return virtual_frame_index; }
    static constexpr uint32_t virtual_frames_count = 2;

    static ptr_ctx* api;  // the one context, like the reference's static vkrenderer::api

    static void check(int status, const char* what);

private:
    std::vector<Renderpass*> renderpasses;
    uint32_t virtual_frame_index = 0;
};
