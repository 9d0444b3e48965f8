#include "hip-renderer.hpp"

#include "compute-renderpass.hpp"

ptr_ctx* hiprenderer::api = nullptr;

void hiprenderer::check(int status, const char* what) {
    if (status != PTR_OK) throw renderer_error(std::string(what) + ": " + ptr_last_error(api));
}

hiprenderer::hiprenderer(int device) {
    if (api) throw renderer_error("hiprenderer: one renderer per process (the context is a singleton, as in the reference)");
    if (ptr_ctx_create(device, &api) != PTR_OK) throw renderer_error(std::string("ptr_ctx_create: ") + ptr_last_error(nullptr));
}

hiprenderer::~hiprenderer() {
    for (auto* pass : renderpasses) delete pass;
    if (api) ptr_ctx_destroy(api);
    api = nullptr;
}

void hiprenderer::begin_frame() {}
void hiprenderer::render() { for (auto* pass : renderpasses) pass->execute(*this); }
void hiprenderer::finish_frame() { virtual_frame_index = (virtual_frame_index + 1) % virtual_frames_count; }
void hiprenderer::wait_idle() { check(ptr_sync(api), "ptr_sync"); }

Buffer* hiprenderer::create_buffer(size_t size) { return new Buffer(size); }
Texture* hiprenderer::create_2d_texture(size_t width, size_t height, int format, Sampler* sampler) { return new Texture(width, height, 1, format, sampler); }
Sampler* hiprenderer::create_sampler(int filter, int address_mode) {
    auto* s = new Sampler();
    check(ptr_sampler_create(api, filter, address_mode, &s->device_sampler), "ptr_sampler_create");
    return s;
}
ComputeRenderpass* hiprenderer::create_compute_renderpass() {
    auto* pass = new ComputeRenderpass();
    renderpasses.push_back(pass);
    return pass;
}

Buffer::Buffer(size_t size) : buffer_size(size) { hiprenderer::check(ptr_buffer_create(hiprenderer::api, size, &device_buffer), "ptr_buffer_create"); }
Buffer::~Buffer() { if (hiprenderer::api && device_buffer) ptr_buffer_destroy(hiprenderer::api, device_buffer); }
void Buffer::write(const void* data, off_t alloc_offset, size_t data_size) const {
    hiprenderer::check(ptr_buffer_write(hiprenderer::api, device_buffer, data, (size_t)alloc_offset, data_size), "ptr_buffer_write");
}
void Buffer::read(void* dst, off_t offset, size_t size) const {
    hiprenderer::check(ptr_buffer_read(hiprenderer::api, device_buffer, dst, (size_t)offset, size), "ptr_buffer_read");
}
uint64_t Buffer::device_address() const {
    uint64_t a = 0;
    hiprenderer::check(ptr_buffer_address(hiprenderer::api, device_buffer, &a), "ptr_buffer_address");
    return a;
}

Texture::Texture(size_t w, size_t h, size_t d, int fmt, Sampler* texture_sampler) : sampler(texture_sampler), width(w), height(h), depth(d), format(fmt) {
    hiprenderer::check(ptr_image_create(hiprenderer::api, (uint32_t)w, (uint32_t)h, fmt, texture_sampler ? texture_sampler->device_sampler : 0, &device_image),
                       "ptr_image_create");
}
Texture::~Texture() { if (hiprenderer::api && device_image) ptr_image_destroy(hiprenderer::api, device_image); }
void Texture::update(const void* new_data) { hiprenderer::check(ptr_image_upload(hiprenderer::api, device_image, new_data), "ptr_image_upload"); }
void Texture::read(void* dst) const { hiprenderer::check(ptr_image_read(hiprenderer::api, device_image, dst), "ptr_image_read"); }
void Texture::read_unorm8(void* dst) const { hiprenderer::check(ptr_image_read_unorm8(hiprenderer::api, device_image, dst), "ptr_image_read_unorm8"); }
uint32_t Texture::bindless_index() const {
    uint32_t i = 0;
    hiprenderer::check(ptr_image_index(hiprenderer::api, device_image, &i), "ptr_image_index");
    return i;
}
