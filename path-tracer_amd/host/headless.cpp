// Headless driver: the reference's main() (src/main.cpp:202-437) without window, swapchain and UI.
// Builds the scene, wires the compute render pass exactly as src/main.cpp:261-270 does, renders `spp`
// frames with the per-frame protocol of src/main.cpp:406-416 and writes the accumulated image.
//
//   ptr_headless --scene file.{ptrscene,gltf,glb} [--apply-transforms | --instanced] [--gpu-bvh] [--copies NX NZ] --width W --height H --spp N [--max-bounce B] [--min-bounce B]
//                [--dof 0|1] [--downscale D] [--debug-bvh 0|1] [--pipeline compute|compute.megakernel|compute.wavefront]
//                [--pos x y z] [--target x y z] [--fov deg] [--aperture a] [--focus f] [--out image.pfm|image.ppm|image.png]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

#include "compute-renderpass.hpp"
#include "gltf.hpp"
#include "hip-renderer.hpp"
#include "image.hpp"
#include "scene.hpp"

namespace {
float clampf(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }

void write_image(const std::string& path, const std::vector<float>& rgba, uint32_t w, uint32_t h) {
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) { std::fprintf(stderr, "cannot write %s\n", path.c_str()); std::exit(2); }
    const auto quantise = [](float c) { return (unsigned)(256.0 * clampf(c, 0.f, 0.999f)); };  // write_color(), src/color.cpp:6-14
    if (path.size() > 4 && path.substr(path.size() - 4) == ".png") {
        // the same 8-bit values as the .ppm, in a container image viewers open
        std::vector<uint8_t> rgb((size_t)w * h * 3);
        for (size_t i = 0; i < (size_t)w * h; i++)
            for (int k = 0; k < 3; k++) rgb[i * 3 + k] = (uint8_t)quantise(rgba[i * 4 + k]);
        const std::vector<uint8_t> file = encode_png_rgb8(rgb.data(), w, h);
        std::fwrite(file.data(), 1, file.size(), f);
    } else if (path.size() > 4 && path.substr(path.size() - 4) == ".ppm") {
        // the reference's write_color(): 256 * clamp(c, 0, 0.999) (src/color.cpp:6-14)
        std::fprintf(f, "P3\n%u\n%u\n255\n", w, h);
        for (size_t i = 0; i < (size_t)w * h; i++)
            std::fprintf(f, "%u %u %u\n", (unsigned)(256.0 * clampf(rgba[i * 4], 0.f, 0.999f)), (unsigned)(256.0 * clampf(rgba[i * 4 + 1], 0.f, 0.999f)),
                         (unsigned)(256.0 * clampf(rgba[i * 4 + 2], 0.f, 0.999f)));
    } else {  // raw float RGBA, row 0 first: "PTRF w h\n" + floats
        std::fprintf(f, "PTRF %u %u\n", w, h);
        std::fwrite(rgba.data(), sizeof(float), rgba.size(), f);
    }
    std::fclose(f);
}
// .gltf / .glb go through the glTF loader, anything else is read as a .ptrscene file
scene_source load_source(const std::string& path, bool apply_transforms, bool instanced) {
    const size_t dot = path.find_last_of('.');
    const std::string ext = dot == std::string::npos ? std::string() : path.substr(dot);
    if ((ext == ".gltf" || ext == ".glb") && instanced) return gltf(path).flatten_instanced();
    if (ext == ".gltf" || ext == ".glb") return gltf(path).flatten(apply_transforms);
    return scene_source::load(path);
}
}  // namespace

int main(int argc, char** argv) {
    std::string scene_path, out_path, pipeline = "compute";
    bool apply_transforms = false, gpu_bvh = false, instanced = false;
    uint32_t copies_x = 1, copies_z = 1;
    uint32_t width = 400, height = 225, spp = 1, max_bounce = 3, min_bounce = 1, dof = 0, debug_bvh = 0;
    int32_t downscale = 1;
    float pos[3] = { 13.f, 2.f, -3.f }, target[3] = { 0.f, 0.f, 0.f }, v_fov = 90.f, aperture = 0.1f, focus = 10.f;  // src/main.cpp:219-223
    for (int i = 1; i < argc; i++) {
        auto is = [&](const char* name) { return std::strcmp(argv[i], name) == 0; };
        auto next = [&]() -> const char* { if (i + 1 >= argc) { std::fprintf(stderr, "%s needs a value\n", argv[i]); std::exit(2); } return argv[++i]; };
        if (is("--scene")) scene_path = next();
        else if (is("--apply-transforms")) apply_transforms = true;
        else if (is("--gpu-bvh")) gpu_bvh = true;
        else if (is("--instanced")) instanced = true;
        else if (is("--copies")) { copies_x = (uint32_t)std::atoi(next()); copies_z = (uint32_t)std::atoi(next()); }
        else if (is("--out")) out_path = next();
        else if (is("--pipeline")) pipeline = next();
        else if (is("--width")) width = (uint32_t)std::atoi(next());
        else if (is("--height")) height = (uint32_t)std::atoi(next());
        else if (is("--spp")) spp = (uint32_t)std::atoi(next());
        else if (is("--max-bounce")) max_bounce = (uint32_t)std::atoi(next());
        else if (is("--min-bounce")) min_bounce = (uint32_t)std::atoi(next());
        else if (is("--dof")) dof = (uint32_t)std::atoi(next());
        else if (is("--debug-bvh")) debug_bvh = (uint32_t)std::atoi(next());
        else if (is("--downscale")) downscale = std::atoi(next());
        else if (is("--fov")) v_fov = (float)std::atof(next());
        else if (is("--aperture")) aperture = (float)std::atof(next());
        else if (is("--focus")) focus = (float)std::atof(next());
        else if (is("--pos")) { for (float& c : pos) c = (float)std::atof(next()); }
        else if (is("--target")) { for (float& c : target) c = (float)std::atof(next()); }
        else { std::fprintf(stderr, "unknown argument %s\n", argv[i]); return 2; }
    }
    if (scene_path.empty()) { std::fprintf(stderr, "usage: ptr_headless --scene file.{ptrscene,gltf,glb} [options]\n"); return 2; }

    try {
        hiprenderer renderer(0);
        const float aspect_ratio = (float)width / (float)height;
        const auto t_build = std::chrono::steady_clock::now();
        scene main_scene(camera(point3{ pos[0], pos[1], pos[2] }, point3{ target[0], target[1], target[2] }, v_fov, aspect_ratio, aperture, focus), width, height,
                         load_source(scene_path, apply_transforms, instanced), pipeline != "compute.megakernel", gpu_bvh);
        main_scene.meta.max_bounce = max_bounce; main_scene.meta.min_bounce = min_bounce;
        main_scene.meta.enable_dof = dof; main_scene.meta.debug_bvh = debug_bvh; main_scene.meta.downscale_factor = downscale;
        if (copies_x * copies_z > 1) main_scene.place_copies_on_grid(copies_x, copies_z);  // the scene repeated on a grid, through a top-level tree
        const double build_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_build).count();

        ComputeRenderpass* raytracing_pass = renderer.create_compute_renderpass();
        raytracing_pass->set_pipeline(pipeline.c_str());
        Texture* accumulation_texture = hiprenderer::create_2d_texture(width, height, PTR_FORMAT_RGBA32_SFLOAT);
        Texture* output_texture = hiprenderer::create_2d_texture(width, height, PTR_FORMAT_RGBA32_SFLOAT);
        raytracing_pass->set_constant(0, main_scene.scene_buffer);
        raytracing_pass->set_constant(8, main_scene.bvh_buffer);
        raytracing_pass->set_constant(16, main_scene.indices_buffer);
        raytracing_pass->set_constant(24, main_scene.positions_buffer);
        raytracing_pass->set_constant(32, main_scene.normals_buffer);
        raytracing_pass->set_constant(40, main_scene.uvs_buffer);
        raytracing_pass->set_constant(48, main_scene.materials_buffer);
        raytracing_pass->set_dispatch_size(width / 8 + 1, height / 8 + 1, 1);

        const auto t0 = std::chrono::steady_clock::now();
        for (uint32_t frame = 0; frame < spp; frame++) {
            raytracing_pass->set_ouput_texture(output_texture);
            raytracing_pass->set_constant(60, accumulation_texture);
            renderer.begin_frame();
            main_scene.upload_metadata();
            renderer.render();
            main_scene.meta.sample_index++;
            std::swap(output_texture, accumulation_texture);
            renderer.finish_frame();
        }
        renderer.wait_idle();
        const double render_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::fprintf(stderr, "ptr_headless: %zu triangles x %zu (%s BVH), scene build %.2f s, %u frames %ux%u in %.3f s (%.3f ms/frame)\n",
                     main_scene.triangle_count, main_scene.copy_count, main_scene.bvh_mode.c_str(), build_s, spp, width, height, render_s, render_s * 1e3 / (spp ? spp : 1));

        if (!out_path.empty()) {
            std::vector<float> rgba((size_t)width * height * 4);
            accumulation_texture->read(rgba.data());  // after the swap the last written image is `accumulation_texture`
            write_image(out_path, rgba, width, height);
        }
        delete accumulation_texture;
        delete output_texture;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "ptr_headless: %s\n", e.what());
        return 1;
    }
    return 0;
}
