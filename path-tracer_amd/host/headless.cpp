// Headless driver: the reference's main() (src/main.cpp:202-437) without window, swapchain and UI.
// Builds the scene, wires the compute render pass exactly as src/main.cpp:261-270 does, renders `spp`
// frames with the per-frame protocol of src/main.cpp:406-416 and writes the accumulated image.
//
//   ptr_headless --scene file.{ptrscene,gltf,glb} [--apply-transforms | --instanced] [--gpu-bvh] [--copies NX NZ] --width W --height H --spp N [--max-bounce B] [--min-bounce B]
//                [--dof 0|1] [--downscale D] [--debug-bvh 0|1] [--pipeline compute|compute.megakernel|compute.wavefront]
//                [--pos x y z] [--target x y z] [--fov deg] [--aperture a] [--focus f] [--out image.pfm|image.ppm|image.png]
//                [--samples-per-dispatch S]   (1..64: one dispatch renders S consecutive samples, same image, fewer launches)
//                [--coalesce N]   (the loop stays one dispatch per sample; the library renders N of them at a time)
//                [--checkpoint FILE [--checkpoint-every N]] [--resume FILE]   (progressive render state: accumulated image + sample count)
//                [--gpus N [--partition tiles|samples] [--transport rccl|file] [--stripe-rows S] [--rendezvous DIR]]
//
// --gpus N (no counterpart in the single-device reference): the process starts N copies of itself, one per GPU, each with
// --rank r; the scene is replicated.  tiles: every rank renders the stripes s % N == r of all `spp` frames, one all-gather
// merges the frame (bit-identical to one GPU).  samples: every rank renders spp / N whole frames of its own samples, one
// all-reduce of the linearised images combines them.  Rank 0 writes the image.  --transport file stages the exchange through
// files (several ranks on one GPU, for rehearsal); rccl uses the GPUs' xGMI links.
#include <dirent.h>
#include <sys/wait.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

#include "compute-renderpass.hpp"
#include "exchange.hpp"
#include "gltf.hpp"
#include "hip-renderer.hpp"
#include "image.hpp"
#include "scene.hpp"

namespace {
float clampf(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }

// A checkpoint of a progressive render: "PTRC w h samples\n" + the accumulated RGBA32F image as the next frame reads it.
// Written to a temporary name first, so that a run killed while writing leaves the previous checkpoint intact.
void write_checkpoint(const std::string& path, const std::vector<float>& rgba, uint32_t w, uint32_t h, uint32_t samples) {
    const std::string tmp = path + ".part";
    FILE* f = std::fopen(tmp.c_str(), "wb");
    if (!f) throw std::runtime_error("cannot write " + tmp);
    std::fprintf(f, "PTRC %u %u %u\n", w, h, samples);
    const bool ok = std::fwrite(rgba.data(), sizeof(float), rgba.size(), f) == rgba.size();
    if (std::fclose(f) != 0 || !ok || std::rename(tmp.c_str(), path.c_str()) != 0) throw std::runtime_error("cannot write " + path);
}
uint32_t read_checkpoint(const std::string& path, std::vector<float>& rgba, uint32_t w, uint32_t h) {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error("cannot read " + path);
    unsigned fw = 0, fh = 0, samples = 0;
    char tag[8] = {};
    const bool header = std::fscanf(f, "%4s %u %u %u", tag, &fw, &fh, &samples) == 4 && std::fgetc(f) == '\n' && std::strcmp(tag, "PTRC") == 0;
    rgba.resize((size_t)w * h * 4);
    const bool ok = header && fw == w && fh == h && std::fread(rgba.data(), sizeof(float), rgba.size(), f) == rgba.size();
    std::fclose(f);
    if (!header) throw std::runtime_error(path + ": not a checkpoint");
    if (!ok) throw std::runtime_error(path + ": a checkpoint of " + std::to_string(fw) + " x " + std::to_string(fh) + ", the render is " + std::to_string(w) + " x " + std::to_string(h));
    return samples;
}

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
// --gpus N without --rank: one child process per GPU, the worst exit code wins.  The children are this program again (fork +
// exec of /proc/self/exe), which is only allowed from a process that has not initialised the GPU: main() calls this before its
// first HIP call, and it refuses to start under a profiler, whose preloaded library initialises the GPU before main() runs.
int launch_ranks(int argc, char** argv, uint32_t gpus, std::string rendezvous) {
    for (const char* name : { "ROCP_TOOL_LIBRARIES", "ROCPROFILER_REGISTER_FORCE_LOAD", "HSA_TOOLS_LIB" }) {
        const char* v = std::getenv(name);
        if (v && *v) {
            std::fprintf(stderr, "ptr_headless --gpus %u: not under a profiler (%s is set): its library initialises the GPU before this process "
                                 "starts the ranks.  Profile one rank instead: --rank R --gpus %u --rendezvous DIR.\n", gpus, name, gpus);
            return 2;
        }
    }
    const bool own_directory = rendezvous.empty();
    if (rendezvous.empty()) {
        char tmpl[] = "/tmp/ptr_headless_XXXXXX";
        if (!mkdtemp(tmpl)) { std::perror("mkdtemp"); return 2; }
        rendezvous = tmpl;
    }
    std::vector<pid_t> children;
    for (uint32_t r = 0; r < gpus; r++) {
        const pid_t pid = fork();
        if (pid < 0) { std::perror("fork"); return 2; }
        if (pid == 0) {
            std::vector<std::string> args(argv, argv + argc);
            args.push_back("--rank"); args.push_back(std::to_string(r));
            args.push_back("--rendezvous"); args.push_back(rendezvous);
            std::vector<char*> raw;
            for (auto& a : args) raw.push_back(&a[0]);
            raw.push_back(nullptr);
            execv("/proc/self/exe", raw.data());
            std::perror("execv");
            _exit(127);
        }
        children.push_back(pid);
    }
    int worst = 0;
    for (pid_t pid : children) {
        int status = 0;
        if (waitpid(pid, &status, 0) < 0 || !WIFEXITED(status) || WEXITSTATUS(status) != 0) worst = WIFEXITED(status) && WEXITSTATUS(status) ? WEXITSTATUS(status) : 1;
    }
    if (own_directory) {   // the exchange's files (communicator id, staged chunks) go with the directory made for them
        if (DIR* d = opendir(rendezvous.c_str())) {
            while (const dirent* e = readdir(d))
                if (std::strcmp(e->d_name, ".") != 0 && std::strcmp(e->d_name, "..") != 0) std::remove((rendezvous + "/" + e->d_name).c_str());
            closedir(d);
        }
        rmdir(rendezvous.c_str());
    }
    return worst;
}
}  // namespace

const char kUsage[] =
    "usage: ptr_headless --scene file.{ptrscene,gltf,glb} | --spheres file.txt | --random-spheres SEED [options]\n"
    "  spheres   --spheres FILE (one per line: x y z radius r g b metalness roughness)   --random-spheres SEED (the reference's old scene of spheres)\n"
    "            alone or beside --scene: they become sphere leaves of the same tree\n"
    "  scene     --apply-transforms | --instanced (glTF node transforms: applied to the geometry | kept as placed copies)\n"
    "            --copies NX NZ (the scene placed on a grid through a top-level tree)   --gpu-bvh (tree built on the GPU)\n"
    "  frame     --width W --height H --spp N --max-bounce N --min-bounce N --downscale N --dof 0|1 --debug-bvh 0|1\n"
    "            --samples-per-dispatch N (1..64 samples per dispatch; same image)   --pipeline compute|compute.megakernel|compute.wavefront\n"
    "            --coalesce N (one dispatch per sample as the reference issues them, rendered N at a time by the library; same image)\n"
    "  camera    --pos X Y Z --target X Y Z --fov DEGREES --aperture A --focus DISTANCE\n"
    "  progress  --checkpoint FILE [--checkpoint-every N] (the accumulated float image + its sample count, written every N samples and at the end)\n"
    "            --resume FILE (continue such a render up to --spp samples: same image as an uninterrupted run)\n"
    "  output    --out file.ppm | file.png (8 bits per channel, 256 * clamp(c, 0, 0.999)) | any other name (\"PTRF w h\\n\" + the float RGBA image)\n"
    "  GPUs      --gpus N (one process per GPU is started)   --partition tiles|samples   --stripe-rows N (a multiple of 8)\n"
    "            --transport rccl|file   --rendezvous DIR   [--rank R --device D: set for the started processes]\n"
    "            (do not profile the --gpus N launcher: profile one rank, --rank R --gpus N --rendezvous DIR)\n";

int main(int argc, char** argv) {
    std::string scene_path, spheres_path, out_path, pipeline = "compute", checkpoint_path, resume_path;
    long random_spheres_seed = -1;
    uint32_t checkpoint_every = 0;
    bool apply_transforms = false, gpu_bvh = false, instanced = false;
    uint32_t copies_x = 1, copies_z = 1;
    uint32_t gpus = 1, stripe_rows = 8, samples_per_dispatch = 1, coalesce = 1;
    int rank = -1, device = -1;
    bool transport_given = false;   // with --gpus 1: run the exchange anyway, alone (rehearses the RCCL calls on a one-GPU box)
    std::string partition = "tiles", transport = "rccl", rendezvous;
    uint32_t width = 400, height = 225, spp = 1, max_bounce = 3, min_bounce = 1, dof = 0, debug_bvh = 0;
    int32_t downscale = 1;
    float pos[3] = { 13.f, 2.f, -3.f }, target[3] = { 0.f, 0.f, 0.f }, v_fov = 90.f, aperture = 0.1f, focus = 10.f;  // src/main.cpp:219-223
    for (int i = 1; i < argc; i++) {
        auto is = [&](const char* name) { return std::strcmp(argv[i], name) == 0; };
        auto next = [&]() -> const char* { if (i + 1 >= argc) { std::fprintf(stderr, "%s needs a value\n", argv[i]); std::exit(2); } return argv[++i]; };
        if (is("--scene")) scene_path = next();
        else if (is("--spheres")) spheres_path = next();
        else if (is("--random-spheres")) random_spheres_seed = std::atol(next());
        else if (is("--apply-transforms")) apply_transforms = true;
        else if (is("--gpu-bvh")) gpu_bvh = true;
        else if (is("--instanced")) instanced = true;
        else if (is("--copies")) { copies_x = (uint32_t)std::atoi(next()); copies_z = (uint32_t)std::atoi(next()); }
        else if (is("--gpus")) gpus = (uint32_t)std::atoi(next());
        else if (is("--rank")) rank = std::atoi(next());
        else if (is("--device")) device = std::atoi(next());
        else if (is("--partition")) partition = next();
        else if (is("--transport")) { transport = next(); transport_given = true; }
        else if (is("--rendezvous")) rendezvous = next();
        else if (is("--stripe-rows")) stripe_rows = (uint32_t)std::atoi(next());
        else if (is("--samples-per-dispatch")) samples_per_dispatch = (uint32_t)std::atoi(next());
        else if (is("--coalesce")) coalesce = (uint32_t)std::atoi(next());
        else if (is("--out")) out_path = next();
        else if (is("--checkpoint")) checkpoint_path = next();
        else if (is("--checkpoint-every")) checkpoint_every = (uint32_t)std::atoi(next());
        else if (is("--resume")) resume_path = next();
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
        else if (is("--help") || is("-h")) { std::fputs(kUsage, stdout); return 0; }
        else { std::fprintf(stderr, "unknown argument %s\n%s", argv[i], kUsage); return 2; }
    }
    if (scene_path.empty() && spheres_path.empty() && random_spheres_seed < 0) { std::fputs(kUsage, stderr); return 2; }

    if (gpus == 0 || (partition != "tiles" && partition != "samples")) { std::fprintf(stderr, "--gpus N >= 1, --partition tiles|samples\n"); return 2; }
    // (Nothing above this line may call HIP -- no hiprenderer, no device query: launch_ranks() starts this program again.)
    if (gpus > 1 && rank < 0) return launch_ranks(argc, argv, gpus, rendezvous);
    if (gpus > 1 && (uint32_t)rank >= gpus) { std::fprintf(stderr, "--rank %d of %u\n", rank, gpus); return 2; }
    if (gpus == 1) rank = 0;
    const bool tiles = partition == "tiles";
    if (gpus > 1 && !tiles && spp % gpus != 0) { std::fprintf(stderr, "--partition samples: --spp must be a multiple of --gpus\n"); return 2; }
    if (gpus > 1 && !tiles && (!checkpoint_path.empty() || !resume_path.empty())) { std::fprintf(stderr, "--checkpoint / --resume: not with --partition samples (every GPU keeps a mean of its own)\n"); return 2; }
    if (gpus > 1 && checkpoint_every) { std::fprintf(stderr, "--checkpoint-every: one GPU only (with --gpus N the checkpoint is written once, after the exchange)\n"); return 2; }
    if (device < 0) device = (gpus > 1 && transport == "rccl") ? rank : 0;

    try {
        hiprenderer renderer(device);
        std::unique_ptr<device_exchange> exchange;
        if (gpus > 1 || transport_given) {
            if (rendezvous.empty()) { char tmpl[] = "/tmp/ptr_headless_XXXXXX"; if (!mkdtemp(tmpl)) throw std::runtime_error("mkdtemp failed"); rendezvous = tmpl; }
            exchange = device_exchange::create(transport, (uint32_t)rank, gpus, rendezvous, device);
            if (tiles) hiprenderer::check(ptr_set_partition(hiprenderer::api, (uint32_t)rank, gpus, stripe_rows), "ptr_set_partition");
            else hiprenderer::check(ptr_set_sample_partition(hiprenderer::api, (uint32_t)rank, gpus), "ptr_set_sample_partition");
        }
        uint32_t frames = (gpus > 1 && !tiles) ? spp / gpus : spp;
        const float aspect_ratio = (float)width / (float)height;
        const auto t_build = std::chrono::steady_clock::now();
        scene_source source = scene_path.empty() ? scene_source() : load_source(scene_path, apply_transforms, instanced);
        if (!spheres_path.empty()) source.spheres = scene_source::load_spheres(spheres_path);
        if (random_spheres_seed >= 0) {
            const auto more = scene_source::random_spheres((uint32_t)random_spheres_seed);
            source.spheres.insert(source.spheres.end(), more.begin(), more.end());
        }
        scene main_scene(camera(point3{ pos[0], pos[1], pos[2] }, point3{ target[0], target[1], target[2] }, v_fov, aspect_ratio, aperture, focus), width, height,
                         source, pipeline != "compute.megakernel", gpu_bvh);
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

        // --resume: the accumulated image of an earlier run and the number of samples in it; this run renders the samples up to --spp
        uint32_t samples_done = 0;
        if (!resume_path.empty()) {
            std::vector<float> rgba;
            samples_done = read_checkpoint(resume_path, rgba, width, height);
            if (samples_done > frames) throw std::runtime_error("--resume: the checkpoint already holds " + std::to_string(samples_done) + " samples");
            accumulation_texture->update(rgba.data());
            main_scene.meta.sample_index = samples_done + 1;
            frames -= samples_done;
        }
        auto checkpoint = [&](uint32_t samples) {   // after the swap the image written last is `accumulation_texture`
            std::vector<float> rgba((size_t)width * height * 4);
            accumulation_texture->read(rgba.data());
            write_checkpoint(checkpoint_path, rgba, width, height, samples);
        };
        uint32_t since_checkpoint = 0;

        const auto t0 = std::chrono::steady_clock::now();
        if (samples_per_dispatch == 0 || samples_per_dispatch > 64) throw std::runtime_error("--samples-per-dispatch: 1..64");
        if (coalesce == 0 || coalesce > 64) throw std::runtime_error("--coalesce: 1..64");
        // the loop below stays the reference's (one dispatch and one image swap per sample); the library holds the
        // dispatches back and renders `coalesce` of them at a time
        hiprenderer::check(ptr_set_option(hiprenderer::api, "dispatch.coalesce", (int)coalesce), "ptr_set_option");
        for (uint32_t frame = 0; frame < frames;) {
            // one dispatch = the next `batch` samples, accumulated in order; the images swap once per dispatch
            const uint32_t batch = std::min(samples_per_dispatch, frames - frame);
            hiprenderer::check(ptr_set_option(hiprenderer::api, "dispatch.samples", (int)batch), "ptr_set_option");
            raytracing_pass->set_ouput_texture(output_texture);
            raytracing_pass->set_constant(60, accumulation_texture);
            renderer.begin_frame();
            main_scene.upload_metadata();
            renderer.render();
            main_scene.meta.sample_index += batch;
            frame += batch;
            std::swap(output_texture, accumulation_texture);
            renderer.finish_frame();
            since_checkpoint += batch;
            if (checkpoint_every && !checkpoint_path.empty() && since_checkpoint >= checkpoint_every && frame < frames) {
                checkpoint(samples_done + frame);
                since_checkpoint = 0;
            }
        }
        hiprenderer::check(ptr_set_option(hiprenderer::api, "dispatch.samples", 1), "ptr_set_option");
        hiprenderer::check(ptr_set_option(hiprenderer::api, "dispatch.coalesce", 1), "ptr_set_option");   // runs what is still held back
        if (exchange && tiles) {
            // one all-gather of the owned stripes; after the swap the last written image is `accumulation_texture`
            const size_t chunk_bytes = stripe_chunk_texels(width, height, stripe_rows, gpus) * 16;
            Buffer packed(chunk_bytes), gathered(chunk_bytes * gpus);
            hiprenderer::check(ptr_image_pack_stripes(hiprenderer::api, accumulation_texture->device_image, (uint32_t)rank, gpus, stripe_rows, packed.device_buffer), "ptr_image_pack_stripes");
            renderer.wait_idle();
            exchange->all_gather(packed.device_address(), gathered.device_address(), chunk_bytes);
            hiprenderer::check(ptr_image_unpack_stripes(hiprenderer::api, accumulation_texture->device_image, gpus, stripe_rows, gathered.device_buffer), "ptr_image_unpack_stripes");
        } else if (exchange) {
            // linearise this GPU's running mean, sum over the GPUs, resolve (alpha carries the number of contributions)
            uint8_t block[64] = {};
            auto run = [&](const char* kernel, Texture* out, Texture* in) {
                const uint32_t o = out->bindless_index(), a = in->bindless_index();
                std::memcpy(block + 56, &o, 4); std::memcpy(block + 60, &a, 4);
                hiprenderer::check(ptr_dispatch(hiprenderer::api, kernel, block, 1, 1, 1), kernel);
            };
            run("linearize", output_texture, accumulation_texture);
            renderer.wait_idle();
            uint64_t address = 0;
            hiprenderer::check(ptr_image_address(hiprenderer::api, output_texture->device_image, &address), "ptr_image_address");
            exchange->all_reduce_sum(address, (size_t)width * height * 4);
            run("resolve", accumulation_texture, output_texture);
        }
        renderer.wait_idle();
        const double render_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::fprintf(stderr, "ptr_headless%s: %zu triangles x %zu (%s BVH), scene build %.2f s, %u frames %ux%u in %.3f s (%.3f ms/frame)\n",
                     exchange ? (" [rank " + std::to_string(rank) + " of " + std::to_string(gpus) + ", " + partition + " over " + exchange->transport() + "]").c_str() : "",
                     main_scene.triangle_count, main_scene.copy_count, main_scene.bvh_mode.c_str(), build_s, frames, width, height, render_s, render_s * 1e3 / (frames ? frames : 1));

        if (!checkpoint_path.empty() && rank == 0) checkpoint(samples_done + frames);
        if (!out_path.empty() && rank == 0) {
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
