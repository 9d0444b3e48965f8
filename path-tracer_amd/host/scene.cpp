#include "scene.hpp"

#include <cstdio>
#include <cstring>
#include <stdexcept>

#include "bvh.hpp"
#include "hip-renderer.hpp"
#include "sphere.hpp"

scene::metadata::metadata(const camera& cam_, uint32_t width_, uint32_t height_) : cam(cam_), width(width_), height(height_) {}

scene::scene(const camera& cam, uint32_t width, uint32_t height, const scene_source& src, bool flatten_bvh, bool gpu_bvh) : meta(cam, width, height) {
    const uint32_t iw = src.index_words;
    if (iw != 3 && iw != 4) throw std::runtime_error("scene: index_words must be 3 or 4");
    if ((src.indices.empty() && src.spheres.empty()) || src.indices.size() % iw) throw std::runtime_error("scene: the index array must hold whole triangles");
    if (!src.spheres.empty() && !flatten_bvh) throw std::runtime_error("scene: spheres are leaves of the flattened layout; the threaded tree has no place for them");
    if (!src.spheres.empty() && (!src.placements.empty() || gpu_bvh)) throw std::runtime_error("scene: spheres go with neither placed copies nor the GPU builder");
    const size_t vertex_count = src.positions.size() / 3;
    if (iw == 3 && vertex_count >= (1u << 24)) throw std::runtime_error("scene: more than 2^24 vertices do not fit the packed index words");
    if (iw == 3 && src.materials.size() >= (1u << 24)) throw std::runtime_error("scene: more than 2^24 materials");
    hiprenderer::check(ptr_set_option(hiprenderer::api, "scene.index_words", (int)iw), "ptr_set_option(scene.index_words)");

    // textures + the one LINEAR / REPEAT sampler the reference creates (src/gltf.cpp:118)
    sampler = hiprenderer::create_sampler(PTR_FILTER_LINEAR, PTR_ADDRESS_REPEAT);
    for (const auto& t : src.textures) {
        Texture* tex = hiprenderer::create_2d_texture(t.width, t.height, PTR_FORMAT_RGBA8_UNORM, sampler);
        tex->update(t.rgba.data());
        textures.push_back(tex);
    }
    uint32_t sampler_index = 0;
    hiprenderer::check(ptr_sampler_index(hiprenderer::api, sampler->device_sampler, &sampler_index), "ptr_sampler_index");

    // one gpu_material per primitive with bindless indices (src/scene.cpp:145-163); 0 = no texture = white
    std::vector<gpu_material> materials;
    auto to_gpu = [&](const scene_source::source_material& m) {
        gpu_material g;
        g.base_color = color(m.base_color[0], m.base_color[1], m.base_color[2]);
        g.base_color.v[3] = m.base_color[3];
        auto bind = [&](uint32_t number, uint32_t& tex_id, uint32_t& sampler_id) {
            if (number == 0) return;
            if (number > textures.size()) throw std::runtime_error("scene: material names a texture that does not exist");
            tex_id = textures[number - 1]->bindless_index();
            sampler_id = sampler_index;
        };
        bind(m.albedo_texture, g.albedo_texture_id, g.albedo_texture_sampler_id);
        bind(m.mr_texture, g.metallic_roughness_texture_id, g.metallic_roughness_texture_sampler_id);
        g.metalness = m.metalness;
        g.roughness = m.roughness;
        return g;
    };
    for (const auto& m : src.materials) materials.push_back(to_gpu(m));
    std::vector<gpu_sphere> sphere_records;
    std::vector<flat_bvh_sphere> sphere_geometry;
    for (const auto& s : src.spheres) {
        gpu_sphere g;
        g.position[0] = s.center[0]; g.position[1] = s.center[1]; g.position[2] = s.center[2];
        g.mat = to_gpu(s.material);
        g.radius = s.radius;
        sphere_records.push_back(g);
        sphere_geometry.push_back(flat_bvh_sphere{ { s.center[0], s.center[1], s.center[2] }, s.radius });
    }

    // triangles for the builder: corners fetched through the 24-bit vertex indices (src/scene.cpp:132-136)
    std::vector<triangle> triangles;
    triangles.reserve(src.indices.size() / iw);
    for (size_t i = 0; i < src.indices.size(); i += iw) {
        vec3 p[3];
        for (int k = 0; k < 3; k++) {
            const uint32_t v = iw == 3 ? (src.indices[i + k] & 0x00ffffffu) : src.indices[i + k];
            if (v >= vertex_count) throw std::runtime_error("scene: vertex index out of range");
            p[k] = vec3(src.positions[v * 3], src.positions[v * 3 + 1], src.positions[v * 3 + 2]);
        }
        triangles.emplace_back(p[0], p[1], p[2]);
    }
    triangle_count = triangles.size();

    std::vector<packed_bvh_node> packed_nodes;
    std::vector<int32_t> ranks;
    if (triangles.empty()) {
        // a scene of spheres only: the threaded tree is one node that holds nothing (the kernel that walks it renders the sky)
        packed_bvh_node none{};
        none.next_id = -1;
        none.primitive_id = -1;
        packed_nodes.push_back(none);
        bvh_mode = "no triangles";
    } else
    try {
        bvh builder(triangles, packed_nodes);
        ranks = bvh::visit_ranks(packed_nodes, triangles.size());
        bvh_mode = "reference_order";
    } catch (const bvh_build_error&) {
        // the reference's builder cannot finish on this input; keep the same node format with a real SAH split
        bvh builder(triangles, packed_nodes, bvh::build_mode::sah_partitioned);
        bvh_mode = "sah_partitioned";
    }

    scene_buffer = hiprenderer::create_buffer(sizeof(meta) * hiprenderer::virtual_frames_count);
    upload_metadata();
    scene_buffer->write(&meta, sizeof(meta), sizeof(meta));

    auto upload = [](const void* data, size_t bytes) {
        Buffer* b = hiprenderer::create_buffer(bytes ? bytes : 48);   // (an array a scene of spheres leaves empty: 48 zero bytes, i.e. one record of whatever it holds)
        if (bytes) b->write(data, 0, bytes);
        return b;
    };
    indices_buffer = upload(src.indices.data(), src.indices.size() * sizeof(uint32_t));
    positions_buffer = upload(src.positions.data(), src.positions.size() * sizeof(float));
    normals_buffer = upload(src.normals.data(), src.normals.size() * sizeof(float));
    uvs_buffer = upload(src.uvs.data(), src.uvs.size() * sizeof(float));
    bvh_buffer = upload(packed_nodes.data(), packed_nodes.size() * sizeof(packed_bvh_node));
    materials_buffer = upload(materials.data(), materials.size() * sizeof(gpu_material));

    if (flatten_bvh && !src.placements.empty()) {
        // the scene is placed copies of its objects: one layout with a tree per object, and a top level over the copies
        std::vector<flat_bvh_object> objects;
        const std::vector<uint8_t> flat = bvh::flatten_objects(triangles, ranks, src.object_first, objects);
        flat_bvh_buffer = upload(flat.data(), flat.size());
        hiprenderer::check(ptr_bvh_bind_flat(hiprenderer::api, bvh_buffer->device_buffer, flat_bvh_buffer->device_buffer), "ptr_bvh_bind_flat");
        std::vector<uint32_t> which;
        std::vector<float> transforms;
        for (const auto& p : src.placements) { which.push_back(p.object); transforms.insert(transforms.end(), p.object_to_world, p.object_to_world + 12); }
        const std::vector<uint8_t> top = bvh::flatten_instances(objects, which, transforms);
        instances_buffer = upload(top.data(), top.size());
        hiprenderer::check(ptr_bvh_bind_instances(hiprenderer::api, flat_bvh_buffer->device_buffer, instances_buffer->device_buffer), "ptr_bvh_bind_instances");
        copy_count = src.placements.size();
        bvh_mode += " + placements";
    } else if (flatten_bvh && gpu_bvh) {
        ptr_handle built = 0;
        hiprenderer::check(ptr_bvh_build_flat(hiprenderer::api, indices_buffer->device_buffer, positions_buffer->device_buffer, (uint32_t)triangle_count, &built),
                           "ptr_bvh_build_flat");
        flat_bvh_buffer = new Buffer(built, 64 + 128 * triangle_count);
        // equal-t ties as the reference's walk resolves them: ranks from the leaf order of the threaded nodes
        hiprenderer::check(ptr_bvh_flat_set_reference_ranks(hiprenderer::api, built, bvh_buffer->device_buffer, (uint32_t)triangle_count),
                           "ptr_bvh_flat_set_reference_ranks");
        hiprenderer::check(ptr_bvh_bind_flat(hiprenderer::api, bvh_buffer->device_buffer, flat_bvh_buffer->device_buffer), "ptr_bvh_bind_flat");
        bvh_mode += " + gpu radix tree";
    } else if (flatten_bvh) {
        const std::vector<uint8_t> flat = bvh::flatten(triangles, ranks, 0, sphere_geometry);
        flat_bvh_buffer = upload(flat.data(), flat.size());
        hiprenderer::check(ptr_bvh_bind_flat(hiprenderer::api, bvh_buffer->device_buffer, flat_bvh_buffer->device_buffer), "ptr_bvh_bind_flat");
        if (!sphere_records.empty()) {
            spheres_buffer = upload(sphere_records.data(), sphere_records.size() * sizeof(gpu_sphere));
            sphere_count = sphere_records.size();
            hiprenderer::check(ptr_bvh_bind_spheres(hiprenderer::api, flat_bvh_buffer->device_buffer, spheres_buffer->device_buffer, (uint32_t)sphere_count), "ptr_bvh_bind_spheres");
            bvh_mode += " + spheres";
        }
    }
}

void scene::place_copies(const std::vector<float>& object_to_world) {
    if (!flat_bvh_buffer) throw std::runtime_error("scene: placed copies need the flattened layout");
    if (instances_buffer) {
        hiprenderer::check(ptr_bvh_bind_instances(hiprenderer::api, flat_bvh_buffer->device_buffer, 0), "ptr_bvh_bind_instances");
        delete instances_buffer;
        instances_buffer = nullptr;
        copy_count = 1;
    }
    if (object_to_world.empty()) return;
    // the top level needs the object's header only: its bounds, root and stack need
    std::vector<uint8_t> header(sizeof(flat_bvh_header));
    flat_bvh_buffer->read(header.data(), 0, header.size());
    const std::vector<uint8_t> top = bvh::flatten_instances(header, object_to_world);
    instances_buffer = hiprenderer::create_buffer(top.size());
    instances_buffer->write(top.data(), 0, top.size());
    hiprenderer::check(ptr_bvh_bind_instances(hiprenderer::api, flat_bvh_buffer->device_buffer, instances_buffer->device_buffer), "ptr_bvh_bind_instances");
    copy_count = object_to_world.size() / 12;
}

void scene::place_copies_on_grid(uint32_t nx, uint32_t nz) {
    if (!flat_bvh_buffer) throw std::runtime_error("scene: placed copies need the flattened layout");
    if (nx == 0 || nz == 0) throw std::runtime_error("scene: the grid needs at least one copy each way");
    flat_bvh_header header;
    flat_bvh_buffer->read(&header, 0, sizeof header);
    const float step_x = (header.scene_max[0] - header.scene_min[0]) * 1.02f, step_z = (header.scene_max[2] - header.scene_min[2]) * 1.02f;
    std::vector<float> m;
    m.reserve((size_t)nx * nz * 12);
    for (uint32_t ix = 0; ix < nx; ix++)
        for (uint32_t iz = 0; iz < nz; iz++) {
            const float tx = ((float)ix - (float)(nx - 1) / 2.f) * step_x, tz = ((float)iz - (float)(nz - 1) / 2.f) * step_z;
            const float rows[12] = { 1.f, 0.f, 0.f, tx, 0.f, 1.f, 0.f, 0.f, 0.f, 0.f, 1.f, tz };
            m.insert(m.end(), rows, rows + 12);
        }
    place_copies(m);
}

void scene::upload_metadata() const { scene_buffer->write(&meta, 0, sizeof(meta)); }

scene::~scene() {
    for (Buffer* b : { scene_buffer, indices_buffer, positions_buffer, normals_buffer, uvs_buffer, bvh_buffer, materials_buffer, flat_bvh_buffer, instances_buffer, spheres_buffer }) delete b;
    for (Texture* t : textures) delete t;
    delete sampler;
}
