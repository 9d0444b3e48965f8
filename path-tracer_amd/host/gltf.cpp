#include "gltf.hpp"

#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <queue>

namespace {

std::string parent_directory(const std::string& path) {
    const size_t slash = path.find_last_of("/\\");
    return slash == std::string::npos ? std::string(".") : path.substr(0, slash);
}

std::vector<uint8_t> read_binary_file(const std::string& path) {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) throw gltf_error("gltf: cannot open " + path);
    std::vector<uint8_t> bytes;
    uint8_t buf[65536];
    size_t n;
    while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) bytes.insert(bytes.end(), buf, buf + n);
    std::fclose(f);
    return bytes;
}

// percent-decoding of a relative URI reference (RFC 3986): "%20" -> ' '
std::string uri_to_path(const std::string& uri) {
    std::string out;
    for (size_t i = 0; i < uri.size(); i++) {
        if (uri[i] == '%' && i + 2 < uri.size() + 0 && std::isxdigit((unsigned char)uri[i + 1]) && std::isxdigit((unsigned char)uri[i + 2])) {
            out.push_back((char)std::strtol(uri.substr(i + 1, 2).c_str(), nullptr, 16));
            i += 2;
        } else out.push_back(uri[i]);
    }
    return out;
}

bool is_data_uri(const std::string& uri) { return uri.compare(0, 5, "data:") == 0; }

std::vector<uint8_t> decode_data_uri(const std::string& uri) {
    const size_t comma = uri.find(',');
    if (comma == std::string::npos || uri.substr(0, comma).find(";base64") == std::string::npos) throw gltf_error("gltf: only base64 data URIs are supported");
    std::vector<uint8_t> out;
    uint32_t acc = 0;
    int bits = 0;
    for (size_t i = comma + 1; i < uri.size(); i++) {
        const char c = uri[i];
        int v;
        if (c >= 'A' && c <= 'Z') v = c - 'A';
        else if (c >= 'a' && c <= 'z') v = c - 'a' + 26;
        else if (c >= '0' && c <= '9') v = c - '0' + 52;
        else if (c == '+' || c == '-') v = 62;
        else if (c == '/' || c == '_') v = 63;
        else if (c == '=') break;
        else throw gltf_error("gltf: invalid character in a base64 data URI");
        acc = (acc << 6) | (uint32_t)v;
        bits += 6;
        if (bits >= 8) { bits -= 8; out.push_back((uint8_t)((acc >> bits) & 0xFFu)); }
    }
    return out;
}

uint32_t le32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

// column-major 4x4: r = a * b
void mat_mul(const double a[16], const double b[16], double r[16]) {
    double t[16];
    for (int c = 0; c < 4; c++)
        for (int rr = 0; rr < 4; rr++) {
            double s = 0.0;
            for (int k = 0; k < 4; k++) s += a[k * 4 + rr] * b[c * 4 + k];
            t[c * 4 + rr] = s;
        }
    std::memcpy(r, t, sizeof t);
}

// local matrix of a node: `matrix`, or translation * rotation * scale
void node_matrix(const json& n, double m[16]) {
    if (n.contains("matrix")) {
        const json& a = n["matrix"];
        if (a.size() != 16) throw gltf_error("gltf: node.matrix must have 16 numbers");
        for (int i = 0; i < 16; i++) m[i] = a[(size_t)i].as_double();
        return;
    }
    double t[3] = { 0, 0, 0 }, q[4] = { 0, 0, 0, 1 }, s[3] = { 1, 1, 1 };
    auto read = [&](const char* key, double* dst, size_t count) {
        if (!n.contains(key)) return;
        const json& a = n[key];
        if (a.size() != count) throw gltf_error(std::string("gltf: node.") + key + " has the wrong length");
        for (size_t i = 0; i < count; i++) dst[i] = a[i].as_double();
    };
    read("translation", t, 3); read("rotation", q, 4); read("scale", s, 3);
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    const double r[9] = { 1 - 2 * (y * y + z * z), 2 * (x * y + z * w),     2 * (x * z - y * w),      // column 0
                          2 * (x * y - z * w),     1 - 2 * (x * x + z * z), 2 * (y * z + x * w),      // column 1
                          2 * (x * z + y * w),     2 * (y * z - x * w),     1 - 2 * (x * x + y * y) };  // column 2
    for (int c = 0; c < 3; c++) {
        for (int rr = 0; rr < 3; rr++) m[c * 4 + rr] = r[c * 3 + rr] * s[c];
        m[c * 4 + 3] = 0.0;
    }
    m[12] = t[0]; m[13] = t[1]; m[14] = t[2]; m[15] = 1.0;
}

}  // namespace

gltf::gltf(const std::string& filepath) {
    const std::string parent_path = parent_directory(filepath);
    std::vector<uint8_t> file = read_binary_file(filepath);
    if (file.size() >= 12 && std::memcmp(file.data(), "glTF", 4) == 0) {
        // .glb: 12-byte header, then chunks (length, type, data): JSON first, optionally BIN
        if (le32(file.data() + 4) != 2) throw gltf_error("gltf: only version 2 .glb containers are supported");
        size_t pos = 12;
        bool have_json = false;
        while (pos + 8 <= file.size()) {
            const uint32_t len = le32(file.data() + pos), type = le32(file.data() + pos + 4);
            if (len > file.size() - pos - 8) throw gltf_error("gltf: .glb chunk runs past the end of the file");
            if (type == 0x4E4F534Au && !have_json) { gltf_json = json::parse(std::string((const char*)file.data() + pos + 8, len)); have_json = true; }
            else if (type == 0x004E4942u && glb_binary.empty()) glb_binary.assign(file.begin() + pos + 8, file.begin() + pos + 8 + len);
            pos += 8 + (size_t)len;
        }
        if (!have_json) throw gltf_error("gltf: .glb container without a JSON chunk");
    } else {
        gltf_json = json::parse(std::string((const char*)file.data(), file.size()));
    }
    if (!gltf_json.is_object()) throw gltf_error("gltf: the document is not a JSON object");
    if (gltf_json.contains("asset") && gltf_json["asset"].contains("version") && gltf_json["asset"]["version"].as_string().compare(0, 2, "2.") != 0)
        throw gltf_error("gltf: only glTF 2.x is supported");

    load_buffers(parent_path);
    load_textures(parent_path);
    load_materials();
    load_meshes();

    if (!gltf_json.contains("scenes") || gltf_json["scenes"].size() == 0) throw gltf_error("gltf: the file has no scene");
    const uint32_t scene_index = gltf_json.contains("scene") ? gltf_json["scene"].as_u32() : 0u;  // the reference always takes scene 0
    const json& scene = gltf_json["scenes"][scene_index];
    if (scene.contains("nodes")) {
        const json& root_children = scene["nodes"];
        root_node.children.resize(root_children.size());
        for (size_t i = 0; i < root_children.size(); i++) load_node(root_children[i].as_u32(), root_node.children[i], 0);
    }
}

void gltf::load_buffers(const std::string& parent_path) {
    if (!gltf_json.contains("buffers")) return;
    const json& list = gltf_json["buffers"];
    buffers.resize(list.size());
    for (size_t i = 0; i < list.size(); i++) {
        const json& b = list[i];
        if (b.contains("uri")) {
            const std::string& uri = b["uri"].as_string();
            buffers[i] = is_data_uri(uri) ? decode_data_uri(uri) : read_binary_file(parent_path + "/" + uri_to_path(uri));
        } else if (i == 0 && !glb_binary.empty()) {
            buffers[i] = glb_binary;
        } else {
            throw gltf_error("gltf: buffer " + std::to_string(i) + " has no uri");
        }
        const size_t declared = (size_t)b["byteLength"].as_int();
        if (buffers[i].size() < declared) throw gltf_error("gltf: buffer " + std::to_string(i) + " is shorter than its byteLength");
    }
}

void gltf::load_node(uint32_t index, node& parent, int depth) {
    if (depth > 512) throw gltf_error("gltf: node hierarchy too deep (a cycle?)");
    const json& gltf_node = gltf_json["nodes"][index];
    node_matrix(gltf_node, parent.matrix);
    if (gltf_node.contains("children")) {
        const json& children = gltf_node["children"];
        parent.children.resize(children.size());
        for (size_t i = 0; i < children.size(); i++) load_node(children[i].as_u32(), parent.children[i], depth + 1);
    }
    if (gltf_node.contains("mesh")) {
        const uint32_t m = gltf_node["mesh"].as_u32();
        if (m >= meshes.size()) throw gltf_error("gltf: node names a mesh that does not exist");
        parent.mesh = &meshes[m];
    }
}

void gltf::load_meshes() {
    if (!gltf_json.contains("meshes")) return;
    const json& gltf_meshes = gltf_json["meshes"];
    meshes.resize(gltf_meshes.size());
    for (size_t mesh_index = 0; mesh_index < gltf_meshes.size(); mesh_index++) {
        const json& primitives = gltf_meshes[mesh_index]["primitives"];
        for (const json& primitive : primitives.elements()) {
            uint32_t material;
            if (primitive.contains("material")) {
                material = primitive["material"].as_u32();
                if (material >= materials.size()) throw gltf_error("gltf: primitive names a material that does not exist");
            } else {
                if (default_material == 0xffffffffu) {
                    default_material = (uint32_t)materials.size();
                    materials.push_back(scene_source::source_material{ { 1.f, 1.f, 1.f, 1.f }, 0u, 0u, 1.f, 1.f });
                }
                material = default_material;
            }
            meshes[mesh_index].add_submesh(load_primitive(primitive), material);
        }
    }
}

Mesh::submesh gltf::load_primitive(const json& primitive) {
    if (primitive.contains("mode") && primitive["mode"].as_u32() != 4u) throw gltf_error("gltf: only triangle-list primitives (mode 4) are supported");
    const json& attributes = primitive["attributes"];
    if (!attributes.contains("POSITION")) throw gltf_error("gltf: primitive without POSITION");
    Mesh::submesh submesh;

    uint32_t vertex_count = 0;
    {
        const std::vector<double> p = load_attribute(attributes["POSITION"].as_u32(), 3, &vertex_count);
        submesh.attributes[ATTRIBUTE_TYPE::POSITION].assign(p.begin(), p.end());  // double -> float: exact for float sources
    }
    submesh.vertex_count = vertex_count;

    if (primitive.contains("indices")) {
        uint32_t n = 0;
        const std::vector<double> idx = load_attribute(primitive["indices"].as_u32(), 1, &n);
        submesh.indices.resize(n);
        for (uint32_t i = 0; i < n; i++) {
            if (idx[i] < 0 || idx[i] >= vertex_count) throw gltf_error("gltf: vertex index out of range");
            submesh.indices[i] = (uint32_t)idx[i];
        }
    } else {
        submesh.indices.resize(vertex_count);
        for (uint32_t i = 0; i < vertex_count; i++) submesh.indices[i] = i;
    }
    submesh.indices.resize(submesh.indices.size() / 3 * 3);
    submesh.index_count = (uint32_t)submesh.indices.size();

    auto& normals = submesh.attributes[ATTRIBUTE_TYPE::NORMAL];
    if (attributes.contains("NORMAL")) {
        uint32_t n = 0;
        const std::vector<double> v = load_attribute(attributes["NORMAL"].as_u32(), 3, &n);
        if (n != vertex_count) throw gltf_error("gltf: NORMAL and POSITION differ in count");
        normals.assign(v.begin(), v.end());
    } else {
        // area-weighted vertex normals (the sum of the un-normalised face normals), in double
        const auto& pos = submesh.attributes[ATTRIBUTE_TYPE::POSITION];
        std::vector<double> acc((size_t)vertex_count * 3, 0.0);
        for (size_t t = 0; t + 2 < submesh.indices.size(); t += 3) {
            const uint32_t i0 = submesh.indices[t], i1 = submesh.indices[t + 1], i2 = submesh.indices[t + 2];
            double e1[3], e2[3];
            for (int k = 0; k < 3; k++) { e1[k] = (double)pos[i1 * 3 + k] - pos[i0 * 3 + k]; e2[k] = (double)pos[i2 * 3 + k] - pos[i0 * 3 + k]; }
            const double n[3] = { e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0] };
            for (uint32_t i : { i0, i1, i2 }) for (int k = 0; k < 3; k++) acc[(size_t)i * 3 + k] += n[k];
        }
        normals.resize((size_t)vertex_count * 3);
        for (uint32_t i = 0; i < vertex_count; i++) {
            const double l = std::sqrt(acc[i * 3] * acc[i * 3] + acc[i * 3 + 1] * acc[i * 3 + 1] + acc[i * 3 + 2] * acc[i * 3 + 2]);
            for (int k = 0; k < 3; k++) normals[(size_t)i * 3 + k] = l > 0.0 ? (float)(acc[i * 3 + k] / l) : (k == 1 ? 1.f : 0.f);
        }
    }

    auto& uvs = submesh.attributes[ATTRIBUTE_TYPE::UV_0];
    if (attributes.contains("TEXCOORD_0")) {
        uint32_t n = 0;
        const std::vector<double> v = load_attribute(attributes["TEXCOORD_0"].as_u32(), 2, &n);
        if (n != vertex_count) throw gltf_error("gltf: TEXCOORD_0 and POSITION differ in count");
        uvs.assign(v.begin(), v.end());
    } else {
        uvs.assign((size_t)vertex_count * 2, 0.f);
    }
    return submesh;
}

// Reads accessor `accessor_index` into doubles (exact for every component type), `expected_components` per element.
std::vector<double> gltf::load_attribute(uint32_t accessor_index, uint32_t expected_components, uint32_t* count) const {
    const json& accessor = gltf_json["accessors"][accessor_index];
    if (accessor.contains("sparse")) throw gltf_error("gltf: sparse accessors are not supported");
    const std::string& type = accessor["type"].as_string();
    const uint32_t components = type == "SCALAR" ? 1 : type == "VEC2" ? 2 : type == "VEC3" ? 3 : type == "VEC4" ? 4 : 0;
    if (components != expected_components) throw gltf_error("gltf: accessor " + std::to_string(accessor_index) + " has type " + type + ", " + std::to_string(expected_components) + " components expected");
    const uint32_t component_type = accessor["componentType"].as_u32();
    size_t component_size;
    switch (component_type) {
    case 5120: case 5121: component_size = 1; break;
    case 5122: case 5123: component_size = 2; break;
    case 5125: case 5126: component_size = 4; break;
    default: throw gltf_error("gltf: unknown accessor componentType");
    }
    const bool normalized = accessor.contains("normalized") && accessor["normalized"].as_bool();
    const uint32_t n = accessor["count"].as_u32();
    *count = n;
    std::vector<double> out((size_t)n * components, 0.0);
    if (!accessor.contains("bufferView")) return out;  // the format defines this as all zeros

    const json& view = gltf_json["bufferViews"][accessor["bufferView"].as_u32()];
    const uint32_t buffer_index = view["buffer"].as_u32();
    if (buffer_index >= buffers.size()) throw gltf_error("gltf: buffer view names a buffer that does not exist");
    const std::vector<uint8_t>& buffer = buffers[buffer_index];
    const size_t view_offset = view.contains("byteOffset") ? (size_t)view["byteOffset"].as_int() : 0;
    const size_t view_length = (size_t)view["byteLength"].as_int();
    const size_t accessor_offset = accessor.contains("byteOffset") ? (size_t)accessor["byteOffset"].as_int() : 0;
    const size_t element_size = component_size * components;
    const size_t stride = view.contains("byteStride") && view["byteStride"].as_int() > 0 ? (size_t)view["byteStride"].as_int() : element_size;
    if (view_offset > buffer.size() || view_length > buffer.size() - view_offset) throw gltf_error("gltf: buffer view runs past the end of its buffer");
    if (n && (accessor_offset > view_length || (size_t)(n - 1) * stride + element_size > view_length - accessor_offset))
        throw gltf_error("gltf: accessor " + std::to_string(accessor_index) + " runs past the end of its buffer view");
    const uint8_t* base = buffer.data() + view_offset + accessor_offset;
    for (uint32_t i = 0; i < n; i++)
        for (uint32_t c = 0; c < components; c++) {
            const uint8_t* p = base + (size_t)i * stride + c * component_size;
            double v;
            switch (component_type) {
            case 5120: { int8_t x; std::memcpy(&x, p, 1); v = normalized ? std::fmax(x / 127.0, -1.0) : x; break; }
            case 5121: { v = normalized ? p[0] / 255.0 : p[0]; break; }
            case 5122: { int16_t x; std::memcpy(&x, p, 2); v = normalized ? std::fmax(x / 32767.0, -1.0) : x; break; }
            case 5123: { uint16_t x; std::memcpy(&x, p, 2); v = normalized ? x / 65535.0 : x; break; }
            case 5125: { uint32_t x; std::memcpy(&x, p, 4); v = x; break; }
            default: { float x; std::memcpy(&x, p, 4); v = x; break; }
            }
            out[(size_t)i * components + c] = v;
        }
    return out;
}

void gltf::load_textures(const std::string& parent_path) {
    if (gltf_json.contains("images")) {
        const json& gltf_images = gltf_json["images"];
        images.resize(gltf_images.size());
        for (size_t i = 0; i < gltf_images.size(); i++) {
            const json& im = gltf_images[i];
            if (im.contains("uri")) {
                const std::string& uri = im["uri"].as_string();
                if (is_data_uri(uri)) {
                    const std::vector<uint8_t> bytes = decode_data_uri(uri);
                    images[i] = decode_image(bytes.data(), bytes.size());
                } else {
                    images[i] = load_image(parent_path + "/" + uri_to_path(uri));
                }
            } else if (im.contains("bufferView")) {
                const json& view = gltf_json["bufferViews"][im["bufferView"].as_u32()];
                const std::vector<uint8_t>& buffer = buffers.at(view["buffer"].as_u32());
                const size_t off = view.contains("byteOffset") ? (size_t)view["byteOffset"].as_int() : 0, len = (size_t)view["byteLength"].as_int();
                if (off > buffer.size() || len > buffer.size() - off) throw gltf_error("gltf: image buffer view runs past the end of its buffer");
                images[i] = decode_image(buffer.data() + off, len);
            } else {
                throw gltf_error("gltf: image " + std::to_string(i) + " has neither uri nor bufferView");
            }
        }
    }
    if (gltf_json.contains("textures")) {
        const json& gltf_textures = gltf_json["textures"];
        textures.resize(gltf_textures.size());
        for (size_t i = 0; i < gltf_textures.size(); i++) {
            // every texture is sampled LINEAR / REPEAT, whatever its sampler says (src/gltf.cpp:118-137)
            const uint32_t source = gltf_textures[i]["source"].as_u32();
            if (source >= images.size()) throw gltf_error("gltf: texture names an image that does not exist");
            textures[i] = source;
        }
    }
}

// src/gltf.cpp:147-195, with the format's defaults where the reference leaves members uninitialised.
void gltf::load_materials() {
    if (!gltf_json.contains("materials")) return;
    const json& gltf_materials = gltf_json["materials"];
    materials.resize(gltf_materials.size());
    for (size_t i = 0; i < gltf_materials.size(); i++) {
        scene_source::source_material& m = materials[i];
        m = scene_source::source_material{ { 1.f, 1.f, 1.f, 1.f }, 0u, 0u, 1.f, 1.f };
        const json& gm = gltf_materials[i];
        if (!gm.contains("pbrMetallicRoughness")) continue;
        const json& pbr = gm["pbrMetallicRoughness"];
        if (pbr.contains("baseColorFactor")) {
            const json& c = pbr["baseColorFactor"];
            if (c.size() != 4) throw gltf_error("gltf: baseColorFactor must have 4 numbers");
            for (size_t k = 0; k < 4; k++) m.base_color[k] = c[k].as_float();
        }
        auto texture_number = [&](const char* key) -> uint32_t {
            if (!pbr.contains(key)) return 0u;
            const uint32_t t = pbr[key]["index"].as_u32();
            if (t >= textures.size()) throw gltf_error("gltf: material names a texture that does not exist");
            return t + 1u;
        };
        m.albedo_texture = texture_number("baseColorTexture");
        m.mr_texture = texture_number("metallicRoughnessTexture");
        if (pbr.contains("metallicFactor")) m.metalness = pbr["metallicFactor"].as_float();
        if (pbr.contains("roughnessFactor")) m.roughness = pbr["roughnessFactor"].as_float();
    }
}

namespace {
// Index words and material records of one mesh, appended to the flat arrays: indices made absolute, one material record
// per submesh (src/scene.cpp:132-163).
void append_submeshes(scene_source& out, const Mesh& mesh, const std::vector<scene_source::source_material>& materials, size_t vertices_offset) {
    const auto& mesh_indices = mesh.get_indices();
    for (const auto& submesh : mesh.get_submeshes()) {
        const size_t vertex_offset = vertices_offset + submesh.vertex_offset;
        const size_t material_id = out.materials.size();
        for (size_t e = 0; e < submesh.element_count; e++) {
            const size_t at = submesh.index_offset + e * 3;
            if (out.index_words == 4) {
                for (int k = 0; k < 3; k++) out.indices.push_back((uint32_t)(mesh_indices[at + k] + vertex_offset));
                out.indices.push_back((uint32_t)material_id);
                continue;
            }
            // vertex id in the low 24 bits, one byte of the material id on top of each word (src/scene.cpp:140-142)
            out.indices.push_back((uint32_t)((mesh_indices[at] + vertex_offset) | (0xff000000u & (material_id << 8))));
            out.indices.push_back((uint32_t)((mesh_indices[at + 1] + vertex_offset) | (0xff000000u & (material_id << 16))));
            out.indices.push_back((uint32_t)((mesh_indices[at + 2] + vertex_offset) | (0xff000000u & (material_id << 24))));
        }
        out.materials.push_back(materials[submesh.material]);
    }
}
}  // namespace

scene_source gltf::flatten(bool apply_transforms, int wide_indices) const {
    scene_source out;
    // what the flat scene will hold, to pick the index layout before the first word is written
    {
        size_t total_vertices = 0, total_materials = 0;
        std::queue<const node*> walk;
        walk.push(&root_node);
        while (!walk.empty()) {
            const node* n = walk.front();
            walk.pop();
            if (n->mesh) { total_vertices += n->mesh->vertices_count(); total_materials += n->mesh->get_submeshes().size(); }
            for (const node& child : n->children) walk.push(&child);
        }
        const bool fits = total_vertices < (1u << 24) && total_materials < (1u << 24);
        if (wide_indices == 0 && !fits) throw gltf_error("gltf: more than 2^24 vertices or material records do not fit the packed index words");
        if (total_vertices > 0xffffffffull) throw gltf_error("gltf: more than 2^32 vertices");
        out.index_words = (wide_indices == 1 || (wide_indices < 0 && !fits)) ? 4u : 3u;
    }
    // textures: one per glTF texture, numbered as the materials number them
    for (uint32_t image_index : textures) {
        const raw_image& im = images[image_index];
        scene_source::source_texture t;
        t.width = (uint32_t)im.width; t.height = (uint32_t)im.height;
        t.rgba = im.data;
        out.textures.push_back(std::move(t));
    }

    struct pending { const node* n; double world[16]; };
    std::queue<pending> nodes_to_load;
    {
        pending root; root.n = &root_node;
        std::memcpy(root.world, root_node.matrix, sizeof root.world);
        nodes_to_load.push(root);
    }
    size_t vertices_offset = 0;
    while (!nodes_to_load.empty()) {
        const pending cur = nodes_to_load.front();
        nodes_to_load.pop();
        if (cur.n->mesh != nullptr) {
            const Mesh* mesh = cur.n->mesh;
            const auto& mesh_positions = mesh->get_attribute(ATTRIBUTE_TYPE::POSITION);
            const auto& mesh_normals = mesh->get_attribute(ATTRIBUTE_TYPE::NORMAL);
            const auto& mesh_uvs_0 = mesh->get_attribute(ATTRIBUTE_TYPE::UV_0);

            if (!apply_transforms) {
                out.positions.insert(out.positions.end(), mesh_positions.begin(), mesh_positions.end());
                out.normals.insert(out.normals.end(), mesh_normals.begin(), mesh_normals.end());
            } else {
                const double* m = cur.world;
                // normals go through the inverse transpose of the upper 3x3 (= the cofactor matrix, up to scale)
                const double cof[9] = { m[5] * m[10] - m[6] * m[9], m[6] * m[8] - m[4] * m[10], m[4] * m[9] - m[5] * m[8],
                                        m[9] * m[2] - m[10] * m[1], m[10] * m[0] - m[8] * m[2], m[8] * m[1] - m[9] * m[0],
                                        m[1] * m[6] - m[2] * m[5], m[2] * m[4] - m[0] * m[6], m[0] * m[5] - m[1] * m[4] };
                const double det = m[0] * cof[0] + m[1] * cof[1] + m[2] * cof[2];
                for (size_t v = 0; v < mesh->vertices_count(); v++) {
                    const double p[3] = { mesh_positions[v * 3], mesh_positions[v * 3 + 1], mesh_positions[v * 3 + 2] };
                    for (int r = 0; r < 3; r++) out.positions.push_back((float)(m[r] * p[0] + m[4 + r] * p[1] + m[8 + r] * p[2] + m[12 + r]));
                    const double n[3] = { mesh_normals[v * 3], mesh_normals[v * 3 + 1], mesh_normals[v * 3 + 2] };
                    double t[3];
                    for (int r = 0; r < 3; r++) t[r] = (cof[r] * n[0] + cof[3 + r] * n[1] + cof[6 + r] * n[2]) * (det < 0 ? -1.0 : 1.0);
                    const double l = std::sqrt(t[0] * t[0] + t[1] * t[1] + t[2] * t[2]);
                    for (int r = 0; r < 3; r++) out.normals.push_back(l > 0.0 ? (float)(t[r] / l) : (float)n[r]);
                }
            }
            out.uvs.insert(out.uvs.end(), mesh_uvs_0.begin(), mesh_uvs_0.end());

            append_submeshes(out, *mesh, materials, vertices_offset);
            vertices_offset += mesh->vertices_count();
        }
        for (const node& child : cur.n->children) {
            pending next; next.n = &child;
            mat_mul(cur.world, child.matrix, next.world);
            nodes_to_load.push(next);
        }
    }
    return out;
}

// Each mesh once, every node that shows it a placement: what a glTF file means by sharing a mesh between nodes.
scene_source gltf::flatten_instanced(int wide_indices) const {
    scene_source out;
    {
        size_t total_vertices = 0, total_materials = 0;
        for (const Mesh& m : meshes) { total_vertices += m.vertices_count(); total_materials += m.get_submeshes().size(); }
        const bool fits = total_vertices < (1u << 24) && total_materials < (1u << 24);
        if (wide_indices == 0 && !fits) throw gltf_error("gltf: more than 2^24 vertices or material records do not fit the packed index words");
        if (total_vertices > 0xffffffffull) throw gltf_error("gltf: more than 2^32 vertices");
        out.index_words = (wide_indices == 1 || (wide_indices < 0 && !fits)) ? 4u : 3u;
    }
    for (uint32_t image_index : textures) {
        const raw_image& im = images[image_index];
        scene_source::source_texture t;
        t.width = (uint32_t)im.width; t.height = (uint32_t)im.height;
        t.rgba = im.data;
        out.textures.push_back(std::move(t));
    }
    std::vector<int32_t> object_of_mesh(meshes.size(), -1);
    out.object_first.push_back(0);
    struct pending { const node* n; double world[16]; };
    std::queue<pending> nodes_to_load;
    {
        pending root; root.n = &root_node;
        std::memcpy(root.world, root_node.matrix, sizeof root.world);
        nodes_to_load.push(root);
    }
    size_t vertices_offset = 0;
    while (!nodes_to_load.empty()) {
        const pending cur = nodes_to_load.front();
        nodes_to_load.pop();
        if (cur.n->mesh != nullptr) {
            const Mesh* mesh = cur.n->mesh;
            const size_t mesh_index = (size_t)(mesh - meshes.data());
            if (object_of_mesh[mesh_index] < 0 && !mesh->get_indices().empty()) {
                const auto& p = mesh->get_attribute(ATTRIBUTE_TYPE::POSITION);
                const auto& n = mesh->get_attribute(ATTRIBUTE_TYPE::NORMAL);
                const auto& uv = mesh->get_attribute(ATTRIBUTE_TYPE::UV_0);
                out.positions.insert(out.positions.end(), p.begin(), p.end());
                out.normals.insert(out.normals.end(), n.begin(), n.end());
                out.uvs.insert(out.uvs.end(), uv.begin(), uv.end());
                append_submeshes(out, *mesh, materials, vertices_offset);
                vertices_offset += mesh->vertices_count();
                object_of_mesh[mesh_index] = (int32_t)out.object_first.size() - 1;
                out.object_first.push_back((uint32_t)out.triangle_count());
            }
            if (object_of_mesh[mesh_index] >= 0) {
                scene_source::placement place;
                place.object = (uint32_t)object_of_mesh[mesh_index];
                const double* m = cur.world;  // column-major; rows of (3x3 part, translation)
                for (int r = 0; r < 3; r++) {
                    place.object_to_world[r * 4] = (float)m[r]; place.object_to_world[r * 4 + 1] = (float)m[4 + r];
                    place.object_to_world[r * 4 + 2] = (float)m[8 + r]; place.object_to_world[r * 4 + 3] = (float)m[12 + r];
                }
                out.placements.push_back(place);
            }
        }
        for (const node& child : cur.n->children) {
            pending next; next.n = &child;
            mat_mul(cur.world, child.matrix, next.world);
            nodes_to_load.push(next);
        }
    }
    if (out.placements.empty()) throw gltf_error("gltf: no node shows a mesh");
    return out;
}
