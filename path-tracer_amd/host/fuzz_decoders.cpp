// Robustness driver for the file readers (JSON, PNG, JPEG, glTF): feeds them seeded random corruptions of valid
// inputs and checks that every outcome is either a result or one of the readers' own exceptions -- never a crash,
// an out-of-bounds access or undefined behaviour (the Makefile builds this with AddressSanitizer + UBSan).
//   fuzz_decoders <file> [<file> ...] [--rounds N] [--seed S]
// A .gltf / .glb argument is mutated in a scratch copy next to the original so that its relative URIs still resolve.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <string>
#include <vector>

#include "gltf.hpp"
#include "image.hpp"
#include "json.hpp"

namespace {

uint64_t g_state = 0x9E3779B97F4A7C15ull;
uint32_t rnd() {  // splitmix64
    uint64_t z = (g_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return (uint32_t)((z ^ (z >> 31)) >> 16);
}

std::vector<uint8_t> read_all(const std::string& path) {
    std::vector<uint8_t> bytes;
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) { std::fprintf(stderr, "cannot open %s\n", path.c_str()); std::exit(2); }
    uint8_t buf[65536];
    size_t n;
    while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) bytes.insert(bytes.end(), buf, buf + n);
    std::fclose(f);
    return bytes;
}

std::vector<uint8_t> mutate(const std::vector<uint8_t>& in) {
    std::vector<uint8_t> out = in;
    if (out.empty()) return out;
    switch (rnd() % 5) {
    case 0:  // flip a few bits
        for (uint32_t k = 1 + rnd() % 4; k > 0; k--) out[rnd() % out.size()] ^= (uint8_t)(1u << (rnd() % 8));
        break;
    case 1:  // overwrite a short run with random bytes
        for (size_t at = rnd() % out.size(), k = 0; k < 1 + rnd() % 16 && at + k < out.size(); k++) out[at + k] = (uint8_t)rnd();
        break;
    case 2:  // truncate
        out.resize(rnd() % out.size());
        break;
    case 3: {  // plant an extreme 32-bit value (lengths, counts, offsets)
        static const uint32_t kValues[] = { 0u, 1u, 0x7fffffffu, 0x80000000u, 0xffffffffu, 0xfffffff0u, 65535u, 65536u };
        const uint32_t v = kValues[rnd() % 8];
        if (out.size() >= 4) {
            const size_t at = rnd() % (out.size() - 3);
            if (rnd() & 1) { out[at] = (uint8_t)(v >> 24); out[at + 1] = (uint8_t)(v >> 16); out[at + 2] = (uint8_t)(v >> 8); out[at + 3] = (uint8_t)v; }
            else std::memcpy(&out[at], &v, 4);
        }
        break;
    }
    default: {  // duplicate a slice somewhere else
        const size_t from = rnd() % out.size(), len = 1 + rnd() % 64, to = rnd() % out.size();
        for (size_t k = 0; k < len && from + k < in.size() && to + k < out.size(); k++) out[to + k] = in[from + k];
        break;
    }
    }
    return out;
}

bool ends_with(const std::string& s, const char* suffix) {
    const size_t n = std::strlen(suffix);
    return s.size() >= n && s.compare(s.size() - n, n, suffix) == 0;
}

}  // namespace

int main(int argc, char** argv) {
    std::vector<std::string> files;
    int rounds = 300;
    for (int i = 1; i < argc; i++) {
        if (!std::strcmp(argv[i], "--rounds") && i + 1 < argc) rounds = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--seed") && i + 1 < argc) g_state ^= std::strtoull(argv[++i], nullptr, 0);
        else files.push_back(argv[i]);
    }
    if (files.empty()) { std::fprintf(stderr, "usage: fuzz_decoders <file>... [--rounds N] [--seed S]\n"); return 2; }
    size_t accepted = 0, rejected = 0;
    for (const std::string& path : files) {
        const std::vector<uint8_t> original = read_all(path);
        const bool is_gltf = ends_with(path, ".gltf") || ends_with(path, ".glb");
        const bool is_json = ends_with(path, ".json") || ends_with(path, ".gltf");
        const std::string scratch = path.substr(0, path.find_last_of('.')) + ".fuzz" + path.substr(path.find_last_of('.'));
        for (int r = 0; r < rounds; r++) {
            const std::vector<uint8_t> bytes = r == 0 ? original : mutate(original);
            try {
                if (is_gltf) {
                    FILE* f = std::fopen(scratch.c_str(), "wb");
                    if (!f) { std::fprintf(stderr, "cannot write %s\n", scratch.c_str()); return 2; }
                    std::fwrite(bytes.data(), 1, bytes.size(), f);
                    std::fclose(f);
                    const gltf model(scratch);
                    const scene_source s = model.flatten(r & 1, (r >> 1) % 3 - 1);
                    if (s.indices.size() % s.index_words) std::abort();
                } else if (is_json) {
                    (void)json::parse(std::string((const char*)bytes.data(), bytes.size())).dump();
                } else {
                    const raw_image im = decode_image(bytes.data(), bytes.size());
                    if (im.data.size() != (size_t)im.width * im.height * 4) std::abort();
                }
                accepted++;
            } catch (const image_error&) { rejected++; }
            catch (const json_error&) { rejected++; }
            catch (const gltf_error&) { rejected++; }
            catch (const std::bad_alloc&) { rejected++; }
            catch (const std::length_error&) { rejected++; }
            catch (const std::runtime_error&) { rejected++; }   // Mesh::add_submesh
            catch (const std::out_of_range&) { rejected++; }    // vector::at in the loader
        }
        if (is_gltf) std::remove(scratch.c_str());
    }
    std::printf("fuzz_decoders: %zu inputs accepted, %zu rejected, 0 crashes\n", accepted, rejected);
    return 0;
}
