// Host-memory half of the multi-GPU exchange: the stripe layout of the tile partition and the file transport.
// No GPU, no RCCL: this file is what the CPU tests exercise with several processes.
#include "exchange.hpp"

#include <chrono>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <thread>

size_t stripe_chunk_texels(uint32_t width, uint32_t height, uint32_t stripe_rows, uint32_t world) {
    const size_t stripes = ((size_t)height + stripe_rows - 1) / stripe_rows;
    return ((stripes + world - 1) / world) * stripe_rows * width;
}

void pack_stripes_host(const float* frame, uint32_t width, uint32_t height, uint32_t stripe_rows, uint32_t rank, uint32_t world, float* chunk) {
    const size_t chunk_stripes = stripe_chunk_texels(width, height, stripe_rows, world) / ((size_t)stripe_rows * width);
    const size_t row_floats = (size_t)width * 4;
    for (size_t j = 0; j < chunk_stripes; j++)
        for (uint32_t y = 0; y < stripe_rows; y++) {
            const size_t row = (j * world + rank) * stripe_rows + y;
            float* dst = chunk + (j * stripe_rows + y) * row_floats;
            if (row < height) std::memcpy(dst, frame + row * row_floats, row_floats * sizeof(float));
            else std::memset(dst, 0, row_floats * sizeof(float));
        }
}

void unpack_stripes_host(float* frame, uint32_t width, uint32_t height, uint32_t stripe_rows, uint32_t world, const float* chunks) {
    const size_t chunk_texels = stripe_chunk_texels(width, height, stripe_rows, world);
    const size_t chunk_stripes = chunk_texels / ((size_t)stripe_rows * width);
    const size_t row_floats = (size_t)width * 4;
    for (uint32_t r = 0; r < world; r++)
        for (size_t j = 0; j < chunk_stripes; j++)
            for (uint32_t y = 0; y < stripe_rows; y++) {
                const size_t row = (j * world + r) * stripe_rows + y;
                if (row < height) std::memcpy(frame + row * row_floats, chunks + r * chunk_texels * 4 + (j * stripe_rows + y) * row_floats, row_floats * sizeof(float));
            }
}

namespace {
std::string part_name(const std::string& dir, const char* what, uint32_t sequence, uint32_t rank) {
    return dir + "/" + what + "_" + std::to_string(sequence) + "_" + std::to_string(rank) + ".bin";
}
// A file appears under its final name only when it is complete (written under another name, then renamed).
void publish(const std::string& path, const void* data, size_t bytes) {
    const std::string tmp = path + ".part";
    FILE* f = std::fopen(tmp.c_str(), "wb");
    if (!f) throw std::runtime_error("exchange: cannot write " + tmp);
    const size_t n = bytes ? std::fwrite(data, 1, bytes, f) : 0;
    std::fclose(f);
    if (n != bytes || std::rename(tmp.c_str(), path.c_str()) != 0) throw std::runtime_error("exchange: cannot publish " + path);
}
void await(const std::string& path, void* data, size_t bytes, double timeout_s = 120.0) {
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        if (FILE* f = std::fopen(path.c_str(), "rb")) {
            const size_t n = bytes ? std::fread(data, 1, bytes, f) : 0;
            std::fclose(f);
            if (n != bytes) throw std::runtime_error("exchange: " + path + " is shorter than expected");
            return;
        }
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s)
            throw std::runtime_error("exchange: timed out waiting for " + path + " (is every rank running?)");
        std::this_thread::sleep_for(std::chrono::milliseconds(2));
    }
}
}  // namespace

void file_all_gather(const std::string& dir, uint32_t sequence, uint32_t rank, uint32_t world, const void* mine, void* all, size_t bytes) {
    publish(part_name(dir, "gather", sequence, rank), mine, bytes);
    for (uint32_t r = 0; r < world; r++) {
        uint8_t* dst = (uint8_t*)all + (size_t)r * bytes;
        if (r == rank) std::memcpy(dst, mine, bytes);
        else await(part_name(dir, "gather", sequence, r), dst, bytes);
    }
}

void file_all_reduce_sum(const std::string& dir, uint32_t sequence, uint32_t rank, uint32_t world, float* data, size_t count) {
    publish(part_name(dir, "reduce", sequence, rank), data, count * sizeof(float));
    std::vector<float> mine(data, data + count), other(count);
    std::fill(data, data + count, 0.0f);
    for (uint32_t r = 0; r < world; r++) {   // rank order on every rank: the same bits everywhere
        const float* src = mine.data();
        if (r != rank) { await(part_name(dir, "reduce", sequence, r), other.data(), count * sizeof(float)); src = other.data(); }
        for (size_t i = 0; i < count; i++) data[i] += src[i];
    }
}
