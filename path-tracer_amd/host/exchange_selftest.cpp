// One rank of a CPU-only check of the tile-partition exchange: fills the stripes it owns of a synthetic frame, gathers over the
// file transport, and compares the merged frame with the whole synthetic frame; then sums a vector over the ranks.
//   exchange_selftest <dir> <rank> <world> <width> <height> <stripe_rows>      exit code 0 = both match
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "exchange.hpp"

static float texel(uint32_t x, uint32_t y, uint32_t c) { return (float)((x * 7u + y * 131u + c * 3u) % 1021u) * 0.25f; }

int main(int argc, char** argv) {
    if (argc != 7) { std::fprintf(stderr, "usage: exchange_selftest dir rank world width height stripe_rows\n"); return 2; }
    const std::string dir = argv[1];
    const uint32_t rank = (uint32_t)std::atoi(argv[2]), world = (uint32_t)std::atoi(argv[3]);
    const uint32_t w = (uint32_t)std::atoi(argv[4]), h = (uint32_t)std::atoi(argv[5]), stripe = (uint32_t)std::atoi(argv[6]);
    try {
        std::vector<float> whole((size_t)w * h * 4), mine((size_t)w * h * 4, 0.0f);
        for (uint32_t y = 0; y < h; y++)
            for (uint32_t x = 0; x < w; x++)
                for (uint32_t c = 0; c < 4; c++) {
                    whole[((size_t)y * w + x) * 4 + c] = texel(x, y, c);
                    if ((y / stripe) % world == rank) mine[((size_t)y * w + x) * 4 + c] = texel(x, y, c);   // a rank renders its own stripes only
                }
        const size_t chunk = stripe_chunk_texels(w, h, stripe, world) * 4;
        std::vector<float> packed(chunk), gathered(chunk * world), merged((size_t)w * h * 4, -1.0f);
        pack_stripes_host(mine.data(), w, h, stripe, rank, world, packed.data());
        file_all_gather(dir, 0, rank, world, packed.data(), gathered.data(), chunk * sizeof(float));
        unpack_stripes_host(merged.data(), w, h, stripe, world, gathered.data());
        if (merged != whole) { std::fprintf(stderr, "rank %u: merged frame differs\n", rank); return 1; }
        std::vector<float> v(1000);
        for (size_t i = 0; i < v.size(); i++) v[i] = (float)(i + 1) * (float)(rank + 1);
        file_all_reduce_sum(dir, 1, rank, world, v.data(), v.size());
        const float ranks_sum = (float)(world * (world + 1) / 2);
        for (size_t i = 0; i < v.size(); i++)
            if (v[i] != (float)(i + 1) * ranks_sum) { std::fprintf(stderr, "rank %u: sum differs at %zu\n", rank, i); return 1; }
    } catch (const std::exception& e) {
        std::fprintf(stderr, "rank %u: %s\n", rank, e.what());
        return 1;
    }
    return 0;
}
