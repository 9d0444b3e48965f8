// The multi-GPU exchange of the render path, one process per GPU (the reference is single-device, src/vk-context.cpp:147-199:
// nothing here has a counterpart there).  The scene is replicated; either every frame is split into stripes of rows
// (tile partition: one all-gather of the owned stripes merges the frame, bit-identical to one GPU) or the samples are dealt
// out (sample partition: one all-reduce of the linearised per-GPU means).
//
// Two transports behind one interface:
//   "rccl"  ncclAllGather / ncclAllReduce on device memory, over xGMI between the GPUs of a node.  Rendezvous: rank 0 writes
//           the communicator id into <dir>/rccl_id, the others wait for the file.
//   "file"  host staging through files in <dir>: for rehearsing several ranks on ONE GPU (RCCL refuses two ranks on a
//           device) and for tests on machines without a GPU.
#pragma once

#include <cstddef>
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

// Host-side layout of the tile partition (the device-side twins are ptr_image_pack_stripes / ptr_image_unpack_stripes).
size_t stripe_chunk_texels(uint32_t width, uint32_t height, uint32_t stripe_rows, uint32_t world);
// The stripes `rank` owns of a width x height frame of float4 texels, in order, zero-padded to the chunk size.
void pack_stripes_host(const float* frame, uint32_t width, uint32_t height, uint32_t stripe_rows, uint32_t rank, uint32_t world, float* chunk);
// `world` chunks in rank order back into the frame's rows.
void unpack_stripes_host(float* frame, uint32_t width, uint32_t height, uint32_t stripe_rows, uint32_t world, const float* chunks);

// File transport on host memory: every rank contributes `bytes` bytes, every rank gets all contributions in rank order.
// `sequence` numbers the collectives of a job (all ranks must use the same numbers in the same order).
void file_all_gather(const std::string& dir, uint32_t sequence, uint32_t rank, uint32_t world, const void* mine, void* all, size_t bytes);
// Elementwise sum of `count` floats over the ranks, summed in rank order on every rank (so all ranks get the same bits).
void file_all_reduce_sum(const std::string& dir, uint32_t sequence, uint32_t rank, uint32_t world, float* data, size_t count);

// The exchange on DEVICE memory (addresses as ptr_buffer_address / ptr_image_address give them).  Not thread-safe; one
// per process.  Every call blocks until the result is in place.
class device_exchange {
public:
    virtual ~device_exchange() = default;
    // send: `bytes` bytes; recv: world * bytes bytes, contributions in rank order
    virtual void all_gather(uint64_t send_address, uint64_t recv_address, size_t bytes) = 0;
    // `count` floats summed in place over the ranks
    virtual void all_reduce_sum(uint64_t address, size_t count) = 0;
    virtual const char* transport() const = 0;

    // transport: "rccl" or "file".  device: the HIP device this process renders on.
    static std::unique_ptr<device_exchange> create(const std::string& transport, uint32_t rank, uint32_t world, const std::string& rendezvous_dir, int device);
};
