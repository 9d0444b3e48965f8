// Device-memory half of the multi-GPU exchange: RCCL over xGMI, or the file transport staged through host memory.
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstring>
#include <stdexcept>

#include "exchange.hpp"

namespace {
void hip_check(hipError_t e, const char* what) {
    if (e != hipSuccess) throw std::runtime_error(std::string("exchange: ") + what + ": " + hipGetErrorString(e));
}
void rccl_check(ncclResult_t r, const char* what) {
    if (r != ncclSuccess) throw std::runtime_error(std::string("exchange: ") + what + ": " + ncclGetErrorString(r));
}

class rccl_exchange final : public device_exchange {
public:
    rccl_exchange(uint32_t rank, uint32_t world, const std::string& dir, int device) {
        hip_check(hipSetDevice(device), "hipSetDevice");
        ncclUniqueId id;
        std::memset(&id, 0, sizeof id);
        // rank 0 makes the id and hands it to the others through the file transport's one-to-all gather
        if (rank == 0) rccl_check(ncclGetUniqueId(&id), "ncclGetUniqueId");
        std::vector<ncclUniqueId> ids(world);
        file_all_gather(dir, 0xffff0000u, rank, world, &id, ids.data(), sizeof id);
        rccl_check(ncclCommInitRank(&comm_, (int)world, ids[0], (int)rank), "ncclCommInitRank");
        hip_check(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking), "hipStreamCreate");
    }
    ~rccl_exchange() override {
        if (comm_) ncclCommDestroy(comm_);
        if (stream_) (void)hipStreamDestroy(stream_);
    }
    void all_gather(uint64_t send, uint64_t recv, size_t bytes) override {
        rccl_check(ncclAllGather((const void*)(uintptr_t)send, (void*)(uintptr_t)recv, bytes / sizeof(float), ncclFloat, comm_, stream_), "ncclAllGather");
        hip_check(hipStreamSynchronize(stream_), "hipStreamSynchronize");
    }
    void all_reduce_sum(uint64_t address, size_t count) override {
        rccl_check(ncclAllReduce((const void*)(uintptr_t)address, (void*)(uintptr_t)address, count, ncclFloat, ncclSum, comm_, stream_), "ncclAllReduce");
        hip_check(hipStreamSynchronize(stream_), "hipStreamSynchronize");
    }
    const char* transport() const override { return "rccl"; }

private:
    ncclComm_t comm_ = nullptr;
    hipStream_t stream_ = nullptr;
};

class file_exchange final : public device_exchange {
public:
    file_exchange(uint32_t rank, uint32_t world, std::string dir, int device) : rank_(rank), world_(world), dir_(std::move(dir)) {
        hip_check(hipSetDevice(device), "hipSetDevice");
    }
    void all_gather(uint64_t send, uint64_t recv, size_t bytes) override {
        std::vector<uint8_t> mine(bytes), all(bytes * world_);
        hip_check(hipMemcpy(mine.data(), (const void*)(uintptr_t)send, bytes, hipMemcpyDeviceToHost), "hipMemcpy");
        file_all_gather(dir_, sequence_++, rank_, world_, mine.data(), all.data(), bytes);
        hip_check(hipMemcpy((void*)(uintptr_t)recv, all.data(), all.size(), hipMemcpyHostToDevice), "hipMemcpy");
    }
    void all_reduce_sum(uint64_t address, size_t count) override {
        std::vector<float> data(count);
        hip_check(hipMemcpy(data.data(), (const void*)(uintptr_t)address, count * sizeof(float), hipMemcpyDeviceToHost), "hipMemcpy");
        file_all_reduce_sum(dir_, sequence_++, rank_, world_, data.data(), count);
        hip_check(hipMemcpy((void*)(uintptr_t)address, data.data(), count * sizeof(float), hipMemcpyHostToDevice), "hipMemcpy");
    }
    const char* transport() const override { return "file"; }

private:
    uint32_t rank_, world_, sequence_ = 0;
    std::string dir_;
};
}  // namespace

std::unique_ptr<device_exchange> device_exchange::create(const std::string& transport, uint32_t rank, uint32_t world, const std::string& dir, int device) {
    if (world == 0 || rank >= world) throw std::runtime_error("exchange: rank / world out of range");
    if (dir.empty()) throw std::runtime_error("exchange: a rendezvous directory shared by all ranks is needed");
    if (transport == "rccl") return std::unique_ptr<device_exchange>(new rccl_exchange(rank, world, dir, device));
    if (transport == "file") return std::unique_ptr<device_exchange>(new file_exchange(rank, world, dir, device));
    throw std::runtime_error("exchange: unknown transport '" + transport + "' (rccl, file)");
}
