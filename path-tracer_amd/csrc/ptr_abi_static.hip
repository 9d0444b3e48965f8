// C ABI of the render hot path (include/ptr_hip.h): the scratch of ptr_dispatch_static -- one fixed array per GPU instead of allocations -- and the entry point itself.
#include "ptr_abi_internal.h"

namespace ptr_abi {

static __device__ __align__(256) uint32_t scratch[kStaticWords];

constexpr int kMaxDevices = 64;
std::mutex g_static_mutex;
uint32_t* g_static_scratch[kMaxDevices] = {};   // device address of `scratch` on each GPU that has a context

// With `device` current: the address of this GPU's copy of the array.
hipError_t static_scratch_address(int device, uint32_t** out) {
    std::lock_guard<std::mutex> lock(g_static_mutex);
    if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
    if (!g_static_scratch[device]) {
        void* p = nullptr;
        const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return e;
        g_static_scratch[device] = (uint32_t*)p;
    }
    *out = g_static_scratch[device];
    return hipSuccess;
}

hipError_t bind_static_scratch(ptr_ctx* c) {
    const int device = c->device;
    hipError_t e = hipSuccess;
    {
        // this GPU's copy of the fixed scratch array, carved up like an allocated frame scratch
        uint32_t* p = nullptr;
        if ((e = static_scratch_address(device, &p)) != hipSuccess) return e;
        auto take = [&](size_t n) { uint32_t* r = p; p += n; return r; };
        ptr::WavefrontScratch& s = c->static_scratch;
        s.hits = (float4*)take((size_t)kStaticPaths * 4);
        s.samples = (float4*)take((size_t)kStaticPaths * 4);
        for (int g = 0; g < 2; g++) {
            s.ray_o[g] = (float4*)take((size_t)kStaticPaths * 4);
            s.ray_d[g] = (float4*)take((size_t)kStaticPaths * 4);
            s.path_t[g] = (float4*)take((size_t)kStaticPaths * 4);
        }
        s.hit_instance = (int32_t*)take(kStaticPaths);
        s.counters = take(64);
        s.stack_spill = (int32_t*)take(kStaticSpillWords);
        s.capacity = kStaticPaths;
        s.extend_blocks = kStaticExtendBlocks;
        s.spill_entries = kStaticSpill;
    }
    return hipSuccess;
}

}  // namespace ptr_abi

extern "C" {

int ptr_dispatch_static(ptr_ctx* c, const char* kernel, const uint8_t constants[64], uint32_t gx, uint32_t gy, uint32_t gz) {
    if (!c) return PTR_E_INVALID;
    if (!kernel || !constants) return fail(c, PTR_E_INVALID, "ptr_dispatch_static: null kernel name or constants");
    int rc = bind_device(c); if (rc) return rc;
    // running held-back dispatches here would allocate their scratch: the caller runs them first, where it may
    if (c->pending.active) return fail(c, PTR_E_STATE, "ptr_dispatch_static: dispatches are being held back (\"dispatch.coalesce\"); call ptr_flush or ptr_sync first");
    if (gx == 0 || gy == 0 || gz == 0) return PTR_OK;
    if ((uint64_t)gx * gy > 0x7fffffffull) return fail(c, PTR_E_RANGE, "ptr_dispatch_static: too many workgroups");
    if (!strcmp(kernel, "compute")) return dispatch_compute(c, constants, gx, gy, 0, true);
    if (!strcmp(kernel, "compute.megakernel")) return dispatch_compute(c, constants, gx, gy, 1, true);
    if (!strcmp(kernel, "compute.wavefront")) return dispatch_compute(c, constants, gx, gy, 2, true);
    return fail(c, PTR_E_INVALID, "ptr_dispatch_static: unknown kernel '%s' (compute, compute.megakernel, compute.wavefront)", kernel);
}

}  // extern "C"
