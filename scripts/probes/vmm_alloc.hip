// vmm_alloc.hip -- probe helper (GPU box): back the input fields of one batch with HIP virtual-memory-management
// allocations (hipMemCreate physical handles mapped into ONE reserved virtual range) so that the placement probe can
// choose HOW the physical backing is cut: one handle for the whole batch, one per tensor, or fixed-size chunks.
//   hipcc --offload-arch=gfx950 -O2 -shared -fPIC scripts/probes/vmm_alloc.hip -o scripts/probes/_build/libvmm_probe.so
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

struct vmm_block {
    void* base = nullptr;
    size_t span = 0;
    std::vector<hipMemGenericAllocationHandle_t> handles;
    std::vector<void*> plain;   // mode -1: hipMalloc per segment
};

static hipMemAllocationProp prop_for(int device) {
    hipMemAllocationProp p = {};
    p.type = hipMemAllocationTypePinned;
    p.location.type = hipMemLocationTypeDevice;
    p.location.id = device;
    return p;
}

#define VCHECK(call)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (call);                                                                        \
        if (e_ != hipSuccess) {                                                                        \
            fprintf(stderr, "vmm_probe: %s -> %s (%s:%d)\n", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return -1;                                                                                 \
        }                                                                                              \
    } while (0)

extern "C" {

size_t vmm_granularity(int device, int recommended) {
    hipMemAllocationProp p = prop_for(device);
    size_t g = 0;
    if (hipMemGetAllocationGranularity(&g, &p, recommended ? hipMemAllocationGranularityRecommended
                                                            : hipMemAllocationGranularityMinimum) != hipSuccess)
        return 0;
    return g;
}

// n_seg segments of sizes[i] bytes, each rounded up to `align` (>= granularity), mapped back to back into one virtual
// range.  mode -1: plain hipMalloc per segment (control); 0: ONE physical handle for everything; 1: one handle per
// segment; 2: handles of `chunk` bytes (the last one of a segment may be shorter).  seg_ptrs[i] = device address.
int vmm_alloc(int device, int n_seg, const size_t* sizes, int mode, size_t chunk, size_t align, void** seg_ptrs,
              vmm_block** out) {
    VCHECK(hipSetDevice(device));
    vmm_block* b = new vmm_block();
    *out = b;
    if (mode < 0) {
        for (int i = 0; i < n_seg; ++i) {
            void* p = nullptr;
            VCHECK(hipMalloc(&p, sizes[i]));
            b->plain.push_back(p);
            seg_ptrs[i] = p;
        }
        return 0;
    }
    hipMemAllocationProp prop = prop_for(device);
    size_t gran = vmm_granularity(device, 0);
    if (!gran) return -2;
    if (align < gran) align = gran;
    std::vector<size_t> padded(n_seg);
    size_t span = 0;
    for (int i = 0; i < n_seg; ++i) {
        padded[i] = (sizes[i] + align - 1) / align * align;
        span += padded[i];
    }
    VCHECK(hipMemAddressReserve(&b->base, span, align, nullptr, 0));
    b->span = span;
    char* va = static_cast<char*>(b->base);
    size_t off = 0;
    for (int i = 0; i < n_seg; ++i) {
        seg_ptrs[i] = va + off;
        off += padded[i];
    }
    auto map_one = [&](size_t at, size_t bytes) -> int {
        hipMemGenericAllocationHandle_t h;
        VCHECK(hipMemCreate(&h, bytes, &prop, 0));
        b->handles.push_back(h);
        VCHECK(hipMemMap(va + at, bytes, 0, h, 0));
        return 0;
    };
    if (mode == 0) {
        if (map_one(0, span)) return -1;
    } else if (mode == 1) {
        off = 0;
        for (int i = 0; i < n_seg; ++i) {
            if (map_one(off, padded[i])) return -1;
            off += padded[i];
        }
    } else {
        if (chunk < gran) chunk = gran;
        chunk = chunk / gran * gran;
        for (size_t at = 0; at < span; at += chunk)
            if (map_one(at, span - at < chunk ? span - at : chunk)) return -1;
    }
    hipMemAccessDesc acc = {};
    acc.location.type = hipMemLocationTypeDevice;
    acc.location.id = device;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    VCHECK(hipMemSetAccess(b->base, span, &acc, 1));
    return 0;
}

// write every byte of [p, p + bytes) with a device memset and wait: is the mapping itself usable?
int vmm_touch(void* p, size_t bytes) {
    VCHECK(hipMemset(p, 0, bytes));
    VCHECK(hipDeviceSynchronize());
    return 0;
}

int vmm_free(vmm_block* b) {
    if (!b) return 0;
    for (void* p : b->plain) (void)hipFree(p);
    if (b->base) {
        (void)hipMemUnmap(b->base, b->span);
        for (auto h : b->handles) (void)hipMemRelease(h);
        (void)hipMemAddressFree(b->base, b->span);
    }
    delete b;
    return 0;
}

}  // extern "C"
