// GPU box: stand-alone probe -- how fast do FIVE planes stream (read-only, non-temporal 16 B per lane, like the sweep) as a
// function of how many loads a thread issues?  (The write side: one store per thread 6.9 TB/s, four 6.0 -- upsample_bench.hip.)
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 scripts/probes/read_bench.hip -o scripts/probes/_build/read_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float v4 __attribute__((ext_vector_type(4)));
struct Planes { const float* p[5]; };
// each block: 256 threads x UNR units of EVERY plane, contiguous per plane; NT: non-temporal loads
template <int UNR, bool NT>
__global__ __launch_bounds__(256) void k_read(Planes pl, size_t units, float* sink) {
    const size_t base = (size_t)blockIdx.x * (256 * UNR) + threadIdx.x;
    v4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
        const size_t i = base + (size_t)u * 256;
        if (i < units) {
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                const v4* q = reinterpret_cast<const v4*>(pl.p[k]) + i;
                acc += NT ? __builtin_nontemporal_load(q) : *q;
            }
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 1.2345e30f) sink[0] = acc.x;
}
template <int UNR, bool NT>
void run(const char* name, Planes pl, size_t units, float* sink, hipStream_t s, hipEvent_t e0, hipEvent_t e1, unsigned lds = 0) {   // lds: dynamic LDS bytes = a cap on the workgroups per CU
    const dim3 grid((unsigned)((units + 256 * UNR - 1) / (256 * UNR)));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_read<UNR, NT>), grid, dim3(256), lds, s, pl, units, sink);
    CK(hipStreamSynchronize(s));
    float sum = 0, best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((k_read<UNR, NT>), grid, dim3(256), lds, s, pl, units, sink);
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10; sum += ms; if (ms < best) best = ms;
    }
    const double bytes = 5.0 * units * 16;
    printf("  %-26s mean %8.1f us  best %8.1f us = %5.2f TB/s (%.3f of 8)\n", name, sum / 3 * 1e3, best * 1e3, bytes / (sum / 3 * 1e-3) / 1e12, bytes / (sum / 3 * 1e-3) / 8e12);
    fflush(stdout);
}
int main() {
    const size_t N = (size_t)1024 * 480 * 640, units = N / 4;      // one plane of the BASELINE batch
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int alloc = 0; alloc < 2; ++alloc) {
        Planes pl; float* keep[5]; float* sink;
        for (int k = 0; k < 5; ++k) { CK(hipMalloc(&keep[k], N * 4)); CK(hipMemsetAsync(keep[k], 0, N * 4, s)); pl.p[k] = keep[k]; }
        CK(hipMalloc(&sink, 16));
        printf("== allocation %d: 5 planes x %.0f MB read per launch (6291 MB)\n", alloc, N * 4 / 1e6);
        run<1, true>("1 unit / thread, nt", pl, units, sink, s, e0, e1);
        run<1, false>("1 unit / thread, plain", pl, units, sink, s, e0, e1);
        run<2, true>("2 units / thread, nt", pl, units, sink, s, e0, e1);
        run<4, true>("4 units / thread, nt", pl, units, sink, s, e0, e1);
        run<4, true>("4 units, nt, 4 WG / CU", pl, units, sink, s, e0, e1, 40000);
        run<4, true>("4 units, nt, 3 WG / CU", pl, units, sink, s, e0, e1, 51200);
        run<4, true>("4 units, nt, 2 WG / CU", pl, units, sink, s, e0, e1, 80000);
        run<8, true>("8 units / thread, nt", pl, units, sink, s, e0, e1);
        run<20, true>("20 units / thread, nt", pl, units, sink, s, e0, e1);
        run<40, true>("40 units / thread, nt", pl, units, sink, s, e0, e1);
        // (the planes stay allocated: the next round lands on other pages)
    }
    return 0;
}
