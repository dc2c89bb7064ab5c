// Microbenchmark (GPU box): issue cost of the VALU instructions the sweep kernel is made of, on gfx950.
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/valu_probe.hip -o /tmp/valu_probe && /tmp/valu_probe
// Every wave runs ITER x 32 independent instructions of one kind (8 accumulator chains); reports
// shader cycles per instruction per SIMD at a given occupancy and the shader clock (s_memtime vs 100 MHz wall clock).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int ITER = 4096;

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define BODY4(OP) REP8(OP) REP8(OP) REP8(OP) REP8(OP)

template <int KIND>
__global__ __launch_bounds__(256) void probe(float* out, long long* cyc, float seed) {
    f2 a[8];
    float b[8];
    for (int i = 0; i < 8; ++i) { a[i] = f2{seed + i, seed - i}; b[i] = seed * (i + 1); }
    f2 x[8], z[8];
    float xs[8], zs[8];
    for (int i = 0; i < 8; ++i) { x[i] = f2{seed * i, seed + 2 * i}; z[i] = f2{seed - 3 * i, seed * 0.1f * i}; xs[i] = seed * 3 * i; zs[i] = seed - 5 * i; }
    const f2 k = f2{seed * 0.5f, seed * 0.25f};
    const float ks = seed * 0.5f;
    const long long t0 = clock64();
    const long long w0 = wall_clock64();
    for (int it = 0; it < ITER; ++it) {
        if constexpr (KIND == 0) {
#define OP(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(b[i]) : "v"(ks));
            BODY4(OP)
#undef OP
        } else if constexpr (KIND == 1) {
#define OP(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(k));
            BODY4(OP)
#undef OP
        } else if constexpr (KIND == 2) {
#define OP(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(k));
            BODY4(OP)
#undef OP
        } else if constexpr (KIND == 3) {
#define OP(i) asm volatile("v_min_f32 %0, %0, %1" : "+v"(b[i]) : "v"(ks));
            BODY4(OP)
#undef OP
        } else if constexpr (KIND == 4) {
#define OP(i) asm volatile("v_rsq_f32 %0, %0" : "+v"(b[i]));
            BODY4(OP)
#undef OP
        } else if constexpr (KIND == 5) {       // the sweep's mix: 4 pk_fma : 3 pk_mul : 1 scalar
#define OP(i) asm volatile("v_pk_fma_f32 %0, %0, %2, %2\n v_pk_mul_f32 %0, %0, %2\n v_min_f32 %1, %1, %3" : "+v"(a[i]), "+v"(b[i]) : "v"(k), "v"(ks));
            REP8(OP) REP8(OP)
#undef OP
#define OP(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(k));
            REP8(OP)
#undef OP
        } else if constexpr (KIND == 6) {
#define OP(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(k));
            BODY4(OP)
#undef OP
        } else if constexpr (KIND == 7) {       // pk_fma with a neg modifier + op_sel splat (as the compiler emits)
#define OP(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1 op_sel_hi:[1,0,0] neg_lo:[0,0,1] neg_hi:[0,0,1]" : "+v"(a[i]) : "v"(k));
            BODY4(OP)
#undef OP
        } else if constexpr (KIND == 8) {       // three distinct 64-bit sources (what real code looks like)
#define OP(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x[i]), "v"(z[i]));
            BODY4(OP)
#undef OP
        } else if constexpr (KIND == 9) {
#define OP(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(b[i]) : "v"(xs[i]), "v"(zs[i]));
            BODY4(OP)
#undef OP
        } else if constexpr (KIND == 10) {
#define OP(i) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(a[i]) : "v"(x[i]), "v"(z[i]));
            BODY4(OP)
#undef OP
        } else if constexpr (KIND == 11) {
#define OP(i) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(b[i]) : "v"(xs[i]), "v"(zs[i]));
            BODY4(OP)
#undef OP
        } else if constexpr (KIND == 12) {      // same FLOPs as KIND 8, unpacked
#define OP(i) asm volatile("v_fma_f32 %0, %2, %4, %0\n v_fma_f32 %1, %3, %5, %1" : "+v"(b[i]), "+v"(xs[i]) : "v"(zs[i]), "v"(zs[(i + 1) & 7]), "v"(a[i].x), "v"(a[i].y));
            REP8(OP) REP8(OP)
#undef OP
        }
    }
    const long long t1 = clock64();
    const long long w1 = wall_clock64();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += a[i].x + a[i].y + b[i] + x[i].x + z[i].y + xs[i] + zs[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { cyc[2 * blockIdx.x] = t1 - t0; cyc[2 * blockIdx.x + 1] = w1 - w0; }
}

template <int KIND>
void run(const char* name, int n_instr, int waves_per_simd) {
    const int blocks = 256 * waves_per_simd;   // 256 CUs x (4 waves per block = 1 per SIMD) x occupancy
    float* out; long long* cyc;
    hipMalloc(&out, sizeof(float) * blocks * 256);
    hipMalloc(&cyc, sizeof(long long) * 2 * blocks);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<KIND><<<blocks, 256>>>(out, cyc, 1.0001f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<KIND><<<blocks, 256>>>(out, cyc, 1.0001f);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(2 * blocks);
    hipMemcpy(h.data(), cyc, sizeof(long long) * 2 * blocks, hipMemcpyDeviceToHost);
    double c = 0, w = 0;
    for (int i = 0; i < blocks; ++i) { c += h[2 * i]; w += h[2 * i + 1]; }
    c /= blocks; w /= blocks;
    const double instr_per_wave = (double)ITER * n_instr;
    // s_memtime ticks per 100 MHz wall tick -> MHz of whatever s_memtime counts; event time gives issue rate
    const double wave_instr_per_simd = instr_per_wave * waves_per_simd;
    printf("%-28s occ %d: %8.3f ms  | %6.2f ns per wave-instr per SIMD | memtime/wall = %.3f (memtime %.0f, wall %.0f)\n",
           name, waves_per_simd, ms, ms * 1e6 / wave_instr_per_simd, c / w, c, w);
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int occ : {1, 4, 8}) {
        run<0>("v_fma_f32", 32, occ);
        run<1>("v_pk_fma_f32", 32, occ);
        run<7>("v_pk_fma_f32 (mods)", 32, occ);
        run<2>("v_pk_mul_f32", 32, occ);
        run<6>("v_pk_add_f32", 32, occ);
        run<3>("v_min_f32", 32, occ);
        run<4>("v_rsq_f32", 32, occ);
        run<5>("mix 24 pk_fma/16 pk_mul/16 min", 56, occ);
        run<8>("v_pk_fma_f32 3 distinct src", 32, occ);
        run<9>("v_fma_f32 3 distinct src", 32, occ);
        run<12>("2x v_fma_f32 (= 1 pk_fma)", 32, occ);
        run<10>("v_pk_mul_f32 distinct", 32, occ);
        run<11>("v_mul_f32 distinct", 32, occ);
    }
    return 0;
}
