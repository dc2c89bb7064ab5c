// GPU box: stand-alone A/B of mappings for the bilinear upsampling kernel (gclm_upsample_fields).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=fast-honor-pragmas scripts/probes/upsample_bench.hip -o scripts/probes/_build/upsample_bench
//   scripts/probes/_build/upsample_bench  > gpurun_out/r05/upsample_bench.log
// Variants (same per-value formulas as csrc/gclm_update.hip upsample_strip; every output is compared with variant "r04"):
//   r04        grid (strip, 4 row groups per block, plane), a wave = 64 float4 units x 8 rows   (the round-4 kernel)
//   band*      a BLOCK owns whole rows (a band of consecutive output rows of one plane), its waves walk the
//              (strip, row group) tasks of the band: the partial 128-byte lines between two strips of a ragged row are
//              written by the same CU within microseconds (one XCD's L2 can merge them) and a block's output is contiguous
//   *pref      all source rows of a task loaded up front (memory-level parallelism), then the same arithmetic
//   *plain     plain stores instead of non-temporal
//   *xcd       blockIdx remapped so that one XCD walks a contiguous range of bands
//   MODE 1 / 2 store-only / load-only (diagnostic)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <string>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float v4 __attribute__((ext_vector_type(4)));
typedef float v4u __attribute__((ext_vector_type(4), aligned(4)));

template <bool PLAIN>
__device__ __forceinline__ void store4(float* p, v4 o) {
    if constexpr (PLAIN) *reinterpret_cast<v4*>(p) = o;
    else __builtin_nontemporal_store(o, reinterpret_cast<v4*>(p));
}

struct ColTerms {
    int xs, i0[4], i1[4];
    float lx[4];
    bool live;
};
__device__ __forceinline__ ColTerms col_terms(int Xu, int w, int W, float sx) {
    ColTerms c;
    c.live = Xu * 4 < W;
    int x0[4], x1[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int X = min(Xu * 4 + k, W - 1);
        const float fx = fmaxf(((float)X + 0.5f) * sx - 0.5f, 0.f);
        x0[k] = min((int)fx, w - 1);
        x1[k] = min(x0[k] + 1, w - 1);
        c.lx[k] = fx - (float)x0[k];
    }
    c.xs = min(x0[0], w - 4);
#pragma unroll
    for (int k = 0; k < 4; ++k) { c.i0[k] = x0[k] - c.xs; c.i1[k] = x1[k] - c.xs; }
    return c;
}
__device__ __forceinline__ float pick(const v4u& t, int i) { return i == 0 ? t.x : (i == 1 ? t.y : (i == 2 ? t.z : t.w)); }
__device__ __forceinline__ void hinterp(const ColTerms& c, const v4u& t, float (&o)[4]) {
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = pick(t, c.i0[k]) * (1.f - c.lx[k]) + pick(t, c.i1[k]) * c.lx[k];
}

// one wave: 64 float4 units x ROWS output rows starting at (Xu, Y0); WIDE case only (w >= 4, 3 w <= 2 W)
template <int ROWS, bool PREF, bool PLAIN, int MODE>
__device__ __forceinline__ void strip(const float* __restrict__ s, float* __restrict__ d, int h, int w, int H, int W, int Xu, int Y0) {
    const float sy = (float)h / (float)H, sx = (float)w / (float)W;
    const ColTerms c = col_terms(Xu, w, W, sx);
    const int Yend = min(Y0 + ROWS, H);
    auto emit = [&](int Y, float ly, const float (&ha)[4], const float (&hb)[4]) {
        const v4 o = {ha[0] * (1.f - ly) + hb[0] * ly, ha[1] * (1.f - ly) + hb[1] * ly, ha[2] * (1.f - ly) + hb[2] * ly,
                      ha[3] * (1.f - ly) + hb[3] * ly};
        if (MODE == 2) { if (c.live && o.x == 1.2345e30f) store4<PLAIN>(d + (size_t)Y * W + Xu * 4, o); }
        else if (c.live) store4<PLAIN>(d + (size_t)Y * W + Xu * 4, o);
    };
    auto loadrow = [&](int y) -> v4u {
        if (MODE == 1) { const float f = (float)(y + c.xs); return v4u{f, f + 1.f, f + 2.f, f + 3.f}; }
        return *reinterpret_cast<const v4u*>(s + (size_t)y * w + c.xs);
    };
    if constexpr (PREF) {
        constexpr int RMAX = 2 * ROWS / 3 + 3;
        const float fyA = fmaxf(((float)Y0 + 0.5f) * sy - 0.5f, 0.f), fyB = fmaxf(((float)(Yend - 1) + 0.5f) * sy - 0.5f, 0.f);
        const int ylo = min((int)fyA, h - 1), yhi = min(min((int)fyB, h - 1) + 1, h - 1);
        const int n = yhi - ylo + 1;
        v4u t[RMAX];
#pragma unroll
        for (int r = 0; r < RMAX; ++r) t[r] = loadrow(min(ylo + r, yhi));
        float hc[4], hn[4];
        hinterp(c, t[0], hc);
        int Y = Y0;
#pragma unroll
        for (int r = 0; r < RMAX; ++r) {
            if (r < n) {
                if (r + 1 < RMAX && r + 1 < n) hinterp(c, t[r + 1 < RMAX ? r + 1 : r], hn);
                else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) hn[k] = hc[k];
                }
                while (Y < Yend) {
                    const float fy = fmaxf(((float)Y + 0.5f) * sy - 0.5f, 0.f);
                    const int y0 = min((int)fy, h - 1);
                    if (y0 - ylo != r) break;
                    emit(Y, fy - (float)y0, hc, hn);
                    ++Y;
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) hc[k] = hn[k];
            }
        }
    } else {
        int ya = -1, yb = -1;
        float ha[4] = {0.f, 0.f, 0.f, 0.f}, hb[4] = {0.f, 0.f, 0.f, 0.f};
        for (int Y = Y0; Y < Yend; ++Y) {
            const float fy = fmaxf(((float)Y + 0.5f) * sy - 0.5f, 0.f);
            const int y0 = min((int)fy, h - 1), y1 = min(y0 + 1, h - 1);
            const float ly = fy - (float)y0;
            if (!(y0 == ya && y1 == yb)) {
                if (y0 == yb) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) ha[k] = hb[k];
                } else if (y0 != ya) {
                    hinterp(c, loadrow(y0), ha);
                }
                ya = y0;
                if (y1 == y0) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) hb[k] = ha[k];
                } else {
                    hinterp(c, loadrow(y1), hb);
                }
                yb = y1;
            }
            emit(Y, ly, ha, hb);
        }
    }
}

template <int ROWS, bool PREF, bool PLAIN, int MODE>
__global__ __launch_bounds__(256) void k_tiled(const float* __restrict__ src, int planes, int h, int w, int H, int W, float* __restrict__ dst) {
    const int Xu = blockIdx.x * 64 + (threadIdx.x & 63), Y0 = (blockIdx.y * 4 + (threadIdx.x >> 6)) * ROWS;
    for (int p = blockIdx.z; p < planes; p += gridDim.z)
        strip<ROWS, PREF, PLAIN, MODE>(src + (size_t)p * h * w, dst + (size_t)p * H * W, h, w, H, W, Xu, Y0);
}

template <int ROWS, bool PREF, bool PLAIN, int MODE, bool XCD>
__global__ __launch_bounds__(256) void k_band(const float* __restrict__ src, int h, int w, int H, int W, int band_rows, int bands_per_plane,
                                              int nstrips, float* __restrict__ dst) {
    int blk = blockIdx.x;
    if constexpr (XCD) {
        const int nblk = gridDim.x, q = nblk >> 3, rem = nblk & 7, x = blk & 7, idx = blk >> 3;
        blk = x * q + min(x, rem) + idx;
    }
    const int p = blk / bands_per_plane, band = blk - p * bands_per_plane;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int ntasks = nstrips * (band_rows / ROWS);
    for (int t = wave; t < ntasks; t += 4) {
        const int rg = t / nstrips, st = t - rg * nstrips;
        const int Y0 = band * band_rows + rg * ROWS;
        if (Y0 >= H) break;
        strip<ROWS, PREF, PLAIN, MODE>(src + (size_t)p * h * w, dst + (size_t)p * H * W, h, w, H, W, st * 64 + lane, Y0);
    }
}

// ------------------------------------------------------------------------------------------------ round-5 kernel
// v2: (1) the wave index is made scalar (readfirstlane), so the row terms and the row-reuse branches are SALU, not exec-masked
// VALU; (2) the eight register selects per source row become a 5-term FMA chain over the 4-float window with per-lane
// weights (10 v_pk_fma instead of 24 v_cmp + 24 v_cndmask); (3) a row's units are ROTATED by the row's phase so that every
// full wave store starts on a 128-byte line: position q of a row maps to unit (q + o) mod Wu, o = units up to the next line.
// Rows 8 apart share the phase (8 Wu = 0 mod 8 units), so in PHASED mode wave i of a 512-thread block owns rows
// Y0 + i + 8 j and keeps its column weights; CONSEC mode (rows that are a whole number of lines) keeps the vertical reuse.
struct Shape;
typedef float v2f __attribute__((ext_vector_type(2)));
struct ColW {
    int xs, u;            // window start (source floats), output unit of this lane
    v2f W[4][2], E[2];    // weights of window float j for outputs (0,1) and (2,3); E: the clamped right edge (x1 == x0)
    bool live;
};
__device__ __forceinline__ ColW col_weights(int q, int o, int w, int W, float sx) {
    ColW c;
    const int Wu = W >> 2;
    int u = q + o;
    if (u >= Wu) u -= Wu;
    c.live = q < Wu;
    if (!c.live) u = 0;
    c.u = u;
    int x0[4], x1[4];
    float lx[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int X = u * 4 + k;
        const float fx = fmaxf(__fmaf_rn((float)X + 0.5f, sx, -0.5f), 0.f);
        x0[k] = min((int)fx, w - 1);
        x1[k] = min(x0[k] + 1, w - 1);
        lx[k] = fx - (float)x0[k];
    }
    c.xs = min(x0[0], w - 4);
    float Wm[4][4], Em[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i0 = x0[k] - c.xs, i1 = x1[k] - c.xs;
#pragma unroll
        for (int j = 0; j < 4; ++j) Wm[j][k] = (j == i1) ? lx[k] : ((j == i0) ? 1.f - lx[k] : 0.f);
        Em[k] = (i0 == i1) ? 1.f - lx[k] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { c.W[j][0] = v2f{Wm[j][0], Wm[j][1]}; c.W[j][1] = v2f{Wm[j][2], Wm[j][3]}; }
    c.E[0] = v2f{Em[0], Em[1]}; c.E[1] = v2f{Em[2], Em[3]};
    return c;
}
// out[k] = fma(t[i0], 1 - lx, t[i1] * lx): descending chain, every other term is an exact zero
__device__ __forceinline__ void hinterp2(const ColW& c, const v4u& t, v2f (&o)[2]) {
#pragma unroll
    for (int hlf = 0; hlf < 2; ++hlf) {
        v2f a = v2f{t.w, t.w} * c.W[3][hlf];
        a = __builtin_elementwise_fma(v2f{t.w, t.w}, c.E[hlf], a);
        a = __builtin_elementwise_fma(v2f{t.z, t.z}, c.W[2][hlf], a);
        a = __builtin_elementwise_fma(v2f{t.y, t.y}, c.W[1][hlf], a);
        a = __builtin_elementwise_fma(v2f{t.x, t.x}, c.W[0][hlf], a);
        o[hlf] = a;
    }
}
struct RowT { int y0, y1; float ly; };
__device__ __forceinline__ RowT row_terms(int Y, int h, float sy) {      // Y wave-uniform
    const float fy = fmaxf(__fmaf_rn((float)Y + 0.5f, sy, -0.5f), 0.f);
    const int y0 = min((int)fy, h - 1);
    RowT r;
    r.ly = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, fy - (float)y0)));
    r.y0 = __builtin_amdgcn_readfirstlane(y0);
    r.y1 = min(r.y0 + 1, h - 1);
    return r;
}
template <bool PLAIN>
__device__ __forceinline__ void vblend_store(float* p, bool live, float ly, const v2f (&ha)[2], const v2f (&hb)[2]) {
    const v2f l2 = v2f{ly, ly}, m2 = v2f{1.f - ly, 1.f - ly};
    const v2f o0 = __builtin_elementwise_fma(ha[0], m2, hb[0] * l2), o1 = __builtin_elementwise_fma(ha[1], m2, hb[1] * l2);
    if (live) store4<PLAIN>(p, v4{o0.x, o0.y, o1.x, o1.y});
}
__device__ __forceinline__ int plane_phase(const float* d) { return (int)((8u - (unsigned)((reinterpret_cast<uintptr_t>(d) >> 4) & 7u)) & 7u); }

// CONSEC: rows Y0 .. Y0+ROWS-1 of one wave, source rows reused; o = the plane's phase (rows are whole lines)
template <int ROWS, bool PREF, bool PLAIN>
__device__ __forceinline__ void strip2_consec(const float* __restrict__ s, float* __restrict__ d, int h, int w, int H, int W, int q, int Y0) {
    const float sy = (float)h / (float)H, sx = (float)w / (float)W;
    const ColW c = col_weights(q, plane_phase(d), w, W, sx);
    const int Yend = min(Y0 + ROWS, H);
    const float* sc = s + c.xs;
    float* dc = d + c.u * 4;
    if constexpr (PREF) {
        constexpr int RMAX = 2 * ROWS / 3 + 3;
        const RowT ra = row_terms(Y0, h, sy), rb = row_terms(Yend - 1, h, sy);
        const int ylo = ra.y0, yhi = rb.y1, n = yhi - ylo + 1;
        v4u t[RMAX];
#pragma unroll
        for (int r = 0; r < RMAX; ++r) t[r] = *reinterpret_cast<const v4u*>(sc + (size_t)min(ylo + r, yhi) * w);
        v2f hc[2], hn[2];
        hinterp2(c, t[0], hc);
        int Y = Y0;
#pragma unroll
        for (int r = 0; r < RMAX; ++r) {
            if (r < n) {
                if (r + 1 < RMAX && r + 1 < n) hinterp2(c, t[r + 1 < RMAX ? r + 1 : r], hn);
                else { hn[0] = hc[0]; hn[1] = hc[1]; }
                while (Y < Yend) {
                    const RowT rt = row_terms(Y, h, sy);
                    if (rt.y0 - ylo != r) break;
                    vblend_store<PLAIN>(dc + (size_t)Y * W, c.live, rt.ly, hc, hn);
                    ++Y;
                }
                hc[0] = hn[0]; hc[1] = hn[1];
            }
        }
    } else {
        int ya = -1, yb = -1;
        v2f ha[2] = {v2f{0.f, 0.f}, v2f{0.f, 0.f}}, hb[2] = {v2f{0.f, 0.f}, v2f{0.f, 0.f}};
        for (int Y = Y0; Y < Yend; ++Y) {
            const RowT rt = row_terms(Y, h, sy);
            if (!(rt.y0 == ya && rt.y1 == yb)) {
                if (rt.y0 == yb) { ha[0] = hb[0]; ha[1] = hb[1]; }
                else if (rt.y0 != ya) hinterp2(c, *reinterpret_cast<const v4u*>(sc + (size_t)rt.y0 * w), ha);
                ya = rt.y0;
                if (rt.y1 == rt.y0) { hb[0] = ha[0]; hb[1] = ha[1]; }
                else hinterp2(c, *reinterpret_cast<const v4u*>(sc + (size_t)rt.y1 * w), hb);
                yb = rt.y1;
            }
            vblend_store<PLAIN>(dc + (size_t)Y * W, c.live, rt.ly, ha, hb);
        }
    }
}
template <int ROWS, bool PREF, bool PLAIN>
__global__ __launch_bounds__(256) void k2_consec(const float* __restrict__ src, int planes, int h, int w, int H, int W, float* __restrict__ dst) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q = blockIdx.x * 64 + (threadIdx.x & 63), Y0 = (blockIdx.y * 4 + wave) * ROWS;
    if (Y0 >= H) return;
    for (int p = blockIdx.z; p < planes; p += gridDim.z)
        strip2_consec<ROWS, PREF, PLAIN>(src + (size_t)p * h * w, dst + (size_t)p * H * W, h, w, H, W, q, Y0);
}
// ROWBLOCK: the four waves of a block take four ADJACENT strips of the same ROWS rows (a block writes whole rows of up to
// 1024 px, row after row) instead of four row groups of one strip
template <int ROWS, bool PREF, bool PLAIN>
__global__ __launch_bounds__(256) void k2_rowblock(const float* __restrict__ src, int planes, int h, int w, int H, int W, float* __restrict__ dst) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q = (blockIdx.x * 4 + wave) * 64 + (threadIdx.x & 63), Y0 = blockIdx.y * ROWS;
    if ((blockIdx.x * 4 + wave) * 64 >= (W >> 2)) return;
    for (int p = blockIdx.z; p < planes; p += gridDim.z)
        strip2_consec<ROWS, PREF, PLAIN>(src + (size_t)p * h * w, dst + (size_t)p * H * W, h, w, H, W, q, Y0);
}
template <int ROWS, bool PREF, bool PLAIN>
void launch2_rowblock(const Shape& g, const float* s, float* d, int, hipStream_t st);
// PHASED: wave i of a 512-thread block owns rows Y0 + i + 8 j (j < ROWS): equal phase, so the column weights are shared
template <int ROWS, bool PLAIN>
__global__ __launch_bounds__(512) void k2_phased(const float* __restrict__ src, int planes, int h, int w, int H, int W, float* __restrict__ dst) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q = blockIdx.x * 64 + (threadIdx.x & 63), Ya = blockIdx.y * (8 * ROWS) + wave;
    if (Ya >= H) return;
    const float sy = (float)h / (float)H, sx = (float)w / (float)W;
    const int Wu = W >> 2;
    for (int p = blockIdx.z; p < planes; p += gridDim.z) {
        const float* s = src + (size_t)p * h * w;
        float* d = dst + (size_t)p * H * W;
        const unsigned g = (unsigned)((reinterpret_cast<uintptr_t>(d) >> 4) & 7u) + (unsigned)Ya * (unsigned)Wu;
        const ColW c = col_weights(q, (int)((8u - (g & 7u)) & 7u), w, W, sx);
        const float* sc = s + c.xs;
        float* dc = d + c.u * 4;
        v4u ta[ROWS], tb[ROWS];
        RowT rt[ROWS];
#pragma unroll
        for (int j = 0; j < ROWS; ++j) {
            rt[j] = row_terms(min(Ya + 8 * j, H - 1), h, sy);
            ta[j] = *reinterpret_cast<const v4u*>(sc + (size_t)rt[j].y0 * w);
            tb[j] = *reinterpret_cast<const v4u*>(sc + (size_t)rt[j].y1 * w);
        }
#pragma unroll
        for (int j = 0; j < ROWS; ++j) {
            v2f ha[2], hb[2];
            hinterp2(c, ta[j], ha);
            hinterp2(c, tb[j], hb);
            const int Y = Ya + 8 * j;
            vblend_store<PLAIN>(dc + (size_t)Y * W, c.live && Y < H, rt[j].ly, ha, hb);
        }
    }
}
// PHASED + ROWBLOCK: 8 waves = 2 rows x 4 adjacent strips; the wave owns rows Ya + 8 j like k2_phased
template <int ROWS, bool PLAIN, int SW>      // SW: strips per block (4 with 2 rows per block, 8 with 1)
__global__ __launch_bounds__(512) void k2_phased_rb(const float* __restrict__ src, int planes, int h, int w, int H, int W, float* __restrict__ dst) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int RW = 8 / SW;                               // rows (phases) per block
    const int strip = blockIdx.x * SW + (wave % SW), r = wave / SW;
    const int groups = 8 / RW;                               // blocks along y that together cover the 8 phases
    const int Ya = (blockIdx.y / groups) * (8 * ROWS) + (blockIdx.y % groups) * RW + r;
    if (strip * 64 >= (W >> 2) || Ya >= H) return;
    const int q = strip * 64 + (threadIdx.x & 63);
    const float sy = (float)h / (float)H, sx = (float)w / (float)W;
    const int Wu = W >> 2;
    for (int p = blockIdx.z; p < planes; p += gridDim.z) {
        const float* s = src + (size_t)p * h * w;
        float* d = dst + (size_t)p * H * W;
        const unsigned g = (unsigned)((reinterpret_cast<uintptr_t>(d) >> 4) & 7u) + (unsigned)Ya * (unsigned)Wu;
        const ColW c = col_weights(q, (int)((8u - (g & 7u)) & 7u), w, W, sx);
        const float* sc = s + c.xs;
        float* dc = d + c.u * 4;
        v4u ta[ROWS], tb[ROWS];
        RowT rt[ROWS];
#pragma unroll
        for (int j = 0; j < ROWS; ++j) {
            rt[j] = row_terms(min(Ya + 8 * j, H - 1), h, sy);
            ta[j] = *reinterpret_cast<const v4u*>(sc + (size_t)rt[j].y0 * w);
            tb[j] = *reinterpret_cast<const v4u*>(sc + (size_t)rt[j].y1 * w);
        }
#pragma unroll
        for (int j = 0; j < ROWS; ++j) {
            v2f ha[2], hb[2];
            hinterp2(c, ta[j], ha);
            hinterp2(c, tb[j], hb);
            const int Y = Ya + 8 * j;
            vblend_store<PLAIN>(dc + (size_t)Y * W, c.live && Y < H, rt[j].ly, ha, hb);
        }
    }
}
template <int ROWS, bool PLAIN, int SW>
void launch2_phased_rb(const Shape& g, const float* s, float* d, int, hipStream_t st);
template <int ROWS, bool PREF, bool PLAIN>
void launch2_consec(const Shape& g, const float* s, float* d, int, hipStream_t st);
template <int ROWS, bool PLAIN>
void launch2_phased(const Shape& g, const float* s, float* d, int, hipStream_t st);

// flat float4 fill of the output, non-temporal: the write ceiling of this box
template <bool PLAIN>
__global__ __launch_bounds__(256) void k_fill(float* dst, size_t units) {
    const v4 o = {1.f, 2.f, 3.f, 4.f};
    for (size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x; i < units && i < (size_t)(blockIdx.x + 1) * 1024; i += 256)
        store4<PLAIN>(dst + 4 * i, o);
}

struct Shape { int planes, h, w, H, W; const char* name; };

struct Variant {
    std::string name;
    void (*launch)(const Shape&, const float*, float*, int band_rows, hipStream_t);
    int band_rows;
};

template <int ROWS, bool PREF, bool PLAIN, int MODE>
void launch_tiled(const Shape& g, const float* s, float* d, int, hipStream_t st) {
    const dim3 grid((g.W / 4 + 63) / 64, (g.H + 4 * ROWS - 1) / (4 * ROWS), g.planes < 65535 ? g.planes : 65535);
    hipLaunchKernelGGL((k_tiled<ROWS, PREF, PLAIN, MODE>), grid, dim3(256), 0, st, s, g.planes, g.h, g.w, g.H, g.W, d);
}
template <int ROWS, bool PREF, bool PLAIN, int MODE, bool XCD>
void launch_band(const Shape& g, const float* s, float* d, int band_rows, hipStream_t st) {
    const int bpp = (g.H + band_rows - 1) / band_rows, nstrips = (g.W / 4 + 63) / 64;
    hipLaunchKernelGGL((k_band<ROWS, PREF, PLAIN, MODE, XCD>), dim3(bpp * g.planes), dim3(256), 0, st, s, g.h, g.w, g.H, g.W, band_rows, bpp,
                       nstrips, d);
}
// fill variants (what makes torch's fill_ 8-16 % faster than the fill above on the same box?): one constant value,
// UNR float4 units per thread with the block's units contiguous, the driver's own memset
template <bool PLAIN, int UNR, bool CONST1>
__global__ __launch_bounds__(256) void k_fill2(float* dst, size_t units) {
    const v4 o = CONST1 ? v4{1.f, 1.f, 1.f, 1.f} : v4{1.f, 2.f, 3.f, 4.f};
    const size_t base = (size_t)blockIdx.x * (256 * UNR) + threadIdx.x;
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
        const size_t i = base + (size_t)u * 256;
        if (i < units) store4<PLAIN>(dst + 4 * i, o);
    }
}
template <bool PLAIN, int UNR, bool CONST1>
void launch_fill2(const Shape& g, const float*, float* d, int, hipStream_t st) {
    const size_t units = (size_t)g.planes * g.H * g.W / 4;
    hipLaunchKernelGGL((k_fill2<PLAIN, UNR, CONST1>), dim3((unsigned)((units + 256 * UNR - 1) / (256 * UNR))), dim3(256), 0, st, d, units);
}
void launch_memset(const Shape& g, const float*, float* d, int, hipStream_t st) {
    (void)hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(d), 0x3f800000, (size_t)g.planes * g.H * g.W, st);
}
template <bool PLAIN>
void launch_fill(const Shape& g, const float*, float* d, int, hipStream_t st) {
    const size_t units = (size_t)g.planes * g.H * g.W / 4;
    hipLaunchKernelGGL(k_fill<PLAIN>, dim3((unsigned)((units + 1023) / 1024)), dim3(256), 0, st, d, units);
}

template <int ROWS, bool PREF, bool PLAIN>
void launch2_consec(const Shape& g, const float* s, float* d, int, hipStream_t st) {
    const dim3 grid((g.W / 4 + 63) / 64, (g.H + 4 * ROWS - 1) / (4 * ROWS), g.planes < 65535 ? g.planes : 65535);
    hipLaunchKernelGGL((k2_consec<ROWS, PREF, PLAIN>), grid, dim3(256), 0, st, s, g.planes, g.h, g.w, g.H, g.W, d);
}
template <int ROWS, bool PLAIN, int SW>
void launch2_phased_rb(const Shape& g, const float* s, float* d, int, hipStream_t st) {
    const int groups = SW;      // 8 / (8 / SW)
    const dim3 grid((g.W / 4 + 64 * SW - 1) / (64 * SW), ((g.H + 8 * ROWS - 1) / (8 * ROWS)) * groups, g.planes < 65535 ? g.planes : 65535);
    hipLaunchKernelGGL((k2_phased_rb<ROWS, PLAIN, SW>), grid, dim3(512), 0, st, s, g.planes, g.h, g.w, g.H, g.W, d);
}
template <int ROWS, bool PREF, bool PLAIN>
void launch2_rowblock(const Shape& g, const float* s, float* d, int, hipStream_t st) {
    const dim3 grid((g.W / 4 + 255) / 256, (g.H + ROWS - 1) / ROWS, g.planes < 65535 ? g.planes : 65535);
    hipLaunchKernelGGL((k2_rowblock<ROWS, PREF, PLAIN>), grid, dim3(256), 0, st, s, g.planes, g.h, g.w, g.H, g.W, d);
}
template <int ROWS, bool PLAIN>
void launch2_phased(const Shape& g, const float* s, float* d, int, hipStream_t st) {
    const dim3 grid((g.W / 4 + 63) / 64, (g.H + 8 * ROWS - 1) / (8 * ROWS), g.planes < 65535 ? g.planes : 65535);
    hipLaunchKernelGGL((k2_phased<ROWS, PLAIN>), grid, dim3(512), 0, st, s, g.planes, g.h, g.w, g.H, g.W, d);
}

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 20;
    const Shape shapes[] = {{1280, 240, 320, 480, 640, "B256 240x320->480x640"},
                            {320, 320, 480, 1080, 1620, "B64 320x480->1080x1620"},
                            {5, 320, 480, 1080, 1620, "B1 320x480->1080x1620"},
                            {5, 240, 320, 480, 640, "B1 240x320->480x640"}};
    std::vector<Variant> vs = {
        {"fill (write ceiling)", launch_fill<false>, 0},
        {"fill plain stores", launch_fill<true>, 0},
        {"fill2 plain x4 const 1.0", launch_fill2<true, 4, true>, 0},
        {"fill2 nt x4 const 1.0", launch_fill2<false, 4, true>, 0},
        {"fill2 plain x4 1,2,3,4", launch_fill2<true, 4, false>, 0},
        {"fill2 plain x1 const", launch_fill2<true, 1, true>, 0},
        {"fill2 plain x8 const", launch_fill2<true, 8, true>, 0},
        {"fill2 plain x16 const", launch_fill2<true, 16, true>, 0},
        {"hipMemsetD32Async", launch_memset, 0},
        {"r04", launch_tiled<8, false, false, 0>, 0},
        {"r04 store-only", launch_tiled<8, false, false, 1>, 0},
        {"r04 load-only", launch_tiled<8, false, false, 2>, 0},
        {"r04 plain", launch_tiled<8, false, true, 0>, 0},
        {"tiled pref", launch_tiled<8, true, false, 0>, 0},
        {"v2 consec8", launch2_consec<8, false, false>, 0},
        {"v2 consec8 pref", launch2_consec<8, true, false>, 0},
        {"v2 consec8 pref plain", launch2_consec<8, true, true>, 0},
        {"v2 consec4", launch2_consec<4, false, false>, 0},
        {"v2 consec4 pref", launch2_consec<4, true, false>, 0},
        {"v2 consec16 pref", launch2_consec<16, true, false>, 0},
        {"v2 rowblock8 pref", launch2_rowblock<8, true, false>, 0},
        {"v2 rowblock4 pref", launch2_rowblock<4, true, false>, 0},
        {"v2 rowblock2 pref", launch2_rowblock<2, true, false>, 0},
        {"v2 rowblock4 pref plain", launch2_rowblock<4, true, true>, 0},
        {"v2 phased_rb4 x4strips", launch2_phased_rb<4, false, 4>, 0},
        {"v2 phased_rb8 x4strips", launch2_phased_rb<8, false, 4>, 0},
        {"v2 phased_rb4 x8strips", launch2_phased_rb<4, false, 8>, 0},
        {"v2 phased_rb8 x8strips", launch2_phased_rb<8, false, 8>, 0},
        {"v2 phased2", launch2_phased<2, false>, 0},
        {"v2 phased4", launch2_phased<4, false>, 0},
        {"v2 phased4 plain", launch2_phased<4, true>, 0},
        {"v2 phased8", launch2_phased<8, false>, 0},
        {"band32", launch_band<8, false, false, 0, false>, 32},
        {"band32 pref", launch_band<8, true, false, 0, false>, 32},
        {"band8 pref", launch_band<8, true, false, 0, false>, 8},
    };
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (const Shape& g : shapes) {
        const size_t ns = (size_t)g.planes * g.h * g.w, nd = (size_t)g.planes * g.H * g.W;
        float *src, *dst, *ref;
        CK(hipMalloc(&src, ns * 4)); CK(hipMalloc(&dst, nd * 4)); CK(hipMalloc(&ref, nd * 4));
        std::vector<float> hs(ns);
        unsigned x = 12345u;
        for (size_t i = 0; i < ns; ++i) { x = x * 1664525u + 1013904223u; hs[i] = (float)(x >> 8) * (1.f / 8388608.f) - 1.f; }
        CK(hipMemcpy(src, hs.data(), ns * 4, hipMemcpyHostToDevice));
        launch_tiled<8, false, false, 0>(g, src, ref, 0, st);
        CK(hipStreamSynchronize(st));
        const size_t ncheck = nd < (size_t)8 * g.H * g.W ? nd : (size_t)8 * g.H * g.W;     // first planes + last plane
        std::vector<float> hr(ncheck), hr2((size_t)g.H * g.W), ho(ncheck), ho2((size_t)g.H * g.W);
        CK(hipMemcpy(hr.data(), ref, ncheck * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hr2.data(), ref + nd - (size_t)g.H * g.W, (size_t)g.H * g.W * 4, hipMemcpyDeviceToHost));
        const double bytes = 4.0 * (double)(ns + nd);
        printf("== %s  (%d planes, %.1f MB algorithmic)\n", g.name, g.planes, bytes / 1e6);
        for (const Variant& v : vs) {
            CK(hipMemsetAsync(dst, 0xff, nd * 4, st));
            for (int i = 0; i < 3; ++i) v.launch(g, src, dst, v.band_rows, st);
            CK(hipStreamSynchronize(st));
            float best = 1e30f, sum = 0.f;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipEventRecord(e0, st));
                for (int i = 0; i < reps; ++i) v.launch(g, src, dst, v.band_rows, st);
                CK(hipEventRecord(e1, st));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                ms /= reps; sum += ms; if (ms < best) best = ms;
            }
            CK(hipGetLastError());
            const bool full = v.name.find("only") == std::string::npos && v.name.find("fill") == std::string::npos && v.name.find("Memset") == std::string::npos;
            double maxd = -1.0;
            if (full) {
                CK(hipMemcpy(ho.data(), dst, ncheck * 4, hipMemcpyDeviceToHost));
                CK(hipMemcpy(ho2.data(), dst + nd - (size_t)g.H * g.W, (size_t)g.H * g.W * 4, hipMemcpyDeviceToHost));
                maxd = 0.0;
                for (size_t i = 0; i < ncheck; ++i) { const double dd = std::fabs((double)ho[i] - (double)hr[i]); if (!(dd <= maxd)) maxd = std::isnan(dd) ? 1e30 : dd; }
                for (size_t i = 0; i < ho2.size(); ++i) { const double dd = std::fabs((double)ho2[i] - (double)hr2[i]); if (!(dd <= maxd)) maxd = std::isnan(dd) ? 1e30 : dd; }
            }
            printf("  %-22s mean %8.2f us  best %8.2f us  = %5.2f TB/s (frac of 8: %.3f)   max|d vs r04| %g\n", v.name.c_str(), sum / 3 * 1e3, best * 1e3,
                   bytes / (sum / 3 * 1e-3) / 1e12, bytes / (sum / 3 * 1e-3) / 8e12, maxd);
            fflush(stdout);
        }
        CK(hipFree(src)); CK(hipFree(dst)); CK(hipFree(ref));
    }
    return 0;
}
