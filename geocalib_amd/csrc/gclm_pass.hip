// gclm_pass.hip -- the fused per-pixel sweep (the hot kernel), gfx950.
//
// One launch evaluates, for every image of the batch at its current parameters, in ONE pass over
// the 3..5 input planes:
//   * the perspective-field prediction (up vector, sin latitude)       perspective_fields.py:47-81,185-211
//   * residuals and scaled-Huber costs / weights x confidences         lm_optimizer.py:248-315
//   * the analytic Jacobian rows wrt (d1, d2, focal[, k1])             perspective_fields.py:84-182,214-275
//   * the reductions sum w J^T r and sum w J^T J                       lm_optimizer.py:317-385
// The reference materialises (B,N,2,P) Jacobians and ~100x the algorithmic bytes; here nothing
// per-pixel ever leaves registers.
//
// Closed form used (checked against the oracle's literal matrix chains):
//   u=(x-cx)/fx, v=(y-cy)/fy, r2=u^2+v^2
//   UP   p=(a-c u, b-c v); d=1+k1 r2; t=u px+v py; q=d p+2 k1 t (u,v)  (pinhole: q=p); up=q/|q|
//        d(up)/d(theta) = n (n . dq/dtheta)/|q| with n=(-up_y, up_x)  [I - up up^T = n n^T in 2-D]
//        => the 2 x P up-Jacobian is rank one: J_up = n s^T,  J^T J = s s^T,  J^T r = s (n . r);
//        dq/dtheta = M dp/dtheta + ..., M = d I + 2 k1 uv uv^T symmetric => s_k = (M nq).dp/dtheta_k
//   LAT  e=1-k1 r2; P=e (u,v); ray=(P,1)/sqrt(|P|^2+1); s=ray.g; r_lat=sin(lat_data)-clamp(s)
//        ds/d(delta_k)=ray.T[:,k];  ds/df=h.(e w-2 k1 (u,v)(uv.w));  ds/dk1=h.(-r2 (u,v)),
//        h=(g_xy-s ray_xy)/sqrt(|P|^2+1),  w=(-u wfx,-v wfy)
//
// Mapping to the machine: grid = (chunks per image, B); a 256-thread workgroup (4 waves of 64)
// streams a contiguous run of float4 groups of one image with fully coalesced 16 B/lane loads
// (1 KiB per wave-instruction per plane); the 64-byte parameter block of the image is read with
// scalar loads (workgroup-uniform -> SGPRs); 16 accumulators per lane are reduced with wave64
// shuffles, then across the 4 waves through LDS, and ONE 64-byte partial record per workgroup is
// written (no atomics: bit-reproducible).  No MFMA: the contraction is N x P -> P x P with P <= 4.
//
// Arithmetic: PMC counters (profiles/r01_pmc_sq_*) show the sweep is VALU-issue-bound as soon as
// it approaches ~6 TB/s (scalar fp32 FMA sustains ~1 wave64 instruction / 3 cycles / SIMD), so the
// float4 path computes PIXEL PAIRS in packed fp32 (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32:
// two pixels per instruction) -- written explicitly on a 2-wide vector type so the pairs live in
// adjacent registers straight out of the dwordx4 loads (LLVM's SLP vectoriser finds some of these
// pairs on its own but pays for them with register shuffles and 2x the VGPRs; it is switched off for
// this file, see the Makefile).
#include "gclm_internal.h"

#ifndef GCLM_MIN_WAVES
#define GCLM_MIN_WAVES 1
#endif
#ifndef GCLM_NT_LOADS
#define GCLM_NT_LOADS 1
#endif

namespace gclm {

namespace {

typedef float f2 __attribute__((ext_vector_type(2)));

// ---- lane-vector helpers: the per-pixel math below is written once for F = float (scalar path)
// ---- and F = f2 (two horizontally adjacent pixels per lane, packed fp32)
__device__ __forceinline__ float vfma(float a, float b, float c) { return fmaf(a, b, c); }
__device__ __forceinline__ f2 vfma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ float vrsq(float a) { return __frsqrt_rn(a); }
__device__ __forceinline__ f2 vrsq(f2 a) { return f2{__frsqrt_rn(a.x), __frsqrt_rn(a.y)}; }
__device__ __forceinline__ float vmax(float a, float b) { return fmaxf(a, b); }
__device__ __forceinline__ f2 vmax(f2 a, f2 b) { return f2{fmaxf(a.x, b.x), fmaxf(a.y, b.y)}; }
__device__ __forceinline__ float vclamp(float a, float lo, float hi) { return fminf(fmaxf(a, lo), hi); }
__device__ __forceinline__ f2 vclamp(f2 a, float lo, float hi) {
    return f2{fminf(fmaxf(a.x, lo), hi), fminf(fmaxf(a.y, lo), hi)};
}
// select(y <= 1, a, b)
__device__ __forceinline__ float vsel_le1(float y, float a, float b) { return y <= 1.0f ? a : b; }
__device__ __forceinline__ f2 vsel_le1(f2 y, f2 a, f2 b) {
    return f2{y.x <= 1.0f ? a.x : b.x, y.y <= 1.0f ? a.y : b.y};
}
__device__ __forceinline__ float vsplat(float, float s) { return s; }
__device__ __forceinline__ f2 vsplat(f2, float s) { return f2{s, s}; }
__device__ __forceinline__ float hsum(float a) { return a; }
__device__ __forceinline__ float hsum(f2 a) { return a.x + a.y; }

// sin(x) for |x| <= pi/2 (odd minimax polynomial, |err| < 1.2e-7 in fp32).  Latitudes are
// asin(clamp(tanh)) outputs of the CNN head (geocalib.py:73-75) and therefore in range.
template <typename F>
__device__ __forceinline__ F sin_halfpi(F x) {
    const F t = x * x;
    F p = vsplat(x, 2.6000457182817627e-06f);
    p = vfma(p, t, vsplat(x, -0.00019806611817330122f));
    p = vfma(p, t, vsplat(x, 0.008333017118275166f));
    p = vfma(p, t, vsplat(x, -0.16666656732559204f));
    return vfma(x * t, p, x);
}

// Scaled Huber on the squared residual x2 (lm_optimizer.py:61-87), in units of a^2: returns
// cost / a^2 (the a^2 factor is applied once per workgroup) and sets the weight.
// The reference's max(eps, 1/sqrt(y)) only matters for y > 7e13; residuals here are bounded
// (|r_up| <= 2, |r_lat| <= 2 => y <= 4/a^2), so it is the identity and is dropped.
template <typename F>
__device__ __forceinline__ F huber(F x2, float inv_a2, F& weight) {
    const F y = x2 * inv_a2;
    const F yy = y + 1e-8f;
    const F isx = vrsq(yy);
    weight = vsel_le1(y, vsplat(y, 1.0f), isx);
    return vsel_le1(y, y, vfma(2.0f * yy, isx, vsplat(y, -1.0f)));
}

struct HuberK {
    float inv_a2u, a2u, inv_a2l, a2l;
};

template <int MODEL, bool HAS_UP, typename F>
__device__ __forceinline__ void pixel_accumulate(const PBlock& P, const HuberK& hk, F xf, float yf, F dux, F duy,
                                                 F dlat, F cu, F cl, F (&acc)[kNAcc]) {
    constexpr bool DIST = MODEL != GCLM_PINHOLE;
    const F u = (xf - P.cx) * P.ifx;
    const float v = (yf - P.cy) * P.ify;                 // one image row per tile: v is lane-scalar
    const F r2 = vfma(u, u, vsplat(u, v * v));
    const F wx = u * (-P.wfx);                           // d(uv)/d(focal parameter) = (wx, wy)
    const float wy = -v * P.wfy;
    const F uvw = vfma(u, wx, vsplat(u, v * wy));
    const float k1x2 = 2.0f * P.k1;

    if constexpr (HAS_UP) {
        const F px = vfma(u, vsplat(u, -P.gc), vsplat(u, P.ga));
        const float py = fmaf(-P.gc, v, P.gb);
        F qx = px, qy = vsplat(u, py), d = vsplat(u, 1.0f), t = vsplat(u, 0.f);
        if constexpr (DIST) {
            d = vfma(r2, vsplat(u, P.k1), vsplat(u, 1.0f));
            t = vfma(u, px, vsplat(u, v * py));
            const F kt = t * k1x2;
            qx = vfma(d, px, kt * u);
            qy = vfma(d, vsplat(u, py), kt * v);
        }
        const F n2 = vmax(vfma(qx, qx, qy * qy), vsplat(u, 1e-24f));
        const F rn = vrsq(n2);
        const F ux = qx * rn, uy = qy * rn;              // predicted up vector
        const F rx = dux - ux, ry = duy - uy;            // residual (lm_optimizer.py:266)
        const F x2 = vfma(rx, rx, ry * ry);
        F wgt;
        const F cost = huber(x2, hk.inv_a2u, wgt);
        wgt = wgt * cu;
        acc[A_CU] = vfma(cost, cu, acc[A_CU]);
        // rank-one Jacobian: s_k = (M nq) . dp/dtheta_k, nq = n/|q|, n = (-uy, ux);  rho = n . r
        const F nx = -uy * rn, ny = ux * rn;
        const F nuv = vfma(nx, u, ny * v);
        const F nw = vfma(nx, wx, ny * wy);
        F mx = nx, my = ny, muv = nuv, mw = nw;
        if constexpr (DIST) {
            const F c2 = nuv * k1x2;
            mx = vfma(d, nx, c2 * u);
            my = vfma(d, ny, c2 * v);
            muv = vfma(mx, u, my * v);
            mw = vfma(mx, wx, my * wy);
        }
        // dp/ddelta_k = (T0k - u T2k, T1k - v T2k)  =>  s_k = m.T[0:2,k] - (m.uv) T2k
        const F s0 = vfma(mx, vsplat(u, P.T00), vfma(my, vsplat(u, P.T10), muv * (-P.T20)));
        const F s1 = vfma(mx, vsplat(u, P.T01), vfma(my, vsplat(u, P.T11), muv * (-P.T21)));
        F s2 = mw * (-P.gc);                             // dp/df = -c w
        [[maybe_unused]] F s3 = vsplat(u, 0.f);
        if constexpr (DIST) {
            // + 2 k1 [ p (uv.w) + t w + (u,v)(p.w) ] . nq          (perspective_fields.py:146-153)
            const F np_ = vfma(nx, px, ny * py);
            const F pw = vfma(px, wx, vsplat(u, py * wy));
            s2 = vfma(vfma(np_, uvw, vfma(t, nw, nuv * pw)), vsplat(u, k1x2), s2);
            s3 = vfma(r2, np_, (t * 2.0f) * nuv);        // dq/dk1 = r2 p + 2 t (u,v)   (:170-180)
        }
        const F rho = vfma(ux, ry, -(uy * rx));
        const F w0 = wgt * s0, w1 = wgt * s1, w2 = wgt * s2;
        acc[A_G0 + 0] = vfma(w0, rho, acc[A_G0 + 0]);
        acc[A_G0 + 1] = vfma(w1, rho, acc[A_G0 + 1]);
        acc[A_G0 + 2] = vfma(w2, rho, acc[A_G0 + 2]);
        acc[A_H00 + 0] = vfma(w0, s0, acc[A_H00 + 0]);
        acc[A_H00 + 1] = vfma(w0, s1, acc[A_H00 + 1]);
        acc[A_H00 + 2] = vfma(w0, s2, acc[A_H00 + 2]);
        acc[A_H00 + 4] = vfma(w1, s1, acc[A_H00 + 4]);
        acc[A_H00 + 5] = vfma(w1, s2, acc[A_H00 + 5]);
        acc[A_H00 + 7] = vfma(w2, s2, acc[A_H00 + 7]);
        if constexpr (DIST) {
            const F w3 = wgt * s3;
            acc[A_G0 + 3] = vfma(w3, rho, acc[A_G0 + 3]);
            acc[A_H00 + 3] = vfma(w0, s3, acc[A_H00 + 3]);
            acc[A_H00 + 6] = vfma(w1, s3, acc[A_H00 + 6]);
            acc[A_H00 + 8] = vfma(w2, s3, acc[A_H00 + 8]);
            acc[A_H00 + 9] = vfma(w3, s3, acc[A_H00 + 9]);
        }
    }

    {   // latitude
        F Px = u, Py = vsplat(u, v), e = vsplat(u, 1.0f);
        if constexpr (DIST) {
            e = vfma(r2, vsplat(u, -P.k1), vsplat(u, 1.0f));
            Px = e * u;
            Py = e * v;
        }
        const F nn = vfma(Px, Px, vfma(Py, Py, vsplat(u, 1.0f)));
        const F rnn = vrsq(nn);
        const F rayx = Px * rnn, rayy = Py * rnn;        // rayz = rnn
        const F s = vfma(rayx, vsplat(u, P.ga), vfma(rayy, vsplat(u, P.gb), rnn * P.gc));
        const F sc = vclamp(s, -1.0f + 1e-6f, 1.0f - 1e-6f);
        const F rl = sin_halfpi(dlat) - sc;              // lm_optimizer.py:262,270-271
        F wgt;
        const F cost = huber(rl * rl, hk.inv_a2l, wgt);
        wgt = wgt * cl;
        acc[A_CL] = vfma(cost, cl, acc[A_CL]);
        const F l0 = vfma(rayx, vsplat(u, P.T00), vfma(rayy, vsplat(u, P.T10), rnn * P.T20));
        const F l1 = vfma(rayx, vsplat(u, P.T01), vfma(rayy, vsplat(u, P.T11), rnn * P.T21));
        const F hx = vfma(-s, rayx, vsplat(u, P.ga)) * rnn, hy = vfma(-s, rayy, vsplat(u, P.gb)) * rnn;
        // ds/df = h.(e w - 2 k1 (u,v)(uv.w)),  ds/dk1 = h.(-r2 (u,v))     (perspective_fields.py:255-272)
        const F hw = vfma(hx, wx, hy * wy);
        F l2 = hw;
        [[maybe_unused]] F hu = vsplat(u, 0.f);
        if constexpr (DIST) {
            hu = vfma(hx, u, hy * v);
            l2 = vfma(e, hw, -((uvw * k1x2) * hu));
        }
        const F w0 = wgt * l0, w1 = wgt * l1, w2 = wgt * l2;
        acc[A_G0 + 0] = vfma(w0, rl, acc[A_G0 + 0]);
        acc[A_G0 + 1] = vfma(w1, rl, acc[A_G0 + 1]);
        acc[A_G0 + 2] = vfma(w2, rl, acc[A_G0 + 2]);
        acc[A_H00 + 0] = vfma(w0, l0, acc[A_H00 + 0]);
        acc[A_H00 + 1] = vfma(w0, l1, acc[A_H00 + 1]);
        acc[A_H00 + 2] = vfma(w0, l2, acc[A_H00 + 2]);
        acc[A_H00 + 4] = vfma(w1, l1, acc[A_H00 + 4]);
        acc[A_H00 + 5] = vfma(w1, l2, acc[A_H00 + 5]);
        acc[A_H00 + 7] = vfma(w2, l2, acc[A_H00 + 7]);
        if constexpr (DIST) {
            const F l3 = -(hu * r2);
            const F w3 = wgt * l3;
            acc[A_G0 + 3] = vfma(w3, rl, acc[A_G0 + 3]);
            acc[A_H00 + 3] = vfma(w0, l3, acc[A_H00 + 3]);
            acc[A_H00 + 6] = vfma(w1, l3, acc[A_H00 + 6]);
            acc[A_H00 + 8] = vfma(w2, l3, acc[A_H00 + 8]);
            acc[A_H00 + 9] = vfma(w3, l3, acc[A_H00 + 9]);
        }
    }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Per-lane tile of one loop iteration: VEC = 4 -> one float4 per plane, processed as two packed
// pixel pairs; VEC = 1 -> one pixel, scalar math (odd widths / unaligned pointers).
template <int VEC>
struct Lane;
template <>
struct Lane<4> {
    using F = f2;
    using V = float4;
    static constexpr int kPairs = 2;
    static __device__ __forceinline__ V ld(const float* p, size_t unit) {
#if GCLM_NT_LOADS
        // every byte is read exactly once per sweep: stream it past the caches (global_load ... nt)
        typedef float v4 __attribute__((ext_vector_type(4)));
        const v4 t = __builtin_nontemporal_load(reinterpret_cast<const v4*>(p + unit * 4));
        return make_float4(t.x, t.y, t.z, t.w);
#else
        return *reinterpret_cast<const float4*>(p + unit * 4);
#endif
    }
    static __device__ __forceinline__ F get(const V& v, int k) { return k == 0 ? f2{v.x, v.y} : f2{v.z, v.w}; }
    static __device__ __forceinline__ V ones() { return make_float4(1.f, 1.f, 1.f, 1.f); }
    static __device__ __forceinline__ F xcoord(int x, int k) {
        const float x0 = (float)(x + 2 * k);
        return f2{x0, x0 + 1.0f};
    }
};
template <>
struct Lane<1> {
    using F = float;
    using V = float;
    static constexpr int kPairs = 1;
    static __device__ __forceinline__ V ld(const float* p, size_t unit) { return p[unit]; }
    static __device__ __forceinline__ F get(const V& v, int) { return v; }
    static __device__ __forceinline__ V ones() { return 1.f; }
    static __device__ __forceinline__ F xcoord(int x, int) { return (float)x; }
};

template <int MODEL, bool HAS_UP, bool HAS_UPC, bool HAS_LATC, int VEC>
__global__ __launch_bounds__(kBlock, GCLM_MIN_WAVES) void sweep_kernel(const SweepArgs a) {
    if (a.skip_if_stopped && a.ctrl->stopped) return;   // batch-global early stop, no host sync
    const int b = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
    const PBlock P = a.pb[b];                            // workgroup-uniform -> scalar loads
    HuberK hk;
    hk.a2u = a.up_scale * a.up_scale;
    hk.inv_a2u = 1.0f / hk.a2u;
    hk.a2l = a.lat_scale * a.lat_scale;
    hk.inv_a2l = 1.0f / hk.a2l;

    const size_t N = (size_t)a.H * a.W;
    const int units = (int)(N / VEC);
    const int u0 = chunk * a.units_per_block;
    const int u1 = min(u0 + a.units_per_block, units);
    const float* upx = HAS_UP ? a.up + (size_t)b * 2 * N : nullptr;
    const float* upy = HAS_UP ? upx + N : nullptr;
    const float* lat = a.lat + (size_t)b * N;
    const float* upc = HAS_UPC ? a.upc + (size_t)b * N : nullptr;
    const float* latc = HAS_LATC ? a.latc + (size_t)b * N : nullptr;

    using L = Lane<VEC>;
    using F = typename L::F;
    using V = typename L::V;
    F acc[kNAcc];
#pragma unroll
    for (int i = 0; i < kNAcc; ++i) acc[i] = F(0.f);

    int unit = u0 + tid;
    int pix = unit * VEC;
    int y = pix / a.W;
    int x = pix - y * a.W;
    const int step_pix = kBlock * VEC;
    const int dy = step_pix / a.W, dx = step_pix - dy * a.W;
    for (; unit < u1; unit += kBlock) {
        V vux, vuy, vcu = L::ones(), vcl = L::ones();
        if constexpr (HAS_UP) {
            vux = L::ld(upx, unit);
            vuy = L::ld(upy, unit);
        }
        const V vlat = L::ld(lat, unit);
        if constexpr (HAS_UP && HAS_UPC) vcu = L::ld(upc, unit);
        if constexpr (HAS_LATC) vcl = L::ld(latc, unit);
        const float yf = (float)y;
#pragma unroll
        for (int k = 0; k < L::kPairs; ++k) {
            pixel_accumulate<MODEL, HAS_UP, F>(P, hk, L::xcoord(x, k), yf, HAS_UP ? L::get(vux, k) : F(0.f),
                                               HAS_UP ? L::get(vuy, k) : F(0.f), L::get(vlat, k), L::get(vcu, k),
                                               L::get(vcl, k), acc);
        }
        x += dx;
        y += dy;
        if (x >= a.W) {
            x -= a.W;
            ++y;
        }
    }

    // wave64 butterfly, then the 4 waves through LDS; one 64-byte record per workgroup
    __shared__ float red[kBlock / 64][kNAcc];
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int i = 0; i < kNAcc; ++i) {
        const float s = wave_sum(hsum(acc[i]));
        if (lane == 0) red[wave][i] = s;
    }
    __syncthreads();
    if (tid < kNAcc) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < kBlock / 64; ++w) s += red[w][tid];
        if (tid == A_CU) s *= hk.a2u;          // costs were accumulated in units of a^2
        if (tid == A_CL) s *= hk.a2l;
        a.partials[((size_t)b * a.nchunks + chunk) * kNAcc + tid] = s;
    }
}

template <int MODEL, int VEC>
hipError_t dispatch(const SweepArgs& a, hipStream_t s) {
    const dim3 grid(a.nchunks, a.B), block(kBlock);
    const bool up = a.up != nullptr, upc = up && a.upc != nullptr, latc = a.latc != nullptr;
#define GCLM_LAUNCH(U, UC, LC) \
    hipLaunchKernelGGL((sweep_kernel<MODEL, U, UC, LC, VEC>), grid, block, 0, s, a)
    if (up) {
        if (upc) { if (latc) GCLM_LAUNCH(true, true, true); else GCLM_LAUNCH(true, true, false); }
        else     { if (latc) GCLM_LAUNCH(true, false, true); else GCLM_LAUNCH(true, false, false); }
    } else {
        if (latc) GCLM_LAUNCH(false, false, true); else GCLM_LAUNCH(false, false, false);
    }
#undef GCLM_LAUNCH
    return hipGetLastError();
}

}  // namespace

hipError_t launch_sweep(int camera_model, const SweepArgs& a, hipStream_t s) {
    if (a.B <= 0) return hipSuccess;
    if (camera_model == GCLM_PINHOLE)
        return a.vec == 4 ? dispatch<GCLM_PINHOLE, 4>(a, s) : dispatch<GCLM_PINHOLE, 1>(a, s);
    if (camera_model == GCLM_SIMPLE_RADIAL)
        return a.vec == 4 ? dispatch<GCLM_SIMPLE_RADIAL, 4>(a, s) : dispatch<GCLM_SIMPLE_RADIAL, 1>(a, s);
    return hipErrorInvalidValue;
}

}  // namespace gclm
