// gclm_pass.hip -- the fused per-pixel sweep (the HBM-bound hot kernel), gfx950.
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
//        => the 2 x P up-Jacobian is rank one: J_up = n s^T,  J^T J = s s^T,  J^T r = s (n . r)
//   LAT  e=1-k1 r2; P=e (u,v); ray=(P,1)/sqrt(|P|^2+1); s=ray.g; r_lat=sin(lat_data)-clamp(s)
//        ds/d(delta_k)=ray.T[:,k];  ds/df=h.(e w-2 k1 (u,v)(uv.w));  ds/dk1=h.(-r2 (u,v)),
//        h=(g_xy-s ray_xy)/sqrt(|P|^2+1),  w=(-u wfx,-v wfy)
//
// Mapping to the machine: grid = (chunks per image, B); a 256-thread workgroup (4 waves of 64)
// streams a contiguous run of float4 groups of one image with fully coalesced 16 B/lane loads
// (1 KiB per wave-instruction per plane), the 64-byte parameter block of the image is read with
// scalar loads (workgroup-uniform -> SGPRs), 16 fp32 accumulators per lane are reduced with
// wave64 DPP shuffles, then across the 4 waves through LDS, and ONE 64-byte partial record per
// workgroup is written (no atomics: bit-reproducible).  No MFMA: the contraction is N x P -> P x P
// with P <= 4.
#include "gclm_internal.h"

namespace gclm {

namespace {

// sin(x) for |x| <= pi/2 (odd minimax polynomial, |err| < 1.2e-7 in fp32).  Latitudes are
// asin(clamp(tanh)) outputs of the CNN head (geocalib.py:73-75) and therefore in range.
__device__ __forceinline__ float sin_halfpi(float x) {
    const float t = x * x;
    float p = 2.6000457182817627e-06f;
    p = fmaf(p, t, -0.00019806611817330122f);
    p = fmaf(p, t, 0.008333017118275166f);
    p = fmaf(p, t, -0.16666656732559204f);
    return fmaf(x * t, p, x);
}

// Scaled Huber on the squared residual x2 (lm_optimizer.py:61-87): returns cost, sets weight.
__device__ __forceinline__ float huber(float x2, float inv_a2, float a2, float& weight) {
    const float y = x2 * inv_a2;
    const float yy = y + 1e-8f;
    const float isx = __frsqrt_rn(yy);
    const float sx = yy * isx;
    const bool inl = y <= 1.0f;
    weight = inl ? 1.0f : fmaxf(isx, 1.1920928955078125e-07f);
    return (inl ? y : fmaf(2.0f, sx, -1.0f)) * a2;
}

struct HuberK {
    float inv_a2u, a2u, inv_a2l, a2l;
};

template <int MODEL, bool HAS_UP>
__device__ __forceinline__ void pixel_accumulate(const PBlock& P, const HuberK& hk, float xf, float yf,
                                                 float dux, float duy, float dlat, float cu, float cl,
                                                 float (&acc)[kNAcc]) {
    constexpr bool DIST = MODEL != GCLM_PINHOLE;
    const float u = (xf - P.cx) * P.ifx;
    const float v = (yf - P.cy) * P.ify;
    const float r2 = fmaf(u, u, v * v);
    const float wx = -u * P.wfx, wy = -v * P.wfy;      // d(uv)/d(focal parameter)
    const float uvw = fmaf(u, wx, v * wy);
    const float k1x2 = 2.0f * P.k1;

    if constexpr (HAS_UP) {
        const float px = fmaf(-P.gc, u, P.ga), py = fmaf(-P.gc, v, P.gb);
        // d p / d delta_k = (T0k - u T2k, T1k - v T2k)
        float a0 = fmaf(-u, P.T20, P.T00), b0 = fmaf(-v, P.T20, P.T10);
        float a1 = fmaf(-u, P.T21, P.T01), b1 = fmaf(-v, P.T21, P.T11);
        float qx = px, qy = py;
        float fx_ = -P.gc * wx, fy_ = -P.gc * wy;      // dq/df (pinhole)
        float kx = 0.f, ky = 0.f;
        if constexpr (DIST) {
            const float d = fmaf(P.k1, r2, 1.0f);
            const float t = fmaf(u, px, v * py);
            const float kt = k1x2 * t;
            qx = fmaf(d, px, kt * u);
            qy = fmaf(d, py, kt * v);
            // M z = d z + 2 k1 (u zx + v zy)(u,v)
            float m;
            m = k1x2 * fmaf(u, a0, v * b0); a0 = fmaf(d, a0, m * u); b0 = fmaf(d, b0, m * v);
            m = k1x2 * fmaf(u, a1, v * b1); a1 = fmaf(d, a1, m * u); b1 = fmaf(d, b1, m * v);
            const float mw = k1x2 * uvw;
            const float Mwx = fmaf(d, wx, mw * u), Mwy = fmaf(d, wy, mw * v);
            const float pw = fmaf(px, wx, py * wy);
            // dq/df = -c M w + 2 k1 [ p (uv.w) + t w + (u,v)(p.w) ]
            fx_ = fmaf(-P.gc, Mwx, k1x2 * (fmaf(px, uvw, fmaf(t, wx, u * pw))));
            fy_ = fmaf(-P.gc, Mwy, k1x2 * (fmaf(py, uvw, fmaf(t, wy, v * pw))));
            // dq/dk1 = r2 p + 2 t (u,v)
            kx = fmaf(r2, px, 2.0f * t * u);
            ky = fmaf(r2, py, 2.0f * t * v);
        }
        const float n2 = fmaxf(fmaf(qx, qx, qy * qy), 1e-24f);
        const float rn = __frsqrt_rn(n2);
        const float ux = qx * rn, uy = qy * rn;          // predicted up vector
        const float rx = dux - ux, ry = duy - uy;        // residual (lm_optimizer.py:266)
        const float x2 = fmaf(rx, rx, ry * ry);
        float wgt;
        float cost = huber(x2, hk.inv_a2u, hk.a2u, wgt);
        wgt *= cu;
        cost *= cu;
        acc[A_CU] += cost;
        // rank-one Jacobian: s_k = (n . dq_k)/|q|,  n = (-uy, ux);  rho = n . r
        const float nx = -uy * rn, ny = ux * rn;
        const float s0 = fmaf(nx, a0, ny * b0);
        const float s1 = fmaf(nx, a1, ny * b1);
        const float s2 = fmaf(nx, fx_, ny * fy_);
        const float rho = fmaf(-uy, rx, ux * ry);
        const float w0 = wgt * s0, w1 = wgt * s1, w2 = wgt * s2;
        acc[A_G0 + 0] = fmaf(w0, rho, acc[A_G0 + 0]);
        acc[A_G0 + 1] = fmaf(w1, rho, acc[A_G0 + 1]);
        acc[A_G0 + 2] = fmaf(w2, rho, acc[A_G0 + 2]);
        acc[A_H00 + 0] = fmaf(w0, s0, acc[A_H00 + 0]);
        acc[A_H00 + 1] = fmaf(w0, s1, acc[A_H00 + 1]);
        acc[A_H00 + 2] = fmaf(w0, s2, acc[A_H00 + 2]);
        acc[A_H00 + 4] = fmaf(w1, s1, acc[A_H00 + 4]);
        acc[A_H00 + 5] = fmaf(w1, s2, acc[A_H00 + 5]);
        acc[A_H00 + 7] = fmaf(w2, s2, acc[A_H00 + 7]);
        if constexpr (DIST) {
            const float s3 = fmaf(nx, kx, ny * ky);
            const float w3 = wgt * s3;
            acc[A_G0 + 3] = fmaf(w3, rho, acc[A_G0 + 3]);
            acc[A_H00 + 3] = fmaf(w0, s3, acc[A_H00 + 3]);
            acc[A_H00 + 6] = fmaf(w1, s3, acc[A_H00 + 6]);
            acc[A_H00 + 8] = fmaf(w2, s3, acc[A_H00 + 8]);
            acc[A_H00 + 9] = fmaf(w3, s3, acc[A_H00 + 9]);
        }
    }

    {   // latitude
        float Px = u, Py = v, e = 1.0f;
        if constexpr (DIST) {
            e = fmaf(-P.k1, r2, 1.0f);
            Px = e * u;
            Py = e * v;
        }
        const float nn = fmaf(Px, Px, fmaf(Py, Py, 1.0f));
        const float rnn = __frsqrt_rn(nn);
        const float rayx = Px * rnn, rayy = Py * rnn;   // rayz = rnn
        const float s = fmaf(rayx, P.ga, fmaf(rayy, P.gb, rnn * P.gc));
        const float sc = fminf(fmaxf(s, -1.0f + 1e-6f), 1.0f - 1e-6f);
        const float rl = sin_halfpi(dlat) - sc;          // lm_optimizer.py:262,270-271
        float wgt;
        float cost = huber(rl * rl, hk.inv_a2l, hk.a2l, wgt);
        wgt *= cl;
        cost *= cl;
        acc[A_CL] += cost;
        const float l0 = fmaf(rayx, P.T00, fmaf(rayy, P.T10, rnn * P.T20));
        const float l1 = fmaf(rayx, P.T01, fmaf(rayy, P.T11, rnn * P.T21));
        const float hx = fmaf(-s, rayx, P.ga) * rnn, hy = fmaf(-s, rayy, P.gb) * rnn;
        float dpx = wx, dpy = wy;
        if constexpr (DIST) {
            const float m = k1x2 * uvw;
            dpx = fmaf(e, wx, -m * u);
            dpy = fmaf(e, wy, -m * v);
        }
        const float l2 = fmaf(hx, dpx, hy * dpy);
        const float w0 = wgt * l0, w1 = wgt * l1, w2 = wgt * l2;
        acc[A_G0 + 0] = fmaf(w0, rl, acc[A_G0 + 0]);
        acc[A_G0 + 1] = fmaf(w1, rl, acc[A_G0 + 1]);
        acc[A_G0 + 2] = fmaf(w2, rl, acc[A_G0 + 2]);
        acc[A_H00 + 0] = fmaf(w0, l0, acc[A_H00 + 0]);
        acc[A_H00 + 1] = fmaf(w0, l1, acc[A_H00 + 1]);
        acc[A_H00 + 2] = fmaf(w0, l2, acc[A_H00 + 2]);
        acc[A_H00 + 4] = fmaf(w1, l1, acc[A_H00 + 4]);
        acc[A_H00 + 5] = fmaf(w1, l2, acc[A_H00 + 5]);
        acc[A_H00 + 7] = fmaf(w2, l2, acc[A_H00 + 7]);
        if constexpr (DIST) {
            const float l3 = -fmaf(hx, u, hy * v) * r2;
            const float w3 = wgt * l3;
            acc[A_G0 + 3] = fmaf(w3, rl, acc[A_G0 + 3]);
            acc[A_H00 + 3] = fmaf(w0, l3, acc[A_H00 + 3]);
            acc[A_H00 + 6] = fmaf(w1, l3, acc[A_H00 + 6]);
            acc[A_H00 + 8] = fmaf(w2, l3, acc[A_H00 + 8]);
            acc[A_H00 + 9] = fmaf(w3, l3, acc[A_H00 + 9]);
        }
    }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

template <int VEC>
struct Ld;
template <>
struct Ld<4> {
    using T = float4;
    static __device__ __forceinline__ T ld(const float* p, size_t unit) {
        return *reinterpret_cast<const float4*>(p + unit * 4);
    }
    static __device__ __forceinline__ float get(const T& v, int k) {
        return k == 0 ? v.x : (k == 1 ? v.y : (k == 2 ? v.z : v.w));
    }
    static __device__ __forceinline__ T ones() { return make_float4(1.f, 1.f, 1.f, 1.f); }
};
template <>
struct Ld<1> {
    using T = float;
    static __device__ __forceinline__ T ld(const float* p, size_t unit) { return p[unit]; }
    static __device__ __forceinline__ float get(const T& v, int) { return v; }
    static __device__ __forceinline__ T ones() { return 1.f; }
};

template <int MODEL, bool HAS_UP, bool HAS_UPC, bool HAS_LATC, int VEC>
__global__ __launch_bounds__(kBlock) void sweep_kernel(const SweepArgs a) {
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

    float acc[kNAcc];
#pragma unroll
    for (int i = 0; i < kNAcc; ++i) acc[i] = 0.f;

    int unit = u0 + tid;
    int pix = unit * VEC;
    int y = pix / a.W;
    int x = pix - y * a.W;
    const int step_pix = kBlock * VEC;
    const int dy = step_pix / a.W, dx = step_pix - dy * a.W;
    using L = Ld<VEC>;
    for (; unit < u1; unit += kBlock) {
        typename L::T vux, vuy, vcu = L::ones(), vcl = L::ones();
        if constexpr (HAS_UP) {
            vux = L::ld(upx, unit);
            vuy = L::ld(upy, unit);
        }
        const typename L::T vlat = L::ld(lat, unit);
        if constexpr (HAS_UP && HAS_UPC) vcu = L::ld(upc, unit);
        if constexpr (HAS_LATC) vcl = L::ld(latc, unit);
        const float yf = (float)y;
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            pixel_accumulate<MODEL, HAS_UP>(P, hk, (float)(x + k), yf, HAS_UP ? L::get(vux, k) : 0.f,
                                            HAS_UP ? L::get(vuy, k) : 0.f, L::get(vlat, k),
                                            L::get(vcu, k), L::get(vcl, k), acc);
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
        const float s = wave_sum(acc[i]);
        if (lane == 0) red[wave][i] = s;
    }
    __syncthreads();
    if (tid < kNAcc) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < kBlock / 64; ++w) s += red[w][tid];
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
