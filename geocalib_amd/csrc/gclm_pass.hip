// gclm_pass.hip -- the fused per-pixel sweep (the hot kernel), gfx950.
//
// One launch evaluates, for every image of the batch at its current parameters, in ONE pass over
// the 3..5 input planes:
//   * the perspective-field prediction (up vector, sin latitude)       perspective_fields.py:47-81,185-211
//   * residuals and scaled-Huber costs / weights x confidences         lm_optimizer.py:248-315
//   * the analytic Jacobian rows wrt (d1, d2, focal[, k1[, k2]])       perspective_fields.py:84-182,214-275
//   * the reductions sum w J^T r and sum w J^T J                       lm_optimizer.py:317-385
// The reference materialises (B,N,2,P) Jacobians and ~100x the algorithmic bytes; here nothing
// per-pixel ever leaves registers.
//
// Closed form (SURVEY.md section 8-A generalised to any radial model; checked against the oracle's
// literal matrix chains).  Every camera model of camera.py is radial: distort(p) = p s(r2),
// undistort(p) = p tau(r2).  With s1 = ds/dr2, s2 = d2s/dr2^2, tau1 = dtau/dr2:
//   u=(x-cx)/fx, v=(y-cy)/fy, r2=u^2+v^2
//   UP   p=(a-c u, b-c v); t=u px+v py; q = M p, M = s I + 2 s1 uv uv^T (symmetric); up=q/|q|
//        d(up)/d(theta) = n (n . dq/dtheta)/|q| with n=(-up_y, up_x)  [I - up up^T = n n^T in 2-D]
//        => the 2 x P up-Jacobian is rank one: J_up = n s^T,  J^T J = s s^T,  J^T r = s (n . r);
//        with nq = n/|q| and m = M nq:   s_k  = m . dp/ddelta_k,
//        s_f  = -c (m.w) + 2 s1 [(nq.p)(uv.w) + (nq.uv)(p.w)] + t [jd (nq.w) + 4 s2 (uv.w)(nq.uv)],
//        s_kj = (ds/dk_j)(nq.p) + 2 (ds1/dk_j) t (nq.uv)            (jd = 2 s1: diagonal of d(off)/duv)
//   LAT  P=tau (u,v); ray=(P,1)/sqrt(|P|^2+1); s=ray.g; r_lat=sin(lat_data)-clamp(s)
//        ds/d(delta_k)=ray.T[:,k];  ds/df = tau (h.w) + 2 tau1 (uv.w)(h.uv);  ds/dk_j = (dtau/dk_j)(h.uv),
//        h=(g_xy-s ray_xy)/sqrt(|P|^2+1),  w=(-u wfx,-v wfy)
//        (evaluated through dot products, the ray is never formed: ray.x = (tau/n)(x_xy.uv) + x_z/n)
//   log-focal loop sweeps (LOGF): w = -(u,v), so x.w = -(x.uv) and the focal columns collapse, see
//   pixel_accumulate_fast.
//   pinhole: s = tau = 1.  simple_radial: s = 1+k1 r2, tau = 1-k1 r2.  radial: s = 1+k1 r2+k2 r4,
//   tau = 1-k1 r2+(3k1^2-k2) r4.  simple_divisional: s = (1-sqrt(1-4k r2))/(2k r2), tau = 1/(1+k r2)
//   with the reference's own guards (camera.py:829-940).
//
// Mapping to the machine: grid = (workgroups per image, B), 256-thread workgroups (4 waves of 64).  Every wave is a
// "job": 64 consecutive lanes of a column-stationary tile (rpi rows x one row-wide strip of float4 groups) that walks down
// the image, so a lane never changes its column: the column terms leave the loop, a wave-instruction still reads 1 KiB of
// consecutive addresses per plane (fully coalesced, non-temporal 16 B/lane loads, uniform base + one 32-bit offset for
// all planes) -- see sweep_kernel.  The 80-byte parameter block of the image is read with scalar loads
// (workgroup-uniform -> SGPRs); 16 (24 for `radial`) accumulators per lane are reduced per wave with DPP adds (no LDS),
// then across the 4 waves through LDS, and ONE partial record per workgroup is written (no atomics:
// bit-reproducible).  No MFMA: the contraction is N x P -> P x P, P <= 5.
//
// Arithmetic: the distortion models are VALU-heavy enough to pull the chip's clock down (power), so every
// flop taken out of the pixel body counts (scripts/probes/valu_probe.hip: a packed op costs ~1.8x a scalar one, i.e. the unit
// is FLOPs at 32 lanes/SIMD/cycle); the float4 path computes PIXEL PAIRS in packed fp32 (v_pk_fma_f32 /
// v_pk_mul_f32 / v_pk_add_f32: two pixels per instruction) -- written explicitly on a 2-wide vector
// type so the pairs live in adjacent registers straight out of the dwordx4 loads (LLVM's SLP
// vectoriser finds some of these pairs on its own but pays for them with register shuffles and 2x
// the VGPRs; it is switched off for this file, see the Makefile).
#include <type_traits>

#include "gclm_internal.h"
#include "gclm_device.h"

// waves per SIMD the float4 sweeps are held to (launch bounds of sweep_kernel); measurement builds override them
#ifndef GCLM_MIN_WAVES
#define GCLM_MIN_WAVES 1
#endif
#ifndef GCLM_DIV_WAVES
#define GCLM_DIV_WAVES 3
#endif
#ifndef GCLM_PINHOLE_WAVES
#define GCLM_PINHOLE_WAVES 6
#endif
#ifndef GCLM_RADIAL_WAVES
#define GCLM_RADIAL_WAVES 3
#endif
#ifndef GCLM_REUSE_TNUV
#define GCLM_REUSE_TNUV 1      // radial / simple_divisional, log-focal: the distortion columns of the up field reuse t (n.uv) and
#endif                         // r2 (n.p) + 2 t (n.uv), which the focal column has formed (3 / 1 packed operations per pixel pair less)
#ifndef GCLM_PINHOLE_LDS
#define GCLM_PINHOLE_LDS 51200 // dynamic LDS bytes of pinhole's float4 sweep launches = THREE workgroups per CU = 3 waves per SIMD where its
#endif                         // 80 VGPRs allow 6: nothing is stored there -- the memory system streams better under fewer concurrent
                               // waves (the no-math build: 6.89 TB/s at 6-8 waves, 6.97 at 3, 7.03 at 2, r06_variant_occupancy.log) and
                               // pinhole's arithmetic still hides at 3 (-0.7 ... -1.9 % per sweep on six allocations of three boxes; 2
                               // waves +1.4 %, simple_radial 4 -> 3 +0.6 ... +1.9 %: r06_variant_occupancy_{math,pinhole}.log).  Same bits.
#ifndef GCLM_DYN_LDS
#define GCLM_DYN_LDS 0         // measurement only: dynamic LDS bytes of a sweep launch, i.e. a cap on the workgroups per CU (with GCLM_NOMATH:
#endif                         // what the access pattern streams at 2 ... 8 waves per SIMD, profiles/r06_variant_occupancy.log)
#ifndef GCLM_LAT_PAIRS
#define GCLM_LAT_PAIRS 1       // row pairs, log-focal: the latitude sums of the two rows are taken together (lat_pair_accumulate)
#endif
#ifndef GCLM_MIRROR_MODELS
#define GCLM_MIRROR_MODELS ((1 << GCLM_RADIAL) | (1 << GCLM_SIMPLE_DIVISIONAL))   // models whose BUILT-IN choice is the row-pair walk
                             // (same-allocation A/B, profiles/r06_variant_row_pairs.log: simple_divisional -14.2 % per sweep at
                             // 2 waves per SIMD; radial -3.9 % once its walker is held to 168 VGPRs = 3 waves, -1.3 % at 2)
#endif
#ifndef GCLM_MIRROR_WAVES
#define GCLM_MIRROR_WAVES 2    // the row-pair walkers (sweep_body: MIRROR) hold two rows' loads: 256 VGPRs
#endif
#ifndef GCLM_MIRROR_WAVES_RADIAL
#define GCLM_MIRROR_WAVES_RADIAL 3    // ... radial's log-focal walker fits 168 VGPRs = 3 waves (16-32 B of scratch: one 8-byte
#endif                                //     reload per iteration); its general-focal form would spill 216-248 B there and keeps 2
#ifndef GCLM_SLAT_PINHOLE
#define GCLM_SLAT_PINHOLE 1    // pinhole has the scratch-plane instantiations too: never the built-in choice (memory-bound), only
#endif                         // gclm_set_slat_plane(h, 1) launches them (measurement)

#ifndef GCLM_TRACE
#define GCLM_TRACE 0                // measurement build only: device timestamps of one workgroup's stages (scripts/probes/trace_probe.py)
#endif
#if GCLM_TRACE
__device__ unsigned long long g_gclm_trace[64][16];
#define GCLM_T(step, slot) do { if (threadIdx.x == 0 && blockIdx.x == GCLM_TRACE_WG && blockIdx.y == 0 && (step) < 64) g_gclm_trace[(step)][(slot)] = wall_clock64(); } while (0)
extern "C" int gclm_debug_trace(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_gclm_trace), sizeof(g_gclm_trace)) == hipSuccess ? 0 : -1;
}
#ifndef GCLM_TRACE_WG
#define GCLM_TRACE_WG 0
#endif
#else
#define GCLM_T(step, slot) do {} while (0)
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
// 1-ulp hardware forms (v_rcp_f32 / v_sqrt_f32) for simple_divisional, whose thirteen quotients would otherwise
// each expand into the ~10-instruction correctly-rounded sequences
__device__ __forceinline__ float vrcp_hw(float a) { return __builtin_amdgcn_rcpf(a); }
__device__ __forceinline__ f2 vrcp_hw(f2 a) { return f2{__builtin_amdgcn_rcpf(a.x), __builtin_amdgcn_rcpf(a.y)}; }
__device__ __forceinline__ float vsqrt_hw(float a) { return __builtin_amdgcn_sqrtf(a); }
__device__ __forceinline__ f2 vsqrt_hw(f2 a) { return f2{__builtin_amdgcn_sqrtf(a.x), __builtin_amdgcn_sqrtf(a.y)}; }
__device__ __forceinline__ float vmax(float a, float b) { return fmaxf(a, b); }
__device__ __forceinline__ f2 vmax(f2 a, f2 b) { return f2{fmaxf(a.x, b.x), fmaxf(a.y, b.y)}; }
__device__ __forceinline__ float vclamp(float a, float lo, float hi) { return fminf(fmaxf(a, lo), hi); }
__device__ __forceinline__ f2 vclamp(f2 a, float lo, float hi) {
    return f2{fminf(fmaxf(a.x, lo), hi), fminf(fmaxf(a.y, lo), hi)};
}
// min(1, 1/sqrt(y)) for y >= 0 in ONE instruction: v_rsq_f32 with the VOP3 clamp bit (the result is clamped to [0, 1];
// rsq >= 0, rsq(0) = +inf -> 1).  Same bits as v_rsq_f32 + v_min_f32 for every y that is not NaN.
__device__ __forceinline__ float vrsq_min1(float y) {
    return __builtin_amdgcn_fmed3f(__frsqrt_rn(y), 0.0f, 1.0f);      // folded into the rsq's clamp bit (no inline asm:
                                                                     // the compiler keeps track of the trans-use hazard)
}
__device__ __forceinline__ f2 vrsq_min1(f2 y) { return f2{vrsq_min1(y.x), vrsq_min1(y.y)}; }
// denominator guard of the reference: x.masked_fill(x == 0, 1e6)
__device__ __forceinline__ float vguard(float a) { return a == 0.f ? 1e6f : a; }
__device__ __forceinline__ f2 vguard(f2 a) { return f2{a.x == 0.f ? 1e6f : a.x, a.y == 0.f ? 1e6f : a.y}; }
// select(d == 0, a, b)
__device__ __forceinline__ float vsel_eq0(float d, float a, float b) { return d == 0.f ? a : b; }
__device__ __forceinline__ f2 vsel_eq0(f2 d, f2 a, f2 b) { return f2{d.x == 0.f ? a.x : b.x, d.y == 0.f ? a.y : b.y}; }
__device__ __forceinline__ float vsplat(float, float s) { return s; }
__device__ __forceinline__ f2 vsplat(f2, float s) { return f2{s, s}; }
__device__ __forceinline__ float hsum(float a) { return a; }
__device__ __forceinline__ float hsum(f2 a) { return a.x + a.y; }

// sin(x) for |x| <= pi/2 (odd minimax polynomial, |err| < 1.2e-7 in fp32).  Latitudes are
// asin(clamp(tanh)) outputs of the CNN head (geocalib.py:73-75) and therefore in range; the reference takes
// torch.sin of whatever the caller passes (lm_optimizer.py:262,270), so anything outside is FOLDED into the
// range first (fold_halfpi) -- behind a wave-uniform branch in the sweep that in-range data never takes.
constexpr float kHalfPi = 1.57079632679489661923f;
// x -> r in [-pi/2, pi/2] with sin(r) = sin(x):  k = rint(x / pi), r = (-1)^k (x - k pi)  (two-term Cody-Waite, exact
// enough for |k| < 1e4: degrees fed as radians give k <= 29).  In-range x (k = 0) comes back bit-identical.
__device__ __forceinline__ float fold_halfpi(float x) {
    const float k = rintf(x * 0.318309886183790671538f);
    float r = fmaf(k, -3.14159274101257324219f, x);            // pi rounded to float ...
    r = fmaf(k, 8.74227765734758577e-08f, r);                 // ... and what the rounding dropped
    return (((int)k) & 1) ? -r : r;
}
__device__ __forceinline__ bool beyond_halfpi(float a) { return fabsf(a) > kHalfPi; }
// the sweep tests t = x^2 (which the polynomial needs anyway) against a threshold a hair beyond (pi/2)^2
constexpr float kHalfPiSq = 2.4675f;
template <typename F>
__device__ __forceinline__ F sin_halfpi(F x, F t) {          // t = x * x
    F p = vsplat(x, 2.6000457182817627e-06f);
    p = vfma(p, t, vsplat(x, -0.00019806611817330122f));
    p = vfma(p, t, vsplat(x, 0.008333017118275166f));
    p = vfma(p, t, vsplat(x, -0.16666656732559204f));
    return vfma(x * t, p, x);
}
template <typename F>
__device__ __forceinline__ F sin_halfpi(F x) { return sin_halfpi(x, x * x); }

// Scaled Huber on the squared residual x2 (lm_optimizer.py:61-87), in units of a^2 (the a^2 factor of the cost
// is applied once per workgroup): adds confidence * cost / a^2 to `cost_acc` and returns confidence * weight.
// Branch-free form, y = x2 / a^2:
//   weight = min(1, 1/sqrt(y))                  (= 1 for y <= 1, = 1/sqrt(y) beyond; y = 0: rsq = +inf -> 1)
//   cost   = y weight (2 - weight)              (= y for y <= 1, = 2 sqrt(y) - 1 beyond)
// The reference's where(y <= 1, ...) selects evaluate sqrt(y + 1e-8): a relative difference below 5e-9 for the
// y > 1 they apply to.  Its max(eps, 1/sqrt(y)) only matters for y > 7e13; residuals here are bounded
// (|r_up| <= 2, |r_lat| <= 2 => y <= 4/a^2), so it is the identity and is dropped.
template <typename F>
__device__ __forceinline__ F huber_accumulate(F x2, float inv_a2, F conf, F& cost_acc) {
    const F y = x2 * inv_a2;
    const F weight = vrsq_min1(y);
    const F wc = weight * conf;
    cost_acc = vfma(y * wc, 2.0f - weight, cost_acc);
    return wc;
}

struct HuberK {
    float inv_a2u, a2u, inv_a2l, a2l;
};

// Radial-model terms at r2 (see the header comment).  Only the members a model needs are computed; for
// pinhole / simple_radial everything else folds away at compile time.
template <typename F>
struct Radial {
    F s, s1x2, jd, s2x4;          // s, 2 s1, diagonal of d(off)/duv (= 2 s1), 4 s2
    F ds[2], ds1x2[2];            // ds/dk_j, 2 d(s1)/dk_j
    F tau, tau1x2, dtau[2];       // undistortion scale, 2 tau1, dtau/dk_j
};

// GUARD (simple_divisional only): the reference's masked_fill(x == 0, 1e6) on the denominators that carry a factor r2
// or r2 k.  They can only fire at the principal point (r2 == 0: at most one pixel of an image) or while k == 0 (the
// first sweep), so the sweep picks, per wave and iteration, a copy of the body without those selects whenever no lane
// can need them (GUARD = false: same operations on the same values, hence the same bits; see sweep_kernel).
template <int MODEL, typename F, bool GUARD = true>
__device__ __forceinline__ void radial_terms(const PBlock& P, F r2, Radial<F>& R) {
    const F one = vsplat(r2, 1.0f), zero = vsplat(r2, 0.f);
    R.s = R.tau = one;
    R.s1x2 = R.jd = R.s2x4 = R.tau1x2 = zero;
    R.ds[0] = R.ds[1] = R.ds1x2[0] = R.ds1x2[1] = R.dtau[0] = R.dtau[1] = zero;
    if constexpr (MODEL == GCLM_SIMPLE_RADIAL) {                 // camera.py:611-660
        R.s = vfma(r2, vsplat(r2, P.k1), one);
        R.s1x2 = R.jd = vsplat(r2, 2.0f * P.k1);
        R.ds[0] = r2;
        R.ds1x2[0] = vsplat(r2, 2.0f);
        R.tau = vfma(r2, vsplat(r2, -P.k1), one);
        R.tau1x2 = vsplat(r2, -2.0f * P.k1);
        R.dtau[0] = -r2;
    } else if constexpr (MODEL == GCLM_RADIAL) {                 // camera.py:712-786
        const F r4 = r2 * r2;
        R.s = vfma(r4, vsplat(r2, P.k2), vfma(r2, vsplat(r2, P.k1), one));
        R.s1x2 = R.jd = vfma(r2, vsplat(r2, 4.0f * P.k2), vsplat(r2, 2.0f * P.k1));
        R.s2x4 = vsplat(r2, 8.0f * P.k2);
        R.ds[0] = r2;  R.ds1x2[0] = vsplat(r2, 2.0f);
        R.ds[1] = r4;  R.ds1x2[1] = 4.0f * r2;
        const float b1 = -P.k1, b2 = 3.0f * P.k1 * P.k1 - P.k2;
        R.tau = vfma(r4, vsplat(r2, b2), vfma(r2, vsplat(r2, b1), one));
        R.tau1x2 = vfma(r2, vsplat(r2, 4.0f * b2), vsplat(r2, 2.0f * b1));
        R.dtau[0] = vfma(r4, vsplat(r2, 6.0f * P.k1), -r2);
        R.dtau[1] = -r4;
    } else if constexpr (MODEL == GCLM_SIMPLE_DIVISIONAL) {      // camera.py:829-940, guards as there
        // The reference divides by thirteen different products of {r2, t1 = sqrt(max(1 - 4 k r2, 1e-6)), k} and
        // replaces a denominator that is exactly 0 by 1e6 (masked_fill).  t1 >= 1e-3, so a product vanishes iff
        // r2 == 0 (indicator r2) or, where k is a factor, r2 k == 0 (indicator rk): every guarded reciprocal is
        // select(indicator == 0, 1e-6, product of 1/r2, 1/t1, 1/k) -- three reciprocals per pixel instead of
        // thirteen; the numerators keep the reference's own (cancelling) form.
        const float k = P.k1;
        const float ik = vrcp_hw(k);                               // inf for k = 0: only read behind the rk selects
        const F tt = vfma(r2, vsplat(r2, -4.0f * k), one);           // 1 - 4 k r2
        const F rk = r2 * k;
        const F tiny = vsplat(r2, 1e-6f);
        const F t0 = vmax(tt, tiny);
        // 1 - sqrt(1 - 4 k r2) cancels: one ulp of the root is 1 / (2 |k| r2) ulps of the difference, and v_sqrt_f32's ulp is not
        // a zero-mean rounding like the correctly rounded root the reference takes (at |k| = 1e-3 the final cost of an image came
        // out 2e-4 off; the oracle's float32 build: 3e-6).  One Newton step on the exact fma residual brings the hardware root to
        // the correctly rounded one's accuracy for three packed operations (+ two for 1 / t1).  The reciprocal is stepped to the
        // SAME point: the Jacobian terms below are cancelling sums of t1 and 1 / t1, and a root and a reciprocal that belong to
        // two points a few 1e-8 apart break those cancellations where the terms are large (fuzz 29/162; profiles/README.md).
        // Ties: for t0 = 1 - 2^-24 (4 k r2 within an ulp of 1: the pixels next to the principal point while |k| ~ 6e-8 f^2) the
        // Newton estimate is EXACTLY the midpoint of two floats and round-to-even picks 1.0, where the true root (always
        // below Newton's estimate) rounds to 1 - 2^-24 as torch.sqrt does: 1 - t1 = 0 instead of 2^-24 made the up vector of those
        // pixels and their k-derivative garbage 40x the reference's own (fuzz 101/270).  A nudge of 2^-21 half-ulps breaks it.
        const F t1h = vsqrt_hw(t0), ih = vrcp_hw(t1h);
        const F t1 = vfma(vfma(t0, vsplat(t0, -0x1p-45f), vfma(-t1h, t1h, t0)), 0.5f * ih, t1h);
        const F it1 = vfma(ih, vfma(-t1, ih, one), ih), it0 = it1 * it1;
        const F ssq = vsel_eq0(t0 - tt, t1, vsqrt_hw(vmax(tt, zero)));      // tt < 1e-6 (|4 k r2| ~ 1): nothing cancels there
        const F ir2 = vrcp_hw(r2), ir4 = ir2 * ir2, ir6 = ir4 * ir2;    // inf for r2 = 0: only read behind the selects
        const F omt = one - t1;
        const F r4 = r2 * r2;
        // guarded reciprocal: select(indicator == 0, a, b); without GUARD the caller has made sure indicator != 0
        auto sel0 = [](F d, F a, F b) { if constexpr (GUARD) return vsel_eq0(d, a, b); else return b; };
        R.s = sel0(rk, one, (one - ssq) * (ir2 * (0.5f * ik)));   // den = 2 k r2
        {   // J_distort scale2pts (:843-851): off = uv (4 d2 - (1-t1) d1)/(d1 d2), d1 = 2 t1 r2, d2 = k r4
            const F d1 = t1 * (2.0f * r2), d2 = r4 * k;
            R.s1x2 = vfma(d2, vsplat(r2, 4.0f), -(omt * d1)) * sel0(rk, tiny, (ir6 * it1) * (0.5f * ik));
        }
        {   // J_up_projection_offset wrt uv (:912-940): diagonal jd and the uv uv^T coefficient
            const F i_r2t1 = ir2 * it1, i_r4t1 = ir4 * it1;
            R.jd = 4.0f * sel0(r2, tiny, 0.5f * i_r2t1) - omt * sel0(rk, tiny, ir4 * ik);
            F pc = -16.0f * sel0(r2, tiny, 0.25f * i_r4t1);
            pc = pc + (32.0f * k) * sel0(r2, tiny, 0.25f * (i_r2t1 * it0));
            pc = pc - 4.0f * sel0(r2, tiny, i_r4t1);
            pc = pc + 4.0f * omt * sel0(rk, tiny, ir6 * ik);
            R.s2x4 = pc;
        }
        {   // J_distort scale2dist (:853-857): (2 d2 - (1-t1) d1)/(d1 d2), d1 = 2 k t1, d2 = 2 r2 k^2
            const F d1 = t1 * (2.0f * k), d2 = r2 * (2.0f * k * k);
            R.ds[0] = vfma(d2, vsplat(r2, 2.0f), -(omt * d1)) * sel0(rk, tiny, (ir2 * it1) * (0.25f * ik * ik * ik));
        }
        {   // J_up_projection_offset wrt dist (:898-911); 4 t0 t1 > 0 needs no guard
            F J = 4.0f * (it0 * it1);
            J = J - 2.0f * sel0(rk, tiny, (ir2 * it1) * ik);
            J = J + omt * sel0(rk, tiny, ir4 * (ik * ik));
            R.ds1x2[0] = J;
        }
        const F den2 = vfma(r2, vsplat(r2, k), one);                  // 1 + k r2
        R.tau = vrcp_hw(vguard(den2));
        R.tau1x2 = (-2.0f * k) * R.tau * R.tau;                        // :878-883
        R.dtau[0] = -r2 * vsel_eq0(den2, tiny, R.tau * R.tau);         // :875-877
    }
}

// |q|^2 kept away from 0 for the rsq behind it (the reference normalises with an epsilon as well, misc.py:275-276): the
// 1e-24 rides in the fma chain -- it vanishes in the rounding unless |q|^2 < 1e-17 -- instead of a v_max_f32 per pixel.
template <typename F>
__device__ __forceinline__ F norm2_eps(F qx, F qy) {
    return vfma(qx, qx, vfma(qy, qy, vsplat(qx, 1e-24f)));
}

template <int MODEL>
struct Layout {
    static constexpr int PM = acc_pm(MODEL);            // full columns of the record (4, radial 5)
    static constexpr int ND = num_dist_params(MODEL);   // distortion parameters of the model
    static constexpr int PN = 3 + ND;                   // columns this model accumulates
    static constexpr int NACC = acc_floats(MODEL);
};

// acc += wgt * J^T r and wgt * J^T J (upper triangle) for a rank-one Jacobian row set s[0..PN)
template <int MODEL, typename F>
__device__ __forceinline__ void accumulate(F (&acc)[Layout<MODEL>::NACC], const F (&s)[Layout<MODEL>::PN], F wgt, F r) {
    constexpr int PM = Layout<MODEL>::PM, PN = Layout<MODEL>::PN;
#pragma unroll
    for (int i = 0; i < PN; ++i) {
        const F wi = wgt * s[i];
        acc[A_G0 + i] = vfma(wi, r, acc[A_G0 + i]);
#pragma unroll
        for (int j = i; j < PN; ++j) acc[acc_h(PM, i, j)] = vfma(wi, s[j], acc[acc_h(PM, i, j)]);
    }
}

// Hand-scheduled form of the same math for the two BASELINE models (pinhole, simple_radial): identical
// operations to pixel_accumulate<> below with the model terms substituted, written out so that the
// compiler reaches 96 / 128 VGPRs (5 / 4 waves per SIMD) without spills or dependency stalls.
//
// LOGF: the focal parameter is log f (every loop sweep of the default conf), so d(uv)/dtheta_f = w = -(u,v)
// exactly and every "x . w" product is minus the matching "x . uv" product that is computed anyway:
// n.w = -n.uv, m.w = -m.uv, uv.w = -r2, p.w = -t, h.w = -h.uv, and the focal column collapses to
//   up:  s2 = c (m.uv) - 2 k1 s3          lat:  l2 = -e (h.uv) - 2 k1 l3
template <int MODEL, bool HAS_UP, bool LOGF, typename F>
__device__ __forceinline__ void pixel_accumulate_fast(const PBlock& P, const HuberK& hk, F u, F px, float v, F dux, F duy,
                                                 F slat, F cu, F cl, F (&acc)[kNAcc]) {
    constexpr bool DIST = MODEL != GCLM_PINHOLE;
    // u = (x - cx)/fx and px = ga - gc u belong to the lane's COLUMN and are loop invariants of the sweep
    // (column-stationary mapping, see sweep_kernel); v = (y - cy)/fy is lane-scalar (one image row per tile)
    const F r2 = vfma(u, u, vsplat(u, v * v));
    [[maybe_unused]] F wx = vsplat(u, 0.f), uvw = vsplat(u, 0.f);
    [[maybe_unused]] float wy = 0.f;
    if constexpr (!LOGF) {
        wx = u * (-P.wfx);                               // d(uv)/d(focal parameter) = (wx, wy)
        wy = -v * P.wfy;
        uvw = vfma(u, wx, vsplat(u, v * wy));
    }
    const float k1x2 = 2.0f * P.k1;

    if constexpr (HAS_UP) {
        const float py = fmaf(-P.gc, v, P.gb);
        F qx = px, qy = vsplat(u, py), d = vsplat(u, 1.0f), t = vsplat(u, 0.f);
        if constexpr (DIST) {
            d = vfma(r2, vsplat(u, P.k1), vsplat(u, 1.0f));
            t = vfma(u, px, vsplat(u, v * py));
            const F kt = t * k1x2;
            qx = vfma(d, px, kt * u);
            qy = vfma(d, vsplat(u, py), kt * v);
        }
        const F n2 = norm2_eps(qx, qy);
        const F rn = vrsq(n2);
        const F ux = qx * rn, uy = qy * rn;              // predicted up vector
        const F rx = dux - ux, ry = duy - uy;            // residual (lm_optimizer.py:266)
        const F x2 = vfma(rx, rx, ry * ry);
        const F wgt = huber_accumulate(x2, hk.inv_a2u, cu, acc[A_CU]);
        // rank-one Jacobian: s_k = (M nq) . dp/dtheta_k, nq = n/|q|, n = (-uy, ux);  rho = n . r
        const F nx = -uy * rn, ny = ux * rn;
        const F nuv = vfma(nx, u, ny * v);
        [[maybe_unused]] F nw = vsplat(u, 0.f);
        if constexpr (!LOGF) nw = vfma(nx, wx, ny * wy);
        F mx = nx, my = ny, muv = nuv, mw = nw;
        if constexpr (DIST) {
            const F c2 = nuv * k1x2;
            mx = vfma(d, nx, c2 * u);
            my = vfma(d, ny, c2 * v);
            muv = vfma(mx, u, my * v);
            if constexpr (!LOGF) mw = vfma(mx, wx, my * wy);
        }
        // dp/ddelta_k = (T0k - u T2k, T1k - v T2k)  =>  s_k = m.T[0:2,k] - (m.uv) T2k
        const F s0 = vfma(mx, vsplat(u, P.T00), vfma(my, vsplat(u, P.T10), muv * (-P.T20)));
        const F s1 = vfma(mx, vsplat(u, P.T01), vfma(my, vsplat(u, P.T11), muv * (-P.T21)));
        F s2 = LOGF ? muv * P.gc : mw * (-P.gc);         // dp/df = -c w
        [[maybe_unused]] F s3 = vsplat(u, 0.f);
        if constexpr (DIST) {
            const F np_ = vfma(nx, px, ny * py);
            // + 2 k1 [ p (uv.w) + t w + (u,v)(p.w) ] . nq          (perspective_fields.py:146-153)
            if constexpr (LOGF) {
                s3 = vfma(r2, np_, (t * 2.0f) * nuv);    // dq/dk1 = r2 p + 2 t (u,v)   (:170-180)
                s2 = vfma(s3, vsplat(u, -k1x2), s2);
            } else {
                const F pw = vfma(px, wx, vsplat(u, py * wy));
                s2 = vfma(vfma(np_, uvw, vfma(t, nw, nuv * pw)), vsplat(u, k1x2), s2);
                s3 = vfma(r2, np_, (t * 2.0f) * nuv);
            }
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

    {   // latitude.  ray = (e u, e v, 1) / n; everything below needs the ray only through dot products with
        // per-image vectors, so it is never formed:  ray.x = ern (x_xy.uv) + rnn x_z,  ern = e/n, rnn = 1/n
        F e = vsplat(u, 1.0f), er2 = r2;
        if constexpr (DIST) {
            e = vfma(r2, vsplat(u, -P.k1), vsplat(u, 1.0f));
            er2 = e * r2;
        }
        const F nn = DIST ? vfma(e, er2, vsplat(u, 1.0f)) : r2 + 1.0f;      // |(e u, e v, 1)|^2
        const F rnn = vrsq(nn);
        const F guv = vfma(u, vsplat(u, P.ga), vsplat(u, v * P.gb));          // g_xy . uv
        F s, l0, l1;
        const F uT0 = vfma(u, vsplat(u, P.T00), vsplat(u, v * P.T10)), uT1 = vfma(u, vsplat(u, P.T01), vsplat(u, v * P.T11));
        if constexpr (DIST) {
            // ray . x = rnn (e (x_xy.uv) + x_z): the 1/n factor last -- one op per pixel pair less than
            // (e rnn)(x_xy.uv) + rnn x_z, which needs e rnn (pinhole: e = 1, no difference; it keeps the form below)
            s = vfma(e, guv, vsplat(u, P.gc)) * rnn;                                                // ray . g
            l0 = vfma(e, uT0, vsplat(u, P.T20)) * rnn;                                              // ray . T[:,0]
            l1 = vfma(e, uT1, vsplat(u, P.T21)) * rnn;
        } else {
            const F ern = DIST ? e * rnn : rnn;
            s = vfma(ern, guv, rnn * P.gc);
            l0 = vfma(ern, uT0, rnn * P.T20);
            l1 = vfma(ern, uT1, rnn * P.T21);
        }
        const F sc = vclamp(s, -1.0f + 1e-6f, 1.0f - 1e-6f);
        const F rl = slat - sc;                          // lm_optimizer.py:262,270-271 (slat = sin(latitude_field))
        const F wgt = huber_accumulate(rl * rl, hk.inv_a2l, cl, acc[A_CL]);
        // h = (g_xy - s ray_xy)/n;  ds/df = h.(e w - 2 k1 (u,v)(uv.w)),  ds/dk1 = h.(-r2 (u,v))   (perspective_fields.py:255-272)
        //   h.uv = rnn (g_xy.uv - s ern r2),   h.w = rnn (g_xy.w - s ern (uv.w))
        F l2;
        [[maybe_unused]] F hu = vsplat(u, 0.f), l3 = vsplat(u, 0.f);
        if constexpr (LOGF || DIST) hu = vfma(-s, DIST ? er2 * rnn : r2 * rnn, guv) * rnn;
        if constexpr (LOGF) {
            l2 = -hu;
            if constexpr (DIST) {
                l3 = -(hu * r2);
                l2 = vfma(l3, vsplat(u, -k1x2), -(e * hu));
            }
        } else {
            const F gw = vfma(wx, vsplat(u, P.ga), vsplat(u, wy * P.gb));
            const F hw = vfma(-s, (DIST ? e * rnn : rnn) * uvw, gw) * rnn;
            l2 = hw;
            if constexpr (DIST) l2 = vfma(e, hw, -((uvw * k1x2) * hu));
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
            if constexpr (!LOGF) l3 = -(hu * r2);
            const F w3 = wgt * l3;
            acc[A_G0 + 3] = vfma(w3, rl, acc[A_G0 + 3]);
            acc[A_H00 + 3] = vfma(w0, l3, acc[A_H00 + 3]);
            acc[A_H00 + 6] = vfma(w1, l3, acc[A_H00 + 6]);
            acc[A_H00 + 8] = vfma(w2, l3, acc[A_H00 + 8]);
            acc[A_H00 + 9] = vfma(w3, l3, acc[A_H00 + 9]);
        }
    }
}

// EMIT (scalar instantiation only): instead of accumulating, write per-pixel quantities out --
//   1: the Jacobian rows of the predicted fields, J_up (2 x PN) = n s^T with n = (-up_y, up_x), J_lat (1 x PN) = l
//      (gclm_jacobian_fields);   2: the residuals r_up (2), r_lat (1) (gclm_residual_fields).
// GMODE (simple_divisional): 0 = guarded radial terms; 1 = guard-free terms, recomputed with the guards when the
// wave-uniform `patch` says a lane of this wave may need them (an if-without-else: the common path carries no copies)
//
// The body is cut in two: `pixel_shared` holds everything that depends on (u, v^2) only -- r2, the radial terms, |ray|^2 and
// its rsq -- and `pixel_stage` the rest.  The one-row walkers call them back to back (pixel_accumulate); the row-pair walker
// (row_math_mirror) evaluates the first ONCE for rows y and H - y of a column.
template <int MODEL, typename F>
struct PixelShared {
    F r2, uvw;              // uvw = uv . d(uv)/d(focal parameter) (general-focal columns only)
    Radial<F> R;            // (what pixel_stage derives from these alone -- tau r2, |ray|^2, its rsq ... -- is merged by the
};                          //  compiler's value numbering when two rows share one PixelShared: same operands, same values)
template <int MODEL, bool LOGF, typename F, int GMODE = 0>
__device__ __forceinline__ void pixel_shared(const PBlock& P, F u, float v, [[maybe_unused]] bool patch, PixelShared<MODEL, F>& S) {
    S.r2 = vfma(u, u, vsplat(u, v * v));
    S.uvw = vsplat(u, 0.f);
    if constexpr (!LOGF) {                               // LOGF: w = -(u,v), see pixel_accumulate_fast
        const F wx = u * (-P.wfx);                       // d(uv)/d(focal parameter) = (wx, wy)
        const float wy = -v * P.wfy;
        S.uvw = vfma(u, wx, vsplat(u, v * wy));
    }
    radial_terms<MODEL, F, GMODE == 0>(P, S.r2, S.R);
    if constexpr (GMODE == 1) {
        if (patch) {
            asm volatile("; simple_divisional: guarded radial terms" ::: "memory");   // a real branch: never if-converted
            radial_terms<MODEL, F, true>(P, S.r2, S.R);
        }
    }
}

// What the latitude field of one row hands to lat_pair_accumulate (row pairs): Huber weight x confidence, residual, the two
// gravity columns, and h.uv -- every other latitude column is (a function of r2) x h.uv
template <typename F>
struct LatRow {
    F wgt, rl, l0, l1, hu;
};
// PART: 0 = both fields, 1 = the up field only, 2 = the latitude field only (the row-pair walker orders them itself);
// EMIT = 3 (with PART 2): the latitude terms go to `lat_out` instead of into the sums
template <int MODEL, bool HAS_UP, bool LOGF, typename F, int EMIT = 0, int PART = 0>
__device__ __forceinline__ void pixel_stage(const PBlock& P, const HuberK& hk, const PixelShared<MODEL, F>& S, F u, F px, float v,
                                            F dux, F duy, F slat, F cu, F cl, F (&acc)[Layout<MODEL>::NACC],
                                            [[maybe_unused]] float* j_up = nullptr, [[maybe_unused]] float* j_lat = nullptr,
                                            [[maybe_unused]] LatRow<F>* lat_out = nullptr) {
    constexpr bool DIST = MODEL != GCLM_PINHOLE;
    constexpr int ND = Layout<MODEL>::ND, PN = Layout<MODEL>::PN;
    const F r2 = S.r2;
    const Radial<F>& R = S.R;
    [[maybe_unused]] F wx = vsplat(u, 0.f);
    [[maybe_unused]] const F uvw = S.uvw;
    [[maybe_unused]] float wy = 0.f;
    if constexpr (!LOGF) {
        wx = u * (-P.wfx);
        wy = -v * P.wfy;
    }

    if constexpr (HAS_UP && PART != 2) {
        const float py = fmaf(-P.gc, v, P.gb);
        F qx = px, qy = vsplat(u, py), t = vsplat(u, 0.f);
        if constexpr (DIST) {
            t = vfma(u, px, vsplat(u, v * py));
            const F kt = R.s1x2 * t;
            qx = vfma(R.s, px, kt * u);
            qy = vfma(R.s, vsplat(u, py), kt * v);
        }
        const F n2 = norm2_eps(qx, qy);
        const F rn = vrsq(n2);
        const F ux = qx * rn, uy = qy * rn;              // predicted up vector
        const F rx = dux - ux, ry = duy - uy;            // residual (lm_optimizer.py:266)
        const F x2 = vfma(rx, rx, ry * ry);
        const F wgt = huber_accumulate(x2, hk.inv_a2u, cu, acc[A_CU]);
        // rank-one Jacobian: s_k = (M nq) . dp/dtheta_k, nq = n/|q|, n = (-uy, ux);  rho = n . r
        const F nx = -uy * rn, ny = ux * rn;
        const F nuv = vfma(nx, u, ny * v);
        [[maybe_unused]] F nw = vsplat(u, 0.f);
        if constexpr (!LOGF) nw = vfma(nx, wx, ny * wy);
        F mx = nx, my = ny, muv = nuv, mw = nw;
        if constexpr (DIST) {
            const F c2 = R.s1x2 * nuv;
            mx = vfma(R.s, nx, c2 * u);
            my = vfma(R.s, ny, c2 * v);
            muv = vfma(mx, u, my * v);
            if constexpr (!LOGF) mw = vfma(mx, wx, my * wy);
        }
        F s[PN];
        // dp/ddelta_k = (T0k - u T2k, T1k - v T2k)  =>  s_k = m.T[0:2,k] - (m.uv) T2k
        s[0] = vfma(mx, vsplat(u, P.T00), vfma(my, vsplat(u, P.T10), muv * (-P.T20)));
        s[1] = vfma(mx, vsplat(u, P.T01), vfma(my, vsplat(u, P.T11), muv * (-P.T21)));
        s[2] = LOGF ? muv * P.gc : mw * (-P.gc);         // dp/df = -c w
        if constexpr (DIST) {
            const F np_ = vfma(nx, px, ny * py);
            // + [p (off.w) + off (p.w)] . nq + t (Joff w) . nq            (perspective_fields.py:146-153)
            [[maybe_unused]] F b_ = vsplat(u, 0.f), ab2 = vsplat(u, 0.f);
            if constexpr (LOGF) {                                // uv.w = -r2, n.w = -n.uv, p.w = -t
                const F a_ = np_ * r2;
                b_ = t * nuv;
                if constexpr (MODEL == GCLM_SIMPLE_DIVISIONAL) {
                    s[2] = vfma(-R.s1x2, a_ + b_, s[2]);
                    s[2] = vfma(-b_, vfma(R.s2x4, r2, R.jd), s[2]);
                } else {
                    ab2 = vfma(b_, vsplat(u, 2.0f), a_);         // r2 (n.p) + 2 t (n.uv) = dq/dk1 . nq as well (below)
                    s[2] = vfma(-R.s1x2, ab2, s[2]);
                    if constexpr (MODEL == GCLM_RADIAL) s[2] = vfma(-(R.s2x4 * r2), b_, s[2]);
                }
            } else {
                const F pw = vfma(px, wx, vsplat(u, py * wy));
                if constexpr (MODEL == GCLM_SIMPLE_DIVISIONAL) {     // jd carries the reference's own guards
                    s[2] = vfma(R.s1x2, vfma(np_, uvw, nuv * pw), s[2]);
                    s[2] = vfma(t, vfma(R.jd, nw, R.s2x4 * (uvw * nuv)), s[2]);
                } else {                                             // polynomial models: jd == 2 s1
                    s[2] = vfma(R.s1x2, vfma(np_, uvw, vfma(t, nw, nuv * pw)), s[2]);
                    if constexpr (MODEL == GCLM_RADIAL) s[2] = vfma(R.s2x4 * t, uvw * nuv, s[2]);
                }
            }
            if constexpr (MODEL == GCLM_RADIAL) {
                // dq/dk1 = r2 p + 2 t uv,  dq/dk2 = r4 p + 4 r2 t uv = r2 (dq/dk1 + 2 t uv)                 (:170-180)
                if constexpr (LOGF && GCLM_REUSE_TNUV) {         // the focal column has formed r2 (n.p) + 2 t (n.uv) already
                    s[3] = ab2;
                    s[4] = r2 * vfma(b_, vsplat(u, 2.0f), ab2);
                } else {
                    const F tn2 = (t * 2.0f) * nuv;
                    s[3] = vfma(r2, np_, tn2);
                    s[4] = r2 * (s[3] + tn2);
                }
            } else {
#pragma unroll
                for (int j = 0; j < ND; ++j) {           // dq/dk_j = (ds/dk_j) p + 2 (ds1/dk_j) t (u,v)   (:170-180)
                    if constexpr (LOGF && GCLM_REUSE_TNUV) s[3 + j] = vfma(R.ds[j], np_, R.ds1x2[j] * b_);       // t (n.uv) is there
                    else s[3 + j] = vfma(R.ds[j], np_, (R.ds1x2[j] * t) * nuv);
                }
            }
        }
        const F rho = vfma(ux, ry, -(uy * rx));
        if constexpr (EMIT == 1) {
            if (j_up) {
#pragma unroll
                for (int k = 0; k < PN; ++k) {
                    j_up[k] = -uy * s[k];
                    j_up[PN + k] = ux * s[k];
                }
            }
        } else if constexpr (EMIT == 2) {
            if (j_up) {
                j_up[0] = rx;
                j_up[1] = ry;
            }
        } else {
            accumulate<MODEL>(acc, s, wgt, rho);
        }
    }

    if constexpr (PART != 1) {
        // latitude (ray = (tau u, tau v, 1)/n only through dot products, see pixel_accumulate_fast; the explicit-ray form
        // measured 1.5-3 % slower for every model: profiles/README.md)
        const F tr2 = DIST ? R.tau * r2 : r2;
        const F nn = DIST ? vfma(R.tau, tr2, vsplat(u, 1.0f)) : r2 + 1.0f;
        const F rnn = vrsq(nn);
        const F guv = vfma(u, vsplat(u, P.ga), vsplat(u, v * P.gb));
        const F uT0 = vfma(u, vsplat(u, P.T00), vsplat(u, v * P.T10)), uT1 = vfma(u, vsplat(u, P.T01), vsplat(u, v * P.T11));
        F s_, l[PN];
        [[maybe_unused]] F trn = rnn;
        if constexpr (DIST && LOGF) {       // ray . x = rnn (tau (x_xy.uv) + x_z), see the fast body
            s_ = vfma(R.tau, guv, vsplat(u, P.gc)) * rnn;
            l[0] = vfma(R.tau, uT0, vsplat(u, P.T20)) * rnn;
            l[1] = vfma(R.tau, uT1, vsplat(u, P.T21)) * rnn;
        } else {
            trn = DIST ? R.tau * rnn : rnn;
            s_ = vfma(trn, guv, rnn * P.gc);
            l[0] = vfma(trn, uT0, rnn * P.T20);
            l[1] = vfma(trn, uT1, rnn * P.T21);
        }
        const F sc = vclamp(s_, -1.0f + 1e-6f, 1.0f - 1e-6f);
        const F rl = slat - sc;                          // lm_optimizer.py:262,270-271 (slat = sin(latitude_field))
        const F wgt = huber_accumulate(rl * rl, hk.inv_a2l, cl, acc[A_CL]);
        // ds/df = tau (h.w) + 2 tau1 (uv.w)(h.uv),  ds/dk_j = (dtau/dk_j)(h.uv)   (perspective_fields.py:255-272)
        //   h.uv = rnn (g_xy.uv - s trn r2),   h.w = rnn (g_xy.w - s trn (uv.w))
        [[maybe_unused]] F hu = vsplat(u, 0.f);
        if constexpr (LOGF || DIST) hu = vfma(-s_, tr2 * rnn, guv) * rnn;
        if constexpr (EMIT == 3) {
            static_assert(LOGF && DIST && PART == 2, "latitude terms out: the log-focal row-pair walker");
            lat_out->wgt = wgt; lat_out->rl = rl; lat_out->l0 = l[0]; lat_out->l1 = l[1]; lat_out->hu = hu;
            return;
        }
        if constexpr (LOGF) {                            // h.w = -h.uv, uv.w = -r2
            l[2] = -hu;
            if constexpr (DIST) l[2] = -(hu * vfma(R.tau1x2, r2, R.tau));
        } else {
            const F gw = vfma(wx, vsplat(u, P.ga), vsplat(u, wy * P.gb));
            const F hw = vfma(-s_, trn * uvw, gw) * rnn;
            l[2] = hw;
            if constexpr (DIST) l[2] = vfma(R.tau, hw, (R.tau1x2 * uvw) * hu);
        }
        if constexpr (DIST) {
#pragma unroll
            for (int j = 0; j < ND; ++j) l[3 + j] = R.dtau[j] * hu;
        }
        if constexpr (EMIT == 1) {
            if (j_lat) {
#pragma unroll
                for (int k = 0; k < PN; ++k) j_lat[k] = l[k];
            }
        } else if constexpr (EMIT == 2) {
            if (j_lat) j_lat[0] = rl;
        } else {
            accumulate<MODEL>(acc, l, wgt, rl);
        }
    }
}

// ROW PAIRS, log-focal: the latitude sums of rows y and H - y of a pixel pair taken TOGETHER.  Every latitude column but the two
// gravity columns is f_j(r2) h.uv with f_2 = -(tau + 2 tau1 r2), f_{3+j} = dtau/dk_j (pixel_stage) -- functions of r2 that the
// two rows share -- so with a_r = w_r hu_r:
//   sum_r w_r l_j l_k = (sum_r a_r hu_r) f_j f_k,   sum_r w_r l_i l_j = (sum_r a_r l_i) f_j (i < 2),   sum_r w_r rl l_j = (sum_r a_r rl_r) f_j
// 45 instead of 56 packed operations per pixel pair for radial (36 instead of 40 for simple_divisional); the sums differ from
// the row-by-row form in their rounding only.
template <int MODEL, typename F>
__device__ __forceinline__ void lat_pair_accumulate(F (&acc)[Layout<MODEL>::NACC], const PixelShared<MODEL, F>& S, const LatRow<F>& A,
                                                    const LatRow<F>& B) {
    constexpr int PM = Layout<MODEL>::PM, ND = Layout<MODEL>::ND;
    const Radial<F>& R = S.R;
    F f[1 + ND];
    f[0] = -vfma(R.tau1x2, S.r2, R.tau);
#pragma unroll
    for (int j = 0; j < ND; ++j) f[1 + j] = R.dtau[j];
    // the gravity columns, row by row
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const LatRow<F>& L = r == 0 ? A : B;
        const F w0 = L.wgt * L.l0, w1 = L.wgt * L.l1;
        acc[A_G0 + 0] = vfma(w0, L.rl, acc[A_G0 + 0]);
        acc[A_G0 + 1] = vfma(w1, L.rl, acc[A_G0 + 1]);
        acc[acc_h(PM, 0, 0)] = vfma(w0, L.l0, acc[acc_h(PM, 0, 0)]);
        acc[acc_h(PM, 0, 1)] = vfma(w0, L.l1, acc[acc_h(PM, 0, 1)]);
        acc[acc_h(PM, 1, 1)] = vfma(w1, L.l1, acc[acc_h(PM, 1, 1)]);
    }
    const F a1 = A.wgt * A.hu, a2 = B.wgt * B.hu;
    const F sAA = vfma(a2, B.hu, a1 * A.hu), sC = vfma(a2, B.rl, a1 * A.rl);
    const F sB0 = vfma(a2, B.l0, a1 * A.l0), sB1 = vfma(a2, B.l1, a1 * A.l1);
#pragma unroll
    for (int m = 0; m <= ND; ++m) {
        acc[A_G0 + 2 + m] = vfma(sC, f[m], acc[A_G0 + 2 + m]);
        acc[acc_h(PM, 0, 2 + m)] = vfma(sB0, f[m], acc[acc_h(PM, 0, 2 + m)]);
        acc[acc_h(PM, 1, 2 + m)] = vfma(sB1, f[m], acc[acc_h(PM, 1, 2 + m)]);
#pragma unroll
        for (int n = m; n <= ND; ++n) acc[acc_h(PM, 2 + m, 2 + n)] = vfma(sAA, f[m] * f[n], acc[acc_h(PM, 2 + m, 2 + n)]);
    }
}

template <int MODEL, bool HAS_UP, bool LOGF, typename F, int EMIT = 0, int GMODE = 0>
__device__ __forceinline__ void pixel_accumulate(const PBlock& P, const HuberK& hk, F u, F px, float v, F dux, F duy,
                                                 F slat, F cu, F cl, F (&acc)[Layout<MODEL>::NACC],
                                                 [[maybe_unused]] float* j_up = nullptr,
                                                 [[maybe_unused]] float* j_lat = nullptr,
                                                 [[maybe_unused]] bool patch = false) {
    PixelShared<MODEL, F> S;
    pixel_shared<MODEL, LOGF, F, GMODE>(P, u, v, patch, S);
    pixel_stage<MODEL, HAS_UP, LOGF, F, EMIT>(P, hk, S, u, px, v, dux, duy, slat, cu, cl, acc, j_up, j_lat);
}

// wave64 sum: four in-row DPP steps (quad_perm x2, row_half_mirror, row_mirror) leave the row sum in every lane of each
// 16-lane row, row_bcast:15 / :31 carry the rows along -- six v_add_f32 with DPP operands, no LDS (a __shfl_xor butterfly:
// six ds_bpermute round trips per value); the total is valid in lane 63 (kWaveSumLane).
constexpr int kWaveSumLane = 63;
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
    const int moved = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false);
    return v + __builtin_bit_cast(float, moved);       // lanes outside ROW_MASK add 0
}
__device__ __forceinline__ float wave_sum(float v) {
    v = dpp_add<0xB1, 0xF>(v);      // quad_perm:[1,0,3,2]
    v = dpp_add<0x4E, 0xF>(v);      // quad_perm:[2,3,0,1]
    v = dpp_add<0x141, 0xF>(v);     // row_half_mirror
    v = dpp_add<0x140, 0xF>(v);     // row_mirror
    v = dpp_add<0x142, 0xA>(v);     // row_bcast:15 -> rows 1, 3
    v = dpp_add<0x143, 0xC>(v);     // row_bcast:31 -> rows 2, 3
    return v;
}

// Per-lane tile of one loop iteration: VEC = 4 -> one float4 per plane, processed as two packed
// pixel pairs; VEC = 1 -> one pixel, scalar math (odd widths / unaligned pointers).
// Loads are "uniform base pointer + 32-bit per-lane byte offset" (global_load ... v_off, s[base:base+1]): one
// v_add_u32 per iteration advances all five planes (per-plane 64-bit pointers cost five v_lshl_add_u64).
template <int VEC>
struct Lane;
template <>
struct Lane<4> {
    using F = f2;
    using V = float4;
    static constexpr int kPairs = 2;
    static __device__ __forceinline__ V ld(const float* base, uint32_t byte_off) {
        typedef float v4 __attribute__((ext_vector_type(4)));
        const v4* p = reinterpret_cast<const v4*>(reinterpret_cast<const char*>(base) + byte_off);
        // every byte is read exactly once per sweep: stream it past the caches (global_load ... nt)
        const v4 t = __builtin_nontemporal_load(p);
        return make_float4(t.x, t.y, t.z, t.w);
    }
    // the lane's four sin(latitude) values into the library's scratch plane (read back once per later sweep: streamed)
    static __device__ __forceinline__ void st(float* base, uint32_t byte_off, const F (&v)[2]) {
        typedef float v4 __attribute__((ext_vector_type(4)));
        __builtin_nontemporal_store(v4{v[0].x, v[0].y, v[1].x, v[1].y},
                                    reinterpret_cast<v4*>(reinterpret_cast<char*>(base) + byte_off));
    }
    static __device__ __forceinline__ F get(const V& v, int k) { return k == 0 ? f2{v.x, v.y} : f2{v.z, v.w}; }
    static __device__ __forceinline__ V ones() { return make_float4(1.f, 1.f, 1.f, 1.f); }
    static __device__ __forceinline__ bool any_zero(F a) { return a.x == 0.f || a.y == 0.f; }
    static __device__ __forceinline__ float max_of(const F (&t)[2]) { return fmaxf(fmaxf(fmaxf(t[0].x, t[0].y), t[1].x), t[1].y); }
    static __device__ __forceinline__ F fold(F a) { return f2{fold_halfpi(a.x), fold_halfpi(a.y)}; }
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
    static __device__ __forceinline__ V ld(const float* base, uint32_t byte_off) {
        return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off);
    }
    static __device__ __forceinline__ void st(float* base, uint32_t byte_off, const F (&v)[1]) {
        *reinterpret_cast<float*>(reinterpret_cast<char*>(base) + byte_off) = v[0];
    }
    static __device__ __forceinline__ F get(const V& v, int) { return v; }
    static __device__ __forceinline__ V ones() { return 1.f; }
    static __device__ __forceinline__ bool any_zero(F a) { return a == 0.f; }
    static __device__ __forceinline__ float max_of(const F (&t)[1]) { return t[0]; }
    static __device__ __forceinline__ F fold(F a) { return fold_halfpi(a); }
    static __device__ __forceinline__ F xcoord(int x, int) { return (float)x; }
};

// One lane's place in the sweep of an image (a pure function of the launch geometry: nothing here depends on the parameters)
template <int VEC>
struct LaneJob {
    bool live;
    int y, y_end, xu;
    uint32_t off, off_step;      // byte offset inside a plane (N < 2^30) and its advance per loop iteration
};
template <int VEC>
__device__ __forceinline__ LaneJob<VEC> lane_job(const SweepArgs& a, const int chunk) {
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int job = chunk * (kBlock / 64) + wave;        // jobs of an image: (rowblock, strip, wave of the tile)
    const int tile = job / a.wpt, tw = job - tile * a.wpt;
    const int rowblock = tile / a.nstrips, strip = tile - rowblock * a.nstrips;
    const int f = tw * 64 + lane;                        // lane index inside the tile
    const int tr = f / a.cu, tc = f - tr * a.cu;
    LaneJob<VEC> j;
    j.xu = strip * a.cu + tc;                            // unit column of the lane
    j.live = job < a.jobs && tr < a.rpi && j.xu < a.wu;
    j.y_end = min((rowblock + 1) * a.rows_per_block, a.hrows);     // hrows = H, or H / 2 for the row-pair walk (MIRROR)
    j.y = rowblock * a.rows_per_block + tr;
    j.off = ((uint32_t)j.y * (uint32_t)a.W + (uint32_t)(j.xu * VEC)) * 4u;
    j.off_step = (uint32_t)a.rpi * (uint32_t)a.W * 4u;
    return j;
}

// The five values a lane reads per loop iteration (one float4 -- or one pixel -- of every plane)
template <int VEC>
struct RowData {
    typename Lane<VEC>::V vux, vuy, vlat, vcu, vcl;
};
template <bool HAS_UP, bool HAS_UPC, bool HAS_LATC, int VEC>
__device__ __forceinline__ RowData<VEC> load_row(const float* upx, const float* upy, const float* lat, const float* upc,
                                                 const float* latc, const uint32_t off) {
    using L = Lane<VEC>;
    RowData<VEC> r;
    r.vcu = L::ones();
    r.vcl = L::ones();
    if constexpr (HAS_UP) {
        r.vux = L::ld(upx, off);
        r.vuy = L::ld(upy, off);
    }
    r.vlat = L::ld(lat, off);
    if constexpr (HAS_UP && HAS_UPC) r.vcu = L::ld(upc, off);
    if constexpr (HAS_LATC) r.vcl = L::ld(latc, off);
    return r;
}

// sin(latitude_field) of the lane's pixels of one row, per SLAT (see row_math)
template <int VEC, int SLAT>
__device__ __forceinline__ void row_slat(const RowData<VEC>& r, typename Lane<VEC>::F (&slat)[Lane<VEC>::kPairs],
                                         [[maybe_unused]] float* slat_out, [[maybe_unused]] const uint32_t off) {
    using L = Lane<VEC>;
    using F = typename L::F;
    // latitudes beyond +-pi/2 (never from the CNN head; a caller's own field may hold them): fold them into the
    // polynomial's range.  Wave-uniform branch on a ballot: in-range data pays one max3 / max / cmp per 4 pixels.
    if constexpr (SLAT == 2) {
#pragma unroll
        for (int k = 0; k < L::kPairs; ++k) slat[k] = L::get(r.vlat, k);
    } else {
        F lt[L::kPairs], t[L::kPairs];
#pragma unroll
        for (int k = 0; k < L::kPairs; ++k) { lt[k] = L::get(r.vlat, k); t[k] = lt[k] * lt[k]; }
        if (__builtin_amdgcn_ballot_w64(L::max_of(t) > kHalfPiSq) != 0) {
#pragma unroll
            for (int k = 0; k < L::kPairs; ++k) { lt[k] = L::fold(lt[k]); t[k] = lt[k] * lt[k]; }
        }
#pragma unroll
        for (int k = 0; k < L::kPairs; ++k) slat[k] = sin_halfpi(lt[k], t[k]);
        if constexpr (SLAT == 1) L::st(slat_out, off, slat);
    }
}

// The math of one loop iteration: image row y of the lane's column(s), from the loaded values into the accumulators.
//
// SLAT -- sin(latitude_field) (lm_optimizer.py:262,270) does not depend on the parameters, yet every sweep of a solve would
// re-evaluate it per pixel (7 packed ops per pixel pair + the range test: ~18 of simple_radial's 265 VALU instructions per
// 4 pixels).  The VALU-bound distortion models therefore keep it in a LIBRARY-owned scratch plane (B x N floats in the
// handle's workspace; the boundary is unchanged: the caller still hands radians):
//   SLAT = 0  compute, keep in registers (pinhole; the one-launch-per-step kernel; the stand-alone stages)
//   SLAT = 1  compute as above AND store the plane (the first sweep of a solve; `slat_out` + `off`)
//   SLAT = 2  the `lat` plane of this launch IS that scratch plane: the loaded value is sin(latitude) (every later sweep)
// Same polynomial, same bits: a float stored and loaded is the float that was computed.
template <int MODEL, bool HAS_UP, bool LOGF, int VEC, int SLAT = 0>
__device__ __forceinline__ void row_math(const PBlock& P, const HuberK& hk, const typename Lane<VEC>::F (&col_u)[Lane<VEC>::kPairs],
                                         const typename Lane<VEC>::F (&col_px)[Lane<VEC>::kPairs], const int y,
                                         const RowData<VEC>& r, typename Lane<VEC>::F (&acc)[Layout<MODEL>::NACC],
                                         [[maybe_unused]] const bool col_zero, [[maybe_unused]] const bool div_k_tiny,
                                         [[maybe_unused]] float* slat_out = nullptr, [[maybe_unused]] const uint32_t off = 0) {
    using L = Lane<VEC>;
    using F = typename L::F;
    F slat[L::kPairs];
    row_slat<VEC, SLAT>(r, slat, slat_out, off);
    const float v = ((float)y - P.cy) * P.ify;
#if GCLM_NOMATH     // measurement only: the memory-system ceiling of this exact access pattern
#pragma unroll
    for (int k = 0; k < L::kPairs; ++k)
        acc[0] = acc[0] + (HAS_UP ? L::get(r.vux, k) + L::get(r.vuy, k) : F(0.f)) + L::get(r.vlat, k) + L::get(r.vcu, k) + L::get(r.vcl, k);
    (void)v;
    (void)hk;
#else
    if constexpr (MODEL == GCLM_SIMPLE_DIVISIONAL) {
        // a guarded denominator of radial_terms can only vanish on the principal point (r2 == 0: this wave holds
        // it in this iteration) or for k ~ 0 (the first sweep): only then are the guarded terms computed
        const bool patch = div_k_tiny || __builtin_amdgcn_ballot_w64(col_zero && v == 0.f) != 0;
#pragma unroll
        for (int k = 0; k < L::kPairs; ++k)
            pixel_accumulate<MODEL, HAS_UP, LOGF, F, 0, 1>(P, hk, col_u[k], col_px[k], v, HAS_UP ? L::get(r.vux, k) : F(0.f),
                                                               HAS_UP ? L::get(r.vuy, k) : F(0.f), slat[k], L::get(r.vcu, k),
                                                               L::get(r.vcl, k), acc, nullptr, nullptr, patch);
    } else {
#pragma unroll
        for (int k = 0; k < L::kPairs; ++k) {
            if constexpr (MODEL == GCLM_PINHOLE || MODEL == GCLM_SIMPLE_RADIAL)
                pixel_accumulate_fast<MODEL, HAS_UP, LOGF, F>(P, hk, col_u[k], col_px[k], v, HAS_UP ? L::get(r.vux, k) : F(0.f),
                                                        HAS_UP ? L::get(r.vuy, k) : F(0.f), slat[k],
                                                        L::get(r.vcu, k), L::get(r.vcl, k), acc);
            else
                pixel_accumulate<MODEL, HAS_UP, LOGF, F>(P, hk, col_u[k], col_px[k], v, HAS_UP ? L::get(r.vux, k) : F(0.f),
                                                   HAS_UP ? L::get(r.vuy, k) : F(0.f), slat[k], L::get(r.vcu, k),
                                                   L::get(r.vcl, k), acc);
        }
    }
#endif
}

// MIRROR (radial / simple_divisional, float4 path, five planes, even H) -- ROW PAIRS.  Everything a radial model adds to the
// pixel body is a function of r2 = u^2 + v^2 alone: the radial terms (simple_divisional: ~165 of its 417 VALU instructions per
// 4 pixels; radial: ~35 of 305), |ray|^2 and its rsq.  The principal point of every camera the library initialises is the
// image centre (camera.py:136-152), so rows y and H - y of a column have v' = -v EXACTLY (integer pixel coordinates,
// cy = H / 2) and the same r2 bit for bit.  A lane therefore walks the upper half of its column and takes row H - y along
// with row y: pixel_shared is evaluated ONCE, pixel_stage twice, the second time with -v (what pixel_stage derives from r2
// alone -- tau r2, |ray|^2, its rsq -- is the same expression on the same values in both calls and is merged by the
// compiler's value numbering).  Same operations on the same values, hence the same per-pixel bits as the one-row walker;
// only the order in which a lane adds its pixels changes.
// SHARE = false: the pair is walked WITHOUT sharing -- the second row gets its own v and its own pixel_shared.  For the one
// iteration that holds row 0 (row 0 has no mirror image and row H / 2 is its own: they pair up with each other) and for an
// image whose principal point is not the centre (an explicit camera, `scales`); sweep_body picks the copy, wave-uniformly.
template <int MODEL, bool HAS_UP, bool LOGF, int VEC, int SLAT, bool SHARE>
__device__ __forceinline__ void row_math_mirror(const PBlock& P, const HuberK& hk, const typename Lane<VEC>::F (&col_u)[Lane<VEC>::kPairs],
                                                const typename Lane<VEC>::F (&col_px)[Lane<VEC>::kPairs], const int y, const int y2,
                                                const RowData<VEC>& r, const RowData<VEC>& rm,
                                                typename Lane<VEC>::F (&acc)[Layout<MODEL>::NACC], const bool col_zero,
                                                const bool div_k_tiny, float* slat_out, const uint32_t off, const uint32_t off2) {
    using L = Lane<VEC>;
    using F = typename L::F;
    F slat[L::kPairs], slat2[L::kPairs];
    row_slat<VEC, SLAT>(r, slat, slat_out, off);
    row_slat<VEC, SLAT>(rm, slat2, slat_out, off2);
    const float v = ((float)y - P.cy) * P.ify;
    const float v2 = SHARE ? -v : ((float)y2 - P.cy) * P.ify;
    constexpr int GM = MODEL == GCLM_SIMPLE_DIVISIONAL ? 1 : 0;
    const bool patch = div_k_tiny || __builtin_amdgcn_ballot_w64(col_zero && (v == 0.f || v2 == 0.f)) != 0;
#pragma unroll
    for (int k = 0; k < L::kPairs; ++k) {
        PixelShared<MODEL, F> S;
        pixel_shared<MODEL, LOGF, F, GM>(P, col_u[k], v, patch, S);
        if constexpr (SHARE && LOGF && GCLM_LAT_PAIRS) {
            // both up fields, then both latitude fields with their sums taken together (lat_pair_accumulate)
            pixel_stage<MODEL, HAS_UP, LOGF, F, 0, 1>(P, hk, S, col_u[k], col_px[k], v, L::get(r.vux, k), L::get(r.vuy, k), slat[k],
                                                      L::get(r.vcu, k), L::get(r.vcl, k), acc);
            pixel_stage<MODEL, HAS_UP, LOGF, F, 0, 1>(P, hk, S, col_u[k], col_px[k], v2, L::get(rm.vux, k), L::get(rm.vuy, k), slat2[k],
                                                      L::get(rm.vcu, k), L::get(rm.vcl, k), acc);
            LatRow<F> la, lb;
            pixel_stage<MODEL, HAS_UP, LOGF, F, 3, 2>(P, hk, S, col_u[k], col_px[k], v, L::get(r.vux, k), L::get(r.vuy, k), slat[k],
                                                      L::get(r.vcu, k), L::get(r.vcl, k), acc, nullptr, nullptr, &la);
            pixel_stage<MODEL, HAS_UP, LOGF, F, 3, 2>(P, hk, S, col_u[k], col_px[k], v2, L::get(rm.vux, k), L::get(rm.vuy, k), slat2[k],
                                                      L::get(rm.vcu, k), L::get(rm.vcl, k), acc, nullptr, nullptr, &lb);
            lat_pair_accumulate<MODEL, F>(acc, S, la, lb);
        } else {
        pixel_stage<MODEL, HAS_UP, LOGF, F>(P, hk, S, col_u[k], col_px[k], v, L::get(r.vux, k), L::get(r.vuy, k), slat[k],
                                            L::get(r.vcu, k), L::get(r.vcl, k), acc);
        if constexpr (!SHARE) pixel_shared<MODEL, LOGF, F, GM>(P, col_u[k], v2, patch, S);
        pixel_stage<MODEL, HAS_UP, LOGF, F>(P, hk, S, col_u[k], col_px[k], v2, L::get(rm.vux, k), L::get(rm.vuy, k), slat2[k],
                                            L::get(rm.vcu, k), L::get(rm.vcl, k), acc);
        }
    }
}

// COLUMN-STATIONARY mapping.  The unit of work is a WAVE JOB: 64 consecutive lanes of a tile of `rpi` rows x `cu`
// units (float4 groups, or pixels in the scalar path) of ONE image, walking down `rows_per_block` rows, `rpi` rows
// per iteration; lane f of the tile sits at (row f / cu, unit f % cu) and NEVER changes its column.  With a single
// strip (cu = units per row: every width up to 2048 px) a tile is rpi whole rows = one contiguous run of memory, so
// a wave still reads 1 KiB of consecutive addresses per plane and instruction (640 px: 160 units per row, 2 rows
// = 320 lanes = 5 waves per tile).  A 256-thread workgroup takes four consecutive jobs of an image, whatever
// tiles they belong to, and writes one partial record.  What it buys: everything that depends on the column
// only -- u = (x - cx)/fx, p_x = ga - gc u, the int -> float conversions behind them -- leaves the loop, the byte
// offset advances by a wave-uniform constant, and the loop bookkeeping is one add and one compare (the row-major
// streaming it replaced spent ~21 of 295 VALU instructions per 4 pixels on these; scripts/isa_stats.py).
//
// PRE > 0 (one-launch-per-step kernel only): the values of the lane's first PRE iterations were requested by the caller
// before its prologue (`pre`, with the lane's job `pj`), so their memory round trip runs under the prologue's.
template <int MODEL, bool HAS_UP, bool HAS_UPC, bool HAS_LATC, bool LOGF, int VEC, int PRE = 0, int SLAT = 0, bool MIRROR = false>
__device__ __forceinline__ void sweep_body(const SweepArgs& a, const PBlock& P, const int b, const int chunk,
                                           [[maybe_unused]] const LaneJob<VEC>* pj = nullptr,
                                           [[maybe_unused]] const RowData<VEC>* pre = nullptr) {
    constexpr int NACC = Layout<MODEL>::NACC;
    const int tid = threadIdx.x;
    HuberK hk;
    hk.a2u = a.up_scale * a.up_scale;
    hk.inv_a2u = 1.0f / hk.a2u;
    hk.a2l = a.lat_scale * a.lat_scale;
    hk.inv_a2l = 1.0f / hk.a2l;

    const size_t N = (size_t)a.H * a.W;
    const float* upx = HAS_UP ? a.up + (size_t)b * 2 * N : nullptr;
    const float* upy = HAS_UP ? upx + N : nullptr;
    const float* lat = a.lat + (size_t)b * N;             // (SLAT == 2: the scratch plane of sin(latitude), see row_math)
    const float* upc = HAS_UPC ? a.upc + (size_t)b * N : nullptr;
    const float* latc = HAS_LATC ? a.latc + (size_t)b * N : nullptr;
    [[maybe_unused]] float* slat_out = SLAT == 1 ? a.slat + (size_t)b * N : nullptr;

    using L = Lane<VEC>;
    using F = typename L::F;
    F acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = F(0.f);

    // this wave's job, the lane's place in its tile (once per kernel), its column terms, its first row
    const int lane = tid & 63, wave = tid >> 6;
    const LaneJob<VEC> j = PRE > 0 ? *pj : lane_job<VEC>(a, chunk);
    const int xu = j.xu;
    const bool live = j.live;
    const int y_end = j.y_end;
    int y = j.y;
    uint32_t off = j.off;
    const uint32_t off_step = j.off_step;
    F col_u[L::kPairs], col_px[L::kPairs];
#pragma unroll
    for (int k = 0; k < L::kPairs; ++k) {
        col_u[k] = (L::xcoord(xu * VEC, k) - P.cx) * P.ifx;                           // camera.py:309-311
        col_px[k] = vfma(col_u[k], vsplat(col_u[k], -P.gc), vsplat(col_u[k], P.ga));  // p_x = ga - gc u
    }
    // simple_divisional: can a guarded denominator vanish?  (k: wave-uniform; the column part of r2 == 0: loop invariant)
    [[maybe_unused]] bool col_zero = false;
    [[maybe_unused]] const bool div_k_tiny = !(fabsf(P.k1) >= 1e-20f);
    if constexpr (MODEL == GCLM_SIMPLE_DIVISIONAL) {
#pragma unroll
        for (int k = 0; k < L::kPairs; ++k) col_zero = col_zero || L::any_zero(col_u[k]);
    }
    if (live) {
        if constexpr (PRE > 0) {
#pragma unroll
            for (int it = 0; it < PRE; ++it) {
                if (y < y_end) {
                    row_math<MODEL, HAS_UP, LOGF, VEC>(P, hk, col_u, col_px, y, pre[it], acc, col_zero, div_k_tiny);
                    y += a.rpi;
                    off += off_step;
                }
            }
        }
        if constexpr (MIRROR) {
            // row pairs (row_math_mirror): this lane's rows are [y, y_end) of the UPPER half, each with its mirror image.
            // Three copies of the body, ONE of them hot: the shared-terms loop.  (A wave-uniform `share` tested inside a single
            // loop costs 32-44 v_mov per iteration: the accumulators of the two bodies meet in phi copies.)
            static_assert(PRE == 0 && VEC == 4 && HAS_UP && HAS_UPC && HAS_LATC, "row pairs: five-plane float4 sweep");
            const bool centered = P.cy + P.cy == (float)a.H;                  // workgroup-uniform (scalar compare)
            const uint32_t row_bytes = (uint32_t)a.W * 4u, col_bytes = (uint32_t)(xu * VEC) * 4u;
            auto pair = [&](auto share) {
                const int y2 = y == 0 ? (a.H >> 1) : a.H - y;
                const uint32_t off2 = (uint32_t)y2 * row_bytes + col_bytes;
                const RowData<VEC> r = load_row<HAS_UP, HAS_UPC, HAS_LATC, VEC>(upx, upy, lat, upc, latc, off);
                const RowData<VEC> rm = load_row<HAS_UP, HAS_UPC, HAS_LATC, VEC>(upx, upy, lat, upc, latc, off2);
                __builtin_amdgcn_sched_barrier(0);
                row_math_mirror<MODEL, HAS_UP, LOGF, VEC, SLAT, decltype(share)::value>(P, hk, col_u, col_px, y, y2, r, rm, acc, col_zero,
                                                                                    div_k_tiny, slat_out, off, off2);
                y += a.rpi;
                off += off_step;
            };
            if (centered) {
                // row 0 has no mirror image (it pairs with row H / 2, nothing shared): the wave that holds it takes its
                // first iteration through the unshared body
                if (__builtin_amdgcn_ballot_w64(y == 0) != 0 && y < y_end) pair(std::false_type{});
                while (y < y_end) pair(std::true_type{});
            } else {
                while (y < y_end) pair(std::false_type{});
            }
        } else
        for (; y < y_end; y += a.rpi, off += off_step) {
            const RowData<VEC> r = load_row<HAS_UP, HAS_UPC, HAS_LATC, VEC>(upx, upy, lat, upc, latc, off);
            // keep every load of the iteration ahead of the math: left alone, the scheduler sinks loads next to
            // their first use to save registers in some instantiations (load -> wait -> use, five times over)
            __builtin_amdgcn_sched_barrier(0);
            row_math<MODEL, HAS_UP, LOGF, VEC, SLAT>(P, hk, col_u, col_px, y, r, acc, col_zero, div_k_tiny, slat_out, off);
        }
    }

    // wave64 DPP sum, then the 4 waves through LDS; one record per workgroup
    __shared__ float red[kBlock / 64][NACC];
    float wsum[NACC];            // all sums first, ONE predicated store block: the 16 DPP chains interleave
#pragma unroll
    for (int i = 0; i < NACC; ++i) wsum[i] = wave_sum(hsum(acc[i]));
    if (lane == kWaveSumLane) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) red[wave][i] = wsum[i];
    }
    __syncthreads();
    if (tid < NACC) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < kBlock / 64; ++w) s += red[w][tid];
        if (tid == A_CU) s *= hk.a2u;          // costs were accumulated in units of a^2
        if (tid == A_CL) s *= hk.a2l;
        a.partials[((size_t)b * a.nchunks + chunk) * NACC + tid] = s;
    }
}

// launch bounds: radial / simple_divisional are held to 168 VGPRs (3 waves per SIMD); simple_radial reaches 112 (4 waves) on its
// own; the log-focal pinhole sweep is held to 80 (6 waves; its launches keep 3 resident: GCLM_PINHOLE_LDS); its general-focal instantiation would spill at 80 and keeps its own
// 96, and so do its scratch-plane instantiations SLAT != 0 (never the library's built-in choice for pinhole)
template <int MODEL, bool HAS_UP, bool HAS_UPC, bool HAS_LATC, bool LOGF, int VEC, int SLAT = 0, bool MIRROR = false>
__global__ __launch_bounds__(kBlock, MIRROR ? (MODEL == GCLM_RADIAL && LOGF ? GCLM_MIRROR_WAVES_RADIAL : GCLM_MIRROR_WAVES) : (VEC == 4 && MODEL == GCLM_RADIAL) ? GCLM_RADIAL_WAVES : (VEC == 4 && MODEL > GCLM_RADIAL) ? GCLM_DIV_WAVES : (VEC == 4 && MODEL == GCLM_SIMPLE_RADIAL) ? 4 : (VEC == 4 && MODEL == GCLM_PINHOLE && LOGF) ? (SLAT == 0 ? GCLM_PINHOLE_WAVES : 5) : GCLM_MIN_WAVES) void sweep_kernel(
    const SweepArgs a) {
    if (stop_fired_before(a.ctrl, a.stop_step)) return;   // batch-global early stop, no host sync
    const int b = blockIdx.y, chunk = blockIdx.x;
    const PBlock P = a.pb[b];                            // workgroup-uniform -> scalar loads
    sweep_body<MODEL, HAS_UP, HAS_UPC, HAS_LATC, LOGF, VEC, 0, SLAT, MIRROR>(a, P, b, chunk);
}

// ONE launch per LM step for small batches (the interactive B = 1 case of the reference's demo, interactive_demo.py:403):
// the per-image update of step k-1 -- reduction of that step's partial records, lambda rule, damped Cholesky, manifold
// update (gclm_device.h: lm_step, the very function update_kernel runs) -- is done REDUNDANTLY in the prologue of every
// workgroup of sweep k, so an LM step is one launch instead of two and the host issues num_steps + 2 launches instead
// of 2 num_steps + 4.  Every workgroup of an image computes the same bits (fixed reduction order); workgroup 0 of the
// image commits the new state and the early-stop counter.  Partial records are double-buffered (launch k reads the
// records of launch k-1 while it writes its own).  Early stop: this kernel is only used when the decision is local to
// a workgroup -- B == 1 -- or off (gclm_api.hip: use_fused): the stop fires in the prologue that detects it, nothing is
// committed, nobody sweeps, and every later launch returns at its first instruction pair.
// The final launch (is_final) does the same with the last update, then builds the (roll, pitch, focal) block of the
// uncertainty sweep (prep_final_kernel's job) and sweeps with it.
template <int MODEL, bool HAS_UP, bool HAS_UPC, bool HAS_LATC, bool LOGF>
__global__ __launch_bounds__(kBlock) void fused_step_kernel(const SweepArgs a, const FusedArgs f) {
    using namespace dev;
    constexpr int PM = Layout<MODEL>::PM, NACC = Layout<MODEL>::NACC;
    const SolveCtx& c = f.c;
    const gclm_config& cfg = c.cfg;
    const int b = blockIdx.y, chunk = blockIdx.x, step = f.step;      // this launch sweeps theta_step (final: theta_final)
    GCLM_T(step, 0);
    __shared__ PBlock Ps;
    __shared__ int go;
    int stop_j = 0;                                                    // > 0: the stop fired after update stop_j (earlier launch)
    if (cfg.early_stop) {
        if (!f.is_final) {
            // every launch after the one that detected the stop leaves its counter at 0 too (see Ctrl).  (Testing the counter
            // only after the partial records have been requested as well -- one round trip instead of two -- was measured: the
            // active launches gain nothing, the skipped ones cost 3.0 instead of 2.3 us each; profiles/archive/r04_latency_trace.log.)
            if (step >= 3 && c.ctrl->notclose[step - 2] == 0) return;
        } else {
            // the first counter in [1, step - 2] that stayed 0: all of them in one round trip, 64 per wave-load
            const int lane = threadIdx.x & 63;
            for (int j0 = 1; j0 <= step - 2 && stop_j == 0; j0 += 64) {
                const int j = j0 + lane;
                const int v = j <= step - 2 ? c.ctrl->notclose[j] : 1;
                const unsigned long long m = __builtin_amdgcn_ballot_w64(v == 0);
                if (m) stop_j = j0 + __builtin_ctzll(m);
            }
        }
    }
    // The lane's first kFusedPre loop iterations are REQUESTED here, before the prologue: which bytes a lane reads depends on
    // the launch geometry only, not on the parameters, so their memory round trip (~1.5 us of a 2.3 us single-image sweep)
    // runs under the round trip of the partial records instead of after the parameter block is built.  A single image is cut
    // into two iterations per workgroup (plan_geometry), i.e. everything it reads is in flight before the update starts.
    constexpr int kFusedPre = 2;
    const LaneJob<4> pj = lane_job<4>(a, chunk);
    RowData<4> pre[kFusedPre];
    auto request_fields = [&]() {
        const size_t N = (size_t)a.H * a.W;
        const float* upx = HAS_UP ? a.up + (size_t)b * 2 * N : nullptr;
        const float* upy = HAS_UP ? upx + N : nullptr;
        const float* lat = a.lat + (size_t)b * N;
        const float* upc = HAS_UPC ? a.upc + (size_t)b * N : nullptr;
        const float* latc = HAS_LATC ? a.latc + (size_t)b * N : nullptr;
#pragma unroll
        for (int it = 0; it < kFusedPre; ++it) {
            pre[it].vux = pre[it].vuy = pre[it].vlat = pre[it].vcu = pre[it].vcl = Lane<4>::ones();
            if (pj.live && pj.y + it * a.rpi < pj.y_end)
                pre[it] = load_row<HAS_UP, HAS_UPC, HAS_LATC, 4>(upx, upy, lat, upc, latc, pj.off + (uint32_t)it * pj.off_step);
        }
    };
    State fin;                                                         // thread 0: the state this launch sweeps at
    bool commit = false, moved = false, stop_now = false;
    bool parts = false;                                                // tangent basis / reciprocals already in LDS
    const bool sph = cfg.use_spherical_manifold != 0;
    // The update of step - 1 is a serial chain per image (reduction -> lambda rule -> damped Cholesky -> manifold / focal /
    // distortion update -> tangent basis of the new gravity): ~4 us on one lane, on the critical path of every launch of a
    // single-image solve.  Its independent pieces run on the leaders of three WAVES (different SIMDs of the CU) at once:
    //   while thread 0 solves the normal equations, thread 64 prepares what the manifold update needs of the OLD gravity;
    //   then thread 0: new gravity + its tangent basis | thread 64: new focal + reciprocals | thread 128: new distortion.
    // The operations and their order per quantity are those of lm_step / build_pblock (gclm_device.h), so the result is the
    // two-launch path's bit for bit (test_one_launch_per_step_equals_the_two_launch_sequence).
    __shared__ StepDelta sh_dl;
    __shared__ GravPre sh_pre;
    __shared__ int sh_stop;
    __shared__ float sh_T[3][2], sh_g[3], sh_f[4], sh_k[2];
    if (step > 0 && stop_j == 0) {
        const int tid = threadIdx.x;
        State prev{};
        if (tid == 0 || tid == 64 || tid == 128) prev = c.state[(step - 1) & 1][b];   // in flight together with the partial records
        float acc[kNAccMax];
        // (the field values are requested BEHIND the partial records: the reduction then does not wait for them)
        reduce_image_partials(f.partials_in + (size_t)b * a.nchunks * NACC, a.nchunks, NACC, acc, request_fields);
        GCLM_T(step, 1);
        if (tid == 0) {
            fin = prev;
            StepDelta dl;
            moved = lm_solve<PM>(cfg, c.H, c.W, step - 1, fin, acc, dl);
            GCLM_T(step, 2);
            stop_now = cfg.early_stop && step - 1 >= 1 && !moved;     // B == 1: this image IS the batch (:619-625)
            sh_dl = dl;
            sh_stop = stop_now ? 1 : 0;
        } else if (tid == 64 && sph) {
            GravPre pre;
            grav_update_pre({prev.gx, prev.gy, prev.gz}, pre);
            sh_pre = pre;
        }
        __syncthreads();
        GCLM_T(step, 3);
        const bool stop_all = sh_stop != 0;                            // block-uniform
        if (!stop_all || f.is_final) {                                 // (a stop in a loop launch: nobody sweeps, nothing to build)
            if (tid == 0) {
                V3 g = {prev.gx, prev.gy, prev.gz};
                if (!stop_all) {
                    const StepDelta dl = sh_dl;
                    if (sph) { const GravPre pre = sh_pre; g = grav_update_post(pre, dl.dg0, dl.dg1); }
                    else g = grav_update(g, dl.dg0, dl.dg1, false);
                }
                float T[3][2];
                pblock_tangent(g, f.is_final ? false : sph, T);
                sh_g[0] = g.x; sh_g[1] = g.y; sh_g[2] = g.z;
#pragma unroll
                for (int i = 0; i < 3; ++i) { sh_T[i][0] = T[i][0]; sh_T[i][1] = T[i][1]; }
            } else if (tid == 64) {
                State t = prev;
                if (!stop_all) lm_apply_focal(cfg, t, sh_dl);
                sh_f[0] = t.fx; sh_f[1] = t.fy; sh_f[2] = 1.0f / t.fx; sh_f[3] = 1.0f / t.fy;
            } else if (tid == 128) {
                State t = prev;
                if (!stop_all) lm_apply_dist(cfg, t, sh_dl);
                sh_k[0] = t.k1; sh_k[1] = t.k2;
            }
            __syncthreads();
            GCLM_T(step, 4);
            parts = true;
        }
        if (tid == 0) {
            if (stop_now) {
                fin = prev;                                            // the tentative theta_step is discarded
            } else {
                fin.gx = sh_g[0]; fin.gy = sh_g[1]; fin.gz = sh_g[2];
                fin.fx = sh_f[0]; fin.fy = sh_f[1]; fin.k1 = sh_k[0]; fin.k2 = sh_k[1];
            }
            commit = !stop_now;
        }
    } else {
        request_fields();
        if (step == 0 && f.init_here) {
            // the first launch of a solve builds the initial estimate itself (every workgroup of the image, same bits) instead
            // of reading what an init_kernel launch left: one launch less at the head of a single-image solve.  Nothing in
            // THIS launch reads Ctrl's counters (step 0), later launches find them reset (stream order).
            if (chunk == 0 && b == 0) {
                for (int i = threadIdx.x; i < GCLM_MAX_STEPS + 4; i += kBlock) c.ctrl->notclose[i] = 0;
                if (threadIdx.x == 0) { c.ctrl->stopped = 0; c.ctrl->final_sel = cfg.num_steps & 1; }
            }
            if (threadIdx.x == 0) {
                fin = init_state(c, f.ia, b);
                if (chunk == 0) c.state[0][b] = fin;
            }
        } else if (threadIdx.x == 0) {
            fin = c.state[stop_j > 0 ? (stop_j & 1) : 0][b];           // an earlier stop's theta_j, or theta_0 (num_steps == 0)
        }
    }
    if (threadIdx.x == 0) {
        PBlock p;
        const bool lf = f.is_final ? c.iso_final != 0 : cfg.use_log_focal != 0;
        if (parts) {
            float T[3][2];
#pragma unroll
            for (int i = 0; i < 3; ++i) { T[i][0] = sh_T[i][0]; T[i][1] = sh_T[i][1]; }
            fill_pblock(fin, T, sh_f[2], sh_f[3], lf, p);
        } else {
            build_pblock(fin, f.is_final ? false : sph, lf, p);
        }
        Ps = p;
        go = (f.is_final || !stop_now) ? 1 : 0;
        if (chunk == 0) {
            // paced launches: tell the host how far image 0 has come and whether the stop fired, so that it stops issuing
            // launches nobody needs (a system-scope store into host-mapped memory; nothing else is ordered by it)
            if (f.progress && b == 0)
                __hip_atomic_store(f.progress, (f.epoch << kPacedEpochShift) | (stop_now ? kPacedStopBit : 0u) | (unsigned)(step + 1),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (commit) {
                c.state[step & 1][b] = fin;
                if (moved) atomicAdd(&c.ctrl->notclose[step - 1], 1);
            }
            if (f.is_final && b == 0) {                                // what prep_final_kernel leaves for finalize_kernel
                const int sj = stop_j > 0 ? stop_j : (stop_now ? step - 1 : 0);
                c.ctrl->stopped = sj > 0 ? 1 : 0;
                c.ctrl->final_sel = sj > 0 ? (sj & 1) : (step & 1);
            }
        }
    }
    __syncthreads();
    GCLM_T(step, 5);
    if (!go) return;
    // workgroup-uniform parameter block: LDS -> SGPRs (the sweep addresses it as scalar operands)
    PBlock P;
    {   // field by field: addressed as a float array, a slice of P stayed an alloca in one instantiation
        auto uni = [](float x) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, x))); };
        P.ifx = uni(Ps.ifx); P.ify = uni(Ps.ify); P.cx = uni(Ps.cx); P.cy = uni(Ps.cy);
        P.ga = uni(Ps.ga); P.gb = uni(Ps.gb); P.gc = uni(Ps.gc); P.k1 = uni(Ps.k1);
        P.T00 = uni(Ps.T00); P.T01 = uni(Ps.T01); P.T10 = uni(Ps.T10); P.T11 = uni(Ps.T11);
        P.T20 = uni(Ps.T20); P.T21 = uni(Ps.T21); P.wfx = uni(Ps.wfx); P.wfy = uni(Ps.wfy);
        P.k2 = uni(Ps.k2); P.pad0 = P.pad1 = P.pad2 = 0.f;
    }
    sweep_body<MODEL, HAS_UP, HAS_UPC, HAS_LATC, LOGF, 4, kFusedPre>(a, P, b, chunk, &pj, pre);
    GCLM_T(step, 6);
}

// Per-pixel Jacobian fields of the prediction (perspective_fields.py:323-365), the reference's
// J_perspective_field as a kernel: one thread per pixel, the parameter block built once per workgroup in LDS.
// J_up (B,H,W,2,PN), J_lat (B,H,W,1,PN), PN = 3 + #dist.  Not on the solve path (which never materialises these).
template <int MODEL>
__global__ __launch_bounds__(kBlock) void jacobian_kernel(const float* cam, const float* grav, int H, int W,
                                                          int spherical, int log_focal, float* J_up, float* J_lat) {
    using namespace dev;
    constexpr int PN = Layout<MODEL>::PN, NACC = Layout<MODEL>::NACC;
    __shared__ PBlock Ps;
    const int b = blockIdx.y;
    if (threadIdx.x == 0) {
        const float* cm = cam + (size_t)b * GCLM_CAM_STRIDE;
        State st{};
        st.w = cm[0]; st.h = cm[1]; st.fx = cm[2]; st.fy = cm[3]; st.cx = cm[4]; st.cy = cm[5]; st.k1 = cm[6]; st.k2 = cm[7];
        const V3 g = normalize3({grav[b * 3], grav[b * 3 + 1], grav[b * 3 + 2]});
        st.gx = g.x; st.gy = g.y; st.gz = g.z;
        PBlock p;
        build_pblock(st, spherical != 0, log_focal != 0, p);
        Ps = p;
    }
    __syncthreads();
    const size_t N = (size_t)H * W;
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= N) return;
    const PBlock P = Ps;
    const int y = (int)(i / W), x = (int)(i - (size_t)y * W);
    HuberK hk{1.f, 1.f, 1.f, 1.f};
    float acc[NACC];
    const size_t px = (size_t)b * N + i;
    const float u = ((float)x - P.cx) * P.ifx, v = ((float)y - P.cy) * P.ify;
    pixel_accumulate<MODEL, true, false, float, 1>(P, hk, u, fmaf(u, -P.gc, P.ga), v, 0.f, 0.f, 0.f, 1.f, 1.f, acc,
                                                      J_up ? J_up + px * 2 * PN : nullptr,
                                                      J_lat ? J_lat + px * PN : nullptr);
}

// Per-pixel residuals (lm_optimizer.py:248-274, calculate_residuals): r_up = up_data - up(theta) (B,N,2),
// r_lat = sin(lat_data) - sin(lat(theta)) (B,N,1), from the same pixel code.  Not on the solve path.
template <int MODEL>
__global__ __launch_bounds__(kBlock) void residual_kernel(const float* up, const float* lat, const float* cam,
                                                          const float* grav, int H, int W, float* r_up, float* r_lat) {
    using namespace dev;
    constexpr int NACC = Layout<MODEL>::NACC;
    __shared__ PBlock Ps;
    const int b = blockIdx.y;
    if (threadIdx.x == 0) {
        const float* cm = cam + (size_t)b * GCLM_CAM_STRIDE;
        State st{};
        st.w = cm[0]; st.h = cm[1]; st.fx = cm[2]; st.fy = cm[3]; st.cx = cm[4]; st.cy = cm[5]; st.k1 = cm[6]; st.k2 = cm[7];
        const V3 g = normalize3({grav[b * 3], grav[b * 3 + 1], grav[b * 3 + 2]});
        st.gx = g.x; st.gy = g.y; st.gz = g.z;
        PBlock p;
        build_pblock(st, false, false, p);
        Ps = p;
    }
    __syncthreads();
    const size_t N = (size_t)H * W;
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= N) return;
    const PBlock P = Ps;
    const int y = (int)(i / W), x = (int)(i - (size_t)y * W);
    HuberK hk{1.f, 1.f, 1.f, 1.f};
    float acc[NACC];
    const size_t px = (size_t)b * N + i;
    const float dux = up ? up[(size_t)b * 2 * N + i] : 0.f, duy = up ? up[(size_t)b * 2 * N + N + i] : 0.f;
    float dl = lat ? lat[px] : 0.f;
    if (beyond_halfpi(dl)) dl = fold_halfpi(dl);            // torch.sin of any latitude (lm_optimizer.py:262,270)
    dl = sin_halfpi(dl);
    const float u = ((float)x - P.cx) * P.ifx, v = ((float)y - P.cy) * P.ify;
    pixel_accumulate<MODEL, true, false, float, 2>(P, hk, u, fmaf(u, -P.gc, P.ga), v, dux, duy, dl, 1.f, 1.f, acc,
                                                   r_up ? r_up + px * 2 : nullptr, r_lat ? r_lat + px : nullptr);
}

// calculate_costs (lm_optimizer.py:276-315) on residual rows of `dim` components (dim = 0: the input already holds
// |r|^2, as scaled_loss / huber_loss take it): Huber cost and weight at scale a, both times the confidence -- the
// sweep's own branch-free form -- and optionally the second derivative of the scaled loss (:87, / a^2 of :76).
__global__ void huber_costs_kernel(const float* residual, size_t n, int dim, float a, const float* conf, float* cost,
                                   float* weight, float* second) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float x2 = dim == 0 ? residual[i] : 0.f;
    for (int d = 0; d < dim; ++d) x2 = fmaf(residual[i * dim + d], residual[i * dim + d], x2);
    const float c = conf ? conf[i] : 1.0f;
    float acc = 0.f;
    const float a2 = a * a;
    const float wc = huber_accumulate(x2, 1.0f / a2, c, acc);
    if (cost) cost[i] = acc * a2;
    if (weight) weight[i] = wc;
    if (second) {
        const float y = x2 / a2;
        second[i] = y <= 1.0f ? 0.f : -wc / (2.0f * y * a2);
    }
}

template <int MODEL, int VEC>
hipError_t dispatch(const SweepArgs& a, hipStream_t s) {
    const dim3 grid(a.nchunks, a.B), block(kBlock);
    const bool up = a.up != nullptr, upc = up && a.upc != nullptr, latc = a.latc != nullptr;
    // the log-focal specialisation only for the vector path (the scalar path is the odd-shape fallback)
    const bool logf = VEC == 4 && a.log_focal != 0;
    // an occupancy cap by LDS reservation for the memory-bound model (GCLM_PINHOLE_LDS above)
    constexpr unsigned lds = (MODEL == GCLM_PINHOLE && VEC == 4 && GCLM_DYN_LDS == 0) ? GCLM_PINHOLE_LDS : GCLM_DYN_LDS;
#define GCLM_LAUNCH(U, UC, LC)                                                                          \
    do {                                                                                                \
        if (logf) hipLaunchKernelGGL((sweep_kernel<MODEL, U, UC, LC, VEC == 4, VEC>), grid, block, lds, s, a); \
        else hipLaunchKernelGGL((sweep_kernel<MODEL, U, UC, LC, false, VEC>), grid, block, lds, s, a);    \
    } while (0)
    // row pairs (sweep_body: MIRROR): the five-plane float4 sweeps of radial / simple_divisional over an even number of rows
    if (a.mirror != 0) {
        if constexpr (VEC == 4 && (MODEL == GCLM_RADIAL || MODEL == GCLM_SIMPLE_DIVISIONAL)) {
            if (!(up && upc && latc) || (a.H & 1) || a.hrows * 2 != a.H || a.slat_mode < 0 || a.slat_mode > 2 ||
                (a.slat_mode != 0 && a.slat == nullptr))
                return hipErrorInvalidValue;
#define GCLM_LAUNCH_MIRROR(LF)                                                                                              \
    do {                                                                                                                    \
        if (a.slat_mode == 0) hipLaunchKernelGGL((sweep_kernel<MODEL, true, true, true, LF, 4, 0, true>), grid, block, GCLM_DYN_LDS, s, a);      \
        else if (a.slat_mode == 1) hipLaunchKernelGGL((sweep_kernel<MODEL, true, true, true, LF, 4, 1, true>), grid, block, GCLM_DYN_LDS, s, a); \
        else hipLaunchKernelGGL((sweep_kernel<MODEL, true, true, true, LF, 4, 2, true>), grid, block, GCLM_DYN_LDS, s, a);             \
    } while (0)
            if (logf) GCLM_LAUNCH_MIRROR(true); else GCLM_LAUNCH_MIRROR(false);
#undef GCLM_LAUNCH_MIRROR
            return hipGetLastError();
        } else {
            return hipErrorInvalidValue;
        }
    }
    // the sin(latitude) scratch plane (row_math: SLAT) exists for the five-plane float4 sweeps of the distortion models
    if constexpr (VEC == 4 && (MODEL != GCLM_PINHOLE || GCLM_SLAT_PINHOLE)) {
        if (a.slat_mode != 0) {
            if (!(up && upc && latc) || a.slat == nullptr || a.slat_mode < 0 || a.slat_mode > 2) return hipErrorInvalidValue;
#define GCLM_LAUNCH_SLAT(LF)                                                                                         \
    do {                                                                                                             \
        if (a.slat_mode == 1) hipLaunchKernelGGL((sweep_kernel<MODEL, true, true, true, LF, 4, 1>), grid, block, GCLM_DYN_LDS, s, a); \
        else hipLaunchKernelGGL((sweep_kernel<MODEL, true, true, true, LF, 4, 2>), grid, block, GCLM_DYN_LDS, s, a);            \
    } while (0)
            if (logf) GCLM_LAUNCH_SLAT(true); else GCLM_LAUNCH_SLAT(false);
#undef GCLM_LAUNCH_SLAT
            return hipGetLastError();
        }
    } else if (a.slat_mode != 0) {
        return hipErrorInvalidValue;
    }
    if (up) {
        if (upc) { if (latc) GCLM_LAUNCH(true, true, true); else GCLM_LAUNCH(true, true, false); }
        else     { if (latc) GCLM_LAUNCH(true, false, true); else GCLM_LAUNCH(true, false, false); }
    } else {
        if (latc) GCLM_LAUNCH(false, false, true); else GCLM_LAUNCH(false, false, false);
    }
#undef GCLM_LAUNCH
    return hipGetLastError();
}

template <int MODEL>
hipError_t dispatch_fused(const SweepArgs& a, const FusedArgs& f, hipStream_t s) {
    const dim3 grid(a.nchunks, a.B), block(kBlock);
    const bool up = a.up != nullptr, upc = up && a.upc != nullptr, latc = a.latc != nullptr;
    const bool logf = a.log_focal != 0;
#define GCLM_LAUNCH(U, UC, LC)                                                                               \
    do {                                                                                                     \
        if (logf) hipLaunchKernelGGL((fused_step_kernel<MODEL, U, UC, LC, true>), grid, block, 0, s, a, f);  \
        else hipLaunchKernelGGL((fused_step_kernel<MODEL, U, UC, LC, false>), grid, block, 0, s, a, f);      \
    } while (0)
    if (up) {
        if (upc) { if (latc) GCLM_LAUNCH(true, true, true); else GCLM_LAUNCH(true, true, false); }
        else     { if (latc) GCLM_LAUNCH(true, false, true); else GCLM_LAUNCH(true, false, false); }
    } else {
        if (latc) GCLM_LAUNCH(false, false, true); else GCLM_LAUNCH(false, false, false);
    }
#undef GCLM_LAUNCH
    return hipGetLastError();
}

template <int MODEL>
hipError_t dispatch_model(const SweepArgs& a, hipStream_t s) {
    return a.vec == 4 ? dispatch<MODEL, 4>(a, s) : dispatch<MODEL, 1>(a, s);
}

}  // namespace

hipError_t launch_sweep(int camera_model, const SweepArgs& a, hipStream_t s) {
    if (a.B <= 0) return hipSuccess;
    switch (camera_model) {
        case GCLM_PINHOLE: return dispatch_model<GCLM_PINHOLE>(a, s);
        case GCLM_SIMPLE_RADIAL: return dispatch_model<GCLM_SIMPLE_RADIAL>(a, s);
        case GCLM_RADIAL: return dispatch_model<GCLM_RADIAL>(a, s);
        case GCLM_SIMPLE_DIVISIONAL: return dispatch_model<GCLM_SIMPLE_DIVISIONAL>(a, s);
        default: return hipErrorInvalidValue;
    }
}

bool sweep_has_mirror(int camera_model) { return camera_model == GCLM_RADIAL || camera_model == GCLM_SIMPLE_DIVISIONAL; }
bool sweep_mirror_builtin(int camera_model) { return sweep_has_mirror(camera_model) && ((GCLM_MIRROR_MODELS >> camera_model) & 1) != 0; }

bool sweep_has_slat_plane(int camera_model) {
    return camera_model > GCLM_PINHOLE ? camera_model <= GCLM_SIMPLE_DIVISIONAL : (camera_model == GCLM_PINHOLE && GCLM_SLAT_PINHOLE != 0);
}

hipError_t launch_fused_step(int camera_model, const SweepArgs& a, const FusedArgs& f, hipStream_t s) {
    if (a.B <= 0) return hipSuccess;
    if (a.vec != 4) return hipErrorInvalidValue;          // the caller only fuses the float4 path
    switch (camera_model) {
        case GCLM_PINHOLE: return dispatch_fused<GCLM_PINHOLE>(a, f, s);
        case GCLM_SIMPLE_RADIAL: return dispatch_fused<GCLM_SIMPLE_RADIAL>(a, f, s);
        case GCLM_RADIAL: return dispatch_fused<GCLM_RADIAL>(a, f, s);
        case GCLM_SIMPLE_DIVISIONAL: return dispatch_fused<GCLM_SIMPLE_DIVISIONAL>(a, f, s);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_residual_fields(int camera_model, const float* d_up, const float* d_lat, const float* d_cam,
                                  const float* d_grav, int B, int H, int W, float* d_r_up, float* d_r_lat, hipStream_t s) {
    if (B <= 0 || H <= 0 || W <= 0) return hipSuccess;
    const dim3 grid((unsigned)(((size_t)H * W + kBlock - 1) / kBlock), B), block(kBlock);
    switch (camera_model) {
#define GCLM_RES(M) \
    case M: hipLaunchKernelGGL(residual_kernel<M>, grid, block, 0, s, d_up, d_lat, d_cam, d_grav, H, W, d_r_up, d_r_lat); break
        GCLM_RES(GCLM_PINHOLE);
        GCLM_RES(GCLM_SIMPLE_RADIAL);
        GCLM_RES(GCLM_RADIAL);
        GCLM_RES(GCLM_SIMPLE_DIVISIONAL);
#undef GCLM_RES
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_huber_costs(const float* d_residual, size_t n, int dim, float scale, const float* d_conf,
                              float* d_cost, float* d_weight, float* d_second, hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(huber_costs_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d_residual, n, dim, scale,
                       d_conf, d_cost, d_weight, d_second);
    return hipGetLastError();
}

hipError_t launch_jacobian_fields(int camera_model, const float* d_cam, const float* d_grav, int B, int H, int W,
                                  int spherical, int log_focal, float* d_J_up, float* d_J_lat, hipStream_t s) {
    if (B <= 0 || H <= 0 || W <= 0) return hipSuccess;
    const dim3 grid((unsigned)(((size_t)H * W + kBlock - 1) / kBlock), B), block(kBlock);
    switch (camera_model) {
#define GCLM_JAC(M) \
    case M: hipLaunchKernelGGL(jacobian_kernel<M>, grid, block, 0, s, d_cam, d_grav, H, W, spherical, log_focal, d_J_up, d_J_lat); break
        GCLM_JAC(GCLM_PINHOLE);
        GCLM_JAC(GCLM_SIMPLE_RADIAL);
        GCLM_JAC(GCLM_RADIAL);
        GCLM_JAC(GCLM_SIMPLE_DIVISIONAL);
#undef GCLM_JAC
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace gclm
