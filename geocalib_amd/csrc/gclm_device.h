// gclm_device.h -- per-image device logic shared by the update kernels (gclm_update.hip) and the prologue of the
// one-launch-per-step sweep for single images (gclm_pass.hip: fused_step_kernel).  gfx950 only.
#pragma once
#include "gclm_internal.h"

namespace gclm {
namespace dev {

// The per-image arithmetic below is compiled into several kernels of two translation units (update_kernel,
// shared_step_kernel, fused_step_kernel ...) that must produce the SAME BITS (the one-launch-per-step path is checked
// bit for bit against the two-launch path).  With contraction on, the backend decides per inlining context which
// a * b + c becomes an fma, so contraction is switched off for this header: every operation here is an IEEE operation
// (the fmas that are wanted are written as fmaf).  Restored at the end of the file -- the per-pixel code keeps it.
// (Needs -ffp-contract=fast-honor-pragmas: plain `fast` ignores this pragma.)
#pragma clang fp contract(off)

constexpr float kPi = 3.14159265358979323846f;

struct V3 { float x, y, z; };

// ---------------------------------------------------------------- gravity / manifold (device)

// Gravity.roll / pitch (gravity.py:63-81)
__device__ inline float grav_roll(V3 g) {
    const float roll = asinf(-g.x / (sqrtf(1.0f - g.z * g.z) + 1e-4f));
    const float sgn = (g.x > 0.f) ? 1.f : ((g.x < 0.f) ? -1.f : 0.f);
    return g.y < 0.f ? roll : -roll - kPi * sgn;
}
__device__ inline float grav_pitch(V3 g) { return asinf(g.z); }

// Gravity.J_rp (gravity.py:69-101): T[i][k], k = roll, pitch
__device__ inline void tangent_rp(V3 g, float (&T)[3][2]) {
    const float r = grav_roll(g), p = grav_pitch(g);
    float sr, cr, sp, cp;
    sincosf(r, &sr, &cr);
    sincosf(p, &sp, &cp);
    T[0][0] = -cr * cp; T[1][0] = sr * cp; T[2][0] = 0.f;
    T[0][1] = sr * sp;  T[1][1] = cr * sp; T[2][1] = cp;
}

// SphericalManifold.householder_vector (misc.py:182-209), pivot = last component
__device__ inline void householder(V3 x, float (&v)[3], float& beta) {
    float sigma = x.x * x.x + x.y * x.y;
    const float norm = sqrtf(sigma + x.z * x.z);
    if (sigma < 1e-7f) sigma += 1e-7f;
    const float vpiv = x.z < 0.f ? x.z - norm : -sigma / (x.z + norm);
    beta = 2.f * vpiv * vpiv / (sigma + vpiv * vpiv);
    v[0] = x.x / vpiv; v[1] = x.y / vpiv; v[2] = 1.f;
}

// SphericalManifold.J_plus (misc.py:226-231)
__device__ inline void tangent_sphere(V3 g, float (&T)[3][2]) {
    float v[3], beta;
    householder(g, v, beta);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int k = 0; k < 2; ++k) T[i][k] = -beta * v[i] * v[k] + (i == k ? 1.f : 0.f);
}

__device__ inline V3 normalize3(V3 g) {
    const float n = fmaxf(sqrtf(g.x * g.x + g.y * g.y + g.z * g.z), 1e-12f);
    return {g.x / n, g.y / n, g.z / n};
}

__device__ inline V3 from_rp(float roll, float pitch) {   // gravity.py:31-40
    float sr, cr, sp, cp;
    sincosf(roll, &sr, &cr);
    sincosf(pitch, &sp, &cp);
    return normalize3({-sr * cp, -cr * cp, sp});
}

// Gravity.update (gravity.py:112-119) / SphericalManifold.plus (misc.py:234-259), in two halves: what depends on the OLD
// gravity only (its norm and Householder vector) and what needs the step.  The one-launch-per-step kernel computes the
// first half on another wave while the normal equations are still being solved (same operations, same bits).
struct GravPre {
    float v[3], beta, nx;
};
__device__ inline void grav_update_pre(V3 g, GravPre& p) {
    p.nx = sqrtf(g.x * g.x + g.y * g.y + g.z * g.z);
    householder(g, p.v, p.beta);
}
__device__ inline V3 grav_update_post(const GravPre& p, float d0, float d1) {
    const float eps = 1e-7f;
    const float nd = sqrtf(d0 * d0 + d1 * d1);
    const float nd_ = nd < eps ? nd + eps : nd;
    const float sinc = nd < eps ? 1.f : sinf(nd_) / nd_;
    const float e[3] = {sinc * d0, sinc * d1, cosf(nd)};
    const float bd = p.beta * (p.v[0] * e[0] + p.v[1] * e[1] + p.v[2] * e[2]);
    return normalize3({p.nx * (e[0] - p.v[0] * bd), p.nx * (e[1] - p.v[1] * bd), p.nx * (e[2] - p.v[2] * bd)});
}
__device__ inline V3 grav_update(V3 g, float d0, float d1, bool spherical) {
    if (!spherical) return from_rp(grav_roll(g) + d0, grav_pitch(g) + d1);
    GravPre p;
    grav_update_pre(g, p);
    return grav_update_post(p, d0, d1);
}

// BaseCamera.update_focal (camera.py:136-152): clamp to fov in [5, 150] deg of the image HEIGHT,
// fx rebuilt from fy by the old ratio.
__device__ inline void update_focal(State& s, float delta, bool as_log) {
    const float fy = as_log ? expf(logf(s.fy) + delta) : s.fy + delta;
    // fov2focal(deg2rad(150), h), fov2focal(deg2rad(5), h) with the float32 values of tan(75 deg), tan(2.5 deg)
    const float min_f = s.h * 0.5f / 3.7320504f;
    const float max_f = s.h * 0.5f / 0.043660946f;
    const float fyc = fminf(fmaxf(fy, min_f), max_f);
    s.fx = fyc * s.fx / s.fy;
    s.fy = fyc;
}

// SimpleRadial.update_dist (camera.py:599-604); slot 7 shadows k1 for one-parameter models
__device__ inline void update_dist(State& s, int camera_model, float d1, float d2) {
    // range (-0.7, 0.7), simple_divisional (-3, 3)  (camera.py:599, 700, 817)
    const float hi = camera_model == GCLM_SIMPLE_DIVISIONAL ? 3.0f : 0.7f;
    s.k1 = fminf(fmaxf(s.k1 + d1, -hi), hi);
    s.k2 = fminf(fmaxf(s.k2 + (camera_model == GCLM_RADIAL ? d2 : d1), -hi), hi);
}

// The parameter block of a state, in two halves: the tangent basis of its gravity (the long chain: a Householder vector,
// or asin / sincos for the (roll, pitch) form) and the rest (two reciprocals).  build_pblock = fill_pblock(tangent).
__device__ inline void pblock_tangent(V3 g, bool spherical, float (&T)[3][2]) {
    if (spherical) tangent_sphere(g, T); else tangent_rp(g, T);
}
__device__ inline void fill_pblock(const State& s, const float (&T)[3][2], float ifx, float ify, bool log_focal, PBlock& p) {
    p.ifx = ifx; p.ify = ify; p.cx = s.cx; p.cy = s.cy;
    p.ga = s.gx; p.gb = s.gy; p.gc = s.gz; p.k1 = s.k1;
    p.T00 = T[0][0]; p.T01 = T[0][1]; p.T10 = T[1][0]; p.T11 = T[1][1]; p.T20 = T[2][0]; p.T21 = T[2][1];
    p.wfx = log_focal ? 1.0f : ifx;
    p.wfy = log_focal ? 1.0f : ify;
    p.k2 = s.k2; p.pad0 = p.pad1 = p.pad2 = 0.f;
}
__device__ inline void build_pblock(const State& s, bool spherical, bool log_focal, PBlock& p) {
    float T[3][2];
    pblock_tangent({s.gx, s.gy, s.gz}, spherical, T);
    fill_pblock(s, T, 1.0f / s.fx, 1.0f / s.fy, log_focal, p);
}

// The initial estimate of image b: the caller's (gclm_solve) or get_trivial_estimation / siclib's heuristic from the priors
// (gclm_calibrate).  init_kernel runs it once per image; the one-launch-per-step kernel runs it in the prologue of its
// first launch instead (every workgroup of the image, same bits), which saves a single-image solve the init launch.
__device__ inline State init_state(const SolveCtx& c, const InitArgs& ia, int b) {
    State s;
    V3 g;
    if (ia.cam) {       // caller-provided initial estimate
        const float* cm = ia.cam + (size_t)b * GCLM_CAM_STRIDE;
        s.w = cm[0]; s.h = cm[1]; s.fx = cm[2]; s.fy = cm[3]; s.cx = cm[4]; s.cy = cm[5]; s.k1 = cm[6]; s.k2 = cm[7];
        g = normalize3({ia.grav[b * 3], ia.grav[b * 3 + 1], ia.grav[b * 3 + 2]});
    } else {            // get_trivial_estimation (lm_optimizer.py:20-58) + BaseCamera.from_dict (camera.py:49-93)
        const float h = (float)c.H, w = (float)c.W;
        const float focal = ia.prior_focal ? ia.prior_focal[b] : 0.7f * fmaxf(h, w);
        const float vfov = 2.0f * atanf(h / (2.0f * focal));           // focal2fov
        const float f = h / 2.0f / tanf(vfov / 2.0f);                   // fov2focal
        s.w = w; s.h = h; s.fy = f; s.cx = w / 2.0f; s.cy = h / 2.0f;
        s.fx = ia.scales ? f * ia.scales[0] / ia.scales[1] : f;
        s.k1 = s.k2 = 0.f;
        if (ia.prior_dist) {
            const int nd = ia.prior_dist_cols;
            s.k1 = ia.prior_dist[(size_t)b * nd];
            if (nd > 1) s.k2 = ia.prior_dist[(size_t)b * nd + 1];
        }
        g = V3{-0.0f, -1.0f, 0.0f};                                   // Gravity.from_rp(0, 0)
        if (c.cfg.heuristic_init && ia.up) {
            // get_heuristic_estimation (siclib/models/optimization/utils.py:27-82): roll = angle of the up
            // vector at the image centre, pitch = latitude at the centre, vfov = |lat(top) - lat(bottom)|
            // on the central column, all clamped; priors still win below
            const size_t N = (size_t)c.H * c.W;
            const int yc = c.H / 2, xc = c.W / 2;
            const float* up = ia.up + (size_t)b * 2 * N;
            const float* lat = ia.lat + (size_t)b * N;
            const float d45 = 45.0f / 180.0f * kPi;
            float roll = -atan2f(up[(size_t)yc * c.W + xc], -up[N + (size_t)yc * c.W + xc]);
            roll = fminf(fmaxf(roll, -d45), d45);
            const float pitch = fminf(fmaxf(lat[(size_t)yc * c.W + xc], -d45), d45);
            float vfov_h = fabsf(lat[xc] - lat[(size_t)(c.H - 1) * c.W + xc]);
            vfov_h = fminf(fmaxf(vfov_h, 20.0f / 180.0f * kPi), 120.0f / 180.0f * kPi);
            if (!ia.prior_focal) {
                const float fh = h / 2.0f / tanf(vfov_h / 2.0f);
                s.fy = fh;
                s.fx = ia.scales ? fh * ia.scales[0] / ia.scales[1] : fh;
            }
            g = from_rp(roll, pitch);
        }
        if (ia.prior_gravity) g = normalize3({ia.prior_gravity[b * 3], ia.prior_gravity[b * 3 + 1], ia.prior_gravity[b * 3 + 2]});
    }
    s.gx = g.x; s.gy = g.y; s.gz = g.z;
    s.lambda = c.cfg.lambda0; s.prev_cost = 0.f; s.fails = 0.f; s.init_cu = s.init_cl = 0.f;
    return s;
}

// ---------------------------------------------------------------- small dense algebra
// Everything here has compile-time sizes and fully unrolled loops so that the tiny matrices live in
// registers: a dynamically indexed local array is placed in scratch memory on gfx950 and made the
// first generic version of the update kernel 2.5x slower (42 us vs 17 us per launch at B = 1024).

// In-place Cholesky solve of an N x N SPD system (fp32 like torch.linalg.cholesky on fp32).
template <int N>
__device__ inline bool chol_solve(float (&A)[N][N], float (&b)[N]) {
    bool ok = true;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        float s = A[j][j];
#pragma unroll
        for (int k = 0; k < j; ++k) s -= A[j][k] * A[j][k];
        ok = ok && (s > 0.f);
        const float l = sqrtf(s);
        A[j][j] = l;
#pragma unroll
        for (int i = j + 1; i < N; ++i) {
            float t = A[i][j];
#pragma unroll
            for (int k = 0; k < j; ++k) t -= A[i][k] * A[j][k];
            A[i][j] = t / l;
        }
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
        float t = b[i];
#pragma unroll
        for (int k = 0; k < i; ++k) t -= A[i][k] * b[k];
        b[i] = t / A[i][i];
    }
#pragma unroll
    for (int i = N - 1; i >= 0; --i) {
        float t = b[i];
#pragma unroll
        for (int k = i + 1; k < N; ++k) t -= A[k][i] * b[k];
        b[i] = t / A[i][i];
    }
    return ok;
}

// Symmetric PM x PM system (PM = 4 or 5 full columns d1,d2,f,k1[,k2]) out of an accumulator record.
template <int PM>
__device__ inline void unpack_system(const float* acc, float (&Hm)[PM][PM], float (&G)[PM]) {
#pragma unroll
    for (int i = 0; i < PM; ++i) {
        G[i] = acc[A_G0 + i];
#pragma unroll
        for (int j = i; j < PM; ++j) Hm[i][j] = Hm[j][i] = acc[acc_h(PM, i, j)];
    }
}

// Which of the PM full columns are free (calculate_gradient_and_hessian, lm_optimizer.py:335-344).  The
// system is solved at its full size with the fixed columns replaced by identity rows -- arithmetically the
// reference's sub-matrix solve (the masked entries only ever contribute exact zeros), without dynamic indexing.
template <int PM>
__device__ inline void active_columns(const gclm_config& cfg, bool (&act)[PM]) {
    const int nd = num_dist_params(cfg.camera_model);
#pragma unroll
    for (int k = 0; k < PM; ++k)
        act[k] = k < 2 ? cfg.estimate_gravity != 0 : (k == 2 ? cfg.estimate_focal != 0 : k - 3 < nd);
}

// The steps of update_estimate (lm_optimizer.py:518-549) out of the full-size solution, reproducing the python
// index arithmetic of :223-235: dist_delta_dims start at focal_delta_dims[-1] + 1, which is 0 when the focal is a
// prior -- i.e. the distortion then takes the GRAVITY steps (reference quirk, kept).
template <int PM>
__device__ inline void split_delta(const gclm_config& cfg, const float (&d)[PM], float& dg0, float& dg1, float& df,
                                   float& dk1, float& dk2) {
    const bool eg = cfg.estimate_gravity != 0, ef = cfg.estimate_focal != 0;
    dg0 = eg ? d[0] : 0.f;
    dg1 = eg ? d[1] : 0.f;
    df = ef ? d[2] : 0.f;
    const float d4 = PM > 4 ? d[PM - 1] : 0.f;
    dk1 = (ef || !eg) ? d[3] : d[0];
    dk2 = (ef || !eg) ? d4 : d[1];
}

// sum(c.mean(-1) for c in costs.values()) (lm_optimizer.py:584,610), float32
__device__ inline float total_cost(const float* acc, float invN, bool has_up, float& cu, float& cl) {
    cu = acc[A_CU] * invN;
    cl = acc[A_CL] * invN;
    return has_up ? cu + cl : cl;
}

// lambda rule + "allclose" test of one image (lm_optimizer.py:95-106, :90-92, :612-627): returns whether the cost
// still MOVED at this step (always false at step 0, where nothing is compared); the caller counts it.
__device__ inline bool cost_rules(const gclm_config& cfg, int step, float total, State& s, bool update_lambda) {
    bool moved = false;
    if (step > 0) {
        if (update_lambda) {
            const float nl = s.lambda * (total > s.prev_cost ? 10.f : 0.1f);
            s.lambda = fminf(fmaxf(nl, 1e-6f), 1e2f);
        }
        // torch.allclose evaluates |new - prev| <= atol + |rtol * prev| in the tensors' dtype (float32: the scalar tolerances
        // do not promote; ATen isclose) -- so does this (rounds 1-3 used double: same decision unless the difference sits
        // within 1e-15 of the threshold, but it was a deviation nobody had written down)
        const float diff = fabsf(total - s.prev_cost);
        moved = !(diff <= cfg.atol + fabsf(cfg.rtol * s.prev_cost));
    }
    s.prev_cost = total;
    return moved;
}

// ... and the batch-global bookkeeping on top of it: every image whose cost moved counts into notclose[step]
__device__ inline void cost_bookkeeping(const gclm_config& cfg, Ctrl* ctrl, int step, float total,
                                        State& s, bool update_lambda) {
    if (cost_rules(cfg, step, total, s, update_lambda)) atomicAdd(&ctrl->notclose[step], 1);
}

// One LM step of one image from its reduced accumulator record, as pure arithmetic on (s, acc), in stages:
//   lm_solve          lambda rule + allclose test, damped normal equations over the estimated columns -> the steps `dl`
//                     (updates s.lambda / prev_cost / fails / init_*; returns whether the cost moved)
//   lm_apply_gravity / lm_apply_focal / lm_apply_dist    manifold / focal / distortion update: independent of each other
// lm_step runs them in sequence on one thread (update_kernel); the one-launch-per-step kernel runs the three updates on
// three waves at once (fused_step_kernel).  `s` is the state at `step` on entry and the state at step + 1 on return.
struct StepDelta {
    float dg0, dg1, df, dk1, dk2;
};
template <int PM>
__device__ inline bool lm_solve(const gclm_config& cfg, int H, int W, int step, State& s, const float (&acc)[kNAccMax],
                                StepDelta& dl) {
    const float invN = 1.0f / (float)((size_t)H * W);
    float cu, cl;
    const float total = total_cost(acc, invN, true, cu, cl);   // A_CU is 0 without an up field
    if (step == 0) { s.init_cu = cu; s.init_cl = cl; }     // infos["initial_*"] (:585-588)
    const bool moved = cost_rules(cfg, step, total, s, !cfg.fix_lambda);

    // damped normal equations over the estimated columns (lm_optimizer.py:109-137)
    float A[PM][PM], d[PM];
    bool act[PM];
    unpack_system<PM>(acc, A, d);
    active_columns<PM>(cfg, act);
#pragma unroll
    for (int i = 0; i < PM; ++i) {
#pragma unroll
        for (int j = 0; j < PM; ++j)
            if (!(act[i] && act[j])) A[i][j] = i == j ? 1.f : 0.f;
        if (act[i]) A[i][i] += fmaxf(A[i][i] * s.lambda, 1e-6f); else d[i] = 0.f;
    }
    bool ok = chol_solve<PM>(A, d);
#pragma unroll
    for (int i = 0; i < PM; ++i) ok = ok && fabsf(d[i]) <= 3.0e38f;      // a NaN / inf step (NaN gradient) is a failed step too
    if (!ok) {
#pragma unroll
        for (int i = 0; i < PM; ++i) d[i] = 0.f;     // zero step for THIS image (reference: whole batch)
        s.fails += 1.f;
    }
    split_delta<PM>(cfg, d, dl.dg0, dl.dg1, dl.df, dl.dk1, dl.dk2);
    return moved;
}
__device__ inline void lm_apply_gravity(const gclm_config& cfg, State& s, const StepDelta& dl) {
    const V3 g = grav_update({s.gx, s.gy, s.gz}, dl.dg0, dl.dg1, cfg.use_spherical_manifold != 0);
    s.gx = g.x; s.gy = g.y; s.gz = g.z;
}
__device__ inline void lm_apply_focal(const gclm_config& cfg, State& s, const StepDelta& dl) {
    update_focal(s, dl.df, cfg.use_log_focal != 0);
}
__device__ inline void lm_apply_dist(const gclm_config& cfg, State& s, const StepDelta& dl) {
    if (cfg.camera_model != GCLM_PINHOLE && cfg.estimate_dist) update_dist(s, cfg.camera_model, dl.dk1, dl.dk2);
}
template <int PM>
__device__ inline bool lm_step(const gclm_config& cfg, int H, int W, int step, State& s, const float (&acc)[kNAccMax]) {
    StepDelta dl;
    const bool moved = lm_solve<PM>(cfg, H, W, step, s, acc, dl);
    lm_apply_gravity(cfg, s, dl);
    lm_apply_focal(cfg, s, dl);
    lm_apply_dist(cfg, s, dl);
    return moved;
}

// ... applied to image b of a solve: state / parameter-block double buffers and the early-stop counters.
template <int PM>
__device__ inline void update_image(const SolveCtx& c, int step, int b, const float (&acc)[kNAccMax]) {
    State s = c.state[step & 1][b];
    if (lm_step<PM>(c.cfg, c.H, c.W, step, s, acc)) atomicAdd(&c.ctrl->notclose[step], 1);
    c.state[(step + 1) & 1][b] = s;
    PBlock p;
    build_pblock(s, c.cfg.use_spherical_manifold != 0, c.cfg.use_log_focal != 0, p);
    c.pb[(step + 1) & 1][b] = p;
}

// ---------------------------------------------------------------- reduction of the sweep's partial records
// Sum the workgroup partials of ONE image with the whole 256-thread block (8 groups x 32 slots), in the order the
// update kernels use: >= kStripeMinChunks records -- 8 stripes of chunks (stripe g: g, g + 8, ...), combined in stripe
// order; fewer -- one ascending walk.  Double accumulation, fixed order: every caller gets the same bits.  Must be
// called by all threads of the block; the sums are valid in EVERY thread on return (read back from LDS).
constexpr int kGroups = 8, kSlots = 32, kStripeMinChunks = 33;
// `after_first_batch()` runs right after the loads of the first batch of records have been ISSUED and before anything waits
// for them: the caller's own independent loads (the one-launch-per-step kernel's first field values) queue up BEHIND the
// records, so the reduction does not wait for them (loads return in order).
// The sums are valid in the threads of WAVE 0 on return (lane i of wave 0 sums slot i over the stripes, v_readlane hands
// the values to the whole wave): round 3 had every one of the 256 threads read all 8 x 24 doubles back from LDS -- 1 us of
// LDS bandwidth at the head of every launch of a single-image solve (device trace, profiles/archive/r04_latency_trace.log).
template <typename Hook>
__device__ inline void reduce_image_partials(const float* image_partials, int nchunks, int nacc, float (&acc)[kNAccMax],
                                             Hook&& after_first_batch) {
    __shared__ double sacc1[kGroups][kSlots + 1];
    const int grp = threadIdx.x / kSlots, slot = threadIdx.x % kSlots;
    const bool striped = nchunks >= kStripeMinChunks;
    const int first = striped ? grp : 0, stride = striped ? kGroups : 1;
    const bool active = slot < nacc && (striped || grp == 0);
    const float* p = image_partials + slot;
    // batches of 32 records per thread: the 19 records of a stripe of a single 640x480 image (150 workgroups) are all
    // in flight together -- one memory round trip; summed in ascending order (the padding adds exact zeros)
    float v[32];
    if (active) {
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            const int c = first + j * stride;
            const float t = p[(size_t)min(c, nchunks - 1) * nacc];      // unconditional load (a clamped index), then
            v[j] = c < nchunks ? t : 0.f;                                // a select: all loads of the batch in flight
        }
    }
    after_first_batch();                                                 // every thread, in uniform control flow
    if (active) {
        double d = 0.0;
#pragma unroll
        for (int j = 0; j < 32; ++j) d += v[j];
        for (int c0 = first + 32 * stride; c0 < nchunks; c0 += 32 * stride) {
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                const int c = c0 + j * stride;
                const float t = p[(size_t)min(c, nchunks - 1) * nacc];
                v[j] = c < nchunks ? t : 0.f;
            }
#pragma unroll
            for (int j = 0; j < 32; ++j) d += v[j];
        }
        sacc1[grp][slot] = d;
    }
    __syncthreads();
    if (threadIdx.x < 64) {                                            // wave 0 (wave-uniform)
        double d = 0.0;
        if ((int)threadIdx.x < nacc) {
            if (striped) { for (int g = 0; g < kGroups; ++g) d += sacc1[g][threadIdx.x]; }
            else d = sacc1[0][threadIdx.x];
        }
        const int mine = __builtin_bit_cast(int, (float)d);            // 0.0f in the lanes beyond nacc
#pragma unroll
        for (int i = 0; i < kNAccMax; ++i) acc[i] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(mine, i));
    }
}
__device__ inline void reduce_image_partials(const float* image_partials, int nchunks, int nacc, float (&acc)[kNAccMax]) {
    reduce_image_partials(image_partials, nchunks, nacc, acc, [] {});
}

#pragma clang fp contract(fast)

}  // namespace dev
}  // namespace gclm
