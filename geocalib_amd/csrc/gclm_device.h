// gclm_device.h -- per-image device logic shared by the update kernels (gclm_update.hip) and the fused
// "last workgroup of an image applies the LM step" tail of the sweep (gclm_pass.hip).  gfx950 only.
#pragma once
#include "gclm_internal.h"

namespace gclm {
namespace dev {

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

// Gravity.update (gravity.py:112-119) / SphericalManifold.plus (misc.py:234-259)
__device__ inline V3 grav_update(V3 g, float d0, float d1, bool spherical) {
    if (!spherical) return from_rp(grav_roll(g) + d0, grav_pitch(g) + d1);
    const float eps = 1e-7f;
    const float nx = sqrtf(g.x * g.x + g.y * g.y + g.z * g.z);
    const float nd = sqrtf(d0 * d0 + d1 * d1);
    const float nd_ = nd < eps ? nd + eps : nd;
    const float sinc = nd < eps ? 1.f : sinf(nd_) / nd_;
    const float e[3] = {sinc * d0, sinc * d1, cosf(nd)};
    float v[3], beta;
    householder(g, v, beta);
    const float bd = beta * (v[0] * e[0] + v[1] * e[1] + v[2] * e[2]);
    return normalize3({nx * (e[0] - v[0] * bd), nx * (e[1] - v[1] * bd), nx * (e[2] - v[2] * bd)});
}

// BaseCamera.update_focal (camera.py:136-152): clamp to fov in [5, 150] deg of the image HEIGHT,
// fx rebuilt from fy by the old ratio.
__device__ inline void update_focal(State& s, float delta, bool as_log) {
    const float fy = as_log ? expf(logf(s.fy) + delta) : s.fy + delta;
    const float min_f = s.h * 0.5f / tanf((150.f / 180.f * kPi) * 0.5f);
    const float max_f = s.h * 0.5f / tanf((5.f / 180.f * kPi) * 0.5f);
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

__device__ inline void build_pblock(const State& s, bool spherical, bool log_focal, PBlock& p) {
    const V3 g = {s.gx, s.gy, s.gz};
    float T[3][2];
    if (spherical) tangent_sphere(g, T); else tangent_rp(g, T);
    p.ifx = 1.0f / s.fx; p.ify = 1.0f / s.fy; p.cx = s.cx; p.cy = s.cy;
    p.ga = g.x; p.gb = g.y; p.gc = g.z; p.k1 = s.k1;
    p.T00 = T[0][0]; p.T01 = T[0][1]; p.T10 = T[1][0]; p.T11 = T[1][1]; p.T20 = T[2][0]; p.T21 = T[2][1];
    p.wfx = log_focal ? 1.0f : 1.0f / s.fx;
    p.wfy = log_focal ? 1.0f : 1.0f / s.fy;
    p.k2 = s.k2; p.pad0 = p.pad1 = p.pad2 = 0.f;
}

// ---------------------------------------------------------------- small dense algebra

// In-place Cholesky solve of an n x n SPD system (fp32 like torch.linalg.cholesky on fp32).
template <int MAXN>
__device__ inline bool chol_solve(int n, float (&A)[MAXN][MAXN], float* b) {
    for (int j = 0; j < n; ++j) {
        float s = A[j][j];
        for (int k = 0; k < j; ++k) s -= A[j][k] * A[j][k];
        if (!(s > 0.f)) return false;
        const float l = sqrtf(s);
        A[j][j] = l;
        for (int i = j + 1; i < n; ++i) {
            float t = A[i][j];
            for (int k = 0; k < j; ++k) t -= A[i][k] * A[j][k];
            A[i][j] = t / l;
        }
    }
    for (int i = 0; i < n; ++i) {
        float t = b[i];
        for (int k = 0; k < i; ++k) t -= A[i][k] * b[k];
        b[i] = t / A[i][i];
    }
    for (int i = n - 1; i >= 0; --i) {
        float t = b[i];
        for (int k = i + 1; k < n; ++k) t -= A[k][i] * b[k];
        b[i] = t / A[i][i];
    }
    return true;
}

// Symmetric PM x PM system (PM = 4 or 5 full columns d1,d2,f,k1[,k2]) out of an accumulator record.
__device__ inline void unpack_system(const float* acc, int pm, float (&Hm)[kMaxP][kMaxP], float (&G)[kMaxP]) {
    for (int i = 0; i < kMaxP; ++i) {
        G[i] = i < pm ? acc[A_G0 + i] : 0.f;
        for (int j = 0; j < kMaxP; ++j) Hm[i][j] = 0.f;
    }
    for (int i = 0; i < pm; ++i)
        for (int j = i; j < pm; ++j) Hm[i][j] = Hm[j][i] = acc[acc_h(pm, i, j)];
}

// Column plan of calculate_gradient_and_hessian (lm_optimizer.py:335-344)
struct Plan {
    int n, cols[kMaxP];
    int focal_dim, dist_dim;   // lm_optimizer.py:223-235 (python indices into delta)
};
__device__ inline Plan make_plan(const gclm_config& cfg) {
    Plan p;
    p.n = 0;
    if (cfg.estimate_gravity) { p.cols[p.n++] = 0; p.cols[p.n++] = 1; }
    if (cfg.estimate_focal) p.cols[p.n++] = 2;
    for (int k = 0; k < num_dist_params(cfg.camera_model); ++k) p.cols[p.n++] = 3 + k;
    p.focal_dim = cfg.estimate_focal ? (cfg.estimate_gravity ? 2 : 0) : -1;
    p.dist_dim = p.focal_dim + 1;          // reproduces the prior_focal + distortion overlap (quirk)
    return p;
}

// sum(c.mean(-1) for c in costs.values()) (lm_optimizer.py:584,610), float32
__device__ inline float total_cost(const float* acc, float invN, bool has_up, float& cu, float& cl) {
    cu = acc[A_CU] * invN;
    cl = acc[A_CL] * invN;
    return has_up ? cu + cl : cl;
}

// lambda rule + batch-global "allclose" bookkeeping shared by every update flavour
// (lm_optimizer.py:95-106, :90-92, :612-627).  Returns the new prev_cost.
__device__ inline void cost_bookkeeping(const gclm_config& cfg, Ctrl* ctrl, int step, float total,
                                        State& s, bool update_lambda) {
    if (step > 0) {
        if (update_lambda) {
            const float nl = s.lambda * (total > s.prev_cost ? 10.f : 0.1f);
            s.lambda = fminf(fmaxf(nl, 1e-6f), 1e2f);
        }
        const double diff = fabs((double)total - (double)s.prev_cost);
        const bool close = diff <= (double)cfg.atol + (double)cfg.rtol * fabs((double)s.prev_cost);
        if (!close) atomicAdd(&ctrl->notclose[step], 1);
    }
    s.prev_cost = total;
}

// One LM step of one image from its reduced accumulator record: lambda rule + allclose bookkeeping,
// damped normal equations over the estimated columns, manifold / focal / distortion update, next
// parameter block.  (update_kernel body; also run by the last workgroup of an image in the fused sweep.)
__device__ inline void update_image(const SolveCtx& c, int step, int b, const float (&acc)[kNAccMax]) {
    const gclm_config& cfg = c.cfg;
    State s = c.state[step & 1][b];
    const float invN = 1.0f / (float)((size_t)c.H * c.W);
    float cu, cl;
    const float total = total_cost(acc, invN, true, cu, cl);   // A_CU is 0 without an up field
    if (step == 0) { s.init_cu = cu; s.init_cl = cl; }     // infos["initial_*"] (:585-588)
    cost_bookkeeping(cfg, c.ctrl, step, total, s, !cfg.fix_lambda);

    // damped normal equations over the estimated columns (lm_optimizer.py:109-137)
    float Hf[kMaxP][kMaxP], Gf[kMaxP];
    unpack_system(acc, acc_pm(cfg.camera_model), Hf, Gf);
    const Plan pl = make_plan(cfg);
    float A[kMaxP][kMaxP], d[kMaxP + 1] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < pl.n; ++i) {
        d[i] = Gf[pl.cols[i]];
        for (int j = 0; j < pl.n; ++j) A[i][j] = Hf[pl.cols[i]][pl.cols[j]];
    }
    for (int i = 0; i < pl.n; ++i) A[i][i] += fmaxf(A[i][i] * s.lambda, 1e-6f);
    if (!chol_solve<kMaxP>(pl.n, A, d)) {
        for (int i = 0; i <= kMaxP; ++i) d[i] = 0.f;   // zero step for THIS image (reference: whole batch)
        s.fails += 1.f;
    }
    // update_estimate (lm_optimizer.py:518-549)
    const float d0 = cfg.estimate_gravity ? d[0] : 0.f, d1 = cfg.estimate_gravity ? d[1] : 0.f;
    const V3 g = grav_update({s.gx, s.gy, s.gz}, d0, d1, cfg.use_spherical_manifold != 0);
    s.gx = g.x; s.gy = g.y; s.gz = g.z;
    update_focal(s, cfg.estimate_focal ? d[pl.focal_dim] : 0.f, cfg.use_log_focal != 0);
    if (cfg.camera_model != GCLM_PINHOLE && cfg.estimate_dist)
        update_dist(s, cfg.camera_model, d[pl.dist_dim], d[pl.dist_dim + 1]);

    c.state[(step + 1) & 1][b] = s;
    PBlock p;
    build_pblock(s, cfg.use_spherical_manifold != 0, cfg.use_log_focal != 0, p);
    c.pb[(step + 1) & 1][b] = p;
}

}  // namespace dev
}  // namespace gclm
