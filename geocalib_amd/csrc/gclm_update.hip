// gclm_update.hip -- per-image / per-group kernels around the sweep: reduction of the workgroup
// partials, lambda rule, damped Cholesky, manifold update, parameter blocks, uncertainty, and the
// synthetic-field generator used for measurement.  All tiny (O(B) threads); the reference does the
// same work with a device->host->device round trip per step (lm_optimizer.py:128-137).
#include "gclm_internal.h"

namespace gclm {

namespace {

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
__device__ inline void update_dist(State& s, float delta, float lo, float hi) {
    s.k1 = fminf(fmaxf(s.k1 + delta, lo), hi);
    s.k2 = fminf(fmaxf(s.k2 + delta, lo), hi);
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
}

// ---------------------------------------------------------------- small dense algebra

// In-place Cholesky solve of an n x n SPD system (fp32 like torch.linalg.cholesky on fp32).
template <int MAXN>
__device__ inline bool chol_solve(int n, float (&A)[MAXN][MAXN], float (&b)[MAXN]) {
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

// Symmetric 4x4 system out of an accumulator record (full column set d1,d2,f,k1).
__device__ inline void unpack_system(const float* acc, float (&Hm)[4][4], float (&G)[4]) {
    const float* h = acc + A_H00;
    Hm[0][0] = h[0]; Hm[0][1] = h[1]; Hm[0][2] = h[2]; Hm[0][3] = h[3];
    Hm[1][1] = h[4]; Hm[1][2] = h[5]; Hm[1][3] = h[6];
    Hm[2][2] = h[7]; Hm[2][3] = h[8]; Hm[3][3] = h[9];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        G[i] = acc[A_G0 + i];
        for (int j = 0; j < i; ++j) Hm[i][j] = Hm[j][i];
    }
}

// Column plan of calculate_gradient_and_hessian (lm_optimizer.py:335-344)
struct Plan {
    int n, cols[4];
    int focal_dim, dist_dim;   // lm_optimizer.py:223-235 (python indices into delta)
};
__device__ inline Plan make_plan(const gclm_config& cfg) {
    Plan p;
    p.n = 0;
    const bool has_dist = cfg.camera_model != GCLM_PINHOLE;
    if (cfg.estimate_gravity) { p.cols[p.n++] = 0; p.cols[p.n++] = 1; }
    if (cfg.estimate_focal) p.cols[p.n++] = 2;
    if (has_dist) p.cols[p.n++] = 3;
    p.focal_dim = cfg.estimate_focal ? (cfg.estimate_gravity ? 2 : 0) : -1;
    p.dist_dim = p.focal_dim + 1;          // reproduces the prior_focal + distortion overlap (quirk)
    return p;
}

// Sum the workgroup partials of the images of this block in a fixed order (double accumulate).
// Block-cooperative: 256 threads = 16 images x 16 accumulator slots; thread (image, slot) walks the
// image's chunk records (64-byte rows: coalesced over the 16 slots), the per-image leader (slot 0)
// then owns the 16 sums.  Returns true for leaders.  Must be called by every thread of the block.
constexpr int kImgPerBlock = 16;
__device__ inline bool coop_reduce_partials(const float* partials, int B, int nchunks, int& b, float (&acc)[kNAcc]) {
    __shared__ float sacc[kImgPerBlock][kNAcc + 1];
    const int li = threadIdx.x / kNAcc, slot = threadIdx.x % kNAcc;
    b = blockIdx.x * kImgPerBlock + li;
    if (b < B) {
        double d = 0.0;
        const float* p = partials + (size_t)b * nchunks * kNAcc + slot;
        for (int c = 0; c < nchunks; ++c) d += p[(size_t)c * kNAcc];
        sacc[li][slot] = (float)d;
    }
    __syncthreads();
    if (b >= B || slot != 0) return false;
#pragma unroll
    for (int i = 0; i < kNAcc; ++i) acc[i] = sacc[li][i];
    return true;
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

// ---------------------------------------------------------------- kernels

__global__ void init_kernel(SolveCtx c, const float* cam, const float* grav) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b == 0) {
        c.ctrl->stopped = 0;
        c.ctrl->final_sel = c.cfg.num_steps & 1;
        for (int i = 0; i < GCLM_MAX_STEPS + 4; ++i) c.ctrl->notclose[i] = 0;
    }
    if (b >= c.B) return;
    State s;
    const float* cm = cam + (size_t)b * GCLM_CAM_STRIDE;
    s.w = cm[0]; s.h = cm[1]; s.fx = cm[2]; s.fy = cm[3]; s.cx = cm[4]; s.cy = cm[5]; s.k1 = cm[6]; s.k2 = cm[7];
    const V3 g = normalize3({grav[b * 3], grav[b * 3 + 1], grav[b * 3 + 2]});
    s.gx = g.x; s.gy = g.y; s.gz = g.z;
    s.lambda = c.cfg.lambda0; s.prev_cost = 0.f; s.fails = 0.f; s.init_cu = s.init_cl = 0.f;
    c.state[0][b] = s;
    PBlock p;
    build_pblock(s, c.cfg.use_spherical_manifold != 0, c.cfg.use_log_focal != 0, p);
    c.pb[0][b] = p;
}

// Independent intrinsics: one thread per image does reduce -> lambda -> damped solve -> update.
__global__ __launch_bounds__(kImgPerBlock * kNAcc) void update_kernel(SolveCtx c, int step) {
    if (c.cfg.early_stop && c.ctrl->stopped) return;          // block-uniform
    int b;
    float acc[kNAcc];
    if (!coop_reduce_partials(c.partials, c.B, c.nchunks, b, acc)) return;
    const gclm_config& cfg = c.cfg;
    State s = c.state[step & 1][b];
    const float invN = 1.0f / (float)((size_t)c.H * c.W);
    float cu, cl;
    const float total = total_cost(acc, invN, true, cu, cl);   // A_CU is 0 without an up field
    if (step == 0) { s.init_cu = cu; s.init_cl = cl; }     // infos["initial_*"] (:585-588)
    cost_bookkeeping(cfg, c.ctrl, step, total, s, !cfg.fix_lambda);

    // damped normal equations over the estimated columns (lm_optimizer.py:109-137)
    float Hf[4][4], Gf[4];
    unpack_system(acc, Hf, Gf);
    const Plan pl = make_plan(cfg);
    float A[4][4], d[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < pl.n; ++i) {
        d[i] = Gf[pl.cols[i]];
        for (int j = 0; j < pl.n; ++j) A[i][j] = Hf[pl.cols[i]][pl.cols[j]];
    }
    for (int i = 0; i < pl.n; ++i) A[i][i] += fmaxf(A[i][i] * s.lambda, 1e-6f);
    if (!chol_solve<4>(pl.n, A, d)) {
        d[0] = d[1] = d[2] = d[3] = 0.f;     // zero step for THIS image (reference: whole batch)
        s.fails += 1.f;
    }
    // update_estimate (lm_optimizer.py:518-549)
    const float d0 = cfg.estimate_gravity ? d[0] : 0.f, d1 = cfg.estimate_gravity ? d[1] : 0.f;
    const V3 g = grav_update({s.gx, s.gy, s.gz}, d0, d1, cfg.use_spherical_manifold != 0);
    s.gx = g.x; s.gy = g.y; s.gz = g.z;
    update_focal(s, cfg.estimate_focal ? d[pl.focal_dim] : 0.f, cfg.use_log_focal != 0);
    if (cfg.camera_model != GCLM_PINHOLE && cfg.estimate_dist) update_dist(s, d[pl.dist_dim], -0.7f, 0.7f);

    c.state[(step + 1) & 1][b] = s;
    PBlock p;
    build_pblock(s, cfg.use_spherical_manifold != 0, cfg.use_log_focal != 0, p);
    c.pb[(step + 1) & 1][b] = p;
}

// Batch-global early stop (lm_optimizer.py:619-625) decided on the device: after update `step`
// every image has compared cost(theta_step) with the previous one.
__global__ void decide_kernel(SolveCtx c, int step) {
    if (c.ctrl->stopped || step < 1) return;
    if (c.ctrl->notclose[step] == 0) {
        c.ctrl->stopped = 1;
        c.ctrl->final_sel = step & 1;      // theta_step: the tentative theta_{step+1} is discarded
    }
}

// Parameter block of the final sweep: (roll, pitch, focal) parametrisation (lm_optimizer.py:481-483).
__global__ void prep_final_kernel(SolveCtx c) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= c.B) return;
    const State s = c.state[c.ctrl->final_sel][b];
    PBlock p;
    build_pblock(s, false, false, p);
    c.pb_final[b] = p;
}

// Inverse of an n x n matrix (n <= 4) by Gauss-Jordan with partial pivoting, in double.
__device__ inline void invert(int n, const float (&A)[4][4], double (&inv)[4][4]) {
    double M[4][8];
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) { M[i][j] = A[i][j]; M[i][n + j] = i == j ? 1.0 : 0.0; }
    for (int col = 0; col < n; ++col) {
        int piv = col;
        for (int r = col + 1; r < n; ++r) if (fabs(M[r][col]) > fabs(M[piv][col])) piv = r;
        if (piv != col) for (int j = 0; j < 2 * n; ++j) { const double t = M[col][j]; M[col][j] = M[piv][j]; M[piv][j] = t; }
        const double dv = M[col][col];
        for (int j = 0; j < 2 * n; ++j) M[col][j] /= dv;
        for (int r = 0; r < n; ++r) if (r != col) {
            const double f = M[r][col];
            for (int j = 0; j < 2 * n; ++j) M[r][j] -= f * M[col][j];
        }
    }
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) inv[i][j] = M[i][n + j];
}

// Final costs + estimate_uncertainty (lm_optimizer.py:632-642, 463-516) from the final sweep.
__global__ __launch_bounds__(kImgPerBlock * kNAcc) void finalize_kernel(SolveCtx c, float* cam, float* grav, float* info) {
    int b;
    float acc[kNAcc];
    if (!coop_reduce_partials(c.partials, c.B, c.nchunks, b, acc)) return;
    const gclm_config& cfg = c.cfg;
    const int sel = c.ctrl->final_sel;
    State s = c.state[sel][b];
    const float invN = 1.0f / (float)((size_t)c.H * c.W);
    float cu, cl;
    const float total = total_cost(acc, invN, true, cu, cl);
    // the final sweep is also the "new cost" evaluation of the last executed step
    if (!c.ctrl->stopped)
        cost_bookkeeping(cfg, c.ctrl, cfg.num_steps, total, s, !cfg.fix_lambda && !cfg.shared_intrinsics);
    float* o = info + (size_t)b * GCLM_INFO_STRIDE;
    o[GCLM_INFO_INITIAL_UP_COST] = s.init_cu;
    o[GCLM_INFO_INITIAL_LAT_COST] = s.init_cl;
    o[GCLM_INFO_INITIAL_COST] = s.init_cu + s.init_cl;
    o[GCLM_INFO_FINAL_UP_COST] = cu;
    o[GCLM_INFO_FINAL_LAT_COST] = cl;
    o[GCLM_INFO_FINAL_COST] = total;
    const Plan pl = make_plan(cfg);
    o[GCLM_INFO_NPARAMS] = (float)pl.n;
    o[GCLM_INFO_LAMBDA] = s.lambda;
    o[GCLM_INFO_STEP_FAILURES] = s.fails;
    for (int i = GCLM_INFO_ROLL_UNC; i <= GCLM_INFO_VFOV_UNC; ++i) o[i] = 0.f;
    if (cfg.compute_uncertainty) {
        float Hf[4][4], Gf[4], A[4][4];
        unpack_system(acc, Hf, Gf);
        for (int i = 0; i < pl.n; ++i)
            for (int j = 0; j < pl.n; ++j) A[i][j] = Hf[pl.cols[i]][pl.cols[j]];
        double Cov[4][4];
        invert(pl.n, A, Cov);                                 // torch.inverse(Hess), :484
        for (int i = 0; i < pl.n; ++i)
            for (int j = 0; j < pl.n; ++j) o[GCLM_INFO_COV + i * pl.n + j] = (float)Cov[i][j];
        if (cfg.estimate_gravity) {
            const double c00 = Cov[0][0], c11 = Cov[1][1], c01 = 0.5 * (Cov[0][1] + Cov[1][0]);
            o[GCLM_INFO_ROLL_UNC] = (float)sqrt(c00);
            o[GCLM_INFO_PITCH_UNC] = (float)sqrt(c11);
            const double tr = 0.5 * (c00 + c11), df = 0.5 * (c00 - c11);
            o[GCLM_INFO_GRAVITY_UNC] = (float)sqrt(tr + sqrt(df * df + c01 * c01));   // max eigvalsh, :495-496
        }
        if (cfg.estimate_focal) {
            const double fu = Cov[pl.focal_dim][pl.focal_dim];
            const double fy = s.fy, hh = s.h;
            const double Jf = -4.0 * hh / (4.0 * fy * fy + hh * hh);                   // misc.py:285-287
            o[GCLM_INFO_FOCAL_UNC] = (float)(sqrt(fu) * 0.5);
            o[GCLM_INFO_VFOV_UNC] = (float)sqrt(Jf * Jf * fu * 0.5);
        }
    }
    float* cm = cam + (size_t)b * GCLM_CAM_STRIDE;
    cm[0] = s.w; cm[1] = s.h; cm[2] = s.fx; cm[3] = s.fy; cm[4] = s.cx; cm[5] = s.cy; cm[6] = s.k1; cm[7] = s.k2;
    grav[b * 3] = s.gx; grav[b * 3 + 1] = s.gy; grav[b * 3 + 2] = s.gz;
}

// stop_at (lm_optimizer.py:575,620,638): first step after which EVERY image's cost was "close".
__global__ void stop_at_kernel(SolveCtx c, float* info) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= c.B) return;
    int stop_at = c.cfg.num_steps;
    for (int j = 1; j <= c.cfg.num_steps; ++j)
        if (c.ctrl->notclose[j] == 0) { stop_at = j; break; }
    info[(size_t)b * GCLM_INFO_STRIDE + GCLM_INFO_STOP_AT] = (float)stop_at;
}

// ---------------------------------------------------------------- shared intrinsics
// Whole group = one arrow-head system (lm_optimizer.py:350-383): 2x2 gravity blocks D_i on the
// diagonal, couplings E_i (2 x ni) to the ni shared intrinsics, C = sum H_ii.  Instead of the
// reference's dense (2B+ni)^2 Cholesky the Schur complement on the intrinsics is formed:
//   S = C~ - sum E_i^T D~_i^-1 E_i,  rhs = c - sum E_i^T D~_i^-1 g_i,  delta_I = S^-1 rhs,
//   delta_g,i = D~_i^-1 (g_i - E_i delta_I)         (~ = with LM damping on the diagonal)
// which is algebraically the same solve and reduces over frames with a plain SUM -- i.e. it can
// be all-reduced across devices when a group's frames are sharded (BASELINE config 5).

// per frame: reduce partials -> frame_sys, costs / allclose bookkeeping
__global__ __launch_bounds__(kImgPerBlock * kNAcc) void shared_frame_kernel(SolveCtx c, int step) {
    if (c.cfg.early_stop && c.ctrl->stopped) return;
    int b;
    float acc[kNAcc];
    if (!coop_reduce_partials(c.partials, c.B, c.nchunks, b, acc)) return;
    State s = c.state[step & 1][b];
    const float invN = 1.0f / (float)((size_t)c.H * c.W);
    float cu, cl;
    const float total = total_cost(acc, invN, true, cu, cl);
    if (step == 0) { s.init_cu = cu; s.init_cl = cl; }     // infos["initial_*"] (:585-588)
    cost_bookkeeping(c.cfg, c.ctrl, step, total, s, false);   // lambda is never updated (:612)
    c.state[step & 1][b] = s;
    float4* out = reinterpret_cast<float4*>(c.frame_sys + (size_t)b * kNAcc);
#pragma unroll
    for (int q = 0; q < kNAcc / 4; ++q) out[q] = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
}

__device__ inline int lower_bound(const int32_t* a, int n, int key) {
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (a[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}
__device__ inline void group_range(const SolveCtx& c, int g, int& f0, int& f1) {
    if (c.group_of_frame) { f0 = lower_bound(c.group_of_frame, c.B, g); f1 = lower_bound(c.group_of_frame, c.B, g + 1); }
    else { f0 = g * c.group_size; f1 = min(f0 + c.group_size, c.B); }
}

// Damped 2x2 gravity block of a frame and its inverse; returns false if not positive definite.
__device__ inline bool frame_block(const float* fs, float lambda, float (&Dinv)[2][2]) {
    const float* h = fs + A_H00;
    const float a = h[0] + fmaxf(h[0] * lambda, 1e-6f), b = h[1], d = h[4] + fmaxf(h[4] * lambda, 1e-6f);
    const float det = a * d - b * b;
    if (!(a > 0.f) || !(det > 0.f)) return false;
    const float id = 1.0f / det;
    Dinv[0][0] = d * id; Dinv[0][1] = -b * id; Dinv[1][0] = -b * id; Dinv[1][1] = a * id;
    return true;
}

// per group: local Schur partials over this device's frames of the group
//   layout (GCLM_SHARED_PARTIAL_STRIDE floats): [0..3] sum E^T Dinv E, [4..5] sum E^T Dinv g,
//   [6..9] sum H_ii, [10..11] sum g_i, [12] #frames; NaN in [0] marks a non-PD frame block.
__global__ void shared_group_kernel(SolveCtx c, int step, float* gp) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= c.n_groups) return;
    if (c.cfg.early_stop && c.ctrl->stopped) return;
    const int ni = c.cfg.camera_model == GCLM_PINHOLE ? 1 : 2;
    int f0, f1;
    group_range(c, g, f0, f1);
    float S[2][2] = {{0.f, 0.f}, {0.f, 0.f}}, r[2] = {0.f, 0.f}, C[2][2] = {{0.f, 0.f}, {0.f, 0.f}}, cg[2] = {0.f, 0.f};
    bool ok = true;
    for (int f = f0; f < f1; ++f) {
        const float* fs = c.frame_sys + (size_t)f * kNAcc;
        float Hf[4][4], Gf[4], Dinv[2][2];
        unpack_system(fs, Hf, Gf);
        ok = frame_block(fs, c.state[step & 1][f].lambda, Dinv) && ok;
        for (int i = 0; i < ni; ++i) {
            // Dinv E[:, i]
            const float e0 = Hf[0][2 + i], e1 = Hf[1][2 + i];
            const float t0 = Dinv[0][0] * e0 + Dinv[0][1] * e1, t1 = Dinv[1][0] * e0 + Dinv[1][1] * e1;
            for (int j = 0; j < ni; ++j) S[j][i] += Hf[0][2 + j] * t0 + Hf[1][2 + j] * t1;
            const float q0 = Dinv[0][0] * Gf[0] + Dinv[0][1] * Gf[1], q1 = Dinv[1][0] * Gf[0] + Dinv[1][1] * Gf[1];
            r[i] += e0 * q0 + e1 * q1;
            cg[i] += Gf[2 + i];
            for (int j = 0; j < ni; ++j) C[i][j] += Hf[2 + i][2 + j];
        }
    }
    float* o = gp + (size_t)g * GCLM_SHARED_PARTIAL_STRIDE;
    o[0] = ok ? S[0][0] : __builtin_nanf("");
    o[1] = S[0][1]; o[2] = S[1][0]; o[3] = S[1][1];
    o[4] = r[0]; o[5] = r[1];
    o[6] = C[0][0]; o[7] = C[0][1]; o[8] = C[1][0]; o[9] = C[1][1];
    o[10] = cg[0]; o[11] = cg[1];
    o[12] = (float)(f1 - f0);
    o[13] = o[14] = o[15] = 0.f;
}

// per frame: solve the (tiny) Schur system of its group from the REDUCED partials (redundantly per
// frame: ni <= 2), back-substitute its own gravity block, update (lm_optimizer.py:597-606).
__global__ void shared_apply_kernel(SolveCtx c, int step, const float* gp) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= c.B) return;
    if (c.cfg.early_stop && c.ctrl->stopped) return;
    const gclm_config& cfg = c.cfg;
    const int ni = cfg.camera_model == GCLM_PINHOLE ? 1 : 2;
    const int g = c.group_of_frame ? c.group_of_frame[b] : b / c.group_size;
    State s = c.state[step & 1][b];
    const float* o = gp + (size_t)g * GCLM_SHARED_PARTIAL_STRIDE;
    float A[2][2], dI[2] = {0.f, 0.f}, dG[2] = {0.f, 0.f};
    bool ok = o[0] == o[0];
    for (int i = 0; i < ni; ++i) {
        dI[i] = o[10 + i] - o[4 + i];
        for (int j = 0; j < ni; ++j) A[i][j] = o[6 + 2 * i + j] - o[2 * i + j];
        A[i][i] += fmaxf(o[6 + 2 * i + i] * s.lambda, 1e-6f);      // damping on sum H_ii (:123-126)
    }
    ok = ok && chol_solve<2>(ni, A, dI);
    const float* fs = c.frame_sys + (size_t)b * kNAcc;
    float Hf[4][4], Gf[4], Dinv[2][2];
    unpack_system(fs, Hf, Gf);
    ok = ok && frame_block(fs, s.lambda, Dinv);
    if (ok) {
        float r0 = Gf[0], r1 = Gf[1];
        for (int i = 0; i < ni; ++i) { r0 -= Hf[0][2 + i] * dI[i]; r1 -= Hf[1][2 + i] * dI[i]; }
        dG[0] = Dinv[0][0] * r0 + Dinv[0][1] * r1;
        dG[1] = Dinv[1][0] * r0 + Dinv[1][1] * r1;
    } else {
        dI[0] = dI[1] = 0.f;
        s.fails += 1.f;
    }
    const V3 gv = grav_update({s.gx, s.gy, s.gz}, dG[0], dG[1], cfg.use_spherical_manifold != 0);
    s.gx = gv.x; s.gy = gv.y; s.gz = gv.z;
    update_focal(s, dI[0], cfg.use_log_focal != 0);
    if (ni == 2) update_dist(s, dI[1], -0.7f, 0.7f);
    c.state[(step + 1) & 1][b] = s;
    PBlock p;
    build_pblock(s, cfg.use_spherical_manifold != 0, cfg.use_log_focal != 0, p);
    c.pb[(step + 1) & 1][b] = p;
}

// ---------------------------------------------------------------- gclm_system() helpers

__global__ void pblock_from_params_kernel(SolveCtx c, const float* cam, const float* grav, int as_rpf, PBlock* out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= c.B) return;
    State s;
    const float* cm = cam + (size_t)b * GCLM_CAM_STRIDE;
    s.w = cm[0]; s.h = cm[1]; s.fx = cm[2]; s.fy = cm[3]; s.cx = cm[4]; s.cy = cm[5]; s.k1 = cm[6]; s.k2 = cm[7];
    const V3 g = normalize3({grav[b * 3], grav[b * 3 + 1], grav[b * 3 + 2]});
    s.gx = g.x; s.gy = g.y; s.gz = g.z;
    PBlock p;
    build_pblock(s, c.cfg.use_spherical_manifold && !as_rpf, c.cfg.use_log_focal && !as_rpf, p);
    out[b] = p;
}

__global__ __launch_bounds__(kImgPerBlock * kNAcc) void system_out_kernel(SolveCtx c, float* cost, float* grad, float* hess) {
    int b;
    float acc[kNAcc];
    if (!coop_reduce_partials(c.partials, c.B, c.nchunks, b, acc)) return;
    const float invN = 1.0f / (float)((size_t)c.H * c.W);
    cost[b * 2] = acc[A_CU] * invN;
    cost[b * 2 + 1] = acc[A_CL] * invN;
    float Hf[4][4], Gf[4];
    unpack_system(acc, Hf, Gf);
    for (int i = 0; i < GCLM_MAX_PARAMS; ++i) {
        grad[b * GCLM_MAX_PARAMS + i] = i < 4 ? Gf[i] : 0.f;
        for (int j = 0; j < GCLM_MAX_PARAMS; ++j)
            hess[(b * GCLM_MAX_PARAMS + i) * GCLM_MAX_PARAMS + j] = (i < 4 && j < 4) ? Hf[i][j] : 0.f;
    }
}

// ---------------------------------------------------------------- synthetic fields (measurement)

__device__ inline uint64_t mix64(uint64_t z) {   // splitmix64 finaliser
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ inline float u01(uint64_t h) { return ((float)(h >> 40) + 0.5f) * (1.0f / 16777216.0f); }

struct GT { float fx, k1; V3 g; };
// gravity is keyed by the image index, the intrinsics by `intr_index` (= image index, or the
// group index when frames of a group share one camera)
__device__ inline GT synth_gt(int model, uint64_t seed, int64_t index, int64_t intr_index, int H) {
    const uint64_t base = mix64(seed ^ mix64((uint64_t)index * 0xD1342543DE82EF95ull + 1));
    const uint64_t ibase = mix64(seed ^ mix64((uint64_t)intr_index * 0xD1342543DE82EF95ull + 1));
    const float d2r = kPi / 180.f;
    const float roll = (u01(mix64(base + 1)) * 90.f - 45.f) * d2r;
    const float pitch = (u01(mix64(base + 2)) * 90.f - 45.f) * d2r;
    const float vfov = (20.f + u01(mix64(ibase + 3)) * 70.f) * d2r;
    GT t;
    t.fx = (float)H * 0.5f / tanf(vfov * 0.5f);
    t.k1 = model == GCLM_PINHOLE ? 0.f : -0.3f + 0.4f * u01(mix64(ibase + 4));
    t.g = from_rp(roll, pitch);
    return t;
}

// One thread per pixel: ground-truth perspective field (perspective_fields.py:278) + Gaussian
// noise, up re-normalised, latitude clamped, confidences ~ U(0,1)  (SURVEY.md 8d).
__global__ void synth_kernel(int model, uint64_t seed, int64_t first, int B, int H, int W, float sigma,
                             int group_size, int run, int run_stride,
                             float* up, float* lat, float* upc, float* latc, float* gt_cam, float* gt_grav) {
    const int b = blockIdx.y;
    const size_t N = (size_t)H * W;
    // global image index of local image b: contiguous, or runs of `run` images every `run_stride`
    const int64_t gidx = first + (run > 0 ? (int64_t)(b / run) * run_stride + (b % run) : b);
    const GT t = synth_gt(model, seed, gidx, group_size > 1 ? gidx / group_size : gidx, H);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (gt_cam) {
            float* cm = gt_cam + (size_t)b * 8;
            cm[0] = (float)W; cm[1] = (float)H; cm[2] = t.fx; cm[3] = t.fx; cm[4] = W * 0.5f; cm[5] = H * 0.5f;
            cm[6] = t.k1; cm[7] = 0.f;
        }
        if (gt_grav) { gt_grav[b * 3] = t.g.x; gt_grav[b * 3 + 1] = t.g.y; gt_grav[b * 3 + 2] = t.g.z; }
    }
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (size_t)gridDim.x * blockDim.x) {
        const int y = (int)(i / W), x = (int)(i - (size_t)y * W);
        const float u = ((float)x - W * 0.5f) / t.fx, v = ((float)y - H * 0.5f) / t.fx;
        const float r2 = u * u + v * v;
        const float px = t.g.x - t.g.z * u, py = t.g.y - t.g.z * v;
        const float d = 1.f + t.k1 * r2, tt = u * px + v * py;
        float qx = d * px + 2.f * t.k1 * tt * u, qy = d * py + 2.f * t.k1 * tt * v;
        const float e = 1.f - t.k1 * r2;
        const float Px = e * u, Py = e * v;
        const float rn = rsqrtf(Px * Px + Py * Py + 1.f);
        float s = (Px * t.g.x + Py * t.g.y + t.g.z) * rn;
        s = fminf(fmaxf(s, -1.f + 1e-6f), 1.f - 1e-6f);
        const uint64_t h = mix64(mix64(seed ^ 0xA5A5A5A5ull) + (uint64_t)gidx * 0x9E3779B97F4A7C15ull + i * 4);
        // Box-Muller, two pairs
        const float a1 = sqrtf(-2.f * logf(u01(mix64(h + 1)))), p1 = 2.f * kPi * u01(mix64(h + 2));
        const float a2 = sqrtf(-2.f * logf(u01(mix64(h + 3)))), p2 = 2.f * kPi * u01(mix64(h + 4));
        const float qn = rsqrtf(fmaxf(qx * qx + qy * qy, 1e-24f));
        qx = qx * qn + sigma * a1 * cosf(p1);
        qy = qy * qn + sigma * a1 * sinf(p1);
        const float qn2 = rsqrtf(fmaxf(qx * qx + qy * qy, 1e-24f));
        float l = asinf(s) + sigma * a2 * cosf(p2);
        const float lim = kPi * 0.5f - 1e-3f;
        l = fminf(fmaxf(l, -lim), lim);
        up[(size_t)b * 2 * N + i] = qx * qn2;
        up[(size_t)b * 2 * N + N + i] = qy * qn2;
        lat[(size_t)b * N + i] = l;
        if (upc) upc[(size_t)b * N + i] = u01(mix64(h + 5));
        if (latc) latc[(size_t)b * N + i] = u01(mix64(h + 6));
    }
}

inline dim3 grid1(int n) { return dim3((n + 127) / 128); }

}  // namespace

#define GCLM_L(kernel, n, s, ...) hipLaunchKernelGGL(kernel, grid1(n), dim3(128), 0, s, __VA_ARGS__)
// cooperative-reduce kernels: 16 images per 256-thread block
#define GCLM_LR(kernel, n, s, ...) \
    hipLaunchKernelGGL(kernel, dim3(((n) + kImgPerBlock - 1) / kImgPerBlock), dim3(kImgPerBlock * kNAcc), 0, s, __VA_ARGS__)

hipError_t launch_init(const SolveCtx& c, const float* d_cam, const float* d_grav, hipStream_t s) {
    GCLM_L(init_kernel, c.B, s, c, d_cam, d_grav);
    return hipGetLastError();
}
hipError_t launch_update(const SolveCtx& c, int step, hipStream_t s) {
    GCLM_LR(update_kernel, c.B, s, c, step);
    return hipGetLastError();
}
hipError_t launch_decide(const SolveCtx& c, int step, hipStream_t s) {
    hipLaunchKernelGGL(decide_kernel, dim3(1), dim3(1), 0, s, c, step);
    return hipGetLastError();
}
hipError_t launch_prep_final(const SolveCtx& c, hipStream_t s) {
    GCLM_L(prep_final_kernel, c.B, s, c);
    return hipGetLastError();
}
hipError_t launch_finalize(const SolveCtx& c, float* d_cam, float* d_grav, float* d_info, hipStream_t s) {
    GCLM_LR(finalize_kernel, c.B, s, c, d_cam, d_grav, d_info);
    GCLM_L(stop_at_kernel, c.B, s, c, d_info);
    return hipGetLastError();
}
hipError_t launch_shared_reduce(const SolveCtx& c, int step, float* d_group_partials, hipStream_t s) {
    if (c.B > 0) GCLM_LR(shared_frame_kernel, c.B, s, c, step);
    GCLM_L(shared_group_kernel, c.n_groups, s, c, step, d_group_partials);
    return hipGetLastError();
}
hipError_t launch_shared_apply(const SolveCtx& c, int step, const float* d_group_partials, hipStream_t s) {
    GCLM_L(shared_apply_kernel, c.B, s, c, step, d_group_partials);
    return hipGetLastError();
}
hipError_t launch_system_out(const SolveCtx& c, float* d_cost, float* d_grad, float* d_hess, hipStream_t s) {
    GCLM_LR(system_out_kernel, c.B, s, c, d_cost, d_grad, d_hess);
    return hipGetLastError();
}
hipError_t launch_pblock_from_params(const SolveCtx& c, const float* d_cam, const float* d_grav, int as_rpf,
                                     PBlock* out, hipStream_t s) {
    GCLM_L(pblock_from_params_kernel, c.B, s, c, d_cam, d_grav, as_rpf, out);
    return hipGetLastError();
}
hipError_t launch_synth(int camera_model, uint64_t seed, int64_t first_index, int B, int H, int W, float sigma,
                        int group_size, int run, int run_stride, float* up, float* lat, float* upc, float* latc, float* gt_cam, float* gt_grav,
                        hipStream_t s) {
    if (B <= 0) return hipSuccess;
    const size_t N = (size_t)H * W;
    const int bx = (int)((N + 255) / 256 < 64 ? (N + 255) / 256 : 64);
    hipLaunchKernelGGL(synth_kernel, dim3(bx, B), dim3(256), 0, s, camera_model, seed, first_index, B, H, W,
                       sigma, group_size, run, run_stride, up, lat, upc, latc, gt_cam, gt_grav);
    return hipGetLastError();
}

}  // namespace gclm
