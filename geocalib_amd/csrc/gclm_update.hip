// gclm_update.hip -- per-image / per-group kernels around the sweep: reduction of the workgroup
// partials, lambda rule, damped Cholesky, manifold update, parameter blocks, uncertainty, and the
// synthetic-field generator used for measurement.  All tiny (O(B) threads); the reference does the
// same work with a device->host->device round trip per step (lm_optimizer.py:128-137).
#include <type_traits>

#include "gclm_device.h"

namespace gclm {

namespace {

using namespace dev;

// Sum the workgroup partials of the images of this block in a fixed order (double accumulate).
// Block-cooperative, 256 threads = 8 groups x 32 slots (the first `nacc` = 16 or 24 are accumulator slots).
// Few chunks per image (large batches): a group is an image, thread (image, slot) walks that image's chunk
// records (coalesced over the slots).  Many chunks per image (small batches of large images, where one serial
// walk is a latency chain of hundreds of dependent loads): the block takes ONE image and the 8 groups are
// stripes of its chunks, combined in stripe order.  The per-image leader then owns the sums.  Returns true for
// leaders.  Must be called by every thread of the block; launch with reduce_blocks(B, nchunks) blocks.
__host__ __device__ inline int reduce_images_per_block(int nchunks) { return nchunks >= kStripeMinChunks ? 1 : kGroups; }
inline int reduce_blocks(int B, int nchunks) {
    const int ipb = reduce_images_per_block(nchunks);
    return (B + ipb - 1) / ipb;
}
__device__ inline bool coop_reduce_partials(const float* partials, int B, int nchunks, int nacc, int& b,
                                            float (&acc)[kNAccMax]) {
    __shared__ double sacc[kGroups][kSlots + 1];
    const int grp = threadIdx.x / kSlots, slot = threadIdx.x % kSlots;
    const bool striped = reduce_images_per_block(nchunks) == 1;
    b = striped ? (int)blockIdx.x : (int)blockIdx.x * kGroups + grp;
    const int first = striped ? grp : 0, stride = striped ? kGroups : 1;
    if (b < B && slot < nacc) {
        double d = 0.0;
        const float* p = partials + (size_t)b * nchunks * nacc + slot;
        // batches of 16 records: all loads of a batch are in flight together (one memory round trip for the 15 records
        // of a 640x480 image instead of four), summed in ascending chunk order as before (the padding adds exact zeros)
        for (int c0 = first; c0 < nchunks; c0 += 16 * stride) {
            float v[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int c = c0 + j * stride;
                const float t = p[(size_t)min(c, nchunks - 1) * nacc];      // unconditional load (a clamped index), then
                v[j] = c < nchunks ? t : 0.f;                                // a select: all loads of the batch in flight
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) d += v[j];
        }
        sacc[grp][slot] = d;
    }
    __syncthreads();
    if (b >= B || slot != 0 || (striped && grp != 0)) return false;
    if (striped) {
#pragma unroll
        for (int i = 0; i < kNAccMax; ++i) {
            double d = 0.0;
            if (i < nacc)
                for (int g = 0; g < kGroups; ++g) d += sacc[g][i];
            acc[i] = (float)d;
        }
    } else {
#pragma unroll
        for (int i = 0; i < kNAccMax; ++i) acc[i] = i < nacc ? (float)sacc[grp][i] : 0.f;
    }
    return true;
}

// ---------------------------------------------------------------- kernels

__global__ void init_kernel(SolveCtx c, InitArgs ia) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b == 0) {
        c.ctrl->stopped = 0;
        c.ctrl->final_sel = c.cfg.num_steps & 1;
        for (int i = 0; i < GCLM_MAX_STEPS + 4; ++i) c.ctrl->notclose[i] = 0;
    }
    if (b >= c.B) return;
    const State s = init_state(c, ia, b);
    c.state[0][b] = s;
    PBlock p;
    build_pblock(s, c.cfg.use_spherical_manifold != 0, c.cfg.use_log_focal != 0, p);
    c.pb[0][b] = p;
}

// Independent intrinsics: reduce the partial records, then one thread per image applies the LM step.
template <int PM>
__global__ __launch_bounds__(kGroups * kSlots) void update_kernel(SolveCtx c, int step) {
    if (c.cfg.early_stop && stop_fired_before(c.ctrl, step)) return;          // block-uniform
    // the leader of an image fetches its state BEFORE the reduction: the load overlaps the partial records' round trip
    // instead of heading the serial chain behind it
    const bool striped = reduce_images_per_block(c.nchunks) == 1;
    const int grp = threadIdx.x / kSlots, pb = striped ? (int)blockIdx.x : (int)blockIdx.x * kGroups + grp;
    State s{};
    if (threadIdx.x % kSlots == 0 && (!striped || grp == 0) && pb < c.B) s = c.state[step & 1][pb];
    int b;
    float acc[kNAccMax];
    if (!coop_reduce_partials(c.partials, c.B, c.nchunks, acc_floats(c.cfg.camera_model), b, acc)) return;
    if (lm_step<PM>(c.cfg, c.H, c.W, step, s, acc)) atomicAdd(&c.ctrl->notclose[step], 1);
    c.state[(step + 1) & 1][b] = s;
    PBlock p;
    build_pblock(s, c.cfg.use_spherical_manifold != 0, c.cfg.use_log_focal != 0, p);
    c.pb[(step + 1) & 1][b] = p;
}

// Parameter block of the final sweep: (roll, pitch, focal) parametrisation (lm_optimizer.py:481-483).
__global__ void prep_final_kernel(SolveCtx c) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    // which state buffer is final: theta_s if the early stop fired after update s (the tentative theta_{s+1}
    // is discarded, :619-625), else theta_{num_steps}
    int sel = c.cfg.num_steps & 1, stopped = 0;
    if (c.cfg.early_stop)
        for (int j = 1; j < c.cfg.num_steps; ++j)
            if (c.ctrl->notclose[j] == 0) { sel = j & 1; stopped = 1; break; }
    if (b == 0) { c.ctrl->stopped = stopped; c.ctrl->final_sel = sel; }    // for finalize_kernel
    if (b >= c.B) return;
    const State s = c.state[sel][b];
    PBlock p;
    build_pblock(s, false, c.iso_final != 0, p);     // iso_final: log-focal columns, rescaled in finalize_kernel
    c.pb_final[b] = p;
}

// Inverse of an N x N SPD matrix through its Cholesky factor, in double, all sizes compile-time (registers).
// torch.inverse (:484) uses a pivoted LU; on the SPD Hessians of this path both agree to rounding.
template <int N>
__device__ inline void spd_inverse(const float (&A)[N][N], double (&inv)[N][N]) {
    double L[N][N];
#pragma unroll
    for (int j = 0; j < N; ++j) {
        double sdiag = A[j][j];
#pragma unroll
        for (int k = 0; k < j; ++k) sdiag -= L[j][k] * L[j][k];
        const double l = sqrt(sdiag);
        L[j][j] = l;
#pragma unroll
        for (int i = j + 1; i < N; ++i) {
            double t = A[i][j];
#pragma unroll
            for (int k = 0; k < j; ++k) t -= L[i][k] * L[j][k];
            L[i][j] = t / l;
        }
    }
#pragma unroll
    for (int col = 0; col < N; ++col) {          // solve L L^T x = e_col
        double x[N];
#pragma unroll
        for (int i = 0; i < N; ++i) {
            double t = i == col ? 1.0 : 0.0;
#pragma unroll
            for (int k = 0; k < i; ++k) t -= L[i][k] * x[k];
            x[i] = t / L[i][i];
        }
#pragma unroll
        for (int i = N - 1; i >= 0; --i) {
            double t = x[i];
#pragma unroll
            for (int k = i + 1; k < N; ++k) t -= L[k][i] * x[k];
            x[i] = t / L[i][i];
        }
#pragma unroll
        for (int i = 0; i < N; ++i) inv[i][col] = x[i];
    }
}

// Final costs + estimate_uncertainty (lm_optimizer.py:632-642, 463-516) from the final sweep.
template <int PM>
__global__ __launch_bounds__(kGroups * kSlots) void finalize_kernel(SolveCtx c, float* cam, float* grav, float* info) {
    int b;
    float acc[kNAccMax];
    if (!coop_reduce_partials(c.partials, c.B, c.nchunks, acc_floats(c.cfg.camera_model), b, acc)) return;
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
    for (int i = 0; i < GCLM_INFO_STRIDE; ++i) o[i] = 0.f;      // every slot is written here: no memset by the caller
    // stop_at (lm_optimizer.py:575,620,638): first step after which EVERY image's cost was "close"; the counter of
    // the last comparison (num_steps) cannot change the answer, so nothing of this launch is needed
    int stop_at = cfg.num_steps;
    for (int j = 1; j < cfg.num_steps; ++j)
        if (c.ctrl->notclose[j] == 0) { stop_at = j; break; }
    o[GCLM_INFO_STOP_AT] = (float)stop_at;
    o[GCLM_INFO_INITIAL_UP_COST] = s.init_cu;
    o[GCLM_INFO_INITIAL_LAT_COST] = s.init_cl;
    o[GCLM_INFO_INITIAL_COST] = s.init_cu + s.init_cl;
    o[GCLM_INFO_FINAL_UP_COST] = cu;
    o[GCLM_INFO_FINAL_LAT_COST] = cl;
    o[GCLM_INFO_FINAL_COST] = total;
    bool act[PM];
    active_columns<PM>(cfg, act);
    int n = 0;
#pragma unroll
    for (int i = 0; i < PM; ++i) n += act[i] ? 1 : 0;
    o[GCLM_INFO_NPARAMS] = (float)n;
    o[GCLM_INFO_LAMBDA] = s.lambda;
    o[GCLM_INFO_STEP_FAILURES] = s.fails;
    if (cfg.compute_uncertainty) {
        float A[PM][PM], Gf[PM];
        unpack_system<PM>(acc, A, Gf);
        if (c.iso_final) {
            // the final sweep ran the log-focal specialisation (d(u,v)/dlog f = -(u,v)); with fx == fy the plain-focal
            // column of :481-483 is that column times 1/f  (d(u,v)/df = -(u,v)/f)
            const float wf = 1.0f / s.fy;
#pragma unroll
            for (int i = 0; i < PM; ++i) { A[2][i] *= wf; A[i][2] *= wf; }
        }
#pragma unroll
        for (int i = 0; i < PM; ++i)
#pragma unroll
            for (int j = 0; j < PM; ++j)
                if (!(act[i] && act[j])) A[i][j] = i == j ? 1.f : 0.f;
        double Cov[PM][PM];
        spd_inverse<PM>(A, Cov);                              // torch.inverse(Hess), :484
        int ci = 0;                                           // compressed (free-parameter) indices
#pragma unroll
        for (int i = 0; i < PM; ++i) {
            if (!act[i]) continue;
            int cj = 0;
#pragma unroll
            for (int j = 0; j < PM; ++j) {
                if (!act[j]) continue;
                o[GCLM_INFO_COV + ci * n + cj] = (float)Cov[i][j];
                ++cj;
            }
            ++ci;
        }
        if (cfg.estimate_gravity) {
            const double c00 = Cov[0][0], c11 = Cov[1][1], c01 = 0.5 * (Cov[0][1] + Cov[1][0]);
            o[GCLM_INFO_ROLL_UNC] = (float)sqrt(c00);
            o[GCLM_INFO_PITCH_UNC] = (float)sqrt(c11);
            const double tr = 0.5 * (c00 + c11), df = 0.5 * (c00 - c11);
            o[GCLM_INFO_GRAVITY_UNC] = (float)sqrt(tr + sqrt(df * df + c01 * c01));   // max eigvalsh, :495-496
        }
        if (cfg.estimate_focal) {
            const double fu = Cov[2][2];
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

// ---------------------------------------------------------------- shared intrinsics
// Whole group = one arrow-head system (lm_optimizer.py:350-383): 2x2 gravity blocks D_i on the
// diagonal, couplings E_i (2 x ni) to the ni shared intrinsics, C = sum H_ii.  Instead of the
// reference's dense (2B+ni)^2 Cholesky the Schur complement on the intrinsics is formed:
//   S = C~ - sum E_i^T D~_i^-1 E_i,  rhs = c - sum E_i^T D~_i^-1 g_i,  delta_I = S^-1 rhs,
//   delta_g,i = D~_i^-1 (g_i - E_i delta_I)         (~ = with LM damping on the diagonal)
// which is algebraically the same solve and reduces over frames with a plain SUM -- i.e. it can
// be all-reduced across devices when a group's frames are sharded (BASELINE config 5).

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
template <int PM>
__device__ inline bool frame_block(const float (&Hf)[PM][PM], float lambda, float (&Dinv)[2][2]) {
    const float a = Hf[0][0] + fmaxf(Hf[0][0] * lambda, 1e-6f), b = Hf[0][1], d = Hf[1][1] + fmaxf(Hf[1][1] * lambda, 1e-6f);
    const float det = a * d - b * b;
    if (!(a > 0.f) || !(det > 0.f)) return false;
    const float id = 1.0f / det;
    Dinv[0][0] = d * id; Dinv[0][1] = -b * id; Dinv[1][0] = -b * id; Dinv[1][1] = a * id;
    return true;
}

// Schur partials of one group (ni = 1..3 shared intrinsics), GCLM_SHARED_PARTIAL_STRIDE = 32 floats:
//   [0..9) sum E^T Dinv E (3x3 row-major), [9..12) sum E^T Dinv g, [12..21) sum H_ii, [21..24) sum g_i, [24] #frames;
//   NaN in [0] marks a non-PD frame block.
constexpr int kNI = 3, kGS = 0, kGR = 9, kGC = 12, kGc = 21, kGN = 24, kGSum = 24;

// One frame's step from the REDUCED partials `o` of its group: solve the (tiny) Schur system (redundantly per frame:
// ni <= 3), back-substitute the frame's own gravity block, update (lm_optimizer.py:597-606).
template <int PM, int NI>
__device__ inline void apply_shared_frame(const SolveCtx& c, int step, int b, const float* o) {
    const gclm_config& cfg = c.cfg;
    constexpr int ni = NI;
    State s = c.state[step & 1][b];
    float A[NI][NI], dS[NI], dI[kNI] = {}, dG[2] = {0.f, 0.f};
    bool ok = o[0] == o[0];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        dS[i] = o[kGc + i] - o[kGR + i];
#pragma unroll
        for (int j = 0; j < NI; ++j) A[i][j] = o[kGC + i * kNI + j] - o[kGS + i * kNI + j];
        A[i][i] += fmaxf(o[kGC + i * kNI + i] * s.lambda, 1e-6f);      // damping on sum H_ii (:123-126)
    }
    ok = chol_solve<NI>(A, dS) && ok;
#pragma unroll
    for (int i = 0; i < NI; ++i) dI[i] = dS[i];
    float Hf[PM][PM], Gf[PM], Dinv[2][2];
    unpack_system<PM>(c.frame_sys + (size_t)b * acc_floats(cfg.camera_model), Hf, Gf);
    // Everything that decides about the SHARED step is uniform over the frames of the group (the NaN marker of a
    // non-PD frame block travels in the reduced partials, the Schur solve is the same on every frame): either the
    // whole group takes dI or nobody does -- one camera per group stays one camera per group.
    ok = frame_block<PM>(Hf, s.lambda, Dinv) && ok;
    for (int i = 0; i < ni; ++i) ok = ok && fabsf(dI[i]) <= 3.0e38f;     // NaN / inf: a failed step of the group
    bool frame_failed = false;
    if (ok) {
        float r0 = Gf[0], r1 = Gf[1];
        for (int i = 0; i < ni; ++i) { r0 -= Hf[0][2 + i] * dI[i]; r1 -= Hf[1][2 + i] * dI[i]; }
        dG[0] = Dinv[0][0] * r0 + Dinv[0][1] * r1;
        dG[1] = Dinv[1][0] * r0 + Dinv[1][1] * r1;
        // a frame whose OWN back-substitution overflows keeps its gravity and still follows the group's intrinsics
        frame_failed = !(fabsf(dG[0]) <= 3.0e38f && fabsf(dG[1]) <= 3.0e38f);
        if (frame_failed) dG[0] = dG[1] = 0.f;
    } else {
        for (int i = 0; i < kNI; ++i) dI[i] = 0.f;
    }
    if (!ok || frame_failed) s.fails += 1.f;
    const V3 gv = grav_update({s.gx, s.gy, s.gz}, dG[0], dG[1], cfg.use_spherical_manifold != 0);
    s.gx = gv.x; s.gy = gv.y; s.gz = gv.z;
    update_focal(s, dI[0], cfg.use_log_focal != 0);
    if (ni >= 2) update_dist(s, cfg.camera_model, dI[1], dI[2]);
    c.state[(step + 1) & 1][b] = s;
    PBlock p;
    build_pblock(s, cfg.use_spherical_manifold != 0, cfg.use_log_focal != 0, p);
    c.pb[(step + 1) & 1][b] = p;
}

// ONE workgroup per group, cooperative over its frames (tiles of kTileFrames):
//   (1) reduce the sweep's partial records of 8 frames at a time (thread = (frame, slot), fixed chunk order, double),
//   (2) one thread per frame: mean costs / allclose bookkeeping, the frame's damped 2x2 block and its contribution
//       to the group's Schur partials,
//   (3) 24 threads sum the contributions over the frames in frame order (fp32, bit-reproducible),
//   then either write the partials for the all-reduce of the split protocol (APPLY = false: gclm_shared_reduce), or --
//   single device, the partials are already complete -- go straight on to the solve and the per-frame update
//   (APPLY = true): a shared-intrinsics LM step is sweep + THIS kernel, the same two launches as an independent one.
constexpr int kTileFrames = 64;
template <int PM, int NI, bool APPLY>
__global__ __launch_bounds__(kGroups * kSlots) void shared_step_kernel(SolveCtx c, int step, float* gp) {
    if (c.cfg.early_stop && stop_fired_before(c.ctrl, step)) return;          // block-uniform
    const int g = blockIdx.x, tid = threadIdx.x;
    const int nacc = acc_floats(c.cfg.camera_model);
    __shared__ float fsys[kTileFrames][kNAccMax];        // reduced per-frame systems of the tile
    __shared__ float contrib[kTileFrames][kGSum + 1];    // per-frame Schur contributions (+1: bank spread)
    __shared__ int bad[kTileFrames];
    __shared__ float gsum[GCLM_SHARED_PARTIAL_STRIDE];
    int f0, f1;
    group_range(c, g, f0, f1);
    if (tid < GCLM_SHARED_PARTIAL_STRIDE) gsum[tid] = 0.f;
    int any_bad = 0;                                     // thread 0 only
    const int grp = tid / kSlots, slot = tid % kSlots;
    const float invN = 1.0f / (float)((size_t)c.H * c.W);
    for (int t0 = f0; t0 < f1; t0 += kTileFrames) {
        const int nt = min(kTileFrames, f1 - t0);
        // (1) partial records -> fsys (and frame_sys in memory for the apply step)
        for (int base = 0; base < nt; base += kGroups) {
            const int fl = base + grp;
            if (fl < nt && slot < nacc) {
                const int b = t0 + fl;
                double d = 0.0;
                const float* p = c.partials + (size_t)b * c.nchunks * nacc + slot;
                for (int q0 = 0; q0 < c.nchunks; q0 += 16) {          // 16 loads in flight, ascending order (see above)
                    float v[16];
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        const float t = p[(size_t)min(q0 + j, c.nchunks - 1) * nacc];      // unconditional (clamped), then a select
                        v[j] = q0 + j < c.nchunks ? t : 0.f;
                    }
#pragma unroll
                    for (int j = 0; j < 16; ++j) d += v[j];
                }
                fsys[fl][slot] = (float)d;
                c.frame_sys[(size_t)b * nacc + slot] = (float)d;
            }
        }
        __syncthreads();
        // (2) per frame
        if (tid < nt) {
            const int b = t0 + tid;
            State s = c.state[step & 1][b];
            float cu, cl;
            const float total = total_cost(fsys[tid], invN, true, cu, cl);
            if (step == 0) { s.init_cu = cu; s.init_cl = cl; }     // infos["initial_*"] (:585-588)
            cost_bookkeeping(c.cfg, c.ctrl, step, total, s, false);   // lambda is never updated (:612)
            c.state[step & 1][b] = s;
            float Hf[PM][PM], Gf[PM], Dinv[2][2];
            unpack_system<PM>(fsys[tid], Hf, Gf);
            const bool ok = frame_block<PM>(Hf, s.lambda, Dinv);
            bad[tid] = ok ? 0 : 1;
            float* o = contrib[tid];
#pragma unroll
            for (int i = 0; i < kGSum; ++i) o[i] = 0.f;
            const float q0 = Dinv[0][0] * Gf[0] + Dinv[0][1] * Gf[1], q1 = Dinv[1][0] * Gf[0] + Dinv[1][1] * Gf[1];
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const float e0 = Hf[0][2 + i], e1 = Hf[1][2 + i];                         // E[:, i]
                const float t0_ = Dinv[0][0] * e0 + Dinv[0][1] * e1, t1_ = Dinv[1][0] * e0 + Dinv[1][1] * e1;
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    o[kGS + j * kNI + i] = Hf[0][2 + j] * t0_ + Hf[1][2 + j] * t1_;
                    o[kGC + i * kNI + j] = Hf[2 + i][2 + j];
                }
                o[kGR + i] = e0 * q0 + e1 * q1;
                o[kGc + i] = Gf[2 + i];
            }
        }
        __syncthreads();
        // (3) sum over the frames of the tile, frame order
        if (tid < kGSum) {
            float a = gsum[tid];
            for (int f = 0; f < nt; ++f) a += contrib[f][tid];
            gsum[tid] = a;
        }
        if (tid == 0)
            for (int f = 0; f < nt; ++f) any_bad |= bad[f];
        __syncthreads();
    }
    if (tid == 0) {
        if (any_bad) gsum[0] = __builtin_nanf("");
        gsum[kGN] = (float)(f1 - f0);
    }
    __syncthreads();
    if constexpr (!APPLY) {
        if (tid < GCLM_SHARED_PARTIAL_STRIDE) gp[(size_t)g * GCLM_SHARED_PARTIAL_STRIDE + tid] = gsum[tid];
    } else {
        for (int b = f0 + tid; b < f1; b += kGroups * kSlots) apply_shared_frame<PM, NI>(c, step, b, gsum);
    }
}

// split protocol, after the all-reduce: per frame, from the REDUCED partials
template <int PM, int NI>
__global__ void shared_apply_kernel(SolveCtx c, int step, const float* gp) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= c.B) return;
    if (c.cfg.early_stop && stop_fired_before(c.ctrl, step)) return;
    const int g = c.group_of_frame ? c.group_of_frame[b] : b / c.group_size;
    apply_shared_frame<PM, NI>(c, step, b, gp + (size_t)g * GCLM_SHARED_PARTIAL_STRIDE);
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

template <int PM>
__global__ __launch_bounds__(kGroups * kSlots) void system_out_kernel(SolveCtx c, float* cost, float* grad, float* hess) {
    int b;
    float acc[kNAccMax];
    if (!coop_reduce_partials(c.partials, c.B, c.nchunks, acc_floats(c.cfg.camera_model), b, acc)) return;
    const float invN = 1.0f / (float)((size_t)c.H * c.W);
    cost[b * 2] = acc[A_CU] * invN;
    cost[b * 2 + 1] = acc[A_CL] * invN;
    float Hf[PM][PM], Gf[PM];
    unpack_system<PM>(acc, Hf, Gf);
#pragma unroll
    for (int i = 0; i < GCLM_MAX_PARAMS; ++i) {
        grad[b * GCLM_MAX_PARAMS + i] = i < PM ? Gf[i < PM ? i : 0] : 0.f;
#pragma unroll
        for (int j = 0; j < GCLM_MAX_PARAMS; ++j)
            hess[(b * GCLM_MAX_PARAMS + i) * GCLM_MAX_PARAMS + j] = (i < PM && j < PM) ? Hf[i < PM ? i : 0][j < PM ? j : 0] : 0.f;
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

struct GT { float fx, k1, k2; V3 g; };
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
    t.k1 = model == GCLM_PINHOLE ? 0.f : -0.3f + (model == GCLM_SIMPLE_DIVISIONAL ? 0.35f : 0.4f) * u01(mix64(ibase + 4));
    t.k2 = model == GCLM_RADIAL ? -0.02f + 0.04f * u01(mix64(ibase + 5)) : 0.f;
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
            cm[6] = t.k1; cm[7] = t.k2;
        }
        if (gt_grav) { gt_grav[b * 3] = t.g.x; gt_grav[b * 3 + 1] = t.g.y; gt_grav[b * 3 + 2] = t.g.z; }
    }
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (size_t)gridDim.x * blockDim.x) {
        const int y = (int)(i / W), x = (int)(i - (size_t)y * W);
        const float u = ((float)x - W * 0.5f) / t.fx, v = ((float)y - H * 0.5f) / t.fx;
        const float r2 = u * u + v * v;
        const float px = t.g.x - t.g.z * u, py = t.g.y - t.g.z * v;
        // distortion scale s(r2), 2 ds/dr2 and undistortion scale e(r2) of the model (camera.py:611-636,
        // 712-746, 829-868)
        float d = 1.f, d1x2 = 0.f, e = 1.f;
        if (model == GCLM_SIMPLE_RADIAL) {
            d = 1.f + t.k1 * r2; d1x2 = 2.f * t.k1; e = 1.f - t.k1 * r2;
        } else if (model == GCLM_RADIAL) {
            d = 1.f + t.k1 * r2 + t.k2 * r2 * r2; d1x2 = 2.f * t.k1 + 4.f * t.k2 * r2;
            e = 1.f - t.k1 * r2 + (3.f * t.k1 * t.k1 - t.k2) * r2 * r2;
        } else if (model == GCLM_SIMPLE_DIVISIONAL) {
            // s = (1 - t) / (2 k r2), t = sqrt(1 - 4 k r2), and its derivative cancel catastrophically in float32 for small
            // k r2 (the reference's own forms, camera.py:829-868, flagged at :913): up to 1.6e-3 in the up vector.  The
            // generator renders the TRUE field, so it takes the algebraically equal conjugate forms
            // s = 2 / (1 + t), 2 ds/dr2 = 8 k / (t (1 + t)^2)  (test_synth_generator_renders_the_reference_field).
            const float ts = sqrtf(fmaxf(1.f - 4.f * t.k1 * r2, 0.f)), t0 = sqrtf(fmaxf(1.f - 4.f * t.k1 * r2, 1e-6f));
            d = 2.f / (1.f + ts);
            d1x2 = 8.f * t.k1 / (t0 * (1.f + t0) * (1.f + t0));
            e = 1.f / (1.f + t.k1 * r2);
        }
        const float tt = u * px + v * py;
        float qx = d * px + d1x2 * tt * u, qy = d * py + d1x2 * tt * v;
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

// ---------------------------------------------------------------- CNN-head epilogue (the step before the path)
// UpDecoder / LatitudeDecoder epilogues (geocalib.py:57,73-75) fused into ONE pass that writes the five
// planes the sweep reads: up = normalize(raw, dim=1), latitude = asin(clamp(tanh(raw), +-(1-1e-5))),
// confidences = sigmoid(log-confidence).  Eager PyTorch runs 8 elementwise kernels and ~18 plane passes.
template <int VEC>
__global__ void pack_fields_kernel(const float* __restrict__ up_raw, const float* __restrict__ up_lc,
                                   const float* __restrict__ lat_raw, const float* __restrict__ lat_lc, int B,
                                   size_t N, float* __restrict__ up, float* __restrict__ upc,
                                   float* __restrict__ lat, float* __restrict__ latc) {
    const size_t units = N / VEC;
    for (int b = blockIdx.y; b < B; b += gridDim.y) {
        const float* ux = up_raw + (size_t)b * 2 * N;
        const float* uy = ux + N;
        float* ox = up + (size_t)b * 2 * N;
        float* oy = ox + N;
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < units; i += (size_t)gridDim.x * blockDim.x) {
            float vx[VEC], vy[VEC], vl[VEC], c1[VEC], c2[VEC];
            if constexpr (VEC == 4) {
                // every byte is read once and written once: non-temporal both ways (+4 ... 8 % over plain float4 accesses,
                // profiles/archive/r05_pack_bench.log)
                typedef float pk4 __attribute__((ext_vector_type(4)));
                auto ld4 = [](const float* p, size_t j) { return __builtin_nontemporal_load(reinterpret_cast<const pk4*>(p) + j); };
                const pk4 a = ld4(ux, i), bq = ld4(uy, i), l = ld4(lat_raw + (size_t)b * N, i);
                vx[0] = a.x; vx[1] = a.y; vx[2] = a.z; vx[3] = a.w;
                vy[0] = bq.x; vy[1] = bq.y; vy[2] = bq.z; vy[3] = bq.w;
                vl[0] = l.x; vl[1] = l.y; vl[2] = l.z; vl[3] = l.w;
                if (up_lc) { const pk4 t = ld4(up_lc + (size_t)b * N, i); c1[0] = t.x; c1[1] = t.y; c1[2] = t.z; c1[3] = t.w; }
                if (lat_lc) { const pk4 t = ld4(lat_lc + (size_t)b * N, i); c2[0] = t.x; c2[1] = t.y; c2[2] = t.z; c2[3] = t.w; }
            } else {
                vx[0] = ux[i]; vy[0] = uy[i]; vl[0] = lat_raw[(size_t)b * N + i];
                if (up_lc) c1[0] = up_lc[(size_t)b * N + i];
                if (lat_lc) c2[0] = lat_lc[(size_t)b * N + i];
            }
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                const float n = fmaxf(sqrtf(vx[k] * vx[k] + vy[k] * vy[k]), 1e-12f);     // F.normalize eps
                vx[k] /= n; vy[k] /= n;
                vl[k] = asinf(fminf(fmaxf(tanhf(vl[k]), -1.0f + 1e-5f), 1.0f - 1e-5f));
                if (up_lc) c1[k] = 1.0f / (1.0f + expf(-c1[k]));
                if (lat_lc) c2[k] = 1.0f / (1.0f + expf(-c2[k]));
            }
            if constexpr (VEC == 4) {
                typedef float pk4 __attribute__((ext_vector_type(4)));
                auto st4 = [](float* p, size_t j, const float (&v)[VEC]) {
                    __builtin_nontemporal_store(pk4{v[0], v[1], v[2], v[3]}, reinterpret_cast<pk4*>(p) + j);
                };
                st4(ox, i, vx); st4(oy, i, vy); st4(lat + (size_t)b * N, i, vl);
                if (up_lc) st4(upc + (size_t)b * N, i, c1);
                if (lat_lc) st4(latc + (size_t)b * N, i, c2);
            } else {
                ox[i] = vx[0]; oy[i] = vy[0]; lat[(size_t)b * N + i] = vl[0];
                if (up_lc) upc[(size_t)b * N + i] = c1[0];
                if (lat_lc) latc[(size_t)b * N + i] = c2[0];
            }
        }
    }
}

// ---------------------------------------------------------------- _post_process (the step after the path)
// GeoCalib._post_process (extractor.py:51-69) brings the fields back to the input resolution with
// F.interpolate(mode="bilinear", align_corners=False); one launch for any number of (h, w) planes of up to 8 tensors.
// Source index as in ATen (area_pixel_compute_source_index): src = max(0, (dst + 0.5) * in/out - 0.5).
// Every path below evaluates, per output value and with these roundings,
//     top/bot = fma(r[x0], 1 - lx, r[x1] * lx)        out = fma(top, 1 - ly, bot * ly)
// so the scalar, gather and window kernels agree bit for bit (test_upsample_paths_agree_bitwise).
__device__ __forceinline__ float up_src(int X, float scale) { return fmaxf(__fmaf_rn((float)X + 0.5f, scale, -0.5f), 0.f); }
__device__ __forceinline__ float up_lerp(float a, float b, float l) { return __fmaf_rn(a, 1.f - l, __fmul_rn(b, l)); }

// Scalar path: any width / alignment; one output value per thread and iteration, rows walked with an incremental
// (row, column) counter instead of a division per pixel.
__device__ __forceinline__ void upsample_plane(const float* __restrict__ s, float* __restrict__ d, int h, int w, int H, int W) {
    const float sy = (float)h / (float)H, sx = (float)w / (float)W;
    const unsigned units = (unsigned)H * (unsigned)W;
    const unsigned stride = gridDim.x * blockDim.x;
    const int dY = (int)(stride / (unsigned)W), dX = (int)(stride - (unsigned)dY * (unsigned)W);
    unsigned q = blockIdx.x * blockDim.x + threadIdx.x;
    int Y = (int)(q / (unsigned)W), X = (int)(q - (unsigned)Y * (unsigned)W);
    for (; q < units; q += stride) {
        const float fy = up_src(Y, sy);
        const int y0 = min((int)fy, h - 1), y1 = min(y0 + 1, h - 1);
        const float ly = fy - (float)y0;
        const float* r0 = s + (size_t)y0 * w;
        const float* r1 = s + (size_t)y1 * w;
        const float fx = up_src(X, sx);
        const int x0 = min((int)fx, w - 1), x1 = min(x0 + 1, w - 1);
        const float lx = fx - (float)x0;
        d[(size_t)Y * W + X] = up_lerp(up_lerp(r0[x0], r0[x1], lx), up_lerp(r1[x0], r1[x1], lx), ly);
        Y += dY; X += dX;
        if (X >= W) { X -= W; ++Y; }
    }
}

// float4 paths (W % 4 == 0, 16-byte aligned planes).  A WAVE owns 64 float4 units (256 output pixels, ONE 1 KiB
// non-temporal store per output row: the output is the traffic, it is written once and read by a later kernel).
//   * The wave index is made scalar (readfirstlane): row terms and the row-reuse branches are SALU, not exec-masked VALU.
//   * WINDOW (upsampling by >= 1.5 horizontally, w >= 4): the eight taps of a lane's four pixels lie within FOUR consecutive
//     source floats, so a source row costs ONE 16-byte load per lane (4-byte aligned) and the horizontal interpolation is
//     a 5-term FMA chain over that window with per-lane weights -- 10 v_pk_fma_f32; round 4 used 24 v_cmp + 24 v_cndmask
//     register selects.  Descending order makes the chain reproduce up_lerp exactly: every other term is an exact zero,
//     the last non-zero term fused is the x0 tap; E carries the clamped right edge (x1 == x0, both weights on t.w).
//   * Position q of a row maps to unit (q + o) mod Wu, o = the units from the row's first byte up to the next 128-byte
//     line: every full wave store then starts on a line.  Rows of 1620 floats (6480 B) are not whole lines; round 4
//     wrote each wave's 1 KiB across nine lines, two of them partial, and the stores alone took 1.75x a flat fill
//     (profiles/archive/r05_upsample_bench.log: 652 -> 439 us for 64 x 5 planes 320x480 -> 1080x1620).
//   * CONSEC (rows are whole 64-byte half lines, Wu % 4 == 0): a wave walks ROWS consecutive output rows; every source
//     row it needs is loaded up front (RMAX, when the vertical ratio bounds their number) and its horizontally interpolated
//     values stay in registers for all output rows that tap it.  The (up to four) waves of a block take adjacent strips
//     of the same rows, so that a block writes whole rows.
//   * PHASED (ragged rows): rows 8 apart share the phase o (8 Wu = 0 mod 8 units), so wave i of a 512-thread block owns
//     rows Y0 + i + 8 j and keeps its column weights; all 2 ROWS window loads are issued before the arithmetic.
//   * GATHER (any other ratio): per-lane dword gathers, consecutive rows, no rotation.
typedef float up_v2 __attribute__((ext_vector_type(2)));
typedef float up_v4 __attribute__((ext_vector_type(4)));
typedef float up_v4u __attribute__((ext_vector_type(4), aligned(4)));
// NW = 4: upsampling by >= 1.5 (above).  NW = 5: upsampling by 1 ... 1.5 -- four adjacent outputs then tap at most FIVE
// consecutive source floats (one 16-byte + one 4-byte load per source row and lane, a 6-term chain); GeoCalib resizes
// the short side to 320 px, so every input between 320 and 480 px on its short side lands here.
template <int NW>
struct UpCol {
    int xs, u;               // window start (source floats), output unit of this lane
    up_v2 W[NW][2], E[2];    // weight of window float j for the outputs (0, 1) and (2, 3)
    bool live;
};
template <int NW>
struct UpWin { float t[NW]; };
template <int NW>
__device__ __forceinline__ UpWin<NW> up_load_window(const float* p) {
    UpWin<NW> r;
    const up_v4u q = *reinterpret_cast<const up_v4u*>(p);
    r.t[0] = q.x; r.t[1] = q.y; r.t[2] = q.z; r.t[3] = q.w;
    if constexpr (NW == 5) r.t[4] = p[4];
    return r;
}
template <int NW>
__device__ __forceinline__ UpCol<NW> up_col_weights(int q, int o, int w, int W, float sx) {
    UpCol<NW> c;
    const int Wu = W >> 2;
    int u = q + (Wu >= 64 ? o : 0);      // rows shorter than one wave store: nothing to align
    if (u >= Wu) u -= Wu;
    c.live = q < Wu;
    if (!c.live) u = 0;
    c.u = u;
    int x0[4], x1[4];
    float lx[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float fx = up_src(u * 4 + k, sx);
        x0[k] = min((int)fx, w - 1);
        x1[k] = min(x0[k] + 1, w - 1);
        lx[k] = fx - (float)x0[k];
    }
    c.xs = min(x0[0], w - NW);
    float Wm[NW][4], Em[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i0 = x0[k] - c.xs, i1 = x1[k] - c.xs;
#pragma unroll
        for (int j = 0; j < NW; ++j) Wm[j][k] = (j == i1) ? lx[k] : ((j == i0) ? 1.f - lx[k] : 0.f);
        Em[k] = (i0 == i1) ? 1.f - lx[k] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < NW; ++j) { c.W[j][0] = up_v2{Wm[j][0], Wm[j][1]}; c.W[j][1] = up_v2{Wm[j][2], Wm[j][3]}; }
    c.E[0] = up_v2{Em[0], Em[1]}; c.E[1] = up_v2{Em[2], Em[3]};
    return c;
}
template <int NW>
__device__ __forceinline__ void up_hwindow(const UpCol<NW>& c, const UpWin<NW>& t, up_v2 (&o)[2]) {
#pragma unroll
    for (int hlf = 0; hlf < 2; ++hlf) {
        const float last = t.t[NW - 1];
        up_v2 a = up_v2{last, last} * c.W[NW - 1][hlf];
        a = __builtin_elementwise_fma(up_v2{last, last}, c.E[hlf], a);
#pragma unroll
        for (int j = NW - 2; j >= 0; --j) a = __builtin_elementwise_fma(up_v2{t.t[j], t.t[j]}, c.W[j][hlf], a);
        o[hlf] = a;
    }
}
struct UpRow { int y0, y1; float ly; };
__device__ __forceinline__ UpRow up_row_terms(int Y, int h, float sy) {      // Y wave-uniform: the results are scalars
    const float fy = up_src(Y, sy);
    const int y0 = min((int)fy, h - 1);
    UpRow r;
    r.ly = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, fy - (float)y0)));
    r.y0 = __builtin_amdgcn_readfirstlane(y0);
    r.y1 = min(r.y0 + 1, h - 1);
    return r;
}
__device__ __forceinline__ void up_vblend_store(float* p, bool live, float ly, const up_v2 (&ha)[2], const up_v2 (&hb)[2]) {
    const up_v2 l2 = up_v2{ly, ly}, m2 = up_v2{1.f - ly, 1.f - ly};
    const up_v2 o0 = __builtin_elementwise_fma(ha[0], m2, hb[0] * l2), o1 = __builtin_elementwise_fma(ha[1], m2, hb[1] * l2);
    if (live) __builtin_nontemporal_store(up_v4{o0.x, o0.y, o1.x, o1.y}, reinterpret_cast<up_v4*>(p));
}
__device__ __forceinline__ unsigned up_line_unit(const float* d) { return (unsigned)((reinterpret_cast<uintptr_t>(d) >> 4) & 7u); }

// RMAX > 0: the rows Y0 .. Y0 + ROWS - 1 tap at most RMAX source rows (the host derives it from the vertical ratio),
// all loaded before the arithmetic; RMAX = 0: loaded as the walk reaches them (vertical downsampling).
template <int ROWS, int NW, int RMAX>
__device__ __forceinline__ void upsample_consec(const float* __restrict__ s, float* __restrict__ d, int h, int w, int H, int W, int q,
                                                int Y0) {
    const float sy = (float)h / (float)H, sx = (float)w / (float)W;
    const UpCol<NW> c = up_col_weights<NW>(q, (int)((8u - up_line_unit(d)) & 7u), w, W, sx);
    const int Yend = min(Y0 + ROWS, H);
    const float* sc = s + c.xs;
    float* dc = d + c.u * 4;
    if constexpr (RMAX > 0) {
        const UpRow ra = up_row_terms(Y0, h, sy), rb = up_row_terms(Yend - 1, h, sy);
        const int ylo = ra.y0, yhi = rb.y1, n = yhi - ylo + 1;
        UpWin<NW> t[RMAX];
#pragma unroll
        for (int r = 0; r < RMAX; ++r) t[r] = up_load_window<NW>(sc + (size_t)min(ylo + r, yhi) * w);
        up_v2 hc[2], hn[2];
        up_hwindow<NW>(c, t[0], hc);
        int Y = Y0;
#pragma unroll
        for (int r = 0; r < RMAX; ++r) {
            if (r < n) {
                if (r + 1 < RMAX && r + 1 < n) up_hwindow<NW>(c, t[r + 1 < RMAX ? r + 1 : r], hn);
                else { hn[0] = hc[0]; hn[1] = hc[1]; }        // y1 == y0: the last source row
                while (Y < Yend) {
                    const UpRow rt = up_row_terms(Y, h, sy);
                    if (rt.y0 - ylo != r) break;
                    up_vblend_store(dc + (size_t)Y * W, c.live, rt.ly, hc, hn);
                    ++Y;
                }
                hc[0] = hn[0]; hc[1] = hn[1];
            }
        }
    } else {
        int ya = -1, yb = -1;
        up_v2 ha[2] = {up_v2{0.f, 0.f}, up_v2{0.f, 0.f}}, hb[2] = {up_v2{0.f, 0.f}, up_v2{0.f, 0.f}};
        for (int Y = Y0; Y < Yend; ++Y) {
            const UpRow rt = up_row_terms(Y, h, sy);
            if (!(rt.y0 == ya && rt.y1 == yb)) {
                if (rt.y0 == yb) { ha[0] = hb[0]; ha[1] = hb[1]; }
                else if (rt.y0 != ya) up_hwindow<NW>(c, up_load_window<NW>(sc + (size_t)rt.y0 * w), ha);
                ya = rt.y0;
                if (rt.y1 == rt.y0) { hb[0] = ha[0]; hb[1] = ha[1]; }
                else up_hwindow<NW>(c, up_load_window<NW>(sc + (size_t)rt.y1 * w), hb);
                yb = rt.y1;
            }
            up_vblend_store(dc + (size_t)Y * W, c.live, rt.ly, ha, hb);
        }
    }
}
template <int ROWS, int NW>
__device__ __forceinline__ void upsample_phased(const float* __restrict__ s, float* __restrict__ d, int h, int w, int H, int W, int q,
                                                int Ya) {
    const float sy = (float)h / (float)H, sx = (float)w / (float)W;
    const unsigned g = up_line_unit(d) + (unsigned)Ya * (unsigned)(W >> 2);      // the row's first unit, mod 8 = its phase
    const UpCol<NW> c = up_col_weights<NW>(q, (int)((8u - (g & 7u)) & 7u), w, W, sx);
    const float* sc = s + c.xs;
    float* dc = d + c.u * 4;
    UpWin<NW> ta[ROWS], tb[ROWS];
    UpRow rt[ROWS];
#pragma unroll
    for (int j = 0; j < ROWS; ++j) {
        rt[j] = up_row_terms(min(Ya + 8 * j, H - 1), h, sy);
        ta[j] = up_load_window<NW>(sc + (size_t)rt[j].y0 * w);
        tb[j] = up_load_window<NW>(sc + (size_t)rt[j].y1 * w);
    }
#pragma unroll
    for (int j = 0; j < ROWS; ++j) {
        up_v2 ha[2], hb[2];
        up_hwindow<NW>(c, ta[j], ha);
        up_hwindow<NW>(c, tb[j], hb);
        const int Y = Ya + 8 * j;
        up_vblend_store(dc + (size_t)Y * W, c.live && Y < H, rt[j].ly, ha, hb);
    }
}
template <int ROWS>
__device__ __forceinline__ void upsample_gather(const float* __restrict__ s, float* __restrict__ d, int h, int w, int H, int W, int Xu,
                                                int Y0) {
    const float sy = (float)h / (float)H, sx = (float)w / (float)W;
    const bool live = Xu * 4 < W;
    int x0[4], x1[4];
    float lx[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float fx = up_src(min(Xu * 4 + k, W - 1), sx);
        x0[k] = min((int)fx, w - 1);
        x1[k] = min(x0[k] + 1, w - 1);
        lx[k] = fx - (float)x0[k];
    }
    auto hrow = [&](int y, float (&o)[4]) {
        const float* r = s + (size_t)y * w;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = up_lerp(r[x0[k]], r[x1[k]], lx[k]);
    };
    int ya = -1, yb = -1;
    float ha[4] = {0.f, 0.f, 0.f, 0.f}, hb[4] = {0.f, 0.f, 0.f, 0.f};
    const int Yend = min(Y0 + ROWS, H);
    for (int Y = Y0; Y < Yend; ++Y) {
        const UpRow rt = up_row_terms(Y, h, sy);
        if (!(rt.y0 == ya && rt.y1 == yb)) {
            if (rt.y0 == yb) {
#pragma unroll
                for (int k = 0; k < 4; ++k) ha[k] = hb[k];
            } else if (rt.y0 != ya) {
                hrow(rt.y0, ha);
            }
            ya = rt.y0;
            if (rt.y1 == rt.y0) {
#pragma unroll
                for (int k = 0; k < 4; ++k) hb[k] = ha[k];
            } else {
                hrow(rt.y1, hb);
            }
            yb = rt.y1;
        }
        if (live)
            __builtin_nontemporal_store(up_v4{up_lerp(ha[0], hb[0], rt.ly), up_lerp(ha[1], hb[1], rt.ly), up_lerp(ha[2], hb[2], rt.ly),
                                              up_lerp(ha[3], hb[3], rt.ly)},
                                        reinterpret_cast<up_v4*>(d + (size_t)Y * W + Xu * 4));
    }
}

// grid = (strips of 64 units, row groups, planes of all tensors); blockIdx.z strides over the planes beyond 65 535
struct UpPlane { const float* s; float* d; };
__device__ __forceinline__ UpPlane up_plane(const UpsampleMulti& m, int P, int h, int w, int H, int W) {
    int t = 0;
    while (t < m.n - 1 && P >= m.planes[t]) { P -= m.planes[t]; ++t; }
    return UpPlane{m.src[t] + (size_t)P * h * w, m.dst[t] + (size_t)P * H * W};
}
enum { kUpScalar = 0, kUpGather = 1, kUpConsec = 2, kUpPhased = 3 };
template <int KIND, int ROWS, int NW, int RMAX>
__global__ __launch_bounds__(KIND == kUpPhased ? 512 : 256) void upsample_kernel(UpsampleMulti m, int total, int h, int w, int H, int W,
                                                                                  int rowblock) {
    if constexpr (KIND == kUpScalar) {
        for (int P = blockIdx.y; P < total; P += gridDim.y) {
            const UpPlane pl = up_plane(m, P, h, w, H, W);
            upsample_plane(pl.s, pl.d, h, w, H, W);
        }
    } else {
        const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        // CONSEC: the waves of a block take ADJACENT strips of the same ROWS rows -- a block writes whole rows (of up to
        // 1024 px), row after row, instead of four row groups of one strip: 0.346 -> 0.323 ms on the x2 case
        // (profiles/archive/r05_upsample_bench.log, "rowblock").  The other kinds: blockIdx.x = strip, waves = row groups / phases.
        // (one image: the old grouping, four row groups of one strip per block -- 4.5 against 5.5 us for 5 planes of 480x640)
        const bool rb = KIND == kUpConsec && rowblock != 0;
        const int strip = rb ? blockIdx.x * 4 + wave : blockIdx.x;
        const int q = strip * 64 + (threadIdx.x & 63);
        const int Y0 = KIND == kUpPhased ? blockIdx.y * (8 * ROWS) + wave : (rb ? blockIdx.y * ROWS : (blockIdx.y * 4 + wave) * ROWS);
        if (Y0 >= H || strip * 64 >= (W >> 2)) return;
        for (int P = blockIdx.z; P < total; P += gridDim.z) {
            const UpPlane pl = up_plane(m, P, h, w, H, W);
            if constexpr (KIND == kUpGather) upsample_gather<ROWS>(pl.s, pl.d, h, w, H, W, q, Y0);
            else if constexpr (KIND == kUpPhased) upsample_phased<ROWS, NW>(pl.s, pl.d, h, w, H, W, q, Y0);
            else upsample_consec<ROWS, NW, RMAX>(pl.s, pl.d, h, w, H, W, q, Y0);
        }
    }
}
// A window of NW source floats holds every tap of four adjacent output pixels when x0[3] <= x0[0] + NW - 2, i.e.
// 3 sx <= NW - 2 in exact arithmetic (NW = 4: upsampling by >= 1.5; NW = 5: by >= 1).  The kernel derives x0 from fp32
// (X + 0.5) sx - 0.5, whose error per value is below fx 2^-23 <= w 2^-23; two of them must not bridge the gap
// (NW - 2) - 3 w / W = ((NW - 2) W - 3 w) / W, so away from the exact ratios (1.5: no source coordinate of a lane's first
// pixel is an integer, (16 u - 1) / 6; 1: every coordinate is an exact integer) the window path needs gap 2^21 > w W.
__host__ inline bool upsample_window_ok(int nw, int w, int W) {
    const long long gap = (long long)(nw - 2) * W - 3LL * w;
    return w >= nw && (gap == 0 || (gap > 0 && (double)gap * 2097152.0 > (double)w * (double)W));
}
struct UpPlan { int kind, nw, pref; };      // pref: 0 none, 1 = 3 h <= 2 H (2 ROWS / 3 + 3 source rows), 2 = h <= H (ROWS + 2)
__host__ inline UpPlan upsample_plan(const UpsampleMulti& m, int h, int w, int H, int W) {
    bool vec4 = W % 4 == 0;
    for (int t = 0; t < m.n; ++t) vec4 = vec4 && (reinterpret_cast<uintptr_t>(m.dst[t]) & 15u) == 0;
    if (!vec4) return UpPlan{kUpScalar, 4, 0};
    const int nw = upsample_window_ok(4, w, W) ? 4 : (upsample_window_ok(5, w, W) ? 5 : 0);
    if (!nw) return UpPlan{kUpGather, 4, 0};
    // Ragged = rows that are not a whole number of 64-BYTE half lines (W / 4 units of 16 B, not divisible by 4): only then do
    // the wave stores of the consecutive-row kernel leave 16 ... 48-byte slivers.  Rows of 2240 B (560 px) alternate between
    // line-aligned and 64 B off, and stream at full rate without rotation (5.9 TB/s against 4.0 through the phased kernel,
    // which gives up the vertical reuse); rows of 6480 B (1620 px) need it (3.9 against 5.6).
    if ((W / 4) % 4 != 0) return UpPlan{kUpPhased, nw, 0};
    return UpPlan{kUpConsec, nw, 3LL * h <= 2LL * H ? 1 : (h <= H ? 2 : 0)};
}

// optimizer_step (lm_optimizer.py:109-137) as a batched device kernel: delta = (H + diag(clamp(lambda diag H, eps)))^-1 G
// by an fp32 Cholesky per system (the reference copies H, G to the CPU for this, twice per LM step).  A system
// that is not positive definite takes a zero step and raises its flag (the reference zeroes the whole batch).
template <int N>
__global__ void lm_step_kernel(const float* G, const float* H, const float* lambda, int lambda_stride, float eps, int B,
                               float* delta, int* failed) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float A[N][N], g[N];
    const float lam = lambda[(size_t)b * lambda_stride];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        g[i] = G[(size_t)b * N + i];
#pragma unroll
        for (int j = 0; j < N; ++j) A[i][j] = H[((size_t)b * N + i) * N + j];
    }
#pragma unroll
    for (int i = 0; i < N; ++i) A[i][i] += fmaxf(A[i][i] * lam, eps);
    const bool ok = chol_solve<N>(A, g);
#pragma unroll
    for (int i = 0; i < N; ++i) delta[(size_t)b * N + i] = ok ? g[i] : 0.f;
    if (failed) failed[b] = ok ? 0 : 1;
}

// calculate_gradient_and_hessian (lm_optimizer.py:317-385) on MATERIALISED tensors: G = sum_px w J^T r,
// H = sum_px w J^T J for J (B,N,R,P), r (B,N,R), w (B,N).  One workgroup per image, fixed summation order
// (per-thread strided partial sums in double, then a tree over the 256 threads in LDS).  The solve itself never
// forms J; this serves callers that hold the tensors.
template <int P>
__global__ __launch_bounds__(256) void gradient_hessian_kernel(const float* J, const float* r, const float* w, int N,
                                                               int R, int accumulate, float* G, float* H) {
    constexpr int NV = P + P * (P + 1) / 2;
    const int b = blockIdx.x, tid = threadIdx.x;
    double acc[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) acc[i] = 0.0;
    for (int px = tid; px < N; px += 256) {
        const size_t q = (size_t)b * N + px;
        const float wt = w[q];
        for (int row = 0; row < R; ++row) {
            const float* j = J + (q * R + row) * P;
            const float res = r[q * R + row];
            float jr[P];
#pragma unroll
            for (int k = 0; k < P; ++k) jr[k] = j[k];
            int o = P;
#pragma unroll
            for (int k = 0; k < P; ++k) {
                const float wk = wt * jr[k];
                acc[k] += (double)(wk * res);
#pragma unroll
                for (int l = k; l < P; ++l) acc[o++] += (double)(wk * jr[l]);
            }
        }
    }
    __shared__ double red[256];
    double total[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        red[tid] = acc[i];
        __syncthreads();
        for (int sft = 128; sft > 0; sft >>= 1) {
            if (tid < sft) red[tid] += red[tid + sft];
            __syncthreads();
        }
        total[i] = red[0];
        __syncthreads();
    }
    if (tid != 0) return;
    int o = P;
#pragma unroll
    for (int k = 0; k < P; ++k) {
        float* g = G + (size_t)b * P + k;
        *g = (accumulate ? *g : 0.f) + (float)total[k];
#pragma unroll
        for (int l = k; l < P; ++l) {
            const float v = (float)total[o++];
            float* hkl = H + ((size_t)b * P + k) * P + l;
            float* hlk = H + ((size_t)b * P + l) * P + k;
            const float nv = (accumulate ? *hkl : 0.f) + v;
            *hkl = nv;
            *hlk = nv;
        }
    }
}

inline dim3 grid1(int n) { return dim3((n + 127) / 128); }

}  // namespace

#define GCLM_L(kernel, n, s, ...) hipLaunchKernelGGL(kernel, grid1(n), dim3(128), 0, s, __VA_ARGS__)
// cooperative-reduce kernels: 256-thread blocks of 8 images (or 1 striped image, see coop_reduce_partials)
#define GCLM_LR(kernel, n, s, ...) \
    hipLaunchKernelGGL(kernel, dim3(reduce_blocks((n), c.nchunks)), dim3(kGroups * kSlots), 0, s, __VA_ARGS__)

hipError_t launch_init(const SolveCtx& c, const InitArgs& ia, hipStream_t s) {
    GCLM_L(init_kernel, c.B > 0 ? c.B : 1, s, c, ia);     // B = 0 (an empty shard of the split protocol): thread 0 still resets Ctrl
    return hipGetLastError();
}
hipError_t launch_update(const SolveCtx& c, int step, hipStream_t s) {
    if (acc_pm(c.cfg.camera_model) == 5) GCLM_LR(update_kernel<5>, c.B, s, c, step);
    else GCLM_LR(update_kernel<4>, c.B, s, c, step);
    return hipGetLastError();
}
hipError_t launch_prep_final(const SolveCtx& c, hipStream_t s) {
    GCLM_L(prep_final_kernel, c.B, s, c);
    return hipGetLastError();
}
hipError_t launch_finalize(const SolveCtx& c, float* d_cam, float* d_grav, float* d_info, hipStream_t s) {
    if (acc_pm(c.cfg.camera_model) == 5) GCLM_LR(finalize_kernel<5>, c.B, s, c, d_cam, d_grav, d_info);
    else GCLM_LR(finalize_kernel<4>, c.B, s, c, d_cam, d_grav, d_info);
    return hipGetLastError();
}
// Parts of ONE batch solved by separate handles (LMOptimizer.overlap_streams): infos["stop_at"] is a property of the
// whole batch -- the first step after which EVERY image's cost was "close" (lm_optimizer.py:619-620) -- so it is
// re-derived from the SUM of the parts' per-step counters and written into every row of every part.
__global__ void merge_stop_kernel(MergeStopArgs a) {
    __shared__ int stop;
    if (threadIdx.x == 0) {
        int s = a.num_steps;
        for (int j = 1; j < a.num_steps; ++j) {
            int moved = 0;
            for (int p = 0; p < a.n; ++p) moved += a.ctrl[p]->notclose[j];
            if (moved == 0) { s = j; break; }
        }
        stop = s;
    }
    __syncthreads();
    const float v = (float)stop;
    for (int p = 0; p < a.n; ++p)
        for (int i = threadIdx.x; i < a.B[p]; i += blockDim.x) a.info[p][(size_t)i * GCLM_INFO_STRIDE + GCLM_INFO_STOP_AT] = v;
}
hipError_t launch_merge_stop(const MergeStopArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(merge_stop_kernel, dim3(1), dim3(256), 0, s, a);
    return hipGetLastError();
}
#define GCLM_LG(kernel, s, ...) hipLaunchKernelGGL(kernel, dim3(c.n_groups), dim3(kGroups * kSlots), 0, s, __VA_ARGS__)
hipError_t launch_shared_reduce(const SolveCtx& c, int step, float* d_group_partials, hipStream_t s) {
    if (c.n_groups <= 0) return hipSuccess;
    switch (c.cfg.camera_model) {
        case GCLM_PINHOLE: GCLM_LG((shared_step_kernel<4, 1, false>), s, c, step, d_group_partials); break;
        case GCLM_RADIAL: GCLM_LG((shared_step_kernel<5, 3, false>), s, c, step, d_group_partials); break;
        default: GCLM_LG((shared_step_kernel<4, 2, false>), s, c, step, d_group_partials); break;
    }
    return hipGetLastError();
}
// single device: reduce + solve + update of every group in ONE launch
hipError_t launch_shared_step(const SolveCtx& c, int step, hipStream_t s) {
    if (c.n_groups <= 0) return hipSuccess;
    switch (c.cfg.camera_model) {
        case GCLM_PINHOLE: GCLM_LG((shared_step_kernel<4, 1, true>), s, c, step, nullptr); break;
        case GCLM_RADIAL: GCLM_LG((shared_step_kernel<5, 3, true>), s, c, step, nullptr); break;
        default: GCLM_LG((shared_step_kernel<4, 2, true>), s, c, step, nullptr); break;
    }
    return hipGetLastError();
}
#undef GCLM_LG
hipError_t launch_shared_apply(const SolveCtx& c, int step, const float* d_group_partials, hipStream_t s) {
    switch (c.cfg.camera_model) {
        case GCLM_PINHOLE: GCLM_L((shared_apply_kernel<4, 1>), c.B, s, c, step, d_group_partials); break;
        case GCLM_RADIAL: GCLM_L((shared_apply_kernel<5, 3>), c.B, s, c, step, d_group_partials); break;
        default: GCLM_L((shared_apply_kernel<4, 2>), c.B, s, c, step, d_group_partials); break;
    }
    return hipGetLastError();
}
hipError_t launch_system_out(const SolveCtx& c, float* d_cost, float* d_grad, float* d_hess, hipStream_t s) {
    if (acc_pm(c.cfg.camera_model) == 5) GCLM_LR(system_out_kernel<5>, c.B, s, c, d_cost, d_grad, d_hess);
    else GCLM_LR(system_out_kernel<4>, c.B, s, c, d_cost, d_grad, d_hess);
    return hipGetLastError();
}
hipError_t launch_pblock_from_params(const SolveCtx& c, const float* d_cam, const float* d_grav, int as_rpf,
                                     PBlock* out, hipStream_t s) {
    GCLM_L(pblock_from_params_kernel, c.B, s, c, d_cam, d_grav, as_rpf, out);
    return hipGetLastError();
}
// Several tensors of (h, w) planes in ONE launch (the four tensors _post_process resizes: up 2 planes per image, latitude,
// two confidences): a single-image calibrate() pays one launch instead of four.  Small jobs (one image) take half the
// rows per wave: twice the waves to fill 256 CUs.
template <int KIND, int NW, int PREF>
static void launch_upsample_kind(const UpsampleMulti& m, int total, int h, int w, int H, int W, bool small, hipStream_t s) {
    const int strips = (W / 4 + 63) / 64, gz = total < 65535 ? total : 65535;
    auto go = [&](auto rows) {
        constexpr int R = decltype(rows)::value;
        constexpr int RMAX = PREF == 1 ? 2 * R / 3 + 3 : (PREF == 2 ? R + 2 : 0);
        if (KIND == kUpConsec && !small) {
            const int waves = strips < 4 ? strips : 4;            // 640 px: 2.5 strips = 3 waves per block, no idle wave
            hipLaunchKernelGGL((upsample_kernel<KIND, R, NW, RMAX>), dim3((strips + 3) / 4, (H + R - 1) / R, gz), dim3(64 * waves), 0, s, m, total,
                               h, w, H, W, 1);
        } else {
            const dim3 grid(strips, KIND == kUpPhased ? (H + 8 * R - 1) / (8 * R) : (H + 4 * R - 1) / (4 * R), gz);
            hipLaunchKernelGGL((upsample_kernel<KIND, R, NW, RMAX>), grid, dim3(KIND == kUpPhased ? 512 : 256), 0, s, m, total, h, w, H, W, 0);
        }
    };
    // one image: half the rows per wave, twice the waves for 256 CUs; the phased five-float window with 8 rows holds 16
    // windows = 138 VGPRs, one 512-thread block per CU: 4 rows (90 VGPRs, two blocks) stream faster
    if (small || (KIND == kUpPhased && NW == 5)) go(std::integral_constant<int, 4>{});
    else go(std::integral_constant<int, 8>{});
}
hipError_t launch_upsample_multi(const UpsampleMulti& m, int h, int w, int H, int W, hipStream_t s) {
    long long total = 0;
    for (int t = 0; t < m.n; ++t) total += m.planes[t];
    if (total == 0 || (size_t)H * W == 0) return hipSuccess;
    if (total > 0x7fffffffLL) return hipErrorInvalidValue;
    const bool small = (double)total * H * W < 16.0e6;
    const int n = (int)total;
    const UpPlan p = upsample_plan(m, h, w, H, W);
    if (p.kind == kUpScalar) {
        unsigned bx = ((unsigned)H * (unsigned)W + 256 * 4 - 1) / (256 * 4);
        hipLaunchKernelGGL((upsample_kernel<kUpScalar, 1, 4, 0>), dim3(bx < 1 ? 1 : bx, (unsigned)(total < 65535 ? total : 65535)), dim3(256), 0, s, m, n,
                           h, w, H, W, 0);
    } else if (p.kind == kUpGather) {
        launch_upsample_kind<kUpGather, 4, 0>(m, n, h, w, H, W, small, s);
    } else if (p.kind == kUpPhased) {
        if (p.nw == 4) launch_upsample_kind<kUpPhased, 4, 0>(m, n, h, w, H, W, small, s);
        else launch_upsample_kind<kUpPhased, 5, 0>(m, n, h, w, H, W, small, s);
    } else if (p.nw == 4) {
        if (p.pref == 1) launch_upsample_kind<kUpConsec, 4, 1>(m, n, h, w, H, W, small, s);
        else if (p.pref == 2) launch_upsample_kind<kUpConsec, 4, 2>(m, n, h, w, H, W, small, s);
        else launch_upsample_kind<kUpConsec, 4, 0>(m, n, h, w, H, W, small, s);
    } else {
        if (p.pref == 1) launch_upsample_kind<kUpConsec, 5, 1>(m, n, h, w, H, W, small, s);
        else if (p.pref == 2) launch_upsample_kind<kUpConsec, 5, 2>(m, n, h, w, H, W, small, s);
        else launch_upsample_kind<kUpConsec, 5, 0>(m, n, h, w, H, W, small, s);
    }
    return hipGetLastError();
}
hipError_t launch_upsample(const float* src, int planes, int h, int w, int H, int W, float* dst, hipStream_t s) {
    UpsampleMulti m{};
    m.src[0] = src; m.dst[0] = dst; m.planes[0] = planes; m.n = 1;
    return launch_upsample_multi(m, h, w, H, W, s);
}

hipError_t launch_pack_fields(const float* up_raw, const float* up_lc, const float* lat_raw, const float* lat_lc,
                              int B, int H, int W, bool vec4, float* up, float* upc, float* lat, float* latc,
                              hipStream_t s) {
    if (B <= 0) return hipSuccess;
    const size_t N = (size_t)H * W;
    const size_t units = vec4 ? N / 4 : N;
    // one unit per thread and image where the image allows (one pass: +3 % over 128 blocks walking a grid-stride loop)
    const int bx = (int)((units + 255) / 256 < 2048 ? (units + 255) / 256 : 2048);
    const dim3 grid(bx, B < 4096 ? B : 4096), block(256);
    if (vec4) hipLaunchKernelGGL(pack_fields_kernel<4>, grid, block, 0, s, up_raw, up_lc, lat_raw, lat_lc, B, N, up, upc, lat, latc);
    else hipLaunchKernelGGL(pack_fields_kernel<1>, grid, block, 0, s, up_raw, up_lc, lat_raw, lat_lc, B, N, up, upc, lat, latc);
    return hipGetLastError();
}

// gclm_read_probe (include/gclm.h): the sweep's load -- non-temporal, 16 B per lane, consecutive lanes on consecutive
// addresses -- over up to 8 planes, four loads per plane in flight per thread, nothing else: the memory-system ceiling of the
// sweep's access pattern on the caller's own buffers.
struct ReadProbeArgs { const float* p[8]; int n; };
__global__ __launch_bounds__(256) void read_probe_kernel(ReadProbeArgs a, size_t units, float* sink) {
    typedef float v4 __attribute__((ext_vector_type(4)));
    constexpr int kUnroll = 4;
    const size_t base = (size_t)blockIdx.x * (256 * kUnroll) + threadIdx.x;
    v4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
        const size_t i = base + (size_t)u * 256;
        if (i < units)
            for (int k = 0; k < a.n; ++k) acc += __builtin_nontemporal_load(reinterpret_cast<const v4*>(a.p[k]) + i);
    }
    if (acc.x + acc.y + acc.z + acc.w == 1.2345e30f && sink) sink[0] = acc.x;      // never: keeps the loads alive
}
hipError_t launch_read_probe(const float* const* planes, int n, size_t floats, hipStream_t s) {
    ReadProbeArgs a{};
    a.n = n;
    for (int k = 0; k < n; ++k) a.p[k] = planes[k];
    const size_t units = floats / 4;
    if (units == 0) return hipSuccess;
    hipLaunchKernelGGL(read_probe_kernel, dim3((unsigned)((units + 1023) / 1024)), dim3(256), 0, s, a, units, (float*)nullptr);
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

hipError_t launch_lm_step(const float* d_G, const float* d_H, const float* d_lambda, int lambda_stride, float eps, int B,
                          int P, float* d_delta, int* d_failed, hipStream_t s) {
    if (B <= 0) return hipSuccess;
    switch (P) {
#define GCLM_STEP(N) \
    case N: hipLaunchKernelGGL(lm_step_kernel<N>, grid1(B), dim3(128), 0, s, d_G, d_H, d_lambda, lambda_stride, eps, B, d_delta, d_failed); break
        GCLM_STEP(1); GCLM_STEP(2); GCLM_STEP(3); GCLM_STEP(4); GCLM_STEP(5);
#undef GCLM_STEP
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_gradient_hessian(const float* d_J, const float* d_r, const float* d_w, int B, int N, int R, int P,
                                   int accumulate, float* d_G, float* d_H, hipStream_t s) {
    if (B <= 0) return hipSuccess;
    switch (P) {
#define GCLM_GH(K) \
    case K: hipLaunchKernelGGL(gradient_hessian_kernel<K>, dim3(B), dim3(256), 0, s, d_J, d_r, d_w, N, R, accumulate, d_G, d_H); break
        GCLM_GH(1); GCLM_GH(2); GCLM_GH(3); GCLM_GH(4); GCLM_GH(5);
#undef GCLM_GH
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace gclm
