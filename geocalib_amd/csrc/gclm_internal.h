// gclm_internal.h -- shared declarations of the three translation units of libgeocalib_hip.so
// (gclm_pass.hip: per-pixel sweep, gclm_update.hip: per-image / per-group solve + update,
//  gclm_api.hip: C ABI and launch sequence).  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gclm.h"

namespace gclm {

constexpr int kBlock = 256;        // 4 waves of 64 (per-pixel helper kernels)
constexpr int kMaxTile = 512;      // most lanes of one column-stationary tile of the sweep (see plan_geometry)
constexpr int kNAcc = 16;          // floats per partial record for models with <= 4 parameters (see enum Acc)
constexpr int kNAccMax = 24;       // ... and for the 5-parameter `radial` model (2 + 5 + 15 = 22, padded)
constexpr int kPBlockFloats = 20;
constexpr int kMaxP = GCLM_MAX_PARAMS;
constexpr int kStateFloats = 16;

// Partial / accumulator record of one sweep over (part of) an image.
//   [0] sum Huber cost up   [1] sum Huber cost latitude
//   [2..5] G = sum w J^T r over columns (d1, d2, f, k1)
//   [6..15] upper triangle of sum w J^T J: 00 01 02 03 11 12 13 22 23 33
enum Acc { A_CU = 0, A_CL = 1, A_G0 = 2, A_H00 = 6 };
// General layout for PM = 4 (record of 16) or PM = 5 (record of 24) full columns (d1, d2, f, k1[, k2]):
//   [2 .. 2+PM) gradient, then the upper triangle of the Hessian row-major.
__host__ __device__ constexpr int acc_pm(int camera_model) { return camera_model == GCLM_RADIAL ? 5 : 4; }
__host__ __device__ constexpr int acc_floats(int camera_model) { return camera_model == GCLM_RADIAL ? kNAccMax : kNAcc; }
__host__ __device__ constexpr int acc_h(int pm, int i, int j) {   // i <= j
    return 2 + pm + i * pm - (i * (i - 1)) / 2 + (j - i);
}
__host__ __device__ constexpr int num_dist_params(int camera_model) {
    return camera_model == GCLM_PINHOLE ? 0 : (camera_model == GCLM_RADIAL ? 2 : 1);
}

// Per-image constants consumed by the sweep (written by the update kernels).
struct __attribute__((aligned(16))) PBlock {
    float ifx, ify, cx, cy;          // normalize(): u = (x - cx) * ifx          camera.py:309-311
    float ga, gb, gc, k1;            // gravity (a,b,c), distortion
    float T00, T01, T10, T11;        // tangent basis T[i][k] (3x2): SphericalManifold.J_plus(g) in the
    float T20, T21, wfx, wfy;        //   loop, Gravity.J_rp() for uncertainty; (wfx,wfy): focal column
    float k2, pad0, pad1, pad2;      //   scale (1,1) for log-focal, (1/fx,1/fy) otherwise; k2: radial model
};
static_assert(sizeof(PBlock) == kPBlockFloats * 4, "PBlock layout");

// Per-image optimiser state (double-buffered across steps).
struct __attribute__((aligned(16))) State {
    float w, h, fx, fy, cx, cy, k1, k2;   // BaseCamera._data layout
    float gx, gy, gz;                      // Gravity._data
    float lambda, prev_cost, fails, init_cu, init_cl;   // init_*: initial mean costs (infos)
};
static_assert(sizeof(State) == kStateFloats * 4, "State layout");

// Device-side control block: batch-global early stop without host synchronisation.
// Batch-global early stop (lm_optimizer.py:619-625) without a host sync or a decision kernel: update `s`
// counts the images whose cost still moved into notclose[s]; the stop fires after the first update s >= 1
// that leaves notclose[s] == 0.  Every later launch of the loop is skipped, so its own counter stays 0 too,
// hence "the stop fired before LM step k"  <=>  k >= 2 && notclose[k-1] == 0.
struct Ctrl {
    int stopped;                           // set by prep_final once the early stop has fired
    int final_sel;                         // state buffer (0/1) that holds the final estimate
    int pad[2];
    int notclose[GCLM_MAX_STEPS + 4];      // per step: number of images whose cost still moved
};

__host__ __device__ inline bool stop_fired_before(const Ctrl* ctrl, int step) {
    return step >= 2 && ctrl->notclose[step - 1] == 0;
}

struct SweepArgs {
    const float* up;        // (B,2,H,W) or nullptr
    const float* lat;       // (B,1,H,W)
    const float* upc;       // (B,H,W) or nullptr
    const float* latc;      // (B,H,W) or nullptr
    const PBlock* pb;       // (B)
    const Ctrl* ctrl;       // nullptr: never skip
    float* partials;        // (B, nchunks, acc_floats(model))
    int B, H, W;
    int nchunks;            // workgroups (partial records) per image = ceil(jobs / 4)
    int vec;                // 4: float4 path, 1: scalar path
    int wu, cu, nstrips;    // units (float4 groups / pixels) per row, per strip, strips per row
    int rpi, rows_per_block;// rows per loop iteration of a tile (tile lanes / cu), rows a tile walks down
    int wpt, jobs;          // waves per tile (rpi * cu lanes rounded up to whole waves), wave jobs per image
    int log_focal;          // the parameter block was built for the log-focal parametrisation (wfx = wfy = 1)
    int stop_step;          // >= 2: LM step of this loop sweep, skipped once the early stop has fired (else 0)
    float up_scale, lat_scale;   // Huber scales a (lm_optimizer.py:158-159)
    float* slat;            // library-owned scratch plane (B,H,W) of sin(latitude_field), or nullptr (gclm_pass.hip: row_math)
    int slat_mode;          // 0: `lat` holds radians, nothing is stored; 1: ... and this sweep fills `slat`;
                            // 2: `lat` IS the filled scratch plane (sin(latitude) is loaded, not computed)
    int hrows;              // rows the tiles walk: H, or H / 2 when a lane takes row H - y along with row y (mirror)
    int mirror;             // 1: the row-pair walk (gclm_pass.hip: row_math_mirror); the geometry was planned for H / 2 rows
};

struct SolveCtx;
struct FusedArgs;

struct Geometry {          // how a sweep is cut into blocks (column-stationary tiles, see gclm_pass.hip)
    int vec, nchunks;
    int wu, cu, nstrips, rpi, rows_per_block, wpt, jobs;
    int mirror, hrows;     // the row-pair walk: tiles cover rows [0, H / 2), every lane takes the mirror row along
};
Geometry plan_geometry(int B, int H, int W, bool aligned16, int sweep_iters = 0, int camera_model = 0, bool mirror = false);
bool sweep_has_mirror(int camera_model);       // gclm_pass.hip: the model's five-plane float4 sweep has the row-pair instantiations
bool sweep_mirror_builtin(int camera_model);   // ... and they are the library's own choice for it

// gclm_pass.hip
hipError_t launch_gradient_hessian(const float* d_J, const float* d_r, const float* d_w, int B, int N, int R, int P,
                                   int accumulate, float* d_G, float* d_H, hipStream_t s);
hipError_t launch_lm_step(const float* d_G, const float* d_H, const float* d_lambda, int lambda_stride, float eps, int B,
                          int P, float* d_delta, int* d_failed, hipStream_t s);
hipError_t launch_residual_fields(int camera_model, const float* d_up, const float* d_lat, const float* d_cam,
                                  const float* d_grav, int B, int H, int W, float* d_r_up, float* d_r_lat, hipStream_t s);
hipError_t launch_huber_costs(const float* d_residual, size_t n, int dim, float scale, const float* d_conf,
                              float* d_cost, float* d_weight, float* d_second, hipStream_t s);
hipError_t launch_jacobian_fields(int camera_model, const float* d_cam, const float* d_grav, int B, int H, int W,
                                  int spherical, int log_focal, float* d_J_up, float* d_J_lat, hipStream_t s);
hipError_t launch_sweep(int camera_model, const SweepArgs& a, hipStream_t s);

// gclm_update.hip
struct SolveCtx {
    gclm_config cfg;
    int B, H, W, nchunks;
    int iso_final;              // the final (uncertainty) sweep runs in the log-focal form and finalize rescales its focal
                                //   column by 1/f: valid when fx == fy for every image (gclm_calibrate without `scales`)
    int n_groups, group_size;   // shared intrinsics
    const int32_t* group_of_frame; // device (B), non-decreasing group id per frame, or nullptr (uniform group_size)
    State* state[2];
    PBlock* pb[2];
    PBlock* pb_final;
    float* partials;
    float* partials2;           // second half of the double buffer (fused small-batch path)
    float* frame_sys;           // (B, acc_floats(model)) reduced per-frame system (shared mode)
    Ctrl* ctrl;
};
struct InitArgs {              // initial estimate: explicit (cam, grav) or trivial estimation from the priors
    const float* cam;           // (B,8) or nullptr -> get_trivial_estimation on the device
    const float* grav;          // (B,3)
    const float* scales;        // (2,) or nullptr
    const float* prior_focal;   // (B,) or nullptr
    const float* prior_gravity; // (B,3) or nullptr
    const float* prior_dist;    // (B, prior_dist_cols) or nullptr
    int prior_dist_cols;
    const float* up;            // heuristic initialisation reads three pixels of the fields
    const float* lat;
};
// One-launch-per-step variant of the sweep for small batches (gclm_pass.hip: fused_step_kernel)
struct FusedArgs {
    SolveCtx c;
    int step;                   // the launch sweeps theta_step after applying update step-1 (final launch: num_steps)
    int is_final;               // the uncertainty sweep: (roll, pitch, focal) block, takes over prep_final_kernel
    const float* partials_in;   // the partial records of the previous launch (the other half of the double buffer)
    unsigned* progress;         // paced launches (gclm_set_paced_launches): host-mapped word image 0 reports to, or null
    unsigned epoch;             // ... and the tag of this solve in it (stale words of earlier solves are ignored)
    InitArgs ia;                // step 0: the initial estimate is built in this launch's prologue (no init_kernel launch) ...
    int init_here;              // ... when set; workgroup 0 of an image commits state[0], image 0's also resets Ctrl
};
// the word a paced launch publishes: [31:20] epoch of the solve, bit 16 "the early stop fired here", [15:0] step + 1
constexpr unsigned kPacedStopBit = 1u << 16;
constexpr unsigned kPacedEpochShift = 20, kPacedEpochMask = 0xfffu;
constexpr int kPacedPatienceUs = 2000;    // host wait for one report before the rest of the solve is issued unpaced ...
constexpr int kPacedCooldown = 64;        // ... and solves of that handle that then do not pace at all
bool sweep_has_slat_plane(int camera_model);   // gclm_pass.hip: the model's five-plane float4 sweep has the SLAT instantiations
constexpr int kMaxMergeParts = 8;
struct MergeStopArgs {          // gclm_merge_stop_at: parts of one batch solved by separate handles
    const Ctrl* ctrl[kMaxMergeParts];
    float* info[kMaxMergeParts];
    int B[kMaxMergeParts];
    int n, num_steps;
};
hipError_t launch_merge_stop(const MergeStopArgs& a, hipStream_t s);
hipError_t launch_fused_step(int camera_model, const SweepArgs& a, const FusedArgs& f, hipStream_t s);
hipError_t launch_init(const SolveCtx& c, const InitArgs& ia, hipStream_t s);
hipError_t launch_update(const SolveCtx& c, int step, hipStream_t s);
hipError_t launch_prep_final(const SolveCtx& c, hipStream_t s);
hipError_t launch_finalize(const SolveCtx& c, float* d_cam, float* d_grav, float* d_info, hipStream_t s);
hipError_t launch_shared_reduce(const SolveCtx& c, int step, float* d_group_partials, hipStream_t s);
hipError_t launch_shared_apply(const SolveCtx& c, int step, const float* d_group_partials, hipStream_t s);
hipError_t launch_shared_step(const SolveCtx& c, int step, hipStream_t s);
hipError_t launch_system_out(const SolveCtx& c, float* d_cost, float* d_grad, float* d_hess, hipStream_t s);
hipError_t launch_pblock_from_params(const SolveCtx& c, const float* d_cam, const float* d_grav, int as_rpf, PBlock* out, hipStream_t s);
hipError_t launch_upsample(const float* src, int planes, int h, int w, int H, int W, float* dst, hipStream_t s);
constexpr int kMaxUpsampleTensors = 8;
struct UpsampleMulti {          // gclm_upsample_fields_multi: several tensors of (h, w) planes in one launch
    const float* src[kMaxUpsampleTensors];
    float* dst[kMaxUpsampleTensors];
    int planes[kMaxUpsampleTensors];
    int n;
};
hipError_t launch_upsample_multi(const UpsampleMulti& m, int h, int w, int H, int W, hipStream_t s);
hipError_t launch_pack_fields(const float* up_raw, const float* up_lc, const float* lat_raw, const float* lat_lc,
                              int B, int H, int W, bool vec4, float* up, float* upc, float* lat, float* latc,
                              hipStream_t s);
hipError_t launch_read_probe(const float* const* planes, int n, size_t floats, hipStream_t s);
hipError_t launch_synth(int camera_model, uint64_t seed, int64_t first_index, int B, int H, int W,
                        float sigma, int group_size, int run, int run_stride, float* up, float* lat, float* upc, float* latc, float* gt_cam,
                        float* gt_grav, hipStream_t s);

}  // namespace gclm
