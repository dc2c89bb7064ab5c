/*
 * gclm.h -- C ABI of libgeocalib_hip.so: GeoCalib's batched Levenberg-Marquardt calibration
 * (perspective-field residuals, analytic Jacobians, J^T W J / J^T W r reductions, damped solve,
 * manifold update) as hand-written HIP kernels for gfx950 (MI355X).
 *
 * The reference (cvg/GeoCalib) has no native/FFI boundary: its seam is the Python attribute
 * `GeoCalib.optimizer` (geocalib/geocalib.py:106,119), an `LMOptimizer` (geocalib/lm_optimizer.py:141).
 * Every entry point below states which reference function(s) it replaces.  All pointers named
 * `d_*` are DEVICE pointers owned by the caller (torch-ROCm tensors); `stream` is a hipStream_t
 * passed as void*.  Calls are asynchronous on `stream` and never synchronise the device -- with ONE exception: a
 * handle's workspace is allocated by the first solve and GROWN by the first solve of a larger shape than any before
 * (batch size, chunks per image, groups; the sin(latitude) scratch plane, an allocation of its own -- see
 * gclm_set_slat_plane): that call runs hipFree + hipMalloc, which synchronises the device once.  Every later call of that
 * shape or a smaller one finds the workspace in place: no allocation and no synchronisation after warm-up
 * (gclm_workspace_bytes reports its size, gclm_release_workspace gives it back).
 * Return value: 0 = ok, negative = error (message via gclm_last_error).  No C++ exception crosses
 * this boundary.  A handle is not thread-safe and owns the whole solve workspace: one handle per (device, stream).
 * Every entry point runs on the handle's device and restores the caller's current HIP device before it returns.
 * Numerical failure (non positive-definite system, or a NaN / inf step) is NOT an error: the image takes a zero
 * step and counts it in GCLM_INFO_STEP_FAILURES, like lm_optimizer.py:129-133 (there batch-global, here per image /
 * per shared group).
 * Degenerate inputs stay contained to their image and never hang (tests/test_gpu_parity.py::
 * test_bad_images_are_contained): an image with all-zero confidences keeps its initial estimate and reports a
 * non-finite covariance (the reference's torch.inverse raises on the singular Hessian); a NaN in a field makes every
 * step of THAT image fail (it keeps its initial estimate, its costs are NaN, the batch-global early stop then never
 * fires) and leaves the other images of the batch bit-identical to a clean run.
 * One call takes at most 65 535 images (grid.y); geocalib_amd.LMOptimizer slices larger batches.
 */
#ifndef GCLM_H
#define GCLM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI version: bumped whenever struct gclm_config or the export list changes (100 = round 1/2, 300 = round 3:
 * struct_size / abi_version / device moved into gclm_config, gclm_create lost its third argument, new entry points
 * gclm_set_sweep_iters, gclm_set_fused_steps, gclm_set_paced_launches, gclm_set_stop_comm, gclm_comm_all_reduce_sum_i32,
 * gclm_abi_config_size; 400 = round 4: gclm_comm_versions, gclm_merge_stop_at and gclm_upsample_fields_multi added, the NULL-handle error strings became thread-local, an
 * empty batch (B = 0, NULL fields) is accepted by gclm_solve / gclm_calibrate; 500 = round 5: gclm_set_slat_plane and gclm_plan_cut added;
 * 600 = round 6: gclm_set_slat_plane_limit, gclm_slat_plane_bytes, gclm_release_workspace and gclm_read_probe added, the
 * scratch plane became an optional allocation of its own, gclm_merge_stop_at skips empty parts; 610 = round 6: gclm_set_row_pairs
 * added -- radial / simple_divisional batches walk row pairs by default, results equal the one-row walk's to summation order).  gclm_create refuses a gclm_config whose first two fields do not
 * carry the library's own sizeof(gclm_config) and GCLM_VERSION, with a message naming both sides. */
#define GCLM_VERSION 610

/* camera_models of geocalib/camera.py:945-950 */
enum gclm_camera_model {
    GCLM_PINHOLE = 0,           /* camera.py:522 */
    GCLM_SIMPLE_RADIAL = 1,     /* camera.py:565 */
    GCLM_RADIAL = 2,            /* camera.py:663 */
    GCLM_SIMPLE_DIVISIONAL = 3  /* camera.py:789 */
};

#define GCLM_MAX_PARAMS 5     /* delta_g1, delta_g2, focal, k1, k2 */
#define GCLM_MAX_STEPS 256
#define GCLM_CAM_STRIDE 8     /* {w,h,fx,fy,cx,cy,k1,k2}: BaseCamera._data, camera.py:25-41 */
#define GCLM_GRAV_STRIDE 3    /* unit gravity: Gravity._data, gravity.py:18-28 */

/* info_out row layout (floats) -- the `infos` dict of lm_optimizer.py:575,586-588,509-516,638-642 */
#define GCLM_INFO_STRIDE 48
enum gclm_info_slot {
    GCLM_INFO_STOP_AT = 0,
    GCLM_INFO_INITIAL_UP_COST = 1,
    GCLM_INFO_INITIAL_LAT_COST = 2,
    GCLM_INFO_INITIAL_COST = 3,
    GCLM_INFO_FINAL_UP_COST = 4,
    GCLM_INFO_FINAL_LAT_COST = 5,
    GCLM_INFO_FINAL_COST = 6,
    GCLM_INFO_ROLL_UNC = 7,
    GCLM_INFO_PITCH_UNC = 8,
    GCLM_INFO_GRAVITY_UNC = 9,
    GCLM_INFO_FOCAL_UNC = 10,
    GCLM_INFO_VFOV_UNC = 11,
    GCLM_INFO_NPARAMS = 12,
    GCLM_INFO_LAMBDA = 13,
    GCLM_INFO_STEP_FAILURES = 14,  /* number of LM steps rejected: Cholesky failed or the step was NaN / inf (zero step) */
    GCLM_INFO_COV = 16             /* covariance, P x P row-major, up to 25 floats */
};

/*
 * Mirrors LMOptimizer.default_conf (lm_optimizer.py:144-162) plus what
 * setup_optimization_and_priors (lm_optimizer.py:189-246) derives from the priors.
 */
typedef struct gclm_config {
    int32_t struct_size;             /* sizeof(gclm_config) of the CALLER's header (set by gclm_default_config) */
    int32_t abi_version;             /* GCLM_VERSION of the CALLER's header (set by gclm_default_config) */
    int32_t device;                  /* HIP device ordinal the handle is bound to (LMOptimizer(...).to(device)) */
    int32_t camera_model;            /* enum gclm_camera_model */
    int32_t shared_intrinsics;       /* lm_optimizer.py:147; groups of `group_size` frames */
    int32_t group_size;              /* frames per shared-intrinsics group; 0 = whole batch (reference) */
    int32_t num_steps;               /* :149 */
    float lambda0;                   /* :150 */
    int32_t fix_lambda;              /* :151 */
    int32_t early_stop;              /* :152 (batch-global, :90-92,:619-625) */
    float atol, rtol;                /* :153-154; the test |new - prev| <= atol + |rtol * prev| is evaluated in float32, as
                                        torch.allclose does for float32 costs (the scalar tolerances do not promote) */
    int32_t use_spherical_manifold;  /* :155 */
    int32_t use_log_focal;           /* :156 */
    float up_loss_fn_scale;          /* :158 */
    float lat_loss_fn_scale;         /* :159 */
    int32_t estimate_gravity;        /* :204-207 (0 when prior_gravity is given) */
    int32_t estimate_focal;          /* :209-212 */
    int32_t estimate_dist;           /* :214-221 */
    int32_t compute_uncertainty;     /* eval mode: estimate_uncertainty (:635-636) */
    int32_t heuristic_init;          /* gclm_calibrate only: siclib's get_heuristic_estimation instead of the trivial
                                        estimate (siclib/models/optimization/utils.py:27-82; needs the up field) */
} gclm_config;

typedef struct gclm_handle gclm_handle;

/* Library / ABI version (GCLM_VERSION of the build) and the sizeof(gclm_config) the library was built with. */
int gclm_version(void);
int gclm_abi_config_size(void);

/* Fill `cfg` with LMOptimizer.default_conf (lm_optimizer.py:144-162), everything estimated, eval mode, device 0,
 * and the library's own struct_size / abi_version.  A caller compiled against another header therefore passes the
 * check of gclm_create only if its struct really has the library's layout: compare gclm_abi_config_size() with the
 * caller's sizeof(gclm_config) (and gclm_version() with GCLM_VERSION) BEFORE calling this with a smaller struct. */
int gclm_default_config(gclm_config* cfg);

/* LMOptimizer.__init__ (lm_optimizer.py:164-171): validate the configuration, bind to cfg->device.  Signature as
 * SURVEY.md section 8(b).  Fails (-5, message via gclm_last_error(NULL)) when cfg->struct_size / cfg->abi_version
 * are not the library's: a stale caller is told so instead of handing over a too-small struct. */
int gclm_create(gclm_handle** out, const gclm_config* cfg);

/* Re-configure (set_camera_model :173, .shared_intrinsics, priors) without dropping the workspace.  The device of a
 * handle cannot change. */
int gclm_configure(gclm_handle* h, const gclm_config* cfg);

int gclm_destroy(gclm_handle* h);

/* Message of the last failing call on this handle; h == NULL: of the last failing gclm_create of the CALLING THREAD
 * (thread-local storage: two threads creating handles do not see each other's message). */
const char* gclm_last_error(const gclm_handle* h);

/* Bytes of device scratch the handle holds (grows on demand in gclm_solve, never per call after warm-up): the core
 * workspace (per-image states, parameter blocks, partial records: ~3 KB per image) plus the sin(latitude) scratch plane
 * where one is held (gclm_slat_plane_bytes: H x W x 4 bytes per image). */
size_t gclm_workspace_bytes(const gclm_handle* h);
size_t gclm_slat_plane_bytes(const gclm_handle* h);
/* Give the handle's device memory back (hipFree: waits for the device, so nothing of the handle is in flight after).  The
 * handle stays valid; its next solve allocates again.  For serving loops that have seen a rare huge batch. */
int gclm_release_workspace(gclm_handle* h);

/*
 * LMOptimizer.optimize (lm_optimizer.py:551-644) for B images of H x W pixels: all LM steps,
 * the final costs and (compute_uncertainty) estimate_uncertainty (:463-516).
 *   d_up        (B,2,H,W) float32 up field, or NULL          (data["up_field"])
 *   d_lat       (B,1,H,W) float32 latitude in radians        (data["latitude_field"], required like :31)
 *   d_up_conf   (B,H,W)   float32 or NULL                    (data["up_confidence"])
 *   d_lat_conf  (B,H,W)   float32 or NULL                    (data["latitude_confidence"])
 *   d_cam_io    (B,8)     in: initial camera (get_trivial_estimation :20-58), out: optimised camera
 *   d_grav_io   (B,3)     in: initial unit gravity, out: optimised gravity
 *   d_info_out  (B,GCLM_INFO_STRIDE) float32, see enum gclm_info_slot
 */
int gclm_solve(gclm_handle* h, const float* d_up, const float* d_lat, const float* d_up_conf,
               const float* d_lat_conf, int B, int H, int W, float* d_cam_io, float* d_grav_io,
               float* d_info_out, void* stream);

/*
 * LMOptimizer.forward (lm_optimizer.py:646-664): get_trivial_estimation (:20-58; roll = pitch = 0,
 * f = 0.7 max(h, w) through the vfov round trip, principal point at the centre, priors substituted) evaluated
 * on the device, then gclm_solve.  No host-side tensor op is needed before the call.
 *   d_scales         (2,)   or NULL   data["scales"]:  fx = f * scales[0] / scales[1]   (camera.py:89-90)
 *   d_prior_focal    (B,)   or NULL   data["prior_focal"]   (requires cfg.estimate_focal == 0)
 *   d_prior_gravity  (B,3)  or NULL   data["prior_gravity"] (requires cfg.estimate_gravity == 0)
 *   d_prior_dist     (B,prior_dist_cols) or NULL   data["prior_dist"]
 *   d_cam_out (B,8), d_grav_out (B,3), d_info_out (B,GCLM_INFO_STRIDE): outputs
 */
int gclm_calibrate(gclm_handle* h, const float* d_up, const float* d_lat, const float* d_up_conf,
                   const float* d_lat_conf, int B, int H, int W, const float* d_scales,
                   const float* d_prior_focal, const float* d_prior_gravity, const float* d_prior_dist,
                   int prior_dist_cols, float* d_cam_out, float* d_grav_out, float* d_info_out, void* stream);

/*
 * One fused sweep at fixed parameters: calculate_residuals + calculate_costs + setup_system
 * (lm_optimizer.py:248-315,387-461).  as_rpf selects the (roll, pitch, focal) parametrisation used
 * by estimate_uncertainty (:481-483).  Outputs, per image: d_cost (B,2) mean up / latitude Huber cost,
 * d_grad (B,5) and d_hess (B,5,5) over the full column set [d1,d2,f,k1,k2] (unused columns zero).
 */
int gclm_system(gclm_handle* h, const float* d_up, const float* d_lat, const float* d_up_conf,
                const float* d_lat_conf, int B, int H, int W, const float* d_cam, const float* d_grav,
                int as_rpf, float* d_cost, float* d_grad, float* d_hess, void* stream);

/*
 * Shared-intrinsics solve with a group's frames split over several devices (BASELINE config 5).
 * The caller drives the LM loop: per step, gclm_shared_reduce leaves, for each of the G groups,
 * the local Schur partials [sum E^T D^-1 E (3 x 3) | sum E^T D^-1 g (3) | sum H_ii (3 x 3) | sum g_i (3) | #frames]
 * over its ni <= 3 shared intrinsics (unused entries zero)
 * in d_partials (G x GCLM_SHARED_PARTIAL_STRIDE floats); the caller all-reduces (sum) that buffer over the
 * ranks holding frames of the same groups (RCCL), then gclm_shared_apply solves and updates its frames.
 * Together they replace the dense arrow-head Cholesky of lm_optimizer.py:350-383,:597-603.
 */
#define GCLM_SHARED_PARTIAL_STRIDE 32
int gclm_shared_begin(gclm_handle* h, const float* d_up, const float* d_lat, const float* d_up_conf,
                      const float* d_lat_conf, int B_local, int H, int W, float* d_cam_io,
                      float* d_grav_io, const int32_t* d_group_of_frame /* (B_local), non-decreasing */,
                      int num_groups, void* stream);
int gclm_shared_reduce(gclm_handle* h, int step, float* d_partials, void* stream);
int gclm_shared_apply(gclm_handle* h, int step, const float* d_partials, void* stream);
int gclm_shared_finish(gclm_handle* h, float* d_info_out, void* stream);

/*
 * The step BEFORE the path: the CNN head epilogues UpDecoder / LatitudeDecoder (geocalib/geocalib.py:57,73-75)
 * fused into one pass that writes the planes gclm_calibrate reads:
 *   up = normalize(up_raw, dim=1)  (eps 1e-12),   latitude = asin(clamp(tanh(lat_raw), +-(1 - 1e-5))),
 *   up_conf = sigmoid(up_logconf),                lat_conf = sigmoid(lat_logconf)     (NULL: skipped)
 * Shapes: up_raw / up (B,2,H,W); lat_raw / lat (B,1,H,W); log-confidences / confidences (B,H,W).  In-place
 * operation (output pointer == input pointer) is allowed.
 */
int gclm_pack_fields(const float* d_up_raw, const float* d_up_logconf, const float* d_lat_raw,
                     const float* d_lat_logconf, int B, int H, int W, float* d_up, float* d_up_conf, float* d_lat,
                     float* d_lat_conf, void* stream);

/*
 * The step AFTER the path: GeoCalib._post_process (geocalib/extractor.py:51-69) resizes the fields and confidences
 * back to the input resolution with F.interpolate(mode="bilinear", align_corners=False).  `planes` = number of
 * contiguous (h, w) planes in d_src (e.g. B*2 for the up field); d_dst holds planes x (H, W).
 * The sources must be FINITE: the vector path reads a window of 4 or 5 consecutive source floats per lane and multiplies
 * the ones an output does not tap by an exact 0, so a NaN / Inf source pixel reaches up to four output pixels beyond the
 * ones F.interpolate would make non-finite (which itself turns a zero-weight non-finite tap into NaN).  The CNN-head
 * outputs this entry point exists for (unit up vectors, asin(tanh) latitudes, sigmoid confidences) are always finite.
 */
int gclm_upsample_fields(const float* d_src, int planes, int h, int w, int H, int W, float* d_dst, void* stream);
/* ... and the same for up to 8 tensors of (h, w) planes in ONE launch (the four tensors _post_process resizes: a
 * single-image calibrate() pays one launch instead of four). */
int gclm_upsample_fields_multi(const float* const* d_srcs, float* const* d_dsts, const int* planes, int n_tensors, int h, int w,
                               int H, int W, void* stream);

/*
 * LMOptimizer.calculate_gradient_and_hessian (geocalib/lm_optimizer.py:317-385) on materialised tensors:
 *   d_G (B,P) = sum_px w J^T r,   d_H (B,P,P) = sum_px w J^T J
 * for d_J (B,N,R,P), d_residual (B,N,R), d_weight (B,N); R rows per pixel (2 for the up field, 1 for latitude).
 * `accumulate` != 0 adds to the existing d_G / d_H (up + latitude, :444-459).  The solve itself never forms J
 * (gclm_solve contracts it in registers); this serves callers that hold the tensors.
 */
int gclm_gradient_hessian(const float* d_J, const float* d_residual, const float* d_weight, int B, int N, int R,
                          int P, int accumulate, float* d_G, float* d_H, void* stream);

/*
 * optimizer_step (geocalib/lm_optimizer.py:109-137) for B systems of P <= 5 unknowns, on the device:
 *   delta = (H + diag(clamp(lambda * diag(H), min = eps)))^-1 G        (fp32 Cholesky per system)
 * d_G (B,P), d_H (B,P,P), d_lambda (B) or one value (lambda_is_scalar), d_delta (B,P).  A system that is not
 * positive definite gets delta = 0 and d_failed[b] = 1 (d_failed may be NULL); the reference moves H and G to the
 * CPU for this step and zeroes the whole batch on a failure.
 */
int gclm_optimizer_step(const float* d_G, const float* d_H, const float* d_lambda, int lambda_is_scalar, float eps,
                        int B, int P, float* d_delta, int* d_failed, void* stream);

/*
 * LMOptimizer.calculate_residuals (geocalib/lm_optimizer.py:248-274) as one launch: per-pixel residuals of the
 * fields against the prediction of (d_cam (B,8), d_grav (B,3)):  d_r_up (B,H*W,2) = up_data - up(theta),
 * d_r_lat (B,H*W,1) = sin(lat_data) - sin(lat(theta)).  Either output may be NULL (then its field may be NULL).
 */
int gclm_residual_fields(int camera_model, const float* d_up, const float* d_lat, const float* d_cam,
                         const float* d_grav, int B, int H, int W, float* d_r_up, float* d_r_lat, void* stream);

/*
 * LMOptimizer.calculate_costs (geocalib/lm_optimizer.py:276-315) for one residual tensor: n rows of `dim`
 * components (dim = 0: the n inputs already are |r|^2) -> scaled Huber cost and weight at `scale` (scaled_loss
 * :61-76, huber_loss :79-87), both multiplied by d_conf (n) when given, and the loss's second derivative.
 * d_cost / d_weight / d_second (n) may each be NULL.
 */
int gclm_huber_costs(const float* d_residual, size_t n, int dim, float scale, const float* d_conf, float* d_cost,
                     float* d_weight, float* d_second, void* stream);

/*
 * The reference's J_perspective_field (geocalib/perspective_fields.py:323-365 -> J_up_field :84-182,
 * J_latitude_field :214-275) as one launch: per-pixel Jacobians of the PREDICTED up / latitude fields of B cameras
 * wrt (delta_1, delta_2, focal[, k1[, k2]]).  d_cam (B,8), d_grav (B,3) as in gclm_solve; spherical / log_focal
 * select the tangent basis (SphericalManifold.J_plus vs Gravity.J_rp) and the focal parametrisation exactly as
 * the reference's flags.  Outputs (either may be NULL): d_J_up (B,H,W,2,P), d_J_lat (B,H,W,1,P) with
 * P = 3 + number of distortion parameters of `camera_model`.  A solve never materialises these (the sweep
 * contracts them in registers); this entry point exists for callers and tests that want the fields themselves.
 */
int gclm_jacobian_fields(int camera_model, const float* d_cam, const float* d_grav, int B, int H, int W,
                         int spherical, int log_focal, float* d_J_up, float* d_J_lat, void* stream);

/*
 * Measurement helper (bench.py, tests): synthetic perspective fields generated on the device,
 * SURVEY.md section 8(d).  Image i depends on (seed, first_index + i) only, so every sharding of
 * a batch sees identical data.  Writes the 5 planes and the ground truth (B,8) / (B,3).
 */
int gclm_synth_fields(int camera_model, uint64_t seed, int64_t first_index, int B, int H, int W,
                      float noise_sigma, float* d_up, float* d_lat, float* d_up_conf,
                      float* d_lat_conf, float* d_gt_cam, float* d_gt_grav, void* stream);

/* Same generator with shared-intrinsics structure and strided index runs (multi-GPU frame split):
 * intrinsics (focal, k1) are keyed by global_index / group_size (group_size <= 1: per image), and local
 * image b maps to global index first_index + (b / run) * run_stride + b % run (run = 0: contiguous). */
int gclm_synth_fields_grouped(int camera_model, uint64_t seed, int64_t first_index, int B, int H, int W,
                              float noise_sigma, int group_size, int run, int run_stride, float* d_up,
                              float* d_lat, float* d_up_conf, float* d_lat_conf, float* d_gt_cam,
                              float* d_gt_grav, void* stream);

/*
 * Measurement helper (bench.py: roofline.read_ceiling_frac; no reference counterpart): ONE launch that streams n_planes
 * (<= 8) device planes of `floats` floats each (16-byte aligned, floats % 4 == 0) with the sweep's own load -- non-temporal,
 * 16 bytes per lane, consecutive lanes on consecutive addresses, four loads per plane in flight per thread -- and no
 * arithmetic beyond the sum that keeps the loads alive.  Its duration is what THESE buffers can stream on THIS box: the
 * memory-system ceiling of the sweep's access pattern for the allocation a measurement was taken on.  d_planes is a HOST
 * array of device pointers.  Asynchronous on `stream`; the caller times it (events on that stream).
 */
int gclm_read_probe(const float* const* d_planes, int n_planes, size_t floats, void* stream);

/*
 * Multi-GPU collectives over RCCL / xGMI, one process per GPU (no reference counterpart: the
 * reference's LM is single-process).  gclm_comm_unique_id is called on rank 0 only and its 128 bytes are handed to
 * the other ranks by the caller.  Both collectives are asynchronous on `stream`.
 *   gclm_comm_all_gather      the ONE gather of packed result rows (configs[2]); recv holds nranks * count floats
 *   gclm_comm_all_reduce_sum  the ONE in-place sum of the (num_groups x GCLM_SHARED_PARTIAL_STRIDE) Schur partials
 *                             per LM step between gclm_shared_reduce and gclm_shared_apply (configs[4])
 */
#define GCLM_COMM_ID_BYTES 128
typedef struct gclm_comm gclm_comm;
int gclm_comm_unique_id(void* id_out /* GCLM_COMM_ID_BYTES */);
int gclm_comm_create(gclm_comm** out, const void* unique_id, int nranks, int rank, int device);
int gclm_comm_destroy(gclm_comm* c);
/* Message of the last failing call on this communicator; c == NULL: of the last failing gclm_comm_unique_id /
 * gclm_comm_create of the CALLING THREAD (thread-local, as gclm_last_error(NULL)). */
const char* gclm_comm_last_error(const gclm_comm* c);
/* NCCL_VERSION_CODE of the rccl.h this library was compiled against and ncclGetVersion() of the librccl it bound at
 * run time.  The library is not LINKED against librccl: on first use it takes the RCCL the process has already loaded
 * (a torch process: torch's own librccl.so, the one torch.distributed's communicators live in), otherwise
 * /opt/rocm/lib/librccl.so.1.  gclm_comm_create refuses (-21) a run-time library of another MAJOR version, and every
 * gclm_comm_* entry point fails with -22 when no librccl can be loaded at all (gclm_comm_last_error(NULL) then carries the
 * loader's message or the name of the missing entry point).  The choice is made by the first gclm_comm_unique_id /
 * gclm_comm_create of the process and holds for its lifetime: a process that also uses torch.distributed MUST import
 * torch before that call.  gclm_comm_versions itself never makes the choice: before one was made it reports the version
 * of the librccl already loaded in the process (what a binding now would pick), or runtime = 0 when none is loaded yet. */
int gclm_comm_versions(int* compiled, int* runtime);
int gclm_comm_all_gather(gclm_comm* c, const float* d_send, float* d_recv, size_t count_per_rank, void* stream);
int gclm_comm_all_reduce_sum(gclm_comm* c, float* d_buf, size_t count, void* stream);
int gclm_comm_all_reduce_sum_i32(gclm_comm* c, int32_t* d_buf, size_t count, void* stream);

/*
 * The reference's early stop is ONE decision over the whole batch (lm_optimizer.py:90-92, 619-625).  For a batch
 * sharded over the ranks of `c` (image sharding, configs[2]) that decision needs every rank's images: with a stop
 * communicator set, gclm_solve / gclm_calibrate sum the per-step counter of images whose cost still moved over the
 * ranks after every update (ONE 4-byte all-reduce per LM step, enqueued on the solve's stream, no host round trip), so
 * every rank stops at the step the single-process solve of the whole batch would stop at, and reports that `stop_at`.
 * Every rank must run the same num_steps.  NULL unsets it.  Without it a sharded solve must run with early_stop = 0.
 */
int gclm_set_stop_comm(gclm_handle* h, gclm_comm* c);

/* A batch of independent images solved as several PARTS by several handles (e.g. on several streams of one device, so
 * that one part's update launches run under another part's sweep; early_stop = 0): every output of a part is what the
 * single call would have produced for those images, except infos["stop_at"] -- the reference's "first step after which
 * EVERY image's cost was close" (lm_optimizer.py:619-620) is one number for the whole batch.  It re-derives stop_at from
 * the SUM of the parts' per-step counters and writes it into every row of every part's info.  At most 8 parts, same
 * device / num_steps.  The counters live in each handle's workspace and are those of the handle's LAST solve, so:
 *   - B[p] MUST be the batch size of part p's last gclm_solve / gclm_calibrate (checked: -2 otherwise); a part with
 *     B[p] == 0 is skipped (it holds no image: its handle need not have solved anything);
 *   - `stream` MUST be ordered after every part's solve (events / stream waits are the caller's business), and no part
 *     handle may start another solve before the merge kernel has run (it would overwrite the counters being summed). */
int gclm_merge_stop_at(gclm_handle* const* parts, float* const* d_info, const int* B, int n_parts, void* stream);

/* Tuning / test hook (no reference counterpart): loop iterations per workgroup of the sweep (how an image is cut
 * into partial records; only the summation order depends on it).  0 restores the built-in choice (20, fewer for
 * small batches).  Replaces the GCLM_SWEEP_ITERS environment variable of rounds 1-2: the solve reads no environment. */
int gclm_set_sweep_iters(gclm_handle* h, int iters);
/* How this handle's next solve of B images of (H, W) would cut an image into workgroup chunks: rows per chunk and
 * chunks per image (`aligned16`: all field pointers 16-byte aligned).  The cut fixes the summation order of an image's
 * partial records, so two calls agree bit for bit on an image exactly when their cuts agree; a caller that solves ONE
 * batch in several parts (gclm_merge_stop_at; LMOptimizer.overlap_streams) asks here instead of mirroring the rule.  The
 * cut depends on the batch size only below 2048 chunks per call (small batches take fewer rows per chunk to fill the
 * GPU).  Either output pointer may be NULL.  No device work, no error message: -1 for a NULL handle, -3 for a
 * non-positive B, H or W. */
int gclm_plan_cut(const gclm_handle* h, int B, int H, int W, int aligned16, int* rows_per_chunk, int* chunks_per_image);

/* sin(latitude_field) (lm_optimizer.py:262,270) does not depend on the parameters, yet each of the num_steps + 1 sweeps of
 * a solve would re-evaluate it per pixel.  For the VALU-bound distortion models the first sweep of a solve therefore
 * stores it in a LIBRARY-owned scratch plane (B x H x W floats inside the handle's workspace, grown on demand like the
 * rest of it) and every later sweep reads that plane instead of `latitude_field`: same bytes read per sweep, one extra
 * plane written per solve (1 / 105 of its traffic), same polynomial hence bit-identical results; the caller's boundary
 * (float32 radians) does not change.
 * mode -1 (default): the library decides (simple_radial / radial / simple_divisional with all five planes on the
 * 16-byte-aligned path, at least one LM step, not the one-launch-per-step path; never pinhole, which is memory-bound);
 * 0: never (saves B x H x W x 4 bytes of workspace); 1: wherever the sweep has the instantiation (also pinhole).
 *
 * The plane is OPTIONAL in every mode.  It is an allocation of its own, of exactly B x H x W x 4 bytes (no headroom; 1.26 GB
 * for 1024 images of 640x480, against 3 MB for the rest of the workspace), and a solve that cannot have it runs the sweeps
 * that compute sin(latitude) themselves -- the same bits, a few per cent slower -- instead of failing: when hipMalloc
 * fails, or when the plane would exceed the limit.  gclm_set_slat_plane_limit: max_bytes = 0 (default) = the built-in rule,
 * at most HALF of the device memory that is free when the plane is (re)allocated; otherwise a plane is only (re)allocated
 * while B x H x W x 4 <= max_bytes (1 = in effect never; a plane the handle already holds keeps serving the solves it is
 * large enough for -- gclm_release_workspace drops it).  A refused size is remembered until the limit or the mode is set again,
 * so a serving loop does not pay a failing allocation per call.  Only the core workspace failing to allocate is an error
 * (-10).  A batch solved as n parts by n handles (LMOptimizer.overlap_streams) holds n planes of B / n images each: one
 * plane's worth in total. */
int gclm_set_slat_plane(gclm_handle* h, int mode);
int gclm_set_slat_plane_limit(gclm_handle* h, size_t max_bytes);

/* Row pairs (radial / simple_divisional).  Everything a radial camera model adds to the per-pixel work depends on
 * r2 = u^2 + v^2 only.  The principal point of every camera the library initialises is the image centre (camera.py:136-152),
 * so rows y and H - y of a column share r2 bit for bit; a lane of the sweep then takes row H - y along with row y and
 * evaluates the radial terms once for both (gclm_pass.hip: row_math_mirror; simple_divisional -14 % per sweep, radial -4 %).  The
 * per-pixel values are those of the one-row walk bit for bit; the order in which a lane adds its pixels differs, so results
 * agree with it to float32 summation order (~1e-7), not bit for bit.  An image whose principal point is NOT the centre
 * (an explicit camera handed to gclm_solve) is detected per image on the device and walks the same pairs without sharing.
 * mode -1 (default): the library decides (the models it pays for, all five planes on the 16-byte-aligned path, an even
 * number of rows, and more workgroups per launch than the one-launch-per-step path takes -- 768: below that a step is
 * latency-bound, and its two-launch form stays bit-identical to the one-launch form); 0: never (the one-row walk of every
 * earlier ABI, bit for bit); 1: wherever the sweep has the instantiation (also small launches, unless
 * gclm_set_fused_steps(h, 1) asks for one launch per step as well). */
int gclm_set_row_pairs(gclm_handle* h, int mode);

/* Small batches (the interactive single-image calibration of the reference's demo, interactive_demo.py:403) run ONE
 * launch per LM step: the per-image update of step k-1 is done in the prologue of every workgroup of sweep k
 * (num_steps + 2 launches per solve instead of 2 num_steps + 4 -- the first launch also builds the initial estimate; results
 * bit-identical to the two-launch sequence).
 * mode -1 (default): the library decides (few workgroups in flight); 0: never; 1: whenever it is valid (independent
 * intrinsics, 16-byte aligned fields of a width divisible by 4, and a single image or early_stop = 0 -- the
 * batch-global stop of lm_optimizer.py:619-625 is only decidable inside a launch when the batch is one image). */
int gclm_set_fused_steps(gclm_handle* h, int mode);

/* Single image with early_stop = 1 on the one-launch-per-step path: the launches after the stop return at their first
 * instruction, but each still takes its turn on the queue (21 of the 32 launches of a default-conf solve that stops at
 * step 9).  depth > 0 PACES the launches: launch k is issued once launch k - depth has reported to a word in
 * host-mapped memory, and none after the stop has been reported -- gclm_solve / gclm_calibrate then BLOCK the calling
 * thread for about the duration of the LM loop (the results are still produced asynchronously on the stream).  For
 * callers that read the result right away (the reference's GeoCalib.calibrate does: extractor.py:51-69).  0 (default)
 * = off: every launch is issued at once, the call returns immediately.  Ignored where it cannot help (more than one
 * image, early_stop = 0, the two-launch path).  Results do not depend on it.  depth in [0, 16]; 3 is a good value.
 * The host wait spins briefly, then yields between polls; if a report does not arrive within 2 ms (the queue is shared
 * with other streams) the rest of that solve is issued unpaced and the handle does not pace its next 64 solves. */
int gclm_set_paced_launches(gclm_handle* h, int depth);

/* Timing helper: when enabled, every sweep launch is bracketed by HIP events on the solve's stream;
 * gclm_last_pass_timing waits for the recorded launches, returns their count and summed duration
 * since the previous read, and resets the record.  Returns <0 if timing was not enabled. */
int gclm_set_timing(gclm_handle* h, int enabled);
int gclm_last_pass_timing(gclm_handle* h, int* n_launches, float* total_ms);

#ifdef __cplusplus
}
#endif
#endif /* GCLM_H */
