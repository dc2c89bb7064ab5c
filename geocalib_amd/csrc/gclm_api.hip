// gclm_api.hip -- C ABI of libgeocalib_hip.so (include/gclm.h) and the launch sequence of a solve.
//
// A solve is 2*num_steps+4 asynchronous launches on the caller's stream and no host round trip (nor memset):
//   init | { sweep(theta_i) ; update_i } x num_steps | prep_final ; sweep(theta_final, rpf) ; finalize
// and num_steps+2 for a single image (one launch per LM step, use_fused; the first launch builds theta_0 itself):
//   fused(step) x num_steps | fused(final) | finalize
// (the reference syncs twice per step: H,G -> CPU Cholesky -> device, and torch.allclose).
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "gclm_internal.h"

using namespace gclm;

#ifndef GCLM_FUSED_MAX_WORKGROUPS
#define GCLM_FUSED_MAX_WORKGROUPS 768      // B * workgroups-per-image up to which an LM step is ONE launch (see use_fused):
                                           // measured with round 4's prologue (profiles/archive/r04_fused_threshold.log, 20 fixed steps) --
                                           // 640x480 (150 workgroups per image): B = 2 -19 %, 4 -15 %, 6 +-0, 8 +6 %, 12 +15 %;
                                           // 320x240 (38 per image): B = 2 ... 12 -20 ... -27 %
#endif
#ifndef GCLM_ROW_PAIRS_DEFAULT
#define GCLM_ROW_PAIRS_DEFAULT -1   // a handle's initial gclm_set_row_pairs mode (1: test builds that run the whole suite on row pairs)
#endif
#ifndef GCLM_ISO_FINAL
#define GCLM_ISO_FINAL 1      // A/B switch: 0 = the final sweep always takes the general focal column
#endif

struct gclm_handle {
    gclm_config cfg;
    int device = 0;
    std::string err;
    // device scratch (owned)
    void* ws = nullptr;
    size_t ws_bytes = 0;
    // carved views
    SolveCtx ctx{};
    float* group_partials = nullptr;
    // split shared-intrinsics session
    struct {
        bool active = false;
        const float *up = nullptr, *lat = nullptr, *upc = nullptr, *latc = nullptr;
        float *cam_io = nullptr, *grav_io = nullptr;
        Geometry geo{};
        bool slat_ready = false;    // the scratch plane holds sin(latitude) of this session's fields
    } sh;
    // The sin(latitude) scratch plane is an allocation of its own, of exactly the size asked for (no headroom), and the only
    // part of the workspace a solve can do without: see ensure_slat.
    float* slat_buf = nullptr;      // owned: (slat_bytes / 4) floats, or null
    size_t slat_bytes = 0;
    size_t slat_refused = 0;        // smallest plane size (bytes) whose allocation failed or was refused by the limit since the
                                    // last change of the limit / mode: such a size is not tried again on every solve
    size_t slat_limit = 0;          // gclm_set_slat_plane_limit: 0 = built-in rule (half of the free device memory), else bytes
    float* slat = nullptr;          // the plane of the CURRENT solve / session (== slat_buf), or null: sweeps compute sin(latitude)
    int slat_plane = -1;            // gclm_set_slat_plane: -1 = built-in choice, 0 = never, 1 = wherever the sweep has it
    int row_pairs = GCLM_ROW_PAIRS_DEFAULT;   // gclm_set_row_pairs: -1 = built-in choice, 0 = never, 1 = wherever the sweep has the row-pair walk
    int sweep_iters = 0;            // gclm_set_sweep_iters: 0 = built-in choice
    int fused_mode = -1;            // gclm_set_fused_steps: -1 = built-in choice, 0 = never, 1 = whenever it is valid
    gclm_comm* stop_comm = nullptr; // gclm_set_stop_comm: the batch-global early stop spans the ranks of this communicator
    int paced_depth = 0;            // gclm_set_paced_launches: 0 = off, d = launch k waits for launch k-d's report
    unsigned* progress_host = nullptr;   // ... the host-mapped word the launches report to (owned), its device alias
    unsigned* progress_dev = nullptr;
    unsigned epoch = 0;             // ... and the tag of the current solve in it
    int paced_cooldown = 0;         // solves left to run UNPACED after a wait timed out (the queue was not ours)
    // optional timing of the sweep launches
    bool timing = false;
    std::vector<hipEvent_t> ev;
    int ev_used = 0;
};

namespace {

thread_local std::string g_create_error;

int fail(gclm_handle* h, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (h) h->err = buf; else g_create_error = buf;
    return code;
}

#define GCLM_HIP(h, expr)                                                                   \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess) return fail(h, -10, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

// A caller built against another include/gclm.h must be told so, not trusted with a struct of another layout.
const char* abi_mismatch(const gclm_config* c, char (&buf)[256]) {
    if (c->struct_size == (int32_t)sizeof(gclm_config) && c->abi_version == GCLM_VERSION) return nullptr;
    snprintf(buf, sizeof(buf), "gclm_config ABI mismatch: caller passes struct_size %d / abi_version %d, this library is "
             "sizeof(gclm_config) %d / GCLM_VERSION %d (rebuild the caller against this include/gclm.h and fill the "
             "struct with gclm_default_config)", (int)c->struct_size, (int)c->abi_version, (int)sizeof(gclm_config), GCLM_VERSION);
    return buf;
}

const char* validate(const gclm_config& c) {
    if (c.camera_model < GCLM_PINHOLE || c.camera_model > GCLM_SIMPLE_DIVISIONAL)
        return "camera_model: unknown (0 pinhole, 1 simple_radial, 2 radial, 3 simple_divisional)";
    if (c.num_steps < 0 || c.num_steps > GCLM_MAX_STEPS) return "num_steps out of range [0, GCLM_MAX_STEPS]";
    if (!(c.up_loss_fn_scale > 0.f) || !(c.lat_loss_fn_scale > 0.f)) return "loss scales must be > 0";
    if (c.group_size < 0) return "group_size must be >= 0";
    if (c.shared_intrinsics && !(c.estimate_gravity && c.estimate_focal))
        return "shared_intrinsics requires gravity and focal to be estimated (lm_optimizer.py:350-383)";
    if (c.shared_intrinsics && c.camera_model != GCLM_PINHOLE && !c.estimate_dist)
        return "shared_intrinsics with a distortion prior is not supported";
    const int n = 2 * (c.estimate_gravity != 0) + (c.estimate_focal != 0) + num_dist_params(c.camera_model);
    if (n == 0) return "No parameters to optimize";
    return nullptr;
}

// Every entry point runs on the handle's device and leaves the CALLER's current device as it found it (a
// single-process multi-GPU program would otherwise see its default device flip under PyTorch, also at garbage-
// collection time through gclm_destroy).
struct DeviceGuard {
    int prev = -1;
    hipError_t status = hipSuccess;
    explicit DeviceGuard(int device) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != device) status = hipSetDevice(device); else prev = -1;    // prev = -1: nothing to restore
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

bool is_aligned16(const void* p) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Carve the core workspace for B images / nchunks partial records per image / G groups.  Growing it frees and re-allocates
// (hipFree synchronises the device ONCE, on the first call of a larger shape than any before; every later call finds the
// workspace in place -- include/gclm.h says so).  A failure here is the only allocation failure a solve reports (-10).
int ensure_workspace(gclm_handle* h, int B, int nchunks, int G) {
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    const size_t o_state0 = take(sizeof(State) * B), o_state1 = take(sizeof(State) * B);
    const size_t o_pb0 = take(sizeof(PBlock) * B), o_pb1 = take(sizeof(PBlock) * B), o_pbf = take(sizeof(PBlock) * B);
    const size_t o_part = take(sizeof(float) * kNAccMax * (size_t)B * nchunks);
    const size_t o_part2 = take(sizeof(float) * kNAccMax * (size_t)B * nchunks);     // double buffer of the fused path
    const size_t o_fsys = take(sizeof(float) * kNAccMax * (size_t)B);
    const size_t o_gp = take(sizeof(float) * GCLM_SHARED_PARTIAL_STRIDE * (size_t)(G > 0 ? G : 1));
    const size_t o_ctrl = take(sizeof(Ctrl));
    if (off > h->ws_bytes) {
        if (h->ws) GCLM_HIP(h, hipFree(h->ws));
        h->ws = nullptr;
        h->ws_bytes = 0;
        const size_t want = off + off / 4;      // headroom: no reallocation for slightly larger calls (small records only:
                                                // 3 MB at 1024 images of 640x480; the scratch plane is NOT in here)
        GCLM_HIP(h, hipMalloc(&h->ws, want));
        h->ws_bytes = want;
    }
    char* base = static_cast<char*>(h->ws);
    SolveCtx& c = h->ctx;
    c.state[0] = reinterpret_cast<State*>(base + o_state0);
    c.state[1] = reinterpret_cast<State*>(base + o_state1);
    c.pb[0] = reinterpret_cast<PBlock*>(base + o_pb0);
    c.pb[1] = reinterpret_cast<PBlock*>(base + o_pb1);
    c.pb_final = reinterpret_cast<PBlock*>(base + o_pbf);
    c.partials = reinterpret_cast<float*>(base + o_part);
    c.partials2 = reinterpret_cast<float*>(base + o_part2);
    c.frame_sys = reinterpret_cast<float*>(base + o_fsys);
    c.ctrl = reinterpret_cast<Ctrl*>(base + o_ctrl);
    h->group_partials = reinterpret_cast<float*>(base + o_gp);
    return 0;
}

// The sin(latitude) scratch plane of a solve of `floats` pixels (0: none wanted): h->slat = the plane, or null when the solve
// runs without it -- never an error.  The plane only saves arithmetic (gclm_pass.hip: row_math, SLAT); a sweep that computes
// sin(latitude) itself produces the same bits, so whenever the plane cannot be had -- hipMalloc fails, or the plane is larger
// than the limit (gclm_set_slat_plane_limit; built-in: half of the device memory that is free once the old plane is
// released) -- the solve goes on without it, as the library did before it had the plane.  Exact size, no headroom: the plane
// is 400x the rest of the workspace.  A size that was refused is remembered (slat_refused) so that a serving loop does not
// pay a failing hipMalloc (and its device synchronisation) per call.
void ensure_slat(gclm_handle* h, size_t floats) {
    h->slat = nullptr;
    if (floats == 0) return;
    const size_t bytes = floats * sizeof(float);
    if (bytes <= h->slat_bytes) { h->slat = h->slat_buf; return; }
    if (h->slat_refused && bytes >= h->slat_refused) return;
    if (h->slat_limit && bytes > h->slat_limit) { h->slat_refused = bytes; return; }     // (the old, smaller plane stays)
    if (!h->slat_limit) {                     // built-in rule: half of what is free once the old plane is released
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || bytes > (free_b + h->slat_bytes) / 2) {
            (void)hipGetLastError();
            h->slat_refused = bytes;          // (the old, smaller plane stays here too)
            return;
        }
    }
    if (h->slat_buf) (void)hipFree(h->slat_buf);
    h->slat_buf = nullptr;
    h->slat_bytes = 0;
    void* p = nullptr;
    const bool ok = hipMalloc(&p, bytes) == hipSuccess && p != nullptr;
    if (!ok) {
        (void)hipGetLastError();          // the failure is handled here: it must not surface in the next launch's status
        h->slat_refused = bytes;
        return;
    }
    h->slat_buf = static_cast<float*>(p);
    h->slat_bytes = bytes;
    h->slat = h->slat_buf;
}

// Does this solve keep sin(latitude_field) in a scratch plane (gclm_pass.hip: row_math, SLAT)?  Built-in choice: the
// VALU-bound distortion models, on the five-plane float4 sweep, whenever at least one loop sweep precedes the final one.
// Pinhole never (memory-bound: the plane's extra write costs what the saved arithmetic gains); not the one-launch-per-step
// path (latency-bound, a handful of workgroups).
bool slat_wanted(const gclm_handle* h, const float* up, const float* upc, const float* latc, const Geometry& g, bool fused) {
    if (h->slat_plane == 0 || fused || g.vec != 4 || !up || !upc || !latc || h->cfg.num_steps < 1) return false;
    if (h->slat_plane < 0 && h->cfg.camera_model == GCLM_PINHOLE) return false;
    return sweep_has_slat_plane(h->cfg.camera_model);
}

// The sweep arguments of the next sweep of a solve that keeps the scratch plane: the first one fills it, the others read it
void apply_slat(const gclm_handle* h, SweepArgs& a, bool& ready) {
    if (!h->slat) return;
    a.slat = h->slat;
    if (ready) { a.slat_mode = 2; a.lat = h->slat; }
    else { a.slat_mode = 1; ready = true; }
}

// Does a sweep over these planes take the row-pair walk (gclm_pass.hip: row_math_mirror)?  radial / simple_divisional on the
// five-plane float4 sweep over an even number of rows; never the one-launch-per-step path (its geometry, its prefetch).
bool mirror_wanted(const gclm_handle* h, bool five_planes, const Geometry& g, int H) {
    if (h->row_pairs == 0 || g.vec != 4 || !five_planes || (H & 1) || H < 4) return false;
    return h->row_pairs > 0 ? sweep_has_mirror(h->cfg.camera_model) : sweep_mirror_builtin(h->cfg.camera_model);
}

bool use_fused(const gclm_handle* h, int B, const Geometry& g);

// The geometry of this call's sweeps: the one-row walk, or -- where the sweep has it and the step is not one launch -- row pairs
Geometry plan_sweeps(const gclm_handle* h, int B, int H, int W, bool aligned16, bool five_planes) {
    const Geometry g = plan_geometry(B, H, W, aligned16, h->sweep_iters, h->cfg.camera_model);
    if (!mirror_wanted(h, five_planes, g, H)) return g;
    if (h->row_pairs < 0) {
        // built-in choice: only launches too large for one launch per step, WHETHER OR NOT that path is valid for the call
        // (early stop over several images, gclm_set_fused_steps(h, 0)): below the threshold a step is latency-bound, and the
        // two-launch sequence stays the one-launch path's twin bit for bit
        if ((long long)B * g.nchunks <= GCLM_FUSED_MAX_WORKGROUPS) return g;
    } else if (h->fused_mode == 1 && use_fused(h, B, g)) {
        return g;                    // asked for both: the step that can be ONE launch stays one launch
    }
    return plan_geometry(B, H, W, aligned16, h->sweep_iters, h->cfg.camera_model, true);
}

SweepArgs sweep_args(const gclm_handle* h, const float* up, const float* lat, const float* upc, const float* latc,
                     const PBlock* pb, const Geometry& g, bool loop_params, int stop_step) {
    SweepArgs a{};
    a.up = up; a.lat = lat; a.upc = up ? upc : nullptr; a.latc = latc;
    a.pb = pb; a.ctrl = h->ctx.ctrl; a.partials = h->ctx.partials;
    a.B = h->ctx.B; a.H = h->ctx.H; a.W = h->ctx.W;
    a.nchunks = g.nchunks; a.vec = g.vec;
    a.wu = g.wu; a.cu = g.cu; a.nstrips = g.nstrips; a.rpi = g.rpi; a.rows_per_block = g.rows_per_block; a.wpt = g.wpt; a.jobs = g.jobs;
    a.hrows = g.hrows; a.mirror = g.mirror;
    // loop sweeps: the configured parametrisation; final sweep: the (roll, pitch, focal) block, in its log-focal form when
    // every image is known to have fx == fy (iso_final: finalize_kernel rescales the focal column)
    a.log_focal = loop_params ? h->cfg.use_log_focal : h->ctx.iso_final;
    a.stop_step = stop_step;
    a.up_scale = h->cfg.up_loss_fn_scale; a.lat_scale = h->cfg.lat_loss_fn_scale;
    return a;
}

int timed_sweep(gclm_handle* h, const SweepArgs& a, hipStream_t s, const FusedArgs* fused = nullptr) {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const bool timing = h->timing && h->ev_used < 2 * 8192;   // bounded: stop recording silently
    if (timing) {
        while ((int)h->ev.size() < h->ev_used + 2) {
            hipEvent_t e;
            GCLM_HIP(h, hipEventCreate(&e));
            h->ev.push_back(e);
        }
        e0 = h->ev[h->ev_used];
        e1 = h->ev[h->ev_used + 1];
        h->ev_used += 2;
        GCLM_HIP(h, hipEventRecord(e0, s));
    }
    if (fused) GCLM_HIP(h, launch_fused_step(h->cfg.camera_model, a, *fused, s));
    else GCLM_HIP(h, launch_sweep(h->cfg.camera_model, a, s));
    if (timing) GCLM_HIP(h, hipEventRecord(e1, s));
    return 0;
}

// One launch per LM step (gclm_pass.hip: fused_step_kernel) pays when a step is bound by its two dependent launches
// rather than by the sweep: few workgroups in flight.  It is VALID when the early-stop decision is local to a
// workgroup (one image) or off, for independent intrinsics on the float4 path.
bool use_fused(const gclm_handle* h, int B, const Geometry& g) {
    const gclm_config& c = h->cfg;
    const bool valid = !c.shared_intrinsics && g.vec == 4 && (B == 1 || !c.early_stop) && B > 0 &&
                       !(c.early_stop && h->stop_comm);       // a stop that spans ranks is never local to a workgroup
    if (!valid || h->fused_mode == 0) return false;
    if (h->fused_mode == 1) return true;
    return (long long)B * g.nchunks <= GCLM_FUSED_MAX_WORKGROUPS;
}

int check_shapes(gclm_handle* h, const float* lat, int B, int H, int W) {
    // (an EMPTY batch has no fields: a zero-size device tensor's pointer is NULL -- run_solve's B == 0 branch)
    if (!lat && B != 0) return fail(h, -3, "latitude_field is required (lm_optimizer.py:31 raises KeyError without it)");
    if (B < 0 || H <= 0 || W <= 0) return fail(h, -3, "bad shape B=%d H=%d W=%d", B, H, W);
    if (B > 65535) return fail(h, -3, "batch %d exceeds 65535 images per call (grid.y): split the batch", B);
    if ((size_t)H * W >= (size_t)1 << 30) return fail(h, -3, "image too large");
    return 0;
}

int setup_groups(gclm_handle* h, int B) {
    SolveCtx& c = h->ctx;
    c.group_of_frame = nullptr;
    if (!h->cfg.shared_intrinsics) { c.n_groups = 0; c.group_size = 1; return 0; }
    const int gs = h->cfg.group_size > 0 ? h->cfg.group_size : (B > 0 ? B : 1);
    if (B % gs != 0) return fail(h, -3, "batch %d is not a multiple of group_size %d", B, gs);
    c.group_size = gs;
    c.n_groups = B / gs;
    return 0;
}

}  // namespace

namespace gclm {
// Cut one sweep into column-stationary tiles and wave jobs (gclm_pass.hip: sweep_kernel).  A row holds `wu` units
// (float4 groups, or pixels in the scalar path).  Up to 512 units per row the strip is the whole row and a tile takes
// `rpi` rows per iteration with rpi * wu lanes rounded up to whole waves -- rpi is chosen for the fewest idle lanes,
// then for the smallest tile (640 px: 160 units, rpi = 2, 320 lanes = 5 waves, no idle lane).  Wider rows are cut
// into strips of a multiple of 64 units, one row per iteration.  Every wave of a tile is a job; a workgroup is four
// consecutive jobs of an image.
Geometry plan_geometry(int B, int H, int W, bool aligned16, int sweep_iters, int camera_model, bool mirror) {
    Geometry g;
    g.vec = (aligned16 && (W % 4 == 0)) ? 4 : 1;
    // row pairs (gclm_pass.hip: row_math_mirror): the tiles walk rows [0, H / 2), every lane takes row H - y along with row y
    g.mirror = (mirror && g.vec == 4 && H % 2 == 0 && H >= 4) ? 1 : 0;
    g.hrows = g.mirror ? H / 2 : H;
    const int rows = g.hrows;
    g.wu = W / g.vec;
    auto waste = [](int used, int lanes) { return (int)(50.0 * (lanes - used) / lanes); };   // idle lanes, steps of 2 %
    if (g.wu <= kMaxTile) {
        g.nstrips = 1;
        g.cu = g.wu;
        g.rpi = 1;
        int best = 1 << 30;
        for (int r = 1; r * g.wu <= kMaxTile; ++r) {
            const int lanes = (r * g.wu + 63) / 64 * 64;
            const int score = waste(r * g.wu, lanes) * 1024 + lanes / 64;
            if (score < best) { best = score; g.rpi = r; }
        }
        g.wpt = (g.rpi * g.wu + 63) / 64;
    } else {
        g.rpi = 1;
        g.cu = 256;
        int best = 1 << 30;
        for (int cu = 128; cu <= kMaxTile; cu += 64) {
            const int ns = (g.wu + cu - 1) / cu;
            const int score = waste(g.wu, ns * cu) * 1024 + cu / 64;
            if (score < best) { best = score; g.cu = cu; }
        }
        g.nstrips = (g.wu + g.cu - 1) / g.cu;
        g.wpt = g.cu / 64;
    }
    // ~20 loop iterations per lane amortise the 16-value workgroup reduction; fewer when the batch alone cannot
    // fill 256 CUs x 4+ workgroups.
    // The distortion models are VALU-bound: their per-workgroup prologue / epilogue is worth amortising over
    // 30 iterations where the image divides into whole blocks of that many rows (640x480: 8 blocks of 60 rows; same-allocation
    // A/B profiles/archive/r04_variant_huber_clamp.log: simple_radial -0.3 %, radial -1.2 %, simple_divisional -0.7 %, pinhole +0.4 %).
    // Row pairs: an iteration is two rows of the image -- half the iterations for the same work per workgroup.
    int builtin = g.mirror ? 10 : 20;
    if (camera_model != GCLM_PINHOLE && rows % (g.rpi * (g.mirror ? 15 : 30)) == 0) builtin = g.mirror ? 15 : 30;
    int iters = (sweep_iters >= 1 && sweep_iters <= 4096) ? sweep_iters : builtin;   // gclm_set_sweep_iters (tuning / tests)
    auto jobs = [&](int it) { return g.nstrips * g.wpt * ((rows + g.rpi * it - 1) / (g.rpi * it)); };
    auto chunks = [&](int it) { return (jobs(it) + kBlock / 64 - 1) / (kBlock / 64); };
    while (iters > 2 && (long long)B * chunks(iters) < 2048) iters = iters > 5 ? iters / 2 : iters - 1;
    g.rows_per_block = g.rpi * iters;
    g.jobs = jobs(iters);
    g.nchunks = chunks(iters);
    return g;
}
}  // namespace gclm

extern "C" {

int gclm_version(void) { return GCLM_VERSION; }

int gclm_abi_config_size(void) { return (int)sizeof(gclm_config); }

int gclm_default_config(gclm_config* cfg) {
    if (!cfg) return -1;
    std::memset(cfg, 0, sizeof(*cfg));
    cfg->struct_size = (int32_t)sizeof(gclm_config);
    cfg->abi_version = GCLM_VERSION;
    cfg->device = 0;
    cfg->camera_model = GCLM_PINHOLE;
    cfg->num_steps = 30;
    cfg->lambda0 = 0.1f;
    cfg->early_stop = 1;
    cfg->atol = 1e-8f;
    cfg->rtol = 1e-8f;
    cfg->use_spherical_manifold = 1;
    cfg->use_log_focal = 1;
    cfg->up_loss_fn_scale = 1e-2f;
    cfg->lat_loss_fn_scale = 1e-2f;
    cfg->estimate_gravity = cfg->estimate_focal = cfg->estimate_dist = 1;
    cfg->compute_uncertainty = 1;
    return 0;
}

int gclm_create(gclm_handle** out, const gclm_config* cfg) {
    if (!out || !cfg) return fail(nullptr, -1, "gclm_create: null argument");
    *out = nullptr;
    char abuf[256];
    if (const char* msg = abi_mismatch(cfg, abuf)) return fail(nullptr, -5, "gclm_create: %s", msg);
    if (const char* msg = validate(*cfg)) return fail(nullptr, -2, "gclm_create: %s", msg);
    const int device = cfg->device;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, -11, "gclm_create: no HIP device visible (the HIP path has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(nullptr, -2, "gclm_create: device %d out of range (%d devices)", device, ndev);
    gclm_handle* h = new (std::nothrow) gclm_handle();
    if (!h) return fail(nullptr, -12, "gclm_create: out of host memory");
    h->cfg = *cfg;
    h->device = device;
    *out = h;
    return 0;
}

int gclm_configure(gclm_handle* h, const gclm_config* cfg) {
    if (!h || !cfg) return -1;
    char abuf[256];
    if (const char* msg = abi_mismatch(cfg, abuf)) return fail(h, -5, "gclm_configure: %s", msg);
    if (const char* msg = validate(*cfg)) return fail(h, -2, "gclm_configure: %s", msg);
    if (cfg->device != h->device) return fail(h, -2, "gclm_configure: a handle cannot move from device %d to %d", h->device, (int)cfg->device);
    h->cfg = *cfg;
    h->sh.active = false;
    return 0;
}

int gclm_destroy(gclm_handle* h) {
    if (!h) return 0;
    DeviceGuard guard(h->device);
    for (hipEvent_t e : h->ev) (void)hipEventDestroy(e);
    if (h->ws) (void)hipFree(h->ws);
    if (h->slat_buf) (void)hipFree(h->slat_buf);
    if (h->progress_host) (void)hipHostFree(h->progress_host);
    delete h;
    return 0;
}

const char* gclm_last_error(const gclm_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

size_t gclm_workspace_bytes(const gclm_handle* h) { return h ? h->ws_bytes + h->slat_bytes : 0; }

size_t gclm_slat_plane_bytes(const gclm_handle* h) { return h ? h->slat_bytes : 0; }

int gclm_set_slat_plane_limit(gclm_handle* h, size_t max_bytes) {
    if (!h) return -1;
    h->slat_limit = max_bytes;
    h->slat_refused = 0;
    h->sh.active = false;
    return 0;
}

int gclm_release_workspace(gclm_handle* h) {
    if (!h) return -1;
    DeviceGuard guard(h->device);
    GCLM_HIP(h, guard.status);
    h->sh.active = false;
    if (h->ws) GCLM_HIP(h, hipFree(h->ws));                // (hipFree waits for the device: nothing of this handle is in flight after)
    h->ws = nullptr;
    h->ws_bytes = 0;
    h->ctx = SolveCtx{};
    h->group_partials = nullptr;
    if (h->slat_buf) GCLM_HIP(h, hipFree(h->slat_buf));
    h->slat_buf = h->slat = nullptr;
    h->slat_bytes = h->slat_refused = 0;
    return 0;
}

int gclm_set_sweep_iters(gclm_handle* h, int iters) {
    if (!h) return -1;
    if (iters < 0 || iters > 4096) return fail(h, -3, "gclm_set_sweep_iters: %d out of range [0, 4096]", iters);
    h->sweep_iters = iters;
    h->sh.active = false;
    return 0;
}

int gclm_plan_cut(const gclm_handle* h, int B, int H, int W, int aligned16, int* rows_per_chunk, int* chunks_per_image) {
    if (!h) return -1;
    if (B < 1 || H < 1 || W < 1) return -3;
    const Geometry geo = plan_sweeps(h, B, H, W, aligned16 != 0, true);       // (as for the five planes of a complete field set)
    if (rows_per_chunk) *rows_per_chunk = geo.rows_per_block;
    if (chunks_per_image) *chunks_per_image = geo.nchunks;
    return 0;
}

int gclm_set_slat_plane(gclm_handle* h, int mode) {
    if (!h) return -1;
    if (mode < -1 || mode > 1) return fail(h, -3, "gclm_set_slat_plane: mode %d not in {-1, 0, 1}", mode);
    h->slat_plane = mode;
    h->slat_refused = 0;
    h->sh.active = false;
    return 0;
}

int gclm_set_row_pairs(gclm_handle* h, int mode) {
    if (!h) return -1;
    if (mode < -1 || mode > 1) return fail(h, -3, "gclm_set_row_pairs: mode %d not in {-1, 0, 1}", mode);
    h->row_pairs = mode;
    h->sh.active = false;
    return 0;
}

int gclm_set_stop_comm(gclm_handle* h, gclm_comm* c) {
    if (!h) return -1;
    h->stop_comm = c;
    return 0;
}

int gclm_merge_stop_at(gclm_handle* const* parts, float* const* d_info, const int* B, int n_parts, void* stream) {
    if (!parts || !d_info || !B || n_parts < 1 || !parts[0]) return -1;
    gclm_handle* h0 = parts[0];
    if (n_parts > kMaxMergeParts) return fail(h0, -3, "gclm_merge_stop_at: %d parts, at most %d", n_parts, kMaxMergeParts);
    MergeStopArgs a{};
    a.n = 0;
    a.num_steps = h0->cfg.num_steps;
    for (int p = 0; p < n_parts; ++p) {
        gclm_handle* h = parts[p];
        if (B[p] < 0) return fail(h0, -3, "gclm_merge_stop_at: part %d has a negative size", p);
        // an EMPTY part holds no image: it adds nothing to the counters and has no row to write (its handle may never have
        // solved anything, or may hold the counters of an older, non-empty solve -- a B = 0 solve does not touch them)
        if (B[p] == 0) continue;
        if (!h || !h->ctx.ctrl || !d_info[p]) return fail(h0, -3, "gclm_merge_stop_at: part %d has not solved anything", p);
        if (h->device != h0->device || h->cfg.num_steps != h0->cfg.num_steps || h->cfg.early_stop)
            return fail(h0, -2, "gclm_merge_stop_at: the parts must share device and num_steps and run with early_stop = 0");
        if (B[p] != h->ctx.B)       // the counters in this handle's workspace are those of its LAST solve
            return fail(h0, -2, "gclm_merge_stop_at: part %d is given as %d images, but the last solve of its handle had %d", p, B[p],
                        h->ctx.B);
        a.ctrl[a.n] = h->ctx.ctrl;
        a.info[a.n] = d_info[p];
        a.B[a.n] = B[p];
        ++a.n;
    }
    if (a.n == 0) return 0;
    DeviceGuard guard(h0->device);
    GCLM_HIP(h0, guard.status);
    GCLM_HIP(h0, launch_merge_stop(a, static_cast<hipStream_t>(stream)));
    return 0;
}

int gclm_set_fused_steps(gclm_handle* h, int mode) {
    if (!h) return -1;
    if (mode < -1 || mode > 1) return fail(h, -3, "gclm_set_fused_steps: mode %d not in {-1, 0, 1}", mode);
    h->fused_mode = mode;
    return 0;
}

int gclm_set_paced_launches(gclm_handle* h, int depth) {
    if (!h) return -1;
    if (depth < 0 || depth > 16) return fail(h, -3, "gclm_set_paced_launches: depth %d out of range [0, 16]", depth);
    if (depth > 0 && !h->progress_host) {
        DeviceGuard guard(h->device);
        GCLM_HIP(h, guard.status);
        void* p = nullptr;
        GCLM_HIP(h, hipHostMalloc(&p, 64, hipHostMallocMapped));
        h->progress_host = static_cast<unsigned*>(p);
        *h->progress_host = 0;
        void* d = nullptr;
        GCLM_HIP(h, hipHostGetDevicePointer(&d, p, 0));
        h->progress_dev = static_cast<unsigned*>(d);
    }
    h->paced_depth = depth;
    return 0;
}

int gclm_set_timing(gclm_handle* h, int enabled) {
    if (!h) return -1;
    h->timing = enabled != 0;
    h->ev_used = 0;
    return 0;
}

int gclm_last_pass_timing(gclm_handle* h, int* n_launches, float* total_ms) {
    if (!h) return -1;
    if (!h->timing) return fail(h, -4, "timing not enabled");
    float tot = 0.f;
    for (int i = 0; i + 1 < h->ev_used; i += 2) {
        GCLM_HIP(h, hipEventSynchronize(h->ev[i + 1]));
        float ms = 0.f;
        GCLM_HIP(h, hipEventElapsedTime(&ms, h->ev[i], h->ev[i + 1]));
        tot += ms;
    }
    if (n_launches) *n_launches = h->ev_used / 2;
    if (total_ms) *total_ms = tot;
    h->ev_used = 0;          // events accumulate over solves until read
    return 0;
}

static int run_solve(gclm_handle* h, const float* d_up, const float* d_lat, const float* d_up_conf,
                     const float* d_lat_conf, int B, int H, int W, const InitArgs& ia, float* d_cam_out,
                     float* d_grav_out, float* d_info_out, void* stream) {
    if (int rc = check_shapes(h, d_lat, B, H, W)) return rc;
    if (B > 0 && (!d_cam_out || !d_grav_out || !d_info_out)) return fail(h, -3, "null output pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (B == 0) {
        // an empty shard still owes its peers the per-step collectives of a rank-spanning early stop (zero counters)
        if (h->cfg.early_stop && h->stop_comm && h->cfg.num_steps > 1) {
            DeviceGuard guard0(h->device);
            GCLM_HIP(h, guard0.status);
            SolveCtx& c0 = h->ctx;
            c0.cfg = h->cfg;
            c0.B = 0; c0.H = H; c0.W = W; c0.nchunks = 1; c0.n_groups = 0; c0.group_size = 1; c0.group_of_frame = nullptr;
            if (int rc = ensure_workspace(h, 1, 1, 0)) return rc;
            GCLM_HIP(h, launch_init(c0, ia, s));                               // thread 0 resets the counters
            for (int step = 1; step < h->cfg.num_steps; ++step)
                if (gclm_comm_all_reduce_sum_i32(h->stop_comm, &c0.ctrl->notclose[step], 1, s) != 0)
                    return fail(h, -20, "early-stop all-reduce failed: %s", gclm_comm_last_error(h->stop_comm));
        }
        return 0;
    }
    DeviceGuard guard(h->device);
    GCLM_HIP(h, guard.status);
    const bool al = is_aligned16(d_up) && is_aligned16(d_lat) && is_aligned16(d_up_conf) && is_aligned16(d_lat_conf);
    const Geometry geo = plan_sweeps(h, B, H, W, al, d_up && d_up_conf && d_lat_conf);
    SolveCtx& c = h->ctx;
    c.cfg = h->cfg;
    c.B = B; c.H = H; c.W = W; c.nchunks = geo.nchunks;
    // trivial / heuristic initial estimate without `scales`: fx == fy at the start, and update_focal keeps the ratio
    // (camera.py:148) -- the final sweep may then run the cheaper log-focal instantiation (see finalize_kernel)
    c.iso_final = (ia.cam == nullptr && ia.scales == nullptr && GCLM_ISO_FINAL) ? 1 : 0;
    if (int rc = setup_groups(h, B)) return rc;
    const bool es = h->cfg.early_stop != 0;
    const bool fused_path = !geo.mirror && use_fused(h, B, geo);      // (plan_sweeps only pairs rows where the step is not one launch)
    const bool keep_slat = slat_wanted(h, d_up, d_up_conf, d_lat_conf, geo, fused_path);
    if (int rc = ensure_workspace(h, B, geo.nchunks, c.n_groups)) return rc;
    ensure_slat(h, keep_slat ? (size_t)B * H * W : 0);
    h->sh.active = false;
    bool slat_ready = false;

    if (!fused_path) GCLM_HIP(h, launch_init(c, ia, s));      // (the one-launch-per-step path builds theta_0 in its first launch)
    if (fused_path) {
        // fused(step) x num_steps | fused(final) | finalize : num_steps + 2 launches, partial records double-buffered
        float* const part[2] = {c.partials, c.partials2};
        // Paced launches (single image with early stop): the launches after the stop are skipped on the device, but each
        // still costs its turn on the queue (2.3 us x 21 of a default-conf solve).  Launch k is therefore only issued
        // once launch k - depth has reported (a word in host-mapped memory), and none once the stop is reported.  What
        // the host sees only decides how many launches it issues: the device-side skip stays in force, so a late or
        // missing report costs time, never correctness; the wait is bounded (kPacedPatienceUs, then a cool-down).
        bool paced = h->paced_depth > 0 && es && B == 1 && h->progress_host;
        if (paced && h->paced_cooldown > 0) { --h->paced_cooldown; paced = false; }
        if (paced) h->epoch = (h->epoch + 1) & kPacedEpochMask;
        bool stop_seen = false, pace = paced;
        auto reported = [&](int launch) {          // has `launch` (or a later one) of THIS solve reported?  sets stop_seen
            const unsigned v = __atomic_load_n(h->progress_host, __ATOMIC_ACQUIRE);
            if (((v >> kPacedEpochShift) & kPacedEpochMask) != h->epoch) return false;
            if (v & kPacedStopBit) stop_seen = true;
            return stop_seen || (v & 0xffffu) >= (unsigned)(launch + 1);
        };
        for (int step = 0; step <= h->cfg.num_steps; ++step) {
            const bool fin = step == h->cfg.num_steps;
            if (pace && !fin && step >= h->paced_depth) {
                // A launch of this solve reports within a few microseconds of the previous one when the queue is ours.
                // Spin briefly (the common case), then yield the core between polls; give up after kPacedPatienceUs.
                // A queue backed up by other streams (a serving loop overlapping the CNN with the LM) trips that: the rest
                // of THIS solve is issued unpaced, and the handle stops pacing for the next kPacedCooldown solves instead
                // of burning the patience on every call.
                const auto t0 = std::chrono::steady_clock::now();
                for (unsigned spin = 0; !reported(step - h->paced_depth); ++spin) {
                    if (spin < 256) continue;
                    std::this_thread::yield();
                    if ((spin & 0xf) == 0 &&
                        std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(kPacedPatienceUs)) {
                        pace = false;
                        h->paced_cooldown = kPacedCooldown;
                        break;
                    }
                }
            }
            if (stop_seen && !fin) continue;       // the final launch finds the stop in the counters, as it always does
            SweepArgs a = sweep_args(h, d_up, d_lat, d_up_conf, d_lat_conf, c.pb[0], geo, !fin, 0);
            a.partials = part[step & 1];
            FusedArgs f;
            f.c = c;
            f.step = step;
            f.is_final = fin ? 1 : 0;
            f.partials_in = part[(step + 1) & 1];
            f.progress = (paced && !fin) ? h->progress_dev : nullptr;
            f.epoch = h->epoch;
            f.ia = ia;
            f.init_here = step == 0 ? 1 : 0;
            if (int rc = timed_sweep(h, a, s, &f)) return rc;
        }
        SolveCtx cf = c;
        cf.partials = part[h->cfg.num_steps & 1];        // the final sweep's records
        GCLM_HIP(h, launch_finalize(cf, d_cam_out, d_grav_out, d_info_out, s));
        return 0;
    }
    for (int step = 0; step < h->cfg.num_steps; ++step) {
        SweepArgs a = sweep_args(h, d_up, d_lat, d_up_conf, d_lat_conf, c.pb[step & 1], geo, true, es ? step : 0);
        apply_slat(h, a, slat_ready);
        if (int rc = timed_sweep(h, a, s)) return rc;
        if (!h->cfg.shared_intrinsics) {
            GCLM_HIP(h, launch_update(c, step, s));
        } else {
            GCLM_HIP(h, launch_shared_step(c, step, s));     // reduce + Schur solve + update: one launch
        }
        // a batch sharded over ranks: "whose cost still moved" is a count over ALL ranks' images (every step, also after
        // the stop -- the counters of skipped steps stay 0 everywhere -- so that the ranks' collectives keep matching)
        if (es && h->stop_comm && step >= 1)
            if (gclm_comm_all_reduce_sum_i32(h->stop_comm, &c.ctrl->notclose[step], 1, s) != 0)
                return fail(h, -20, "early-stop all-reduce failed: %s", gclm_comm_last_error(h->stop_comm));
    }
    GCLM_HIP(h, launch_prep_final(c, s));
    SweepArgs a = sweep_args(h, d_up, d_lat, d_up_conf, d_lat_conf, c.pb_final, geo, false, 0);
    apply_slat(h, a, slat_ready);
    if (int rc = timed_sweep(h, a, s)) return rc;
    GCLM_HIP(h, launch_finalize(c, d_cam_out, d_grav_out, d_info_out, s));
    return 0;
}

int gclm_solve(gclm_handle* h, const float* d_up, const float* d_lat, const float* d_up_conf,
               const float* d_lat_conf, int B, int H, int W, float* d_cam_io, float* d_grav_io,
               float* d_info_out, void* stream) {
    if (!h) return -1;
    InitArgs ia{};
    ia.cam = d_cam_io;
    ia.grav = d_grav_io;
    if (B != 0 && (!d_cam_io || !d_grav_io)) return fail(h, -3, "gclm_solve: null camera / gravity pointer");
    return run_solve(h, d_up, d_lat, d_up_conf, d_lat_conf, B, H, W, ia, d_cam_io, d_grav_io, d_info_out, stream);
}

int gclm_calibrate(gclm_handle* h, const float* d_up, const float* d_lat, const float* d_up_conf,
                   const float* d_lat_conf, int B, int H, int W, const float* d_scales,
                   const float* d_prior_focal, const float* d_prior_gravity, const float* d_prior_dist,
                   int prior_dist_cols, float* d_cam_out, float* d_grav_out, float* d_info_out, void* stream) {
    if (!h) return -1;
    if (d_prior_dist && (prior_dist_cols < 1 || prior_dist_cols > 2)) return fail(h, -3, "gclm_calibrate: prior_dist_cols must be 1 or 2");
    // the free-parameter flags must agree with the priors (setup_optimization_and_priors, :204-221); an empty batch has
    // no prior rows either (NULL), and only takes part in the stop collectives
    if (B != 0 && ((d_prior_focal != nullptr) == (h->cfg.estimate_focal != 0) || (d_prior_gravity != nullptr) == (h->cfg.estimate_gravity != 0)))
        return fail(h, -2, "gclm_calibrate: estimate_focal / estimate_gravity disagree with the priors passed");
    InitArgs ia{};
    ia.scales = d_scales;
    ia.prior_focal = d_prior_focal;
    ia.prior_gravity = d_prior_gravity;
    ia.prior_dist = d_prior_dist;
    ia.prior_dist_cols = prior_dist_cols;
    ia.up = d_up;
    ia.lat = d_lat;
    if (h->cfg.heuristic_init && !d_up && B != 0) return fail(h, -3, "gclm_calibrate: heuristic_init needs the up field");
    return run_solve(h, d_up, d_lat, d_up_conf, d_lat_conf, B, H, W, ia, d_cam_out, d_grav_out, d_info_out, stream);
}

int gclm_system(gclm_handle* h, const float* d_up, const float* d_lat, const float* d_up_conf,
                const float* d_lat_conf, int B, int H, int W, const float* d_cam, const float* d_grav,
                int as_rpf, float* d_cost, float* d_grad, float* d_hess, void* stream) {
    if (!h) return -1;
    if (int rc = check_shapes(h, d_lat, B, H, W)) return rc;
    if (!d_cam || !d_grav || !d_cost || !d_grad || !d_hess) return fail(h, -3, "gclm_system: null pointer");
    if (B == 0) return 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    DeviceGuard guard(h->device);
    GCLM_HIP(h, guard.status);
    const bool al = is_aligned16(d_up) && is_aligned16(d_lat) && is_aligned16(d_up_conf) && is_aligned16(d_lat_conf);
    const Geometry geo = plan_sweeps(h, B, H, W, al, d_up && d_up_conf && d_lat_conf);
    SolveCtx& c = h->ctx;
    c.cfg = h->cfg;
    c.B = B; c.H = H; c.W = W; c.nchunks = geo.nchunks;
    c.n_groups = 0; c.group_size = 1; c.group_of_frame = nullptr; c.iso_final = 0;
    if (int rc = ensure_workspace(h, B, geo.nchunks, 0)) return rc;
    h->slat = nullptr;
    h->sh.active = false;
    GCLM_HIP(h, launch_pblock_from_params(c, d_cam, d_grav, as_rpf, c.pb_final, s));
    const SweepArgs a = sweep_args(h, d_up, d_lat, d_up_conf, d_lat_conf, c.pb_final, geo, !as_rpf, 0);
    GCLM_HIP(h, launch_sweep(h->cfg.camera_model, a, s));
    GCLM_HIP(h, launch_system_out(c, d_cost, d_grad, d_hess, s));
    return 0;
}

int gclm_shared_begin(gclm_handle* h, const float* d_up, const float* d_lat, const float* d_up_conf,
                      const float* d_lat_conf, int B_local, int H, int W, float* d_cam_io,
                      float* d_grav_io, const int32_t* d_group_of_frame, int num_groups, void* stream) {
    if (!h) return -1;
    if (!h->cfg.shared_intrinsics) return fail(h, -2, "gclm_shared_begin: handle is not configured for shared_intrinsics");
    if (h->cfg.early_stop) return fail(h, -2, "gclm_shared_begin: early_stop needs a global decision; run with early_stop=0");
    if (int rc = check_shapes(h, d_lat, B_local, H, W)) return rc;
    if (!d_cam_io || !d_grav_io || !d_group_of_frame || num_groups <= 0) return fail(h, -3, "gclm_shared_begin: bad arguments");
    hipStream_t s = static_cast<hipStream_t>(stream);
    DeviceGuard guard(h->device);
    GCLM_HIP(h, guard.status);
    const bool al = is_aligned16(d_up) && is_aligned16(d_lat) && is_aligned16(d_up_conf) && is_aligned16(d_lat_conf);
    const int Bp = B_local > 0 ? B_local : 1;
    h->sh.geo = plan_sweeps(h, Bp, H, W, al, d_up && d_up_conf && d_lat_conf);
    SolveCtx& c = h->ctx;
    c.cfg = h->cfg;
    c.B = B_local; c.H = H; c.W = W; c.nchunks = h->sh.geo.nchunks;
    c.n_groups = num_groups; c.group_size = 1; c.group_of_frame = d_group_of_frame; c.iso_final = 0;
    const bool keep_slat = B_local > 0 && slat_wanted(h, d_up, d_up_conf, d_lat_conf, h->sh.geo, false);
    if (int rc = ensure_workspace(h, Bp, h->sh.geo.nchunks, num_groups)) return rc;
    ensure_slat(h, keep_slat ? (size_t)B_local * H * W : 0);
    h->sh.slat_ready = false;
    h->sh.up = d_up; h->sh.lat = d_lat; h->sh.upc = d_up_conf; h->sh.latc = d_lat_conf;
    h->sh.cam_io = d_cam_io; h->sh.grav_io = d_grav_io;
    h->sh.active = true;
    InitArgs ia{};
    ia.cam = d_cam_io;
    ia.grav = d_grav_io;
    GCLM_HIP(h, launch_init(c, ia, s));
    return 0;
}

int gclm_shared_reduce(gclm_handle* h, int step, float* d_partials, void* stream) {
    if (!h) return -1;
    if (!h->sh.active) return fail(h, -4, "gclm_shared_reduce: no active session (call gclm_shared_begin)");
    if (!d_partials || step < 0 || step >= h->cfg.num_steps) return fail(h, -3, "gclm_shared_reduce: bad arguments");
    hipStream_t s = static_cast<hipStream_t>(stream);
    DeviceGuard guard(h->device);
    GCLM_HIP(h, guard.status);
    SolveCtx& c = h->ctx;
    if (c.B > 0) {
        SweepArgs a = sweep_args(h, h->sh.up, h->sh.lat, h->sh.upc, h->sh.latc, c.pb[step & 1], h->sh.geo, true, 0);
        apply_slat(h, a, h->sh.slat_ready);
        if (int rc = timed_sweep(h, a, s)) return rc;
    }
    GCLM_HIP(h, launch_shared_reduce(c, step, d_partials, s));
    return 0;
}

int gclm_shared_apply(gclm_handle* h, int step, const float* d_partials, void* stream) {
    if (!h) return -1;
    if (!h->sh.active) return fail(h, -4, "gclm_shared_apply: no active session");
    if (!d_partials || step < 0 || step >= h->cfg.num_steps) return fail(h, -3, "gclm_shared_apply: bad arguments");
    hipStream_t s = static_cast<hipStream_t>(stream);
    DeviceGuard guard(h->device);
    GCLM_HIP(h, guard.status);
    if (h->ctx.B > 0) GCLM_HIP(h, launch_shared_apply(h->ctx, step, d_partials, s));
    return 0;
}

int gclm_shared_finish(gclm_handle* h, float* d_info_out, void* stream) {
    if (!h) return -1;
    if (!h->sh.active) return fail(h, -4, "gclm_shared_finish: no active session");
    if (!d_info_out) return fail(h, -3, "gclm_shared_finish: null info pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    DeviceGuard guard(h->device);
    GCLM_HIP(h, guard.status);
    SolveCtx& c = h->ctx;
    h->sh.active = false;
    if (c.B == 0) return 0;
    GCLM_HIP(h, launch_prep_final(c, s));
    SweepArgs a = sweep_args(h, h->sh.up, h->sh.lat, h->sh.upc, h->sh.latc, c.pb_final, h->sh.geo, false, 0);
    apply_slat(h, a, h->sh.slat_ready);
    if (int rc = timed_sweep(h, a, s)) return rc;
    GCLM_HIP(h, launch_finalize(c, h->sh.cam_io, h->sh.grav_io, d_info_out, s));
    return 0;
}

int gclm_gradient_hessian(const float* d_J, const float* d_residual, const float* d_weight, int B, int N, int R,
                          int P, int accumulate, float* d_G, float* d_H, void* stream) {
    if (!d_J || !d_residual || !d_weight || !d_G || !d_H || B < 0 || N < 0 || R < 1 || R > 4 || P < 1 ||
        P > GCLM_MAX_PARAMS)
        return -3;
    hipError_t e = launch_gradient_hessian(d_J, d_residual, d_weight, B, N, R, P, accumulate, d_G, d_H,
                                           static_cast<hipStream_t>(stream));
    return e == hipSuccess ? 0 : -10;
}

int gclm_optimizer_step(const float* d_G, const float* d_H, const float* d_lambda, int lambda_is_scalar, float eps,
                        int B, int P, float* d_delta, int* d_failed, void* stream) {
    if (!d_G || !d_H || !d_lambda || !d_delta || B < 0 || P < 1 || P > GCLM_MAX_PARAMS) return -3;
    hipError_t e = launch_lm_step(d_G, d_H, d_lambda, lambda_is_scalar ? 0 : 1, eps, B, P, d_delta, d_failed,
                                  static_cast<hipStream_t>(stream));
    return e == hipSuccess ? 0 : -10;
}

int gclm_residual_fields(int camera_model, const float* d_up, const float* d_lat, const float* d_cam,
                         const float* d_grav, int B, int H, int W, float* d_r_up, float* d_r_lat, void* stream) {
    if (!d_cam || !d_grav || (!d_r_up && !d_r_lat) || B < 0 || H <= 0 || W <= 0) return -3;
    if ((d_r_up && !d_up) || (d_r_lat && !d_lat)) return -3;
    if (camera_model < GCLM_PINHOLE || camera_model > GCLM_SIMPLE_DIVISIONAL || B > 65535) return -3;
    hipError_t e = launch_residual_fields(camera_model, d_up, d_lat, d_cam, d_grav, B, H, W, d_r_up, d_r_lat,
                                          static_cast<hipStream_t>(stream));
    return e == hipSuccess ? 0 : -10;
}

int gclm_huber_costs(const float* d_residual, size_t n, int dim, float scale, const float* d_conf, float* d_cost,
                     float* d_weight, float* d_second, void* stream) {
    if (!d_residual || (!d_cost && !d_weight && !d_second) || dim < 0 || dim > 4 || !(scale > 0.f)) return -3;
    hipError_t e = launch_huber_costs(d_residual, n, dim, scale, d_conf, d_cost, d_weight, d_second,
                                      static_cast<hipStream_t>(stream));
    return e == hipSuccess ? 0 : -10;
}

int gclm_jacobian_fields(int camera_model, const float* d_cam, const float* d_grav, int B, int H, int W,
                         int spherical, int log_focal, float* d_J_up, float* d_J_lat, void* stream) {
    if (!d_cam || !d_grav || (!d_J_up && !d_J_lat) || B < 0 || H <= 0 || W <= 0) return -3;
    if (camera_model < GCLM_PINHOLE || camera_model > GCLM_SIMPLE_DIVISIONAL || B > 65535) return -3;
    hipError_t e = launch_jacobian_fields(camera_model, d_cam, d_grav, B, H, W, spherical, log_focal, d_J_up, d_J_lat,
                                          static_cast<hipStream_t>(stream));
    return e == hipSuccess ? 0 : -10;
}

int gclm_upsample_fields(const float* d_src, int planes, int h, int w, int H, int W, float* d_dst, void* stream) {
    if (!d_src || !d_dst || planes < 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0) return -3;
    hipError_t e = launch_upsample(d_src, planes, h, w, H, W, d_dst, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? 0 : -10;
}

int gclm_upsample_fields_multi(const float* const* d_srcs, float* const* d_dsts, const int* planes, int n_tensors, int h, int w,
                               int H, int W, void* stream) {
    if (!d_srcs || !d_dsts || !planes || n_tensors < 1 || n_tensors > kMaxUpsampleTensors || h <= 0 || w <= 0 || H <= 0 || W <= 0)
        return -1;
    UpsampleMulti m{};
    m.n = n_tensors;
    for (int t = 0; t < n_tensors; ++t) {
        if (planes[t] < 0 || (planes[t] > 0 && (!d_srcs[t] || !d_dsts[t]))) return -1;
        m.src[t] = d_srcs[t]; m.dst[t] = d_dsts[t]; m.planes[t] = planes[t];
    }
    return launch_upsample_multi(m, h, w, H, W, static_cast<hipStream_t>(stream)) == hipSuccess ? 0 : -10;
}

int gclm_pack_fields(const float* d_up_raw, const float* d_up_logconf, const float* d_lat_raw,
                     const float* d_lat_logconf, int B, int H, int W, float* d_up, float* d_up_conf, float* d_lat,
                     float* d_lat_conf, void* stream) {
    if (!d_up_raw || !d_lat_raw || !d_up || !d_lat || B < 0 || H <= 0 || W <= 0) return -3;
    if ((d_up_logconf && !d_up_conf) || (d_lat_logconf && !d_lat_conf)) return -3;
    const bool vec4 = ((size_t)H * W) % 4 == 0 && is_aligned16(d_up_raw) && is_aligned16(d_up_logconf) &&
                      is_aligned16(d_lat_raw) && is_aligned16(d_lat_logconf) && is_aligned16(d_up) &&
                      is_aligned16(d_up_conf) && is_aligned16(d_lat) && is_aligned16(d_lat_conf);
    hipError_t e = launch_pack_fields(d_up_raw, d_up_logconf, d_lat_raw, d_lat_logconf, B, H, W, vec4, d_up, d_up_conf,
                                      d_lat, d_lat_conf, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? 0 : -10;
}

int gclm_read_probe(const float* const* d_planes, int n_planes, size_t floats, void* stream) {
    if (!d_planes || n_planes < 1 || n_planes > 8 || floats % 4 != 0) return -3;
    for (int k = 0; k < n_planes; ++k)
        if (!d_planes[k] || !is_aligned16(d_planes[k])) return -3;
    return launch_read_probe(d_planes, n_planes, floats, static_cast<hipStream_t>(stream)) == hipSuccess ? 0 : -10;
}

int gclm_synth_fields_grouped(int camera_model, uint64_t seed, int64_t first_index, int B, int H, int W,
                              float noise_sigma, int group_size, int run, int run_stride, float* d_up,
                              float* d_lat, float* d_up_conf, float* d_lat_conf, float* d_gt_cam,
                              float* d_gt_grav, void* stream) {
    if (!d_up || !d_lat || B < 0 || H <= 0 || W <= 0 || group_size < 0 || run < 0) return -3;
    if (camera_model < GCLM_PINHOLE || camera_model > GCLM_SIMPLE_DIVISIONAL) return -2;
    hipError_t e = launch_synth(camera_model, seed, first_index, B, H, W, noise_sigma, group_size, run, run_stride,
                                d_up, d_lat, d_up_conf, d_lat_conf, d_gt_cam, d_gt_grav,
                                static_cast<hipStream_t>(stream));
    return e == hipSuccess ? 0 : -10;
}

int gclm_synth_fields(int camera_model, uint64_t seed, int64_t first_index, int B, int H, int W,
                      float noise_sigma, float* d_up, float* d_lat, float* d_up_conf,
                      float* d_lat_conf, float* d_gt_cam, float* d_gt_grav, void* stream) {
    return gclm_synth_fields_grouped(camera_model, seed, first_index, B, H, W, noise_sigma, 1, 0, 0, d_up, d_lat,
                                     d_up_conf, d_lat_conf, d_gt_cam, d_gt_grav, stream);
}

}  // extern "C"
