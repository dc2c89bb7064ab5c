/*
 * lm_oracle.c -- CPU restatement of GeoCalib's batched Levenberg-Marquardt calibration path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity *checker* for the HIP product path in
 * geocalib_amd/csrc.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it.  Nothing in geocalib_amd/ imports, links or executes it.
 *
 * It restates, in plain C, what the reference computes (all citations relative to
 * /root/reference/geocalib/):
 *   lm_optimizer.py:20-58    get_trivial_estimation      -> oracle_init()
 *   lm_optimizer.py:189-246  setup_optimization_and_priors -> struct plan / make_plan()
 *   lm_optimizer.py:248-274  calculate_residuals         -> pixel_eval() (forward part)
 *   lm_optimizer.py:276-315  calculate_costs (+ :61-87 scaled Huber) -> huber()
 *   lm_optimizer.py:317-461  calculate_gradient_and_hessian / setup_system -> pass()
 *   lm_optimizer.py:109-137  optimizer_step (damped Cholesky) -> lm_step()
 *   lm_optimizer.py:518-549  update_estimate             -> apply_update()
 *   lm_optimizer.py:551-644  optimize (loop, lambda rule, early stop, infos) -> lm_oracle_solve()
 *   lm_optimizer.py:463-516  estimate_uncertainty        -> uncertainty()
 *   perspective_fields.py:47-81,84-182   get_up_field / J_up_field
 *   perspective_fields.py:185-211,214-275 get_latitude_field / J_latitude_field
 *   camera.py:136-152 update_focal, :309-323 normalize/J_normalize, :325-344 pixel grid,
 *             :522-562 Pinhole, :565-660 SimpleRadial, :663-786 Radial, :789-942 SimpleDivisional
 *   gravity.py:31-40 from_rp, :63-101 roll/pitch/J_rp, :112-119 update
 *   misc.py:182-259 SphericalManifold, :263-281 J_vecnorm, :285-287 J_focal2fov, :291-318 J_up_projection
 *
 * The per-pixel Jacobians are deliberately formed the way the reference forms them (explicit
 * 2x2 / 3x2 matrix chains), NOT in the simplified closed form the HIP kernels use, so that
 * kernel-vs-oracle parity also checks the algebraic simplifications.
 *
 * Arithmetic: `real` is float (default, like the reference) or double (-DORACLE_F64); reductions
 * over pixels are accumulated in double in both builds.  The oracle is pinned against outputs of
 * the reference itself (tests/golden/make_golden.py -> tests/golden npz files; tests/test_oracle.py).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#ifdef ORACLE_F64
typedef double real;
#define R(x) x
#define rsqrt_ sqrt
#define rsin sin
#define rcos cos
#define rasin asin
#define ratan atan
#define rtan tan
#define rexp exp
#define rlog log
#define rfabs fabs
#define rpow pow
#else
typedef float real;
#define R(x) x##f
#define rsqrt_ sqrtf
#define rsin sinf
#define rcos cosf
#define rasin asinf
#define ratan atanf
#define rtan tanf
#define rexp expf
#define rlog logf
#define rfabs fabsf
#define rpow powf
#endif

enum { CAM_PINHOLE = 0, CAM_SIMPLE_RADIAL = 1, CAM_RADIAL = 2, CAM_SIMPLE_DIVISIONAL = 3 };

#define MAXP 5    /* delta_g1, delta_g2, focal, k1, k2 */
#define INFO_STRIDE 48

/* Mirrors LMOptimizer.default_conf (lm_optimizer.py:144-162) + which inputs exist. */
typedef struct {
    int camera_model;
    int shared_intrinsics;
    int num_steps;
    double lambda0;
    int fix_lambda;
    int early_stop;
    double atol, rtol;
    int use_spherical_manifold;
    int use_log_focal;
    double up_loss_fn_scale, lat_loss_fn_scale;
    int training;            /* nn.Module.training: skips estimate_uncertainty (:635) */
    int num_threads;         /* OpenMP threads over images / pixel rows; <=0: library default */
    int heuristic_init;      /* siclib/models/optimization/utils.py:27-82 instead of the trivial estimate */
} oracle_conf;

typedef struct {
    int B, H, W;
    const float *up;         /* (B,2,H,W) or NULL */
    const float *lat;        /* (B,1,H,W) radians or NULL */
    const float *up_conf;    /* (B,H,W) or NULL */
    const float *lat_conf;   /* (B,H,W) or NULL */
    const float *scales;     /* (2,) or NULL */
    const float *prior_focal;   /* (B,) or NULL */
    const float *prior_gravity; /* (B,3) or NULL */
    const float *prior_dist;    /* (B,nd) or NULL */
} oracle_data;

/* lm_optimizer.py:189-246 */
typedef struct {
    int has_dist, ndist;
    int est_grav, est_focal, est_dist;
    int n_params;            /* columns kept by calculate_gradient_and_hessian (:335-344) */
    int cols[MAXP];          /* indices into the full [d1,d2,f,k1,k2] Jacobian */
    int grav_dims[2];        /* gravity_delta_dims (or -1) */
    int focal_dim;           /* focal_delta_dims[0] (or -1) */
    int dist_dims[2];        /* dist_delta_dims */
    int n_intrinsic;
} plan_t;

static int num_dist(int model) {
    return model == CAM_PINHOLE ? 0 : (model == CAM_RADIAL ? 2 : 1);
}

static void make_plan(const oracle_conf *c, const oracle_data *d, plan_t *p) {
    memset(p, 0, sizeof(*p));
    p->has_dist = c->camera_model != CAM_PINHOLE;
    p->ndist = num_dist(c->camera_model);
    p->est_grav = d->prior_gravity == NULL;
    p->est_focal = d->prior_focal == NULL;
    p->est_dist = p->has_dist && d->prior_dist == NULL;
    /* :223-235, python negative indices kept as -1 */
    p->grav_dims[0] = p->est_grav ? 0 : -1;
    p->grav_dims[1] = p->est_grav ? 1 : -1;
    int gmax = p->est_grav ? 1 : -1;
    p->focal_dim = p->est_focal ? gmax + 1 : -1;
    for (int k = 0; k < p->ndist; ++k) p->dist_dims[k] = p->focal_dim + 1 + k;
    p->n_intrinsic = p->est_focal + (p->has_dist ? p->ndist : 0);
    /* :335-344 */
    int n = 0;
    if (p->est_grav) { p->cols[n++] = 0; p->cols[n++] = 1; }
    if (p->est_focal) p->cols[n++] = 2;
    if (p->has_dist) for (int k = 0; k < p->ndist; ++k) p->cols[n++] = 3 + k;
    p->n_params = n;
}

/* ---------------------------------------------------------------- camera model primitives */

typedef struct { real w, h, fx, fy, cx, cy, k1, k2; } cam_t;
typedef struct { real x, y, z; } vec3;

/* distort(uv, return_scale=True): camera.py:532-535, 611-617, 712-719, 829-839 */
static real distort_scale(int model, const cam_t *c, real u, real v) {
    real r2 = u * u + v * v;
    switch (model) {
    case CAM_SIMPLE_RADIAL: return R(1.0) + c->k1 * r2;
    case CAM_RADIAL: return R(1.0) + c->k1 * r2 + c->k2 * r2 * r2;
    case CAM_SIMPLE_DIVISIONAL: {
        real t = R(1.0) - R(4.0) * c->k1 * r2;
        if (t < 0) t = 0;
        real radial = R(1.0) - rsqrt_(t);
        real denom = R(2.0) * c->k1 * r2;
        return denom == 0 ? R(1.0) : radial / denom;
    }
    default: return R(1.0);
    }
}

/* up_projection_offset = J_distort(uv, "scale2pts"): camera.py:271-273, 625-626, 729-732, 847-851 */
static void up_offset(int model, const cam_t *c, real u, real v, real off[2]) {
    real r2 = u * u + v * v, s = 0;
    switch (model) {
    case CAM_SIMPLE_RADIAL: s = R(2.0) * c->k1; break;
    case CAM_RADIAL: s = R(2.0) * c->k1 + R(4.0) * c->k2 * r2; break;
    case CAM_SIMPLE_DIVISIONAL: {
        real t0 = R(1.0) - R(4.0) * c->k1 * r2;
        if (t0 < R(1e-6)) t0 = R(1e-6);
        t0 = rsqrt_(t0);
        real d1 = t0 * R(2.0) * r2, d2 = c->k1 * r2 * r2, denom = d1 * d2;
        if (denom == 0) denom = R(1e6);
        s = (R(4.0) * d2 - (R(1.0) - t0) * d1) / denom;
        break;
    }
    default: s = 0;
    }
    off[0] = s * u; off[1] = s * v;
}

/* J_up_projection_offset(uv, "uv"): camera.py:557-562, 653-656, 772-782, 888-940 */
static void J_offset_uv(int model, const cam_t *c, real u, real v, real J[2][2]) {
    real r2 = u * u + v * v;
    real pp[2][2] = {{u * u, u * v}, {v * u, v * v}};
    real di = 0, pc = 0;
    switch (model) {
    case CAM_SIMPLE_RADIAL: di = R(2.0) * c->k1; break;
    case CAM_RADIAL: di = R(2.0) * c->k1 + R(4.0) * c->k2 * r2; pc = R(8.0) * c->k2; break;
    case CAM_SIMPLE_DIVISIONAL: {
        real k1 = c->k1;
        real t0 = R(1.0) - R(4.0) * k1 * r2;
        if (t0 < R(1e-6)) t0 = R(1e-6);
        real t1 = rsqrt_(t0), t032 = t0 * t1, den;
#define SAFE(x) ((den = (x)) == 0 ? R(1e6) : den)
        di = R(4.0) / SAFE(R(2.0) * r2 * t1);
        pc = -R(16.0) / SAFE(R(4.0) * t1 * r2 * r2);
        pc += (R(32.0) * k1) / SAFE(R(4.0) * r2 * t032);
        pc -= R(4.0) / SAFE(r2 * r2 * t1);
        pc += R(4.0) * (R(1.0) - t1) / SAFE(k1 * r2 * r2 * r2);
        di -= (R(1.0) - t1) / SAFE(k1 * r2 * r2);
#undef SAFE
        break;
    }
    default: break;
    }
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j) J[i][j] = pc * pp[i][j] + (i == j ? di : 0);
}

/* J_up_projection_offset(uv, "dist") (2 x K): camera.py:657-658, 783-784, 898-911 */
static void J_offset_dist(int model, const cam_t *c, real u, real v, real J[2][2]) {
    real r2 = u * u + v * v;
    memset(J, 0, sizeof(real) * 4);
    switch (model) {
    case CAM_SIMPLE_RADIAL: J[0][0] = R(2.0) * u; J[1][0] = R(2.0) * v; break;
    case CAM_RADIAL:
        J[0][0] = R(2.0) * u; J[1][0] = R(2.0) * v;
        J[0][1] = R(4.0) * r2 * u; J[1][1] = R(4.0) * r2 * v; break;
    case CAM_SIMPLE_DIVISIONAL: {
        real k1 = c->k1;
        real t0 = R(1.0) - R(4.0) * k1 * r2;
        if (t0 < R(1e-6)) t0 = R(1e-6);
        real t1 = rsqrt_(t0), den, Jv;
        den = R(4.0) * t0 * t1; if (den == 0) den = R(1e6);
        Jv = R(16.0) / den;
        den = r2 * t1 * k1; if (den == 0) den = R(1e6);
        Jv -= R(2.0) / den;
        den = (r2 * k1) * (r2 * k1); if (den == 0) den = R(1e6);
        Jv += (R(1.0) - t1) / den;
        J[0][0] = Jv * u; J[1][0] = Jv * v;
        break;
    }
    default: break;
    }
}

/* J_distort(uv, "scale2dist") (K): camera.py:623-624, 725-728, 853-857 */
static void J_scale_dist(int model, const cam_t *c, real u, real v, real J[2]) {
    real r2 = u * u + v * v;
    J[0] = J[1] = 0;
    switch (model) {
    case CAM_SIMPLE_RADIAL: J[0] = r2; break;
    case CAM_RADIAL: J[0] = r2; J[1] = r2 * r2; break;
    case CAM_SIMPLE_DIVISIONAL: {
        real k1 = c->k1;
        real t0 = R(1.0) - R(4.0) * k1 * r2;
        if (t0 < R(1e-6)) t0 = R(1e-6);
        t0 = rsqrt_(t0);
        real d1 = R(2.0) * k1 * t0, d2 = R(2.0) * r2 * k1 * k1, denom = d1 * d2;
        if (denom == 0) denom = R(1e6);
        J[0] = (R(2.0) * d2 - (R(1.0) - t0) * d1) / denom;
        break;
    }
    default: break;
    }
}

/* undistort(uv): camera.py:546-548, 631-636, 737-746, 863-868 */
static void undistort(int model, const cam_t *c, real u, real v, real out[2]) {
    real r2 = u * u + v * v, radial = R(1.0);
    switch (model) {
    case CAM_SIMPLE_RADIAL: radial = R(1.0) + (-c->k1) * r2; break;
    case CAM_RADIAL: {
        real b1 = -c->k1, b2 = R(3.0) * c->k1 * c->k1 - c->k2;
        radial = R(1.0) + b1 * r2 + b2 * r2 * r2; break;
    }
    case CAM_SIMPLE_DIVISIONAL: {
        real denom = R(1.0) + c->k1 * r2;
        if (denom == 0) denom = R(1e6);
        radial = R(1.0) / denom; break;
    }
    default: break;
    }
    out[0] = u * radial; out[1] = v * radial;
}

/* J_undistort(uv, "pts") (2x2): camera.py:550-555, 645-649, 760-767, 878-883 */
static void J_undistort_pts(int model, const cam_t *c, real u, real v, real J[2][2]) {
    real r2 = u * u + v * v, di = R(1.0), pc = 0;
    switch (model) {
    case CAM_SIMPLE_RADIAL: { real b1 = -c->k1; di = R(1.0) + b1 * r2; pc = R(2.0) * b1; break; }
    case CAM_RADIAL: {
        real b1 = -c->k1, b2 = R(3.0) * c->k1 * c->k1 - c->k2;
        pc = R(4.0) * r2 * b2 + R(2.0) * b1;
        di = R(1.0) + b1 * r2 + b2 * r2 * r2; break;
    }
    case CAM_SIMPLE_DIVISIONAL: {
        real t0 = R(1.0) + c->k1 * r2;
        if (t0 == 0) t0 = R(1e6);
        di = R(1.0) / t0; pc = -R(2.0) * c->k1 / (t0 * t0); break;
    }
    default: break;
    }
    J[0][0] = pc * u * u + di; J[0][1] = pc * u * v;
    J[1][0] = pc * v * u;      J[1][1] = pc * v * v + di;
}

/* J_undistort(uv, "dist") (2xK): camera.py:643-644, 756-759, 875-877 */
static void J_undistort_dist(int model, const cam_t *c, real u, real v, real J[2][2]) {
    real r2 = u * u + v * v, r4 = r2 * r2;
    memset(J, 0, sizeof(real) * 4);
    switch (model) {
    case CAM_SIMPLE_RADIAL: J[0][0] = -r2 * u; J[1][0] = -r2 * v; break;
    case CAM_RADIAL: {
        real a = R(6.0) * r4 * c->k1 - r2;
        J[0][0] = a * u; J[1][0] = a * v; J[0][1] = -r4 * u; J[1][1] = -r4 * v; break;
    }
    case CAM_SIMPLE_DIVISIONAL: {
        real denom = (R(1.0) + c->k1 * r2) * (R(1.0) + c->k1 * r2);
        if (denom == 0) denom = R(1e6);
        J[0][0] = -r2 / denom * u; J[1][0] = -r2 / denom * v; break;
    }
    default: break;
    }
}

/* ---------------------------------------------------------------- gravity / manifold */

/* gravity.py:63-67 */
static real grav_roll(vec3 g) {
    real roll = rasin(-g.x / (rsqrt_(R(1.0) - g.z * g.z) + R(1e-4)));
    real sgn = (g.x > 0) - (g.x < 0);
    real offset = -(real)M_PI * sgn;
    return g.y < 0 ? roll : -roll + offset;
}
static real grav_pitch(vec3 g) { return rasin(g.z); }

/* gravity.py:69-101: T[i][k], k=0 roll, k=1 pitch */
static void J_rp(vec3 g, real T[3][2]) {
    real r = grav_roll(g), p = grav_pitch(g);
    real cr = rcos(r), sr = rsin(r), cp = rcos(p), sp = rsin(p);
    T[0][0] = -cr * cp; T[1][0] = sr * cp; T[2][0] = 0;
    T[0][1] = sr * sp;  T[1][1] = cr * sp; T[2][1] = cp;
}

/* misc.py:182-209 */
static void householder(vec3 x, real v[3], real *beta) {
    real sigma = x.x * x.x + x.y * x.y;
    real xpiv = x.z;
    real norm = rsqrt_(x.x * x.x + x.y * x.y + x.z * x.z);
    if (sigma < R(1e-7)) sigma = sigma + R(1e-7);
    real vpiv = xpiv < 0 ? xpiv - norm : -sigma / (xpiv + norm);
    *beta = R(2.0) * vpiv * vpiv / (sigma + vpiv * vpiv);
    v[0] = x.x / vpiv; v[1] = x.y / vpiv; v[2] = R(1.0);
}

/* misc.py:226-231 */
static void J_plus(vec3 g, real T[3][2]) {
    real v[3], beta;
    householder(g, v, &beta);
    for (int i = 0; i < 3; ++i)
        for (int k = 0; k < 2; ++k) T[i][k] = -beta * v[i] * v[k] + (i == k ? R(1.0) : 0);
}

static vec3 normalize3(vec3 g) {   /* F.normalize(dim=-1), eps 1e-12 */
    real n = rsqrt_(g.x * g.x + g.y * g.y + g.z * g.z);
    if (n < R(1e-12)) n = R(1e-12);
    vec3 o = {g.x / n, g.y / n, g.z / n};
    return o;
}

/* gravity.py:31-40 */
static vec3 from_rp(real roll, real pitch) {
    real sr = rsin(roll), cr = rcos(roll), sp = rsin(pitch), cp = rcos(pitch);
    vec3 g = {-sr * cp, -cr * cp, sp};
    return normalize3(g);
}

/* gravity.py:112-119 + misc.py:234-259 */
static vec3 grav_update(vec3 g, real d0, real d1, int spherical) {
    if (!spherical) return from_rp(grav_roll(g) + d0, grav_pitch(g) + d1);
    const real eps = R(1e-7);
    real nx = rsqrt_(g.x * g.x + g.y * g.y + g.z * g.z);
    real nd = rsqrt_(d0 * d0 + d1 * d1);
    real nd_ = nd < eps ? nd + eps : nd;
    real sinc = nd < eps ? R(1.0) : rsin(nd_) / nd_;
    real e[3] = {sinc * d0, sinc * d1, rcos(nd)};
    real v[3], beta;
    householder(g, v, &beta);
    real dot = v[0] * e[0] + v[1] * e[1] + v[2] * e[2];
    vec3 o = {nx * (e[0] - v[0] * (beta * dot)), nx * (e[1] - v[1] * (beta * dot)),
              nx * (e[2] - v[2] * (beta * dot))};
    return normalize3(o);
}

/* ---------------------------------------------------------------- per-pixel evaluation */

typedef struct {
    real r_up[2], r_lat;         /* residuals data - prediction */
    real J_up[2][MAXP], J_lat[MAXP];   /* full columns [d1, d2, f, k1, k2] */
} pix_t;

/* Forward + (optionally) Jacobian for one pixel.  T is J_plus(g) or J_rp(g) (3x2). */
static void pixel_eval(int model, const cam_t *c, vec3 g, real T[3][2], int log_focal,
                       real x, real y, const real *d_up, const real *d_lat, int want_J, pix_t *o) {
    const int has_dist = model != CAM_PINHOLE;
    const int nd = num_dist(model);
    real u = (x - c->cx) / c->fx, v = (y - c->cy) / c->fy;   /* camera.py:309-311 */
    real uv[2] = {u, v};
    /* ---- up field: perspective_fields.py:63-79 */
    real p[2] = {g.x - g.z * u, g.y - g.z * v};
    real M[2][2] = {{1, 0}, {0, 1}}, off[2] = {0, 0}, q[2];
    if (has_dist) {
        real d = distort_scale(model, c, u, v);
        up_offset(model, c, u, v, off);
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 2; ++j) M[i][j] = (i == j ? d : 0) + off[i] * uv[j];
    }
    for (int i = 0; i < 2; ++i) q[i] = M[i][0] * p[0] + M[i][1] * p[1];
    real nq = rsqrt_(q[0] * q[0] + q[1] * q[1]);
    real nq_c = nq < R(1e-12) ? R(1e-12) : nq;           /* F.normalize eps */
    if (d_up) { o->r_up[0] = d_up[0] - q[0] / nq_c; o->r_up[1] = d_up[1] - q[1] / nq_c; }
    /* ---- latitude: perspective_fields.py:203-209, lm_optimizer.py:262,270 */
    real w2[2];
    undistort(model, c, u, v, w2);
    real uv1[3] = {w2[0], w2[1], R(1.0)};
    real n3 = rsqrt_(uv1[0] * uv1[0] + uv1[1] * uv1[1] + R(1.0));
    real ray[3] = {uv1[0] / n3, uv1[1] / n3, uv1[2] / n3};
    real s = ray[0] * g.x + ray[1] * g.y + ray[2] * g.z;
    if (d_lat) {
        real sc = s < R(-1.0) + R(1e-6) ? R(-1.0) + R(1e-6) : (s > R(1.0) - R(1e-6) ? R(1.0) - R(1e-6) : s);
        o->r_lat = rsin(*d_lat) - rsin(rasin(sc));
    }
    if (!want_J) return;

    /* ---- J_up_field: perspective_fields.py:104-182 */
    real Jn[2][2];                                         /* J_vecnorm(q): misc.py:263-281 */
    {
        real nn = nq == 0 ? nq + R(1e-6) : nq;
        real n3_ = nn * nn * nn;
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 2; ++j) Jn[i][j] = (i == j ? R(1.0) / nn : 0) - q[i] * q[j] / n3_;
    }
    real Jabc[2][3] = {{1, 0, -u}, {0, 1, -v}};            /* misc.py:309-315 */
    real MJ[2][3];
    for (int i = 0; i < 2; ++i)
        for (int k = 0; k < 3; ++k) MJ[i][k] = M[i][0] * Jabc[0][k] + M[i][1] * Jabc[1][k];
    for (int k = 0; k < 2; ++k) {
        real pd[2];
        for (int i = 0; i < 2; ++i) pd[i] = MJ[i][0] * T[0][k] + MJ[i][1] * T[1][k] + MJ[i][2] * T[2][k];
        for (int i = 0; i < 2; ++i) o->J_up[i][k] = Jn[i][0] * pd[0] + Jn[i][1] * pd[1];
    }
    real Jq_uv[2][2];                                      /* :144-153 */
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j) Jq_uv[i][j] = -g.z * M[i][j];
    if (has_dist) {
        real Joff[2][2];
        J_offset_uv(model, c, u, v, Joff);
        real inner = u * p[0] + v * p[1];
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 2; ++j)
                Jq_uv[i][j] += p[i] * off[j] + inner * Joff[i][j] + off[i] * p[j];
    }
    real wv[2] = {-(x - c->cx) / (c->fx * c->fx), -(y - c->cy) / (c->fy * c->fy)};  /* camera.py:317 */
    if (log_focal) { wv[0] *= c->fx; wv[1] *= c->fy; }    /* :157-160 */
    {
        real pf[2] = {Jq_uv[0][0] * wv[0] + Jq_uv[0][1] * wv[1], Jq_uv[1][0] * wv[0] + Jq_uv[1][1] * wv[1]};
        for (int i = 0; i < 2; ++i) o->J_up[i][2] = Jn[i][0] * pf[0] + Jn[i][1] * pf[1];
    }
    if (has_dist) {                                        /* :170-180 */
        real Jd[2], Jod[2][2];
        J_scale_dist(model, c, u, v, Jd);
        J_offset_dist(model, c, u, v, Jod);
        for (int k = 0; k < nd; ++k) {
            real pj = p[0] * Jod[0][k] + p[1] * Jod[1][k];
            real col[2] = {p[0] * Jd[k] + u * pj, p[1] * Jd[k] + v * pj};
            for (int i = 0; i < 2; ++i) o->J_up[i][3 + k] = Jn[i][0] * col[0] + Jn[i][1] * col[1];
        }
    }
    /* ---- J_latitude_field: perspective_fields.py:234-275 */
    real Jn3[3][2];
    {
        real nn = n3, n3c = nn * nn * nn;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 2; ++j) Jn3[i][j] = (i == j ? R(1.0) / nn : 0) - uv1[i] * uv1[j] / n3c;
    }
    real gv[3] = {g.x, g.y, g.z};
    for (int k = 0; k < 2; ++k) o->J_lat[k] = ray[0] * T[0][k] + ray[1] * T[1][k] + ray[2] * T[2][k];
    real Ju[2][2];
    J_undistort_pts(model, c, u, v, Ju);
    {
        real wf[2] = {Ju[0][0] * wv[0] + Ju[0][1] * wv[1], Ju[1][0] * wv[0] + Ju[1][1] * wv[1]};
        real acc = 0;
        for (int i = 0; i < 3; ++i) acc += (Jn3[i][0] * wf[0] + Jn3[i][1] * wf[1]) * gv[i];
        o->J_lat[2] = acc;
    }
    if (has_dist) {
        real Jud[2][2];
        J_undistort_dist(model, c, u, v, Jud);
        for (int k = 0; k < nd; ++k) {
            real acc = 0;
            for (int i = 0; i < 3; ++i) acc += (Jn3[i][0] * Jud[0][k] + Jn3[i][1] * Jud[1][k]) * gv[i];
            o->J_lat[3 + k] = acc;
        }
    }
}

/* lm_optimizer.py:61-87: scaled Huber on squared residual x; returns cost, writes weight */
static real huber(real x, double a, real *weight) {
    real a2 = (real)(a * a);   /* python float a**2, then tensor / scalar */
    real y = x / a2;
    real sx = rsqrt_(y + R(1e-8));
    real isx = R(1.0) / sx;
    const real eps = R(1.1920928955078125e-07);
    if (isx < eps) isx = eps;
    if (y <= R(1.0)) { *weight = R(1.0); return y * a2; }
    *weight = isx;
    return (R(2.0) * sx - R(1.0)) * a2;
}

typedef struct {
    double cost_up, cost_lat;      /* mean over pixels */
    double G[MAXP], Hm[MAXP][MAXP];
} sys_t;

/* One sweep over one image: costs (+ gradient/Hessian over the plan's columns). */
static void image_pass(const oracle_conf *cf, const oracle_data *d, const plan_t *pl, int b,
                       const cam_t *c, vec3 g, int as_rpf, int want_sys, sys_t *out) {
    const int H = d->H, W = d->W, N = H * W, P = pl->n_params;
    real T[3][2];
    int spherical = cf->use_spherical_manifold && !as_rpf;
    int log_focal = cf->use_log_focal && !as_rpf;
    if (spherical) J_plus(g, T); else J_rp(g, T);
    memset(out, 0, sizeof(*out));
    const float *up0 = d->up ? d->up + (size_t)b * 2 * N : NULL;
    const float *up1 = d->up ? up0 + N : NULL;
    const float *lat = d->lat ? d->lat + (size_t)b * N : NULL;
    const float *uc = d->up_conf ? d->up_conf + (size_t)b * N : NULL;
    const float *lc = d->lat_conf ? d->lat_conf + (size_t)b * N : NULL;
    double cu = 0, cl = 0, G[MAXP] = {0}, Hm[MAXP][MAXP] = {{0}};
    for (int y = 0; y < H; ++y) {
        for (int x = 0; x < W; ++x) {      /* camera.py:325-344: integer grid, row-major */
            int i = y * W + x;
            pix_t px;
            real dup[2] = {0, 0}, dlat = 0;
            if (up0) { dup[0] = up0[i]; dup[1] = up1[i]; }
            if (lat) dlat = lat[i];
            pixel_eval(cf->camera_model, c, g, T, log_focal, (real)x, (real)y,
                       up0 ? dup : NULL, lat ? &dlat : NULL, want_sys, &px);
            if (up0) {
                real wgt, x2 = px.r_up[0] * px.r_up[0] + px.r_up[1] * px.r_up[1];
                real cost = huber(x2, cf->up_loss_fn_scale, &wgt);
                if (uc) { wgt *= uc[i]; cost *= uc[i]; }
                cu += cost;
                if (want_sys)
                    for (int k = 0; k < P; ++k) {
                        int ck = pl->cols[k];
                        real gk = px.J_up[0][ck] * px.r_up[0] + px.J_up[1][ck] * px.r_up[1];
                        G[k] += (double)(wgt * gk);
                        for (int l = 0; l < P; ++l) {
                            int cl_ = pl->cols[l];
                            real h = px.J_up[0][ck] * px.J_up[0][cl_] + px.J_up[1][ck] * px.J_up[1][cl_];
                            Hm[k][l] += (double)(wgt * h);
                        }
                    }
            }
            if (lat) {
                real wgt, x2 = px.r_lat * px.r_lat;
                real cost = huber(x2, cf->lat_loss_fn_scale, &wgt);
                if (lc) { wgt *= lc[i]; cost *= lc[i]; }
                cl += cost;
                if (want_sys)
                    for (int k = 0; k < P; ++k) {
                        int ck = pl->cols[k];
                        G[k] += (double)(wgt * (px.J_lat[ck] * px.r_lat));
                        for (int l = 0; l < P; ++l)
                            Hm[k][l] += (double)(wgt * (px.J_lat[ck] * px.J_lat[pl->cols[l]]));
                    }
            }
        }
    }
    out->cost_up = cu / N;
    out->cost_lat = cl / N;
    memcpy(out->G, G, sizeof(G));
    memcpy(out->Hm, Hm, sizeof(Hm));
}

/* Dense Cholesky solve (lower), n x n, row-major A (overwritten); returns 0 on failure. */
static int chol_solve(int n, real *A, real *b) {
    for (int j = 0; j < n; ++j) {
        real s = A[j * n + j];
        for (int k = 0; k < j; ++k) s -= A[j * n + k] * A[j * n + k];
        if (!(s > 0)) return 0;
        real l = rsqrt_(s);
        A[j * n + j] = l;
        for (int i = j + 1; i < n; ++i) {
            real t = A[i * n + j];
            for (int k = 0; k < j; ++k) t -= A[i * n + k] * A[j * n + k];
            A[i * n + j] = t / l;
        }
    }
    for (int i = 0; i < n; ++i) {
        real t = b[i];
        for (int k = 0; k < i; ++k) t -= A[i * n + k] * b[k];
        b[i] = t / A[i * n + i];
    }
    for (int i = n - 1; i >= 0; --i) {
        real t = b[i];
        for (int k = i + 1; k < n; ++k) t -= A[k * n + i] * b[k];
        b[i] = t / A[i * n + i];
    }
    return 1;
}

/* lm_optimizer.py:109-137.  H (n x n), G (n) in; delta out.  Returns 0 if Cholesky failed. */
static int lm_step(int n, const real *Hm, const real *G, real lambda, real *delta) {
    real *A = (real *)malloc(sizeof(real) * n * n);
    memcpy(A, Hm, sizeof(real) * n * n);
    for (int i = 0; i < n; ++i) {
        real dg = Hm[i * n + i] * lambda;
        if (dg < R(1e-6)) dg = R(1e-6);
        A[i * n + i] = Hm[i * n + i] + dg;
    }
    memcpy(delta, G, sizeof(real) * n);
    int ok = chol_solve(n, A, delta);
    free(A);
    return ok;
}

/* camera.py:136-152 + utils.py:272-279,297 */
static void update_focal(cam_t *c, real delta, int as_log) {
    real fx = as_log ? rexp(rlog(c->fx) + delta) : c->fx + delta;
    real fy = as_log ? rexp(rlog(c->fy) + delta) : c->fy + delta;
    real min_f = c->h / R(2.0) / rtan((R(150.0) / R(180.0) * (real)M_PI) / R(2.0));
    real max_f = c->h / R(2.0) / rtan((R(5.0) / R(180.0) * (real)M_PI) / R(2.0));
    fx = fx < min_f ? min_f : (fx > max_f ? max_f : fx);
    fy = fy < min_f ? min_f : (fy > max_f ? max_f : fy);
    (void)fx;
    real fx_new = fy * c->fx / c->fy;
    c->fx = fx_new; c->fy = fy;
}

/* camera.py:599-604, 700-705, 817-822 */
static void update_dist(int model, cam_t *c, const real *delta) {
    real lo = model == CAM_SIMPLE_DIVISIONAL ? R(-3.0) : R(-0.7), hi = -lo;
    real k1 = c->k1 + delta[0];
    c->k1 = k1 < lo ? lo : (k1 > hi ? hi : k1);
    /* one-parameter models: `dist` is _data[..., 6:] (two slots, camera.py:580-582, 803-805), and
     * update_dist broadcasts the single delta over both, so slot 7 shadows k1 (unused by the math) */
    real k2 = c->k2 + (model == CAM_RADIAL ? delta[1] : delta[0]);
    c->k2 = k2 < lo ? lo : (k2 > hi ? hi : k2);
}

/* python-style index into a delta row of length n (negative wraps) */
static real delta_at(const real *delta, int n, int idx) {
    if (idx < 0) idx += n;
    return (idx >= 0 && idx < n) ? delta[idx] : 0;
}

/* lm_optimizer.py:518-549 */
static void apply_update(const oracle_conf *cf, const plan_t *pl, cam_t *c, vec3 *g,
                         const real *delta, int n) {
    real d0 = 0, d1 = 0, df = 0;
    if (pl->est_grav) { d0 = delta_at(delta, n, pl->grav_dims[0]); d1 = delta_at(delta, n, pl->grav_dims[1]); }
    *g = grav_update(*g, d0, d1, cf->use_spherical_manifold);
    if (pl->est_focal) df = delta_at(delta, n, pl->focal_dim);
    update_focal(c, df, cf->use_log_focal);
    if (pl->has_dist && pl->est_dist) {
        real dd[2] = {delta_at(delta, n, pl->dist_dims[0]),
                      pl->ndist > 1 ? delta_at(delta, n, pl->dist_dims[1]) : 0};
        update_dist(cf->camera_model, c, dd);
    }
}

/* lm_optimizer.py:20-58 + camera.py:49-93 */
static void oracle_init(const oracle_conf *cf, const oracle_data *d, int b, cam_t *c, vec3 *g) {
    real h = (real)d->H, w = (real)d->W;
    real focal = d->prior_focal ? (real)d->prior_focal[b] : R(0.7) * (h > w ? h : w);
    real vfov = R(2.0) * ratan(h / (R(2.0) * focal));          /* focal2fov */
    real f = h / R(2.0) / rtan(vfov / R(2.0));                 /* fov2focal */
    memset(c, 0, sizeof(*c));
    c->w = w; c->h = h; c->fx = f; c->fy = f; c->cx = w / R(2.0); c->cy = h / R(2.0);
    if (d->scales) c->fx = c->fx * (real)d->scales[0] / (real)d->scales[1];
    int nd = num_dist(cf->camera_model);
    if (d->prior_dist) {
        c->k1 = (real)d->prior_dist[b * nd];
        if (nd > 1) c->k2 = (real)d->prior_dist[b * nd + 1];
    }
    *g = from_rp(0, 0);
    if (cf->heuristic_init && d->up && d->lat) {
        /* siclib get_heuristic_estimation: roll from the up vector at the centre, pitch = latitude at the
         * centre, vfov = |lat(top) - lat(bottom)| on the central column; clamps as there */
        const int N = d->H * d->W, yc = d->H / 2, xc = d->W / 2;
        const float *up = d->up + (size_t)b * 2 * N, *lat = d->lat + (size_t)b * N;
        const real d45 = R(45.0) / R(180.0) * (real)M_PI;
#ifdef ORACLE_F64
        real roll = -atan2((real)up[yc * d->W + xc], -(real)up[N + yc * d->W + xc]);
#else
        real roll = -atan2f(up[yc * d->W + xc], -up[N + yc * d->W + xc]);
#endif
        roll = roll < -d45 ? -d45 : (roll > d45 ? d45 : roll);
        real pitch = (real)lat[yc * d->W + xc];
        pitch = pitch < -d45 ? -d45 : (pitch > d45 ? d45 : pitch);
        real vf = rfabs((real)lat[xc] - (real)lat[(d->H - 1) * d->W + xc]);
        const real lo = R(20.0) / R(180.0) * (real)M_PI, hi = R(120.0) / R(180.0) * (real)M_PI;
        vf = vf < lo ? lo : (vf > hi ? hi : vf);
        if (!d->prior_focal) {
            real fh = h / R(2.0) / rtan(vf / R(2.0));
            c->fy = fh;
            c->fx = d->scales ? fh * (real)d->scales[0] / (real)d->scales[1] : fh;
        }
        *g = from_rp(roll, pitch);
    }
    if (d->prior_gravity) {
        vec3 pg = {(real)d->prior_gravity[b * 3], (real)d->prior_gravity[b * 3 + 1], (real)d->prior_gravity[b * 3 + 2]};
        *g = normalize3(pg);
    }
}

/* general inverse by Gauss-Jordan with partial pivoting (torch.inverse) */
static void invert(int n, const double *A, double *inv) {
    double M[MAXP][2 * MAXP];
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) { M[i][j] = A[i * n + j]; M[i][n + j] = i == j; }
    for (int col = 0; col < n; ++col) {
        int piv = col;
        for (int r = col + 1; r < n; ++r) if (fabs(M[r][col]) > fabs(M[piv][col])) piv = r;
        if (piv != col) for (int j = 0; j < 2 * n; ++j) { double t = M[col][j]; M[col][j] = M[piv][j]; M[piv][j] = t; }
        double dv = M[col][col];
        for (int j = 0; j < 2 * n; ++j) M[col][j] /= dv;
        for (int r = 0; r < n; ++r) if (r != col) {
            double f = M[r][col];
            for (int j = 0; j < 2 * n; ++j) M[r][j] -= f * M[col][j];
        }
    }
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) inv[i * n + j] = M[i][n + j];
}

/*
 * info layout per image (INFO_STRIDE floats):
 *  0 stop_at, 1 initial_up_cost, 2 initial_latitude_cost, 3 initial_cost,
 *  4 final_up_cost, 5 final_latitude_cost, 6 final_cost,
 *  7 roll_unc, 8 pitch_unc, 9 gravity_unc, 10 focal_unc, 11 vfov_unc, 12 n_params,
 *  13 final lambda, 16.. covariance (P x P row-major)
 */
int lm_oracle_info_stride(void) { return INFO_STRIDE; }
int lm_oracle_real_bytes(void) { return (int)sizeof(real); }

static double total_cost(const oracle_data *d, const sys_t *s) {
    /* sum(c.mean(-1) for c in costs.values()) in float like the reference */
    real t = 0;
    if (d->up) t += (real)s->cost_up;
    if (d->lat) t += (real)s->cost_lat;
    return t;
}

/* Optional per-step trace: for step i and image b, TRACE_STRIDE doubles:
 * [cost_up, cost_lat, lambda, G(5), H(25), delta(5), cam fx fy k1 k2, g xyz] */
#define TRACE_STRIDE 48
int lm_oracle_trace_stride(void) { return TRACE_STRIDE; }

int lm_oracle_solve(const oracle_conf *cf, const oracle_data *d, float *cam_out /*B x 8*/,
                    float *grav_out /*B x 3*/, float *info /*B x INFO_STRIDE*/,
                    double *trace /* num_steps x B x TRACE_STRIDE or NULL */) {
    const int B = d->B;
    plan_t pl;
    make_plan(cf, d, &pl);
    const int P = pl.n_params;
    if (P <= 0 || (!d->up && !d->lat)) return -1;
    if (cf->shared_intrinsics && !(pl.est_grav && pl.est_focal)) return -2;
#ifdef _OPENMP
    if (cf->num_threads > 0) omp_set_num_threads(cf->num_threads);
#endif
    cam_t *cam = (cam_t *)malloc(sizeof(cam_t) * B);
    vec3 *grav = (vec3 *)malloc(sizeof(vec3) * B);
    sys_t *sys = (sys_t *)malloc(sizeof(sys_t) * B);
    real *lamb = (real *)malloc(sizeof(real) * B);
    real *prev_cost = (real *)malloc(sizeof(real) * B);
    real *new_cost = (real *)malloc(sizeof(real) * B);
    real *delta = (real *)calloc((size_t)B * MAXP, sizeof(real));
    memset(info, 0, sizeof(float) * (size_t)B * INFO_STRIDE);
    for (int b = 0; b < B; ++b) { oracle_init(cf, d, b, &cam[b], &grav[b]); lamb[b] = (real)cf->lambda0; }

    int stop_at = cf->num_steps;
    /* pass at theta_0 */
#pragma omp parallel for schedule(dynamic, 1)
    for (int b = 0; b < B; ++b) image_pass(cf, d, &pl, b, &cam[b], grav[b], 0, 1, &sys[b]);
    for (int b = 0; b < B; ++b) {
        prev_cost[b] = (real)total_cost(d, &sys[b]);
        info[b * INFO_STRIDE + 1] = (float)sys[b].cost_up;
        info[b * INFO_STRIDE + 2] = (float)sys[b].cost_lat;
        info[b * INFO_STRIDE + 3] = (float)prev_cost[b];
    }
    for (int it = 0; it < cf->num_steps; ++it) {
        /* ---- delta from the system at theta_it (:590-603) */
        if (!cf->shared_intrinsics) {
            int any_fail = 0;
            for (int b = 0; b < B; ++b) {
                real Hm[MAXP * MAXP], G[MAXP];
                for (int k = 0; k < P; ++k) { G[k] = (real)sys[b].G[k]; for (int l = 0; l < P; ++l) Hm[k * P + l] = (real)sys[b].Hm[k][l]; }
                if (!lm_step(P, Hm, G, lamb[b], &delta[b * MAXP])) any_fail = 1;
            }
            if (any_fail) memset(delta, 0, sizeof(real) * (size_t)B * MAXP);  /* :129-133 */
        } else {
            /* arrow-head system (:350-383): [2B gravity dims | n_intrinsic] */
            const int ni = pl.n_intrinsic, n = 2 * B + ni;
            real *A = (real *)calloc((size_t)n * n, sizeof(real));
            real *g = (real *)calloc(n, sizeof(real));
            real *dl = (real *)calloc(n, sizeof(real));
            for (int b = 0; b < B; ++b) {
                for (int i = 0; i < 2; ++i) {
                    g[2 * b + i] = (real)sys[b].G[i];
                    for (int j = 0; j < 2; ++j) A[(2 * b + i) * n + 2 * b + j] = (real)sys[b].Hm[i][j];
                    for (int j = 0; j < ni; ++j) {
                        A[(2 * b + i) * n + 2 * B + j] = (real)sys[b].Hm[i][2 + j];
                        A[(2 * B + j) * n + 2 * b + i] = (real)sys[b].Hm[2 + j][i];
                    }
                }
                for (int i = 0; i < ni; ++i) {
                    g[2 * B + i] += (real)sys[b].G[2 + i];
                    for (int j = 0; j < ni; ++j) A[(2 * B + i) * n + 2 * B + j] += (real)sys[b].Hm[2 + i][2 + j];
                }
            }
            if (!lm_step(n, A, g, lamb[0], dl)) memset(dl, 0, sizeof(real) * n);
            for (int b = 0; b < B; ++b) {   /* :599-603 */
                delta[b * MAXP + 0] = dl[2 * b]; delta[b * MAXP + 1] = dl[2 * b + 1];
                for (int j = 0; j < ni; ++j) delta[b * MAXP + 2 + j] = dl[2 * B + j];
            }
            free(A); free(g); free(dl);
        }
        if (trace)
            for (int b = 0; b < B; ++b) {
                double *t = trace + ((size_t)it * B + b) * TRACE_STRIDE;
                t[0] = sys[b].cost_up; t[1] = sys[b].cost_lat; t[2] = lamb[cf->shared_intrinsics ? 0 : b];
                for (int k = 0; k < MAXP; ++k) { t[3 + k] = sys[b].G[k]; t[33 + k] = delta[b * MAXP + k]; for (int l = 0; l < MAXP; ++l) t[8 + k * MAXP + l] = sys[b].Hm[k][l]; }
            }
        /* ---- update (:606) and evaluate at theta_{it+1} (:607-610; reused as next system) */
        int last = it == cf->num_steps - 1;
        for (int b = 0; b < B; ++b)
            apply_update(cf, &pl, &cam[b], &grav[b], &delta[b * MAXP], cf->shared_intrinsics ? P : P);
        if (trace)
            for (int b = 0; b < B; ++b) {
                double *t = trace + ((size_t)it * B + b) * TRACE_STRIDE;
                t[38] = cam[b].fx; t[39] = cam[b].fy; t[40] = cam[b].k1; t[41] = cam[b].k2;
                t[42] = grav[b].x; t[43] = grav[b].y; t[44] = grav[b].z;
            }
        (void)last;
#pragma omp parallel for schedule(dynamic, 1)
        for (int b = 0; b < B; ++b) image_pass(cf, d, &pl, b, &cam[b], grav[b], 0, 1, &sys[b]);
        int all_close = 1;
        for (int b = 0; b < B; ++b) {
            new_cost[b] = (real)total_cost(d, &sys[b]);
            if (!cf->fix_lambda && !cf->shared_intrinsics) {          /* :95-106, :612-613 */
                real nl = lamb[b] * (new_cost[b] > prev_cost[b] ? R(10.0) : R(0.1));
                lamb[b] = nl < R(1e-6) ? R(1e-6) : (nl > R(1e2) ? R(1e2) : nl);
            }
            /* torch.allclose(new, prev) (:90-92) evaluates |new - prev| <= atol + |rtol * prev| in the TENSORS' dtype (the
             * tolerances are scalars and do not promote: ATen TensorCompare.cpp isclose): float32 in the reference */
            volatile real scaled = (real)cf->rtol * prev_cost[b];      /* volatile: no fma contraction of the two roundings */
            real diff = rfabs(new_cost[b] - prev_cost[b]);
            real allowed = (real)cf->atol + rfabs(scaled);
            if (!(diff <= allowed)) all_close = 0;
        }
        int brk = 0;
        if (all_close) {                                              /* :619-625 */
            if (it + 1 < stop_at) stop_at = it + 1;
            if (cf->early_stop) brk = 1;
        }
        memcpy(prev_cost, new_cost, sizeof(real) * B);
        if (brk) break;
    }
    /* ---- final costs (:632-642) and uncertainty (:463-516) */
#pragma omp parallel for schedule(dynamic, 1)
    for (int b = 0; b < B; ++b) image_pass(cf, d, &pl, b, &cam[b], grav[b], 1, !cf->training, &sys[b]);
    for (int b = 0; b < B; ++b) {
        float *o = info + (size_t)b * INFO_STRIDE;
        o[0] = (float)stop_at;
        o[4] = (float)sys[b].cost_up; o[5] = (float)sys[b].cost_lat; o[6] = (float)total_cost(d, &sys[b]);
        o[12] = (float)P;
        o[13] = (float)lamb[cf->shared_intrinsics ? 0 : b];
        if (!cf->training) {
            double Hd[MAXP * MAXP], Cov[MAXP * MAXP];
            for (int k = 0; k < P; ++k) for (int l = 0; l < P; ++l) Hd[k * P + l] = (double)(real)sys[b].Hm[k][l];
            invert(P, Hd, Cov);
            for (int k = 0; k < P * P; ++k) o[16 + k] = (float)Cov[k];
            if (pl.est_grav) {
                double c00 = Cov[0], c11 = Cov[P + 1], c01 = 0.5 * (Cov[1] + Cov[P]);
                o[7] = (float)sqrt(c00); o[8] = (float)sqrt(c11);
                double tr = 0.5 * (c00 + c11), df = 0.5 * (c00 - c11);
                o[9] = (float)sqrt(tr + sqrt(df * df + c01 * c01));   /* max eigvalsh of 2x2 */
            }
            if (pl.est_focal) {
                int fd = pl.focal_dim;
                double fu = Cov[fd * P + fd];
                double fy = cam[b].fy, hh = cam[b].h;
                double Jf = -4.0 * hh / (4.0 * fy * fy + hh * hh);   /* misc.py:285-287 */
                o[10] = (float)(sqrt(fu) / 2.0);
                o[11] = (float)sqrt(Jf * Jf * fu / 2.0);
            }
        }
        float *co = cam_out + (size_t)b * 8;
        co[0] = (float)cam[b].w; co[1] = (float)cam[b].h; co[2] = (float)cam[b].fx; co[3] = (float)cam[b].fy;
        co[4] = (float)cam[b].cx; co[5] = (float)cam[b].cy; co[6] = (float)cam[b].k1; co[7] = (float)cam[b].k2;
        grav_out[b * 3] = (float)grav[b].x; grav_out[b * 3 + 1] = (float)grav[b].y; grav_out[b * 3 + 2] = (float)grav[b].z;
    }
    free(cam); free(grav); free(sys); free(lamb); free(prev_cost); free(new_cost); free(delta);
    return 0;
}

/* Render the perspective field of a camera (perspective_fields.py:278-320): up (2,H,W), lat (1,H,W). */
int lm_oracle_render(int model, int H, int W, const float *cam8, const float *grav3, float *up, float *lat) {
    cam_t c = {cam8[0], cam8[1], cam8[2], cam8[3], cam8[4], cam8[5], cam8[6], cam8[7]};
    vec3 g = {grav3[0], grav3[1], grav3[2]};
    real T[3][2] = {{0}};
    const int N = H * W;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            pix_t px;
            real z2[2] = {0, 0}, z1 = 0;
            pixel_eval(model, &c, g, T, 0, (real)x, (real)y, z2, &z1, 0, &px);
            up[y * W + x] = (float)-px.r_up[0];
            up[N + y * W + x] = (float)-px.r_up[1];
            lat[y * W + x] = (float)rasin(-px.r_lat);   /* sin(0) - sin(asin(clamp s)) */
        }
    return 0;
}

/* Per-pixel Jacobians of the predicted fields wrt (d1, d2, focal[, k1[, k2]]) (perspective_fields.py:323-365:
 * J_up_field :84-182, J_latitude_field :214-275): J_up (H,W,2,P), J_lat (H,W,1,P), P = 3 + #dist; returns P. */
int lm_oracle_jacobians(int model, int H, int W, const float *cam8, const float *grav3, int spherical,
                        int log_focal, double *J_up, double *J_lat) {
    cam_t c = {cam8[0], cam8[1], cam8[2], cam8[3], cam8[4], cam8[5], cam8[6], cam8[7]};
    vec3 g = {grav3[0], grav3[1], grav3[2]};
    real T[3][2];
    const int P = 3 + num_dist(model);
    if (spherical) J_plus(g, T); else J_rp(g, T);
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            pix_t px;
            real z2[2] = {0, 0}, z1 = 0;
            pixel_eval(model, &c, g, T, log_focal, (real)x, (real)y, z2, &z1, 1, &px);
            for (int k = 0; k < P; ++k) {
                J_up[((size_t)(y * W + x) * 2 + 0) * P + k] = (double)px.J_up[0][k];
                J_up[((size_t)(y * W + x) * 2 + 1) * P + k] = (double)px.J_up[1][k];
                J_lat[(size_t)(y * W + x) * P + k] = (double)px.J_lat[k];
            }
        }
    return P;
}

/* calculate_residuals (lm_optimizer.py:248-274) of one image: r_up (N,2) = up_data - up(theta), r_lat (N) =
 * sin(lat_data) - sin(lat(theta)); either field / output pair may be NULL. */
int lm_oracle_residuals(int model, int H, int W, const float *cam8, const float *grav3, const float *up /*2,H,W*/,
                        const float *lat /*H,W*/, double *r_up, double *r_lat) {
    cam_t c = {cam8[0], cam8[1], cam8[2], cam8[3], cam8[4], cam8[5], cam8[6], cam8[7]};
    vec3 g = {grav3[0], grav3[1], grav3[2]};
    real T[3][2] = {{0}};
    const int N = H * W;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const int i = y * W + x;
            pix_t px;
            real dup[2] = {0, 0}, dlat = 0;
            if (up) { dup[0] = up[i]; dup[1] = up[N + i]; }
            if (lat) dlat = lat[i];
            pixel_eval(model, &c, g, T, 0, (real)x, (real)y, dup, &dlat, 0, &px);
            if (up && r_up) { r_up[2 * i] = (double)px.r_up[0]; r_up[2 * i + 1] = (double)px.r_up[1]; }
            if (lat && r_lat) r_lat[i] = (double)px.r_lat;
        }
    return 0;
}

/* calculate_costs (lm_optimizer.py:276-315) on n squared residual norms: cost and weight, times conf if given. */
int lm_oracle_huber_costs(const double *x2, int n, double scale, const float *conf, double *cost, double *weight) {
    for (int i = 0; i < n; ++i) {
        real w, c = huber((real)x2[i], scale, &w);
        const real cf = conf ? (real)conf[i] : R(1.0);
        cost[i] = (double)(c * cf);
        weight[i] = (double)(w * cf);
    }
    return 0;
}

/* Single-pass system at given parameters (used by kernel-level parity tests). */
int lm_oracle_system(const oracle_conf *cf, const oracle_data *d, const float *cam8 /*B x 8*/,
                     const float *grav3 /*B x 3*/, int as_rpf, double *cost_up, double *cost_lat,
                     double *G /*B x MAXP*/, double *Hm /*B x MAXP x MAXP*/) {
    plan_t pl;
    make_plan(cf, d, &pl);
    if (cf->num_threads > 0) omp_set_num_threads(cf->num_threads);
#pragma omp parallel for schedule(dynamic, 1)
    for (int b = 0; b < d->B; ++b) {
        cam_t c = {cam8[b * 8], cam8[b * 8 + 1], cam8[b * 8 + 2], cam8[b * 8 + 3], cam8[b * 8 + 4], cam8[b * 8 + 5], cam8[b * 8 + 6], cam8[b * 8 + 7]};
        vec3 g = {grav3[b * 3], grav3[b * 3 + 1], grav3[b * 3 + 2]};
        sys_t s;
        image_pass(cf, d, &pl, b, &c, g, as_rpf, 1, &s);
        cost_up[b] = s.cost_up; cost_lat[b] = s.cost_lat;
        memcpy(G + (size_t)b * MAXP, s.G, sizeof(s.G));
        memcpy(Hm + (size_t)b * MAXP * MAXP, s.Hm, sizeof(s.Hm));
    }
    return pl.n_params;
}
