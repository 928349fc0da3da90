/* mate_oracle.c -- CPU restatement (plain C, f64) of the MultiAgentTracking
 * step path of the upstream reference.  TEST INFRASTRUCTURE ONLY (see
 * mate_oracle.h): parity oracle + CPU baseline, never the product path.
 *
 * Every function cites the reference file:line it restates.  Arithmetic is
 * kept operation-for-operation (build with -ffp-contract=off): numpy evaluates
 * each ufunc separately, except 2-element dot products, which the container's
 * OpenBLAS evaluates as fma(a1, b1, a0*b0) (probed) -- that is
 * what dot2()/norm2() below do so that the f64 golden vectors match to the ulp
 * wherever no libm transcendental is involved.
 */
#include "mate_oracle.h"

#include <math.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

#define TERRAIN_SIZE 1000.0          /* constants.py:52 */
#define TERRAIN_WIDTH 2000.0         /* constants.py:55 */
#define WAREHOUSE_RADIUS 75.0        /* constants.py:67 */
#define MAX_VIEWING_ANGLE 180.0      /* constants.py:78 */
#define NUM_RESET_RETRIES 500        /* environment.py:53 */
#define RAD2DEG (180.0 / 3.14159265358979323846)  /* utils.py:63 */
#define DEG2RAD (3.14159265358979323846 / 180.0)  /* utils.py:68 */
#define PRESERVED_DIM 13             /* constants.py:95 */

static const double WAREHOUSES[MO_NW][2] = {  /* constants.py:70-72 */
    {+925.0, +925.0}, {-925.0, +925.0}, {-925.0, -925.0}, {+925.0, -925.0}};

enum { S_TRANSMIT = 1, S_GOAL = 2, S_ACT_CAM = 3, S_ACT_TGT = 4, S_RESET = 5, S_RESET_VIEW = 6 };

struct mo_env {
    int Nc, Nt, No;
    /* ---- configuration (environment.py:113-269, 1396-1544) ---- */
    double tau;                 /* obstacle transmittance */
    int max_episode_steps;
    int sparse_reward;
    int num_cargoes_per_target;
    int shuffle_entities;
    int targets_start_with_cargoes;
    double high_capacity_target_split;
    double target_step_size;    /* config target/step_size */
    double target_sight_range;
    double freight_scale, bounty_scale, reward_scale, max_team_reward; /* environment.py:521-529 */
    double cam_range[MO_MAXC][4], tgt_range[MO_MAXT][4], obs_range[MO_MAXO][4]; /* xlo xhi ylo yhi */
    double obs_radius_range[2];
    double cfg_cam_radius, cfg_cam_theta_min, cfg_cam_rmax, cfg_cam_rot, cfg_cam_zoom;
    /* ---- per-episode static state ---- */
    double cam_x[MO_MAXC], cam_y[MO_MAXC], cam_r[MO_MAXC];
    double cam_theta_min[MO_MAXC], cam_rmax[MO_MAXC], cam_rot[MO_MAXC], cam_zoom[MO_MAXC];
    double obs_x[MO_MAXO], obs_y[MO_MAXO], obs_r[MO_MAXO];
    int tgt_capacity[MO_MAXT];
    double tgt_step[MO_MAXT], tgt_sight[MO_MAXT];
    unsigned char cam_obs_mask[MO_MAXC][MO_MAXO];   /* environment.py:752-755 */
    int lut_n[2][MO_MAXC];
    double *lut_phi[2][MO_MAXC], *lut_rho[2][MO_MAXC]; /* [inner|outer] */
    /* ---- dynamic state ---- */
    double cam_phi[MO_MAXC], cam_theta[MO_MAXC], cam_sight[MO_MAXC];
    double tgt_x[MO_MAXT], tgt_y[MO_MAXT];
    unsigned char tgt_colliding[MO_MAXT];
    unsigned char empty_bits[MO_MAXT][MO_NW];
    int goal_bits[MO_MAXT][MO_NW];
    int goals[MO_MAXT];
    int freights[MO_MAXT], bounties[MO_MAXT];
    int target_steps[MO_MAXT], tracked_steps[MO_MAXT];
    int remaining[MO_NW][MO_NW], awaiting[MO_NW];
    int num_delivered;
    double episode_reward, delayed_episode_reward;
    int episode_step;
    /* ---- per-step outputs ---- */
    unsigned char m_ct[MO_MAXC][MO_MAXT], m_tc[MO_MAXT][MO_MAXC], m_to[MO_MAXT][MO_MAXO];
    unsigned char m_tt[MO_MAXT][MO_MAXT], m_cc[MO_MAXC][MO_MAXC];
    unsigned char tracked[MO_MAXT], target_dones[MO_MAXT];
    double tw_dist[MO_MAXT][MO_NW];
    double coverage_rate, real_coverage_rate, mean_transport_rate;
    double reward_cam, reward_tgt, reward_dense, reward_delayed, normalized_reward_tgt;
    int done;
    /* ---- RNG identity ---- */
    uint64_t seed;
    uint32_t env_index, tick, episode;
    uint32_t reset_draws;
    /* tape mode of reset (mo_reset_tape): the uniforms the reference's RNG proxies logged, in call order */
    const double *reset_tape;
    uint32_t reset_tape_n;
    int reset_tape_overrun;
};

/* ======================================================================== RNG */
void mo_philox4x32(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t out[4]) {
    /* Philox-4x32-10 (Salmon et al., SC'11), the published algorithm. */
    for (int round = 0; round < 10; ++round) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

static inline double u53(uint32_t hi, uint32_t lo) {
    return ((double)(hi >> 5) * 67108864.0 + (double)(lo >> 6)) * (1.0 / 9007199254740992.0);
}

/* A tick-keyed draw of the engine's Philox streams: ticks 2k and 2k + 1 share the block with counter k and take its first /
 * second 64 bits (mate_amd/csrc/engine_kernels.hpp, Ctx::draw). */
static double draw_stream(const mo_env *e, uint32_t tick, uint32_t stream, uint32_t sub) {
    uint32_t r[4];
    mo_philox4x32((uint32_t)e->seed, (uint32_t)(e->seed >> 32), e->env_index, tick >> 1, stream, sub, r);
    return (tick & 1u) ? u53(r[2], r[3]) : u53(r[0], r[1]);
}

static double draw_reset(mo_env *e) {
    uint32_t r[4];
    uint32_t idx = e->reset_draws++;
    if (e->reset_tape) {
        if (idx >= e->reset_tape_n) { e->reset_tape_overrun = 1; return 0.0; }
        return e->reset_tape[idx];
    }
    mo_philox4x32((uint32_t)e->seed, (uint32_t)(e->seed >> 32), e->env_index, e->episode, S_RESET, idx >> 1, r);
    return (idx & 1) ? u53(r[2], r[3]) : u53(r[0], r[1]);
}

static int randint_reset(mo_env *e, int n) {
    int j = (int)(draw_reset(e) * (double)n);
    return j >= n ? n - 1 : j;
}

void mo_random_actions(uint64_t seed, uint64_t env_index, uint64_t tick, int Nc, int Nt, double rot_step,
                       double zoom_step, double step_size, float *cam_act, float *tgt_act) {
    uint32_t r[4];
    const int h = ((uint32_t)tick & 1u) ? 2 : 0;      /* the half of the block this tick takes (draw_stream) */
    for (int c = 0; c < Nc; ++c) {
        mo_philox4x32((uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)env_index, (uint32_t)tick >> 1, S_ACT_CAM, (uint32_t)c, r);
        cam_act[2 * c + 0] = (float)((double)(r[h] >> 8) * 5.9604644775390625e-08 * (2.0 * rot_step) - rot_step);
        cam_act[2 * c + 1] = (float)((double)(r[h + 1] >> 8) * 5.9604644775390625e-08 * (2.0 * zoom_step) - zoom_step);
    }
    for (int t = 0; t < Nt; ++t) {
        mo_philox4x32((uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)env_index, (uint32_t)tick >> 1, S_ACT_TGT, (uint32_t)t, r);
        tgt_act[2 * t + 0] = (float)((double)(r[h] >> 8) * 5.9604644775390625e-08 * (2.0 * step_size) - step_size);
        tgt_act[2 * t + 1] = (float)((double)(r[h + 1] >> 8) * 5.9604644775390625e-08 * (2.0 * step_size) - step_size);
    }
}

/* ================================================================ math utils */
static inline double dot2(double ax, double ay, double bx, double by) { return fma(ay, by, ax * bx); }
static inline double norm2(double x, double y) { return sqrt(fma(y, y, x * x)); }  /* np.linalg.norm of a 2-vector */
static inline double pymod(double a, double b) { /* Python / numpy float % */
    double m = fmod(a, b);
    if (m != 0.0) {
        if ((b < 0.0) != (m < 0.0)) m += b;
    } else {
        m = copysign(0.0, b);
    }
    return m;
}
static inline double clipd(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }

double mo_normalize_angle(double angle) { return pymod(angle + 180.0, 360.0) - 180.0; } /* utils.py:155-158 */
static inline double atan2_deg(double y, double x) { return atan2(y, x) * RAD2DEG; }      /* utils.py:124-131 */

/* Vector2D with its lazy polar/cartesian caches (utils.py:161-271). */
typedef struct {
    double ox, oy;
    int has_v, has_n, has_a;
    double vx, vy, n, a;
} ray_t;

static void ray_from_vector(ray_t *r, double vx, double vy, double ox, double oy) {
    r->ox = ox; r->oy = oy; r->has_v = 1; r->vx = vx; r->vy = vy; r->has_n = r->has_a = 0; r->n = r->a = 0.0;
}
static void ray_materialize(ray_t *r) { /* utils.py:177-181,144-152 */
    if (!r->has_v) {
        double phi_rad = r->a * DEG2RAD;
        r->vx = r->n * cos(phi_rad);
        r->vy = r->n * sin(phi_rad);
        r->has_v = 1;
    }
}
static double ray_angle(ray_t *r) { /* utils.py:206-210 */
    if (!r->has_a) { r->a = atan2_deg(r->vy, r->vx); r->has_a = 1; }
    return r->a;
}
static double ray_norm(ray_t *r) { /* utils.py:217-221 */
    if (!r->has_n) { r->n = norm2(r->vx, r->vy); r->has_n = 1; }
    return r->n;
}
static void ray_set_angle(ray_t *r, double value) { /* utils.py:212-215 */
    r->a = mo_normalize_angle(value); r->has_a = 1; r->has_v = 0;
}
static void ray_set_norm(ray_t *r, double value) { /* utils.py:223-229 */
    double angle = ray_angle(r);
    r->n = fabs(value); r->has_n = 1; r->has_v = 0;
    if (value < 0.0) ray_set_angle(r, angle + 180.0);
}
static void ray_from_polar(ray_t *r, double norm, double angle, double ox, double oy) { /* utils.py:171-173 */
    r->ox = ox; r->oy = oy; r->has_v = r->has_n = r->has_a = 0; r->vx = r->vy = 0.0;
    r->a = mo_normalize_angle(angle); r->has_a = 1;
    r->n = fabs(norm); r->has_n = 1;
    if (norm < 0.0) ray_set_angle(r, r->a + 180.0);
}
static void ray_set_vector(ray_t *r, double vx, double vy) { r->vx = vx; r->vy = vy; r->has_v = 1; r->has_n = r->has_a = 0; }

/* Obstacle.obstruct (entities.py:158-184): clip `ray` by the circle (cx,cy,rad). */
static void obstruct(ray_t *ray, double cx, double cy, double rad, int keep_tangential, int outer) {
    double relx = cx - ray->ox, rely = cy - ray->oy;
    double rel_norm = norm2(relx, rely);
    double norm = ray_norm(ray);
    if (norm == 0.0 || rel_norm < rad) { /* return -ray */
        ray_materialize(ray);
        ray_set_vector(ray, -ray->vx, -ray->vy);
        return;
    }
    if (rel_norm >= norm + rad) return;
    ray_materialize(ray);
    double inner = dot2(relx, rely, ray->vx, ray->vy);
    if (inner >= 0.0) {
        double c = inner / (rel_norm * norm);
        double cosv = c < 1.0 ? c : 1.0;
        double perpendicular = rel_norm * sqrt(1.0 - cosv * cosv);
        if (rad > perpendicular) {
            double half_chord = sqrt(rad * rad - perpendicular * perpendicular);
            double cand = outer ? rel_norm * cosv + half_chord : rel_norm * cosv - half_chord;
            double new_norm = cand > 0.0 ? cand : 0.0;
            if (new_norm < norm) {
                double oldx = ray->vx, oldy = ray->vy;
                ray_set_norm(ray, new_norm);
                if (keep_tangential) {
                    ray_materialize(ray);
                    double rx = (ray->ox + ray->vx) - cx, ry = (ray->oy + ray->vy) - cy;
                    double s = (norm - new_norm) * half_chord / (rad * rad);
                    ray_set_vector(ray, oldx + rx * s, oldy + ry * s);
                }
            }
        }
    }
}

void mo_obstruct(double ox, double oy, double vx, double vy, double cx, double cy, double r, int keep_tangential,
                 int outer, double out[2]) {
    ray_t ray;
    ray_from_vector(&ray, vx, vy, ox, oy);
    obstruct(&ray, cx, cy, r, keep_tangential, outer);
    ray_materialize(&ray);
    out[0] = ray.vx; out[1] = ray.vy;
}

void mo_clamp_step(double ax, double ay, double step_size, double out[2]) { /* entities.py:648-650 */
    ray_t step;
    ray_from_vector(&step, ax, ay, 0.0, 0.0);
    if (ray_norm(&step) > step_size) ray_set_norm(&step, step_size);
    ray_materialize(&step);
    out[0] = step.vx; out[1] = step.vy;
}

void mo_camera_simulate(double phi, double theta, double dphi, double dtheta, double theta_min, double rmax,
                        double rot_step, double zoom_step, double out[3]) { /* entities.py:347-360 */
    double da = clipd(dphi, -rot_step, rot_step);
    double dv = clipd(dtheta, -zoom_step, zoom_step);
    double nphi = mo_normalize_angle(phi + da);
    double ntheta = clipd(theta + dv, theta_min, MAX_VIEWING_ANGLE);
    double area = theta_min * (rmax * rmax);       /* entities.py:285 */
    out[0] = nphi; out[1] = ntheta; out[2] = sqrt(area / ntheta);
}

/* np.interp on a scalar (numpy compiled_base.c arr_interp), which is what
 * scipy.interpolate.interp1d(kind='linear') dispatches to (entities.py:476,511). */
double mo_interp(const double *xp, const double *fp, int n, double x) {
    if (x != x) return x;
    if (x > xp[n - 1]) return fp[n - 1];
    if (x < xp[0]) return fp[0];
    int lo = 0, hi = n; /* largest j with xp[j] <= x */
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (xp[mid] <= x) lo = mid; else hi = mid;
    }
    int j = lo;
    if (j == n - 1) return fp[j];
    if (xp[j] == x) return fp[j];
    double slope = (fp[j + 1] - fp[j]) / (xp[j + 1] - xp[j]);
    double res = slope * (x - xp[j]) + fp[j];
    if (res != res) {
        res = slope * (x - xp[j + 1]) + fp[j + 1];
        if (res != res && fp[j] == fp[j + 1]) res = fp[j];
    }
    return res;
}

/* ------------------------------------------------------------------ LUT build */
typedef struct { double a, n; int seq; } knot_t;
static int knot_cmp(const void *pa, const void *pb) {
    const knot_t *a = (const knot_t *)pa, *b = (const knot_t *)pb;
    if (a->a < b->a) return -1;
    if (a->a > b->a) return 1;
    return a->seq - b->seq; /* list.sort is stable */
}

/* Camera.add_obstacles (entities.py:362-479) for one camera.  `outer` selects
 * boundary_outer.  Obstacles are applied in index order (the reference iterates
 * a Python set; the order only perturbs last-place rounding). Returns #knots. */
int mo_build_lut_raw(double cx, double cy, double rmax, const double *oxyr, int nobs, double tau, int outer,
                     double *phis, double *rhos, int cap) {
    int inrange[MO_MAXO], nin = 0;
    for (int o = 0; o < nobs; ++o) { /* entities.py:363-368 */
        double d = norm2(cx - oxyr[3 * o], cy - oxyr[3 * o + 1]);
        if (d < rmax + oxyr[3 * o + 2]) inrange[nin++] = o;
    }
    int maxrays = 360 + nin * (4 + 185 + 42) + 8;
    ray_t *rays = (ray_t *)malloc(sizeof(ray_t) * (size_t)maxrays);
    int nr = 0;
    for (int i = 0; i < 360; ++i) ray_from_polar(&rays[nr++], rmax, -180.0 + (double)i * 1.0, cx, cy); /* entities.py:336-343 */
    int degenerate = 0;
    for (int k = 0; k < nin && tau != 1.0; ++k) {
        int o = inrange[k];
        double ox = oxyr[3 * o], oy = oxyr[3 * o + 1], orad = oxyr[3 * o + 2];
        ray_t rel;
        ray_from_vector(&rel, ox - cx, oy - cy, cx, cy);
        double rel_norm = ray_norm(&rel);
        if (orad > rel_norm) { degenerate = 1; break; } /* entities.py:378-387 */
        double half = asin(orad / rel_norm) * RAD2DEG;   /* entities.py:389 */
        double far = rel_norm + orad;
        double max_rho = far < rmax ? far : rmax;         /* entities.py:390 */
        double rel_angle = ray_angle(&rel);
        double a_left = rel_angle - half, a_right = rel_angle + half;
        int num = (int)(2.0 * half);
        if (num < 16) num = 16;
        num += 1;                                          /* entities.py:413 */
        if (!outer) {
            ray_from_polar(&rays[nr++], rmax, a_left - 0.01, cx, cy);
            ray_from_polar(&rays[nr++], rmax, a_left + 0.01, cx, cy);
            ray_from_polar(&rays[nr++], rmax, a_right - 0.01, cx, cy);
            ray_from_polar(&rays[nr++], rmax, a_right + 0.01, cx, cy);
        }
        /* np.linspace(a_left, a_right, num, endpoint=True): i*step + start, last = stop */
        double step = (a_right - a_left) / (double)(num - 1);
        for (int i = 0; i < num; ++i) {
            double a = (i == num - 1) ? a_right : ((double)i * step + a_left);
            ray_from_polar(&rays[nr++], max_rho, a, cx, cy);
        }
        if (outer) { /* entities.py:431-448 */
            double nr2 = sqrt(rel_norm * rel_norm + orad * orad);
            double near_rho = nr2 < rmax ? nr2 : rmax;
            for (int side = 0; side < 2; ++side) {
                ray_t nearv, farv;
                ray_from_polar(&nearv, near_rho, side == 0 ? a_left : a_right, cx, cy);
                ray_from_polar(&farv, rmax, side == 0 ? a_left - 0.01 : a_right + 0.01, cx, cy);
                ray_materialize(&nearv); ray_materialize(&farv);
                for (int i = 0; i < 21; ++i) {
                    double t = (i == 20) ? 1.0 : (double)i * (1.0 / 20.0);
                    double x = (1.0 - t) * nearv.vx + t * farv.vx;
                    double y = (1.0 - t) * nearv.vy + t * farv.vy;
                    ray_from_vector(&rays[nr++], x, y, cx, cy);
                }
            }
        }
    }
    if (degenerate) { /* camera inside an obstacle: the view is fully blocked */
        free(rays);
        if (cap < 2) return -1;
        phis[0] = -180.0; rhos[0] = 0.0; phis[1] = 180.0; rhos[1] = 0.0;
        return 2;
    }
    if (tau != 1.0) {
        for (int k = 0; k < nin; ++k) { /* entities.py:450-455 */
            int o = inrange[k];
            for (int i = 0; i < nr; ++i) obstruct(&rays[i], oxyr[3 * o], oxyr[3 * o + 1], oxyr[3 * o + 2], 0, outer);
        }
    }
    knot_t *knots = (knot_t *)malloc(sizeof(knot_t) * (size_t)nr);
    for (int i = 0; i < nr; ++i) { knots[i].a = ray_angle(&rays[i]); knots[i].n = ray_norm(&rays[i]); knots[i].seq = i; }
    qsort(knots, (size_t)nr, sizeof(knot_t), knot_cmp); /* entities.py:458 */
    int m = 0;
    for (int i = 0; i < nr; ++i) {                       /* entities.py:460-466 */
        if (m > 0 && knots[m - 1].a == knots[i].a) {
            if (knots[m - 1].n > knots[i].n) knots[m - 1] = knots[i];
        } else {
            knots[m++] = knots[i];
        }
    }
    int total = m + 1;
    if (total > cap) { free(rays); free(knots); return -total; }
    for (int i = 0; i < m; ++i) { phis[i] = knots[i].a; rhos[i] = knots[i].n; }
    phis[m] = knots[0].a + 360.0; rhos[m] = knots[0].n;  /* entities.py:470-471 */
    free(rays); free(knots);
    return total;
}

/* Camera.perceive + sight_range_at (entities.py:491-511). */
int mo_camera_perceive(double cx, double cy, double phi, double theta, double sight, double px, double py,
                       double u, double tau, const double *lut_phi, const double *lut_rho, int lut_n) {
    double rx = px - cx, ry = py - cy;
    double rn = norm2(rx, ry);
    if (rn > sight) return 0;
    double ang = atan2_deg(ry, rx);
    double ra = fabs(phi - ang);
    double alt = 360.0 - ra;
    if (alt < ra) ra = alt;
    if (ra * 2.0 > theta) return 0;
    /* np_random.binomial(1, tau): one uniform, legacy inversion sampler */
    int see_through = (tau <= 0.5) ? (u > 1.0 - tau) : (u <= tau);
    if (see_through) return 1;
    double limit = mo_interp(lut_phi, lut_rho, lut_n, mo_normalize_angle(ang));
    return rn <= limit * (1.0 + 1e-6);
}

/* ================================================================ environment */
mo_env *mo_create(int Nc, int Nt, int No) {
    if (Nc < 0 || Nc > MO_MAXC || Nt < 1 || Nt > MO_MAXT || No < 0 || No > MO_MAXO) return NULL;
    mo_env *e = (mo_env *)calloc(1, sizeof(mo_env));
    e->Nc = Nc; e->Nt = Nt; e->No = No;
    e->max_episode_steps = 10000;
    e->num_cargoes_per_target = 8;
    e->shuffle_entities = 1;
    e->targets_start_with_cargoes = 1;
    e->high_capacity_target_split = 0.5;
    e->target_step_size = 20.0;
    e->target_sight_range = 500.0;
    e->freight_scale = 100.0; e->bounty_scale = 100.0; e->reward_scale = 200.0;
    e->max_team_reward = 200.0 * 8 * Nt;
    for (int t = 0; t < Nt; ++t) { e->goals[t] = -1; e->tgt_capacity[t] = 1; e->tgt_step[t] = 20.0; e->tgt_sight[t] = 500.0; }
    return e;
}

static void free_luts(mo_env *e) {
    for (int k = 0; k < 2; ++k)
        for (int c = 0; c < MO_MAXC; ++c) {
            free(e->lut_phi[k][c]); free(e->lut_rho[k][c]);
            e->lut_phi[k][c] = e->lut_rho[k][c] = NULL; e->lut_n[k][c] = 0;
        }
}
void mo_destroy(mo_env *e) { if (e) { free_luts(e); free(e); } }

int mo_camera_obs_dim(const mo_env *e) { return PRESERVED_DIM + 9 + 5 * e->Nt + 4 * e->No + 7 * e->Nc; } /* constants.py:267-282 */
int mo_target_obs_dim(const mo_env *e) { return PRESERVED_DIM + 14 + 7 * e->Nc + 4 * e->No + 5 * e->Nt; } /* constants.py:285-300 */
int mo_state_dim(const mo_env *e) { return PRESERVED_DIM + 9 * e->Nc + 14 * e->Nt + 3 * e->No + 2 * e->Nt + MO_NW * MO_NW; }

/* named field table */
enum { TD, TI, TB, TU32, TU64 };
typedef struct { const char *name; size_t off; int count; int type; } field_t;
#define F(name, member, count, type) {name, offsetof(mo_env, member), count, type}
static const field_t FIELDS[] = {
    F("transmittance", tau, 1, TD), F("max_episode_steps", max_episode_steps, 1, TI),
    F("sparse_reward", sparse_reward, 1, TI), F("num_cargoes_per_target", num_cargoes_per_target, 1, TI),
    F("shuffle_entities", shuffle_entities, 1, TI), F("targets_start_with_cargoes", targets_start_with_cargoes, 1, TI),
    F("high_capacity_target_split", high_capacity_target_split, 1, TD),
    F("target_step_size", target_step_size, 1, TD), F("target_sight_range", target_sight_range, 1, TD),
    F("freight_scale", freight_scale, 1, TD), F("bounty_scale", bounty_scale, 1, TD),
    F("reward_scale", reward_scale, 1, TD), F("max_target_team_episode_reward", max_team_reward, 1, TD),
    F("cam_range", cam_range, MO_MAXC * 4, TD), F("tgt_range", tgt_range, MO_MAXT * 4, TD),
    F("obs_range", obs_range, MO_MAXO * 4, TD), F("obs_radius_range", obs_radius_range, 2, TD),
    F("cfg_cam_radius", cfg_cam_radius, 1, TD), F("cfg_cam_min_viewing_angle", cfg_cam_theta_min, 1, TD),
    F("cfg_cam_max_sight_range", cfg_cam_rmax, 1, TD), F("cfg_cam_rotation_step", cfg_cam_rot, 1, TD),
    F("cfg_cam_zooming_step", cfg_cam_zoom, 1, TD),
    F("cam_x", cam_x, MO_MAXC, TD), F("cam_y", cam_y, MO_MAXC, TD), F("cam_radius", cam_r, MO_MAXC, TD),
    F("cam_min_viewing_angle", cam_theta_min, MO_MAXC, TD), F("cam_max_sight_range", cam_rmax, MO_MAXC, TD),
    F("cam_rotation_step", cam_rot, MO_MAXC, TD), F("cam_zooming_step", cam_zoom, MO_MAXC, TD),
    F("obs_x", obs_x, MO_MAXO, TD), F("obs_y", obs_y, MO_MAXO, TD), F("obs_radius", obs_r, MO_MAXO, TD),
    F("tgt_capacity", tgt_capacity, MO_MAXT, TI), F("tgt_step_size", tgt_step, MO_MAXT, TD),
    F("tgt_sight_range", tgt_sight, MO_MAXT, TD),
    F("camera_obstacle_view_mask", cam_obs_mask, MO_MAXC * MO_MAXO, TB),
    F("cam_phi", cam_phi, MO_MAXC, TD), F("cam_theta", cam_theta, MO_MAXC, TD), F("cam_sight", cam_sight, MO_MAXC, TD),
    F("tgt_x", tgt_x, MO_MAXT, TD), F("tgt_y", tgt_y, MO_MAXT, TD), F("tgt_colliding", tgt_colliding, MO_MAXT, TB),
    F("tgt_empty_bits", empty_bits, MO_MAXT * MO_NW, TB), F("tgt_goal_bits", goal_bits, MO_MAXT * MO_NW, TI),
    F("tgt_goals", goals, MO_MAXT, TI), F("freights", freights, MO_MAXT, TI), F("bounties", bounties, MO_MAXT, TI),
    F("target_steps", target_steps, MO_MAXT, TI), F("tracked_steps", tracked_steps, MO_MAXT, TI),
    F("remaining_cargoes", remaining, MO_NW * MO_NW, TI), F("awaiting_cargo_counts", awaiting, MO_NW, TI),
    F("num_delivered_cargoes", num_delivered, 1, TI), F("episode_reward", episode_reward, 1, TD),
    F("delayed_episode_reward", delayed_episode_reward, 1, TD), F("episode_step", episode_step, 1, TI),
    F("camera_target_view_mask", m_ct, MO_MAXC * MO_MAXT, TB), F("target_camera_view_mask", m_tc, MO_MAXT * MO_MAXC, TB),
    F("target_obstacle_view_mask", m_to, MO_MAXT * MO_MAXO, TB), F("target_target_view_mask", m_tt, MO_MAXT * MO_MAXT, TB),
    F("camera_camera_view_mask", m_cc, MO_MAXC * MO_MAXC, TB), F("tracked_bits", tracked, MO_MAXT, TB),
    F("target_dones", target_dones, MO_MAXT, TB), F("target_warehouse_distances", tw_dist, MO_MAXT * MO_NW, TD),
    F("coverage_rate", coverage_rate, 1, TD), F("real_coverage_rate", real_coverage_rate, 1, TD),
    F("mean_transport_rate", mean_transport_rate, 1, TD), F("reward_cam", reward_cam, 1, TD),
    F("reward_tgt", reward_tgt, 1, TD), F("reward_dense", reward_dense, 1, TD), F("reward_delayed", reward_delayed, 1, TD),
    F("normalized_reward_tgt", normalized_reward_tgt, 1, TD), F("done", done, 1, TI),
    F("seed", seed, 1, TU64), F("env_index", env_index, 1, TU32), F("tick", tick, 1, TU32), F("episode", episode, 1, TU32),
};
#define NFIELDS ((int)(sizeof(FIELDS) / sizeof(FIELDS[0])))

static const field_t *find_field(const char *name) {
    for (int i = 0; i < NFIELDS; ++i)
        if (strcmp(FIELDS[i].name, name) == 0) return &FIELDS[i];
    return NULL;
}

int mo_set(mo_env *e, const char *field, const double *data, int n) {
    const field_t *f = find_field(field);
    if (!f || n > f->count || n < 0) return -1;
    char *base = (char *)e + f->off;
    for (int i = 0; i < n; ++i) {
        switch (f->type) {
            case TD: ((double *)base)[i] = data[i]; break;
            case TI: ((int *)base)[i] = (int)data[i]; break;
            case TB: ((unsigned char *)base)[i] = data[i] != 0.0; break;
            case TU32: ((uint32_t *)base)[i] = (uint32_t)data[i]; break;
            case TU64: ((uint64_t *)base)[i] = (uint64_t)data[i]; break;
        }
    }
    return n;
}

int mo_get(const mo_env *e, const char *field, double *data, int n) {
    const field_t *f = find_field(field);
    if (!f || n > f->count || n < 0) return -1;
    const char *base = (const char *)e + f->off;
    for (int i = 0; i < n; ++i) {
        switch (f->type) {
            case TD: data[i] = ((const double *)base)[i]; break;
            case TI: data[i] = (double)((const int *)base)[i]; break;
            case TB: data[i] = (double)((const unsigned char *)base)[i]; break;
            case TU32: data[i] = (double)((const uint32_t *)base)[i]; break;
            case TU64: data[i] = (double)((const uint64_t *)base)[i]; break;
        }
    }
    return n;
}

int mo_get_lut(const mo_env *e, int c, int outer, double *phis, double *rhos, int cap) {
    int n = e->lut_n[outer][c];
    if (n > cap) return -n;
    memcpy(phis, e->lut_phi[outer][c], sizeof(double) * (size_t)n);
    memcpy(rhos, e->lut_rho[outer][c], sizeof(double) * (size_t)n);
    return n;
}

int mo_set_lut(mo_env *e, int c, int outer, const double *phis, const double *rhos, int n) {
    free(e->lut_phi[outer][c]); free(e->lut_rho[outer][c]);
    e->lut_phi[outer][c] = (double *)malloc(sizeof(double) * (size_t)n);
    e->lut_rho[outer][c] = (double *)malloc(sizeof(double) * (size_t)n);
    memcpy(e->lut_phi[outer][c], phis, sizeof(double) * (size_t)n);
    memcpy(e->lut_rho[outer][c], rhos, sizeof(double) * (size_t)n);
    e->lut_n[outer][c] = n;
    return n;
}

void mo_build_luts(mo_env *e) { /* environment.py:739-741,750-755 */
    double oxyr[3 * MO_MAXO];
    for (int o = 0; o < e->No; ++o) { oxyr[3 * o] = e->obs_x[o]; oxyr[3 * o + 1] = e->obs_y[o]; oxyr[3 * o + 2] = e->obs_r[o]; }
    int cap = 360 + e->No * 240 + 16;
    double *phis = (double *)malloc(sizeof(double) * (size_t)cap), *rhos = (double *)malloc(sizeof(double) * (size_t)cap);
    for (int c = 0; c < e->Nc; ++c) {
        for (int outer = 0; outer < 2; ++outer) {
            int n = mo_build_lut_raw(e->cam_x[c], e->cam_y[c], e->cam_rmax[c], oxyr, e->No, e->tau, outer, phis, rhos, cap);
            mo_set_lut(e, c, outer, phis, rhos, n);
        }
        for (int o = 0; o < e->No; ++o) { /* entities.py:365: strict < */
            double d = norm2(e->cam_x[c] - e->obs_x[o], e->cam_y[c] - e->obs_y[o]);
            e->cam_obs_mask[c][o] = d < e->cam_rmax[c] + e->obs_r[o];
        }
    }
    free(phis); free(rhos);
}

/* _update_view (environment.py:1356-1388) */
static void update_view(mo_env *e, const double *tape_ct, uint32_t stream, uint32_t tick) {
    int Nc = e->Nc, Nt = e->Nt, No = e->No;
    memset(e->m_ct, 0, sizeof(e->m_ct)); memset(e->m_tc, 0, sizeof(e->m_tc)); memset(e->m_to, 0, sizeof(e->m_to));
    memset(e->m_tt, 0, sizeof(e->m_tt)); memset(e->m_cc, 0, sizeof(e->m_cc));
    for (int t = 0; t < Nt; ++t) {
        for (int c = 0; c < Nc; ++c) {
            double u = tape_ct ? tape_ct[c * Nt + t] : draw_stream(e, tick, stream, (uint32_t)(c * Nt + t));
            e->m_ct[c][t] = (unsigned char)mo_camera_perceive(e->cam_x[c], e->cam_y[c], e->cam_phi[c], e->cam_theta[c],
                                                               e->cam_sight[c], e->tgt_x[t], e->tgt_y[t], u, e->tau,
                                                               e->lut_phi[0][c], e->lut_rho[0][c], e->lut_n[0][c]);
            /* Sensor.perceive (entities.py:229-232) */
            e->m_tc[t][c] = norm2(e->tgt_x[t] - e->cam_x[c], e->tgt_y[t] - e->cam_y[c]) <= e->tgt_sight[t] + e->cam_r[c];
        }
        for (int o = 0; o < No; ++o)
            e->m_to[t][o] = norm2(e->tgt_x[t] - e->obs_x[o], e->tgt_y[t] - e->obs_y[o]) <= e->tgt_sight[t] + e->obs_r[o];
        for (int t2 = 0; t2 < Nt; ++t2)
            e->m_tt[t][t2] = (t == t2) || norm2(e->tgt_x[t] - e->tgt_x[t2], e->tgt_y[t] - e->tgt_y[t2]) <= e->tgt_sight[t] + 0.0;
    }
    for (int c = 0; c < Nc; ++c)
        for (int c2 = 0; c2 < Nc; ++c2)
            e->m_cc[c][c2] = (c == c2) || mo_camera_perceive(e->cam_x[c], e->cam_y[c], e->cam_phi[c], e->cam_theta[c],
                                                               e->cam_sight[c], e->cam_x[c2], e->cam_y[c2], 0.0, 0.0,
                                                               e->lut_phi[0][c], e->lut_rho[0][c], e->lut_n[0][c]);
    for (int t = 0; t < Nt; ++t) {
        unsigned char any = 0;
        for (int c = 0; c < Nc; ++c) any |= e->m_ct[c][t];
        e->tracked[t] = any;
    }
}

void mo_update_view(mo_env *e, const double *tape_ct) { update_view(e, tape_ct, S_TRANSMIT, e->tick); }

/* Target.simulate (entities.py:645-668).  Circles = obstacles then cameras in
 * index order (the reference walks a Python set of spatial-hash candidates; the
 * hash only prunes circles that cannot touch the step). */
static void target_simulate(mo_env *e, int t, double ax, double ay) {
    ray_t step;
    ray_from_vector(&step, ax, ay, e->tgt_x[t], e->tgt_y[t]);
    if (ray_norm(&step) > e->tgt_step[t]) ray_set_norm(&step, e->tgt_step[t]);
    ray_materialize(&step);
    double desx = step.ox + step.vx, desy = step.oy + step.vy;
    for (int o = 0; o < e->No; ++o) obstruct(&step, e->obs_x[o], e->obs_y[o], e->obs_r[o], 1, 0);
    for (int c = 0; c < e->Nc; ++c) obstruct(&step, e->cam_x[c], e->cam_y[c], e->cam_r[c], 1, 0);
    ray_materialize(&step);
    double nx = clipd(step.ox + step.vx, -TERRAIN_SIZE, TERRAIN_SIZE);
    double ny = clipd(step.oy + step.vy, -TERRAIN_SIZE, TERRAIN_SIZE);
    e->tgt_x[t] = nx; e->tgt_y[t] = ny;
    e->tgt_colliding[t] = (fabs(nx - desx) > 1e-6) || (fabs(ny - desy) > 1e-6);
}

typedef double (*uniform_fn)(mo_env *, int t, const double *goal_u);
static double goal_uniform_step(mo_env *e, int t, const double *goal_u) {
    return goal_u ? goal_u[t] : draw_stream(e, e->tick, S_GOAL, (uint32_t)t);
}
static double goal_uniform_reset(mo_env *e, int t, const double *goal_u) { (void)t; (void)goal_u; return draw_reset(e); }

static int row_any(const int row[MO_NW]) { return row[0] || row[1] || row[2] || row[3]; }
static int pick_goal(const int row[MO_NW], double u) { /* np_random.choice(flatnonzero(row > 0)) */
    int cand[MO_NW], k = 0;
    for (int g = 0; g < MO_NW; ++g) if (row[g] > 0) cand[k++] = g;
    int j = (int)(u * (double)k);
    if (j >= k) j = k - 1;
    return cand[j];
}

/* _assign_goals (environment.py:1271-1324) */
static void assign_goals(mo_env *e, const double *goal_u, uniform_fn uf, double *reward, double *delayed) {
    int Nt = e->Nt;
    int old_goals[MO_MAXT];
    memcpy(old_goals, e->goals, sizeof(old_goals));
    double dl = 0.0;
    int cnt = 0;
    for (int t = 0; t < Nt; ++t) cnt += (e->tracked[t] && e->bounties[t] > 0);
    double rw = -(double)cnt;
    for (int t = 0; t < Nt; ++t) { int b = e->bounties[t] - (int)e->tracked[t]; e->bounties[t] = b > 0 ? b : 0; }
    for (int t = 0; t < Nt; ++t) {
        int goal = e->goals[t];
        int capacity = e->tgt_capacity[t];
        int inside[MO_NW];
        for (int w = 0; w < MO_NW; ++w) {
            double dx = e->tgt_x[t] - WAREHOUSES[w][0], dy = e->tgt_y[t] - WAREHOUSES[w][1];
            e->tw_dist[t][w] = sqrt(dx * dx + dy * dy); /* np.linalg.norm(axis=-1): no fma */
            double sup = fabs(dx) > fabs(dy) ? fabs(dx) : fabs(dy);
            inside[w] = sup <= WAREHOUSE_RADIUS;
        }
        for (int w = 0; w < MO_NW; ++w) {
            if (!inside[w]) continue;
            if (goal >= 0) {
                if (goal == w) {
                    int weight = e->goal_bits[t][goal];
                    double total_bounty = (double)weight * e->bounty_scale;
                    double r = (double)(e->freights[t] + e->bounties[t]);
                    rw += r;
                    dl += r - (total_bounty - (double)e->bounties[t]);
                    e->num_delivered += weight;
                    e->awaiting[goal] -= weight;
                } else {
                    continue;
                }
            }
            e->freights[t] = e->bounties[t] = 0;
            e->tracked_steps[t] = e->target_steps[t] = 0;
            for (int g = 0; g < MO_NW; ++g) e->goal_bits[t][g] = 0;
            e->goals[t] = -1;
            if (row_any(e->remaining[w])) {
                int new_goal = pick_goal(e->remaining[w], uf(e, t, goal_u));
                int rem = e->remaining[w][new_goal];
                int weight = capacity < rem ? capacity : rem;
                e->remaining[w][new_goal] -= weight;
                e->goal_bits[t][new_goal] = weight;
                e->freights[t] = (int)((double)weight * e->freight_scale);
                e->bounties[t] = (int)((double)weight * e->bounty_scale);
                e->goals[t] = new_goal;
                break;
            }
        }
        for (int w = 0; w < MO_NW; ++w)
            if (inside[w]) e->empty_bits[t][w] = !row_any(e->remaining[w]);
    }
    for (int t = 0; t < Nt; ++t) e->target_dones[t] = (e->goals[t] != old_goals[t]) && (old_goals[t] >= 0);
    *reward = rw; *delayed = dl;
}

/* coverage metrics computed inside joint_observation (environment.py:966-979) */
static void update_metrics(mo_env *e) {
    int nb = 0, tb = 0, tr = 0;
    for (int t = 0; t < e->Nt; ++t) {
        int wb = e->bounties[t] > 0;
        nb += wb; tb += (wb && e->tracked[t]); tr += e->tracked[t];
    }
    e->coverage_rate = (double)tr / (double)e->Nt;
    e->real_coverage_rate = nb > 0 ? (double)tb / (double)nb : 0.0;
    e->mean_transport_rate = e->num_delivered > 0 ? e->delayed_episode_reward / (e->reward_scale * (double)e->num_delivered) : 0.0;
}

void mo_step(mo_env *e, const double *cam_act, const double *tgt_act, const double *tape_ct, const double *goal_u) {
    /* _simulate (environment.py:1326-1354) */
    for (int c = 0; c < e->Nc; ++c) {
        double out[3];
        mo_camera_simulate(e->cam_phi[c], e->cam_theta[c], cam_act[2 * c], cam_act[2 * c + 1], e->cam_theta_min[c],
                           e->cam_rmax[c], e->cam_rot[c], e->cam_zoom[c], out);
        e->cam_phi[c] = out[0]; e->cam_theta[c] = out[1]; e->cam_sight[c] = out[2];
    }
    for (int t = 0; t < e->Nt; ++t) target_simulate(e, t, tgt_act[2 * t], tgt_act[2 * t + 1]);
    update_view(e, tape_ct, S_TRANSMIT, e->tick);
    /* step (environment.py:612-632) */
    double rw, dl;
    assign_goals(e, goal_u, goal_uniform_step, &rw, &dl);
    e->episode_reward += rw;
    e->delayed_episode_reward += dl;
    update_metrics(e);
    e->reward_dense = rw; e->reward_delayed = dl;
    double r = e->sparse_reward ? dl : rw;
    e->reward_tgt = r; e->reward_cam = -r;
    e->normalized_reward_tgt = r / e->max_team_reward;
    for (int t = 0; t < e->Nt; ++t) { e->target_steps[t] += 1; e->tracked_steps[t] += e->tracked[t]; }
    e->episode_step += 1;
    int awaiting_any = e->awaiting[0] || e->awaiting[1] || e->awaiting[2] || e->awaiting[3];
    e->done = !(e->episode_step <= e->max_episode_steps && awaiting_any);
    e->tick += 1;
}

/* state vectors (entities.py:313-324, 631-637, 147-148) */
static void camera_state(const mo_env *e, int c, double *out, int priv) {
    double phi_rad = e->cam_phi[c] * DEG2RAD;
    out[0] = e->cam_x[c]; out[1] = e->cam_y[c]; out[2] = e->cam_r[c];
    out[3] = e->cam_sight[c] * cos(phi_rad); out[4] = e->cam_sight[c] * sin(phi_rad);
    out[5] = e->cam_theta[c];
    if (priv) { out[6] = e->cam_rmax[c]; out[7] = e->cam_rot[c]; out[8] = e->cam_zoom[c]; }
}
static void target_state(const mo_env *e, int t, double *out, int priv) {
    int loaded = 0;
    for (int g = 0; g < MO_NW; ++g) loaded |= e->goal_bits[t][g] != 0;
    out[0] = e->tgt_x[t]; out[1] = e->tgt_y[t]; out[2] = e->tgt_sight[t]; out[3] = (double)loaded;
    if (priv) {
        out[4] = e->tgt_step[t]; out[5] = (double)e->tgt_capacity[t];
        for (int g = 0; g < MO_NW; ++g) { out[6 + g] = (double)e->goal_bits[t][g]; out[10 + g] = (double)e->empty_bits[t][g]; }
    }
}
static void preserved(const mo_env *e, double index, double *out) { /* environment.py:499-501 */
    out[0] = e->Nc; out[1] = e->Nt; out[2] = e->No; out[3] = index;
    for (int w = 0; w < MO_NW; ++w) { out[4 + 2 * w] = WAREHOUSES[w][0]; out[5 + 2 * w] = WAREHOUSES[w][1]; }
    out[12] = WAREHOUSE_RADIUS;
}

void mo_observe(const mo_env *e, double *cam_obs, double *tgt_obs) { /* environment.py:908-964 */
    int Nc = e->Nc, Nt = e->Nt, No = e->No;
    int Dc = mo_camera_obs_dim(e), Dt = mo_target_obs_dim(e);
    double cpub[MO_MAXC][6], tpub[MO_MAXT][4];
    for (int c = 0; c < Nc; ++c) camera_state(e, c, cpub[c], 0);
    for (int t = 0; t < Nt; ++t) target_state(e, t, tpub[t], 0);
    for (int c = 0; c < Nc; ++c) {
        double *row = cam_obs + (size_t)c * Dc;
        memset(row, 0, sizeof(double) * (size_t)Dc);
        preserved(e, (double)c, row);
        camera_state(e, c, row + PRESERVED_DIM, 1);
        double *p = row + PRESERVED_DIM + 9;
        for (int t = 0; t < Nt; ++t, p += 5)
            if (e->m_ct[c][t]) { memcpy(p, tpub[t], sizeof(double) * 4); p[4] = 1.0; }
        for (int o = 0; o < No; ++o, p += 4)
            if (e->cam_obs_mask[c][o]) { p[0] = e->obs_x[o]; p[1] = e->obs_y[o]; p[2] = e->obs_r[o]; p[3] = 1.0; }
        for (int c2 = 0; c2 < Nc; ++c2, p += 7)
            if (e->m_cc[c][c2]) { memcpy(p, cpub[c2], sizeof(double) * 6); p[6] = 1.0; }
    }
    for (int t = 0; t < Nt; ++t) {
        double *row = tgt_obs + (size_t)t * Dt;
        memset(row, 0, sizeof(double) * (size_t)Dt);
        preserved(e, (double)t, row);
        target_state(e, t, row + PRESERVED_DIM, 1);
        double *p = row + PRESERVED_DIM + 14;
        for (int c = 0; c < Nc; ++c, p += 7)
            if (e->m_tc[t][c]) { memcpy(p, cpub[c], sizeof(double) * 6); p[6] = 1.0; }
        for (int o = 0; o < No; ++o, p += 4)
            if (e->m_to[t][o]) { p[0] = e->obs_x[o]; p[1] = e->obs_y[o]; p[2] = e->obs_r[o]; p[3] = 1.0; }
        for (int t2 = 0; t2 < Nt; ++t2, p += 5)
            if (e->m_tt[t][t2]) { memcpy(p, tpub[t2], sizeof(double) * 4); p[4] = 1.0; }
    }
}

/* The reference's EnhancedObservation (mode 1, wrappers/enhanced_observation.py:72-126) and SharedFieldOfView
 * (mode 2, wrappers/shared_field_of_view.py:72-148) applied to joint_observation(), per team (0 = plain). */
void mo_observe_mode(const mo_env *e, int cam_mode, int tgt_mode, double *cam_obs, double *tgt_obs) {
    int Nc = e->Nc, Nt = e->Nt, No = e->No;
    int Dc = mo_camera_obs_dim(e), Dt = mo_target_obs_dim(e);
    mo_observe(e, cam_obs, tgt_obs);
    double cpub[MO_MAXC][6], tpub[MO_MAXT][4];
    for (int c = 0; c < Nc; ++c) camera_state(e, c, cpub[c], 0);
    for (int t = 0; t < Nt; ++t) target_state(e, t, tpub[t], 0);
    if (cam_mode) {
        for (int c = 0; c < Nc; ++c) {
            double *p = cam_obs + (size_t)c * Dc + PRESERVED_DIM + 9;
            for (int t = 0; t < Nt; ++t, p += 5) {
                int vis = 1;
                if (cam_mode == 2) { vis = 0; for (int k = 0; k < Nc; ++k) vis |= e->m_ct[k][t]; }       /* target_mask.any(axis=0) */
                for (int i = 0; i < 4; ++i) p[i] = vis ? tpub[t][i] : 0.0;
                p[4] = vis ? 1.0 : 0.0;
            }
            for (int o = 0; o < No; ++o, p += 4) {
                int vis = 1;
                if (cam_mode == 2) { vis = 0; for (int k = 0; k < Nc; ++k) vis |= e->cam_obs_mask[k][o]; }
                p[0] = vis ? e->obs_x[o] : 0.0; p[1] = vis ? e->obs_y[o] : 0.0; p[2] = vis ? e->obs_r[o] : 0.0; p[3] = vis ? 1.0 : 0.0;
            }
            for (int c2 = 0; c2 < Nc; ++c2, p += 7) { memcpy(p, cpub[c2], sizeof(double) * 6); p[6] = 1.0; }   /* teammates: always */
        }
    }
    if (tgt_mode) {
        double empty[MO_NW];
        for (int w = 0; w < MO_NW; ++w) {
            if (tgt_mode == 1) {                       /* np.logical_not(remaining_cargoes).all(axis=-1) */
                int any = 0;
                for (int r = 0; r < MO_NW; ++r) any |= e->remaining[w][r] != 0;
                empty[w] = any ? 0.0 : 1.0;
            } else {                                   /* empty_bits.any(axis=0) */
                int any = 0;
                for (int t = 0; t < Nt; ++t) any |= tgt_obs[(size_t)t * Dt + PRESERVED_DIM + 10 + w] != 0.0;
                empty[w] = any ? 1.0 : 0.0;
            }
        }
        for (int t = 0; t < Nt; ++t) {
            double *row = tgt_obs + (size_t)t * Dt;
            for (int w = 0; w < MO_NW; ++w) row[PRESERVED_DIM + 10 + w] = empty[w];
            double *p = row + PRESERVED_DIM + 14;
            for (int c = 0; c < Nc; ++c, p += 7) {
                int vis = 1;
                if (tgt_mode == 2) { vis = 0; for (int k = 0; k < Nt; ++k) vis |= e->m_tc[k][c]; }
                for (int i = 0; i < 6; ++i) p[i] = vis ? cpub[c][i] : 0.0;
                p[6] = vis ? 1.0 : 0.0;
            }
            for (int o = 0; o < No; ++o, p += 4) {
                int vis = 1;
                if (tgt_mode == 2) { vis = 0; for (int k = 0; k < Nt; ++k) vis |= e->m_to[k][o]; }
                p[0] = vis ? e->obs_x[o] : 0.0; p[1] = vis ? e->obs_y[o] : 0.0; p[2] = vis ? e->obs_r[o] : 0.0; p[3] = vis ? 1.0 : 0.0;
            }
            for (int t2 = 0; t2 < Nt; ++t2, p += 5) { memcpy(p, tpub[t2], sizeof(double) * 4); p[4] = 1.0; }
        }
    }
}

/* DiscreteCamera.action / DiscreteTarget.action (wrappers/discrete_action_spaces.py:59-74, 165-180):
 * continuous = action_high * normalized_grid[index]; the grids are inputs (NumPy builds them). */
void mo_decode_discrete(const mo_env *e, const int *cam_idx, const double *cam_grid, const int *tgt_idx, const double *tgt_grid,
                        double *cam_act, double *tgt_act) {
    if (cam_idx && cam_grid && cam_act)
        for (int c = 0; c < e->Nc; ++c) {
            cam_act[2 * c] = e->cam_rot[c] * cam_grid[2 * cam_idx[c]];
            cam_act[2 * c + 1] = e->cam_zoom[c] * cam_grid[2 * cam_idx[c] + 1];
        }
    if (tgt_idx && tgt_grid && tgt_act)
        for (int t = 0; t < e->Nt; ++t) {
            double high = e->tgt_step[t];                                      /* Target.step_size, entities.py:612-615 */
            tgt_act[2 * t] = high * tgt_grid[2 * tgt_idx[t]];
            tgt_act[2 * t + 1] = high * tgt_grid[2 * tgt_idx[t] + 1];
        }
}

/* AuxiliaryCameraRewards.compute_soft_coverage_score(s) + the per-camera reduction of its step()
 * (wrappers/auxiliary_camera_rewards.py:128-139, 181-239): the distance from each target to the nearest POINT of the
 * camera's sector outline -- 16 points up each flank, the knots of the OUTER occlusion table strictly inside the sector
 * (Camera.boundary_between(outer=True), entities.py:513-543) and the two sector ends interpolated on the INNER table --
 * signed by the view mask and divided by the radius of the sector's inscribed circle.  matrix [Nc][Nt], scores [Nc]. */
void mo_soft_coverage(const mo_env *e, double *matrix, double *scores) {
    for (int c = 0; c < e->Nc; ++c) {
        const double theta = e->cam_theta[c], sight = e->cam_sight[c];
        const double dist_max = theta < 180.0 ? sight / (1.0 + 1.0 / sin(theta / 2.0 * DEG2RAD)) : sight / 2.0;
        const double left = mo_normalize_angle(e->cam_phi[c] - theta / 2.0);
        const double right = left + ((e->cam_phi[c] + theta / 2.0) - (e->cam_phi[c] - theta / 2.0));
        const int n = e->lut_n[1][c];
        const double *phis_all = e->lut_phi[1][c], *rhos_all = e->lut_rho[1][c];
        double *phis = (double *)malloc(sizeof(double) * (size_t)(n + 34)), *rhos = (double *)malloc(sizeof(double) * (size_t)(n + 34));
        int m = 16;
        phis[m] = left; rhos[m] = mo_interp(e->lut_phi[0][c], e->lut_rho[0][c], e->lut_n[0][c], mo_normalize_angle(left)); ++m;
        if (right <= 180.0) {
            for (int i = 0; i < n; ++i) if (left < phis_all[i] && phis_all[i] < right) { phis[m] = phis_all[i]; rhos[m] = rhos_all[i]; ++m; }
        } else {
            for (int i = 0; i < n; ++i) if (left < phis_all[i] && phis_all[i] <= 180.0) { phis[m] = phis_all[i]; rhos[m] = rhos_all[i]; ++m; }
            for (int i = 0; i < n; ++i) if (phis_all[i] > -180.0 && phis_all[i] < right - 360.0) { phis[m] = phis_all[i]; rhos[m] = rhos_all[i]; ++m; }
        }
        phis[m] = right; rhos[m] = mo_interp(e->lut_phi[0][c], e->lut_rho[0][c], e->lut_n[0][c], mo_normalize_angle(right)); ++m;
        for (int k = 0; k < 16; ++k) {                       /* np.linspace(0, rho, 16, endpoint=False) up both flanks */
            phis[k] = phis[16]; rhos[k] = (double)k * (rhos[16] / 16.0);
            phis[m + k] = phis[m - 1]; rhos[m + k] = (double)k * (rhos[m - 1] / 16.0);
        }
        m += 16;
        int any = 0; double sum = 0.0, best = -INFINITY;
        for (int t = 0; t < e->Nt; ++t) {
            const double dx = e->tgt_x[t] - e->cam_x[c], dy = e->tgt_y[t] - e->cam_y[c];
            double dist = INFINITY;
            for (int i = 0; i < m; ++i) {
                const double a = phis[i] * DEG2RAD;
                const double d = hypot(dx - rhos[i] * cos(a), dy - rhos[i] * sin(a));
                if (d < dist) dist = d;
            }
            if (!e->m_ct[c][t]) dist = -dist;
            const double score = dist / dist_max;
            if (matrix) matrix[c * e->Nt + t] = score;
            if (e->m_ct[c][t]) { any = 1; sum += score; }
            if (score > best) best = score;
        }
        if (scores) scores[c] = any ? sum : tanh(best);
        free(phis); free(rhos);
    }
}

void mo_state(const mo_env *e, double *out) { /* environment.py:894-906 */
    double *p = out;
    preserved(e, 0.0, p); p += PRESERVED_DIM;
    for (int c = 0; c < e->Nc; ++c, p += 9) camera_state(e, c, p, 1);
    for (int t = 0; t < e->Nt; ++t, p += 14) target_state(e, t, p, 1);
    for (int o = 0; o < e->No; ++o, p += 3) { p[0] = e->obs_x[o]; p[1] = e->obs_y[o]; p[2] = e->obs_r[o]; }
    for (int t = 0; t < e->Nt; ++t) *p++ = (double)e->freights[t];
    for (int t = 0; t < e->Nt; ++t) *p++ = (double)e->bounties[t];
    for (int s = 0; s < MO_NW; ++s) for (int r = 0; r < MO_NW; ++r) *p++ = (double)e->remaining[s][r];
}

/* ------------------------------------------------------------ rule-based agents (row f1)
 * GreedyCameraAgent / GreedyTargetAgent (mate/agents/greedy.py:13-227, 229-365) for the agents of ONE
 * environment, called in the order of mate.group_step (wrappers/single_team.py:79-92): observe -> send ->
 * receive -> act.  An agent knows its own private state, the opponents its observation row shows (the
 * view masks of the previous step) and its teammates' messages; every random draw comes from the tape. */
struct mo_policy {
    int episode;                                   /* episode the memories belong to (-1: none) */
    int memory_period;                             /* greedy.py:21 */
    double noise_scale;                            /* greedy.py:236 */
    double mem[MO_MAXC][MO_MAXT][2];
    int t2f[MO_MAXC][MO_MAXT];
    double prev_action[MO_MAXC][2];
    int delay[MO_MAXC][MO_MAXC];                   /* [sender][recipient] */
    int neighbor[MO_MAXC][MO_MAXC];                /* [camera][teammate] */
    int has_state[MO_MAXC];                        /* message2send still holds 'state' */
    double tgt_prev[MO_MAXT][2], tgt_noise[MO_MAXT][2];
    int tgt_goal[MO_MAXT], tgt_nonempty[MO_MAXT], tgt_need[MO_MAXT];
};

mo_policy *mo_policy_create(void) {
    mo_policy *p = (mo_policy *)calloc(1, sizeof(mo_policy));
    if (p) { p->episode = -1; p->memory_period = 25; p->noise_scale = 0.5; }
    return p;
}
void mo_policy_destroy(mo_policy *p) { free(p); }

static inline double sin_deg(double x) { return sin(x * DEG2RAD); }   /* utils.py:134-136 */

void mo_policy_act(mo_policy *a, const mo_env *e, const mo_policy_tape *tape, double *cam_act, double *tgt_act) {
    const int Nc = e->Nc, Nt = e->Nt;
    const int fresh = a->episode != (int)e->episode;
    /* ---- reset (greedy.py:43-61, 262-283) and observe / process_messages (:100-113, :326-332) */
    for (int c = 0; c < Nc; ++c) {
        if (fresh) {
            for (int t = 0; t < Nt; ++t) {
                int s = e->m_ct[c][t];
                a->mem[c][t][0] = s ? e->tgt_x[t] : 0.0; a->mem[c][t][1] = s ? e->tgt_y[t] : 0.0;
                a->t2f[c][t] = s ? a->memory_period : 0;
            }
            a->prev_action[c][0] = a->prev_action[c][1] = 0.0;
            for (int s = 0; s < Nc; ++s) { a->delay[c][s] = 0; a->neighbor[c][s] = 0; }
            a->has_state[c] = 1;
        }
        for (int t = 0; t < Nt; ++t) {
            int left = a->t2f[c][t] - 1;
            if (left < 0) left = 0;
            if (e->m_ct[c][t]) { left = a->memory_period; a->mem[c][t][0] = e->tgt_x[t]; a->mem[c][t][1] = e->tgt_y[t]; }
            a->t2f[c][t] = left;
        }
    }
    for (int t = 0; t < Nt; ++t) {
        if (fresh) {
            double step = e->tgt_step[t];
            a->tgt_prev[t][0] = e->tgt_x[t]; a->tgt_prev[t][1] = e->tgt_y[t];
            a->tgt_noise[t][0] = 0.5 * (-step + (2.0 * step) * tape->tgt_reset_sample_u[2 * t]);       /* 0.5 * action_space.sample() */
            a->tgt_noise[t][1] = 0.5 * (-step + (2.0 * step) * tape->tgt_reset_sample_u[2 * t + 1]);
            a->tgt_goal[t] = e->goals[t];
            a->tgt_nonempty[t] = 0xf;
            a->tgt_need[t] = 0;
        }
        int seen_empty = 0;
        for (int w = 0; w < MO_NW; ++w) seen_empty |= (e->empty_bits[t][w] != 0) << w;
        if (seen_empty & a->tgt_nonempty[t]) { a->tgt_nonempty[t] &= ~seen_empty; a->tgt_need[t] = 1; }
    }
    a->episode = (int)e->episode;
    /* ---- cameras: send_responses (:158-190) */
    int send[MO_MAXC][MO_MAXC];
    for (int s = 0; s < Nc; ++s) {
        unsigned seen_now = 0;
        for (int t = 0; t < Nt; ++t) seen_now |= (unsigned)(e->m_ct[s][t] != 0) << t;
        const int has_content = a->has_state[s] || seen_now;
        for (int c = 0; c < Nc; ++c) {
            int d = a->delay[s][c] - 1;
            if (d < 0) d = 0;
            int bits = 0;
            if (has_content && s != c && d == 0) {
                unsigned list = 0;
                if (seen_now && a->neighbor[s][c]) {            /* filterout_beyond_range */
                    double threshold = 1.1 * e->cam_rmax[c];
                    for (int t = 0; t < Nt; ++t)
                        if (((seen_now >> t) & 1u) && norm2(e->tgt_x[t] - e->cam_x[c], e->tgt_y[t] - e->cam_y[c]) < threshold) list |= 1u << t;
                }
                bits = (int)list | (a->has_state[s] ? (int)0x80000000u : 0);
                if (bits) d = tape->cam_delay[s * Nc + c];       /* np_random.randint(memory_period // 4, 2 * memory_period) */
            }
            a->delay[s][c] = d;
            send[s][c] = bits;
        }
    }
    /* ---- cameras: receive_responses (:192-226), then message2send.clear() */
    for (int c = 0; c < Nc; ++c)
        for (int s = 0; s < Nc; ++s) {
            int bits = send[s][c];
            if (bits & (int)0x80000000u) a->neighbor[c][s] = 1;
            for (int t = 0; t < Nt; ++t)
                if ((bits >> t) & 1) { a->mem[c][t][0] = e->tgt_x[t]; a->mem[c][t][1] = e->tgt_y[t]; a->t2f[c][t] = a->memory_period; }
        }
    for (int c = 0; c < Nc; ++c) {
        int seen_now = 0;
        for (int t = 0; t < Nt; ++t) seen_now |= e->m_ct[c][t] != 0;
        if (a->has_state[c] || seen_now) a->has_state[c] = 0;
    }
    /* ---- targets: broadcast the non-empty sets (:334-358) */
    {
        int set[MO_MAXT];
        for (int t = 0; t < Nt; ++t) {
            set[t] = a->tgt_nonempty[t];
            for (int s = 0; s < Nt; ++s) if (a->tgt_need[s]) set[t] &= a->tgt_nonempty[s];
        }
        for (int t = 0; t < Nt; ++t) { a->tgt_nonempty[t] = set[t]; a->tgt_need[t] = 0; }
    }
    /* ---- cameras: act (:69-156) */
    for (int c = 0; c < Nc; ++c) {
        double obs_state[9];
        camera_state(e, c, obs_state, 1);
        /* what the agent derives from its observation row (agents/utils.py:206-255) */
        const double sight = norm2(obs_state[3], obs_state[4]);
        const double orientation = atan2_deg(obs_state[4], obs_state[3]);
        const double theta = obs_state[5], rmax = obs_state[6];
        const double q2 = sight / rmax;
        const double min_va = theta * (q2 * q2);
        const double threshold = 1.1 * rmax;
        int best = -1; double best_d = 0.0;
        for (int t = 0; t < Nt; ++t) {
            if (a->t2f[c][t] <= 0) continue;
            double dnorm = norm2(a->mem[c][t][0] - e->cam_x[c], a->mem[c][t][1] - e->cam_y[c]);
            if (!(dnorm < threshold)) continue;
            if (best < 0 || dnorm < best_d) { best = t; best_d = dnorm; }
        }
        double a0, a1;
        if (best >= 0) {
            double rx = a->mem[c][best][0] - e->cam_x[c], ry = a->mem[c][best][1] - e->cam_y[c];
            double best_orientation = atan2_deg(ry, rx);
            double distance = best_d, best_va;
            if (distance * (1.0 + sin_deg(min_va / 2.0)) >= rmax) best_va = min_va;
            else {
                double area_product = theta * (sight * sight);
                if (distance <= sqrt(area_product / 180.0) / 2.0) best_va = 180.0;
                else {
                    double b = 180.0;
                    for (int it = 0; it < 20; ++it) {
                        double half = b / 2.0;
                        double sr = distance * (1.0 + sin_deg(half < 90.0 ? half : 90.0));
                        b = area_product / (sr * sr);
                    }
                    best_va = clipd(b, min_va, 180.0);
                }
            }
            a0 = clipd(mo_normalize_angle(best_orientation - orientation), -e->cam_rot[c], e->cam_rot[c]);
            a1 = clipd(best_va - theta, -e->cam_zoom[c], e->cam_zoom[c]);
        } else if (tape->cam_binom_u[c] > 1.0 - 0.1) {          /* np_random.binomial(1, 0.1) */
            a0 = -e->cam_rot[c] + (2.0 * e->cam_rot[c]) * tape->cam_sample_u[2 * c];
            a1 = -e->cam_zoom[c] + (2.0 * e->cam_zoom[c]) * tape->cam_sample_u[2 * c + 1];
        } else { a0 = a->prev_action[c][0]; a1 = a->prev_action[c][1]; }
        a->prev_action[c][0] = a0; a->prev_action[c][1] = a1;
        cam_act[2 * c] = a0; cam_act[2 * c + 1] = a1;
    }
    /* ---- targets: act (:285-324) */
    for (int t = 0; t < Nt; ++t) {
        const int state_goal = e->goals[t];
        const double step_size = e->tgt_step[t];
        int goal = a->tgt_goal[t];
        if (state_goal >= 0) goal = state_goal;
        const int nonempty = a->tgt_nonempty[t];
        if (goal < 0 || (state_goal < 0 && !((nonempty >> goal) & 1))) {
            goal = -1;
            int k = 0;
            for (int w = 0; w < MO_NW; ++w) k += (nonempty >> w) & 1;
            if (k > 0) {                                          /* np_random.choice(list(non_empty_warehouses)) */
                int j = (int)(tape->tgt_choice_u[t] * (double)k);
                if (j >= k) j = k - 1;
                for (int w = 0, seen = 0; w < MO_NW; ++w) if ((nonempty >> w) & 1) { if (seen == j) goal = w; ++seen; }
            }
        }
        a->tgt_goal[t] = goal;
        const double x = e->tgt_x[t], y = e->tgt_y[t];
        const double pax = x - a->tgt_prev[t][0], pay = y - a->tgt_prev[t][1];
        double ax = 0.0, ay = 0.0;
        if (goal >= 0) { ax = WAREHOUSES[goal][0] - x; ay = WAREHOUSES[goal][1] - y; }
        const double len = norm2(ax, ay);
        if (len > step_size) { double k2 = step_size / len; ax *= k2; ay *= k2; }
        const double prob = norm2(pax, pay) > 0.2 * step_size ? 0.05 : 0.75;
        const double u = tape->tgt_binom_u[t];
        double nx = a->tgt_noise[t][0], ny = a->tgt_noise[t][1];
        if ((prob <= 0.5) ? (u > 1.0 - prob) : (u <= prob)) {
            nx = a->noise_scale * (-step_size + (2.0 * step_size) * tape->tgt_sample_u[2 * t]);
            ny = a->noise_scale * (-step_size + (2.0 * step_size) * tape->tgt_sample_u[2 * t + 1]);
        }
        tgt_act[2 * t] = clipd(ax + nx, -step_size, step_size);
        tgt_act[2 * t + 1] = clipd(ay + ny, -step_size, step_size);
        a->tgt_prev[t][0] = x; a->tgt_prev[t][1] = y;
        a->tgt_noise[t][0] = nx; a->tgt_noise[t][1] = ny;
    }
}

/* ---------------------------------------------------------------------- reset */
static void shuffle_perm(mo_env *e, int *perm, int n) {
    for (int i = 0; i < n; ++i) perm[i] = i;
    if (!e->shuffle_entities) return;
    for (int i = n - 1; i >= 1; --i) { int j = randint_reset(e, i + 1); int tmp = perm[i]; perm[i] = perm[j]; perm[j] = tmp; }
}

typedef struct { double x, y, r, sight; int is_camera; } placed_t;

/* Entity.overlap / Camera.overlap (entities.py:96-100, 484-489) */
static int overlaps(const placed_t *a, const placed_t *b, double min_distance) {
    double d = norm2(a->x - b->x, a->y - b->y);
    if (d * (1.0 + 1e-6) < a->r + b->r + min_distance) return 1;
    if (a->is_camera && b->is_camera) {
        double m = a->sight < b->sight ? a->sight : b->sight;
        return d < 0.1 * m;
    }
    return 0;
}

/* reset (environment.py:679-834) driven by ONE stream of uniforms: the engine's own Philox reset stream, or -- tape
 * mode -- the uniforms recorded from the reference (tests/golden/make_golden.py `reset_fixture`: every RandomState the
 * reference's reset() touches is replaced by a proxy that draws uniforms, logs them in call order and turns them into
 * shuffles / choices / integers by the rules below; the reference consumes the proxies unchanged).  The rules:
 *   shuffle(x), permutation(n)          Fisher-Yates, i = n-1 .. 1: j = int(u * (i + 1)), swap(x[i], x[j])
 *   choice(n, size=k, replace=False)    partial Fisher-Yates on 0..n-1, i = 0 .. k-1: j = i + int(u * (n - i)), swap
 *   choice(array)                       array[int(u * len)]
 *   randint(lo, hi)                     lo + int(u * (hi - lo));   uniform(a, b) and Box.sample: a + (b - a) * u
 * The draw ORDER below is the reference's call order (pinned by tests/golden/reset_*.npz) and part of the engine
 * specification: the HIP reset kernel consumes the same stream in the same order.  The see-through draws of the
 * first _update_view (environment.py:766) come from `tape_ct` ([Nc*Nt], tape mode) or the reset-view Philox stream. */
static void reset_impl(mo_env *e, const double *tape_ct);
void mo_reset(mo_env *e) { e->reset_tape = NULL; reset_impl(e, NULL); }
int mo_reset_tape(mo_env *e, const double *tape, int n, const double *tape_ct) {
    e->reset_tape = tape; e->reset_tape_n = (uint32_t)n; e->reset_tape_overrun = 0;
    reset_impl(e, tape_ct);
    e->reset_tape = NULL;
    return e->reset_tape_overrun ? -1 : (int)e->reset_draws;
}
/* the first view of the episode again (e.g. after the occlusion tables were replaced): reset-view stream */
void mo_update_view_reset(mo_env *e) { update_view(e, NULL, S_RESET_VIEW, e->episode); update_metrics(e); }
static void reset_impl(mo_env *e, const double *tape_ct) {
    int Nc = e->Nc, Nt = e->Nt, No = e->No;
    e->episode += 1;
    e->reset_draws = 0;
    int pc[MO_MAXC], pt[MO_MAXT], po[MO_MAXO];
    shuffle_perm(e, pc, Nc); shuffle_perm(e, pt, Nt); shuffle_perm(e, po, No);   /* :707-710 */
    for (int t = 0; t < Nt; ++t) e->tgt_capacity[t] = 1;
    int nh = (int)((double)Nt * e->high_capacity_target_split);                   /* :1530-1533 */
    if (nh > 0) {
        if (e->shuffle_entities) {                                               /* :715-717 choice without replacement */
            int idx[MO_MAXT];
            for (int i = 0; i < Nt; ++i) idx[i] = i;
            for (int i = 0; i < nh; ++i) { int j = i + randint_reset(e, Nt - i); int tmp = idx[i]; idx[i] = idx[j]; idx[j] = tmp; e->tgt_capacity[idx[i]] = 2; }
        } else {
            for (int i = 0; i < nh; ++i) e->tgt_capacity[i] = 2;
        }
    }
    for (int t = 0; t < Nt; ++t) { e->tgt_step[t] = e->target_step_size / (double)e->tgt_capacity[t]; e->tgt_sight[t] = e->target_sight_range; }

    placed_t placed[MO_NW + MO_MAXC + MO_MAXO + MO_MAXT];
    int np_ = 0;
    for (int w = 0; w < MO_NW; ++w) { placed[np_].x = WAREHOUSES[w][0]; placed[np_].y = WAREHOUSES[w][1]; placed[np_].r = 0.75 * WAREHOUSE_RADIUS; placed[np_].sight = 0; placed[np_].is_camera = 0; ++np_; } /* :724-727 */
    int total = Nc + No + Nt;
    for (int k = 0; k < total; ++k) {                                            /* :728-737 */
        int kind = k < Nc ? 0 : (k < Nc + No ? 1 : 2);
        int i = kind == 0 ? k : (kind == 1 ? k - Nc : k - Nc - No);
        const double *range = kind == 0 ? e->cam_range[pc[i]] : (kind == 1 ? e->obs_range[po[i]] : e->tgt_range[pt[i]]);
        double min_distance = kind == 2 ? 0.0 : e->target_step_size;
        placed_t cur; memset(&cur, 0, sizeof(cur));
        double phi = 0.0, theta = 0.0;
        int ok = 0;
        for (int attempt = 0; attempt < NUM_RESET_RETRIES && !ok; ++attempt) {
            double radius = 0.0;
            /* Camera(Sensor, Obstacle) resets through Obstacle.reset as well: it samples its (degenerate) radius box first (entities.py:235, 150-152) */
            if (kind == 0) radius = e->cfg_cam_radius + (e->cfg_cam_radius - e->cfg_cam_radius) * draw_reset(e);
            if (kind == 1) radius = e->obs_radius_range[0] + (e->obs_radius_range[1] - e->obs_radius_range[0]) * draw_reset(e); /* entities.py:151 */
            double x = range[0] + (range[1] - range[0]) * draw_reset(e);       /* entities.py:61 */
            double y = range[2] + (range[3] - range[2]) * draw_reset(e);
            double lim = TERRAIN_SIZE - 1.2 * radius;                             /* entities.py:62-65 */
            cur.x = clipd(x, -lim, lim); cur.y = clipd(y, -lim, lim); cur.r = radius; cur.is_camera = kind == 0; cur.sight = 0.0;
            if (kind == 0) {                                                      /* entities.py:326-334 */
                int nsteps = (int)(360.0 / e->cfg_cam_rot);
                phi = mo_normalize_angle(e->cfg_cam_rot * (double)randint_reset(e, nsteps));
                theta = e->cfg_cam_theta_min + (MAX_VIEWING_ANGLE - e->cfg_cam_theta_min) * draw_reset(e);
                cur.sight = sqrt(e->cfg_cam_theta_min * (e->cfg_cam_rmax * e->cfg_cam_rmax) / theta);
            }
            ok = 1;
            for (int q = 0; q < np_ && ok; ++q) if (overlaps(&cur, &placed[q], min_distance)) ok = 0;
        }
        if (!ok && kind == 1) cur.r = 0.0;                                       /* :735-736 */
        placed[np_++] = cur;
        if (kind == 0) {
            e->cam_x[i] = cur.x; e->cam_y[i] = cur.y; e->cam_r[i] = cur.r;
            e->cam_theta_min[i] = e->cfg_cam_theta_min; e->cam_rmax[i] = e->cfg_cam_rmax;
            e->cam_rot[i] = e->cfg_cam_rot; e->cam_zoom[i] = e->cfg_cam_zoom;
            e->cam_phi[i] = phi; e->cam_theta[i] = theta; e->cam_sight[i] = cur.sight;
        } else if (kind == 1) {
            e->obs_x[i] = cur.x; e->obs_y[i] = cur.y; e->obs_r[i] = cur.r;
        } else {
            e->tgt_x[i] = cur.x; e->tgt_y[i] = cur.y; e->tgt_colliding[i] = 0;
            for (int g = 0; g < MO_NW; ++g) { e->goal_bits[i][g] = 0; e->empty_bits[i][g] = 0; }
        }
    }
    mo_build_luts(e);                                                            /* :739-764 */
    update_view(e, tape_ct, S_RESET_VIEW, e->episode);                           /* :766 */

    memset(e->remaining, 0, sizeof(e->remaining));                               /* :768-775 */
    for (;;) {
        for (int k = 0; k < e->num_cargoes_per_target * Nt; ++k) {
            /* choice(4, size=2, replace=False): partial Fisher-Yates on (0, 1, 2, 3) */
            int s = randint_reset(e, MO_NW);
            int r = 1 + randint_reset(e, MO_NW - 1);
            if (r == s) r = 0;
            e->remaining[s][r] += 1;
        }
        for (int g = 0; g < MO_NW; ++g) { e->awaiting[g] = 0; for (int s = 0; s < MO_NW; ++s) e->awaiting[g] += e->remaining[s][g]; }
        int all = 1;
        for (int s = 0; s < MO_NW; ++s) all &= row_any(e->remaining[s]);
        if (all) break;
        if (e->reset_tape_overrun) break;            /* an exhausted tape yields zeros for ever: reported by mo_reset_tape */
    }
    for (int t = 0; t < Nt; ++t) {                                               /* :777-783 */
        e->goals[t] = -1; e->target_steps[t] = e->tracked_steps[t] = 0; e->freights[t] = e->bounties[t] = 0;
        for (int g = 0; g < MO_NW; ++g) { e->goal_bits[t][g] = 0; e->tw_dist[t][g] = 0.0; }
    }
    double rw, dl;
    assign_goals(e, NULL, goal_uniform_reset, &rw, &dl);                         /* :784 */
    for (int t = 0; t < Nt; ++t) e->target_dones[t] = 0;
    e->num_delivered = 0; e->episode_reward = 0.0; e->delayed_episode_reward = 0.0;
    if (e->targets_start_with_cargoes) {                                         /* :789-807 */
        for (int t = 0; t < Nt; ++t) {
            if (e->goals[t] >= 0) continue;
            int perm[MO_NW] = {0, 1, 2, 3};
            for (int i = MO_NW - 1; i >= 1; --i) { int j = randint_reset(e, i + 1); int tmp = perm[i]; perm[i] = perm[j]; perm[j] = tmp; }
            for (int q = 0; q < MO_NW; ++q) {
                int w = perm[q];
                if (!row_any(e->remaining[w])) continue;
                int goal = pick_goal(e->remaining[w], draw_reset(e));
                int rem = e->remaining[w][goal];
                int weight = e->tgt_capacity[t] < rem ? e->tgt_capacity[t] : rem;
                e->remaining[w][goal] -= weight;
                e->goal_bits[t][goal] = weight;
                e->freights[t] = (int)((double)weight * e->freight_scale);
                e->bounties[t] = (int)((double)weight * e->bounty_scale);
                e->goals[t] = goal;
                break;
            }
        }
    }
    e->episode_step = 0;
    e->done = 0;
    e->reward_cam = e->reward_tgt = e->reward_dense = e->reward_delayed = e->normalized_reward_tgt = 0.0;
    update_metrics(e);
}

/* ---------------------------------------------------------------------- batch */
struct mo_batch { int n; mo_env **envs; };

mo_batch *mo_batch_create(const mo_env *proto, int n, uint64_t seed, uint64_t first_env_index) {
    mo_batch *b = (mo_batch *)calloc(1, sizeof(mo_batch));
    b->n = n;
    b->envs = (mo_env **)calloc((size_t)n, sizeof(mo_env *));
    for (int i = 0; i < n; ++i) {
        mo_env *e = (mo_env *)malloc(sizeof(mo_env));
        memcpy(e, proto, sizeof(mo_env));
        memset(e->lut_phi, 0, sizeof(e->lut_phi)); memset(e->lut_rho, 0, sizeof(e->lut_rho)); memset(e->lut_n, 0, sizeof(e->lut_n));
        e->seed = seed; e->env_index = (uint32_t)(first_env_index + (uint64_t)i); e->tick = 0; e->episode = 0;
        b->envs[i] = e;
    }
    return b;
}
void mo_batch_destroy(mo_batch *b) {
    if (!b) return;
    for (int i = 0; i < b->n; ++i) mo_destroy(b->envs[i]);
    free(b->envs); free(b);
}
mo_env *mo_batch_env(mo_batch *b, int i) { return b->envs[i]; }

/* one field of every environment: out[i * n .. i * n + n) = mo_get(env i, field, n) */
int mo_batch_get(const mo_batch *b, const char *field, double *out, int n) {
    for (int i = 0; i < b->n; ++i)
        if (mo_get(b->envs[i], field, out + (size_t)i * (size_t)n, n) != n) return -1;
    return b->n;
}

void mo_batch_reset(mo_batch *b, int threads) {
    (void)threads;
#pragma omp parallel for num_threads(threads) schedule(dynamic, 8)
    for (int i = 0; i < b->n; ++i) mo_reset(b->envs[i]);
}

void mo_batch_step(mo_batch *b, const float *cam_act, const float *tgt_act, int auto_reset, int threads) {
    (void)threads;
#pragma omp parallel for num_threads(threads) schedule(dynamic, 8)
    for (int i = 0; i < b->n; ++i) {
        mo_env *e = b->envs[i];
        float fa[2 * MO_MAXC], ft[2 * MO_MAXT];
        double ca[2 * MO_MAXC], ta[2 * MO_MAXT];
        const float *pc = cam_act ? cam_act + (size_t)i * 2 * e->Nc : fa;
        const float *pt = tgt_act ? tgt_act + (size_t)i * 2 * e->Nt : ft;
        if (!cam_act || !tgt_act) {
            float *qc = cam_act ? NULL : fa, *qt = tgt_act ? NULL : ft;
            float dummy_c[2 * MO_MAXC], dummy_t[2 * MO_MAXT];
            double rot = e->Nc > 0 ? e->cam_rot[0] : 0.0, zoom = e->Nc > 0 ? e->cam_zoom[0] : 0.0;
            mo_random_actions(e->seed, e->env_index, e->tick, e->Nc, e->Nt, rot, zoom, e->target_step_size,
                              qc ? qc : dummy_c, qt ? qt : dummy_t);
        }
        for (int k = 0; k < 2 * e->Nc; ++k) ca[k] = (double)pc[k];
        for (int k = 0; k < 2 * e->Nt; ++k) ta[k] = (double)pt[k];
        mo_step(e, ca, ta, NULL, NULL);
        if (auto_reset && e->done) { int d = e->done; mo_reset(e); e->done = d; }
    }
}

void mo_batch_observe(mo_batch *b, float *cam_obs, float *tgt_obs, int threads) {
    (void)threads;
#pragma omp parallel for num_threads(threads) schedule(static)
    for (int i = 0; i < b->n; ++i) {
        mo_env *e = b->envs[i];
        int Dc = mo_camera_obs_dim(e), Dt = mo_target_obs_dim(e);
        double co[MO_MAXC * 512], to[MO_MAXT * 512];
        mo_observe(e, co, to);
        for (int k = 0; k < e->Nc * Dc; ++k) cam_obs[(size_t)i * e->Nc * Dc + k] = (float)co[k];
        for (int k = 0; k < e->Nt * Dt; ++k) tgt_obs[(size_t)i * e->Nt * Dt + k] = (float)to[k];
    }
}
