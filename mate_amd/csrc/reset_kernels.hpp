// reset_kernels.hpp -- episode boundary on the GPU: MultiAgentTracking.reset (environment.py:679-834).
//
// One 256-thread workgroup per environment that needs a reset:
//   R1 (thread 0)      placement by rejection sampling + cargo matrix + initial goals; consumes the
//                      Philox reset stream in exactly the order of oracle/mate_oracle.c::mo_reset
//   R2 (256 threads)   Camera.add_obstacles (entities.py:362-479): build every camera's occlusion
//                      table -- generate rays, clip them by every obstacle, bitonic-sort by angle in
//                      LDS, dedupe, compact to HBM together with a per-degree bucket index
//   R3 (wave 0)        first _update_view + joint_observation of the new episode
#pragma once
#include "engine_kernels.hpp"

namespace mate {

enum ResetKind : int32_t { RESET_ALL = 0, RESET_MASK = 1, RESET_DONE = 2, RESET_FLAGGED = 3, RESET_LIST = 4,
                           RESET_PAIRS = 5 };   // the (environment, camera) tables a small-LDS table launch deferred (Ptrs::lut_overflow)
enum ResetPhase : int32_t { PH_PLACE = 1, PH_LUT = 2, PH_VIEW = 4, PH_PER_CAMERA = 8, PH_MORE = 16 };

struct ResetLds {   // byte offsets inside the workgroup's dynamic LDS, after the wave-0 context
    int32_t off_keys, off_vals, off_okeys, off_ovals, off_bucket, off_meta, off_scan, sort_cap, total_bytes;
    int32_t off_pre;       // 256 uniforms of the placement's reset stream, drawn by the whole wave before one lane consumes them
    int32_t sort_in_hbm;   // the four sort arrays (4 x sort_cap doubles) do not fit the 160 KiB LDS next to the rest: they live in
                           // Ptrs::sort_scratch, one slice per workgroup of a capped grid (scenarios beyond 20 obstacles per camera table)
};

// The reset stream: uniforms in the reference's call order (oracle/mate_oracle.c above reset_impl states how each
// becomes a shuffle / choice / integer).  Philox keyed by (seed, environment, episode), or -- tape mode,
// mate_engine_reset_tape -- the uniforms recorded from the reference's own reset() (tests/golden/reset_*.npz).
struct ResetRng {
    uint32_t k0, k1, env, episode, n;
    const double *tape;      // this environment's row of the tape, or nullptr
    uint32_t tape_len;
    const double *pre;       // the first `pre_count` draws of the Philox stream, precomputed by all lanes (reset_kernel)
    uint32_t pre_count;
    __device__ double draw() {
        const uint32_t idx = n++;
        if (tape) return idx < tape_len ? tape[idx] : 0.0;      // an overrun shows in the returned draw count
        if (idx < pre_count) return pre[idx];
        const U4 r = philox(k0, k1, env, episode, S_RESET, idx >> 1);
        return (idx & 1u) ? u53(r.z, r.w) : u53(r.x, r.y);
    }
    __device__ int randint(int m) { int j = (int)(draw() * (double)m); return j >= m ? m - 1 : j; }
};

__device__ __forceinline__ bool row_any(const int32_t *row) { return row[0] || row[1] || row[2] || row[3]; }
__device__ __forceinline__ int pick_goal(const int32_t *row, double u) {  // np_random.choice(flatnonzero(row > 0))
    int k = 0;
    for (int g = 0; g < 4; ++g) k += row[g] > 0;
    int j = (int)(u * (double)k);
    if (j >= k) j = k - 1;
    int pick = 0;
    for (int g = 0, seen = 0; g < 4; ++g) if (row[g] > 0) { if (seen == j) pick = g; ++seen; }
    return pick;
}

// R1: the placement is sequential (every draw depends on how many the attempts before it took), so ALL 64 lanes run it
// redundantly -- same draws, same values, same stores to the wave context's LDS records, every branch wave-uniform -- and
// share only the one data-parallel part: the overlap test of a candidate against everything placed so far (one placed
// circle per lane + a ballot instead of a loop of up to 29 square roots on one lane; it was 60 % of the placement's latency).
template <typename ObsT>
__device__ __forceinline__ void reset_place(Ctx<ObsT> &c, double *placed /* [5][cap] x y r sight iscam */, int placed_cap,
                                            const double *pre = nullptr, uint32_t pre_count = 0) {
    const Params &p = c.p;
    const int Nc = p.Nc, Nt = p.Nt, No = p.No;
    const uint32_t episode = (uint32_t)c.ei(EI_EPISODE) + 1u;
    c.ei(EI_EPISODE) = (int32_t)episode;
    ResetRng rng{p.seed_lo, p.seed_hi, c.env_global(), episode, 0u,
                 c.g.reset_tape ? c.g.reset_tape + c.env * (int64_t)c.g.reset_tape_len : nullptr, (uint32_t)c.g.reset_tape_len, pre, pre_count};
    double *px = placed, *py = placed + placed_cap, *pr = placed + 2 * placed_cap, *ps = placed + 3 * placed_cap, *pk = placed + 4 * placed_cap;
    int32_t *perm = reinterpret_cast<int32_t *>(placed + 5 * placed_cap);   // [Nc + Nt + No]
    int32_t *perm_c = perm, *perm_t = perm + Nc, *perm_o = perm + Nc + Nt;
    // shuffles (environment.py:707-710): Fisher-Yates on the range indices
    for (int which = 0; which < 3; ++which) {
        int32_t *pp = which == 0 ? perm_c : (which == 1 ? perm_t : perm_o);
        const int n = which == 0 ? Nc : (which == 1 ? Nt : No);
        for (int i = 0; i < n; ++i) pp[i] = i;
        if (p.shuffle) for (int i = n - 1; i >= 1; --i) { const int j = rng.randint(i + 1); const int tmp = pp[i]; pp[i] = pp[j]; pp[j] = tmp; }
    }
    // capacities (environment.py:712-722)
    uint64_t capword = 0;
    if (p.n_high > 0) {
        if (p.shuffle) {
            int32_t *idx = perm + Nc + Nt + No;   // scratch [Nt]
            for (int i = 0; i < Nt; ++i) idx[i] = i;
            for (int i = 0; i < p.n_high; ++i) { const int j = i + rng.randint(Nt - i); const int tmp = idx[i]; idx[i] = idx[j]; idx[j] = tmp; capword |= 1ull << idx[i]; }
        } else {
            for (int i = 0; i < p.n_high; ++i) capword |= 1ull << i;
        }
    }
    reinterpret_cast<uint64_t *>(c.st)[3 * Nc + 3 * No] = capword;
    // placement (environment.py:724-737)
    int np = 0;
    for (int w = 0; w < 4; ++w) {
        px[np] = (w == 0 || w == 3) ? kWarehouseCenter : -kWarehouseCenter;
        py[np] = (w < 2) ? kWarehouseCenter : -kWarehouseCenter;
        pr[np] = 0.75 * kWarehouseRadius; ps[np] = 0.0; pk[np] = 0.0; ++np;
    }
    const int total = Nc + No + Nt;
    for (int k = 0; k < total; ++k) {
        const int kind = k < Nc ? 0 : (k < Nc + No ? 1 : 2);
        const int i = kind == 0 ? k : (kind == 1 ? k - Nc : k - Nc - No);
        const int ridx = kind == 0 ? perm_c[i] : (kind == 1 ? Nc + perm_o[i] : Nc + No + perm_t[i]);
        const double *range = c.g.reset_ranges + 4 * ridx;
        const double xlo = range[0], xhi = range[1], ylo = range[2], yhi = range[3];
        const double min_distance = kind == 2 ? 0.0 : p.tgt_step;
        double x = 0, y = 0, rad = 0, sight = 0, phi = 0, theta = 0;
        bool ok = false;
        for (int attempt = 0; attempt < 500 && !ok; ++attempt) {
            rad = 0.0;
            // Camera(Sensor, Obstacle) resets through Obstacle.reset too: it samples its (degenerate) radius box first (entities.py:235, 150-152)
            if (kind == 0) rad = p.cam_radius + (p.cam_radius - p.cam_radius) * rng.draw();
            if (kind == 1) rad = p.obs_r_lo + (p.obs_r_hi - p.obs_r_lo) * rng.draw();   // entities.py:151
            const double sx = xlo + (xhi - xlo) * rng.draw();                               // entities.py:61
            const double sy = ylo + (yhi - ylo) * rng.draw();
            const double lim = kTerrain - 1.2 * rad;                                        // entities.py:62-65
            x = clipd(sx, -lim, lim); y = clipd(sy, -lim, lim);
            sight = 0.0;
            if (kind == 0) {                                                                // entities.py:326-334
                const int nsteps = (int)(360.0 / p.rot);
                phi = normalize_angle(p.rot * (double)rng.randint(nsteps));
                theta = p.theta_min + (kMaxViewingAngle - p.theta_min) * rng.draw();
                sight = sqrt(p.area / theta);
            }
            // overlap against everything placed so far (entities.py:96-100, 484-489): one placed circle per lane
            bool bad = false;
            for (int q0 = 0; q0 < np; q0 += 64) {
                const int q = q0 + c.lane;
                if (q < np) {
                    const double d = norm2(x - px[q], y - py[q]);
                    if (d * (1.0 + 1e-6) < rad + pr[q] + min_distance) bad = true;
                    else if (kind == 0 && pk[q] != 0.0) { const double m = sight < ps[q] ? sight : ps[q]; if (d < 0.1 * m) bad = true; }
                }
            }
            ok = __ballot(bad) == 0ull;
        }
        if (!ok && kind == 1) rad = 0.0;                                                    // environment.py:735-736
        px[np] = x; py[np] = y; pr[np] = rad; ps[np] = sight; pk[np] = kind == 0 ? 1.0 : 0.0; ++np;
        if (kind == 0) { c.st[i] = x; c.st[Nc + i] = y; c.phi(i) = phi; c.theta(i) = theta; }
        else if (kind == 1) { c.st[2 * Nc + i] = x; c.st[2 * Nc + No + i] = y; c.st[2 * Nc + 2 * No + i] = rad; }
        else { c.tx(i) = x; c.ty(i) = y; }
    }
    // cargo matrix (environment.py:768-775)
    int32_t *remaining = &c.ei(EI_REMAINING);
    for (int i = 0; i < 16; ++i) remaining[i] = 0;
    for (;;) {
        for (int k = 0; k < p.num_cargoes_per_target * Nt; ++k) {
            // choice(4, size=2, replace=False): partial Fisher-Yates on (0, 1, 2, 3)
            const int s = rng.randint(4);
            int r = 1 + rng.randint(3);
            if (r == s) r = 0;
            remaining[4 * s + r] += 1;
        }
        bool all = true;
        for (int s = 0; s < 4; ++s) all = all && row_any(remaining + 4 * s);
        for (int g = 0; g < 4; ++g) c.ei(EI_AWAITING + g) = remaining[g] + remaining[4 + g] + remaining[8 + g] + remaining[12 + g];
        if (all) break;
        if (rng.tape && rng.n > rng.tape_len) break;      // an exhausted tape yields zeros for ever: reported through reset_draws, not looped on
    }
    for (int t = 0; t < Nt; ++t) {                                                          // environment.py:777-783
        c.ti(t, TI_BOUNTY) = 0; c.ti(t, TI_FREIGHT) = 0; c.ti(t, TI_GW) = 0; c.ti(t, TI_TSTEPS) = 0; c.ti(t, TI_TRSTEPS) = 0;
    }
    // _assign_goals at reset (environment.py:784): only targets that spawn inside a warehouse pick up here
    for (int t = 0; t < Nt; ++t) {
        const double x = c.tx(t), y = c.ty(t);
        for (int w = 0; w < 4; ++w) {
            const double wx = (w == 0 || w == 3) ? kWarehouseCenter : -kWarehouseCenter;
            const double wy = (w < 2) ? kWarehouseCenter : -kWarehouseCenter;
            if (!(fmax(fabs(x - wx), fabs(y - wy)) <= kWarehouseRadius)) continue;
            int gw = c.ti(t, TI_GW);
            bool picked = false;
            if ((gw & 0xff) == 0) {   // no goal yet (a goal-carrying target cannot exist here)
                int32_t *row = remaining + 4 * w;
                if (row_any(row)) {
                    const int goal = pick_goal(row, rng.draw());
                    const int cap = 1 + (int)((capword >> t) & 1ull);
                    const int weight = cap < row[goal] ? cap : row[goal];
                    row[goal] -= weight;
                    c.ti(t, TI_FREIGHT) = (int)((double)weight * p.freight_scale);
                    c.ti(t, TI_BOUNTY) = (int)((double)weight * p.bounty_scale);
                    gw = (goal + 1) | (weight << 8);
                    picked = true;
                }
            }
            const bool empty = !row_any(remaining + 4 * w);
            gw = (gw & ~(1 << (16 + w))) | ((int)empty << (16 + w));
            c.ti(t, TI_GW) = gw;
            if (picked) break;
        }
    }
    c.ei(EI_DELIVERED) = 0; c.ep_reward() = 0.0; c.ep_delayed() = 0.0;
    if (p.start_with_cargoes) {                                                             // environment.py:789-807
        for (int t = 0; t < Nt; ++t) {
            if ((c.ti(t, TI_GW) & 0xff) != 0) continue;
            int32_t *wp = perm + Nc + Nt + No;   // LDS scratch [4]
            for (int i = 0; i < 4; ++i) wp[i] = i;
            for (int i = 3; i >= 1; --i) { const int j = rng.randint(i + 1); const int tmp = wp[i]; wp[i] = wp[j]; wp[j] = tmp; }
            for (int q = 0; q < 4; ++q) {
                int32_t *row = remaining + 4 * wp[q];
                if (!row_any(row)) continue;
                const int goal = pick_goal(row, rng.draw());
                const int cap = 1 + (int)((capword >> t) & 1ull);
                const int weight = cap < row[goal] ? cap : row[goal];
                row[goal] -= weight;
                c.ti(t, TI_FREIGHT) = (int)((double)weight * p.freight_scale);
                c.ti(t, TI_BOUNTY) = (int)((double)weight * p.bounty_scale);
                c.ti(t, TI_GW) = (c.ti(t, TI_GW) & ~0xffff) | (goal + 1) | (weight << 8);
                break;
            }
        }
    }
    // (pipelined restarts: live from the next launch with this reset's list parity on, see Ptrs::pipelined)
    c.ei(EI_EPSTEP) = 0; c.ei(EI_DONE) = c.g.pipelined ? (kDoneTag | ((c.g.parity & 1) << 3)) : 0;
    if (c.g.reset_draws) c.g.reset_draws[c.env] = (int32_t)((rng.tape && rng.n > rng.tape_len) ? -1 : (int32_t)rng.n);
}

// R2: occlusion table of camera `cam` by the whole workgroup.
template <typename ObsT>
__device__ __forceinline__ bool build_lut(Ctx<ObsT> &c, int cam, double *keys, double *vals, double *okeys, double *ovals, uint16_t *lbucket,
                          double *meta, int32_t *scan, int sort_cap, bool outer = false, bool in_hbm = false, int defer_above = 0) {
    const Params &p = c.p;
    const int tid = threadIdx.x, nthreads = blockDim.x;
    const int No = p.No;
    // meta rows (stride = No): 0 relx 1 rely 2 rel_norm 3 rad 4 a_left 5 a_right 6 step 7 max_rho ; ints: num, offset
    double *m_relx = meta, *m_rely = meta + No, *m_rn = meta + 2 * No, *m_rad = meta + 3 * No;
    double *m_al = meta + 4 * No, *m_ar = meta + 5 * No, *m_step = meta + 6 * No, *m_rho = meta + 7 * No;
    int32_t *m_num = reinterpret_cast<int32_t *>(meta + 8 * No);
    int32_t *m_off = m_num + No;
    int32_t *hdr = m_off + No;    // [0] nrays [1] degenerate [2] kept count
    const double cx = c.cam_x(cam), cy = c.cam_y(cam);
    for (int o = tid; o < No; o += nthreads) {
        const double relx = c.obs_x(o) - cx, rely = c.obs_y(o) - cy, rad = c.obs_r(o);
        const double rn = norm2(relx, rely);
        const bool in_range = rn < p.rmax + rad;                 // entities.py:365 (strict)
        int num = 0;
        m_relx[o] = relx; m_rely[o] = rely; m_rn[o] = rn; m_rad[o] = rad;
        if (in_range && p.tau != 1.0) {
            if (rad > rn) { num = -1; }                          // entities.py:378: camera inside the obstacle
            else {
                const double half = asin(rad / rn) * kRad2Deg;   // entities.py:389
                const double far = rn + rad;
                m_rho[o] = far < p.rmax ? far : p.rmax;          // entities.py:390
                const double ra = atan2_deg(rely, relx);
                const double al = ra - half, ar = ra + half;
                int n = (int)(2.0 * half);
                if (n < 16) n = 16;
                num = n + 1;                                     // entities.py:413
                m_al[o] = al; m_ar[o] = ar; m_step[o] = (ar - al) / (double)(num - 1);
            }
        }
        m_num[o] = in_range ? (num == 0 ? -2 : num) : 0;         // -2: in range but transparent (tau == 1)
    }
    __syncthreads();
    if (tid == 0) {
        int off = 360, degenerate = 0;
        uint64_t bits = 0;
        for (int o = 0; o < No; ++o) {
            const int num = m_num[o];
            if (num != 0) bits |= 1ull << o;
            m_off[o] = off;
            if (num > 0) off += outer ? num + 42 : 4 + num;          // boundary_outer: arc + two 21-point flanks (entities.py:419-448)
            if (num == -1) degenerate = 1;
        }
        hdr[0] = off; hdr[1] = degenerate;
        if (!outer) reinterpret_cast<uint64_t *>(c.st)[2 * p.Nc + 3 * No + cam] = bits;   // camera_obstacle_view_mask row
    }
    __syncthreads();
    const int nr = hdr[0];
    // Two-tier launches: the sort arrays of this launch hold `defer_above` rays (half of the worst case, so that four
    // workgroups fit a CU instead of two); a table with more -- obstacles filling most of a camera's horizon -- is put on a list
    // and built by the full-size launch behind this one.
    if (defer_above > 0 && nr >= defer_above) {            // (the closing knot takes slot nr)
        if (tid == 0) { const int slot = atomicAdd(c.g.lut_overflow, 1); c.g.lut_overflow[1 + slot] = (int32_t)(c.env * p.Nc + cam); }
        __syncthreads();
        return true;
    }
    const int64_t lc = c.env * p.Nc + cam;
    double2 *knots = outer ? c.g.lut_knots_outer + lc * c.g.kmax_outer : c.g.lut_knots + lc * p.kmax;
    int32_t *knot_count = outer ? c.g.lut_count_outer + lc : c.g.lut_count + lc;
    uint16_t *bucket = c.g.lut_bucket + lc * p.nbucket;
    if (hdr[1]) {   // fully blocked view
        if (tid == 0) { knots[0] = make_double2(-180.0, 0.0); knots[1] = make_double2(180.0, 0.0); *knot_count = 2; }
        if (outer) { __syncthreads(); return false; }
        for (int d = tid; d < p.nbucket; d += nthreads) bucket[d] = d >= 360 ? 1 : 0;
        for (int d = tid; d < kLutCells; d += nthreads) {
            double2 *rec = c.g.lut_deg + (lc * kLutCells + d) * kDegWords;
            degree_record_set(rec, 0, -180.0, 0.0, 0.0);      // (one segment from -180 to 180, range 0)
            for (int i = 1; i < kDegSlots - 1; ++i) degree_record_set(rec, i, __longlong_as_double(0x7ff0000000000000ll), 0.0, 0.0);
        }
        __syncthreads();
        return false;
    }
    int P = 512;
    while (P < nr) P <<= 1;          // nr <= sort_cap by construction
    (void)sort_cap;
    // The 360 integer-degree rays are generated in angle order: only the obstacle rays (edge + arc rays, typically ~20 per
    // obstacle in range) are sorted -- in okeys / ovals -- and then merged with the degree grid by rank.  (Sorting all
    // 360 + 21 No rays bitonically was 55 % of a reset's GPU time.)
    const int nobs = nr - 360;
    int P2 = 64;
    while (P2 < nobs) P2 <<= 1;
    for (int i = tid; i < 360 + P2; i += nthreads) {
        double key = __longlong_as_double(0x7ff0000000000000ll), val = 0.0;
        if (i < nr) {
            double a, n0;
            if (i < 360) { a = -180.0 + (double)i; n0 = p.rmax; }                // entities.py:336-339
            else {
                int o = 0;
                for (int q = 0; q < No; ++q) if (m_num[q] > 0 && i >= m_off[q]) o = q;
                int j = i - m_off[o];
                const double al = m_al[o], ar = m_ar[o];
                if (!outer && j < 4) { a = (j < 2 ? al : ar) + ((j & 1) ? 0.01 : -0.01); n0 = p.rmax; }   // entities.py:395-406
                else {
                    if (!outer) j -= 4;
                    if (j < m_num[o]) { a = (j == m_num[o] - 1) ? ar : ((double)j * m_step[o] + al); n0 = m_rho[o]; }  // :409-415, :419-428
                    else {
                        // boundary_outer flanks (entities.py:430-448): 21 points on the segment from the tangent point
                        // to the range limit just outside the obstacle's shadow, kept as cartesian vectors
                        const int f = j - m_num[o];
                        const bool right = f >= 21;
                        const double t = (double)(right ? f - 21 : f) * 0.05;
                        const double near_rho0 = sqrt(m_rn[o] * m_rn[o] + m_rad[o] * m_rad[o]);
                        const double near_rho = near_rho0 < p.rmax ? near_rho0 : p.rmax;
                        double sn0, cs0, sn1, cs1;
                        sincos_deg(normalize_angle(right ? ar : al), sn0, cs0);
                        sincos_deg(normalize_angle(right ? ar + 0.01 : al - 0.01), sn1, cs1);
                        const double x = (1.0 - t) * (near_rho * cs0) + t * (p.rmax * cs1);
                        const double y = (1.0 - t) * (near_rho * sn0) + t * (p.rmax * sn1);
                        a = atan2_deg(y, x); n0 = norm2(x, y);
                        // t = 0 is the tangent direction itself: the reference gets it back from an atan2(sin, cos) round
                        // trip that usually, but not always, returns the angle bit for bit (and then merges it with the
                        // arc's end ray); the exact angle is used here
                        if (f == 0 || f == 21) { a = right ? ar : al; n0 = near_rho; }
                    }
                }
                a = normalize_angle(a);
            }
            double sn, cs;
#ifdef MATE_ABLATE_DEGREE_SINCOS      // (experiment: what a sin / cos TABLE for the 360 integer-degree rays could save at most -- their evaluation made free, wrong values)
            if (i < 360) { sn = 0.0; cs = 1.0; } else
#endif
            sincos_deg(a, sn, cs);
            for (int q = 0; q < No; ++q) {                                        // entities.py:450-455
                if (m_num[q] <= 0) continue;
                // a ray whose LINE passes the circle at more than its radius (+ 1e-6 relative and absolute: a million times the
                // rounding of clip_polar's own perpendicular) is returned unchanged by Obstacle.obstruct: two fmas screen it out
                const double relx = m_relx[q], rely = m_rely[q], rad = m_rad[q];
                if (fabs(fma(relx, sn, -(rely * cs))) > fma(rad, 1e-6, rad) + 1e-6) continue;
                n0 = clip_polar(n0, cs, sn, relx, rely, m_rn[q], rad, outer);
            }
            key = a; val = n0;
        }
        if (i < 360) { keys[i] = key; vals[i] = val; } else { okeys[i - 360] = key; ovals[i - 360] = val; }
    }
    __syncthreads();
    // bitonic sort of the obstacle rays by angle (entities.py:458); equal angles are merged below, so stability is moot
    for (int k = 2; k <= P2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < P2; i += nthreads) {
                const int l = i ^ j;
                if (l > i) {
                    const double ki = okeys[i], kl = okeys[l];
                    const bool up = (i & k) == 0;
                    if ((ki > kl) == up && ki != kl) {
                        okeys[i] = kl; okeys[l] = ki;
                        const double vi = ovals[i]; ovals[i] = ovals[l]; ovals[l] = vi;
                    }
                }
            }
            // partners at a distance below 64 sit in the same wave (i = tid + n * nthreads): between two such stages no workgroup
            // barrier is needed
            const int next_j = j > 1 ? (j >> 1) : k;
            if (j >= 64 || next_j >= 64 || in_hbm) __syncthreads(); else wave_sync();
        }
    }
    __syncthreads();
    // merge by rank into keys / vals: a degree ray goes behind the obstacle rays with a smaller angle, an obstacle ray behind the
    // degree rays with an angle <= its own (ties: the degree ray first; both counts use exact comparisons, so the positions are
    // a permutation)
    {
        double held[6];                                                   // this thread's degree rays leave keys / vals first
#pragma unroll
        for (int n = 0; n < 6; ++n) { const int d = tid + n * nthreads; held[n] = d < 360 ? vals[d] : 0.0; }
        __syncthreads();
#pragma unroll
        for (int n = 0; n < 6; ++n) {
            const int d = tid + n * nthreads;
            if (d < 360) {
                const double a = -180.0 + (double)d;
                int lo = 0, hi = nobs;                                    // first obstacle ray with angle >= a
                while (lo < hi) { const int mid = (lo + hi) >> 1; if (okeys[mid] < a) lo = mid + 1; else hi = mid; }
                keys[d + lo] = a; vals[d + lo] = held[n];
            }
        }
        for (int sidx = tid; sidx < nobs; sidx += nthreads) {
            const double x = okeys[sidx];
            int before = (int)floor(x + 180.0) + 1;                       // degree rays -180 .. with angle <= x (x is in [-180, 180))
            before = before < 0 ? 0 : (before > 360 ? 360 : before);
            if (before > 0 && (double)(before - 1) - 180.0 > x) --before;         // x + 180 rounded up across an integer
            if (before < 360 && (double)before - 180.0 <= x) ++before;
            keys[sidx + before] = x; vals[sidx + before] = ovals[sidx];
        }
    }
    __syncthreads();
    // dedupe equal angles keeping the smaller norm (entities.py:460-466) + compaction
    const int per = (P + nthreads - 1) / nthreads;
    const int lo = tid * per, hi = (lo + per < nr) ? lo + per : nr;
    int local = 0;
    for (int i = lo; i < hi; ++i) local += (i == 0 || keys[i] != keys[i - 1]);
    // exclusive prefix over the threads: inside each wave by shuffles, across the (at most 16) waves through `scan`
    int incl = local;
    {
        const int ln = tid & 63;
        for (int off = 1; off < 64; off <<= 1) { const int up = __shfl_up(incl, off, 64); if (ln >= off) incl += up; }
        if (ln == 63) scan[tid >> 6] = incl;
    }
    __syncthreads();
    int pos = incl - local;
    {
        const int nw = (nthreads + 63) >> 6;
        int total = 0;
        for (int wv = 0; wv < nw; ++wv) { const int v = scan[wv]; if (wv < (tid >> 6)) pos += v; total += v; }
        if (tid == 0) hdr[2] = total;
    }
    __syncthreads();
    for (int i = lo; i < hi; ++i) {
        if (i == 0 || keys[i] != keys[i - 1]) {
            const double a = keys[i];
            double rho = vals[i];
            for (int q = i + 1; q < nr && keys[q] == a; ++q) rho = vals[q] < rho ? vals[q] : rho;
            okeys[pos] = a; ovals[pos] = rho;
            if (!outer && a == floor(a)) lbucket[(int)a + 180] = (uint16_t)pos;     // per-degree index
            ++pos;
        }
    }
    __syncthreads();
    const int m = hdr[2];
    if (tid == 0) {
        okeys[m] = okeys[0] + 360.0; ovals[m] = ovals[0];                 // entities.py:470-471
        lbucket[360] = (uint16_t)m; lbucket[361] = (uint16_t)m;
        *knot_count = m + 1;
    }
    __syncthreads();
    for (int i = tid; i <= m; i += nthreads) knots[i] = make_double2(okeys[i], ovals[i]);
    if (outer) { __syncthreads(); return false; }                          // boundary_between(outer=True) only reads the knots
    for (int d = tid; d < p.nbucket; d += nthreads) bucket[d] = d <= 361 ? lbucket[d] : (uint16_t)m;
    const double inf = __longlong_as_double(0x7ff0000000000000ll);
    for (int cell = tid; cell < kLutCells; cell += nthreads) {       // per-cell records of the fast lookup path
        // the cell's knots: from the last one at or below its start (a cell that starts on an integer degree starts on a knot) up to,
        // not including, the first one at or above the next cell's start, which closes the last segment
        const int d = cell / kCellsPerDegree, sub = cell - d * kCellsPerDegree;
        int start = lbucket[d];
        if (sub > 0) {
            const double a = cell_start(cell);
            while (okeys[start] < a) ++start;
            if (okeys[start] != a) --start;
        }
        int endk = lbucket[d + 1];
        if (sub < kCellsPerDegree - 1) {
            const double a = cell_start(cell + 1);
            endk = start;
            while (okeys[endk] < a) ++endk;
        }
        double2 *rec = c.g.lut_deg + (lc * kLutCells + cell) * kDegWords;
        if (endk - start + 1 <= kDegSlots) {
            for (int i = 0; i < kDegSlots - 1; ++i) {          // the degree's knots as segments (np.interp's slope, its own division)
                const int idx = start + i;
                if (idx < endk) degree_record_set(rec, i, okeys[idx], ovals[idx], (ovals[idx + 1] - ovals[idx]) / (okeys[idx + 1] - okeys[idx]));
                else degree_record_set(rec, i, inf, 0.0, 0.0);
            }
        } else {                                   // overflow: (NaN, first knot) (knots of the degree incl. the next integer one, pivot stride) + 8 pivot angles
            const int count = endk - start + 1, last = count - 1;
            const int q = count <= kPivotKnots ? pivot_stride(count) : 0;
            rec[0] = make_double2(__longlong_as_double(0x7ff8000000000000ll), (double)start);
            rec[1] = make_double2((double)count, (double)q);
            for (int i = 0; i < 4; ++i) {
                const int ka = (2 * i + 1) * q, kb = (2 * i + 2) * q;
                rec[2 + i] = make_double2(q > 0 && ka <= last ? okeys[start + ka] : inf, q > 0 && kb <= last ? okeys[start + kb] : inf);
            }
        }
    }
    __syncthreads();
    return false;
}

// One launch can run all three phases for an environment in one workgroup, or the host splits them
// (launch_reset): placement by one wave per environment, then ONE WORKGROUP PER (environment, camera) for the
// occlusion tables -- they are independent once the geometry is placed, and with 8 cameras the tables are
// 2/3 of a reset's latency -- then the first view by one wave per environment.  Under RESET_FLAGGED the
// selection is the `done` word itself, which the placement clears: with PH_MORE the placement launch appends
// the environments it resets to `flag_list`, and the later launches (RESET_LIST) walk that list with a small
// grid instead of scanning N x Nc workgroups that reserve the sort's LDS only to find nothing to do.
template <typename ObsT>
__global__ __launch_bounds__(256) void reset_kernel(const Params *__restrict__ pp, const Ptrs g, const ResetLds rl, const int32_t phases) {
    const Params &p = *pp;
    extern __shared__ __align__(16) unsigned char smem[];
    // device-resident step counter (Params::dev_tick): the immediate auto-reset launch behind a step takes the list
    // parity that step used and advances the counter for the next one (this kernel touches the counter only here)
    const int32_t parity = (p.dev_tick_on && g.reset_kind == RESET_DONE) ? g.ctrl[0] : g.parity;
    // (a reset split into several launches advances it in the last one: tick_advance = 0 in the others; only the auto-reset launches of
    // the flows that run with the counter on the device carry one: list-driven behind the per-step flows, by flag behind K-frame launches)
    if (p.dev_tick_on && g.tick_advance != 0u && blockIdx.x == 0 && threadIdx.x == 0) { g.dev_tick_ptr[0] += g.tick_advance; g.dev_tick_ptr[1] += 1u; }
    if (g.reset_kind == RESET_DONE && (int64_t)blockIdx.x >= (int64_t)g.done_count[parity] * ((phases & PH_PER_CAMERA) ? p.Nc : 1)) return;   // idle: nothing finished
    if (g.reset_kind == RESET_LIST && (int64_t)blockIdx.x >= (int64_t)g.flag_count[0] * ((phases & PH_PER_CAMERA) ? p.Nc : 1)) return;
    const bool pairs = g.reset_kind == RESET_PAIRS;
    if (pairs && (int)blockIdx.x >= g.lut_overflow[0]) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned char *wave_base = smem;
    double *keys = reinterpret_cast<double *>(smem + rl.off_keys);
    double *vals = reinterpret_cast<double *>(smem + rl.off_vals);
    double *okeys = reinterpret_cast<double *>(smem + rl.off_okeys);
    double *ovals = reinterpret_cast<double *>(smem + rl.off_ovals);
    double *place_scratch = keys;     // reset_place's list of placed circles: always in the LDS
    double *pre_draws = reinterpret_cast<double *>(smem + rl.off_pre);
    if (rl.sort_in_hbm) {      // same code on global memory: __syncthreads orders a workgroup's global accesses as well
        keys = g.sort_scratch + (int64_t)blockIdx.x * 4 * rl.sort_cap;
        vals = keys + rl.sort_cap; okeys = vals + rl.sort_cap; ovals = okeys + rl.sort_cap;
    }
    uint16_t *lbucket = reinterpret_cast<uint16_t *>(smem + rl.off_bucket);
    double *meta = reinterpret_cast<double *>(smem + rl.off_meta);
    int32_t *scan = reinterpret_cast<int32_t *>(smem + rl.off_scan);
    const bool per_camera = (phases & PH_PER_CAMERA) != 0;
    const int fan = (per_camera && !pairs) ? p.Nc : 1;
    int64_t count = g.N;
    if (g.reset_kind == RESET_DONE) count = g.done_count[parity];
    if (g.reset_kind == RESET_LIST) count = g.flag_count[0];
    if (pairs) count = g.lut_overflow[0];
    for (int64_t v = blockIdx.x; v < count * fan; v += gridDim.x) {
        int64_t item = per_camera ? v / p.Nc : v;
        int only_cam = per_camera ? (int)(v - item * p.Nc) : -1;
        if (pairs) { const int32_t entry = g.lut_overflow[1 + v]; item = entry / p.Nc; only_cam = entry - (int32_t)item * p.Nc; }
        const int64_t env = g.reset_kind == RESET_DONE ? (int64_t)g.done_list[(int64_t)parity * g.N + item]
                          : g.reset_kind == RESET_LIST ? (int64_t)g.flag_list[item] : item;
        if (g.reset_kind == RESET_MASK && !g.reset_mask[env]) continue;
        if (g.reset_kind == RESET_DONE) {
            // A listed environment that somebody else restarted since (reset(env_mask), reset_tape, a whole-batch reset under a
            // device-resident step counter) is in the middle of a NEW episode: leave it alone.  The launch that places voids
            // the list entry, so that the table / view launches of a split reset skip it too.
            if (env < 0) continue;
            if (phases & PH_PLACE) {
                const int done = reinterpret_cast<const int32_t *>(g.dyn + env * p.DW + p.DF)[p.Nt * TI_STRIDE + EI_DONE];
                if (done == 0) {
                    if (threadIdx.x == 0) g.done_list[(int64_t)parity * g.N + item] = -1;
                    continue;
                }
            }
        }
        if (g.reset_kind == RESET_FLAGGED) {
            const int done = reinterpret_cast<const int32_t *>(g.dyn + env * p.DW + p.DF)[p.Nt * TI_STRIDE + EI_DONE];
            if (done == 0) continue;
        }
        __syncthreads();
        Ctx<ObsT> c(p, g, wave_base, lane, env);
        if (wave == 0) {
            load_records(c);
            wave_sync();
            constexpr int kPreDraws = 256;      // a reset of the shipped scenarios takes 130-220 draws; beyond, draw() falls back to one Philox call per draw
            if ((phases & PH_PLACE) && !g.reset_tape) {
                // one Philox-4x32-10 block gives two 53-bit uniforms: the 64 lanes draw the stream's first 256 at once (the single
                // placing lane used to spend half of its time on one block per draw)
                const uint32_t episode = (uint32_t)c.ei(EI_EPISODE) + 1u;
                for (int b = lane; b < kPreDraws / 2; b += 64) {
                    const U4 r = philox(p.seed_lo, p.seed_hi, c.env_global(), episode, S_RESET, (uint32_t)b);
                    pre_draws[2 * b] = u53(r.x, r.y); pre_draws[2 * b + 1] = u53(r.z, r.w);
                }
                wave_sync();
            }
            if (phases & PH_PLACE) reset_place(c, place_scratch, 4 + p.Nc + p.No + p.Nt, pre_draws, g.reset_tape ? 0u : (uint32_t)kPreDraws);
            if ((phases & PH_PLACE) && lane == 0) {
                if ((g.reset_kind == RESET_FLAGGED || g.reset_kind == RESET_MASK) && (phases & PH_MORE)) g.flag_list[atomicAdd(g.flag_count, 1)] = (int32_t)env;
            }
            wave_sync();
        }
        __syncthreads();
        if (phases & PH_LUT) {
            if (per_camera) {
                const bool deferred = build_lut(c, only_cam, keys, vals, okeys, ovals, lbucket, meta, scan, rl.sort_cap, false, rl.sort_in_hbm != 0, g.lut_defer_above);
                if (g.lut_knots_outer && !deferred)       // (deferred here, with the inner table built: the full-size launch rebuilds both)
                    build_lut(c, only_cam, keys, vals, okeys, ovals, lbucket, meta, scan, rl.sort_cap, true, rl.sort_in_hbm != 0, g.lut_defer_above);
                // this workgroup owns exactly one word of the static record: the camera's obstacle mask row
                if (threadIdx.x == 0) {
                    const int w = 2 * p.Nc + 3 * p.No + only_cam;
                    reinterpret_cast<uint64_t *>(g.stat + env * p.SW)[w] = reinterpret_cast<const uint64_t *>(c.st)[w];
                }
                continue;
            }
            for (int cam = 0; cam < p.Nc; ++cam) {
                build_lut(c, cam, keys, vals, okeys, ovals, lbucket, meta, scan, rl.sort_cap, false, rl.sort_in_hbm != 0);
                if (g.lut_knots_outer) build_lut(c, cam, keys, vals, okeys, ovals, lbucket, meta, scan, rl.sort_cap, true, rl.sort_in_hbm != 0);
            }
        }
        __syncthreads();
        if (wave == 0) {
            // static record back to HBM
            if (phases & (PH_PLACE | PH_LUT)) {
                double *s = g.stat + env * p.SW;
                for (int i = lane; i < p.SW; i += 64) s[i] = c.st[i];
            }
            if (phases & PH_VIEW) {
                __threadfence();   // this wave reads the tables other waves / workgroups just wrote
                build_entities(c);
                simulate_cameras(c, StepDraws{0.0, 0.0}, false);   // camera sight + scratch only
                wave_sync();
                update_view(c, (uint32_t)c.ei(EI_EPISODE), S_RESET_VIEW, false);
                score_only(c, g.scalars);
                fill_scratch(c);
                { PackDescriptors d; pack_observations<false>(c, d); }
            }
            if (phases & (PH_PLACE | PH_VIEW)) store_dynamic(c);
        }
    }
}

// Leaving the pipelined-restart mode: every "restarted, live from launch parity q on" tag becomes a plain live environment.
__global__ void untag_kernel(const Params *__restrict__ pp, const Ptrs g) {
    const Params &p = *pp;
    const int64_t env = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= g.N) return;
    int32_t *e = reinterpret_cast<int32_t *>(g.dyn + env * p.DW + p.DF) + p.Nt * TI_STRIDE;
    if (e[EI_DONE] & kDoneTag) e[EI_DONE] = 0;
}

// seed(): every counter that enters a Philox counter word goes back to zero, so that (seed, environment index) alone
// determines the next reset() and everything after it (environment.py:1203-1227 re-creates its RandomStates).
__global__ void rewind_kernel(const Params *__restrict__ pp, const Ptrs g) {
    const Params &p = *pp;
    const int64_t env = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= g.N) return;
    int32_t *e = reinterpret_cast<int32_t *>(g.dyn + env * p.DW + p.DF) + p.Nt * TI_STRIDE;
    e[EI_EPISODE] = 0; e[EI_TICK] = 0;
}

// ---------------------------------------------------------------------------------------------
// canonical f64 export / import (one lane per environment; not on the hot path)
struct Exporter {
    const Params &p;
    __device__ int width() const { return p.export_width; }
};

__global__ void export_kernel(const Params *__restrict__ pp, const Ptrs g, double *dst) {
    const Params &p = *pp;
    const int64_t env = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= g.N) return;
    const double *st = g.stat + env * p.SW;
    const double *dy = g.dyn + env * p.DW;
    const int32_t *di = reinterpret_cast<const int32_t *>(dy + p.DF);
    const uint64_t *stw = reinterpret_cast<const uint64_t *>(st);
    double *o = dst + env * p.export_width;
    const int Nc = p.Nc, Nt = p.Nt, No = p.No;
    for (int i = 0; i < 2 * Nc + 3 * No; ++i) *o++ = st[i];                       // cam_x cam_y obs_x obs_y obs_r
    const uint64_t capword = stw[3 * Nc + 3 * No];
    for (int t = 0; t < Nt; ++t) *o++ = (double)(1 + (int)((capword >> t) & 1ull));
    for (int c = 0; c < Nc; ++c) for (int q = 0; q < No; ++q) *o++ = (double)((stw[2 * Nc + 3 * No + c] >> q) & 1ull);
    for (int i = 0; i < 2 * Nc + 2 * Nt; ++i) *o++ = dy[i];                       // cam_phi cam_theta tgt_x tgt_y
    for (int t = 0; t < Nt; ++t) *o++ = (double)((di[t * TI_STRIDE + TI_GW] >> 24) & 1);
    for (int t = 0; t < Nt; ++t) for (int w = 0; w < 4; ++w) *o++ = (double)((di[t * TI_STRIDE + TI_GW] >> (16 + w)) & 1);
    for (int t = 0; t < Nt; ++t) {
        const int gw = di[t * TI_STRIDE + TI_GW];
        const int goal = (gw & 0xff) - 1, weight = (gw >> 8) & 0xff;
        for (int w = 0; w < 4; ++w) *o++ = (double)(goal == w ? weight : 0);
    }
    for (int t = 0; t < Nt; ++t) *o++ = (double)((di[t * TI_STRIDE + TI_GW] & 0xff) - 1);
    for (int t = 0; t < Nt; ++t) *o++ = (double)di[t * TI_STRIDE + TI_FREIGHT];
    for (int t = 0; t < Nt; ++t) *o++ = (double)di[t * TI_STRIDE + TI_BOUNTY];
    for (int t = 0; t < Nt; ++t) *o++ = (double)di[t * TI_STRIDE + TI_TSTEPS];
    for (int t = 0; t < Nt; ++t) *o++ = (double)di[t * TI_STRIDE + TI_TRSTEPS];
    const int32_t *e = di + Nt * TI_STRIDE;
    for (int i = 0; i < 20; ++i) *o++ = (double)e[i];                             // remaining[16], awaiting[4]
    *o++ = (double)e[EI_DELIVERED];
    *o++ = dy[2 * Nc + 2 * Nt]; *o++ = dy[2 * Nc + 2 * Nt + 1];
    *o++ = (double)e[EI_EPSTEP]; *o++ = (double)(uint32_t)e[EI_TICK]; *o++ = (double)(uint32_t)e[EI_EPISODE]; *o++ = (double)e[EI_DONE];
}

__global__ void import_kernel(const Params *__restrict__ pp, const Ptrs g, const double *src) {
    const Params &p = *pp;
    const int64_t env = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= g.N) return;
    double *st = g.stat + env * p.SW;
    double *dy = g.dyn + env * p.DW;
    int32_t *di = reinterpret_cast<int32_t *>(dy + p.DF);
    uint64_t *stw = reinterpret_cast<uint64_t *>(st);
    const double *o = src + env * p.export_width;
    const int Nc = p.Nc, Nt = p.Nt, No = p.No;
    for (int i = 0; i < 2 * Nc + 3 * No; ++i) st[i] = *o++;
    uint64_t capword = 0;
    for (int t = 0; t < Nt; ++t) capword |= (uint64_t)((*o++) >= 2.0) << t;
    stw[3 * Nc + 3 * No] = capword;
    for (int c = 0; c < Nc; ++c) { uint64_t bits = 0; for (int q = 0; q < No; ++q) bits |= (uint64_t)((*o++) != 0.0) << q; stw[2 * Nc + 3 * No + c] = bits; }
    for (int i = 0; i < 2 * Nc + 2 * Nt; ++i) dy[i] = *o++;
    const double *colliding = o; o += Nt;
    const double *empty = o; o += 4 * Nt;
    const double *goal_bits = o; o += 4 * Nt;
    const double *goals = o; o += Nt;
    for (int t = 0; t < Nt; ++t) {
        const int goal = (int)goals[t];
        int weight = 0;
        for (int w = 0; w < 4; ++w) if (goal_bits[4 * t + w] != 0.0) weight = (int)goal_bits[4 * t + w];
        int gw = (goal + 1) | (weight << 8) | ((int)(colliding[t] != 0.0) << 24);
        for (int w = 0; w < 4; ++w) gw |= (int)(empty[4 * t + w] != 0.0) << (16 + w);
        di[t * TI_STRIDE + TI_GW] = gw;
    }
    for (int t = 0; t < Nt; ++t) di[t * TI_STRIDE + TI_FREIGHT] = (int)*o++;
    for (int t = 0; t < Nt; ++t) di[t * TI_STRIDE + TI_BOUNTY] = (int)*o++;
    for (int t = 0; t < Nt; ++t) di[t * TI_STRIDE + TI_TSTEPS] = (int)*o++;
    for (int t = 0; t < Nt; ++t) di[t * TI_STRIDE + TI_TRSTEPS] = (int)*o++;
    int32_t *e = di + Nt * TI_STRIDE;
    for (int i = 0; i < 20; ++i) e[i] = (int)*o++;
    e[EI_DELIVERED] = (int)*o++;
    dy[2 * Nc + 2 * Nt] = *o++; dy[2 * Nc + 2 * Nt + 1] = *o++;
    e[EI_EPSTEP] = (int)*o++; e[EI_TICK] = (int)(uint32_t)*o++; e[EI_EPISODE] = (int)(uint32_t)*o++; e[EI_DONE] = (int)*o++;
    e[EI_PAD] = 0;
}

}  // namespace mate
