// aux_kernels.hpp -- the one geometric term of the reference's AuxiliaryCameraRewards wrapper: the soft coverage
// score (wrappers/auxiliary_camera_rewards.py:128-139 reduction, 181-239 score).  Not on the step path: a caller
// that shapes camera rewards launches it after a step, on the state and the packed masks that step left behind.
//
// One wave per (environment, camera).  The score of a target is its distance to the nearest POINT of the
// camera's sector outline: 16 points up each flank, the two sector ends (interpolated on the inner occlusion
// table) and every knot of the OUTER table strictly inside the sector (Camera.boundary_between(outer=True),
// entities.py:513-543).  A minimum over a point set does not depend on the order of the points, so the lanes stride
// over the flank points and the raw outer table with a membership predicate instead of compacting the outline first;
// every lane keeps one running minimum per target in registers and six shuffles per target fold the wave.
#pragma once
#include "engine_kernels.hpp"

namespace mate {

constexpr int kFlankPoints = 16;     // np.linspace(0, rho, num=16, endpoint=False) up each flank
constexpr int kAuxMaxTargets = 16;

__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double o = __shfl_xor(v, off, 64);
        v = o < v ? o : v;
    }
    return v;
}

__global__ __launch_bounds__(256) void soft_coverage_kernel(const Params *pp, const Ptrs g, const uint32_t *masks,
                                                            double *matrix, double *scores) {
    const Params &p = *pp;
    __shared__ double rel_xy[4][2 * kAuxMaxTargets];
    __shared__ double row[4][kAuxMaxTargets];
    __shared__ int32_t seen_row[4][kAuxMaxTargets];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t item = (int64_t)blockIdx.x * 4 + wave;          // (environment, camera); waves never meet at a barrier
    if (item >= g.N * p.Nc) return;
    const int64_t env = item / p.Nc;
    const int cam = (int)(item - env * p.Nc);
    const double *st = g.stat + env * p.SW;
    const double *dy = g.dyn + env * p.DW;
    const double cx = st[cam], cy = st[p.Nc + cam];
    const double phi = dy[cam], theta = dy[p.Nc + cam];
    if (lane < p.Nt) {                                            // direction = target - camera (:230)
        rel_xy[wave][lane] = dy[2 * p.Nc + lane] - cx;
        rel_xy[wave][kAuxMaxTargets + lane] = dy[2 * p.Nc + p.Nt + lane] - cy;
    }
    wave_sync();

    const double sight = sqrt(p.area / theta);                    // Camera.sight_range, entities.py:360
    const double half = theta / 2.0;
    const double dist_max = theta < 180.0 ? sight / (1.0 + 1.0 / sin(half * kDeg2Rad)) : sight / 2.0;   // :203-206
    const double left = normalize_angle(phi - half);              // entities.py:516-519
    const double right = left + ((phi + half) - (phi - half));
    const double2 *inner = g.lut_knots + item * p.kmax;
    const uint16_t *bucket = g.lut_bucket + item * p.nbucket;
    const int n_inner = g.lut_count[item];
    const double rho_left = lut_lookup(inner, bucket, n_inner, left);                       // entities.py:538-541
    const double rho_right = lut_lookup(inner, bucket, n_inner, normalize_angle(right));
    const double2 *outer = g.lut_knots_outer + item * (int64_t)g.kmax_outer;
    const int n_outer = g.lut_count_outer[item];

    double best[kAuxMaxTargets];
#pragma unroll
    for (int t = 0; t < kAuxMaxTargets; ++t) best[t] = INFINITY;
    const int fixed = 2 * kFlankPoints + 2;
    for (int i = lane; i < fixed + n_outer; i += 64) {
        double ang, rho;
        bool member = true;
        if (i < 2 * kFlankPoints) {                               // flanks: phi of the end, rho = k * (rho_end / 16) (:213-221)
            const bool far_side = i >= kFlankPoints;
            ang = far_side ? right : left;
            rho = (double)(i & (kFlankPoints - 1)) * ((far_side ? rho_right : rho_left) / (double)kFlankPoints);
        } else if (i < fixed) {                                   // the two sector ends
            ang = i == fixed - 1 ? right : left;
            rho = i == fixed - 1 ? rho_right : rho_left;
        } else {                                                  // knots strictly inside the sector (entities.py:528-536)
            const double2 k = outer[i - fixed];
            ang = k.x; rho = k.y;
            member = right <= 180.0 ? (left < ang && ang < right)
                                    : ((left < ang && ang <= 180.0) || (ang > -180.0 && ang < right - 360.0));
        }
        if (!member) continue;
        double sn, cs;
        sincos_deg(ang, sn, cs);                                  // polar2cartesian, utils.py:144-152
        const double x = rho * cs, y = rho * sn;
#pragma unroll
        for (int t = 0; t < kAuxMaxTargets; ++t) {
            if (t < p.Nt) {
                const double ddx = rel_xy[wave][t] - x, ddy = rel_xy[wave][kAuxMaxTargets + t] - y;
                const double d2 = ddx * ddx + ddy * ddy;
                best[t] = d2 < best[t] ? d2 : best[t];
            }
        }
    }
    double mine = 0.0;
#pragma unroll
    for (int t = 0; t < kAuxMaxTargets; ++t) {
        if (t < p.Nt) {
            const double m = wave_min(best[t]);
            if (lane == t) mine = m;
        }
    }
    if (lane < p.Nt) {
        const int bit = cam * p.Nt + lane;                        // camera_target_view_mask in the packed words
        const bool seen = (masks[env * p.MW + (bit >> 5)] >> (bit & 31)) & 1u;
        double dist = sqrt(mine);                                 // min of hypot == hypot at the min
        if (!seen) dist = -dist;                                  // :233-234
        double score = dist / dist_max;
        if (n_outer < 2) score = NAN;                             // outer table never built for this environment
        if (matrix) matrix[item * p.Nt + lane] = score;
        row[wave][lane] = score;
        seen_row[wave][lane] = seen;
    }
    wave_sync();
    if (lane == 0 && scores) {                                    // :131-139: sum over the tracked targets, else tanh(max)
        bool any = false;
        double sum = 0.0, top = -INFINITY;
        for (int t = 0; t < p.Nt; ++t) {
            const double s = row[wave][t];
            if (seen_row[wave][t]) { any = true; sum += s; }
            top = s > top ? s : top;
        }
        scores[item] = any ? sum : tanh(top);
    }
}

}  // namespace mate
