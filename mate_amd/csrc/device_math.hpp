// device_math.hpp -- scalar f64 geometry used by the gfx950 kernels.
//
// Semantics follow the upstream reference (file:line cited per function); arithmetic is kept
// operation-for-operation (the translation unit is built with -ffp-contract=off) so that the
// only differences to the NumPy path are last-place differences of atan2/sincos/asin.
// 2-element dot products use fma(a1, b1, a0*b0): that is how the BLAS ddot behind
// np.inner / np.linalg.norm rounds them (probed in the build container).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mate {

constexpr double kRad2Deg = 180.0 / 3.14159265358979323846;  // utils.py:63
constexpr double kDeg2Rad = 3.14159265358979323846 / 180.0;  // utils.py:68
constexpr double kTerrain = 1000.0;                          // constants.py:52
constexpr double kWarehouseRadius = 75.0;                    // constants.py:67
constexpr double kWarehouseCenter = 925.0;                   // constants.py:70
constexpr double kMaxViewingAngle = 180.0;                   // constants.py:78

enum Stream : uint32_t { S_TRANSMIT = 1, S_GOAL = 2, S_ACT_CAM = 3, S_ACT_TGT = 4, S_RESET = 5, S_RESET_VIEW = 6 };

struct U4 { uint32_t x, y, z, w; };

// Philox-4x32-10 (Salmon et al., SC'11).  Same counter/key convention as oracle/mate_oracle.c.
__device__ __forceinline__ U4 philox(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return U4{c0, c1, c2, c3};
}

__device__ __forceinline__ double u53(uint32_t hi, uint32_t lo) {
    return ((double)(hi >> 5) * 67108864.0 + (double)(lo >> 6)) * (1.0 / 9007199254740992.0);
}

// uniform action component in [-m, m): f64 arithmetic, rounded once to f32 (the action dtype)
__device__ __forceinline__ double action_component(uint32_t r, double m) {
    return (double)(float)((double)(r >> 8) * 5.9604644775390625e-08 * (2.0 * m) - m);
}

__device__ __forceinline__ double dot2(double ax, double ay, double bx, double by) { return fma(ay, by, ax * bx); }
__device__ __forceinline__ double norm2(double x, double y) { return sqrt(fma(y, y, x * x)); }
__device__ __forceinline__ double clipd(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }

// Python float `%` with a positive divisor (utils.py:158 uses it with 360.0).
__device__ __forceinline__ double pymod_pos(double a, double b) {
    const double q = trunc(a / b);
    double r = fma(-q, b, a);          // exact when q is the true truncated quotient
    if (a >= 0.0) { if (r < 0.0) r += b; else if (r >= b) r -= b; }
    else          { if (r > 0.0) r -= b; else if (r <= -b) r += b; }
    if (r < 0.0) r += b;               // sign follows the divisor
    return r == 0.0 ? 0.0 : r;
}
__device__ __forceinline__ double normalize_angle(double a) { return pymod_pos(a + 180.0, 360.0) - 180.0; }  // utils.py:155-158
__device__ __forceinline__ double atan2_deg(double y, double x) { return atan2(y, x) * kRad2Deg; }           // utils.py:124-131

// Vector2D with its lazy polar <-> cartesian caches (utils.py:161-271).
struct Ray {
    double ox, oy, vx, vy, n, a;
    bool hv, hn, ha;
};
__device__ __forceinline__ void ray_materialize(Ray &r) {  // utils.py:177-181,144-152
    if (!r.hv) {
        double s, c;
        sincos(r.a * kDeg2Rad, &s, &c);
        r.vx = r.n * c; r.vy = r.n * s; r.hv = true;
    }
}
__device__ __forceinline__ double ray_angle(Ray &r) { if (!r.ha) { r.a = atan2_deg(r.vy, r.vx); r.ha = true; } return r.a; }
__device__ __forceinline__ double ray_norm(Ray &r) { if (!r.hn) { r.n = norm2(r.vx, r.vy); r.hn = true; } return r.n; }
__device__ __forceinline__ void ray_set_norm(Ray &r, double value) {  // utils.py:223-229 (value >= 0 on this path)
    (void)ray_angle(r);
    r.n = fabs(value); r.hn = true; r.hv = false;
}
__device__ __forceinline__ void ray_set_vector(Ray &r, double vx, double vy) { r.vx = vx; r.vy = vy; r.hv = true; r.hn = false; r.ha = false; }

// Obstacle.obstruct(ray, keep_tangential=True) (entities.py:158-184): clip the step by one circle.
__device__ __noinline__ void obstruct_tangential(Ray &ray, double cx, double cy, double rad) {
    const double relx = cx - ray.ox, rely = cy - ray.oy;
    const double rel_norm = norm2(relx, rely);
    const double norm = ray_norm(ray);
    if (norm == 0.0 || rel_norm < rad) {  // return -ray
        ray_materialize(ray);
        ray_set_vector(ray, -ray.vx, -ray.vy);
        return;
    }
    if (rel_norm >= norm + rad) return;
    ray_materialize(ray);
    const double inner = dot2(relx, rely, ray.vx, ray.vy);
    if (inner >= 0.0) {
        const double c0 = inner / (rel_norm * norm);
        const double cosv = c0 < 1.0 ? c0 : 1.0;
        const double perpendicular = rel_norm * sqrt(1.0 - cosv * cosv);
        if (rad > perpendicular) {
            const double half_chord = sqrt(rad * rad - perpendicular * perpendicular);
            const double cand = rel_norm * cosv - half_chord;
            const double new_norm = cand > 0.0 ? cand : 0.0;
            if (new_norm < norm) {
                const double oldx = ray.vx, oldy = ray.vy;
                ray_set_norm(ray, new_norm);
                ray_materialize(ray);
                const double rx = (ray.ox + ray.vx) - cx, ry = (ray.oy + ray.vy) - cy;
                const double s = (norm - new_norm) * half_chord / (rad * rad);
                ray_set_vector(ray, oldx + rx * s, oldy + ry * s);
            }
        }
    }
}

// Obstacle.obstruct(ray) without tangential part on a ray that stays polar (angle fixed, norm
// shrinks): the occlusion-table builder (entities.py:450-455).  (cs, sn) = cos/sin of the angle.
__device__ __forceinline__ double clip_polar(double norm, double cs, double sn, double relx, double rely, double rel_norm, double rad) {
    if (norm == 0.0 || rel_norm < rad) return norm;   // degenerate (camera inside/touching): left unchanged
    if (rel_norm >= norm + rad) return norm;
    const double vx = norm * cs, vy = norm * sn;
    const double inner = dot2(relx, rely, vx, vy);
    if (inner >= 0.0) {
        const double c0 = inner / (rel_norm * norm);
        const double cosv = c0 < 1.0 ? c0 : 1.0;
        const double perpendicular = rel_norm * sqrt(1.0 - cosv * cosv);
        if (rad > perpendicular) {
            const double half_chord = sqrt(rad * rad - perpendicular * perpendicular);
            const double cand = rel_norm * cosv - half_chord;
            const double new_norm = cand > 0.0 ? cand : 0.0;
            if (new_norm < norm) return new_norm;
        }
    }
    return norm;
}

}  // namespace mate
