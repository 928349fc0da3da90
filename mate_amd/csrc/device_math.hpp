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
// IEEE square root and quotient for the magnitudes this engine feeds them (0 or 2^-700 < |x| < 2^700, divisor != 0):
// the compiler's own expansions (v_rsq / v_rcp + Newton steps + residual correction) without the range scaling and
// special-operand fix-ups around them -- 13 and 8 instructions instead of 18 and 11.  Bit-identical to sqrt() and `/`
// on 4.3e9 random operands spanning 2^-32 .. 2^32 (tools/math_check.hip).  The kernels are VALU-issue bound at
// large batches, so instruction count is throughput.
__device__ __forceinline__ double sqrt_pos(double x) {
    const double r = __builtin_amdgcn_rsq(x);
    double g = x * r, h = r * 0.5;
    const double e = fma(-h, g, 0.5);
    g = fma(g, e, g); h = fma(h, e, h);
    double d = fma(-g, g, x);
    g = fma(d, h, g);
    d = fma(-g, g, x);
    g = fma(d, h, g);
    return x == 0.0 ? x : g;
}
__device__ __forceinline__ double div_nz(double a, double b) {
    double r = __builtin_amdgcn_rcp(b);
    r = fma(r, fma(-b, r, 1.0), r);
    r = fma(r, fma(-b, r, 1.0), r);
    const double q = a * r;
    return fma(fma(-b, q, a), r, q);
}
// The hardware's f32 root as it is (one unit in the last place; the argument is a normal number -- a squared sight range): the
// correctly rounded sqrtf expands to fourteen instructions, for a value that is multiplied by a sine good to 3e-7 next.
__device__ __forceinline__ float sqrt_f32_1ulp(float x) { return __builtin_amdgcn_sqrtf(x); }
// a / n for a count n that a shape-specialised kernel knows at compile time: a power of two (eight targets) is an exact multiply
__device__ __forceinline__ double div_by_count(double a, int n) {
    if (__builtin_constant_p(n) && n > 0 && (n & (n - 1)) == 0) return a * (1.0 / (double)n);
    return div_nz(a, (double)n);
}
__device__ __forceinline__ double norm2(double x, double y) { return sqrt_pos(fma(y, y, x * x)); }
// np.clip on finite operands with lo <= hi, neither bound a zero (action limits, viewing-angle limits, the terrain): two
// v_max / v_min instead of two compares and four selects.
__device__ __forceinline__ double clipd(double v, double lo, double hi) { return __builtin_fmin(__builtin_fmax(v, lo), hi); }
// The same with WAVE-UNIFORM bounds (scenario constants, literals) as two instructions: fmax / fmin above cost five, because
// the compiler first canonicalises every operand (v_max_f64 x, x) in case it is a signalling NaN.  The bounds travel in
// scalar registers (one per instruction: the constant-bus limit), the value is finite by construction.
__device__ __forceinline__ double clip_uniform(double v, double lo, double hi) {
    double r;
    asm("v_max_f64 %0, %1, %2\n\tv_min_f64 %0, %0, %3" : "=&v"(r) : "v"(v), "s"(lo), "s"(hi));
    return r;
}

// Python float `%` with a positive divisor (utils.py:158 uses it with 360.0).
__device__ __forceinline__ double pymod_pos(double a, double b) {
    const double q = trunc(a / b);
    double r = fma(-q, b, a);          // exact when q is the true truncated quotient
    if (a >= 0.0) { if (r < 0.0) r += b; else if (r >= b) r -= b; }
    else          { if (r > 0.0) r -= b; else if (r <= -b) r += b; }
    if (r < 0.0) r += b;               // sign follows the divisor
    return r == 0.0 ? 0.0 : r;
}
__device__ __forceinline__ double normalize_angle(double a) {   // utils.py:155-158: (a + 180) % 360 - 180
    const double x = a + 180.0;
    double r;
    if (x >= 0.0 && x < 360.0) r = x;                 // fmod is the identity here
    else if (x >= 360.0 && x < 720.0) r = x - 360.0;  // exact (Sterbenz)
    else if (x < 0.0 && x >= -360.0) r = x + 360.0;   // Python: fmod keeps x, then adds the divisor once
    else r = pymod_pos(x, 360.0);
    return r - 180.0;
}
// p * z + C for an f64 literal C.  A 64-bit literal cannot be encoded in a VALU instruction and the compiler parks
// every polynomial coefficient in a VGPR pair (two v_mov per coefficient, and again in every loop iteration with
// MachineLICM off); here the literal goes through an SGPR pair written by two scalar moves, which issue beside the
// vector instructions of other waves.  Same fma, same bits.
template <unsigned long long BITS>
__device__ __forceinline__ double fma_sc(double p, double z) {
    unsigned long long bits = BITS;
    asm("" : "+s"(bits));                          // the literal, opaque, in a scalar register pair
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(p), "v"(z), "s"(bits));
    return r;
}
#define FMA_SC(p, z, c) fma_sc<__builtin_bit_cast(unsigned long long, (double)(c))>((p), (z))
#define FMA_SCX(p, z, hex) fma_sc<(hex)>((p), (z))

// atan2 for finite operands that are not both zero: the device library's algorithm (t = min/max of the magnitudes,
// a degree-19 minimax polynomial in t^2 with the library's own coefficients, octant and quadrant unfolding), without
// its infinity / NaN / signed-zero cases and with the coefficients routed through SGPRs -- 45 vector instructions
// instead of 105, bit-identical to atan2() on finite non-zero operands (tools/math_check.hip).
__device__ __forceinline__ double atan2_finite(double y, double x) {
    const double ax = fabs(x), ay = fabs(y);
    const double mx = ax > ay ? ax : ay, mn = ax > ay ? ay : ax;
    const double t = div_nz(mn, mx);
    const double z = t * t;
    double p = __longlong_as_double(0x3eeba404b5e68a13ll);
    p = FMA_SCX(p, z, 0xbf23e260bd3237f4ull); p = FMA_SCX(p, z, 0x3f4b2bb069efb384ull); p = FMA_SCX(p, z, 0xbf67952daf56de9bull);
    p = FMA_SCX(p, z, 0x3f7d6d43a595c56full); p = FMA_SCX(p, z, 0xbf8c6ea4a57d9582ull); p = FMA_SCX(p, z, 0x3f967e295f08b19full);
    p = FMA_SCX(p, z, 0xbf9e9ae6fc27006aull); p = FMA_SCX(p, z, 0x3fa2c15b5711927aull); p = FMA_SCX(p, z, 0xbfa59976e82d3ff0ull);
    p = FMA_SCX(p, z, 0x3fa82d5d6ef28734ull); p = FMA_SCX(p, z, 0xbfaae5ce6a214619ull); p = FMA_SCX(p, z, 0x3fae1bb48427b883ull);
    p = FMA_SCX(p, z, 0xbfb110e48b207f05ull); p = FMA_SCX(p, z, 0x3fb3b13657b87036ull); p = FMA_SCX(p, z, 0xbfb745d119378e4full);
    p = FMA_SCX(p, z, 0x3fbc71c717e1913cull); p = FMA_SCX(p, z, 0xbfc2492492376b7dull); p = FMA_SCX(p, z, 0x3fc99999999952ccull);
    p = FMA_SCX(p, z, 0xbfd5555555555523ull);
    double r = fma(t, z * p, t);
    const double half_pi = __longlong_as_double(0x3ff921fb54442d18ll), pi = __longlong_as_double(0x400921fb54442d18ll);
    if (ay > ax) r = half_pi - r;
    if (x < 0.0) r = pi - r;
    if (y == 0.0) r = x < 0.0 ? pi : 0.0;
    return copysign(r, y);
}
__device__ __forceinline__ double atan2_deg(double y, double x) { return atan2_finite(y, x) * kRad2Deg; }           // utils.py:124-131
// The same for Camera.perceive's relative vectors (never both components zero: cameras and targets are circles that do not
// overlap): the octant / quadrant unfolding as ONE addition base + (+-r), base in {0, pi/2, pi} -- 10 vector instructions
// instead of 16.  Identical bits to atan2_finite except in the second octant pair (|y| > |x|, x < 0), where pi/2 + r is rounded
// once instead of pi - (pi/2 - r) twice (at most one unit in the last place, towards the true value).
__device__ __forceinline__ double atan2_sector(double y, double x) {
    const double ax = fabs(x), ay = fabs(y);
    const bool steep = ay > ax, back = x < 0.0;
    const double t = div_nz(steep ? ax : ay, steep ? ay : ax);
    const double z = t * t;
    double p = __longlong_as_double(0x3eeba404b5e68a13ll);
    p = FMA_SCX(p, z, 0xbf23e260bd3237f4ull); p = FMA_SCX(p, z, 0x3f4b2bb069efb384ull); p = FMA_SCX(p, z, 0xbf67952daf56de9bull);
    p = FMA_SCX(p, z, 0x3f7d6d43a595c56full); p = FMA_SCX(p, z, 0xbf8c6ea4a57d9582ull); p = FMA_SCX(p, z, 0x3f967e295f08b19full);
    p = FMA_SCX(p, z, 0xbf9e9ae6fc27006aull); p = FMA_SCX(p, z, 0x3fa2c15b5711927aull); p = FMA_SCX(p, z, 0xbfa59976e82d3ff0ull);
    p = FMA_SCX(p, z, 0x3fa82d5d6ef28734ull); p = FMA_SCX(p, z, 0xbfaae5ce6a214619ull); p = FMA_SCX(p, z, 0x3fae1bb48427b883ull);
    p = FMA_SCX(p, z, 0xbfb110e48b207f05ull); p = FMA_SCX(p, z, 0x3fb3b13657b87036ull); p = FMA_SCX(p, z, 0xbfb745d119378e4full);
    p = FMA_SCX(p, z, 0x3fbc71c717e1913cull); p = FMA_SCX(p, z, 0xbfc2492492376b7dull); p = FMA_SCX(p, z, 0x3fc99999999952ccull);
    p = FMA_SCX(p, z, 0xbfd5555555555523ull);
    const double r = fma(t, z * p, t);
    // base: the high words of 0, pi/2, pi differ, the low word is pi's or zero; the sign of r flips when exactly one of the
    // two reflections applies
    const uint32_t base_hi = steep ? 0x3ff921fbu : (back ? 0x400921fbu : 0u);
    const uint32_t base_lo = (steep || back) ? 0x54442d18u : 0u;
    const double base = __hiloint2double((int)base_hi, (int)base_lo);
    const uint32_t flip = (steep != back) ? 0x80000000u : 0u;
    const double sr = __hiloint2double(__double2hiint(r) ^ (int)flip, __double2loint(r));
    return copysign(base + sr, y);
}

// sin and cos of an angle given in DEGREES, |deg| <= 720.  polar2cartesian (utils.py:144-152) converts
// with phi * (pi/180) first, and so does this.  Two-term Cody-Waite reduction to [-pi/4, pi/4] plus
// Taylor polynomials evaluated with fma: ~1 ulp, a third of the instructions of the general-range
// libm sincos (which carries a Payne-Hanek path these bounded angles never need).
__device__ __forceinline__ void sincos_deg(double deg, double &sn, double &cs) {
    const double x = deg * kDeg2Rad;
    const double k = rint(x * 0.63661977236758134308);            // 2/pi
    double r = fma(-k, 1.57079632679489655800e+00, x);            // pi/2 high
    r = fma(-k, 6.12323399573676603587e-17, r);                   // pi/2 low
    const double z = r * r;
    double ps = 2.81145725434552075980e-15;                       // 1/17!
    ps = FMA_SC(ps, z, -7.64716373181981647590e-13);              // -1/15!
    ps = FMA_SC(ps, z, 1.60590438368216145994e-10);               // 1/13!
    ps = FMA_SC(ps, z, -2.50521083854417187751e-08);              // -1/11!
    ps = FMA_SC(ps, z, 2.75573192239858906526e-06);               // 1/9!
    ps = FMA_SC(ps, z, -1.98412698412698412698e-04);              // -1/7!
    ps = FMA_SC(ps, z, 8.33333333333333333333e-03);               // 1/5!
    ps = FMA_SC(ps, z, -1.66666666666666666667e-01);              // -1/3!
    const double s0 = fma(r * z, ps, r);
    double pc = -1.56192069685862264622e-16;                      // -1/18!
    pc = FMA_SC(pc, z, 4.77947733238738529744e-14);               // 1/16!
    pc = FMA_SC(pc, z, -1.14707455977297247139e-11);              // -1/14!
    pc = FMA_SC(pc, z, 2.08767569878680989792e-09);               // 1/12!
    pc = FMA_SC(pc, z, -2.75573192239858906526e-07);              // -1/10!
    pc = FMA_SC(pc, z, 2.48015873015873015873e-05);               // 1/8!
    pc = FMA_SC(pc, z, -1.38888888888888888889e-03);              // -1/6!
    pc = FMA_SC(pc, z, 4.16666666666666666667e-02);               // 1/4!
    const double c0 = fma(z * z, pc, fma(-0.5, z, 1.0));
    const int q = (int)k & 3;
    const double sa = (q & 1) ? c0 : s0, ca = (q & 1) ? s0 : c0;
    sn = (q & 2) ? -sa : sa;
    cs = ((q + 1) & 2) ? -ca : ca;
}

// sin and cos of an angle given in degrees (|deg| <= 720) to f32: the Cody-Waite reduction of sincos_deg in f64 (the
// argument's ABSOLUTE accuracy matters: a component near zero must keep its relative precision), then Taylor
// polynomials of the reduced argument in f32.  Relative error of either result < 3e-7.
__device__ __forceinline__ void sincos_deg_f32(double deg, float &sn, float &cs) {
    const double x = deg * kDeg2Rad;
    const double k = rint(x * 0.63661977236758134308);
    double rd = fma(-k, 1.57079632679489655800e+00, x);
    rd = fma(-k, 6.12323399573676603587e-17, rd);
    const float r = (float)rd;
    const float z = r * r;
    float ps = 2.75573192239858906526e-06f;                       // 1/9!
    ps = fmaf(ps, z, -1.98412698412698412698e-04f); ps = fmaf(ps, z, 8.33333333333333333333e-03f); ps = fmaf(ps, z, -1.66666666666666666667e-01f);
    const float s0 = fmaf(r * z, ps, r);
    float pc = 2.48015873015873015873e-05f;                       // 1/8!
    pc = fmaf(pc, z, -1.38888888888888888889e-03f); pc = fmaf(pc, z, 4.16666666666666666667e-02f); pc = fmaf(pc, z, -0.5f);
    const float c0 = fmaf(z, pc, 1.0f);
    const int q = (int)k & 3;
    const float sa = (q & 1) ? c0 : s0, ca = (q & 1) ? s0 : c0;
    sn = (q & 2) ? -sa : sa;
    cs = ((q + 1) & 2) ? -ca : ca;
}

// Obstacle.obstruct(ray, keep_tangential=True) (entities.py:158-184) on a step kept in cartesian form.
// The reference shortens the ray through its polar form (new_norm * (cos a, sin a), a = atan2(v)); that
// is v * (new_norm / |v|) up to last-place rounding, which is how it is done here (no atan2/sincos).
// (vx, vy) step vector from origin (ox, oy); n = |v| (recomputed when n_known is false).
// `Vector2D.norm = value` (utils.py:223-229) the reference's way: the vector is re-made from its polar form, value * (cos, sin) of
// atan2_deg(v) (utils.py:144-152).  The kernels rescale the vector instead (below); -DMATE_POLAR_CLAMP (python -m mate_amd.build
// --variant polar -DMATE_POLAR_CLAMP) builds them with THIS on the two places the hot path sets a norm -- the over-long step and the
// ray truncated by an obstacle -- for the census that weighs the departure (profiles/r05_polar_census.txt).
__device__ __forceinline__ void set_norm_polar(double &vx, double &vy, double value) {
    double sn, cs;
    sincos_deg(atan2_deg(vy, vx), sn, cs);
    vx = value * cs; vy = value * sn;
}

__device__ __forceinline__ void obstruct_tangential(double ox, double oy, double &vx, double &vy, double &n, bool &n_known,
                                                    double cx, double cy, double rad) {
    const double relx = cx - ox, rely = cy - oy;
    const double rel_norm = norm2(relx, rely);
    if (!n_known) { n = norm2(vx, vy); n_known = true; }
    if (n == 0.0 || rel_norm < rad) { vx = -vx; vy = -vy; return; }   // return -ray
    if (rel_norm >= n + rad) return;
    const double inner = dot2(relx, rely, vx, vy);
    if (inner >= 0.0) {
        const double c0 = div_nz(inner, rel_norm * n);
        const double cosv = c0 < 1.0 ? c0 : 1.0;
        const double perpendicular = rel_norm * sqrt_pos(1.0 - cosv * cosv);
        if (rad > perpendicular) {
            const double half_chord = sqrt_pos(rad * rad - perpendicular * perpendicular);
            const double cand = rel_norm * cosv - half_chord;
            const double new_norm = cand > 0.0 ? cand : 0.0;
            if (new_norm < n) {
#ifdef MATE_POLAR_CLAMP
                double tvx = vx, tvy = vy;
                set_norm_polar(tvx, tvy, new_norm);
                const double rx = (ox + tvx) - cx, ry = (oy + tvy) - cy;
#else
                const double scale = div_nz(new_norm, n);
                const double rx = (ox + vx * scale) - cx, ry = (oy + vy * scale) - cy;
#endif
                const double s = div_nz((n - new_norm) * half_chord, rad * rad);
                vx = vx + rx * s; vy = vy + ry * s;
                n_known = false;
            }
        }
    }
}

// Obstacle.obstruct(ray) without tangential part on a ray that stays polar (angle fixed, norm
// shrinks): the occlusion-table builder (entities.py:450-455).  (cs, sn) = cos/sin of the angle.
__device__ __forceinline__ double clip_polar(double norm, double cs, double sn, double relx, double rely, double rel_norm, double rad, bool outer = false) {
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
            const double cand = outer ? rel_norm * cosv + half_chord : rel_norm * cosv - half_chord;   // far / near crossing, entities.py:172-175
            const double new_norm = cand > 0.0 ? cand : 0.0;
            if (new_norm < norm) return new_norm;
        }
    }
    return norm;
}

}  // namespace mate
