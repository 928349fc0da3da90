// Bit-exactness census of the lean f64 sqrt / divide against the compiler's IEEE expansions, on the GPU:
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/math_check.hip -o /tmp/math_check && /tmp/math_check
#include <hip/hip_runtime.h>
#include "../mate_amd/csrc/device_math.hpp"
#include <cstdio>
#include <cstdint>
#include <cmath>

using mate::sqrt_pos; using mate::div_nz; using mate::atan2_finite;
#define lean_sqrt sqrt_pos
#define lean_div div_nz
__device__ uint64_t splitmix(uint64_t &s) { uint64_t z = (s += 0x9E3779B97f4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }

__global__ void census(unsigned long long *bad, int iters) {
    uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x + 12345;
    unsigned long long bs = 0, bd = 0, ba = 0;
    for (int i = 0; i < iters; ++i) {
        const uint64_t u = splitmix(s), v = splitmix(s);
        // magnitudes 2^-40 .. 2^40 (everything the engine feeds: distances, squares of distances, angles, scales)
        const double x = ldexp((double)(u >> 11) * (1.0 / 9007199254740992.0) + 1.0, (int)(u & 63) - 32) ;
        const double y = ldexp((double)(v >> 11) * (1.0 / 9007199254740992.0) + 1.0, (int)(v & 63) - 32) * ((v >> 10) & 1 ? -1.0 : 1.0);
        if (__double_as_longlong(lean_sqrt(x)) != __double_as_longlong(sqrt(x))) ++bs;
        if (__double_as_longlong(lean_div(x, y)) != __double_as_longlong(x / y)) ++bd;
        if (__double_as_longlong(lean_div(y, x)) != __double_as_longlong(y / x)) ++bd;
        const double xs = (u >> 9) & 1 ? -x : x;
        if (__double_as_longlong(atan2_finite(y, xs)) != __double_as_longlong(atan2(y, xs))) ++ba;
        if (__double_as_longlong(atan2_finite(xs, y)) != __double_as_longlong(atan2(xs, y))) ++ba;
    }
    if (lean_sqrt(0.0) != 0.0) ++bs;
    if (atan2_finite(0.0, 3.0) != 0.0 || atan2_finite(0.0, -3.0) != atan2(0.0, -3.0) || atan2_finite(2.0, 0.0) != atan2(2.0, 0.0) || atan2_finite(-2.0, 0.0) != atan2(-2.0, 0.0)) ++ba;
    atomicAdd(&bad[0], bs); atomicAdd(&bad[1], bd); atomicAdd(&bad[2], ba);
}

int main() {
    unsigned long long *d, h[3] = {0, 0, 0};
    (void)hipMalloc(&d, 24); (void)hipMemset(d, 0, 24);
    const int iters = 4096;
    hipLaunchKernelGGL(census, dim3(4096), dim3(256), 0, 0, d, iters);
    (void)hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
    printf("%.3g samples: sqrt_pos mismatches %llu, div_nz mismatches %llu (of 2x), atan2_finite mismatches %llu (of 2x)\n", 4096.0 * 256 * iters, h[0], h[1], h[2]);
    return (h[0] || h[1] || h[2]) ? 1 : 0;
}
