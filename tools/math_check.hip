// Bit-exactness census of the lean f64 sqrt / divide against the compiler's IEEE expansions, on the GPU:
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/math_check.hip -o /tmp/math_check && /tmp/math_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>

__device__ __forceinline__ double lean_sqrt(double x) {      // the expansion's core without the range scaling: x = 0 or 2^-700 < x < 2^700
    const double r = __builtin_amdgcn_rsq(x);
    double g = x * r, h = r * 0.5;
    double e = fma(-h, g, 0.5);
    g = fma(g, e, g); h = fma(h, e, h);
    double d = fma(-g, g, x);
    g = fma(d, h, g);
    d = fma(-g, g, x);
    g = fma(d, h, g);
    return x == 0.0 ? x : g;
}
__device__ __forceinline__ double lean_div(double a, double b) {   // normal-range operands, b != 0
    double r = __builtin_amdgcn_rcp(b);
    r = fma(r, fma(-b, r, 1.0), r);
    r = fma(r, fma(-b, r, 1.0), r);
    const double q = a * r;
    return fma(fma(-b, q, a), r, q);
}
__device__ uint64_t splitmix(uint64_t &s) { uint64_t z = (s += 0x9E3779B97f4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }

__global__ void census(unsigned long long *bad, int iters) {
    uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x + 12345;
    unsigned long long bs = 0, bd = 0;
    for (int i = 0; i < iters; ++i) {
        const uint64_t u = splitmix(s), v = splitmix(s);
        // magnitudes 2^-40 .. 2^40 (everything the engine feeds: distances, squares of distances, angles, scales)
        const double x = ldexp((double)(u >> 11) * (1.0 / 9007199254740992.0) + 1.0, (int)(u & 63) - 32) ;
        const double y = ldexp((double)(v >> 11) * (1.0 / 9007199254740992.0) + 1.0, (int)(v & 63) - 32) * ((v >> 10) & 1 ? -1.0 : 1.0);
        if (__double_as_longlong(lean_sqrt(x)) != __double_as_longlong(sqrt(x))) ++bs;
        if (__double_as_longlong(lean_div(x, y)) != __double_as_longlong(x / y)) ++bd;
        if (__double_as_longlong(lean_div(y, x)) != __double_as_longlong(y / x)) ++bd;
    }
    if (lean_sqrt(0.0) != 0.0) ++bs;
    atomicAdd(&bad[0], bs); atomicAdd(&bad[1], bd);
}

int main() {
    unsigned long long *d, h[2] = {0, 0};
    hipMalloc(&d, 16); hipMemset(d, 0, 16);
    const int iters = 4096;
    hipLaunchKernelGGL(census, dim3(4096), dim3(256), 0, 0, d, iters);
    hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("%.3g samples: lean_sqrt mismatches %llu, lean_div mismatches %llu\n", 4096.0 * 256 * iters, h[0], h[1]);
    return (h[0] || h[1]) ? 1 : 0;
}
