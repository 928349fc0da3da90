// The cache-policy bits of the observation stores (gfx950: sc0, sc1, nt on global_store): pure-store rate of the fused rollout's
// access pattern for each combination.   hipcc --offload-arch=gfx950 -O3 -o tools/store_bits tools/store_bits.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float vec4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
template <int BITS> __device__ __forceinline__ void st(vec4 v, vec4 *p) {
    if (BITS == 0) asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(p), "v"(v) : "memory");
    if (BITS == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(p), "v"(v) : "memory");
    if (BITS == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0" :: "v"(p), "v"(v) : "memory");
    if (BITS == 3) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
    if (BITS == 4) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
    if (BITS == 5) asm volatile("global_store_dwordx4 %0, %1, off sc0 nt" :: "v"(p), "v"(v) : "memory");
    if (BITS == 6) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" :: "v"(p), "v"(v) : "memory");
    if (BITS == 7) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" :: "v"(p), "v"(v) : "memory");
}
template <int BITS>
__global__ __launch_bounds__(256, 4) void rows(vec4 *cam, vec4 *tgt, int N, int steps, int cam_chunks, int tgt_chunks) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long env = (long)blockIdx.x * 4 + wave;
    if (env >= N) return;
    const vec4 val = {1.f, 2.f, 3.f, (float)lane};
    for (int r = 0; r < steps; ++r) {
        const long row = (long)r * N + env;
        vec4 *c = cam + row * cam_chunks, *t = tgt + row * tgt_chunks;
#pragma unroll
        for (int k = 0; k < 2; ++k) { const int i = lane + 64 * k; if (i < cam_chunks) st<BITS>(val, c + i); }
#pragma unroll
        for (int k = 0; k < 5; ++k) { const int i = lane + 64 * k; if (i < tgt_chunks) st<BITS>(val, t + i); }
    }
}
template <int BITS> float run(vec4 *cam, vec4 *tgt, int N, int steps, hipEvent_t e0, hipEvent_t e1) {
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        CHECK(hipEventRecord(e0));
        rows<BITS><<<(N + 3) / 4, 256>>>(cam, tgt, N, steps, 126, 262);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < best) best = ms;
    }
    return best;
}
int main() {
    const int N = 4096, steps = 256;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const double bytes = (double)steps * N * 388 * 16;
    const char *names[8] = {"(none)", "nt", "sc0", "sc1", "sc0 sc1", "sc0 nt", "sc1 nt", "sc0 sc1 nt"};
    for (int s = 0; s < 3; ++s) {
        vec4 *cam, *tgt;
        CHECK(hipMalloc(&cam, (size_t)steps * N * 126 * 16)); CHECK(hipMalloc(&tgt, (size_t)steps * N * 262 * 16));
        float ms[8] = {run<0>(cam, tgt, N, steps, e0, e1), run<1>(cam, tgt, N, steps, e0, e1), run<2>(cam, tgt, N, steps, e0, e1), run<3>(cam, tgt, N, steps, e0, e1),
                       run<4>(cam, tgt, N, steps, e0, e1), run<5>(cam, tgt, N, steps, e0, e1), run<6>(cam, tgt, N, steps, e0, e1), run<7>(cam, tgt, N, steps, e0, e1)};
        printf("set %d:", s);
        for (int b = 0; b < 8; ++b) printf("  %s %.0f", names[b], bytes / ms[b] / 1e6);
        printf("  GB/s\n");
    }
    return 0;
}
