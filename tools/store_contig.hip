// Does the store rate of the rollout's access pattern depend on how the blocks are ALLOCATED?  Plain hipMalloc against
// hipExtMallocWithFlags(hipDeviceMallocContiguous) (physically contiguous) on the same box, several blocks of each.
//   hipcc --offload-arch=gfx950 -O3 -o tools/store_contig tools/store_contig.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float vec4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ __launch_bounds__(256, 4) void rows(vec4 *cam, vec4 *tgt, int N, int steps, int cam_chunks, int tgt_chunks) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long env = (long)blockIdx.x * 4 + wave;
    if (env >= N) return;
    const vec4 val = {1.f, 2.f, 3.f, (float)lane};
    for (int r = 0; r < steps; ++r) {
        const long row = (long)r * N + env;
        vec4 *c = cam + row * cam_chunks, *t = tgt + row * tgt_chunks;
#pragma unroll
        for (int k = 0; k < 2; ++k) { const int i = lane + 64 * k; if (i < cam_chunks) __builtin_nontemporal_store(val, c + i); }
#pragma unroll
        for (int k = 0; k < 5; ++k) { const int i = lane + 64 * k; if (i < tgt_chunks) __builtin_nontemporal_store(val, t + i); }
    }
}
int main(int argc, char **argv) {
    const int N = 4096, steps = 256, sets = argc > 1 ? atoi(argv[1]) : 4;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const size_t cam_bytes = (size_t)steps * N * 126 * 16, tgt_bytes = (size_t)steps * N * 262 * 16;
    const double bytes = (double)(cam_bytes + tgt_bytes);
    for (int mode = 0; mode < 3; ++mode) {
        for (int s = 0; s < sets; ++s) {
            vec4 *cam = nullptr, *tgt = nullptr;
            hipError_t a, b;
            if (mode == 1) { a = hipExtMallocWithFlags((void **)&cam, cam_bytes, hipDeviceMallocContiguous); b = hipExtMallocWithFlags((void **)&tgt, tgt_bytes, hipDeviceMallocContiguous); }
            else if (mode == 2) { a = hipMalloc(&cam, cam_bytes + tgt_bytes); b = a; tgt = (vec4 *)((char *)cam + cam_bytes); }
            else { a = hipMalloc(&cam, cam_bytes); b = hipMalloc(&tgt, tgt_bytes); }
            if (a != hipSuccess || b != hipSuccess) { printf("mode %d set %d: allocation failed (%s / %s)\n", mode, s, hipGetErrorString(a), hipGetErrorString(b)); (void)hipGetLastError(); continue; }
            float best = 1e30f;
            for (int rep = 0; rep < 4; ++rep) {
                CHECK(hipEventRecord(e0));
                rows<<<N / 4, 256>>>(cam, tgt, N, steps, 126, 262);
                CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (rep && ms < best) best = ms;
            }
            printf("%s set %d (cam %p): %.0f GB/s\n", mode == 0 ? "hipMalloc x2      " : mode == 1 ? "contiguous flag x2" : "hipMalloc, one    ", s, (void *)cam, bytes / best / 1e6);
        }
    }
    return 0;
}
