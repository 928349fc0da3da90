// The fused rollout's row stores in two layouts of the [steps][N][row] blocks: step-major (row r * N + env: what the kernels write
// today -- at any moment the 4096 waves write 4096 rows that lie side by side) and environment-major (row env * steps + r: every
// wave streams ONE contiguous 1 MB run for the whole launch).  Store-only launches, non-temporal 16-byte stores, one wave per
// environment, the waves skewed against each other as in the real kernel (a wave waits a pseudo-random few hundred cycles per step).
//   hipcc --offload-arch=gfx950 -O3 -o tools/store_layout tools/store_layout.hip ; tools/store_layout [allocations]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float vec4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
template <bool ENV_MAJOR>
__global__ __launch_bounds__(256, 4) void rows(vec4 *cam, vec4 *tgt, int N, int steps, int cam_chunks, int tgt_chunks, int work) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long env = (long)blockIdx.x * 4 + wave;
    if (env >= N) return;
    vec4 val = {1.f, 2.f, 3.f, (float)lane};
    unsigned h = (unsigned)env * 2654435761u;
    for (int r = 0; r < steps; ++r) {
        // stand-in for the step's arithmetic: `work` dependent FMAs (+-25 % by environment and step), so that the waves drift apart
        h = h * 1664525u + 1013904223u;
        const int n = work + (int)((h >> 16) % (unsigned)(work / 2 + 1)) - work / 4;
        float a = val.x;
        for (int i = 0; i < n; ++i) a = __builtin_fmaf(a, 1.0000001f, 1e-7f);
        val.x = a;
        const long row = ENV_MAJOR ? env * steps + r : (long)r * N + env;
        vec4 *c = cam + row * cam_chunks, *t = tgt + row * tgt_chunks;
#pragma unroll
        for (int k = 0; k < 2; ++k) { const int i = lane + 64 * k; if (i < cam_chunks) __builtin_nontemporal_store(val, c + i); }
#pragma unroll
        for (int k = 0; k < 5; ++k) { const int i = lane + 64 * k; if (i < tgt_chunks) __builtin_nontemporal_store(val, t + i); }
    }
}
int main(int argc, char **argv) {
    const int N = 4096, steps = 256, trials = argc > 1 ? atoi(argv[1]) : 5;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const size_t cam_bytes = (size_t)steps * N * 126 * 16, tgt_bytes = (size_t)steps * N * 262 * 16;
    const double bytes = (double)(cam_bytes + tgt_bytes);
    printf("%-10s %28s %28s\n", "work/step", "step-major ms (GB/s)", "environment-major ms (GB/s)");
    for (int t = 0; t < trials; ++t) {
        vec4 *cam, *tgt; CHECK(hipMalloc(&cam, cam_bytes)); CHECK(hipMalloc(&tgt, tgt_bytes));
        for (int work : {0, 400, 1200}) {
            float best[2] = {1e30f, 1e30f};
            for (int rep = 0; rep < 3; ++rep)
                for (int m = 0; m < 2; ++m) {
                    CHECK(hipEventRecord(e0));
                    if (m) rows<true><<<N / 4, 256>>>(cam, tgt, N, steps, 126, 262, work); else rows<false><<<N / 4, 256>>>(cam, tgt, N, steps, 126, 262, work);
                    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                    if (rep && ms < best[m]) best[m] = ms;
                }
            printf("%-10d %16.3f (%6.0f) %20.3f (%6.0f)\n", work, best[0], bytes / best[0] / 1e6, best[1], bytes / best[1] / 1e6);
        }
        // (the blocks are NOT freed: the next trial's come from further into the device's memory)
    }
    return 0;
}
