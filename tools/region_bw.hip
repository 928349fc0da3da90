// How much of HBM's bandwidth does ONE small region get?  All CUs store (non-temporal) into a region of 2 MiB .. 1 GiB again and
// again: if the rate grows with the region, the interleave across stacks / channels is coarser than the small regions.
//   hipcc --offload-arch=gfx950 -O3 -o tools/region_bw tools/region_bw.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float vec4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ __launch_bounds__(256) void hammer(vec4 *p, size_t chunks, int passes) {
    const vec4 v = {1.f, 2.f, 3.f, 4.f};
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (int k = 0; k < passes; ++k)
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < chunks; i += stride) __builtin_nontemporal_store(v, p + i);
}
int main() {
    char *base; const size_t total = (size_t)8 << 30;
    CHECK(hipMalloc(&base, total));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (size_t region : {(size_t)2 << 20, (size_t)8 << 20, (size_t)32 << 20, (size_t)128 << 20, (size_t)512 << 20, (size_t)2048 << 20}) {
        printf("region %5zu MiB:", region >> 20);
        for (int where = 0; where < 6; ++where) {
            char *p = base + (size_t)where * ((total - region) / 5 / (2 << 20)) * (2 << 20);
            const int passes = (int)(((size_t)4 << 30) / region);
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                CHECK(hipEventRecord(e0));
                hammer<<<2048, 256>>>((vec4 *)p, region / 16, passes);
                CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (rep && ms < best) best = ms;
            }
            printf(" %5.0f", (double)region * passes / best / 1e6);
        }
        printf("  GB/s (six places in an 8 GiB allocation)\n");
    }
    return 0;
}
