// What launching more, smaller workgroups costs: kernels that do nothing but hold their wave for a fixed number of cycles, for the
// launch shapes of the per-step kernels at 4096 environments -- (1024 x 256) one wave per environment, four per workgroup;
// (4096 x 128) two waves per environment; (2048 x 256) two waves per environment, two environments per workgroup; ...
//   hipcc --offload-arch=gfx950 -O3 -o tools/dispatch_probe tools/dispatch_probe.hip ; tools/dispatch_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void hold(long long cycles, int barriers, float *sink) {
    extern __shared__ float lds[];
    const long long t0 = __builtin_amdgcn_s_memtime();
    lds[threadIdx.x] = (float)threadIdx.x;
    for (int b = 0; b < barriers; ++b) __syncthreads();
    while (__builtin_amdgcn_s_memtime() - t0 < cycles) __builtin_amdgcn_s_sleep(1);
    if (sink && lds[threadIdx.x ^ 1] < 0.f) sink[0] = 1.f;
}
int main() {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    struct Shape { int grid, block, lds; } shapes[] = {{1024, 256, 18432}, {2048, 256, 9216}, {4096, 128, 4608}, {4096, 64, 4608}, {8192, 64, 2304}, {2048, 128, 4608}, {512, 256, 18432}};
    printf("%-14s %10s %10s %10s %10s   (us per launch, dispatch events, best of 20)\n", "grid x block", "hold 0", "hold 10k", "hold 20k", "20k+2 bar");
    for (auto s : shapes) {
        char name[32]; snprintf(name, sizeof name, "%d x %d", s.grid, s.block);
        printf("%-14s", name);
        for (int v = 0; v < 4; ++v) {
            const long long cyc = v == 0 ? 0 : v == 1 ? 10000 : 20000;
            float best = 1e30f;
            for (int rep = 0; rep < 24; ++rep) {
                hipExtLaunchKernelGGL(hold, dim3(s.grid), dim3(s.block), s.lds, 0, e0, e1, 0, cyc, v == 3 ? 2 : 0, (float *)nullptr);
                CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (rep >= 4 && ms < best) best = ms;
            }
            printf(" %10.2f", best * 1e3f);
        }
        printf("\n");
    }
    return 0;
}
