// The rollout's store pattern on blocks built from separately created physical chunks (hipMemCreate / hipMemMap), mapped in
// order or in a shuffled order, for several chunk sizes -- against plain hipMalloc on the same box.
//   hipcc --offload-arch=gfx950 -O3 -o tools/store_vmm tools/store_vmm.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
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
static char *vmm_block(size_t bytes, size_t chunk, bool shuffle, unsigned seed) {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    size_t gran = 0;
    CHECK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum));
    if (chunk < gran) chunk = gran;
    chunk = (chunk + gran - 1) / gran * gran;
    const size_t n = (bytes + chunk - 1) / chunk;
    void *va = nullptr;
    CHECK(hipMemAddressReserve(&va, n * chunk, 0, nullptr, 0));
    std::vector<size_t> order(n);
    for (size_t i = 0; i < n; ++i) order[i] = i;
    std::vector<hipMemGenericAllocationHandle_t> h(n);
    for (size_t i = 0; i < n; ++i) CHECK(hipMemCreate(&h[i], chunk, &prop, 0));      // created in order ...
    if (shuffle) { std::mt19937 rng(seed); std::shuffle(order.begin(), order.end(), rng); }
    for (size_t i = 0; i < n; ++i) CHECK(hipMemMap((char *)va + order[i] * chunk, chunk, 0, h[i], 0));      // ... mapped in a permuted one
    hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    CHECK(hipMemSetAccess(va, n * chunk, &acc, 1));
    return (char *)va;
}
int main(int argc, char **argv) {
    const int N = 4096, steps = 256;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const size_t cam_bytes = (size_t)steps * N * 126 * 16, tgt_bytes = (size_t)steps * N * 262 * 16;
    const double bytes = (double)(cam_bytes + tgt_bytes);
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    size_t gmin = 0, grec = 0;
    CHECK(hipMemGetAllocationGranularity(&gmin, &prop, hipMemAllocationGranularityMinimum));
    CHECK(hipMemGetAllocationGranularity(&grec, &prop, hipMemAllocationGranularityRecommended));
    printf("allocation granularity: minimum %zu, recommended %zu bytes\n", gmin, grec);
    auto measure = [&](const char *what, vec4 *cam, vec4 *tgt) {
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            CHECK(hipEventRecord(e0));
            rows<<<N / 4, 256>>>(cam, tgt, N, steps, 126, 262);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep && ms < best) best = ms;
        }
        printf("%-44s %.0f GB/s\n", what, bytes / best / 1e6);
    };
    const int trials = argc > 1 ? atoi(argv[1]) : 8;
    const size_t chunk = (size_t)(argc > 2 ? atoi(argv[2]) : 2) << 20;
    for (int t = 0; t < trials; ++t) {
        vec4 *cam, *tgt; CHECK(hipMalloc(&cam, cam_bytes)); CHECK(hipMalloc(&tgt, tgt_bytes)); measure("hipMalloc x2", cam, tgt);
        char *blk = vmm_block(cam_bytes + tgt_bytes, chunk, true, 1000u + t);
        char what[96];
        snprintf(what, sizeof what, "chunks of %zu MiB, mapped shuffled", chunk >> 20);
        measure(what, (vec4 *)blk, (vec4 *)(blk + cam_bytes));
    }
    return 0;
}
