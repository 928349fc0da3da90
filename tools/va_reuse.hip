// Minimal repro of "a reused virtual range loses stores" (profiles/HISTORY.md 3.1b, mate_engine_block_free): a block of 2 MiB chunks is mapped,
// written, unmapped CHUNK BY CHUNK, its chunks released, hipDeviceSynchronize; then NEW chunks are mapped at the SAME virtual
// address -- (a) the range kept reserved, (b) the range freed (hipMemAddressFree) and reserved again -- a kernel writes a new pattern
// and the block is read back three ways: by a kernel, by hipMemcpy to the host, by a device-to-device copy into hipMalloc memory.
//   hipcc --offload-arch=gfx950 -O3 -o tools/va_reuse tools/va_reuse.hip ; tools/va_reuse [chunks] [rounds]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr size_t kChunk = (size_t)2 << 20;
__global__ void fill(unsigned *p, size_t n, unsigned tag) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = tag ^ (unsigned)i; }
__global__ void check(const unsigned *p, size_t n, unsigned tag, unsigned long long *bad) {
    unsigned long long mine = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) mine += p[i] != (tag ^ (unsigned)i);
    if (mine) atomicAdd(bad, mine);
}
static hipMemAllocationProp prop;
static std::vector<hipMemGenericAllocationHandle_t> map_new(void *va, size_t n) {
    std::vector<hipMemGenericAllocationHandle_t> h(n);
    for (size_t i = 0; i < n; ++i) { CHECK(hipMemCreate(&h[i], kChunk, &prop, 0)); CHECK(hipMemMap((char *)va + i * kChunk, kChunk, 0, h[i], 0)); }
    hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    CHECK(hipMemSetAccess(va, n * kChunk, &acc, 1));
    return h;
}
static void unmap_all(void *va, std::vector<hipMemGenericAllocationHandle_t> &h) {
    CHECK(hipDeviceSynchronize());
    for (size_t i = 0; i < h.size(); ++i) CHECK(hipMemUnmap((char *)va + i * kChunk, kChunk));
    for (auto x : h) CHECK(hipMemRelease(x));
    h.clear();
    CHECK(hipDeviceSynchronize());
}
int main(int argc, char **argv) {
    const size_t n = argc > 1 ? (size_t)atol(argv[1]) : 64;
    const int rounds = argc > 2 ? atoi(argv[2]) : 12;
    prop = {}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    const size_t words = n * kChunk / 4;
    unsigned long long *bad; CHECK(hipMalloc(&bad, 8));
    unsigned *mirror; CHECK(hipMalloc(&mirror, n * kChunk));
    std::vector<unsigned> host(words);
    for (int mode = 0; mode < 2; ++mode) {
        printf("%s\n", mode == 0 ? "(a) range kept reserved, new chunks mapped at it" : "(b) range freed and reserved again");
        void *va = nullptr;
        CHECK(hipMemAddressReserve(&va, n * kChunk, kChunk, nullptr, 0));
        int same_va = 0;
        unsigned long long lost_kernel = 0, lost_host = 0, lost_d2d = 0;
        for (int r = 0; r < rounds; ++r) {
            auto h = map_new(va, n);
            const unsigned tag = 0x9e3779b9u * (unsigned)(r + 1 + 100 * mode);
            fill<<<1024, 256>>>((unsigned *)va, words, tag);
            CHECK(hipMemset(bad, 0, 8));
            check<<<1024, 256>>>((const unsigned *)va, words, tag, bad);
            unsigned long long b = 0; CHECK(hipMemcpy(&b, bad, 8, hipMemcpyDeviceToHost)); lost_kernel += b;
            CHECK(hipMemcpy(host.data(), va, n * kChunk, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < words; ++i) lost_host += host[i] != (tag ^ (unsigned)i);
            CHECK(hipMemcpy(mirror, va, n * kChunk, hipMemcpyDeviceToDevice));
            CHECK(hipMemset(bad, 0, 8));
            check<<<1024, 256>>>(mirror, words, tag, bad);
            CHECK(hipMemcpy(&b, bad, 8, hipMemcpyDeviceToHost)); lost_d2d += b;
            unmap_all(va, h);
            if (mode == 1) {
                void *old = va;
                CHECK(hipMemAddressFree(va, n * kChunk));
                CHECK(hipMemAddressReserve(&va, n * kChunk, kChunk, nullptr, 0));
                same_va += va == old;
            }
        }
        printf("    %d rounds of %zu chunks: words lost as seen by a kernel %llu, by hipMemcpy to the host %llu, by a device-to-device copy %llu", rounds, n, lost_kernel, lost_host, lost_d2d);
        if (mode == 1) printf("; the new reservation was the old address %d times", same_va);
        printf("\n");
        CHECK(hipMemAddressFree(va, n * kChunk));
    }
    return 0;
}
