// Which PHYSICAL chunks may be written side by side?  (profiles/HISTORY.md 3.1b, "the stores": the same block allocated twice takes the fused
// rollout's row stores at 4.3 or 5.5 TB/s; physically contiguous memory is the slowest, shuffled 2 MiB chunks mostly fast.)
//
// A pool of 2 MiB physical chunks is created in allocation order (a proxy for physical order: the driver hands chunks out in
// address order, more or less).  The fused rollout's target-row pattern -- 4096 waves, wave `env` streaming the 4192 bytes of row
// r * 4096 + env, step after step -- is then run on virtual ranges mapped from CHOSEN chunks:
//   windows: a range of K steps' rows (K = 2: 17 chunks), written over and over, mapped from chunks base + i * stride of the pool,
//            for strides 1, 2, 4, ... and several bases, and from random chunks: ONE bit-field of the chunk index varies at a time.
//            What it says: which chunk-index bits must differ among the ~9-17 chunks a step writes concurrently.
//   blocks:  a whole 256-step block (2097 chunks) from one pool segment, mapped by identity, shuffled, and by the stride
//            permutations i -> (i * s) mod n -- a CONSTRUCTION (no probing, no candidates) if some stride is reliably fast.
//   hipcc --offload-arch=gfx950 -O3 -o tools/chunk_order tools/chunk_order.hip ; tools/chunk_order [pool chunks] [steps per window]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
typedef float vec4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr size_t kChunk = (size_t)2 << 20;

__global__ __launch_bounds__(256, 4) void rows(vec4 *blk, int N, int steps, int wrap, int row_chunks) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long env = (long)blockIdx.x * 4 + wave;
    if (env >= N) return;
    const vec4 val = {1.f, 2.f, 3.f, (float)lane};
    for (int r = 0; r < steps; ++r) {
        vec4 *t = blk + ((long)(r % wrap) * N + env) * row_chunks;
#pragma unroll
        for (int k = 0; k < 5; ++k) { const int i = lane + 64 * k; if (i < row_chunks) __builtin_nontemporal_store(val, t + i); }
    }
}
__global__ void fill(vec4 *blk, long n) {
    const vec4 val = {0.f, 0.f, 0.f, 0.f};
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) __builtin_nontemporal_store(val, blk + i);
}

static hipMemAllocationProp g_prop;
static std::vector<hipMemGenericAllocationHandle_t> g_pool;
static hipEvent_t e0, e1;

struct Mapping { char *va; size_t n; };
static Mapping map_chunks(const std::vector<size_t> &which) {
    void *va = nullptr;
    CHECK(hipMemAddressReserve(&va, which.size() * kChunk, kChunk, nullptr, 0));
    for (size_t i = 0; i < which.size(); ++i) CHECK(hipMemMap((char *)va + i * kChunk, kChunk, 0, g_pool[which[i]], 0));
    hipMemAccessDesc acc = {}; acc.location = g_prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    CHECK(hipMemSetAccess(va, which.size() * kChunk, &acc, 1));
    return {(char *)va, which.size()};
}
static void unmap(const Mapping &m) {
    CHECK(hipDeviceSynchronize());
    for (size_t i = 0; i < m.n; ++i) CHECK(hipMemUnmap(m.va + i * kChunk, kChunk));
    CHECK(hipMemAddressFree(m.va, m.n * kChunk));
}
static double pattern_rate(const Mapping &m, int N, int steps, int wrap) {
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        CHECK(hipEventRecord(e0));
        rows<<<N / 4, 256>>>((vec4 *)m.va, N, steps, wrap, 262);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < best) best = ms;
    }
    return (double)steps * N * 262 * 16 / best / 1e6;
}
static double fill_rate(const Mapping &m) {
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0));
        fill<<<4096, 256>>>((vec4 *)m.va, (long)(m.n * kChunk / 16));
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < best) best = ms;
    }
    return (double)m.n * kChunk / best / 1e6;
}

int main(int argc, char **argv) {
    const size_t P = argc > 1 ? (size_t)atol(argv[1]) : 8192;
    const int K = argc > 2 ? atoi(argv[2]) : 2;
    const int N = 4096;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    g_prop = {};
    g_prop.type = hipMemAllocationTypePinned; g_prop.location.type = hipMemLocationTypeDevice; g_prop.location.id = 0;
    g_pool.resize(P);
    for (size_t i = 0; i < P; ++i) CHECK(hipMemCreate(&g_pool[i], kChunk, &g_prop, 0));
    printf("pool: %zu chunks of 2 MiB (%.1f GB), created in order\n", P, P * kChunk / 1e9);

    // ---- windows: K steps of rows written 128 / K times over
    const size_t W = ((size_t)K * N * 262 * 16 + kChunk - 1) / kChunk;
    const int loops = 256;
    printf("\nwindow = %d steps of rows = %zu chunks, written %d times; rate [GB/s] by chunk stride (rows) and pool base (columns)\n", K, W, loops / K);
    const size_t bases[4] = {0, P / 4, P / 2, 3 * P / 4};
    printf("%8s", "stride");
    for (size_t b : bases) printf(" %9zu", b);
    printf("\n");
    for (size_t stride = 1; W * stride <= P / 4; stride *= 2) {
        printf("%8zu", stride);
        for (size_t b : bases) {
            std::vector<size_t> which(W);
            for (size_t i = 0; i < W; ++i) which[i] = b + i * stride;
            const Mapping m = map_chunks(which);
            printf(" %9.0f", pattern_rate(m, N, loops, K));
            unmap(m);
        }
        printf("\n");
        fflush(stdout);
    }
    {
        std::mt19937_64 rng(12345);
        printf("%8s", "random");
        for (int t = 0; t < 4; ++t) {
            std::vector<size_t> all(P);
            for (size_t i = 0; i < P; ++i) all[i] = i;
            std::shuffle(all.begin(), all.end(), rng);
            all.resize(W);
            const Mapping m = map_chunks(all);
            printf(" %9.0f", pattern_rate(m, N, loops, K));
            unmap(m);
        }
        printf("\n");
    }
    // one bit of the chunk index at a time: half of the window from base, the other half from base + 2^b (consecutive inside a half)
    printf("\nwindow halves 2^b chunks apart (each half consecutive); rate [GB/s] by b\n");
    for (size_t b = 4; ((size_t)1 << b) + W <= P; ++b) {
        std::vector<size_t> which(W);
        for (size_t i = 0; i < W; ++i) which[i] = (i & 1 ? ((size_t)1 << b) : 0) + i / 2;
        const Mapping m = map_chunks(which);
        printf("  b=%2zu %7.0f", b, pattern_rate(m, N, loops, K));
        unmap(m);
        if ((b - 3) % 5 == 0) printf("\n");
    }
    printf("\n");

    // ---- whole blocks: 256 steps, one pool segment, several chunk orders
    const size_t nb = ((size_t)256 * N * 262 * 16 + kChunk - 1) / kChunk;
    if (2 * nb <= P) {
        printf("\nwhole 256-step blocks (%zu chunks) from pool segment [base, base + %zu); pattern / plain-fill rate [GB/s]\n", nb, nb);
        const size_t seg[3] = {0, (P - nb) / 2, P - nb};
        for (size_t base : seg) {
            auto run = [&](const char *what, const std::vector<size_t> &order) {
                std::vector<size_t> which(nb);
                for (size_t i = 0; i < nb; ++i) which[i] = base + order[i];
                const Mapping m = map_chunks(which);
                const double pr = pattern_rate(m, N, 256, 256), fr = fill_rate(m);
                printf("  base %5zu  %-28s %6.0f / %6.0f\n", base, what, pr, fr);
                fflush(stdout);
                unmap(m);
            };
            std::vector<size_t> order(nb);
            for (size_t i = 0; i < nb; ++i) order[i] = i;
            run("identity", order);
            for (unsigned seed : {1u, 2u, 3u}) {
                std::mt19937_64 rng(seed);
                for (size_t i = 0; i < nb; ++i) order[i] = i;
                std::shuffle(order.begin(), order.end(), rng);
                char what[64]; snprintf(what, sizeof what, "shuffled (seed %u)", seed);
                run(what, order);
            }
            for (size_t s : {(size_t)9, (size_t)33, (size_t)129, (size_t)513, (size_t)1297}) {       // (coprime to nb = 2097 = 3^2 * 233: checked below)
                size_t a = s, bb = nb;
                while (bb) { const size_t t = a % bb; a = bb; bb = t; }
                if (a != 1) s += 2;
                for (size_t i = 0; i < nb; ++i) order[i] = (i * s) % nb;
                char what[64]; snprintf(what, sizeof what, "stride %zu mod n", s);
                run(what, order);
            }
        }
    }
    return 0;
}
