// What the observation stores of the fused rollout cost by themselves: the launch shape of rollout_kernel at the headline
// batch (1024 workgroups x 4 waves, one environment per wave, 16 waves per CU), every step writing the environment's rows
// (504 + 1048 floats for MATE-4v8-9, non-temporal 16-byte stores, row r*N + env as the kernel lays them out) -- with a
// dependent chain of `spin` f64 fma per step and wave standing in for the simulation.  spin = 0 is the store roofline of
// this access pattern; the curve over spin shows where stores stop and arithmetic starts to bound the launch.
//   hipcc --offload-arch=gfx950 -O3 -o tools/store_roof tools/store_roof.hip && tools/store_roof
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float vec4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <bool NT, int MAP = 0>
__global__ __launch_bounds__(256, 4) void rows(vec4 *cam, vec4 *tgt, int N, int steps, int cam_chunks, int tgt_chunks, int spin, double *sink) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // MAP 1: workgroups are dealt to the 8 XCDs round-robin -- give every XCD a CONTIGUOUS range of environments
    const long blk = MAP == 1 ? (long)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3) : (long)blockIdx.x;
    const long env = blk * 4 + wave;
    if (env >= N) return;
    double a = 1.0 + lane * 1e-9, b = 0.999999;
    for (int r = 0; r < steps; ++r) {
        for (int i = 0; i < spin; ++i) a = fma(a, b, 1e-9);
        const float v = (float)a;
        const vec4 val = {v, v, v, v};
        const long row = (long)r * N + env;
        vec4 *c = cam + row * cam_chunks, *t = tgt + row * tgt_chunks;
        if (MAP == 3) {
            // ... and the two lines a row SHARES with its neighbours (head and tail, written by other waves at other times) go
            // through plain stores -- the L2 merges the halves -- while the full lines stream past the caches
            const int sc = (int)(((size_t)c >> 4) & 7), st = (int)(((size_t)t >> 4) & 7);
            const int cam_last = (cam_chunks + sc) & ~7, tgt_last = (tgt_chunks + st) & ~7;     // first chunk slot of the (partial) last line
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int j = lane + 64 * k, i = j - sc;
                if (i >= 0 && i < cam_chunks) { if ((sc && j < 8) || j >= cam_last) c[i] = val; else __builtin_nontemporal_store(val, c + i); }
            }
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                const int j = lane + 64 * k, i = j - st;
                if (i >= 0 && i < tgt_chunks) { if ((st && j < 8) || j >= tgt_last) t[i] = val; else __builtin_nontemporal_store(val, t + i); }
            }
            continue;
        }
        if (MAP == 2) {
            // every store instruction covers a 128-byte-ALIGNED kilobyte: the lanes' chunks are shifted by the row's offset inside
            // its first line (a row starts at a multiple of 32 bytes, not of 128)
            const int sc = (int)(((size_t)c >> 4) & 7), st = (int)(((size_t)t >> 4) & 7);
#pragma unroll
            for (int k = 0; k < 3; ++k) { const int i = lane + 64 * k - sc; if (i >= 0 && i < cam_chunks) __builtin_nontemporal_store(val, c + i); }
#pragma unroll
            for (int k = 0; k < 5; ++k) { const int i = lane + 64 * k - st; if (i >= 0 && i < tgt_chunks) __builtin_nontemporal_store(val, t + i); }
            continue;
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) { const int i = lane + 64 * k; if (i < cam_chunks) { if (NT) __builtin_nontemporal_store(val, c + i); else c[i] = val; } }
#pragma unroll
        for (int k = 0; k < 6; ++k) { const int i = lane + 64 * k; if (i < tgt_chunks) { if (NT) __builtin_nontemporal_store(val, t + i); else t[i] = val; } }
    }
    if (a == 123.456) sink[0] = a;
}

int main(int argc, char **argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 4096, steps = argc > 2 ? atoi(argv[2]) : 256;
    const int cam_chunks = argc > 4 ? atoi(argv[4]) : 126, tgt_chunks = argc > 5 ? atoi(argv[5]) : 262;      // (16-byte chunks per environment-step: 4 x 126 and 8 x 131 floats)
    if (getenv("PRE_GB")) { void *dummy; CHECK(hipMalloc(&dummy, (size_t)atoi(getenv("PRE_GB")) << 30)); printf("%s GiB allocated first (%p)\n", getenv("PRE_GB"), dummy); }
    vec4 *cam, *tgt; double *sink;
    CHECK(hipMalloc(&cam, (size_t)steps * N * cam_chunks * 16));
    CHECK(hipMalloc(&tgt, (size_t)steps * N * tgt_chunks * 16));
    CHECK(hipMalloc(&sink, 8));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const double bytes = (double)steps * N * (cam_chunks + tgt_chunks) * 16;
    for (int nt = 1; nt >= 0; --nt)
    for (int spin : {0, 250, 500, 1000, 1500, 2000, 2500, 3000, 4000}) {
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            CHECK(hipEventRecord(e0));
            if (nt) rows<true><<<(N + 3) / 4, 256>>>(cam, tgt, N, steps, cam_chunks, tgt_chunks, spin, sink);
            else rows<false><<<(N + 3) / 4, 256>>>(cam, tgt, N, steps, cam_chunks, tgt_chunks, spin, sink);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep && ms < best) best = ms;
        }
        printf("%s spin %4d fma/step/wave: %.3f ms per %d-step launch, %.0f GB/s stored, %.3g env-steps/s\n", nt ? "non-temporal" : "plain       ", spin, best, steps,
               bytes / best / 1e6, (double)steps * N / best * 1e3);
    }
    // ---- does the rate depend on where the blocks lie?  several sets of blocks, the pure-store launch on each
    const int sets = argc > 3 ? atoi(argv[3]) : 6;
    std::vector<vec4 *> cams, tgts;
    for (int s = 0; s < sets; ++s) {
        vec4 *c2, *t2;
        if (hipMalloc(&c2, (size_t)steps * N * cam_chunks * 16) != hipSuccess || hipMalloc(&t2, (size_t)steps * N * tgt_chunks * 16) != hipSuccess) break;
        cams.push_back(c2); tgts.push_back(t2);
    }
    for (int pass = 0; pass < 2; ++pass)
    for (size_t s = 0; s < cams.size(); ++s) {
        float best = 1e30f, worst = 0.f;
        for (int rep = 0; rep < 5; ++rep) {
            CHECK(hipEventRecord(e0));
            rows<true><<<(N + 3) / 4, 256>>>(cams[s], tgts[s], N, steps, cam_chunks, tgt_chunks, 0, sink);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep && ms < best) best = ms;
            if (rep && ms > worst) worst = ms;
        }
        float pbest = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            CHECK(hipEventRecord(e0));
            rows<false><<<(N + 3) / 4, 256>>>(cams[s], tgts[s], N, steps, cam_chunks, tgt_chunks, 0, sink);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep && ms < pbest) pbest = ms;
        }
        float xbest = 1e30f, ybest = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            float ms;
            CHECK(hipEventRecord(e0));
            rows<true, 1><<<(N + 3) / 4, 256>>>(cams[s], tgts[s], N, steps, cam_chunks, tgt_chunks, 0, sink);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep && ms < xbest) xbest = ms;
            CHECK(hipEventRecord(e0));
            rows<true, 2><<<(N + 3) / 4, 256>>>(cams[s], tgts[s], N, steps, cam_chunks, tgt_chunks, 0, sink);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep && ms < ybest) ybest = ms;
        }
        {
            float cb = 1e30f, tb = 1e30f, ms;
            for (int rep = 0; rep < 4; ++rep) {
                CHECK(hipEventRecord(e0));
                rows<true><<<(N + 3) / 4, 256>>>(cams[s], tgts[s], N, steps, cam_chunks, 0, 0, sink);
                CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (rep && ms < cb) cb = ms;
                CHECK(hipEventRecord(e0));
                rows<true><<<(N + 3) / 4, 256>>>(cams[s], tgts[s], N, steps, 0, tgt_chunks, 0, sink);
                CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (rep && ms < tb) tb = ms;
            }
            printf("   one block at a time (non-temporal): camera block %.0f GB/s, target block %.0f GB/s\n", (double)steps * N * cam_chunks * 16 / cb / 1e6, (double)steps * N * tgt_chunks * 16 / tb / 1e6);
        }
        float zbest = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            float ms;
            CHECK(hipEventRecord(e0));
            rows<true, 3><<<(N + 3) / 4, 256>>>(cams[s], tgts[s], N, steps, cam_chunks, tgt_chunks, 0, sink);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep && ms < zbest) zbest = ms;
        }
        printf("   non-temporal: XCD-contiguous environments %.0f GB/s | line-aligned store windows %.0f GB/s | ... and shared lines through plain stores %.0f GB/s\n", bytes / xbest / 1e6, bytes / ybest / 1e6, bytes / zbest / 1e6);
        printf("set %zu cam %p tgt %p: stores only, non-temporal best %.3f worst %.3f ms %.0f GB/s | plain best %.3f ms %.0f GB/s\n", s, (void *)cams[s], (void *)tgts[s], best, worst, bytes / best / 1e6, pbest, bytes / pbest / 1e6);
    }
    // ---- is it the blocks, or the PAIR?  camera block of set i with the target block of set j
    printf("non-temporal GB/s, camera block of set i (row) with target block of set j (column):\n");
    for (size_t i = 0; i < cams.size(); ++i) {
        for (size_t j = 0; j < tgts.size(); ++j) {
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                CHECK(hipEventRecord(e0));
                rows<true><<<(N + 3) / 4, 256>>>(cams[i], tgts[j], N, steps, cam_chunks, tgt_chunks, 0, sink);
                CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (rep && ms < best) best = ms;
            }
            printf(" %5.0f", bytes / best / 1e6);
        }
        printf("\n");
    }
    // ---- one allocation holding both blocks: does the rate depend on the GAP between them, or on the allocation?
    for (int trial = 0; trial < 3; ++trial) {
        char *big;
        const size_t cam_bytes = (size_t)steps * N * cam_chunks * 16, tgt_bytes = (size_t)steps * N * tgt_chunks * 16;
        if (hipMalloc(&big, cam_bytes + tgt_bytes + (512u << 20)) != hipSuccess) break;
        printf("one allocation %p, plain-store GB/s by gap between the blocks [MiB]:", (void *)big);
        for (size_t gap : {0u, 2u, 4u, 8u, 16u, 32u, 34u, 64u, 100u, 128u, 256u, 510u}) {
            vec4 *c2 = (vec4 *)big, *t2 = (vec4 *)(big + cam_bytes + (gap << 20));
            float best = 1e30f;
            for (int rep = 0; rep < 4; ++rep) {
                CHECK(hipEventRecord(e0));
                rows<false><<<(N + 3) / 4, 256>>>(c2, t2, N, steps, cam_chunks, tgt_chunks, 0, sink);
                CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (rep && ms < best) best = ms;
            }
            printf(" %zu:%.0f", gap, bytes / best / 1e6);
        }
        printf("\n");      // (kept allocated: the next trial lies elsewhere)
    }
    // ---- ... and does it vary INSIDE a set?  16-step slices (0.4 GB) of each set, launched separately
    const int slice = 16;
    for (size_t s = 0; s < cams.size() && steps % slice == 0; ++s) {
        printf("set %zu, GB/s per %d-step slice:", s, slice);
        for (int first = 0; first < steps; first += slice) {
            float best = 1e30f;
            for (int rep = 0; rep < 4; ++rep) {
                CHECK(hipEventRecord(e0));
                rows<true><<<(N + 3) / 4, 256>>>(cams[s] + (size_t)first * N * cam_chunks, tgts[s] + (size_t)first * N * tgt_chunks, N, slice, cam_chunks, tgt_chunks, 0, sink);
                CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (rep && ms < best) best = ms;
            }
            printf(" %.0f", bytes * slice / steps / best / 1e6);
        }
        printf("\n");
    }
    return 0;
}
