// valu_rates.hip -- issue cost of the vector instructions the engine's kernels are made of, on the box it runs on:
// cycles per wave64 instruction with one wave per SIMD (latency-free streams of independent instructions), per opcode.
//   hipcc --offload-arch=gfx950 -O2 -o valu_rates tools/valu_rates.hip && ./valu_rates
// What DESIGN.md's "instruction count is throughput" weighs the counts with.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x
#define BODY(NAME, ASM)                                                                         \
    __global__ void NAME(long long *out, int iters) {                                                    \
        double a = threadIdx.x + 1.5, b = threadIdx.x * 0.25 + 2.0, c = 0.5, d = 1.25;                    \
        double e = 3.5, f = 2.25, g = 0.75, h = 5.0;                                                      \
        unsigned i0 = threadIdx.x * 77u + 1u, i1 = (threadIdx.x & 63u) * 4u, i2 = 5u, i3 = 9u;                    \
        __shared__ unsigned lds_pad[1024]; if (iters < 0) lds_pad[threadIdx.x] = i0;                    \
        float f0 = threadIdx.x, f1 = 2.f, f2 = 3.f, f3 = .5f;                                             \
        unsigned long long w0 = threadIdx.x, w1 = 3;                                                      \
        const long long t0 = __builtin_amdgcn_s_memtime();                                                \
        for (int it = 0; it < iters; ++it) {                                                              \
            asm volatile(REP8(ASM) : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(i0), "+v"(i1), "+v"(i2), \
                         "+v"(i3), "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(w0), "+v"(w1) : : "memory", "vcc", "s20", "s21", "s22", "s23");                                \
        }                                                                                                 \
        const long long t1 = __builtin_amdgcn_s_memtime();                                                \
        if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;                                                  \
        if (a + b + c + d + e + f + g + h + i0 + i1 + i2 + i3 + f0 + f1 + f2 + f3 + (double)w0 + (double)w1 == 1.2345) out[0] = 0;          \
    }

// four independent destinations per line so that no instruction waits for the previous one
BODY(k_fma_f64, "v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3\n v_fma_f64 %4, %4, %2, %3\n v_fma_f64 %5, %5, %2, %3\n")
BODY(k_mul_f64, "v_mul_f64 %0, %0, %2\n v_mul_f64 %1, %1, %2\n v_mul_f64 %4, %4, %2\n v_mul_f64 %5, %5, %2\n")
BODY(k_add_f64, "v_add_f64 %0, %0, %2\n v_add_f64 %1, %1, %2\n v_add_f64 %4, %4, %2\n v_add_f64 %5, %5, %2\n")
BODY(k_max_f64, "v_max_f64 %0, %0, %2\n v_max_f64 %1, %1, %2\n v_max_f64 %4, %4, %2\n v_max_f64 %5, %5, %2\n")
BODY(k_rcp_f64, "v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1\n v_rcp_f64 %4, %4\n v_rcp_f64 %5, %5\n")
BODY(k_rsq_f64, "v_rsq_f64 %0, %0\n v_rsq_f64 %1, %1\n v_rsq_f64 %4, %4\n v_rsq_f64 %5, %5\n")
BODY(k_cvt_f64_u32, "v_cvt_f64_u32 %0, %8\n v_cvt_f64_u32 %1, %9\n v_cvt_f64_u32 %4, %10\n v_cvt_f64_u32 %5, %11\n")
BODY(k_cvt_f32_f64, "v_cvt_f32_f64 %12, %0\n v_cvt_f32_f64 %13, %1\n v_cvt_f32_f64 %14, %4\n v_cvt_f32_f64 %15, %5\n")
BODY(k_cvt_f64_f32, "v_cvt_f64_f32 %0, %12\n v_cvt_f64_f32 %1, %13\n v_cvt_f64_f32 %4, %14\n v_cvt_f64_f32 %5, %15\n")
BODY(k_ldexp_f64, "v_ldexp_f64 %0, %0, 3\n v_ldexp_f64 %1, %1, 3\n v_ldexp_f64 %4, %4, 3\n v_ldexp_f64 %5, %5, 3\n")
BODY(k_mad_u64_u32, "v_mad_u64_u32 %16, vcc, %8, %9, 0\n v_mad_u64_u32 %17, vcc, %10, %11, 0\n v_mad_u64_u32 %16, vcc, %9, %10, 0\n v_mad_u64_u32 %17, vcc, %8, %11, 0\n")
BODY(k_mul_lo_u32, "v_mul_lo_u32 %8, %8, %9\n v_mul_lo_u32 %10, %10, %9\n v_mul_lo_u32 %11, %11, %9\n v_mul_lo_u32 %8, %8, %9\n")
BODY(k_mul_hi_u32, "v_mul_hi_u32 %8, %8, %9\n v_mul_hi_u32 %10, %10, %9\n v_mul_hi_u32 %11, %11, %9\n v_mul_hi_u32 %8, %8, %9\n")
BODY(k_xor_b32, "v_xor_b32 %8, %8, %9\n v_xor_b32 %10, %10, %9\n v_xor_b32 %11, %11, %9\n v_xor_b32 %8, %8, %10\n")
BODY(k_and_b32, "v_and_b32 %8, %8, %9\n v_and_b32 %10, %10, %9\n v_and_b32 %11, %11, %9\n v_and_b32 %8, %8, %10\n")
BODY(k_add_sdwa, "v_add_u32_sdwa %8, %9, %10 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n v_add_u32_sdwa %11, %9, %10 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n v_add_u32_sdwa %8, %9, %11 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n v_add_u32_sdwa %11, %9, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n")
BODY(k_lshl_add_u64, "v_lshl_add_u64 %16, %16, 4, %17\n v_lshl_add_u64 %17, %17, 4, %16\n v_lshl_add_u64 %16, %16, 4, %17\n v_lshl_add_u64 %17, %17, 4, %16\n")
BODY(k_cndmask, "v_cndmask_b32 %8, %8, %9, vcc\n v_cndmask_b32 %10, %10, %9, vcc\n v_cndmask_b32 %11, %11, %9, vcc\n v_cndmask_b32 %8, %8, %10, vcc\n")
BODY(k_cndmask_indep, "v_cndmask_b32 %8, %9, %10, vcc\n v_cndmask_b32 %11, %9, %10, vcc\n v_cndmask_b32 %8, %9, %10, vcc\n v_cndmask_b32 %11, %9, %10, vcc\n")
BODY(k_cndmask_e64, "v_cndmask_b32_e64 %8, %9, %10, s[20:21]\n v_cndmask_b32_e64 %11, %9, %10, s[20:21]\n v_cndmask_b32_e64 %8, %9, %10, s[20:21]\n v_cndmask_b32_e64 %11, %9, %10, s[20:21]\n")
BODY(k_cmp_cndmask, "v_cmp_gt_u32 vcc, %8, %9\n v_cndmask_b32 %10, %9, %10, vcc\n v_cmp_gt_u32 vcc, %11, %9\n v_cndmask_b32 %8, %9, %11, vcc\n")
BODY(k_bfi_b32, "v_bfi_b32 %8, %9, %10, %8\n v_bfi_b32 %11, %9, %10, %11\n v_bfi_b32 %8, %9, %10, %8\n v_bfi_b32 %11, %9, %10, %11\n")
BODY(k_add_u32, "v_add_u32 %8, %8, %9\n v_add_u32 %10, %10, %9\n v_add_u32 %11, %11, %9\n v_add_u32 %8, %8, %10\n")
BODY(k_lshl_add_u32, "v_lshl_add_u32 %8, %8, 3, %9\n v_lshl_add_u32 %10, %10, 3, %9\n v_lshl_add_u32 %11, %11, 3, %9\n v_lshl_add_u32 %8, %8, 3, %10\n")
BODY(k_add_f32, "v_add_f32 %12, %12, %13\n v_add_f32 %14, %14, %13\n v_add_f32 %15, %15, %13\n v_add_f32 %12, %12, %14\n")
BODY(k_cmp_f32, "v_cmp_gt_f32 vcc, %12, %13\n v_cmp_gt_f32 vcc, %14, %13\n v_cmp_gt_f32 vcc, %15, %13\n v_cmp_gt_f32 vcc, %12, %14\n")
BODY(k_cmp_e64, "v_cmp_gt_u32_e64 s[20:21], %8, %9\n v_cmp_gt_u32_e64 s[22:23], %10, %9\n v_cmp_gt_u32_e64 s[20:21], %11, %9\n v_cmp_gt_u32_e64 s[22:23], %8, %10\n")
BODY(k_readlane, "v_readlane_b32 s20, %8, 3\n v_readlane_b32 s21, %9, 3\n v_readlane_b32 s22, %10, 3\n v_readlane_b32 s23, %11, 3\n")
BODY(k_salu, "s_add_u32 s20, s20, s21\n s_add_u32 s22, s22, s21\n s_add_u32 s23, s23, s21\n s_add_u32 s20, s20, s22\n")
BODY(k_valu_salu, "v_add_u32 %8, %8, %9\n s_add_u32 s20, s20, s21\n v_add_u32 %10, %10, %9\n s_add_u32 s22, s22, s21\n")
BODY(k_saveexec, "s_and_saveexec_b64 s[20:21], vcc\n s_or_b64 exec, exec, s[20:21]\n s_and_saveexec_b64 s[22:23], vcc\n s_or_b64 exec, exec, s[22:23]\n")
BODY(k_ds_read_indep, "ds_read_b32 %8, %9\n ds_read_b32 %10, %9 offset:256\n ds_read_b32 %11, %9 offset:512\n ds_read_b32 %8, %9 offset:768\n")
BODY(k_ds_write, "ds_write_b32 %9, %8\n ds_write_b32 %9, %10 offset:256\n ds_write_b32 %9, %11 offset:512\n ds_write_b32 %9, %8 offset:768\n")
BODY(k_ds_read_and, "ds_read_b32 %8, %9\n ds_read_b32 %10, %9 offset:256\n s_waitcnt lgkmcnt(0)\n v_and_b32 %11, %8, %10\n")
BODY(k_fma_f32, "v_fma_f32 %12, %12, %13, %14\n v_fma_f32 %15, %15, %13, %14\n v_fma_f32 %12, %12, %13, %15\n v_fma_f32 %15, %15, %13, %12\n")
BODY(k_pk_fma_f32, "v_pk_fma_f32 %0, %0, %2, %3\n v_pk_fma_f32 %1, %1, %2, %3\n v_pk_fma_f32 %4, %4, %2, %3\n v_pk_fma_f32 %5, %5, %2, %3\n")
BODY(k_mov_b32, "v_mov_b32 %8, %9\n v_mov_b32 %10, %9\n v_mov_b32 %11, %9\n v_mov_b32 %8, %10\n")
BODY(k_mov_b64, "v_mov_b64 %0, %2\n v_mov_b64 %1, %2\n v_mov_b64 %4, %2\n v_mov_b64 %5, %2\n")
BODY(k_cmp_f64, "v_cmp_gt_f64 vcc, %0, %2\n v_cmp_gt_f64 vcc, %1, %2\n v_cmp_gt_f64 vcc, %4, %2\n v_cmp_gt_f64 vcc, %5, %2\n")
BODY(k_sqrt_f32, "v_sqrt_f32 %12, %12\n v_sqrt_f32 %13, %13\n v_sqrt_f32 %14, %14\n v_sqrt_f32 %15, %15\n")
BODY(k_rndne_f64, "v_rndne_f64 %0, %0\n v_rndne_f64 %1, %1\n v_rndne_f64 %4, %4\n v_rndne_f64 %5, %5\n")
BODY(k_ds_read, "ds_read_b32 %8, %9\n ds_read_b32 %10, %9\n ds_read_b32 %11, %9\n ds_read_b32 %8, %9\n s_waitcnt lgkmcnt(0)\n")

struct Case { const char *name; void (*fn)(long long *, int); };
int main() {
    Case cases[] = {{"v_fma_f64", k_fma_f64}, {"v_mul_f64", k_mul_f64}, {"v_add_f64", k_add_f64}, {"v_max_f64", k_max_f64}, {"v_rcp_f64", k_rcp_f64}, {"v_rsq_f64", k_rsq_f64},
                    {"v_cvt_f64_u32", k_cvt_f64_u32}, {"v_cvt_f32_f64", k_cvt_f32_f64}, {"v_cvt_f64_f32", k_cvt_f64_f32}, {"v_ldexp_f64", k_ldexp_f64},
                    {"v_mad_u64_u32", k_mad_u64_u32}, {"v_mul_lo_u32", k_mul_lo_u32}, {"v_mul_hi_u32", k_mul_hi_u32}, {"v_xor_b32", k_xor_b32}, {"v_and_b32", k_and_b32},
                    {"v_add_u32_sdwa", k_add_sdwa}, {"v_lshl_add_u64", k_lshl_add_u64}, {"v_cndmask_b32", k_cndmask}, {"v_cndmask indep", k_cndmask_indep}, {"v_cndmask_e64 sgpr", k_cndmask_e64}, {"v_cmp+v_cndmask", k_cmp_cndmask},
                    {"v_bfi_b32", k_bfi_b32}, {"v_add_u32", k_add_u32}, {"v_lshl_add_u32", k_lshl_add_u32}, {"v_add_f32", k_add_f32}, {"v_cmp_gt_f32", k_cmp_f32},
                    {"v_cmp_e64 sgpr", k_cmp_e64}, {"v_readlane_b32", k_readlane}, {"s_add_u32", k_salu}, {"v_add_u32+s_add_u32", k_valu_salu}, {"saveexec+restore", k_saveexec},
                    {"ds_read_b32 no wait", k_ds_read_indep}, {"ds_write_b32", k_ds_write}, {"2 ds_read+wait+and", k_ds_read_and}, {"v_fma_f32", k_fma_f32}, {"v_pk_fma_f32", k_pk_fma_f32},
                    {"v_mov_b32", k_mov_b32}, {"v_mov_b64", k_mov_b64}, {"v_cmp_gt_f64", k_cmp_f64}, {"v_sqrt_f32", k_sqrt_f32}, {"v_rndne_f64", k_rndne_f64}, {"ds_read_b32 x4+wait", k_ds_read}};
    long long *out;
    hipMalloc(&out, 4096 * sizeof(long long));
    const int iters = 2000;
    // s_memtime ticks at a constant 100 MHz on this part; calibrate against the shader clock with a known v_mov stream
    for (int waves_per_simd : {1, 4}) {
        printf("== %d wave(s) per SIMD (workgroups of %d threads, one per CU)\n", waves_per_simd, 256 * waves_per_simd);
        for (auto &c : cases) {
            hipLaunchKernelGGL(c.fn, dim3(256), dim3(256 * waves_per_simd), 0, 0, out, 10);      // warm
            hipLaunchKernelGGL(c.fn, dim3(256), dim3(256 * waves_per_simd), 0, 0, out, iters);
            hipDeviceSynchronize();
            std::vector<long long> h(256);
            hipMemcpy(h.data(), out, 256 * sizeof(long long), hipMemcpyDeviceToHost);
            double mean = 0;
            for (long long v : h) mean += (double)v;
            mean /= 256.0;
            printf("%-22s %8.3f memtime-ticks per instruction per wave (x%d waves sharing the SIMD)\n", c.name, mean / (iters * 32.0), waves_per_simd);
        }
    }
    return 0;
}
