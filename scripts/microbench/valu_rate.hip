// valu_rate.hip -- issue cost (cycles per wave64 instruction per SIMD) of the VALU operations the EVP kernels use.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int OP>
__global__ void __launch_bounds__(256) k(double* out, int iters) {
    double a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const double c = 1.0000001, d = 1e-9;
    for (int k = 0; k < iters; ++k) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (OP == 0) {   // fma f64
                asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                             "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));
            } else if (OP == 1) {   // mul f64
                asm volatile("v_mul_f64 %0, %0, %8\n v_mul_f64 %1, %1, %8\n v_mul_f64 %2, %2, %8\n v_mul_f64 %3, %3, %8\n"
                             "v_mul_f64 %4, %4, %8\n v_mul_f64 %5, %5, %8\n v_mul_f64 %6, %6, %8\n v_mul_f64 %7, %7, %8\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));
            } else if (OP == 2) {   // add f64
                asm volatile("v_add_f64 %0, %0, %9\n v_add_f64 %1, %1, %9\n v_add_f64 %2, %2, %9\n v_add_f64 %3, %3, %9\n"
                             "v_add_f64 %4, %4, %9\n v_add_f64 %5, %5, %9\n v_add_f64 %6, %6, %9\n v_add_f64 %7, %7, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));
            } else if (OP == 3) {   // v_mov_b32 (lo halves)
                asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n"
                             "v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0\n"
                             : "+v"(((int*)&a0)[0]), "+v"(((int*)&a1)[0]), "+v"(((int*)&a2)[0]), "+v"(((int*)&a3)[0]),
                               "+v"(((int*)&a4)[0]), "+v"(((int*)&a5)[0]), "+v"(((int*)&a6)[0]), "+v"(((int*)&a7)[0]));
            } else if (OP == 4) {   // dpp wave_shr
                asm volatile("s_nop 1\n v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %3 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
                             "v_mov_b32_dpp %4, %5 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %7 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
                             "v_mov_b32_dpp %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %2 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
                             "v_mov_b32_dpp %5, %4 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %6 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
                             : "+v"(((int*)&a0)[0]), "+v"(((int*)&a1)[0]), "+v"(((int*)&a2)[0]), "+v"(((int*)&a3)[0]),
                               "+v"(((int*)&a4)[0]), "+v"(((int*)&a5)[0]), "+v"(((int*)&a6)[0]), "+v"(((int*)&a7)[0]));
            } else if (OP == 5) {   // rcp f64
                asm volatile("v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1\n v_rcp_f64 %2, %2\n v_rcp_f64 %3, %3\n"
                             "v_rcp_f64 %4, %4\n v_rcp_f64 %5, %5\n v_rcp_f64 %6, %6\n v_rcp_f64 %7, %7\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if (OP == 6) {   // mov b64
                asm volatile("v_mov_b64 %0, %1\n v_mov_b64 %1, %2\n v_mov_b64 %2, %3\n v_mov_b64 %3, %4\n"
                             "v_mov_b64 %4, %5\n v_mov_b64 %5, %6\n v_mov_b64 %6, %7\n v_mov_b64 %7, %0\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if (OP == 7) {   // cndmask b32 with vcc
                asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n"
                             "v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %5, %5, %6, vcc\n v_cndmask_b32 %6, %6, %7, vcc\n v_cndmask_b32 %7, %7, %0, vcc\n"
                             : "+v"(((int*)&a0)[0]), "+v"(((int*)&a1)[0]), "+v"(((int*)&a2)[0]), "+v"(((int*)&a3)[0]),
                               "+v"(((int*)&a4)[0]), "+v"(((int*)&a5)[0]), "+v"(((int*)&a6)[0]), "+v"(((int*)&a7)[0]) :: "vcc");
            } else if (OP == 8) {   // dependent fma chain (latency)
                asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %0, %0, %8, %9\n"
                             "v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %0, %0, %8, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));
            } else if (OP == 10) {   // cndmask e64 with sgpr pair mask
                asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]\n v_cndmask_b32_e64 %1, %1, %2, s[20:21]\n v_cndmask_b32_e64 %2, %2, %3, s[20:21]\n v_cndmask_b32_e64 %3, %3, %4, s[20:21]\n"
                             "v_cndmask_b32_e64 %4, %4, %5, s[20:21]\n v_cndmask_b32_e64 %5, %5, %6, s[20:21]\n v_cndmask_b32_e64 %6, %6, %7, s[20:21]\n v_cndmask_b32_e64 %7, %7, %0, s[20:21]\n"
                             : "+v"(((int*)&a0)[0]), "+v"(((int*)&a1)[0]), "+v"(((int*)&a2)[0]), "+v"(((int*)&a3)[0]),
                               "+v"(((int*)&a4)[0]), "+v"(((int*)&a5)[0]), "+v"(((int*)&a6)[0]), "+v"(((int*)&a7)[0]) :: "s20", "s21");
            } else if (OP == 11) {   // max f64
                asm volatile("v_max_f64 %0, %0, %8\n v_max_f64 %1, %1, %8\n v_max_f64 %2, %2, %8\n v_max_f64 %3, %3, %8\n"
                             "v_max_f64 %4, %4, %8\n v_max_f64 %5, %5, %8\n v_max_f64 %6, %6, %8\n v_max_f64 %7, %7, %8\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));
            } else if (OP == 12) {   // cmp f64 -> vcc
                asm volatile("v_cmp_lt_f64 vcc, %0, %8\n v_cmp_lt_f64 vcc, %1, %8\n v_cmp_lt_f64 vcc, %2, %8\n v_cmp_lt_f64 vcc, %3, %8\n"
                             "v_cmp_lt_f64 vcc, %4, %8\n v_cmp_lt_f64 vcc, %5, %8\n v_cmp_lt_f64 vcc, %6, %8\n v_cmp_lt_f64 vcc, %7, %8\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d) : "vcc");
            } else if (OP == 13) {   // cmp + 2 cndmask (one double select)
                asm volatile("v_cmp_lt_f64 vcc, %0, %8\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n"
                             "v_cmp_lt_f64 vcc, %5, %8\n v_cndmask_b32 %6, %6, %7, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n"
                             "v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %5, %5, %8, %9\n"
                             : "+v"(a0), "+v"(((int*)&a1)[0]), "+v"(((int*)&a2)[0]), "+v"(((int*)&a3)[0]),
                               "+v"(((int*)&a4)[0]), "+v"(a5), "+v"(((int*)&a6)[0]), "+v"(((int*)&a7)[0]) : "v"(c), "v"(d) : "vcc");
            } else if (OP == 14) {   // s_mov_b64 / s_and_b64 / s_cselect_b64 mix
                asm volatile("s_mov_b64 s[20:21], s[22:23]\n s_and_b64 s[22:23], s[20:21], exec\n s_cselect_b64 s[20:21], s[22:23], exec\n s_or_b64 s[22:23], s[20:21], exec\n"
                             "s_mov_b64 s[20:21], s[22:23]\n s_and_b64 s[22:23], s[20:21], exec\n s_cselect_b64 s[20:21], s[22:23], exec\n s_or_b64 s[22:23], s[20:21], exec\n"
                             ::: "s20", "s21", "s22", "s23", "scc");
            } else if (OP == 15) {   // not-taken branches
                asm volatile("s_cmp_eq_u32 %0, -1\n s_cbranch_scc1 9f\n s_cmp_eq_u32 %0, -2\n s_cbranch_scc1 9f\n"
                             "s_cmp_eq_u32 %0, -3\n s_cbranch_scc1 9f\n s_cmp_eq_u32 %0, -4\n s_cbranch_scc1 9f\n 9:\n" :: "s"(k) : "scc");
            } else if (OP == 16) {   // taken branches (skip one s_nop)
                asm volatile("s_branch 1f\n s_nop 0\n 1: s_branch 2f\n s_nop 0\n 2: s_branch 3f\n s_nop 0\n 3: s_branch 4f\n s_nop 0\n 4:\n"
                             "s_branch 5f\n s_nop 0\n 5: s_branch 6f\n s_nop 0\n 6: s_branch 7f\n s_nop 0\n 7: s_branch 8f\n s_nop 0\n 8:\n" ::: "scc");
            } else if (OP == 9) {   // salu
                int s0 = k, s1 = u;
                asm volatile("s_add_i32 %0, %0, %1\n s_add_i32 %1, %1, %0\n s_add_i32 %0, %0, %1\n s_add_i32 %1, %1, %0\n"
                             "s_add_i32 %0, %0, %1\n s_add_i32 %1, %1, %0\n s_add_i32 %0, %0, %1\n s_add_i32 %1, %1, %0\n" : "+s"(s0), "+s"(s1) :: "scc");
                if (s0 == 0x7fffffff) a0 += 1;
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int OP>
void run(const char* name, double* d) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 2000;
    for (int nb : {256, 512, 1024}) {
        hipLaunchKernelGGL((k<OP>), dim3(nb), dim3(256), 0, 0, d, 10);
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k<OP>), dim3(nb), dim3(256), 0, 0, d, iters);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double instr_per_simd = (double)iters * 64 * (nb / 256.0);
        printf("%-34s waves/SIMD %d: %.3f ms  -> %.2f ns per wave-instruction per SIMD (%.1f cycles at 2.4 GHz)\n", name, nb / 256, ms,
               ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
    }
}

int main() {
    double* d; CK(hipMalloc(&d, 1024 * 256 * 8));
    run<0>("v_fma_f64", d); run<1>("v_mul_f64", d); run<2>("v_add_f64", d); run<3>("v_mov_b32", d); run<4>("dpp wave_shr", d);
    run<5>("v_rcp_f64", d); run<6>("v_mov_b64", d); run<7>("v_cndmask_b32", d); run<8>("fma dep chain", d); run<9>("s_add_i32", d);
    run<10>("cndmask e64 sgpr", d); run<11>("v_max_f64", d); run<12>("v_cmp_lt_f64", d); run<13>("cmp+2cnd+fma mix", d);
    run<14>("salu b64 mix", d); run<15>("cmp+branch not taken (x2 instr)", d); run<16>("s_branch taken", d);
    return 0;
}
