// tools/valu_bank.hip -- does the VGPR bank of the source operands change the issue cost?
//   hipcc -O3 --offload-arch=gfx950 tools/valu_bank.hip -o tools/_build/valu_bank
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
template <int KIND>
__global__ __launch_bounds__(256) void k(float *out, int iters)
{
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) {   // fma, sources in three different banks (reg % 4): v1 v2 v3 -> v20..27
            REP8(asm volatile("v_fma_f32 v20, v1, v2, v3\n v_fma_f32 v21, v5, v6, v7\n v_fma_f32 v22, v9, v10, v11\n v_fma_f32 v23, v13, v14, v15\n"
                              "v_fma_f32 v24, v1, v6, v11\n v_fma_f32 v25, v5, v10, v15\n v_fma_f32 v26, v9, v14, v3\n v_fma_f32 v27, v13, v2, v7\n"
                              ::: "v20","v21","v22","v23","v24","v25","v26","v27");)
        } else if (KIND == 1) {   // fma, all three sources in the same bank
            REP8(asm volatile("v_fma_f32 v20, v0, v4, v8\n v_fma_f32 v21, v1, v5, v9\n v_fma_f32 v22, v2, v6, v10\n v_fma_f32 v23, v3, v7, v11\n"
                              "v_fma_f32 v24, v4, v8, v12\n v_fma_f32 v25, v5, v9, v13\n v_fma_f32 v26, v6, v10, v14\n v_fma_f32 v27, v7, v11, v15\n"
                              ::: "v20","v21","v22","v23","v24","v25","v26","v27");)
        } else if (KIND == 2) {   // mul, two sources different banks
            REP8(asm volatile("v_mul_f32 v20, v1, v2\n v_mul_f32 v21, v5, v6\n v_mul_f32 v22, v9, v10\n v_mul_f32 v23, v13, v14\n"
                              "v_mul_f32 v24, v1, v6\n v_mul_f32 v25, v5, v10\n v_mul_f32 v26, v9, v14\n v_mul_f32 v27, v13, v2\n"
                              ::: "v20","v21","v22","v23","v24","v25","v26","v27");)
        } else if (KIND == 3) {   // mul, same bank
            REP8(asm volatile("v_mul_f32 v20, v0, v4\n v_mul_f32 v21, v1, v5\n v_mul_f32 v22, v2, v6\n v_mul_f32 v23, v3, v7\n"
                              "v_mul_f32 v24, v4, v8\n v_mul_f32 v25, v5, v9\n v_mul_f32 v26, v6, v10\n v_mul_f32 v27, v7, v11\n"
                              ::: "v20","v21","v22","v23","v24","v25","v26","v27");)
        } else if (KIND == 4) {   // dependent chain of fmac (latency-bound per wave)
            REP8(asm volatile("v_fmac_f32 v20, v1, v2\n v_fmac_f32 v20, v5, v6\n v_fmac_f32 v20, v9, v10\n v_fmac_f32 v20, v13, v14\n"
                              "v_fmac_f32 v20, v1, v6\n v_fmac_f32 v20, v5, v10\n v_fmac_f32 v20, v9, v14\n v_fmac_f32 v20, v13, v2\n"
                              ::: "v20");)
        } else if (KIND == 5) {   // two interleaved dependent chains
            REP8(asm volatile("v_fmac_f32 v20, v1, v2\n v_fmac_f32 v21, v5, v6\n v_fmac_f32 v20, v9, v10\n v_fmac_f32 v21, v13, v14\n"
                              "v_fmac_f32 v20, v1, v6\n v_fmac_f32 v21, v5, v10\n v_fmac_f32 v20, v9, v14\n v_fmac_f32 v21, v13, v2\n"
                              ::: "v20","v21");)
        } else if (KIND == 6) {   // mul with an SGPR operand
            REP8(asm volatile("v_mul_f32 v20, s4, v2\n v_mul_f32 v21, s5, v6\n v_mul_f32 v22, s6, v10\n v_mul_f32 v23, s7, v14\n"
                              "v_mul_f32 v24, s4, v6\n v_mul_f32 v25, s5, v10\n v_mul_f32 v26, s6, v14\n v_mul_f32 v27, s7, v2\n"
                              ::: "v20","v21","v22","v23","v24","v25","v26","v27");)
        } else if (KIND == 7) {   // dependent chain through v_rcp (transcendental latency)
            REP8(asm volatile("v_rcp_f32 v20, v20\n v_rcp_f32 v20, v20\n v_rcp_f32 v20, v20\n v_rcp_f32 v20, v20\n"
                              "v_rcp_f32 v20, v20\n v_rcp_f32 v20, v20\n v_rcp_f32 v20, v20\n v_rcp_f32 v20, v20\n"
                              ::: "v20");)
        } else if (KIND == 8) {   // max / cndmask / cmp mix
            REP8(asm volatile("v_max_f32 v20, v1, v2\n v_cmp_le_f32 vcc, v5, v6\n v_cndmask_b32 v22, v9, v10, vcc\n v_sub_f32 v23, v13, v14\n"
                              "v_add_f32 v24, v1, v6\n v_med3_f32 v25, v5, v10, v15\n v_max_f32 v26, v9, v14\n v_cndmask_b32 v27, v13, v2, vcc\n"
                              ::: "v20","v21","v22","v23","v24","v25","v26","v27","vcc");)
        }
    }
    if (iters < 0) out[0] = 1.0f;
}
template <int KIND> void run(const char *name, float *d, int bpc)
{
    const int iters = 2000, blocks = 256 * bpc;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 10); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, iters); (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1); float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double wave_instr = (double)blocks * 4 * iters * 64, per_simd = wave_instr / 1024.0 / (ms * 1e-3);
    printf("%-28s waves/SIMD=%d  %7.3f ms  %5.2f cycles/wave-instr @2.4GHz (%.2f ns)\n", name, bpc, ms, 2.4e9 / per_simd, 1e9 / per_simd);
}
int main()
{
    float *d; (void)hipMalloc(&d, 4);
    for (int bpc : {1, 4, 8}) {
        run<0>("fma 3 banks", d, bpc); run<1>("fma same bank", d, bpc); run<2>("mul 2 banks", d, bpc); run<3>("mul same bank", d, bpc);
        run<6>("mul sgpr operand", d, bpc); run<8>("max/cmp/cndmask/med3 mix", d, bpc);
        run<4>("fmac dependent chain", d, bpc); run<5>("fmac 2 chains", d, bpc); run<7>("rcp dependent chain", d, bpc);
    }
    return 0;
}
