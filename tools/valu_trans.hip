// tools/valu_trans.hip -- what does a transcendental cost inside a stream of plain VALU work?
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
#define M7 "v_mul_f32 v21, v1, v2\n v_mul_f32 v22, v5, v6\n v_mul_f32 v23, v9, v10\n v_mul_f32 v24, v13, v14\n v_mul_f32 v25, v1, v6\n v_mul_f32 v26, v5, v10\n v_mul_f32 v27, v9, v14\n"
#define CL "v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31"
template <int KIND>
__global__ __launch_bounds__(256) void k(float *out, int iters)
{
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) {        // 1 rcp (self-dependent only) + 7 independent muls, x4   [32 instr]
            REP8(asm volatile("v_rcp_f32 v20, v20\n" M7 "v_rcp_f32 v28, v28\n" M7 "v_rcp_f32 v29, v29\n" M7 "v_rcp_f32 v30, v30\n" M7 ::: CL);)
        } else if (KIND == 1) { // the same 4 rcp grouped back to back, then 28 muls
            REP8(asm volatile("v_rcp_f32 v20, v20\n v_rcp_f32 v28, v28\n v_rcp_f32 v29, v29\n v_rcp_f32 v30, v30\n" M7 M7 M7 M7 ::: CL);)
        } else if (KIND == 2) { // rcp whose result is used by the very next instruction
            REP8(asm volatile("v_rcp_f32 v20, v3\n v_mul_f32 v31, v20, v2\n" M7 "v_rcp_f32 v28, v7\n v_mul_f32 v31, v28, v2\n" M7
                              "v_rcp_f32 v29, v11\n v_mul_f32 v31, v29, v2\n" M7 "v_rcp_f32 v30, v15\n v_mul_f32 v31, v30, v2\n" M7 ::: CL);)
        } else if (KIND == 3) { // 32 muls (baseline for the same instruction count as kind 0)
            REP8(asm volatile("v_mul_f32 v20, v3, v2\n" M7 "v_mul_f32 v28, v7, v2\n" M7 "v_mul_f32 v29, v11, v2\n" M7 "v_mul_f32 v30, v15, v2\n" M7 ::: CL);)
        } else if (KIND == 4) { // rcp fed by a plain VALU result and feeding one (chain through the two pipes)
            REP8(asm volatile("v_mul_f32 v20, v20, v2\n v_rcp_f32 v20, v20\n v_mul_f32 v20, v20, v2\n" M7 "v_mul_f32 v28, v28, v2\n v_rcp_f32 v28, v28\n v_mul_f32 v28, v28, v2\n" M7 ::: CL);)
        }
    }
    if (iters < 0) out[0] = 1.0f;
}
template <int KIND> void run(const char *name, float *d, int bpc, int n_instr, int n_trans)
{
    const int iters = 1000, blocks = 256 * bpc;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 10); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, iters); (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1); float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double ns_per_group = ms * 1e6 / iters / 8 / bpc;     // one asm group (n_instr instructions) per wave per SIMD
    printf("%-34s waves/SIMD=%d  %6.1f ns per %2d-instr group (%d trans)  -> %.2f ns/instr\n", name, bpc, ns_per_group, n_instr, n_trans, ns_per_group / n_instr);
}
int main()
{
    float *d; (void)hipMalloc(&d, 4);
    for (int bpc : {4, 8}) {
        run<3>("32 mul", d, bpc, 32, 0);
        run<0>("4x(rcp + 7 mul), rcp isolated", d, bpc, 32, 4);
        run<1>("4 rcp grouped + 28 mul", d, bpc, 32, 4);
        run<2>("4x(rcp, use next, 7 mul)", d, bpc, 36, 4);
        run<4>("2x(mul->rcp->mul chain, 7 mul)", d, bpc, 20, 2);
    }
    return 0;
}
