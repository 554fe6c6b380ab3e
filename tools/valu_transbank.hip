// tools/valu_transbank.hip -- does a transcendental block other waves' instructions that touch the VGPR BANK
// (register index mod 4) of its operands?  4 rcp + 28 mul per group; the muls either avoid the rcp's bank or sit in it.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
// rcp on v20, v24, v28, v32 (bank 0, in place)
// A: muls read/write only banks 1,2,3
#define MA "v_mul_f32 v21, v1, v2\n v_mul_f32 v22, v5, v6\n v_mul_f32 v23, v9, v10\n v_mul_f32 v25, v13, v14\n v_mul_f32 v26, v1, v6\n v_mul_f32 v27, v5, v10\n v_mul_f32 v29, v9, v14\n"
// B: muls read bank 0 sources (v0,v4,v8,v12,v16) and write bank 0 (v36..v60 step 4)
#define MB "v_mul_f32 v36, v0, v4\n v_mul_f32 v40, v8, v12\n v_mul_f32 v44, v16, v0\n v_mul_f32 v48, v4, v8\n v_mul_f32 v52, v12, v16\n v_mul_f32 v56, v0, v8\n v_mul_f32 v60, v4, v12\n"
// C: muls read banks 1,2 but WRITE bank 0
#define MC "v_mul_f32 v36, v1, v2\n v_mul_f32 v40, v5, v6\n v_mul_f32 v44, v9, v10\n v_mul_f32 v48, v13, v14\n v_mul_f32 v52, v1, v6\n v_mul_f32 v56, v5, v10\n v_mul_f32 v60, v9, v14\n"
// D: muls read bank 0, write banks 1,2,3
#define MD "v_mul_f32 v21, v0, v4\n v_mul_f32 v22, v8, v12\n v_mul_f32 v23, v16, v0\n v_mul_f32 v25, v4, v8\n v_mul_f32 v26, v12, v16\n v_mul_f32 v27, v0, v8\n v_mul_f32 v29, v4, v12\n"
#define CL "v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v32","v36","v40","v44","v48","v52","v56","v60"
#define GROUP(M) "v_rcp_f32 v20, v20\n" M "v_rcp_f32 v24, v24\n" M "v_rcp_f32 v28, v28\n" M "v_rcp_f32 v32, v32\n" M
#define NOTR(M) "v_mul_f32 v20, v20, v2\n" M "v_mul_f32 v24, v24, v2\n" M "v_mul_f32 v28, v28, v2\n" M "v_mul_f32 v32, v32, v2\n" M
template <int KIND>
__global__ __launch_bounds__(256) void k(float *out, int iters)
{
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) { REP8(asm volatile(GROUP(MA) ::: CL);) }
        else if (KIND == 1) { REP8(asm volatile(GROUP(MB) ::: CL);) }
        else if (KIND == 2) { REP8(asm volatile(GROUP(MC) ::: CL);) }
        else if (KIND == 3) { REP8(asm volatile(GROUP(MD) ::: CL);) }
        else if (KIND == 4) { REP8(asm volatile(NOTR(MA) ::: CL);) }
        else if (KIND == 5) { REP8(asm volatile(NOTR(MB) ::: CL);) }
    }
    if (iters < 0) out[0] = 1.0f;
}
template <int KIND> void run(const char *name, float *d, int bpc)
{
    const int iters = 1000, blocks = 256 * bpc;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 10); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, iters); (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1); float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double groups = (double)iters * 8 * bpc;   // per SIMD
    std::printf("%-44s waves/SIMD=%d  %7.1f ns per 32-instr group\n", name, bpc, ms * 1e6 / groups);
}
int main()
{
    float *d; (void)hipMalloc(&d, 4);
    for (int bpc : {1, 2, 4, 8}) {
        run<4>("32 mul, banks 1-3 (no trans)", d, bpc);
        run<5>("32 mul, bank 0 (no trans)", d, bpc);
        run<0>("4 rcp(bank 0) + 28 mul avoiding bank 0", d, bpc);
        run<1>("4 rcp(bank 0) + 28 mul all in bank 0", d, bpc);
        run<2>("4 rcp(bank 0) + 28 mul WRITING bank 0", d, bpc);
        run<3>("4 rcp(bank 0) + 28 mul READING bank 0", d, bpc);
    }
    return 0;
}
