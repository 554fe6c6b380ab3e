// tools/valu_mix.hip -- compiler-generated arithmetic shaped like the shading code, with and
// without transcendental instructions, to see what a realistic dependent mix sustains.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
template <int KIND>
__device__ __forceinline__ float inv(float x)
{
    if (KIND == 0) return x * 0.37f + 0.11f;          // no transcendental: stands in for rcp
    return __builtin_amdgcn_rcpf(x);
}
template <int KIND>
__device__ __forceinline__ float isq(float x)
{
    if (KIND == 0) return x * 0.21f + 0.13f;
    return __builtin_amdgcn_rsqf(x);
}
template <int KIND, int UNROLL>
__global__ __launch_bounds__(256) void k(float *out, int iters, float seed)
{
    float x = seed + threadIdx.x * 1e-3f, y = seed * 0.5f + blockIdx.x * 1e-4f;
    float n0 = 0.1f + x, n1 = 0.2f - y, n2 = 0.9f, A = 0.3f + x * y, s0 = 0.5f, d0 = 0.4f;
    float acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0;
    float c0 = seed, c1 = seed * 2, c2 = seed * 3, l0 = -seed, l1 = seed * 0.7f, l2 = seed * 1.5f;
#pragma unroll UNROLL
    for (int i = 0; i < iters; ++i) {
        float ax = c0 - x, ay = c1 - y, az = c2;
        float bx = l0 - x, by = l1 - y, bz = l2;
        float da = (ax * ax + ay * ay) + az * az, db = (bx * bx + by * by) + bz * bz;
        float ia = isq<KIND>(da), ib = isq<KIND>(db);
        float la = da * ia, lb = db * ib;
        la = fma_(fma_(-la, la, da), 0.5f * ia, la); lb = fma_(fma_(-lb, lb, db), 0.5f * ib, lb);
        float ra = fma_(fma_(-la, ia, 1.0f), ia, ia), rb = fma_(fma_(-lb, ib, 1.0f), ib, ib);
        float q;
        q = ax * ra; float wox = fma_(fma_(-la, q, ax), ra, q);
        q = ay * ra; float woy = fma_(fma_(-la, q, ay), ra, q);
        q = az * ra; float woz = fma_(fma_(-la, q, az), ra, q);
        q = bx * rb; float wix = fma_(fma_(-lb, q, bx), rb, q);
        q = by * rb; float wiy = fma_(fma_(-lb, q, by), rb, q);
        q = bz * rb; float wiz = fma_(fma_(-lb, q, bz), rb, q);
        float hx = (wix + wox) * 0.5f, hy = (wiy + woy) * 0.5f, hz = (wiz + woz) * 0.5f;
        float dh = (hx * hx + hy * hy) + hz * hz;
        float ih = isq<KIND>(dh);
        hx *= ih; hy *= ih; hz *= ih;
        #pragma unroll
        for (int m = 0; m < 2; ++m) {
            float nh = (n0 * hx + n1 * hy) + n2 * hz, vn = (wox * n0 + woy * n1) + woz * n2, ln = (wix * n0 + wiy * n1) + wiz * n2;
            nh = fmaxf(nh, 0.001f); vn = fmaxf(vn, 0.001f); ln = fmaxf(ln, 0.001f);
            float nh2 = nh * nh, on = 1.0f - nh2;
            float iq = inv<KIND>(vn * ln), ivn = iq * ln, iln = iq * vn;
            float uv = (1.0f - vn * vn) * (ivn * ivn), ul = (1.0f - ln * ln) * (iln * iln);
            float xv = fma_(A, uv, 1.0f), xl = fma_(A, ul, 1.0f);
            float iwv = isq<KIND>(xv), iwl = isq<KIND>(xl);
            float M = (1.0f + xv * iwv) * (1.0f + xl * iwl);
            float den = fmaxf(fma_(nh2, A, on), 0.001f);
            float R = inv<KIND>(M * (3.14159f * den * den));
            float gd = A * R;
            float F = fma_(1.0f - s0, 0.3f, s0);
            float f = fma_(1.0f - F, d0, F * gd * iq);
            acc0 += f * ln; acc1 = fma_(f, hx, acc1); acc2 = fma_(gd, vn, acc2); acc3 = fma_(R, nh, acc3);
            n0 += 1e-6f; n1 -= 1e-6f;
        }
        x += 1e-7f; y -= 1e-7f;
    }
    if (acc0 + acc1 + acc2 + acc3 == 12345.6f) out[0] = acc0;
}
template <int KIND, int UNROLL> void run(const char *name, float *d, int bpc, int instr_per_iter)
{
    const int iters = 4000, blocks = 256 * bpc;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<KIND, UNROLL>), dim3(blocks), dim3(256), 0, 0, d, 10, 1.0f); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); hipLaunchKernelGGL((k<KIND, UNROLL>), dim3(blocks), dim3(256), 0, 0, d, iters, 1.0f); (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1); float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double per_iter_ns = ms * 1e6 / iters / (bpc);
    printf("%-24s waves/SIMD=%d  %8.3f ms  %7.1f ns per wave-iteration per SIMD", name, bpc, ms, per_iter_ns);
    if (instr_per_iter) printf("  = %.2f ns/instr (%d VALU/iter)", per_iter_ns / instr_per_iter, instr_per_iter);
    printf("\n");
}
int main(int argc, char **argv)
{
    float *d; (void)hipMalloc(&d, 4);
    const int n0 = argc > 1 ? atoi(argv[1]) : 0, n1 = argc > 2 ? atoi(argv[2]) : 0;
    for (int bpc : {4, 8}) {
        run<0, 1>("no-trans", d, bpc, n0); run<1, 1>("with-trans", d, bpc, n1);
        run<0, 4>("no-trans x4 unrolled", d, bpc, n0); run<1, 4>("with-trans x4 unrolled", d, bpc, n1);
        run<0, 16>("no-trans x16 unrolled", d, bpc, n0); run<1, 16>("with-trans x16 unrolled", d, bpc, n1);
    }
    return 0;
}
