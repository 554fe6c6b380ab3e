// tools/k3_asm_bench.cpp -- times k_rendering_loss<GRAD> (device-table variant) from code objects given on the command
// line (.hsaco files assembled from edited compiler output): research harness for the register-assignment question
// (DESIGN.md section 8).  Not product code.   hipcc -O2 tools/k3_asm_bench.cpp -o tools/_build/k3_asm_bench
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
struct L1Params { float a, b, c; };
static unsigned rng_state = 12345u;
static float urand() { rng_state = rng_state * 1664525u + 1013904223u; return (float)(rng_state >> 8) * (1.0f / 16777216.0f); }
static float grand() { float u = urand() + 1e-7f, v = urand(); return std::sqrt(-2.0f * std::log(u)) * std::cos(6.2831853f * v); }
int main(int argc, char **argv)
{
    const int B = 8, H = 256, W = 256, S = std::getenv("K3_S") ? std::atoi(std::getenv("K3_S")) : 9, reps = 30;
    const size_t plane = (size_t)H * W, n = (size_t)B * 12 * plane;
    std::vector<float> in(n), tg(n), sc((size_t)B * S * 9), xr(W);
    for (int which = 0; which < 2; ++which) {
        std::vector<float> &m = which ? tg : in;
        for (int b = 0; b < B; ++b)
            for (size_t p = 0; p < plane; ++p) {
                float nx = 0.3f * grand(), ny = 0.3f * grand(), nz = 1.0f + std::fabs(0.3f * grand());
                const float il = 1.0f / std::sqrt(nx * nx + ny * ny + nz * nz);
                float *q = &m[(size_t)b * 12 * plane + p];
                q[0 * plane] = nx * il; q[1 * plane] = ny * il; q[2 * plane] = nz * il;
                const float r = urand();
                const bool untied = std::getenv("K3_UNTIED") != nullptr;      // independent roughness channels: three-lobe path
                for (int k = 0; k < 3; ++k) { q[(3 + k) * plane] = urand(); q[(6 + k) * plane] = untied ? urand() : r; q[(9 + k) * plane] = urand(); }
            }
    }
    for (size_t i = 0; i < sc.size() / 9; ++i) {
        float *q = &sc[i * 9];
        const float r1 = std::sqrt(0.001f + 0.899f * urand()), ph = 6.2831853f * urand(), d = 0.8f + 2.0f * urand();
        q[0] = r1 * std::cos(ph) * d; q[1] = r1 * std::sin(ph) * d; q[2] = std::sqrt(1 - r1 * r1) * d + 1e-3f;
        const float r2 = std::sqrt(0.001f + 0.899f * urand()), p2 = 6.2831853f * urand(), d2 = 0.8f + 2.0f * urand();
        q[3] = r2 * std::cos(p2) * d2; q[4] = r2 * std::sin(p2) * d2; q[5] = std::sqrt(1 - r2 * r2) * d2 + 1e-3f;
        q[6] = q[7] = q[8] = 20.0f;
    }
    for (int i = 0; i < W; ++i) xr[i] = -1.0f + 2.0f * i / (W - 1);
    float *d_in, *d_tg, *d_sc, *d_xr, *d_grad, *d_loss; unsigned long long *d_ws;
    CK(hipMalloc(&d_in, n * 4)); CK(hipMalloc(&d_tg, n * 4)); CK(hipMalloc(&d_grad, n * 4)); CK(hipMalloc(&d_sc, sc.size() * 4));
    CK(hipMalloc(&d_xr, W * 4)); CK(hipMalloc(&d_loss, 4)); CK(hipMalloc(&d_ws, 65 * 8)); CK(hipMemset(d_ws, 0, 65 * 8));
    CK(hipMemcpy(d_in, in.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_tg, tg.data(), n * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_sc, sc.data(), sc.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_xr, xr.data(), W * 4, hipMemcpyHostToDevice));
    const double count = (double)B * S * 3.0 * (double)plane;
    float eps = 0.1f, inv_count = (float)(1.0 / count), fixed_scale = 16777216.0f; double loss_scale = std::ldexp(1.0, -24) / count;
    L1Params l1{0.0f, 0.0f, 0.01f};
    int iS = S, iH = H, iW = W;
    void *args[] = {&d_in, &d_tg, &d_sc, &d_xr, &eps, &inv_count, &loss_scale, &fixed_scale, &l1, &d_grad, &d_ws, &d_loss, &iS, &iH, &iW};
    const char *kname = "_ZN12_GLOBAL__N_116k_rendering_lossILb1ELb0ELb0EEEvPKfS2_S2_S2_ffdfNS_8L1ParamsEPfPyS4_iii";
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int rounds = std::getenv("K3_ROUNDS") ? std::atoi(std::getenv("K3_ROUNDS")) : 2;
    std::vector<std::vector<float>> times(argc);
    std::vector<float> losses(argc, 0.0f);
    for (int round = 0; round < rounds; ++round)
        for (int a = 1; a < argc; ++a) {
            hipModule_t mod; hipFunction_t fn;
            if (hipModuleLoad(&mod, argv[a]) != hipSuccess || hipModuleGetFunction(&fn, mod, kname) != hipSuccess) { times[a].push_back(-1); continue; }
            for (int i = 0; i < 5; ++i) CK(hipModuleLaunchKernel(fn, (unsigned)(plane / 256), B, 1, 256, 1, 1, 0, 0, args, nullptr));
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            for (int i = 0; i < reps; ++i) CK(hipModuleLaunchKernel(fn, (unsigned)(plane / 256), B, 1, 256, 1, 1, 0, 0, args, nullptr));
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            times[a].push_back(ms * 1e3f / reps);
            CK(hipMemcpy(&losses[a], d_loss, 4, hipMemcpyDeviceToHost));
            CK(hipModuleUnload(mod));
        }
    for (int a = 1; a < argc; ++a) {
        std::sort(times[a].begin(), times[a].end());
        std::printf("%-40s %8.2f us (min of %d)   loss %.7f\n", argv[a], times[a].empty() ? -1.0f : times[a][0], rounds, losses[a]);
    }
    return 0;
}
