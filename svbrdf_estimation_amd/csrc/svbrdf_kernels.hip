// svbrdf_kernels.hip -- hand-written CDNA4 (gfx950) kernels for the SVBRDF rendering loss
// and the C ABI declared in include/svbrdf_hip.h.
//
// Replaces, for one hot path only (mworchel/svbrdf-estimation, development/multiImage_pytorch/):
//   K1 render_fwd            LocalRenderer.render                  renderers.py:67-104
//   K2 render_bwd            the autograd graph of render()        (66 nodes, ~336 ATen calls)
//   K3 rendering_loss        RenderingLoss.forward + its backward  losses.py:29-52
//
// Build:  hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -shared   (csrc/Makefile)
//
// Numerics contract.  The GGX denominator NH^2*(a^2 + (1-NH^2)/NH^2) (renderers.py:26)
// amplifies a 1-ULP change of NH by 1e3..1e4 next to a highlight, so everything on the path
//     pixel coords -> wo, wi -> h -> NH, VN, LN -> 1-NH^2
// reproduces the reference's fp32 rounding sequence exactly: products rounded one by one
// (-ffp-contract=off, no FMA contraction), dot products summed (p0+p1)+p2 like
// torch.sum(dim=-3), correctly rounded sqrt and division.  Downstream of those values the
// computation is well conditioned.
//
// Data layout.  Maps stay in the reference's BCHW planar layout (W contiguous): lane l of a
// wave owns VEC horizontally adjacent pixels, so each of the 12 planes is read with one
// fully coalesced global_load_dword{,x2,x4} per wave (64*VEC*4 contiguous bytes).  The nine
// scene scalars of a render are uniform per workgroup (a workgroup never straddles batch
// items) and are fetched with scalar loads into SGPRs; all 3-vector math is intra-lane.
// No MFMA: the path is elementwise.

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cmath>

#include "svbrdf_hip.h"

namespace {

constexpr int kThreads = 256;          // 4 waves of 64
constexpr float kPi = 3.14159274101257324219f;  // float32(math.pi), renderers.py:20,27
constexpr float kMinDot = 0.001f;      // renderers.py:48-52
constexpr float kMinRough = 0.001f;    // renderers.py:87
constexpr float kMinDen = 0.001f;      // renderers.py:26

// ------------------------------------------------------------------------------------------
// per-pixel device code
// ------------------------------------------------------------------------------------------

// torch.sum(a*b, dim=-3): three separately rounded products, summed (p0+p1)+p2
__device__ __forceinline__ float dot3(float ax, float ay, float az, float bx, float by, float bz)
{
    const float p0 = ax * bx, p1 = ay * by, p2 = az * bz;
    return (p0 + p1) + p2;
}

struct Geom {
    float wox, woy, woz;
    float wix, wiy, wiz;
    float hx, hy, hz;
    float fall;   // 1 / (sqrt(|L|^2))^2      renderers.py:99
    float p;      // (1 - VH)^5               renderers.py:32
};

// map-independent part of render(): renderers.py:73-82, 91-93, 45, 49, 99.
// `sc` is wave-uniform (scalar loads).
__device__ __forceinline__ Geom geometry(const float *__restrict__ sc, float x, float y)
{
    Geom g;
    const float rcx = sc[0] - x, rcy = sc[1] - y, rcz = sc[2];   // z of the patch is 0
    const float rlx = sc[3] - x, rly = sc[4] - y, rlz = sc[5];
    const float lc = sqrtf(dot3(rcx, rcy, rcz, rcx, rcy, rcz));
    const float d2 = dot3(rlx, rly, rlz, rlx, rly, rlz);
    const float ll = sqrtf(d2);
    g.wox = rcx / lc; g.woy = rcy / lc; g.woz = rcz / lc;
    g.wix = rlx / ll; g.wiy = rly / ll; g.wiz = rlz / ll;
    const float sx = (g.wix + g.wox) * 0.5f, sy = (g.wiy + g.woy) * 0.5f, sz = (g.wiz + g.woz) * 0.5f;
    const float lh = sqrtf(dot3(sx, sy, sz, sx, sy, sz));
    g.hx = sx / lh; g.hy = sy / lh; g.hz = sz / lh;
    const float VH = fmaxf(dot3(g.wox, g.woy, g.woz, g.hx, g.hy, g.hz), kMinDot);
    const float t = 1.0f - VH;
    const float t2 = t * t;
    g.p = (t2 * t2) * t;
    g.fall = 1.0f / (ll * ll);
    return g;
}

struct Maps {           // one pixel of a [12,H,W] SVBRDF
    float n[3], d[3], r[3], s[3];
};

struct Ctx {            // forward values the adjoint needs
    float nh_raw, vn_raw, ln_raw;
    float NH, VN, LN, LNp, uV, uL, q4;
    float r[3], A[3], F[3], G1V[3], G1L[3], wV[3], wL[3], D[3], den[3], den_raw[3];
    float spec[3], f[3], E[3];
};

// map-dependent part of render(): renderers.py:43-65, 87, 95-100
template <bool KEEP>
__device__ __forceinline__ void shade(const Geom &g, const float *__restrict__ col, const Maps &m,
                                      float rad[3], Ctx &c)
{
    const float nh_raw = dot3(m.n[0], m.n[1], m.n[2], g.hx, g.hy, g.hz);
    const float vn_raw = dot3(g.wox, g.woy, g.woz, m.n[0], m.n[1], m.n[2]);
    const float ln_raw = dot3(g.wix, g.wiy, g.wiz, m.n[0], m.n[1], m.n[2]);
    const float NH = fmaxf(nh_raw, kMinDot), VN = fmaxf(vn_raw, kMinDot), LN = fmaxf(ln_raw, kMinDot);
    const float LNp = fmaxf(ln_raw, 0.0f);
    const float NH2 = NH * NH, VN2 = VN * VN, LN2 = LN * LN;
    const float oV = 1.0f - VN2, oL = 1.0f - LN2;
    const float iN = (1.0f - NH2) / NH2;
    const float q4 = (4.0f * VN) * LN;
    if (KEEP) {
        c.nh_raw = nh_raw; c.vn_raw = vn_raw; c.ln_raw = ln_raw;
        c.NH = NH; c.VN = VN; c.LN = LN; c.LNp = LNp; c.q4 = q4;
        c.uV = oV / VN2; c.uL = oL / LN2;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float r = fmaxf(m.r[k], kMinRough);
        const float a = r * r;
        const float A = a * a;
        const float F = m.s[k] + (1.0f - m.s[k]) * g.p;
        const float wV = sqrtf(1.0f + (A * oV) / VN2);
        const float wL = sqrtf(1.0f + (A * oL) / LN2);
        const float G1V = 2.0f / (1.0f + wV);
        const float G1L = 2.0f / (1.0f + wL);
        const float den_raw = NH2 * (A + iN);
        const float den = fmaxf(den_raw, kMinDen);
        const float D = A / (kPi * (den * den));
        const float spec = ((F * (G1V * G1L)) * D) / q4;
        const float f = (((1.0f - F) * m.d[k]) / kPi) + spec;
        const float E = col[k] * g.fall;
        rad[k] = (f * E) * LNp;
        if (KEEP) {
            c.r[k] = r; c.A[k] = A; c.F[k] = F; c.wV[k] = wV; c.wL[k] = wL;
            c.G1V[k] = G1V; c.G1L[k] = G1L; c.den_raw[k] = den_raw; c.den[k] = den;
            c.D[k] = D; c.spec[k] = spec; c.f[k] = f; c.E[k] = E;
        }
    }
}

struct Grad {           // d/d(maps) of one pixel
    float n[3], d[3], r[3], s[3];
};

// adjoint of shade() with PyTorch's sub-gradient conventions: clamp(min=m) passes the
// gradient iff x >= m (inclusive); xi() has zero gradient (renderers.py:15-16).
__device__ __forceinline__ void shade_bwd(const Geom &g, const Ctx &c, const Maps &m,
                                          const float g_rad[3], Grad &acc)
{
    float g_LNp = 0.0f, g_NH = 0.0f, g_VN = 0.0f, g_LN = 0.0f;
    const float NH2 = c.NH * c.NH;
    const float inv_q4 = 1.0f / c.q4;
    const float inv_VN = 1.0f / c.VN, inv_LN = 1.0f / c.LN;
    const float inv_VN3 = (inv_VN * inv_VN) * inv_VN, inv_LN3 = (inv_LN * inv_LN) * inv_LN;
    const float inv_pi = 1.0f / kPi;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float g_f = (g_rad[k] * c.E[k]) * c.LNp;
        const float G = c.G1V[k] * c.G1L[k];
        const float gfq = g_f * inv_q4;
        const float g_F = g_f * ((G * c.D[k]) * inv_q4 - m.d[k] * inv_pi);
        const float g_G = gfq * (c.F[k] * c.D[k]);
        const float g_D = gfq * (c.F[k] * G);
        const float g_sp = g_f * c.spec[k];
        const float g_tV = -(g_G * c.G1L[k]) * (c.G1V[k] * c.G1V[k]) / (4.0f * c.wV[k]);
        const float g_tL = -(g_G * c.G1V[k]) * (c.G1L[k] * c.G1L[k]) / (4.0f * c.wL[k]);
        const float inv_pd2 = 1.0f / (kPi * (c.den[k] * c.den[k]));
        float g_A = (g_tV * c.uV + g_tL * c.uL) + g_D * inv_pd2;
        const float g_den = (c.den_raw[k] >= kMinDen) ? (-2.0f * g_D) * (c.A[k] * inv_pd2) / c.den[k] : 0.0f;
        g_LNp += (g_rad[k] * c.f[k]) * c.E[k];
        acc.d[k] += g_f * ((1.0f - c.F[k]) * inv_pi);
        acc.s[k] += g_F * (1.0f - g.p);
        g_VN += (g_tV * (-2.0f * c.A[k])) * inv_VN3 - g_sp * inv_VN;
        g_LN += (g_tL * (-2.0f * c.A[k])) * inv_LN3 - g_sp * inv_LN;
        g_A += g_den * NH2;
        g_NH += (g_den * (c.A[k] - 1.0f)) * (2.0f * c.NH);
        if (m.r[k] >= kMinRough)
            acc.r[k] += (g_A * 4.0f) * ((c.r[k] * c.r[k]) * c.r[k]);
    }
    if (!(c.nh_raw >= kMinDot)) g_NH = 0.0f;
    if (!(c.vn_raw >= kMinDot)) g_VN = 0.0f;
    if (!(c.ln_raw >= kMinDot)) g_LN = 0.0f;
    if (!(c.ln_raw >= 0.0f)) g_LNp = 0.0f;
    const float gl = g_LN + g_LNp;
    acc.n[0] += (g_NH * g.hx + g_VN * g.wox) + gl * g.wix;
    acc.n[1] += (g_NH * g.hy + g_VN * g.woy) + gl * g.wiy;
    acc.n[2] += (g_NH * g.hz + g_VN * g.woz) + gl * g.wiz;
}

// ------------------------------------------------------------------------------------------
// vector load/store helpers: VEC horizontally adjacent pixels of one plane
// ------------------------------------------------------------------------------------------

template <int VEC> struct VecT;
template <> struct VecT<1> { using type = float; };
template <> struct VecT<2> { using type = float2; };
template <> struct VecT<4> { using type = float4; };

template <int VEC>
__device__ __forceinline__ void load_vec(const float *__restrict__ p, float out[VEC])
{
    using V = typename VecT<VEC>::type;
    const V v = *reinterpret_cast<const V *>(p);
    const float *f = reinterpret_cast<const float *>(&v);
#pragma unroll
    for (int i = 0; i < VEC; ++i) out[i] = f[i];
}

template <int VEC>
__device__ __forceinline__ void store_vec(float *__restrict__ p, const float in[VEC])
{
    using V = typename VecT<VEC>::type;
    V v;
    float *f = reinterpret_cast<float *>(&v);
#pragma unroll
    for (int i = 0; i < VEC; ++i) f[i] = in[i];
    *reinterpret_cast<V *>(p) = v;
}

template <int VEC>
__device__ __forceinline__ void load_maps(const float *__restrict__ base, size_t plane, size_t pix,
                                          Maps m[VEC])
{
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float t[VEC];
        load_vec<VEC>(base + (size_t)(0 + k) * plane + pix, t);
#pragma unroll
        for (int v = 0; v < VEC; ++v) m[v].n[k] = t[v];
        load_vec<VEC>(base + (size_t)(3 + k) * plane + pix, t);
#pragma unroll
        for (int v = 0; v < VEC; ++v) m[v].d[k] = t[v];
        load_vec<VEC>(base + (size_t)(6 + k) * plane + pix, t);
#pragma unroll
        for (int v = 0; v < VEC; ++v) m[v].r[k] = t[v];
        load_vec<VEC>(base + (size_t)(9 + k) * plane + pix, t);
#pragma unroll
        for (int v = 0; v < VEC; ++v) m[v].s[k] = t[v];
    }
}

template <int VEC>
__device__ __forceinline__ void store_grads(float *__restrict__ base, size_t plane, size_t pix,
                                            const Grad g[VEC])
{
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float t[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) t[v] = g[v].n[k];
        store_vec<VEC>(base + (size_t)(0 + k) * plane + pix, t);
#pragma unroll
        for (int v = 0; v < VEC; ++v) t[v] = g[v].d[k];
        store_vec<VEC>(base + (size_t)(3 + k) * plane + pix, t);
#pragma unroll
        for (int v = 0; v < VEC; ++v) t[v] = g[v].r[k];
        store_vec<VEC>(base + (size_t)(6 + k) * plane + pix, t);
#pragma unroll
        for (int v = 0; v < VEC; ++v) t[v] = g[v].s[k];
        store_vec<VEC>(base + (size_t)(9 + k) * plane + pix, t);
    }
}

__device__ __forceinline__ void zero_grad(Grad &g)
{
#pragma unroll
    for (int k = 0; k < 3; ++k) g.n[k] = g.d[k] = g.r[k] = g.s[k] = 0.0f;
}

// pixel (i, j) sits at (xrow[j], -xrow[i], 0): renderers.py:73-76 (needs H == W)
template <int VEC>
__device__ __forceinline__ void pixel_coords(const float *__restrict__ xrow, size_t pix, int W,
                                             float x[VEC], float &y)
{
    const int i = (int)(pix / (size_t)W), j = (int)(pix % (size_t)W);
    load_vec<VEC>(xrow + j, x);
    y = -xrow[i];
}

// ------------------------------------------------------------------------------------------
// K1: render forward.  grid = (ceil(H*W / (256*VEC)), B); S renders per map in one pass.
// ------------------------------------------------------------------------------------------
template <int VEC>
__global__ __launch_bounds__(kThreads) void k_render_fwd(const float *__restrict__ maps,
                                                         const float *__restrict__ scenes,
                                                         const float *__restrict__ xrow,
                                                         float *__restrict__ out, int S, int H, int W)
{
    const size_t plane = (size_t)H * W;
    const size_t pix = ((size_t)blockIdx.x * kThreads + threadIdx.x) * VEC;
    const int b = blockIdx.y;
    if (pix >= plane) return;
    Maps m[VEC];
    load_maps<VEC>(maps + (size_t)b * 12 * plane, plane, pix, m);
    float x[VEC], y;
    pixel_coords<VEC>(xrow, pix, W, x, y);
    const float *__restrict__ sc = scenes + (size_t)b * S * 9;
    float *__restrict__ o = out + (size_t)b * S * 3 * plane + pix;
    for (int s = 0; s < S; ++s, sc += 9, o += 3 * plane) {
        float rad[VEC][3];
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            const Geom g = geometry(sc, x[v], y);
            Ctx unused;
            shade<false>(g, sc + 6, m[v], rad[v], unused);
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float t[VEC];
#pragma unroll
            for (int v = 0; v < VEC; ++v) t[v] = rad[v][k];
            store_vec<VEC>(o + (size_t)k * plane, t);
        }
    }
}

// ------------------------------------------------------------------------------------------
// K2: render backward.  Same grid; the S scenes of one map are accumulated in-thread
// (no atomics), the forward is recomputed in registers.
// ------------------------------------------------------------------------------------------
template <int VEC>
__global__ __launch_bounds__(kThreads) void k_render_bwd(const float *__restrict__ maps,
                                                         const float *__restrict__ scenes,
                                                         const float *__restrict__ xrow,
                                                         const float *__restrict__ grad_out,
                                                         float *__restrict__ grad_maps, int S, int H, int W)
{
    const size_t plane = (size_t)H * W;
    const size_t pix = ((size_t)blockIdx.x * kThreads + threadIdx.x) * VEC;
    const int b = blockIdx.y;
    if (pix >= plane) return;
    Maps m[VEC];
    load_maps<VEC>(maps + (size_t)b * 12 * plane, plane, pix, m);
    float x[VEC], y;
    pixel_coords<VEC>(xrow, pix, W, x, y);
    Grad acc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) zero_grad(acc[v]);
    const float *__restrict__ sc = scenes + (size_t)b * S * 9;
    const float *__restrict__ go = grad_out + (size_t)b * S * 3 * plane + pix;
    for (int s = 0; s < S; ++s, sc += 9, go += 3 * plane) {
        float gr[3][VEC];
#pragma unroll
        for (int k = 0; k < 3; ++k) load_vec<VEC>(go + (size_t)k * plane, gr[k]);
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            const Geom g = geometry(sc, x[v], y);
            Ctx c;
            float rad[3];
            shade<true>(g, sc + 6, m[v], rad, c);
            const float g_rad[3] = {gr[0][v], gr[1][v], gr[2][v]};
            shade_bwd(g, c, m[v], g_rad, acc[v]);
        }
    }
    store_grads<VEC>(grad_maps + (size_t)b * 12 * plane, plane, pix, acc);
}

// ------------------------------------------------------------------------------------------
// K3: fused rendering loss, forward + backward in one pass over the maps.
//   per pixel: read 12 input + 12 target planes once, loop the S scenes in registers
//   (geometry shared by input and target), accumulate |dlog| and the 12 map gradients,
//   write 12 gradient planes once  -> 144 B/pixel of HBM traffic, independent of S.
//   Loss: per-thread fp32 sum -> wave shuffle -> LDS -> one partial per workgroup;
//   k_loss_finalize sums the partials in a fixed order in fp64 (deterministic).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

template <int VEC, bool WITH_GRAD>
__global__ __launch_bounds__(kThreads) void k_rendering_loss(const float *__restrict__ input,
                                                             const float *__restrict__ target,
                                                             const float *__restrict__ scenes,
                                                             const float *__restrict__ xrow, float eps,
                                                             float inv_count, float *__restrict__ grad_input,
                                                             float *__restrict__ partials, int S, int H, int W)
{
    __shared__ float wave_part[kThreads / 64];
    const size_t plane = (size_t)H * W;
    const size_t pix = ((size_t)blockIdx.x * kThreads + threadIdx.x) * VEC;
    const int b = blockIdx.y;
    const bool active = pix < plane;
    float lsum = 0.0f;
    if (active) {
        Maps mi[VEC], mt[VEC];
        load_maps<VEC>(input + (size_t)b * 12 * plane, plane, pix, mi);
        load_maps<VEC>(target + (size_t)b * 12 * plane, plane, pix, mt);
        float x[VEC], y;
        pixel_coords<VEC>(xrow, pix, W, x, y);
        Grad acc[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) zero_grad(acc[v]);
        const float *__restrict__ sc = scenes + (size_t)b * S * 9;
        for (int s = 0; s < S; ++s, sc += 9) {
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                const Geom g = geometry(sc, x[v], y);
                Ctx ci, ct;
                float ri[3], rt[3], g_rad[3];
                shade<WITH_GRAD>(g, sc + 6, mi[v], ri, ci);
                shade<false>(g, sc + 6, mt[v], rt, ct);
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const float ai = ri[k] + eps, at = rt[k] + eps;   // losses.py:46-48
                    const float delta = logf(ai) - logf(at);
                    lsum += fabsf(delta);                              // losses.py:50 (L1)
                    const float sg = (delta > 0.0f) ? inv_count : ((delta < 0.0f) ? -inv_count : 0.0f);
                    g_rad[k] = sg / ai;
                }
                if (WITH_GRAD) shade_bwd(g, ci, mi[v], g_rad, acc[v]);
            }
        }
        if (WITH_GRAD) store_grads<VEC>(grad_input + (size_t)b * 12 * plane, plane, pix, acc);
    }
    lsum = wave_sum(lsum);
    if ((threadIdx.x & 63) == 0) wave_part[threadIdx.x >> 6] = lsum;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.0f;
#pragma unroll
        for (int w = 0; w < kThreads / 64; ++w) t += wave_part[w];
        partials[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = t;
    }
}

// one workgroup: fixed-order fp64 sum of the per-workgroup partials -> mean
__global__ __launch_bounds__(kThreads) void k_loss_finalize(const float *__restrict__ partials, int n,
                                                            double inv_count, float *__restrict__ loss_out)
{
    __shared__ double red[kThreads];
    double t = 0.0;
    for (int i = threadIdx.x; i < n; i += kThreads) t += (double)partials[i];
    red[threadIdx.x] = t;
    __syncthreads();
    for (int off = kThreads / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss_out[0] = (float)(red[0] * inv_count);
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------

thread_local char g_err[256] = "";

int fail(int code, const char *what)
{
    std::snprintf(g_err, sizeof(g_err), "%s", what);
    return code;
}

int check_dims(int B, int S, int H, int W)
{
    if (B <= 0 || S <= 0 || H <= 0 || W <= 0) return fail(SVBRDF_ERR_DIMS, "B, S, H, W must be positive");
    if (H != W) return fail(SVBRDF_ERR_DIMS, "H must equal W (renderers.py:75 transposes the x grid)");
    if (B > 65535) return fail(SVBRDF_ERR_DIMS, "B exceeds 65535 (grid.y)");
    if ((long long)H * W > (1LL << 30)) return fail(SVBRDF_ERR_DIMS, "H*W too large");
    return 0;
}

bool aligned(const void *p, size_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; }

// widest vector width (<= want) every pointer and the row length allow
int pick_vec(int want, int W, std::initializer_list<const void *> ptrs)
{
    int vec = want;
    while (vec > 1) {
        bool ok = (W % vec) == 0;
        for (const void *p : ptrs) ok = ok && (p == nullptr || aligned(p, sizeof(float) * vec));
        if (ok) break;
        vec >>= 1;
    }
    return vec;
}

int env_vec(const char *name, int dflt)
{
    const char *e = std::getenv(name);
    if (!e) return dflt;
    const int v = std::atoi(e);
    return (v == 1 || v == 2 || v == 4) ? v : dflt;
}

int launch_status(const char *what)
{
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        std::snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}

dim3 grid_for(int B, int H, int W, int vec)
{
    const long long plane = (long long)H * W;
    const long long per_block = (long long)kThreads * vec;
    return dim3((unsigned)((plane + per_block - 1) / per_block), (unsigned)B, 1);
}

}  // namespace

extern "C" {

int svbrdf_abi_version(void) { return SVBRDF_ABI_VERSION; }

const char *svbrdf_last_error(void) { return g_err; }

int svbrdf_make_xrow(float *xrow_host, int W)
{
    if (!xrow_host) return fail(SVBRDF_ERR_NULL, "xrow_host is null");
    if (W <= 0) return fail(SVBRDF_ERR_DIMS, "W must be positive");
    if (W == 1) { xrow_host[0] = -1.0f; return 0; }
    // torch.linspace(-1, 1, W) on the reference's CPU path: symmetric halves, one FMA each
    const float step = 2.0f / (float)(W - 1);
    for (int i = 0; i < W; ++i)
        xrow_host[i] = (i < W / 2) ? std::fmaf(step, (float)i, -1.0f) : std::fmaf(-step, (float)(W - 1 - i), 1.0f);
    return 0;
}

int svbrdf_render_fwd(const float *maps, const float *scenes, const float *xrow, float *out,
                      int B, int S, int H, int W, void *stream)
{
    if (!maps || !scenes || !xrow || !out) return fail(SVBRDF_ERR_NULL, "render_fwd: null pointer");
    if (int e = check_dims(B, S, H, W)) return e;
    if (!aligned(maps, 4) || !aligned(scenes, 4) || !aligned(xrow, 4) || !aligned(out, 4))
        return fail(SVBRDF_ERR_ALIGN, "render_fwd: pointers must be 4-byte aligned");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int vec = pick_vec(env_vec("SVBRDF_K1_VEC", 4), W, {maps, xrow, out});
    const dim3 grid = grid_for(B, H, W, vec), block(kThreads);
    if (vec == 4) hipLaunchKernelGGL(k_render_fwd<4>, grid, block, 0, st, maps, scenes, xrow, out, S, H, W);
    else if (vec == 2) hipLaunchKernelGGL(k_render_fwd<2>, grid, block, 0, st, maps, scenes, xrow, out, S, H, W);
    else hipLaunchKernelGGL(k_render_fwd<1>, grid, block, 0, st, maps, scenes, xrow, out, S, H, W);
    return launch_status("render_fwd launch");
}

int svbrdf_render_bwd(const float *maps, const float *scenes, const float *xrow, const float *grad_out,
                      float *grad_maps, int B, int S, int H, int W, void *stream)
{
    if (!maps || !scenes || !xrow || !grad_out || !grad_maps) return fail(SVBRDF_ERR_NULL, "render_bwd: null pointer");
    if (int e = check_dims(B, S, H, W)) return e;
    if (!aligned(maps, 4) || !aligned(scenes, 4) || !aligned(xrow, 4) || !aligned(grad_out, 4) || !aligned(grad_maps, 4))
        return fail(SVBRDF_ERR_ALIGN, "render_bwd: pointers must be 4-byte aligned");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int vec = pick_vec(env_vec("SVBRDF_K2_VEC", 2), W, {maps, xrow, grad_out, grad_maps});
    const dim3 grid = grid_for(B, H, W, vec), block(kThreads);
    if (vec == 4) hipLaunchKernelGGL(k_render_bwd<4>, grid, block, 0, st, maps, scenes, xrow, grad_out, grad_maps, S, H, W);
    else if (vec == 2) hipLaunchKernelGGL(k_render_bwd<2>, grid, block, 0, st, maps, scenes, xrow, grad_out, grad_maps, S, H, W);
    else hipLaunchKernelGGL(k_render_bwd<1>, grid, block, 0, st, maps, scenes, xrow, grad_out, grad_maps, S, H, W);
    return launch_status("render_bwd launch");
}

size_t svbrdf_rendering_loss_workspace_bytes(int B, int S, int H, int W)
{
    if (B <= 0 || S <= 0 || H <= 0 || W <= 0) return 0;
    // one fp32 partial per workgroup; sized for the narrowest vector width (most workgroups)
    const dim3 g = grid_for(B, H, W, 1);
    return (size_t)g.x * g.y * sizeof(float);
}

int svbrdf_rendering_loss_fwd_bwd(const float *input, const float *target, const float *scenes,
                                  const float *xrow, float eps, float *loss_out, float *grad_input,
                                  void *workspace, size_t workspace_bytes, int B, int S, int H, int W,
                                  void *stream)
{
    if (!input || !target || !scenes || !xrow || !loss_out || !workspace)
        return fail(SVBRDF_ERR_NULL, "rendering_loss: null pointer");
    if (int e = check_dims(B, S, H, W)) return e;
    if (!aligned(input, 4) || !aligned(target, 4) || !aligned(scenes, 4) || !aligned(xrow, 4) ||
        !aligned(loss_out, 4) || !aligned(workspace, 4) || (grad_input && !aligned(grad_input, 4)))
        return fail(SVBRDF_ERR_ALIGN, "rendering_loss: pointers must be 4-byte aligned");
    if (workspace_bytes < svbrdf_rendering_loss_workspace_bytes(B, S, H, W))
        return fail(SVBRDF_ERR_WORKSPACE, "rendering_loss: workspace too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int vec = pick_vec(env_vec("SVBRDF_K3_VEC", 1), W, {input, target, xrow, grad_input});
    const dim3 grid = grid_for(B, H, W, vec), block(kThreads);
    const double count = (double)B * S * 3.0 * (double)H * (double)W;
    const float inv_count = (float)(1.0 / count);
    float *partials = static_cast<float *>(workspace);
#define SVBRDF_LAUNCH_K3(V)                                                                               \
    do {                                                                                                  \
        if (grad_input)                                                                                   \
            hipLaunchKernelGGL((k_rendering_loss<V, true>), grid, block, 0, st, input, target, scenes,    \
                               xrow, eps, inv_count, grad_input, partials, S, H, W);                      \
        else                                                                                              \
            hipLaunchKernelGGL((k_rendering_loss<V, false>), grid, block, 0, st, input, target, scenes,   \
                               xrow, eps, inv_count, grad_input, partials, S, H, W);                      \
    } while (0)
    if (vec == 4) SVBRDF_LAUNCH_K3(4);
    else if (vec == 2) SVBRDF_LAUNCH_K3(2);
    else SVBRDF_LAUNCH_K3(1);
#undef SVBRDF_LAUNCH_K3
    if (int e = launch_status("rendering_loss launch")) return e;
    hipLaunchKernelGGL(k_loss_finalize, dim3(1), block, 0, st, partials, (int)(grid.x * grid.y), 1.0 / count, loss_out);
    return launch_status("loss_finalize launch");
}

}  // extern "C"
