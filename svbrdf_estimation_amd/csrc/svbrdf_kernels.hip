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
// arithmetic primitives
// ------------------------------------------------------------------------------------------

// torch.sum(a*b, dim=-3): three separately rounded products, summed (p0+p1)+p2
__device__ __forceinline__ float dot3(float ax, float ay, float az, float bx, float by, float bz)
{
    const float p0 = ax * bx, p1 = ay * by, p2 = az * bz;
    return (p0 + p1) + p2;
}

__device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ float rcp_(float x) { return __builtin_amdgcn_rcpf(x); }    // v_rcp_f32, 1 ULP
__device__ __forceinline__ float rsq_(float x) { return __builtin_amdgcn_rsqf(x); }    // v_rsq_f32, 1 ULP

// Correctly rounded a/b for several numerators over ONE denominator: the reciprocal is
// refined once (v_rcp + 2 FMA, Newton) and shared; each quotient then costs a multiply and
// two exact-residual FMA corrections (Markstein).  Same result as the compiler's IEEE
// division sequence (v_div_scale/v_div_fmas/v_div_fixup, ~11 instructions per quotient) for
// operands away from the overflow/denormal range, which holds here: |a| <= ~1e3 and
// 1e-3 <~ b <~ 1e3 (lengths of camera/light offsets).  svbrdf_debug_check_arith() measures
// the agreement with the IEEE `/` on the device.
struct Recip {
    float b, y;
};
__device__ __forceinline__ Recip make_recip(float b)
{
    const float y0 = rcp_(b);
    const float e = fma_(-b, y0, 1.0f);
    return Recip{b, fma_(e, y0, y0)};
}
__device__ __forceinline__ float div_rn(float a, const Recip &r)
{
    float q = a * r.y;
    float e = fma_(-r.b, q, a);
    q = fma_(e, r.y, q);
    e = fma_(-r.b, q, a);
    return fma_(e, r.y, q);
}

// Correctly rounded sqrt for x in the normal range (Markstein: rsq seed, one coupled
// Newton step on (g ~ sqrt x, h ~ 1/(2 sqrt x)), final exact-residual correction).
__device__ __forceinline__ float sqrt_rn(float x)
{
    const float y = rsq_(x);
    float g = x * y;
    float h = 0.5f * y;
    const float r = fma_(-h, g, 0.5f);
    g = fma_(g, r, g);
    h = fma_(h, r, h);
    const float d = fma_(-g, g, x);
    return fma_(d, h, g);
}

// ------------------------------------------------------------------------------------------
// per-pixel device code
// ------------------------------------------------------------------------------------------

struct Geom {           // map-independent, shared by input and target and by the 3 channels
    float wox, woy, woz;
    float wix, wiy, wiz;
    float hx, hy, hz;
    float p;            // (1 - VH)^5               renderers.py:32
    float E[3];         // light_color * falloff     renderers.py:98-100
};

// renderers.py:73-82, 91-93, 45, 49, 99.  `sc` (9 floats) is wave-uniform.  Everything up to
// h reproduces the reference's rounding sequence exactly (see the header of this file).
__device__ __forceinline__ Geom geometry(const float sc[9], float x, float y)
{
    Geom g;
    const float rcx = sc[0] - x, rcy = sc[1] - y, rcz = sc[2];   // z of the patch is 0
    const float rlx = sc[3] - x, rly = sc[4] - y, rlz = sc[5];
    const float lc = sqrt_rn(dot3(rcx, rcy, rcz, rcx, rcy, rcz));
    const float ll = sqrt_rn(dot3(rlx, rly, rlz, rlx, rly, rlz));
    const Recip ic = make_recip(lc), il = make_recip(ll);
    g.wox = div_rn(rcx, ic); g.woy = div_rn(rcy, ic); g.woz = div_rn(rcz, ic);
    g.wix = div_rn(rlx, il); g.wiy = div_rn(rly, il); g.wiz = div_rn(rlz, il);
    const float sx = (g.wix + g.wox) * 0.5f, sy = (g.wiy + g.woy) * 0.5f, sz = (g.wiz + g.woz) * 0.5f;
    const Recip ih = make_recip(sqrt_rn(dot3(sx, sy, sz, sx, sy, sz)));
    g.hx = div_rn(sx, ih); g.hy = div_rn(sy, ih); g.hz = div_rn(sz, ih);
    // from here on the computation is well conditioned: 1-ULP primitives are enough
    const float VH = fmaxf(dot3(g.wox, g.woy, g.woz, g.hx, g.hy, g.hz), kMinDot);
    const float t = 1.0f - VH;
    const float t2 = t * t;
    g.p = (t2 * t2) * t;
    const float fall = rcp_(ll * ll);
#pragma unroll
    for (int k = 0; k < 3; ++k) g.E[k] = sc[6 + k] * fall;
    return g;
}

struct Maps {           // one pixel of a [12,H,W] SVBRDF as stored
    float n[3], d[3], r[3], s[3];
};

struct Grad {           // d/d(maps) of one pixel
    float n[3], d[3], r[3], s[3];
};

struct MapK {           // scene-independent per-pixel constants, hoisted out of the scene loop
    float n[3];
    float A[3];         // r^4 with r = max(r_hat, 1e-3)        renderers.py:87, 23-24
    float s[3], oms[3]; // specular, 1 - specular
    float dpi[3];       // diffuse / pi
    float r4m[3];       // dA/dr_hat = 4 r^3, 0 where r_hat < 1e-3 (clamp mask)
};

template <bool BWD>
__device__ __forceinline__ MapK prepare(const Maps &m)
{
    MapK k;
    constexpr float inv_pi = 1.0f / kPi;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float r = fmaxf(m.r[c], kMinRough);
        const float a = r * r;
        k.n[c] = m.n[c];
        k.A[c] = a * a;
        k.s[c] = m.s[c];
        k.oms[c] = 1.0f - m.s[c];
        k.dpi[c] = m.d[c] * inv_pi;
        if (BWD) k.r4m[c] = (m.r[c] >= kMinRough) ? 4.0f * (a * r) : 0.0f;
    }
    return k;
}

struct Ctx {            // forward values the adjoint needs
    float nh_raw, vn_raw, ln_raw;
    float NH, NH2, uV, uL, iVN, iLN, iq, LNp;
    float F[3], Gp[3], D[3], spec[3], f[3];
    float rV[3], rL[3], iwV[3], iwL[3];
    float ipd2[3], pd[3];
    bool den_on[3];
};

// map-dependent part of render(): renderers.py:43-65, 95-100.
// The three clamped dot products and 1-NH^2 follow the reference's rounding exactly; the
// rest uses v_rcp/v_rsq (1 ULP each) -- the result differs from the op-by-op evaluation
// by a few 1e-7 relative, 50x inside the parity budget.
template <bool KEEP>
__device__ __forceinline__ void shade(const Geom &g, const MapK &m, float rad[3], Ctx &c)
{
    const float nh_raw = dot3(m.n[0], m.n[1], m.n[2], g.hx, g.hy, g.hz);
    const float vn_raw = dot3(g.wox, g.woy, g.woz, m.n[0], m.n[1], m.n[2]);
    const float ln_raw = dot3(g.wix, g.wiy, g.wiz, m.n[0], m.n[1], m.n[2]);
    const float NH = fmaxf(nh_raw, kMinDot), VN = fmaxf(vn_raw, kMinDot), LN = fmaxf(ln_raw, kMinDot);
    const float LNp = fmaxf(ln_raw, 0.0f);
    const float NH2 = NH * NH, VN2 = VN * VN, LN2 = LN * LN;
    const float iN = (1.0f - NH2) * rcp_(NH2);            // (1 - NH^2) / NH^2
    const float iVN = rcp_(VN), iLN = rcp_(LN);
    const float uV = (1.0f - VN2) * (iVN * iVN);          // (1 - VN^2) / VN^2
    const float uL = (1.0f - LN2) * (iLN * iLN);
    const float iq = iVN * iLN;                           // 4/(4 VN LN): the 4 cancels against G1V*G1L
    if (KEEP) {
        c.nh_raw = nh_raw; c.vn_raw = vn_raw; c.ln_raw = ln_raw;
        c.NH = NH; c.NH2 = NH2; c.uV = uV; c.uL = uL; c.iVN = iVN; c.iLN = iLN; c.iq = iq; c.LNp = LNp;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float A = m.A[k];
        const float F = fma_(m.oms[k], g.p, m.s[k]);                  // Schlick
        const float xV = fma_(A, uV, 1.0f), xL = fma_(A, uL, 1.0f);   // 1 + A (1-XN^2)/XN^2
        const float iwV = rsq_(xV), iwL = rsq_(xL);
        const float rV = rcp_(fma_(xV, iwV, 1.0f));                   // 1/(1 + sqrt(xV)) = G1V/2
        const float rL = rcp_(fma_(xL, iwL, 1.0f));
        const float Gp = rV * rL;                                     // G/4
        const float den_raw = NH2 * (A + iN);
        const float den = fmaxf(den_raw, kMinDen);
        const float pd = kPi * den;
        const float ipd2 = rcp_(pd * den);
        const float D = A * ipd2;                                     // GGX
        const float spec = ((F * Gp) * D) * iq;
        const float f = fma_(1.0f - F, m.dpi[k], spec);
        rad[k] = f * (g.E[k] * LNp);
        if (KEEP) {
            c.F[k] = F; c.Gp[k] = Gp; c.D[k] = D; c.spec[k] = spec; c.f[k] = f;
            c.rV[k] = rV; c.rL[k] = rL; c.iwV[k] = iwV; c.iwL[k] = iwL;
            c.ipd2[k] = ipd2; c.pd[k] = pd; c.den_on[k] = den_raw >= kMinDen;
        }
    }
}

// adjoint of shade() with PyTorch's sub-gradient conventions: clamp(min=m) passes the
// gradient iff x >= m (inclusive); xi() has zero gradient (renderers.py:15-16).
__device__ __forceinline__ void shade_bwd(const Geom &g, const MapK &m, const Ctx &c,
                                          const float g_rad[3], Grad &acc)
{
    float g_LNp = 0.0f, sNH = 0.0f, sV = 0.0f, sL = 0.0f, sSp = 0.0f;
    constexpr float inv_pi = 1.0f / kPi;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float gE = g_rad[k] * g.E[k];
        const float g_f = gE * c.LNp;
        g_LNp = fma_(gE, c.f[k], g_LNp);
        const float gfq = g_f * c.iq;
        const float GD = c.Gp[k] * c.D[k];
        const float g_F = fma_(gfq, GD, -(g_f * m.dpi[k]));
        acc.s[k] = fma_(g_F, 1.0f - g.p, acc.s[k]);
        acc.d[k] = fma_(g_f * (1.0f - c.F[k]), inv_pi, acc.d[k]);
        const float gfqF = gfq * c.F[k];
        const float g_Gp = gfqF * c.D[k];          // d/dGp
        const float g_D = gfqF * c.Gp[k];
        sSp = fma_(g_f, c.spec[k], sSp);
        // Gp = rV*rL, rX = 1/(1+wX), wX = sqrt(xX), xX = 1 + A*uX:  dGp/dxV = -Gp*rV/(2 wV)
        const float gG = g_Gp * c.Gp[k];
        const float g_xV = (-0.5f * gG) * (c.rV[k] * c.iwV[k]);
        const float g_xL = (-0.5f * gG) * (c.rL[k] * c.iwL[k]);
        float g_A = fma_(g_xV, c.uV, g_xL * c.uL);
        sV = fma_(g_xV, m.A[k], sV);               // d xV / d uV = A
        sL = fma_(g_xL, m.A[k], sL);
        // D = A/(pi den^2)
        g_A = fma_(g_D, c.ipd2[k], g_A);
        const float iden = c.ipd2[k] * c.pd[k];    // 1/den
        const float g_den = c.den_on[k] ? (-2.0f * g_D) * (c.D[k] * iden) : 0.0f;
        // den_raw = m*(A + (1-m)/m), m = NH^2:  d/dA = m, d/dm = A - 1
        g_A = fma_(g_den, c.NH2, g_A);
        sNH = fma_(g_den, m.A[k] - 1.0f, sNH);
        acc.r[k] = fma_(g_A, m.r4m[k], acc.r[k]);
    }
    // uX = 1/XN^2 - 1: d uX/d XN = -2/XN^3 ; spec ~ 1/(VN LN)
    float g_NH = (sNH * 2.0f) * c.NH;
    float g_VN = -c.iVN * fma_(2.0f * sV, c.iVN * c.iVN, sSp);
    float g_LN = -c.iLN * fma_(2.0f * sL, c.iLN * c.iLN, sSp);
    if (!(c.nh_raw >= kMinDot)) g_NH = 0.0f;
    if (!(c.vn_raw >= kMinDot)) g_VN = 0.0f;
    if (!(c.ln_raw >= kMinDot)) g_LN = 0.0f;
    if (!(c.ln_raw >= 0.0f)) g_LNp = 0.0f;
    const float gl = g_LN + g_LNp;
    acc.n[0] = fma_(g_NH, g.hx, fma_(g_VN, g.wox, fma_(gl, g.wix, acc.n[0])));
    acc.n[1] = fma_(g_NH, g.hy, fma_(g_VN, g.woy, fma_(gl, g.wiy, acc.n[1])));
    acc.n[2] = fma_(g_NH, g.hz, fma_(g_VN, g.woz, fma_(gl, g.wiz, acc.n[2])));
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

// the nine scalars of one render (camera xyz | light xyz | light rgb); the pointer is
// wave-uniform, so these become scalar loads into SGPRs
__device__ __forceinline__ void load_scene(const float *__restrict__ p, float sc[9])
{
#pragma unroll
    for (int i = 0; i < 9; ++i) sc[i] = p[i];
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
    MapK mk[VEC];
    {
        Maps m[VEC];
        load_maps<VEC>(maps + (size_t)b * 12 * plane, plane, pix, m);
#pragma unroll
        for (int v = 0; v < VEC; ++v) mk[v] = prepare<false>(m[v]);
    }
    float x[VEC], y;
    pixel_coords<VEC>(xrow, pix, W, x, y);
    const float *__restrict__ scp = scenes + (size_t)b * S * 9;
    float *__restrict__ o = out + (size_t)b * S * 3 * plane + pix;
    for (int s = 0; s < S; ++s, scp += 9, o += 3 * plane) {
        float sc[9];
        load_scene(scp, sc);
        float rad[VEC][3];
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            const Geom g = geometry(sc, x[v], y);
            Ctx unused;
            shade<false>(g, mk[v], rad[v], unused);
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
    MapK mk[VEC];
    {
        Maps m[VEC];
        load_maps<VEC>(maps + (size_t)b * 12 * plane, plane, pix, m);
#pragma unroll
        for (int v = 0; v < VEC; ++v) mk[v] = prepare<true>(m[v]);
    }
    float x[VEC], y;
    pixel_coords<VEC>(xrow, pix, W, x, y);
    Grad acc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) zero_grad(acc[v]);
    const float *__restrict__ scp = scenes + (size_t)b * S * 9;
    const float *__restrict__ go = grad_out + (size_t)b * S * 3 * plane + pix;
    for (int s = 0; s < S; ++s, scp += 9, go += 3 * plane) {
        float sc[9];
        load_scene(scp, sc);
        float gr[3][VEC];
#pragma unroll
        for (int k = 0; k < 3; ++k) load_vec<VEC>(go + (size_t)k * plane, gr[k]);
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            const Geom g = geometry(sc, x[v], y);
            Ctx c;
            float rad[3];
            shade<true>(g, mk[v], rad, c);
            const float g_rad[3] = {gr[0][v], gr[1][v], gr[2][v]};
            shade_bwd(g, mk[v], c, g_rad, acc[v]);
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
    constexpr float kLn2 = 0.693147180559945309417f;
    const size_t plane = (size_t)H * W;
    const size_t pix = ((size_t)blockIdx.x * kThreads + threadIdx.x) * VEC;
    const int b = blockIdx.y;
    const bool active = pix < plane;
    float lsum = 0.0f;
    if (active) {
        MapK mi[VEC], mt[VEC];
        {
            Maps m[VEC];
            load_maps<VEC>(input + (size_t)b * 12 * plane, plane, pix, m);
#pragma unroll
            for (int v = 0; v < VEC; ++v) mi[v] = prepare<WITH_GRAD>(m[v]);
            load_maps<VEC>(target + (size_t)b * 12 * plane, plane, pix, m);
#pragma unroll
            for (int v = 0; v < VEC; ++v) mt[v] = prepare<false>(m[v]);
        }
        float x[VEC], y;
        pixel_coords<VEC>(xrow, pix, W, x, y);
        Grad acc[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) zero_grad(acc[v]);
        const float *__restrict__ scp = scenes + (size_t)b * S * 9;
        float sc[9], sc_next[9];
        load_scene(scp, sc_next);
        for (int s = 0; s < S; ++s) {
#pragma unroll
            for (int i = 0; i < 9; ++i) sc[i] = sc_next[i];
            // prefetch the next render's scalars (SGPRs) behind this iteration's arithmetic
            scp += (s + 1 < S) ? 9 : 0;
            load_scene(scp, sc_next);
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                const Geom g = geometry(sc, x[v], y);
                Ctx ci, ct;
                float ri[3], rt[3], g_rad[3];
                shade<false>(g, mt[v], rt, ct);
                shade<WITH_GRAD>(g, mi[v], ri, ci);
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    // losses.py:46-50: |log(ri + eps) - log(rt + eps)|, v_log_f32 = log2
                    const float ai = ri[k] + eps, at = rt[k] + eps;
                    const float delta = kLn2 * (__builtin_amdgcn_logf(ai) - __builtin_amdgcn_logf(at));
                    lsum += fabsf(delta);
                    // d|delta|/d ri = sign(delta)/(N*ai), sign(0) = 0 as in torch
                    const float sg = __builtin_amdgcn_fmed3f(delta * 1.0e30f, -1.0f, 1.0f);
                    g_rad[k] = sg * (inv_count * rcp_(ai));
                }
                if (WITH_GRAD) shade_bwd(g, mi[v], ci, g_rad, acc[v]);
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

// data[i] *= *scale, skipped entirely (no memory traffic) when *scale == 1: lets the autograd
// wrapper apply an upstream gradient that lives on the device without a host sync.
__global__ __launch_bounds__(kThreads) void k_scale_inplace(float *__restrict__ data, const float *__restrict__ scale,
                                                            size_t n)
{
    const float s = scale[0];
    if (s == 1.0f) return;
    const size_t stride = (size_t)gridDim.x * kThreads;
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride) data[i] *= s;
}

// ------------------------------------------------------------------------------------------
// arithmetic self-check: div_rn / sqrt_rn against the compiler's IEEE `/` and sqrtf
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned hash32(unsigned x)
{
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

__global__ __launch_bounds__(kThreads) void k_check_arith(unsigned long long n, unsigned seed, float lo, float hi,
                                                          unsigned long long *__restrict__ counts)
{
    unsigned long long bad_div = 0, bad_sqrt = 0;
    const unsigned long long stride = (unsigned long long)gridDim.x * kThreads;
    for (unsigned long long i = (unsigned long long)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride) {
        const unsigned h0 = hash32((unsigned)i * 2654435761U + seed), h1 = hash32(h0 ^ (unsigned)(i >> 32) ^ 0x9e3779b9U);
        // denominators log-uniform in [lo, hi], numerators uniform in [-hi, hi] with random mantissas
        const float u0 = (float)(h0 >> 8) * (1.0f / 16777216.0f), u1 = (float)(h1 >> 8) * (1.0f / 16777216.0f);
        const float b = lo * exp2f(u0 * log2f(hi / lo));
        const float a = (2.0f * u1 - 1.0f) * hi;
        const Recip r = make_recip(b);
        if (div_rn(a, r) != a / b) ++bad_div;
        if (sqrt_rn(b) != sqrtf(b)) ++bad_sqrt;
    }
    if (bad_div) atomicAdd(&counts[0], bad_div);
    if (bad_sqrt) atomicAdd(&counts[1], bad_sqrt);
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

int svbrdf_scale_inplace(float *data, const float *scale_dev, size_t n, void *stream)
{
    if (!data || !scale_dev) return fail(SVBRDF_ERR_NULL, "scale_inplace: null pointer");
    if (n == 0) return 0;
    const size_t blocks = (n + kThreads - 1) / kThreads;
    hipLaunchKernelGGL(k_scale_inplace, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(kThreads), 0,
                       static_cast<hipStream_t>(stream), data, scale_dev, n);
    return launch_status("scale_inplace launch");
}

int svbrdf_debug_check_arith(unsigned long long n, unsigned seed, float lo, float hi,
                             unsigned long long *counts_dev, void *stream)
{
    if (!counts_dev) return fail(SVBRDF_ERR_NULL, "check_arith: null pointer");
    if (!(lo > 0.0f) || !(hi > lo)) return fail(SVBRDF_ERR_DIMS, "check_arith: need 0 < lo < hi");
    hipLaunchKernelGGL(k_check_arith, dim3(2048), dim3(kThreads), 0, static_cast<hipStream_t>(stream), n, seed, lo, hi,
                       counts_dev);
    return launch_status("check_arith launch");
}

}  // extern "C"
