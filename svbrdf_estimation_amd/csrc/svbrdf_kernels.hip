// svbrdf_kernels.hip -- hand-written CDNA4 (gfx950) kernels for the SVBRDF rendering loss
// and the C ABI declared in include/svbrdf_hip.h.
//
// Replaces, for one hot path only (mworchel/svbrdf-estimation, development/multiImage_pytorch/):
//   K1 render_fwd            LocalRenderer.render                  renderers.py:67-104
//   K2 render_bwd            the autograd graph of render()        (66 nodes, ~336 ATen calls)
//   K3 rendering_loss        RenderingLoss.forward + its backward  losses.py:29-52
//      (+ SVBRDFL1Loss / MixedLoss, losses.py:7-19, 54-63, and the network head, models.py:338-346, folded in)
//   K4 mix_materials         SvbrdfDataset.mix                     dataset.py:142-160
//   (float64 maps and second order -- auxiliary, not on the north-star path -- live in svbrdf_aux_f64.hip)
//
// Build:  hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -fPIC -shared   (csrc/Makefile)
//
// Numerics contract.  The GGX denominator NH^2*(a^2 + (1-NH^2)/NH^2) (renderers.py:26)
// amplifies a 1-ULP change of NH by 1e3..1e4 next to a highlight, so everything on the path
//     pixel coords -> wo, wi -> h -> NH, VN, LN -> 1-NH^2
// reproduces the reference's fp32 rounding sequence exactly: products rounded one by one
// (-ffp-contract=off, no FMA contraction), dot products summed (p0+p1)+p2 like
// torch.sum(dim=-3), correctly rounded sqrt and division.  Downstream of those values -- and
// for the three dot products that do not feed NH (n.wo, n.wi, wo.h) -- the
// computation is well conditioned and uses explicit FMAs and 1-ULP primitives.
//
// Data layout.  Maps stay in the reference's BCHW planar layout (W contiguous): lane l of a
// wave owns VEC horizontally adjacent pixels (K1/K2: VEC in {1,2,4}; K3: one pixel), so each
// of the 12 planes is read with one fully coalesced global_load_dword{,x2,x4} per wave
// (64*VEC*4 contiguous bytes).  The nine scene scalars of a render are uniform per workgroup
// (a workgroup never straddles batch items): K1/K2 read them with scalar loads, K3 keeps the
// row of the render in flight in VGPRs (an SGPR operand costs 1.75x on a VALU-bound kernel,
// tools/valu_bank.hip) and stages the table in LDS for the forward-only variants.  All
// 3-vector math is intra-lane.  No MFMA: the path is elementwise.

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>

#include "svbrdf_hip.h"

// Translation units.  The order the compiler leaves the instructions in moves these VALU-bound kernels by up to 20 %
// at an identical instruction mix, and no single set of LLVM scheduler options suits every kernel (measured grid:
// csrc/Makefile, profiles/r02_sched_grid.txt).  The Makefile therefore compiles this file three times and links the
// objects (and the auxiliary unit svbrdf_aux_f64.hip) into libsvbrdf_hip.so:
//   SVBRDF_TU=0  everything except the forward+adjoint loss kernels        (bottom-up post-RA scheduler)
//   SVBRDF_TU=1  the forward+adjoint RenderingLoss kernels <GRAD, L1=0, HEAD=0> and their launcher
//   SVBRDF_TU=3  the forward+adjoint MixedLoss / head-fused kernels (L1 or HEAD set) and their launcher
//                (both: no post-RA scheduler, iterative-minreg strategy, round 1's register-assignment options;
//                separate units so that each can take its own set again when the optimum moves)
//   SVBRDF_TU=4  none of the kernels or entry points of this file: svbrdf_aux_f64.hip (float64 maps, second order:
//                auxiliary, not on the north-star path) includes it for the shared inline device code and host checks
// SVBRDF_TU=2 (default: a one-command build of the file): units 0, 1 and 3 in one.
#ifndef SVBRDF_TU
#define SVBRDF_TU 2
#endif
#define SVBRDF_TU_MAIN (SVBRDF_TU == 0 || SVBRDF_TU == 2)
#define SVBRDF_TU_ADJOINT_PLAIN (SVBRDF_TU == 1 || SVBRDF_TU == 2)
#define SVBRDF_TU_ADJOINT_EXTRA (SVBRDF_TU == 3 || SVBRDF_TU == 2)
#define SVBRDF_TU_AUX (SVBRDF_TU == 4)      // svbrdf_aux_f64.hip includes this file for the shared inline code only

namespace {

constexpr int kThreads = 256;          // 4 waves of 64 (K1, K2)
constexpr int kLossThreads = 256;      // K3 workgroup (64 and 128 threads: A/B builds of round 4, both slower)
constexpr float kPi = 3.14159274101257324219f;  // float32(math.pi), renderers.py:20,27
constexpr float kMinDot = 0.001f;      // renderers.py:48-52
constexpr float kMinRough = 0.001f;    // renderers.py:87

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
__device__ __forceinline__ float log2_(float x) { return __builtin_amdgcn_logf(x); }   // v_log_f32

// Correctly rounded a/b for several numerators over ONE denominator: the reciprocal is
// refined once (v_rcp + 2 FMA, Newton) and shared; each quotient then costs a multiply and
// one exact-residual FMA correction (Markstein).  The compiler's IEEE division sequence
// (v_div_scale/v_div_fmas/v_div_fixup) costs ~11 instructions per quotient.  Valid for
// operands away from the overflow/denormal range, which holds here: |a| <= ~1e3 and
// 1e-3 <~ b <~ 1e3 (lengths of camera/light offsets).  svbrdf_debug_check_arith() compares
// both primitives with the IEEE-correct `/` and sqrtf on the device: 0 mismatches in
// 3 x 2^31 operand pairs (tests/test_gpu_parity.py).
// (The timing-only ablation builds of round 1 -- inexact geometry, no target shading, no logs, no adjoint, no lobe, no loads,
// no stores, transcendentals as multiplies: profiles/r01_k3_ablation.txt -- and the per-term log form of the loss are no
// longer in the source; git show 3d7c7a1 has them.)
#ifndef SVBRDF_SCALE_BLOCKS
#define SVBRDF_SCALE_BLOCKS 1024
#endif
#ifndef SVBRDF_TIMING
#define SVBRDF_TIMING 0            // 1: timing-only build (per-wave stamps instead of gradients), see tools/k3_timeline.py
#endif
// The exact-rounding contract (header of this file) covers the path into NH -- coords, wo, wi, h, n.h, 1 - NH^2 -- because the
// GGX denominator amplifies NH's last bit by 1e3..1e4.  n.wo, n.wi and wo.h feed only well-conditioned terms (Smith G, the
// cosine factor, Fresnel): a last-bit difference there moves a radiance by ~1e-7 relative, a hundredth of the 1e-5 bound.
// They are one multiply + two FMAs instead of the reference's three rounded products and two adds: -10 VALU per render
// (K3 at config 2: -2 % time, same-box A/B profiles/r04_k3_ab_algebra.txt).
// The 1/pi of the diffuse gradient and dA/dr_hat = 4 r^3 of the roughness gradient are per-pixel constants: applied once
// after the scene loop instead of once per render and channel (-6 VALU per render; -1.5 % time in the same A/B).
// f = (1-F) d/pi + F GD (renderers.py:18-20, 62-65) evaluated as d/pi + F (GD - d/pi): the difference is shared with the
// adjoint's d f/d F, -6 VALU per render (-0.5 ... -1.6 % time, profiles/r04_k3_ab_lerp.txt); both maps use the same form,
// so identical maps still give a loss of exactly 0.
#ifndef SVBRDF_K3_MIN_WAVES
#define SVBRDF_K3_MIN_WAVES 4      // waves/SIMD the register allocator must leave room for (128 VGPRs)
#endif

struct Recip {
    float b, y;
};
__device__ __forceinline__ float div_rn(float a, const Recip &r)
{
    const float q = a * r.y;
    return fma_(fma_(-r.b, q, a), r.y, q);
}

// len = correctly rounded sqrt(x) for x in the normal range -- rsq seed y (1 ULP), g = x*y,
// one exact-residual correction g + (x - g*g)*y/2 -- together with the refined reciprocal
// of len (Newton step from the same seed, no extra v_rcp) for the divisions that follow.
// `seed` returns y ~ 1/sqrt(x) for well-conditioned uses.
__device__ __forceinline__ Recip length_rn(float x, float &seed)
{
    const float y = rsq_(x);
    const float g = x * y;
    seed = y;
    const float len = fma_(fma_(-g, g, x), 0.5f * y, g);
    const float e = fma_(-len, y, 1.0f);
    return Recip{len, fma_(e, y, y)};
}

// Measured on gfx950 (tools/valu_bank.hip): a VALU instruction with an SGPR or 32-bit literal
// source issues ~1.75x slower than the same instruction on VGPR sources.  The handful of
// constants the inner loop uses over and over therefore live in VGPRs (the empty asm makes
// them opaque so the compiler cannot fold them back into literals).
struct VConst {
    float tiny;     // 0.001: the clamps of renderers.py:26 (GGX denominator), 48-52 (dots), 87 (roughness)
    float pi, inv_pi;
    float ln2;
    float huge;     // 1e30, for sign()
};
__device__ __forceinline__ float vreg(float c)
{
    asm volatile("" : "+v"(c));
    return c;
}
__device__ __forceinline__ VConst make_vconst()
{
    return VConst{vreg(kMinDot), vreg(kPi), vreg(1.0f / kPi), vreg(0.693147180559945309417f), vreg(1.0e30f)};
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
__device__ __forceinline__ Geom geometry(const VConst &K, const float sc[9], float x, float y)
{
    Geom g;
    const float rcx = sc[0] - x, rcy = sc[1] - y, rcz = sc[2];   // z of the patch is 0
    const float rlx = sc[3] - x, rly = sc[4] - y, rlz = sc[5];
    float yc, yl, yh;
    const Recip ic = length_rn(dot3(rcx, rcy, rcz, rcx, rcy, rcz), yc);
    const Recip il = length_rn(dot3(rlx, rly, rlz, rlx, rly, rlz), yl);
    g.wox = div_rn(rcx, ic); g.woy = div_rn(rcy, ic); g.woz = div_rn(rcz, ic);
    g.wix = div_rn(rlx, il); g.wiy = div_rn(rly, il); g.wiz = div_rn(rlz, il);
    // h = normalize((wi + wo)/2) (renderers.py:45): the halving is exact and commutes with every rounding
    // that follows (products scale by 4, the correctly rounded sqrt by 2, the quotients not at all), so
    // normalize(wi + wo) gives the same bits without the three multiplies
    const float sx = g.wix + g.wox, sy = g.wiy + g.woy, sz = g.wiz + g.woz;
    const Recip ih = length_rn(dot3(sx, sy, sz, sx, sy, sz), yh);
    g.hx = div_rn(sx, ih); g.hy = div_rn(sy, ih); g.hz = div_rn(sz, ih);
    // from here on the computation is well conditioned: 1-ULP primitives are enough
    // (wo.h feeds only the Fresnel factor: no exact dot product needed)
    const float VH = fmaxf(fma_(g.wox, g.hx, fma_(g.woy, g.hy, g.woz * g.hz)), K.tiny);
    const float t = 1.0f - VH;
    const float t2 = t * t;
    g.p = (t2 * t2) * t;
    const float fall = yl * yl;                       // 1/|L|^2 (renderers.py:99), rsq seed squared
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
    float oA[3];        // 1 - A
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
        k.oA[c] = 1.0f - k.A[c];
        k.s[c] = m.s[c];
        k.oms[c] = 1.0f - m.s[c];
        k.dpi[c] = m.d[c] * inv_pi;
        if (BWD) k.r4m[c] = (m.r[c] >= kMinRough) ? 4.0f * (a * r) : 0.0f;
    }
    return k;
}

// The network emits ONE roughness channel repeated three times (utils.py:78-80
// decode_svbrdf) and the Deschaintre data stores grey roughness, so the three colour
// channels usually share alpha -- and with it the whole GGX D and Smith G evaluation.
__device__ __forceinline__ bool tied_roughness(const Maps &m) { return m.r[0] == m.r[1] && m.r[1] == m.r[2]; }

struct Dots {           // clamped dot products of one (pixel, scene, map set)
    float nh_raw, vn_raw, ln_raw;
    float NH, NH2, oN;  // NH^2 and 1 - NH^2, rounded exactly like the reference's
    float VN, LN;       // clamp(n.wo, 1e-3), clamp(n.wi, 1e-3)
    float VN2, LN2;     // their squares
    float LNp;          // clamp(n.wi, 0)                       renderers.py:96
};

// renderers.py:48-52, 96.  The three dot products and 1-NH^2 follow the reference's
// rounding exactly; everything derived from them is well conditioned.
__device__ __forceinline__ Dots dots(const VConst &K, const Geom &g, const MapK &m)
{
    Dots d;
    d.nh_raw = dot3(m.n[0], m.n[1], m.n[2], g.hx, g.hy, g.hz);
    d.vn_raw = fma_(g.wox, m.n[0], fma_(g.woy, m.n[1], g.woz * m.n[2]));      // well conditioned: one multiply + two FMAs
    d.ln_raw = fma_(g.wix, m.n[0], fma_(g.wiy, m.n[1], g.wiz * m.n[2]));
    d.NH = fmaxf(d.nh_raw, K.tiny);
    d.VN = fmaxf(d.vn_raw, K.tiny);
    d.LN = fmaxf(d.ln_raw, K.tiny);
    d.LNp = fmaxf(d.ln_raw, 0.0f);
    d.NH2 = d.NH * d.NH;
    d.oN = 1.0f - d.NH2;
    d.VN2 = d.VN * d.VN;
    d.LN2 = d.LN * d.LN;
    return d;
}

// Everything of the specular term that depends on the material only through alpha^2 = A, with the
// 1/(4 VN LN) of renderers.py:62 folded in.  Smith's G1(X) = 2/(1 + sqrt(1 + A (1-XN^2)/XN^2))
// (renderers.py:34-41) equals 2 XN/(XN + sX) with sX = sqrt(XN^2 (1-A) + A), so
//   GD = G D/(4 VN LN) = A / (pi den^2 PV PL),   PX = XN + sX,
//   den = NH^2 (A + (1-NH^2)/NH^2) = NH^2 A + (1-NH^2)   (renderers.py:22-27; one rounding instead of three)
// -- the division by VN LN cancels, which saves its reciprocal (a transcendental costs ~7 plain
// VALU slots in this kernel) and everything that was derived from it.  All terms are sums of
// non-negative numbers: no cancellation.  For the adjoint, per unit of d(loss)/d(GD):
//   KA = dGD/dA (den clamp mask applied), KV/KL = dGD/dVN, dGD/dLN, KN = dGD/d(NH^2).
struct Lobe {
    float GD, KA, KV, KL, KN;
};

template <bool BWD>
__device__ __forceinline__ Lobe lobe(const VConst &K, float A, float oA, const Dots &d)
{
    Lobe l;
    const float yV = fma_(d.VN2, oA, A), yL = fma_(d.LN2, oA, A);
    // sqrt as y*rsq(y) in the forward-only and the forward+backward instantiation alike, so
    // that input and target shading are the SAME arithmetic (identical maps -> loss exactly 0)
    const float iV = rsq_(yV), iL = rsq_(yL);
    const float sV = yV * iV, sL = yL * iL;
    const float PV = d.VN + sV, PL = d.LN + sL;
    const float PP = PV * PL;                           // 4 VN LN / G
    const float den_raw = fma_(d.NH2, A, d.oN);
    const float den = fmaxf(den_raw, K.tiny);           // renderers.py:26 clamp
    const float pd = K.pi * den;
    const float Q = pd * den;                           // A/D
    const float R = rcp_(PP * Q);
    l.GD = A * R;
    if (BWD) {
        const float RQ = R * Q;                         // 1/PP
        const float tV = (RQ * PL) * l.GD;              // GD/PV
        const float tL = (RQ * PV) * l.GD;              // GD/PL
        // dPX/dXN = 1 + XN (1-A)/sX,  dPX/dA = (1 - XN^2)/(2 sX),  1/sX = iX
        l.KV = -tV * fma_(d.VN * oA, iV, 1.0f);
        l.KL = -tL * fma_(d.LN * oA, iL, 1.0f);
        // dGD/dden = -2 GD/den, 1/den = (R PP) pd; zero where the clamp is active
        const float Kden = (den_raw >= K.tiny) ? (-2.0f * l.GD) * ((R * PP) * pd) : 0.0f;
        const float hV = (1.0f - d.VN2) * iV, hL = (1.0f - d.LN2) * iL;
        l.KA = fma_(-0.5f, fma_(tV, hV, tL * hL), fma_(Kden, d.NH2, R));
        l.KN = -Kden * oA;                              // d den_raw/d(NH^2) = A - 1
    }
    return l;
}

// radiance of one pixel under one scene: renderers.py:43-65, 95-100.  NL = 3: one lobe per
// colour channel (independent roughness channels); NL = 1: tied roughness, one lobe.
template <int NL, bool BWD>
__device__ __forceinline__ void shade(const VConst &K, const Geom &g, const MapK &m, const Dots &d, Lobe lb[NL],
                                      float F[3], float f[3], float rad[3])
{
#pragma unroll
    for (int l = 0; l < NL; ++l) lb[l] = lobe<BWD>(K, m.A[l], m.oA[l], d);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        F[k] = fma_(m.oms[k], g.p, m.s[k]);                               // Schlick, renderers.py:29-32
        // f = (1-F) d/pi + F GD (renderers.py:18-20, 62-65; GD carries the 1/(4 VN LN)) as d/pi + F (GD - d/pi)
        f[k] = fma_(F[k], lb[NL == 3 ? k : 0].GD - m.dpi[k], m.dpi[k]);
        rad[k] = f[k] * (g.E[k] * d.LNp);
    }
}

// adjoint of shade() with PyTorch's sub-gradient conventions: clamp(min=m) passes the
// gradient iff x >= m (inclusive); xi() has zero gradient (renderers.py:15-16).
// DEFER_D / DEFER_R: acc.d accumulates g_f (1-F) without the 1/pi and acc.r accumulates gGD KA without dA/dr_hat; the
// caller applies the two per-pixel constants once after its scene loop (apply_deferred_scales).
template <int NL, bool DEFER_D = false, bool DEFER_R = false>
__device__ __forceinline__ void shade_bwd(const VConst &K, const Geom &g, const MapK &m, const Dots &d, const Lobe lb[NL],
                                          const float F[3], const float f[3], const float g_rad[3], Grad &acc)
{
    // (sums start from their first term, not from 0: `0 + x` is an instruction the compiler must keep -- it turns a
    // -0 into +0 -- and this code is bound by the instruction count)
    float g_LNp, W[NL];
    const float inv_pi = K.inv_pi;
    const float omp = 1.0f - g.p;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const Lobe &l = lb[NL == 3 ? k : 0];
        const float gE = g_rad[k] * g.E[k];
        const float g_f = gE * d.LNp;
        g_LNp = k == 0 ? gE * f[k] : fma_(gE, f[k], g_LNp);
        const float g_F = g_f * (l.GD - m.dpi[k]);                        // f = d/pi + F (GD - d/pi)
        acc.s[k] = fma_(g_F, omp, acc.s[k]);
        acc.d[k] = DEFER_D ? fma_(g_f, 1.0f - F[k], acc.d[k]) : fma_(g_f * (1.0f - F[k]), inv_pi, acc.d[k]);
        const float gGD = g_f * F[k];                                     // d loss/d GD
        acc.r[k] = DEFER_R ? fma_(gGD, l.KA, acc.r[k]) : fma_(gGD * l.KA, m.r4m[k], acc.r[k]);
        if (NL == 3 || k == 0) W[NL == 3 ? k : 0] = gGD;
        else W[0] += gGD;
    }
    float g_VN = W[0] * lb[0].KV, g_LN = W[0] * lb[0].KL, sN = W[0] * lb[0].KN;
#pragma unroll
    for (int l = 1; l < NL; ++l) {
        g_VN = fma_(W[l], lb[l].KV, g_VN);
        g_LN = fma_(W[l], lb[l].KL, g_LN);
        sN = fma_(W[l], lb[l].KN, sN);
    }
    float g_NH = (sN * 2.0f) * d.NH;
    if (!(d.nh_raw >= K.tiny)) g_NH = 0.0f;
    if (!(d.vn_raw >= K.tiny)) g_VN = 0.0f;
    if (!(d.ln_raw >= K.tiny)) g_LN = 0.0f;
    if (!(d.ln_raw >= 0.0f)) g_LNp = 0.0f;
    const float gl = g_LN + g_LNp;
    acc.n[0] = fma_(g_NH, g.hx, fma_(g_VN, g.wox, fma_(gl, g.wix, acc.n[0])));
    acc.n[1] = fma_(g_NH, g.hy, fma_(g_VN, g.woy, fma_(gl, g.wiy, acc.n[1])));
    acc.n[2] = fma_(g_NH, g.hz, fma_(g_VN, g.woz, fma_(gl, g.wiz, acc.n[2])));
}

// ------------------------------------------------------------------------------------------
// vector load/store helpers: VEC horizontally adjacent pixels of one plane
// ------------------------------------------------------------------------------------------

typedef float vec2f __attribute__((ext_vector_type(2)));
typedef float vec4f __attribute__((ext_vector_type(4)));
template <int VEC> struct VecT;
template <> struct VecT<1> { using type = float; };
template <> struct VecT<2> { using type = vec2f; };
template <> struct VecT<4> { using type = vec4f; };

// NT = non-temporal (streaming) access.  Measured on the HBM-bound stand-alone renders (288 renders per
// launch, 1.1 / 2.0 GB): K1 5.55 -> 6.45 TB/s with nt LOADS (its stores stay cached for the consumer),
// K2 5.9 -> 6.2 TB/s with nt loads AND stores; no effect on the VALU-bound fused loss.
template <int VEC, bool NT = false>
__device__ __forceinline__ void load_vec(const float *__restrict__ p, float out[VEC])
{
    using V = typename VecT<VEC>::type;
    const V v = NT ? __builtin_nontemporal_load(reinterpret_cast<const V *>(p)) : *reinterpret_cast<const V *>(p);
    const float *f = reinterpret_cast<const float *>(&v);
#pragma unroll
    for (int i = 0; i < VEC; ++i) out[i] = f[i];
}

template <int VEC, bool NT = false>
__device__ __forceinline__ void store_vec(float *__restrict__ p, const float in[VEC])
{
    using V = typename VecT<VEC>::type;
    V v;
    float *f = reinterpret_cast<float *>(&v);
#pragma unroll
    for (int i = 0; i < VEC; ++i) f[i] = in[i];
    if (NT) __builtin_nontemporal_store(v, reinterpret_cast<V *>(p));
    else *reinterpret_cast<V *>(p) = v;
}

template <int VEC, bool NT = false>
__device__ __forceinline__ void load_maps(const float *__restrict__ base, size_t plane, size_t pix,
                                          Maps m[VEC])
{
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float t[VEC];
        load_vec<VEC, NT>(base + (size_t)(0 + k) * plane + pix, t);
#pragma unroll
        for (int v = 0; v < VEC; ++v) m[v].n[k] = t[v];
        load_vec<VEC, NT>(base + (size_t)(3 + k) * plane + pix, t);
#pragma unroll
        for (int v = 0; v < VEC; ++v) m[v].d[k] = t[v];
        load_vec<VEC, NT>(base + (size_t)(6 + k) * plane + pix, t);
#pragma unroll
        for (int v = 0; v < VEC; ++v) m[v].r[k] = t[v];
        load_vec<VEC, NT>(base + (size_t)(9 + k) * plane + pix, t);
#pragma unroll
        for (int v = 0; v < VEC; ++v) m[v].s[k] = t[v];
    }
}

template <int VEC, bool NT = false>
__device__ __forceinline__ void store_grads(float *__restrict__ base, size_t plane, size_t pix,
                                            const Grad g[VEC])
{
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float t[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) t[v] = g[v].n[k];
        store_vec<VEC, NT>(base + (size_t)(0 + k) * plane + pix, t);
#pragma unroll
        for (int v = 0; v < VEC; ++v) t[v] = g[v].d[k];
        store_vec<VEC, NT>(base + (size_t)(3 + k) * plane + pix, t);
#pragma unroll
        for (int v = 0; v < VEC; ++v) t[v] = g[v].r[k];
        store_vec<VEC, NT>(base + (size_t)(6 + k) * plane + pix, t);
#pragma unroll
        for (int v = 0; v < VEC; ++v) t[v] = g[v].s[k];
        store_vec<VEC, NT>(base + (size_t)(9 + k) * plane + pix, t);
    }
}

// K3's 36 plane accesses (one pixel per lane): the plane base is wave-uniform (kernel argument + batch item + channel),
// the lane contributes a 32-bit pixel index.  Buffer instructions express exactly that: one 128-bit resource per tensor
// and batch item in SGPRs (base, byte size), the lane's byte offset in ONE VGPR computed once, the plane's byte offset
// as the scalar offset operand -- all address arithmetic on the scalar unit.  With flat 64-bit addresses the compiler
// spends ~100 VALU instructions per pixel on v_mad_u64_u32 / v_lshl_add_u64 pairs, in a kernel that is bound by VALU
// issue.  One item's planes must stay below 2 GiB (checked on the host: H*W <= 2^25).
#ifndef SVBRDF_K3_STORE_AUX
// Cache policy of K3's gradient stores: sc0 sc1 = write-through.  A launch cannot end before its dirty lines have left
// the eight XCDs' L2s; with plain stores the 25 MB of gradients of config 2 pile up there and the release at the end of
// the kernel costs ~1.4 us (same-box A/B, profiles/r04_k3_ab.txt: 38.2 -> 36.8 us per launch one at a time, equal
// with two launches in flight; nt alone gains 1.1).  A/B builds: 0 = plain, 2 = nt, 17 = sc0 sc1, 19 = all three.
#define SVBRDF_K3_STORE_AUX 17
#endif
#ifndef SVBRDF_K3_LOAD_AUX
#define SVBRDF_K3_LOAD_AUX 0        // cache policy of K3's plane loads (A/B builds: 2 = nt)
#endif
struct PlaneBuf {
    __amdgpu_buffer_rsrc_t rsrc;
    unsigned lane_bytes;        // pixel index * 4
    unsigned plane_bytes;       // H*W*4
};
__device__ __forceinline__ PlaneBuf plane_buf(const float *item_base, int planes, size_t plane, size_t pix)
{
    PlaneBuf p;
    p.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(item_base), 0, (int)(planes * plane * sizeof(float)),
                                               0x00020000);     // gfx9 raw dword buffer
    p.lane_bytes = (unsigned)pix * 4u;
    p.plane_bytes = (unsigned)plane * 4u;
    return p;
}
__device__ __forceinline__ float plane_load(const PlaneBuf &p, int k)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(p.rsrc, p.lane_bytes, k * p.plane_bytes, SVBRDF_K3_LOAD_AUX));
}
__device__ __forceinline__ void plane_store(const PlaneBuf &p, int k, float v)
{
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), p.rsrc, p.lane_bytes, k * p.plane_bytes,
                                          SVBRDF_K3_STORE_AUX);
}

__device__ __forceinline__ void load_maps_k3(const float *__restrict__ base, size_t plane, size_t pix, Maps &m)
{
    const PlaneBuf p = plane_buf(base, 12, plane, pix);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        m.n[k] = plane_load(p, 0 + k);
        m.d[k] = plane_load(p, 3 + k);
        m.r[k] = plane_load(p, 6 + k);
        m.s[k] = plane_load(p, 9 + k);
    }
}

__device__ __forceinline__ void store_grads_k3(float *__restrict__ base, size_t plane, size_t pix, const Grad &g)
{
    const PlaneBuf p = plane_buf(base, 12, plane, pix);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        plane_store(p, 0 + k, g.n[k]);
        plane_store(p, 3 + k, g.d[k]);
        plane_store(p, 6 + k, g.r[k]);
        plane_store(p, 9 + k, g.s[k]);
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

// scene rows passed BY VALUE in a launch's kernel-argument block (see k_rendering_loss_inl)
struct SceneBlock {
    float v[SVBRDF_HOST_SCENES_MAX_ROWS * 9];
};

// ------------------------------------------------------------------------------------------
// Sensor-noise epilogue of the input-photo synthesis (dataset.py:215-217: rendering + N(0, sigma_image), clamp [0,1]),
// fused into K1's store.  The reference draws the field from torch's CPU generator; here it is counter-based:
// Philox4x32-10 (Salmon et al., SC'11; the generator behind torch's device randn) keyed with the caller's 64-bit seed,
// counter = (group index lo, hi, offset lo, hi) where group = linear index of the OUTPUT element / 4.  One call gives four
// uniforms -> two Box-Muller pairs -> the four normals of elements 4g .. 4g+3, so the field is a pure function of
// (seed, offset, element index): independent of the vector width the launch picked, reproducible, and a VEC = 4 lane
// (four adjacent pixels of one plane row, 16-byte aligned) spends exactly one Philox call per plane it stores.
// ------------------------------------------------------------------------------------------
struct PhiloxKey {
    unsigned k0, k1;        // seed
    unsigned c2, c3;        // offset (upper counter words)
};

__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1,
                                              unsigned out[4])
{
#pragma unroll
    for (int round = 0; round < 10; ++round) {
        // one full 32 x 32 -> 64 multiply per product (v_mad_u64_u32): integer multiplies are quarter rate, and a
        // separate v_mul_hi_u32 + v_mul_lo_u32 pair would cost two of them
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
        const unsigned hi0 = (unsigned)(p0 >> 32), lo0 = (unsigned)p0, hi1 = (unsigned)(p1 >> 32), lo1 = (unsigned)p1;
        c0 = hi1 ^ c1 ^ k0;
        c1 = lo1;
        c2 = hi0 ^ c3 ^ k1;
        c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// four standard normals of output-element group g: u = (top 24 bits + 0.5) / 2^24 in (0,1);
// (n0, n1) = sqrt(-2 ln u0) (cos, sin)(2 pi u1), (n2, n3) likewise from u2, u3.  v_sin/v_cos take revolutions.
[[maybe_unused]] __device__ __forceinline__ void normal4(const PhiloxKey &key, unsigned long long g, float n[4])
{
    unsigned r[4];
    philox4x32_10((unsigned)g, (unsigned)(g >> 32), key.c2, key.c3, key.k0, key.k1, r);
    constexpr float kInv24 = 1.0f / 16777216.0f;
    constexpr float kM2Ln2 = -1.38629436111989061883f;      // -2 ln 2: -2 ln u = kM2Ln2 * log2 u
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const float u0 = fma_((float)(r[2 * h] >> 8), kInv24, 0.5f * kInv24);
        const float u1 = fma_((float)(r[2 * h + 1] >> 8), kInv24, 0.5f * kInv24);
        const float rad = __builtin_amdgcn_sqrtf(kM2Ln2 * log2_(u0));
        n[2 * h] = rad * __builtin_amdgcn_cosf(u1);
        n[2 * h + 1] = rad * __builtin_amdgcn_sinf(u1);
    }
}

// EPI: what happens to a radiance on its way to memory.  0 = stored as computed (LocalRenderer.render);
// 1 = clamp to [0,1]; 2 = + sigma[render] * N(0,1), then clamp to [0,1] (render_inputs, dataset.py:215-217)
struct Epilogue {
    const float *__restrict__ sigma;    // per render, wave-uniform (device table or kernel-argument block)
    PhiloxKey key;
};

// `elem` = linear index in `out` of t[0].  A lane's VEC values are adjacent elements of one plane row starting at a multiple
// of VEC, and a plane holds H*W = W*W elements with W a multiple of VEC (pick_vec), so they always share ONE group of four:
// one Philox call per lane and stored plane whatever the width.
template <int VEC, int EPI>
__device__ __forceinline__ void finish_radiance(const Epilogue &ep, float sig, unsigned long long elem, float t[VEC])
{
    if constexpr (EPI == 2) {
        float n[4];
        normal4(ep.key, elem >> 2, n);
        if constexpr (VEC == 4) {
#pragma unroll
            for (int v = 0; v < 4; ++v) t[v] = fma_(sig, n[v], t[v]);
        } else if constexpr (VEC == 2) {
            const bool upper = (elem & 2ull) != 0;
            t[0] = fma_(sig, upper ? n[2] : n[0], t[0]);
            t[1] = fma_(sig, upper ? n[3] : n[1], t[1]);
        } else {
            const unsigned lane = (unsigned)elem & 3u;
            const float lo = (lane & 1u) ? n[1] : n[0], hi = (lane & 1u) ? n[3] : n[2];
            t[0] = fma_(sig, (lane & 2u) ? hi : lo, t[0]);
        }
    }
    if constexpr (EPI != 0) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) t[v] = t[v] < 0.0f ? 0.0f : (t[v] > 1.0f ? 1.0f : t[v]);   // torch.clamp(min=0, max=1): a NaN stays a NaN
    }
}

// ------------------------------------------------------------------------------------------
// K1: render forward.  grid = (ceil(H*W / (256*VEC)), B); S renders per map in one pass.
// ------------------------------------------------------------------------------------------
template <int VEC, int NL, int EPI = 0>
__device__ __forceinline__ void render_fwd_loop(const MapK mk[VEC], const float x[VEC], float y,
                                                const float *__restrict__ scp, float *__restrict__ o,
                                                size_t plane, int S, [[maybe_unused]] const Epilogue &ep,
                                                [[maybe_unused]] unsigned long long elem)
{
    const VConst K = make_vconst();
    for (int s = 0; s < S; ++s, scp += 9, o += 3 * plane) {
        float sc[9];
        load_scene(scp, sc);
        float rad[VEC][3];
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            const Geom g = geometry(K, sc, x[v], y);
            const Dots d = dots(K, g, mk[v]);
            Lobe lb[NL];
            float F[3], f[3];
            shade<NL, false>(K, g, mk[v], d, lb, F, f, rad[v]);
        }
        [[maybe_unused]] float sig = 0.0f;
        if constexpr (EPI == 2) sig = ep.sigma[s];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float t[VEC];
#pragma unroll
            for (int v = 0; v < VEC; ++v) t[v] = rad[v][k];
            if constexpr (EPI != 0) finish_radiance<VEC, EPI>(ep, sig, elem + (unsigned long long)(3 * s + k) * plane, t);
            store_vec<VEC>(o + (size_t)k * plane, t);
        }
    }
}

// `offsets` (or null): ragged form, the renders of map b are rows offsets[b] .. offsets[b+1] of scenes / out
// (any number per map, zero included) instead of the regular S per map.  `shared`: ONE list of S scenes for every
// map (rows 0 .. S-1 of `scenes`) instead of S rows per map -- LocalRenderer.render's "one scene, B maps"
// (renderers.py:98) without materialising B copies of the row.
template <int VEC, int EPI = 0>
__device__ __forceinline__ void render_fwd_body(const float *__restrict__ maps, const float *__restrict__ scenes,
                                                const float *__restrict__ xrow, float *__restrict__ out,
                                                const int *__restrict__ offsets, bool shared, int S, int H, int W,
                                                Epilogue ep = Epilogue{nullptr, PhiloxKey{0, 0, 0, 0}})
{
    const size_t plane = (size_t)H * W;
    const size_t pix = ((size_t)blockIdx.x * kThreads + threadIdx.x) * VEC;
    const int b = blockIdx.y;
    if (pix >= plane) return;
    MapK mk[VEC];
    bool tied = true;
    {
        Maps m[VEC];
        load_maps<VEC, true>(maps + (size_t)b * 12 * plane, plane, pix, m);
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            mk[v] = prepare<false>(m[v]);
            tied = tied && tied_roughness(m[v]);
        }
    }
    float x[VEC], y;
    pixel_coords<VEC>(xrow, pix, W, x, y);
    size_t first = (size_t)b * S;
    if (offsets) {
        first = (size_t)offsets[b];
        S = offsets[b + 1] - offsets[b];
    }
    const float *__restrict__ scp = scenes + (shared ? 0 : first * 9);
    float *__restrict__ o = out + first * 3 * plane + pix;
    const unsigned long long elem = (unsigned long long)first * 3ull * plane + pix;    // linear index of o[0] in `out`
    if constexpr (EPI == 2) ep.sigma += first;                                 // one noise level per render, like the scenes
    if (__all(tied)) render_fwd_loop<VEC, 1, EPI>(mk, x, y, scp, o, plane, S, ep, elem);      // wave-uniform branch
    else render_fwd_loop<VEC, 3, EPI>(mk, x, y, scp, o, plane, S, ep, elem);
}

template <int VEC>
__global__ __launch_bounds__(kThreads) void k_render_fwd(const float *__restrict__ maps,
                                                         const float *__restrict__ scenes,
                                                         const float *__restrict__ xrow,
                                                         float *__restrict__ out, const int *__restrict__ offsets,
                                                         int shared, int S, int H, int W)
{
    render_fwd_body<VEC>(maps, scenes, xrow, out, offsets, shared != 0, S, H, W);
}

// scene rows BY VALUE in the kernel-argument block (see SceneBlock below: the table is the first argument and is
// read through the kernarg segment pointer with the same wave-uniform scalar loads).  A reference-shaped
// `render(scene, svbrdf)` call (renderers.py:67-104: three synchronous H2D copies per call, :79,91,98) is then ONE
// dispatch with no copy command in front of it.
template <int VEC>
__global__ __launch_bounds__(kThreads) void k_render_fwd_inl([[maybe_unused]] const SceneBlock table,
                                                             const float *__restrict__ maps,
                                                             const float *__restrict__ xrow, float *__restrict__ out,
                                                             int shared, int S, int H, int W)
{
    const float *__restrict__ rows = (const float *)__builtin_amdgcn_kernarg_segment_ptr();
    render_fwd_body<VEC>(maps, rows, xrow, out, nullptr, shared != 0, S, H, W);
}

// K1 with the sensor-noise epilogue (render_inputs, dataset.py:206-219): EPI = 1 clamp, 2 noise + clamp.  The noise levels
// (one per render) come from a device table, or ride behind the scene rows in the argument block.
#ifndef SVBRDF_PHOTO_MIN_WAVES
#define SVBRDF_PHOTO_MIN_WAVES 4    // the Philox state next to four pixels' shading: 137 VGPRs (3 waves/SIMD) left to itself
#endif
#define SVBRDF_PHOTO_ATTRS __attribute__((amdgpu_waves_per_eu(SVBRDF_PHOTO_MIN_WAVES, 8)))
struct PhotoBlock {
    float v[SVBRDF_HOST_SCENES_MAX_ROWS * 9];
    float sigma[SVBRDF_HOST_SCENES_MAX_ROWS];
};

template <int VEC, int EPI>
__global__ __launch_bounds__(kThreads) SVBRDF_PHOTO_ATTRS void k_render_inputs(const float *__restrict__ maps,
                                                            const float *__restrict__ scenes,
                                                            const float *__restrict__ sigma, PhiloxKey key,
                                                            const float *__restrict__ xrow, float *__restrict__ out,
                                                            int S, int H, int W)
{
    render_fwd_body<VEC, EPI>(maps, scenes, xrow, out, nullptr, false, S, H, W, Epilogue{sigma, key});
}

template <int VEC, int EPI>
__global__ __launch_bounds__(kThreads) SVBRDF_PHOTO_ATTRS void k_render_inputs_inl([[maybe_unused]] const PhotoBlock table, PhiloxKey key,
                                                                const float *__restrict__ maps,
                                                                const float *__restrict__ xrow, float *__restrict__ out,
                                                                int S, int H, int W)
{
    const float *__restrict__ rows = (const float *)__builtin_amdgcn_kernarg_segment_ptr();
    render_fwd_body<VEC, EPI>(maps, rows, xrow, out, nullptr, false, S, H, W,
                              Epilogue{rows + SVBRDF_HOST_SCENES_MAX_ROWS * 9, key});
}

// ------------------------------------------------------------------------------------------
// K2: render backward.  Same grid; the S scenes of one map are accumulated in-thread
// (no atomics), the forward is recomputed in registers.
// ------------------------------------------------------------------------------------------
template <int VEC, int NL>
__device__ __forceinline__ void render_bwd_loop(const MapK mk[VEC], const float x[VEC], float y,
                                                const float *__restrict__ scp, const float *__restrict__ go,
                                                size_t plane, int S, Grad acc[VEC])
{
    const VConst K = make_vconst();
    for (int s = 0; s < S; ++s, scp += 9, go += 3 * plane) {
        float sc[9];
        load_scene(scp, sc);
        float gr[3][VEC];
#pragma unroll
        for (int k = 0; k < 3; ++k) load_vec<VEC, true>(go + (size_t)k * plane, gr[k]);
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            const Geom g = geometry(K, sc, x[v], y);
            const Dots d = dots(K, g, mk[v]);
            Lobe lb[NL];
            float F[3], f[3], rad[3];
            shade<NL, true>(K, g, mk[v], d, lb, F, f, rad);
            const float g_rad[3] = {gr[0][v], gr[1][v], gr[2][v]};
            shade_bwd<NL>(K, g, mk[v], d, lb, F, f, g_rad, acc[v]);
        }
    }
}

template <int VEC>
__device__ __forceinline__ void render_bwd_body(const float *__restrict__ maps, const float *__restrict__ scenes,
                                                const float *__restrict__ xrow, const float *__restrict__ grad_out,
                                                float *__restrict__ grad_maps, const int *__restrict__ offsets,
                                                bool shared, int S, int H, int W)
{
    const size_t plane = (size_t)H * W;
    const size_t pix = ((size_t)blockIdx.x * kThreads + threadIdx.x) * VEC;
    const int b = blockIdx.y;
    if (pix >= plane) return;
    MapK mk[VEC];
    bool tied = true;
    {
        Maps m[VEC];
        load_maps<VEC, true>(maps + (size_t)b * 12 * plane, plane, pix, m);
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            mk[v] = prepare<true>(m[v]);
            tied = tied && tied_roughness(m[v]);
        }
    }
    float x[VEC], y;
    pixel_coords<VEC>(xrow, pix, W, x, y);
    Grad acc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) zero_grad(acc[v]);
    size_t first = (size_t)b * S;
    if (offsets) {
        first = (size_t)offsets[b];
        S = offsets[b + 1] - offsets[b];
    }
    const float *__restrict__ scp = scenes + (shared ? 0 : first * 9);
    const float *__restrict__ go = grad_out + first * 3 * plane + pix;
    if (__all(tied)) render_bwd_loop<VEC, 1>(mk, x, y, scp, go, plane, S, acc);
    else render_bwd_loop<VEC, 3>(mk, x, y, scp, go, plane, S, acc);
    store_grads<VEC, true>(grad_maps + (size_t)b * 12 * plane, plane, pix, acc);
}

template <int VEC>
__global__ __launch_bounds__(kThreads) void k_render_bwd(const float *__restrict__ maps,
                                                         const float *__restrict__ scenes,
                                                         const float *__restrict__ xrow,
                                                         const float *__restrict__ grad_out,
                                                         float *__restrict__ grad_maps, const int *__restrict__ offsets,
                                                         int shared, int S, int H, int W)
{
    render_bwd_body<VEC>(maps, scenes, xrow, grad_out, grad_maps, offsets, shared != 0, S, H, W);
}

template <int VEC>
__global__ __launch_bounds__(kThreads) void k_render_bwd_inl([[maybe_unused]] const SceneBlock table,
                                                             const float *__restrict__ maps,
                                                             const float *__restrict__ xrow,
                                                             const float *__restrict__ grad_out,
                                                             float *__restrict__ grad_maps, int shared, int S, int H, int W)
{
    const float *__restrict__ rows = (const float *)__builtin_amdgcn_kernarg_segment_ptr();
    render_bwd_body<VEC>(maps, rows, xrow, grad_out, grad_maps, nullptr, shared != 0, S, H, W);
}

// ------------------------------------------------------------------------------------------
// K3: fused rendering loss, forward + backward in one pass over the maps.
//   per pixel: read 12 input + 12 target planes once, loop the S scenes in registers
//   (geometry shared by input and target), accumulate |dlog| and the 12 map gradients,
//   write 12 gradient planes once  -> 144 B/pixel of HBM traffic, independent of S.
//   Loss: per-thread fp32 sum -> wave shuffle -> LDS -> one value per workgroup, added as a
//   fixed-point integer to one of kLossSlots 64-bit accumulators (integer addition is
//   associative, so the result is bitwise reproducible whatever the arrival order).  Each
//   accumulator word also counts its arrivals in its top 16 bits, so ONE returning atomic
//   per workgroup both adds and tells the workgroup whether it completed its slot; slot
//   completers draw a global ticket (<= kLossSlots atomics in all) and the last of them sums
//   the slots, writes the mean and re-zeroes the scratch.  (A single shared accumulator +
//   ticket serialises 2 atomics per workgroup on one address and cost ~45 us at 2048
//   workgroups -- measured; profiles/r01_k3_sweep.txt.)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

constexpr int kLossSlots = 64;                    // sharded accumulators (power of two)
constexpr int kLossCountShift = 48;               // word = arrivals << 48 | fixed-point sum
constexpr unsigned long long kLossSumMask = (1ULL << kLossCountShift) - 1;
constexpr unsigned long long kLossTicketMask = 0xffffffffULL;   // ws[kLossSlots]: low half counts slot completions,
constexpr unsigned long long kLossNonFiniteFlag = 1ULL << 32;   // bit 32 = some workgroup's partial sum was not finite

// one (pixel, scene) of the fused loss: both shadings, log/L1, adjoint of the input shading
template <int NL, bool WITH_GRAD, int DEFER = 0>
__device__ __forceinline__ void loss_pixel_scene(const VConst &K, const Geom &g, const MapK &mi,
                                                 const MapK &mt, float eps, float inv_count, float &lsum, Grad &acc)
{
    float rt[3];
    {
        const Dots dt = dots(K, g, mt);
        Lobe lt[NL];
        float Ft[3], ft[3];
        shade<NL, false>(K, g, mt, dt, lt, Ft, ft, rt);
    }
    const Dots di = dots(K, g, mi);
    Lobe li[NL];
    float Fi[3], fi[3], ri[3], g_rad[3];
    shade<NL, WITH_GRAD>(K, g, mi, di, li, Fi, fi, ri);
    // losses.py:46-50: |log(ri + eps) - log(rt + eps)| and its derivative sign/(N (ri + eps)).
    // Transcendentals are what this kernel pays most for (~16 issue cycles each against ~2.5 for a
    // plain instruction once several waves share the SIMD: profiles/r01_k3_cycles.txt),
    // so the nine of the reference's formulation (six logs, three reciprocals) are done with four:
    //  * the three 1/ai come from ONE v_rcp of their product (6 multiplies).  Operands are scaled
    //    by 2^-10 (exact) so that the product stays in range for eps <= ai <= 7e15, eps >= 1e-9
    //    (checked on the host);
    //  * log(ai) - log(at) = -log(at/ai): one log per channel on the quotient formed with that 1/ai.
    // ai == at must give exactly 0 like the reference's log(x) - log(x) (identical maps: zero loss,
    // zero gradient, sign(0) = 0), hence the explicit select.
    {
        constexpr float c = 9.765625e-04f;               // 2^-10
        const float ec = eps * c, nc = inv_count * c;
        float b[3], bt[3], ib[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) { b[k] = fma_(ri[k], c, ec); bt[k] = fma_(rt[k], c, ec); }   // = (r + eps) * c, one rounding
        {
            const float P = b[0] * b[1];
            const float r = rcp_(P * b[2]);
            const float t = r * b[2];
            ib[0] = t * b[1]; ib[1] = t * b[0]; ib[2] = r * P;
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            // lg = log2(at/ai) = -(log(ai) - log(at))/ln2.  The loss sums |lg| (scaled by ln2 ONCE, after the scene
            // loop); d|delta|/d ri = sign(delta)/(N ai) = -sign(lg) nc/b: the magnitude nc*ib takes lg's sign bit with
            // one v_and_or, and the minus rides as a source modifier on the product that consumes it.  sign(0) = 0 as in
            // torch: equal operands select lg = 0 explicitly, and lg == 0 -- equal operands, or a quotient of unequal
            // ones that rounds to exactly 1 -- selects magnitude 0 (a +0 there would otherwise always read as "input
            // darker": a systematic sign at near-ties where the reference's rounded log difference gives 0).
            const float lg = (b[k] != bt[k]) ? log2_(bt[k] * ib[k]) : 0.0f;
            lsum += fabsf(lg);
            const float mag = (lg != 0.0f) ? nc * ib[k] : 0.0f;
            g_rad[k] = -__builtin_bit_cast(float, __builtin_bit_cast(unsigned, mag) | (__builtin_bit_cast(unsigned, lg) & 0x80000000u));
        }
    }
    if (WITH_GRAD) shade_bwd<NL, (DEFER & 1) != 0, (DEFER & 2) != 0>(K, g, mi, di, li, Fi, fi, g_rad, acc);
}

// The same for independent roughness channels (three lobes per map), one colour channel after the other: target lobe,
// input lobe, loss term and adjoint of channel k are finished before channel k+1 starts, so that only the four sums
// over the lobes (d loss/d VN, LN, NH^2, LN+) stay live between channels.  The joint form above keeps three lobes'
// partials (15 values) and three F/f pairs alive across the loss term and cost the three-lobe loop 19 scratch
// accesses and ~60 register moves per iteration (510 VALU; 77.8 us at config 2).  Price: the three 1/(render+eps)
// are three v_rcp here instead of one.
template <bool WITH_GRAD, int DEFER = 0>
__device__ __forceinline__ void loss_pixel_scene_by_channel(const VConst &K, const Geom &g, const MapK &mi, const MapK &mt,
                                                            float eps, float inv_count, float &lsum, Grad &acc)
{
    const Dots dt = dots(K, g, mt);
    const Dots di = dots(K, g, mi);
    constexpr float c = 9.765625e-04f;               // 2^-10, as in loss_pixel_scene
    const float ec = eps * c, nc = inv_count * c;
    const float omp = 1.0f - g.p;
    float g_LNp = 0.0f, g_VN = 0.0f, g_LN = 0.0f, sN = 0.0f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const Lobe lt = lobe<false>(K, mt.A[k], mt.oA[k], dt);
        const float Ft = fma_(mt.oms[k], g.p, mt.s[k]);
        const float ft = fma_(Ft, lt.GD - mt.dpi[k], mt.dpi[k]);
        const float rt = ft * (g.E[k] * dt.LNp);
        const Lobe li = lobe<WITH_GRAD>(K, mi.A[k], mi.oA[k], di);
        const float Fi = fma_(mi.oms[k], g.p, mi.s[k]);
        const float fi = fma_(Fi, li.GD - mi.dpi[k], mi.dpi[k]);
        const float ri = fi * (g.E[k] * di.LNp);
        const float b = fma_(ri, c, ec), bt = fma_(rt, c, ec);
        const float ib = rcp_(b);
        const float lg = (b != bt) ? log2_(bt * ib) : 0.0f;             // see loss_pixel_scene
        lsum += fabsf(lg);
        if (WITH_GRAD) {
            const float mag = (lg != 0.0f) ? nc * ib : 0.0f;
            const float g_rad = -__builtin_bit_cast(float, __builtin_bit_cast(unsigned, mag) | (__builtin_bit_cast(unsigned, lg) & 0x80000000u));
            const float gE = g_rad * g.E[k];
            const float g_f = gE * di.LNp;
            g_LNp = fma_(gE, fi, g_LNp);
            const float g_F = g_f * (li.GD - mi.dpi[k]);
            acc.s[k] = fma_(g_F, omp, acc.s[k]);
            acc.d[k] = (DEFER & 1) ? fma_(g_f, 1.0f - Fi, acc.d[k]) : fma_(g_f * (1.0f - Fi), K.inv_pi, acc.d[k]);
            const float gGD = g_f * Fi;
            acc.r[k] = (DEFER & 2) ? fma_(gGD, li.KA, acc.r[k]) : fma_(gGD * li.KA, mi.r4m[k], acc.r[k]);
            g_VN = fma_(gGD, li.KV, g_VN);
            g_LN = fma_(gGD, li.KL, g_LN);
            sN = fma_(gGD, li.KN, sN);
        }
    }
    if (WITH_GRAD) {
        float g_NH = (sN * 2.0f) * di.NH;
        if (!(di.nh_raw >= K.tiny)) g_NH = 0.0f;
        if (!(di.vn_raw >= K.tiny)) g_VN = 0.0f;
        if (!(di.ln_raw >= K.tiny)) g_LN = 0.0f;
        if (!(di.ln_raw >= 0.0f)) g_LNp = 0.0f;
        const float gl = g_LN + g_LNp;
        acc.n[0] = fma_(g_NH, g.hx, fma_(g_VN, g.wox, fma_(gl, g.wix, acc.n[0])));
        acc.n[1] = fma_(g_NH, g.hy, fma_(g_VN, g.woy, fma_(gl, g.wiy, acc.n[1])));
        acc.n[2] = fma_(g_NH, g.hz, fma_(g_VN, g.woz, fma_(gl, g.wiz, acc.n[2])));
    }
}

template <int NL, bool WITH_GRAD, int DEFER = 0>
__device__ __forceinline__ void loss_pixel_scene_any(const VConst &K, const Geom &g, const MapK &mi, const MapK &mt,
                                                     float eps, float inv_count, float &lsum, Grad &acc)
{
    if (NL == 3)
        loss_pixel_scene_by_channel<WITH_GRAD, DEFER>(K, g, mi, mt, eps, inv_count, lsum, acc);
    else
        loss_pixel_scene<NL, WITH_GRAD, DEFER>(K, g, mi, mt, eps, inv_count, lsum, acc);
}

// Scene loop of K3, software-pipelined: the geometry of render s+1 (three rsq-headed dependent
// chains: lengths -> reciprocals -> quotients) is computed in the same iteration as the
// shading / loss / adjoint of render s.  The two are independent instruction streams, which gives
// the scheduler work to put behind the transcendentals' latency (measured: a v_rsq/v_rcp whose
// result is consumed right away costs ~11 plain VALU slots, tools/valu_mix.hip).
//
// Where the nine scene scalars of a render come from is a register-pressure question, decided by
// measurement (config 2, us per launch):            forward+adjoint   forward only
//   LDS stage (one copy per workgroup, ds_read)          59.1             39.8
//   global loads, double-buffered two renders ahead      56.5             43.8
// The adjoint kernel sits at the 128-VGPR limit of 4 waves/SIMD and spills a little either way; the
// LDS variant keeps the scalars live across the interleaved streams and spills more.  So: LDS for
// the forward-only kernels, prefetched global loads for the forward+adjoint kernels.
// Rejected with same-box evidence in round 4 and no longer in the source (HISTORY.md A.1; git show 3d7c7a1 has them): issue
// priorities by remaining work in the tail of a launch (s_setprio), a peeled last pass without the unused successor
// geometry, the un-pipelined and the rolled loop, and re-loading the target maps in every pass.
#ifndef SVBRDF_K3_STAGGER
#define SVBRDF_K3_STAGGER 64        // s_sleep units (64 cycles) between the load layers of a launch's first round; 0 = off
#endif
#ifndef SVBRDF_K3_EARLY_COORDS
#define SVBRDF_K3_EARLY_COORDS 1    // pixel coordinates loaded in front of the plane loads (rendering_loss_body)
#endif
template <int NL, bool WITH_GRAD, int DEFER = 0>
__device__ __forceinline__ float loss_scene_loop(const MapK &mi, const MapK &mt_in, float x, float y,
                                                 const float *__restrict__ scp, const float *sc_lds, int S,
                                                 float eps, float inv_count, Grad &acc)
{
    constexpr int ST = 9;               // floats per scene row
    float lsum = 0.0f;
    const VConst K = make_vconst();
    eps = vreg(eps);
    inv_count = vreg(inv_count);
    float sc[9];
#if SVBRDF_TIMING
    const long long tm0 = clock64(), wc0 = wall_clock64();
#endif
    if (WITH_GRAD) {
        load_scene(scp, sc);
        Geom ga = geometry(K, sc, x, y), gb;                     // render 0
        load_scene(scp + (S > 1 ? ST : 0), sc);                  // scalars of render 1
        // One pass = shade render s with geometry G_CUR while the geometry of render s+1 goes into G_NEXT.  The loop
        // body holds TWO passes with the roles of ga / gb swapped, so the pipelined geometry never has to be copied
        // from a "next" to a "current" register set (9 v_mov per render in the rolled loop).
#define SVBRDF_K3_PASS(G_CUR, G_NEXT, SI)                                                                             \
        {                                                                                                          \
            asm volatile("" ::"s"(sc[0]), "s"(sc[8]));           /* the wait for sc lands here, before the next loads */ \
            float cur[9];                                                                                          \
            _Pragma("unroll") for (int i = 0; i < 9; ++i) cur[i] = sc[i];                                          \
            load_scene(scp + ((SI) + 2 < S ? 2 * ST : ((SI) + 1 < S ? ST : 0)), sc);                               \
            scp += ((SI) + 1 < S) ? ST : 0;                                                                        \
            __builtin_amdgcn_sched_barrier(0);                                                                     \
            /* two independent streams from here to the end of the pass: */                                        \
            G_NEXT = geometry(K, cur, x, y);                     /* render s+1 (a harmless repeat on the last pass) */ \
            loss_pixel_scene_any<NL, WITH_GRAD, DEFER>(K, G_CUR, mi, mt_in, eps, inv_count, lsum, acc);                   \
        }
        for (int s = 0;;) {
            SVBRDF_K3_PASS(ga, gb, s)
            if (++s >= S) break;
            SVBRDF_K3_PASS(gb, ga, s)
            if (++s >= S) break;
        }
#undef SVBRDF_K3_PASS
    } else {
        load_scene(sc_lds, sc);
        Geom g_next = geometry(K, sc, x, y);
        for (int s = 0; s < S; ++s) {
            const Geom g = g_next;
            load_scene(sc_lds + (s + 1 < S ? s + 1 : s) * 9, sc);
            g_next = geometry(K, sc, x, y);
            loss_pixel_scene_any<NL, WITH_GRAD, DEFER>(K, g, mi, mt_in, eps, inv_count, lsum, acc);
        }
    }
    lsum *= 0.693147180559945309417f;       // the scene loop sums |log2|: natural log once per pixel
    if (WITH_GRAD && DEFER) {               // the per-pixel constants shade_bwd left out (see its DEFER_D / DEFER_R)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (DEFER & 1) acc.d[k] *= K.inv_pi;
            if (DEFER & 2) acc.r[k] *= mi.r4m[k];
        }
    }
#if SVBRDF_TIMING
    if (WITH_GRAD) {        // timing build: the normal-gradient planes carry loop cycles / 100 MHz ticks / start stamp
        acc.n[0] = (float)(clock64() - tm0);
        acc.n[1] = (float)(wall_clock64() - wc0);
        acc.n[2] = (float)(wc0 & 0xffffff);        // 100 MHz ticks, one clock for the whole chip
    }
#endif
    return lsum;
}

struct L1Params {       // SVBRDFL1Loss folded into the same pass (losses.py:7-19, 62-63)
    float sum_scale;    // l1_weight * S: puts the L1 sums on the rendering loss's 1/(B S 3 H W) scale
    float grad_scale;   // l1_weight / (B 3 H W)
    float eps;          // 0.01, losses.py:13
};

// sign with sign(0) = 0 (torch.sign), times `scale`
__device__ __forceinline__ float signed_scale(float delta, float scale)
{
    return __builtin_amdgcn_fmed3f(delta * 1.0e30f, -1.0f, 1.0f) * scale;
}

// Network head folded into the loss (the north star's "normal-map decode"):
// the generator's 9-channel output after tanh, layout normals_xy(0:2) | diffuse(2:5) |
// roughness(5) | specular(6:9) (utils.py:49-53), is decoded exactly like
// models.py:338-346 -> utils.py:73-98: normals = normalize(3 nx, 3 ny, 1), roughness repeated
// to three channels, d/r/s mapped from [-1,1] to [0,1] by (x+1)/2.  The normal feeds NH, so its
// normalisation reproduces the reference's rounding (exact dot, correctly rounded sqrt and
// division) like the rest of the geometry.
struct Head {
    float nx, ny, nz;      // decoded unit normal
    float inv_len;         // 1/|(3nx, 3ny, 1)|
};

__device__ __forceinline__ Head decode_head(const float e[9], Maps &m)
{
    Head h;
    const float vx = e[0] * 3.0f, vy = e[1] * 3.0f;
    float seed;
    const Recip il = length_rn((vx * vx + vy * vy) + 1.0f, seed);   // sum(pow(v,2)) with v.z = 1
    h.nx = div_rn(vx, il); h.ny = div_rn(vy, il); h.nz = div_rn(1.0f, il);
    h.inv_len = il.y;
    m.n[0] = h.nx; m.n[1] = h.ny; m.n[2] = h.nz;
    const float r = (e[5] + 1.0f) * 0.5f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        m.d[k] = (e[2 + k] + 1.0f) * 0.5f;
        m.r[k] = r;
        m.s[k] = (e[6 + k] + 1.0f) * 0.5f;
    }
    return h;
}

// chain rule through decode_head: 12-channel gradient -> 9-channel gradient
__device__ __forceinline__ void head_bwd(const Head &h, const Grad &g, float ge[9])
{
    // n = v/|v|: dL/dv = (g_n - n (n.g_n))/|v|, v = (3 nx, 3 ny, 1)
    const float ng = fma_(h.nx, g.n[0], fma_(h.ny, g.n[1], h.nz * g.n[2]));
    ge[0] = 3.0f * (fma_(-h.nx, ng, g.n[0]) * h.inv_len);
    ge[1] = 3.0f * (fma_(-h.ny, ng, g.n[1]) * h.inv_len);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        ge[2 + k] = 0.5f * g.d[k];
        ge[6 + k] = 0.5f * g.s[k];
    }
    ge[5] = 0.5f * ((g.r[0] + g.r[1]) + g.r[2]);
}

// Loss reduction, workgroup level: one lane adds its workgroup's partial sum `t` (fixed point) to the workgroup's slot
// with ONE returning atomic that also counts the slot's arrivals; returns true for the workgroup that completed the
// last slot of the launch (the finisher).
__device__ __forceinline__ bool loss_arrive(float t, float fixed_scale, unsigned long long *__restrict__ ws)
{
    const unsigned nblocks = gridDim.x * gridDim.y, bid = blockIdx.y * gridDim.x + blockIdx.x;
    const unsigned slot = bid & (kLossSlots - 1);
    const unsigned slot_blocks = (nblocks - slot + kLossSlots - 1) / kLossSlots;
    // The scale keeps every legitimate partial sum below 2^47 (loss_impl).  A partial sum that is NaN, infinite
    // or beyond that (NaN/inf maps, or radiances no renderer input can produce) must neither be cast (undefined
    // for NaN/inf) nor reach the arrival count in the word's top bits: it contributes 0 and raises the sticky
    // non-finite flag next to the ticket counter instead, and the finisher reports NaN -- as the reference's
    // log/L1 chain would (isfinite(loss) guards keep working) -- and leaves the scratch zeroed as always.
    // The bound is per workgroup: the slot_blocks partial sums of one slot must not carry into the arrival count
    // together either, so each stays below 2^47 / slot_blocks -- still above every legitimate value (the host picks
    // the scale so that slot_blocks maximal partial sums fit 2^47), but finite absurd maps (|dlog| of 70-110 per
    // term) can no longer add up past bit 48, keep the finisher from firing and leave the scratch dirty.
    const float scaled = fma_(t, fixed_scale, 0.5f);
    const float limit = 140737488355328.0f * 0.999f * rcp_((float)slot_blocks);     // 2^47 / slot_blocks
    const bool finite = scaled >= 0.0f && scaled < limit;                            // false for NaN
    const unsigned long long fixed = finite ? (unsigned long long)scaled : 0ULL;
    if (!finite) {
        atomicOr(&ws[kLossSlots], kLossNonFiniteFlag);
        __threadfence();        // the flag is visible device-wide before this workgroup's arrival is
    }
    // device-scope returning atomic, performed at the memory side: add + arrival count in one
    const unsigned long long old = atomicAdd(&ws[slot], (1ULL << kLossCountShift) | fixed);
    if ((unsigned)(old >> kLossCountShift) + 1 == slot_blocks) {
        const unsigned nslots = nblocks < (unsigned)kLossSlots ? nblocks : (unsigned)kLossSlots;
        const unsigned long long ticket = atomicAdd(&ws[kLossSlots], 1ULL);
        if ((unsigned)(ticket & kLossTicketMask) + 1 == nslots) return true;
    }
    return false;
}

// the finisher's first wave (`lane` = 0..63): every slot is complete (its completer drew its ticket after its add had
// returned): fetch-and-clear all slots in parallel, integer wave reduction, write the mean
__device__ __forceinline__ void loss_finish(unsigned lane, unsigned long long *__restrict__ ws, float *__restrict__ loss_out,
                                            double loss_scale)
{
    const unsigned nblocks = gridDim.x * gridDim.y;
    unsigned long long v = 0;
    if (lane < (unsigned)kLossSlots && lane < nblocks) v = atomicExch(&ws[lane], 0ULL) & kLossSumMask;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    if (lane == 0) {
        __threadfence();
        const unsigned long long tail = atomicExch(&ws[kLossSlots], 0ULL);
        loss_out[0] = (tail & kLossNonFiniteFlag) ? __builtin_nanf("") : (float)((double)v * loss_scale);
    }
}

// the gradient planes of one pixel: 12 channels, or the 9 of the encoded head output (chain rule through decode_head)
template <bool HEAD>
__device__ __forceinline__ void store_pixel_grad(const Head &head, const Grad &acc, float *__restrict__ grad_input, int b,
                                                 size_t plane, size_t pix)
{
    if (HEAD) {
        float ge[9];
        head_bwd(head, acc, ge);
        float *__restrict__ gp = grad_input + (size_t)b * 9 * plane;
        const PlaneBuf pb = plane_buf(gp, 9, plane, pix);
#pragma unroll
        for (int k = 0; k < 9; ++k) plane_store(pb, k, ge[k]);
    } else {
        store_grads_k3(grad_input + (size_t)b * 12 * plane, plane, pix, acc);
    }
}

// One thread = one pixel, all S renders of it (VEC = 1: the kernel is VALU-bound, wider loads measured no gain and cost
// occupancy); workgroup = 256 pixels.  (Round 4 also built workgroups of 64 pixels x G waves that share the renders of a
// pixel -- more, shorter waves against the ramp and tail of a launch -- and rejected them: the per-wave fixed work is worth
// 0.75 renders, profiles/r04_k3_ab.txt, HISTORY.md A.1; the variant is in the history, git show 3d7c7a1.)
// WITH_L1 adds SVBRDFL1Loss on the 24 values already in registers.
template <bool WITH_GRAD, bool WITH_L1, bool HEAD, bool EARLY_COORDS = false>
__device__ __forceinline__ void rendering_loss_body(const float *__restrict__ input, const float *__restrict__ target,
                                                    const float *__restrict__ scenes, const float *__restrict__ xrow,
                                                    float eps, float inv_count, double loss_scale, float fixed_scale,
                                                    L1Params l1, float *__restrict__ grad_input,
                                                    unsigned long long *__restrict__ ws, float *__restrict__ loss_out,
                                                    int S, int H, int W)
{
    extern __shared__ __attribute__((aligned(16))) float sc_lds[];      // [S][9] scene scalars of batch item b
    constexpr float kLn2 = 0.693147180559945309417f;
    const size_t plane = (size_t)H * W;
    const size_t pix = (size_t)blockIdx.x * kLossThreads + threadIdx.x;
    const int b = blockIdx.y;
    const bool active = pix < plane;
    float lsum = 0.0f;
    Grad acc;
    Head head;
#if SVBRDF_TIMING
    const long long t_entry = wall_clock64();
#endif
#if SVBRDF_K3_STAGGER
    if (WITH_GRAD && EARLY_COORDS && S >= 6) {     // (by-value-table kernels only, like the early coordinate loads)
        // The first resident round of a launch -- 1024 workgroups, 4096 waves -- issues 25 MB of plane loads within 0.4 us,
        // and when the maps come from HBM every one of those waves gets its last plane at about the same time, 4-5 us
        // later: nobody computes until then.  Issued in four layers (workgroup >> 8 = which of a CU's four workgroup
        // slots it takes, the dispatcher filling the CUs breadth-first), SVBRDF_K3_STAGGER x 64 cycles apart, the first
        // layer's loads meet an idle memory system and its waves are in their scene loops while the later layers' data
        // arrives -- the stagger that age arbitration produces anyway, from the start.  Same-box A/B, medians of three
        // (profiles/r04_k3_ab_stagger_hbm.txt, r04_k3_ab_stagger_long.txt): with the maps from HBM -3 ... -4 % per launch
        // for the rendering loss, -1 % for the mixed loss and batch 16; neutral with cache-resident maps; +1 ... +2 % with
        // two launches in flight.  64 units best (48: half the gain; 80 and more cost one-round launches and the L1
        // variants; eight or sixteen finer layers: no better, r04_k3_ab_stagger_layers.txt).  Only for launches whose
        // waves live long enough (>= 6 renders per pixel: the reference's loss has 9, config 5 has 32).  A different
        // placement order would make this a harmless delay, not an error.
        const unsigned layer = (blockIdx.y * gridDim.x + blockIdx.x) >> 8;
        if (layer == 1) __builtin_amdgcn_s_sleep(SVBRDF_K3_STAGGER);
        else if (layer == 2) { __builtin_amdgcn_s_sleep(SVBRDF_K3_STAGGER); __builtin_amdgcn_s_sleep(SVBRDF_K3_STAGGER); }
        else if (layer == 3) { __builtin_amdgcn_s_sleep(SVBRDF_K3_STAGGER); __builtin_amdgcn_s_sleep(SVBRDF_K3_STAGGER); __builtin_amdgcn_s_sleep(SVBRDF_K3_STAGGER); }
    }
#endif
    if (!WITH_GRAD) {    // forward-only kernels stage the scene table of batch item b in LDS (see loss_scene_loop)
        for (int i = threadIdx.x; i < S * 9; i += kLossThreads) sc_lds[i] = scenes[(size_t)b * S * 9 + i];
        __syncthreads();
    }
    if (active) {
        Maps in[1], tg[1];
        // The pixel's two coordinates come from the xrow table.  Left to the compiler they are loaded AFTER the wait for
        // the 24 plane loads (and the scene scalars after them): three memory round trips in series in front of every
        // wave's first geometry, and in the first resident round of a launch each of them queues behind the burst of
        // all the waves' plane loads.  Issued here, in front of the plane loads, they are in flight together with them.
        // Plain loads followed by a scheduling barrier: the machine scheduler (which otherwise sinks them to their use)
        // cannot move them across it, and the compiler's own s_waitcnt bookkeeping covers them -- safe by construction.
        // (Rounds 3-4 issued them with inline asm outside that bookkeeping, correct only while they stayed older than every
        // tracked load; same instruction order, same loops, eight instructions fewer this way.)  tests/test_isa_guard.py
        // checks that no plane load is issued before them.
        [[maybe_unused]] float x_early = 0.0f, y_early = 0.0f;
        // (by-value-table kernels only: the device-table variants sit at the register limit and answer two more live
        // values in the prologue with spills between the plane loads)
        const bool early_coords = SVBRDF_K3_EARLY_COORDS && EARLY_COORDS && WITH_GRAD && (W & (W - 1)) == 0;
        if (early_coords) {
            const unsigned p32 = (unsigned)pix, sh = (unsigned)__builtin_ctz((unsigned)W);
            x_early = xrow[p32 & (unsigned)(W - 1)];
            y_early = xrow[p32 >> sh];
            __builtin_amdgcn_sched_barrier(0);
        }
        if (HEAD) {     // input is the [B,9,H,W] post-tanh generator output
            float e[9];
            const float *__restrict__ ip = input + (size_t)b * 9 * plane;
            const PlaneBuf pb = plane_buf(ip, 9, plane, pix);
#pragma unroll
            for (int k = 0; k < 9; ++k) e[k] = plane_load(pb, k);
            head = decode_head(e, in[0]);
        } else {
            load_maps_k3(input + (size_t)b * 12 * plane, plane, pix, in[0]);
        }
        load_maps_k3(target + (size_t)b * 12 * plane, plane, pix, tg[0]);
        zero_grad(acc);
        float l1sum = 0.0f;
        // deferred per-pixel constants of the adjoint (shade_bwd): 1/pi of the diffuse gradient always (an L1 start value
        // is multiplied by pi here), dA/dr_hat of the roughness gradient only where acc.r starts from zero (it can be 0)
        constexpr int kDefer = WITH_GRAD ? (WITH_L1 ? 1 : 3) : 0;
        if (WITH_L1) {
            // losses.py:7-19.  Same economy of transcendentals as in loss_pixel_scene: the six 1/(x + eps)
            // of the log terms' derivatives come from ONE v_rcp of their product (all six lie in
            // [eps_l1, 1 + eps_l1]: no scaling needed), and log(a) - log(b) is one log of the quotient formed
            // with that reciprocal (equal operands select exactly 0): 7 transcendentals per pixel instead of 18.
            float a[6], b[6], ia[6];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                a[k] = in[0].d[k] + l1.eps; b[k] = tg[0].d[k] + l1.eps;
                a[3 + k] = in[0].s[k] + l1.eps; b[3 + k] = tg[0].s[k] + l1.eps;
            }
            {
                float pre[6];               // prefix products a0, a0 a1, ...
                pre[0] = a[0];
#pragma unroll
                for (int k = 1; k < 6; ++k) pre[k] = pre[k - 1] * a[k];
                float r = rcp_(pre[5]);     // 1/(a0 ... a5)
#pragma unroll
                for (int k = 5; k > 0; --k) { ia[k] = r * pre[k - 1]; r *= a[k]; }
                ia[0] = r;
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float dn = in[0].n[k] - tg[0].n[k], dr = in[0].r[k] - tg[0].r[k];
                const float dd = (a[k] == b[k]) ? 0.0f : -kLn2 * log2_(b[k] * ia[k]);
                const float ds = (a[3 + k] == b[3 + k]) ? 0.0f : -kLn2 * log2_(b[3 + k] * ia[3 + k]);
                l1sum += (fabsf(dn) + fabsf(dr)) + (fabsf(dd) + fabsf(ds));
                if (WITH_GRAD) {
                    acc.n[k] = signed_scale(dn, l1.grad_scale);
                    acc.r[k] = signed_scale(dr, l1.grad_scale);
                    acc.d[k] = signed_scale(dd, l1.grad_scale) * ia[k];
                    if (kDefer & 1) acc.d[k] *= kPi;
                    acc.s[k] = signed_scale(ds, l1.grad_scale) * ia[3 + k];
                }
            }
        }
        const bool tied = tied_roughness(in[0]) && tied_roughness(tg[0]);
        const MapK mi = prepare<WITH_GRAD>(in[0]), mt = prepare<false>(tg[0]);
        float x[1], y;
        if (early_coords) {
            x[0] = x_early;
            y = -y_early;
        } else if ((W & (W - 1)) == 0) {
            // power-of-two width (256, 512: every BASELINE configuration): row and column by shift and mask instead of a
            // 64-bit division (~25 VALU instructions, several of them quarter-rate integer multiplies)
            const unsigned p32 = (unsigned)pix, sh = (unsigned)__builtin_ctz((unsigned)W);
            x[0] = xrow[p32 & (unsigned)(W - 1)];
            y = -xrow[p32 >> sh];
        } else {
            pixel_coords<1>(xrow, pix, W, x, y);
        }
        {
            // torch.clamp propagates NaN (renderers.py:48-52, 87), v_max_f32 returns the other operand: a NaN (or
            // infinite) normal or roughness value would vanish in the clamps and leave a finite loss beside NaN
            // gradients.  t - t is 0 for finite t and NaN otherwise; added to the pixel's x coordinate (already live
            // through the scene loop: no extra register) it leaves finite maps untouched and makes every radiance of
            // the pixel, hence the loss, NaN when such a value is NaN (as the reference's is) or infinite (where the
            // reference gives NaN, inf or a clamped value depending on the sign: NaN is the safe report).  Diffuse and
            // specular propagate through the shading arithmetic by themselves.
            const float chk = (((in[0].n[0] + in[0].n[1]) + (in[0].n[2] + in[0].r[0])) + (in[0].r[1] + in[0].r[2])) +
                              (((tg[0].n[0] + tg[0].n[1]) + (tg[0].n[2] + tg[0].r[0])) + (tg[0].r[1] + tg[0].r[2]));
            x[0] += chk - chk;
        }
        const float *__restrict__ scp = scenes + (size_t)b * S * 9;
        if (__all(tied))     // wave-uniform: every lane's input AND target roughness channels are tied
            lsum = loss_scene_loop<1, WITH_GRAD, kDefer>(mi, mt, x[0], y, scp, sc_lds, S, eps, inv_count, acc);
        else
            lsum = loss_scene_loop<3, WITH_GRAD, kDefer>(mi, mt, x[0], y, scp, sc_lds, S, eps, inv_count, acc);
        if (WITH_L1) lsum = fma_(l1sum, l1.sum_scale, lsum);
#if SVBRDF_TIMING
        if (WITH_GRAD && !HEAD) {       // timing build: entry / exit stamps of the wave beside the loop's (loss_scene_loop)
            acc.d[0] = (float)(t_entry & 0xffffff);
            acc.d[1] = (float)(wall_clock64() & 0xffffff);
            // where the wave ran (tools/k3_placement.py: is the first resident round placed breadth-first over the CUs, as the
            // load stagger above assumes?): HW_ID (hwreg 4: CU_ID [11:8], SH_ID [12], SE_ID [15:13]) and XCC_ID (hwreg 20, [3:0])
            const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20);
            acc.d[2] = (float)(((xcc & 15u) << 8) | (((hw >> 13) & 7u) << 5) | (((hw >> 12) & 1u) << 4) | ((hw >> 8) & 15u));
        }
#endif
        if (WITH_GRAD) store_pixel_grad<HEAD>(head, acc, grad_input, b, plane, pix);
    }
    {
        __shared__ float wave_part[kLossThreads / 64];
        __shared__ int finisher;
        lsum = wave_sum(lsum);
        if ((threadIdx.x & 63) == 0) wave_part[threadIdx.x >> 6] = lsum;
        __syncthreads();
        if (threadIdx.x == 0) {
            float t = 0.0f;
#pragma unroll
            for (int w = 0; w < kLossThreads / 64; ++w) t += wave_part[w];
            finisher = loss_arrive(t, fixed_scale, ws) ? 1 : 0;
        }
        __syncthreads();
        if (finisher && threadIdx.x < 64) loss_finish(threadIdx.x, ws, loss_out, loss_scale);
    }
}

#define SVBRDF_K3_ATTRS __launch_bounds__(kLossThreads) __attribute__((amdgpu_waves_per_eu(SVBRDF_K3_MIN_WAVES, 8)))

// scene table in device memory (any B*S)
template <bool WITH_GRAD, bool WITH_L1, bool HEAD>
__global__ SVBRDF_K3_ATTRS void k_rendering_loss(const float *__restrict__ input, const float *__restrict__ target,
                                                 const float *__restrict__ scenes, const float *__restrict__ xrow,
                                                 float eps, float inv_count, double loss_scale, float fixed_scale,
                                                 L1Params l1, float *__restrict__ grad_input,
                                                 unsigned long long *__restrict__ ws, float *__restrict__ loss_out,
                                                 int S, int H, int W)
{
    rendering_loss_body<WITH_GRAD, WITH_L1, HEAD>(input, target, scenes, xrow, eps, inv_count, loss_scale, fixed_scale,
                                                  l1, grad_input, ws, loss_out, S, H, W);
}

// Scene table passed BY VALUE in the kernel-argument block (B*S <= SVBRDF_HOST_SCENES_MAX_ROWS).
// The table is drawn on the host for every call (losses.py:35), so it has to cross to the device
// once per call either way; as a kernel argument it rides in the dispatch packet's own argument
// buffer: no device allocation, no hipMemcpyAsync command in front of the kernel, no pinned
// staging slot and no event to guard its reuse.  Measured on config 2: the memcpy route costs
// ~24 us of host time per call and makes the step host-bound at ~60 us; this route leaves one
// dispatch per step and the loop GPU-bound at the kernel's own ~53 us.  The body reads the rows
// with the same wave-uniform scalar loads, from the kernarg segment instead of a global buffer.
// Capacity: 288 rows (10,368 bytes) -- configs[3] (16 x 9 rows) and config 5 at batch 8 (8 x 32) fit; the
// runtime takes argument blocks of 32 KB and more on gfx950 (tools: a by-value struct of 32,000 bytes launches and
// reads back correctly), and marshalling 10 KB costs the launch ~0.2 us.
// `table` is the FIRST argument, i.e. it sits at offset 0 of the kernarg segment, and is read through
// the segment pointer: taking the address of the by-value parameter itself makes the compiler
// copy the whole block into scratch in the adjoint variants (seen in the resource report).
template <bool WITH_GRAD, bool WITH_L1, bool HEAD>
__global__ SVBRDF_K3_ATTRS void k_rendering_loss_inl([[maybe_unused]] const SceneBlock table,
                                                     const float *__restrict__ input, const float *__restrict__ target,
                                                     const float *__restrict__ xrow, float eps, float inv_count,
                                                     double loss_scale, float fixed_scale, L1Params l1,
                                                     float *__restrict__ grad_input, unsigned long long *__restrict__ ws,
                                                     float *__restrict__ loss_out, int S, int H, int W)
{
    const float *__restrict__ rows = (const float *)__builtin_amdgcn_kernarg_segment_ptr();
    rendering_loss_body<WITH_GRAD, WITH_L1, HEAD, true>(input, target, rows, xrow, eps, inv_count, loss_scale,
                                                           fixed_scale, l1, grad_input, ws, loss_out, S, H, W);
}


#if defined(SVBRDF_ISA_PROBE)
// tests/test_isa_guard.py: dot3 alone, to check in the assembly that its products are not contracted into FMAs
extern "C" __global__ void svbrdf_isa_probe_dot3(const float *__restrict__ a, float *__restrict__ o)
{
    o[threadIdx.x] = dot3(a[threadIdx.x], a[threadIdx.x + 64], a[threadIdx.x + 128], a[threadIdx.x + 192], a[threadIdx.x + 256],
                          a[threadIdx.x + 320]);
}
#endif

#if SVBRDF_TU_MAIN
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
        // b plays the squared length: the kernels divide by sqrt(b)
        float seed;
        const Recip r = length_rn(b, seed);
        const float len = sqrtf(b);
        if (r.b != len) ++bad_sqrt;
        if (div_rn(a, r) != a / len) ++bad_div;
    }
    if (bad_div) atomicAdd(&counts[0], bad_div);
    if (bad_sqrt) atomicAdd(&counts[1], bad_sqrt);
}

// ------------------------------------------------------------------------------------------
// K4: material mixing (dataset.py:142-160 SvbrdfDataset.mix), one pass over two SVBRDFs:
//   normals projected to z = 1 (n / max(0.01, n.z)), blended, renormalised; diffuse / roughness / specular blended;
//   weights alpha and fp32(1 - alpha) per batch item.  HBM-bound (24 planes in, 12 out = 144 B per pixel, ~40 flop).
// Operation order and roundings are the reference's: every product rounded on its own (-ffp-contract=off), the
// squared length summed (p0 + p1) + p2 like torch.sum over the channel axis, IEEE division and square root.
// ------------------------------------------------------------------------------------------
template <int VEC>
__global__ __launch_bounds__(kThreads) void k_mix_materials(const float *__restrict__ m0, const float *__restrict__ m1,
                                                            const float *__restrict__ alpha, float *__restrict__ out,
                                                            size_t plane)
{
    const size_t pix = ((size_t)blockIdx.x * kThreads + threadIdx.x) * VEC;
    const int b = blockIdx.y;
    if (pix >= plane) return;
    const float a = alpha[b], oma = 1.0f - a;
    const float *__restrict__ p0 = m0 + (size_t)b * 12 * plane + pix;
    const float *__restrict__ p1 = m1 + (size_t)b * 12 * plane + pix;
    float *__restrict__ po = out + (size_t)b * 12 * plane + pix;
    float n0[3][VEC], n1[3][VEC], nm[3][VEC];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        load_vec<VEC, true>(p0 + (size_t)k * plane, n0[k]);
        load_vec<VEC, true>(p1 + (size_t)k * plane, n1[k]);
    }
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
        const float z0 = fmaxf(0.01f, n0[2][v]), z1 = fmaxf(0.01f, n1[2][v]);
#pragma unroll
        for (int k = 0; k < 3; ++k) nm[k][v] = a * (n0[k][v] / z0) + oma * (n1[k][v] / z1);
        const float len = sqrtf((nm[0][v] * nm[0][v] + nm[1][v] * nm[1][v]) + nm[2][v] * nm[2][v]);
#pragma unroll
        for (int k = 0; k < 3; ++k) nm[k][v] = nm[k][v] / len;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) store_vec<VEC, true>(po + (size_t)k * plane, nm[k]);
#pragma unroll
    for (int k = 3; k < 12; ++k) {
        float x0[VEC], x1[VEC], y[VEC];
        load_vec<VEC, true>(p0 + (size_t)k * plane, x0);
        load_vec<VEC, true>(p1 + (size_t)k * plane, x1);
#pragma unroll
        for (int v = 0; v < VEC; ++v) y[v] = a * x0[v] + oma * x1[v];
        store_vec<VEC, true>(po + (size_t)k * plane, y);
    }
}

// ------------------------------------------------------------------------------------------
// measurement aid: the shader clock the chip holds while other kernels run.  One wave spins for `ticks` ticks of
// the constant 100 MHz counter (s_memrealtime) and reports how many shader cycles (s_memtime) went by: launched
// on a stream of its own beside the fused loss it reads the clock under THAT load (DVFS lowers it under VALU-dense
// kernels); one wave on one SIMD does not disturb the measured kernels.  out[0] = shader cycles, out[1] = ticks.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_clock_probe(unsigned long long *__restrict__ out, unsigned long long ticks)
{
    if (threadIdx.x != 0) return;
    const unsigned long long r0 = wall_clock64(), c0 = clock64();
    unsigned long long r1 = r0;
    while (r1 - r0 < ticks) {
        __builtin_amdgcn_s_sleep(32);
        r1 = wall_clock64();
    }
    const unsigned long long c1 = clock64();
    out[0] = c1 - c0;
    out[1] = r1 - r0;
}

// ------------------------------------------------------------------------------------------
// measurement aid: the HBM copy rate of THIS box ("fraction of both nominal and measured-copy peak").  A plain
// streaming copy, 16 bytes per lane and access, non-temporal both ways, UNROLL independent loads in flight per lane
// before the first store; a workgroup owns one contiguous UNROLL x 4 KiB chunk.  Bytes moved = 2 x n x 4.
// SVBRDF_COPY_UNROLL / SVBRDF_COPY_NT select the A/B variants of tools/copy_peak.py.  Shipped: UNROLL = 1, non-temporal --
// 6.55-6.59 TB/s on 1 and 4 GiB (profiles/r06_copy_peak.txt; unroll 2: 6.1-6.2, 4: 6.3, 8: 4.4-4.5; plain accesses
// 0.3-0.6 TB/s lower at every unroll; hipMemcpyAsync device-to-device 5.0-5.2; the guide quotes 6.29 for a float4 copy).
// ------------------------------------------------------------------------------------------
template <int UNROLL, bool NT>
__global__ __launch_bounds__(kThreads) void k_copy_vec4(vec4f *__restrict__ dst, const vec4f *__restrict__ src, size_t n4)
{
    const size_t base = (size_t)blockIdx.x * (kThreads * UNROLL) + threadIdx.x;
    vec4f v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
        const size_t i = base + (size_t)u * kThreads;
        if (i < n4) v[u] = NT ? __builtin_nontemporal_load(src + i) : src[i];
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
        const size_t i = base + (size_t)u * kThreads;
        if (i < n4) {
            if (NT) __builtin_nontemporal_store(v[u], dst + i);
            else dst[i] = v[u];
        }
    }
}

#endif  // SVBRDF_TU_MAIN (kernels)

}  // namespace

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------

// the thread-local error text of svbrdf_last_error() lives in the main unit; the auxiliary float64 unit
// (svbrdf_aux_f64.hip) reports through the same two functions
extern "C" __attribute__((visibility("hidden"))) int svbrdf_internal_fail(int code, const char *what);
extern "C" __attribute__((visibility("hidden"))) int svbrdf_internal_launch_status(const char *what);
#if SVBRDF_TU_MAIN
namespace {
thread_local char g_err[256] = "";
std::atomic<unsigned long long> g_launches{0};      // svbrdf_debug_launch_count()
}
int svbrdf_internal_fail(int code, const char *what)
{
    std::snprintf(g_err, sizeof(g_err), "%s", what);
    return code;
}
int svbrdf_internal_launch_status(const char *what)
{
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        std::snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
        return (int)e;
    }
    g_launches.fetch_add(1, std::memory_order_relaxed);      // every entry point enqueues exactly one kernel and ends here
    return 0;
}
#endif

namespace {

#if SVBRDF_TU_MAIN || SVBRDF_TU_AUX
[[maybe_unused]] int fail(int code, const char *what) { return svbrdf_internal_fail(code, what); }

[[maybe_unused]] int check_dims(int B, int S, int H, int W)
{
    if (B <= 0 || S <= 0 || H <= 0 || W <= 0) return fail(SVBRDF_ERR_DIMS, "B, S, H, W must be positive");
    if (H != W) return fail(SVBRDF_ERR_DIMS, "H must equal W (renderers.py:75 transposes the x grid)");
    if (B > 65535) return fail(SVBRDF_ERR_DIMS, "B exceeds 65535 (grid.y)");
    if ((long long)H * W > (1LL << 30)) return fail(SVBRDF_ERR_DIMS, "H*W too large");
    return 0;
}

[[maybe_unused]] bool aligned(const void *p, size_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; }

// widest vector width (<= want) every pointer and the row length allow
[[maybe_unused]] int pick_vec(int want, int W, std::initializer_list<const void *> ptrs)
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

[[maybe_unused]] int env_vec(const char *name, int dflt)
{
    const char *e = std::getenv(name);
    if (!e) return dflt;
    const int v = std::atoi(e);
    return (v == 1 || v == 2 || v == 4) ? v : dflt;
}

[[maybe_unused]] int launch_status(const char *what) { return svbrdf_internal_launch_status(what); }

[[maybe_unused]] dim3 grid_for(int B, int H, int W, int vec)
{
    const long long plane = (long long)H * W;
    const long long per_block = (long long)kThreads * vec;
    return dim3((unsigned)((plane + per_block - 1) / per_block), (unsigned)B, 1);
}

#endif  // SVBRDF_TU_MAIN || SVBRDF_TU_AUX (host helpers)


// launches one K3 variant; `rows` = the host scene table for the by-value kernels (NULL: device table `scenes`).
// WHICH selects the variants this translation unit instantiates: 0 = <G, L1=0, HEAD=0> only, 1 = the other three,
// 2 = all four.
template <bool G, int WHICH>
void launch_k3(bool with_l1, bool head, const float *rows, dim3 grid, size_t lds_bytes, hipStream_t st,
               const float *input, const float *target, const float *scenes, const float *xrow, float eps,
               float inv_count, double loss_scale, float fixed_scale, L1Params l1, float *grad_input,
               unsigned long long *ws, float *loss_out, int B, int S, int H, int W)
{
    SceneBlock block_arg;      // only the first B*S rows are ever read
    if (rows) std::memcpy(block_arg.v, rows, (size_t)B * S * 9 * sizeof(float));
#define SVBRDF_LAUNCH_K3_AS(KERNEL, KERNEL_INL, THREADS)                                                        \
    do {                                                                                                        \
        if (rows)                                                                                               \
            hipLaunchKernelGGL(KERNEL_INL, grid, dim3(THREADS), lds_bytes, st, block_arg, input, target, xrow,  \
                               eps, inv_count, loss_scale, fixed_scale, l1, grad_input, ws, loss_out, S, H, W); \
        else                                                                                                    \
            hipLaunchKernelGGL(KERNEL, grid, dim3(THREADS), lds_bytes, st, input, target, scenes, xrow, eps,    \
                               inv_count, loss_scale, fixed_scale, l1, grad_input, ws, loss_out, S, H, W);      \
    } while (0)
#define SVBRDF_LAUNCH_K3(L, HD)                                                                                 \
    SVBRDF_LAUNCH_K3_AS((k_rendering_loss<G, L, HD>), (k_rendering_loss_inl<G, L, HD>), kLossThreads)
    if constexpr (WHICH != 0) {
        if (head) { if (with_l1) SVBRDF_LAUNCH_K3(true, true); else SVBRDF_LAUNCH_K3(false, true); }
        else if (with_l1) SVBRDF_LAUNCH_K3(true, false);
    }
    if constexpr (WHICH != 1) {
        if (!head && !with_l1) SVBRDF_LAUNCH_K3(false, false);
    }
#undef SVBRDF_LAUNCH_K3
#undef SVBRDF_LAUNCH_K3_AS
}

}  // namespace

// the forward+adjoint variants live in their own translation units (see the top of this file)
#define SVBRDF_K3_ADJOINT_ARGS                                                                                        \
    int with_l1, int head, const float *rows, unsigned gx, unsigned gy, size_t lds_bytes, void *stream,                 \
        const float *input, const float *target, const float *scenes, const float *xrow, float eps, float inv_count,   \
        double loss_scale, float fixed_scale, float l1_sum_scale, float l1_grad_scale, float l1_eps, float *grad_input, \
        unsigned long long *ws, float *loss_out, int B, int S, int H, int W
extern "C" __attribute__((visibility("hidden"))) void svbrdf_internal_launch_k3_adjoint_plain(SVBRDF_K3_ADJOINT_ARGS);
extern "C" __attribute__((visibility("hidden"))) void svbrdf_internal_launch_k3_adjoint_extra(SVBRDF_K3_ADJOINT_ARGS);
#if SVBRDF_TU_ADJOINT_PLAIN
void svbrdf_internal_launch_k3_adjoint_plain(SVBRDF_K3_ADJOINT_ARGS)
{
    launch_k3<true, 0>(with_l1 != 0, head != 0, rows, dim3(gx, gy, 1), lds_bytes,
                       static_cast<hipStream_t>(stream), input, target, scenes, xrow, eps, inv_count, loss_scale, fixed_scale,
                       L1Params{l1_sum_scale, l1_grad_scale, l1_eps}, grad_input, ws, loss_out, B, S, H, W);
}
#endif
#if SVBRDF_TU_ADJOINT_EXTRA
void svbrdf_internal_launch_k3_adjoint_extra(SVBRDF_K3_ADJOINT_ARGS)
{
    launch_k3<true, 1>(with_l1 != 0, head != 0, rows, dim3(gx, gy, 1), lds_bytes,
                       static_cast<hipStream_t>(stream), input, target, scenes, xrow, eps, inv_count, loss_scale, fixed_scale,
                       L1Params{l1_sum_scale, l1_grad_scale, l1_eps}, grad_input, ws, loss_out, B, S, H, W);
}
#endif


#if SVBRDF_TU_MAIN
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

// `host_rows` > 0: `scenes` is a HOST table of that many rows and travels by value in the argument block
static int render_fwd_impl(const float *maps, const float *scenes, const float *xrow, float *out, const int *offsets,
                           int shared, int host_rows, int B, int S, int H, int W, void *stream)
{
    if (!maps || !scenes || !xrow || !out) return fail(SVBRDF_ERR_NULL, "render_fwd: null pointer");
    if (int e = check_dims(B, S, H, W)) return e;
    if (!aligned(maps, 4) || !aligned(scenes, 4) || !aligned(xrow, 4) || !aligned(out, 4) || !aligned(offsets, 4))
        return fail(SVBRDF_ERR_ALIGN, "render_fwd: pointers must be 4-byte aligned");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int vec = pick_vec(env_vec("SVBRDF_K1_VEC", 4), W, {maps, xrow, out});
    const dim3 grid = grid_for(B, H, W, vec), block(kThreads);
    if (host_rows > 0) {
        SceneBlock block_arg;      // only the first host_rows rows are ever read
        std::memcpy(block_arg.v, scenes, (size_t)host_rows * 9 * sizeof(float));
        if (vec == 4) hipLaunchKernelGGL(k_render_fwd_inl<4>, grid, block, 0, st, block_arg, maps, xrow, out, shared, S, H, W);
        else if (vec == 2) hipLaunchKernelGGL(k_render_fwd_inl<2>, grid, block, 0, st, block_arg, maps, xrow, out, shared, S, H, W);
        else hipLaunchKernelGGL(k_render_fwd_inl<1>, grid, block, 0, st, block_arg, maps, xrow, out, shared, S, H, W);
        return launch_status("render_fwd_host_scenes launch");
    }
    if (vec == 4) hipLaunchKernelGGL(k_render_fwd<4>, grid, block, 0, st, maps, scenes, xrow, out, offsets, shared, S, H, W);
    else if (vec == 2) hipLaunchKernelGGL(k_render_fwd<2>, grid, block, 0, st, maps, scenes, xrow, out, offsets, shared, S, H, W);
    else hipLaunchKernelGGL(k_render_fwd<1>, grid, block, 0, st, maps, scenes, xrow, out, offsets, shared, S, H, W);
    return launch_status("render_fwd launch");
}

static int host_rows_for(const char *who, int scenes_shared, int B, int S, int *rows)
{
    const long long n = scenes_shared ? (long long)S : (long long)B * S;
    if (n > SVBRDF_HOST_SCENES_MAX_ROWS)
        return fail(SVBRDF_ERR_DIMS, who);
    *rows = (int)n;
    return 0;
}

int svbrdf_render_fwd(const float *maps, const float *scenes, const float *xrow, float *out,
                      int B, int S, int H, int W, void *stream)
{
    return render_fwd_impl(maps, scenes, xrow, out, nullptr, 0, 0, B, S, H, W, stream);
}

int svbrdf_render_fwd_host_scenes(const float *maps, const float *scenes_host, int scenes_shared, const float *xrow,
                                  float *out, int B, int S, int H, int W, void *stream)
{
    if (int e = check_dims(B, S, H, W)) return e;
    int rows = 0;
    if (int e = host_rows_for("render_fwd_host_scenes: the table exceeds SVBRDF_HOST_SCENES_MAX_ROWS (upload it and use "
                              "svbrdf_render_fwd)", scenes_shared, B, S, &rows)) return e;
    return render_fwd_impl(maps, scenes_host, xrow, out, nullptr, scenes_shared != 0, rows, B, S, H, W, stream);
}

int svbrdf_render_fwd_ragged(const float *maps, const float *scenes, const int *offsets, const float *xrow, float *out,
                             int B, int R, int H, int W, void *stream)
{
    if (!offsets) return fail(SVBRDF_ERR_NULL, "render_fwd_ragged: offsets is null");
    if (R < 0) return fail(SVBRDF_ERR_DIMS, "render_fwd_ragged: R must be >= 0");
    if (R == 0) return check_dims(B, 1, H, W);          // nothing to render
    return render_fwd_impl(maps, scenes, xrow, out, offsets, 0, 0, B, 1, H, W, stream);
}

// K1 + sensor-noise epilogue (render_inputs, dataset.py:206-219).  `sigma` null: clamp only.
static int render_inputs_impl(const char *who, bool on_host, const float *maps, const float *scenes, const float *sigma,
                              unsigned long long seed, unsigned long long offset, const float *xrow, float *out,
                              int B, int S, int H, int W, void *stream)
{
    if (!maps || !scenes || !xrow || !out) return fail(SVBRDF_ERR_NULL, "render_inputs: null pointer");
    if (int e = check_dims(B, S, H, W)) return e;
    if (on_host && (long long)B * S > SVBRDF_HOST_SCENES_MAX_ROWS)
        return fail(SVBRDF_ERR_DIMS, "render_inputs_host_scenes: the table exceeds SVBRDF_HOST_SCENES_MAX_ROWS (upload scenes "
                                     "and noise levels and use svbrdf_render_inputs)");
    if (!aligned(maps, 4) || !aligned(scenes, 4) || !aligned(sigma, 4) || !aligned(xrow, 4) || !aligned(out, 4))
        return fail(SVBRDF_ERR_ALIGN, "render_inputs: pointers must be 4-byte aligned");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int vec = pick_vec(env_vec("SVBRDF_K1_VEC", 4), W, {maps, xrow, out});
    const dim3 grid = grid_for(B, H, W, vec), block(kThreads);
    const PhiloxKey key{(unsigned)seed, (unsigned)(seed >> 32), (unsigned)offset, (unsigned)(offset >> 32)};
#define SVBRDF_LAUNCH_PHOTOS(V, E)                                                                                     \
    do {                                                                                                               \
        if (on_host)                                                                                                   \
            hipLaunchKernelGGL((k_render_inputs_inl<V, E>), grid, block, 0, st, block_arg, key, maps, xrow, out, S, H, W); \
        else                                                                                                           \
            hipLaunchKernelGGL((k_render_inputs<V, E>), grid, block, 0, st, maps, scenes, sigma, key, xrow, out, S, H, W); \
    } while (0)
    PhotoBlock block_arg;          // only the first B*S rows / levels are ever read
    if (on_host) {
        std::memcpy(block_arg.v, scenes, (size_t)B * S * 9 * sizeof(float));
        if (sigma) std::memcpy(block_arg.sigma, sigma, (size_t)B * S * sizeof(float));
    }
    if (sigma) {
        if (vec == 4) SVBRDF_LAUNCH_PHOTOS(4, 2); else if (vec == 2) SVBRDF_LAUNCH_PHOTOS(2, 2); else SVBRDF_LAUNCH_PHOTOS(1, 2);
    } else {
        if (vec == 4) SVBRDF_LAUNCH_PHOTOS(4, 1); else if (vec == 2) SVBRDF_LAUNCH_PHOTOS(2, 1); else SVBRDF_LAUNCH_PHOTOS(1, 1);
    }
#undef SVBRDF_LAUNCH_PHOTOS
    return launch_status(who);
}

int svbrdf_render_inputs(const float *maps, const float *scenes, const float *noise_std, unsigned long long seed,
                         unsigned long long offset, const float *xrow, float *out, int B, int S, int H, int W, void *stream)
{
    return render_inputs_impl("render_inputs launch", false, maps, scenes, noise_std, seed, offset, xrow, out, B, S, H, W,
                              stream);
}

int svbrdf_render_inputs_host_scenes(const float *maps, const float *scenes_host, const float *noise_std_host,
                                     unsigned long long seed, unsigned long long offset, const float *xrow, float *out,
                                     int B, int S, int H, int W, void *stream)
{
    return render_inputs_impl("render_inputs_host_scenes launch", true, maps, scenes_host, noise_std_host, seed, offset,
                              xrow, out, B, S, H, W, stream);
}

static int render_bwd_impl(const float *maps, const float *scenes, const float *xrow, const float *grad_out,
                           float *grad_maps, const int *offsets, int shared, int host_rows, int B, int S, int H, int W,
                           void *stream)
{
    if (!maps || !scenes || !xrow || !grad_out || !grad_maps) return fail(SVBRDF_ERR_NULL, "render_bwd: null pointer");
    if (int e = check_dims(B, S, H, W)) return e;
    if (!aligned(maps, 4) || !aligned(scenes, 4) || !aligned(xrow, 4) || !aligned(grad_out, 4) || !aligned(grad_maps, 4) ||
        !aligned(offsets, 4))
        return fail(SVBRDF_ERR_ALIGN, "render_bwd: pointers must be 4-byte aligned");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int vec = pick_vec(env_vec("SVBRDF_K2_VEC", 2), W, {maps, xrow, grad_out, grad_maps});
    const dim3 grid = grid_for(B, H, W, vec), block(kThreads);
    if (host_rows > 0) {
        SceneBlock block_arg;
        std::memcpy(block_arg.v, scenes, (size_t)host_rows * 9 * sizeof(float));
        if (vec == 4) hipLaunchKernelGGL(k_render_bwd_inl<4>, grid, block, 0, st, block_arg, maps, xrow, grad_out, grad_maps, shared, S, H, W);
        else if (vec == 2) hipLaunchKernelGGL(k_render_bwd_inl<2>, grid, block, 0, st, block_arg, maps, xrow, grad_out, grad_maps, shared, S, H, W);
        else hipLaunchKernelGGL(k_render_bwd_inl<1>, grid, block, 0, st, block_arg, maps, xrow, grad_out, grad_maps, shared, S, H, W);
        return launch_status("render_bwd_host_scenes launch");
    }
    if (vec == 4) hipLaunchKernelGGL(k_render_bwd<4>, grid, block, 0, st, maps, scenes, xrow, grad_out, grad_maps, offsets, shared, S, H, W);
    else if (vec == 2) hipLaunchKernelGGL(k_render_bwd<2>, grid, block, 0, st, maps, scenes, xrow, grad_out, grad_maps, offsets, shared, S, H, W);
    else hipLaunchKernelGGL(k_render_bwd<1>, grid, block, 0, st, maps, scenes, xrow, grad_out, grad_maps, offsets, shared, S, H, W);
    return launch_status("render_bwd launch");
}

int svbrdf_render_bwd(const float *maps, const float *scenes, const float *xrow, const float *grad_out,
                      float *grad_maps, int B, int S, int H, int W, void *stream)
{
    return render_bwd_impl(maps, scenes, xrow, grad_out, grad_maps, nullptr, 0, 0, B, S, H, W, stream);
}

int svbrdf_render_bwd_host_scenes(const float *maps, const float *scenes_host, int scenes_shared, const float *xrow,
                                  const float *grad_out, float *grad_maps, int B, int S, int H, int W, void *stream)
{
    if (int e = check_dims(B, S, H, W)) return e;
    int rows = 0;
    if (int e = host_rows_for("render_bwd_host_scenes: the table exceeds SVBRDF_HOST_SCENES_MAX_ROWS (upload it and use "
                              "svbrdf_render_bwd)", scenes_shared, B, S, &rows)) return e;
    return render_bwd_impl(maps, scenes_host, xrow, grad_out, grad_maps, nullptr, scenes_shared != 0, rows, B, S, H, W, stream);
}

int svbrdf_render_bwd_ragged(const float *maps, const float *scenes, const int *offsets, const float *xrow,
                             const float *grad_out, float *grad_maps, int B, int R, int H, int W, void *stream)
{
    if (!offsets) return fail(SVBRDF_ERR_NULL, "render_bwd_ragged: offsets is null");
    if (R < 0) return fail(SVBRDF_ERR_DIMS, "render_bwd_ragged: R must be >= 0");
    // R == 0 still launches: every map's gradient is written (zeros)
    const float *go = (R == 0 && !grad_out) ? maps : grad_out, *sc = (R == 0 && !scenes) ? maps : scenes;
    return render_bwd_impl(maps, sc, xrow, go, grad_maps, offsets, 0, 0, B, 1, H, W, stream);
}

size_t svbrdf_rendering_loss_workspace_bytes(int B, int S, int H, int W)
{
    if (B <= 0 || S <= 0 || H <= 0 || W <= 0) return 0;
    return (kLossSlots + 1) * sizeof(unsigned long long);   // sharded fixed-point accumulators + ticket
}

static int loss_impl(const char *who, bool head, bool scenes_on_host, const float *input, const float *target,
                     const float *scenes, const float *xrow, float eps, float l1_weight, float eps_l1,
                     float *loss_out, float *grad_input, void *workspace, size_t workspace_bytes, int B, int S,
                     int H, int W, void *stream)
{
    if (!input || !target || !scenes || !xrow || !loss_out || !workspace) return fail(SVBRDF_ERR_NULL, who);
    if (int e = check_dims(B, S, H, W)) return e;
    if (!(eps >= 1e-9f) || !(eps <= 1e9f))
        return fail(SVBRDF_ERR_DIMS, "loss: eps_render must lie in [1e-9, 1e9] (the reference uses 0.1)");
    if (scenes_on_host && (long long)B * S > SVBRDF_HOST_SCENES_MAX_ROWS)
        return fail(SVBRDF_ERR_DIMS, "host_scenes: B*S exceeds SVBRDF_HOST_SCENES_MAX_ROWS (upload the table and use the device-pointer entry)");
    if (!aligned(input, 4) || !aligned(target, 4) || !aligned(scenes, 4) || !aligned(xrow, 4) ||
        !aligned(loss_out, 4) || !aligned(workspace, 8) || (grad_input && !aligned(grad_input, 4)))
        return fail(SVBRDF_ERR_ALIGN, "loss: pointers must be 4-byte aligned (workspace 8-byte)");
    if (workspace_bytes < svbrdf_rendering_loss_workspace_bytes(B, S, H, W))
        return fail(SVBRDF_ERR_WORKSPACE, "loss: workspace too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const long long plane = (long long)H * W;
    if (plane > (1LL << 25))
        return fail(SVBRDF_ERR_DIMS, "loss: H*W exceeds 2^25 (one item's 12 planes are addressed with 32-bit byte offsets)");
    const dim3 grid((unsigned)((plane + kLossThreads - 1) / kLossThreads), (unsigned)B, 1);    // 256 pixels per workgroup
    const double count = (double)B * S * 3.0 * (double)plane;
    const float inv_count = (float)(1.0 / count);
    unsigned long long *ws = static_cast<unsigned long long *>(workspace);
    // Fixed-point scale 2^k of the per-workgroup partial sums: as fine as 2^-24, coarser only if
    // a slot could otherwise outgrow its 48 bits (|dlog| <= 32 per term is far beyond any
    // radiance this renderer can produce: log(1e13/0.1)).
    int k = 24;
    const double l1_share = 1.0 + 4.0 * std::fabs((double)l1_weight);
    const double worst_per_slot = (count * 32.0 / (double)kLossSlots + 32.0 * kLossThreads * 3 * S) * l1_share;
    while (k > 0 && worst_per_slot * std::ldexp(1.0, k) >= std::ldexp(1.0, kLossCountShift - 1)) --k;
    const float fixed_scale = (float)std::ldexp(1.0, k);
    const double loss_scale = std::ldexp(1.0, -k) / count;
    if ((unsigned long long)grid.x * grid.y >= (1ULL << 16) * kLossSlots)
        return fail(SVBRDF_ERR_DIMS, "loss: too many workgroups for the arrival counters");
    const size_t lds_bytes = grad_input ? 0 : (size_t)S * 9 * sizeof(float);   // forward-only kernels stage scenes in LDS
    if (lds_bytes > 60 * 1024) return fail(SVBRDF_ERR_DIMS, "loss: too many scenes per item for the LDS stage (max 1706)");
    const L1Params l1{l1_weight * (float)S, (float)((double)l1_weight / ((double)B * 3.0 * (double)plane)), eps_l1};
    const float *rows = scenes_on_host ? scenes : nullptr;
    if (grad_input)
        (l1_weight != 0.0f || head ? svbrdf_internal_launch_k3_adjoint_extra : svbrdf_internal_launch_k3_adjoint_plain)(
            l1_weight != 0.0f, head, rows, grid.x, grid.y, lds_bytes, stream, input, target, scenes, xrow, eps,
            inv_count, loss_scale, fixed_scale, l1.sum_scale, l1.grad_scale, l1.eps, grad_input, ws, loss_out, B, S, H, W);
    else
        launch_k3<false, 2>(l1_weight != 0.0f, head, rows, grid, lds_bytes, st, input, target, scenes, xrow, eps,
                            inv_count, loss_scale, fixed_scale, l1, grad_input, ws, loss_out, B, S, H, W);
    return launch_status(who);
}

int svbrdf_rendering_loss_fwd_bwd(const float *input, const float *target, const float *scenes,
                                  const float *xrow, float eps, float *loss_out, float *grad_input,
                                  void *workspace, size_t workspace_bytes, int B, int S, int H, int W,
                                  void *stream)
{
    return loss_impl("rendering_loss", false, false, input, target, scenes, xrow, eps, 0.0f, 0.01f, loss_out, grad_input,
                     workspace, workspace_bytes, B, S, H, W, stream);
}

int svbrdf_mixed_loss_fwd_bwd(const float *input, const float *target, const float *scenes, const float *xrow,
                              float eps_render, float l1_weight, float eps_l1, float *loss_out,
                              float *grad_input, void *workspace, size_t workspace_bytes, int B, int S, int H,
                              int W, void *stream)
{
    return loss_impl("mixed_loss", false, false, input, target, scenes, xrow, eps_render, l1_weight, eps_l1, loss_out,
                     grad_input, workspace, workspace_bytes, B, S, H, W, stream);
}

int svbrdf_head_loss_fwd_bwd(const float *encoded9, const float *target, const float *scenes, const float *xrow,
                             float eps_render, float l1_weight, float eps_l1, float *loss_out, float *grad_encoded9,
                             void *workspace, size_t workspace_bytes, int B, int S, int H, int W, void *stream)
{
    return loss_impl("head_loss", true, false, encoded9, target, scenes, xrow, eps_render, l1_weight, eps_l1, loss_out,
                     grad_encoded9, workspace, workspace_bytes, B, S, H, W, stream);
}

int svbrdf_host_scenes_max_rows(void) { return SVBRDF_HOST_SCENES_MAX_ROWS; }

int svbrdf_mixed_loss_fwd_bwd_host_scenes(const float *input, const float *target, const float *scenes_host,
                                          const float *xrow, float eps_render, float l1_weight, float eps_l1,
                                          float *loss_out, float *grad_input, void *workspace, size_t workspace_bytes,
                                          int B, int S, int H, int W, void *stream)
{
    return loss_impl("mixed_loss_host_scenes", false, true, input, target, scenes_host, xrow, eps_render, l1_weight,
                     eps_l1, loss_out, grad_input, workspace, workspace_bytes, B, S, H, W, stream);
}

int svbrdf_head_loss_fwd_bwd_host_scenes(const float *encoded9, const float *target, const float *scenes_host,
                                         const float *xrow, float eps_render, float l1_weight, float eps_l1,
                                         float *loss_out, float *grad_encoded9, void *workspace, size_t workspace_bytes,
                                         int B, int S, int H, int W, void *stream)
{
    return loss_impl("head_loss_host_scenes", true, true, encoded9, target, scenes_host, xrow, eps_render, l1_weight,
                     eps_l1, loss_out, grad_encoded9, workspace, workspace_bytes, B, S, H, W, stream);
}

int svbrdf_scale_inplace(float *data, const float *scale_dev, size_t n, void *stream)
{
    // The common case (upstream gradient 1) exits at once, so the launch itself is the cost: 1024 workgroups
    // (4 per CU, grid-stride) still stream at HBM rate when the scale is not 1 and dispatch in a third of the
    // time of 4096.
    constexpr size_t kScaleBlocks = SVBRDF_SCALE_BLOCKS;
    if (!data || !scale_dev) return fail(SVBRDF_ERR_NULL, "scale_inplace: null pointer");
    if (n == 0) return 0;
    const size_t blocks = (n + kThreads - 1) / kThreads;
    hipLaunchKernelGGL(k_scale_inplace, dim3((unsigned)(blocks < kScaleBlocks ? blocks : kScaleBlocks)), dim3(kThreads), 0,
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

int svbrdf_mix_materials(const float *svbrdf0, const float *svbrdf1, const float *alpha, float *out, int B, int H, int W,
                         void *stream)
{
    if (!svbrdf0 || !svbrdf1 || !alpha || !out) return fail(SVBRDF_ERR_NULL, "mix_materials: null pointer");
    if (B <= 0 || H <= 0 || W <= 0 || B > 65535 || (long long)H * W > (1LL << 30))
        return fail(SVBRDF_ERR_DIMS, "mix_materials: bad dimensions");
    if (!aligned(svbrdf0, 4) || !aligned(svbrdf1, 4) || !aligned(alpha, 4) || !aligned(out, 4))
        return fail(SVBRDF_ERR_ALIGN, "mix_materials: pointers must be 4-byte aligned");
    const size_t plane = (size_t)H * W;
    int vec = 4;            // whole planes are contiguous here: the vector width only needs H*W and the bases to agree
    while (vec > 1 && !((plane % vec) == 0 && aligned(svbrdf0, 4 * vec) && aligned(svbrdf1, 4 * vec) && aligned(out, 4 * vec))) vec >>= 1;
    const dim3 grid((unsigned)((plane + (size_t)kThreads * vec - 1) / ((size_t)kThreads * vec)), (unsigned)B, 1), block(kThreads);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (vec == 4) hipLaunchKernelGGL(k_mix_materials<4>, grid, block, 0, st, svbrdf0, svbrdf1, alpha, out, plane);
    else if (vec == 2) hipLaunchKernelGGL(k_mix_materials<2>, grid, block, 0, st, svbrdf0, svbrdf1, alpha, out, plane);
    else hipLaunchKernelGGL(k_mix_materials<1>, grid, block, 0, st, svbrdf0, svbrdf1, alpha, out, plane);
    return launch_status("mix_materials launch");
}

int svbrdf_debug_copy(float *dst, const float *src, size_t n, void *stream)
{
    if (!dst || !src) return fail(SVBRDF_ERR_NULL, "debug_copy: null pointer");
    if (n == 0 || (n & 3) != 0) return fail(SVBRDF_ERR_DIMS, "debug_copy: n must be a positive multiple of 4 floats");
    if (!aligned(dst, 16) || !aligned(src, 16)) return fail(SVBRDF_ERR_ALIGN, "debug_copy: pointers must be 16-byte aligned");
    const int unroll = [] { const char *e = std::getenv("SVBRDF_COPY_UNROLL"); const int v = e ? std::atoi(e) : 0;
                            return (v == 1 || v == 2 || v == 4 || v == 8) ? v : 1; }();
    const bool nt = [] { const char *e = std::getenv("SVBRDF_COPY_NT"); return !(e && e[0] == '0'); }();
    const size_t n4 = n / 4, per_block = (size_t)kThreads * unroll, blocks = (n4 + per_block - 1) / per_block;
    if (blocks > 0x7fffffffULL) return fail(SVBRDF_ERR_DIMS, "debug_copy: n too large");
    const dim3 grid((unsigned)blocks), block(kThreads);
    hipStream_t st = static_cast<hipStream_t>(stream);
    vec4f *d = reinterpret_cast<vec4f *>(dst);
    const vec4f *sp = reinterpret_cast<const vec4f *>(src);
#define SVBRDF_COPY(U)                                                                            \
    do {                                                                                          \
        if (nt) hipLaunchKernelGGL((k_copy_vec4<U, true>), grid, block, 0, st, d, sp, n4);        \
        else hipLaunchKernelGGL((k_copy_vec4<U, false>), grid, block, 0, st, d, sp, n4);          \
    } while (0)
    if (unroll == 8) SVBRDF_COPY(8); else if (unroll == 4) SVBRDF_COPY(4); else if (unroll == 2) SVBRDF_COPY(2); else SVBRDF_COPY(1);
#undef SVBRDF_COPY
    return launch_status("debug_copy launch");
}

unsigned long long svbrdf_debug_launch_count(void) { return g_launches.load(std::memory_order_relaxed); }

int svbrdf_debug_clock_probe(unsigned long long *out_dev, unsigned long long ticks_100mhz, void *stream)
{
    if (!out_dev) return fail(SVBRDF_ERR_NULL, "clock_probe: null pointer");
    if (ticks_100mhz == 0 || ticks_100mhz > 100000000ULL) return fail(SVBRDF_ERR_DIMS, "clock_probe: 0 < ticks <= 1e8 (1 s)");
    hipLaunchKernelGGL(k_clock_probe, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), out_dev, ticks_100mhz);
    return launch_status("clock_probe launch");
}

}  // extern "C"
#endif  // SVBRDF_TU_MAIN
