// svbrdf_aux_f64.hip -- AUXILIARY translation unit of libsvbrdf_hip.so: float64 maps and second order.
//
// NOT part of the north-star path (BASELINE.json: fp32 maps, first-order training).  The reference never feeds double maps
// or create_graph=True to its renderer; these kernels exist because render() there is dtype-agnostic and twice
// differentiable as a property of being built from torch ops (renderers.py:67-104).  Frozen since round 4: nothing is added
// here, and the code lives in a file and a translation unit of its own so that edits cannot move the schedule or the
// compile time of the kernels that matter (svbrdf_kernels.hip: K1-K4; csrc/Makefile).
//
// The shared per-pixel device code (geometry(), pixel_coords(), load_scene(), make_vconst(): float32, exact-rounded) and
// the host-side argument checks come from svbrdf_kernels.hip, included with SVBRDF_TU = 4: that selects none of its
// kernels and none of its C ABI entry points, only the inline device functions and host helpers.
#define SVBRDF_TU 4
#include "svbrdf_kernels.hip"

// ------------------------------------------------------------------------------------------
// float64 maps.  The reference's render() is dtype-agnostic (renderers.py:67-104), but what it does with double maps is
// MIXED precision: the pixel grid is torch.linspace's default float32 (:73) and camera / light positions and the light
// colour go through torch.Tensor(...) = float32 (:79, :91, :98), so wo, wi, h, VH, (1-VH)^5 and colour * falloff are
// computed in float32 exactly as for float32 maps, and only what touches the maps -- the dot products with the normal,
// D, G, F, the BRDF, the radiance -- is promoted to double.  Same here: geometry() as above (float32, exact-rounded),
// shading in double, op by op in the reference's order (no algebraic merging: this is the slow path of a caller who
// asked for double -- gradient checks, notebooks -- not the training path).  K1 / K2 only; the losses compose them
// through autograd (losses.py).  One pixel per thread.
// ------------------------------------------------------------------------------------------
namespace {
constexpr double kPiD = 3.14159265358979323846;     // math.pi, renderers.py:20,27 (a python float: double in double ops)

// Scalar of the shading code: `double`, or `Dual` = value + directional derivative (forward-mode), which turns the
// adjoint below into its own derivative along a direction u of the maps: the tangent of the rendering is J u and the
// tangent of the accumulated gradient is d/dmaps <J^T grad_out, u> -- exactly the two products autograd needs to
// differentiate THROUGH the backward of render() (create_graph=True; the reference gets them from plain autograd,
// renderers.py:67-104 being built from differentiable ops).  Masks (clamps, sub-gradient selections) compare values and
// have zero derivative, as in torch's double-backward formulas.
struct Dual {
    double v, d;
};
__device__ __forceinline__ Dual operator+(Dual a, Dual b) { return Dual{a.v + b.v, a.d + b.d}; }
[[maybe_unused]] __device__ __forceinline__ Dual operator+(Dual a, double b) { return Dual{a.v + b, a.d}; }
__device__ __forceinline__ Dual operator+(double a, Dual b) { return Dual{a + b.v, b.d}; }
__device__ __forceinline__ Dual operator-(Dual a, Dual b) { return Dual{a.v - b.v, a.d - b.d}; }
__device__ __forceinline__ Dual operator-(Dual a, double b) { return Dual{a.v - b, a.d}; }
__device__ __forceinline__ Dual operator-(double a, Dual b) { return Dual{a - b.v, -b.d}; }
__device__ __forceinline__ Dual operator-(Dual a) { return Dual{-a.v, -a.d}; }
__device__ __forceinline__ Dual operator*(Dual a, Dual b) { return Dual{a.v * b.v, a.d * b.v + a.v * b.d}; }
__device__ __forceinline__ Dual operator*(Dual a, double b) { return Dual{a.v * b, a.d * b}; }
__device__ __forceinline__ Dual operator*(double a, Dual b) { return Dual{a * b.v, a * b.d}; }
__device__ __forceinline__ Dual operator/(Dual a, Dual b)
{
    const double q = a.v / b.v;
    return Dual{q, (a.d - q * b.d) / b.v};
}
__device__ __forceinline__ Dual operator/(Dual a, double b) { return Dual{a.v / b, a.d / b}; }
__device__ __forceinline__ Dual operator/(double a, Dual b)
{
    const double q = a / b.v;
    return Dual{q, -q * b.d / b.v};
}
__device__ __forceinline__ Dual &operator+=(Dual &a, Dual b) { a.v += b.v; a.d += b.d; return a; }
__device__ __forceinline__ Dual &operator-=(Dual &a, Dual b) { a.v -= b.v; a.d -= b.d; return a; }
__device__ __forceinline__ double value_of(double x) { return x; }
__device__ __forceinline__ double value_of(Dual x) { return x.v; }
__device__ __forceinline__ double sqrt_(double x) { return sqrt(x); }
__device__ __forceinline__ Dual sqrt_(Dual x)
{
    const double r = sqrt(x.v);
    return Dual{r, x.d / (2.0 * r)};
}
// torch.clamp(x, min=c): the gradient passes iff x >= c
__device__ __forceinline__ double clamp_min(double x, double c) { return fmax(x, c); }
__device__ __forceinline__ Dual clamp_min(Dual x, double c) { return x.v >= c ? x : Dual{c, 0.0}; }
template <typename T> __device__ __forceinline__ T zero_of();
template <> __device__ __forceinline__ double zero_of<double>() { return 0.0; }
template <> __device__ __forceinline__ Dual zero_of<Dual>() { return Dual{0.0, 0.0}; }

template <typename T>
struct MapsT {
    T n[3], d[3], r[3], s[3];
};
using MapsD = MapsT<double>;

__device__ __forceinline__ void load_maps_f64(const double *__restrict__ base, size_t plane, size_t pix, MapsD &m)
{
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        m.n[k] = base[(size_t)(0 + k) * plane + pix];
        m.d[k] = base[(size_t)(3 + k) * plane + pix];
        m.r[k] = base[(size_t)(6 + k) * plane + pix];
        m.s[k] = base[(size_t)(9 + k) * plane + pix];
    }
}

// torch.sum(a*b, dim=-3) with a double and b float32-valued: products in double, summed (p0+p1)+p2
template <typename T>
__device__ __forceinline__ T dot3d(const T a[3], double bx, double by, double bz)
{
    const T p0 = a[0] * bx, p1 = a[1] * by, p2 = a[2] * bz;
    return (p0 + p1) + p2;
}

// forward of one (pixel, scene) in the reference's operation order; with grad_rad != nullptr also the adjoint, accumulated
// into g (SURVEY.md section 8a's backward, PyTorch's sub-gradient conventions: clamp(min=m) passes iff x >= m)
template <typename T>
__device__ __forceinline__ void shade_f64(const Geom &g, const MapsT<T> &m, T rad[3], const T *grad_rad, MapsT<T> *acc)
{
    const double wo[3] = {g.wox, g.woy, g.woz}, wi[3] = {g.wix, g.wiy, g.wiz}, h[3] = {g.hx, g.hy, g.hz};
    const T nh_raw = dot3d(m.n, h[0], h[1], h[2]);
    const T vn_raw = dot3d(m.n, wo[0], wo[1], wo[2]);               // dot_product(wo, normals): the products commute
    const T ln_raw = dot3d(m.n, wi[0], wi[1], wi[2]);
    const T NH = clamp_min(nh_raw, 0.001), VN = clamp_min(vn_raw, 0.001), LN = clamp_min(ln_raw, 0.001), LNp = clamp_min(ln_raw, 0.0);
    const double p = g.p;                                           // (1 - VH)^5, float32 (see the header above)
    const T NH2 = NH * NH, VN2 = VN * VN, LN2 = LN * LN;
    const T qV = (1.0 - VN2) / VN2, qL = (1.0 - LN2) / LN2, qN = (1.0 - NH2) / NH2;
    const T four = 4.0 * VN * LN;
    T g_VN = zero_of<T>(), g_LN = zero_of<T>(), g_NH = zero_of<T>(), g_LNp = zero_of<T>();
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const T r = clamp_min(m.r[k], 0.001);                       // renderers.py:87
        const T a = r * r, A = a * a;                               // roughness**2, alpha**2
        const T F = m.s[k] + (1.0 - m.s[k]) * p;                    // :29-32
        const T wV = sqrt_(1.0 + A * qV), wL = sqrt_(1.0 + A * qL);
        const T G1V = 2.0 / (1.0 + wV), G1L = 2.0 / (1.0 + wL);     // xi() == 1: its arguments are clamped >= 1e-3
        const T G = G1V * G1L;
        const T den_raw = NH2 * (A + qN);
        const T den = clamp_min(den_raw, 0.001);
        const T D = A / (kPiD * (den * den));
        const T spec = F * G * D / four;                            // :62
        const T diff = (1.0 - F) * m.d[k] / kPiD;                   // :18-20
        const T f = diff + spec;
        const double E = g.E[k];                                    // light colour * falloff, float32
        rad[k] = (f * E) * LNp;                                     // :100
        if (grad_rad) {
            const T g_f = grad_rad[k] * E * LNp;
            g_LNp += grad_rad[k] * f * E;
            acc->d[k] += g_f * (1.0 - F) / kPiD;
            const T g_F = g_f * (G * D / four - m.d[k] / kPiD);
            acc->s[k] += g_F * (1.0 - p);
            const T g_G = g_f * F * D / four, g_D = g_f * F * G / four;
            g_VN -= g_f * spec / VN;                                // the 1/(4 VN LN) factor
            g_LN -= g_f * spec / LN;
            // G1 = 2/(1+w), w = sqrt(1 + A q):  dG1/dA = -q G1^2/(4w),  dG1/dq = -A G1^2/(4w)
            const T dV = -G1V * G1V / (4.0 * wV), dL = -G1L * G1L / (4.0 * wL);
            T g_A = g_G * (G1L * dV * qV + G1V * dL * qL);
            const T g_qV = g_G * G1L * dV * A, g_qL = g_G * G1V * dL * A;
            g_VN += g_qV * (-2.0 / (VN2 * VN));                     // q = (1 - X^2)/X^2 = X^-2 - 1
            g_LN += g_qL * (-2.0 / (LN2 * LN));
            // D = A/(pi den^2), den = clamp(NH^2 (A + (1-NH^2)/NH^2)) = clamp(NH^2 A + 1 - NH^2)
            g_A += g_D / (kPiD * (den * den));
            const T g_den = (value_of(den_raw) >= 0.001) ? -2.0 * g_D * A / (kPiD * den * den * den) : zero_of<T>();
            g_A += g_den * NH2;
            g_NH += g_den * (A - 1.0) * 2.0 * NH;
            acc->r[k] += (value_of(m.r[k]) >= 0.001) ? g_A * 4.0 * (a * r) : zero_of<T>();
        }
    }
    if (grad_rad) {
        if (!(value_of(nh_raw) >= 0.001)) g_NH = zero_of<T>();
        if (!(value_of(vn_raw) >= 0.001)) g_VN = zero_of<T>();
        if (!(value_of(ln_raw) >= 0.001)) g_LN = zero_of<T>();
        if (!(value_of(ln_raw) >= 0.0)) g_LNp = zero_of<T>();
        const T gl = g_LN + g_LNp;
#pragma unroll
        for (int c = 0; c < 3; ++c) acc->n[c] += g_NH * h[c] + g_VN * wo[c] + gl * wi[c];
    }
}

template <bool BWD>
__global__ __launch_bounds__(kThreads) void k_render_f64(const double *__restrict__ maps, const float *__restrict__ scenes,
                                                         const float *__restrict__ xrow, const double *__restrict__ grad_out,
                                                         double *__restrict__ out, int S, int H, int W)
{
    const size_t plane = (size_t)H * W;
    const size_t pix = (size_t)blockIdx.x * kThreads + threadIdx.x;
    const int b = blockIdx.y;
    if (pix >= plane) return;
    MapsD m, acc;
    load_maps_f64(maps + (size_t)b * 12 * plane, plane, pix, m);
#pragma unroll
    for (int k = 0; k < 3; ++k) acc.n[k] = acc.d[k] = acc.r[k] = acc.s[k] = 0.0;
    float x[1], y;
    pixel_coords<1>(xrow, pix, W, x, y);
    const VConst K = make_vconst();
    for (int s = 0; s < S; ++s) {
        float sc[9];
        load_scene(scenes + ((size_t)b * S + s) * 9, sc);
        const Geom g = geometry(K, sc, x[0], y);
        double rad[3];
        const size_t o = (((size_t)b * S + s) * 3) * plane + pix;
        if (BWD) {
            const double gr[3] = {grad_out[o], grad_out[o + plane], grad_out[o + 2 * plane]};
            shade_f64<double>(g, m, rad, gr, &acc);
        } else {
            shade_f64<double>(g, m, rad, nullptr, nullptr);
#pragma unroll
            for (int k = 0; k < 3; ++k) out[o + (size_t)k * plane] = rad[k];
        }
    }
    if (BWD) {
        double *__restrict__ gm = out + (size_t)b * 12 * plane + pix;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            gm[(size_t)(0 + k) * plane] = acc.n[k];
            gm[(size_t)(3 + k) * plane] = acc.d[k];
            gm[(size_t)(6 + k) * plane] = acc.r[k];
            gm[(size_t)(9 + k) * plane] = acc.s[k];
        }
    }
}
// Derivative of the render's backward along a direction `tangent` of the maps (see Dual): per pixel, S dual-number
// evaluations of shade_f64.  grad_maps_tangent [B,12,H,W] = d/dmaps <J^T grad_out, tangent>; out_tangent [B,S,3,H,W] = J tangent.
__global__ __launch_bounds__(kThreads) void k_render_f64_jvp(const double *__restrict__ maps, const double *__restrict__ tangent,
                                                             const float *__restrict__ scenes, const float *__restrict__ xrow,
                                                             const double *__restrict__ grad_out,
                                                             double *__restrict__ grad_maps_tangent,
                                                             double *__restrict__ out_tangent, int S, int H, int W)
{
    const size_t plane = (size_t)H * W;
    const size_t pix = (size_t)blockIdx.x * kThreads + threadIdx.x;
    const int b = blockIdx.y;
    if (pix >= plane) return;
    MapsD mv, mt;
    load_maps_f64(maps + (size_t)b * 12 * plane, plane, pix, mv);
    load_maps_f64(tangent + (size_t)b * 12 * plane, plane, pix, mt);
    MapsT<Dual> m, acc;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        m.n[k] = Dual{mv.n[k], mt.n[k]}; m.d[k] = Dual{mv.d[k], mt.d[k]};
        m.r[k] = Dual{mv.r[k], mt.r[k]}; m.s[k] = Dual{mv.s[k], mt.s[k]};
        acc.n[k] = acc.d[k] = acc.r[k] = acc.s[k] = Dual{0.0, 0.0};
    }
    float x[1], y;
    pixel_coords<1>(xrow, pix, W, x, y);
    const VConst K = make_vconst();
    for (int s = 0; s < S; ++s) {
        float sc[9];
        load_scene(scenes + ((size_t)b * S + s) * 9, sc);
        const Geom g = geometry(K, sc, x[0], y);
        const size_t o = (((size_t)b * S + s) * 3) * plane + pix;
        const Dual gr[3] = {Dual{grad_out[o], 0.0}, Dual{grad_out[o + plane], 0.0}, Dual{grad_out[o + 2 * plane], 0.0}};
        Dual rad[3];
        shade_f64<Dual>(g, m, rad, gr, &acc);
#pragma unroll
        for (int k = 0; k < 3; ++k) out_tangent[o + (size_t)k * plane] = rad[k].d;
    }
    double *__restrict__ gm = grad_maps_tangent + (size_t)b * 12 * plane + pix;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        gm[(size_t)(0 + k) * plane] = acc.n[k].d;
        gm[(size_t)(3 + k) * plane] = acc.d[k].d;
        gm[(size_t)(6 + k) * plane] = acc.r[k].d;
        gm[(size_t)(9 + k) * plane] = acc.s[k].d;
    }
}
}  // namespace

extern "C" {

static int render_f64_impl(bool bwd, const double *maps, const float *scenes, const float *xrow, const double *grad_out,
                           double *out, int B, int S, int H, int W, void *stream)
{
    if (!maps || !scenes || !xrow || !out || (bwd && !grad_out)) return fail(SVBRDF_ERR_NULL, "render_f64: null pointer");
    if (int e = check_dims(B, S, H, W)) return e;
    if (!aligned(maps, 8) || !aligned(out, 8) || !aligned(scenes, 4) || !aligned(xrow, 4) || (bwd && !aligned(grad_out, 8)))
        return fail(SVBRDF_ERR_ALIGN, "render_f64: double buffers must be 8-byte aligned, float buffers 4-byte");
    const dim3 grid = grid_for(B, H, W, 1), block(kThreads);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (bwd) hipLaunchKernelGGL(k_render_f64<true>, grid, block, 0, st, maps, scenes, xrow, grad_out, out, S, H, W);
    else hipLaunchKernelGGL(k_render_f64<false>, grid, block, 0, st, maps, scenes, xrow, grad_out, out, S, H, W);
    return launch_status(bwd ? "render_bwd_f64 launch" : "render_fwd_f64 launch");
}

int svbrdf_render_fwd_f64(const double *maps, const float *scenes, const float *xrow, double *out, int B, int S, int H, int W,
                          void *stream)
{
    return render_f64_impl(false, maps, scenes, xrow, nullptr, out, B, S, H, W, stream);
}

int svbrdf_render_bwd_f64(const double *maps, const float *scenes, const float *xrow, const double *grad_out,
                          double *grad_maps, int B, int S, int H, int W, void *stream)
{
    return render_f64_impl(true, maps, scenes, xrow, grad_out, grad_maps, B, S, H, W, stream);
}

int svbrdf_render_bwd_jvp_f64(const double *maps, const double *tangent, const float *scenes, const float *xrow,
                              const double *grad_out, double *grad_maps_tangent, double *out_tangent, int B, int S, int H, int W,
                              void *stream)
{
    if (!maps || !tangent || !scenes || !xrow || !grad_out || !grad_maps_tangent || !out_tangent)
        return fail(SVBRDF_ERR_NULL, "render_bwd_jvp_f64: null pointer");
    if (int e = check_dims(B, S, H, W)) return e;
    if (!aligned(maps, 8) || !aligned(tangent, 8) || !aligned(grad_out, 8) || !aligned(grad_maps_tangent, 8) ||
        !aligned(out_tangent, 8) || !aligned(scenes, 4) || !aligned(xrow, 4))
        return fail(SVBRDF_ERR_ALIGN, "render_bwd_jvp_f64: double buffers must be 8-byte aligned, float buffers 4-byte");
    const dim3 grid = grid_for(B, H, W, 1), block(kThreads);
    hipLaunchKernelGGL(k_render_f64_jvp, grid, block, 0, static_cast<hipStream_t>(stream), maps, tangent, scenes, xrow, grad_out,
                       grad_maps_tangent, out_tangent, S, H, W);
    return launch_status("render_bwd_jvp_f64 launch");
}

}  // extern "C"
