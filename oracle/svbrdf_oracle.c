/*
 * svbrdf_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, gcc) of the reference hot path
 *   development/multiImage_pytorch/renderers.py:67-104  LocalRenderer.render
 *   development/multiImage_pytorch/losses.py:29-52      RenderingLoss.forward
 * and of its analytic backward w.r.t. the four SVBRDF maps.  It exists only
 * so that tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg can
 * check / time the HIP path against it.  The product path
 * (svbrdf_estimation_amd/) never imports, links or calls anything here.
 *
 * Parity pin: the reference itself holds no golden vectors for this path
 * (its only unit tests, utils.py:149-247, pin gamma and channel order).  This
 * oracle is therefore pinned against outputs of the reference ITSELF, imported
 * read-only in the build container by tests/golden/make_golden.py; the
 * resulting fixtures live in tests/golden/ (npz files) and tests/test_oracle_golden.py
 * compares this file against them (renderings <= 1e-5 rel, gradients <= 1e-4
 * rel as per SURVEY.md section 8c).
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fopenmp -shared).
 * -ffp-contract=off is REQUIRED: the GGX denominator NH^2*(a^2+(1-NH^2)/NH^2)
 * (renderers.py:26) amplifies a 1-ULP change of NH by 1e3..1e4.
 */

#include <math.h>
#include <stddef.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ---- float instantiation (mirrors the reference's fp32 op order) ---- */
#define REAL float
#define OUT_T float
#define FN(x) x##_f32
#define SQRT_R(x) sqrtf(x)
#define POW_R(x, y) powf((x), (y))
#define LOG_R(x) logf(x)
#include "svbrdf_core.inc"
#undef REAL
#undef OUT_T
#undef FN
#undef SQRT_R
#undef POW_R
#undef LOG_R

/* ---- double instantiation (same formulas, fp32-valued inputs) ------- */
#define REAL double
#define OUT_T double
#define FN(x) x##_f64
#define SQRT_R(x) sqrt(x)
#define POW_R(x, y) pow((x), (y))
#define LOG_R(x) log(x)
#include "svbrdf_core.inc"
#undef REAL
#undef OUT_T
#undef FN
#undef SQRT_R
#undef POW_R
#undef LOG_R

#define EXPORT __attribute__((visibility("default")))

static int check_dims(int B, int S, int H, int W)
{
    return (B > 0 && S > 0 && H > 0 && W > 0 && H == W) ? 0 : -2;
}

EXPORT int svbrdf_oracle_version(void) { return 1; }

EXPORT void svbrdf_oracle_set_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

EXPORT int svbrdf_oracle_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* renderers.py:73 torch.linspace(-1, 1, W) as the CPU kernel of torch 2.10
 * evaluates it: step = fl32(2/(W-1)); lower half fma(step, i, -1), upper half
 * fma(-step, W-1-i, +1) -- bit-exact for every W probed (2..512). */
EXPORT void svbrdf_oracle_make_xrow(float *xrow, int W)
{
    int i;
    float step;
    if (W == 1) { xrow[0] = -1.0f; return; }
    step = 2.0f / (float)(W - 1);
    for (i = 0; i < W; ++i)
        xrow[i] = (i < W / 2) ? fmaf(step, (float)i, -1.0f)
                              : fmaf(-step, (float)(W - 1 - i), 1.0f);
}

EXPORT int svbrdf_oracle_render_fwd(const float *maps, const float *scenes, const float *xrow,
                                    float *out, int B, int S, int H, int W)
{
    int e = check_dims(B, S, H, W);
    return e ? e : render_fwd_f32(maps, scenes, xrow, out, B, S, H, W);
}

EXPORT int svbrdf_oracle_render_bwd(const float *maps, const float *scenes, const float *xrow,
                                    const float *grad_out, float *grad_maps,
                                    int B, int S, int H, int W)
{
    int e = check_dims(B, S, H, W);
    return e ? e : render_bwd_f32(maps, scenes, xrow, grad_out, grad_maps, B, S, H, W);
}

EXPORT int svbrdf_oracle_rendering_loss(const float *input, const float *target,
                                        const float *scenes, const float *xrow, float eps,
                                        double *loss_out, float *grad_input,
                                        int B, int S, int H, int W)
{
    int e = check_dims(B, S, H, W);
    return e ? e : rendering_loss_f32(input, target, scenes, xrow, eps, 0.0f, 0.01f, loss_out, grad_input, NULL, B, S, H, W);
}

/* losses.py:54-63 MixedLoss = l1_weight * SVBRDFL1Loss + RenderingLoss */
EXPORT int svbrdf_oracle_mixed_loss(const float *input, const float *target,
                                    const float *scenes, const float *xrow, float eps,
                                    float l1_weight, float eps_l1,
                                    double *loss_out, float *grad_input,
                                    int B, int S, int H, int W)
{
    int e = check_dims(B, S, H, W);
    return e ? e : rendering_loss_f32(input, target, scenes, xrow, eps, l1_weight, eps_l1, loss_out, grad_input, NULL, B, S, H, W);
}

/* per pixel: the smallest |log(render(input)+eps) - log(render(target)+eps)| over scenes and channels,
 * evaluated in double.  Where it is at the rounding level of an fp32 log (<~1e-6) the sign() in the
 * L1 gradient is numerically undetermined -- for the reference's autograd as much as for any other
 * fp32 evaluation -- and tests exclude those pixels from gradient comparisons. */
EXPORT int svbrdf_oracle_loss_tie_map(const float *input, const float *target, const float *scenes,
                                      const float *xrow, float eps, double *min_abs_delta,
                                      int B, int S, int H, int W)
{
    double loss;
    int e = check_dims(B, S, H, W);
    return e ? e : rendering_loss_f64(input, target, scenes, xrow, eps, 0.0f, 0.01f, &loss, NULL, min_abs_delta, B, S, H, W);
}

/* ---- network head: models.py:338-346 -> utils.py:73-98 ----------------
 * encoded [B,9,H,W] after tanh: normals_xy(0:2) | diffuse(2:5) | roughness(5) | specular(6:9);
 * decoded [B,12,H,W]: normalize(3nx, 3ny, 1) | (d+1)/2 | (r+1)/2 x3 | (s+1)/2.
 * fp32, the reference's op order: mul(3.0), pow(.,2) = x*x, sum (p0+p1)+p2, sqrt, div. */
#include <stdlib.h>

static void head_decode(const float *enc, float *maps, float *len_out, int B, int H, int W)
{
    const size_t plane = (size_t)H * W;
    long bi;
#pragma omp parallel for schedule(static)
    for (bi = 0; bi < (long)B * H; ++bi) {
        int b = (int)(bi / H), i = (int)(bi % H), j, k;
        const float *e = enc + (size_t)b * 9 * plane;
        float *m = maps + (size_t)b * 12 * plane;
        for (j = 0; j < W; ++j) {
            size_t p = (size_t)i * W + j;
            float vx = e[0 * plane + p] * 3.0f, vy = e[1 * plane + p] * 3.0f, vz = 1.0f;
            float len = sqrtf((vx * vx + vy * vy) + vz * vz);
            float r = (e[5 * plane + p] + 1.0f) / 2.0f;
            m[0 * plane + p] = vx / len;
            m[1 * plane + p] = vy / len;
            m[2 * plane + p] = vz / len;
            for (k = 0; k < 3; ++k) {
                m[(3 + k) * plane + p] = (e[(2 + k) * plane + p] + 1.0f) / 2.0f;
                m[(6 + k) * plane + p] = r;
                m[(9 + k) * plane + p] = (e[(6 + k) * plane + p] + 1.0f) / 2.0f;
            }
            len_out[(size_t)b * plane + p] = len;
        }
    }
}

EXPORT int svbrdf_oracle_head_decode(const float *enc, float *maps, int B, int H, int W)
{
    float *len;
    if (B <= 0 || H <= 0 || W <= 0) return -2;
    len = (float *)malloc((size_t)B * H * W * sizeof(float));
    if (!len) return -5;
    head_decode(enc, maps, len, B, H, W);
    free(len);
    return 0;
}

/* chain rule of head_decode, in double: g12 [B,12,H,W] -> g9 [B,9,H,W] */
static void head_chain(const float *maps, const float *len, const double *g12, double *g9, int B, int H, int W)
{
    const size_t plane = (size_t)H * W;
    long bp;
#pragma omp parallel for schedule(static)
    for (bp = 0; bp < (long)B * (long)plane; ++bp) {
        size_t b = (size_t)bp / plane, p = (size_t)bp % plane;
        const float *m = maps + b * 12 * plane;
        const double *g = g12 + b * 12 * plane;
        double *o = g9 + b * 9 * plane;
        double n0 = m[p], n1 = m[plane + p], n2 = m[2 * plane + p];
        double ng = n0 * g[p] + n1 * g[plane + p] + n2 * g[2 * plane + p];
        double il = 1.0 / (double)len[b * plane + p];
        int k;
        o[0 * plane + p] = 3.0 * (g[p] - n0 * ng) * il;
        o[1 * plane + p] = 3.0 * (g[plane + p] - n1 * ng) * il;
        for (k = 0; k < 3; ++k) {
            o[(2 + k) * plane + p] = 0.5 * g[(3 + k) * plane + p];
            o[(6 + k) * plane + p] = 0.5 * g[(9 + k) * plane + p];
        }
        o[5 * plane + p] = 0.5 * ((g[6 * plane + p] + g[7 * plane + p]) + g[8 * plane + p]);
    }
}

/* mixed loss of the decoded head output and its gradient w.r.t. the 9 encoded channels.
 * f64 != 0: the loss/gradient w.r.t. the decoded maps are evaluated in double. */
EXPORT int svbrdf_oracle_head_loss(const float *enc, const float *target, const float *scenes,
                                   const float *xrow, float eps, float l1_weight, float eps_l1,
                                   double *loss_out, double *grad9, int f64, int B, int S, int H, int W)
{
    const size_t n12 = (size_t)B * 12 * H * W;
    float *maps, *len, *g32 = NULL;
    double *g64 = NULL;
    int rc = check_dims(B, S, H, W);
    size_t i;
    if (rc) return rc;
    maps = (float *)malloc(n12 * sizeof(float));
    len = (float *)malloc((size_t)B * H * W * sizeof(float));
    g64 = grad9 ? (double *)malloc(n12 * sizeof(double)) : NULL;
    if (!maps || !len || (grad9 && !g64)) { free(maps); free(len); free(g64); return -5; }
    head_decode(enc, maps, len, B, H, W);
    if (f64) {
        rc = rendering_loss_f64(maps, target, scenes, xrow, eps, l1_weight, eps_l1, loss_out, g64, NULL, B, S, H, W);
    } else {
        if (grad9) g32 = (float *)malloc(n12 * sizeof(float));
        rc = rendering_loss_f32(maps, target, scenes, xrow, eps, l1_weight, eps_l1, loss_out, g32, NULL, B, S, H, W);
        if (grad9) for (i = 0; i < n12; ++i) g64[i] = (double)g32[i];
        free(g32);
    }
    if (!rc && grad9) head_chain(maps, len, g64, grad9, B, H, W);
    free(maps); free(len); free(g64);
    return rc;
}

EXPORT int svbrdf_oracle_render_fwd_f64(const float *maps, const float *scenes, const float *xrow,
                                        double *out, int B, int S, int H, int W)
{
    int e = check_dims(B, S, H, W);
    return e ? e : render_fwd_f64(maps, scenes, xrow, out, B, S, H, W);
}

EXPORT int svbrdf_oracle_render_bwd_f64(const float *maps, const float *scenes, const float *xrow,
                                        const float *grad_out, double *grad_maps,
                                        int B, int S, int H, int W)
{
    int e = check_dims(B, S, H, W);
    return e ? e : render_bwd_f64(maps, scenes, xrow, grad_out, grad_maps, B, S, H, W);
}

EXPORT int svbrdf_oracle_rendering_loss_f64(const float *input, const float *target,
                                            const float *scenes, const float *xrow, float eps,
                                            double *loss_out, double *grad_input,
                                            int B, int S, int H, int W)
{
    int e = check_dims(B, S, H, W);
    return e ? e : rendering_loss_f64(input, target, scenes, xrow, eps, 0.0f, 0.01f, loss_out, grad_input, NULL, B, S, H, W);
}

EXPORT int svbrdf_oracle_mixed_loss_f64(const float *input, const float *target,
                                        const float *scenes, const float *xrow, float eps,
                                        float l1_weight, float eps_l1,
                                        double *loss_out, double *grad_input,
                                        int B, int S, int H, int W)
{
    int e = check_dims(B, S, H, W);
    return e ? e : rendering_loss_f64(input, target, scenes, xrow, eps, l1_weight, eps_l1, loss_out, grad_input, NULL, B, S, H, W);
}
