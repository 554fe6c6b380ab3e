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
    return e ? e : rendering_loss_f32(input, target, scenes, xrow, eps, 0.0f, 0.01f, loss_out, grad_input, B, S, H, W);
}

/* losses.py:54-63 MixedLoss = l1_weight * SVBRDFL1Loss + RenderingLoss */
EXPORT int svbrdf_oracle_mixed_loss(const float *input, const float *target,
                                    const float *scenes, const float *xrow, float eps,
                                    float l1_weight, float eps_l1,
                                    double *loss_out, float *grad_input,
                                    int B, int S, int H, int W)
{
    int e = check_dims(B, S, H, W);
    return e ? e : rendering_loss_f32(input, target, scenes, xrow, eps, l1_weight, eps_l1, loss_out, grad_input, B, S, H, W);
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
    return e ? e : rendering_loss_f64(input, target, scenes, xrow, eps, 0.0f, 0.01f, loss_out, grad_input, B, S, H, W);
}

EXPORT int svbrdf_oracle_mixed_loss_f64(const float *input, const float *target,
                                        const float *scenes, const float *xrow, float eps,
                                        float l1_weight, float eps_l1,
                                        double *loss_out, double *grad_input,
                                        int B, int S, int H, int W)
{
    int e = check_dims(B, S, H, W);
    return e ? e : rendering_loss_f64(input, target, scenes, xrow, eps, l1_weight, eps_l1, loss_out, grad_input, B, S, H, W);
}
