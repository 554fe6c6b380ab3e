/*
 * svbrdf_hip.h -- C ABI of libsvbrdf_hip.so, the MI355X (gfx950) rendering-loss engine.
 *
 * Drop-in boundary for ONE hot path of mworchel/svbrdf-estimation:
 *     LocalRenderer.render            development/multiImage_pytorch/renderers.py:67-104
 *     RenderingLoss.forward (+autograd backward)
 *                                     development/multiImage_pytorch/losses.py:29-52
 * The reference has no FFI (it is pure PyTorch); the "binding" a maintainer adds is
 * the ctypes stub shown in INTEGRATION.md, which replaces the bodies of those two
 * methods.  Signatures use plain pointers and sizes only -- no torch types.
 *
 * Conventions
 *   - every pointer except `workspace`/`xrow_host` is DEVICE memory owned by the
 *     caller, C-contiguous fp32, 4-byte aligned (16-byte alignment enables the
 *     vectorised path; anything else takes the scalar path -- same results).
 *   - layouts:  maps/input/target/grad  [B,12,H,W]  channel order
 *               normals(0:3) diffuse(3:6) roughness(6:9) specular(9:12)
 *               (utils.py:36-58 pack_svbrdf/unpack_svbrdf)
 *               scenes  [B,S,9] = camera xyz | light xyz | light rgb   per render
 *               (environment.py:4-16 Camera/Light/Scene)
 *               xrow    [W]     = torch.linspace(-1, 1, W)  (renderers.py:73); pixel
 *               (i,j) sits at (xrow[j], -xrow[i], 0) (renderers.py:74-76), so H == W.
 *               renderings / grad_out [B,S,3,H,W]
 *   - the library never allocates, frees or retains device memory and never
 *     synchronises; kernels are enqueued on `stream` (a hipStream_t; NULL = the
 *     default stream) and the call returns immediately.
 *   - return value: 0 success; <0 argument error (-1 null pointer, -2 bad dims or
 *     H != W, -3 misaligned pointer, -4 workspace too small); >0 hipError_t of the
 *     launch.  svbrdf_last_error() gives a thread-local message.
 *   - stateless and re-entrant; results are bitwise run-to-run reproducible (fixed
 *     shape two-stage loss reduction, no float atomics).
 */
#ifndef SVBRDF_HIP_H
#define SVBRDF_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SVBRDF_ABI_VERSION 7

#if defined(__GNUC__)
#define SVBRDF_API __attribute__((visibility("default")))
#else
#define SVBRDF_API
#endif

#define SVBRDF_OK 0
#define SVBRDF_ERR_NULL (-1)
#define SVBRDF_ERR_DIMS (-2)
#define SVBRDF_ERR_ALIGN (-3)
#define SVBRDF_ERR_WORKSPACE (-4)

SVBRDF_API int svbrdf_abi_version(void);

/* thread-local, static storage; valid until the next failing call on this thread */
SVBRDF_API const char *svbrdf_last_error(void);

/* Host helper: fills xrow_host[W] (HOST memory) with the bit pattern of
 * torch.linspace(-1, 1, W) as the reference's CPU path produces it
 * (renderers.py:73).  Upload it once per W. */
SVBRDF_API int svbrdf_make_xrow(float *xrow_host, int W);

/* K1 -- replaces LocalRenderer.render (renderers.py:67-104), S scenes per map in
 * one launch: out[b,s] = render(scene[b,s], maps[b]). */
SVBRDF_API int svbrdf_render_fwd(const float *maps, const float *scenes, const float *xrow,
                      float *out, int B, int S, int H, int W, void *stream);

/* K2 -- replaces the autograd graph of render(): grad_maps[b] = sum_s J(b,s)^T
 * grad_out[b,s]; grad_maps is overwritten. */
SVBRDF_API int svbrdf_render_bwd(const float *maps, const float *scenes, const float *xrow,
                      const float *grad_out, float *grad_maps,
                      int B, int S, int H, int W, void *stream);

/* K1 / K2 with the scene rows in HOST memory -- what a reference-shaped `render(scene, svbrdf)` call has: the scene is a
 * Python object (environment.py:4-16) and the reference uploads its three vectors with three synchronous H2D copies per
 * call (renderers.py:79,91,98).  Here the rows travel BY VALUE in the launch's kernel-argument block: one dispatch per
 * call, no copy command, nothing retained past return.  `scenes_host` (HOST pointer) holds
 *     scenes_shared != 0 :  S rows, the SAME scenes for every map   (render's "one scene, B maps", renderers.py:98)
 *     scenes_shared == 0 :  B*S rows, as svbrdf_render_fwd
 * at most svbrdf_host_scenes_max_rows() of them (SVBRDF_ERR_DIMS beyond: upload the table and use the device entry).
 * Same kernel bodies, bitwise the same results as svbrdf_render_fwd / svbrdf_render_bwd on an uploaded table. */
SVBRDF_API int svbrdf_render_fwd_host_scenes(const float *maps, const float *scenes_host, int scenes_shared,
                                             const float *xrow, float *out, int B, int S, int H, int W, void *stream);
SVBRDF_API int svbrdf_render_bwd_host_scenes(const float *maps, const float *scenes_host, int scenes_shared,
                                             const float *xrow, const float *grad_out, float *grad_maps,
                                             int B, int S, int H, int W, void *stream);

/* Ragged forms of K1 / K2 (SURVEY section 8b "arbitrary (map, scene) pairs in one launch"): R renders in all, grouped
 * by map -- the renders of map b are rows offsets[b] .. offsets[b+1]-1 of `scenes` [R,9] and of `out` / `grad_out`
 * [R,3,H,W]; `offsets` is a DEVICE array of B+1 int32 with offsets[0] = 0 and offsets[B] = R (CSR row pointers; a map
 * may have zero renders).  One launch, each map read once, no atomics: bitwise reproducible like the regular forms, of
 * which these are the generalisation (offsets[b] = b*S).  The proposal's per-render `map_index` is this layout's
 * inverse; grouping by map is what lets the backward accumulate in registers instead of with float atomics.
 * svbrdf_render_bwd_ragged overwrites ALL of grad_maps (zeros for a map without renders). */
SVBRDF_API int svbrdf_render_fwd_ragged(const float *maps, const float *scenes, const int *offsets, const float *xrow,
                                        float *out, int B, int R, int H, int W, void *stream);
SVBRDF_API int svbrdf_render_bwd_ragged(const float *maps, const float *scenes, const int *offsets, const float *xrow,
                                        const float *grad_out, float *grad_maps, int B, int R, int H, int W,
                                        void *stream);

/* Bytes of device scratch svbrdf_rendering_loss_fwd_bwd needs for these dims (65 64-bit
 * words: sharded fixed-point loss accumulators with arrival counts, and a ticket).  The scratch must be
 * 8-byte aligned and ZERO-INITIALISED ONCE by the caller (hipMemset) before its first use;
 * every completed call leaves it zeroed again, so it can be reused call after call on
 * the same stream.  Do not share one scratch buffer between concurrently running calls. */
SVBRDF_API size_t svbrdf_rendering_loss_workspace_bytes(int B, int S, int H, int W);

/* K3 -- replaces RenderingLoss.forward AND its backward (losses.py:29-52):
 *   loss_out[0] = mean_{b,s,c,i,j} | log(render(input)+eps) - log(render(target)+eps) |
 *   grad_input  = d loss / d input  (for upstream gradient 1.0), or NULL for
 *                 forward-only.
 * A NaN or infinite value anywhere in `input` / `target` (or a radiance beyond any the
 * renderer can produce) gives loss_out[0] = NaN, like the reference's clamp/log chain, and
 * the scratch is still left zeroed for the next call.
 * `scenes` are the light/view samples the caller drew (losses.py:35).  `eps` (losses.py:45:
 * 0.1) must lie in [1e-9, 1e9] (SVBRDF_ERR_DIMS otherwise): the kernel derives the three
 * 1/(render+eps) of a pixel from one reciprocal of their product. */
SVBRDF_API int svbrdf_rendering_loss_fwd_bwd(const float *input, const float *target,
                                  const float *scenes, const float *xrow, float eps,
                                  float *loss_out, float *grad_input,
                                  void *workspace, size_t workspace_bytes,
                                  int B, int S, int H, int W, void *stream);

/* K3 with SVBRDFL1Loss folded in -- replaces MixedLoss.forward AND its backward
 * (losses.py:54-63 = l1_weight * SVBRDFL1Loss (losses.py:7-19) + RenderingLoss):
 *   loss_out[0] = l1_weight * [ mean|n_i-n_t| + mean|log(d_i+eps_l1)-log(d_t+eps_l1)|
 *                               + mean|r_i-r_t| + mean|log(s_i+eps_l1)-log(s_t+eps_l1)| ]
 *                 + rendering loss (as above)
 * on the 24 planes the kernel reads anyway: zero extra HBM traffic.  Same scratch and
 * conventions as svbrdf_rendering_loss_fwd_bwd.  Reference values: l1_weight 0.1,
 * eps_l1 0.01, eps_render 0.1. */
SVBRDF_API int svbrdf_mixed_loss_fwd_bwd(const float *input, const float *target, const float *scenes,
                                         const float *xrow, float eps_render, float l1_weight, float eps_l1,
                                         float *loss_out, float *grad_input, void *workspace,
                                         size_t workspace_bytes, int B, int S, int H, int W, void *stream);

/* K3 with the network head folded in (the north star's "normal-map
 * decode"): `encoded9` is the generator's [B,9,H,W] output AFTER tanh, channels
 * normals_xy(0:2) | diffuse(2:5) | roughness(5) | specular(6:9) (utils.py:49-53).  The kernel
 * decodes it exactly like models.py:338-346 (utils.decode_svbrdf, then (x+1)/2 for
 * diffuse/roughness/specular), evaluates the mixed loss against the 12-channel `target`
 * (l1_weight = 0: rendering loss only) and returns d loss / d encoded9 in `grad_encoded9`
 * ([B,9,H,W], or NULL).  Replaces ~15 elementwise launches of the model head and their
 * backward; HBM traffic 120 instead of 144 bytes per pixel. */
SVBRDF_API int svbrdf_head_loss_fwd_bwd(const float *encoded9, const float *target, const float *scenes,
                                        const float *xrow, float eps_render, float l1_weight, float eps_l1,
                                        float *loss_out, float *grad_encoded9, void *workspace,
                                        size_t workspace_bytes, int B, int S, int H, int W, void *stream);

/* Scene table handed over in HOST memory.  The reference draws the table on the CPU for every
 * call (losses.py:35 -> environment.py:18-55), so the caller of this path always holds it in
 * host memory first.  These two entry points take it there: the B*S rows are copied into the
 * kernel-argument block of the launch (consumed before the call returns: `scenes_host` may be
 * reused or freed immediately), so the call enqueues ONE kernel dispatch and nothing else -- no
 * device buffer for the table, no H2D copy command, no pinned staging.  Limit:
 * B*S <= SVBRDF_HOST_SCENES_MAX_ROWS (288 rows = a 10 KB argument block: configs[3]'s 16 x 9 and config 5's
 * 8 x 32 rows fit; the HIP runtime on gfx950 takes argument blocks of at least 32 KB, measured); larger tables return
 * SVBRDF_ERR_DIMS and go through the device-pointer entry points above.  Everything else
 * (results, scratch, stream semantics) is identical to svbrdf_mixed_loss_fwd_bwd /
 * svbrdf_head_loss_fwd_bwd; l1_weight = 0 gives the plain RenderingLoss. */
#define SVBRDF_HOST_SCENES_MAX_ROWS 288
SVBRDF_API int svbrdf_host_scenes_max_rows(void);
SVBRDF_API int svbrdf_mixed_loss_fwd_bwd_host_scenes(const float *input, const float *target,
                                                     const float *scenes_host, const float *xrow,
                                                     float eps_render, float l1_weight, float eps_l1,
                                                     float *loss_out, float *grad_input, void *workspace,
                                                     size_t workspace_bytes, int B, int S, int H, int W,
                                                     void *stream);
SVBRDF_API int svbrdf_head_loss_fwd_bwd_host_scenes(const float *encoded9, const float *target,
                                                    const float *scenes_host, const float *xrow,
                                                    float eps_render, float l1_weight, float eps_l1,
                                                    float *loss_out, float *grad_encoded9, void *workspace,
                                                    size_t workspace_bytes, int B, int S, int H, int W,
                                                    void *stream);

/* data[i] *= scale_dev[0] for i < n, on the device and without a host sync; when the
 * scalar is exactly 1.0 the kernel exits without touching `data`.  Used by the autograd
 * wrapper to apply the upstream gradient of the loss (the chain rule through
 * RenderingLoss.forward's scalar output) to grad_input. */
SVBRDF_API int svbrdf_scale_inplace(float *data, const float *scale_dev, size_t n, void *stream);

/* Test aid: evaluates the kernels' Newton square root and shared-reciprocal division
 * (the primitives that stand in for the reference's torch.sqrt / torch.div, i.e.
 * normalize() of renderers.py:11-12, on the ill-conditioned coords -> NH path) on n
 * pseudo-random operands -- squared lengths x log-uniform in [lo, hi], numerators a
 * uniform in [-hi, hi] -- and ADDS the number of results that differ from the
 * IEEE-correct a / sqrtf(x) to counts_dev[0] and from sqrtf(x) to counts_dev[1].  counts_dev: two zero-initialised device uint64. */
SVBRDF_API int svbrdf_debug_check_arith(unsigned long long n, unsigned seed, float lo, float hi,
                                        unsigned long long *counts_dev, void *stream);

/* K4 -- replaces SvbrdfDataset.mix (dataset.py:142-160), the material-mixing augmentation, for a whole batch:
 *   out[b] = mix(svbrdf0[b], svbrdf1[b], alpha[b])   all [B,12,H,W] device, alpha [B] device (the reference draws
 *   it per sample from U(0.1, 0.9), dataset.py:144).  Normals are projected to z = 1 (n / max(0.01, n.z)), blended
 *   with weights alpha and fp32(1 - alpha) and renormalised; diffuse, roughness and specular are blended.  Same
 *   operation order and roundings as the reference.  H and W are independent here.  `out` may not alias the inputs. */
SVBRDF_API int svbrdf_mix_materials(const float *svbrdf0, const float *svbrdf1, const float *alpha, float *out,
                                    int B, int H, int W, void *stream);

/* K1 with the sensor-noise epilogue (ABI version 7) -- replaces the loop body of SvbrdfDataset.render_inputs
 * (dataset.py:206-219: render, + N(0, std_image) from the CPU generator, torch.clamp(0, 1)) for a whole batch of samples
 * and all their photos in ONE launch that writes each photo once:
 *   out[b,s] = clamp(render(scene[b,s], maps[b]) + noise_std[b,s] * n(seed, offset, element), 0, 1)     [B,S,3,H,W]
 * `noise_std` holds one level per render ([B*S], the reference draws it per image, dataset.py:215); NULL = no noise,
 * clamp only: bitwise clamp(svbrdf_render_fwd(...)).  The standard-normal field n is counter-based and a pure function of
 * (seed, offset, linear index e of the element in `out`): Philox4x32-10 with key = (seed lo, seed hi) and counter =
 * (g lo, g hi, offset lo, offset hi), g = e / 4, gives four 32-bit words x0..x3; u_i = ((x_i >> 8) + 0.5) / 2^24;
 * n(4g) = r(u0) cos(2 pi u1), n(4g+1) = r(u0) sin(2 pi u1), n(4g+2) = r(u2) cos(2 pi u3), n(4g+3) = r(u2) sin(2 pi u3),
 * r(u) = sqrt(-2 ln u).  The field does not depend on the vector width of the launch; a caller advances `offset` (or the
 * seed) between calls, like a device generator.  Different numbers than the reference's CPU field, the same distribution.
 * `svbrdf_render_inputs`: scenes and levels are DEVICE tables.  `_host_scenes`: both are HOST arrays of B*S rows
 * (<= SVBRDF_HOST_SCENES_MAX_ROWS) that travel by value in the launch's argument block, as in
 * svbrdf_render_fwd_host_scenes (one scene table row per render; not shared between maps: every sample draws its own
 * views, dataset.py:172-204). */
SVBRDF_API int svbrdf_render_inputs(const float *maps, const float *scenes, const float *noise_std, unsigned long long seed,
                                    unsigned long long offset, const float *xrow, float *out, int B, int S, int H, int W,
                                    void *stream);
SVBRDF_API int svbrdf_render_inputs_host_scenes(const float *maps, const float *scenes_host, const float *noise_std_host,
                                                unsigned long long seed, unsigned long long offset, const float *xrow,
                                                float *out, int B, int S, int H, int W, void *stream);

/* Measurement aid (ABI version 7): dst[i] = src[i] for i < n floats (n a multiple of 4, both pointers 16-byte aligned,
 * no overlap) as a plain streaming copy, 16 bytes per lane and access, non-temporal loads and stores.  bench.py and the
 * speed guard time it on buffers beyond the Infinity Cache: 2 * 4 * n bytes / duration is the copy bandwidth of THIS
 * box, the "measured-copy peak" the HBM-bound kernels are priced against beside the nominal 8 TB/s. */
SVBRDF_API int svbrdf_debug_copy(float *dst, const float *src, size_t n, void *stream);

/* Test aid (ABI version 7): how many kernels this library has enqueued in this process (every entry point enqueues
 * exactly one; failed calls are not counted).  tests/test_gpu_perf_guard.py asserts ONE launch per training step. */
SVBRDF_API unsigned long long svbrdf_debug_launch_count(void);

/* Measurement aid: one wave spins for `ticks_100mhz` ticks of the chip's constant 100 MHz counter on `stream`
 * and writes out_dev[0] = shader-clock cycles elapsed, out_dev[1] = 100 MHz ticks elapsed (two device uint64).
 * cycles / ticks * 0.1 = the shader clock in GHz while whatever else is running runs (bench.py launches it on a
 * stream of its own beside the fused loss: the clock the chip holds under that kernel). */
SVBRDF_API int svbrdf_debug_clock_probe(unsigned long long *out_dev, unsigned long long ticks_100mhz, void *stream);

/* ---------------------------------------------------------------------------------------------------------------
 * AUXILIARY entry points -- NOT part of the north-star path (BASELINE.json: fp32 maps, first-order training).
 * The three svbrdf_*_f64 symbols below serve callers who hand the renderer double maps or differentiate through its
 * backward (gradient checks, notebook-style direct map optimisation); the reference itself never does either.  They are
 * compiled from a translation unit of their own (csrc/svbrdf_aux_f64.hip), take none of the tuned kernels' paths, are
 * frozen since ABI version 6, and a replacement library that serves training only may return SVBRDF_ERR_DIMS from them.
 * --------------------------------------------------------------------------------------------------------------- */

/* float64 maps (ABI version 5).  LocalRenderer.render is dtype-agnostic in the reference (renderers.py:67-104); with double
 * maps it computes in MIXED precision: pixel grid (torch.linspace, :73), camera / light positions and light colour
 * (torch.Tensor(...), :79,:91,:98) are float32, so wo, wi, h, (1-VH)^5 and colour*falloff are the float32 values of the
 * float32 path, and everything that touches the maps is promoted to double.  These two entry points do exactly that:
 *   maps / grad_maps [B,12,H,W] double, scenes [B,S,9] float32 DEVICE, xrow [W] float32, out / grad_out [B,S,3,H,W] double.
 * Shading op by op in the reference's order (the slow path of gradient checks and notebooks, not of training); the losses
 * for double maps are composed from these through autograd on the host side.  Same error convention as above. */
SVBRDF_API int svbrdf_render_fwd_f64(const double *maps, const float *scenes, const float *xrow, double *out,
                                     int B, int S, int H, int W, void *stream);
SVBRDF_API int svbrdf_render_bwd_f64(const double *maps, const float *scenes, const float *xrow, const double *grad_out,
                                     double *grad_maps, int B, int S, int H, int W, void *stream);

/* Second order (ABI version 6): what autograd needs to differentiate THROUGH the backward of render() -- the reference's
 * render is built from differentiable torch ops (renderers.py:8-104), so `backward(create_graph=True)` /
 * `torch.autograd.grad(..., create_graph=True)` work there (gradient penalties, Hessian-vector products in the notebooks'
 * style of direct map optimisation).  For a direction `tangent` [B,12,H,W] of the maps, one forward-mode (dual number)
 * evaluation of the float64 shading and its adjoint gives
 *   out_tangent       [B,S,3,H,W] = J(maps) tangent                      (derivative of <J^T grad_out, tangent> w.r.t. grad_out)
 *   grad_maps_tangent [B,12,H,W]  = d/dmaps <J(maps)^T grad_out, tangent>  (the Hessian of <grad_out, render(maps)> times tangent)
 * with clamps and sub-gradient selections treated as torch's double-backward formulas treat them (masks compare values,
 * zero derivative).  Buffers and error convention as svbrdf_render_bwd_f64; float32 callers are served by the host side
 * through these in double. */
SVBRDF_API int svbrdf_render_bwd_jvp_f64(const double *maps, const double *tangent, const float *scenes, const float *xrow,
                                         const double *grad_out, double *grad_maps_tangent, double *out_tangent,
                                         int B, int S, int H, int W, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SVBRDF_HIP_H */
