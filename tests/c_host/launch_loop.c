/* Launch loops for tests/test_gpu_perf_guard.py: n back-to-back calls of ONE entry point of the C ABI (include/svbrdf_hip.h)
 * from C, so that the host side of a launch is the ABI function itself (a few microseconds) and not a Python / ctypes call
 * (20-35 us on a slow host: more than the fused loss kernel runs).  The entry point comes in as a function pointer -- the
 * test takes it from the library it measures -- so this file links against nothing and needs no HIP header.
 * Built by the test with gcc into a temporary shared object.  Test infrastructure, not product code. */
#include <stddef.h>

typedef int (*loss_fn_t)(const float *, const float *, const float *, const float *, float, float, float, float *, float *,
                         void *, size_t, int, int, int, int, void *);
typedef int (*render_fwd_fn_t)(const float *, const float *, const float *, float *, int, int, int, int, void *);
typedef int (*render_bwd_fn_t)(const float *, const float *, const float *, const float *, float *, int, int, int, int, void *);
typedef int (*mix_fn_t)(const float *, const float *, const float *, float *, int, int, int, void *);

/* launch k = first .. first+n-1 uses buffer set k % sets; returns the first non-zero return code, or 0 */
int perf_loop_loss(loss_fn_t fn, int n, int first, int sets, const float *const *in, const float *const *tg, float *const *grad,
                   const float *scenes_host, const float *xrow, float eps, float l1_weight, float eps_l1, float *loss,
                   void *ws, size_t ws_bytes, int B, int S, int H, int W, void *stream)
{
    int k, rc;
    for (k = first; k < first + n; ++k) {
        const int j = k % sets;
        rc = fn(in[j], tg[j], scenes_host, xrow, eps, l1_weight, eps_l1, loss, grad[j], ws, ws_bytes, B, S, H, W, stream);
        if (rc != 0) return rc;
    }
    return 0;
}

int perf_loop_render_fwd(render_fwd_fn_t fn, int n, const float *maps, const float *scenes, const float *xrow, float *out,
                         int B, int S, int H, int W, void *stream)
{
    int k, rc;
    for (k = 0; k < n; ++k)
        if ((rc = fn(maps, scenes, xrow, out, B, S, H, W, stream)) != 0) return rc;
    return 0;
}

int perf_loop_render_bwd(render_bwd_fn_t fn, int n, const float *maps, const float *scenes, const float *xrow,
                         const float *grad_out, float *grad_maps, int B, int S, int H, int W, void *stream)
{
    int k, rc;
    for (k = 0; k < n; ++k)
        if ((rc = fn(maps, scenes, xrow, grad_out, grad_maps, B, S, H, W, stream)) != 0) return rc;
    return 0;
}

int perf_loop_mix(mix_fn_t fn, int n, int first, int sets, const float *const *a, const float *const *b, const float *alpha,
                  float *const *out, int B, int H, int W, void *stream)
{
    int k, rc;
    for (k = first; k < first + n; ++k) {
        const int j = k % sets;
        if ((rc = fn(a[j], b[j], alpha, out[j], B, H, W, stream)) != 0) return rc;
    }
    return 0;
}

typedef int (*copy_fn_t)(float *, const float *, size_t, void *);
typedef int (*render_inputs_fn_t)(const float *, const float *, const float *, unsigned long long, unsigned long long,
                                  const float *, float *, int, int, int, int, void *);

/* svbrdf_debug_copy: the copy bandwidth of the box */
int perf_loop_copy(copy_fn_t fn, int n, float *dst, const float *src, size_t n_floats, void *stream)
{
    int k, rc;
    for (k = 0; k < n; ++k)
        if ((rc = fn(dst, src, n_floats, stream)) != 0) return rc;
    return 0;
}

/* svbrdf_render_inputs (device tables): K1 + sensor noise + clamp; the offset advances like a device generator's */
int perf_loop_render_inputs(render_inputs_fn_t fn, int n, int first, const float *maps, const float *scenes, const float *noise_std,
                            unsigned long long seed, const float *xrow, float *out, int B, int S, int H, int W, void *stream)
{
    int k, rc;
    for (k = first; k < first + n; ++k)
        if ((rc = fn(maps, scenes, noise_std, seed, 4ull * (unsigned long long)k, xrow, out, B, S, H, W, stream)) != 0) return rc;
    return 0;
}
