/* A plain-C host of libsvbrdf_hip.so: what a non-Python caller of the drop-in boundary looks like.
 * Renders the KAT-1 case (tests/golden/make_golden.py g9_kat: a flat grey-ish material on a 2x2 patch, camera and
 * light straight above) through svbrdf_render_fwd and prints the 3x2x2 radiance with enough digits to round-trip.
 * Built and run by tests/test_gpu_parity.py::test_c_host_program_calls_the_abi; compiled (only) as part of the CPU
 * suite to check that include/svbrdf_hip.h is valid C. */
#include <stdio.h>
#include <stdlib.h>

#include <hip/hip_runtime_api.h>

#include "svbrdf_hip.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

int main(void)
{
    enum { H = 2, W = 2, PLANE = H * W };
    float maps[12 * PLANE], xrow[W], out[3 * PLANE];
    const float scene[9] = {0.0f, 0.0f, 2.0f, 0.0f, 0.0f, 2.0f, 50.0f, 50.0f, 50.0f};   /* camera | light | colour */
    const float channel[12] = {0.0f, 0.0f, 1.0f, 0.5f, 0.4f, 0.3f, 0.5f, 0.5f, 0.5f, 0.04f, 0.04f, 0.04f};
    float *d_maps, *d_scene, *d_xrow, *d_out;
    int c, p, rc;
    if (svbrdf_abi_version() != SVBRDF_ABI_VERSION) { fprintf(stderr, "ABI mismatch\n"); return 1; }
    for (c = 0; c < 12; ++c)
        for (p = 0; p < PLANE; ++p) maps[c * PLANE + p] = channel[c];
    if (svbrdf_make_xrow(xrow, W) != SVBRDF_OK) { fprintf(stderr, "%s\n", svbrdf_last_error()); return 1; }
    CHECK_HIP(hipMalloc((void **)&d_maps, sizeof maps));
    CHECK_HIP(hipMalloc((void **)&d_scene, sizeof scene));
    CHECK_HIP(hipMalloc((void **)&d_xrow, sizeof xrow));
    CHECK_HIP(hipMalloc((void **)&d_out, sizeof out));
    CHECK_HIP(hipMemcpy(d_maps, maps, sizeof maps, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_scene, scene, sizeof scene, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_xrow, xrow, sizeof xrow, hipMemcpyHostToDevice));
    rc = svbrdf_render_fwd(d_maps, d_scene, d_xrow, d_out, 1, 1, H, W, NULL);
    if (rc != SVBRDF_OK) { fprintf(stderr, "svbrdf_render_fwd: %d %s\n", rc, svbrdf_last_error()); return 1; }
    CHECK_HIP(hipDeviceSynchronize());
    CHECK_HIP(hipMemcpy(out, d_out, sizeof out, hipMemcpyDeviceToHost));
    /* argument errors come back as negative codes with a message, never as a crash */
    if (svbrdf_render_fwd(NULL, d_scene, d_xrow, d_out, 1, 1, H, W, NULL) != SVBRDF_ERR_NULL) return 1;
    if (svbrdf_render_fwd(d_maps, d_scene, d_xrow, d_out, 1, 1, H, W + 1, NULL) != SVBRDF_ERR_DIMS) return 1;
    for (p = 0; p < 3 * PLANE; ++p) printf("%.9g\n", (double)out[p]);
    hipFree(d_maps); hipFree(d_scene); hipFree(d_xrow); hipFree(d_out);
    return 0;
}
