/* Plain-C client of libspcube_hip.so: proves that include/spcube_hip.h is a C header (no C++,
 * no torch types) and that the boundary works without Python.
 *   abi_check <path/to/libspcube_hip.so>          symbols + argument checking (no GPU needed)
 *   abi_check <path/to/libspcube_hip.so> --gpu    additionally runs the reference's 3x3x3 moment
 *                                                 cube (spectral_cube/tests/test_moments.py:56-70,
 *                                                 golden table :19-49) through spc_moments_f32.
 * Built and run by tests/test_host_logic.py / tests/test_gpu_ops.py with gcc. */
#include <dlfcn.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../include/spcube_hip.h"

#define SYM(name) \
    do { if (!dlsym(lib, #name)) { fprintf(stderr, "missing symbol %s\n", #name); return 2; } } while (0)

typedef int (*moments_fn)(int, void*, const spc_cube_f32*, const spc_mask*, const double*, double, double,
                          const spc_moment_outputs*, void*, size_t);
typedef int (*malloc_fn)(int, size_t, void**);
typedef int (*free_fn)(int, void*);
typedef int (*copy_fn)(int, void*, const void*, size_t, void*);
typedef const char* (*err_fn)(void);
typedef int (*count_fn)(int*);

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: abi_check lib.so [--gpu]\n"); return 1; }
    void* lib = dlopen(argv[1], RTLD_NOW | RTLD_GLOBAL);
    if (!lib) { fprintf(stderr, "dlopen failed: %s\n", dlerror()); return 2; }
    SYM(spc_abi_version); SYM(spc_last_error); SYM(spc_device_count); SYM(spc_get_device_info);
    SYM(spc_malloc); SYM(spc_free); SYM(spc_host_alloc); SYM(spc_host_free);
    SYM(spc_memcpy_h2d); SYM(spc_memcpy_d2h); SYM(spc_memcpy_d2d); SYM(spc_memcpy3d_h2d); SYM(spc_memcpy3d_d2d); SYM(spc_memset);
    SYM(spc_stream_create); SYM(spc_stream_destroy); SYM(spc_stream_sync); SYM(spc_device_sync);
    SYM(spc_event_create); SYM(spc_event_destroy); SYM(spc_event_record); SYM(spc_event_sync);
    SYM(spc_stream_wait_event); SYM(spc_event_elapsed_ms);
    SYM(spc_moments_workspace_bytes); SYM(spc_workspace_bytes); SYM(spc_moments_f32); SYM(spc_moment_order_f32); SYM(spc_moments_spatial_f32);
    SYM(spc_spectral_conv_f32); SYM(spc_spectral_conv_moments_f32); SYM(spc_spatial_conv_sep_f32);
    SYM(spc_spatial_conv2d_f32); SYM(spc_spectral_lerp_f32); SYM(spc_resample_bilinear_f32); SYM(spc_resample_bilinear_lerp_f32); SYM(spc_resample_spline_f32); SYM(spc_map_check); SYM(spc_spatial_conv_sep_mfma_f32); SYM(spc_spatial_conv_sep_mfma_moments_f32);
    SYM(spc_stats_global_f32); SYM(spc_stats_axis_f32); SYM(spc_fits_to_f32); SYM(spc_map_conv2d_f64); SYM(spc_map_arith_f64); SYM(spc_percentile_axis0_f32); SYM(spc_percentile_axis2_f32); SYM(spc_fill_masked_f32); SYM(spc_sigma_clip_axis0_f32); SYM(spc_moment_order_spatial_f32); SYM(spc_mask_include_u8); SYM(spc_clip_outside_f32); SYM(spc_scale_f32);
    SYM(spc_argextrema_axis_f32); SYM(spc_percentile_global_f32); SYM(spc_key_histogram_f32); SYM(spc_key_to_f32); SYM(spc_fill_masked_transpose_f32); SYM(spc_pool_trim); SYM(spc_pool_stats); SYM(spc_clip_bounds_f32); SYM(spc_wcs_pixel_map_f64); SYM(spc_stats_planes_f32); SYM(spc_moments_f64); SYM(spc_moment_order_f64); SYM(spc_fits_to_f64); SYM(spc_stats_global_f64); SYM(spc_stats_axis_f64); SYM(spc_spectral_conv_f64); SYM(spc_spatial_conv_f64); SYM(spc_spectral_lerp_f64); SYM(spc_resample_bilinear_f64); SYM(spc_scale_f64); SYM(spc_narrow_f64_to_f32); SYM(spc_percentile_axis0_f64); SYM(spc_sigma_clip_axis0_f64); SYM(spc_mask_include_f64);
    SYM(spc_comm_unique_id); SYM(spc_comm_init); SYM(spc_comm_destroy); SYM(spc_allgather_rows); SYM(spc_allgather_rows_batch);

    int (*ver)(void) = (int (*)(void))dlsym(lib, "spc_abi_version");
    if (ver() != SPC_ABI_VERSION) { fprintf(stderr, "ABI version %d != header %d\n", ver(), SPC_ABI_VERSION); return 3; }

    /* argument checking happens before any device work: status code + thread-local message */
    moments_fn moments = (moments_fn)dlsym(lib, "spc_moments_f32");
    err_fn last_error = (err_fn)dlsym(lib, "spc_last_error");
    spc_cube_f32 bad; memset(&bad, 0, sizeof bad);
    spc_moment_outputs outs; memset(&outs, 0, sizeof outs);
    double cen0 = 0.0;
    if (moments(0, NULL, &bad, NULL, &cen0, 1.0, 0.0, &outs, NULL, 0) != SPC_ERR_INVALID || strlen(last_error()) == 0) {
        fprintf(stderr, "NULL cube was not rejected with SPC_ERR_INVALID\n"); return 4;
    }
    printf("abi ok: version %d, %s\n", ver(), "all symbols present");
    if (argc < 3 || strcmp(argv[2], "--gpu") != 0) return 0;

    count_fn count = (count_fn)dlsym(lib, "spc_device_count");
    int ndev = 0;
    if (count(&ndev) != SPC_OK || ndev < 1) { fprintf(stderr, "no GPU: %s\n", last_error()); return 5; }
    malloc_fn dmalloc = (malloc_fn)dlsym(lib, "spc_malloc");
    free_fn dfree = (free_fn)dlsym(lib, "spc_free");
    copy_fn h2d = (copy_fn)dlsym(lib, "spc_memcpy_h2d");
    copy_fn d2h = (copy_fn)dlsym(lib, "spc_memcpy_d2h");
    /* the reference's moment cube: data = arange(27).reshape(3,3,3), velocity axis cdelt 3 (cen = 0,3,6) */
    float data[27]; for (int i = 0; i < 27; ++i) data[i] = (float)i;
    double cen[3] = {0.0, 3.0, 6.0};
    void *d_data, *d_cen, *d_m0, *d_m1, *d_m2;
    if (dmalloc(0, sizeof data, &d_data) || dmalloc(0, sizeof cen, &d_cen) || dmalloc(0, 9 * 8, &d_m0) ||
        dmalloc(0, 9 * 8, &d_m1) || dmalloc(0, 9 * 8, &d_m2)) { fprintf(stderr, "%s\n", last_error()); return 6; }
    h2d(0, d_data, data, sizeof data, NULL); h2d(0, d_cen, cen, sizeof cen, NULL);
    spc_cube_f32 cube = {(const float*)d_data, 3, 3, 3, 3, 9};
    outs.d_m0 = (double*)d_m0; outs.d_m1 = (double*)d_m1; outs.d_m2 = (double*)d_m2;
    if (moments(0, NULL, &cube, NULL, (const double*)d_cen, 3.0, 2.0, &outs, NULL, 0) != SPC_OK) {
        fprintf(stderr, "spc_moments_f32 failed: %s\n", last_error()); return 7;
    }
    double m0[9], m1[9], m2[9];
    d2h(0, m0, d_m0, sizeof m0, NULL); d2h(0, m1, d_m1, sizeof m1, NULL); d2h(0, m2, d_m2, sizeof m2, NULL);
    /* the reference's golden table for axis 0 (test_moments.py:19-21, 30-32, 41-43), in units of dv = 3 */
    static const double M0V[9] = {27, 30, 33, 36, 39, 42, 45, 48, 51};
    static const double M1V[9] = {1.66666667, 1.6, 1.54545455, 1.5, 1.46153846, 1.42857143, 1.4, 1.375, 1.35294118};
    static const double M2V[9] = {0.22222222, 0.30666667, 0.36914601, 0.41666667, 0.45364892, 0.4829932,
                                  0.50666667, 0.52604167, 0.54209919};
    for (int p = 0; p < 9; ++p) {
        if (fabs(m0[p] - 3.0 * M0V[p]) > 1e-9 || fabs(m1[p] - (3.0 * M1V[p] + 2.0)) > 3e-8 * 3.0 ||
            fabs(m2[p] - 9.0 * M2V[p]) > 1e-7) {
            fprintf(stderr, "pixel %d: got %.10g %.10g %.10g\n", p, m0[p], m1[p], m2[p]); return 8;
        }
    }
    dfree(0, d_data); dfree(0, d_cen); dfree(0, d_m0); dfree(0, d_m1); dfree(0, d_m2);
    /* caller-owned scratch: spectral_smooth of the reference's 5x2x2 delta cube (test_regrid.py:138-172: the
     * smoothed spike equals the kernel samples) with a workspace sized by spc_workspace_bytes; a call with a
     * workspace that is too small is rejected before any device work */
    typedef size_t (*wsbytes_fn)(int, int64_t, int64_t, int64_t, int64_t, int64_t);
    typedef int (*sconv_fn)(int, void*, const spc_cube_f32*, const spc_mask*, const double*, int, float*, int64_t, int64_t,
                            void*, size_t);
    wsbytes_fn wsbytes = (wsbytes_fn)dlsym(lib, "spc_workspace_bytes");
    sconv_fn sconv = (sconv_fn)dlsym(lib, "spc_spectral_conv_f32");
    float delta[20]; memset(delta, 0, sizeof delta);
    for (int p = 0; p < 4; ++p) delta[2 * 4 + p] = 1.0f;                     /* channel 2 of (5,2,2) */
    double g[9], gs = 0.0;
    for (int j = 0; j < 9; ++j) { g[j] = exp(-0.5 * (j - 4) * (j - 4)); gs += g[j]; }
    void *d_delta, *d_sm, *d_ws;
    const size_t need = wsbytes(SPC_WS_SPECTRAL_CONV, 5, 2, 2, 9, 0);
    if (need == 0 || dmalloc(0, sizeof delta, &d_delta) || dmalloc(0, sizeof delta, &d_sm) || dmalloc(0, need, &d_ws)) {
        fprintf(stderr, "workspace query / allocation failed: %s\n", last_error()); return 9;
    }
    h2d(0, d_delta, delta, sizeof delta, NULL);
    spc_cube_f32 dc = {(const float*)d_delta, 5, 2, 2, 2, 4};
    if (sconv(0, NULL, &dc, NULL, g, 9, (float*)d_sm, 0, 0, d_ws, 0) != SPC_ERR_INVALID || !strstr(last_error(), "d_workspace")) {
        fprintf(stderr, "a too small workspace was not rejected\n"); return 10;
    }
    if (sconv(0, NULL, &dc, NULL, g, 9, (float*)d_sm, 0, 0, d_ws, need) != SPC_OK) {
        fprintf(stderr, "spc_spectral_conv_f32 failed: %s\n", last_error()); return 11;
    }
    float sm[20];
    d2h(0, sm, d_sm, sizeof sm, NULL);
    for (int z = 0; z < 5; ++z)
        for (int p = 0; p < 4; ++p)
            if (fabs(sm[z * 4 + p] - g[z + 2] / gs) > 1e-7) { fprintf(stderr, "smoothed delta: channel %d got %.9g\n", z, sm[z * 4 + p]); return 12; }
    dfree(0, d_delta); dfree(0, d_sm); dfree(0, d_ws);
    printf("gpu ok: 3x3x3 reference moment cube and the smoothed delta cube reproduced through the C ABI\n");
    return 0;
}
