/*
 * calibrate_c_abi.c -- the C ABI of libgeocalib_hip.so used from plain C: no Python, no torch.
 *
 *   gcc -std=c99 -D__HIP_PLATFORM_AMD__ examples/calibrate_c_abi.c -Iinclude -I/opt/rocm/include \
 *       -Lgeocalib_amd/lib -lgeocalib_hip -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,/opt/rocm/lib -o calibrate_c_abi
 *   ./calibrate_c_abi [camera_model 0..3] [B] [H] [W]
 *
 * Generates B synthetic perspective fields on the device (gclm_synth_fields), calibrates them with
 * gclm_calibrate (20 LM steps, like BASELINE) and prints, per image, ground truth vs estimate.  This is what a
 * binding of the reference's `GeoCalib.optimizer` seam (geocalib/geocalib.py:119) does underneath.
 */
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>

#include "gclm.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

int main(int argc, char** argv) {
    const int model = argc > 1 ? atoi(argv[1]) : GCLM_PINHOLE;
    const int B = argc > 2 ? atoi(argv[2]) : 4, H = argc > 3 ? atoi(argv[3]) : 240, W = argc > 4 ? atoi(argv[4]) : 320;
    const size_t N = (size_t)H * W;
    float *up, *lat, *upc, *latc, *gt_cam, *gt_grav, *cam, *grav, *info;
    CHECK_HIP(hipMalloc((void**)&up, sizeof(float) * 2 * N * B));
    CHECK_HIP(hipMalloc((void**)&lat, sizeof(float) * N * B));
    CHECK_HIP(hipMalloc((void**)&upc, sizeof(float) * N * B));
    CHECK_HIP(hipMalloc((void**)&latc, sizeof(float) * N * B));
    CHECK_HIP(hipMalloc((void**)&gt_cam, sizeof(float) * GCLM_CAM_STRIDE * B));
    CHECK_HIP(hipMalloc((void**)&gt_grav, sizeof(float) * GCLM_GRAV_STRIDE * B));
    CHECK_HIP(hipMalloc((void**)&cam, sizeof(float) * GCLM_CAM_STRIDE * B));
    CHECK_HIP(hipMalloc((void**)&grav, sizeof(float) * GCLM_GRAV_STRIDE * B));
    CHECK_HIP(hipMalloc((void**)&info, sizeof(float) * GCLM_INFO_STRIDE * B));
    hipStream_t stream;
    CHECK_HIP(hipStreamCreate(&stream));

    if (gclm_synth_fields(model, 42, 0, B, H, W, 0.02f, up, lat, upc, latc, gt_cam, gt_grav, stream) != 0) {
        fprintf(stderr, "gclm_synth_fields failed\n");
        return 3;
    }
    if (gclm_version() != GCLM_VERSION || gclm_abi_config_size() != (int)sizeof(gclm_config)) {
        fprintf(stderr, "libgeocalib_hip.so is ABI %d (config %d bytes), this program was built for %d (%d bytes)\n",
                gclm_version(), gclm_abi_config_size(), GCLM_VERSION, (int)sizeof(gclm_config));
        return 4;
    }
    gclm_config cfg;
    gclm_default_config(&cfg);
    cfg.device = 0;
    cfg.camera_model = model;
    cfg.num_steps = 20;
    cfg.early_stop = 0;
    gclm_handle* h = NULL;
    if (gclm_create(&h, &cfg) != 0) {
        fprintf(stderr, "gclm_create: %s\n", gclm_last_error(NULL));
        return 4;
    }
    if (gclm_calibrate(h, up, lat, upc, latc, B, H, W, NULL, NULL, NULL, NULL, 0, cam, grav, info, stream) != 0) {
        fprintf(stderr, "gclm_calibrate: %s\n", gclm_last_error(h));
        return 5;
    }
    float* hc = (float*)malloc(sizeof(float) * GCLM_CAM_STRIDE * B * 2);
    float* hg = (float*)malloc(sizeof(float) * GCLM_GRAV_STRIDE * B * 2);
    float* hi = (float*)malloc(sizeof(float) * GCLM_INFO_STRIDE * B);
    CHECK_HIP(hipMemcpyAsync(hc, cam, sizeof(float) * GCLM_CAM_STRIDE * B, hipMemcpyDeviceToHost, stream));
    CHECK_HIP(hipMemcpyAsync(hc + GCLM_CAM_STRIDE * B, gt_cam, sizeof(float) * GCLM_CAM_STRIDE * B, hipMemcpyDeviceToHost, stream));
    CHECK_HIP(hipMemcpyAsync(hg, grav, sizeof(float) * GCLM_GRAV_STRIDE * B, hipMemcpyDeviceToHost, stream));
    CHECK_HIP(hipMemcpyAsync(hg + GCLM_GRAV_STRIDE * B, gt_grav, sizeof(float) * GCLM_GRAV_STRIDE * B, hipMemcpyDeviceToHost, stream));
    CHECK_HIP(hipMemcpyAsync(hi, info, sizeof(float) * GCLM_INFO_STRIDE * B, hipMemcpyDeviceToHost, stream));
    CHECK_HIP(hipStreamSynchronize(stream));
    printf("gclm %d, workspace %zu bytes\n", gclm_version(), gclm_workspace_bytes(h));
    for (int b = 0; b < B; ++b) {
        const float* e = hc + b * GCLM_CAM_STRIDE;
        const float* t = hc + (B + b) * GCLM_CAM_STRIDE;
        printf("image %d: f %.4f (gt %.4f) k1 %.5f (gt %.5f) g (%.5f %.5f %.5f) gt (%.5f %.5f %.5f) final_cost %.6e "
               "focal_sigma %.4f stop_at %.0f\n", b, e[3], t[3], e[6], t[6], hg[b * 3], hg[b * 3 + 1], hg[b * 3 + 2],
               hg[(B + b) * 3], hg[(B + b) * 3 + 1], hg[(B + b) * 3 + 2], hi[b * GCLM_INFO_STRIDE + GCLM_INFO_FINAL_COST],
               hi[b * GCLM_INFO_STRIDE + GCLM_INFO_FOCAL_UNC], hi[b * GCLM_INFO_STRIDE + GCLM_INFO_STOP_AT]);
    }
    gclm_destroy(h);
    return 0;
}
