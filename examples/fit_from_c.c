/* fit_from_c.c -- the C ABI used from plain C (no Python, no torch): what a ccall/cgo/JNI binding
 * does.  Reads a little-endian corner file
 *     int32 P, int32 N, int32 width, int32 height, float obj[3*P], float img[2*P*N]
 * runs the one-shot drop-in for fit_model / OpenCV.calibrateCamera (src/detect_fit.jl:27-61 of the
 * reference: flags word of :40, criteria (30, 1e-3)) and prints the NamedTuple fields.
 *
 *   gcc -O2 -Iinclude examples/fit_from_c.c -o fit_from_c -Lcameracalibrations_amd -lccal_hip \
 *       -Wl,-rpath,$PWD/cameracalibrations_amd -lm
 *   ./fit_from_c corners.bin [with_distortion=1] [aspect=1.0]
 */
#include <stdio.h>
#include <stdlib.h>

#include "ccal.h"

int main(int argc, char **argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s corners.bin [with_distortion] [aspect]\n", argv[0]); return 2; }
    const int with_distortion = argc > 2 ? atoi(argv[2]) : 1;
    const double aspect = argc > 3 ? atof(argv[3]) : 1.0;
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 2; }
    int32_t hdr[4];
    if (fread(hdr, sizeof(int32_t), 4, f) != 4) { fprintf(stderr, "short header\n"); return 2; }
    const int32_t P = hdr[0], N = hdr[1], width = hdr[2], height = hdr[3];
    float *obj = malloc(sizeof(float) * 3 * (size_t)P), *img = malloc(sizeof(float) * 2 * (size_t)P * N);
    double *rvecs = malloc(sizeof(double) * 3 * (size_t)N), *tvecs = malloc(sizeof(double) * 3 * (size_t)N);
    if (!obj || !img || !rvecs || !tvecs) return 2;
    if (fread(obj, sizeof(float), 3 * (size_t)P, f) != 3 * (size_t)P ||
        fread(img, sizeof(float), 2 * (size_t)P * N, f) != 2 * (size_t)P * N) { fprintf(stderr, "short file\n"); return 2; }
    fclose(f);
    if (ccal_device_count() < 1) { fprintf(stderr, "no gfx950 device\n"); return 3; }
    /* src/detect_fit.jl:40 */
    const uint32_t flags = CCAL_ZERO_TANGENT_DIST | CCAL_FIX_K3 | CCAL_FIX_K2 | (with_distortion ? 0u : CCAL_FIX_K1) |
                           CCAL_FIX_ASPECT_RATIO;
    double K4[4], dist5[5], rms = 0.0;
    int32_t iters = 0;
    ccal_stats st;
    const int rc = ccal_calibrate(obj, img, P, N, width, height, aspect, flags, 30, 1e-3, 0, K4, dist5, rvecs, tvecs, &rms,
                                  &iters, &st);
    if (rc < 0) { fprintf(stderr, "ccal_calibrate: %d: %s\n", rc, ccal_last_error()); return 1; }
    printf("status %d iters %d rms %.17g\n", rc, (int)iters, rms);
    printf("frow %.17g fcol %.17g crow %.17g ccol %.17g k %.17g\n", K4[0], K4[1], K4[2], K4[3], dist5[0]);
    for (int32_t i = 0; i < N; i++)
        printf("R %.17g %.17g %.17g t %.17g %.17g %.17g\n", rvecs[3 * i], rvecs[3 * i + 1], rvecs[3 * i + 2], tvecs[3 * i],
               tvecs[3 * i + 1], tvecs[3 * i + 2]);
    /* the batched form of the same seam: five calibrations in one call (here: the same corners five times; in a caller's hands
     * one set per camera) -- small fits share merged launches on the device; every result must equal the single call's */
    enum { NCAL = 5 };
    const float *imgs[NCAL];
    int32_t Ns[NCAL], status[NCAL];
    double *outs[NCAL];
    for (int c = 0; c < NCAL; c++) {
        imgs[c] = img;
        Ns[c] = N;
        outs[c] = malloc(sizeof(double) * (11 + 6 * (size_t)N));
        if (!outs[c]) return 2;
    }
    const int rcm = ccal_calibrate_many(obj, P, NCAL, imgs, Ns, width, height, aspect, flags, 30, 1e-3, 0, outs, status);
    if (rcm < 0) { fprintf(stderr, "ccal_calibrate_many: %d: %s\n", rcm, ccal_last_error()); return 1; }
    int same = 1;
    for (int c = 0; c < NCAL; c++) {
        const double *o = outs[c];
        same = same && status[c] == rc && o[0] == K4[0] && o[1] == K4[1] && o[2] == K4[2] && o[3] == K4[3] && o[4] == dist5[0] &&
               o[9] == rms && (int32_t)o[10] == iters;
        for (int32_t i = 0; i < 3 * N; i++) same = same && o[11 + i] == rvecs[i] && o[11 + 3 * (size_t)N + i] == tvecs[i];
        free(outs[c]);
    }
    printf("calibrate_many %d calibrations identical %d\n", (int)NCAL, same);
    ccal_thread_cleanup();
    free(obj); free(img); free(rvecs); free(tvecs);
    return same ? 0 : 1;
}
