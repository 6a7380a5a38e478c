// ccal_detect.hpp -- device checkerboard-corner detector (see ccal_detect.hip)
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>

namespace ccal {

class Detector {
  public:
    static constexpr int MAX_PEAKS = 8192;
    ~Detector();
    // gray: host uint8 image, element (row, col) at gray[row * sr + col * sc] (Julia column-major:
    // sr = 1, sc = rows; numpy C order: sr = cols, sc = 1).  corners: [2, n1*n2] (row, col), first
    // n_corners dimension fastest.  stage_out (optional, 2 * rows * cols doubles): blurred image and
    // response, canonical [rows][cols] layout.  Returns 0 ok (found tells whether a board was found),
    // < 0 error.
    int detect(hipStream_t stream, const uint8_t *gray, int rows, int cols, long long sr, long long sc, int n1, int n2,
               double *corners, int *found, double *stage_out, std::string &err);

    // The same in three steps, so that a caller can keep several images in flight (one Detector and one stream per image slot:
    // ccal_detect_corners_many): begin = upload + the image-wide kernels up to the peak list (nothing waited for);
    // mid = wait for the peak list, sort it, launch the sub-pixel refinement; end = wait for it, merge, lattice, ordering.
    // `gray` must stay valid until end() has returned.  detect() = begin + mid + end.
    int begin(hipStream_t stream, const uint8_t *gray, int rows, int cols, long long sr, long long sc, std::string &err);
    int mid(hipStream_t stream, int n1, int n2, double *stage_out, std::string &err);
    int end(hipStream_t stream, int n1, int n2, double *corners, int *found, std::string &err);

  private:
    // between the steps of one image
    const uint8_t *cur_gray = nullptr;
    int cur_rows = 0, cur_cols = 0, cur_n = 0;
    long long cur_sr = 0, cur_sc = 0;
    bool cur_refining = false;
    // the upload + six image-wide launches of begin() as one replayable graph per image shape; the completion words count a
    // device-side sequence number that pin_seq mirrors (seq_resync: set the device's from the host's before the next image)
    void *g_begin = nullptr;
    int g_rows = 0, g_cols = 0;
    long long g_sr = 0, g_sc = 0;
    bool seq_resync = true;
    int last_rows = 0, last_cols = 0;
    long long last_sr = 0, last_sc = 0;
    void *cur_peaks = nullptr;     // std::vector<Peak> of the image in flight (opaque here: Peak lives in the .hip)
    void release();
    uint8_t *d_raw = nullptr;
    double *d_img = nullptr, *d_tmp = nullptr, *d_g = nullptr, *d_R = nullptr, *d_out = nullptr, *d_taps = nullptr;
    void *d_list = nullptr, *d_misc = nullptr;
    size_t cap_px = 0;
    // pinned, device-mapped host block: completion word | peak count | peak list | refined corners | image staging.
    // The host waits on the completion word (a one-thread kernel behind the stage raises it) instead of paying the
    // wake-up latency of a stream synchronisation twice per image, and no copy of a call blocks.
    unsigned char *h_pin = nullptr;
    size_t pin_img_cap = 0;
    unsigned long long pin_seq = 0;
    int wait_flag(hipStream_t stream, unsigned long long want, std::string &err);
};

}  // namespace ccal
