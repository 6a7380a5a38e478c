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

  private:
    void release();
    uint8_t *d_raw = nullptr;
    double *d_img = nullptr, *d_tmp = nullptr, *d_g = nullptr, *d_R = nullptr, *d_out = nullptr, *d_taps = nullptr;
    void *d_list = nullptr, *d_misc = nullptr;
    size_t cap_px = 0;
};

}  // namespace ccal
