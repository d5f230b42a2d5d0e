// Host-side image pre-processing of Detector.pre_process (src/lib/detector.py:207-239): the affine crop /
// scale of the original u8 frame to the network input (cv2.warpAffine, INTER_LINEAR, constant border 0)
// followed by ((x / 255 - mean) / std) and HWC -> CHW.  Runs on the CPU like the reference's (there it
// executes in the DataLoader worker processes of test.py:22-51, so it must not need a GPU).
//
// cv2 is absent from this environment, so the warp restates OpenCV's published fixed-point algorithm
// (modules/imgproc/src/imgwarp.cpp, warpAffine + remapBilinear for CV_8U): the inverse map is evaluated in
// 10-bit fixed point per row / column, coordinates are quantised to 1/32 pixel, the four taps are blended
// with 15-bit integer weights (32-fx)(32-fy)*32 ... and rounded (+2^14 >> 15).  PARITY UNPINNED against
// cv2 itself (no reference vectors exist); pinned against the independent numpy restatement in oracle/.
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include <thread>
#include <vector>

#include "ct_common.h"

namespace {
inline int cv_round(double v) { return (int)lrint(v); }        // saturate_cast<int>(double): round half to even
}

// dst -> src map of cv::warpAffine: the 2x3 forward map inverted with OpenCV's own sequence of float64 operations
// (shared with the device path, csrc/preprocess.hip; this file is compiled with -ffp-contract=off)
void ct_affine_inverse(const double *trans, double *M)
{
    for (int i = 0; i < 6; ++i) M[i] = trans[i];
    double D = M[0] * M[4] - M[1] * M[3];
    D = D != 0 ? 1. / D : 0;
    const double A11 = M[4] * D, A22 = M[0] * D;
    M[0] = A11; M[1] *= -D;
    M[3] *= -D; M[4] = A22;
    const double b1 = -M[0] * M[2] - M[1] * M[5];
    const double b2 = -M[3] * M[2] - M[4] * M[5];
    M[2] = b1; M[5] = b2;
}

extern "C" int ct_preprocess_image(const uint8_t *img, int h, int w, int stride, int channels, const double *trans,
                                   int dst_w, int dst_h, const float *mean, const float *stdv, float *out, int flip_copy)
{
    if (!img || !trans || !mean || !stdv || !out) CT_FAIL_ARG("ct_preprocess_image: null pointer");
    if (h <= 0 || w <= 0 || dst_w <= 0 || dst_h <= 0 || channels < 1 || channels > 4 || stride < w * channels)
        CT_FAIL_ARG("ct_preprocess_image: bad shape");
    double M[6];
    ct_affine_inverse(trans, M);

    const int AB_BITS = 10, AB_SCALE = 1 << AB_BITS, INTER_BITS = 5, INTER_TAB_SIZE = 1 << INTER_BITS;
    const int round_delta = AB_SCALE / INTER_TAB_SIZE / 2;
    std::vector<int> adelta(dst_w), bdelta(dst_w);
    for (int x = 0; x < dst_w; ++x) {
        adelta[x] = cv_round(M[0] * x * AB_SCALE);
        bdelta[x] = cv_round(M[3] * x * AB_SCALE);
    }
    // ((v / 255. - mean) / std).astype(float32) has 256 possible inputs per channel: float64 arithmetic, one rounding
    // at the end, evaluated once per (channel, value) instead of once per pixel
    float lut[4][256];
    for (int c = 0; c < channels; ++c)
        for (int v = 0; v < 256; ++v) lut[c][v] = (float)(((double)v / 255.0 - (double)mean[c]) / (double)stdv[c]);
    const size_t plane = (size_t)dst_w * dst_h;
    // rows are independent: CENTERTRACK_HOST_THREADS > 1 lets that many host threads share them (results do not depend
    // on the split).  Default 1: the function usually runs in DataLoader worker processes that are the parallelism
    // already (test.py:75), and on a 4-vCPU container 4 threads measured slower than 1.
    int nthreads = 1;
    if (const char *e = getenv("CENTERTRACK_HOST_THREADS")) nthreads = atoi(e);
    if (nthreads > dst_h / 64) nthreads = dst_h / 64;
    if (nthreads < 1) nthreads = 1;
    auto do_rows = [&](int y_begin, int y_end) {
    for (int y = y_begin; y < y_end; ++y) {
        const int X0 = cv_round((M[1] * y + M[2]) * AB_SCALE) + round_delta;
        const int Y0 = cv_round((M[4] * y + M[5]) * AB_SCALE) + round_delta;
        float *orow = out + (size_t)y * dst_w;
        for (int x = 0; x < dst_w; ++x) {
            const int X = (X0 + adelta[x]) >> (AB_BITS - INTER_BITS);
            const int Y = (Y0 + bdelta[x]) >> (AB_BITS - INTER_BITS);
            int sx = X >> INTER_BITS, sy = Y >> INTER_BITS;
            if (sx < -32768) sx = -32768; if (sx > 32767) sx = 32767;       // saturate_cast<short>
            if (sy < -32768) sy = -32768; if (sy > 32767) sy = 32767;
            const int fx = X & (INTER_TAB_SIZE - 1), fy = Y & (INTER_TAB_SIZE - 1);
            const int w00 = (32 - fx) * (32 - fy) * 32, w01 = fx * (32 - fy) * 32;
            const int w10 = (32 - fx) * fy * 32, w11 = fx * fy * 32;
            int vals[4];
            if ((unsigned)sx < (unsigned)(w - 1) && (unsigned)sy < (unsigned)(h - 1)) {
                // all four taps inside the image (the common case): no border logic
                const uint8_t *p0 = img + (size_t)sy * stride + (size_t)sx * channels, *p1 = p0 + stride;
                for (int c = 0; c < channels; ++c)
                    vals[c] = (p0[c] * w00 + p0[channels + c] * w01 + p1[c] * w10 + p1[channels + c] * w11 + (1 << 14)) >> 15;
            } else {
                const bool x0 = sx >= 0 && sx < w, x1 = sx + 1 >= 0 && sx + 1 < w;
                const bool y0 = sy >= 0 && sy < h, y1 = sy + 1 >= 0 && sy + 1 < h;
                for (int c = 0; c < channels; ++c) {
                    const int p00 = (x0 && y0) ? img[(size_t)sy * stride + sx * channels + c] : 0;
                    const int p01 = (x1 && y0) ? img[(size_t)sy * stride + (sx + 1) * channels + c] : 0;
                    const int p10 = (x0 && y1) ? img[(size_t)(sy + 1) * stride + sx * channels + c] : 0;
                    const int p11 = (x1 && y1) ? img[(size_t)(sy + 1) * stride + (sx + 1) * channels + c] : 0;
                    vals[c] = (p00 * w00 + p01 * w01 + p10 * w10 + p11 * w11 + (1 << 14)) >> 15;
                }
            }
            for (int c = 0; c < channels; ++c) {
                int v = vals[c];
                if (v < 0) v = 0; if (v > 255) v = 255;
                const float f = lut[c][v];
                orow[(size_t)c * plane + x] = f;
                if (flip_copy) orow[(size_t)(channels + c) * plane + (dst_w - 1 - x)] = f;
            }
        }
    }
    };
    if (nthreads == 1) {
        do_rows(0, dst_h);
    } else {
        std::vector<std::thread> pool;
        const int per = (dst_h + nthreads - 1) / nthreads;
        for (int t = 1; t < nthreads; ++t) pool.emplace_back(do_rows, t * per, (t + 1) * per < dst_h ? (t + 1) * per : dst_h);
        do_rows(0, per < dst_h ? per : dst_h);
        for (auto &th : pool) th.join();
    }
    return CT_OK;
}
