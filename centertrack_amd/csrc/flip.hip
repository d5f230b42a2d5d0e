// flip_test on the device: the mirrored copy of the input batch and the merge of the two halves of the
// head maps (Detector._flip_output, src/lib/detector.py:311-332; flip_tensor / flip_lr / flip_lr_off,
// src/lib/model/utils.py:28-50) as ONE launch each, so that a flip_test frame is a fixed sequence of
// libcentertrack_hip launches like every other frame (one HIP graph, no tensor-library op in between).
//
// merge:  dst[b,c,y,x] = ( src[b,c,y,x] + sgn(c) * src[B+b, perm(c), y, w-1-x] ) / 2
//   CT_FLIP_AVG           hm, wh, dep, dim         perm = id,            sgn = +1
//   CT_FLIP_NEG_EVEN      amodel_offset            perm = id,            sgn = -1 on even channels (the x offsets)
//   CT_FLIP_JOINTS        hm_hp [J]                perm = left/right partner joint, sgn = +1        (flip_lr)
//   CT_FLIP_JOINT_OFFSETS hps [2J]                 perm = partner joint's (x,y), sgn = -1 on x       (flip_lr_off)
// Heads the reference keeps from the un-flipped image only (reg, tracking, ltrb*, rot, ...) need no work:
// the decoder reads the first B images of the 2B-image map in place.
// fp32: (a + s*b) * 0.5f is bit-identical to torch's (a + flip(b)) / 2 (s = +-1 and the halving are exact).
#include "ct_common.h"

namespace {

constexpr int FLIP_MAX_HEADS = 8, FLIP_MAX_JOINTS = 64;

struct FlipArgs {
    ct_flip_head heads[FLIP_MAX_HEADS];
    unsigned char partner[FLIP_MAX_JOINTS];
    int nheads, B, h, w;
};

template <bool VEC>
__global__ __launch_bounds__(256) void flip_merge_kernel(FlipArgs a)
{
    const ct_flip_head hd = a.heads[blockIdx.y];
    const int wq = VEC ? a.w >> 2 : a.w;
    const size_t plane = (size_t)a.h * a.w;
    const size_t total = (size_t)a.B * hd.C * a.h * wq;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int xq = (int)(idx % wq);
        size_t q = idx / wq;
        const int y = (int)(q % a.h); q /= a.h;
        const int c = (int)(q % hd.C);
        const int b = (int)(q / hd.C);
        int cs = c;
        float sgn = 1.0f;
        if (hd.mode == CT_FLIP_NEG_EVEN) {
            sgn = (c & 1) ? 1.0f : -1.0f;
        } else if (hd.mode == CT_FLIP_JOINTS) {
            cs = a.partner[c];
        } else if (hd.mode == CT_FLIP_JOINT_OFFSETS) {
            cs = 2 * a.partner[c >> 1] + (c & 1);
            sgn = (c & 1) ? 1.0f : -1.0f;
        }
        const float *s0 = hd.src + (size_t)b * hd.src_batch_stride + (size_t)c * plane + (size_t)y * a.w;
        const float *s1 = hd.src + (size_t)(a.B + b) * hd.src_batch_stride + (size_t)cs * plane + (size_t)y * a.w;
        float *d = hd.dst + ((size_t)b * hd.C + c) * plane + (size_t)y * a.w;
        if (VEC) {
            const int x = xq << 2;
            const f32x4 u = *reinterpret_cast<const f32x4 *>(s0 + x);
            const f32x4 m = *reinterpret_cast<const f32x4 *>(s1 + (a.w - 4 - x));
            *reinterpret_cast<f32x4 *>(d + x) = f32x4{(u[0] + sgn * m[3]) * 0.5f, (u[1] + sgn * m[2]) * 0.5f,
                                                      (u[2] + sgn * m[1]) * 0.5f, (u[3] + sgn * m[0]) * 0.5f};
        } else {
            d[xq] = (s0[xq] + sgn * s1[a.w - 1 - xq]) * 0.5f;
        }
    }
}

template <bool VEC>
__global__ __launch_bounds__(256) void flip_rows_kernel(const float *src, float *dst, size_t rows, int W)
{
    const int wq = VEC ? W >> 2 : W;
    const size_t total = rows * wq;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int xq = (int)(idx % wq);
        const size_t r = idx / wq;
        if (VEC) {
            const f32x4 m = *reinterpret_cast<const f32x4 *>(src + r * W + (W - 4 - (xq << 2)));
            *reinterpret_cast<f32x4 *>(dst + r * W + (xq << 2)) = f32x4{m[3], m[2], m[1], m[0]};
        } else {
            dst[r * W + xq] = src[r * W + (W - 1 - xq)];
        }
    }
}

unsigned blocks_for(size_t threads)
{
    size_t b = (threads + 255) / 256;
    if (b > 4096) b = 4096;
    return (unsigned)(b < 1 ? 1 : b);
}

}  // namespace

extern "C" int ct_flip_merge(const ct_flip_head *heads, int nheads, const int *flip_idx, int npairs, int B, int h, int w,
                             void *stream)
{
    if (!heads || nheads <= 0 || nheads > FLIP_MAX_HEADS) CT_FAIL_ARG("ct_flip_merge: 1..%d heads per call", FLIP_MAX_HEADS);
    if (B <= 0 || h <= 0 || w <= 0) CT_FAIL_ARG("ct_flip_merge: bad shape");
    FlipArgs a;
    a.nheads = nheads; a.B = B; a.h = h; a.w = w;
    for (int j = 0; j < FLIP_MAX_JOINTS; ++j) a.partner[j] = (unsigned char)j;
    for (int k = 0; k < npairs; ++k) {
        const int u = flip_idx ? flip_idx[2 * k] : -1, v = flip_idx ? flip_idx[2 * k + 1] : -1;
        if (u < 0 || v < 0 || u >= FLIP_MAX_JOINTS || v >= FLIP_MAX_JOINTS) CT_FAIL_ARG("ct_flip_merge: bad joint pair %d", k);
        // successive swaps like the reference's loop over flip_idx (the published tables are disjoint pairs)
        const unsigned char t = a.partner[u]; a.partner[u] = a.partner[v]; a.partner[v] = t;
    }
    bool vec = (w & 3) == 0;
    size_t most = 0;
    for (int i = 0; i < nheads; ++i) {
        const ct_flip_head &hd = heads[i];
        if (!hd.src || !hd.dst || hd.C <= 0) CT_FAIL_ARG("ct_flip_merge: head %d: null pointer / no channels", i);
        if (hd.mode < CT_FLIP_AVG || hd.mode > CT_FLIP_JOINT_OFFSETS) CT_FAIL_ARG("ct_flip_merge: head %d: unknown mode %d", i, hd.mode);
        if (hd.mode == CT_FLIP_JOINTS && hd.C > FLIP_MAX_JOINTS) CT_FAIL_ARG("ct_flip_merge: more than %d joints", FLIP_MAX_JOINTS);
        if (hd.mode == CT_FLIP_JOINT_OFFSETS && ((hd.C & 1) || hd.C / 2 > FLIP_MAX_JOINTS)) CT_FAIL_ARG("ct_flip_merge: hps needs 2J channels, J <= %d", FLIP_MAX_JOINTS);
        if (hd.src_batch_stride < (size_t)hd.C * h * w) CT_FAIL_ARG("ct_flip_merge: head %d: batch stride smaller than an image", i);
        vec = vec && (((uintptr_t)hd.src | (uintptr_t)hd.dst) & 15) == 0 && (hd.src_batch_stride & 3) == 0;
        a.heads[i] = hd;
        const size_t n = (size_t)B * hd.C * h * w;
        if (n > most) most = n;
    }
    const dim3 grid(blocks_for(vec ? most / 4 : most), (unsigned)nheads);
    if (vec) hipLaunchKernelGGL(flip_merge_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(flip_merge_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, a);
    CT_CHECK_LAUNCH("ct_flip_merge");
    return CT_OK;
}

extern "C" int ct_flip_images(const float *src, float *dst, size_t rows, int W, void *stream)
{
    if (!src || !dst || rows == 0 || W <= 0) CT_FAIL_ARG("ct_flip_images: bad arguments");
    if (src == dst) CT_FAIL_ARG("ct_flip_images: in-place mirroring is not supported");
    const bool vec = (W & 3) == 0 && ((((uintptr_t)src | (uintptr_t)dst) & 15) == 0);
    if (vec) hipLaunchKernelGGL(flip_rows_kernel<true>, dim3(blocks_for(rows * (W / 4))), dim3(256), 0, (hipStream_t)stream, src, dst, rows, W);
    else hipLaunchKernelGGL(flip_rows_kernel<false>, dim3(blocks_for(rows * W)), dim3(256), 0, (hipStream_t)stream, src, dst, rows, W);
    CT_CHECK_LAUNCH("ct_flip_images");
    return CT_OK;
}
