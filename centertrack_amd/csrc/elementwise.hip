// HBM-bound glue ops of the DLA backbone / IDAUp neck on NHWC views (16-byte accesses
// along the channel axis, grid-stride), plus the NCHW <-> NHWC converters of the drop-in ops.
#include "ct_common.h"

namespace {

__global__ __launch_bounds__(256) void maxpool2x2_kernel(const float *x, int N, int H, int W, int C, int ldx,
                                                         float *y, int ldy)
{
    const int Ho = H >> 1, Wo = W >> 1, C4 = C >> 2;
    const size_t total = (size_t)N * Ho * Wo * C4;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int c4 = (int)(idx % C4);
        size_t p = idx / C4;
        const int ox = (int)(p % Wo); p /= Wo;
        const int oy = (int)(p % Ho);
        const int n = (int)(p / Ho);
        const float *src = x + (((size_t)n * H + 2 * oy) * W + 2 * ox) * ldx + c4 * 4;
        const f32x4 a = *reinterpret_cast<const f32x4 *>(src);
        const f32x4 b = *reinterpret_cast<const f32x4 *>(src + ldx);
        const f32x4 c = *reinterpret_cast<const f32x4 *>(src + (size_t)W * ldx);
        const f32x4 d = *reinterpret_cast<const f32x4 *>(src + (size_t)W * ldx + ldx);
        f32x4 m;
#pragma unroll
        for (int i = 0; i < 4; ++i) m[i] = fmaxf(fmaxf(a[i], b[i]), fmaxf(c[i], d[i]));
        *reinterpret_cast<f32x4 *>(y + (((size_t)n * Ho + oy) * Wo + ox) * ldy + c4 * 4) = m;
    }
}

// out[oy,ox,c] = skip[oy,ox,c] + sum_{jy,jx in {0,1}} x[iy-jy, ix-jx, c] * w[c, ky+f*jy, kx+f*jx]
// with (iy, ky) = divmod(oy + f/2, f): the depth-wise ConvTranspose2d(k=2f, s=f, p=f/2); w is the
// kernel transposed to [2f*2f][C] so that a thread's 4 channels are one 16-byte load.
__global__ __launch_bounds__(256) void upsample_add_kernel(const float *x, int N, int H, int W, int C, int ldx,
                                                           const float *w, int f, const float *skip, int lds,
                                                           float *y, int ldy)
{
    const int Ho = H * f, Wo = W * f, C4 = C >> 2, kw = 2 * f, p = f >> 1;
    const size_t total = (size_t)N * Ho * Wo * C4;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int c = (int)(idx % C4) * 4;
        size_t q = idx / C4;
        const int ox = (int)(q % Wo); q /= Wo;
        const int oy = (int)(q % Ho);
        const int n = (int)(q / Ho);
        const int iy = (oy + p) / f, ky = (oy + p) - iy * f;
        const int ix = (ox + p) / f, kx = (ox + p) - ix * f;
        const size_t opix = ((size_t)n * Ho + oy) * Wo + ox;
        f32x4 acc = *reinterpret_cast<const f32x4 *>(skip + opix * lds + c);
#pragma unroll
        for (int jy = 0; jy < 2; ++jy) {
            const int yy = iy - jy;
            if (yy < 0 || yy >= H) continue;
#pragma unroll
            for (int jx = 0; jx < 2; ++jx) {
                const int xx = ix - jx;
                if (xx < 0 || xx >= W) continue;
                const f32x4 v = *reinterpret_cast<const f32x4 *>(x + (((size_t)n * H + yy) * W + xx) * ldx + c);
                const int widx = (ky + f * jy) * kw + (kx + f * jx);
                const f32x4 wv = *reinterpret_cast<const f32x4 *>(w + (size_t)widx * C + c);     // w is [2f*2f][C]
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] += v[i] * wv[i];
            }
        }
        *reinterpret_cast<f32x4 *>(y + opix * ldy + c) = acc;
    }
}

// [C][HW] -> [HW][ld] (per image) through a 32x33 LDS tile
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float *x, int C, int HW, float *y, int ldy)
{
    __shared__ float tile[32][33];
    const int n = blockIdx.z;
    const int c0 = blockIdx.y * 32, p0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const float *src = x + (size_t)n * C * HW;
    float *dst = y + (size_t)n * HW * ldy;
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r, p = p0 + tx;
        tile[r][tx] = (c < C && p < HW) ? src[(size_t)c * HW + p] : 0.0f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int p = p0 + r, c = c0 + tx;
        if (p < HW && c < C) dst[(size_t)p * ldy + c] = tile[tx][r];
    }
}

__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const float *x, int C, int HW, int ldx, float *y)
{
    __shared__ float tile[32][33];
    const int n = blockIdx.z;
    const int c0 = blockIdx.y * 32, p0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const float *src = x + (size_t)n * HW * ldx;
    float *dst = y + (size_t)n * C * HW;
    for (int r = ty; r < 32; r += 8) {
        const int p = p0 + r, c = c0 + tx;
        tile[r][tx] = (p < HW && c < C) ? src[(size_t)p * ldx + c] : 0.0f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r, p = p0 + tx;
        if (c < C && p < HW) dst[(size_t)c * HW + p] = tile[tx][r];
    }
}

// prior heat-map: max-splat of (2r+1)^2 Gaussians, sigma = (2r+1)/6, evaluated in float64.
// One workgroup per 32x8 pixel tile: the stream's blob list is read once, culled against the tile
// into LDS (wave ballots), and every pixel only visits the blobs that can touch its tile.
constexpr int PHM_TW = 32, PHM_TH = 8, PHM_CAP = 256;
__global__ __launch_bounds__(256) void render_pre_hm_kernel(const int *params, const int *counts, int cap, int B,
                                                            int H, int W, float *out, int also_flipped)
{
    __shared__ int blob[PHM_CAP * 3];
    __shared__ int nblob;
    const int tilesX = (W + PHM_TW - 1) / PHM_TW, tilesY = (H + PHM_TH - 1) / PHM_TH;
    int bid = blockIdx.x;
    const int tx = bid % tilesX; bid /= tilesX;
    const int ty = bid % tilesY;
    const int b = bid / tilesY;
    const int x0 = tx * PHM_TW, y0 = ty * PHM_TH;
    const int n = min(counts[b], cap);
    const int *p = params + (size_t)b * cap * 3;
    const int lane = threadIdx.x & 63;
    const int x = x0 + (threadIdx.x & (PHM_TW - 1)), y = y0 + (threadIdx.x / PHM_TW);
    float v = 0.0f;
    // blobs are culled against the tile PHM_CAP at a time (any number of blobs per stream: K is not bounded)
    for (int c0 = 0; c0 == 0 || c0 < n; c0 += PHM_CAP) {
        const int c1 = min(n, c0 + PHM_CAP);
        if (threadIdx.x == 0) nblob = 0;
        __syncthreads();
        for (int i0 = c0; i0 < c1; i0 += 256) {
            const int i = i0 + threadIdx.x;
            int cx = 0, cy = 0, r = -1;
            if (i < c1) { cx = p[3 * i]; cy = p[3 * i + 1]; r = p[3 * i + 2]; }
            const bool hit = i < c1 && r >= 0 && cx + r >= x0 && cx - r < x0 + PHM_TW && cy + r >= y0 && cy - r < y0 + PHM_TH;
            const unsigned long long mask = __ballot(hit);
            int base = 0;
            if (mask) {
                const int leader = __ffsll((long long)mask) - 1;
                if (lane == leader) base = atomicAdd(&nblob, (int)__popcll(mask));
                base = __shfl(base, leader);
            }
            if (hit) {
                const int s = base + (int)__popcll(mask & ((1ull << lane) - 1ull));
                blob[3 * s] = cx; blob[3 * s + 1] = cy; blob[3 * s + 2] = r;
            }
        }
        __syncthreads();
        const int m = nblob;
        for (int i = 0; i < m; ++i) {
            const int cx = blob[3 * i], cy = blob[3 * i + 1], r = blob[3 * i + 2];
            const int dx = x - cx, dy = y - cy;
            if (dx < -r || dx > r || dy < -r || dy > r) continue;
            const double sigma = (double)(2 * r + 1) / 6.0;
            const double g = exp(-(double)(dx * dx + dy * dy) / (2.0 * sigma * sigma));
            v = fmaxf(v, (float)g);
        }
        __syncthreads();
    }
    if (x >= W || y >= H) return;
    out[((size_t)b * H + y) * W + x] = v;
    if (also_flipped) out[((size_t)(B + b) * H + y) * W + (W - 1 - x)] = v;
}

unsigned grid_for(size_t total)
{
    size_t b = (total + 255) / 256;
    if (b > 256 * 16) b = 256 * 16;
    if (b < 1) b = 1;
    return (unsigned)b;
}

}  // namespace

extern "C" int ct_maxpool2x2(const float *x, int N, int H, int W, int C, int ldx, float *y, int ldy, void *stream)
{
    if (!x || !y) CT_FAIL_ARG("ct_maxpool2x2: null pointer");
    if (C % 4 || ldx % 4 || ldy % 4 || ((uintptr_t)x & 15) || ((uintptr_t)y & 15))
        CT_FAIL_ARG("ct_maxpool2x2: C/ld must be multiples of 4 and pointers 16-byte aligned");
    if (H < 2 || W < 2 || N <= 0) CT_FAIL_ARG("ct_maxpool2x2: bad shape");
    const size_t total = (size_t)N * (H / 2) * (W / 2) * (C / 4);
    hipLaunchKernelGGL(maxpool2x2_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, N, H, W, C, ldx,
                       y, ldy);
    CT_CHECK_LAUNCH("ct_maxpool2x2");
    return CT_OK;
}

extern "C" int ct_upsample_add(const float *x, int N, int H, int W, int C, int ldx, const float *w, int f,
                               const float *skip, int lds, float *y, int ldy, void *stream)
{
    if (!x || !w || !skip || !y) CT_FAIL_ARG("ct_upsample_add: null pointer");
    if (f != 2 && f != 4 && f != 8) CT_FAIL_ARG("ct_upsample_add: f=%d unsupported", f);
    if (C % 4 || ldx % 4 || lds % 4 || ldy % 4 || ((uintptr_t)x & 15) || ((uintptr_t)skip & 15) || ((uintptr_t)y & 15))
        CT_FAIL_ARG("ct_upsample_add: C/ld must be multiples of 4 and pointers 16-byte aligned");
    const size_t total = (size_t)N * H * f * W * f * (C / 4);
    hipLaunchKernelGGL(upsample_add_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, N, H, W, C,
                       ldx, w, f, skip, lds, y, ldy);
    CT_CHECK_LAUNCH("ct_upsample_add");
    return CT_OK;
}

extern "C" int ct_nchw_to_nhwc(const float *x, int N, int C, int H, int W, float *y, int ldy, void *stream)
{
    if (!x || !y || ldy < C) CT_FAIL_ARG("ct_nchw_to_nhwc: bad arguments");
    const int HW = H * W;
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(ct_cdiv(HW, 32), ct_cdiv(C, 32), N), dim3(256), 0,
                       (hipStream_t)stream, x, C, HW, y, ldy);
    CT_CHECK_LAUNCH("ct_nchw_to_nhwc");
    return CT_OK;
}

extern "C" int ct_nhwc_to_nchw(const float *x, int N, int C, int H, int W, int ldx, float *y, void *stream)
{
    if (!x || !y || ldx < C) CT_FAIL_ARG("ct_nhwc_to_nchw: bad arguments");
    const int HW = H * W;
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(ct_cdiv(HW, 32), ct_cdiv(C, 32), N), dim3(256), 0,
                       (hipStream_t)stream, x, C, HW, ldx, y);
    CT_CHECK_LAUNCH("ct_nhwc_to_nchw");
    return CT_OK;
}

extern "C" int ct_render_pre_hm(const int *params, const int *counts, int cap, int B, int H, int W, float *out,
                                int also_flipped, void *stream)
{
    if (!params || !counts || !out || cap <= 0 || B <= 0 || H <= 0 || W <= 0) CT_FAIL_ARG("ct_render_pre_hm: bad arguments");
    const long blocks = (long)B * ((H + PHM_TH - 1) / PHM_TH) * ((W + PHM_TW - 1) / PHM_TW);
    hipLaunchKernelGGL(render_pre_hm_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, params, counts,
                       cap, B, H, W, out, also_flipped);
    CT_CHECK_LAUNCH("ct_render_pre_hm");
    return CT_OK;
}
