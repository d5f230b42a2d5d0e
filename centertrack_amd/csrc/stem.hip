// The three 7x7 stems of DLA (image, previous image, prior heat-map), fused:
//   y[n,p,:] = relu(bn0(conv7x7(x)))  + relu(bn1(conv7x7(pre_img))) + relu(bn2(conv7x7(pre_hm)))
// Inputs are the reference's NCHW planes; output is NHWC with 16 channels.
//
// GEMM view per stem: M = pixels, N = 16 couts, K = Cin*49 (147 / 147 / 49, padded to a
// multiple of 4) on v_mfma_f32_16x16x4_f32.  A workgroup (4 waves) owns an 8 x 32 pixel
// tile: all 7 input planes of the tile (+3 px halo, zero padded) and all three weight
// matrices sit in LDS; an A operand is a single ds_read_b32 at plane[tab[k] + pixel], where
// tab[] maps k = (ci,ky,kx) to its patch offset.  The plane pitch is 40 floats so that the
// k -> k+1 row wrap (kx 6 -> 0) lands on a different bank.
#include "ct_common.h"

namespace {

// Tile = TH x 32 pixels, TH = 8 (2 rows per wave) or 16 (4 rows per wave, round 4): at 512 x 512 x 1 stream the 8-row grid is
// 1024 workgroups of 140 VGPRs -- 768 resident, then a second round of 256 that runs one wave per SIMD (43 us for 18 us of
// MFMA time); 16-row tiles are 512 workgroups, all resident, two waves per SIMD, and the 22 KB of weights are staged half
// as often.  Same per-pixel arithmetic in the same order: bit-identical.
constexpr int ST_TW = 32;
constexpr int ST_PW = 40;                              // plane pitch 40 (38 used)
template <int TH> struct StCfg {
    static constexpr int PH = TH + 6;
    static constexpr int PS = PH * ST_PW;              // floats per plane
    static constexpr int NPV = (3 * PH * 38 + 255) / 256;      // plane values staged per thread (3 planes)
    static constexpr int HALVES = TH / 8;              // the workgroup walks its tile in slabs of 8 rows (2 rows = 4 m-tiles per wave)
};
__host__ __device__ constexpr int st_k4(int s) { return s == 2 ? 52 : 148; }   // K padded to 4
__host__ __device__ constexpr int st_koff(int s) { return s * 148; }            // offsets into the k tables
constexpr int ST_KTOT = 348;
constexpr int ST_WLP = 16;

CT_DEFINE_STAMPS(stem)      // (tools/conv_phases.py; expands to nothing in the shipped build)

struct StemArgs {
    const float *in[3];   // nullptr = stem not part of this launch
    const float *w[3];
    const float *scale;   // [3][16]
    const float *shift;   // [3][16]
    const float *add;     // NHWC [N,H,W,16] (pitch ldadd) the stems of this launch are accumulated on, or nullptr (= 0)
    float *y;
    int N, H, W, ldy, ldadd, tilesX, tilesY;
};

// per-thread staging register count of the weights (148*16/256 -> 10)
constexpr int ST_NWV = 10;

template <int TH>
__attribute__((amdgpu_waves_per_eu(TH == 16 ? 2 : 1)))      // (16-row tiles: the launch needs two workgroups per CU resident)
__global__ __launch_bounds__(256) void stem_kernel(StemArgs a)
{
    using SC = StCfg<TH>;
    constexpr int ST_TH = TH, ST_PH = SC::PH, ST_PS = SC::PS, ST_NPV = SC::NPV, MT = 4, HV = SC::HALVES;
    __shared__ __attribute__((aligned(16))) float planes[7 * ST_PS];
    __shared__ __attribute__((aligned(16))) float wl[ST_KTOT * ST_WLP];
    __shared__ int tab[ST_KTOT];
    CT_STAMP_RT(0);
    CT_STAMP(1);
    CT_STAMP_HW(8);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    int bid = blockIdx.x;
    const int tx = bid % a.tilesX; bid /= a.tilesX;
    const int ty = bid % a.tilesY; bid /= a.tilesY;
    const int n = bid;
    const int oy0 = ty * ST_TH, ox0 = tx * ST_TW;
    const size_t HW = (size_t)a.H * a.W;

    // k -> patch offset table of all three stems (pure index arithmetic)
    for (int k = tid; k < ST_KTOT; k += 256) {
        const int s = (k >= 296) ? 2 : ((k >= 148) ? 1 : 0);
        const int kk = k - st_koff(s);
        const int cin = (s == 2) ? 1 : 3;
        int off = 0;
        if (kk < cin * 49) {
            const int c = kk / 49, r = kk - c * 49;
            off = (3 * s + c) * ST_PS + (r / 7) * ST_PW + (r % 7);
        }
        tab[k] = off;
    }

    // Staging of stem s (its input planes of the tile with a 3 px zero-padded halo, and its weights
    // transposed to [k][16]) is split into "all global loads into registers" and "LDS stores", so that
    // the loads of stem s+1 are in flight while the MFMAs of stem s run.
    float pv[ST_NPV], wv[ST_NWV];
    auto stage_load = [&](int s) {
        const int cin = (s == 2) ? 1 : 3;
        const int K = cin * 49;
        const float *src = a.in[s] + (size_t)n * cin * HW;
        const float *w = a.w[s];
        const int np = cin * ST_PH * 38, nw = st_k4(s) * 16;
#pragma unroll
        for (int i = 0; i < ST_NPV; ++i) {
            const int it = tid + 256 * i;
            const int c = it / (ST_PH * 38);
            const int r = it - c * (ST_PH * 38);
            const int py = r / 38, px = r - py * 38;
            const int iy = oy0 - 3 + py, ix = ox0 - 3 + px;
            const bool ok = it < np && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
            const float v = src[ok ? ((size_t)c * HW + (size_t)iy * a.W + ix) : 0];
            pv[i] = ok ? v : 0.0f;
        }
#pragma unroll
        for (int i = 0; i < ST_NWV; ++i) {
            const int it = tid + 256 * i;
            const int k = it >> 4, j = it & 15;
            const bool ok = it < nw && k < K;
            const float v = w[ok ? (j * K + k) : 0];
            wv[i] = ok ? v : 0.0f;
        }
    };
    auto stage_store = [&](int s) {
        const int cin = (s == 2) ? 1 : 3;
        const int np = cin * ST_PH * 38, nw = st_k4(s) * 16;
#pragma unroll
        for (int i = 0; i < ST_NPV; ++i) {
            const int it = tid + 256 * i;
            if (it < np) {
                const int c = it / (ST_PH * 38);
                const int r = it - c * (ST_PH * 38);
                const int py = r / 38, px = r - py * 38;
                planes[(3 * s + c) * ST_PS + py * ST_PW + px] = pv[i];
            }
        }
#pragma unroll
        for (int i = 0; i < ST_NWV; ++i) {
            const int it = tid + 256 * i;
            if (it < nw) wl[st_koff(s) * 16 + it] = wv[i];
        }
    };

    // slab h (8 rows): wave -> rows 8h + 2w, 8h + 2w + 1; m-tile mt -> (row 8h + 2w + mt/2, column block mt&1)
    int pbase[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) pbase[mt] = (2 * wave + (mt >> 1)) * ST_PW + (mt & 1) * 16 + li;

    // The three terms are summed in the order 0, 1, 2 starting from `add` (or 0): a launch of stems {0, 1} into a
    // partial map followed by a launch of stem {2} on top of it gives the bits of ONE launch of all three
    // ((0 + r0) + r1) + r2 -- the detector computes the first two terms of frame t+1, which do not depend on the
    // tracker, while the host still associates frame t (round 3).
    f32x4 out[HV][MT];
#pragma unroll
    for (int h = 0; h < HV; ++h)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        out[h][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (a.add) {
            const int oy = oy0 + 8 * h + 2 * wave + (mt >> 1);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ox = ox0 + (mt & 1) * 16 + lg * 4 + e;
                if (oy < a.H && ox < a.W) out[h][mt][e] = a.add[(((size_t)n * a.H + oy) * a.W + ox) * a.ldadd + li];
            }
        }
    }

    // stems of this launch, in order (uniform): first, and for every stem the next active one (3 = none)
    const int first = a.in[0] ? 0 : (a.in[1] ? 1 : 2);
    stage_load(first);
    CT_STAMP(2);
    stage_store(first);
    __syncthreads();
    CT_STAMP(3);
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        if (s < first) continue;
        const int nxt = (s < 1 && a.in[1]) ? 1 : ((s < 2 && a.in[2]) ? 2 : 3);
        const bool have_next = a.in[s] != nullptr && nxt < 3;
        if (have_next) stage_load(nxt);
        if (a.in[s]) {
            const float sc = a.scale[s * 16 + li], sh = a.shift[s * 16 + li];
#pragma unroll
            for (int h = 0; h < HV; ++h) {
                f32x4 acc[MT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
                const int nsteps = st_k4(s) / 4;
#pragma unroll 4
                for (int st = 0; st < nsteps; ++st) {
                    const int k = st_koff(s) + 4 * st + lg;
                    const int off = tab[k] + 8 * h * ST_PW;
                    const float b = wl[k * ST_WLP + li];
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
                        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(planes[off + pbase[mt]], b, acc[mt], 0, 0, 0);
                }
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int e = 0; e < 4; ++e) out[h][mt][e] += fmaxf(acc[mt][e] * sc + sh, 0.0f);
            }
        }
        if (have_next) stage_store(nxt);
        if (s < 2) __syncthreads();
        if (s == 0) CT_STAMP(4);
        if (s == 1) CT_STAMP(5);
    }
    CT_STAMP(9);

#pragma unroll
    for (int h = 0; h < HV; ++h)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int oy = oy0 + 8 * h + 2 * wave + (mt >> 1);
        if (oy >= a.H) continue;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int ox = ox0 + (mt & 1) * 16 + lg * 4 + e;
            if (ox < a.W) a.y[(((size_t)n * a.H + oy) * a.W + ox) * a.ldy + li] = out[h][mt][e];
        }
    }
    CT_STAMP(6);
    CT_STAMP_RT(7);
}

}  // namespace

extern "C" int ct_stem_forward(const float *x, const float *pre_img, const float *pre_hm, int N, int H, int W,
                               const float *w_x, const float *w_img, const float *w_hm, const float *scale3,
                               const float *shift3, float *y, int ldy, void *stream)
{
    if (!x) CT_FAIL_ARG("ct_stem_forward: null pointer");
    return ct_stem_forward_parts(x, pre_img, pre_hm, nullptr, 0, N, H, W, w_x, w_img, w_hm, scale3, shift3, y, ldy, stream);
}

extern "C" int ct_stem_forward_parts(const float *x, const float *pre_img, const float *pre_hm, const float *add, int ldadd,
                                     int N, int H, int W, const float *w_x, const float *w_img, const float *w_hm,
                                     const float *scale3, const float *shift3, float *y, int ldy, void *stream)
{
    if (!scale3 || !shift3 || !y) CT_FAIL_ARG("ct_stem_forward: null pointer");
    if (!x && !pre_img && !pre_hm) CT_FAIL_ARG("ct_stem_forward_parts: no stem selected");
    if ((x && !w_x) || (pre_img && !w_img) || (pre_hm && !w_hm)) CT_FAIL_ARG("ct_stem_forward: input given without its weights");
    if (N <= 0 || H <= 0 || W <= 0 || ldy < 16 || (add && ldadd < 16)) CT_FAIL_ARG("ct_stem_forward: bad shape");
    StemArgs a;
    a.in[0] = x; a.in[1] = pre_img; a.in[2] = pre_hm;
    a.w[0] = w_x; a.w[1] = w_img; a.w[2] = w_hm;
    a.scale = scale3; a.shift = shift3; a.y = y; a.add = add; a.ldadd = ldadd;
    a.N = N; a.H = H; a.W = W; a.ldy = ldy;
    a.tilesX = ct_cdiv(W, ST_TW);
    // 16-row tiles where the 8-row grid is between one and two rounds of the chip (768 resident workgroups of 140 VGPRs):
    // its second round would run one wave per SIMD
    const long blocks8 = (long)N * a.tilesX * ct_cdiv(H, 8);
    const int want = ct_tune_get(CT_TUNE_STEM_ROWS);
    const bool rows16 = want == 16 || (want == 0 && blocks8 > 768 && blocks8 <= 1536);
    a.tilesY = ct_cdiv(H, rows16 ? 16 : 8);
    const long blocks = (long)N * a.tilesX * a.tilesY;
    if (blocks > 0x7fffffffL) CT_FAIL_ARG("ct_stem_forward: grid too large");
    if (rows16) hipLaunchKernelGGL(stem_kernel<16>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(stem_kernel<8>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    CT_CHECK_LAUNCH("ct_stem_forward");
    return CT_OK;
}
