// Heat-map decode: 3x3 max-pool pseudo-NMS + exact top-K + all head gathers + box
// assembly in two launches, one packed [B,K,F] output (one D2H copy instead of 8-14).
//
// The reference does per-class top-K over h*w then top-K over C*K (utils.py:71-87); the
// result is the global top-K of all (class, pixel) scores, which is what is computed here:
//   stage 1  one workgroup per (image, class, 4096-pixel segment): rows of the segment
//            (+1 halo row each side) -> LDS, NMS from LDS, order-preserving 64-bit keys
//            (score bits : ~pixel index) -> exact radix select of the segment's top-K.
//   stage 2  one workgroup per image: radix select of the top-K over all candidates with
//            keys (score bits : ~(class*h*w + pixel)), bitonic sort of the K winners,
//            gathers + box arithmetic, packed row store.
// Keys are distinct, so the order is fully deterministic: score descending, then lower
// class, then lower pixel index (torch.topk leaves exact ties unspecified).  Byte/index
// work: HBM/L2-bound, no matrix cores.
#include "ct_common.h"

namespace {

constexpr int SEG = 4096;          // pixels per stage-1 workgroup
constexpr int MAXK = 512;
constexpr size_t S1_BCAST = SEG * sizeof(unsigned long long);
constexpr size_t S1_HIST = S1_BCAST + 16;
constexpr size_t S1_SLOT = S1_HIST + 1024;
constexpr size_t S1_ROWS = S1_SLOT + 16;

__device__ __forceinline__ unsigned f2ord(float f)
{
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned o)
{
    return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}

// Exact K-th largest of n distinct 64-bit keys read through `get(i)`; returns the
// threshold T such that exactly min(K, n) keys are >= T.  All threads of the block call it.
template <typename Get>
__device__ unsigned long long radix_select_kth(Get get, int n, int K, unsigned *hist /*[256] LDS*/,
                                               unsigned long long *bcast /*[2] LDS*/)
{
    if (K >= n) return 0ull;
    unsigned long long prefix = 0ull;
    int need = K;
    for (int d = 7; d >= 0; --d) {
        for (int i = threadIdx.x; i < 256; i += blockDim.x) hist[i] = 0u;
        __syncthreads();
        const int sh = d * 8;
        const unsigned long long himask = (d == 7) ? 0ull : (~0ull << (sh + 8));
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            const unsigned long long k = get(i);
            if ((k & himask) == prefix) atomicAdd(&hist[(unsigned)(k >> sh) & 255u], 1u);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            int above = 0, b = 255;
            for (; b > 0; --b) {
                const int c = (int)hist[b];
                if (above + c >= need) break;
                above += c;
            }
            bcast[0] = (unsigned long long)b;
            bcast[1] = (unsigned long long)(need - above);
        }
        __syncthreads();
        prefix |= bcast[0] << sh;
        need = (int)bcast[1];
        __syncthreads();
    }
    return prefix;
}

struct Stage1Args {
    const float *hm;
    unsigned long long *cand;   // [B*C*nseg][K]
    int B, C, h, w, K, nseg;
};

__global__ __launch_bounds__(256) void decode_stage1_kernel(Stage1Args a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // all LDS in the dynamic region so every carve offset is a multiple of 16 bytes
    unsigned long long *keys = reinterpret_cast<unsigned long long *>(smem);                    // [SEG]
    unsigned long long *bcast = reinterpret_cast<unsigned long long *>(smem + S1_BCAST);        // [2]
    unsigned *hist = reinterpret_cast<unsigned *>(smem + S1_HIST);                              // [256]
    int &slot = *reinterpret_cast<int *>(smem + S1_SLOT);
    float *rows = reinterpret_cast<float *>(smem + S1_ROWS);                                    // [nrows*w]

    int bid = blockIdx.x;
    const int seg = bid % a.nseg; bid /= a.nseg;
    const int c = bid % a.C;
    const int b = bid / a.C;
    const int HW = a.h * a.w;
    const int p_lo = seg * SEG;
    const int p_hi = min(HW, p_lo + SEG);
    const int nloc = p_hi - p_lo;
    const float *map = a.hm + ((size_t)b * a.C + c) * HW;

    const int y_lo = p_lo / a.w, y_hi = (p_hi - 1) / a.w;
    const int r_lo = max(0, y_lo - 1), r_hi = min(a.h - 1, y_hi + 1);
    const int nstage = (r_hi - r_lo + 1) * a.w;
    for (int i = threadIdx.x; i < nstage; i += 256) rows[i] = map[r_lo * a.w + i];
    if (threadIdx.x == 0) slot = 0;
    __syncthreads();

    for (int i = threadIdx.x; i < nloc; i += 256) {
        const int p = p_lo + i;
        const int y = p / a.w, x = p - y * a.w;
        const float v = rows[(y - r_lo) * a.w + x];
        float m = v;
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy) {
            const int yy = y + dy;
            if (yy < 0 || yy >= a.h) continue;
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) {
                const int xx = x + dx;
                if (xx < 0 || xx >= a.w) continue;
                m = fmaxf(m, rows[(yy - r_lo) * a.w + xx]);
            }
        }
        const float s = (m == v) ? v : v * 0.0f;      // heat * keep   (utils.py:57-58)
        keys[i] = ((unsigned long long)f2ord(s) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)p);
    }
    __syncthreads();

    const unsigned long long T = radix_select_kth([&](int i) { return keys[i]; }, nloc, a.K, hist, bcast);
    unsigned long long *out = a.cand + (size_t)blockIdx.x * a.K;
    for (int i = threadIdx.x; i < nloc; i += 256) {
        const unsigned long long k = keys[i];
        if (k >= T) {
            const int s = atomicAdd(&slot, 1);
            if (s < a.K) out[s] = k;
        }
    }
    __syncthreads();
    for (int i = slot + threadIdx.x; i < a.K; i += 256) out[i] = 0ull;   // pad (only when nloc < K)
}

struct Stage2Args {
    const unsigned long long *cand;
    const float *hm_unused;
    const float *heads[CT_NUM_HEADS];
    int head_ch[CT_NUM_HEADS];
    float *out;
    long long *inds;
    int B, C, h, w, K, nseg, F;
};

__global__ __launch_bounds__(1024) void decode_stage2_kernel(Stage2Args a)
{
    __shared__ unsigned hist[256];
    __shared__ unsigned long long bcast[2];
    __shared__ unsigned long long win[MAXK];
    __shared__ int slot;
    const int b = blockIdx.x;
    const int HW = a.h * a.w;
    const int M2 = a.C * a.nseg * a.K;
    const unsigned long long *cand = a.cand + (size_t)b * M2;
    const int per_class = a.nseg * a.K;

    // key2 = score bits : ~(class*HW + pixel); pads (key 0) stay 0
    auto get = [&](int i) -> unsigned long long {
        const unsigned long long k = cand[i];
        if (k == 0ull) return 0ull;
        const unsigned cls = (unsigned)(i / per_class);
        const unsigned p = 0xFFFFFFFFu - (unsigned)(k & 0xFFFFFFFFull);
        return (k & 0xFFFFFFFF00000000ull) | (unsigned long long)(0xFFFFFFFFu - (cls * (unsigned)HW + p));
    };
    int KP = 1;
    while (KP < a.K) KP <<= 1;
    for (int i = threadIdx.x; i < KP; i += blockDim.x) win[i] = 0ull;
    if (threadIdx.x == 0) slot = 0;
    __syncthreads();
    const unsigned long long T = radix_select_kth(get, M2, a.K, hist, bcast);
    for (int i = threadIdx.x; i < M2; i += blockDim.x) {
        const unsigned long long k = get(i);
        if (k >= T && k != 0ull) {
            const int s = atomicAdd(&slot, 1);
            if (s < a.K) win[s] = k;
        }
    }
    __syncthreads();
    // bitonic sort, descending
    for (int size = 2; size <= KP; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int i = threadIdx.x; i < KP / 2; i += blockDim.x) {
                const int lo = 2 * i - (i & (stride - 1));
                const int hi = lo + stride;
                const bool desc = ((lo & size) == 0);
                const unsigned long long x = win[lo], y = win[hi];
                if ((x < y) == desc) { win[lo] = y; win[hi] = x; }
            }
            __syncthreads();
        }
    }
    for (int r = threadIdx.x; r < a.K; r += blockDim.x) {
        const unsigned long long k = win[r];
        const float score = ord2f((unsigned)(k >> 32));
        const unsigned flat = 0xFFFFFFFFu - (unsigned)(k & 0xFFFFFFFFull);
        const int cls = (int)(flat / (unsigned)HW);
        const int p = (int)(flat - (unsigned)cls * (unsigned)HW);
        const float ys0 = (float)(p / a.w), xs0 = (float)(p % a.w);
        float *row = a.out + ((size_t)b * a.K + r) * a.F;
        if (a.inds) a.inds[(size_t)b * a.K + r] = p;
        auto hv = [&](int hd, int ch) { return a.heads[hd][((size_t)b * a.head_ch[hd] + ch) * HW + p]; };
        int f = 0;
        row[f++] = score; row[f++] = (float)cls; row[f++] = xs0; row[f++] = ys0;
        float xs = xs0 + 0.5f, ys = ys0 + 0.5f;                       // decode.py:102-110
        if (a.heads[CT_HEAD_REG]) { xs = xs0 + hv(CT_HEAD_REG, 0); ys = ys0 + hv(CT_HEAD_REG, 1); }
        const bool has_box = a.heads[CT_HEAD_WH] || a.heads[CT_HEAD_LTRB] || a.heads[CT_HEAD_LTRB_AMODAL];
        float bb[4] = {0.f, 0.f, 0.f, 0.f};
        if (a.heads[CT_HEAD_WH]) {                                    // decode.py:112-128
            float ww = hv(CT_HEAD_WH, 0), hh = hv(CT_HEAD_WH, 1);
            if (ww < 0.f) ww = 0.f;
            if (hh < 0.f) hh = 0.f;
            bb[0] = xs - ww / 2; bb[1] = ys - hh / 2; bb[2] = xs + ww / 2; bb[3] = ys + hh / 2;
        }
        if (a.heads[CT_HEAD_LTRB]) {                                  // decode.py:131-139
            bb[0] = xs0 + hv(CT_HEAD_LTRB, 0); bb[1] = ys0 + hv(CT_HEAD_LTRB, 1);
            bb[2] = xs0 + hv(CT_HEAD_LTRB, 2); bb[3] = ys0 + hv(CT_HEAD_LTRB, 3);
        }
        float am[4] = {0.f, 0.f, 0.f, 0.f};
        if (a.heads[CT_HEAD_LTRB_AMODAL]) {                           // decode.py:150-159
            am[0] = xs0 + hv(CT_HEAD_LTRB_AMODAL, 0); am[1] = ys0 + hv(CT_HEAD_LTRB_AMODAL, 1);
            am[2] = xs0 + hv(CT_HEAD_LTRB_AMODAL, 2); am[3] = ys0 + hv(CT_HEAD_LTRB_AMODAL, 3);
            for (int i = 0; i < 4; ++i) bb[i] = am[i];
        }
        if (has_box) for (int i = 0; i < 4; ++i) row[f++] = bb[i];
        if (a.heads[CT_HEAD_LTRB_AMODAL]) for (int i = 0; i < 4; ++i) row[f++] = am[i];
        const int rest[7] = {CT_HEAD_TRACKING, CT_HEAD_DEP, CT_HEAD_ROT, CT_HEAD_DIM, CT_HEAD_AMODEL_OFFSET,
                             CT_HEAD_NUSCENES_ATT, CT_HEAD_VELOCITY};
        for (int q = 0; q < 7; ++q) {
            const int hd = rest[q];
            if (!a.heads[hd]) continue;
            for (int ch = 0; ch < a.head_ch[hd]; ++ch) row[f++] = hv(hd, ch);
        }
    }
}

const int kHeadCh[CT_NUM_HEADS] = {2, 2, 2, 4, 4, 1, 8, 3, 2, 8, 3};

int check(const ct_decode_desc *d, const char *who)
{
    if (!d || !d->hm) CT_FAIL_ARG("%s: null heat-map", who);
    if (d->B <= 0 || d->C <= 0 || d->h <= 0 || d->w <= 0) CT_FAIL_ARG("%s: bad shape", who);
    if (d->K <= 0 || d->K > MAXK) CT_FAIL_ARG("%s: K=%d out of range (1..%d)", who, d->K, MAXK);
    if ((long)d->h * d->w < d->K) CT_FAIL_ARG("%s: h*w=%ld < K=%d (torch.topk would raise too)", who, (long)d->h * d->w, d->K);
    if ((double)d->C * d->h * d->w >= 4294967295.0) CT_FAIL_ARG("%s: C*h*w too large", who);
    return CT_OK;
}

}  // namespace

extern "C" int ct_decode_row_floats(const ct_decode_desc *d)
{
    int f = 4;
    if (d->heads[CT_HEAD_WH] || d->heads[CT_HEAD_LTRB] || d->heads[CT_HEAD_LTRB_AMODAL]) f += 4;
    if (d->heads[CT_HEAD_LTRB_AMODAL]) f += 4;
    const int rest[7] = {CT_HEAD_TRACKING, CT_HEAD_DEP, CT_HEAD_ROT, CT_HEAD_DIM, CT_HEAD_AMODEL_OFFSET,
                         CT_HEAD_NUSCENES_ATT, CT_HEAD_VELOCITY};
    for (int q = 0; q < 7; ++q)
        if (d->heads[rest[q]]) f += kHeadCh[rest[q]];
    return f;
}

extern "C" size_t ct_decode_workspace_bytes(const ct_decode_desc *d)
{
    if (check(d, "ct_decode_workspace_bytes") != CT_OK) return 0;
    const int nseg = ct_cdiv(d->h * d->w, SEG);
    return (size_t)d->B * d->C * nseg * d->K * sizeof(unsigned long long);
}

extern "C" int ct_decode(const ct_decode_desc *d, void *stream)
{
    int rc = check(d, "ct_decode");
    if (rc != CT_OK) return rc;
    if (!d->out) CT_FAIL_ARG("ct_decode: null output");
    const int nseg = ct_cdiv(d->h * d->w, SEG);
    const size_t need = (size_t)d->B * d->C * nseg * d->K * sizeof(unsigned long long);
    if (!d->workspace || d->workspace_bytes < need) {
        ct_set_error("ct_decode: needs %zu workspace bytes, got %zu", need, d->workspace_bytes);
        return CT_ERR_WORKSPACE;
    }
    if (d->w > 2048) CT_FAIL_ARG("ct_decode: w=%d > 2048 unsupported", d->w);
    hipStream_t s = (hipStream_t)stream;
    Stage1Args a1;
    a1.hm = d->hm; a1.cand = (unsigned long long *)d->workspace;
    a1.B = d->B; a1.C = d->C; a1.h = d->h; a1.w = d->w; a1.K = d->K; a1.nseg = nseg;
    const int max_rows = ct_cdiv(SEG, d->w) + 3;
    const size_t lds1 = S1_ROWS + (size_t)max_rows * d->w * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(decode_stage1_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        attr_set = true;
    }
    if (lds1 > 96 * 1024) CT_FAIL_ARG("ct_decode: stage-1 LDS %zu too large", lds1);
    hipLaunchKernelGGL(decode_stage1_kernel, dim3((unsigned)(d->B * d->C * nseg)), dim3(256), lds1, s, a1);
    CT_CHECK_LAUNCH("ct_decode(stage 1)");
    Stage2Args a2;
    a2.cand = a1.cand; a2.hm_unused = nullptr;
    for (int i = 0; i < CT_NUM_HEADS; ++i) { a2.heads[i] = d->heads[i]; a2.head_ch[i] = kHeadCh[i]; }
    a2.out = d->out; a2.inds = (long long *)d->inds;
    a2.B = d->B; a2.C = d->C; a2.h = d->h; a2.w = d->w; a2.K = d->K; a2.nseg = nseg;
    a2.F = ct_decode_row_floats(d);
    hipLaunchKernelGGL(decode_stage2_kernel, dim3((unsigned)d->B), dim3(1024), 0, s, a2);
    CT_CHECK_LAUNCH("ct_decode(stage 2)");
    return CT_OK;
}
