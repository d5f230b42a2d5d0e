// Heat-map decode: 3x3 max-pool pseudo-NMS + exact top-K + all head gathers + box
// assembly in two launches, one packed [B,K,F] output (one D2H copy instead of 8-14).
//
// The reference does per-class top-K over h*w then top-K over C*K (utils.py:71-87); the
// result is the global top-K of all (class, pixel) scores, which is what is computed here.
// Keys are order-preserving 64-bit integers (score bits : ~flat index), all distinct, so
// the order is fully deterministic: score descending, then lower class, then lower pixel
// index (torch.topk leaves exact ties unspecified).
//   stage 1  one workgroup per (image, class, segment of <= 1024 pixels): rows of the
//            segment (+1 halo row each side) -> LDS, NMS from LDS, the strictly positive
//            survivors (~10 % of the pixels) are compacted with wave ballots into an LDS
//            list; if more than K survive, rank counting keeps the segment's top K.
//   stage 2a (round 3; only when an image is expected to hold more candidates than stage 2 sorts in LDS) several
//            workgroups per image: each sorts one slice of <= SLICE slots in LDS (bitonic) and keeps its K
//            best -- the union of the slices' top K contains the image's top K -- so that stage 2 reads
//            G x K keys instead of C x nseg x K with ONE workgroup (COCO x 4 streams: 128 000 slots per
//            image on 4 workgroups = 92.9 us; r02_g_kstats_coco_512_b4.txt).
//   stage 2  one workgroup per image: all non-empty candidates are compacted into LDS and sorted (bitonic,
//            <= FCAP keys: one scan + ~66 compare-exchange rounds instead of two scans and two rank-counting
//            loops); the K best are gathered: all head values + box arithmetic, packed row store.  More than
//            FCAP candidates: threshold from the per-thread maxima + rank counting, or the exact radix select
//            (wave-aggregated LDS histogram atomics) for adversarial layouts.  If an image has fewer than K
//            strictly positive survivors (near-empty maps) the K winners are instead selected
//            over ALL pixels with the NMS recomputed from HBM (slow, exact, rare).
// Byte/index work: HBM/L2-bound, no matrix cores.
#include "ct_common.h"

namespace {
CT_DEFINE_STAMPS(decode)    // (tools/decode_phases.py; expands to nothing in the shipped build)

constexpr int SEG_MAX = 1024;      // pixels per stage-1 workgroup (<=)
constexpr int MAXK = 512;

__device__ __forceinline__ unsigned f2ord(float f)
{
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned o)
{
    return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}

// histogram increment with same-digit lanes of a wave combined into one LDS atomic (scores of a
// heat map share their exponent bits, so plain per-lane atomics serialise on a few bins)
__device__ __forceinline__ void hist_add(unsigned *hist, unsigned digit, bool active)
{
    unsigned long long todo = __ballot(active);
    const int lane = threadIdx.x & 63;
#pragma unroll 1
    for (int it = 0; it < 4 && todo; ++it) {
        const int leader = __ffsll((long long)todo) - 1;
        const unsigned d0 = __shfl(digit, leader);
        const unsigned long long same = __ballot(active && digit == d0) & todo;
        if (lane == leader) atomicAdd(&hist[d0], (unsigned)__popcll(same));
        todo &= ~same;
    }
    if (active && ((todo >> lane) & 1ull)) atomicAdd(&hist[digit], 1u);
}

// Exact K-th largest of the n keys read through `get(i)` (key 0 = "no candidate", skipped;
// all other keys distinct); returns T such that exactly K keys are >= T.  The caller guarantees
// that at least K non-zero keys exist.  All threads of the block call it.
template <typename Get>
__device__ unsigned long long radix_select_kth(Get get, int n, int K, unsigned *hist /*[256] LDS*/,
                                               unsigned long long *bcast /*[2] LDS*/)
{
    __shared__ int scan_tmp[256];
    __shared__ int wsum[4];
    unsigned long long prefix = 0ull;
    int need = K;
    for (int d = 7; d >= 0; --d) {
        for (int i = threadIdx.x; i < 256; i += blockDim.x) hist[i] = 0u;
        __syncthreads();
        const int sh = d * 8;
        const unsigned long long himask = (d == 7) ? 0ull : (~0ull << (sh + 8));
        const int nround = (n + blockDim.x - 1) / blockDim.x * blockDim.x;
        for (int i = threadIdx.x; i < nround; i += blockDim.x) {
            const unsigned long long k = (i < n) ? get(i) : 0ull;
            hist_add(hist, (unsigned)(k >> sh) & 255u, k != 0ull && (k & himask) == prefix);
        }
        __syncthreads();
        // the bin holding the need-th largest key: parallel suffix sums over the 256 bins (4 waves)
        if (threadIdx.x < 256) {
            const int b = threadIdx.x, ln = b & 63;
            const int c = (int)hist[b];
            int suf = c;                                   // inclusive suffix sum inside the wave
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int t = __shfl_down(suf, o);
                if (ln + o < 64) suf += t;
            }
            if (ln == 0) wsum[b >> 6] = suf;               // total of this wave's 64 bins
            scan_tmp[b] = suf;
        }
        __syncthreads();
        if (threadIdx.x < 256) {
            const int b = threadIdx.x;
            int hi = 0;
            for (int wv = (b >> 6) + 1; wv < 4; ++wv) hi += wsum[wv];
            const int incl = scan_tmp[b] + hi;             // keys in bins >= b
            const int above = incl - (int)hist[b];         // keys in bins >  b
            if (above < need && incl >= need) {
                bcast[0] = (unsigned long long)b;
                // every key of the chosen bin is wanted -> the prefix alone is the threshold
                bcast[1] = (incl - above == need - above) ? ~0ull : (unsigned long long)(need - above);
            }
        }
        __syncthreads();
        prefix |= bcast[0] << sh;
        const unsigned long long nd = bcast[1];
        __syncthreads();
        if (nd == ~0ull) break;
        need = (int)nd;
    }
    return prefix;
}

struct Stage1Args {
    const float *hm;
    unsigned long long *cand;   // [B*C*nseg][K], unused slots = 0
    int B, C, h, w, K, nseg, seg;
    size_t hm_bs;               // floats between consecutive images of hm
};

__global__ __launch_bounds__(256) void decode_stage1_kernel(Stage1Args a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned long long *keys = reinterpret_cast<unsigned long long *>(smem);                    // [SEG_MAX]
    float *rows = reinterpret_cast<float *>(smem + SEG_MAX * sizeof(unsigned long long));      // [nrows*w]
    __shared__ int npos;

    int bid = blockIdx.x;
    const int seg = bid % a.nseg; bid /= a.nseg;
    const int c = bid % a.C;
    const int b = bid / a.C;
    const int HW = a.h * a.w;
    const int p_lo = seg * a.seg;
    const int p_hi = min(HW, p_lo + a.seg);
    const int nloc = p_hi - p_lo;
    const float *map = a.hm + (size_t)b * a.hm_bs + (size_t)c * HW;

    const int y_lo = p_lo / a.w, y_hi = (p_hi - 1) / a.w;
    const int r_lo = max(0, y_lo - 1), r_hi = min(a.h - 1, y_hi + 1);
    const int nstage = (r_hi - r_lo + 1) * a.w;
    for (int i = threadIdx.x; i < nstage; i += 256) rows[i] = map[r_lo * a.w + i];
    if (threadIdx.x == 0) npos = 0;
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int nround = (nloc + 255) / 256 * 256;
    for (int i = threadIdx.x; i < nround; i += 256) {
        bool pos = false;
        unsigned long long key = 0ull;
        if (i < nloc) {
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
            pos = s > 0.0f;
            key = ((unsigned long long)f2ord(s) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)p);
        }
        const unsigned long long mask = __ballot(pos);
        int base = 0;
        if (lane == 0 && mask) base = atomicAdd(&npos, (int)__popcll(mask));
        base = __shfl(base, 0);
        if (pos) keys[base + (int)__popcll(mask & ((1ull << lane) - 1ull))] = key;
    }
    __syncthreads();

    const int n = npos;
    unsigned long long *out = a.cand + (size_t)blockIdx.x * a.K;
    if (n <= a.K) {
        for (int i = threadIdx.x; i < a.K; i += 256) out[i] = (i < n) ? keys[i] : 0ull;
    } else {
        // rank counting: keys are distinct, the rank of a key = number of larger keys
        for (int i = threadIdx.x; i < n; i += 256) {
            const unsigned long long k = keys[i];
            int r = 0;
            for (int j = 0; j < n; ++j) r += (keys[j] > k) ? 1 : 0;
            if (r < a.K) out[r] = k;
        }
    }
}

struct Stage2Args {
    const unsigned long long *cand;
    const float *hm;
    const float *heads[CT_NUM_HEADS];
    int head_ch[CT_NUM_HEADS];
    size_t head_bs[CT_NUM_HEADS];   // floats between consecutive images of each head map
    size_t hm_bs;
    float *out;
    long long *inds;
    int B, C, h, w, K, nseg, F;      // F = floats between consecutive rows
    int M2;                          // candidate slots per image
    int keys_final;                  // the slots already hold (score : ~(class*HW + pixel)) keys (written by stage 2a)
    float *host_out;                 // optional pinned host copy of the rows, flag raised when all images are done
    int *done_flag;
    unsigned *done_counter;
    // sparse heads (round 5): stage 2 stops after the selection and hands the K sorted winner keys of every image to
    // sparse_heads_kernel, which evaluates the regression heads at those pixels and assembles the rows itself
    unsigned long long *winners;     // [B][K], nullptr = dense head maps (rows assembled here)
    unsigned present;                // bit hd: head hd is part of the row (a dense map OR a sparse head)
};

// One packed row (ct_decode_desc: score, cls, xs0, ys0, box, amodal box, the remaining heads) from the winner's score /
// class / pixel and its head values hv(head, channel) -- decode.py:99-159; shared by the dense gather of stage 2 and
// the sparse heads' epilogue
template <typename HV>
__device__ __forceinline__ void emit_row(float *row, float *hrow, unsigned present, float score, int cls, float xs0,
                                         float ys0, HV hv)
{
    constexpr int HCH[CT_NUM_HEADS] = {2, 2, 2, 4, 4, 1, 8, 3, 2, 8, 3};
    auto has = [&](int hd) { return (present >> hd) & 1u; };
    int f = 0;
    // (round 4: assembling the rows in LDS and storing them coalesced was measured and dropped -- the host copy's cost
    //  is the PCIe write round trip the system fence below waits for, ~3.3 us however the stores are shaped, and the
    //  lane-per-row stores start it a microsecond earlier: 6.6 against 9.0 us from the gathers to the raised flag; a
    //  wave-local transpose without the workgroup barrier: 5.7 against 5.0 us for the stores, 18.1-19.5 against 17.1-17.8
    //  us for the kernel inside the frame)
    auto put = [&](float v) { row[f] = v; if (hrow) hrow[f] = v; ++f; };
    put(score); put((float)cls); put(xs0); put(ys0);
    float xs = xs0 + 0.5f, ys = ys0 + 0.5f;                       // decode.py:102-110
    if (has(CT_HEAD_REG)) { xs = xs0 + hv(CT_HEAD_REG, 0); ys = ys0 + hv(CT_HEAD_REG, 1); }
    const bool has_box = has(CT_HEAD_WH) || has(CT_HEAD_LTRB) || has(CT_HEAD_LTRB_AMODAL);
    float bb[4] = {0.f, 0.f, 0.f, 0.f};
    if (has(CT_HEAD_WH)) {                                        // decode.py:112-128
        float ww = hv(CT_HEAD_WH, 0), hh = hv(CT_HEAD_WH, 1);
        if (ww < 0.f) ww = 0.f;
        if (hh < 0.f) hh = 0.f;
        bb[0] = xs - ww / 2; bb[1] = ys - hh / 2; bb[2] = xs + ww / 2; bb[3] = ys + hh / 2;
    }
    if (has(CT_HEAD_LTRB)) {                                      // decode.py:131-139
        bb[0] = xs0 + hv(CT_HEAD_LTRB, 0); bb[1] = ys0 + hv(CT_HEAD_LTRB, 1);
        bb[2] = xs0 + hv(CT_HEAD_LTRB, 2); bb[3] = ys0 + hv(CT_HEAD_LTRB, 3);
    }
    float am[4] = {0.f, 0.f, 0.f, 0.f};
    if (has(CT_HEAD_LTRB_AMODAL)) {                               // decode.py:150-159
        am[0] = xs0 + hv(CT_HEAD_LTRB_AMODAL, 0); am[1] = ys0 + hv(CT_HEAD_LTRB_AMODAL, 1);
        am[2] = xs0 + hv(CT_HEAD_LTRB_AMODAL, 2); am[3] = ys0 + hv(CT_HEAD_LTRB_AMODAL, 3);
        for (int i = 0; i < 4; ++i) bb[i] = am[i];
    }
    if (has_box) for (int i = 0; i < 4; ++i) put(bb[i]);
    if (has(CT_HEAD_LTRB_AMODAL)) for (int i = 0; i < 4; ++i) put(am[i]);
    const int rest[7] = {CT_HEAD_TRACKING, CT_HEAD_DEP, CT_HEAD_ROT, CT_HEAD_DIM, CT_HEAD_AMODEL_OFFSET,
                         CT_HEAD_NUSCENES_ATT, CT_HEAD_VELOCITY};
#pragma unroll
    for (int q = 0; q < 7; ++q) {
        const int hd = rest[q];
        if (!has(hd)) continue;
#pragma unroll
        for (int ch = 0; ch < HCH[hd]; ++ch) put(hv(hd, ch));
    }
}

// every row of a workgroup is out (device + host copy): the last of `nwg` workgroups raises the host flag
__device__ __forceinline__ void raise_done_flag(int *done_flag, unsigned *done_counter, int nwg)
{
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0 && nwg == 1) {
        // one workgroup: its rows are all there is (fenced above) -- no arrival counter to go through
        __hip_atomic_store(done_flag, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    } else if (threadIdx.x == 0) {
        // acq_rel at system scope: the last arriver ACQUIRES the other workgroups' host_out stores (released by their
        // own increments) before it releases the flag to the host, which skips the runtime wait once it sees it
        const unsigned prev = __hip_atomic_fetch_add(done_counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_SYSTEM);
        if (prev == (unsigned)nwg - 1u) {
            *done_counter = 0u;
            __hip_atomic_store(done_flag, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// NMS'd score key of flat index f = cls*HW + p, straight from HBM (slow path only)
__device__ __forceinline__ unsigned long long key_from_map(const float *hm_b, int C, int h, int w, unsigned f)
{
    const int HW = h * w;
    const int cls = (int)(f / (unsigned)HW);
    const int p = (int)(f - (unsigned)cls * (unsigned)HW);
    const int y = p / w, x = p - y * w;
    const float *map = hm_b + (size_t)cls * HW;
    const float v = map[p];
    float m = v;
    for (int dy = -1; dy <= 1; ++dy) {
        const int yy = y + dy;
        if (yy < 0 || yy >= h) continue;
        for (int dx = -1; dx <= 1; ++dx) {
            const int xx = x + dx;
            if (xx < 0 || xx >= w) continue;
            m = fmaxf(m, map[yy * w + xx]);
        }
    }
    const float s = (m == v) ? v : v * 0.0f;
    return ((unsigned long long)f2ord(s) << 32) | (unsigned long long)(0xFFFFFFFFu - f);
}

// Visit every candidate slot with the loads of 8 slots per thread in flight at once (a plain
// one-load-per-iteration loop is latency-bound: one L2 round trip per slot and thread).  f(i, raw)
// is called by all threads the same number of times (slots past the end come as raw = 0).
template <typename F>
__device__ __forceinline__ void scan_cands(const unsigned long long *cand, int M2, int tid, int NT, F f)
{
    constexpr int U = 8;
    for (int i0 = tid; i0 - tid < M2; i0 += NT * U) {
        unsigned long long raw[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + u * NT;
            raw[u] = (i < M2) ? cand[i] : 0ull;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) f(i0 + u * NT, raw[u]);
    }
}

constexpr int FCAP = 4096;        // capacity of the stage-2 candidate list held (and sorted) in LDS
constexpr int SLICE = 8192;       // slots per stage-2a workgroup (8 per thread, kept in registers)

// descending bitonic sort of n (a power of two) 64-bit keys in LDS by all threads of the workgroup
__device__ __forceinline__ void bitonic_desc(unsigned long long *v, int n, int tid, int NT)
{
    for (int size = 2; size <= n; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int i = tid; i < (n >> 1); i += NT) {
                const int lo = 2 * i - (i & (stride - 1));
                const int hi = lo + stride;
                const bool desc = ((lo & size) == 0);
                const unsigned long long x = v[lo], y = v[hi];
                if ((x < y) == desc) { v[lo] = y; v[hi] = x; }
            }
            __syncthreads();
        }
    }
}


// ---- top-K of <= 8 keys per thread (1024 threads: <= 8192 keys) by a linear score histogram ------------------
// The keys sit in REGISTERS (one global round trip, no second scan, no sort of the whole set): 1024 bins over the
// score range (0, 1) -- heat maps are post-sigmoid -- are filled with LDS atomics, one suffix scan finds the bin b*
// that holds the K-th largest key, every key in a higher bin is a winner, the keys of b* itself (a handful) are
// ranked by counting and the best `need` of them complete the set.  Exact for any input -- the bin is a monotone
// function of the key, nothing else is assumed; a boundary bin with more than TIECAP keys (scores clustered in
// < 1/1024 of the range) reports overflow and the caller falls back to the general path.
// Round 4: plain LDS atomics (one per live key) instead of the wave-aggregated ones of the radix select -- the scores
// of one wave's keys rarely share a bin, the aggregation loop cost more than the conflicts it removed -- and the
// winners / boundary keys compacted with ONE atomic per wave and list (wave prefix sums of the per-lane counts)
// instead of one per wave, list and key slot.  Logarithmic bins (64 per octave) were measured too: finer where the
// low-score survivors crowd, 8x coarser in (0.5, 1) where the winners of a confident map sit -- 23.3 against 17.2 us
// per frame on the benchmark's maps; linear stays.
constexpr int SEL_U = 8;          // keys per thread
constexpr int SEL_BINS = 1024;
constexpr int TIECAP = 1024;

struct SelShared {
    unsigned hist[SEL_BINS];
    int wsum[16];
    unsigned long long tie[TIECAP];
    int bstar, need, total, nwin, ntie;
};

__device__ __forceinline__ int sel_bin(unsigned long long k)
{
    const float sc = ord2f((unsigned)(k >> 32));
    const int b = (int)(sc * (float)SEL_BINS);
    return b < 0 ? 0 : (b > SEL_BINS - 1 ? SEL_BINS - 1 : b);
}

// keys[u]: 0 = empty.  On return: -1 = fewer than K keys in total (sh.total holds their number; with `keep_all` they
// are all in win[0 .. total)), -2 = boundary bin overflow (nothing written), else K: win[0 .. K) hold the K largest
// keys, unsorted.  blockDim.x must be 1024.  win must hold >= K slots.
__device__ __forceinline__ int select_topk_regs(const unsigned long long (&keys)[SEL_U], int K, SelShared &sh,
                                                unsigned long long *win, bool keep_all)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    sh.hist[tid] = 0u;
    if (tid == 0) { sh.bstar = -1; sh.need = 0; sh.total = 0; sh.nwin = 0; sh.ntie = 0; }
    __syncthreads();
    int bins[SEL_U];
#pragma unroll
    for (int u = 0; u < SEL_U; ++u) {
        bins[u] = sel_bin(keys[u]);
        if (keys[u] != 0ull) atomicAdd(&sh.hist[bins[u]], 1u);
    }
    __syncthreads();
    // suffix sums over the 1024 bins: thread t owns bin t
    const int c = (int)sh.hist[tid];
    int suf = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_down(suf, o);
        if (lane + o < 64) suf += t;
    }
    if (lane == 0) sh.wsum[wave] = suf;
    __syncthreads();
    int hi = 0;
    for (int w = wave + 1; w < 16; ++w) hi += sh.wsum[w];
    const int incl = suf + hi, above = incl - c;          // keys in bins >= t / > t
    if (tid == 0) sh.total = incl;
    if (above < K && incl >= K) { sh.bstar = tid; sh.need = K - above; }
    __syncthreads();
    const int total = sh.total;
    if (total < K) {
        if (keep_all) {
#pragma unroll
            for (int u = 0; u < SEL_U; ++u) {
                const bool hit = keys[u] != 0ull;
                const unsigned long long mask = __ballot(hit);
                int base = 0;
                if (mask) {
                    const int leader = __ffsll((long long)mask) - 1;
                    if (lane == leader) base = atomicAdd(&sh.nwin, (int)__popcll(mask));
                    base = __shfl(base, leader);
                }
                if (hit) win[base + (int)__popcll(mask & ((1ull << lane) - 1ull))] = keys[u];
            }
            __syncthreads();
        }
        return -1;
    }
    const int bstar = sh.bstar, need = sh.need;
    // winners (bins above b*) and boundary keys (bin b*) of this lane as bit masks over its key slots; their counts,
    // packed (winners : low half, boundary keys : high half -- at most 512 per wave), prefix-summed over the wave;
    // one atomic per wave and list reserves the slots
    unsigned hitm = 0u, tiem = 0u;
#pragma unroll
    for (int u = 0; u < SEL_U; ++u) {
        const bool live = keys[u] != 0ull;
        hitm |= (live && bins[u] > bstar) ? (1u << u) : 0u;
        tiem |= (live && bins[u] == bstar) ? (1u << u) : 0u;
    }
    const int packed = __popc(hitm) | (__popc(tiem) << 16);
    int pre = packed;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(pre, o);
        if (lane >= o) pre += t;
    }
    const int tot = __shfl(pre, 63);
    int bh = 0, bt = 0;
    if (lane == 63) {
        if (tot & 0xffff) bh = atomicAdd(&sh.nwin, tot & 0xffff);
        if (tot >> 16) bt = atomicAdd(&sh.ntie, tot >> 16);
    }
    int sw = __shfl(bh, 63) + ((pre - packed) & 0xffff);
    int st = __shfl(bt, 63) + ((pre - packed) >> 16);
#pragma unroll
    for (int u = 0; u < SEL_U; ++u) {
        if ((hitm >> u) & 1u) win[sw++] = keys[u];
        if ((tiem >> u) & 1u) {
            if (st < TIECAP) sh.tie[st] = keys[u];
            ++st;
        }
    }
    __syncthreads();
    const int m = sh.ntie;
    if (m > TIECAP) return -2;
    const int first = K - need;                            // = number of keys above the boundary bin
    for (int i = tid; i < m; i += 1024) {
        const unsigned long long k = sh.tie[i];
        int r = 0;
        for (int j = 0; j < m; ++j) r += (sh.tie[j] > k) ? 1 : 0;
        if (r < need) win[first + r] = k;
    }
    __syncthreads();
    return K;
}

struct Stage2aArgs {
    const unsigned long long *cand;   // [B][M2] stage-1 keys (score : ~pixel), class = slot / per_class
    unsigned long long *cand2;        // [B][G][K] final-form keys, sorted, unused slots = 0
    int M2, G, K, per_class, HW;
};

__global__ __launch_bounds__(1024) void decode_stage2a_kernel(Stage2aArgs a)
{
    __shared__ SelShared sh;
    __shared__ unsigned long long win[MAXK];
    __shared__ unsigned rhist[256];
    __shared__ unsigned long long rbcast[2];
    __shared__ int rslot;
    const int b = blockIdx.x / a.G, g = blockIdx.x - b * a.G;
    const int tid = threadIdx.x, NT = blockDim.x;
    const int lo = g * SLICE, n = min(SLICE, a.M2 - lo);
    const unsigned long long *cand = a.cand + (size_t)b * a.M2 + lo;
    unsigned long long keys[SEL_U];
#pragma unroll
    for (int u = 0; u < SEL_U; ++u) {
        const int i = tid + u * NT;
        unsigned long long k = (i < n) ? cand[i] : 0ull;
        if (k != 0ull) {
            const unsigned cls = (unsigned)((lo + i) / a.per_class);
            const unsigned p = 0xFFFFFFFFu - (unsigned)(k & 0xFFFFFFFFull);
            k = (k & 0xFFFFFFFF00000000ull) | (unsigned long long)(0xFFFFFFFFu - (cls * (unsigned)a.HW + p));
        }
        keys[u] = k;
    }
    for (int i = tid; i < a.K; i += NT) win[i] = 0ull;
    __syncthreads();
    unsigned long long *out = a.cand2 + ((size_t)b * a.G + g) * a.K;
    const int rc = select_topk_regs(keys, a.K, sh, win, true);
    if (rc != -2) {
        for (int i = tid; i < a.K; i += NT) out[i] = win[i];
        return;
    }
    // scores clustered inside one histogram bin: exact radix select over the slice (>= K keys exist)
    auto get = [&](int i) -> unsigned long long {
        unsigned long long k = cand[i];
        if (k != 0ull) {
            const unsigned cls = (unsigned)((lo + i) / a.per_class);
            const unsigned p = 0xFFFFFFFFu - (unsigned)(k & 0xFFFFFFFFull);
            k = (k & 0xFFFFFFFF00000000ull) | (unsigned long long)(0xFFFFFFFFu - (cls * (unsigned)a.HW + p));
        }
        return k;
    };
    if (tid == 0) rslot = 0;
    __syncthreads();
    const unsigned long long T = radix_select_kth(get, n, a.K, rhist, rbcast);
#pragma unroll
    for (int u = 0; u < SEL_U; ++u)
        if (keys[u] >= T && keys[u] != 0ull) {
            const int sl = atomicAdd(&rslot, 1);
            if (sl < a.K) out[sl] = keys[u];
        }
}

__global__ __launch_bounds__(1024) void decode_stage2_kernel(Stage2Args a)
{
    __shared__ unsigned hist[256];
    __shared__ unsigned long long bcast[2];
    __shared__ unsigned long long win[MAXK];
    __shared__ unsigned long long lmax[1024];
    __shared__ unsigned long long filt[FCAP];
    __shared__ SelShared sel;
    __shared__ unsigned long long Lsh;
    __shared__ int slot, total, nf;
    const int b = blockIdx.x;
    const int tid = threadIdx.x, NT = blockDim.x, lane = tid & 63;
    const int HW = a.h * a.w;
    const int M2 = a.M2;
    CT_STAMP_RT(0);
    CT_STAMP(1);
    const unsigned long long *cand = a.cand + (size_t)b * M2;
    const int per_class = a.nseg * a.K;

    // key2 = score bits : ~(class*HW + pixel); empty slots (key 0) stay 0
    auto key2 = [&](int i, unsigned long long k) -> unsigned long long {
        if (k == 0ull || a.keys_final) return k;
        const unsigned cls = (unsigned)(i / per_class);
        const unsigned p = 0xFFFFFFFFu - (unsigned)(k & 0xFFFFFFFFull);
        return (k & 0xFFFFFFFF00000000ull) | (unsigned long long)(0xFFFFFFFFu - (cls * (unsigned)HW + p));
    };
    auto get = [&](int i) -> unsigned long long { return key2(i, cand[i]); };
    const bool in_regs = M2 <= SEL_U * 1024 && NT == 1024;
    unsigned long long keys[SEL_U];
    if (in_regs) {                                         // (the candidates' round trip runs under the LDS set-up below)
#pragma unroll
        for (int u = 0; u < SEL_U; ++u) {
            const int i = tid + u * NT;
            keys[u] = (i < M2) ? cand[i] : 0ull;
        }
    }
    int KP = 1;
    while (KP < a.K) KP <<= 1;
    for (int i = tid; i < KP; i += NT) win[i] = 0ull;
    if (tid == 0) { slot = 0; total = 0; nf = 0; Lsh = 1ull; }
    __syncthreads();
    bool need_sort = true;
    bool selected = false;
    if (in_regs) {
        // ---- the common case (round 3): <= 8 candidates per thread, kept in registers; histogram select ----
#pragma unroll
        for (int u = 0; u < SEL_U; ++u) keys[u] = key2(tid + u * NT, keys[u]);
        CT_STAMP(2);
        const int rc = select_topk_regs(keys, a.K, sel, filt, false);
        CT_STAMP(3);
        if (rc == a.K) {
            // sort the K winners by counting (distinct keys): eight lanes per winner, each over an eighth of the list
            for (int i0 = 0; i0 < a.K; i0 += NT / 8) {
                const int i = i0 + (tid >> 3), part = tid & 7;
                const unsigned long long k = (i < a.K) ? filt[i] : 0ull;
                int r = 0;
                for (int j = part; j < a.K; j += 8) r += (filt[j] > k) ? 1 : 0;
                r += __shfl_xor(r, 1);
                r += __shfl_xor(r, 2);
                r += __shfl_xor(r, 4);
                if (i < a.K && part == 0) win[r] = k;
            }
            need_sort = false;
            selected = true;
        } else if (rc == -1) {
            if (tid == 0) total = sel.total;              // fewer than K candidates: the exact slow path below
            selected = true;
        }
        __syncthreads();
    }
    if (!selected) {
    // ---- pass 0: every non-empty candidate into LDS (ballot compaction), counted ------------------
    scan_cands(cand, M2, tid, NT, [&](int i, unsigned long long raw) {
        const unsigned long long k = key2(i, raw);
        const bool hit = k != 0ull;
        const unsigned long long mask = __ballot(hit);
        int base = 0;
        if (mask) {
            const int leader = __ffsll((long long)mask) - 1;
            if (lane == leader) base = atomicAdd(&total, (int)__popcll(mask));
            base = __shfl(base, leader);
        }
        if (hit) {
            const int sl = base + (int)__popcll(mask & ((1ull << lane) - 1ull));
            if (sl < FCAP) filt[sl] = k;
        }
    });
    __syncthreads();
    }
    if (selected && need_sort == false) {
        // (winners already in win[])
    } else if (!selected && total >= a.K && total <= FCAP) {
        // sort them all, the K best come out in order
        int NP = KP;
        while (NP < total) NP <<= 1;
        for (int i = total + tid; i < NP; i += NT) filt[i] = 0ull;
        __syncthreads();
        bitonic_desc(filt, NP, tid, NT);
        for (int i = tid; i < a.K; i += NT) win[i] = filt[i];
        need_sort = false;
    } else if (!selected && total >= a.K) {
        // ---- more candidates than the LDS list holds: every thread's largest candidate first ----------
        {
            unsigned long long mx = 0ull;
            scan_cands(cand, M2, tid, NT, [&](int i, unsigned long long raw) {
                const unsigned long long k = key2(i, raw);
                mx = (k > mx) ? k : mx;
            });
            lmax[tid] = mx;
        }
        __syncthreads();
        // L = K-th largest of the per-thread maxima (distinct keys): a lower bound of the K-th largest
        // candidate, so {k >= L} contains the top K and, for randomly spread scores, little more
        // (ranked among the first 256 threads' maxima only: ranking all 1024 costs more than the
        //  ~4x longer filtered list it would save)
        if (tid < 256) {
            const int G = NT < 256 ? NT : 256;
            const unsigned long long my = lmax[tid];
            int r = 0;
            for (int j = 0; j < G; ++j) r += (lmax[j] > my) ? 1 : 0;
            if (my != 0ull && r == a.K - 1) Lsh = my;       // (fewer than K non-empty threads: L stays 1)
        }
        __syncthreads();
        const unsigned long long L = Lsh;
        scan_cands(cand, M2, tid, NT, [&](int i, unsigned long long raw) {
            const unsigned long long k = key2(i, raw);
            const bool hit = k >= L && k != 0ull;
            const unsigned long long mask = __ballot(hit);
            int base = 0;
            if (mask) {
                const int leader = __ffsll((long long)mask) - 1;
                if (lane == leader) base = atomicAdd(&nf, (int)__popcll(mask));
                base = __shfl(base, leader);
            }
            if (hit) {
                const int sl = base + (int)__popcll(mask & ((1ull << lane) - 1ull));
                if (sl < FCAP) filt[sl] = k;
            }
        });
        __syncthreads();
        const int n = nf;
        if (n <= FCAP) {
            // exact ranks among the filtered keys; the K best land in win[] already sorted
            for (int i = tid; i < n; i += NT) {
                const unsigned long long k = filt[i];
                int r = 0;
                for (int j = 0; j < n; ++j) r += (filt[j] > k) ? 1 : 0;
                if (r < a.K) win[r] = k;
            }
            need_sort = false;
        } else {
            // adversarial layout (the top keys sit in few threads' stripes): plain radix select
            const unsigned long long T = radix_select_kth(get, M2, a.K, hist, bcast);
            for (int i = tid; i < M2; i += NT) {
                const unsigned long long k = get(i);
                if (k >= T && k != 0ull) {
                    const int s = atomicAdd(&slot, 1);
                    if (s < a.K) win[s] = k;
                }
            }
        }
    } else {
        // fewer than K positive survivors: exact selection over every (class, pixel) of the image
        const float *hm_b = a.hm + (size_t)b * a.hm_bs;
        const int NALL = a.C * HW;
        auto getall = [&](int i) -> unsigned long long { return key_from_map(hm_b, a.C, a.h, a.w, (unsigned)i); };
        const unsigned long long T = radix_select_kth(getall, NALL, a.K, hist, bcast);
        for (int i = tid; i < NALL; i += NT) {
            const unsigned long long k = getall(i);
            if (k >= T) {
                const int s = atomicAdd(&slot, 1);
                if (s < a.K) win[s] = k;
            }
        }
    }
    __syncthreads();
    CT_STAMP(4);
    if (need_sort) bitonic_desc(win, KP, tid, NT);          // (block-uniform)
    if (a.winners) {
        // sparse heads: the sorted winners go to sparse_heads_kernel (rows, host copy and flag are its job)
        for (int r = threadIdx.x; r < a.K; r += blockDim.x) {
            const unsigned long long k = win[r];
            a.winners[(size_t)b * a.K + r] = k;
            if (a.inds) a.inds[(size_t)b * a.K + r] = (int)((0xFFFFFFFFu - (unsigned)(k & 0xFFFFFFFFull)) % (unsigned)HW);
        }
        CT_STAMP(5);
        CT_STAMP(6);
        CT_STAMP(7);
        CT_STAMP_RT(8);
        return;
    }
    for (int r = threadIdx.x; r < a.K; r += blockDim.x) {
        const unsigned long long k = win[r];
        const float score = ord2f((unsigned)(k >> 32));
        const unsigned flat = 0xFFFFFFFFu - (unsigned)(k & 0xFFFFFFFFull);
        const int cls = (int)(flat / (unsigned)HW);
        const int p = (int)(flat - (unsigned)cls * (unsigned)HW);
        const float ys0 = (float)(p / a.w), xs0 = (float)(p % a.w);
        float *row = a.out + ((size_t)b * a.K + r) * a.F;
        float *hrow = a.host_out ? a.host_out + ((size_t)b * a.K + r) * a.F : nullptr;
        if (a.inds) a.inds[(size_t)b * a.K + r] = p;
        // every head value of this row is fetched up front, branch-free (absent heads read hm[0] and are
        // ignored): ~39 independent loads in flight instead of a chain of dependent round trips
        constexpr int HCH[CT_NUM_HEADS] = {2, 2, 2, 4, 4, 1, 8, 3, 2, 8, 3};
        float hval[CT_NUM_HEADS][8];
#pragma unroll
        for (int hd = 0; hd < CT_NUM_HEADS; ++hd) {
            const float *hp = a.heads[hd];
            const bool on = hp != nullptr;
            const float *base = on ? hp + (size_t)b * a.head_bs[hd] + p : a.hm;
#pragma unroll
            for (int ch = 0; ch < HCH[hd]; ++ch) hval[hd][ch] = base[on ? (size_t)ch * HW : 0];
        }
        emit_row(row, hrow, a.present, score, cls, xs0, ys0, [&](int hd, int ch) { return hval[hd][ch]; });
    }
    CT_STAMP(5);
    if (a.done_flag) {
        raise_done_flag(a.done_flag, a.done_counter, a.B);
        CT_STAMP(6);
    }
    CT_STAMP(7);
    CT_STAMP_RT(8);
}

// ---- sparse heads (round 5, opt-in) --------------------------------------------------------------------------
// generic_decode reads the regression heads (reg, wh, tracking, ltrb*, dep, rot, dim, amodel_offset, ...) at the K winner
// pixels ONLY (decode.py:99-180: every one of them goes through _tranpose_and_gather_feat(head, inds)); the reference
// still computes their dense maps because a framework conv has no other form.  With ct_decode_desc.sparse the maps are
// never made: after the selection this kernel evaluates conv3x3 64 -> 256 + bias + ReLU + conv1x1 256 -> c of every
// listed head at the winners -- a [16 winners x 576] x [576 x 256] MFMA product per (workgroup, head), the winner's
// 3x3 x 64 patch gathered from the NHWC feature map (zero outside the map: padding 1) -- applies the dense epilogue's
// transforms (dep: detector.py:305-307) and assembles the packed rows exactly like the dense gather (emit_row).
// One workgroup = 16 waves = 16 winners of one image x ALL sparse heads; wave w owns hidden channels 16w .. 16w+15.
// 4 heads x 100 winners = 0.12 GFLOP instead of 9.7 GFLOP of dense maps at 512 x 512.
constexpr int SP_HID = 256;          // hidden channels of a head (head_conv, opts.py:294-295)

struct SparseArgs {
    const float *feat; int ldf; size_t feat_bs;
    int nheads;
    int head[CT_NUM_HEADS];
    const float *w1[CT_NUM_HEADS], *b1[CT_NUM_HEADS], *w2[CT_NUM_HEADS], *b2[CT_NUM_HEADS];
    float depth_scale;
    int zero_tracking;
    int flip_B;                      // flip_test: image b's mirrored twin is image flip_B + b of feat (0 = off)
    int flip_mode[CT_NUM_HEADS];     // per sparse head: 0 = image b alone, 1 = (v + v') / 2, 2 = the same with even channels of v' negated
    const unsigned long long *winners;
    float *partial;                  // [B * tiles][nheads][2 passes][4][16][8] partial head outputs (one hidden quarter each)
    float *out, *host_out;
    int *done_flag; unsigned *done_counter;
    int B, h, w, K, F, tiles;        // tiles = ceil(K / 16) winner tiles per image
    unsigned present;
};

// One workgroup = 4 waves = (16 winners of one image) x (one head) x (one quarter of its 256 hidden channels): wave w owns
// hidden channels 64 * quarter + 16 * w ...  First version (one workgroup per winner tile, all heads, 16 waves): 68 us at
// one stream -- seven workgroups each pulling all 2.4 MB of head weights through one CU's L1 (rocprofv3,
// gpurun_out/r05_e); spread over tiles x heads x quarters every workgroup streams 147 KB, issued in full BEFORE the
// winners are even read.  The partial 1x1 outputs of the four quarters meet in a scratch block; sparse_rows_kernel (the
// next launch) sums them in quarter order (deterministic), applies bias / transforms and assembles the rows.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void sparse_heads_kernel(SparseArgs a)
{
    __shared__ __attribute__((aligned(16))) float A[36 * 256];            // [tap * 4 + slab][16 winners][16 ch], swizzled
    __shared__ __attribute__((aligned(16))) float Hd[16 * 68];            // this quarter's 64 hidden activations per winner
    __shared__ int wy[16], wx[16];
    constexpr int HCH[CT_NUM_HEADS] = {2, 2, 2, 4, 4, 1, 8, 3, 2, 8, 3};
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bid = blockIdx.x;
    const int quarter = bid & 3; bid >>= 2;
    const int hi = bid % a.nheads; bid /= a.nheads;
    const int tile = bid;                                   // b * tiles + t
    const int b = tile / a.tiles, t = tile - b * a.tiles;
    const int HW = a.h * a.w;
    const int hd = a.head[hi];
    const int cout = HCH[hd];
    // ---- this wave's weights: 36 (tap, slab) fragments of 1 KiB for n-tile 4 * quarter + wave, all requested now ----
    const float *wp = a.w1[hi] + ((size_t)(quarter * 4 + wave) << 8) + (lane << 2);
    f32x4 bq[36];
#pragma unroll
    for (int st = 0; st < 36; ++st) bq[st] = *reinterpret_cast<const f32x4 *>(wp + (size_t)st * (16 << 8));
    const int li = lane & 15, lg = lane >> 4;
    const float bias1 = a.b1[hi][quarter * 64 + wave * 16 + li];
    if (tid < 16) {
        const int r = t * 16 + tid;
        int y = -4, x = -4;
        if (r < a.K) {
            const unsigned long long k = a.winners[(size_t)b * a.K + r];
            const unsigned flat = 0xFFFFFFFFu - (unsigned)(k & 0xFFFFFFFFull);
            const int p = (int)(flat % (unsigned)HW);
            y = p / a.w; x = p - y * a.w;
        }
        wy[tid] = y; wx[tid] = x;
    }
    __syncthreads();
    // flip_test (detector.py:311-332): wh / dep / dim / amodel_offset are averaged with the MIRRORED image's value at the
    // mirrored pixel -- a second pass over that image's patch at (y, w - 1 - x); every other head reads image b alone
    const int npass = (a.flip_B > 0 && a.flip_mode[hi] != 0) ? 2 : 1;
    for (int pass = 0; pass < npass; ++pass) {
    // ---- the 16 winners' 3x3 x 64 patches: 16 x 9 x 16 float4 items (rows past K and taps outside the map: zero) ----
    {
        const float *fb = a.feat + (size_t)(pass ? a.flip_B + b : b) * a.feat_bs;
        f32x4 pv[9];
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            const int it = tid + 256 * j;
            const int m = it / 144, rem = it - m * 144;
            const int tap = rem >> 4, q = rem & 15;
            const int cx = pass ? a.w - 1 - wx[m] : wx[m];
            const int y = wy[m] + tap / 3 - 1, x = cx + tap % 3 - 1;
            const bool ok = y >= 0 && y < a.h && x >= 0 && x < a.w && wy[m] >= 0;
            const f32x4 v = *reinterpret_cast<const f32x4 *>(fb + (ok ? ((size_t)y * a.w + x) * a.ldf + q * 4 : 0));
            pv[j] = ok ? v : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (pass) __syncthreads();                                // (pass 0's hidden tile / patch reads are done)
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            const int it = tid + 256 * j;
            const int m = it / 144, rem = it - m * 144;
            const int tap = rem >> 4, q = rem & 15;
            *reinterpret_cast<f32x4 *>(A + (tap * 4 + (q >> 2)) * 256 + m * 16 + (((q & 3) ^ ((m >> 1) & 2)) << 2)) = pv[j];
        }
    }
    __syncthreads();
    {
        const int aoff = li * 16 + ((lg ^ ((li >> 1) & 2)) << 2);
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int st = 0; st < 36; ++st) {
            const f32x4 af = *reinterpret_cast<const f32x4 *>(A + st * 256 + aoff);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[e], bq[st][e], acc, 0, 0, 0);
        }
        // C layout: row (winner) = lg * 4 + e, column (hidden channel of this quarter) = 16 * wave + li
#pragma unroll
        for (int e = 0; e < 4; ++e) Hd[(lg * 4 + e) * 68 + wave * 16 + li] = fmaxf(acc[e] + bias1, 0.0f);
    }
    __syncthreads();
    // ---- this quarter's share of conv1x1 256 -> cout: 2 lanes per (winner, channel), 32 hidden values each ----
    {
        const int pair = tid >> 1, part = tid & 1;
        const int m = pair >> 3, c = pair & 7;
        float sum = 0.f;
        if (c < cout) {
            const float *w2 = a.w2[hi] + c * SP_HID + quarter * 64 + part * 32;
            const float *hh = Hd + m * 68 + part * 32;
#pragma unroll
            for (int j = 0; j < 32; ++j) sum += hh[j] * w2[j];
        }
        sum += __shfl_xor(sum, 1);
        if (part == 0) a.partial[(((((size_t)tile * a.nheads + hi) * 2 + pass) * 4 + quarter) * 16 + m) * 8 + c] = sum;
    }
    }
}

// The partial head outputs of sparse_heads_kernel -> packed rows: one workgroup per image, one lane per winner.  (Round 5's
// first form let the LAST workgroup of every winner tile do this behind an agent-scope arrival counter: at 32 streams the
// 1 792 release / acquire fences -- an L2 write-back and invalidate each -- evicted the head weights under every other
// workgroup: 536 us per launch.  A kernel boundary is the cheaper way to make the partials visible.)
__global__ __launch_bounds__(128) void sparse_rows_kernel(SparseArgs a)
{
    constexpr int HCH[CT_NUM_HEADS] = {2, 2, 2, 4, 4, 1, 8, 3, 2, 8, 3};
    const int b = blockIdx.x;
    const int HW = a.h * a.w;
    for (int r = threadIdx.x; r < a.K; r += blockDim.x) {
        const unsigned long long k = a.winners[(size_t)b * a.K + r];
        const unsigned flat = 0xFFFFFFFFu - (unsigned)(k & 0xFFFFFFFFull);
        const int cls = (int)(flat / (unsigned)HW);
        const int p = (int)(flat - (unsigned)cls * (unsigned)HW);
        const int y = p / a.w, x = p - y * a.w;
        const float score = ord2f((unsigned)(k >> 32));
        const int tile = b * a.tiles + (r >> 4), m = r & 15;
        float V[CT_NUM_HEADS][8];
#pragma unroll
        for (int hd = 0; hd < CT_NUM_HEADS; ++hd)
#pragma unroll
            for (int c = 0; c < 8; ++c) V[hd][c] = 0.f;
        for (int h2 = 0; h2 < a.nheads; ++h2) {
            const int hd2 = a.head[h2];
            const bool both = a.flip_B > 0 && a.flip_mode[h2] != 0;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                if (c >= HCH[hd2]) continue;
                auto value = [&](int pass) {
                    const float *pp = a.partial + ((((size_t)tile * a.nheads + h2) * 2 + pass) * 4 * 16 + m) * 8 + c;
                    float v = pp[0];                       // quarters summed in order: deterministic
                    v += pp[128];
                    v += pp[256];
                    v += pp[384];
                    v += a.b2[h2][c];
                    if (hd2 == CT_HEAD_DEP) v = (1.0f / (1.0f / (1.0f + expf(-v)) + 1e-6f) - 1.0f) * a.depth_scale;
                    return v;
                };
                float v = value(0);
                if (both) {
                    const float v1 = value(1);
                    v = (v + ((a.flip_mode[h2] == 2 && !(c & 1)) ? -v1 : v1)) / 2;     // (a + s * b) / 2 like ct_flip_merge / torch
                }
                if (hd2 == CT_HEAD_TRACKING && a.zero_tracking) v = 0.0f;
                // (hd2 is uniform but not a compile-time constant: a switch keeps V in registers)
#pragma unroll
                for (int hd = 0; hd < CT_NUM_HEADS; ++hd)
                    if (hd == hd2) V[hd][c] = v;
            }
        }
        float *row = a.out + ((size_t)b * a.K + r) * a.F;
        float *hrow = a.host_out ? a.host_out + ((size_t)b * a.K + r) * a.F : nullptr;
        emit_row(row, hrow, a.present, score, cls, (float)x, (float)y, [&](int hd3, int ch) { return V[hd3][ch]; });
    }
    if (a.done_flag) raise_done_flag(a.done_flag, a.done_counter, a.B);
}

const int kHeadCh[CT_NUM_HEADS] = {2, 2, 2, 4, 4, 1, 8, 3, 2, 8, 3};

int check(const ct_decode_desc *d, const char *who)
{
    if (!d || !d->hm) CT_FAIL_ARG("%s: null heat-map", who);
    if (d->B <= 0 || d->C <= 0 || d->h <= 0 || d->w <= 0) CT_FAIL_ARG("%s: bad shape", who);
    if (d->K <= 0 || d->K > MAXK) CT_FAIL_ARG("%s: K=%d out of range (1..%d)", who, d->K, MAXK);
    if ((long)d->h * d->w < d->K) CT_FAIL_ARG("%s: h*w=%ld < K=%d (torch.topk would raise too)", who, (long)d->h * d->w, d->K);
    if ((double)d->C * d->h * d->w >= 4294967295.0) CT_FAIL_ARG("%s: C*h*w too large", who);
    // the sparse-heads descriptor is validated HERE, for every entry point that reads it (ct_decode_workspace_bytes sizes the
    // workspace from nheads, ct_decode_row_floats walks head[]: a bad descriptor must be an error, not a wrapped size)
    if (const ct_sparse_heads_desc *sp = d->sparse) {
        if (sp->nheads < 1 || sp->nheads > CT_NUM_HEADS || !sp->feat || sp->ldf < 64 || (sp->ldf & 3) || ((uintptr_t)sp->feat & 15))
            CT_FAIL_ARG("%s: sparse heads need 1..%d heads and a 16-byte aligned NHWC feature view of >= 64 channels", who, CT_NUM_HEADS);
        for (int i = 0; i < sp->nheads; ++i) {
            if (sp->head[i] < 0 || sp->head[i] >= CT_NUM_HEADS || !sp->w1[i] || !sp->b1[i] || !sp->w2[i] || !sp->b2[i])
                CT_FAIL_ARG("%s: sparse head %d is incomplete", who, i);
            if (d->heads[sp->head[i]]) CT_FAIL_ARG("%s: head %d is both dense and sparse", who, sp->head[i]);
        }
        if (sp->flip_B < 0 || (sp->flip_B > 0 && sp->flip_B != d->B)) CT_FAIL_ARG("%s: sparse flip_B must be 0 or B (feat then holds 2 * B images)", who);
    }
    return CT_OK;
}

}  // namespace

// bit hd: head hd is part of the packed row -- it has a dense map or it is one of the sparse heads
static unsigned present_mask(const ct_decode_desc *d)
{
    unsigned m = 0;
    for (int i = 0; i < CT_NUM_HEADS; ++i)
        if (d->heads[i]) m |= 1u << i;
    if (d->sparse)
        for (int i = 0; i < d->sparse->nheads && i < CT_NUM_HEADS; ++i)
            if (d->sparse->head[i] >= 0 && d->sparse->head[i] < CT_NUM_HEADS) m |= 1u << d->sparse->head[i];
    return m;
}

extern "C" int ct_decode_row_floats(const ct_decode_desc *d)
{
    if (!d) return 0;
    if (d->sparse && (d->sparse->nheads < 1 || d->sparse->nheads > CT_NUM_HEADS)) {
        ct_set_error("ct_decode_row_floats: sparse nheads=%d out of range (1..%d)", d->sparse->nheads, CT_NUM_HEADS);
        return 0;
    }
    const unsigned m = present_mask(d);
    auto has = [&](int hd) { return (m >> hd) & 1u; };
    int f = 4;
    if (has(CT_HEAD_WH) || has(CT_HEAD_LTRB) || has(CT_HEAD_LTRB_AMODAL)) f += 4;
    if (has(CT_HEAD_LTRB_AMODAL)) f += 4;
    const int rest[7] = {CT_HEAD_TRACKING, CT_HEAD_DEP, CT_HEAD_ROT, CT_HEAD_DIM, CT_HEAD_AMODEL_OFFSET,
                         CT_HEAD_NUSCENES_ATT, CT_HEAD_VELOCITY};
    for (int q = 0; q < 7; ++q)
        if (has(rest[q])) f += kHeadCh[rest[q]];
    return f;
}

// pixels per stage-1 workgroup: as large as possible (<= SEG_MAX) while the launch still has
// >= 256 workgroups, never below 256
static int pick_seg(const ct_decode_desc *d)
{
    const long HW = (long)d->h * d->w;
    int seg = SEG_MAX;
    while (seg > 256 && (long)d->B * d->C * ((HW + seg - 1) / seg) < 256) seg >>= 1;
    return seg;
}

// stage 2a groups per image (0 = stage 2 reads the stage-1 candidates itself): one group per 8192 candidate slots
static int pick_groups(const ct_decode_desc *d, int seg, int nseg)
{
    const long M2 = (long)d->C * nseg * d->K;
    (void)seg;
    if (M2 <= SLICE) return 0;                            // stage 2 holds them all in registers
    const long G = (M2 + SLICE - 1) / SLICE;
    return (int)G;
}

// sparse heads (descriptor validated by check()): [B][K] winner keys | partial head outputs
static size_t sparse_bytes(const ct_decode_desc *d)
{
    if (!d->sparse) return 0;
    const size_t tiles = (size_t)d->B * ct_cdiv(d->K, 16);
    return (size_t)d->B * d->K * 8 + tiles * (size_t)d->sparse->nheads * 2 * 4 * 16 * 8 * sizeof(float);
}

extern "C" size_t ct_decode_workspace_bytes(const ct_decode_desc *d)
{
    if (check(d, "ct_decode_workspace_bytes") != CT_OK) return 0;
    const int seg = pick_seg(d);
    const int nseg = ct_cdiv(d->h * d->w, seg);
    const long M2 = (long)d->C * nseg * d->K;
    return ((size_t)d->B * M2 + (size_t)d->B * pick_groups(d, seg, nseg) * d->K) * sizeof(unsigned long long) + sparse_bytes(d);
}

extern "C" int ct_decode(const ct_decode_desc *d, void *stream)
{
    int rc = check(d, "ct_decode");
    if (rc != CT_OK) return rc;
    if (!d->out) CT_FAIL_ARG("ct_decode: null output");
    if (d->done_flag && (!d->done_counter || !d->host_out)) CT_FAIL_ARG("ct_decode: done_flag needs host_out and done_counter");
    const int seg = pick_seg(d);
    const int nseg = ct_cdiv(d->h * d->w, seg);
    const long M2 = (long)d->C * nseg * d->K;
    const int G = pick_groups(d, seg, nseg);
    const size_t need = ((size_t)d->B * M2 + (size_t)d->B * G * d->K) * sizeof(unsigned long long) + sparse_bytes(d);
    const ct_sparse_heads_desc *sp = d->sparse;
    if (!d->workspace || d->workspace_bytes < need) {
        ct_set_error("ct_decode: needs %zu workspace bytes, got %zu", need, d->workspace_bytes);
        return CT_ERR_WORKSPACE;
    }
    if (d->w > 4096) CT_FAIL_ARG("ct_decode: w=%d > 4096 unsupported", d->w);
    hipStream_t s = (hipStream_t)stream;
    Stage1Args a1;
    a1.hm = d->hm; a1.cand = (unsigned long long *)d->workspace;
    a1.B = d->B; a1.C = d->C; a1.h = d->h; a1.w = d->w; a1.K = d->K; a1.nseg = nseg; a1.seg = seg;
    const size_t HWs = (size_t)d->h * d->w;
    a1.hm_bs = d->hm_batch_stride ? (size_t)d->hm_batch_stride : (size_t)d->C * HWs;
    const int max_rows = ct_cdiv(seg, d->w) + 3;
    const size_t lds1 = SEG_MAX * sizeof(unsigned long long) + (size_t)max_rows * d->w * sizeof(float);
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
    a2.cand = a1.cand; a2.hm = d->hm;
    for (int i = 0; i < CT_NUM_HEADS; ++i) {
        a2.heads[i] = d->heads[i]; a2.head_ch[i] = kHeadCh[i];
        a2.head_bs[i] = d->head_batch_stride[i] ? (size_t)d->head_batch_stride[i] : (size_t)kHeadCh[i] * HWs;
    }
    a2.hm_bs = a1.hm_bs;
    a2.out = d->out; a2.inds = (long long *)d->inds;
    a2.B = d->B; a2.C = d->C; a2.h = d->h; a2.w = d->w; a2.K = d->K; a2.nseg = nseg;
    a2.F = d->out_stride ? d->out_stride : ct_decode_row_floats(d);   // floats between consecutive rows
    if (a2.F < ct_decode_row_floats(d)) CT_FAIL_ARG("ct_decode: out_stride %d < row floats", d->out_stride);
    a2.M2 = (int)M2; a2.keys_final = 0;
    a2.host_out = d->host_out; a2.done_flag = d->done_flag; a2.done_counter = d->done_counter;
    a2.present = present_mask(d);
    a2.winners = sp ? (unsigned long long *)d->workspace + (size_t)d->B * M2 + (size_t)d->B * G * d->K : nullptr;
    if (G > 0) {
        Stage2aArgs aa;
        aa.cand = a1.cand; aa.cand2 = a1.cand + (size_t)d->B * M2;
        aa.M2 = (int)M2; aa.G = G; aa.K = d->K; aa.per_class = nseg * d->K; aa.HW = d->h * d->w;
        hipLaunchKernelGGL(decode_stage2a_kernel, dim3((unsigned)(d->B * G)), dim3(1024), 0, s, aa);
        CT_CHECK_LAUNCH("ct_decode(stage 2a)");
        a2.cand = aa.cand2; a2.M2 = G * d->K; a2.keys_final = 1;
    }
    hipLaunchKernelGGL(decode_stage2_kernel, dim3((unsigned)d->B), dim3(1024), 0, s, a2);
    CT_CHECK_LAUNCH("ct_decode(stage 2)");
    if (sp) {
        SparseArgs sa;
        sa.feat = sp->feat; sa.ldf = sp->ldf; sa.feat_bs = (size_t)d->h * d->w * sp->ldf;
        sa.nheads = sp->nheads;
        for (int i = 0; i < CT_NUM_HEADS; ++i) {
            const bool on = i < sp->nheads;
            sa.head[i] = on ? sp->head[i] : 0;
            sa.w1[i] = on ? sp->w1[i] : nullptr; sa.b1[i] = on ? sp->b1[i] : nullptr;
            sa.w2[i] = on ? sp->w2[i] : nullptr; sa.b2[i] = on ? sp->b2[i] : nullptr;
        }
        sa.depth_scale = sp->depth_scale; sa.zero_tracking = sp->zero_tracking;
        sa.flip_B = sp->flip_B;
        for (int i = 0; i < CT_NUM_HEADS; ++i) sa.flip_mode[i] = (i < sp->nheads) ? sp->flip_mode[i] : 0;
        sa.partial = (float *)(a2.winners + (size_t)d->B * d->K);
        sa.winners = a2.winners; sa.out = d->out; sa.host_out = d->host_out;
        sa.done_flag = d->done_flag; sa.done_counter = d->done_counter;
        sa.B = d->B; sa.h = d->h; sa.w = d->w; sa.K = d->K; sa.F = a2.F; sa.tiles = ct_cdiv(d->K, 16);
        sa.present = a2.present;
        hipLaunchKernelGGL(sparse_heads_kernel, dim3((unsigned)(d->B * sa.tiles * sp->nheads * 4)), dim3(256), 0, s, sa);
        CT_CHECK_LAUNCH("ct_decode(sparse heads)");
        hipLaunchKernelGGL(sparse_rows_kernel, dim3((unsigned)d->B), dim3(128), 0, s, sa);
        CT_CHECK_LAUNCH("ct_decode(sparse rows)");
    }
    return CT_OK;
}
