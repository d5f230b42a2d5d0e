// Dense NHWC convolution as an implicit GEMM on the gfx950 fp32 matrix cores.
//
//   D[pixel, cout] = sum_{tap, ci} X[pixel + tap, ci] * W[cout, ci, tap]
//
// GEMM roles: M = output pixels (MFMA A operand / D rows), N = couts (B operand / D
// cols), K = (tap, ci).  One workgroup (4 waves) owns a TH x 16 pixel tile x BN couts.
//   * A side: the input patch of the tile (halo included, zero padded) for one chunk of
//     16*NKK channels is staged ONCE into LDS as [kk][patch pixel][16 ch] and re-used by
//     all KS*KS taps: an A fragment is just a ds_read_b128 at a tap-shifted patch address
//     (lane l: pixel l&15, channels 4*(l>>4)..+3), so nothing is re-staged per tap.
//     A 16-byte-slot XOR swizzle (slot ^= (P>>1)&2) makes those reads conflict free for
//     stride 1 and 2-way for stride 2 (checked by brute force).
//   * B side: weights are pre-packed so that one (tap, 16-ch slab, 16-cout tile) fragment
//     is a contiguous 1 KiB block; every wave loads its fragments straight from L2/L1
//     into VGPRs (global_load_dwordx4, lane-linear), one step ahead of the MFMAs.
//   * v_mfma_f32_16x16x4_f32, 4 per (m-tile, n-tile, 16 channels); fp32 in, fp32 acc,
//     bit-identical to an fmaf chain, so the result only differs from the reference by
//     summation order.
//   * LDS patch is double buffered: one barrier per channel chunk.
//   * split-K over channel chunks (gridDim.y) writes raw partials to a workspace; a
//     second kernel reduces them in a fixed order (deterministic) and applies the epilogue.
#include "ct_common.h"
#define CT_KS_STAMP(i) CT_STAMP(i)
#include "ksplit_core.h"

CT_DEFINE_STAMPS(conv)      // (tools/conv_phases.py; expands to nothing in the shipped build)

namespace {

struct ConvArgs {
    const float *x;
    const float *wp;
    int N, H, W, Cin, ldx;
    int tilesX, tilesY, coutBlocks, xcdPer;
    int NT;          // CoutPad / 16
    int nchunks;     // Cin / (16*NKK)
    int chunksPerSplit;
    float *ws;       // split-K workspace or nullptr
    int wsCout;      // channel pitch of the workspace
    EpiArgs epi;
    float *pool_y;   // 3x3 stride-2 launches: 2x2 max-pool of the input as a side output (NHWC, pitch pool_ld), or nullptr
    int pool_ld;
    // 3x3 stride-2 launches (round 4): Tree.project (dla.py:196-203,217-218: conv1x1 + BN of the 2x2 max-pooled input, no
    // ReLU) as a second output of the same workgroups.  The pooled A fragment costs nothing: the 2x2 window of output
    // pixel (oy, ox) is exactly taps (1,1) (1,2) (2,1) (2,2) of the 3x3 stride-2 window, i.e. the element-wise maximum of
    // four A fragments the tap loop reads anyway; one more "tap" of MFMAs per 16-channel slab contracts it with the 1x1
    // weights.  proj_wp: ct_pack_conv_weight of [Cout, Cin, 1, 1] (same n-tiles as wp), or nullptr.
    const float *proj_wp;
    EpiArgs epi2;
};

// 2x2 / stride-2 max-pool of the input from the staged LDS patch of a 3x3 stride-2 conv tile: output pixel (oy, ox) of
// the tile covers input pixels (2oy .. 2oy+1, 2ox .. 2ox+1) = patch pixels (2*oyl + 1 .., 2*oxl + 1 ..) (the patch
// starts one pixel above / left of the tile).  `buf`: the chunk's [slabs][PP][16] swizzled patch, `c0`: its first
// channel.  Called by the cout-block-0 workgroups only (every channel chunk of every tile passes exactly one of them).
template <int TH, int PW, int SLAB, int SLABS, int NTHR>
__device__ __forceinline__ void pool_from_patch(const float *buf, int c0, float *pool_y, int pool_ld, int n, int oy0,
                                                int ox0, int Ho, int Wo)
{
    constexpr int ITEMS = TH * 16 * SLABS * 4;
    for (int it = threadIdx.x; it < ITEMS; it += NTHR) {
        const int q = it & 3;
        const int kk = (it >> 2) % SLABS;
        const int px = (it >> 2) / SLABS;
        const int oxl = px & 15, oyl = px >> 4;
        const int oy = oy0 + oyl, ox = ox0 + oxl;
        if (oy >= Ho || ox >= Wo) continue;
        const int P0 = (2 * oyl + 1) * PW + 2 * oxl + 1;
        f32x4 m;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int P = P0 + (j >> 1) * PW + (j & 1);
            const f32x4 v = *reinterpret_cast<const f32x4 *>(buf + kk * SLAB + P * 16 + ((q ^ ((P >> 1) & 2)) << 2));
            if (j == 0) m = v;
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) m[e] = fmaxf(m[e], v[e]);
            }
        }
        *reinterpret_cast<f32x4 *>(pool_y + (((size_t)n * Ho + oy) * Wo + ox) * pool_ld + c0 + kk * 16 + q * 4) = m;
    }
}

template <int KS, int STRIDE, int WGM, int WGN, int WM, int WN, int NKK>
struct ConvCfg {
    static constexpr int TH = WGM * WM;
    static constexpr int BN = 16 * WN * WGN;
    static constexpr int PH = (TH - 1) * STRIDE + KS;
    static constexpr int PW = 15 * STRIDE + KS;
    static constexpr int PP = PH * PW;
    static constexpr int SLAB = PP * 16;                  // floats per 16-channel slab
    static constexpr int BUF = NKK * SLAB;                // floats per chunk buffer
    static constexpr int ITEMS = NKK * PP * 4;            // float4 items per chunk
    static constexpr int NR = (ITEMS + 255) / 256;
    static constexpr size_t LDS_BYTES = 2ull * BUF * sizeof(float);
};

// POOL: the instantiation that also writes the 2x2 max-pool of its input (3x3 stride-2 only; a separate
// instantiation because the extra code costs the plain kernels 12-20 VGPRs)
template <int KS, int STRIDE, int WGM, int WGN, int WM, int WN, int NKK, int PIPE, bool POOL = false>
__global__ __launch_bounds__(256) void conv_mfma_kernel(ConvArgs a)
{
    using C = ConvCfg<KS, STRIDE, WGM, WGN, WM, WN, NKK>;
    static_assert(WGM * WGN == 4, "4 waves per workgroup");
    constexpr int PAD = KS / 2;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    CT_STAMP_RT(0);
    CT_STAMP(1);
    CT_STAMP_HW(8);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // uniform: kept in an SGPR
    const int wm = wave / WGN, wn = wave % WGN;

    int bid = blockIdx.x;
    const int cb = ct_block_cout(bid, a.coutBlocks, a.xcdPer);
    const int tx = bid % a.tilesX; bid /= a.tilesX;
    const int ty = bid % a.tilesY; bid /= a.tilesY;
    const int n = bid;
    const int oy0 = ty * C::TH, ox0 = tx * 16;
    const int iy0 = oy0 * STRIDE - PAD, ix0 = ox0 * STRIDE - PAD;

    const int split = blockIdx.y;
    const int c_begin = split * a.chunksPerSplit;
    const int c_end = min(a.nchunks, c_begin + a.chunksPerSplit);

    const float *xin = a.x + (size_t)n * a.H * a.W * a.ldx;

    // ---- staging assignment (round 6): thread -> (channel quad q = tid & 3, pixel slot tid >> 2); a slot covers patch pixels
    // P, P + 64, .. of every 16-channel slab.  What depends on the pixel (row / column, inside the image?, swizzled LDS
    // address) is computed once per pixel round; the slab is an immediate offset of the loads and of the LDS stores (the item
    // list this replaces cost two integer divisions and an address per item: two thirds of the level-0 launch's vector
    // instructions, which are paid on top of its MFMA time).
    constexpr int RP = (C::PP + 63) / 64;           // pixel rounds
    constexpr int SNR = RP * NKK;                   // staged vectors per thread
    const int sq = tid & 3, sslot = tid >> 2;
    int goff[RP], loff[RP];
#pragma unroll
    for (int j = 0; j < RP; ++j) {
        const int P = sslot + 64 * j;
        const int py = P / C::PW, px = P - py * C::PW;
        const int iy = iy0 + py, ix = ix0 + px;
        loff[j] = (P < C::PP) ? P * 16 + ((sq ^ ((P >> 1) & 2)) << 2) : -1;
        goff[j] = (P < C::PP && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) ? ((iy * a.W + ix) * a.ldx + sq * 4) * 4 : (int)0x80000000;
    }
    f32x4 stage[SNR];
    // Branch-free staging through a buffer descriptor of this image: the pixel's byte offset in the vector offset, the chunk
    // in the scalar offset, the slab in the immediate; out-of-image / padding pixels carry an out-of-range offset and the
    // hardware returns zeros for them
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(xin), 0, (int)((((unsigned)a.H * a.W - 1u) * a.ldx + a.Cin) * 4u), 0x00020000);
    auto stage_load = [&](int chunk) {
        const int coff = chunk * (16 * NKK * 4);
#pragma unroll
        for (int k = 0; k < NKK; ++k)
#pragma unroll
            for (int j = 0; j < RP; ++j)
                stage[k * RP + j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, goff[j] + k * 64, coff, 0));
    };
    auto stage_store = [&](int buf) {
        float *dst = lds + buf * C::BUF;
#pragma unroll
        for (int j = 0; j < RP; ++j)
            if (loff[j] >= 0) {
#pragma unroll
                for (int k = 0; k < NKK; ++k) *reinterpret_cast<f32x4 *>(dst + loff[j] + k * C::SLAB) = stage[k * RP + j];
            }
    };

    // ---- fragment addressing -----------------------------------------------------------
    const int li = lane & 15, lg = lane >> 4;
    int pbase[WM];      // patch pixel of (row of m-tile, lane pixel) for tap (0,0)
#pragma unroll
    for (int mt = 0; mt < WM; ++mt) pbase[mt] = ((wm * WM + mt) * STRIDE) * C::PW + li * STRIDE;

    const int nt0 = cb * (C::BN / 16) + wn * WN;    // first n-tile of this wave
    const int NCH16 = a.Cin >> 4;
    // B fragments: one contiguous 1 KiB block per (tap, 16-ch slab, n-tile); n-tiles past the
    // padded Cout are clamped to the last valid tile (their results are never stored).
    // (round 6: the lane's part of the address is a constant vector offset, the (tap, slab) part a scalar offset)
    const int slab_bytes = a.NT << 10;                 // bytes between consecutive 16-ch slabs
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.wp), 0, KS * KS * NCH16 * slab_bytes, 0x00020000);
    int bvo[WN];
#pragma unroll
    for (int nt = 0; nt < WN; ++nt) bvo[nt] = (min(nt0 + nt, a.NT - 1) << 10) + (lane << 4);
    auto load_b = [&](f32x4 (&b)[WN], int chunk, int kk, int tap) {
        const int so = (tap * NCH16 + chunk * NKK + kk) * slab_bytes;
#pragma unroll
        for (int nt = 0; nt < WN; ++nt) b[nt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, bvo[nt], so, 0));
    };

    // epilogue operands (BN scale / shift of this wave's couts) first: they arrive while the main loop runs
    float psc[WN], psh[WN];
#pragma unroll
    for (int nt = 0; nt < WN; ++nt) ct_load_scale_shift(a.epi, (nt0 + nt) * 16, lane, psc[nt], psh[nt]);

    f32x4 acc[WM][WN];
#pragma unroll
    for (int mt = 0; mt < WM; ++mt)
#pragma unroll
        for (int nt = 0; nt < WN; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    // projection of the pooled input (POOL instantiations, when a.proj_wp is set): its own accumulators, the running
    // maximum of the four window taps and the 1x1 weight fragments of the current slab
    constexpr bool PROJ = POOL && KS == 3 && STRIDE == 2;
    const bool proj = PROJ && a.proj_wp != nullptr;              // (uniform)
    f32x4 accp[PROJ ? WM : 1][PROJ ? WN : 1];
    f32x4 pmax[PROJ ? WM : 1];
    f32x4 bproj[PROJ ? WN : 1];
    float psc2[PROJ ? WN : 1], psh2[PROJ ? WN : 1];
    if (PROJ) {
#pragma unroll
        for (int nt = 0; nt < WN; ++nt) {
            psc2[nt] = 1.0f; psh2[nt] = 0.0f;
            if (proj) ct_load_scale_shift(a.epi2, (nt0 + nt) * 16, lane, psc2[nt], psh2[nt]);
        }
    }
    if (PROJ) {
#pragma unroll
        for (int mt = 0; mt < WM; ++mt)
#pragma unroll
            for (int nt = 0; nt < WN; ++nt) accp[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    // Software pipeline over steps (kk, tap): B fragments are fetched RING-1 steps ahead of
    // their MFMAs into a static register ring; the next chunk's patch is fetched into
    // registers at the top of a chunk and written to the other LDS buffer at its end.
    constexpr int S = NKK * KS * KS;                   // steps per chunk
    constexpr int RING = (S % 3 == 0) ? 3 : ((S % 2 == 0) ? 2 : 1);
    if (c_begin < c_end) {
        // all first-use global loads go out together (one memory round trip before the first MFMA)
        stage_load(c_begin);
        f32x4 breg[RING][WN];
#pragma unroll
        for (int p = 0; p < RING - 1; ++p) load_b(breg[p], c_begin, p / (KS * KS), p % (KS * KS));   // S >= RING
        CT_STAMP(2);
        stage_store(0);
        __syncthreads();
        CT_STAMP(3);
        for (int c = c_begin; c < c_end; ++c) {
            const int cur = (c - c_begin) & 1;
            const int cnext = min(c + 1, c_end - 1);   // clamped: the last chunk re-fetches itself (unused)
            stage_load(cnext);
            if (PIPE == 1) __builtin_amdgcn_sched_barrier(0x386);
            const float *buf = lds + cur * C::BUF;
            if constexpr (POOL && KS == 3 && STRIDE == 2) {
                if (cb == 0 && a.pool_y)        // (uniform) the input's 2x2 max-pool as a side output
                    pool_from_patch<C::TH, C::PW, C::SLAB, NKK, 256>(buf, c * (16 * NKK), a.pool_y, a.pool_ld, n, oy0, ox0,
                                                                     a.epi.Ho, a.epi.Wo);
            }
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const int kk = s / (KS * KS), tap = s % (KS * KS);
                {   // prefetch step s + RING - 1 (static position in the ring)
                    constexpr int dummy = 0; (void)dummy;
                    const int sp = s + RING - 1;
                    const int pc = (sp >= S) ? cnext : c;
                    const int ps = (sp >= S) ? sp - S : sp;
                    if (RING > 1) load_b(breg[sp % RING], pc, ps / (KS * KS), ps % (KS * KS));
                }
                if (RING == 1) load_b(breg[0], c, kk, tap);
                // pin the prefetch loads here (hipcc otherwise sinks them towards their use)
                if (PIPE == 1) __builtin_amdgcn_sched_barrier(0x386);
                const int ky = tap / KS, kx = tap % KS;
                if constexpr (PROJ) {
                    if (tap == 0 && proj) {     // the slab's 1x1 weights: 8 taps ahead of their MFMAs
                        const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.proj_wp), 0, NCH16 * slab_bytes, 0x00020000);
#pragma unroll
                        for (int nt = 0; nt < WN; ++nt)
                            bproj[nt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(prs, bvo[nt], (c * NKK + kk) * slab_bytes, 0));
                    }
                }
                f32x4 af[WM];
#pragma unroll
                for (int mt = 0; mt < WM; ++mt) {
                    const int P = pbase[mt] + ky * C::PW + kx;
                    af[mt] = *reinterpret_cast<const f32x4 *>(buf + kk * C::SLAB + P * 16 +
                                                              ((lg ^ ((P >> 1) & 2)) << 2));
                }
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int mt = 0; mt < WM; ++mt)
#pragma unroll
                        for (int nt = 0; nt < WN; ++nt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[mt][e], breg[s % RING][nt][e],
                                                                              acc[mt][nt], 0, 0, 0);
                if constexpr (PROJ) {
                    // taps (1,1) (1,2) (2,1) (2,2) = the 2x2 pooling window of this lane's output pixel
                    if (tap == 4) {
#pragma unroll
                        for (int mt = 0; mt < WM; ++mt) pmax[mt] = af[mt];
                    } else if (tap == 5 || tap == 7 || tap == 8) {
#pragma unroll
                        for (int mt = 0; mt < WM; ++mt)
#pragma unroll
                            for (int e = 0; e < 4; ++e) pmax[mt][e] = fmaxf(pmax[mt][e], af[mt][e]);
                    }
                    if (tap == 8 && proj) {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
#pragma unroll
                            for (int mt = 0; mt < WM; ++mt)
#pragma unroll
                                for (int nt = 0; nt < WN; ++nt)
                                    accp[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(pmax[mt][e], bproj[nt][e], accp[mt][nt], 0, 0, 0);
                    }
                }
            }
            if (c + 1 < c_end) stage_store(cur ^ 1);   // (a single-chunk launch allocates one buffer only)
            __syncthreads();
        }
    }

    CT_STAMP(4);
    CT_STAMP(5);
    // ---- epilogue ----------------------------------------------------------------------
    if (a.ws) {
        // raw partial sums: ws[split][n*Ho*Wo + pixel][wsCout]
        const size_t Mtot = (size_t)a.N * a.epi.Ho * a.epi.Wo;
        float *wsp = a.ws + (size_t)split * Mtot * a.wsCout;
#pragma unroll
        for (int mt = 0; mt < WM; ++mt) {
            const int oy = oy0 + wm * WM + mt;
            if (oy >= a.epi.Ho) continue;
#pragma unroll
            for (int nt = 0; nt < WN; ++nt) {
                const int co = (nt0 + nt) * 16 + li;
                if (co >= a.wsCout) continue;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int ox = ox0 + lg * 4 + e;
                    if (ox < a.epi.Wo)
                        wsp[(((size_t)n * a.epi.Ho + oy) * a.epi.Wo + ox) * a.wsCout + co] = acc[mt][nt][e];
                }
            }
        }
    } else {
#pragma unroll
        for (int mt = 0; mt < WM; ++mt)
#pragma unroll
            for (int nt = 0; nt < WN; ++nt)
                ct_store_tile(a.epi, acc[mt][nt], n, oy0 + wm * WM + mt, ox0, (nt0 + nt) * 16, lane, psc[nt], psh[nt]);
    }
    if constexpr (PROJ) {
        if (proj) {
#pragma unroll
            for (int mt = 0; mt < WM; ++mt)
#pragma unroll
                for (int nt = 0; nt < WN; ++nt)
                    ct_store_tile(a.epi2, accp[mt][nt], n, oy0 + wm * WM + mt, ox0, (nt0 + nt) * 16, lane, psc2[nt], psh2[nt]);
        }
    }
    CT_STAMP(6);
    CT_STAMP_RT(7);
}


// ---------------------------------------------------------------------------------------------
// K-split variant for layers with few output tiles (deep levels: 16x16 .. 64x64 maps with
// 128..1280 input channels): see ksplit_core.h.
template <int KS, int STRIDE, int WM, int WN, int WK, bool POOL = false>
__global__ __launch_bounds__(64 * WK) void conv_ksplit_kernel(ConvArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    CT_STAMP_RT(0);
    CT_STAMP(1);
    CT_STAMP_HW(8);
    const int lane = threadIdx.x & 63;
    int bid = blockIdx.x;
    const int cb = ct_block_cout(bid, a.coutBlocks, a.xcdPer);
    const int tx = bid % a.tilesX; bid /= a.tilesX;
    const int ty = bid % a.tilesY; bid /= a.tilesY;
    const int n = bid;
    const int oy0 = ty * WM, ox0 = tx * 16;
    const int split = blockIdx.y;
    const int c_begin = split * a.chunksPerSplit;
    const int c_end = min(a.nchunks, c_begin + a.chunksPerSplit);
    const float *xin = a.x + (size_t)n * a.H * a.W * a.ldx;
    const int nt0 = cb * WN;
    using KC = KsCfg<KS, STRIDE, WM, WN, WK>;
    auto hook = [&](int c, const float *buf) {
        if constexpr (POOL && KS == 3 && STRIDE == 2) {
            if (cb == 0 && a.pool_y)            // (uniform) the input's 2x2 max-pool as a side output
                pool_from_patch<WM, KC::PW, KC::SLAB, WK, 64 * WK>(buf, c * (16 * WK), a.pool_y, a.pool_ld, n, oy0, ox0,
                                                                   a.epi.Ho, a.epi.Wo);
        }
    };
    // epilogue operands of the tiles THIS wave finalises (tile j * WK + wave of the workgroup's WM x WN tiles), loaded first
    constexpr int NJ = (WM * WN + WK - 1) / WK;
    const int wave_id = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float psc[NJ], psh[NJ], psc2[NJ], psh2[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int t = min(j * WK + wave_id, WM * WN - 1);
        ct_load_scale_shift(a.epi, (nt0 + t % WN) * 16, lane, psc[j], psh[j]);
        psc2[j] = 1.0f; psh2[j] = 0.0f;
        if constexpr (POOL && KS == 3 && STRIDE == 2) {
            if (a.proj_wp) ct_load_scale_shift(a.epi2, (nt0 + t % WN) * 16, lane, psc2[j], psh2[j]);
        }
    }
    auto fin_main = [&](int mt, int nt, f32x4 sum, int j) {
            const int oy = oy0 + mt;
            if (a.ws) {
                const int li = lane & 15, lg = lane >> 4;
                const size_t Mtot = (size_t)a.N * a.epi.Ho * a.epi.Wo;
                float *wsp = a.ws + (size_t)split * Mtot * a.wsCout;
                const int co = (nt0 + nt) * 16 + li;
                if (oy < a.epi.Ho && co < a.wsCout) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int ox = ox0 + lg * 4 + e;
                        if (ox < a.epi.Wo) wsp[(((size_t)n * a.epi.Ho + oy) * a.epi.Wo + ox) * a.wsCout + co] = sum[e];
                    }
                }
            } else {
                ct_store_tile(a.epi, sum, n, oy, ox0, (nt0 + nt) * 16, lane, psc[j], psh[j]);
            }
        };
    if constexpr (POOL && KS == 3 && STRIDE == 2) {
        // (the POOL instantiations also carry Tree.project of the pooled input, see ConvArgs::proj_wp)
        auto fin_proj = [&](int mt, int nt, f32x4 sum, int j) {
            ct_store_tile(a.epi2, sum, n, oy0 + mt, ox0, (nt0 + nt) * 16, lane, psc2[j], psh2[j]);
        };
        ksplit_conv_tile<KS, STRIDE, WM, WN, WK, 2>(xin, a.H, a.W, a.ldx, a.Cin, a.wp, a.NT, nt0, oy0, ox0, c_begin, c_end, lds,
                                                    fin_main, hook, a.proj_wp, fin_proj);
    } else {
        ksplit_conv_tile<KS, STRIDE, WM, WN, WK, 2>(xin, a.H, a.W, a.ldx, a.Cin, a.wp, a.NT, nt0, oy0, ox0, c_begin, c_end, lds,
                                                    fin_main, hook);
    }
    CT_STAMP(6);
    CT_STAMP_RT(7);
}

// Deterministic split-K reduction + epilogue: one thread per (pixel, 4 couts).
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float *ws, int splits, size_t Mtot, int wsCout,
                                                            int N, EpiArgs e)
{
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int quads = wsCout >> 2;
    if (idx >= Mtot * quads) return;
    const size_t m = idx / quads;
    const int c4 = (int)(idx - m * quads) << 2;
    f32x4 s = *reinterpret_cast<const f32x4 *>(ws + m * wsCout + c4);
    for (int k = 1; k < splits; ++k) {
        const f32x4 t = *reinterpret_cast<const f32x4 *>(ws + ((size_t)k * Mtot + m) * wsCout + c4);
        s += t;
    }
    const size_t hw = (size_t)e.Ho * e.Wo;
    const int n = (int)(m / hw);
    const size_t p = m - (size_t)n * hw;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int co = c4 + i;
        if (co >= e.Cout) break;
        const float sc = e.scale ? e.scale[co] : 1.0f;
        const float sh = e.shift ? e.shift[co] : 0.0f;
        const float r = e.res ? e.res[m * e.ldr + co] : 0.0f;
        const float v = ct_epilogue_value(e, s[i], co, sc, sh, r);
        if (e.flags & CT_OUT_NCHW)
            e.y[((size_t)n * e.Cout + co) * hw + p] = v;
        else
            e.y[m * e.ldy + co] = v;
    }
}

__global__ __launch_bounds__(256) void pack_weight_kernel(const float *w, float *p, int Cout, int Cin, int ks,
                                                          int NT)
{
    // p[((tap*Cin16 + c16)*NT + nt)*256 + g*64 + j*4 + e] = w[co = nt*16+j][ci = c16*16 + 4g + e][tap]
    const size_t total = (size_t)ks * ks * (Cin >> 4) * NT * 256;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int e = idx & 3, j = (idx >> 2) & 15, g = (idx >> 6) & 3;
    size_t t = idx >> 8;
    const int nt = t % NT; t /= NT;
    const int c16 = t % (Cin >> 4);
    const int tap = (int)(t / (Cin >> 4));
    const int co = nt * 16 + j, ci = c16 * 16 + 4 * g + e;
    p[idx] = (co < Cout) ? w[((size_t)co * Cin + ci) * ks * ks + tap] : 0.0f;
}

struct Plan {
    int pipe;
    int ks;        // >= 0: K-split kernel configuration (kKs table), -1: regular kernel
    int cfg;       // see launch_tile2
    int TH, BN, nkk;
    int tilesX, tilesY, coutBlocks, nchunks, splits, chunksPerSplit;
    int Ho, Wo, NT;
};

// K-split configurations {WM, WN, WK}: tile = WM rows x 16 px x 16*WN couts, WK waves split K
constexpr int kNumKs = 5;
const int kKs[kNumKs][3] = {{2, 2, 4}, {1, 2, 4}, {1, 2, 8}, {2, 4, 4}, {2, 2, 8}};

int make_plan(const ct_conv_desc *d, Plan *p)
{
    if (!d || !d->x || !d->w_packed || !d->y) CT_FAIL_ARG("ct_conv2d: null pointer");
    if (d->ks != 1 && d->ks != 3) CT_FAIL_ARG("ct_conv2d: ks=%d unsupported (1 or 3)", d->ks);
    if (d->stride != 1 && !(d->stride == 2 && d->ks == 3))
        CT_FAIL_ARG("ct_conv2d: stride=%d with ks=%d unsupported", d->stride, d->ks);
    if (d->Cin % 16 || d->Cin <= 0) CT_FAIL_ARG("ct_conv2d: Cin=%d must be a positive multiple of 16", d->Cin);
    if (d->ldx % 4 || ((uintptr_t)d->x & 15)) CT_FAIL_ARG("ct_conv2d: input view must be 16-byte aligned (ld %% 4 == 0)");
    if (d->Cout <= 0 || d->N <= 0 || d->H <= 0 || d->W <= 0) CT_FAIL_ARG("ct_conv2d: bad shape");
    const int pad = d->ks / 2;
    p->Ho = (d->H + 2 * pad - d->ks) / d->stride + 1;
    p->Wo = (d->W + 2 * pad - d->ks) / d->stride + 1;
    p->NT = ct_cdiv(d->Cout, 16);
    static const int kTH[8] = {16, 8, 4, 4, 2, 4, 8, 4}, kBN[8] = {16, 32, 64, 128, 64, 32, 16, 16};
    p->tilesX = ct_cdiv(p->Wo, 16);
    auto tiles_of = [&](int cfg) {
        return (long)d->N * p->tilesX * ct_cdiv(p->Ho, kTH[cfg]) * ct_cdiv(d->Cout, kBN[cfg]);
    };
    if (d->Cout <= 16) p->cfg = 0;
    else if (d->Cout <= 32) p->cfg = (tiles_of(1) >= ct_tune_get(CT_TUNE_CONV_SMALL_TILES)) ? 1 : 5;
    else {
        p->cfg = 2;
        if (tiles_of(2) < ct_tune_get(CT_TUNE_CONV_SMALL_TILES) || tiles_of(2) >= 2048) p->cfg = 4;
    }
    if (ct_tune_get(CT_TUNE_CONV_CFG) >= 0) p->cfg = ct_tune_get(CT_TUNE_CONV_CFG);
    if (d->algo >= 1 && d->algo <= 8) p->cfg = d->algo - 1;
    else if (d->algo != 0 && !(d->algo >= 101 && d->algo < 101 + kNumKs)) CT_FAIL_ARG("ct_conv2d: unknown algo %d", d->algo);
    p->pipe = ct_tune_get(CT_TUNE_CONV_PIPE);
    p->TH = kTH[p->cfg];
    p->BN = kBN[p->cfg];
    p->tilesY = ct_cdiv(p->Ho, p->TH);
    p->coutBlocks = ct_cdiv(d->Cout, p->BN);
    const int c16 = d->Cin / 16;
    if (d->ks == 1) {
        p->nkk = (c16 % 4 == 0) ? 4 : ((c16 % 2 == 0) ? 2 : 1);
        if (p->cfg <= 1 && p->nkk > 2) p->nkk = 2;      // big pixel tiles: keep the LDS patch <= 64 KiB
    }
    else if (d->stride == 2) p->nkk = 1;
    else p->nkk = (c16 % 2 == 0) ? 2 : 1;
    p->nchunks = c16 / p->nkk;
    long tiles = (long)d->N * p->tilesX * p->tilesY * p->coutBlocks;
    // ---- K-split kernel for layers with few tiles (see conv_ksplit_kernel) -------------------
    p->ks = -1;
    {
        auto ks_ok = [&](int id) {
            if (id < 0 || id >= kNumKs) return false;
            if (d->Cin % (16 * kKs[id][2])) return false;
            if (d->stride == 2 && kKs[id][0] == 2 && kKs[id][2] == 8) return false;      // LDS patch too large
            return true;
        };
        auto ks_wgs = [&](int id) {
            return (long)d->N * p->tilesX * ct_cdiv(p->Ho, kKs[id][0]) * ct_cdiv(d->Cout, 16 * kKs[id][1]);
        };
        int want = ct_tune_get(CT_TUNE_CONV_KS);
        if (d->algo >= 101) {
            want = d->algo - 101;
            if (!ks_ok(want)) CT_FAIL_ARG("ct_conv2d: algo %d cannot run Cin=%d stride=%d", d->algo, d->Cin, d->stride);
        } else if (d->algo >= 1) want = -2;
        if (want >= 0) {
            if (ks_ok(want)) p->ks = want;
        } else if (want == -1 && d->split_k <= 0 && tiles_of(2) < ct_tune_get(CT_TUNE_CONV_KS_BELOW)) {
            static const int order[5] = {3, 0, 1, 4, 2};           // largest tile first
            long best_waves = 0;
            for (int i = 0; i < 5; ++i) {
                const int id = order[i];
                if (!ks_ok(id)) continue;
                if (16 * kKs[id][1] > p->NT * 16 && kKs[id][1] > 2) continue;     // tile wider than the layer
                const long waves = ks_wgs(id) * kKs[id][2];
                if (waves >= ct_tune_get(CT_TUNE_CONV_KS_WAVES)) { p->ks = id; break; }
                if (waves > best_waves) { best_waves = waves; p->ks = id; }
            }
        }
        if (p->ks >= 0) {
            p->TH = kKs[p->ks][0];
            p->BN = 16 * kKs[p->ks][1];
            p->tilesY = ct_cdiv(p->Ho, p->TH);
            p->coutBlocks = ct_cdiv(d->Cout, p->BN);
            p->nchunks = d->Cin / (16 * kKs[p->ks][2]);
            tiles = (long)d->N * p->tilesX * p->tilesY * p->coutBlocks;
        }
    }
    int splits = d->split_k;
    if (p->ks >= 0 && splits <= 0) splits = 1;
    if (splits <= 0) {
        splits = 1;
        // (the fused projection has no split-K form: an automatic split would turn a valid descriptor into a plan-time
        // error, so only an EXPLICIT split_k > 1 is refused below; ct_conv2d_workspace_bytes reports 0 accordingly)
        if (d->workspace && tiles < 256 && !d->proj_w_packed) {
            splits = (int)((ct_tune_get(CT_TUNE_SPLITK_TARGET) + tiles - 1) / tiles);
            // keep at least ~2 chunks (>= 18 MFMA steps for 3x3) per split
            const int maxs = p->nchunks >= 2 ? p->nchunks / 2 : 1;
            if (splits > maxs) splits = maxs;
            if (splits > 32) splits = 32;
        }
    }
    if (splits > p->nchunks) splits = p->nchunks;
    if (splits < 1) splits = 1;
    p->chunksPerSplit = ct_cdiv(p->nchunks, splits);
    p->splits = ct_cdiv(p->nchunks, p->chunksPerSplit);
    return CT_OK;
}

size_t ws_bytes(const ct_conv_desc *d, const Plan &p)
{
    if (p.splits <= 1) return 0;
    return (size_t)p.splits * d->N * p.Ho * p.Wo * (size_t)(p.NT * 16) * sizeof(float);
}

template <int KS, int STRIDE, int WGM, int WGN, int WM, int WN, int NKK, int PIPE>
int launch_cfg(const ConvArgs &a, dim3 grid, hipStream_t s)
{
    using C = ConvCfg<KS, STRIDE, WGM, WGN, WM, WN, NKK>;
    const size_t lds = (a.chunksPerSplit == 1) ? C::LDS_BYTES / 2 : C::LDS_BYTES;
    if constexpr (KS == 3 && STRIDE == 2) {
        if (a.pool_y || a.proj_wp) {
            auto kp = conv_mfma_kernel<KS, STRIDE, WGM, WGN, WM, WN, NKK, PIPE, true>;
            static bool attr_set_p = false;
            if (!attr_set_p && C::LDS_BYTES > 48 * 1024) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kp), hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)C::LDS_BYTES);
                attr_set_p = true;
            }
            hipLaunchKernelGGL(kp, grid, dim3(256), lds, s, a);
            return CT_OK;
        }
    }
    auto k = conv_mfma_kernel<KS, STRIDE, WGM, WGN, WM, WN, NKK, PIPE>;
    static bool attr_set = false;   // > 64 KiB dynamic LDS needs the opt-in attribute
    if (!attr_set && C::LDS_BYTES > 48 * 1024) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)C::LDS_BYTES);
        attr_set = true;
    }
    hipLaunchKernelGGL(k, grid, dim3(256), lds, s, a);
    return CT_OK;
}

// tile configurations: {WGM, WGN, WM, WN} -> TH = WGM*WM rows x 16 px, BN = 16*WGN*WN couts
//   0: 256 px x 16   1: 128 px x 32   2: 64 px x 64   3: 64 px x 128   4: 32 px x 64   5: 64 px x 32
//   6: 128 px x 16   7: 64 px x 16
template <int KS, int STRIDE, int NKK, int PIPE>
int launch_tile2(int cfg, const ConvArgs &a, dim3 grid, hipStream_t s)
{
    switch (cfg) {
    case 0: return launch_cfg<KS, STRIDE, 4, 1, 4, 1, NKK, PIPE>(a, grid, s);
    case 1: return launch_cfg<KS, STRIDE, 4, 1, 2, 2, NKK, PIPE>(a, grid, s);
    case 2: return launch_cfg<KS, STRIDE, 2, 2, 2, 2, NKK, PIPE>(a, grid, s);
    case 3: return launch_cfg<KS, STRIDE, 2, 2, 2, 4, NKK, PIPE>(a, grid, s);
    case 4: return launch_cfg<KS, STRIDE, 1, 4, 2, 1, NKK, PIPE>(a, grid, s);
    case 6: return launch_cfg<KS, STRIDE, 4, 1, 2, 1, NKK, PIPE>(a, grid, s);
    case 7: return launch_cfg<KS, STRIDE, 4, 1, 1, 1, NKK, PIPE>(a, grid, s);
    default: return launch_cfg<KS, STRIDE, 2, 2, 2, 1, NKK, PIPE>(a, grid, s);
    }
}
template <int KS, int STRIDE, int NKK>
int launch_tile(int cfg, int pipe, const ConvArgs &a, dim3 grid, hipStream_t s)
{
    return pipe ? launch_tile2<KS, STRIDE, NKK, 1>(cfg, a, grid, s) : launch_tile2<KS, STRIDE, NKK, 0>(cfg, a, grid, s);
}

template <int KS, int STRIDE, int WM, int WN, int WK>
int launch_ks_cfg(const ConvArgs &a, dim3 grid, hipStream_t s)
{
    using C = KsCfg<KS, STRIDE, WM, WN, WK>;
    static_assert(C::LDS_BYTES <= 160 * 1024, "LDS patch too large");
    size_t lds = C::LDS_BYTES;
    if (a.chunksPerSplit == 1) {
        lds = sizeof(float) * (size_t)((C::BUF > C::RED) ? C::BUF : C::RED);
    }
    if constexpr (KS == 3 && STRIDE == 2) {
        if (a.pool_y || a.proj_wp) {
            auto kp = conv_ksplit_kernel<KS, STRIDE, WM, WN, WK, true>;
            static bool attr_set_p = false;
            if (!attr_set_p && C::LDS_BYTES > 48 * 1024) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kp), hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)C::LDS_BYTES);
                attr_set_p = true;
            }
            hipLaunchKernelGGL(kp, grid, dim3(64 * WK), lds, s, a);
            return CT_OK;
        }
    }
    auto k = conv_ksplit_kernel<KS, STRIDE, WM, WN, WK>;
    static bool attr_set = false;
    if (!attr_set && C::LDS_BYTES > 48 * 1024) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)C::LDS_BYTES);
        attr_set = true;
    }
    hipLaunchKernelGGL(k, grid, dim3(64 * WK), lds, s, a);
    return CT_OK;
}
template <int KS, int STRIDE>
int launch_ks(int id, const ConvArgs &a, dim3 grid, hipStream_t s)
{
    switch (id) {
    case 0: return launch_ks_cfg<KS, STRIDE, 2, 2, 4>(a, grid, s);
    case 1: return launch_ks_cfg<KS, STRIDE, 1, 2, 4>(a, grid, s);
    case 2: return launch_ks_cfg<KS, STRIDE, 1, 2, 8>(a, grid, s);
    case 3: return launch_ks_cfg<KS, STRIDE, 2, 4, 4>(a, grid, s);
    default:
        if constexpr (STRIDE == 1) return launch_ks_cfg<KS, STRIDE, 2, 2, 8>(a, grid, s);
        else return CT_ERR_ARG;
    }
}

}  // namespace

extern "C" size_t ct_packed_weight_elems(int Cout, int Cin, int ks)
{
    return (size_t)ks * ks * (Cin / 16) * ct_cdiv(Cout, 16) * 256;
}

extern "C" int ct_pack_conv_weight(const float *w_oihw, float *packed, int Cout, int Cin, int ks, void *stream)
{
    if (!w_oihw || !packed) CT_FAIL_ARG("ct_pack_conv_weight: null pointer");
    if (Cin % 16 || Cin <= 0 || Cout <= 0 || ks <= 0) CT_FAIL_ARG("ct_pack_conv_weight: Cin=%d must be a multiple of 16", Cin);
    const size_t total = ct_packed_weight_elems(Cout, Cin, ks);
    hipLaunchKernelGGL(pack_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       w_oihw, packed, Cout, Cin, ks, ct_cdiv(Cout, 16));
    CT_CHECK_LAUNCH("ct_pack_conv_weight");
    return CT_OK;
}

extern "C" size_t ct_conv2d_workspace_bytes(const ct_conv_desc *d)
{
    if (d && d->algo >= 201 && d->algo <= 211) return 0;
    Plan p;
    ct_conv_desc t = *d;
    float dummy;
    if (!t.workspace) t.workspace = &dummy;   // ask "how much would the auto plan want"
    if (!t.y) t.y = &dummy;
    if (make_plan(&t, &p) != CT_OK) return 0;
    return ws_bytes(&t, p);
}

int ct_conv2d_winograd(const ct_conv_desc *d, void *stream);       // wino_mfma.hip

extern "C" int ct_conv2d(const ct_conv_desc *d, void *stream)
{
    if (d && d->algo >= 201 && d->algo <= 211) {
        if (!d->x || !d->y) CT_FAIL_ARG("ct_conv2d: null pointer");
        if (d->ldx % 4 || ((uintptr_t)d->x & 15)) CT_FAIL_ARG("ct_conv2d: input view must be 16-byte aligned (ld %% 4 == 0)");
        return ct_conv2d_winograd(d, stream);
    }
    Plan p;
    int rc = make_plan(d, &p);
    if (rc != CT_OK) return rc;
    const size_t need = ws_bytes(d, p);
    if (need > 0 && (!d->workspace || d->workspace_bytes < need)) {
        ct_set_error("ct_conv2d: split_k=%d needs %zu workspace bytes, got %zu", p.splits, need, d->workspace_bytes);
        return CT_ERR_WORKSPACE;
    }
    if ((d->flags & CT_OUT_NCHW) && d->res) CT_FAIL_ARG("ct_conv2d: residual with NCHW output unsupported");
    ConvArgs a;
    a.x = d->x; a.wp = d->w_packed;
    a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.ldx = d->ldx;
    a.tilesX = p.tilesX; a.tilesY = p.tilesY; a.coutBlocks = p.coutBlocks; a.xcdPer = ct_xcd_per(p.coutBlocks);
    a.NT = p.NT; a.nchunks = p.nchunks; a.chunksPerSplit = p.chunksPerSplit;
    a.ws = p.splits > 1 ? d->workspace : nullptr;
    a.wsCout = p.NT * 16;
    a.epi.scale = d->scale; a.epi.shift = d->shift; a.epi.res = d->res; a.epi.y = d->y;
    a.epi.ldr = d->ldr; a.epi.ldy = d->ldy; a.epi.Cout = d->Cout; a.epi.Ho = p.Ho; a.epi.Wo = p.Wo;
    a.epi.flags = d->flags; a.epi.sig_lo = d->sig_lo; a.epi.sig_hi = d->sig_hi;
    a.epi.dep_lo = d->dep_lo; a.epi.dep_hi = d->dep_hi; a.epi.depth_scale = d->depth_scale;
    a.pool_y = nullptr; a.pool_ld = 0;
    if (d->pool_y) {
        if (!(d->ks == 3 && d->stride == 2)) CT_FAIL_ARG("ct_conv2d: pool_y is a side output of the 3x3 stride-2 shapes");
        if ((d->H & 1) || (d->W & 1) || d->pool_ld % 4 || d->pool_ld < d->Cin || ((uintptr_t)d->pool_y & 15))
            CT_FAIL_ARG("ct_conv2d: pool_y needs even H / W and a 16-byte aligned view of >= Cin channels");
        a.pool_y = d->pool_y; a.pool_ld = d->pool_ld;
    }
    a.proj_wp = nullptr;
    a.epi2 = a.epi;
    if (d->proj_w_packed) {
        if (!(d->ks == 3 && d->stride == 2)) CT_FAIL_ARG("ct_conv2d: proj_* is a second output of the 3x3 stride-2 shapes");
        if (!d->proj_y || (d->H & 1) || (d->W & 1) || d->proj_ldy < d->Cout)
            CT_FAIL_ARG("ct_conv2d: proj_y needs even H / W and a view of >= Cout channels");
        if (p.splits > 1) CT_FAIL_ARG("ct_conv2d: the fused projection cannot be combined with split-K (split_k=%d)", p.splits);
        a.proj_wp = d->proj_w_packed;
        a.epi2.scale = d->proj_scale; a.epi2.shift = d->proj_shift; a.epi2.res = nullptr; a.epi2.y = d->proj_y;
        a.epi2.ldr = 0; a.epi2.ldy = d->proj_ldy; a.epi2.flags = 0;
        a.epi2.sig_lo = a.epi2.sig_hi = a.epi2.dep_lo = a.epi2.dep_hi = 0;
    }
    const long blocks = (long)d->N * p.tilesX * p.tilesY * p.coutBlocks;
    if (blocks > 0x7fffffffL) CT_FAIL_ARG("ct_conv2d: grid too large");
    dim3 grid((unsigned)blocks, (unsigned)p.splits);
    hipStream_t s = (hipStream_t)stream;
    if (p.ks >= 0) {
        if (d->ks == 1) rc = launch_ks<1, 1>(p.ks, a, grid, s);
        else if (d->stride == 2) rc = launch_ks<3, 2>(p.ks, a, grid, s);
        else rc = launch_ks<3, 1>(p.ks, a, grid, s);
    } else if (d->ks == 1) {
        if (p.nkk == 4) rc = launch_tile<1, 1, 4>(p.cfg, p.pipe, a, grid, s);
        else if (p.nkk == 2) rc = launch_tile<1, 1, 2>(p.cfg, p.pipe, a, grid, s);
        else rc = launch_tile<1, 1, 1>(p.cfg, p.pipe, a, grid, s);
    } else if (d->stride == 2) {
        rc = launch_tile<3, 2, 1>(p.cfg, p.pipe, a, grid, s);
    } else {
        if (p.nkk == 2) rc = launch_tile<3, 1, 2>(p.cfg, p.pipe, a, grid, s);
        else rc = launch_tile<3, 1, 1>(p.cfg, p.pipe, a, grid, s);
    }
    CT_CHECK_LAUNCH("ct_conv2d");
    if (p.splits > 1) {
        const size_t Mtot = (size_t)d->N * p.Ho * p.Wo;
        const size_t n = Mtot * (size_t)(a.wsCout / 4);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d->workspace,
                           p.splits, Mtot, a.wsCout, d->N, a.epi);
        CT_CHECK_LAUNCH("ct_conv2d(split-K reduce)");
    }
    return rc;
}
