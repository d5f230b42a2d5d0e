// K-split-in-workgroup implicit-GEMM convolution tile (shared by conv_mfma.hip's conv_ksplit_kernel
// and the fused offset-conv stage of dcn_mfma.hip).
//
// ALL WK waves of a workgroup own the SAME small output tile (WM rows x 16 px  x  16*WN couts) and
// split the reduction: wave k contracts the k-th 16-channel slab of every 16*WK-channel chunk (all
// taps); the WK partial tiles are summed through LDS in wave order (deterministic) and a caller-
// supplied finaliser runs once per 16x16 tile.  This replaces the global split-K (partials through
// HBM + a second reduce launch) for layers with few output tiles and gives every SIMD >= 1-2 waves
// even when M x N is only 256 x 512.
#pragma once
#include <type_traits>

#include "ct_common.h"

#ifndef CT_KS_STAMP
#define CT_KS_STAMP(i)      // (conv_mfma.hip maps these to CT_STAMP for tools/conv_phases.py; the DCN kernels stamp their own phases)
#endif

template <int KS, int STRIDE, int WM, int WN, int WK>
struct KsCfg {
    static constexpr int NTHR = 64 * WK;
    static constexpr int BN = 16 * WN;
    static constexpr int PH = (WM - 1) * STRIDE + KS;
    static constexpr int PW = 15 * STRIDE + KS;
    static constexpr int PP = PH * PW;
    static constexpr int SLAB = PP * 16;
    static constexpr int BUF = WK * SLAB;
    static constexpr int ITEMS = WK * PP * 4;
    static constexpr int NR = (ITEMS + NTHR - 1) / NTHR;
    static constexpr int RED = WK * WM * WN * 256;
    static constexpr size_t LDS_BYTES = sizeof(float) * (size_t)((2 * BUF > RED) ? 2 * BUF : RED);
    // floats needed when a launch holds a single chunk (no double buffering)
    static constexpr int LDS1_FLOATS = (BUF > RED) ? BUF : RED;
};

// xin: image base (NHWC, pitch ldx); the tile's first output pixel is (oy0, ox0); wp: packed weights
// (NT n-tiles of 16 couts); this workgroup's first n-tile is nt0; chunks [c_begin, c_end) of 16*WK
// channels; lds: >= KsCfg::LDS_BYTES (or LDS1_FLOATS floats for a single chunk), free for reuse on
// return AFTER a __syncthreads().  fin(mt, nt, sum, j) is called by exactly one wave per tile: wave w finalises tiles
// j * WK + w, j = 0, 1, .. (j is a compile-time position: callers index per-tile registers they loaded early with it).
// PD = B prefetch distance in (tap) steps: 2 for the multi-chunk backbone layers (registers); the single-chunk offset
// conv of the DCN launches may fetch every tap's fragments up front (PD = KS*KS - 1): its 9-tap loop is otherwise
// paced by the L2 latency of the weights (512 MFMA clocks per tap against a > 1000-clock round trip).
struct KsNoHook {
    __device__ __forceinline__ void operator()(int, const float *) const {}
};

struct KsNoFin2 {
    __device__ __forceinline__ void operator()(int, int, f32x4, int) const {}
};

// hook(chunk, buf): called by all threads at the top of every chunk with the chunk's staged patch in LDS ([WK slabs][PP]
// [16 ch], swizzled like conv_mfma.hip's) -- lets a caller derive a side output from the input tile (2x2 max-pool)
// wp2 / fin2 (3x3 stride-2 tiles only, round 4): a SECOND product on the same tile -- the 1x1 convolution of the 2x2
// max-pooled input (Tree.project of the pooled tensor, dla.py:196-203,217): the pooling window of output pixel (oy, ox)
// is taps (1,1) (1,2) (2,1) (2,2) of its 3x3 stride-2 window, so the pooled A fragment is the element-wise maximum of four
// fragments the tap loop reads anyway; one extra "tap" of MFMAs per chunk contracts it with wp2 (packed [Cout, Cin, 1, 1],
// same n-tiles).  wp2 == nullptr: off (uniform).  fin2(mt, nt, sum) runs after a second reduction pass.
template <int KS, int STRIDE, int WM, int WN, int WK, int PD = 2, typename Fin, typename Hook = KsNoHook, typename Fin2 = KsNoFin2>
__device__ __forceinline__ void ksplit_conv_tile(const float *xin, int H, int W, int ldx, int Cin, const float *wp, int NT,
                                                 int nt0, int oy0, int ox0, int c_begin, int c_end, float *lds, Fin fin,
                                                 Hook hook = Hook(), const float *wp2 = nullptr, Fin2 fin2 = Fin2())
{
    constexpr bool SIDE = KS == 3 && STRIDE == 2 && !std::is_same<Fin2, KsNoFin2>::value;
    const bool side = SIDE && wp2 != nullptr;                    // (uniform)
    using C = KsCfg<KS, STRIDE, WM, WN, WK>;
    constexpr int PAD = KS / 2;
    constexpr int S = KS * KS;                       // steps (taps) per chunk and wave
    constexpr int D = PD;                            // B prefetch distance (steps)
    constexpr int R = D + 1;                         // register ring
    constexpr int U = (S % R == 0) ? 1 : R;          // chunk unroll so that ring slots stay static

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // = K slab inside a chunk (uniform: SGPR)
    const int iy0 = oy0 * STRIDE - PAD, ix0 = ox0 * STRIDE - PAD;

    // staging (round 6): thread -> (channel quad q = tid & 3, slot tid >> 2); a slot is patch pixel P (+ SP per pixel round) of
    // slab kk0 (+ KG per slab round).  Row / column, the inside-the-image test and the swizzled LDS address are computed once
    // per pixel round; the slab is an immediate offset of the loads and of the LDS stores.  The patch comes through a buffer
    // descriptor of the image: out-of-image pixels carry an out-of-range offset and the hardware returns zeros for them.
    constexpr int SLOTS = C::NTHR / 4;                   // 32 / 64 / 128
    constexpr int SP = SLOTS < 64 ? SLOTS : 64;          // pixel slots per round
    constexpr int KG = SLOTS / SP;                       // slab groups side by side in the thread block
    constexpr int RP = (C::PP + SP - 1) / SP;
    constexpr int RK = (WK + KG - 1) / KG;
    constexpr int SNR = RP * RK;
    const int sq = tid & 3, sslot = tid >> 2;
    const int skk0 = sslot / SP, sP0 = sslot - skk0 * SP;
    int goff[RP], loff[RP];
#pragma unroll
    for (int j = 0; j < RP; ++j) {
        const int P = sP0 + SP * j;
        const int py = P / C::PW, px = P - py * C::PW;
        const int iy = iy0 + py, ix = ix0 + px;
        loff[j] = (P < C::PP) ? skk0 * C::SLAB + P * 16 + ((sq ^ ((P >> 1) & 2)) << 2) : -1;
        goff[j] = (P < C::PP && iy >= 0 && iy < H && ix >= 0 && ix < W) ? ((iy * W + ix) * ldx + skk0 * 16 + sq * 4) * 4 : (int)0x80000000;
    }
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(xin), 0, (int)((((unsigned)H * W - 1u) * ldx + Cin) * 4u), 0x00020000);
    f32x4 stage[SNR];
    auto stage_load = [&](int chunk) {
        const int coff = chunk * (16 * WK * 4);
#pragma unroll
        for (int k = 0; k < RK; ++k)
#pragma unroll
            for (int j = 0; j < RP; ++j)
                if (skk0 + k * KG < WK)                         // (compile-time true unless WK is no multiple of KG)
                    stage[k * RP + j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, goff[j] + k * (KG * 64), coff, 0));
    };
    auto stage_store = [&](int buf) {
        float *dst = lds + buf * C::BUF;
#pragma unroll
        for (int j = 0; j < RP; ++j)
            if (loff[j] >= 0) {
#pragma unroll
                for (int k = 0; k < RK; ++k)
                    if (skk0 + k * KG < WK) *reinterpret_cast<f32x4 *>(dst + loff[j] + k * (KG * C::SLAB)) = stage[k * RP + j];
            }
    };

    const int li = lane & 15, lg = lane >> 4;
    int pbase[WM];
#pragma unroll
    for (int mt = 0; mt < WM; ++mt) pbase[mt] = (mt * STRIDE) * C::PW + li * STRIDE;
    const int NCH16 = Cin >> 4;
    const int slab_bytes = NT << 10;
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(wp), 0, KS * KS * NCH16 * slab_bytes, 0x00020000);
    int bvo[WN];
#pragma unroll
    for (int nt = 0; nt < WN; ++nt) bvo[nt] = (min(nt0 + nt, NT - 1) << 10) + (lane << 4);
    // B fragment of (chunk, tap) for THIS wave's slab; chunks past the end clamp (loaded, never used)
    auto load_b = [&](f32x4 (&b)[WN], int chunk, int tap) {
        const int so = (tap * NCH16 + min(chunk, c_end - 1) * WK + wave) * slab_bytes;
#pragma unroll
        for (int nt = 0; nt < WN; ++nt) b[nt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, bvo[nt], so, 0));
    };

    f32x4 acc[WM][WN];
#pragma unroll
    for (int mt = 0; mt < WM; ++mt)
#pragma unroll
        for (int nt = 0; nt < WN; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 acc2[SIDE ? WM : 1][SIDE ? WN : 1];
    f32x4 pmax[SIDE ? WM : 1];
    f32x4 b2[SIDE ? WN : 1];
    if (SIDE) {
#pragma unroll
        for (int mt = 0; mt < WM; ++mt)
#pragma unroll
            for (int nt = 0; nt < WN; ++nt) acc2[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    if (c_begin < c_end) {
        // all first-use global loads go out together (one memory round trip before the first MFMA)
        stage_load(c_begin);
        f32x4 breg[R][WN];
#pragma unroll
        for (int p = 0; p < D; ++p) load_b(breg[p % R], c_begin + p / S, p % S);
        CT_KS_STAMP(2);
        stage_store(0);
        __syncthreads();
        CT_KS_STAMP(3);
        for (int c0 = c_begin; c0 < c_end; c0 += U) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int c = c0 + u;
                if (c < c_end) {
                    const int cur = (c - c_begin) & 1;
                    stage_load(min(c + 1, c_end - 1));
                    __builtin_amdgcn_sched_barrier(0x386);
                    hook(c, lds + cur * C::BUF);
                    const float *buf = lds + cur * C::BUF + wave * C::SLAB;
#pragma unroll
                    for (int s = 0; s < S; ++s) {
                        const int g = u * S + s;                    // static position in the unrolled body
                        const int sp = s + D;
                        load_b(breg[(g + D) % R], c + sp / S, sp % S);
                        __builtin_amdgcn_sched_barrier(0x386);
                        const int ky = s / KS, kx = s % KS;
                        if constexpr (SIDE) {
                            if (s == 0 && side) {      // this wave's slab of the 1x1 weights: 8 taps ahead of its MFMAs
                                const __amdgpu_buffer_rsrc_t w2rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(wp2), 0, NCH16 * slab_bytes, 0x00020000);
#pragma unroll
                                for (int nt = 0; nt < WN; ++nt)
                                    b2[nt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(w2rs, bvo[nt], (c * WK + wave) * slab_bytes, 0));
                            }
                        }
                        f32x4 af[WM];
#pragma unroll
                        for (int mt = 0; mt < WM; ++mt) {
                            const int P = pbase[mt] + ky * C::PW + kx;
                            af[mt] = *reinterpret_cast<const f32x4 *>(buf + P * 16 + ((lg ^ ((P >> 1) & 2)) << 2));
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e)
#pragma unroll
                            for (int mt = 0; mt < WM; ++mt)
#pragma unroll
                                for (int nt = 0; nt < WN; ++nt)
                                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[mt][e], breg[g % R][nt][e],
                                                                                      acc[mt][nt], 0, 0, 0);
                        if constexpr (SIDE) {
                            if (s == 4) {
#pragma unroll
                                for (int mt = 0; mt < WM; ++mt) pmax[mt] = af[mt];
                            } else if (s == 5 || s == 7 || s == 8) {
#pragma unroll
                                for (int mt = 0; mt < WM; ++mt)
#pragma unroll
                                    for (int e = 0; e < 4; ++e) pmax[mt][e] = fmaxf(pmax[mt][e], af[mt][e]);
                            }
                            if (s == 8 && side) {
#pragma unroll
                                for (int e = 0; e < 4; ++e)
#pragma unroll
                                    for (int mt = 0; mt < WM; ++mt)
#pragma unroll
                                        for (int nt = 0; nt < WN; ++nt)
                                            acc2[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(pmax[mt][e], b2[nt][e], acc2[mt][nt], 0, 0, 0);
                            }
                        }
                    }
                    if (c + 1 < c_end) stage_store(cur ^ 1);
                    __syncthreads();
                }
            }
        }
    }

    CT_KS_STAMP(4);
    // ---- cross-wave reduction through LDS (wave order 0..WK-1), then the finaliser ------------
    constexpr int T = WM * WN;
    float *red = lds;
#pragma unroll
    for (int mt = 0; mt < WM; ++mt)
#pragma unroll
        for (int nt = 0; nt < WN; ++nt)
            *reinterpret_cast<f32x4 *>(red + ((wave * T + mt * WN + nt) * 64 + lane) * 4) = acc[mt][nt];
    __syncthreads();
    CT_KS_STAMP(5);
#pragma unroll
    for (int t0 = 0; t0 < T; t0 += WK) {
        const int t = t0 + wave;
        if (t < T) {
            f32x4 sum = *reinterpret_cast<const f32x4 *>(red + (t * 64 + lane) * 4);
#pragma unroll
            for (int w = 1; w < WK; ++w) sum += *reinterpret_cast<const f32x4 *>(red + ((w * T + t) * 64 + lane) * 4);
            fin(t / WN, t % WN, sum, t0 / WK);
        }
    }
    if constexpr (SIDE) {
        if (side) {                                  // second product: same reduction, its own finaliser
            __syncthreads();
#pragma unroll
            for (int mt = 0; mt < WM; ++mt)
#pragma unroll
                for (int nt = 0; nt < WN; ++nt)
                    *reinterpret_cast<f32x4 *>(red + ((wave * T + mt * WN + nt) * 64 + lane) * 4) = acc2[mt][nt];
            __syncthreads();
#pragma unroll
            for (int t0 = 0; t0 < T; t0 += WK) {
                const int t = t0 + wave;
                if (t < T) {
                    f32x4 sum = *reinterpret_cast<const f32x4 *>(red + (t * 64 + lane) * 4);
#pragma unroll
                    for (int w = 1; w < WK; ++w) sum += *reinterpret_cast<const f32x4 *>(red + ((w * T + t) * 64 + lane) * 4);
                    fin2(t / WN, t % WN, sum, t0 / WK);
                }
            }
        }
    }
}
