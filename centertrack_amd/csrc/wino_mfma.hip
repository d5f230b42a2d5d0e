// 3x3 stride-1 convolution by Winograd F(2x2, 3x3) on the fp32 matrix cores.
//
//   Y = A^T [ (G g G^T) .* (B^T d B) ] A          (Lavin & Gray): 16 multiplies per 2x2 outputs instead of 36
//
// fp32 MFMA on gfx950 runs at the fp32 vector rate, so the dense 3x3 layers are MFMA-bound and the 2.25x
// cut in multiplies is the one algorithmic lever left (cuDNN, the reference's backend, makes the same
// choice for these layers).  The element-wise product over channels is 16 independent GEMMs, one per
// position (r,c) of the 4x4 transformed tile:  M[pos][tile, co] = sum_ci V[pos][tile, ci] * U[pos][ci, co].
//
// One workgroup (4 waves) = 16 Winograd tiles (2 tile rows x 8 tile columns = 4 x 16 output pixels, the
// MFMA M dimension) x 16*WN couts.  Wave r owns the four positions (r, 0..3):
//   * the raw input patch (6 x 18 pixels, zero padded) of a 64-channel chunk is staged once in LDS (same
//     swizzled [slab][pixel][16 ch] layout as conv_mfma.hip); the input transform B^T d B is done ON THE FLY
//     while building an A fragment: 8 ds_read_b128 + 8 vector adds give the 4 fragments V[(r,0..3)] of a slab;
//   * U = G g G^T is precomputed by ct_pack_winograd_weight into MFMA fragment order [pos][Cin/16][NT][256]
//     (1 KiB per fragment, loaded L2 -> VGPR two steps ahead);
//   * output transform: columns (c) in registers, rows (r) across the four waves through LDS, then
//     scale/shift/residual/ReLU and the NHWC store.
#include "ct_common.h"

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {

struct WinoArgs {
    const float *x;
    const float *up;          // packed U
    int N, H, W, Cin, ldx;
    int tilesX, tilesY, coutBlocks, xcdPer;
    int headMajor;            // HEADS: workgroups ordered head-major (all pixel tiles of head 0, then head 1, ..)
    int NT;                   // CoutPad / 16
    int nchunks;              // Cin / 64
    EpiArgs epi;
    // HEADS instances (ct_heads_fused): the workgroup's 256 couts are the hidden layer of head `cb`; its 1x1 output
    // layer is applied in the epilogue and only the head's <= 8 channels are written (NCHW)
    const float *hw2;         // [heads][8][256] output-layer weights (rows past the head's channels: zero)
    const float *hb2;         // [heads][8]
    float *hout;              // [N, hctot, H, W]
    int hctot;
    int hcout[CT_MAX_FUSED_HEADS], hcoff[CT_MAX_FUSED_HEADS];
};

// WM = m-tiles (blocks of 4 x 16 output pixels = 16 Winograd tiles) stacked vertically per workgroup
template <int WM>
struct WCfg {
    static constexpr int PH = 4 * WM + 2, PW = 18, PP = PH * PW;   // raw patch of a (4*WM) x 16 output block
    static constexpr int SLAB = PP * 16;                           // floats per 16-channel slab
    static constexpr int BUF = 4 * SLAB;                           // floats per 64-channel chunk
    static constexpr int ITEMS = 4 * PP * 4;                       // float4 items per chunk
    static constexpr int NR = (ITEMS + 255) / 256;
};
CT_DEFINE_STAMPS(wino)      // (tools/conv_phases.py; expands to nothing in the shipped build)

// B^T rows as (i1, i2, s2): V = d[i1] + s2 * d[i2]   (r=0: d0-d2, r=1: d1+d2, r=2: d2-d1, r=3: d1-d3)
__device__ __forceinline__ void bt_row(int r, int &i1, int &i2, float &s2)
{
    i1 = (r == 0) ? 0 : ((r == 2) ? 2 : 1);
    i2 = (r == 0) ? 2 : ((r == 1) ? 2 : ((r == 2) ? 1 : 3));
    s2 = (r == 1) ? 1.0f : -1.0f;
}

// MULTI = more than one 64-channel chunk (the next chunk's patch is prefetched into registers during the
// MFMAs); the single-chunk instance (Cin = 64: level 2, heads) needs ~40 VGPRs less -> one more wave per SIMD.
// KS = K-split inside the workgroup for the deep levels (32x32 / 16x16 maps have too few pixel tiles to fill
// the chip): 4*KS waves, wave (r, kp) contracts slabs kp*4/KS .. of every chunk for row r; the KS partial
// results are summed (in kp order) by the output transform.
// NB > 1 (single chunk = 64 input channels, no K split): the workgroup walks NB consecutive cout blocks with the
// same pixel tile.  The patch is staged and input-transformed ONCE (the 16 transformed fragments of a lane stay
// in 64 VGPRs), every further block costs only its B loads, MFMAs and output transform -- for a short-K layer
// like the heads conv (64 -> 1280) the per-block VALU work drops from ~6 to ~3 instructions per MFMA.
// HEADS (NB = 8, 32 couts per block: one workgroup = 64 pixels x the 256 hidden channels of ONE head, base_model.py:
// 24-65): conv3x3 + bias + ReLU never leaves the workgroup -- every lane multiplies the 4 hidden channels it holds
// after the output transform with the head's 1x1 weights (kept in LDS) and accumulates <= 8 output channels per
// pixel over the 8 blocks; the partial sums of the 4 channel quads (lanes) and the 2 n-tiles (waves) are added in a
// fixed order, then bias, sigmoid / depth transform (detector.py:300-308) and ONE NCHW store per output value.  The
// 256-channel intermediate (84 MB per frame at 512x512 with 5 heads) is neither written nor read back.
#ifndef CT_NB_WAVES
#define CT_NB_WAVES 3       // waves per SIMD the NB > 1 shapes (fused heads) are compiled for (variant builds: 4 = 128 VGPRs)
#endif
// The workgroup body, shared by the plain launch (wino_conv_kernel) and the grouped partial launch (wino_offsets_kernel):
// `bid` = workgroup index inside its layer, `nblocks` = workgroups of the layer, `chunk0` = first 64-channel chunk this
// workgroup contracts (the K-split-through-partial-maps form runs one chunk per workgroup), `ydst` = output base.
template <int WM, int WN, int KS, bool MULTI, int NB = 1, bool HEADS = false>
__device__ __forceinline__ void wino_body(const WinoArgs &a, int bid, const unsigned nblocks, const int chunk0, float *const ydst)
{
    static_assert(NB == 1 || (!MULTI && KS == 1), "NB > 1 is a single-chunk, unsplit shape");
    static_assert(!HEADS || (NB == 8 && WM == 1 && WN == 2), "the fused heads run on 64 px x 8 blocks of 32 couts");
    using C = WCfg<WM>;
    constexpr int NTHR = 256 * KS;
    constexpr int W_PW = C::PW, W_PP = C::PP, W_SLAB = C::SLAB, W_BUF = C::BUF;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    CT_STAMP_RT(0);
    CT_STAMP(1);
    CT_STAMP_HW(8);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave index: uniform, kept in an SGPR
    const int wave = wv & 3, kp = wv >> 2;
    const int li = lane & 15, lg = lane >> 4;

    int cb;
    if (HEADS && a.headMajor == 2) {
        // XCD-affine head-major order (round 6; knob "heads_order" = 2): workgroup ids are dealt to the 8 XCDs
        // round-robin (id % 8; observed, not promised: a speed matter only), so XCD x is handed the CONTIGUOUS eighth
        // [x * per, (x + 1) * per) of the head-major list -- at most two heads' Winograd weights (1 MB each) ever pass
        // through one L2 instead of all of them through every L2: FETCH_SIZE x 2 of the launch 47.9 -> 34.4 MB at one stream
        // (profiles/r06_m_fetch_heads_order{1,2}.txt; what remains is the feature map, read once per head with its halo), same
        // time within noise (profiles/r06_m_ab_switches.txt), bit-identical results.  (Also measured there and not kept: this
        // kernel's multi-chunk 64 x 32 shape capped at 168 VGPRs for a third wave per SIMD -- 4 spilled registers, no change.)
        const unsigned per = nblocks >> 3;
        const unsigned ubid = (unsigned)bid;
        const unsigned lin = ubid < per * 8u ? (ubid & 7u) * per + (ubid >> 3) : ubid;     // (the last nblocks % 8 keep their place)
        const int per_head = (int)(nblocks / (unsigned)a.coutBlocks);
        cb = (int)lin / per_head;
        bid = (int)lin - cb * per_head;
    } else if (HEADS && a.headMajor) {
        // every head has its own 1 MB of Winograd weights: with the head as the SLOWEST index the workgroups resident
        // at any time work on <= 3 heads (3 MB per XCD L2 of 4 MB) instead of cycling through all of them (5.2 MB for
        // the five MOT heads: every XCD kept re-streaming the weights, 119 MB of fetches per launch in round 2)
        const int per_head = (int)(nblocks / (unsigned)a.coutBlocks);
        cb = bid / per_head;
        bid -= cb * per_head;
    } else {
        cb = ct_block_cout(bid, a.coutBlocks, a.xcdPer);
    }
    const int tx = bid % a.tilesX; bid /= a.tilesX;
    const int ty = bid % a.tilesY; bid /= a.tilesY;
    const int n = bid;
    const int oy0 = ty * 4 * WM, ox0 = tx * 16;
    const float *xin = a.x + (size_t)n * a.H * a.W * a.ldx;

    // ---- staging of the raw patch (rows oy0-1 .. oy0+4*WM, cols ox0-1 .. ox0+16) -------------------------
    // thread -> (channel quad q = tid & 3, slot tid >> 2): a slot is a patch pixel P (+ SP per round) of the 16-channel slab
    // kk0 (+ KG per round).  Everything that depends on the pixel -- its row / column, whether it lies inside the image, its
    // swizzled LDS address -- is computed ONCE per pixel round (one or two of them); the slab is an immediate offset of the
    // loads and the LDS stores (round 6: the item list it replaces cost two integer divisions and an address per item).
    constexpr int SLOTS = NTHR / 4;                   // 64 / 128 / 256
    constexpr int KG = SLOTS >= 256 ? 2 : 1;          // slab groups side by side in the thread block
    constexpr int SP = SLOTS / KG;                    // pixel slots per round: 64 or 128
    constexpr int RP = (W_PP + SP - 1) / SP;          // pixel rounds
    constexpr int RK = 4 / KG;                        // slab rounds
    constexpr int W_NR = RP * RK;
    const int sq = tid & 3, sslot = tid >> 2;
    const int skk0 = sslot / SP, sP0 = sslot - skk0 * SP;
    int goff[RP], loff[RP];
#pragma unroll
    for (int j = 0; j < RP; ++j) {
        const int P = sP0 + SP * j;
        const int py = P / W_PW, px = P - py * W_PW;
        const int iy = oy0 - 1 + py, ix = ox0 - 1 + px;
        loff[j] = (P < W_PP) ? skk0 * W_SLAB + P * 16 + ((sq ^ ((P >> 1) & 2)) << 2) : -1;
        goff[j] = (P < W_PP && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) ? ((iy * a.W + ix) * a.ldx + skk0 * 16 + sq * 4) * 4 : (int)0x80000000;
    }
    // the patch comes through a buffer descriptor of this image: byte offset of the pixel in the vector offset, the chunk in
    // the scalar offset, the slab in the immediate; the padding pixels carry an out-of-range offset and the hardware returns
    // zeros for them -- no address arithmetic and no select per staged vector (fp32 MFMAs and vector instructions share the
    // SIMD's lanes on this part: every VALU instruction of the loop is paid on top of the MFMA time)
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(xin), 0, (int)((((unsigned)a.H * a.W - 1u) * a.ldx + a.Cin) * 4u), 0x00020000);
    f32x4 stage[W_NR];
    auto stage_load = [&](int chunk) {
        const int coff = (chunk0 + chunk) * 256;
#pragma unroll
        for (int k = 0; k < RK; ++k)
#pragma unroll
            for (int j = 0; j < RP; ++j)
                stage[k * RP + j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, goff[j] + k * (KG * 64), coff, 0));
    };
    auto stage_store = [&](int buf) {
        float *dst = lds + buf * W_BUF;
#pragma unroll
        for (int j = 0; j < RP; ++j)
            if (loff[j] >= 0) {
#pragma unroll
                for (int k = 0; k < RK; ++k) *reinterpret_cast<f32x4 *>(dst + loff[j] + k * (KG * W_SLAB)) = stage[k * RP + j];
            }
    };

    // ---- A side: lane li = Winograd tile (row li>>3, column li&7), lg = channel quad; wave = row r of the
    //      transformed tile.  Patch pixel of d[i][j] of this lane's tile: (2*trow + i, 2*tcol + j).
    int ri1, ri2;
    float rs2;
    bt_row(wave, ri1, ri2, rs2);
    const int trow = li >> 3, tcol = li & 7;
    // LDS float offsets (inside a slab) of d[i1][0..3], d[i2][0..3] of m-tile 0; m-tile mt adds 4*mt patch
    // rows = 72*mt pixels, an even multiple of 4 pixels, so the swizzle term ((P >> 1) & 2) is unchanged
    int pa[2][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int P1 = (2 * trow + ri1) * W_PW + 2 * tcol + j;
        const int P2 = (2 * trow + ri2) * W_PW + 2 * tcol + j;
        pa[0][j] = P1 * 16 + ((lg ^ ((P1 >> 1) & 2)) << 2);
        pa[1][j] = P2 * 16 + ((lg ^ ((P2 >> 1) & 2)) << 2);
    }
    constexpr int MT_OFF = 4 * W_PW * 16;            // floats between the patches of consecutive m-tiles

    // ---- B side ----------------------------------------------------------------------------------------
    const int NCH16 = a.Cin >> 4;
    const int slab_bytes = a.NT << 10;              // one (position, 16-channel slab): NT fragments of 1 KiB
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.up), 0, 16 * NCH16 * slab_bytes, 0x00020000);
    constexpr int SPK = 4 / KS;                     // slabs of a chunk per K part
    constexpr int S = 16 / KS, D = 3, R = 4;       // steps per chunk and wave, B prefetched 3 steps ahead, ring of 4
    float *exch = lds;                              // [kp KS][r 4][q 2][mt WM][nt WN][lane 64] float4
    constexpr int TN = WM * WN;
    float *ybuf = lds + KS * 4 * 2 * TN * 256;      // per-wave 32 x 16 transpose slabs behind the exchange buffer
    float *w2l = ybuf + 4 * KS * 32 * 16;           // HEADS: the head's [8][256] output-layer weights ...
    float *hred = w2l + 8 * 256;                    // ... and the n-tile-1 waves' partial sums [2][64][2][8]
    float hacc[HEADS ? 2 : 1][HEADS ? 8 : 1];       // HEADS: output channels of this lane's two pixels, its 4-channel quads
    if (HEADS) {
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
            for (int j = 0; j < 8; ++j) hacc[h2][j] = 0.0f;
    }
    const bool wide = (a.epi.Cout % 4 == 0) && (a.epi.ldy % 4 == 0) && (!a.epi.res || a.epi.ldr % 4 == 0) &&
                      (((uintptr_t)ydst & 15) == 0) && (!a.epi.res || ((uintptr_t)a.epi.res & 15) == 0);
    const int wr = kp * 4 + wave;                   // this wave's slot in the exchange buffer
    // epilogue operands of the (q, m-tile, n-tile) jobs this wave finalises (BN scale / shift of 4 couts), loaded before
    // the main loop (ct_common.h, ct_load_scale_shift: a ~1 us round trip off the end of every workgroup)
    constexpr int NJW = (2 * WM * WN + 4 * KS - 1) / (4 * KS);
    constexpr bool PRE = NB == 1 && !HEADS && !(KS == 4 && WN == 2);   // (64 x 32 x K-split 4: 1024 threads cap a wave at 128 VGPRs, already spilling)
    f32x4 psc[PRE ? NJW : 1], psh[PRE ? NJW : 1];
    if (PRE && wide) {
#pragma unroll
        for (int jw = 0; jw < NJW; ++jw) {
            const int job = min(jw * 4 * KS + wv, 2 * WM * WN - 1);
            const int tn = job >> 1;
            const int nt = tn - (tn / WN) * WN;
            const int c4 = (cb * WN + nt) * 16 + (lane & 3) * 4;
            const bool ok = c4 < a.epi.Cout;
            psc[jw] = (a.epi.scale && ok) ? *reinterpret_cast<const f32x4 *>(a.epi.scale + c4) : f32x4{1.f, 1.f, 1.f, 1.f};
            psh[jw] = (a.epi.shift && ok) ? *reinterpret_cast<const f32x4 *>(a.epi.shift + c4) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }

    // input transform of slab kk of the current LDS buffer: rows combined first, then the four column combinations
    auto transform = [&](f32x4 (&v)[WM][4], const float *buf, int kk) {
#pragma unroll
        for (int mt = 0; mt < WM; ++mt) {
            f32x4 e[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 d1 = *reinterpret_cast<const f32x4 *>(buf + kk * W_SLAB + mt * MT_OFF + pa[0][j]);
                const f32x4 d2 = *reinterpret_cast<const f32x4 *>(buf + kk * W_SLAB + mt * MT_OFF + pa[1][j]);
                e[j] = d1 + rs2 * d2;
            }
            v[mt][0] = e[0] - e[2];
            v[mt][1] = e[1] + e[2];
            v[mt][2] = e[2] - e[1];
            v[mt][3] = e[1] - e[3];
        }
    };

    f32x4 vall[NB > 1 ? 4 : 1][WM][4];             // NB > 1: all four slabs of the (single) chunk, transformed once
    if (NB > 1) {
        stage_load(0);
        CT_STAMP(2);
        stage_store(0);
        __syncthreads();
        CT_STAMP(3);
#pragma unroll
        for (int kk = 0; kk < (NB > 1 ? 4 : 1); ++kk) transform(vall[kk], lds, kk);
        __syncthreads();                            // the patch is dead from here on: its LDS becomes the exchange buffer
        if (HEADS) {                                // (visible to every wave after the first block's exchange barrier)
            const float *src = a.hw2 + (size_t)cb * 2048;
            *reinterpret_cast<f32x4 *>(w2l + tid * 4) = *reinterpret_cast<const f32x4 *>(src + tid * 4);
            *reinterpret_cast<f32x4 *>(w2l + 1024 + tid * 4) = *reinterpret_cast<const f32x4 *>(src + 1024 + tid * 4);
        }
    }

#pragma unroll 1
    for (int nb = 0; nb < NB; ++nb) {
    const int nt0 = (cb * NB + nb) * WN;
    if (NB > 1 && nt0 >= a.NT) break;               // (uniform: cout blocks past the end of a ragged Cout)
    int bvo[WN];                                    // the lane's part of a fragment address: constant over the steps
#pragma unroll
    for (int nt = 0; nt < WN; ++nt) bvo[nt] = (min(nt0 + nt, a.NT - 1) << 10) + (lane << 4);
    // step s of a chunk (for this wave) = (slab kp*SPK + (s>>2), column c = s&3); position = wave*4 + c
    auto load_b = [&](f32x4 (&b)[WN], int chunk, int s) {
        const int c = s & 3, kk = kp * SPK + (s >> 2);
        const int so = ((wave * 4 + c) * NCH16 + (chunk0 + min(chunk, a.nchunks - 1)) * 4 + kk) * slab_bytes;     // (uniform)
#pragma unroll
        for (int nt = 0; nt < WN; ++nt) b[nt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, bvo[nt], so, 0));
    };

    f32x4 acc[WM][4][WN];                          // [m-tile][column c of the transformed tile][n-tile]
#pragma unroll
    for (int mt = 0; mt < WM; ++mt)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int nt = 0; nt < WN; ++nt) acc[mt][c][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (NB > 1) {
        f32x4 breg[R][WN];
#pragma unroll
        for (int p = 0; p < D; ++p) load_b(breg[p], 0, p);
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            load_b(breg[(s + D) % R], 0, (s + D) & 15);          // (the last D prefetches re-read valid slabs)
            __builtin_amdgcn_sched_barrier(0x386);
#pragma unroll
            for (int ee = 0; ee < 4; ++ee)
#pragma unroll
                for (int mt = 0; mt < WM; ++mt)
#pragma unroll
                    for (int nt = 0; nt < WN; ++nt)
                        acc[mt][s & 3][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vall[(NB > 1) ? (s >> 2) : 0][mt][s & 3][ee],
                                                                                 breg[s % R][nt][ee], acc[mt][s & 3][nt], 0, 0, 0);
        }
    } else {
        stage_load(0);
        f32x4 breg[R][WN];
#pragma unroll
        for (int p = 0; p < D; ++p) load_b(breg[p], p / S, p % S);
        CT_STAMP(2);
        stage_store(0);
        __syncthreads();
        CT_STAMP(3);
        const int nchunks = MULTI ? a.nchunks : 1;
        for (int ch = 0; ch < nchunks; ++ch) {
            const int cur = ch & 1;
            if (ch < 2) CT_STAMP(12 + 6 * ch);          // (loop-internal stamps of the first two chunks: tools/conv_phases.py --loop)
            if (MULTI) {
                stage_load(min(ch + 1, a.nchunks - 1));
                __builtin_amdgcn_sched_barrier(0x386);
            }
            if (ch < 2) CT_STAMP(13 + 6 * ch);
            const float *buf = lds + cur * W_BUF;
#pragma unroll
            for (int ks = 0; ks < SPK; ++ks) {
                const int kk = kp * SPK + ks;
                f32x4 v[WM][4];
                transform(v, buf, kk);
                if (ch < 2 && ks == 0) CT_STAMP(14 + 6 * ch);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int s = ks * 4 + c;
                    const int sp = s + D;
                    load_b(breg[(s + D) % R], ch + sp / S, sp % S);
                    __builtin_amdgcn_sched_barrier(0x386);
#pragma unroll
                    for (int ee = 0; ee < 4; ++ee)
#pragma unroll
                        for (int mt = 0; mt < WM; ++mt)
#pragma unroll
                            for (int nt = 0; nt < WN; ++nt)
                                acc[mt][c][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[mt][c][ee], breg[s % R][nt][ee],
                                                                                     acc[mt][c][nt], 0, 0, 0);
                }
            }
            if (ch < 2) CT_STAMP(15 + 6 * ch);
            if (MULTI && ch + 1 < a.nchunks) stage_store(cur ^ 1);
            if (ch < 2) CT_STAMP(16 + 6 * ch);
            __syncthreads();
            if (ch < 2) CT_STAMP(17 + 6 * ch);
        }
    }

    if (NB == 1) CT_STAMP(4);
    // ---- output transform: columns in registers ... ---------------------------------------------------
    // T[q][nt]: q=0: M0+M1+M2, q=1: M1-M2-M3   (this wave's row r)
#pragma unroll
    for (int mt = 0; mt < WM; ++mt)
#pragma unroll
        for (int nt = 0; nt < WN; ++nt) {
            const f32x4 t0 = acc[mt][0][nt] + acc[mt][1][nt] + acc[mt][2][nt];
            const f32x4 t1 = acc[mt][1][nt] - acc[mt][2][nt] - acc[mt][3][nt];
            *reinterpret_cast<f32x4 *>(exch + (((wr * 2 + 0) * TN + mt * WN + nt) * 64 + lane) * 4) = t0;
            *reinterpret_cast<f32x4 *>(exch + (((wr * 2 + 1) * TN + mt * WN + nt) * 64 + lane) * 4) = t1;
        }
    __syncthreads();
    if (NB == 1) CT_STAMP(5);
    // ... rows across the waves: Y[0][q] = T0+T1+T2, Y[1][q] = T1-T2-T3; 2*WM*WN (q, mt, nt) jobs over 4 waves
#pragma unroll
    for (int w0 = 0; w0 < 2 * TN; w0 += 4 * KS) {
        const int job = w0 + wv;
        if (job < 2 * TN) {
            const int q = job & 1, tn = job >> 1;
            const int mt = tn / WN, nt = tn - mt * WN;
            f32x4 t[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                t[r] = *reinterpret_cast<const f32x4 *>(exch + (((r * 2 + q) * TN + tn) * 64 + lane) * 4);
#pragma unroll
                for (int k = 1; k < KS; ++k)        // K parts, in order
                    t[r] += *reinterpret_cast<const f32x4 *>(exch + ((((k * 4 + r) * 2 + q) * TN + tn) * 64 + lane) * 4);
            }
            const f32x4 y0 = t[0] + t[1] + t[2];
            const f32x4 y1 = t[1] - t[2] - t[3];
            const int co = (nt0 + nt) * 16 + li;
            if (HEADS) {
                float *yb = ybuf + wv * (32 * 16);
#pragma unroll
                for (int ee = 0; ee < 4; ++ee) {
                    const int tile = lg * 4 + ee;
                    yb[(tile * 2 + 0) * 16 + li] = y0[ee];
                    yb[(tile * 2 + 1) * 16 + li] = y1[ee];
                }
                __builtin_amdgcn_s_waitcnt(0xc07f);               // lgkmcnt(0): the wave's own LDS writes have landed
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    const int item = lane + 64 * h2;              // (tile*2 + p) * 4 + cout quad
                    const int cq = item & 3, tp = item >> 2;
                    const int c4 = (nt0 + nt) * 16 + cq * 4;      // hidden channel (global); local = c4 - cb * 256
                    const f32x4 raw = *reinterpret_cast<const f32x4 *>(yb + tp * 16 + cq * 4);
                    const f32x4 sh4 = *reinterpret_cast<const f32x4 *>(a.epi.shift + c4);
                    f32x4 o;
#pragma unroll
                    for (int i = 0; i < 4; ++i) o[i] = fmaxf(raw[i] + sh4[i], 0.0f);      // + bias, ReLU
                    const float *wrow = w2l + (c4 - cb * 256);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const f32x4 w = *reinterpret_cast<const f32x4 *>(wrow + j * 256);
                        hacc[h2][j] += (o[0] * w[0] + o[1] * w[1]) + (o[2] * w[2] + o[3] * w[3]);
                    }
                }
            } else if (wide) {
                // 32 pixels x 16 couts of this job: transpose through this wave's private LDS slab so that a lane
                // stores 4 consecutive couts of one pixel as ONE 16-byte store (the epilogue of a short-K layer
                // like the heads conv is store-issue bound: 8 dword stores per lane become 2 dwordx4 stores)
                float *yb = ybuf + wv * (32 * 16);
#pragma unroll
                for (int ee = 0; ee < 4; ++ee) {
                    const int tile = lg * 4 + ee;
                    yb[(tile * 2 + 0) * 16 + li] = y0[ee];
                    yb[(tile * 2 + 1) * 16 + li] = y1[ee];
                }
                __builtin_amdgcn_s_waitcnt(0xc07f);               // lgkmcnt(0): the wave's own LDS writes have landed
                // (round 6) stores and residual loads through buffer descriptors of image n: the lane's part of the offset
                // (its pixel inside the 4 x 16 block, its cout quad) is computed once per item, the block / job part is a
                // scalar offset; lanes past the map's edge or the last cout carry an out-of-range offset (store dropped,
                // residual read as zero).  The item-by-item 64-bit address arithmetic this replaces was a third of the
                // kernel's vector instructions on the short-K layers.
                const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(
                    ydst + (size_t)n * a.epi.Ho * a.epi.Wo * a.epi.ldy, 0,
                    (int)((((unsigned)a.epi.Ho * a.epi.Wo - 1u) * a.epi.ldy + a.epi.Cout) * 4u), 0x00020000);
                const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<float *>(a.epi.res ? a.epi.res + (size_t)n * a.epi.Ho * a.epi.Wo * a.epi.ldr : ydst), 0,
                    a.epi.res ? (int)((((unsigned)a.epi.Ho * a.epi.Wo - 1u) * a.epi.ldr + a.epi.Cout) * 4u) : 0, 0x00020000);
                const int srow = oy0 + 4 * mt, scol = ox0 + q, sc0 = (nt0 + nt) * 16;     // (uniform)
                const int so_y = ((srow * a.epi.Wo + scol) * a.epi.ldy + sc0) * 4;
                const int so_r = ((srow * a.epi.Wo + scol) * a.epi.ldr + sc0) * 4;
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    const int item = lane + 64 * h2;              // (tile*2 + p) * 4 + cout quad
                    const int cq = item & 3, tp = item >> 2;
                    const int tile = tp >> 1, pp = tp & 1;
                    const int lrow = 2 * (tile >> 3) + pp, lcol = 2 * (tile & 7);
                    const bool ok = lrow < a.epi.Ho - srow && lcol < a.epi.Wo - scol && cq * 4 < a.epi.Cout - sc0;
                    const int lpix = lrow * a.epi.Wo + lcol;
                    const int vo_y = ok ? (lpix * a.epi.ldy + cq * 4) * 4 : (int)0x80000000;
                    const f32x4 raw = *reinterpret_cast<const f32x4 *>(yb + tp * 16 + cq * 4);
                    f32x4 sc4, sh4;
                    if (PRE) {
                        sc4 = psc[PRE ? w0 / (4 * KS) : 0];
                        sh4 = psh[PRE ? w0 / (4 * KS) : 0];
                    } else {
                        const int c4 = min(sc0 + cq * 4, a.epi.Cout - 4);      // (clamped: lanes past the end store nothing)
                        sc4 = a.epi.scale ? *reinterpret_cast<const f32x4 *>(a.epi.scale + c4) : f32x4{1.f, 1.f, 1.f, 1.f};
                        sh4 = a.epi.shift ? *reinterpret_cast<const f32x4 *>(a.epi.shift + c4) : f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                    f32x4 r4 = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (a.epi.res)                                  // (uniform)
                        r4 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                 rrs, ok ? (lpix * a.epi.ldr + cq * 4) * 4 : (int)0x80000000, so_r, 0));
                    f32x4 o;
#pragma unroll
                    for (int i = 0; i < 4; ++i) o[i] = ct_epilogue_plain(a.epi, raw[i], sc4[i], sh4[i], r4[i]);
                    // (the block offset goes into the VECTOR offset, the scalar offset stays the constant 0: hipcc pads the
                    //  "VALU overwrites the data registers of a store wider than 64 bits" hazard only for stores WITHOUT a scalar
                    //  offset register -- the published exemption -- and gfx950 does not honour the exemption: with `so_y` in the
                    //  scalar field the Winograd OFFSETS kernel stored `2 | pp` (the next instruction's result) instead of o[0]
                    //  in lanes 12-15 of every 16, once per ~15 launches of a 4-stream plan; tools/determinism.py,
                    //  profiles/r06_an_store_hazard.txt, tests/test_hip_determinism.py)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), yrs, vo_y + so_y, 0, 0);
                }
            } else if (co < a.epi.Cout) {
                const float sc = a.epi.scale ? a.epi.scale[co] : 1.0f;
                const float sh = a.epi.shift ? a.epi.shift[co] : 0.0f;
#pragma unroll
                for (int ee = 0; ee < 4; ++ee) {
                    const int tile = lg * 4 + ee;
                    const int ox = ox0 + 2 * (tile & 7) + q;
#pragma unroll
                    for (int p = 0; p < 2; ++p) {
                        const int oy = oy0 + 4 * mt + 2 * (tile >> 3) + p;
                        if (oy < a.epi.Ho && ox < a.epi.Wo) {
                            const size_t pix = ((size_t)n * a.epi.Ho + oy) * a.epi.Wo + ox;
                            const float r = a.epi.res ? a.epi.res[pix * a.epi.ldr + co] : 0.0f;
                            ydst[pix * a.epi.ldy + co] = ct_epilogue_plain(a.epi, p ? y1[ee] : y0[ee], sc, sh, r);
                        }
                    }
                }
            }
        }
    }
    if (NB > 1) __syncthreads();                    // every job has read the exchange buffer: the next block may overwrite it
    }                                               // (cout block nb)
    if (NB > 1) CT_STAMP(5);
    if (!HEADS) { CT_STAMP(6); CT_STAMP_RT(7); }
    if (HEADS) {
        // wave wv handled (q = wv & 1, n-tile wv >> 1) of every block; lane: pixels tp = (lane + 64 h2) >> 2, quad lane & 3
        const int q = wv & 1, ntile = wv >> 1;
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float v = hacc[h2][j];
                v += __shfl_xor(v, 1);              // the 4 channel quads of a pixel sit in 4 neighbouring lanes
                v += __shfl_xor(v, 2);
                hacc[h2][j] = v;
            }
        if (ntile == 1) {
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
                for (int j = 0; j < 8; ++j) hred[((q * 64 + lane) * 2 + h2) * 8 + j] = hacc[h2][j];
        }
        __syncthreads();
        if (ntile == 0 && (lane & 3) == 0) {
            const int hc = a.hcout[cb], g0 = a.hcoff[cb];
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                const int tp = (lane + 64 * h2) >> 2;
                const int tile = tp >> 1, pp = tp & 1;
                const int oy = oy0 + 2 * (tile >> 3) + pp;
                const int ox = ox0 + 2 * (tile & 7) + q;
                if (oy >= a.epi.Ho || ox >= a.epi.Wo) continue;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (j >= hc) break;
                    const int gc = g0 + j;
                    const float v = (hacc[h2][j] + hred[((q * 64 + lane) * 2 + h2) * 8 + j]) + a.hb2[cb * 8 + j];
                    a.hout[(((size_t)n * a.hctot + gc) * a.epi.Ho + oy) * a.epi.Wo + ox] =
                        ct_epilogue_value(a.epi, v, gc, 1.0f, 0.0f, 0.0f);
                }
            }
        }
    }
    if (HEADS) { CT_STAMP(6); CT_STAMP_RT(7); }
}

template <int WM, int WN, int KS, bool MULTI, int NB = 1, bool HEADS = false>
// (the single-chunk 256-thread shapes need 136 VGPRs unconstrained: capped at 128 = 4 waves per SIMD, no spills)
__global__ __launch_bounds__(256 * KS)
__attribute__((amdgpu_waves_per_eu(NB > 1 ? CT_NB_WAVES : ((!MULTI && KS == 1 && WM * WN <= 2) ? 4 : KS))))
void wino_conv_kernel(WinoArgs a)
{
    wino_body<WM, WN, KS, MULTI, NB, HEADS>(a, (int)blockIdx.x, gridDim.x, 0, a.epi.y);
}

// Grouped PARTIAL launch (round 6; ct_dcn_v2_group, phase CT_DCN_OFFSETS, layers with ct_dcn_desc.w_off_winograd): the
// offset/mask convs (DCN.conv_offset_mask, 3x3 Cin -> 27; dla.py:513 -> upstream dcn_v2.py) of up to four independent
// layers in ONE launch, K-split over 64-channel chunks through partial maps: one workgroup = one 64-pixel Winograd block
// x ONE chunk of ONE layer, whatever Cin -- 16 MFMA steps each, so the workgroups of a 512-channel layer at 16 x 16 and of a
// 64-channel layer at 128 x 128 take the same time and a slot's convs fill the chip together.  Raw sums (no bias, no
// sigmoid: the DCN launch applies both while it sums the chunks, fuse_offset == 2), channel pitch 32 (the packed
// weights are zero past channel 27).  At 4 streams the per-layer conv launches this replaces cost 10-18 us each (16 per
// frame batch: 218 of the 952 us of the DCN sequence, profiles/r06_a_dcn_slots_b4.txt).
struct WinoGroup {
    WinoArgs p[4];
    int first[5];
    int tiles[4];                 // workgroups per chunk of layer i
    // XCD-aware order (as in the DCN MAIN launch, dcn_mfma.hip): ids go round-robin to the 8 XCDs; XCD x runs the chunks
    // c = x mod sx[i] only (sx = gcd(Cin / 64, 8): the chunk's weights and input channels pass through ONE L2) and, of the
    // 8 / sx XCDs sharing a chunk set, `pband[i]` consecutive pixel blocks each.  first[i] is a multiple of 8.
    int sx[4], pband[4];
    int n;
};

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) void wino_offsets_kernel(WinoGroup g)
{
    int bid = blockIdx.x;
    int pi = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i)
        if (i < g.n && bid >= g.first[i]) pi = i;
    const WinoArgs a = g.p[pi];       // (by value: bursts of wide scalar loads instead of dependent reloads)
    bid -= g.first[pi];
    const int tiles = g.tiles[pi];
    const int sx = g.sx[pi], sper = (a.Cin >> 6) / sx;
    const int xcd = bid & 7;
    int j = bid >> 3;
    const int sh = j % sper; j /= sper;
    const int split = (xcd & (sx - 1)) + sx * sh;
    bid = (xcd / sx) * g.pband[pi] + j;
    if (j >= g.pband[pi] || bid >= tiles) return;             // (uniform: padding of the id range)
    float *ydst = a.epi.y + (size_t)split * a.N * a.H * a.W * 32;
    wino_body<1, 2, 1, false>(a, bid, (unsigned)tiles, split, ydst);
}

// U[pos][co][ci] = (G g G^T)[r][c], G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]
__global__ __launch_bounds__(256) void pack_winograd_kernel(const float *w, float *p, int Cout, int Cin, int NT)
{
    const size_t total = (size_t)16 * (Cin >> 4) * NT * 256;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int e = idx & 3, j = (idx >> 2) & 15, g = (idx >> 6) & 3;
    size_t t = idx >> 8;
    const int nt = t % NT; t /= NT;
    const int c16 = t % (Cin >> 4);
    const int pos = (int)(t / (Cin >> 4));
    const int co = nt * 16 + j, ci = c16 * 16 + 4 * g + e;
    float u = 0.0f;
    if (co < Cout) {
        const float G[4][3] = {{1.f, 0.f, 0.f}, {.5f, .5f, .5f}, {.5f, -.5f, .5f}, {0.f, 0.f, 1.f}};
        const int r = pos >> 2, c = pos & 3;
        const float *g3 = w + ((size_t)co * Cin + ci) * 9;
        for (int aa = 0; aa < 3; ++aa)
            for (int bb = 0; bb < 3; ++bb) u += G[r][aa] * g3[aa * 3 + bb] * G[c][bb];
    }
    p[idx] = u;
}

template <int WM, int WN, int KS, bool MULTI, int NB = 1, bool HEADS = false>
int launch_wino2(const WinoArgs &a, dim3 grid, hipStream_t s)
{
    using C = WCfg<WM>;
    auto k = wino_conv_kernel<WM, WN, KS, MULTI, NB, HEADS>;
    const size_t patch = sizeof(float) * (size_t)C::BUF * (a.nchunks > 1 ? 2 : 1);
    const size_t exch = sizeof(float) * (size_t)(KS * 4 * 2 * WM * WN * 256 + 4 * KS * 32 * 16 + (HEADS ? 8 * 256 + 2 * 64 * 2 * 8 : 0));
    const size_t lds = patch > exch ? patch : exch;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(k, grid, dim3(256 * KS), lds, s, a);
    return CT_OK;
}
template <int WM, int WN, int KS>
int launch_wino(const WinoArgs &a, dim3 grid, hipStream_t s)
{
    if (KS > 1) return launch_wino2<WM, WN, KS, true>(a, grid, s);      // (K-split instances are for the deep levels)
    return a.nchunks > 1 ? launch_wino2<WM, WN, KS, true>(a, grid, s) : launch_wino2<WM, WN, KS, false>(a, grid, s);
}

}  // namespace

extern "C" size_t ct_packed_winograd_elems(int Cout, int Cin) { return (size_t)16 * (Cin / 16) * ct_cdiv(Cout, 16) * 256; }

extern "C" int ct_pack_winograd_weight(const float *w_oihw, float *packed, int Cout, int Cin, void *stream)
{
    if (!w_oihw || !packed) CT_FAIL_ARG("ct_pack_winograd_weight: null pointer");
    if (Cin % 16 || Cin <= 0 || Cout <= 0) CT_FAIL_ARG("ct_pack_winograd_weight: Cin=%d must be a multiple of 16", Cin);
    const size_t total = ct_packed_winograd_elems(Cout, Cin);
    hipLaunchKernelGGL(pack_winograd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       w_oihw, packed, Cout, Cin, ct_cdiv(Cout, 16));
    CT_CHECK_LAUNCH("ct_pack_winograd_weight");
    return CT_OK;
}

// called by ct_conv2d for algo 201..207
int ct_conv2d_winograd(const ct_conv_desc *d, void *stream)
{
    if (d->ks != 3 || d->stride != 1) CT_FAIL_ARG("ct_conv2d: the Winograd algo is for 3x3 stride-1 convolutions");
    if (!d->w_winograd) CT_FAIL_ARG("ct_conv2d: algo %d needs w_winograd (ct_pack_winograd_weight)", d->algo);
    if (d->Cin % 64) CT_FAIL_ARG("ct_conv2d: the Winograd algo needs Cin %% 64 == 0 (got %d)", d->Cin);
    if (d->flags & CT_OUT_NCHW) CT_FAIL_ARG("ct_conv2d: the Winograd algo writes NHWC only");
    if (d->sig_hi > d->sig_lo || d->dep_hi > d->dep_lo) CT_FAIL_ARG("ct_conv2d: the Winograd algo has no sigmoid epilogue");
    // algo 201: 64 px x 64 couts, 202: 64 x 32, 203: 128 x 32, 204: 128 x 16 per workgroup of 4 waves;
    // 205 / 206: 64 x 32 with K split over 2 / 4 wave groups (8 / 16 waves), 207: 64 x 16 with K split 4;
    // 208..211: 64 x 32 walking 2 / 4 / 5 / 8 cout blocks per workgroup on one input transform (Cin == 64 only)
    const int WM = (d->algo == 203 || d->algo == 204) ? 2 : 1;
    const int WN = (d->algo == 201) ? 4 : ((d->algo == 204 || d->algo == 207) ? 1 : 2);
    const int NB = d->algo == 208 ? 2 : (d->algo == 209 ? 4 : (d->algo == 210 ? 5 : (d->algo == 211 ? 8 : 1)));
    if (NB > 1 && d->Cin != 64) CT_FAIL_ARG("ct_conv2d: algo %d is for Cin == 64 (got %d)", d->algo, d->Cin);
    WinoArgs a;
    a.x = d->x; a.up = d->w_winograd;
    a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.ldx = d->ldx;
    a.tilesX = ct_cdiv(d->W, 16); a.tilesY = ct_cdiv(d->H, 4 * WM); a.coutBlocks = ct_cdiv(d->Cout, 16 * WN * NB); a.xcdPer = ct_xcd_per(a.coutBlocks); a.headMajor = 0;
    a.NT = ct_cdiv(d->Cout, 16); a.nchunks = d->Cin / 64;
    a.epi.scale = d->scale; a.epi.shift = d->shift; a.epi.res = d->res; a.epi.y = d->y;
    a.epi.ldr = d->ldr; a.epi.ldy = d->ldy; a.epi.Cout = d->Cout; a.epi.Ho = d->H; a.epi.Wo = d->W;
    a.epi.flags = d->flags; a.epi.sig_lo = a.epi.sig_hi = 0; a.epi.dep_lo = a.epi.dep_hi = 0; a.epi.depth_scale = 1.0f;
    const long blocks = (long)d->N * a.tilesX * a.tilesY * a.coutBlocks;
    if (blocks > 0x7fffffffL) CT_FAIL_ARG("ct_conv2d: grid too large");
    const dim3 grid((unsigned)blocks);
    hipStream_t st = (hipStream_t)stream;
    int rc;
    switch (d->algo) {
    case 201: rc = launch_wino<1, 4, 1>(a, grid, st); break;
    case 202: rc = launch_wino<1, 2, 1>(a, grid, st); break;
    case 203: rc = launch_wino<2, 2, 1>(a, grid, st); break;
    case 204: rc = launch_wino<2, 1, 1>(a, grid, st); break;
    case 205: rc = launch_wino<1, 2, 2>(a, grid, st); break;
    case 206: rc = launch_wino<1, 2, 4>(a, grid, st); break;
    case 208: rc = launch_wino2<1, 2, 1, false, 2>(a, grid, st); break;
    case 209: rc = launch_wino2<1, 2, 1, false, 4>(a, grid, st); break;
    case 210: rc = launch_wino2<1, 2, 1, false, 5>(a, grid, st); break;
    case 211: rc = launch_wino2<1, 2, 1, false, 8>(a, grid, st); break;
    default: rc = launch_wino<1, 1, 4>(a, grid, st); break;
    }
    CT_CHECK_LAUNCH("ct_conv2d(winograd)");
    return rc;
}

// called by ct_dcn_v2_group (dcn_mfma.hip), phase CT_DCN_OFFSETS, for the layers that carry Winograd offset weights
int ct_wino_offsets_group(const ct_wino_off_layer *L, int n, void *stream)
{
    if (n < 1 || n > 4) CT_FAIL_ARG("ct_wino_offsets_group: 1..4 layers");
    WinoGroup g;
    long blocks = 0;
    for (int i = 0; i < n; ++i) {
        const ct_wino_off_layer &l = L[i];
        if (!l.x || !l.w_winograd || !l.part || l.Cin % 64 || l.Cin <= 0 || l.ldx % 4 || ((uintptr_t)l.x & 15) || ((uintptr_t)l.part & 15))
            CT_FAIL_ARG("ct_wino_offsets_group: layer %d: bad arguments", i);
        WinoArgs &a = g.p[i];
        a.x = l.x; a.up = l.w_winograd;
        a.N = l.N; a.H = l.H; a.W = l.W; a.Cin = l.Cin; a.ldx = l.ldx;
        a.tilesX = ct_cdiv(l.W, 16); a.tilesY = ct_cdiv(l.H, 4); a.coutBlocks = 1; a.xcdPer = 0; a.headMajor = 0;
        a.NT = 2; a.nchunks = 1;
        a.epi.scale = nullptr; a.epi.shift = nullptr; a.epi.res = nullptr; a.epi.y = l.part;
        a.epi.ldr = 0; a.epi.ldy = 32; a.epi.Cout = 32; a.epi.Ho = l.H; a.epi.Wo = l.W;
        a.epi.flags = 0; a.epi.sig_lo = a.epi.sig_hi = 0; a.epi.dep_lo = a.epi.dep_hi = 0; a.epi.depth_scale = 1.0f;
        a.hw2 = nullptr; a.hb2 = nullptr; a.hout = nullptr; a.hctot = 0;
        for (int j = 0; j < CT_MAX_FUSED_HEADS; ++j) { a.hcout[j] = 0; a.hcoff[j] = 0; }
        g.tiles[i] = l.N * a.tilesX * a.tilesY;
        g.first[i] = (int)blocks;
        int sx = 1;
        while (sx < 8 && (l.Cin / 64) % (2 * sx) == 0) sx *= 2;
        g.sx[i] = sx;
        g.pband[i] = ct_cdiv(g.tiles[i], 8 / sx);
        blocks += 8L * (l.Cin / 64 / sx) * g.pband[i];
        if (blocks > 0x7fffffffL) CT_FAIL_ARG("ct_wino_offsets_group: grid too large");
    }
    for (int i = n; i <= 4; ++i) g.first[i] = (int)blocks;
    for (int i = n; i < 4; ++i) { g.p[i] = g.p[0]; g.tiles[i] = g.tiles[0]; g.sx[i] = g.sx[0]; g.pband[i] = g.pband[0]; }
    g.n = n;
    using C = WCfg<1>;
    const size_t patch = sizeof(float) * (size_t)C::BUF;
    const size_t exch = sizeof(float) * (size_t)(4 * 2 * 2 * 256 + 4 * 32 * 16);
    hipLaunchKernelGGL(wino_offsets_kernel, dim3((unsigned)blocks), dim3(256), patch > exch ? patch : exch, (hipStream_t)stream, g);
    CT_CHECK_LAUNCH("ct_dcn_v2(offset/mask convs, Winograd)");
    return CT_OK;
}

// The heads of the network in one launch (ct_heads_desc): conv3x3 64 -> 256 + bias + ReLU -> conv1x1 256 -> c + bias (+ the
// sigmoid / depth transform) for every listed head
extern "C" int ct_heads_fused(const ct_heads_desc *d, void *stream)
{
    if (!d || !d->x || !d->w0_winograd || !d->b0 || !d->w2 || !d->b2 || !d->out) CT_FAIL_ARG("ct_heads_fused: null pointer");
    if (d->Cin != 64) CT_FAIL_ARG("ct_heads_fused: the heads read a 64-channel feature map (got %d)", d->Cin);
    if (d->ldx % 4 || ((uintptr_t)d->x & 15)) CT_FAIL_ARG("ct_heads_fused: input view must be 16-byte aligned");
    if (d->nheads < 1 || d->nheads > CT_MAX_FUSED_HEADS) CT_FAIL_ARG("ct_heads_fused: 1..%d heads", CT_MAX_FUSED_HEADS);
    if (d->N <= 0 || d->H <= 0 || d->W <= 0 || d->ctot <= 0) CT_FAIL_ARG("ct_heads_fused: bad shape");
    if (((uintptr_t)d->b0 & 15) || ((uintptr_t)d->w2 & 15)) CT_FAIL_ARG("ct_heads_fused: b0 / w2 must be 16-byte aligned");
    WinoArgs a;
    a.x = d->x; a.up = d->w0_winograd;
    a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = 64; a.ldx = d->ldx;
    a.tilesX = ct_cdiv(d->W, 16); a.tilesY = ct_cdiv(d->H, 4); a.coutBlocks = d->nheads; a.xcdPer = 0;
    a.headMajor = ct_tune_get(CT_TUNE_HEADS_ORDER);
    a.NT = d->nheads * 16; a.nchunks = 1;
    a.epi.scale = nullptr; a.epi.shift = d->b0; a.epi.res = nullptr; a.epi.y = nullptr;
    a.epi.ldr = 0; a.epi.ldy = 0; a.epi.Cout = d->nheads * 256; a.epi.Ho = d->H; a.epi.Wo = d->W;
    a.epi.flags = 0; a.epi.sig_lo = d->sig_lo; a.epi.sig_hi = d->sig_hi; a.epi.dep_lo = d->dep_lo; a.epi.dep_hi = d->dep_hi;
    a.epi.depth_scale = d->depth_scale;
    a.hw2 = d->w2; a.hb2 = d->b2; a.hout = d->out; a.hctot = d->ctot;
    for (int i = 0; i < CT_MAX_FUSED_HEADS; ++i) { a.hcout[i] = 0; a.hcoff[i] = 0; }
    for (int i = 0; i < d->nheads; ++i) {
        if (d->cout[i] < 1 || d->cout[i] > 8 || d->coff[i] < 0 || d->coff[i] + d->cout[i] > d->ctot)
            CT_FAIL_ARG("ct_heads_fused: head %d: 1..8 channels inside [0, ctot)", i);
        a.hcout[i] = d->cout[i]; a.hcoff[i] = d->coff[i];
    }
    const long blocks = (long)d->N * a.tilesX * a.tilesY * a.coutBlocks;
    if (blocks > 0x7fffffffL) CT_FAIL_ARG("ct_heads_fused: grid too large");
    const int rc = launch_wino2<1, 2, 1, false, 8, true>(a, dim3((unsigned)blocks), (hipStream_t)stream);
    CT_CHECK_LAUNCH("ct_heads_fused");
    return rc;
}
