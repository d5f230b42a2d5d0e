// Modulated deformable convolution v2 (3x3, stride 1, pad 1, dilation 1, one deformable
// group) fused with its BatchNorm + ReLU, as ONE kernel: no im2col buffer ever reaches HBM.
//
//   y[p,co] = act( scale[co] * sum_{k,ci} W[co,ci,k] * m_k(p) * bilinear(x[.,ci], p+tap_k+d_k(p))
//                  + shift[co] )
//
// Layout choice (MI355X-first): activations are NHWC, so the four bilinear corners of a
// sampling point are four contiguous channel vectors -> 16-byte loads, and the bilinear
// weights / validity / mask of a (pixel, tap) are computed ONCE per workgroup (a 9-tap
// table in LDS) instead of once per input channel as the upstream NCHW im2col kernel does.
//
// One workgroup = 4 waves = 64 output pixels (4 rows x 16) x BN couts.  Loop over
// (channel chunk of 32, tap): every thread gathers 4 corners x 2 float4 for its
// (pixel, channel quad), blends them with the table weights and writes the A tile
// [2][64 px][16 ch] to LDS (double buffered, XOR-swizzled 16-B slots, conflict-free
// ds_read_b128 fragments); B fragments (packed weights, 1 KiB contiguous per wave load) go
// straight from L2 to VGPRs one step ahead; v_mfma_f32_16x16x4_f32 accumulates in fp32.
// The gather of step s+1 is in flight while the MFMAs of step s run.  Chunk-outer /
// tap-inner order keeps the 9 taps' footprint of one chunk in L1.
#include <type_traits>

#include "ct_common.h"
#include "ksplit_core.h"

namespace {

struct DcnArgs {
    const float *x;
    const float *om;
    const float *wp;
    int N, H, W, Cin, ldx, ldom;
    int tilesX, tilesY, coutBlocks;
    int NT;
    int nchunks, chunksPerSplit;     // chunks of 32 channels
    int tiles;                       // N * tilesY * tilesX * coutBlocks: workgroups per split
    int ptiles, splits;              // N * tilesY * tilesX pixel tiles; K splits (the persistent form strides over pixel tiles)
    // XCD-aware workgroup order (round 6; 0 = plain order): workgroup ids go round-robin to the 8 XCDs (id % 8) and each XCD has
    // its own L2.  XCD x contracts the K splits s = x mod xsx only (xsx = gcd(splits, 8): an eighth of a 512-channel layer's
    // weights and input channels per L2 instead of all of them) and, of the 8 / xsx XCDs that share a split set, each one
    // `pband` consecutive pixel tiles (neighbouring tiles share their halo rows in ONE L2).  The layer's id range is padded to
    // 8 * xper ids; ids past the last tile of a band exit.
    int xsx, xper, pband;
    float *ws;
    int wsCout;
    const float *w_off;              // fused offset/mask conv (FUSE kernels; per layer: nullptr = read `om`):
    const float *b_off;              //   packed 3x3 Cin -> 27 weights, bias
    // offset/mask conv split over 64-channel chunks (ct_dcn_desc::fuse_offset == 2): omPart = [omSplits][N*H*W][32] raw
    // partial sums, written by the CT_DCN_OFFSETS launch (offsOnly workgroups: one 32-pixel tile x one chunk each) and
    // summed (+ bias, mask sigmoid) by the MAIN launch while it builds its sampling table
    float *omPart;
    int omSplits;
    int offsOnly;
    EpiArgs epi;
};

// Several independent layers in ONE launch (ct_dcn_v2_group): workgroup ids [first[i], first[i+1]) belong to layer
// i.  At one stream most DCN layers of the network are 128 .. 512-workgroup problems on a 256-CU chip; the IDAUp
// tree has up to three of them ready at the same time (dla.py:539-574: every `proj` only needs a finished level).
constexpr int DCN_MAX_GROUP = 4;
struct DcnGroup {
    DcnArgs p[DCN_MAX_GROUP];
    int first[DCN_MAX_GROUP + 1];
    int n;
};

// BM = pixels per workgroup (64 = 4 rows x 16, 32 = 2 rows x 16); the 4 waves are a (BM/32) x (128/BM) grid of
// (pixel-row pairs) x (cout groups); wave tile = 2 m-tiles x WN n-tiles => BN = 16 * WN * 128 / BM couts.
// FUSE: the offset/mask conv of upstream's DCN.forward (conv_offset_mask + sigmoid of the mask channels) may be
// computed by the workgroup itself for its own pixels (per layer, when a.w_off is set: ksplit_conv_tile, result
// kept in LDS) instead of being read from a map another launch wrote: one launch and one HBM round trip of the
// 27-channel map less per layer.
// NKK = 16-channel slabs per step: 2 (32 channels x 1 tap, 16 * WN MFMAs per wave between barriers) or 4 (64 channels:
// 32 * WN MFMAs per barrier -- every step pays the same few hundred cycles of LDS round trips, waits and barrier skew, so
// the matrix pipe's share of a step grows with the work per barrier: measured 47 % busy at 16, 60 % at 32 MFMAs per step).
#ifndef CT_OFF_PD
#define CT_OFF_PD 2       // B prefetch distance of the offset/mask conv tiles (variant builds: tools/build_variant.py)
#endif
CT_DEFINE_STAMPS(dcn)       // (tools/dcn_phases.py; expands to nothing in the shipped build)
// Ablations of the MAIN loop (variant builds of tools/build_variant.py ONLY; results are wrong, the timing says which resource
// bounds a step): 1 = no corner loads (the gather path), 2 = no MFMAs, 4 = no weight loads in the loop, 8 = no barrier,
// 16 = no A-fragment reads from LDS
#ifndef CT_ABL
#define CT_ABL 0
#endif
// cache policy of the split-K partial / raw-tile stores (variant builds): 0 = plain, 16 = sc1 (write-through: the slab leaves the
// XCD's L2 at once instead of staying dirty until the kernel boundary writes it back, MI355X_MICROARCH.md "publish-large")
#ifndef CT_WS_AUX
#define CT_WS_AUX 0
#endif

// sigmoid of the mask channels (upstream dcn_v2.py: mask = torch.sigmoid(mask)): v_exp_f32 + v_rcp_f32, 1 ulp each -- the
// table build runs once per workgroup on the same SIMD lanes the MFMAs use, so its instruction count is kernel time
__device__ __forceinline__ float dcn_mask_sigmoid(float v) { return __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

// bilinear blend of the four corners' channel quads, w[c] = corner weight x mask: ((w0 c0 + w1 c1) + w2 c2) + w3 c3 per
// channel (the order of the scalar expression), written on channel PAIRS so that every operation is one packed
// instruction (v_pk_mul_f32 / v_pk_fma_f32: two lanes' worth of work in the issue slot of one) -- 8 instead of 12-13
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x4 dcn_blend(const f32x4 w, const f32x4 (&c)[4])
{
    // (written as instructions: left to the compiler the same expression comes out as 4 v_mul + 4 v_fma + 4 v_pk_fma; the
    //  weight pair (w0, w1) / (w2, w3) is one 64-bit operand whose low or high half op_sel broadcasts to both lanes)
    const f32x2 w01 = w.lo, w23 = w.hi;
    f32x2 lo, hi;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(lo) : "v"(w01), "v"(c[0].lo));
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(hi) : "v"(w01), "v"(c[0].hi));
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(lo) : "v"(w01), "v"(c[1].lo));
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(hi) : "v"(w01), "v"(c[1].hi));
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(lo) : "v"(w23), "v"(c[2].lo));
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(hi) : "v"(w23), "v"(c[2].hi));
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(lo) : "v"(w23), "v"(c[3].lo));
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(hi) : "v"(w23), "v"(c[3].hi));
    return f32x4{lo[0], lo[1], hi[0], hi[1]};
}

// One entry of the sampling table: (pixel, tap) -> 4 corner BYTE offsets + 4 weights (bilinear x mask; an invalid corner has
// offset 0 and weight 0).  fyk / fxk = the tap's undeformed position (float)(oy - 1 + ky), (float)(ox - 1 + kx); `in` = the
// pixel lies inside the map.  Round 6: written without branches and with 24-bit multiplies -- a workgroup builds 9 entries
// per pixel, all four waves of a SIMD do it at the same time (the dispatcher starts a launch's workgroups in rounds), and
// vector instructions are paid on top of the fp32 MFMA time on this part: the 7.6 k clocks of this phase (of 56 k in a
// 64 -> 64 workgroup's life at four streams, profiles/r06_q_dcn_phases_b4.txt) were instruction issue, not latency.
// Entries are stored TAP-major (index k * BM + m): thread `it` of a pass owns tap it / BM, pixel it % BM -- no division by 9.
__device__ __forceinline__ void dcn_tab_entry(float dy, float dx, float mk, float fyk, float fxk, bool in, int H, int W, int ldx4,
                                              int rowb, int4 &o, f32x4 &w)
{
    const float ys = fyk + dy, xs = fxk + dx;
    const float yf = floorf(ys), xf = floorf(xs);
    const int y0 = (int)yf, x0 = (int)xf;
    const float ly = ys - yf, lx = xs - xf, hy = 1.0f - ly, hx = 1.0f - lx;
    const bool ok = in & (ys > -1.0f) & (xs > -1.0f) & (ys < (float)H) & (xs < (float)W);
    const bool vy0 = ok & (y0 >= 0), vy1 = ok & (y0 + 1 <= H - 1), vx0 = x0 >= 0, vx1 = x0 + 1 <= W - 1;
    const int base = __mul24(__mul24(y0, W) + x0, ldx4);            // (only the valid corners use it)
    o.x = (vy0 & vx0) ? base : 0;
    o.y = (vy0 & vx1) ? base + ldx4 : 0;
    o.z = (vy1 & vx0) ? base + rowb : 0;
    o.w = (vy1 & vx1) ? base + rowb + ldx4 : 0;
    w[0] = (vy0 & vx0) ? hy * hx * mk : 0.0f;
    w[1] = (vy0 & vx1) ? hy * lx * mk : 0.0f;
    w[2] = (vy1 & vx0) ? ly * hx * mk : 0.0f;
    w[3] = (vy1 & vx1) ? ly * lx * mk : 0.0f;
}

// accumulators of one tile (WM rows of 16 pixels x WN tiles of 16 couts per wave) -> split-K partials / raw tiles (a.ws) or the
// finished output (BN + ReLU)
template <int WM, int WN>
__device__ __forceinline__ void dcn_store_acc(const DcnArgs &a, const f32x4 (&acc)[WM][WN], int n, int oy0, int ox0, int split, int nt0,
                                              int wm, int lane, const float (&psc)[WN], const float (&psh)[WN])
{
    const int li = lane & 15, lg = lane >> 4;
    if (a.ws) {
        // raw tiles / split-K partials: buffer stores -- the lane's (pixel, cout) offset once per n-tile, row and pixel steps as
        // scalar offsets; lanes past the map's edge or the padded Cout get an out-of-range offset the hardware drops
        const size_t Mtot = (size_t)a.N * a.H * a.W;
        float *wsp = a.ws + (size_t)split * Mtot * a.wsCout;
        const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(wsp, 0, (int)(Mtot * a.wsCout * 4), 0x00020000);
        const int pixb = a.wsCout * 4, rowb2 = a.W * pixb;
        const int oyb = oy0 + wm * WM;
        const int px0 = ox0 + lg * 4;
#pragma unroll
        for (int nt = 0; nt < WN; ++nt) {
            const int co = (nt0 + nt) * 16 + li;
            const int vbase = (((n * a.H + oyb) * a.W + px0) * a.wsCout + co) * 4;
            int vo[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) vo[e] = (co < a.wsCout && px0 + e < a.W) ? vbase : (int)0x80000000;
#pragma unroll
            for (int mt = 0; mt < WM; ++mt) {
                if (oyb + mt >= a.H) continue;                      // (uniform)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[mt][nt][e]), srs, vo[e], mt * rowb2 + e * pixb, CT_WS_AUX);
            }
        }
    } else {
        // finished output: BN + ReLU (the DCN entry points take no other epilogue), stored like the partials above
        const EpiArgs &e = a.epi;
        const unsigned img = (unsigned)a.H * a.W;
        const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(
            e.y + (size_t)n * img * e.ldy, 0, (int)(((img - 1u) * e.ldy + e.Cout) * 4u), 0x00020000);
        const int pixb = e.ldy * 4, rowb2 = a.W * pixb;
        const int oyb = oy0 + wm * WM;
        const int px0 = ox0 + lg * 4;
#pragma unroll
        for (int nt = 0; nt < WN; ++nt) {
            const int co = (nt0 + nt) * 16 + li;
            const int vbase = (__mul24(__mul24(oyb, a.W) + px0, e.ldy) + co) * 4;
            int vo[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) vo[i] = (co < e.Cout && px0 + i < a.W) ? vbase : (int)0x80000000;
#pragma unroll
            for (int mt = 0; mt < WM; ++mt) {
                if (oyb + mt >= a.H) continue;                      // (uniform)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(ct_epilogue_plain(e, acc[mt][nt][i], psc[nt], psh[nt], 0.0f)),
                                                          yrs, vo[i], mt * rowb2 + i * pixb, 0);
            }
        }
    }
}

template <int BM, int WN, bool FUSE, int NKK = 2>
__global__ __launch_bounds__(256) void dcn_mfma_kernel(DcnGroup g)
{
    CT_STAMP_RT(0);
    CT_STAMP(1);
    constexpr int WM = 2;
    constexpr int WGM = BM / 32, WGN = 4 / WGM;
    constexpr int ROWS = BM / 16;
    constexpr int SLAB = BM * 16;            // floats
    constexpr int BUF = NKK * SLAB;
    constexpr int NTHR = 256;
    static_assert(!FUSE || BM == 32, "the fused offset conv works on 32-pixel tiles");
    static_assert(NKK == 2 || NKK == 4, "32- or 64-channel steps");
    constexpr int UPC2 = NKK / 2;            // 32-channel chunks (the split-K bookkeeping unit) per step unit
    // dynamic LDS: [ om tile float[BM*32] (FUSE) ] | A tile double buffer | table offsets int4[BM*9] | table
    // weights float4[BM*9]; the offset-conv stage's scratch aliases everything after the om tile
    extern __shared__ __attribute__((aligned(16))) float dlds[];
    float *om_lds = dlds;
    float *lds_a = dlds + (FUSE ? BM * 32 : 0);
    int *tab_off = reinterpret_cast<int *>(lds_a + 2 * BUF);
    float *tab_w = lds_a + 2 * BUF + BM * 9 * 4;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // uniform: kept in an SGPR
    const int tl = tid;
    const int wm = wave / WGN, wn = wave % WGN;

    // ---- which layer of the group, which tile, which K split --------------------------------------
    int bid = blockIdx.x;
    int pi = 0;
#pragma unroll
    for (int i = 1; i < DCN_MAX_GROUP; ++i)
        if (i < g.n && bid >= g.first[i]) pi = i;
    const DcnArgs a = g.p[pi];       // (by value: one burst of wide scalar loads instead of ~17 dependent reloads before the loop)
    bid -= g.first[pi];
    int split, cb;
    if (a.xsx && !(FUSE && a.offsOnly)) {
        const int xcd = bid & 7;
        int j = bid >> 3;
        cb = j % a.coutBlocks; j /= a.coutBlocks;
        const int sper = a.splits / a.xsx;
        const int sh = j % sper; j /= sper;
        split = (xcd & (a.xsx - 1)) + a.xsx * sh;
        const int tile = (xcd / a.xsx) * a.pband + j;
        if (j >= a.pband || tile >= a.ptiles) return;       // (uniform: padding of the id range)
        bid = tile;
    } else {
        split = bid / a.tiles;
        bid -= split * a.tiles;
        cb = bid % a.coutBlocks; bid /= a.coutBlocks;
    }
    const int tx = bid % a.tilesX; bid /= a.tilesX;
    const int ty = bid % a.tilesY; bid /= a.tilesY;
    const int n = bid;
    const int oy0 = ty * ROWS, ox0 = tx * 16;
    // units of 16 * NKK channels (the host keeps chunksPerSplit a multiple of UPC2)
    const int nunits = a.nchunks / UPC2;
    const int c_begin = split * (a.chunksPerSplit / UPC2);
    const int c_end = min(nunits, c_begin + a.chunksPerSplit / UPC2);

    const float *xin = a.x + (size_t)n * a.H * a.W * a.ldx;
    if (FUSE && a.offsOnly) {
        // ---- CT_DCN_OFFSETS: chunk `split` (64 input channels) of the offset/mask conv of this tile, raw ----
        float *part = a.omPart + ((size_t)split * a.N + n) * a.H * a.W * 32;
        auto fin = [&](int mt, int nt, f32x4 sum, int) {
            const int co = nt * 16 + (lane & 15);
            const int oy = oy0 + mt;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ox = ox0 + (lane >> 4) * 4 + e;
                if (oy < a.H && ox < a.W) part[((size_t)oy * a.W + ox) * 32 + co] = sum[e];
            }
        };
        ksplit_conv_tile<3, 1, 2, 2, 4, CT_OFF_PD>(xin, a.H, a.W, a.ldx, a.Cin, a.w_off, 2, 0, oy0, ox0, split, split + 1, lds_a, fin);
        return;
    }
    const bool fuse = FUSE && a.w_off != nullptr;                // (uniform)
    const bool parts = a.omSplits > 0;                           // (uniform)
    const float *omn = (fuse || parts) ? nullptr : a.om + (size_t)n * a.H * a.W * a.ldom;

    if (fuse) {
        // ---- offset / mask conv of this tile (all input channels, whatever K range this split owns) ----
        auto fin = [&](int mt, int nt, f32x4 sum, int) {
            const int co = nt * 16 + (lane & 15);
            const float b = (co < 27) ? a.b_off[co] : 0.0f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = sum[e] + b;
                if (co >= 18) v = dcn_mask_sigmoid(v);
                om_lds[(mt * 16 + (lane >> 4) * 4 + e) * 32 + co] = v;
            }
        };
        ksplit_conv_tile<3, 1, 2, 2, 4, CT_OFF_PD>(xin, a.H, a.W, a.ldx, a.Cin, a.w_off, 2, 0, oy0, ox0, 0, a.Cin >> 6, lds_a, fin);
        __syncthreads();
    }
    CT_STAMP(2);
    // ---- B fragment addressing (set up first so that the weights of step 0 are in flight while
    //      the sampling table is built) --------------------------------------------------------
    const int li = lane & 15, lg = lane >> 4;
    const int nt0 = cb * (WGN * WN) + wn * WN;
    const int NCH16 = a.Cin >> 4;
    // B fragment loads through a buffer descriptor (round 6): the per-lane part of the address is a constant VGPR (n-tiles past
    // the padded Cout clamp to the last valid tile), the (tap, slab) part an SGPR offset -- no vector instruction per step.
    // On this part fp32 MFMAs and vector instructions do not overlap on a SIMD (profiles/r06_b_dcn_loop_ablations_b8.txt:
    // the launch takes MFMA time PLUS the time of everything else), so every VALU instruction of a step is paid in full.
    const int slab_bytes = a.NT << 10;                          // one (tap, 16-channel slab): NT fragments of 1 KiB
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.wp), 0, 9 * NCH16 * slab_bytes, 0x00020000);
    int bvo[WN];
#pragma unroll
    for (int nt = 0; nt < WN; ++nt) bvo[nt] = (min(nt0 + nt, a.NT - 1) << 10) + (lane << 4);
    auto load_b = [&](f32x4 (&b)[NKK][WN], int chunk, int tap) {
        const int so = (tap * NCH16 + chunk * NKK) * slab_bytes;          // (uniform)
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk)
#pragma unroll
            for (int nt = 0; nt < WN; ++nt)
                b[kk][nt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, bvo[nt], so + kk * slab_bytes, 0));
    };
    // (chunk, tap) of a step index (chunk-outer / tap-inner order keeps the 9 taps' footprint of a chunk in L1),
    // clamped to the last valid step (extra fetches are never used)
    const int nsteps = (c_end - c_begin) * 9;
    const int nmax = nsteps;
    auto step_ct = [&](int st, int &chunk, int &tap) {
        const int u = min(max(st, 0), max(nsteps - 1, 0));
        const int c = u / 9;
        chunk = min(c_begin + c, nunits - 1);
        tap = u - c * 9;
    };
    f32x4 bq[2][NKK][WN];
    {
        int ch, tp;
        step_ct(0, ch, tp);
        load_b(bq[0], ch, tp);
    }
    // epilogue operands (BN scale / shift) of this wave's couts: loaded here, needed after the last step
    float psc[WN], psh[WN];
#pragma unroll
    for (int nt = 0; nt < WN; ++nt) {
        psc[nt] = 1.0f; psh[nt] = 0.0f;
        if (!a.ws) ct_load_scale_shift(a.epi, (nt0 + nt) * 16, lane, psc[nt], psh[nt]);
    }

    // ---- sampling table: (pixel m, tap k) -> 4 corner BYTE offsets + 4 weights (mask folded in), tap-major ----
    // (all offset/mask loads of a thread are issued before any of them is used: one round trip; a wave whose lanes hold no
    //  entry in a pass skips it -- 288 entries on 256 threads: the second pass is wave 0's alone)
    {
        constexpr int E = BM * 9;
        constexpr int TI = (E + NTHR - 1) / NTHR;
        const int ldx4 = a.ldx * 4, rowb = a.W * ldx4;
        float tdy[TI], tdx[TI], tmk[TI];
        // the short last pass (32 of the 288 entries) rotates over the waves with the workgroup id: the four workgroups of a CU
        // build their tables at the same time (they start together), wave w of each on SIMD w -- with the last pass always on
        // wave 0, SIMD 0 issued two passes for all four of them while the other SIMDs waited at the barrier
        const int rot = (blockIdx.x >> 8) & 3;
        const int wrot = (wave - rot) & 3, trot = (tid - rot * 64) & 255;
#pragma unroll
        for (int i = 0; i < TI; ++i) {
            if (wrot * 64 + NTHR * i >= E) continue;                 // (uniform)
            const int it = trot + NTHR * i;
            const int k = it / BM, m = it & (BM - 1);
            const int oy = oy0 + (m >> 4), ox = ox0 + (m & 15);
            const bool in = it < E && oy < a.H && ox < a.W;
            if (parts) {
                // partial sums of the K-split offset conv: chunk order, then the bias (and the mask's sigmoid)
                const size_t plane = (size_t)a.N * a.H * a.W * 32;
                const float *pp = a.omPart + (in ? (((size_t)n * a.H + oy) * a.W + ox) * 32 : 0);
                const int kk = in ? k : 0;
                float dy = pp[2 * kk], dx = pp[2 * kk + 1], mk = pp[18 + kk];
                for (int sp = 1; sp < a.omSplits; ++sp) {
                    dy += pp[sp * plane + 2 * kk];
                    dx += pp[sp * plane + 2 * kk + 1];
                    mk += pp[sp * plane + 18 + kk];
                }
                tdy[i] = dy + a.b_off[2 * kk];
                tdx[i] = dx + a.b_off[2 * kk + 1];
                tmk[i] = dcn_mask_sigmoid(mk + a.b_off[18 + kk]);
                continue;
            }
            const float *omp = fuse ? om_lds + (in ? m * 32 : 0) : omn + (in ? ((size_t)oy * a.W + ox) * a.ldom : 0);
            tdy[i] = omp[in ? 2 * k : 0];
            tdx[i] = omp[in ? 2 * k + 1 : 0];
            tmk[i] = omp[in ? 18 + k : 0];
        }
#pragma unroll
        for (int i = 0; i < TI; ++i) {
            if (wrot * 64 + NTHR * i >= E) continue;                 // (uniform)
            const int it = trot + NTHR * i;
            const int k = it / BM, m = it & (BM - 1);
            const int ky = (k * 11) >> 5, kx = k - 3 * ky;            // k / 3, k % 3 for k < 9
            const int oy = oy0 + (m >> 4), ox = ox0 + (m & 15);
            int4 o;
            f32x4 wgt;
            dcn_tab_entry(tdy[i], tdx[i], tmk[i], (float)(oy - 1 + ky), (float)(ox - 1 + kx), it < E && oy < a.H && ox < a.W,
                          a.H, a.W, ldx4, rowb, o, wgt);
            if (it < E) {
                *reinterpret_cast<int4 *>(tab_off + it * 4) = o;
                *reinterpret_cast<f32x4 *>(tab_w + it * 4) = wgt;
            }
        }
    }
    __syncthreads();
    CT_STAMP(3);

    // ---- gather assignment: thread -> pixel, channel quad, slab(s) ----
    // BM = 64: thread -> (pixel tl>>2, quad tl&3), every slab; BM = 32: (pixel tl>>3, quad tl&3), slabs (tl>>2)&1, +2, ..
    constexpr int GK = NKK * BM / 64;                   // slabs gathered per thread
    constexpr int SSTR = (BM == 64) ? 1 : 2;            // ... and their stride
    const int gm = (BM == 64) ? (tl >> 2) : (tl >> 3), gq = tl & 3;
    const int gk0 = (BM == 64) ? 0 : ((tl >> 2) & 1);
    const int lslot = gm * 16 + ((gq ^ ((gm >> 1) & 2)) << 2);   // float offset inside a slab
    float *lds_g = lds_a;
    // the corners come through a buffer descriptor of this image: table byte offset + the lane's (slab, quad) constant in the
    // vector offset (one add per corner), the chunk in the scalar offset, the slab in the immediate
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(xin), 0, (int)((((unsigned)a.H * a.W - 1u) * a.ldx + a.Cin) * 4u), 0x00020000);
    const int tconst = (gk0 * 16 + gq * 4) * 4;
    const int taddr = gm * 16;                                   // byte offset of this pixel's entry of tap 0 (tap-major table)
    // two gather stages in flight (register slots 0/1): the corner loads of step s+2 are issued
    // before the MFMAs of step s and consumed (blend + LDS store) after the MFMAs of step s+1
    f32x4 cv[2][GK][4];
    f32x4 gw[2];
    auto gather_load = [&](int slot, int chunk, int tap) {
        const int ta = taddr + tap * (BM * 16);
        const int4 o = *reinterpret_cast<const int4 *>(reinterpret_cast<const char *>(tab_off) + ta);
        gw[slot] = *reinterpret_cast<const f32x4 *>(reinterpret_cast<const char *>(tab_w) + ta);
        const int so = chunk * (16 * NKK * 4);
        if (CT_ABL & 1) {
#pragma unroll
            for (int kk = 0; kk < GK; ++kk)
#pragma unroll
                for (int c = 0; c < 4; ++c) cv[slot][kk][c] = f32x4{(float)o.x, (float)o.y, (float)(o.z + kk), (float)(o.w + c)};
            return;
        }
#pragma unroll
        for (int kk = 0; kk < GK; ++kk) {
            cv[slot][kk][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, o.x + tconst + kk * SSTR * 64, so, 0));
            cv[slot][kk][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, o.y + tconst + kk * SSTR * 64, so, 0));
            cv[slot][kk][2] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, o.z + tconst + kk * SSTR * 64, so, 0));
            cv[slot][kk][3] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, o.w + tconst + kk * SSTR * 64, so, 0));
        }
    };
    auto gather_store = [&](int slot, int buf) {
#pragma unroll
        for (int kk = 0; kk < GK; ++kk) {
            const f32x4 v = dcn_blend(gw[slot], cv[slot][kk]);
            *reinterpret_cast<f32x4 *>(lds_g + buf * BUF + (gk0 + kk * SSTR) * SLAB + lslot) = v;
        }
    };

    int aoff[WM];
#pragma unroll
    for (int mt = 0; mt < WM; ++mt) {
        const int m = (wm * WM + mt) * 16 + li;
        aoff[mt] = m * 16 + ((lg ^ ((m >> 1) & 2)) << 2);
    }

    f32x4 acc[WM][WN];
#pragma unroll
    for (int mt = 0; mt < WM; ++mt)
#pragma unroll
        for (int nt = 0; nt < WN; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (nmax > 0) {
        int ch, tp;
        step_ct(0, ch, tp);
        gather_load(0, ch, tp);
        step_ct(1, ch, tp);
        gather_load(1, ch, tp);
        gather_store(0, 0);
        __syncthreads();
        CT_STAMP(4);
        // one step (P = s & 1 is static, two steps per loop iteration): A fragments of step s (LDS -> VGPR, all of them
        // up front) | weights of s+1, corners of s+2 (global) | MFMAs of s | blend + LDS store of s+1's tile | barrier
        auto step = [&](auto ptag, int s) {
            constexpr int P = decltype(ptag)::value;
            int c1, t1, c2, t2;
            step_ct(s + 1, c1, t1);
            step_ct(s + 2, c2, t2);
            f32x4 af[NKK][WM];
#pragma unroll
            for (int kk = 0; kk < NKK; ++kk)
#pragma unroll
                for (int mt = 0; mt < WM; ++mt) {
                    if (CT_ABL & 16) af[kk][mt] = f32x4{(float)s, 1.f, 2.f, (float)(kk + mt)};
                    else af[kk][mt] = *reinterpret_cast<const f32x4 *>(lds_g + P * BUF + kk * SLAB + aoff[mt]);
                }
            if (!(CT_ABL & 4)) load_b(bq[P ^ 1], c1, t1);
            else if (P == 0) {
#pragma unroll
                for (int kk = 0; kk < NKK; ++kk)
#pragma unroll
                    for (int nt = 0; nt < WN; ++nt) bq[1][kk][nt] = bq[0][kk][nt];
            }
            // slot P held step s, already blended into LDS buffer P by the previous iteration
            gather_load(P, c2, t2);
            // step s+1's A tile (corners loaded one step ago); past the last step this blends the clamped re-fetch
            // into a buffer nobody reads
            f32x4 v[GK];
#pragma unroll
            for (int kk = 0; kk < GK; ++kk)
                v[kk] = dcn_blend(gw[P ^ 1], cv[P ^ 1][kk]);
            // keep these global loads ahead of this step's MFMAs (hipcc otherwise sinks them to their
            // first use and exposes the full latency): neither VMEM nor MFMA may cross
            __builtin_amdgcn_sched_barrier(0x386);
#pragma unroll
            for (int kk = 0; kk < NKK; ++kk)
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int mt = 0; mt < WM; ++mt)
#pragma unroll
                        for (int nt = 0; nt < WN; ++nt) {
                            if (CT_ABL & 2) asm volatile("" : "+v"(acc[mt][nt]) : "v"(af[kk][mt][e]), "v"(bq[P][kk][nt][e]));
                            else acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[kk][mt][e], bq[P][kk][nt][e],
                                                                              acc[mt][nt], 0, 0, 0);
                        }
            __builtin_amdgcn_sched_barrier(0x386);
#pragma unroll
            for (int kk = 0; kk < GK; ++kk)
                *reinterpret_cast<f32x4 *>(lds_g + (P ^ 1) * BUF + (gk0 + kk * SSTR) * SLAB + lslot) = v[kk];
            if (!(CT_ABL & 8)) __syncthreads();
            __builtin_amdgcn_sched_barrier(0);                   // (steps are scheduled one by one)
        };
        // Both steps of an iteration sit in ONE basic block and an odd last step is peeled: with a branch between them
        // (`if (s + 1 < nmax)`) the waitcnt insertion drained every load in flight -- vmcnt(0), the corners of s+1
        // included -- at the top of every other step.
        int s = 0;
        for (; s + 1 < nmax; s += 2) {
            step(std::integral_constant<int, 0>{}, s);
            step(std::integral_constant<int, 1>{}, s + 1);
        }
        if (s < nmax) step(std::integral_constant<int, 0>{}, s);
    }

    CT_STAMP(5);
    CT_STAMP_VAL(8, pi);
    CT_STAMP_VAL(9, nmax);
    dcn_store_acc<WM, WN>(a, acc, n, oy0, ox0, split, nt0, wm, lane, psc, psh);
    CT_STAMP(6);
    CT_STAMP_RT(7);
}

// ---- persistent form of the MAIN launch (round 6) ----------------------------------------------------------------------
// Phase stamps of the kernel above at four streams (profiles/r06_q_dcn_phases_b4.txt): inside its loop a workgroup runs AT the
// matrix pipe's rate -- 2060 clocks per (chunk, tap) step with four workgroups per CU, 4 x 512 clocks of MFMA issue per SIMD --
// but a third of a 64 -> 64 workgroup's life is not the loop (kernel arguments 3.3 k clocks, sampling table 7.6 k, first
// gathers 3.1 k, epilogue 4.9 k against 37 k of loop), and the dispatcher starts a launch's workgroups in rounds that stay
// in phase: all of them build tables at the same time, then all of them loop.  Here a launch is `slots` resident workgroups;
// each owns one (layer, cout block, K split) column and strides over its pixel tiles as ONE stream of (chunk, tap) steps:
// the corner loads two steps ahead and the weight loads one step ahead run across tile boundaries, the next tile's
// offset / mask values are fetched during the first two steps of a tile and its sampling table (a second LDS copy) is
// built right after them, the epilogue is stores between two steps.  Same tile shape, same K order, same splits as the
// kernel above: bit-identical results.
struct DcnPersist {
    DcnArgs p[DCN_MAX_GROUP];
    int first[DCN_MAX_GROUP + 1];    // workgroup ids [first[i], first[i+1]) work on layer i ...
    int per_col[DCN_MAX_GROUP];      // ... per_col[i] of them on each of its (cout block, K split) columns
    int n;
};

template <int WN, int NKK>
__global__ __launch_bounds__(256) void dcn_persist_kernel(DcnPersist g)
{
    constexpr int BM = 32, WM = 2, WGN = 4;
    constexpr int SLAB = BM * 16, BUF = NKK * SLAB, NTHR = 256, UPC2 = NKK / 2;
    constexpr int E = BM * 9;
    constexpr int TABB = 2 * E * 16;                 // bytes of one table: int4 offsets[E], then float4 weights[E]
    // dynamic LDS: A tile double buffer | two sampling tables
    extern __shared__ __attribute__((aligned(16))) float dlds[];
    float *lds_a = dlds;
    char *tab = reinterpret_cast<char *>(dlds + 2 * BUF);

    // (stamps, debug builds: 0 / 15 real time at start / end, 1 start, 2 first table + gathers done, then per tile i < 4:
    //  3+3i first two steps + next table, 4+3i loop, 5+3i stores; 16 tiles walked, 17 steps per tile, 18 layer)
    CT_STAMP_RT(0);
    CT_STAMP(1);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave;                             // 1 x 4 waves: every wave holds all 32 pixels x its own WN cout tiles

    int bid = blockIdx.x;
    int pi = 0;
#pragma unroll
    for (int i = 1; i < DCN_MAX_GROUP; ++i)
        if (i < g.n && bid >= g.first[i]) pi = i;
    const DcnArgs a = g.p[pi];       // (by value: bursts of wide scalar loads instead of dependent reloads)
    bid -= g.first[pi];
    const int per = g.per_col[pi];
    const int col = bid / per;
    int t = bid - col * per;                         // first pixel tile; then t += per
    const int cb = col % a.coutBlocks, split = col / a.coutBlocks;
    const int ptiles = a.ptiles;
    if (t >= ptiles) return;                         // (uniform; the host never launches such workgroups)
    const int tilesX = a.tilesX, txy = tilesX * a.tilesY;
    auto coords = [&](int tt, int &n, int &oy0, int &ox0) {
        n = tt / txy;
        const int r = tt - n * txy;
        const int ty = r / tilesX;
        oy0 = ty * 2;
        ox0 = (r - ty * tilesX) * 16;
    };
    const int nunits = a.nchunks / UPC2;
    const int c_begin = split * (a.chunksPerSplit / UPC2);
    const int c_end = min(nunits, c_begin + a.chunksPerSplit / UPC2);
    const int nsteps = (c_end - c_begin) * 9;        // per tile; even (host)

    // ---- weights: as in dcn_mfma_kernel ----
    const int li = lane & 15, lg = lane >> 4;
    const int nt0 = cb * (WGN * WN) + wn * WN;
    const int NCH16 = a.Cin >> 4;
    const int slab_bytes = a.NT << 10;
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.wp), 0, 9 * NCH16 * slab_bytes, 0x00020000);
    int bvo[WN];
#pragma unroll
    for (int nt = 0; nt < WN; ++nt) bvo[nt] = (min(nt0 + nt, a.NT - 1) << 10) + (lane << 4);
    auto load_b = [&](f32x4 (&b)[NKK][WN], int chunk, int tap) {
        const int so = (tap * NCH16 + chunk * NKK) * slab_bytes;
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk)
#pragma unroll
            for (int nt = 0; nt < WN; ++nt)
                b[kk][nt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, bvo[nt], so + kk * slab_bytes, 0));
    };
    float psc[WN], psh[WN];
#pragma unroll
    for (int nt = 0; nt < WN; ++nt) {
        psc[nt] = 1.0f; psh[nt] = 0.0f;
        if (!a.ws) ct_load_scale_shift(a.epi, (nt0 + nt) * 16, lane, psc[nt], psh[nt]);
    }

    // ---- sampling table of one tile, in two halves: fetch (global loads into registers) and build (-> LDS) ----
    // Tap-major entries: pass 0 = tap tid >> 5 of pixel tid & 31 (all threads), pass 1 = tap 8 of pixel tid (the first 32 lanes
    // of wave 0) -- the same pixel in both passes.  Everything of an entry that does not depend on the tile is computed here,
    // once per workgroup; per tile the offset / mask values come through ONE buffer descriptor (the tile's first pixel in the
    // scalar offset, the entry's own offset a constant VGPR).
    const int H = a.H, W = a.W, omSplits = a.omSplits;          // (kept in SGPRs: the tile loop reads them at every boundary)
    const bool parts = omSplits > 0;                 // (uniform)
    const int ldx4 = a.ldx * 4, rowb = a.W * ldx4;
    const unsigned imgb = (unsigned)a.H * a.W * (unsigned)ldx4;      // bytes of one image of the input view
    const int my = (tid >> 4) & 1, mx = tid & 15;
    const int k0 = tid >> 5, ky0 = (k0 * 11) >> 5, kx0 = k0 - 3 * ky0;
    const float fy0 = (float)(my - 1 + ky0), fx0 = (float)(mx - 1 + kx0);
    const int opitch = parts ? 32 : a.ldom;
    const int epix = (my * a.W + mx) * opitch * 4;
    const unsigned oplane = (unsigned)a.N * a.H * a.W * 32u * 4u;   // bytes of one partial map (parts)
    const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(parts ? a.omPart : a.om), 0,
        parts ? (int)((unsigned)a.omSplits * oplane) : (int)((((unsigned)a.N * a.H * a.W - 1u) * a.ldom + 27u) * 4u), 0x00020000);
    float bdy0 = 0.f, bdx0 = 0.f, bmk0 = 0.f, bdy1 = 0.f, bdx1 = 0.f, bmk1 = 0.f;      // conv_offset_mask.bias of the two taps
    if (parts) {
        bdy0 = a.b_off[2 * k0]; bdx0 = a.b_off[2 * k0 + 1]; bmk0 = a.b_off[18 + k0];
        bdy1 = a.b_off[16]; bdx1 = a.b_off[17]; bmk1 = a.b_off[26];
    }
    constexpr int TI = 2;
    auto om_fetch = [&](int n, int oy0, int ox0, float (&tdy)[TI], float (&tdx)[TI], float (&tmk)[TI]) {
        const bool in0 = (oy0 + my < H) & (ox0 + mx < W), in1 = in0 & (tid < 32);
        const int so = (int)((((unsigned)n * H + oy0) * W + ox0) * (unsigned)opitch * 4u);
        const int v0 = in0 ? epix + 8 * k0 : (int)0x80000000, m0 = in0 ? epix + (18 + k0) * 4 : (int)0x80000000;
        const int v1 = in1 ? epix + 64 : (int)0x80000000, m1 = in1 ? epix + 104 : (int)0x80000000;
        tdy[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ors, v0, so, 0));
        tdx[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ors, v0 + 4, so, 0));
        tmk[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ors, m0, so, 0));
        tdy[1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ors, v1, so, 0));
        tdx[1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ors, v1 + 4, so, 0));
        tmk[1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ors, m1, so, 0));
    };
    auto table_build = [&](char *tb, int n, int oy0, int ox0, float (&tdy)[TI], float (&tdx)[TI], float (&tmk)[TI]) {
        const bool in0 = (oy0 + my < H) & (ox0 + mx < W), in1 = in0 & (tid < 32);
        if (parts) {
            // the other chunks of the K-split offset conv (chunk order), then the bias and the mask's sigmoid
            const int v0 = in0 ? epix + 8 * k0 : (int)0x80000000, m0 = in0 ? epix + (18 + k0) * 4 : (int)0x80000000;
            const int v1 = in1 ? epix + 64 : (int)0x80000000, m1 = in1 ? epix + 104 : (int)0x80000000;
            int so = (int)((((unsigned)n * H + oy0) * W + ox0) * 128u);
            for (int sp = 1; sp < omSplits; ++sp) {
                so += (int)oplane;
                const float a0 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ors, v0, so, 0));
                const float a1 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ors, v0 + 4, so, 0));
                const float a2 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ors, m0, so, 0));
                const float a3 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ors, v1, so, 0));
                const float a4 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ors, v1 + 4, so, 0));
                const float a5 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ors, m1, so, 0));
                tdy[0] += a0; tdx[0] += a1; tmk[0] += a2; tdy[1] += a3; tdx[1] += a4; tmk[1] += a5;
            }
            tdy[0] += bdy0; tdx[0] += bdx0; tmk[0] = dcn_mask_sigmoid(tmk[0] + bmk0);
        }
        const float oyf = (float)oy0, oxf = (float)ox0;
        int4 o;
        f32x4 wgt;
        dcn_tab_entry(tdy[0], tdx[0], tmk[0], oyf + fy0, oxf + fx0, in0, H, W, ldx4, rowb, o, wgt);
        *reinterpret_cast<int4 *>(tb + tid * 16) = o;
        *reinterpret_cast<f32x4 *>(tb + E * 16 + tid * 16) = wgt;
        if (wave == 0) {                                 // (uniform) tap 8: ky = kx = 2
            if (parts) { tdy[1] += bdy1; tdx[1] += bdx1; tmk[1] = dcn_mask_sigmoid(tmk[1] + bmk1); }
            dcn_tab_entry(tdy[1], tdx[1], tmk[1], oyf + (float)(my + 1), oxf + (float)(mx + 1), in1, H, W, ldx4, rowb, o, wgt);
            if (tid < 32) {
                *reinterpret_cast<int4 *>(tb + (256 + tid) * 16) = o;
                *reinterpret_cast<f32x4 *>(tb + E * 16 + (256 + tid) * 16) = wgt;
            }
        }
    };

    // ---- gather assignment (BM = 32): thread -> (pixel tl >> 3, quad tl & 3), slabs (tl >> 2) & 1, +2, .. ----
    constexpr int GK = NKK / 2;
    const int gm = tid >> 3, gq = tid & 3, gk0 = (tid >> 2) & 1;
    const int lslot = gm * 16 + ((gq ^ ((gm >> 1) & 2)) << 2);
    float *lds_g = lds_a;
    // ONE buffer descriptor for the whole batch: the image goes into the scalar offset beside the chunk
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(a.x), 0, (int)((unsigned)(a.N - 1) * imgb + (((unsigned)a.H * a.W - 1u) * a.ldx + a.Cin) * 4u), 0x00020000);
    const int tconst = (gk0 * 16 + gq * 4) * 4;
    const int taddr = gm * 16;
    f32x4 cv[2][GK][4];
    f32x4 gw[2];
    auto gather_load = [&](int slot, const char *tb, int xoff, int chunk, int tap) {
        const int ta = taddr + tap * (BM * 16);
        const int4 o = *reinterpret_cast<const int4 *>(tb + ta);
        gw[slot] = *reinterpret_cast<const f32x4 *>(tb + E * 16 + ta);
        const int so = xoff + chunk * (16 * NKK * 4);
#pragma unroll
        for (int kk = 0; kk < GK; ++kk) {
            cv[slot][kk][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, o.x + tconst + kk * 128, so, 0));
            cv[slot][kk][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, o.y + tconst + kk * 128, so, 0));
            cv[slot][kk][2] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, o.z + tconst + kk * 128, so, 0));
            cv[slot][kk][3] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, o.w + tconst + kk * 128, so, 0));
        }
    };
    int aoff[WM];
#pragma unroll
    for (int mt = 0; mt < WM; ++mt) {
        const int m = mt * 16 + li;
        aoff[mt] = m * 16 + ((lg ^ ((m >> 1) & 2)) << 2);
    }
    f32x4 acc[WM][WN];
#pragma unroll
    for (int mt = 0; mt < WM; ++mt)
#pragma unroll
        for (int nt = 0; nt < WN; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- the tile this workgroup starts with: table, the first two gathers, the first weights ----
    int n_c, oy_c, ox_c;
    coords(t, n_c, oy_c, ox_c);
    int tsel_c = 0;                                  // byte offset of the current tile's table (0 / TABB)
    {
        float tdy[TI], tdx[TI], tmk[TI];
        om_fetch(n_c, oy_c, ox_c, tdy, tdx, tmk);
        table_build(tab, n_c, oy_c, ox_c, tdy, tdx, tmk);
    }
    f32x4 bq[2][NKK][WN];
    load_b(bq[0], c_begin, 0);
    __syncthreads();
    int xoff_c = (int)((unsigned)n_c * imgb);
    {
        // (steps 0 and 1 of a tile: taps 0 and 1 of its first unit -- nsteps >= 18)
        gather_load(0, tab, xoff_c, c_begin, 0);
        gather_load(1, tab, xoff_c, c_begin, 1);
#pragma unroll
        for (int kk = 0; kk < GK; ++kk) {
            const f32x4 v = dcn_blend(gw[0], cv[0][kk]);
            *reinterpret_cast<f32x4 *>(lds_g + (gk0 + kk * 2) * SLAB + lslot) = v;
        }
    }
    __syncthreads();
    CT_STAMP(2);
    int ntile = 0;
    // The four workgroups of a CU share its SIMDs, and the issue arbiter serves the oldest wave first: identical workgroups
    // drift apart (lives of 35 .. 55 us at 45 on average, profiles/r06_u), and the launch ends on stragglers that have nobody
    // left to hide their latencies behind.  Priority by progress -- a workgroup starts at 3 and drops a level with every
    // quarter of its tiles done -- keeps the four abreast (variant builds may switch it off: -DCT_DCN_PRIO=0).
#ifndef CT_DCN_PRIO
#define CT_DCN_PRIO 1
#endif
    const int ntiles = (ptiles - t + per - 1) / per;
    if (CT_DCN_PRIO) __builtin_amdgcn_s_setprio(3);

    for (;;) {
        // the tile after this one (the last tile of the workgroup names itself: its fetches are valid and never used)
        const int t_n = (t + per < ptiles) ? t + per : t;
        int n_n, oy_n, ox_n;
        coords(t_n, n_n, oy_n, ox_n);
        const int tsel_n = TABB - tsel_c;
        const int xoff_n = (int)((unsigned)n_n * imgb);
        // step s of the stream (P = s & 1 static): as in dcn_mfma_kernel, except that steps s+1 / s+2 past the tile's end
        // are steps 0 / 1 of the NEXT tile (its table, its image)
        auto step = [&](auto ptag, int s) {
            constexpr int P = decltype(ptag)::value;
            int u1 = s + 1, u2 = s + 2;
            const bool x1 = u1 >= nsteps, x2 = u2 >= nsteps;         // (uniform)
            u1 = x1 ? u1 - nsteps : u1;
            u2 = x2 ? u2 - nsteps : u2;
            const int q1 = u1 / 9, q2 = u2 / 9;
            const int c1 = c_begin + q1, t1 = u1 - q1 * 9;
            const int c2 = c_begin + q2, t2 = u2 - q2 * 9;
            f32x4 af[NKK][WM];
#pragma unroll
            for (int kk = 0; kk < NKK; ++kk)
#pragma unroll
                for (int mt = 0; mt < WM; ++mt)
                    af[kk][mt] = *reinterpret_cast<const f32x4 *>(lds_g + P * BUF + kk * SLAB + aoff[mt]);
            load_b(bq[P ^ 1], c1, t1);
            gather_load(P, tab + (x2 ? tsel_n : tsel_c), x2 ? xoff_n : xoff_c, c2, t2);
            f32x4 v[GK];
#pragma unroll
            for (int kk = 0; kk < GK; ++kk)
                v[kk] = dcn_blend(gw[P ^ 1], cv[P ^ 1][kk]);
            __builtin_amdgcn_sched_barrier(0x386);
#pragma unroll
            for (int kk = 0; kk < NKK; ++kk)
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int mt = 0; mt < WM; ++mt)
#pragma unroll
                        for (int nt = 0; nt < WN; ++nt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[kk][mt][e], bq[P][kk][nt][e], acc[mt][nt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0x386);
#pragma unroll
            for (int kk = 0; kk < GK; ++kk)
                *reinterpret_cast<f32x4 *>(lds_g + (P ^ 1) * BUF + (gk0 + kk * 2) * SLAB + lslot) = v[kk];
            __syncthreads();
            __builtin_amdgcn_sched_barrier(0);
        };
        // steps 0 / 1 with the next tile's offset / mask values in flight beside them, then its table
        {
            float tdy[TI], tdx[TI], tmk[TI];
            om_fetch(n_n, oy_n, ox_n, tdy, tdx, tmk);
            step(std::integral_constant<int, 0>{}, 0);
            step(std::integral_constant<int, 1>{}, 1);
            if (ntile < 2) CT_STAMP(19 + ntile);
            table_build(tab + tsel_n, n_n, oy_n, ox_n, tdy, tdx, tmk);
        }
        if (ntile < 4) CT_STAMP(3 + 3 * ntile);
        for (int s = 2; s < nsteps; s += 2) {
            step(std::integral_constant<int, 0>{}, s);
            step(std::integral_constant<int, 1>{}, s + 1);
        }
        if (ntile < 4) CT_STAMP(4 + 3 * ntile);
        dcn_store_acc<WM, WN>(a, acc, n_c, oy_c, ox_c, split, nt0, 0, lane, psc, psh);
        if (ntile < 4) CT_STAMP(5 + 3 * ntile);
        ++ntile;
        CT_STAMP_VAL(16, ntile);
        if (t_n == t) break;
        if (CT_DCN_PRIO) {
            const int q = (4 * ntile) / ntiles;          // (uniform)
            if (q == 1) __builtin_amdgcn_s_setprio(2);
            else if (q == 2) __builtin_amdgcn_s_setprio(1);
            else if (q >= 3) __builtin_amdgcn_s_setprio(0);
        }
#pragma unroll
        for (int mt = 0; mt < WM; ++mt)
#pragma unroll
            for (int nt = 0; nt < WN; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        t = t_n; n_c = n_n; oy_c = oy_n; ox_c = ox_n; tsel_c = tsel_n; xoff_c = xoff_n;
    }
    CT_STAMP_VAL(17, nsteps);
    CT_STAMP_VAL(18, pi);
    CT_STAMP_RT(15);
}

// Measured and dropped in round 6 (profiles/r06_c_kbench_ws_b8.txt, r06_d_ws_ablations_b8.txt, r06_e_priorities_b8.txt): a
// warp-specialised form of this kernel -- 64 px x 64 couts, four "matrix" waves that only read A / B fragments one step ahead
// and issue MFMAs, four "gather" waves that gather, blend and write the A tile two steps ahead, one barrier per step; parity
// green, 92 VGPRs.  64 -> 64 @ 128 x 128 x 8 streams: 146-148 us against 110-117 us for the 4-wave kernel.  The ablations say
// why: without the MFMAs it takes 71 us, the MFMAs need 61.5 us, together 146 -- the gather waves make no progress while the
// matrix wave of their SIMD issues (an fp32 MFMA runs on the SIMD's fp32 lanes: v_mfma_f32_16x16x4_f32 is 64 FLOP / clk / SIMD,
// the vector rate, and vector instructions of OTHER waves do not slip in between), so specialisation serialises the gather
// latency behind every MFMA burst instead of hiding it; s_setprio on either role changed nothing.  What a step costs on
// this part is MFMA cycles PLUS 4 cycles per vector instruction, whichever wave issues it -- hence the instruction diet of
// the kernel above (buffer addressing, packed blend, one-pass table) rather than a different division of labour.
// Measured and dropped in round 2 (tools/kbench.py, profiles/r02_kbench_dcn_*.txt; every variant was parity-green):
//   * 8 waves per workgroup, two K groups in phase (same steps, summed through LDS) and in ANTI-phase (one group's
//     MFMAs beside the other's gather / blend / weight loads per barrier interval): 64->64 @128x128 x 8 streams 142-144
//     us against 129 us for this kernel;
//   * gathering the bilinear corners from an LDS-staged window of the input (+-2 px halo, zero-filled border, global
//     fall-back per wave-step) on 32- and 64-pixel tiles: 4x fewer vector-memory instructions (SQ_INSTS_VMEM_RD 0.51 M
//     against 2.03 M per launch) and the same 132 us.
//   * forcing one MFMA / two VALU alternation inside a step (sched_group_barrier; blend of s+1 between the MFMAs of
//     s): 157-159 us against 113-118 us at 8 streams (VALU between dependent MFMAs costs more than the idle issue
//     slots it fills), 19.8 against 20.5 us at one stream; A fragments double-buffered in registers (LDS reads of tile
//     s+1 issued one step early, no LDS round trip between a barrier and the first MFMA): equal to this kernel at
//     both sizes (tools/kbench.py, libraries built with each variant, same box).
// PMC of this kernel on that layer: MFMA busy 0.46, 4 waves per SIMD resident, 58 % of the wave cycles in
// s_waitcnt, 4.1 VALU + 3.7 SALU instructions per MFMA.  A loop of the same MFMAs with LDS fragment reads and a barrier
// every 16 MFMAs sustains 137-148 TFLOP/s on the same box (tools/micro/mfma_peak.py), so the matrix side is not what
// holds these kernels near 75-95 TFLOP/s.
}  // namespace

extern "C" size_t ct_dcn_v2_offsets_bytes(const ct_dcn_desc *d)
{
    if (!d || (d->fuse_offset != 2 && d->fuse_offset != 3) || d->Cin % 64 || d->Cin <= 0 || d->N <= 0 || d->H <= 0 || d->W <= 0)
        return 0;
    return (size_t)(d->fuse_offset == 3 ? 1 : d->Cin / 64) * d->N * d->H * d->W * 32 * sizeof(float);
}

namespace {

struct DcnPlan {
    int fuse;
    int parts;                       // offset/mask conv K-split into Cin / 64 partial maps (fuse_offset == 2)
    int BM, BN, NKK, tilesX, tilesY, coutBlocks, NT, nchunks, splits, chunksPerSplit;
    int use_ws;                      // partial / raw tiles go through the workspace (split-K, or a fused IDAUp step of a group)
    int persist;                     // MAIN launch in its persistent form (dcn_persist_kernel; algo 5xxxx / 6xxxx)
};

// `grouped`: the layer is one of several in a ct_dcn_v2_group launch (32-pixel x 64-cout tiles for all of them; a
// fused IDAUp step always goes through the workspace so that ONE reduce launch finishes every layer of the group)
int make_plan(const ct_dcn_desc *d, DcnPlan *p, bool grouped)
{
    if (!d || !d->x || !d->w_packed || !d->y) CT_FAIL_ARG("ct_dcn_v2: null pointer");
    if (d->fuse_offset < 0 || d->fuse_offset > 3) CT_FAIL_ARG("ct_dcn_v2: fuse_offset=%d (0 .. 3)", d->fuse_offset);
    p->fuse = d->fuse_offset == 1;
    p->parts = d->fuse_offset == 2 ? d->Cin / 64 : (d->fuse_offset == 3 ? 1 : 0);
    if (d->fuse_offset && (!d->b_off || (d->fuse_offset != 3 && !d->w_off_packed && !(d->fuse_offset == 2 && d->w_off_winograd))))
        CT_FAIL_ARG("ct_dcn_v2: fuse_offset needs w_off_packed and b_off");
    if (d->fuse_offset && d->Cin % 64) CT_FAIL_ARG("ct_dcn_v2: fuse_offset needs Cin %% 64 == 0 (got %d)", d->Cin);
    if (p->parts) {
        const size_t need = ct_dcn_v2_offsets_bytes(d);
        if (!d->om_partial || d->om_partial_bytes < need) {
            ct_set_error("ct_dcn_v2: fuse_offset=%d needs %zu bytes of om_partial, got %zu", d->fuse_offset, need, d->om_partial ? d->om_partial_bytes : (size_t)0);
            return CT_ERR_WORKSPACE;
        }
    }
    if (!d->fuse_offset && !d->om) CT_FAIL_ARG("ct_dcn_v2: null offset/mask map");
    if (d->up_w) {
        if (!d->up_skip || !d->up_y) CT_FAIL_ARG("ct_dcn_v2: up_w given without up_skip / up_y");
        if (d->up_f != 2 && d->up_f != 4 && d->up_f != 8) CT_FAIL_ARG("ct_dcn_v2: up_f=%d unsupported", d->up_f);
        if (d->Cout % 4 || d->up_lds % 4 || d->up_ldy % 4 || ((uintptr_t)d->up_skip & 15) || ((uintptr_t)d->up_y & 15))
            CT_FAIL_ARG("ct_dcn_v2: up-sample views must be 16-byte aligned with Cout %% 4 == 0");
    }
    if (d->Cin % 32 || d->Cin <= 0) CT_FAIL_ARG("ct_dcn_v2: Cin=%d must be a positive multiple of 32", d->Cin);
    if (d->ldx % 4 || ((uintptr_t)d->x & 15)) CT_FAIL_ARG("ct_dcn_v2: input view must be 16-byte aligned");
    if (!d->fuse_offset && d->ldom < 27) CT_FAIL_ARG("ct_dcn_v2: offset/mask map needs >= 27 channels");
    if (d->Cout <= 0 || d->N <= 0 || d->H <= 0 || d->W <= 0) CT_FAIL_ARG("ct_dcn_v2: bad shape");
    if (d->flags & CT_OUT_NCHW) CT_FAIL_ARG("ct_dcn_v2: NCHW output unsupported");
    p->NT = ct_cdiv(d->Cout, 16);
    p->tilesX = ct_cdiv(d->W, 16);
    p->BM = 64;
    p->BN = 64;
    p->NKK = 2;
    // algo: 0 heuristic; 64 / 128 = 64-pixel tile with 64 / 128 couts; 3264 / 32128 = 32-pixel tile; 43264 / 432128 =
    // 32-pixel tile stepping through 64 channels (32 * WN MFMAs per barrier)
    int algo = d->algo;
    if (algo != 0 && algo != 64 && algo != 128 && algo != 3264 && algo != 32128 && algo != 43264 && algo != 432128 &&
        algo != 53264 && algo != 532128 && algo != 63264 && algo != 632128)
        CT_FAIL_ARG("ct_dcn_v2: unknown algo %d", d->algo);
    p->persist = 0;
    if (algo == 53264 || algo == 532128 || algo == 63264 || algo == 632128) {
        // persistent form of 3264 / 32128 / 43264 / 432128: same tiles, same K order, same splits
        if (d->fuse_offset == 1) CT_FAIL_ARG("ct_dcn_v2: the persistent shapes (algo %d) read their offsets from a map (fuse_offset 0 / 2 / 3)", algo);
        p->persist = 1;
        algo = algo == 53264 ? 3264 : algo == 532128 ? 32128 : algo == 63264 ? 43264 : 432128;
    }
    if (algo == 43264) { p->NKK = 4; algo = 3264; }
    else if (algo == 432128) { p->NKK = 4; algo = 32128; }
    if (grouped) {
        // (64-pixel tiles -- half the weight-fragment traffic per flop -- only without a fused offset conv: that stage
        //  works on 32-pixel tiles)
        if (algo != 0 && algo != 3264 && algo != 32128 && !((algo == 64 || algo == 128) && d->fuse_offset != 1))
            CT_FAIL_ARG("ct_dcn_v2_group: the layers of a group run on 32-pixel tiles (algo 3264 / 32128 / 43264 / 432128), or on 64-pixel tiles (64 / 128) when no offset conv is fused");
        if (algo == 0) algo = 3264;
    }
    if (algo == 3264) { p->BM = 32; p->BN = 64; }
    else if (algo == 32128) { p->BM = 32; p->BN = 128; }
    else if (algo == 128) p->BN = 128;
    else if (algo == 64) p->BN = 64;
    else if (ct_tune_get(CT_TUNE_DCN_BN) == 128 && d->Cout >= 128) p->BN = 128;
    else if (ct_tune_get(CT_TUNE_DCN_BN) == 64) p->BN = 64;
    else if (d->Cout >= 128 && (long)d->N * p->tilesX * ct_cdiv(d->H, 4) * ct_cdiv(d->Cout, 128) >= 512) p->BN = 128;
    if (p->fuse) {
        if (algo == 64 || algo == 128) CT_FAIL_ARG("ct_dcn_v2: fuse_offset runs on the 32-pixel tiles (algo 3264 / 32128 / 43264 / 432128)");
        if (p->BM != 32) { p->BM = 32; p->BN = 64; }
    }
    if (p->NKK == 4 && d->Cin % 64) CT_FAIL_ARG("ct_dcn_v2: the 64-channel-step shapes need Cin %% 64 == 0 (got %d)", d->Cin);
    p->tilesY = ct_cdiv(d->H, p->BM / 16);
    p->coutBlocks = ct_cdiv(d->Cout, p->BN);
    p->nchunks = d->Cin / 32;
    const long tiles = (long)d->N * p->tilesX * p->tilesY * p->coutBlocks;
    int splits = d->split_k;
    if (splits <= 0) {
        splits = 1;
        if (grouped) {
            splits = p->nchunks / 2;              // 18 (chunk, tap) steps per workgroup: the layers of a group finish together
        } else if (d->workspace && tiles < 256) {
            splits = (int)((512 + tiles - 1) / tiles);
            if (splits > 16) splits = 16;
        }
    }
    const int upc = p->NKK / 2;                       // 32-channel chunks per step unit: a split owns whole units
    const int nunits = p->nchunks / upc;
    if (splits > nunits) splits = nunits;
    if (splits < 1) splits = 1;
    p->chunksPerSplit = ct_cdiv(nunits, splits) * upc;
    p->splits = ct_cdiv(p->nchunks, p->chunksPerSplit);
    p->use_ws = (p->splits > 1 || (grouped && d->up_w)) ? 1 : 0;
    if (p->persist) {
        // every split owns the same EVEN number of step units (the stream of steps keeps its LDS / register parity across
        // tiles), and one buffer descriptor spans the batch
        const int ups = p->chunksPerSplit / upc;
        if (p->BM != 32 || (ups & 1) || nunits % ups)
            CT_FAIL_ARG("ct_dcn_v2: persistent shape needs an even number of %d-channel units per split (Cin=%d, split_k=%d)", 16 * p->NKK, d->Cin, p->splits);
        if ((double)d->N * d->H * d->W * d->ldx * 4.0 >= 4294967296.0) CT_FAIL_ARG("ct_dcn_v2: persistent shape: input view of 4 GiB or more");
    }
    return CT_OK;
}

size_t ws_bytes(const ct_dcn_desc *d, const DcnPlan &p)
{
    if (!p.use_ws) return 0;
    return (size_t)p.splits * d->N * d->H * d->W * (size_t)(p.NT * 16) * sizeof(float);
}

// Split-K reduction (+ BN + ReLU), optionally fused with the IDAUp step that consumes a `proj` DCN (dla.py:543-545):
// one thread per output pixel and 4 channels reduces the partials of the (at most four) input pixels it needs, applies
// BN + ReLU and accumulates the depth-wise transposed conv on top of the skip tensor -- the same operation order as
// the plain reduction followed by upsample_add_kernel (bit-identical result), in one launch and without
// materialising the DCN output.  One launch finishes every layer of a group.
struct UpArgs {
    const float *w;      // [2f*2f][C]
    const float *skip;
    float *y;
    int f, lds, ldy;     // f == 0: plain reduction into e.y
};
struct RedArgs {
    const float *ws;
    int splits, wsCout;
    size_t Mtot;
    int N, H, W;
    int rowBlocks;       // workgroups per output row (IDAUp step) / per input row (plain): a workgroup never straddles rows, so
                         // (image, row) are scalar divisions of the workgroup id and a thread only splits (pixel, channel quad)
    EpiArgs e;
    UpArgs u;
};
struct RedGroup {
    RedArgs p[DCN_MAX_GROUP];
    int first[DCN_MAX_GROUP + 1];
    int n;
};

// (round 6: 32-bit index arithmetic, rows decoded per workgroup -- the kernel moves 66 MB in the first FINISH launch of a
//  four-stream frame and spent its time in three 64-bit divisions per thread: 22.6 us = 2.9 TB/s)
__global__ __launch_bounds__(256) void dcn_reduce_kernel(RedGroup g)
{
    int bid = blockIdx.x;
    int pi = 0;
#pragma unroll
    for (int i = 1; i < DCN_MAX_GROUP; ++i)
        if (i < g.n && bid >= g.first[i]) pi = i;
    const RedArgs r = g.p[pi];
    const EpiArgs &e = r.e;
    bid -= g.first[pi];
    const int row = bid / r.rowBlocks;                    // (scalar)
    const unsigned inrow = (unsigned)(bid - row * r.rowBlocks) * 256u + threadIdx.x;
    const float *ws = r.ws;
    const int splits = r.splits, wsCout = r.wsCout;
    const unsigned plane = (unsigned)r.Mtot * wsCout;     // floats of one split's partial map (< 2^30: host)
    if (r.u.f == 0) {
        const unsigned quads = wsCout >> 2;
        const unsigned x = inrow / quads;
        if (x >= (unsigned)r.W) return;
        const unsigned c4 = (inrow - x * quads) << 2;
        const unsigned m = (unsigned)row * r.W + x;       // row = n * H + y
        const float *src = ws + (size_t)(m * wsCout + c4);
        f32x4 s = *reinterpret_cast<const f32x4 *>(src);
        for (int k = 1; k < splits; ++k) s += *reinterpret_cast<const f32x4 *>(src + (size_t)k * plane);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int co = c4 + i;
            if (co >= e.Cout) break;
            const float sc = e.scale ? e.scale[co] : 1.0f;
            const float sh = e.shift ? e.shift[co] : 0.0f;
            e.y[(size_t)m * e.ldy + co] = ct_epilogue_value(e, s[i], co, sc, sh, 0.0f);
        }
        return;
    }
    const UpArgs &u = r.u;
    const int H = r.H, W = r.W;
    const int C = e.Cout, f = u.f, kw = 2 * f, p = f >> 1;
    const unsigned C4 = C >> 2;
    const int lf = 31 - __builtin_clz(f);                 // f = 2 / 4 / 8
    const int Ho = H * f, Wo = W * f;
    const unsigned ox = inrow / C4;
    if (ox >= (unsigned)Wo) return;
    const int c = (int)(inrow - ox * C4) * 4;
    const int n = row / Ho, oy = row - n * Ho;            // (scalar)
    const int iy = (oy + p) >> lf, ky = (oy + p) & (f - 1);
    const int ix = ((int)ox + p) >> lf, kx = ((int)ox + p) & (f - 1);
    const unsigned opix = (unsigned)row * Wo + ox;
    f32x4 acc = *reinterpret_cast<const f32x4 *>(u.skip + (size_t)opix * u.lds + c);
    const f32x4 sc = e.scale ? *reinterpret_cast<const f32x4 *>(e.scale + c) : f32x4{1.f, 1.f, 1.f, 1.f};
    const f32x4 sh = e.shift ? *reinterpret_cast<const f32x4 *>(e.shift + c) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int jy = 0; jy < 2; ++jy) {
        const int yy = iy - jy;
        if (yy < 0 || yy >= H) continue;                  // (uniform)
#pragma unroll
        for (int jx = 0; jx < 2; ++jx) {
            const int xx = ix - jx;
            if (xx < 0 || xx >= W) continue;
            const unsigned m = ((unsigned)n * H + yy) * W + xx;
            const float *src = ws + (size_t)(m * wsCout + c);
            f32x4 s = *reinterpret_cast<const f32x4 *>(src);
            for (int k = 1; k < splits; ++k) s += *reinterpret_cast<const f32x4 *>(src + (size_t)k * plane);
            const int widx = (ky + f * jy) * kw + (kx + f * jx);
            const f32x4 wv = *reinterpret_cast<const f32x4 *>(u.w + widx * C + c);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] += ct_epilogue_value(e, s[i], c + i, sc[i], sh[i], 0.0f) * wv[i];
        }
    }
    *reinterpret_cast<f32x4 *>(u.y + (size_t)opix * u.ldy + c) = acc;
}

void fill_args(const ct_dcn_desc *d, const DcnPlan &p, DcnArgs *a)
{
    a->x = d->x; a->om = d->om; a->wp = d->w_packed;
    a->N = d->N; a->H = d->H; a->W = d->W; a->Cin = d->Cin; a->ldx = d->ldx; a->ldom = d->ldom;
    a->tilesX = p.tilesX; a->tilesY = p.tilesY; a->coutBlocks = p.coutBlocks; a->NT = p.NT;
    a->nchunks = p.nchunks; a->chunksPerSplit = p.chunksPerSplit;
    a->tiles = d->N * p.tilesX * p.tilesY * p.coutBlocks;
    a->ws = p.use_ws ? d->workspace : nullptr;
    a->wsCout = p.NT * 16;
    a->epi.scale = d->scale; a->epi.shift = d->shift; a->epi.res = nullptr; a->epi.y = d->y;
    a->epi.ldr = 0; a->epi.ldy = d->ldy; a->epi.Cout = d->Cout; a->epi.Ho = d->H; a->epi.Wo = d->W;
    a->epi.flags = d->flags & CT_RELU; a->epi.sig_lo = a->epi.sig_hi = 0; a->epi.dep_lo = a->epi.dep_hi = 0;
    a->epi.depth_scale = 1.0f;
    a->w_off = p.fuse ? d->w_off_packed : nullptr;
    a->b_off = d->fuse_offset ? d->b_off : nullptr;
    a->omPart = p.parts ? d->om_partial : nullptr;
    a->omSplits = p.parts;
    a->offsOnly = 0;
    a->ptiles = d->N * p.tilesX * p.tilesY;
    a->splits = p.splits;
    a->xsx = 0; a->xper = 0; a->pband = 0;
    if (ct_tune_get(CT_TUNE_DCN_XCD)) {
        int sx = 1;
        while (sx < 8 && p.splits % (2 * sx) == 0) sx *= 2;          // gcd(splits, 8)
        a->xsx = sx;
        a->pband = ct_cdiv(a->ptiles, 8 / sx);
        a->xper = p.coutBlocks * (p.splits / sx) * a->pband;         // ids per XCD
    }
}

// dynamic LDS of one workgroup: A double buffers + the two tables (+ om tile and the offset-conv scratch when fused)
size_t lds_bytes(int BM, bool fuse_any, bool single_chunk, int NKK = 2)
{
    using OffCfg = KsCfg<3, 1, 2, 2, 4>;
    size_t regionA = sizeof(float) * (size_t)(2 * NKK * BM * 16 + 2 * BM * 9 * 4);
    if (!fuse_any) return regionA;
    // (every fused layer has Cin == 64: the offset conv holds a single chunk and needs no double buffer)
    const size_t scratch = single_chunk ? sizeof(float) * (size_t)OffCfg::LDS1_FLOATS : OffCfg::LDS_BYTES;
    return sizeof(float) * (size_t)(BM * 32) + (scratch > regionA ? scratch : regionA);
}

int launch_group(const ct_dcn_desc *descs, int n, bool grouped, int phases, void *stream)
{
    if (!descs || n < 1 || n > DCN_MAX_GROUP) CT_FAIL_ARG("ct_dcn_v2_group: 1..%d layers per launch", DCN_MAX_GROUP);
    DcnPlan plans[DCN_MAX_GROUP];
    DcnGroup g;
    RedGroup rg;
    g.n = n;
    rg.n = 0;
    long blocks = 0, rblocks = 0;
    bool fuse_any = false, single_chunk = true;
    const ct_dcn_desc *legacy_up = nullptr;
    for (int i = 0; i < n; ++i) {
        const ct_dcn_desc *d = descs + i;
        DcnPlan &p = plans[i];
        int rc = make_plan(d, &p, grouped);
        if (rc != CT_OK) return rc;
        // (one MAIN launch = one kernel instantiation; the OFFSETS / FINISH launches do not depend on the tile shape, so
        //  layers whose MAIN launches differ -- e.g. in channels per step -- may share them)
        if ((phases & CT_DCN_MAIN) && (p.BM != plans[0].BM || p.BN != plans[0].BN || p.NKK != plans[0].NKK || p.persist != plans[0].persist))
            CT_FAIL_ARG("ct_dcn_v2_group: layer %d resolves to another tile shape than layer 0", i);
        const size_t need = ws_bytes(d, p);
        if (need > 0 && (!d->workspace || d->workspace_bytes < need)) {
            ct_set_error("ct_dcn_v2: layer %d: split_k=%d needs %zu workspace bytes, got %zu", i, p.splits, need, d->workspace_bytes);
            return CT_ERR_WORKSPACE;
        }
        fill_args(d, p, &g.p[i]);
        fuse_any = fuse_any || p.fuse;
        if (p.fuse && d->Cin != 64) single_chunk = false;
        g.first[i] = (int)blocks;                  // (a multiple of 8 when the XCD-aware order is on: id % 8 stays the XCD)
        blocks += g.p[i].xsx ? 8L * g.p[i].xper : (long)g.p[i].tiles * p.splits;
        if (blocks > 0x7fffffffL) CT_FAIL_ARG("ct_dcn_v2: grid too large");
        if (p.use_ws) {
            RedArgs &r = rg.p[rg.n];
            r.ws = d->workspace; r.splits = p.splits; r.wsCout = g.p[i].wsCout;
            r.Mtot = (size_t)d->N * d->H * d->W;
            r.N = d->N; r.H = d->H; r.W = d->W;
            r.e = g.p[i].epi;
            long rows;
            if (d->up_w) {
                r.u.w = d->up_w; r.u.skip = d->up_skip; r.u.y = d->up_y; r.u.f = d->up_f; r.u.lds = d->up_lds; r.u.ldy = d->up_ldy;
                r.rowBlocks = (d->W * d->up_f * (d->Cout / 4) + 255) / 256;
                rows = (long)d->N * d->H * d->up_f;
            } else {
                r.u.w = nullptr; r.u.skip = nullptr; r.u.y = nullptr; r.u.f = 0; r.u.lds = r.u.ldy = 0;
                r.rowBlocks = (d->W * (r.wsCout / 4) + 255) / 256;
                rows = (long)d->N * d->H;
            }
            // (32-bit indices in the kernel: one split's partial map, the output and the skip view in elements)
            if ((double)r.Mtot * r.wsCout >= 1073741824.0 ||
                (d->up_w && (double)r.Mtot * d->up_f * d->up_f * (d->up_lds > d->up_ldy ? d->up_lds : d->up_ldy) >= 4294967296.0))
                CT_FAIL_ARG("ct_dcn_v2: layer %d: partial map / IDAUp output too large for the finishing launch", i);
            rg.first[rg.n] = (int)rblocks;
            rblocks += rows * r.rowBlocks;
            if (rblocks > 0x7fffffffL) CT_FAIL_ARG("ct_dcn_v2: grid too large");
            ++rg.n;
        } else if (d->up_w) {
            legacy_up = d;      // (single launch without split-K: the plain IDAUp step on the finished DCN output)
        }
    }
    g.first[n] = (int)blocks;
    for (int i = n + 1; i <= DCN_MAX_GROUP; ++i) g.first[i] = (int)blocks;
    for (int i = n; i < DCN_MAX_GROUP; ++i) g.p[i] = g.p[0];
    const DcnPlan &p0 = plans[0];
    hipStream_t s = (hipStream_t)stream;
    if (phases & CT_DCN_OFFSETS) {
        // the K-split offset/mask convs of the layers that asked for them (fuse_offset == 2), all in one launch
        DcnGroup og;
        og.n = 0;
        long oblocks = 0;
        ct_wino_off_layer wl[DCN_MAX_GROUP];
        int nw = 0;
        for (int i = 0; i < n; ++i) {
            const ct_dcn_desc *d = descs + i;
            if (d->fuse_offset != 2) continue;            // (3: another launch wrote the raw sums)
            if (d->w_off_winograd) {                      // Winograd form: all such layers of the group in one launch (wino_mfma.hip)
                ct_wino_off_layer &l = wl[nw++];
                l.x = d->x; l.N = d->N; l.H = d->H; l.W = d->W; l.Cin = d->Cin; l.ldx = d->ldx;
                l.w_winograd = d->w_off_winograd; l.part = d->om_partial;
                continue;
            }
            DcnArgs &a = og.p[og.n];
            a = g.p[i];
            a.tilesX = ct_cdiv(d->W, 16); a.tilesY = ct_cdiv(d->H, 2); a.coutBlocks = 1;
            a.tiles = d->N * a.tilesX * a.tilesY;
            a.w_off = d->w_off_packed;
            a.offsOnly = 1;
            og.first[og.n] = (int)oblocks;
            oblocks += (long)a.tiles * plans[i].parts;
            ++og.n;
        }
        if (og.n > 0) {
            if (oblocks > 0x7fffffffL) CT_FAIL_ARG("ct_dcn_v2: grid too large");
            for (int i = og.n; i <= DCN_MAX_GROUP; ++i) og.first[i] = (int)oblocks;
            for (int i = og.n; i < DCN_MAX_GROUP; ++i) og.p[i] = og.p[0];
            hipLaunchKernelGGL((dcn_mfma_kernel<32, 1, true>), dim3((unsigned)oblocks), dim3(256), lds_bytes(32, true, true), s, og);
            CT_CHECK_LAUNCH("ct_dcn_v2(offset/mask conv)");
        }
        if (nw > 0) {
            const int rc = ct_wino_offsets_group(wl, nw, stream);
            if (rc != CT_OK) return rc;
        }
    }
    const dim3 grid((unsigned)blocks);
    if (!(phases & CT_DCN_MAIN)) {
        // finish only: the partials were written by an earlier CT_DCN_MAIN call on the same descriptors
    } else if (p0.persist) {
        // `slots` resident workgroups shared out so that every layer's workgroups run the same number of steps: layer i has
        // cols[i] (cout block, K split) columns of P[i] pixel tiles, a tile costs its steps plus a boundary's worth; the
        // smallest makespan T with sum_i cols[i] * ceil(P[i] / floor(T / cost[i])) <= slots
        DcnPersist pg;
        pg.n = n;
        long cols[DCN_MAX_GROUP], P[DCN_MAX_GROUP], cost[DCN_MAX_GROUP];
        long slots = ct_tune_get(CT_TUNE_DCN_SLOTS), lo = 0, hi = 0, all = 0;
        for (int i = 0; i < n; ++i) {
            cols[i] = (long)plans[i].coutBlocks * plans[i].splits;
            P[i] = g.p[i].ptiles;
            cost[i] = (long)(plans[i].chunksPerSplit / (plans[i].NKK / 2)) * 9 + 3;
            if (cost[i] > lo) lo = cost[i];
            hi += cost[i] * P[i];
            all += cols[i];
        }
        if (slots < all) slots = all;                 // (at least one workgroup per column)
        auto need = [&](long T) {
            long w = 0;
            for (int i = 0; i < n; ++i) w += cols[i] * ((P[i] + T / cost[i] - 1) / (T / cost[i]));
            return w;
        };
        while (lo < hi) {
            const long mid = (lo + hi) / 2;
            if (need(mid) <= slots) hi = mid; else lo = mid + 1;
        }
        long pblocks = 0;
        for (int i = 0; i < n; ++i) {
            pg.p[i] = g.p[i];
            const long k = lo / cost[i];
            pg.per_col[i] = (int)((P[i] + k - 1) / k);
            pg.first[i] = (int)pblocks;
            pblocks += cols[i] * pg.per_col[i];
        }
        for (int i = n; i <= DCN_MAX_GROUP; ++i) pg.first[i] = (int)pblocks;
        for (int i = n; i < DCN_MAX_GROUP; ++i) { pg.p[i] = pg.p[0]; pg.per_col[i] = 1; }
        const dim3 pgrid((unsigned)pblocks);
        const size_t lds = sizeof(float) * (size_t)(2 * p0.NKK * 32 * 16) + 2 * (size_t)(2 * 32 * 9 * 16);
        if (p0.NKK == 4) {
            if (p0.BN == 128) hipLaunchKernelGGL((dcn_persist_kernel<2, 4>), pgrid, dim3(256), lds, s, pg);
            else hipLaunchKernelGGL((dcn_persist_kernel<1, 4>), pgrid, dim3(256), lds, s, pg);
        } else {
            if (p0.BN == 128) hipLaunchKernelGGL((dcn_persist_kernel<2, 2>), pgrid, dim3(256), lds, s, pg);
            else hipLaunchKernelGGL((dcn_persist_kernel<1, 2>), pgrid, dim3(256), lds, s, pg);
        }
    } else if (p0.NKK == 4) {
        if (p0.BM != 32) CT_FAIL_ARG("ct_dcn_v2: the 64-channel-step shapes run on 32-pixel tiles");
        const size_t lds = lds_bytes(32, fuse_any, single_chunk, 4);
        if (fuse_any) {
            if (p0.BN == 128) hipLaunchKernelGGL((dcn_mfma_kernel<32, 2, true, 4>), grid, dim3(256), lds, s, g);
            else hipLaunchKernelGGL((dcn_mfma_kernel<32, 1, true, 4>), grid, dim3(256), lds, s, g);
        } else {
            if (p0.BN == 128) hipLaunchKernelGGL((dcn_mfma_kernel<32, 2, false, 4>), grid, dim3(256), lds, s, g);
            else hipLaunchKernelGGL((dcn_mfma_kernel<32, 1, false, 4>), grid, dim3(256), lds, s, g);
        }
    } else if (fuse_any) {
        const size_t lds = lds_bytes(32, true, single_chunk);
        if (p0.BN == 128) hipLaunchKernelGGL((dcn_mfma_kernel<32, 2, true>), grid, dim3(256), lds, s, g);
        else hipLaunchKernelGGL((dcn_mfma_kernel<32, 1, true>), grid, dim3(256), lds, s, g);
    } else if (p0.BM == 32) {
        const size_t lds = lds_bytes(32, false, false);
        if (p0.BN == 128) hipLaunchKernelGGL((dcn_mfma_kernel<32, 2, false>), grid, dim3(256), lds, s, g);
        else hipLaunchKernelGGL((dcn_mfma_kernel<32, 1, false>), grid, dim3(256), lds, s, g);
    } else {
        const size_t lds = lds_bytes(64, false, false);
        if (p0.BN == 128) hipLaunchKernelGGL((dcn_mfma_kernel<64, 4, false>), grid, dim3(256), lds, s, g);
        else hipLaunchKernelGGL((dcn_mfma_kernel<64, 2, false>), grid, dim3(256), lds, s, g);
    }
    CT_CHECK_LAUNCH("ct_dcn_v2");
    if (!(phases & CT_DCN_FINISH)) return CT_OK;
    if (rg.n > 0) {
        rg.first[rg.n] = (int)rblocks;
        for (int i = rg.n + 1; i <= DCN_MAX_GROUP; ++i) rg.first[i] = (int)rblocks;
        for (int i = rg.n; i < DCN_MAX_GROUP; ++i) rg.p[i] = rg.p[0];
        hipLaunchKernelGGL(dcn_reduce_kernel, dim3((unsigned)rblocks), dim3(256), 0, s, rg);
        CT_CHECK_LAUNCH("ct_dcn_v2(split-K reduce / IDAUp step)");
    }
    if (legacy_up)
        return ct_upsample_add(legacy_up->y, legacy_up->N, legacy_up->H, legacy_up->W, legacy_up->Cout, legacy_up->ldy,
                               legacy_up->up_w, legacy_up->up_f, legacy_up->up_skip, legacy_up->up_lds, legacy_up->up_y,
                               legacy_up->up_ldy, stream);
    return CT_OK;
}

}  // namespace

extern "C" size_t ct_dcn_v2_workspace_bytes(const ct_dcn_desc *d)
{
    DcnPlan p;
    ct_dcn_desc t = *d;
    float dummy;
    if (!t.workspace) t.workspace = &dummy;
    if (!t.y) t.y = &dummy;
    if (make_plan(&t, &p, false) != CT_OK) return 0;
    return ws_bytes(&t, p);
}

extern "C" size_t ct_dcn_v2_group_workspace_bytes(const ct_dcn_desc *d)
{
    DcnPlan p;
    ct_dcn_desc t = *d;
    float dummy;
    if (!t.workspace) t.workspace = &dummy;
    if (!t.y) t.y = &dummy;
    if (make_plan(&t, &p, true) != CT_OK) return 0;
    return ws_bytes(&t, p);
}

extern "C" int ct_dcn_v2_group_plan(const ct_dcn_desc *d, size_t *workspace_bytes, int *splits)
{
    if (!d) CT_FAIL_ARG("ct_dcn_v2_group_plan: null descriptor");
    DcnPlan p;
    ct_dcn_desc t = *d;
    float dummy;
    if (!t.workspace) t.workspace = &dummy;
    if (!t.y) t.y = &dummy;
    const int rc = make_plan(&t, &p, true);
    if (rc != CT_OK) return rc;
    if (workspace_bytes) *workspace_bytes = ws_bytes(&t, p);
    if (splits) *splits = p.splits;
    return CT_OK;
}

extern "C" int ct_dcn_v2(const ct_dcn_desc *d, void *stream)
{
    return launch_group(d, 1, false, CT_DCN_OFFSETS | CT_DCN_MAIN | CT_DCN_FINISH, stream);
}

extern "C" int ct_dcn_v2_group(const ct_dcn_desc *descs, int n, int phases, void *stream)
{
    if (!(phases & (CT_DCN_OFFSETS | CT_DCN_MAIN | CT_DCN_FINISH)) || (phases & ~(CT_DCN_OFFSETS | CT_DCN_MAIN | CT_DCN_FINISH)))
        CT_FAIL_ARG("ct_dcn_v2_group: phases must be a combination of CT_DCN_OFFSETS, CT_DCN_MAIN and CT_DCN_FINISH");
    return launch_group(descs, n, true, phases, stream);
}
