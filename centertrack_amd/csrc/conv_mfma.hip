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

namespace {

struct ConvArgs {
    const float *x;
    const float *wp;
    int N, H, W, Cin, ldx;
    int tilesX, tilesY, coutBlocks;
    int NT;          // CoutPad / 16
    int nchunks;     // Cin / (16*NKK)
    int chunksPerSplit;
    float *ws;       // split-K workspace or nullptr
    int wsCout;      // channel pitch of the workspace
    EpiArgs epi;
};

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

template <int KS, int STRIDE, int WGM, int WGN, int WM, int WN, int NKK>
__global__ __launch_bounds__(256) void conv_mfma_kernel(ConvArgs a)
{
    using C = ConvCfg<KS, STRIDE, WGM, WGN, WM, WN, NKK>;
    static_assert(WGM * WGN == 4, "4 waves per workgroup");
    constexpr int PAD = KS / 2;
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;

    int bid = blockIdx.x;
    const int cb = bid % a.coutBlocks; bid /= a.coutBlocks;
    const int tx = bid % a.tilesX; bid /= a.tilesX;
    const int ty = bid % a.tilesY; bid /= a.tilesY;
    const int n = bid;
    const int oy0 = ty * C::TH, ox0 = tx * 16;
    const int iy0 = oy0 * STRIDE - PAD, ix0 = ox0 * STRIDE - PAD;

    const int split = blockIdx.y;
    const int c_begin = split * a.chunksPerSplit;
    const int c_end = min(a.nchunks, c_begin + a.chunksPerSplit);

    const float *xin = a.x + (size_t)n * a.H * a.W * a.ldx;

    // ---- staging assignment: item -> (slab kk, patch pixel P, channel quad q) ----------
    int goff[C::NR];    // element offset into xin (without the chunk offset), -1 = zero fill
    int loff[C::NR];    // float offset inside one chunk buffer, -1 = no item
#pragma unroll
    for (int r = 0; r < C::NR; ++r) {
        const int it = tid + 256 * r;
        if (it < C::ITEMS) {
            const int q = it & 3;
            const int pp = it >> 2;
            const int kk = pp / C::PP;
            const int P = pp - kk * C::PP;
            const int py = P / C::PW, px = P - py * C::PW;
            const int iy = iy0 + py, ix = ix0 + px;
            loff[r] = kk * C::SLAB + P * 16 + ((q ^ ((P >> 1) & 2)) << 2);
            goff[r] = (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W)
                          ? ((iy * a.W + ix) * a.ldx + kk * 16 + q * 4)
                          : -1;
        } else {
            loff[r] = -1;
            goff[r] = -1;
        }
    }
    f32x4 stage[C::NR];
    auto stage_load = [&](int chunk) {
        const int coff = chunk * (16 * NKK);
#pragma unroll
        for (int r = 0; r < C::NR; ++r) {
            stage[r] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (goff[r] >= 0) stage[r] = *reinterpret_cast<const f32x4 *>(xin + goff[r] + coff);
        }
    };
    auto stage_store = [&](int buf) {
        float *dst = lds + buf * C::BUF;
#pragma unroll
        for (int r = 0; r < C::NR; ++r)
            if (loff[r] >= 0) *reinterpret_cast<f32x4 *>(dst + loff[r]) = stage[r];
    };

    // ---- fragment addressing -----------------------------------------------------------
    const int li = lane & 15, lg = lane >> 4;
    int pbase[WM];      // patch pixel of (row of m-tile, lane pixel) for tap (0,0)
#pragma unroll
    for (int mt = 0; mt < WM; ++mt) pbase[mt] = ((wm * WM + mt) * STRIDE) * C::PW + li * STRIDE;

    const int nt0 = cb * (C::BN / 16) + wn * WN;    // first n-tile of this wave
    const int NCH16 = a.Cin >> 4;
    auto load_b = [&](f32x4 (&b)[WN], int chunk, int kk, int tap) {
        const size_t slab = (size_t)tap * NCH16 + (size_t)chunk * NKK + kk;
#pragma unroll
        for (int nt = 0; nt < WN; ++nt) {
            b[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (nt0 + nt < a.NT)
                b[nt] = *reinterpret_cast<const f32x4 *>(a.wp + ((slab * a.NT + nt0 + nt) << 8) + (lane << 2));
        }
    };

    f32x4 acc[WM][WN];
#pragma unroll
    for (int mt = 0; mt < WM; ++mt)
#pragma unroll
        for (int nt = 0; nt < WN; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (c_begin < c_end) {
        stage_load(c_begin);
        stage_store(0);
        __syncthreads();
        f32x4 bcur[WN], bnext[WN];
        load_b(bcur, c_begin, 0, 0);
        for (int c = c_begin; c < c_end; ++c) {
            const int cur = (c - c_begin) & 1;
            const bool more = (c + 1 < c_end);
            if (more) stage_load(c + 1);
            const float *buf = lds + cur * C::BUF;
#pragma unroll
            for (int kk = 0; kk < NKK; ++kk) {
#pragma unroll
                for (int tap = 0; tap < KS * KS; ++tap) {
                    // prefetch the next step's B fragments (possibly of the next chunk)
                    {
                        int ntap = tap + 1, nkk = kk, nc = c;
                        if (ntap == KS * KS) { ntap = 0; nkk = kk + 1; }
                        if (nkk == NKK) { nkk = 0; nc = c + 1; }
                        if (nc < c_end) load_b(bnext, nc, nkk, ntap);
                    }
                    const int ky = tap / KS, kx = tap % KS;
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
                                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[mt][e], bcur[nt][e],
                                                                                  acc[mt][nt], 0, 0, 0);
#pragma unroll
                    for (int nt = 0; nt < WN; ++nt) bcur[nt] = bnext[nt];
                }
            }
            if (more) stage_store(cur ^ 1);
            __syncthreads();
        }
    }

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
                ct_store_tile(a.epi, acc[mt][nt], n, oy0 + wm * WM + mt, ox0, (nt0 + nt) * 16, lane);
    }
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
    int cfg;       // 0: BN16 (TH16), 1: BN32 (TH8), 2: BN64 (TH4), 3: BN128 (TH4)
    int TH, BN, nkk;
    int tilesX, tilesY, coutBlocks, nchunks, splits, chunksPerSplit;
    int Ho, Wo, NT;
};

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
    if (d->Cout <= 16) { p->cfg = 0; p->TH = 16; p->BN = 16; }
    else if (d->Cout <= 32) { p->cfg = 1; p->TH = 8; p->BN = 32; }
    else { p->cfg = 2; p->TH = 4; p->BN = 64; }
    p->tilesX = ct_cdiv(p->Wo, 16);
    p->tilesY = ct_cdiv(p->Ho, p->TH);
    if (p->cfg == 2 && d->Cout >= 128) {
        const long tiles128 = (long)d->N * p->tilesX * p->tilesY * ct_cdiv(d->Cout, 128);
        if (tiles128 >= 512) { p->cfg = 3; p->BN = 128; }
    }
    p->coutBlocks = ct_cdiv(d->Cout, p->BN);
    const int c16 = d->Cin / 16;
    if (d->ks == 1) p->nkk = (c16 % 4 == 0) ? 4 : ((c16 % 2 == 0) ? 2 : 1);
    else if (d->stride == 2) p->nkk = 1;
    else p->nkk = (c16 % 2 == 0) ? 2 : 1;
    p->nchunks = c16 / p->nkk;
    const long tiles = (long)d->N * p->tilesX * p->tilesY * p->coutBlocks;
    int splits = d->split_k;
    if (splits <= 0) {
        splits = 1;
        if (d->workspace && tiles < 256) {
            splits = (int)((512 + tiles - 1) / tiles);
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

template <int KS, int STRIDE, int WGM, int WGN, int WM, int WN, int NKK>
int launch_cfg(const ConvArgs &a, dim3 grid, hipStream_t s)
{
    using C = ConvCfg<KS, STRIDE, WGM, WGN, WM, WN, NKK>;
    auto k = conv_mfma_kernel<KS, STRIDE, WGM, WGN, WM, WN, NKK>;
    static bool attr_set = false;   // > 64 KiB dynamic LDS needs the opt-in attribute
    if (!attr_set && C::LDS_BYTES > 48 * 1024) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)C::LDS_BYTES);
        attr_set = true;
    }
    hipLaunchKernelGGL(k, grid, dim3(256), C::LDS_BYTES, s, a);
    return CT_OK;
}

template <int KS, int STRIDE, int NKK>
int launch_tile(int cfg, const ConvArgs &a, dim3 grid, hipStream_t s)
{
    switch (cfg) {
    case 0: return launch_cfg<KS, STRIDE, 4, 1, 4, 1, NKK>(a, grid, s);
    case 1: return launch_cfg<KS, STRIDE, 4, 1, 2, 2, NKK>(a, grid, s);
    case 2: return launch_cfg<KS, STRIDE, 2, 2, 2, 2, NKK>(a, grid, s);
    default: return launch_cfg<KS, STRIDE, 2, 2, 2, 4, NKK>(a, grid, s);
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
    Plan p;
    ct_conv_desc t = *d;
    float dummy;
    if (!t.workspace) t.workspace = &dummy;   // ask "how much would the auto plan want"
    if (!t.y) t.y = &dummy;
    if (make_plan(&t, &p) != CT_OK) return 0;
    return ws_bytes(&t, p);
}

extern "C" int ct_conv2d(const ct_conv_desc *d, void *stream)
{
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
    a.tilesX = p.tilesX; a.tilesY = p.tilesY; a.coutBlocks = p.coutBlocks;
    a.NT = p.NT; a.nchunks = p.nchunks; a.chunksPerSplit = p.chunksPerSplit;
    a.ws = p.splits > 1 ? d->workspace : nullptr;
    a.wsCout = p.NT * 16;
    a.epi.scale = d->scale; a.epi.shift = d->shift; a.epi.res = d->res; a.epi.y = d->y;
    a.epi.ldr = d->ldr; a.epi.ldy = d->ldy; a.epi.Cout = d->Cout; a.epi.Ho = p.Ho; a.epi.Wo = p.Wo;
    a.epi.flags = d->flags; a.epi.sig_lo = d->sig_lo; a.epi.sig_hi = d->sig_hi;
    a.epi.dep_lo = d->dep_lo; a.epi.dep_hi = d->dep_hi; a.epi.depth_scale = d->depth_scale;
    const long blocks = (long)d->N * p.tilesX * p.tilesY * p.coutBlocks;
    if (blocks > 0x7fffffffL) CT_FAIL_ARG("ct_conv2d: grid too large");
    dim3 grid((unsigned)blocks, (unsigned)p.splits);
    hipStream_t s = (hipStream_t)stream;
    if (d->ks == 1) {
        if (p.nkk == 4) rc = launch_tile<1, 1, 4>(p.cfg, a, grid, s);
        else if (p.nkk == 2) rc = launch_tile<1, 1, 2>(p.cfg, a, grid, s);
        else rc = launch_tile<1, 1, 1>(p.cfg, a, grid, s);
    } else if (d->stride == 2) {
        rc = launch_tile<3, 2, 1>(p.cfg, a, grid, s);
    } else {
        if (p.nkk == 2) rc = launch_tile<3, 1, 2>(p.cfg, a, grid, s);
        else rc = launch_tile<3, 1, 1>(p.cfg, a, grid, s);
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
