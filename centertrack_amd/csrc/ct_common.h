// Shared device/host helpers of libcentertrack_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "centertrack_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

void ct_set_error(const char *fmt, ...);

#define CT_FAIL_ARG(...)            \
    do {                            \
        ct_set_error(__VA_ARGS__);  \
        return CT_ERR_ARG;          \
    } while (0)

#define CT_CHECK_LAUNCH(name)                                              \
    do {                                                                   \
        hipError_t e__ = hipGetLastError();                                \
        if (e__ != hipSuccess) {                                           \
            ct_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return CT_ERR_LAUNCH;                                          \
        }                                                                  \
    } while (0)

static inline int ct_cdiv(int a, int b) { return (a + b - 1) / b; }

// tuning knobs (ct_set_tuning); defaults chosen from MI355X measurements, see DESIGN.md
enum { CT_TUNE_CONV_CFG = 0, CT_TUNE_CONV_PIPE, CT_TUNE_CONV_SMALL_TILES, CT_TUNE_SPLITK_TARGET, CT_TUNE_DCN_BN,
       CT_TUNE_CONV_KS, CT_TUNE_CONV_KS_BELOW, CT_TUNE_CONV_KS_WAVES, CT_TUNE_XCD_REMAP, CT_TUNE_HEADS_ORDER, CT_TUNE_STEM_ROWS, CT_TUNE_DCN_SLOTS, CT_TUNE_DCN_XCD, CT_TUNE_COUNT };
int ct_tune_get(int key);
void ct_affine_inverse(const double *trans, double *M);   // host_preprocess.cpp
// wino_mfma.hip: the offset/mask convs of up to 4 DCN layers in one Winograd launch, K-split over 64-channel chunks into
// raw partial maps part[Cin/64][N,H,W,32] (dcn_mfma.hip's CT_DCN_OFFSETS phase)
struct ct_wino_off_layer {
    const float *x; int N, H, W, Cin, ldx;
    const float *w_winograd;     // ct_pack_winograd_weight(conv_offset_mask.weight [27,Cin,3,3])
    float *part;
};
int ct_wino_offsets_group(const ct_wino_off_layer *layers, int n, void *stream);

// XCD-aware decode of the linear workgroup id into (cout block, pixel-tile index).  Workgroups are dealt to
// the 8 XCDs round-robin (id % 8) and each XCD has its own 4 MB L2, so with `per` = coutBlocks / 8 > 0
// XCD x only ever touches the weights of cout blocks {x, x+8, ..}: 1/8 of the packed weights per L2
// instead of all of them (heads: 5.2 MB of Winograd weights, level 5: 9.4 MB).  per == 0: plain order.
static inline int ct_xcd_per(int coutBlocks)
{
    return (ct_tune_get(CT_TUNE_XCD_REMAP) && coutBlocks >= 16 && (coutBlocks & 7) == 0) ? (coutBlocks >> 3) : 0;
}
#ifdef __HIPCC__
__device__ __forceinline__ int ct_block_cout(int &bid, int coutBlocks, int per)
{
    int cb;
    if (per) {
        const int j = bid >> 3;
        cb = (bid & 7) + 8 * (j % per);
        bid = j / per;
    } else {
        cb = bid % coutBlocks;
        bid /= coutBlocks;
    }
    return cb;
}
#endif

// ---- per-workgroup phase stamps (debug builds only: -DCT_STAMPS, tools/conv_phases.py / tools/dcn_phases.py; never part
// of the shipped library).  Every translation unit that stamps owns a buffer [CT_STAMP_BLOCKS][CT_STAMP_WORDS]; thread 0
// of a workgroup writes s_memtime (CT_STAMP), s_memrealtime (100 MHz: CT_STAMP_RT) or a value (CT_STAMP_VAL) into word i
// of its row; CT_DEFINE_STAMPS(name) exports ct_<name>_read_stamps / ct_<name>_clear_stamps for the host tool.
#if defined(CT_STAMPS) && defined(__HIPCC__)
#define CT_STAMP_WORDS 24
#define CT_STAMP_BLOCKS 8192
static __device__ unsigned long long ct_stamps_buf[CT_STAMP_WORDS * CT_STAMP_BLOCKS];
#define CT_STAMP_AT(i, v) do { if (threadIdx.x == 0 && blockIdx.x < CT_STAMP_BLOCKS && blockIdx.y == 0) ct_stamps_buf[blockIdx.x * CT_STAMP_WORDS + (i)] = (unsigned long long)(v); } while (0)
#define CT_STAMP(i) CT_STAMP_AT(i, __builtin_amdgcn_s_memtime())
#define CT_STAMP_RT(i) CT_STAMP_AT(i, __builtin_amdgcn_s_memrealtime())
#define CT_STAMP_VAL(i, v) CT_STAMP_AT(i, v)
// HW_ID (wave / SIMD / CU / SE of the stamping wave) and XCC_ID: where the dispatcher put the workgroup
#define CT_STAMP_HW(i) CT_STAMP_AT(i, ((unsigned long long)__builtin_amdgcn_s_getreg(((32 - 1) << 11) | 20) << 32) | (unsigned)__builtin_amdgcn_s_getreg(((32 - 1) << 11) | 4))
#define CT_DEFINE_STAMPS(name)                                                                                           \
    extern "C" int ct_##name##_read_stamps(unsigned long long *host, int nblocks)                                        \
    {                                                                                                                    \
        return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(ct_stamps_buf), sizeof(unsigned long long) * CT_STAMP_WORDS * nblocks); \
    }                                                                                                                    \
    extern "C" int ct_##name##_clear_stamps(void)                                                                        \
    {                                                                                                                    \
        void *p = nullptr;                                                                                               \
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(ct_stamps_buf)) != hipSuccess) return 1;                                  \
        return (int)hipMemset(p, 0, sizeof(unsigned long long) * CT_STAMP_WORDS * CT_STAMP_BLOCKS);                      \
    }
#else
#define CT_STAMP(i)
#define CT_STAMP_RT(i)
#define CT_STAMP_VAL(i, v)
#define CT_STAMP_HW(i)
#define CT_DEFINE_STAMPS(name)
#endif

// Epilogue description shared by the conv / dcn / split-K-reduce kernels.
struct EpiArgs {
    const float *scale;   // [>=Cout] or nullptr
    const float *shift;   // [>=Cout] or nullptr
    const float *res;     // NHWC residual view or nullptr
    float *y;
    int ldr, ldy;
    int Cout;
    int Ho, Wo;           // output grid (per image)
    int flags;
    int sig_lo, sig_hi, dep_lo, dep_hi;
    float depth_scale;
};

__device__ __forceinline__ float ct_sigmoid(float v) { return 1.0f / (1.0f + __expf(-v)); }

// scale / shift / residual / ReLU only (kernels whose host entry rejects the sigmoid and depth transforms)
__device__ __forceinline__ float ct_epilogue_plain(const EpiArgs &e, float v, float sc, float sh, float r)
{
    v = v * sc + sh + r;
    return (e.flags & CT_RELU) ? fmaxf(v, 0.0f) : v;
}

// Apply scale/shift/residual/activation to one accumulator value of channel `co`.
__device__ __forceinline__ float ct_epilogue_value(const EpiArgs &e, float v, int co, float sc, float sh,
                                                   float r)
{
    v = v * sc + sh + r;
    if (e.flags & CT_RELU) v = fmaxf(v, 0.0f);
    if (co >= e.sig_lo && co < e.sig_hi) v = 1.0f / (1.0f + expf(-v));
    if (co >= e.dep_lo && co < e.dep_hi) v = (1.0f / (1.0f / (1.0f + expf(-v)) + 1e-6f) - 1.0f) * e.depth_scale;
    return v;
}

// Scale / shift of channel co0 + (lane & 15), loaded EARLY (round 4): at one stream every launch is a single round of
// workgroups, so a workgroup's own latency chain is the kernel's duration; the epilogue's scale / shift loads were a
// ~1 us round trip into a cold L2 at its very end (s_memtime stamps: 0.9-1.1 us between the last MFMA and the last
// store of the backbone kernels, 2.1 us in the DCN kernel).  Issued before the main loop they cost two registers per
// n-tile and arrive long before they are needed.
__device__ __forceinline__ void ct_load_scale_shift(const EpiArgs &e, int co0, int lane, float &sc, float &sh)
{
    const int co = co0 + (lane & 15);
    const bool ok = co < e.Cout;
    sc = (e.scale && ok) ? e.scale[co] : 1.0f;
    sh = (e.shift && ok) ? e.shift[co] : 0.0f;
}

// RULE for every store through a buffer descriptor in this library (round 6): a store WIDER than 64 bits
// (`__builtin_amdgcn_raw_buffer_store_b96 / _b128`) keeps its scalar-offset argument at the constant 0 and carries the whole offset in
// the vector offset.  With a scalar-offset REGISTER hipcc does not pad the "VALU overwrites the store's data registers" hazard (the
// published exemption) and gfx950 needs the pad: the store sends the next instruction's result.  32-bit stores (below) are not
// affected.  tools/isa_hazards.py checks what the compiler emitted for every kernel (tests/test_isa_hazards.py).
// Store the 16x16 MFMA tile `acc` (C/D layout: col = lane&15, row = (lane>>4)*4 + e) whose
// rows are the 16 consecutive output pixels (n, oy, ox0..ox0+15) and whose columns are the
// couts co0..co0+15; sc / sh: the channel's scale / shift (ct_load_scale_shift).
__device__ __forceinline__ void ct_store_tile(const EpiArgs &e, f32x4 acc, int n, int oy, int ox0, int co0,
                                              int lane, float sc, float sh)
{
    if (oy >= e.Ho) return;
    const int co = co0 + (lane & 15);
    const int xr = ox0 + ((lane >> 4) << 2);
    if (e.flags & CT_OUT_NCHW) {
        if (co >= e.Cout) return;
        const size_t pix = ((size_t)n * e.Ho + oy) * e.Wo;
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ox = xr + i;
            float r = 0.0f;
            if (e.res && ox < e.Wo) r = e.res[(pix + ox) * e.ldr + co];
            v[i] = ct_epilogue_value(e, acc[i], co, sc, sh, r);
        }
        float *dst = e.y + ((size_t)n * e.Cout + co) * ((size_t)e.Ho * e.Wo) + (size_t)oy * e.Wo + xr;
        if (xr + 3 < e.Wo && ((e.Wo & 3) == 0)) {
            *reinterpret_cast<f32x4 *>(dst) = f32x4{v[0], v[1], v[2], v[3]};
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (xr + i < e.Wo) dst[i] = v[i];
        }
        return;
    }
    // NHWC (round 6): stores and residual loads through buffer descriptors of image n -- the lane's (pixel, cout) byte offset
    // once per tile, the four pixels of the lane as scalar offsets; lanes past the row's end or the last cout carry an
    // out-of-range offset (store dropped by the hardware, residual read as zero).  Replaces a 64-bit address per value:
    // vector instructions are paid on top of the fp32 MFMA time on this part.
    const unsigned img = (unsigned)e.Ho * e.Wo;
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(
        e.y + (size_t)n * img * e.ldy, 0, (int)(((img - 1u) * e.ldy + e.Cout) * 4u), 0x00020000);
    const int lpix = oy * e.Wo + xr;
    const bool cok = co < e.Cout;
    const int vy = (lpix * e.ldy + co) * 4;
    float r[4] = {0.f, 0.f, 0.f, 0.f};
    if (e.res) {                                                  // (uniform)
        const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float *>(e.res) + (size_t)n * img * e.ldr, 0, (int)(((img - 1u) * e.ldr + e.Cout) * 4u), 0x00020000);
        const int vr = (lpix * e.ldr + co) * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            r[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rrs, (cok && xr + i < e.Wo) ? vr : (int)0x80000000, i * e.ldr * 4, 0));
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float v = ct_epilogue_value(e, acc[i], co, sc, sh, r[i]);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), yrs, (cok && xr + i < e.Wo) ? vy : (int)0x80000000, i * e.ldy * 4, 0);
    }
}

__device__ __forceinline__ void ct_store_tile(const EpiArgs &e, f32x4 acc, int n, int oy, int ox0, int co0, int lane)
{
    float sc, sh;
    ct_load_scale_shift(e, co0, lane, sc, sh);
    ct_store_tile(e, acc, n, oy, ox0, co0, lane, sc, sh);
}
