// Box probes (diagnostics; NOT part of the product library): what box did a measurement run on?  Built into
// tools/micro/libct_probes.so by tools/micro/probes.py and used by tools/box_calib.py (bench.py's "box_calibration").
// No reference equivalent; nothing in centertrack_amd/ or include/ knows about this file.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

static char probe_error[256];
extern "C" const char *ctp_last_error(void) { return probe_error; }
#define CT_FAIL_ARG(...) do { snprintf(probe_error, sizeof probe_error, __VA_ARGS__); return 1; } while (0)
#define CT_CHECK_LAUNCH(name)                                                                          \
    do {                                                                                               \
        hipError_t e__ = hipGetLastError();                                                            \
        if (e__ != hipSuccess) {                                                                       \
            snprintf(probe_error, sizeof probe_error, "%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return 2;                                                                                  \
        }                                                                                              \
    } while (0)
#define CT_OK 0

// ---- box calibration: a pure MFMA loop (2 accumulator chains per wave, no memory traffic) ----
__global__ __launch_bounds__(256) void calib_mfma_kernel(int iters, float *out)
{
    f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
    const f32x4 b = f32x4{1.f, 1.0001f, 0.9999f, 1.0002f};
    const f32x4 a0 = f32x4{1.f, 2.f, 3.f, 4.f} * (1.0f + threadIdx.x * 1e-6f), a1 = a0 * 0.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[e], b[e], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[e], b[e], acc1, 0, 0, 0);
            }
    }
    const f32x4 s = acc0 + acc1;
    if (s[0] == 12345.678f) out[blockIdx.x * 256 + threadIdx.x] = s[1] + s[2] + s[3];
}

extern "C" int ctp_mfma(int blocks, int iters, float *out, void *stream)
{
    if (blocks <= 0 || iters <= 0 || !out) CT_FAIL_ARG("ctp_mfma: bad arguments");
    hipLaunchKernelGGL(calib_mfma_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, iters, out);
    CT_CHECK_LAUNCH("ctp_mfma");
    return CT_OK;
}

// ---- box probes (diagnostics of bench.py's "box_calibration": what separates the boxes of a pool) ----
// (1) dependent-load latency: ONE lane follows a ring of `hops` indices through `ring` (one element per 128-byte line,
//     a random cycle built by the host) -- footprint chosen by the caller: inside one L2 (<= 2 MB), inside the
//     Infinity Cache (64 MB), HBM (>= 1 GiB), or PINNED HOST memory (the PCIe read round trip).  out[0] = last index
//     (keeps the chain alive), out[1] = elapsed ticks of s_memrealtime (100 MHz, constant), out[2] = s_memtime clocks.
__global__ void calib_chase_kernel(const unsigned *ring, int hops, unsigned start, unsigned long long *out)
{
    unsigned idx = start;
    for (int i = 0; i < 64; ++i) idx = ring[(size_t)idx * 32];      // first touches (page walks) outside the interval
    const unsigned long long r0 = __builtin_amdgcn_s_memtime(), t0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < hops; ++i) idx = ring[(size_t)idx * 32];
    const unsigned long long r1 = __builtin_amdgcn_s_memtime(), t1 = __builtin_amdgcn_s_memrealtime();
    out[0] = idx;
    out[1] = t1 - t0;
    out[2] = r1 - r0;
}

extern "C" int ctp_chase(const unsigned *ring, int hops, unsigned start, unsigned long long *out, void *stream)
{
    if (!ring || hops <= 0 || !out) CT_FAIL_ARG("ctp_chase: bad arguments");
    hipLaunchKernelGGL(calib_chase_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, ring, hops, start, out);
    CT_CHECK_LAUNCH("ctp_chase");
    return CT_OK;
}

// (1b) the same ring followed by MANY lanes at once (every lane of `blocks` x 256 starts at its own line): dependent random
//      64-byte accesses under load -- what a gather-heavy launch sees, as opposed to one lane on an idle fabric.
__global__ __launch_bounds__(256) void calib_chase_many_kernel(const unsigned *ring, int hops, unsigned nlines, unsigned *out)
{
    const unsigned gid = blockIdx.x * 256 + threadIdx.x;
    unsigned idx = (unsigned)(((unsigned long long)gid * 2654435761ull) % nlines);
    for (int i = 0; i < hops; ++i) idx = ring[(size_t)idx * 32];
    if (idx == 0xffffffffu) out[gid] = idx;
}

extern "C" int ctp_chase_many(const unsigned *ring, int hops, unsigned nlines, int blocks, unsigned *out, void *stream)
{
    if (!ring || hops <= 0 || nlines == 0 || blocks <= 0 || !out) CT_FAIL_ARG("ctp_chase_many: bad arguments");
    hipLaunchKernelGGL(calib_chase_many_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, ring, hops, nlines, out);
    CT_CHECK_LAUNCH("ctp_chase_many");
    return CT_OK;
}

// (1c) write side: ONE lane stores to a new 128-byte line and waits for the acknowledgement (s_waitcnt 0) before the next
//      store -- the store round trip every launch pays at its end (a kernel retires when its last store is acknowledged);
//      and `blocks` x 256 lanes streaming 16-byte stores with no loads at all (fill = 1).
__global__ void calib_write_ack_kernel(float *buf, int hops, unsigned long long *out)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < hops; ++i) {
        __builtin_nontemporal_store(1.0f + i, buf + (size_t)i * 32);
        __builtin_amdgcn_s_waitcnt(0);
    }
    out[0] = __builtin_amdgcn_s_memrealtime() - t0;
}

__global__ __launch_bounds__(256) void calib_fill_kernel(float4 *dst, size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256;
    const float4 v = make_float4(1.f, 2.f, 3.f, 4.f);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) dst[i] = v;
}

extern "C" int ctp_write(void *buf, size_t bytes, int hops, int blocks, int fill, unsigned long long *out, void *stream)
{
    if (!buf || (bytes & 15) || (!fill && (!out || hops <= 0 || (size_t)hops * 128 > bytes)) || (fill && blocks <= 0))
        CT_FAIL_ARG("ctp_write: bad arguments");
    if (fill)
        hipLaunchKernelGGL(calib_fill_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (float4 *)buf, bytes / 16);
    else
        hipLaunchKernelGGL(calib_write_ack_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (float *)buf, hops, out);
    CT_CHECK_LAUNCH("ctp_write");
    return CT_OK;
}

// (1d) instruction fetch: 16 384 independent-ish VALU instructions in a straight line (64 KB of code, more than the 64 KB
//      instruction cache two CUs share once the prologue is counted), executed once by every wave of `blocks` workgroups --
//      a launch of big straight-line code (the stem, decode stage 2) starts with a cold instruction cache every time.
#define CT_REP4(x) x x x x
#define CT_REP16(x) CT_REP4(CT_REP4(x))
#define CT_REP256(x) CT_REP16(CT_REP16(x))
#define CT_REP4096(x) CT_REP16(CT_REP256(x))
__global__ __launch_bounds__(256) void calib_ifetch_kernel(float *out, float seed)
{
    float a = seed + threadIdx.x, b = seed * 0.5f, c = seed * 0.25f, d = seed * 0.125f;
    CT_REP4096(asm volatile("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4"
                            : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(seed));)
    if (a + b + c + d == 12345.678f) out[blockIdx.x * 256 + threadIdx.x] = a;
}

extern "C" int ctp_ifetch(int blocks, float *out, void *stream)
{
    if (blocks <= 0 || !out) CT_FAIL_ARG("ctp_ifetch: bad arguments");
    hipLaunchKernelGGL(calib_ifetch_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, 1.0f);
    CT_CHECK_LAUNCH("ctp_ifetch");
    return CT_OK;
}

// (2) streaming copy at a chosen occupancy: `blocks` workgroups of `threads` lanes, each lane moves 16-byte vectors with
//     `inflight` loads issued before the first store -- 256 x 256 x 1 is "one wave per SIMD, one load in flight", the
//     regime of the frame's latency-bound launches; 2048 x 256 x 4 is the bandwidth regime.
template <int INFLIGHT>
__global__ __launch_bounds__(256) void calib_stream_kernel(const float4 *src, float4 *dst, size_t n)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (INFLIGHT - 1) * stride < n; i += INFLIGHT * stride) {
        float4 v[INFLIGHT];
#pragma unroll
        for (int k = 0; k < INFLIGHT; ++k) v[k] = src[i + k * stride];
#pragma unroll
        for (int k = 0; k < INFLIGHT; ++k) dst[i + k * stride] = v[k];
    }
    for (; i < n; i += stride) dst[i] = src[i];
}

extern "C" int ctp_stream(const void *src, void *dst, size_t bytes, int blocks, int inflight, void *stream)
{
    if (!src || !dst || blocks <= 0 || (bytes & 15)) CT_FAIL_ARG("ctp_stream: bad arguments");
    const size_t n = bytes / 16;
    if (inflight >= 4)
        hipLaunchKernelGGL(calib_stream_kernel<4>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float4 *)src, (float4 *)dst, n);
    else
        hipLaunchKernelGGL(calib_stream_kernel<1>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float4 *)src, (float4 *)dst, n);
    CT_CHECK_LAUNCH("ctp_stream");
    return CT_OK;
}

// (3) `n` dependent launches of a `blocks`-workgroup kernel that adds 1 to what its predecessor wrote: the kernel
//     boundary (dispatch + argument fetch + first load + drain) of this box, per launch.
__global__ __launch_bounds__(256) void calib_chain_kernel(float *buf)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    buf[i] = buf[i] + 1.0f;
}

extern "C" int ctp_launches(int n, int blocks, float *buf, void *stream)
{
    if (n <= 0 || blocks <= 0 || !buf) CT_FAIL_ARG("ctp_launches: bad arguments");
    for (int k = 0; k < n; ++k) hipLaunchKernelGGL(calib_chain_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, buf);
    CT_CHECK_LAUNCH("ctp_launches");
    return CT_OK;
}

// (4) where the dispatcher puts workgroups: every workgroup records the hardware ids of the CU it runs on (HW_ID: shader
//     engine / array / CU; XCC_ID: the XCD) and its start / end time (100 MHz), and spins for `spin_ticks` in between.
//     `lds_bytes` of dynamic LDS cap the workgroups a CU can hold (64 KB: two), like the stem's 249 VGPRs do.  An MI355X
//     has 32 of 36 physical CUs per XCD enabled; WHICH four are fused off differs from chip to chip, and the dispatcher
//     hands workgroups to shader engines, not to CUs -- a launch sized "two workgroups per CU" finishes in one round only
//     if the engines hold equal numbers of CUs.
__global__ __launch_bounds__(256) void calib_cu_map_kernel(int spin_ticks, unsigned *out)
{
    extern __shared__ float cu_map_lds[];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);          // HW_REG_HW_ID
        const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);        // HW_REG_XCC_ID
        cu_map_lds[0] = 0.f;
        while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin_ticks) __builtin_amdgcn_s_sleep(8);
        const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
        out[4 * blockIdx.x + 0] = hw;
        out[4 * blockIdx.x + 1] = xcc;
        out[4 * blockIdx.x + 2] = (unsigned)t0;
        out[4 * blockIdx.x + 3] = (unsigned)t1;
    }
    __syncthreads();
}

extern "C" int ctp_cu_map(int blocks, int lds_bytes, int spin_ticks, unsigned *out, void *stream)
{
    if (blocks <= 0 || lds_bytes < 0 || lds_bytes > 160 * 1024 || spin_ticks < 0 || !out) CT_FAIL_ARG("ctp_cu_map: bad arguments");
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(calib_cu_map_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr = true;
    }
    hipLaunchKernelGGL(calib_cu_map_kernel, dim3(blocks), dim3(256), (size_t)lds_bytes, (hipStream_t)stream, spin_ticks, out);
    CT_CHECK_LAUNCH("ctp_cu_map");
    return CT_OK;
}

// (5) per-XCD memory rate: workgroup i copies ITS OWN contiguous chunk of `chunk_bytes` (one 16-byte load in flight per
//     lane) and records the XCD it ran on and its start / end time.  A launch ends when its slowest workgroup does: an XCD
//     whose path to memory is slower than its seven siblings' stretches every memory-bound launch of a frame although
//     the aggregate bandwidth of the chip (probes 2) barely moves.
__global__ __launch_bounds__(256) void calib_xcd_stream_kernel(const float4 *src, float4 *dst, size_t chunk_vec, unsigned *out)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    const float4 *s = src + (size_t)blockIdx.x * chunk_vec;
    float4 *d = dst + (size_t)blockIdx.x * chunk_vec;
    for (size_t i = threadIdx.x; i < chunk_vec; i += 256) d[i] = s[i];
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
        out[4 * blockIdx.x + 0] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
        out[4 * blockIdx.x + 1] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        out[4 * blockIdx.x + 2] = (unsigned)t0;
        out[4 * blockIdx.x + 3] = (unsigned)t1;
    }
}

extern "C" int ctp_xcd_stream(const void *src, void *dst, size_t chunk_bytes, int blocks, unsigned *out, void *stream)
{
    if (!src || !dst || !out || blocks <= 0 || (chunk_bytes & 15)) CT_FAIL_ARG("ctp_xcd_stream: bad arguments");
    hipLaunchKernelGGL(calib_xcd_stream_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float4 *)src, (float4 *)dst,
                       chunk_bytes / 16, out);
    CT_CHECK_LAUNCH("ctp_xcd_stream");
    return CT_OK;
}

