// HIP runtime helpers of the per-frame host loop (graph capture / replay, async copies).
#include "ct_common.h"

// ---- HIP graph + async copy helpers ----------------------------------------------------------
// The per-frame host loop of the detector is four runtime calls (H2D of the prior-heat-map blobs, D2D of
// the frame, graph launch, D2H of the packed detections); issuing them straight through the HIP runtime
// costs a fraction of the same calls made through a tensor library's dispatcher.
extern "C" int ct_graph_begin(void *stream)
{
    hipError_t e = hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeRelaxed);
    if (e != hipSuccess) {
        ct_set_error("ct_graph_begin: %s", hipGetErrorString(e));
        return CT_ERR_LAUNCH;
    }
    return CT_OK;
}

extern "C" void *ct_graph_end(void *stream)
{
    hipGraph_t g = nullptr;
    hipError_t e = hipStreamEndCapture((hipStream_t)stream, &g);
    if (e != hipSuccess || !g) {
        ct_set_error("ct_graph_end: capture failed: %s", hipGetErrorString(e));
        return nullptr;
    }
    hipGraphExec_t exec = nullptr;
    e = hipGraphInstantiate(&exec, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (e != hipSuccess) {
        ct_set_error("ct_graph_end: instantiate failed: %s", hipGetErrorString(e));
        return nullptr;
    }
    return (void *)exec;
}

extern "C" int ct_graph_launch(void *exec, void *stream)
{
    hipError_t e = hipGraphLaunch((hipGraphExec_t)exec, (hipStream_t)stream);
    if (e != hipSuccess) {
        ct_set_error("ct_graph_launch: %s", hipGetErrorString(e));
        return CT_ERR_LAUNCH;
    }
    return CT_OK;
}

extern "C" void ct_graph_destroy(void *exec)
{
    if (exec) (void)hipGraphExecDestroy((hipGraphExec_t)exec);
}

// kind: 0 device->device, 1 host->device, 2 device->host (host memory should be pinned)
extern "C" int ct_memcpy_async(void *dst, const void *src, size_t bytes, int kind, void *stream)
{
    const hipMemcpyKind k = kind == 0 ? hipMemcpyDeviceToDevice : (kind == 1 ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost);
    hipError_t e = hipMemcpyAsync(dst, src, bytes, k, (hipStream_t)stream);
    if (e != hipSuccess) {
        ct_set_error("ct_memcpy_async: %s", hipGetErrorString(e));
        return CT_ERR_LAUNCH;
    }
    return CT_OK;
}

// (round 5: a KERNEL, not hipMemsetAsync -- inside a captured frame graph the runtime's memset node wrote garbage instead of
//  the value on this stack (zero_tracking streams replayed through a graph: tracking rows of 1e-21 .. 1e12 instead of 0,
//  tools/calls/dbg_zero.py; eager launches were fine).  Kernel nodes are what every other launch of the graph is.)
__global__ __launch_bounds__(256) void fill_words_kernel(unsigned *dst, unsigned pattern, size_t nwords)
{
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nwords; i += stride) dst[i] = pattern;
}

extern "C" int ct_memset_async(void *dst, int value, size_t bytes, void *stream)
{
    if (!dst && bytes) CT_FAIL_ARG("ct_memset_async: null pointer");
    if (bytes == 0) return CT_OK;
    if (((uintptr_t)dst & 3) == 0 && (bytes & 3) == 0) {
        const unsigned b = (unsigned)value & 0xffu;
        const size_t n = bytes / 4;
        const unsigned blocks = (unsigned)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
        hipLaunchKernelGGL(fill_words_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (unsigned *)dst,
                           b | (b << 8) | (b << 16) | (b << 24), n);
        CT_CHECK_LAUNCH("ct_memset_async");
        return CT_OK;
    }
    hipError_t e = hipMemsetAsync(dst, value, bytes, (hipStream_t)stream);
    if (e != hipSuccess) {
        ct_set_error("ct_memset_async: %s", hipGetErrorString(e));
        return CT_ERR_LAUNCH;
    }
    return CT_OK;
}

extern "C" int ct_stream_synchronize(void *stream)
{
    hipError_t e = hipStreamSynchronize((hipStream_t)stream);
    if (e != hipSuccess) {
        ct_set_error("ct_stream_synchronize: %s", hipGetErrorString(e));
        return CT_ERR_LAUNCH;
    }
    return CT_OK;
}

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

extern "C" int ct_calib_mfma(int blocks, int iters, float *out, void *stream)
{
    if (blocks <= 0 || iters <= 0 || !out) CT_FAIL_ARG("ct_calib_mfma: bad arguments");
    hipLaunchKernelGGL(calib_mfma_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, iters, out);
    CT_CHECK_LAUNCH("ct_calib_mfma");
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

extern "C" int ct_calib_chase(const unsigned *ring, int hops, unsigned start, unsigned long long *out, void *stream)
{
    if (!ring || hops <= 0 || !out) CT_FAIL_ARG("ct_calib_chase: bad arguments");
    hipLaunchKernelGGL(calib_chase_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, ring, hops, start, out);
    CT_CHECK_LAUNCH("ct_calib_chase");
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

extern "C" int ct_calib_stream(const void *src, void *dst, size_t bytes, int blocks, int inflight, void *stream)
{
    if (!src || !dst || blocks <= 0 || (bytes & 15)) CT_FAIL_ARG("ct_calib_stream: bad arguments");
    const size_t n = bytes / 16;
    if (inflight >= 4)
        hipLaunchKernelGGL(calib_stream_kernel<4>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float4 *)src, (float4 *)dst, n);
    else
        hipLaunchKernelGGL(calib_stream_kernel<1>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float4 *)src, (float4 *)dst, n);
    CT_CHECK_LAUNCH("ct_calib_stream");
    return CT_OK;
}

// (3) `n` dependent launches of a `blocks`-workgroup kernel that adds 1 to what its predecessor wrote: the kernel
//     boundary (dispatch + argument fetch + first load + drain) of this box, per launch.
__global__ __launch_bounds__(256) void calib_chain_kernel(float *buf)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    buf[i] = buf[i] + 1.0f;
}

extern "C" int ct_calib_launches(int n, int blocks, float *buf, void *stream)
{
    if (n <= 0 || blocks <= 0 || !buf) CT_FAIL_ARG("ct_calib_launches: bad arguments");
    for (int k = 0; k < n; ++k) hipLaunchKernelGGL(calib_chain_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, buf);
    CT_CHECK_LAUNCH("ct_calib_launches");
    return CT_OK;
}

// ---- end-of-frame flag in pinned host memory ----
__global__ void signal_host_kernel(int *flag, int value)
{
    __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

extern "C" int ct_signal_host(int *flag, int value, void *stream)
{
    if (!flag) CT_FAIL_ARG("ct_signal_host: null flag");
    hipLaunchKernelGGL(signal_host_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, flag, value);
    CT_CHECK_LAUNCH("ct_signal_host");
    return CT_OK;
}
