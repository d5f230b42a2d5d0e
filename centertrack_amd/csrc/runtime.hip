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

// (round 5: KERNELS, not hipMemsetAsync -- inside a captured frame graph the runtime's memset node wrote garbage instead of the
//  value (zero_tracking streams replayed through a graph: tracking rows of 1e-21 .. 1e12 instead of 0; eager launches were
//  fine; root cause inside the runtime not established).  Regression test: tests/test_hip_sparse_heads.py -- the
//  zero_tracking case compares a graph-replayed stream against the dense path.  Kernel nodes are what every other launch of
//  the graph is; sizes or addresses that are not word multiples go through the byte kernel, never through the runtime.)
__global__ __launch_bounds__(256) void fill_words_kernel(unsigned *dst, unsigned pattern, size_t nwords)
{
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nwords; i += stride) dst[i] = pattern;
}

__global__ __launch_bounds__(256) void fill_bytes_kernel(unsigned char *dst, unsigned char value, size_t nbytes)
{
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nbytes; i += stride) dst[i] = value;
}

extern "C" int ct_memset_async(void *dst, int value, size_t bytes, void *stream)
{
    if (!dst && bytes) CT_FAIL_ARG("ct_memset_async: null pointer");
    if (bytes == 0) return CT_OK;
    const unsigned b = (unsigned)value & 0xffu;
    if (((uintptr_t)dst & 3) == 0 && (bytes & 3) == 0) {
        const size_t n = bytes / 4;
        const unsigned blocks = (unsigned)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
        hipLaunchKernelGGL(fill_words_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (unsigned *)dst,
                           b | (b << 8) | (b << 16) | (b << 24), n);
    } else {
        const unsigned blocks = (unsigned)((bytes + 255) / 256 < 2048 ? (bytes + 255) / 256 : 2048);
        hipLaunchKernelGGL(fill_bytes_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (unsigned char *)dst, (unsigned char)b, bytes);
    }
    CT_CHECK_LAUNCH("ct_memset_async");
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
