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

extern "C" int ct_memset_async(void *dst, int value, size_t bytes, void *stream)
{
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
