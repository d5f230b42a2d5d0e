// oracle/_ref shim -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
//
// Plain-C entry point around the REFERENCE's own compiled DCNv2 CPU forward, `dcn_v2_cpu_forward`, declared in
// upstream's src/cpu/vision.h and defined in src/cpu/dcn_v2_cpu.cpp (+ dcn_v2_im2col_cpu.cpp) of the un-vendored
// submodule CharlesShang/DCNv2 (reference .gitmodules:10-13, path src/lib/model/networks/DCNv2).  `make -C oracle ref`
// compiles those two upstream sources FROM WHERE THEY LIE together with this file into oracle/_ref/libdcn_v2_ref.so;
// nothing of upstream is copied into this repository.  The export has the contract of oracle/dcn_v2_ref.c's
// dcn_v2_forward_ref, so tests/test_oracle_dcn.py can put the upstream build, the C restatement and the two torch
// restatements side by side -- the day the submodule is checked out, SURVEY row a6 turns from "parity unpinned" to
// pinned by running that one make target.
//
// The submodule is ABSENT from /root/reference today: this file has only been compiled against a stub declaration of
// the upstream prototype (the one below, as published on upstream's master branch).
#include <ATen/ATen.h>

#include "cpu/vision.h"   // at::Tensor dcn_v2_cpu_forward(input, weight, bias, offset, mask, kh, kw, sh, sw, ph, pw, dh, dw, deformable_group)

extern "C" int dcn_v2_forward_upstream(const float *x, const float *offset, const float *mask, const float *weight,
                                       const float *bias, float *out, int B, int Ci, int H, int W, int Co, int kh,
                                       int kw, int stride, int pad, int dil)
{
    try {
        const auto opt = at::TensorOptions().dtype(at::kFloat).device(at::kCPU);
        const int K = kh * kw;
        const int Ho = (H + 2 * pad - (dil * (kh - 1) + 1)) / stride + 1;
        const int Wo = (W + 2 * pad - (dil * (kw - 1) + 1)) / stride + 1;
        // upstream reads contiguous NCHW tensors; from_blob wraps the caller's buffers without copying
        at::Tensor tx = at::from_blob(const_cast<float *>(x), {B, Ci, H, W}, opt);
        at::Tensor to = at::from_blob(const_cast<float *>(offset), {B, 2 * K, Ho, Wo}, opt);
        at::Tensor tm = at::from_blob(const_cast<float *>(mask), {B, K, Ho, Wo}, opt);
        at::Tensor tw = at::from_blob(const_cast<float *>(weight), {Co, Ci, kh, kw}, opt);
        at::Tensor tb = bias ? at::from_blob(const_cast<float *>(bias), {Co}, opt) : at::zeros({Co}, opt);
        at::Tensor y = dcn_v2_cpu_forward(tx, tw, tb, to, tm, kh, kw, stride, stride, pad, pad, dil, dil, 1).contiguous();
        if (y.numel() != (int64_t)B * Co * Ho * Wo) return 2;
        std::memcpy(out, y.data_ptr<float>(), sizeof(float) * (size_t)y.numel());
        return 0;
    } catch (const std::exception &) {
        return 1;
    }
}
