/* Oracle (TEST INFRASTRUCTURE ONLY): scalar C restatement of the DCNv2 CPU
 * forward.  PARITY UNPINNED -- the op lives in the un-vendored submodule
 * CharlesShang/DCNv2 (reference .gitmodules:10-13, branch master, no SHA); this
 * follows its published algorithm: src/cpu/dcn_v2_im2col_cpu.cpp
 * (dmcn_im2col_bilinear_cpu, modulated_deformable_im2col_cpu_kernel) and
 * src/cpu/dcn_v2_cpu.cpp (per-sample im2col + GEMM + bias), SURVEY.md Appendix B.
 * Call site in the reference: src/lib/model/networks/dla.py:513,516.
 *
 * A second, independent restatement (oracle/dcn_v2.py, vectorised torch) must
 * agree with this one; tests/test_oracle_dcn.py checks that.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

static float im2col_bilinear(const float *im, int height, int width, float h, float w)
{
    int h_low = (int)floorf(h);
    int w_low = (int)floorf(w);
    int h_high = h_low + 1;
    int w_high = w_low + 1;
    float lh = h - (float)h_low;
    float lw = w - (float)w_low;
    float hh = 1.0f - lh, hw = 1.0f - lw;
    float v1 = 0.f, v2 = 0.f, v3 = 0.f, v4 = 0.f;
    if (h_low >= 0 && w_low >= 0) v1 = im[h_low * width + w_low];
    if (h_low >= 0 && w_high <= width - 1) v2 = im[h_low * width + w_high];
    if (h_high <= height - 1 && w_low >= 0) v3 = im[h_high * width + w_low];
    if (h_high <= height - 1 && w_high <= width - 1) v4 = im[h_high * width + w_high];
    float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
    return w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4;
}

/* x [B,Ci,H,W]; offset [B,2*kh*kw,Ho,Wo]; mask [B,kh*kw,Ho,Wo]; weight [Co,Ci,kh,kw];
 * bias [Co] or NULL; out [B,Co,Ho,Wo].  Returns 0, or -1 on allocation failure. */
int dcn_v2_forward_ref(const float *x, const float *offset, const float *mask,
                       const float *weight, const float *bias, float *out,
                       int B, int Ci, int H, int W, int Co, int kh, int kw,
                       int stride, int pad, int dil)
{
    const int Ho = (H + 2 * pad - (dil * (kh - 1) + 1)) / stride + 1;
    const int Wo = (W + 2 * pad - (dil * (kw - 1) + 1)) / stride + 1;
    const int K = Ci * kh * kw;
    const long N = (long)Ho * Wo;
    float *col = (float *)malloc(sizeof(float) * (size_t)K * (size_t)N);
    if (!col) return -1;
    for (int b = 0; b < B; ++b) {
        const float *xb = x + (long)b * Ci * H * W;
        const float *ob = offset + (long)b * 2 * kh * kw * N;
        const float *mb = mask + (long)b * kh * kw * N;
        /* modulated deformable im2col: column row index = ci*kh*kw + i*kw + j */
        for (int ci = 0; ci < Ci; ++ci)
            for (int i = 0; i < kh; ++i)
                for (int j = 0; j < kw; ++j) {
                    const int k = i * kw + j;
                    float *crow = col + ((long)ci * kh * kw + k) * N;
                    for (int ho = 0; ho < Ho; ++ho)
                        for (int wo = 0; wo < Wo; ++wo) {
                            const long p = (long)ho * Wo + wo;
                            const float oh = ob[(2 * k) * N + p];
                            const float ow = ob[(2 * k + 1) * N + p];
                            const float m = mb[k * N + p];
                            const float h_im = (float)(ho * stride - pad + i * dil) + oh;
                            const float w_im = (float)(wo * stride - pad + j * dil) + ow;
                            float val = 0.f;
                            if (h_im > -1 && w_im > -1 && h_im < H && w_im < W)
                                val = im2col_bilinear(xb + (long)ci * H * W, H, W, h_im, w_im);
                            crow[p] = val * m;
                        }
                }
        /* out[b] = W[Co,K] @ col[K,N] + bias */
        float *outb = out + (long)b * Co * N;
        for (int co = 0; co < Co; ++co) {
            float *orow = outb + (long)co * N;
            const float bv = bias ? bias[co] : 0.f;
            for (long p = 0; p < N; ++p) orow[p] = bv;
            for (int kk = 0; kk < K; ++kk) {
                const float wv = weight[(long)co * K + kk];
                const float *crow = col + (long)kk * N;
                for (long p = 0; p < N; ++p) orow[p] += wv * crow[p];
            }
        }
    }
    free(col);
    return 0;
}
