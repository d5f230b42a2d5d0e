// Calibration micro-benchmark (tools only): sustained v_mfma_f32_16x16x4_f32 rate of gfx950 in loops shaped like
// the DCN / conv kernels of this library -- (0) MFMAs alone, 2 / 4 accumulator chains; (1) + the A fragments read
// from LDS with ds_read_b128 every 8 MFMAs; (2) + a workgroup barrier every 16 / 32 MFMAs.
#include <hip/hip_runtime.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CH, int MODE, int PER_BAR>
__global__ __launch_bounds__(256) void mfma_loop(int iters, float *out)
{
    __shared__ __attribute__((aligned(16))) float lds[4096];
    const int tid = threadIdx.x;
    for (int i = tid; i < 4096; i += 256) lds[i] = 1.0f + i * 1e-6f;
    __syncthreads();
    f32x4 acc[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 b = f32x4{1.f, 1.0001f, 0.9999f, 1.0002f};
    f32x4 a0 = f32x4{1.f, 2.f, 3.f, 4.f}, a1 = a0 * 0.5f;
    const float *ap = lds + (tid & 63) * 4 + (tid >> 6) * 1024;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int blk = 0; blk < PER_BAR / 8; ++blk) {
            if (MODE >= 1) {
                a0 = *reinterpret_cast<const f32x4 *>(ap + ((it + blk) & 1) * 256);
                a1 = *reinterpret_cast<const f32x4 *>(ap + 512 + ((it + blk) & 1) * 256);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[(2 * e) % CH] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[e], b[e], acc[(2 * e) % CH], 0, 0, 0);
                acc[(2 * e + 1) % CH] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[e], b[e], acc[(2 * e + 1) % CH], 0, 0, 0);
            }
        }
        if (MODE >= 2) __syncthreads();
    }
    f32x4 s = acc[0];
#pragma unroll
    for (int c = 1; c < CH; ++c) s += acc[c];
    if (s[0] == 12345.678f) out[blockIdx.x * 256 + tid] = s[1] + s[2] + s[3];
}

extern "C" int mfma_peak(int variant, int blocks, int iters, float *out, void *stream)
{
    hipStream_t s = (hipStream_t)stream;
    switch (variant) {
    case 0: hipLaunchKernelGGL((mfma_loop<2, 0, 16>), dim3(blocks), dim3(256), 0, s, iters, out); break;
    case 1: hipLaunchKernelGGL((mfma_loop<4, 0, 16>), dim3(blocks), dim3(256), 0, s, iters, out); break;
    case 2: hipLaunchKernelGGL((mfma_loop<2, 1, 16>), dim3(blocks), dim3(256), 0, s, iters, out); break;
    case 3: hipLaunchKernelGGL((mfma_loop<2, 2, 16>), dim3(blocks), dim3(256), 0, s, iters, out); break;
    case 4: hipLaunchKernelGGL((mfma_loop<2, 2, 32>), dim3(blocks), dim3(256), 0, s, iters, out); break;
    case 5: hipLaunchKernelGGL((mfma_loop<4, 2, 32>), dim3(blocks), dim3(256), 0, s, iters, out); break;
    default: return 1;
    }
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
