// Micro-benchmark (tools only; VERDICT r4 "next 9", optional): fp32 products emulated on the bf16 matrix cores.
// fp32 MFMA on gfx950 runs at the vector rate (157 TFLOP/s, 1/16 of the bf16 MFMA rate): a contraction whose operands are
// split into bf16 terms x = hi + lo (+ lo2) and whose cross products are accumulated in fp32 by v_mfma_f32_16x16x32_bf16
// does the same arithmetic on the fast pipe -- 3 terms (hi*hi + hi*lo + lo*hi: ~2^-16 relative per product) or 6 terms
// (+ lo*lo + hi*lo2 + lo2*hi: ~2^-24).  This file measures, on one 16 x 16 output tile per wave: (a) what the result of a
// K = 576 contraction (a 3x3 conv over 64 channels) looks like against fp64 in each form, (b) the sustained issue rate of
// the three instruction mixes in a register-only loop (splitting included / excluded).  Nothing in the product uses it.
#include <hip/hip_runtime.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned short to_bf16(float x)       // round to nearest even
{
    unsigned u = __float_as_uint(x);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float from_bf16(unsigned short h) { return __uint_as_float((unsigned)h << 16); }

// x[8] -> hi / lo / lo2 fragments
__device__ __forceinline__ void split8(const float (&x)[8], bf16x8 &hi, bf16x8 &lo, bf16x8 &lo2)
{
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const unsigned short h = to_bf16(x[j]);
        const float r1 = x[j] - from_bf16(h);
        const unsigned short l = to_bf16(r1);
        const float r2 = r1 - from_bf16(l);
        hi[j] = (short)h; lo[j] = (short)l; lo2[j] = (short)to_bf16(r2);
    }
}

// MODE 0: v_mfma_f32_16x16x4_f32; 3 / 6: bf16 terms.  A [M][K] row-major, B [K][N] row-major, C [M][N]; one wave per tile.
template <int MODE>
__global__ __launch_bounds__(64) void gemm_tile_kernel(const float *A, const float *B, float *C, int M, int N, int K)
{
    const int lane = threadIdx.x, li = lane & 15, lg = lane >> 4;
    const int tn = blockIdx.x % (N / 16), tm = blockIdx.x / (N / 16);
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    if (MODE == 0) {
        for (int k0 = 0; k0 < K; k0 += 4)
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[(size_t)(tm * 16 + li) * K + k0 + lg], B[(size_t)(k0 + lg) * N + tn * 16 + li], acc, 0, 0, 0);
    } else {
        for (int k0 = 0; k0 < K; k0 += 32) {
            float a[8], b[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {               // lane (li, lg): A row li, B column li, k = k0 + 8 * lg + j
                a[j] = A[(size_t)(tm * 16 + li) * K + k0 + 8 * lg + j];
                b[j] = B[(size_t)(k0 + 8 * lg + j) * N + tn * 16 + li];
            }
            bf16x8 ah, al, al2, bh, bl, bl2;
            split8(a, ah, al, al2);
            split8(b, bh, bl, bl2);
            // small terms first: the fp32 accumulator then rounds them before the large term swamps them
            if (MODE == 6) {
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bl, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl2, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al2, bh, acc, 0, 0, 0);
            }
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc, 0, 0, 0);
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) C[(size_t)(tm * 16 + lg * 4 + e) * N + tn * 16 + li] = acc[e];
}

// issue-rate loops: per iteration one K = 32 slice of a 16 x 16 tile on 2 independent accumulator chains.
//   MODE 0: 8 + 8 fp32 MFMAs; MODE 3 / 6: 3 / 6 bf16 MFMAs per chain; SPLIT: the A operand (8 values per lane) is split into
//   its bf16 terms inside the loop (the B operand = weights is pre-split once, like a packed weight tensor would be)
template <int MODE, bool SPLIT>
__global__ __launch_bounds__(256) void rate_kernel(int iters, float *out)
{
    const int tid = threadIdx.x;
    f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
    float a[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = 1.0f + 0.013f * j + tid * 1e-5f;
    bf16x8 ah, al, al2, bh, bl, bl2;
    split8(a, ah, al, al2);
    split8(a, bh, bl, bl2);
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], a[7 - e], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[7 - e], a[e], acc1, 0, 0, 0);
            }
        } else {
            if (SPLIT) {
#pragma unroll
                for (int j = 0; j < 8; ++j) a[j] += acc0[j & 3] * 1e-30f;       // (keeps the split inside the loop)
                split8(a, ah, al, al2);
            }
            if (MODE == 6) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bl, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bl2, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl2, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al2, bh, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al2, bl, acc1, 0, 0, 0);
            }
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc1, 0, 0, 0);
        }
    }
    const f32x4 s = acc0 + acc1;
    if (s[0] == 12345.678f) out[blockIdx.x * 256 + tid] = s[1] + s[2] + s[3];
}

extern "C" int split_gemm(int mode, const float *A, const float *B, float *C, int M, int N, int K, void *stream)
{
    if (M % 16 || N % 16 || K % 32) return 1;
    const dim3 grid((M / 16) * (N / 16));
    hipStream_t s = (hipStream_t)stream;
    if (mode == 0) hipLaunchKernelGGL(gemm_tile_kernel<0>, grid, dim3(64), 0, s, A, B, C, M, N, K);
    else if (mode == 3) hipLaunchKernelGGL(gemm_tile_kernel<3>, grid, dim3(64), 0, s, A, B, C, M, N, K);
    else if (mode == 6) hipLaunchKernelGGL(gemm_tile_kernel<6>, grid, dim3(64), 0, s, A, B, C, M, N, K);
    else return 1;
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

extern "C" int split_rate(int mode, int split, int blocks, int iters, float *out, void *stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (mode == 0) hipLaunchKernelGGL((rate_kernel<0, false>), dim3(blocks), dim3(256), 0, s, iters, out);
    else if (mode == 3 && !split) hipLaunchKernelGGL((rate_kernel<3, false>), dim3(blocks), dim3(256), 0, s, iters, out);
    else if (mode == 3) hipLaunchKernelGGL((rate_kernel<3, true>), dim3(blocks), dim3(256), 0, s, iters, out);
    else if (mode == 6 && !split) hipLaunchKernelGGL((rate_kernel<6, false>), dim3(blocks), dim3(256), 0, s, iters, out);
    else if (mode == 6) hipLaunchKernelGGL((rate_kernel<6, true>), dim3(blocks), dim3(256), 0, s, iters, out);
    else return 1;
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
