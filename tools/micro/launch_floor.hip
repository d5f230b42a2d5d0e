// Calibration for DESIGN.md section 8: what a dependent kernel boundary costs inside a captured HIP graph, and what a
// device-wide barrier inside ONE kernel costs (256 co-resident workgroups, atomic counter + acquire / release fences
// across the 8 XCDs) -- the two ways a chain of small dependent layers can be sequenced on an MI355X.
#include <hip/hip_runtime.h>

extern "C" {

__global__ __launch_bounds__(256) void tiny_kernel(float *buf, int round)
{
    // every workgroup reads what its neighbour wrote in the previous launch and writes its own cell
    const int g = gridDim.x, b = blockIdx.x;
    if (threadIdx.x == 0) buf[(round & 1) * 4096 + b] = buf[((round + 1) & 1) * 4096 + (b + 1) % g] + 1.0f;
}

int launch_tiny(int blocks, int rounds, float *buf, void *stream)
{
    for (int r = 0; r < rounds; ++r) hipLaunchKernelGGL(tiny_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, buf, r);
    return (int)hipGetLastError();
}

__global__ __launch_bounds__(256) void barrier_kernel(float *buf, unsigned *ctr, unsigned *err, int rounds)
{
    const unsigned g = gridDim.x, b = blockIdx.x;
    for (int r = 0; r < rounds; ++r) {
        if (threadIdx.x == 0) {
            buf[(r & 1) * 4096 + b] = buf[((r + 1) & 1) * 4096 + (b + 1) % g] + 1.0f;
            __threadfence();                                            // release: my cell before my arrival
            atomicAdd(ctr, 1u);
            const unsigned target = g * (unsigned)(r + 1);
            unsigned spins = 0;
            while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
                if (++spins > (1u << 22)) { atomicAdd(err, 1u); break; }      // (bounded: never hangs the box)
                __builtin_amdgcn_s_sleep(1);
            }
            __threadfence();                                            // acquire: the neighbours' cells
        }
        __syncthreads();
    }
}

// XCD-local variant: workgroups are dealt to the 8 XCDs round-robin (id % 8; checked against XCC_ID below), every XCD has its own
// L2, and atomics execute in the L2 -- so the workgroups of ONE XCD can meet in their own L2 without the agent-scope write-backs
// and cache-bypassing polls a device-wide barrier needs: counter per XCD (own cache line), workgroup-scope atomics and sc0 loads
// (L1 bypass, served by the local L2), stores only waited for (write-through L1), neighbour = the next workgroup of the same XCD.
__global__ __launch_bounds__(256) void barrier_xcd_kernel(float *buf, unsigned *ctr, unsigned *err, int rounds)
{
    const unsigned g = gridDim.x, b = blockIdx.x;
    const unsigned x = b & 7u, per = g >> 3;                                // (gridDim.x is a multiple of 8)
    unsigned *my = ctr + 32 * x;
    if (threadIdx.x == 0) {
        const unsigned xcc = __builtin_amdgcn_s_getreg(((4 - 1) << 11) | 20) & 15u;     // XCC_ID
        if (xcc != x) atomicAdd(err + 1, 1u);
    }
    const unsigned nb = ((b >> 3) + 1) % per * 8 + x;                        // next workgroup of this XCD
    for (int r = 0; r < rounds; ++r) {
        if (threadIdx.x == 0) {
            const float v = __hip_atomic_load(&buf[((r + 1) & 1) * 4096 + nb], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_store(&buf[(r & 1) * 4096 + b], v + 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __builtin_amdgcn_s_waitcnt(0);                                  // my cell has left for the L2
            __hip_atomic_fetch_add(my, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const unsigned target = per * (unsigned)(r + 1);
            unsigned spins = 0;
            while (__hip_atomic_load(my, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) {
                if (++spins > (1u << 12)) { atomicAdd(err, 1u); break; }     // (a few ms at most: the premise may not hold)
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
    }
}

int launch_barrier_xcd(int blocks, int rounds, float *buf, unsigned *ctr, unsigned *err, void *stream)
{
    hipLaunchKernelGGL(barrier_xcd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, buf, ctr, err, rounds);
    return (int)hipGetLastError();
}

int launch_barrier(int blocks, int rounds, float *buf, unsigned *ctr, unsigned *err, void *stream)
{
    hipLaunchKernelGGL(barrier_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, buf, ctr, err, rounds);
    return (int)hipGetLastError();
}

}
