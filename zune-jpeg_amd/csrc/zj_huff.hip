// zj_huff.hip -- kernels of the GPU entropy stage (zj_huff.h: algorithm; zj_huff_device.h: the per-thread code).
// gfx950: one sub-sequence per lane, 256 per workgroup; the workgroup's stream bytes (<= 32 KB) and the decoding tables
// (<= 18 KB) are staged in LDS with coalesced loads, after which a lane touches global memory only to publish its exit
// state or to scatter coefficients.
#include <hip/hip_runtime.h>

#include "zj_huff_device.h"
#include "zj_launch.h"

namespace zj {

__global__ __launch_bounds__(HUFF_WG) void zj_huff_sync_kernel(HuffArgs a)
{
    __shared__ HuffLds L;
    const uint32_t i = blockIdx.x * HUFF_WG + threadIdx.x;
    const HuffScan* g = huff_hdr(a.blob);
    const uint32_t nsub = g->nsub;
    const bool need = huff_sync_needed(a, i, nsub, huff_subs(a.blob));
    if (!__syncthreads_or(need ? 1 : 0)) { // nobody's entry state moved: no staging either
        if (i < nsub) a.changed[(size_t)(a.round & 1) * nsub + i] = 0;
        return;
    }
    huff_stage<HUFF_WG>(a.blob, (int)blockIdx.x, (int)threadIdx.x, L);
    __syncthreads();
    huff_sync_thread(a, L, i);
}

__global__ __launch_bounds__(HUFF_WG) void zj_huff_write_kernel(HuffArgs a)
{
    __shared__ HuffLds L;
    huff_stage<HUFF_WG>(a.blob, (int)blockIdx.x, (int)threadIdx.x, L);
    __syncthreads();
    huff_write_thread(a, L, blockIdx.x * HUFF_WG + threadIdx.x);
}

constexpr int HUFF_SCAN_NT = 1024;
__global__ __launch_bounds__(HUFF_SCAN_NT) void zj_huff_scan_kernel(HuffArgs a)
{
    __shared__ HuffAgg agg[HUFF_SCAN_NT];
    const uint32_t nsub = huff_hdr(a.blob)->nsub;
    const uint32_t chunk = (nsub + HUFF_SCAN_NT - 1) / HUFF_SCAN_NT;
    const uint32_t t = threadIdx.x;
    agg[t] = huff_scan_chunk(a, t, chunk);
    __syncthreads();
    if (t == 0) huff_scan_combine(agg, HUFF_SCAN_NT);
    __syncthreads();
    huff_scan_apply(a, t, chunk, agg[t]);
}

__global__ __launch_bounds__(256) void zj_huff_cut_kernel(HuffArgs a)
{
    uint32_t first = 0;
    const uint32_t pieces = huff_cut_plan(a, &first);
    for (uint32_t p = threadIdx.x; p < pieces; p += 256) huff_cut_clear(a, first, p);
}

hipError_t launch_huff_sync(const HuffArgs& a, uint32_t nsub, hipStream_t s)
{
    hipLaunchKernelGGL(zj_huff_sync_kernel, dim3((nsub + HUFF_WG - 1) / HUFF_WG), dim3(HUFF_WG), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_huff_finish(const HuffArgs& a, uint32_t nsub, hipStream_t s)
{
    hipLaunchKernelGGL(zj_huff_scan_kernel, dim3(1), dim3(HUFF_SCAN_NT), 0, s, a);
    hipLaunchKernelGGL(zj_huff_write_kernel, dim3((nsub + HUFF_WG - 1) / HUFF_WG), dim3(HUFF_WG), 0, s, a);
    hipLaunchKernelGGL(zj_huff_cut_kernel, dim3(1), dim3(256), 0, s, a);
    return hipGetLastError();
}

} // namespace zj
