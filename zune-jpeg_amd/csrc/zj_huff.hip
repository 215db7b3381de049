// zj_huff.hip -- kernels of the GPU entropy stage (zj_huff.h: algorithm; zj_huff_device.h: the per-thread code).
// gfx950: one sub-sequence per lane, 256 per workgroup; every lane that has work stages its own 144 stream bytes in a
// private stretch of LDS (37 KB per workgroup, odd stride), the workgroup stages the decoding tables (<= 18 KB); after
// that a lane touches global memory only to publish its exit state or to scatter coefficients.
#include <hip/hip_runtime.h>

#include "zj_huff_device.h"
#include "zj_launch.h"

namespace zj {

// Every kernel takes the working sets of a BATCH of scans (blockIdx.y picks one): the scans of several files run as one
// launch each, sized for the largest; workgroups past a scan's end leave at once.
__global__ __launch_bounds__(HUFF_WG) void zj_huff_sync_kernel(const HuffBatch args, int round)
{
    __shared__ HuffLds L;
    HuffArgs a = args.a[blockIdx.y];
    a.round = round;
    const HuffScan* g = huff_hdr(a.blob);
    const uint32_t nsub = g->nsub;
    // the rounds are launched ahead of any look at their outcome: once a round changed nothing, the rest are no-ops
    if (round >= 2 && a.ctl[HUFF_CTL_ROUND0 + round - 1] == 0) return;
    // rounds 0 and 1 decode (nearly) every sub-sequence in place; later rounds the entries of their work list
    uint32_t work = nsub;
    if (round >= 2) {
        uint32_t entries = a.ctl[HUFF_CTL_ROUND0 + round - 1];
        if (entries > HUFF_LIST_FACTOR * nsub) entries = HUFF_LIST_FACTOR * nsub;
        work = entries * huff_spread(a, nsub, entries);
    }
    if (blockIdx.x * HUFF_WG >= work) return;
    const uint32_t i = huff_sync_pick(a, blockIdx.x * HUFF_WG + threadIdx.x, nsub, huff_subs(a.blob));
    if (round == 0) huff_clear_planes(a, blockIdx.x * HUFF_WG + threadIdx.x, ((nsub + HUFF_WG - 1) / HUFF_WG) * HUFF_WG);
    huff_stage<HUFF_WG>(a.blob, (int)threadIdx.x, i < nsub, i, L);
    __syncthreads();
    huff_sync_thread(a, L, threadIdx.x, i);
}

// the periodic-run rule between two rounds (zj_huff.h)
__global__ __launch_bounds__(HUFF_WG) void zj_huff_periodic_kernel(const HuffBatch args, int next_round)
{
    const HuffArgs a = args.a[blockIdx.y];
    if (huff_hdr(a.blob)->nper == 0) return;
    huff_periodic_thread(a, blockIdx.x * HUFF_WG + threadIdx.x, next_round);
}

__global__ __launch_bounds__(HUFF_WG) void zj_huff_write_kernel(const HuffBatch args)
{
    __shared__ HuffLds L;
    const HuffArgs a = args.a[blockIdx.y];
    const uint32_t nsub = huff_hdr(a.blob)->nsub, i = blockIdx.x * HUFF_WG + threadIdx.x;
    if (blockIdx.x * HUFF_WG >= nsub) return;
    huff_stage<HUFF_WG>(a.blob, (int)threadIdx.x, i < nsub, i, L);
    __syncthreads();
    huff_write_thread(a, L, threadIdx.x, i);
}

__global__ __launch_bounds__(HUFF_SCAN_WG) void zj_huff_scan_kernel(const HuffBatch args)
{
    // Hillis-Steele over the workgroup's 1024 elements, double-buffered in LDS
    __shared__ HuffAgg buf[2][HUFF_SCAN_WG];
    __shared__ uint32_t ticket;
    const HuffArgs a = args.a[blockIdx.y];
    const uint32_t nwg = (huff_hdr(a.blob)->nsub + HUFF_SCAN_WG - 1) / HUFF_SCAN_WG; // of THIS scan
    if (blockIdx.x >= nwg) return;
    const uint32_t t = threadIdx.x, i = blockIdx.x * HUFF_SCAN_WG + t;
    buf[0][t] = huff_scan_element(a, i);
    __syncthreads();
    int cur = 0;
    for (uint32_t d = 1; d < HUFF_SCAN_WG; d <<= 1) {
        HuffAgg v = buf[cur][t];
        if (t >= d) v = huff_scan_op(buf[cur][t - d], v);
        buf[cur ^ 1][t] = v;
        cur ^= 1;
        __syncthreads();
    }
    huff_scan_store(a, i, t ? buf[cur][t - 1] : huff_scan_identity());
    if (t == HUFF_SCAN_WG - 1) {
        a.wgagg[blockIdx.x] = buf[cur][t];
        __threadfence();
        ticket = atomicAdd(&a.ctl[HUFF_CTL_TICKET], 1u);
    }
    __syncthreads();
    if (ticket == nwg - 1 && t == 0) { // the last workgroup to finish: every total is visible
        __threadfence();
        huff_scan_totals(a, nwg);
    }
}

__global__ __launch_bounds__(256) void zj_huff_cut_kernel(const HuffBatch args)
{
    const HuffArgs a = args.a[blockIdx.y];
    uint32_t first = 0;
    const uint32_t pieces = huff_cut_plan(a, &first);
    for (uint32_t p = threadIdx.x; p < pieces; p += 256) huff_cut_clear(a, first, p);
}

// periodic: some scan of the batch has periodic runs -- the rule runs in front of the rounds huff_periodic_before() names
hipError_t launch_huff_sync(const HuffBatch& b, int njobs, uint32_t max_nsub, int round, bool periodic, hipStream_t s)
{
    if (periodic && huff_periodic_before(round))
        hipLaunchKernelGGL(zj_huff_periodic_kernel, dim3((max_nsub + HUFF_WG - 1) / HUFF_WG, njobs), dim3(HUFF_WG), 0, s, b, round);
    // from round 2 on a work list may hold up to HUFF_LIST_FACTOR x nsub entries (workgroups past its end leave at once)
    const uint32_t threads = round >= 2 && periodic ? HUFF_LIST_FACTOR * max_nsub : max_nsub;
    hipLaunchKernelGGL(zj_huff_sync_kernel, dim3((threads + HUFF_WG - 1) / HUFF_WG, njobs), dim3(HUFF_WG), 0, s, b, round);
    return hipGetLastError();
}
hipError_t launch_huff_finish(const HuffBatch& b, int njobs, uint32_t max_nsub, hipStream_t s)
{
    hipLaunchKernelGGL(zj_huff_scan_kernel, dim3((max_nsub + HUFF_SCAN_WG - 1) / HUFF_SCAN_WG, njobs), dim3(HUFF_SCAN_WG), 0, s, b);
    hipLaunchKernelGGL(zj_huff_write_kernel, dim3((max_nsub + HUFF_WG - 1) / HUFF_WG, njobs), dim3(HUFF_WG), 0, s, b);
    hipLaunchKernelGGL(zj_huff_cut_kernel, dim3(1, njobs), dim3(256), 0, s, b);
    return hipGetLastError();
}

} // namespace zj
