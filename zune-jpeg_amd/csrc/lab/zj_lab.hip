// zj_lab.hip -- "kernel lab": the IDCT and colour stages timed in isolation (registers/LDS only, no
// HBM traffic) in several formulations, to pick the cheapest exact one on gfx950.  tools/lab.py.
// Not part of the decode path.
#include <hip/hip_runtime.h>

#include "../zj_device.h"
#include "zj_lab_launch.h"

namespace zj {

// ---- the block transforms in isolation -------------------------------------------------------------
// MODE 0: idct_block (wide: 24-bit multiply-adds, exact for every input; round 1's transform)
// MODE 1: classify_block + idct_block_packed (v_dot2_i32_i16 on 16-bit pairs), what a full block costs in round 2
// MODE 2: idct_block_packed alone
struct LabTab { uint32_t t[3 * TAB_DW]; };
template <int MODE>
__global__ __launch_bounds__(256) void lab_idct(const LabTab tabs, int* out, int iters)
{
    __shared__ uint32_t tab_l[3 * TAB_DW];
    if (threadIdx.x < 3 * TAB_DW) tab_l[threadIdx.x] = tabs.t[threadIdx.x];
    __syncthreads();
    const uint32_t* tab = tab_l + TAB_DW * (threadIdx.x % 3);
    U4 raw[8];
    uint32_t seed = threadIdx.x * 2654435761u + blockIdx.x;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        seed = seed * 1664525u + 1013904223u; raw[i].x = seed & 0x00070007;
        seed = seed * 1664525u + 1013904223u; raw[i].y = seed & 0x00030003;
        seed = seed * 1664525u + 1013904223u; raw[i].z = seed & 0x00010001;
        seed = seed * 1664525u + 1013904223u; raw[i].w = seed & 0x00010001;
    }
    uint32_t acc = 0;
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) {
            U4 px[8];
            idct_block(raw, reinterpret_cast<const uint16_t*>(tab), px);
#pragma unroll
            for (int i = 0; i < 8; i++) { raw[i].x = px[i].x & 0x00070007; raw[i].y = px[i].y & 0x00030003; raw[i].z = px[i].z & 0x00010001; raw[i].w = px[i].w & 0x00010001; }
        } else {
            uint32_t b[16];
            const int cls = MODE == 1 ? classify_block(reinterpret_cast<const uint32_t*>(raw), tab + 32) : 1;
            if (cls == 1) idct_block_packed(raw, tab, b);
            else {
#pragma unroll
                for (int i = 0; i < 16; i++) b[i] = (uint32_t)cls;
            }
            // dependent chain: the next block's coefficients come from this block's pixels (kept small: guard passes)
#pragma unroll
            for (int i = 0; i < 8; i++) { raw[i].x = b[2 * i] & 0x00070007; raw[i].y = b[2 * i + 1] & 0x00030003; raw[i].z = (b[2 * i] >> 8) & 0x00010001; raw[i].w = (b[2 * i + 1] >> 8) & 0x00010001; }
        }
        acc += raw[0].x;
    }
#pragma unroll
    for (int i = 0; i < 8; i++) acc ^= raw[i].x ^ raw[i].y ^ raw[i].z ^ raw[i].w;
    if (acc == (uint32_t)iters * 0x00010001u) out[blockIdx.x * blockDim.x + threadIdx.x] = (int)acc;
}

// ---- colour stage in isolation: 16 px per lane per iteration (HV arrangement) -----------------
template <int V>
__global__ __launch_bounds__(256) void lab_color(int* out, int iters)
{
    uint32_t yp[8], cbp[8], crp[8];
    uint32_t seed = threadIdx.x * 2654435761u + blockIdx.x;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        seed = seed * 1664525u + 1013904223u; yp[i] = seed & 0x00ff00ff;
        seed = seed * 1664525u + 1013904223u; cbp[i] = seed & 0x00ff00ff;
        seed = seed * 1664525u + 1013904223u; crp[i] = seed & 0x00ff00ff;
    }
    uint32_t acc = 0;
    for (int it = 0; it < iters; it++) {
        uint32_t d[12];
        RGB2 c[8];
#pragma unroll
        for (int k = 0; k < 8; k++) c[k] = ycc_to_rgb_pair(yp[k], cbp[k], crp[k]);
#pragma unroll
        for (int k = 0; k < 4; k++) pack_rgb4_eo(c[k], c[4 + k], d[3 * k], d[3 * k + 1], d[3 * k + 2]);
#pragma unroll
        for (int k = 0; k < 8; k++) { yp[k] = d[k] & 0x00ff00ff; cbp[k] = (d[k] >> 8) & 0x00ff00ff; crp[k] ^= d[(k + 4) % 12] & 0x00ff00ff; }
        acc ^= d[8] ^ d[9] ^ d[10] ^ d[11];
    }
    if (acc == 0x12345678u) out[blockIdx.x * blockDim.x + threadIdx.x] = (int)acc;
}

// ---- memory-pattern lab: 384 bytes in, 384 bytes out per lane ----------------------------------
// RD 0: coalesced 16-byte reads (lane stride 16 B)        1: the decoder's pattern, 8 x 16 B per 128-B block (lane stride 128 B)
// WR 0: coalesced 16-byte writes                           1: the decoder's pattern, 3 x 16 B per 48-B item (lane stride 48 B)
//    2: 48-B items re-laid through LDS so every store instruction writes contiguous 16-B chunks
template <int RD, int WR>
__global__ __launch_bounds__(256) void lab_mem(const U4* __restrict__ in, U4* __restrict__ out, long long T)
{
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    U4 v[24];
    if (RD == 0) {
#pragma unroll
        for (int c = 0; c < 24; c++) v[c] = in[c * T + t];
    } else {
#pragma unroll
        for (int b = 0; b < 3; b++)
#pragma unroll
            for (int k = 0; k < 8; k++) v[b * 8 + k] = in[(b * T + t) * 8 + k];
    }
    if (WR == 0) {
#pragma unroll
        for (int c = 0; c < 24; c++) out[c * T + t] = v[c];
    } else if (WR == 1) {
#pragma unroll
        for (int i = 0; i < 8; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) out[(i * T + t) * 3 + j] = v[i * 3 + j];
    } else {
        // per-wave LDS transpose: 64 lanes x 48 B = 3 KB contiguous in memory per item index i
        __shared__ U4 stage[4][192];
        U4* st = stage[threadIdx.x >> 6];
        const int lane = threadIdx.x & 63;
        const long long wave_base = (t - lane) * 3; // in U4 units, for item i add i*T*3
#pragma unroll
        for (int i = 0; i < 8; i++) {
            st[lane * 3 + 0] = v[i * 3 + 0]; st[lane * 3 + 1] = v[i * 3 + 1]; st[lane * 3 + 2] = v[i * 3 + 2];
            __builtin_amdgcn_wave_barrier();
            const U4 a = st[lane], b = st[64 + lane], c = st[128 + lane];
            __builtin_amdgcn_wave_barrier();
            U4* o = out + i * T * 3 + wave_base;
            o[lane] = a; o[64 + lane] = b; o[128 + lane] = c;
        }
    }
}

typedef void (*lab_fn)(const LabTab, int*, int);
static const struct { const char* name; lab_fn fn; } LAB[] = {
    {"idct wide   (24-bit multiply-adds, round 1)", lab_idct<0>},
    {"idct packed (v_dot2_i32_i16) incl. classify", lab_idct<1>},
    {"idct packed (v_dot2_i32_i16) transform only", lab_idct<2>},
};
// tile-shaped copy: each workgroup writes 32 row segments of S bytes (row pitch 12288 B = 4096 px RGB)
// and reads the same amount contiguously (RDT = 0) or as 4+2+2 block-row segments like the decoder (RDT = 1)
template <int S, int XCD>
__global__ __launch_bounds__(256) void lab_tile(const U4* __restrict__ in, U4* __restrict__ out, long long ntiles)
{
    constexpr int CPR = S / 16;            // 16-byte chunks per row segment
    constexpr int NCH = 32 * CPR;          // chunks per tile
    constexpr int TPR = 12288 / S;         // tiles per row
    long long id = blockIdx.x;
    if (XCD && (ntiles & 7) == 0) id = (id & 7) * (ntiles >> 3) + (id >> 3);
    const long long strip = id / TPR, tx = id % TPR;
    const U4* src = in + id * NCH;
    U4* dst = out + strip * 32 * 768 + tx * CPR;
    U4 v[(NCH + 255) / 256];
#pragma unroll
    for (int k = 0; k < (NCH + 255) / 256; k++) { const int c = threadIdx.x + 256 * k; if (c < NCH) v[k] = src[c]; }
#pragma unroll
    for (int k = 0; k < (NCH + 255) / 256; k++) {
        const int c = threadIdx.x + 256 * k;
        if (c < NCH) dst[(c / CPR) * 768 + (c % CPR)] = v[k];
    }
}

// the same 32 KB per workgroup written as ROWS row segments of S bytes (ROWS * S = 32768) at the 12288-byte pitch:
// isolates the row-segment length from the bytes a workgroup moves
template <int S>
__global__ __launch_bounds__(256) void lab_tile32k(const U4* __restrict__ in, U4* __restrict__ out, long long ntiles)
{
    constexpr int ROWS = 32768 / S, CPR = S / 16, TPR = 12288 / S; // S in {512, 1024, 2048, 4096}; 12288 / S whole
    long long id = blockIdx.x;
    if ((ntiles & 7) == 0) id = (id & 7) * (ntiles >> 3) + (id >> 3);
    const long long band = id / TPR, tx = id % TPR; // a band = ROWS full rows
    const U4* src = in + id * 2048;
    U4* dst = out + band * ROWS * 768 + tx * CPR;
    U4 v[8];
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = src[threadIdx.x + 256 * k];
#pragma unroll
    for (int k = 0; k < 8; k++) { const int c = threadIdx.x + 256 * k; dst[(c / CPR) * 768 + (c % CPR)] = v[k]; }
}

// plain streaming copies: N x 16 bytes, 1 or 4 chunks per lane (consecutive lanes -> consecutive chunks)
template <int PER>
__global__ __launch_bounds__(256) void lab_stream(const U4* __restrict__ in, U4* __restrict__ out, long long n)
{
    const long long base = ((long long)blockIdx.x * 256 * PER) + threadIdx.x;
    U4 v[PER];
#pragma unroll
    for (int k = 0; k < PER; k++) v[k] = in[base + 256 * k];
#pragma unroll
    for (int k = 0; k < PER; k++) out[base + 256 * k] = v[k];
}

// What makes a burst / tile copy slower than the 1-load-1-store stream?  Same bytes per workgroup (32 KB) in
// three shapes: MODE 0 lane-contiguous 128 B per lane (like one block per lane), 8 loads then 8 stores;
// MODE 1 the lab_stream<8> addresses, but load k+1 is issued before store k and nothing else (a software pipeline);
// MODE 2 a loop of eight 1-load-1-store rounds with a wait in between (fine-grained alternation, long-lived waves)
template <int MODE>
__global__ __launch_bounds__(256) void lab_stream8(const U4* __restrict__ in, U4* __restrict__ out, long long n)
{
    const long long wg = (long long)blockIdx.x * 256 * 8;
    if (MODE == 0) {
        const long long base = wg + threadIdx.x * 8;
        U4 v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = in[base + k];
#pragma unroll
        for (int k = 0; k < 8; k++) out[base + k] = v[k];
    } else if (MODE == 1) {
        const long long base = wg + threadIdx.x;
        U4 cur = in[base];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            U4 nxt = cur;
            if (k < 7) nxt = in[base + 256 * (k + 1)];
            asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
            out[base + 256 * k] = cur;
            cur = nxt;
        }
    } else {
        const long long base = wg + threadIdx.x;
        for (int k = 0; k < 8; k++) {
            const U4 v = in[base + 256 * k];
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            out[base + 256 * k] = v;
        }
    }
}
// grid-stride form of the 1-load-1-store stream: G workgroups, each walking n / (256 G) chunks (persistent waves)
__global__ __launch_bounds__(256) void lab_stream_persistent(const U4* __restrict__ in, U4* __restrict__ out, long long n)
{
    for (long long c = (long long)blockIdx.x * 256 + threadIdx.x; c < n; c += (long long)gridDim.x * 256) out[c] = in[c];
}


// ---- load-side lab: what does the decoder's READ pattern cost, and what would LDS-DMA change? ----------------------
// A workgroup reads one "tile" of 192 coefficient blocks (24 KB contiguous here) and stores nothing.
//   MODE 0  one lane per block, 8 x 16-byte global loads at a 128-byte lane stride (the decoder today: every wave
//           instruction touches 64 different 128-byte lines, every line is visited by 8 instructions)
//   MODE 1  waves 0-2: 8 x global_load_lds_dwordx4 each -- one wave instruction moves 1 KB of CONTIGUOUS memory (8 whole
//           blocks) straight into LDS; the wave waits for its own data, then every lane reads its block's 8 chunks
//           with ds_read_b128.  The 16-byte chunks of block b sit at slot c ^ ((b >> 1) & 7) of the block's 128 bytes
//           (the permutation is applied to the SOURCE address, the LDS image of a DMA instruction is lane-linear), which
//           makes the lane-per-block ds_read_b128 conflict-free in the hardware's lane groups.
//   MODE 2  the same, all four waves issue 6 DMA instructions each, workgroup barrier before the reads
// PAD: extra LDS so that 6 workgroups fit a CU, as in the fused kernel (26.4 KB)
template <int MODE>
__global__ __launch_bounds__(256, 6) void lab_rd(const U4* __restrict__ in, U4* __restrict__ out, long long ntiles)
{
    __shared__ __attribute__((aligned(16))) U4 raw[192 * 8 + 152];
    long long id = blockIdx.x;
    if ((ntiles & 7) == 0) id = (id & 7) * (ntiles >> 3) + (id >> 3);
    const U4* src = in + id * (192 * 8);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    U4 v[8];
    if (MODE == 0) {
        if (tid < 192) {
#pragma unroll
            for (int k = 0; k < 8; k++) v[k] = src[tid * 8 + k];
        }
        if (tid == 255) raw[0] = v[0]; // keeps the allocation (and the occupancy) of the other modes
    } else {
        const int pos = lane & 7, sub = lane >> 3;
        if (MODE == 1) {
            if (w < 3) {
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const int blk = 64 * w + 8 * i + sub;
                    const U4* g = src + blk * 8 + (pos ^ ((blk >> 1) & 7));
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                                     (__attribute__((address_space(3))) void*)(raw + (64 * w + 8 * i) * 8), 16, 0, 0);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        } else {
#pragma unroll
            for (int i = 0; i < 6; i++) {
                const int blk = 48 * w + 8 * i + sub;
                const U4* g = src + blk * 8 + (pos ^ ((blk >> 1) & 7));
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                                 (__attribute__((address_space(3))) void*)(raw + (48 * w + 8 * i) * 8), 16, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        if (tid < 192) {
#pragma unroll
            for (int k = 0; k < 8; k++) v[k] = raw[tid * 8 + (k ^ ((tid >> 1) & 7))];
        }
    }
    if (tid < 192) {
        // MODE 0 and the DMA modes must see the same values: chunk k of block tid
        uint32_t acc = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) acc += (v[k].x ^ (v[k].y + k)) + (v[k].z ^ v[k].w) * (k + 1);
        if (acc == 0x9e3779b9u) out[id * 192 + tid].x = acc; // (practically) never: the kernel only reads
    }
}
// the same with the sums written out, so that tools/lab.py can check MODE 1 / 2 against MODE 0 (one u32 per block)
template <int MODE>
__global__ __launch_bounds__(256, 6) void lab_rd_check(const U4* __restrict__ in, uint32_t* __restrict__ sums, long long ntiles)
{
    __shared__ __attribute__((aligned(16))) U4 raw[192 * 8 + 152];
    const long long id = blockIdx.x;
    const U4* src = in + id * (192 * 8);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    U4 v[8];
    if (MODE == 0) {
        if (tid < 192) {
#pragma unroll
            for (int k = 0; k < 8; k++) v[k] = src[tid * 8 + k];
        }
        if (tid == 255) raw[0] = v[0];
    } else {
        const int pos = lane & 7, sub = lane >> 3;
        const int per = MODE == 1 ? 64 : 48, n = MODE == 1 ? 8 : 6;
        if (MODE == 2 || w < 3) {
            for (int i = 0; i < n; i++) {
                const int blk = per * w + 8 * i + sub;
                const U4* g = src + blk * 8 + (pos ^ ((blk >> 1) & 7));
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                                 (__attribute__((address_space(3))) void*)(raw + (per * w + 8 * i) * 8), 16, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid < 192) {
#pragma unroll
            for (int k = 0; k < 8; k++) v[k] = raw[tid * 8 + (k ^ ((tid >> 1) & 7))];
        }
    }
    if (tid < 192) {
        uint32_t acc = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) acc += (v[k].x ^ (v[k].y + k)) + (v[k].z ^ v[k].w) * (k + 1);
        sums[id * 192 + tid] = acc;
    }
}

typedef void (*labmem_fn)(const U4*, U4*, long long);
static const struct { const char* name; labmem_fn fn; } LABMEM[] = {
    {"copy  rd=coalesced      wr=coalesced", lab_mem<0, 0>},
    {"copy  rd=block/lane     wr=coalesced", lab_mem<1, 0>},
    {"copy  rd=coalesced      wr=48B/lane", lab_mem<0, 1>},
    {"copy  rd=block/lane     wr=48B/lane (decoder)", lab_mem<1, 1>},
    {"copy  rd=block/lane     wr=48B via LDS transpose", lab_mem<1, 2>},
    {"stream copy 1 x 16 B per lane", nullptr},
    {"stream copy 4 x 16 B per lane", nullptr},
    {"stream copy 8 x 16 B per lane", nullptr},
    {"burst copy 8 x 16 B, lane-contiguous 128 B per lane", nullptr},
    {"burst copy 8 x 16 B, software-pipelined (load k+1 | store k)", nullptr},
    {"burst copy 8 rounds of load;wait;store (long-lived waves)", nullptr},
    {"stream copy, persistent: 2048 workgroups grid-stride", nullptr},
    {"stream copy, persistent: 8192 workgroups grid-stride", nullptr},
};
typedef void (*labtile_fn)(const U4*, U4*, long long);
static const struct { const char* name; labtile_fn fn; int seg; } LABTILE[] = {
    {"tile copy  row segment  768 B  xcd-contiguous", lab_tile<768, 1>, 768},
    {"tile copy  row segment 1536 B  xcd-contiguous", lab_tile<1536, 1>, 1536},
    {"tile copy  row segment 3072 B  xcd-contiguous", lab_tile<3072, 1>, 3072},
    {"tile copy  row segment 6144 B  xcd-contiguous", lab_tile<6144, 1>, 6144},
    {"tile copy  row segment 12288 B xcd-contiguous", lab_tile<12288, 1>, 12288},
    {"tile copy  row segment  768 B  round-robin", lab_tile<768, 0>, 768},
    {"tile copy  row segment 3072 B  round-robin", lab_tile<3072, 0>, 3072},
    {"tile copy  row segment 12288 B round-robin", lab_tile<12288, 0>, 12288},
    {"32 KB per workgroup as 64 rows x  512 B", lab_tile32k<512>, 1024},
    {"32 KB per workgroup as 32 rows x 1024 B", lab_tile32k<1024>, 1024},
    {"32 KB per workgroup as 16 rows x 2048 B", lab_tile32k<2048>, 1024},
    {"32 KB per workgroup as  8 rows x 4096 B", lab_tile32k<4096>, 1024},
    {"read only: lane per block, 8 x 16 B at 128-B stride (decoder)", lab_rd<0>, 768},
    {"read only: LDS-DMA 1 KB/instruction, waves 0-2 x 8, ds_read_b128", lab_rd<1>, 768},
    {"read only: LDS-DMA 1 KB/instruction, 4 waves x 6, barrier", lab_rd<2>, 768},
};
int labmem_count() { return (int)(sizeof(LABMEM) / sizeof(LABMEM[0])) + (int)(sizeof(LABTILE) / sizeof(LABTILE[0])); }
const char* labmem_name(int i) { const int n = (int)(sizeof(LABMEM) / sizeof(LABMEM[0])); return i < n ? LABMEM[i].name : LABTILE[i - n].name; }
hipError_t launch_labmem(int i, const void* in, void* out, long long bytes, hipStream_t s)
{
    if (i < 0 || i >= labmem_count()) return hipErrorInvalidValue;
    const int n = (int)(sizeof(LABMEM) / sizeof(LABMEM[0]));
    if (i >= n) {
        const long long ntiles = bytes / (32ll * LABTILE[i - n].seg);
        hipLaunchKernelGGL(LABTILE[i - n].fn, dim3((unsigned)ntiles), dim3(256), 0, s, (const U4*)in, (U4*)out, ntiles);
        return hipGetLastError();
    }
    const long long T = bytes / 384;
    if (LABMEM[i].fn == nullptr) {
        const long long nch = bytes / 16;
        if (i >= 8) {
            const unsigned g8 = (unsigned)(nch / (256 * 8));
            if (i == 8) hipLaunchKernelGGL(lab_stream8<0>, dim3(g8), dim3(256), 0, s, (const U4*)in, (U4*)out, nch);
            else if (i == 9) hipLaunchKernelGGL(lab_stream8<1>, dim3(g8), dim3(256), 0, s, (const U4*)in, (U4*)out, nch);
            else if (i == 10) hipLaunchKernelGGL(lab_stream8<2>, dim3(g8), dim3(256), 0, s, (const U4*)in, (U4*)out, nch);
            else hipLaunchKernelGGL(lab_stream_persistent, dim3(i == 11 ? 2048 : 8192), dim3(256), 0, s, (const U4*)in, (U4*)out, nch);
            return hipGetLastError();
        }
        const int per = i == 5 ? 1 : (i == 6 ? 4 : 8);
        const unsigned g = (unsigned)(nch / (256 * per));
        if (per == 1) hipLaunchKernelGGL(lab_stream<1>, dim3(g), dim3(256), 0, s, (const U4*)in, (U4*)out, nch);
        else if (per == 4) hipLaunchKernelGGL(lab_stream<4>, dim3(g), dim3(256), 0, s, (const U4*)in, (U4*)out, nch);
        else hipLaunchKernelGGL(lab_stream<8>, dim3(g), dim3(256), 0, s, (const U4*)in, (U4*)out, nch);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(LABMEM[i].fn, dim3((unsigned)(T / 256)), dim3(256), 0, s, (const U4*)in, (U4*)out, T);
    return hipGetLastError();
}

// MODE 0 / 1 / 2 of lab_rd_check over `ntiles` tiles of 24 KB: one u32 per block into sums (device memory)
hipError_t launch_lab_rd_check(int mode, const void* in, uint32_t* sums, long long ntiles, hipStream_t s)
{
    if (mode == 0) hipLaunchKernelGGL(lab_rd_check<0>, dim3((unsigned)ntiles), dim3(256), 0, s, (const U4*)in, sums, ntiles);
    else if (mode == 1) hipLaunchKernelGGL(lab_rd_check<1>, dim3((unsigned)ntiles), dim3(256), 0, s, (const U4*)in, sums, ntiles);
    else hipLaunchKernelGGL(lab_rd_check<2>, dim3((unsigned)ntiles), dim3(256), 0, s, (const U4*)in, sums, ntiles);
    return hipGetLastError();
}

int lab_count() { return (int)(sizeof(LAB) / sizeof(LAB[0])) + 1; }
const char* lab_name(int i) { return i < lab_count() - 1 ? LAB[i].name : "colour 16px/lane (ycc->rgb + pack, EO)"; }
hipError_t launch_lab(int i, const int32_t qt[3][64], int* out, int blocks, int iters, hipStream_t s)
{
    if (i < 0 || i >= lab_count()) return hipErrorInvalidValue;
    LabTab tabs;
    for (int c = 0; c < 3; c++) build_table(qt[c], tabs.t + TAB_DW * c);
    if (i == lab_count() - 1) hipLaunchKernelGGL(lab_color<0>, dim3(blocks), dim3(256), 0, s, out, iters);
    else hipLaunchKernelGGL(LAB[i].fn, dim3(blocks), dim3(256), 0, s, tabs, out, iters);
    return hipGetLastError();
}

} // namespace zj
