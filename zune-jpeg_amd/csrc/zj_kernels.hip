// zj_kernels.hip -- gfx950 kernels of the pixel path and their launchers.
//
//   zj_fused_kernel<HS,VS,OUT>  whole hot path per tile: dequantize + IDCT -> LDS planar staging ->
//                               up-sample + colour-convert -> global store.  One HBM read of the
//                               coefficients, one HBM write of the pixels (6 B/px for 4:2:0->RGB).
//   zj_idct_strip_kernel        IDCTPtr-compatible strip IDCT  (src/idct/scalar.rs:19)
//   zj_upsample_{h,v}_kernel    UpSampler-compatible flat-array filters (src/upsampler/scalar.rs)
//   zj_rgb16_kernel             ColorConvert16Ptr (src/color_convert/scalar.rs:52)
#include <hip/hip_runtime.h>
#include <stdio.h>

#include "zj_device.h"
#include "zj_launch.h"

#ifndef ZJ_WAVES_PER_SIMD
#define ZJ_WAVES_PER_SIMD 5
#endif

namespace zj {

// ------------------------------------------------------------------------------------------------
// fused tile kernel
// ------------------------------------------------------------------------------------------------
// ZJ_PRIO bit 0: raise the wave priority while a tile's coefficient loads are being issued (+1 % measured,
// tools/ab_libs.sh); bit 1: raise it for the colour phase (no gain).  Default: bit 0.
#ifndef ZJ_STEAL_ROTATE
#define ZJ_STEAL_ROTATE 1
#endif
#ifndef ZJ_PRIO
#define ZJ_PRIO 1
#endif
#define ZJ_SETPRIO(bit, level) do { if (ZJ_PRIO & (bit)) __builtin_amdgcn_s_setprio(level); } while (0)

template <int HS, int VS, int OUT, int COMPACT, bool FAST>
// 2nd launch bound = waves per SIMD: 5 workgroups of 4 waves per CU need <= 96 VGPRs; LDS (32.7 KB
// per workgroup for 4:2:0) allows exactly 5.  Measured with tools/occupancy.py: 5 and 4 workgroups per CU
// run the same, 3 cost 7 %, 2 cost 29 % -- the bound keeps the kernel on the flat part.
__global__ __launch_bounds__((Cfg<HS, VS, OUT>::NT), ZJ_WAVES_PER_SIMD) void zj_fused_kernel(const Params p)
{
    using C = Cfg<HS, VS, OUT>;
    // COMPACT: bits 0-1 = how phase 1 spreads the IDCT (0 one lane per block, 1 compaction, 3 work stealing),
    //          bit 2    = transposed stores in phase 2
    constexpr int IDCT_MODE = COMPACT & 3;
    constexpr bool TS = (COMPACT & 4) != 0;
    __shared__ __attribute__((aligned(16))) char lds_raw[TS ? C::LDS_BYTES_TS : (IDCT_MODE ? C::LDS_BYTES_COMPACT : C::LDS_BYTES)];
    int16_t* lds = reinterpret_cast<int16_t*>(lds_raw);
    ZJ_SETPRIO(1, 3); // issue the tile's loads ahead of other waves' arithmetic
    const TileId t = decode_tile(p, (int)blockIdx.x);
    const int tid = (int)threadIdx.x;
    const BlockLoc L = locate<C>(p, t, tid, lds);
    U4 raw[8];
    load_block(L, raw, p.debug); // HBM loads in flight across the barrier below
    ZJ_SETPRIO(1, 0);
    if (IDCT_MODE == 3) {
        const int32_t q0 = p.qt[64 * L.comp]; // the DC-only shortcut needs q[0] before the tables are staged
        phase_setup<C, HS, VS>(p, tid, lds);
        // the donor wave changes from tile to tile: a wave stays on its SIMD, so a fixed donor would relieve one SIMD only
        const int donor = ZJ_STEAL_ROTATE ? (t.tile + t.strip + t.frame) % (C::NT / 64) : C::NT / 64 - 1;
        const StealState st = steal_stage<C>(L, raw, q0, tid, lds, p.clamp_dc, donor);
        __syncthreads();
        steal_idct<C>(L, raw, st, tid, lds, donor);
    } else if (IDCT_MODE) {
        const int32_t q0 = p.qt[64 * L.comp]; // the DC-only shortcut needs q[0] before the tables are staged
        phase_setup<C, HS, VS>(p, tid, lds);
        classify_stage<C>(L, raw, q0, tid, lds, p.clamp_dc);
        __syncthreads();
        idct_queue<C>(tid, lds);
    } else {
        phase_setup<C, HS, VS>(p, tid, lds);
        __syncthreads();
        finish_block<C>(L, raw, lds, p.debug, p.clamp_dc);
    }
    __syncthreads();
    ZJ_SETPRIO(2, 2); // (off) let a tile's last phase, the one that frees the workgroup slot, go first
    if (TS) {
        // each wave stages its 64 items of a round in LDS (in place), then stores them as contiguous pieces;
        // LDS operations of one wave execute in order, so no barrier is needed between the two halves
        for (int round = 0; round * C::NT < C::NITEMS; round++) {
            phase_color<C, HS, VS, OUT, FAST, true>(p, t, tid, lds, round);
            color_copyout<C, OUT>(p, t, tid, lds, round);
        }
    } else {
        phase_color<C, HS, VS, OUT, FAST>(p, t, tid, lds);
    }
}

// Persistent form of the same pipeline: a fixed grid of workgroups, each walking many tiles.  The next
// tile's coefficient loads are issued as soon as the IDCT has consumed the current ones, so they are in
// flight during the whole colour phase (HBM latency hidden, smoother load/store mix); tables are staged
// once per workgroup instead of once per tile.
template <int HS, int VS, int OUT>
__global__ __launch_bounds__((Cfg<HS, VS, OUT>::NT), 4) void zj_fused_persistent_kernel(const Params p)
{
    using C = Cfg<HS, VS, OUT>;
    __shared__ __attribute__((aligned(16))) char lds_raw[C::LDS_BYTES];
    int16_t* lds = reinterpret_cast<int16_t*>(lds_raw);
    const int tid = (int)threadIdx.x;
    const TileWalk w = persistent_walk(p, (int)blockIdx.x, (int)gridDim.x);
    int id = w.first;
    if (id >= w.last) return;
    TileId t = tile_from_id(p, id);
    BlockLoc L = locate<C>(p, t, tid, lds);
    U4 raw[8];
    load_block(L, raw);
    phase_setup<C, HS, VS>(p, tid, lds);
    __syncthreads();
    for (;;) {
        finish_block<C>(L, raw, lds, p.debug, p.clamp_dc);
        const int nid = id + w.step;
        const bool more = nid < w.last;
        TileId tn = t;
        if (more) { // prefetch: in flight across the colour phase below
            tn = tile_from_id(p, nid);
            L = locate<C>(p, tn, tid, lds);
            load_block(L, raw);
        }
        __syncthreads();
        phase_color<C, HS, VS, OUT, true>(p, t, tid, lds);
        if (!more) break;
        __syncthreads(); // the staging area is rewritten by the next IDCT
        t = tn;
        id = nid;
    }
}

static int g_persistent_wgs = 0;
void set_persistent_grid(int wgs) { g_persistent_wgs = wgs; }
static int persistent_grid()
{
    if (g_persistent_wgs > 0) return g_persistent_wgs;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    return cus * 4;
}

// occupancy probe of tools/occupancy.py: extra dynamic LDS per workgroup (never set by the product)
static int g_pad_lds = 0;
void set_pad_lds(int bytes) { g_pad_lds = bytes < 0 ? 0 : bytes; }
int fused_occupancy_420_rgb(int pad_lds)
{
    int n = -1;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, zj_fused_kernel<2, 2, OUT_RGB, 0, true>, Cfg<2, 2, OUT_RGB>::NT, (size_t)pad_lds) != hipSuccess) return -1;
    return n;
}

template <int HS, int VS, int OUT>
static hipError_t launch_fused_t(const Params& p, int compact, int fast, hipStream_t s)
{
    using C = Cfg<HS, VS, OUT>;
    if (p.total_tiles <= 0) return hipSuccess;
    const dim3 grid((unsigned)p.total_tiles), block(C::NT);
    if (OUT == OUT_RGBA || OUT == OUT_RGB_CHW) { // extensions: the one-pass kernel only
        if (fast) hipLaunchKernelGGL((zj_fused_kernel<HS, VS, OUT, 0, true>), grid, block, 0, s, p);
        else hipLaunchKernelGGL((zj_fused_kernel<HS, VS, OUT, 0, false>), grid, block, 0, s, p);
        return hipGetLastError();
    }
    if (g_pad_lds > 0 && fast && compact == 0) {
        hipLaunchKernelGGL((zj_fused_kernel<HS, VS, OUT, 0, true>), grid, block, (size_t)g_pad_lds, s, p);
        return hipGetLastError();
    }
    if (!fast) hipLaunchKernelGGL((zj_fused_kernel<HS, VS, OUT, 0, false>), grid, block, 0, s, p); // any width
    else if (compact == 2) {
        int wgs = persistent_grid();
        if (wgs > p.total_tiles) wgs = p.total_tiles;
        hipLaunchKernelGGL((zj_fused_persistent_kernel<HS, VS, OUT>), dim3((unsigned)wgs), block, 0, s, p);
    } else if ((compact == 4 || compact == 7) && ts_eligible<C>(p, OUT, true)) {
        // transposed stores exist for the 3-byte interleaved outputs only; TSC folds to 0 elsewhere (never reached)
        constexpr int TSC = (OUT == OUT_RGB || OUT == OUT_YCBCR) ? 4 : 0;
        if (compact == 4) hipLaunchKernelGGL((zj_fused_kernel<HS, VS, OUT, TSC, true>), grid, block, 0, s, p);
        else hipLaunchKernelGGL((zj_fused_kernel<HS, VS, OUT, (TSC | 3), true>), grid, block, 0, s, p);
    } else if (compact == 3 || compact == 7) hipLaunchKernelGGL((zj_fused_kernel<HS, VS, OUT, 3, true>), grid, block, 0, s, p);
    else if (compact == 4) hipLaunchKernelGGL((zj_fused_kernel<HS, VS, OUT, 0, true>), grid, block, 0, s, p);
    else if (compact) hipLaunchKernelGGL((zj_fused_kernel<HS, VS, OUT, 1, true>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((zj_fused_kernel<HS, VS, OUT, 0, true>), grid, block, 0, s, p);
    return hipGetLastError();
}

hipError_t launch_fused(int hs, int vs, int out, int compact, int fast, const Params& p, hipStream_t s)
{
#define ZJ_CASE(H, V, O) if (hs == H && vs == V && out == O) return launch_fused_t<H, V, O>(p, compact, fast, s);
    ZJ_CASE(1, 1, OUT_RGB) ZJ_CASE(1, 1, OUT_GRAY) ZJ_CASE(1, 1, OUT_YCBCR)
    ZJ_CASE(2, 1, OUT_RGB) ZJ_CASE(2, 1, OUT_GRAY) ZJ_CASE(2, 1, OUT_YCBCR)
    ZJ_CASE(1, 2, OUT_RGB) ZJ_CASE(1, 2, OUT_GRAY) ZJ_CASE(1, 2, OUT_YCBCR)
    ZJ_CASE(2, 2, OUT_RGB) ZJ_CASE(2, 2, OUT_GRAY) ZJ_CASE(2, 2, OUT_YCBCR)
    ZJ_CASE(1, 1, OUT_RGBA) ZJ_CASE(2, 1, OUT_RGBA) ZJ_CASE(1, 2, OUT_RGBA) ZJ_CASE(2, 2, OUT_RGBA)
    ZJ_CASE(1, 1, OUT_RGB_CHW) ZJ_CASE(2, 1, OUT_RGB_CHW) ZJ_CASE(1, 2, OUT_RGB_CHW) ZJ_CASE(2, 2, OUT_RGB_CHW)
#undef ZJ_CASE
    return hipErrorInvalidValue;
}

const char* fused_kernel_name(int hs, int vs, int out, int variant, int fast)
{
    // the demangled name rocprofv3 prints for the instantiation launch_fused() picks
    static char buf[8][96];
    static int slot = 0;
    char* b = buf[slot++ & 7];
    if (out == OUT_RGBA || out == OUT_RGB_CHW) variant = 0;
    if (fast && variant == 2) snprintf(b, 96, "void zj::zj_fused_persistent_kernel<%d, %d, %d>(zj::Params)", hs, vs, out);
    else snprintf(b, 96, "void zj::zj_fused_kernel<%d, %d, %d, %d, %s>(zj::Params)", hs, vs, out, (fast && variant != 2) ? variant : 0, fast ? "true" : "false");
    return b;
}

// ------------------------------------------------------------------------------------------------
// strip-level kernels (fn-pointer compatible API; not the hot path)
// ------------------------------------------------------------------------------------------------
// dequantize_and_idct_int (src/idct/scalar.rs:19-282): one lane per block.  `out` pre-zeroed.
__global__ __launch_bounds__(64) void zj_idct_strip_kernel(const int16_t* __restrict__ coeff,
                                                           const int32_t* __restrict__ qt,
                                                           int16_t* __restrict__ out, long long nblocks,
                                                           long long chunks, long long bpc, long long stride)
{
    const long long j = (long long)blockIdx.x * 64 + threadIdx.x;
    if (j >= nblocks) return;
    const long long c = j / bpc, k = j % bpc; // chunk, block within chunk (scalar.rs:32-40)
    const U4* src = reinterpret_cast<const U4*>(coeff + c * chunks + k * 64);
    U4 raw[8], px[8];
#pragma unroll
    for (int i = 0; i < 8; i++) raw[i] = src[i];
    const uint32_t* cw = reinterpret_cast<const uint32_t*>(raw);
    uint32_t any = cw[0] & 0xffff0000u;
#pragma unroll
    for (int i = 1; i < 32; i++) any |= cw[i];
    if (any == 0) { // DC-only shortcut (scalar.rs:45-74)
        const uint32_t v = dc_only_value(cw[0], qt[0]);
#pragma unroll
        for (int i = 0; i < 8; i++) { px[i].x = v; px[i].y = v; px[i].z = v; px[i].w = v; }
    } else {
        idct_block(raw, qt, px);
    }
    int16_t* dst = out + c * chunks + k * 8; // pos = x = 8k (scalar.rs:277-278)
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const uint32_t* w = reinterpret_cast<const uint32_t*>(&px[r]);
        int16_t* d = dst + r * stride; // arbitrary stride: 2-byte stores
#pragma unroll
        for (int i = 0; i < 4; i++) { d[2 * i] = (int16_t)(w[i] & 0xffff); d[2 * i + 1] = (int16_t)(w[i] >> 16); }
    }
}

// upsample_horizontal (src/upsampler/scalar.rs:5-60), flat array; `out` pre-zeroed.
__global__ void zj_upsample_h_kernel(const int16_t* __restrict__ in, long long n, int16_t* __restrict__ out,
                                     long long out_len, long long m /* windows */)
{
    const long long w = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (w < m) {
        const int s = (int16_t)(uint16_t)(3 * in[w + 1] + 2);
        const long long o = 2 + 2 * w;
        // the last two outputs are owned by the epilogue below (scalar.rs:46-57 runs after the loop)
        if (o < out_len - 2) out[o] = (int16_t)((int16_t)(uint16_t)(s + in[w]) >> 2);
        if (o + 1 < out_len - 2) out[o + 1] = (int16_t)((int16_t)(uint16_t)(s + in[w + 2]) >> 2);
    }
    if (w == 0) {
        out[0] = in[0];
        out[1] = (int16_t)tri1(in[0], in[1]);
        out[out_len - 2] = (int16_t)tri1(in[n - 2], in[n - 1]);
        out[out_len - 1] = in[n - 1];
    }
}

// upsample_vertical (src/upsampler/scalar.rs:64-147): 8 input rows of `stride`; `out` pre-zeroed.
__global__ void zj_upsample_v_kernel(const int16_t* __restrict__ in, long long stride,
                                     int16_t* __restrict__ out, long long out_len)
{
    const long long x = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int k = blockIdx.y; // pair 0..7
    int n, f;
    vsched(k, n, f);
    const long long i = 2ll * k * stride;
    const long long rem = out_len - i - stride;
    const long long cnt = stride < rem ? stride : rem; // zip stops at the shortest (scalar.rs:110-127)
    if (x >= cnt) return;
    const int a = in[n * stride + x], b = in[f * stride + x];
    out[i + x] = (int16_t)tri1(a, b);
    out[i + stride + x] = (int16_t)tri1(b, a);
}

// ycbcr_to_rgb_16_scalar (src/color_convert/scalar.rs:52-89): 16 pixels -> 48 bytes
__global__ void zj_rgb16_kernel(const int16_t* __restrict__ ycc /* y[16] cb[16] cr[16] */, uint8_t* __restrict__ out)
{
    const int i = threadIdx.x;
    if (i >= 16) return;
    const uint32_t y = (uint16_t)ycc[i], cb = (uint16_t)ycc[16 + i], cr = (uint16_t)ycc[32 + i];
    const RGB2 c = ycc_to_rgb_pair(y, cb, cr);
    out[3 * i] = (uint8_t)sat_pk_u8(c.r); out[3 * i + 1] = (uint8_t)sat_pk_u8(c.g); out[3 * i + 2] = (uint8_t)sat_pk_u8(c.b);
}

hipError_t launch_idct_strip(const int16_t* coeff, const int32_t* qt, int16_t* out, long long nblocks,
                             long long chunks, long long bpc, long long stride, hipStream_t s)
{
    if (nblocks <= 0) return hipSuccess;
    hipLaunchKernelGGL(zj_idct_strip_kernel, dim3((unsigned)((nblocks + 63) / 64)), dim3(64), 0, s, coeff, qt,
                       out, nblocks, chunks, bpc, stride);
    return hipGetLastError();
}
hipError_t launch_upsample_h(const int16_t* in, long long n, int16_t* out, long long out_len, long long m, hipStream_t s)
{
    const long long work = m > 1 ? m : 1;
    hipLaunchKernelGGL(zj_upsample_h_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, s, in, n, out, out_len, m);
    return hipGetLastError();
}
hipError_t launch_upsample_v(const int16_t* in, long long stride, int16_t* out, long long out_len, hipStream_t s)
{
    hipLaunchKernelGGL(zj_upsample_v_kernel, dim3((unsigned)((stride + 255) / 256), 8), dim3(256), 0, s, in, stride, out, out_len);
    return hipGetLastError();
}
hipError_t launch_rgb16(const int16_t* ycc, uint8_t* out, hipStream_t s)
{
    hipLaunchKernelGGL(zj_rgb16_kernel, dim3(1), dim3(64), 0, s, ycc, out);
    return hipGetLastError();
}

} // namespace zj
