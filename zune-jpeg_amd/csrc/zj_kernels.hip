// zj_kernels.hip -- gfx950 kernels of the pixel path and their launchers.
//
//   zj_fused_kernel<HS,VS,OUT,GEN,FAST,TS>
//                               whole hot path per tile: dequantize + IDCT -> LDS planar staging ->
//                               up-sample + colour-convert -> global store.  One HBM read of the
//                               coefficients, one HBM write of the pixels (6 B/px for 4:2:0->RGB).
//                               GEN: packed (round 2) | wide (round 1); TS: staged, lane-contiguous stores
//   zj_idct_strip_kernel        IDCTPtr-compatible strip IDCT  (src/idct/scalar.rs:19)
//   zj_upsample_{h,v}_kernel    UpSampler-compatible flat-array filters (src/upsampler/scalar.rs)
//   zj_rgb16_kernel             ColorConvert16Ptr (src/color_convert/scalar.rs:52)
#include <hip/hip_runtime.h>
#include <atomic>
#include <stdio.h>

#include "zj_device.h"
#include "zj_launch.h"

// launch bound, waves per SIMD.  Wide generation: 5 (the 24-bit transform holds 64 x i32 between its passes: 89 VGPRs;
// 32.4 KB of LDS per workgroup allow 5 as well).  Packed generation: 6 -- its hot path fits 80 VGPRs (the spills the
// bound causes, 44 bytes, are all inside the wide code it falls back to), a 256-pixel tile needs 25.8 KB of LDS (26.4 before round 3 moved the halo columns out of the chroma rows), and six
// workgroups per CU measure 1 % faster than five (tools/ab_libs.sh, profiles/r02_*).
#ifndef ZJ_WAVES_PER_SIMD
#define ZJ_WAVES_PER_SIMD 5
#endif
#ifndef ZJ_WAVES_PER_SIMD_PACKED
#define ZJ_WAVES_PER_SIMD_PACKED 6
#endif

namespace zj {

// ------------------------------------------------------------------------------------------------
// fused tile kernel
// ------------------------------------------------------------------------------------------------
// ZJ_PRIO bit 0: raise the wave priority while a tile's coefficient loads are being issued (+1 % measured,
// tools/ab_libs.sh); bit 1: raise it for the colour phase (no gain).  Default: bit 0.
#ifndef ZJ_PRIO
#define ZJ_PRIO 1
#endif
#define ZJ_SETPRIO(bit, level) do { if (ZJ_PRIO & (bit)) __builtin_amdgcn_s_setprio(level); } while (0)

// the round-1 pipeline for one tile: int16 staging, 24-bit IDCT, 48-byte-per-lane stores
template <class C, int HS, int VS, int OUT, bool FAST, bool RAG = false>
__device__ __forceinline__ void tile_wide(const Params& p, const TileId t, const int tid, char* lds)
{
    ZJ_SETPRIO(1, 3); // issue the tile's loads ahead of other waves' arithmetic
    const BlockLoc L = locate<C, GEN_WIDE>(p, t, tid, lds);
    U4 raw[8];
    load_block(L, raw, ZJ_PDBG(p)); // HBM loads in flight across the barrier below
    ZJ_SETPRIO(1, 0);
    phase_setup<C, HS, VS, GEN_WIDE>(p, tid, lds);
    __syncthreads();
    finish_block<C, GEN_WIDE, false>(L, raw, lds, ZJ_PDBG(p), p.clamp_dc);
    __syncthreads();
    ZJ_SETPRIO(2, 2); // (off) let a tile's last phase, the one that frees the workgroup slot, go first
    phase_color<C, HS, VS, OUT, GEN_WIDE, FAST, false, RAG>(p, t, tid, lds);
}

// The workgroups of a launch's FIRST wave (one per occupancy slot of the chip) all start in the same cycle and would move
// through load / transform / colour / store in lock step: memory idles while they compute and the SIMDs idle while they
// load.  In a long launch that synchrony dissolves after a tile or two; a launch of ONE frame (2048 tiles on 1536 slots)
// never gets that far (profiles/r03_ab_history.txt: 25 us against 18 us per frame in a batch).  Here the k-th workgroup of
// every slot group waits k * delay before its first load, so that the workgroups sharing a CU start a phase apart.  The
// host turns it on for launches of one to two waves of workgroups only (zj_api.cpp: launch_params) and passes the slot
// count of the launched instantiation (stagger_wgs = CUs x workgroups per CU) and the divisor as a multiplier.
__device__ __forceinline__ void stagger_start(const Params& p, const int bid)
{
    if (p.stagger_delay <= 0 || bid >= p.stagger_wgs) return; // uniform
    const Magic g = {p.stagger_magic, p.stagger_shift};
    const int k = (int)magic_div((uint32_t)bid, g);           // 0 .. slots per CU - 1: the dispatcher fills the CUs round-robin
    const int n = k * p.stagger_delay;
    for (int i = 0; i < n; i++) __builtin_amdgcn_s_sleep(2);  // ~128 cycles = 53 ns per step
}

// What precedes a workgroup's first load is on its critical path: the workgroup holds one of the CU's six slots while it
// waits, so throughput = slots / workgroup lifetime (round 4: the division by a kernel argument).  Left to itself the
// compiler loads each kernel argument where it is first used -- total_tiles, then the multipliers, then a shift, then the
// base pointers, then the strides: five to six scalar-cache round trips in a row before the first coefficient load can
// issue.  The empty asm below makes the first 128 bytes of the arguments (everything the tile decode, the addresses and
// the stagger need) operands at the kernel's entry, so their loads are issued together -- merged into a few wide s_load --
// and waited for once; later uses find the values in registers (invariant loads, the same SSA values).
#ifndef ZJ_ASSUME_NO_HALO
#define ZJ_ASSUME_NO_HALO 1 // 0: A/B knob for the __builtin_assume in front of the block waves' `locate`
#endif
#ifndef ZJ_PIN_ARGS
#define ZJ_PIN_ARGS 1 // 0: A/B knob (tools/build_variant.sh nopin "-DZJ_PIN_ARGS=0")
#endif
__device__ __forceinline__ void pin_head(const Params& p)
{
    if (!ZJ_PIN_ARGS) return;
    asm volatile("; kernel arguments 0x00-0x87 are resident"
                 :: "s"(p.y), "s"(p.cb), "s"(p.cr), "s"(p.out), "s"(p.y_frame_stride), "s"(p.c_frame_stride), "s"(p.out_frame_stride),
                    "s"(p.width), "s"(p.height), "s"(p.mcu_x), "s"(p.n_strips), "s"(p.tiles_per_row), "s"(p.regular_px), "s"(p.zero_fill),
                    "s"(p.total_tiles), "s"(p.tpr_magic), "s"(p.tpr_shift), "s"(p.ns_magic), "s"(p.ns_shift), "s"(p.stagger_wgs),
                    "s"(p.stagger_delay), "s"(p.stagger_magic), "s"(p.stagger_shift), "s"(p.plain), "s"(p.out_pitch));
}

// the body of both kernel families: zj_fused_kernel (RAG = false) and zj_fused_ragged_kernel (GEN_PACKED, FAST, RAG)
template <int HS, int VS, int OUT, int GEN, bool FAST, bool TS, bool RAG, bool SEAM = false>
__device__ __forceinline__ void fused_body(const Params& p, char* lds)
{
    using C = Cfg<HS, VS, OUT>;
    pin_head(p);
    const TileId t = decode_tile(p, (int)blockIdx.x);
    // Which hardware wave plays which role (4:2:0: two luma waves, the chroma wave, the halo wave -- ~1390 VALU instructions per
    // tile for the first three, ~900 for the last).  `tid` is the LOGICAL thread number everywhere below (block, item,
    // staging slot), so any rotation of whole waves is equivalent; barriers are workgroup-wide.  Measured in round 4
    // (profiles/r04_ab_history.txt): rotating the roles with the workgroup, so that every SIMD sees every role, is 3.7 %
    // SLOWER than fixed roles (ZJ_ROTATE=1); a constant shift (ZJ_ROLE_SHIFT: which role the first wave plays) is the
    // other knob.  Default: roles as the threads are numbered.
#ifndef ZJ_ROTATE
#define ZJ_ROTATE 0
#endif
#ifndef ZJ_ROLE_SHIFT
#define ZJ_ROLE_SHIFT 0
#endif
    int tid = (int)threadIdx.x;
    if ((ZJ_ROTATE || ZJ_ROLE_SHIFT) && C::NW > 1) {
        const unsigned b = blockIdx.x;
        const int rot = ZJ_ROTATE ? (int)((b ^ (b >> 3) ^ (b >> 8)) % (unsigned)C::NW) : ZJ_ROLE_SHIFT % C::NW;
        tid += 64 * rot;
        if ((C::NT & (C::NT - 1)) == 0) tid &= C::NT - 1; else if (tid >= C::NT) tid -= C::NT;
        __builtin_assume(tid >= 0 && tid < C::NT); // what the compiler knew of threadIdx.x (it shapes the colour rounds)
    }
    stagger_start(p, (int)blockIdx.x);
    if (GEN == GEN_WIDE) { tile_wide<C, HS, VS, OUT, FAST, RAG>(p, t, tid, lds); return; }
    // luma enters arithmetic for the RGB family only; gray / YCbCr outputs keep its low byte (Q7)
    constexpr bool NEED_Y16 = OUT == OUT_RGB || OUT == OUT_RGBA || OUT == OUT_RGB_CHW;
    ZJ_SETPRIO(1, 3);
    // the wave of the halo blocks works with one lane per block column (zj_device.h: halo_*); a wave-uniform split
    const bool halo_wave = C::HALO_PURE && (__builtin_amdgcn_readfirstlane(tid) >> 6) == C::HALO_T0 / 64;
    // The halo wave and the block waves run DISJOINT code from their first load to the second barrier (both sides execute
    // the same two barriers).  Written as two consecutive `if (halo_wave)` the compiler keeps the halo wave's twelve
    // registers (its eight coefficients, its addresses) live THROUGH the block waves' transform -- it cannot know that the
    // two conditions exclude that path -- and the packed IDCT has 68 registers instead of 80 (round 4, seen as spills the
    // moment the transform got a second form).
    if (halo_wave) {
        HaloLane H = halo_locate<C>(p, t, tid - C::HALO_T0, lds);
        int32_t hs8[8];
        halo_load(H, hs8);
        ZJ_SETPRIO(1, 0);
        phase_setup<C, HS, VS, GEN_PACKED>(p, tid, lds);
        // cut points of the instruction ledger (diagnostic build only, tools/valu_ledger.sh): the kernel ends here, so that
        // the hardware's instruction counters of two builds-with-a-cut differ by exactly one phase
        if (ZJ_ABL(ZJ_PDBG(p), 32)) return;   // ... after tile decode, block addresses, load issue, table staging
        __syncthreads();
        halo_pass1<C>(H, hs8, lds);
        ZJ_WAVE_FENCE();
        halo_pass2<C>(H, lds, p.clamp_dc);
        ZJ_WAVE_FENCE();
        halo_filter<C, HS, VS>(p, t, tid - C::HALO_T0, lds);
    } else {
        // (with the halo blocks in a wave of their own no block wave ever holds one: telling the compiler prunes the halo
        // cases out of `locate`, which sits in front of the chroma wave's first load)
        if (ZJ_ASSUME_NO_HALO && C::HALO_PURE && C::NT == C::HALO_T0 + 64) __builtin_assume(tid < C::HALO_T0);
        const BlockLoc L = locate<C, GEN_PACKED>(p, t, tid, lds);
        U4 raw[8];
        load_block(L, raw, ZJ_PDBG(p));
        ZJ_SETPRIO(1, 0);
        phase_setup<C, HS, VS, GEN_PACKED>(p, tid, lds);
        if (ZJ_ABL(ZJ_PDBG(p), 32)) return;
        __syncthreads();
        finish_block<C, GEN_PACKED, NEED_Y16>(L, raw, lds, ZJ_PDBG(p), p.clamp_dc);
    }
    // The coefficient loads are consumed inside exec-masked regions, so on the paths that skip those regions the
    // compiler still counts them as outstanding and would put `s_waitcnt vmcnt(0)` in front of every later reuse of
    // their registers -- in the store rounds below that wait also drains the round's own stores (gfx9 has one counter
    // for loads and stores) and serialises them.  One explicit wait here (free: the data arrived before the IDCT) tells
    // the wait-count pass that nothing is pending.
    __builtin_amdgcn_s_waitcnt(0x0f70); // vmcnt(0) only
    if (ZJ_ABL(ZJ_PDBG(p), 64)) return;   // ... after classification, IDCT, LDS staging (block waves) / the halo wave's work
    __syncthreads();
    // a DC-only luma block of this tile decodes outside 0..255 (Q1: the scalar shortcut does not clamp): the byte
    // staging cannot carry it, the whole tile is redone by the wide code (never seen on valid 8-bit JPEG data)
    const bool redo = NEED_Y16 && __builtin_amdgcn_readfirstlane((int)*lds_flag<C>(lds)) != 0;
    if (redo) {
        __syncthreads(); // everyone has read the flag before the wide layout overwrites it
        tile_wide<C, HS, VS, OUT, FAST, RAG>(p, t, tid, lds);
    } else if (TS) {
        // each wave stages its 64 items of a round in LDS, then stores them as contiguous pieces; LDS operations
        // of one wave execute in order, so no barrier is needed between the halves or between rounds
        ZJ_SETPRIO(2, 2);
        __builtin_amdgcn_s_waitcnt(0x0f70); // vmcnt(0), free here; see above
#pragma unroll
        for (int round = 0; round * C::NT < C::NITEMS; round++) {
            ItemOut io;
            // (only 4:2:2 has a partly filled round: everywhere else the logical numbering is the hardware's, hw = -1 says so)
            const int hw = C::ROUND_ROT ? uniform(tid >> 6) : -1, ltid = round_tid<C>(tid, round);
            phase_color<C, HS, VS, OUT, GEN_PACKED, FAST, true, RAG>(p, t, ltid, lds, round, &io);
            if (round == 0 && ZJ_ABL(ZJ_PDBG(p), 128)) { ZJ_USE(io.s0.x ^ io.s0.y ^ io.s0.z ^ io.s0.w ^ io.s1.x ^ io.s1.y ^ io.s1.z ^ io.s1.w ^ io.s2.x ^ io.s2.y ^ io.s2.z ^ io.s2.w); return; } // ... after round 0's filters, colour math, packing
            stage_item<C>(io, ltid, lds, round, hw);
            ZJ_WAVE_FENCE();
            color_copyout<C, OUT, RAG, SEAM>(p, t, ltid, lds, round, hw);
            ZJ_WAVE_FENCE();
            if (round == 0 && ZJ_ABL(ZJ_PDBG(p), 256)) return; // ... after round 0's staging and stores
        }
    } else {
        ZJ_SETPRIO(2, 2);
        __builtin_amdgcn_s_waitcnt(0x0f70); // vmcnt(0), free here; see above
        phase_color<C, HS, VS, OUT, GEN_PACKED, FAST, false, RAG>(p, t, tid, lds);
    }
}

template <int HS, int VS, int OUT, int GEN, bool FAST, bool TS>
// launch bounds: see ZJ_WAVES_PER_SIMD / ZJ_WAVES_PER_SIMD_PACKED above
__global__ __launch_bounds__((Cfg<HS, VS, OUT>::NT), (GEN == GEN_PACKED ? ZJ_WAVES_PER_SIMD_PACKED : ZJ_WAVES_PER_SIMD)) void zj_fused_kernel(const Params p)
{
    using C = Cfg<HS, VS, OUT>;
    __shared__ __attribute__((aligned(16))) char lds[GEN == GEN_PACKED ? C::LDS_PACKED : C::LDS_WIDE];
    fused_body<HS, VS, OUT, GEN, FAST, TS, false>(p, lds);
}

// Ragged widths (width % 16 != 0, width >= 64; the reference's medium images are 2500 wide): the packed generation's fast
// path for every ordinary 16-pixel group of a row, the generic stores for the few groups at the row's end -- one launch,
// one kernel (zj_device.h: phase_color, RAG).  A family of its own so that the aligned kernels above keep their code.
// Aligned widths whose rows do not start on 128-byte boundaries (zj_device.h: seam_launch, color_copyout SEAM): the packed
// generation's fast path with staged stores, the lines shared between neighbouring tiles written back.  A family of its
// own for the same reason as the ragged one.
template <int HS, int VS, int OUT>
__global__ __launch_bounds__((Cfg<HS, VS, OUT>::NT), ZJ_WAVES_PER_SIMD_PACKED) void zj_fused_seam_kernel(const Params p)
{
    using C = Cfg<HS, VS, OUT>;
    __shared__ __attribute__((aligned(16))) char lds[C::LDS_PACKED];
    fused_body<HS, VS, OUT, GEN_PACKED, true, C::TSCAP, false, C::TSCAP>(p, lds);
}

#ifndef ZJ_WAVES_PER_SIMD_RAG
#define ZJ_WAVES_PER_SIMD_RAG ZJ_WAVES_PER_SIMD_PACKED
#endif
template <int HS, int VS, int OUT, bool TS>
__global__ __launch_bounds__((Cfg<HS, VS, OUT>::NT), ZJ_WAVES_PER_SIMD_RAG) void zj_fused_ragged_kernel(const Params p)
{
    using C = Cfg<HS, VS, OUT>;
    __shared__ __attribute__((aligned(16))) char lds[C::LDS_PACKED];
    fused_body<HS, VS, OUT, GEN_PACKED, true, TS, true>(p, lds);
}

// Zeros for the rows below a frame's last complete strip (zj_launch.h: ZeroRows).  blockIdx.y = frame * nr + range,
// blockIdx.x = a 16 KB piece of the range; bytes up to the first 16-byte boundary and after the last one go out singly
// (the rows of a ragged width start anywhere).
__global__ __launch_bounds__(256) void zj_zero_rows_kernel(const ZeroRows z)
{
    const int fr = (int)blockIdx.y / z.nr, r = (int)blockIdx.y - fr * z.nr;
    uint8_t* const base = (z.out ? z.out + (long long)fr * z.frame_stride : ZJ_GLOBAL_PTR(uint8_t, z.fptr[fr])) + z.off[r];
    const unsigned long long len = z.len[r], begin = (unsigned long long)blockIdx.x * 16384ull;
    if (begin >= len) return;
    const unsigned n = (unsigned)(len - begin < 16384ull ? len - begin : 16384ull);
    uint8_t* const p = base + begin;
    unsigned head = (16u - ((unsigned)reinterpret_cast<uintptr_t>(p) & 15u)) & 15u;
    if (head > n) head = n;
    const unsigned nq = (n - head) >> 4, tail = head + (nq << 4);
    const unsigned tid = threadIdx.x;
    if (tid < head) p[tid] = 0;
    const U4 zero = {0, 0, 0, 0};
    for (unsigned i = tid; i < nq; i += 256) *reinterpret_cast<U4*>(p + head + 16u * i) = zero;
    if (tail + tid < n) p[tail + tid] = 0;
}

hipError_t launch_zero_rows(const ZeroRows& z, hipStream_t s)
{
    if (z.nr <= 0 || z.nframes <= 0) return hipSuccess;
    unsigned long long longest = 0;
    for (int r = 0; r < z.nr; r++) longest = z.len[r] > longest ? z.len[r] : longest;
    if (longest == 0) return hipSuccess;
    const dim3 grid((unsigned)((longest + 16383ull) / 16384ull), (unsigned)(z.nframes * z.nr)), block(256);
    hipLaunchKernelGGL(zj_zero_rows_kernel, grid, block, 0, s, z);
    return hipGetLastError();
}

#if defined(ZJ_ABLATION)
// occupancy probe of tools/occupancy.py (diagnostic build only): extra dynamic LDS per workgroup
static int g_pad_lds = 0;
void set_pad_lds(int bytes) { g_pad_lds = bytes < 0 ? 0 : bytes; }
int fused_occupancy_420_rgb(int pad_lds)
{
    int n = -1;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, zj_fused_kernel<2, 2, OUT_RGB, GEN_PACKED, true, true>, Cfg<2, 2, OUT_RGB>::NT, (size_t)pad_lds) != hipSuccess) return -1;
    return n;
}
#else
static constexpr int g_pad_lds = 0;
#endif

// variant: 0 = packed generation (staged stores where they apply), 1 = wide generation (round 1), 2 = packed with
// direct stores.  All are bit-exact; 1 and 2 exist for A/B measurements and as the parity cross-check.
// `fast` is zj_plan.h's launch_mode: 0 generic stores, 1 aligned fast path, 2 ragged fast path (packed generation only)
static void pick(int variant, int out, bool fast, bool ts_ok, int& gen, bool& ts)
{
    gen = variant == 1 ? GEN_WIDE : GEN_PACKED;
    ts = gen == GEN_PACKED && variant == 0 && fast && (out == OUT_RGB || out == OUT_YCBCR || out == OUT_RGBA) && ts_ok;
}

template <int HS, int VS, int OUT>
static hipError_t launch_fused_t(const Params& p, int variant, int fast, hipStream_t s)
{
    using C = Cfg<HS, VS, OUT>;
    if (p.total_tiles <= 0) return hipSuccess;
    const dim3 grid((unsigned)p.total_tiles), block(C::NT);
    int gen; bool ts;
    pick(variant, OUT, fast != 0, ts_eligible<C>(p, OUT, fast != 0, fast == 2), gen, ts);
    const size_t dyn = (size_t)g_pad_lds;
    constexpr bool TSC = C::TSCAP; // staged stores exist for the 3-byte interleaved outputs only
    if (fast == 2 && gen == GEN_PACKED) {
        if (ts && TSC) hipLaunchKernelGGL((zj_fused_ragged_kernel<HS, VS, OUT, TSC>), grid, block, dyn, s, p);
        else hipLaunchKernelGGL((zj_fused_ragged_kernel<HS, VS, OUT, false>), grid, block, dyn, s, p);
        return hipGetLastError();
    }
    if (fast == 2) fast = 0; // (the wide generation has no ragged form)
    if (gen == GEN_WIDE) {
#if defined(ZJ_VARIANTS_ALL)
        if (fast) hipLaunchKernelGGL((zj_fused_kernel<HS, VS, OUT, GEN_WIDE, true, false>), grid, block, dyn, s, p);
        else hipLaunchKernelGGL((zj_fused_kernel<HS, VS, OUT, GEN_WIDE, false, false>), grid, block, dyn, s, p);
#else
        return hipErrorNotSupported; // (the product build leaves round 1's generation out: make VARIANTS=all)
#endif
    } else if (!fast) hipLaunchKernelGGL((zj_fused_kernel<HS, VS, OUT, GEN_PACKED, false, false>), grid, block, dyn, s, p);
    else if (ts && TSC && seam_launch<C>(p)) {
        if constexpr (ZJ_SEAM_WB != 0 && C::YBR == 4 && TSC) hipLaunchKernelGGL((zj_fused_seam_kernel<HS, VS, OUT>), grid, block, dyn, s, p); // (the only shapes seam_launch admits)
    } else if (ts && TSC) hipLaunchKernelGGL((zj_fused_kernel<HS, VS, OUT, GEN_PACKED, true, TSC>), grid, block, dyn, s, p);
    else hipLaunchKernelGGL((zj_fused_kernel<HS, VS, OUT, GEN_PACKED, true, false>), grid, block, dyn, s, p);
    return hipGetLastError();
}

hipError_t launch_fused(int hs, int vs, int out, int variant, int fast, const Params& p, hipStream_t s)
{
#define ZJ_CASE(H, V, O) if (hs == H && vs == V && out == O) return launch_fused_t<H, V, O>(p, variant, fast, s);
    ZJ_CASE(1, 1, OUT_RGB) ZJ_CASE(1, 1, OUT_GRAY) ZJ_CASE(1, 1, OUT_YCBCR)
    ZJ_CASE(2, 1, OUT_RGB) ZJ_CASE(2, 1, OUT_GRAY) ZJ_CASE(2, 1, OUT_YCBCR)
    ZJ_CASE(1, 2, OUT_RGB) ZJ_CASE(1, 2, OUT_GRAY) ZJ_CASE(1, 2, OUT_YCBCR)
    ZJ_CASE(2, 2, OUT_RGB) ZJ_CASE(2, 2, OUT_GRAY) ZJ_CASE(2, 2, OUT_YCBCR)
    ZJ_CASE(1, 1, OUT_RGBA) ZJ_CASE(2, 1, OUT_RGBA) ZJ_CASE(1, 2, OUT_RGBA) ZJ_CASE(2, 2, OUT_RGBA)
    ZJ_CASE(1, 1, OUT_RGB_CHW) ZJ_CASE(2, 1, OUT_RGB_CHW) ZJ_CASE(1, 2, OUT_RGB_CHW) ZJ_CASE(2, 2, OUT_RGB_CHW)
#undef ZJ_CASE
    return hipErrorInvalidValue;
}

// does this build carry round 1's kernel generation (variant 1: the A/B and N-version cross-check)?  make VARIANTS=all
bool fused_has_wide()
{
#if defined(ZJ_VARIANTS_ALL)
    return true;
#else
    return false;
#endif
}

// workgroups of the instantiation launch_fused() picks that fit one CU (occupancy query, once per instantiation)
template <int HS, int VS, int OUT>
static int slots_t(const Params& p, int variant, int fast)
{
    using C = Cfg<HS, VS, OUT>;
    int gen; bool ts;
    pick(variant, OUT, fast != 0, ts_eligible<C>(p, OUT, fast != 0), gen, ts);
    constexpr bool TSC = C::TSCAP;
    // 0 = not asked yet, -1 = the query failed.  Per device (a node may mix parts) and atomic: launch_params asks on every
    // 4:2:0 launch, from zj_multi's slot threads and the pool's submitters at once (two threads asking first both store the
    // same answer)
    constexpr int MAX_DEV = 64;
    static std::atomic<int> cache_all[MAX_DEV][6];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) dev = 0;
    std::atomic<int>* const cache = cache_all[dev];
    const int which = gen == GEN_WIDE ? (fast ? 0 : 1) : (!fast ? 2 : ((ts && TSC) ? 3 : 4));
    int have = cache[which].load(std::memory_order_relaxed);
    if (have == 0) {
        int n = 0;
        hipError_t e;
#if defined(ZJ_VARIANTS_ALL)
        if (which == 0) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, zj_fused_kernel<HS, VS, OUT, GEN_WIDE, true, false>, C::NT, 0);
        else if (which == 1) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, zj_fused_kernel<HS, VS, OUT, GEN_WIDE, false, false>, C::NT, 0);
        else
#else
        if (which < 2) e = hipErrorNotSupported;
        else
#endif
        if (which == 2) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, zj_fused_kernel<HS, VS, OUT, GEN_PACKED, false, false>, C::NT, 0);
        else if (which == 3) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, zj_fused_kernel<HS, VS, OUT, GEN_PACKED, true, TSC>, C::NT, 0);
        else e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, zj_fused_kernel<HS, VS, OUT, GEN_PACKED, true, false>, C::NT, 0);
        have = (e == hipSuccess && n > 0) ? n : -1;
        cache[which].store(have, std::memory_order_relaxed);
    }
    return have > 0 ? have : 0;
}

int fused_slots_per_cu(int hs, int vs, int out, int variant, int fast, const Params& p)
{
#define ZJ_CASE(H, V, O) if (hs == H && vs == V && out == O) return slots_t<H, V, O>(p, variant, fast);
    ZJ_CASE(2, 2, OUT_RGB) ZJ_CASE(2, 2, OUT_YCBCR) ZJ_CASE(2, 2, OUT_RGBA) ZJ_CASE(2, 2, OUT_RGB_CHW) // (the shapes that stagger)
#undef ZJ_CASE
    return 0;
}

template <int HS, int VS, int OUT>
static bool ts_ok_t(const Params& p, int fast) { return ts_eligible<Cfg<HS, VS, OUT>>(p, OUT, fast != 0); }
template <int HS, int VS, int OUT>
static bool seam_t(const Params& p) { return seam_launch<Cfg<HS, VS, OUT>>(p); }

const char* fused_kernel_name(int hs, int vs, int out, int variant, int fast, const Params& p)
{
    // the demangled name rocprofv3 prints for the instantiation launch_fused() picks
    thread_local char buf[8][112];
    thread_local int slot = 0;
    char* b = buf[slot++ & 7];
    bool ok = false;
#define ZJ_CASE(H, V, O) if (hs == H && vs == V && out == O) ok = ts_ok_t<H, V, O>(p, fast);
    ZJ_CASE(1, 1, OUT_RGB) ZJ_CASE(2, 1, OUT_RGB) ZJ_CASE(1, 2, OUT_RGB) ZJ_CASE(2, 2, OUT_RGB)
    ZJ_CASE(1, 1, OUT_YCBCR) ZJ_CASE(2, 1, OUT_YCBCR) ZJ_CASE(1, 2, OUT_YCBCR) ZJ_CASE(2, 2, OUT_YCBCR)
    ZJ_CASE(1, 1, OUT_RGBA) ZJ_CASE(2, 1, OUT_RGBA) ZJ_CASE(1, 2, OUT_RGBA) ZJ_CASE(2, 2, OUT_RGBA)
#undef ZJ_CASE
    int gen; bool ts;
    pick(variant, out, fast != 0, ok || (fast == 2 && (out == OUT_RGB || out == OUT_YCBCR || out == OUT_RGBA)), gen, ts);
    bool seam = false;
    if (fast == 1 && ts && hs == 2 && vs == 2) seam = out == OUT_RGB ? seam_t<2, 2, OUT_RGB>(p) : (out == OUT_YCBCR ? seam_t<2, 2, OUT_YCBCR>(p) : seam_t<2, 2, OUT_RGBA>(p));
    if (seam) snprintf(b, 112, "void zj::zj_fused_seam_kernel<%d, %d, %d>(zj::Params)", hs, vs, out);
    else if (fast == 2 && gen == GEN_PACKED) snprintf(b, 112, "void zj::zj_fused_ragged_kernel<%d, %d, %d, %s>(zj::Params)", hs, vs, out, ts ? "true" : "false");
    else snprintf(b, 112, "void zj::zj_fused_kernel<%d, %d, %d, %d, %s, %s>(zj::Params)", hs, vs, out, gen, (fast == 1) ? "true" : "false", ts ? "true" : "false");
    return b;
}

// ------------------------------------------------------------------------------------------------
// strip-level kernels (fn-pointer compatible API; not the hot path)
// ------------------------------------------------------------------------------------------------
// dequantize_and_idct_int (src/idct/scalar.rs:19-282): one lane per block.  `out` pre-zeroed.
struct Tab1 { uint32_t t[TAB_DW]; }; // one component's table (build_table), by value like Params::tab
__global__ __launch_bounds__(64) void zj_idct_strip_kernel(const int16_t* __restrict__ coeff, const Tab1 tab,
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
        const uint32_t v = dc_only_value(cw[0], (int32_t)(tab.t[0] & 0xffffu));
#pragma unroll
        for (int i = 0; i < 8; i++) { px[i].x = v; px[i].y = v; px[i].z = v; px[i].w = v; }
    } else {
        idct_block(raw, reinterpret_cast<const uint16_t*>(tab.t), px);
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

hipError_t launch_idct_strip(const int16_t* coeff, const int32_t qt[64], int16_t* out, long long nblocks,
                             long long chunks, long long bpc, long long stride, hipStream_t s)
{
    if (nblocks <= 0) return hipSuccess;
    Tab1 tab;
    build_table(qt, tab.t);
    hipLaunchKernelGGL(zj_idct_strip_kernel, dim3((unsigned)((nblocks + 63) / 64)), dim3(64), 0, s, coeff, tab,
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
