// zj_persist.hip -- LAB (libzjlab.so, tools/persist_lab.py; never in the product): the structural experiment DESIGN.md
// section 9 named since round 2 and VERDICT r3 asked to settle -- PERSISTENT workgroups whose next tile's raw coefficients are
// prefetched by LDS-DMA while the current tile is transformed, coloured and stored.
//
// It is the real skeleton: the product's own device code (zj_device.h: locate / finish_block / halo_* / phase_color /
// stage_item / color_copyout) for the 4:2:0 -> RGB tile, bit-checked against the product kernel's output by the tool.
//
//   MODE 0  persistent + prefetch.  A workgroup loops over its tiles.  For tile k+1 every lane of the three block waves
//           issues its block's eight 16-byte rows as global_load_lds_dwordx4 (instruction i of wave w lands in the 1 KB
//           slab P[w][i], lane-linear, so the later ds_read_b128 of "my block's row i" is conflict-free without a
//           swizzle), the halo wave its eight coefficients as dword DMAs.  No VGPR is held across the colour phase
//           (round 3's register-prefetch attempt died of exactly that).  Cost: 24 KB + 2 KB of LDS on top of the tile's
//           25.8 KB -> 3 workgroups per CU instead of 6.
//   MODE 1  persistent only (loads into VGPRs at the top of each tile, as the product does): 6 workgroups per CU.
//           Separates "no relaunch, tables staged once" from the prefetch.
//
// Differences from the product kernel, all stated in the tool's output: the Q1 redo path (an unclamped DC-only luma value
// outside 0..255 sends the tile to the wide code) is left out -- the synthetic frames never take it and the output
// comparison would show it.
#include <hip/hip_runtime.h>

#include "../zj_device.h"
#include "zj_lab_launch.h"

namespace zj {

template <int MODE>
__global__ __launch_bounds__(256, (MODE == 0 ? 3 : 4)) void zj_persist_kernel(const Params p)
{
    using C = Cfg<2, 2, OUT_RGB>;
    static_assert(C::NT == 256 && C::HALO_PURE && C::HALO_T0 == 192, "the 256-pixel 4:2:0 tile: 2 luma waves, 1 chroma wave, the halo wave");
    constexpr int WORK = (C::LDS_PACKED + 1023) / 1024 * 1024;
    constexpr int P_OFF = WORK, P_BYTES = 3 * 8 * 1024, PH_OFF = P_OFF + P_BYTES, PH_BYTES = 8 * 256;
    __shared__ __attribute__((aligned(1024))) char lds[MODE == 0 ? PH_OFF + PH_BYTES : WORK];
    const int tid = (int)threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid) >> 6;
    const bool halo_wave = wave == 3;

    // tiles of this workgroup: XCD x = bid % 8 owns a contiguous eighth of the launch's tiles (neighbouring tiles share halo
    // blocks in one L2, as in the product's xcd_order); its workgroups walk that range with a stride
    const int n = p.total_tiles, G = (int)gridDim.x, bid = (int)blockIdx.x;
    int first, stride, last;
    if ((n & 7) == 0 && (G & 7) == 0) { const int per = n >> 3; first = (bid & 7) * per + (bid >> 3); stride = G >> 3; last = (bid & 7) * per + per; }
    else { first = bid; stride = G; last = n; }
    if (first >= last) return;

    phase_setup<C, 2, 2, GEN_PACKED>(p, tid, lds); // tables, vertical LUT: ONCE per workgroup
    BlockLoc L, Ln;
    HaloLane H, Hn;
    L.valid = false; Ln = L;
    int hsel = 0, hsel_n = 0;

    auto issue = [&](const TileId t, BlockLoc& Lx, HaloLane& Hx, int& hs) {
        if (halo_wave) {
            Hx = halo_locate<C>(p, t, tid - C::HALO_T0, lds);
            hs = (int)((reinterpret_cast<uintptr_t>(Hx.src) >> 1) & 1);
            if (MODE == 0) {
                const char* g = reinterpret_cast<const char*>(reinterpret_cast<uintptr_t>(Hx.src) & ~(uintptr_t)3);
#pragma unroll
                for (int k = 0; k < 8; k++)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + 16 * k),
                                                     (__attribute__((address_space(3))) void*)(lds + PH_OFF + 256 * k), 4, 0, 0);
            }
        } else {
            Lx = locate<C, GEN_PACKED>(p, t, tid, lds);
            if (MODE == 0 && Lx.valid) {
#pragma unroll
                for (int i = 0; i < 8; i++)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Lx.src + i),
                                                     (__attribute__((address_space(3))) void*)(lds + P_OFF + (wave * 8 + i) * 1024), 16, 0, 0);
            }
        }
    };

    int id = first;
    TileId t = tile_from_id(p, id);
    if (MODE == 0) issue(t, L, H, hsel);
    bool first_iter = true, prev_plain = false;
    __syncthreads(); // tables staged
    for (;;) {
        U4 raw[8];
        int32_t hs8[8];
        if (MODE == 0) {
            // my DMA of this tile is older than the (at most 6) pixel stores of the previous one; gfx9 has ONE counter for
            // loads and stores and retires them in order, so "at most 6 outstanding" means the DMA has landed -- when the
            // previous tile issued exactly 6 stores in this wave (an interior tile); otherwise wait for everything
            if (first_iter || !prev_plain) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            if (halo_wave) {
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const uint32_t w = *reinterpret_cast<const uint32_t*>(lds + PH_OFF + 256 * k + 4 * lane);
                    hs8[k] = hsel ? (int32_t)w >> 16 : (int32_t)(int16_t)(w & 0xffffu);
                }
            } else if (L.valid) {
#pragma unroll
                for (int i = 0; i < 8; i++) raw[i] = *reinterpret_cast<const U4*>(lds + P_OFF + (wave * 8 + i) * 1024 + 16 * lane);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // my reads of P are done before the next DMA overwrites it
        } else {
            if (halo_wave) { H = halo_locate<C>(p, t, tid - C::HALO_T0, lds); halo_load(H, hs8); }
            else { L = locate<C, GEN_PACKED>(p, t, tid, lds); load_block(L, raw); }
        }
        const int idn = id + stride;
        const bool has_next = idn < last;
        TileId tn = t;
        if (has_next) {
            tn = tile_from_id(p, idn);
            if (MODE == 0) issue(tn, Ln, Hn, hsel_n);
        }
        // the previous tile's colour phase (other waves) may still read the planes and the staging area this tile's
        // transform writes
        if (!first_iter) __syncthreads();
        if (halo_wave) {
            halo_pass1<C>(H, hs8, lds);
            ZJ_WAVE_FENCE();
            halo_pass2<C>(H, lds, p.clamp_dc);
            ZJ_WAVE_FENCE();
            halo_filter<C, 2, 2>(p, t, tid - C::HALO_T0, lds);
        } else {
            finish_block<C, GEN_PACKED, false>(L, raw, lds, 0, p.clamp_dc);
        }
        if (MODE != 0) __builtin_amdgcn_s_waitcnt(0x0f70);
        __syncthreads();
#pragma unroll
        for (int round = 0; round * C::NT < C::NITEMS; round++) {
            ItemOut io;
            phase_color<C, 2, 2, OUT_RGB, GEN_PACKED, true, true>(p, t, tid, lds, round, &io);
            stage_item<C>(io, tid, lds, round);
            ZJ_WAVE_FENCE();
            color_copyout<C, OUT_RGB>(p, t, tid, lds, round);
            ZJ_WAVE_FENCE();
        }
        if (!has_next) break;
        {   // did this wave issue exactly six stores for this tile?  (color_copyout's plain_round condition)
            const int P = p.mcu_x * 16, x0 = t.tile * C::TWY;
            prev_plain = (P - x0) / 16 >= C::NGRP && p.height - t.strip * C::SH >= C::SH &&
                         !(!p.plain && !p.zero_fill && x0 + 16 * C::NGRP == P);
        }
        id = idn; t = tn; L = Ln; H = Hn; hsel = hsel_n;
        first_iter = false;
    }
}

hipError_t launch_persist(int mode, const Params& p, int groups, hipStream_t s)
{
    if (p.total_tiles <= 0 || groups <= 0) return hipErrorInvalidValue;
    if (mode == 0) hipLaunchKernelGGL(zj_persist_kernel<0>, dim3((unsigned)groups), dim3(256), 0, s, p);
    else hipLaunchKernelGGL(zj_persist_kernel<1>, dim3((unsigned)groups), dim3(256), 0, s, p);
    return hipGetLastError();
}
int persist_occupancy(int mode)
{
    int n = -1;
    if (mode == 0) { if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, zj_persist_kernel<0>, 256, 0) != hipSuccess) return -1; }
    else if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, zj_persist_kernel<1>, 256, 0) != hipSuccess) return -1;
    return n;
}

} // namespace zj
