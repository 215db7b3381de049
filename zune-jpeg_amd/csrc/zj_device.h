// zj_device.h -- device-side code of the MI355X pixel path (dequantize + 8x8 integer IDCT, chroma
// up-sampling, YCbCr->RGB/gray/YCbCr), written for gfx950 (wave64, LDS, packed-i16 VALU, v_dot2_i32_i16).
//
// The arithmetic restates zune-jpeg's SCALAR arms bit-exactly (paths relative to the reference):
//   src/idct/scalar.rs:19-282, src/upsampler/scalar.rs:5-166, src/color_convert/scalar.rs:52-169,
//   src/worker.rs:32-251.  Quirk numbers (Q1..Q8) refer to SURVEY.md section 8a.
//
// Two kernel generations share this file:
//   GEN_WIDE    round 1: 24-bit multiply-add IDCT, int16 planar staging in LDS, 48-byte-per-lane stores.  Exact
//               for every input without any range check; it is also the fall-back of the packed generation.
//   GEN_PACKED  round 2: the IDCT runs on packed i16 pairs with v_dot2_i32_i16 whenever a cheap L1 bound proves
//               that this is exact (classify_block), luma is staged as bytes, and the 3-byte interleaved outputs
//               leave through an LDS transpose so that every store instruction writes contiguous memory.
//
// The same header is compiled (a) by hipcc into libzjhip.so and (b) by g++ into the CPU
// *emulation* harness under tests/emu/, which runs every workgroup phase thread by thread so the
// indexing can be checked against the oracle without a GPU.  (b) is test infrastructure only: the
// product library contains no CPU path.  ZJ_EMU selects (b).
#pragma once

#include <stdint.h>

#if defined(ZJ_EMU)
#define ZJ_DEV inline
#define ZJ_HD inline
#else
#include <hip/hip_runtime.h>
#define ZJ_DEV __device__ __forceinline__
#define ZJ_HD __host__ __device__ __forceinline__
#endif

namespace zj {

// ------------------------------------------------------------------------------------------------
// small portable vector / intrinsic layer
// ------------------------------------------------------------------------------------------------
typedef uint16_t u16x2 __attribute__((vector_size(4)));  // packed pair, wrap-around + - *
typedef int16_t s16x2 __attribute__((vector_size(4)));   // packed pair, arithmetic >>, min/max

// an address held as an integer -> a pointer the compiler knows to be in GLOBAL memory (address space 1), so that what is
// loaded / stored through it stays global_load / global_store (scalar base + 32-bit lane offset) instead of flat_*
#if defined(ZJ_EMU)
#define ZJ_GLOBAL_PTR(T, v) (reinterpret_cast<T*>(v))
#else
#define ZJ_GLOBAL_PTR(T, v) ((T*)(__attribute__((address_space(1))) T*)(v))
#endif

struct alignas(16) U4 { uint32_t x, y, z, w; };
struct alignas(8) U2 { uint32_t x, y; };
typedef uint32_t V4 __attribute__((vector_size(16)));    // the same 16 bytes for builtins that want a vector
#if defined(ZJ_ABLATION)
#define ZJ_ABL(debug, bit) ((debug) & (bit))
#define ZJ_PDBG(p) ((p).debug)
#else
#define ZJ_ABL(debug, bit) 0  // the ablation switches exist only in the diagnostic build (tools/ablate.py)
#define ZJ_PDBG(p) 0          // ... and so does Params::debug: nothing of it travels in a product launch
#endif
// ZJ_NT (tools/ab_libs.sh): bit 0 = non-temporal pixel stores everywhere, bit 1 = non-temporal coefficient loads,
// bit 2 = non-temporal stores where a store instruction writes whole, lane-contiguous lines (the staged stores of
// color_copyout, the grayscale rows).  Default 4: these lines are written once and nobody reads them
// again (+1.5 % measured); the 48-byte-per-lane direct stores of the other paths LOSE 2.5 % with it (round 1).
#ifndef ZJ_NT
#define ZJ_NT 4
#endif
// 1: aligned widths whose rows do not start on 128-byte boundaries (a pitch or a base that is not a multiple of 128) are
//    decoded by the SEAM instantiation (4:2:0): the pieces of a 128-byte line that a tile's row segment shares with the
//    neighbouring tile are stored write-back, so that the L2 puts the two halves together, and only the lines the tile
//    owns stream out non-temporally (color_copyout; tools/store_probe).  0 (default): the family is not even compiled --
//    over four boxes it measured between +8 % and -5 % on such frames (profiles/r05_store_probe.txt); the layout that
//    repairs the seams for good is zj_frame_desc.out_pitch.
#ifndef ZJ_SEAM_WB
#define ZJ_SEAM_WB 0
#endif
// 1: a colour round that does not fill the workgroup is served by its last wave (Cfg::ROUND_ROT); 0: by its first (A/B knob)
#ifndef ZJ_ROUND_ROT
#define ZJ_ROUND_ROT 1
#endif

ZJ_DEV uint32_t as_u32(u16x2 v) { uint32_t r; __builtin_memcpy(&r, &v, 4); return r; }
ZJ_DEV uint32_t as_u32(s16x2 v) { uint32_t r; __builtin_memcpy(&r, &v, 4); return r; }
ZJ_DEV u16x2 as_u16x2(uint32_t v) { u16x2 r; __builtin_memcpy(&r, &v, 4); return r; }
ZJ_DEV s16x2 as_s16x2(uint32_t v) { s16x2 r; __builtin_memcpy(&r, &v, 4); return r; }

// 24-bit multiply: exact (== wrapping 32-bit product) iff both operands fit in signed 24 bits.
// The emulation build implements the hardware semantics literally so range violations show up.
#if !defined(ZJ_EMU)
// the LLVM intrinsic itself (no clang builtin exists); selects v_mul_i32_i24 / v_mad_i32_i24 / SDWA forms
extern "C" __device__ __attribute__((const)) int zj_llvm_mul_i24(int, int) __asm("llvm.amdgcn.mul.i24");
#endif
ZJ_DEV int32_t mul24(int32_t a, int32_t b)
{
#if defined(ZJ_EMU)
    int32_t a24 = (int32_t)((uint32_t)a << 8) >> 8, b24 = (int32_t)((uint32_t)b << 8) >> 8;
    return (int32_t)((uint32_t)a24 * (uint32_t)b24);
#else
    return zj_llvm_mul_i24(a, b);
#endif
}
ZJ_DEV int32_t wadd(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
ZJ_DEV int32_t wsub(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); }
ZJ_DEV int32_t mad24(int32_t a, int32_t b, int32_t c) { return wadd(mul24(a, b), c); }
ZJ_DEV int32_t wshl(int32_t a, int s) { return (int32_t)((uint32_t)a << s); }

// v_dot2_i32_i16: a.lo*b.lo + a.hi*b.hi + c, the halves read as signed 16-bit, the sum wrapping mod 2^32 (no
// clamp) -- gfx950's counterpart of the pmaddwd the reference's AVX2 arm is built on.  `k` is a compile-time pair of
// constants.  Written as the three-address VOP3P form in inline asm: left to itself the compiler picks the
// two-address v_dot2c_i32_i16 and pays a v_mov for every accumulator that is used twice or starts at zero
// (132 moves per block, a fifth of the transform).
ZJ_DEV int32_t dot2(uint32_t a, uint32_t k, int32_t c)
{
#if defined(ZJ_EMU)
    const int32_t al = (int16_t)(a & 0xffff), ah = (int16_t)(a >> 16), bl = (int16_t)(k & 0xffff), bh = (int16_t)(k >> 16);
    return (int32_t)((uint32_t)(al * bl) + (uint32_t)(ah * bh) + (uint32_t)c);
#else
    int32_t r;
    asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(k), "v"(c));
    return r;
#endif
}
// the same with a zero accumulator (inline constant)
ZJ_DEV int32_t dot2z(uint32_t a, uint32_t k)
{
#if defined(ZJ_EMU)
    return dot2(a, k, 0);
#else
    int32_t r;
    asm("v_dot2_i32_i16 %0, %1, %2, 0" : "=v"(r) : "v"(a), "s"(k));
    return r;
#endif
}
// a packed pair of signed 16-bit constants (lo, hi)
constexpr uint32_t pk16(int lo, int hi) { return ((uint32_t)lo & 0xffffu) | (((uint32_t)hi & 0xffffu) << 16); }

// v_sad_u16: |a.lo - b.lo| + |a.hi - b.hi| + c, the halves read as UNSIGNED 16-bit
ZJ_DEV uint32_t sad_u16(uint32_t a, uint32_t b, uint32_t c)
{
#if defined(ZJ_EMU)
    const int32_t al = (int32_t)(a & 0xffff), ah = (int32_t)(a >> 16), bl = (int32_t)(b & 0xffff), bh = (int32_t)(b >> 16);
    return (uint32_t)((al > bl ? al - bl : bl - al) + (ah > bh ? ah - bh : bh - ah)) + c;
#else
    return __builtin_amdgcn_sad_u16(a, b, c);
#endif
}

// {(a >> sh)[15:0], (b >> sh)[15:0]} as a packed pair: an arithmetic shift and a second one whose SDWA form writes
// its low half into the destination's high word (2 instructions; the compiler's own sequence is shift, shift, bfi)
template <int SH>
ZJ_DEV uint32_t pack_sar(int32_t a, int32_t b)
{
#if defined(ZJ_EMU)
    return ((uint32_t)(a >> SH) & 0xffffu) | ((uint32_t)(b >> SH) << 16);
#else
    // the shift count as an inline constant: with an SGPR operand the e32 shift issues in 4.1 cycles instead of 2.3
    // (profiles/r01_ubench_valu_issue_cost.txt: simple VOP2 instructions are fast only with VGPR / inline operands)
    uint32_t r;
    asm("v_ashrrev_i32_e32 %0, %1, %2" : "=v"(r) : "n"(SH), "v"(a));
    asm("v_ashrrev_i32_sdwa %0, %1, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(r) : "n"(SH), "v"(b));
    return r;
#endif
}

// v_perm_b32: bytes {s0[3..0] -> 7..4, s1[3..0] -> 3..0}; selector byte i picks result byte i.
// v_alignbyte_b32: bytes [s, s + 4) of the 8-byte value hi:lo (s = 0..3)
ZJ_DEV uint32_t alignbyte(uint32_t hi, uint32_t lo, uint32_t s)
{
#if defined(ZJ_EMU)
    return (uint32_t)((((uint64_t)hi << 32) | lo) >> (8 * (s & 3u)));
#else
    return __builtin_amdgcn_alignbyte(hi, lo, s);
#endif
}
ZJ_DEV uint32_t perm(uint32_t s0, uint32_t s1, uint32_t sel)
{
#if defined(ZJ_EMU)
    uint64_t src = ((uint64_t)s0 << 32) | s1;
    uint32_t r = 0;
    for (int i = 0; i < 4; i++) {
        uint32_t s = (sel >> (8 * i)) & 0xff;
        uint32_t b = s < 8 ? (uint32_t)((src >> (8 * s)) & 0xff) : (s == 12 ? 0u : 0xffu);
        r |= b << (8 * i);
    }
    return r;
#else
    return __builtin_amdgcn_perm(s0, s1, sel);
#endif
}
// (hi:lo) >> 16, i.e. {lo.hi16, hi.lo16}
ZJ_DEV uint32_t align16(uint32_t hi, uint32_t lo) { return (lo >> 16) | (hi << 16); }

ZJ_DEV s16x2 pk_max(s16x2 a, s16x2 b)
{
#if defined(ZJ_EMU)
    s16x2 r; for (int i = 0; i < 2; i++) r[i] = a[i] > b[i] ? a[i] : b[i]; return r;
#else
    return __builtin_elementwise_max(a, b);
#endif
}
ZJ_DEV s16x2 pk_min(s16x2 a, s16x2 b)
{
#if defined(ZJ_EMU)
    s16x2 r; for (int i = 0; i < 2; i++) r[i] = a[i] < b[i] ? a[i] : b[i]; return r;
#else
    return __builtin_elementwise_min(a, b);
#endif
}
ZJ_DEV u16x2 splat(int v) { u16x2 r = {(uint16_t)v, (uint16_t)v}; return r; }
ZJ_DEV s16x2 sar(u16x2 a, int s) { s16x2 t = (s16x2)a; s16x2 sh = {(int16_t)s, (int16_t)s}; return t >> sh; }

ZJ_DEV int uniform(int v)
{
#if defined(ZJ_EMU)
    return v;
#else
    return __builtin_amdgcn_readfirstlane(v);
#endif
}

// pin(): keeps a partial sum opaque so the compiler cannot re-associate a multiply-add chain into
// mul + mul + add3 (one instruction more per output); every VALU instruction costs ~4 cycles per
// wave on gfx950 (profiles/r01_ubench_valu_issue_cost.txt), so instruction count is the metric.
#if defined(ZJ_EMU)
#define ZJ_PIN(x) ((void)0)
#else
#define ZJ_PIN(x) asm volatile("" : "+v"(x))
#endif
// keeps a rarely taken, wave-uniform branch a branch (an empty asm statement cannot be if-converted into selects)
#if defined(ZJ_EMU)
#define ZJ_NO_IF_CONVERT() ((void)0)
#else
#define ZJ_NO_IF_CONVERT() asm volatile("; zj-rare-branch") // (no memory clobber: that would force register arrays into scratch; the comment marks the block for tools/valu_ledger.py)
#endif
#if defined(ZJ_EMU)
#define ZJ_BRANCH_TAG(text) ((void)0)
#else
#define ZJ_BRANCH_TAG(text) asm volatile(text)
#endif
#if defined(ZJ_EMU) || defined(ZJ_IDCT_NOBARRIER)
#define ZJ_SCHED_BARRIER() ((void)0)
#else
#define ZJ_SCHED_BARRIER() __builtin_amdgcn_sched_barrier(0)
#endif
#ifndef ZJ_IDCT_G
#define ZJ_IDCT_G 2 // 1, 2 and 4 measure the same within noise in the full kernel (tools/ab_libs.sh); 8 spills
#endif
// experiment knob: scheduling barriers between the stages of the colour phase (filters | colour math | packing)
#if defined(ZJ_COLOR_STAGES) && !defined(ZJ_EMU)
#define ZJ_COLOR_SB() __builtin_amdgcn_sched_barrier(0)
#else
#define ZJ_COLOR_SB() ((void)0)
#endif
// a use the compiler cannot move: everything the value depends on is issued before this point
#if defined(ZJ_EMU)
#define ZJ_USE(x) ((void)(x))
#else
#define ZJ_USE(x) asm volatile("" ::"v"(x))
#endif
// a compiler-level fence between the LDS writes and reads of one wave's transpose (the hardware executes the LDS
// operations of a wave in order; nothing is needed there)
#if defined(ZJ_EMU)
#define ZJ_WAVE_FENCE() ((void)0)
#else
#define ZJ_WAVE_FENCE() __builtin_amdgcn_wave_barrier()
#endif

ZJ_DEV int32_t lo16s(uint32_t v) { return (int32_t)(int16_t)(v & 0xffff); }
ZJ_DEV int32_t hi16s(uint32_t v) { return (int32_t)v >> 16; }

// v_sat_pk_u8_i16: {0, 0, sat_u8(hi16), sat_u8(lo16)} -- the two 16-bit lanes clamped to 0..255 and
// packed into bytes 0 and 1 (replaces v_pk_max_i16 + v_pk_min_i16 and half of the byte packing)
ZJ_DEV uint32_t sat_pk_u8(uint32_t v)
{
#if defined(ZJ_EMU)
    int lo = (int16_t)(v & 0xffff), hi = (int16_t)(v >> 16);
    lo = lo < 0 ? 0 : (lo > 255 ? 255 : lo);
    hi = hi < 0 ? 0 : (hi > 255 ? 255 : hi);
    return (uint32_t)lo | ((uint32_t)hi << 8);
#else
    uint32_t r;
    asm("v_sat_pk_u8_i16_e32 %0, %1" : "=v"(r) : "v"(v));
    return r;
#endif
}

// lo = sat_pk_u8(a) in bits 0..15, sat_pk_u8(b) in bits 16..31 (SDWA write into the high word)
ZJ_DEV uint32_t sat_pk_u8_2(uint32_t a, uint32_t b)
{
#if defined(ZJ_EMU)
    return sat_pk_u8(a) | (sat_pk_u8(b) << 16);
#else
    uint32_t r;
    asm("v_sat_pk_u8_i16_e32 %0, %1" : "=v"(r) : "v"(a));
    asm("v_sat_pk_u8_i16_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(r) : "v"(b));
    return r;
#endif
}

// ------------------------------------------------------------------------------------------------
// Quantisation tables as the kernels see them: per component 44 dwords,
//   [0, 32)   the 64 entries as packed u16 pairs, natural order (pair i = entries 2i, 2i+1)
//   [32, 44)  the constants of the packed-IDCT guard (classify_block): wt[4], wb[4], th[4]
// They travel BY VALUE in the kernel arguments (Params::tab): no device-side table, hence no ordering
// between an upload and the launches of other streams to get wrong.
// ------------------------------------------------------------------------------------------------
constexpr int TAB_DW = 44;                 // dwords per component
constexpr int TAB_BYTES = 3 * TAB_DW * 4;  // 528
constexpr int GUARD_LIMIT = 5904;          // see classify_block

// host side (zj_plan.h, the emulator): q[64] natural order, each 0..255
inline void build_table(const int32_t q[64], uint32_t tab[TAB_DW])
{
    for (int i = 0; i < 32; i++) tab[i] = ((uint32_t)q[2 * i] & 0xffffu) | (((uint32_t)q[2 * i + 1] & 0xffffu) << 16);
    for (int jj = 0; jj < 4; jj++) {
        int wt = 0, wb = 0;
        for (int k = 0; k < 8; k++)
            for (int h = 0; h < 2; h++) {
                const int v = q[8 * k + 2 * jj + h];
                if (k < 4) wt = v > wt ? v : wt; else wb = v > wb ? v : wb;
            }
        tab[32 + jj] = (uint32_t)wt;
        tab[36 + jj] = (uint32_t)wb;
        tab[40 + jj] = (uint32_t)(262144 * (wt + wb) - GUARD_LIMIT);
    }
}

// ------------------------------------------------------------------------------------------------
// 8x8 dequantize + IDCT of ONE block held by ONE lane (64 coefficients in 8 x 16-byte registers).
//
// Algebra: every step of the reference butterfly (scalar.rs:79-274) is +,-,* by constants and <<
// in wrapping i32, i.e. ring operations mod 2^32, so any re-association is bit-exact.  The odd
// part is expanded into its 4x4 integer matrix so that the only multiplicands are the pass inputs
// themselves: dequantized coefficients |c*q| <= 32768*255 < 2^23 in pass 1 and (x >> 10) in
// [-2^21, 2^21) in pass 2.
//
// WIDE form (idct_block): both always fit the signed 24-bit operand of v_mul_i32_i24 / v_mad_i32_i24,
// which makes the full-rate 24-bit multiplier exact for EVERY input (including the wrap-around
// adversarial ones) without any range check.  Requires 0 <= q <= 255 (8-bit DQT, headers.rs:154-174),
// enforced by the host side.
// ------------------------------------------------------------------------------------------------
// The transform is written as two halves -- every multiply / multiply-add first, the butterfly adds after --
// so that idct_block can put the add/shift halves of two transforms next to each other behind a scheduling
// barrier: simple VOP2 instructions issue faster next to each other than interleaved with multiplies
// (profiles/r01_ubench_valu_issue_cost.txt; -3.4 % on the isolated IDCT, profiles/r01_lab_*).
struct IdctHalf { int32_t t0, t1, t2, t3, u0, u1, u2, u3; };
ZJ_DEV IdctHalf idct_1d_mul(const int32_t s[8], const int32_t bias)
{
    IdctHalf h;
    // even part: t3 = (s2+s6)*2217 + s2*3135, t2 = (s2+s6)*2217 - s6*7567      (scalar.rs:81-87)
    h.t3 = mul24(s[2], 2217 + 3135); ZJ_PIN(h.t3); h.t3 = mad24(s[6], 2217, h.t3);
    h.t2 = mul24(s[2], 2217); ZJ_PIN(h.t2); h.t2 = mad24(s[6], 2217 - 7567, h.t2);
    // t0 = fsh(s0+s4) + bias, t1 = fsh(s0-s4) + bias                           (:93-99)
    int32_t A = wadd(wshl(s[0], 12), bias); ZJ_PIN(A);
    h.t0 = wadd(wshl(s[4], 12), A); ZJ_PIN(h.t0); // v_lshl_add_u32
    h.t1 = mad24(s[4], -4096, A);
    // odd part (scalar.rs:109-148) as the integer matrix it is; a=s7 b=s5 c=s3 d=s1
    //   u3 = d*6149 + p1 + p4 ... expanded: e.g. coefficient of d in u3 = 6149 + 4816 - 3685 - 1597
    const int32_t a = s[7], b = s[5], c = s[3], d = s[1];
    h.u3 = mul24(d, 6149 + 4816 - 3685 - 1597); ZJ_PIN(h.u3); h.u3 = mad24(a, 4816 - 3685, h.u3); ZJ_PIN(h.u3); h.u3 = mad24(b, 4816 - 1597, h.u3); ZJ_PIN(h.u3); h.u3 = mad24(c, 4816, h.u3);
    h.u2 = mul24(c, 12586 + 4816 - 10497 - 8034); ZJ_PIN(h.u2); h.u2 = mad24(b, 4816 - 10497, h.u2); ZJ_PIN(h.u2); h.u2 = mad24(a, 4816 - 8034, h.u2); ZJ_PIN(h.u2); h.u2 = mad24(d, 4816, h.u2);
    h.u1 = mul24(b, 8410 + 4816 - 10497 - 1597); ZJ_PIN(h.u1); h.u1 = mad24(c, 4816 - 10497, h.u1); ZJ_PIN(h.u1); h.u1 = mad24(d, 4816 - 1597, h.u1); ZJ_PIN(h.u1); h.u1 = mad24(a, 4816, h.u1);
    h.u0 = mul24(a, 1223 + 4816 - 3685 - 8034); ZJ_PIN(h.u0); h.u0 = mad24(d, 4816 - 3685, h.u0); ZJ_PIN(h.u0); h.u0 = mad24(c, 4816 - 8034, h.u0); ZJ_PIN(h.u0); h.u0 = mad24(b, 4816, h.u0);
    return h;
}
ZJ_DEV void idct_1d_add(const IdctHalf& h, int32_t o[8])
{
    const int32_t x0 = wadd(h.t0, h.t3), x3 = wsub(h.t0, h.t3), x1 = wadd(h.t1, h.t2), x2 = wsub(h.t1, h.t2);
    o[0] = wadd(x0, h.u3); o[7] = wsub(x0, h.u3);
    o[1] = wadd(x1, h.u2); o[6] = wsub(x1, h.u2);
    o[2] = wadd(x2, h.u1); o[5] = wsub(x2, h.u1);
    o[3] = wadd(x3, h.u0); o[4] = wsub(x3, h.u0);
}
ZJ_DEV void idct_1d(const int32_t s[8], const int32_t bias, int32_t o[8])
{
    const IdctHalf h = idct_1d_mul(s, bias);
    idct_1d_add(h, o);
}

// raw[r] = coefficient row r (8 x i16, natural order).  qt: the block's 64 table entries (u16, LDS or kernel
// arguments).  out[r] = pixel row r as 8 packed i16 (4 dwords): level-shifted (+128) and clamped to 0..255
// (SCALE_BITS scalar.rs:6, clamp :302-305).  The caller handles DC-only blocks (scalar.rs:45-74).
ZJ_DEV void idct_block(const U4 raw[8], const uint16_t* qt, U4 out[8])
{
    const uint32_t* w = reinterpret_cast<const uint32_t*>(raw);
    int32_t tmp[64];
    constexpr int G = ZJ_IDCT_G; // transforms per group: multiply halves of G lines, barrier, their add/shift halves
    // pass 1: columns (scalar.rs:79-167), bias 512, >> 10
#pragma unroll
    for (int c0 = 0; c0 < 8; c0 += G) {
        IdctHalf h[G];
#pragma unroll
        for (int g = 0; g < G; g++) {
            const int col = c0 + g;
            int32_t s[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const uint32_t pair = w[k * 4 + (col >> 1)];
                const int32_t cf = (col & 1) ? hi16s(pair) : lo16s(pair);
                s[k] = mul24(cf, (int32_t)qt[k * 8 + col]); // dequantize (scalar.rs:308); q is 0..255
            }
            h[g] = idct_1d_mul(s, 512);
        }
        ZJ_SCHED_BARRIER();
#pragma unroll
        for (int g = 0; g < G; g++) {
            int32_t o[8];
            idct_1d_add(h[g], o);
#pragma unroll
            for (int k = 0; k < 8; k++) tmp[k * 8 + c0 + g] = o[k] >> 10;
        }
        ZJ_SCHED_BARRIER();
    }
    // pass 2: rows (scalar.rs:170-274), bias SCALE_BITS, >> 17, clamp
    constexpr int32_t bias2 = 512 + 65536 + (128 << 17);
    uint32_t* ow = reinterpret_cast<uint32_t*>(out);
#pragma unroll
    for (int r0 = 0; r0 < 8; r0 += G) {
        IdctHalf h[G];
#pragma unroll
        for (int g = 0; g < G; g++) h[g] = idct_1d_mul(&tmp[(r0 + g) * 8], bias2);
        ZJ_SCHED_BARRIER();
        int32_t o[G][8];
#pragma unroll
        for (int g = 0; g < G; g++) idct_1d_add(h[g], o[g]);
        ZJ_SCHED_BARRIER();
#pragma unroll
        for (int g = 0; g < G; g++)
#pragma unroll
            for (int k = 0; k < 8; k += 2) {
                // x >> 17 == (x >> 16) >> 1: one v_perm_b32 gathers the two high halves (x >> 16 as i16), one packed
                // shift finishes both, then both lanes are clamped at once (4 instructions per pair instead of 5)
                const uint32_t hi = perm((uint32_t)o[g][k + 1], (uint32_t)o[g][k], 0x07060302u);
                const s16x2 z = {0, 0}, m = {255, 255};
                ow[(r0 + g) * 4 + (k >> 1)] = as_u32(pk_min(pk_max(sar(as_u16x2(hi), 1), z), m));
            }
        ZJ_SCHED_BARRIER();
    }
}

// ------------------------------------------------------------------------------------------------
// PACKED form (idct_block_packed): the same integers from 16-bit operands.
//
// v_dot2_i32_i16 multiplies two signed 16-bit pairs and adds both products to a 32-bit accumulator: two of
// the butterfly's multiply-adds per instruction, 22 instructions per 1-D transform instead of 36.  The matrix
// entries all fit 16 bits (|entry| <= 5683); the INPUTS must too:
//   (1) every dequantized coefficient s = c*q of the block fits i16  (v_pk_mul_lo_u16 then yields s itself),
//   (2) every pass-1 result (x >> 10) fits i16, i.e. |x| < 2^25.
// With B_j = sum_k |s[k][j]| (column j):  |x| <= 5683 * B_j + 512, so B_j <= 5904 gives (2), and (1) follows
// from |s| <= B_j.  classify_block bounds B_j from the QUANTIZED coefficients with v_sad_u16 (32 instructions
// for the block): for column pair jj, a_k = |c[k][2jj]| + |c[k][2jj+1]|, weighted by the largest table entry
// of rows 0-3 (wt) and of rows 4-7 (wb) of that column pair:
//   B_2jj, B_2jj+1  <=  wt * sum_{k<4} a_k + wb * sum_{k>=4} a_k  <=  GUARD_LIMIT  for all four jj.
// Blocks that pass take the packed transform; the others (extreme contrast at low quantisation, adversarial
// test vectors) take idct_block.  Both produce the reference's integers, so which one ran is unobservable.
// ------------------------------------------------------------------------------------------------
struct PackedHalf { int32_t x0, x1, x2, x3, u0, u1, u2, u3; };
// p04 = (s0, s4), p26 = (s2, s6), p13 = (s1, s3), p57 = (s5, s7)
ZJ_DEV PackedHalf idct_1d_dot(const uint32_t p04, const uint32_t p26, const uint32_t p13, const uint32_t p57, const int32_t bias)
{
    PackedHalf h;
    const int32_t t0 = dot2(p04, pk16(4096, 4096), bias);   // fsh(s0+s4) + bias      (scalar.rs:93-99)
    const int32_t t1 = dot2(p04, pk16(4096, -4096), bias);  // fsh(s0-s4) + bias
    // t3 = 5352*s2 + 2217*s6, t2 = 2217*s2 - 5350*s6 (scalar.rs:81-87), folded into x0..x3
    h.x0 = dot2(p26, pk16(5352, 2217), t0);
    h.x3 = dot2(p26, pk16(-5352, -2217), t0);
    h.x1 = dot2(p26, pk16(2217, -5350), t1);
    h.x2 = dot2(p26, pk16(-2217, 5350), t1);
    // odd part (scalar.rs:109-148) as the 4x4 integer matrix; d = s1, c = s3, b = s5, a = s7
    h.u3 = dot2(p57, pk16(3219, 1131), dot2z(p13, pk16(5683, 4816)));
    h.u2 = dot2(p57, pk16(-5681, -3218), dot2z(p13, pk16(4816, -1129)));
    h.u1 = dot2(p57, pk16(1132, 4816), dot2z(p13, pk16(3219, -5681)));
    h.u0 = dot2(p57, pk16(4816, -5680), dot2z(p13, pk16(1131, -3218)));
    return h;
}
ZJ_DEV void idct_1d_dot_add(const PackedHalf& h, int32_t o[8])
{
    o[0] = wadd(h.x0, h.u3); o[7] = wsub(h.x0, h.u3);
    o[1] = wadd(h.x1, h.u2); o[6] = wsub(h.x1, h.u2);
    o[2] = wadd(h.x2, h.u1); o[5] = wsub(h.x2, h.u1);
    o[3] = wadd(h.x3, h.u0); o[4] = wsub(h.x3, h.u0);
}

// 0: DC-only (scalar.rs:45)   1: the packed transform is exact for this block   2: take the wide transform
// w = the block's 32 coefficient dwords, g = the component's guard constants (tab + 32)
ZJ_DEV int classify_block(const uint32_t* w, const uint32_t* g)
{
    uint32_t any = w[0] & 0xffff0000u; // all but coefficient 0 are zero?
#pragma unroll
    for (int i = 1; i < 32; i++) any |= w[i];
    if (any == 0) return 0;
    // v_sad_u16 against 0x8000 per half: 65536 - (|lo| + |hi|) for signed halves (|-32768| included)
    int32_t bad = 0;
#pragma unroll
    for (int jj = 0; jj < 4; jj++) {
        uint32_t st = 0, sb = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) st = sad_u16(w[4 * k + jj], 0x80008000u, st);
#pragma unroll
        for (int k = 4; k < 8; k++) sb = sad_u16(w[4 * k + jj], 0x80008000u, sb);
        // wt * (262144 - st) + wb * (262144 - sb) <= GUARD_LIMIT  <=>  wt*st + wb*sb - th >= 0
        const int32_t d = mad24((int32_t)g[4 + jj], (int32_t)sb, mad24((int32_t)g[jj], (int32_t)st, -(int32_t)g[8 + jj]));
        bad |= d;
    }
    return bad < 0 ? 2 : 1;
}

// raw: as idct_block.  qp: the component's 32 packed table pairs.  out[2r], out[2r+1] = pixel row r as 8 BYTES,
// level-shifted and clamped to 0..255.  Precondition: classify_block(...) == 1.
ZJ_DEV void idct_block_packed(const U4 raw[8], const uint32_t* qp, uint32_t out[16])
{
    const uint32_t* w = reinterpret_cast<const uint32_t*>(raw);
    uint32_t D[32]; // dequantized pairs (s[k][2jj], s[k][2jj+1]), scalar.rs:308
#pragma unroll
    for (int i = 0; i < 32; i++) D[i] = as_u32(as_u16x2(w[i]) * as_u16x2(qp[i]));
    // pass 1: columns (scalar.rs:79-167), bias 512, >> 10.  Two columns per group, chosen so that their results
    // pair up as pass 2 wants them: T[i][g] = (tmp[i][CA[g]], tmp[i][CB[g]]) = p04, p26, p13, p57 of row i.
    // (Round 4 tried a sparse form of this pass -- a column with nothing below its first row yields 4 * s0 in every row,
    // exactly; columns 5..7 qualify in every wave of q90 data -- behind a wave-uniform test: as a second instantiation it
    // gained 1.3 % on 4:2:0 and cost 18 % on dense input (the full form lost its register allocation), as branches inside
    // this body it gained nothing; profiles/r04_ab_history.txt.  The kernel sits on its memory floor by then.)
    uint32_t T[8][4];
#pragma unroll
    for (int g = 0; g < 4; g++) {
        const int ca = g == 0 ? 0 : (g == 1 ? 2 : (g == 2 ? 1 : 5)), cb = g == 0 ? 4 : (g == 1 ? 6 : (g == 2 ? 3 : 7));
        PackedHalf h[2];
#pragma unroll
        for (int e = 0; e < 2; e++) {
            const int col = e ? cb : ca, jj = col >> 1;
            const uint32_t sel = (col & 1) ? 0x07060302u : 0x05040100u; // the column's halves of two row registers
            const uint32_t p04 = perm(D[4 * 4 + jj], D[0 * 4 + jj], sel), p26 = perm(D[6 * 4 + jj], D[2 * 4 + jj], sel);
            const uint32_t p13 = perm(D[3 * 4 + jj], D[1 * 4 + jj], sel), p57 = perm(D[7 * 4 + jj], D[5 * 4 + jj], sel);
            h[e] = idct_1d_dot(p04, p26, p13, p57, 512);
        }
        ZJ_SCHED_BARRIER();
        int32_t oa[8], ob[8];
        idct_1d_dot_add(h[0], oa);
        idct_1d_dot_add(h[1], ob);
#pragma unroll
        for (int i = 0; i < 8; i++) T[i][g] = pack_sar<10>(oa[i], ob[i]);
        ZJ_SCHED_BARRIER();
    }
    // pass 2: rows (scalar.rs:170-274), bias SCALE_BITS, >> 17, clamp -> bytes
    constexpr int32_t bias2 = 512 + 65536 + (128 << 17);
#pragma unroll
    for (int r0 = 0; r0 < 8; r0 += 2) {
        PackedHalf h[2];
#pragma unroll
        for (int e = 0; e < 2; e++) h[e] = idct_1d_dot(T[r0 + e][0], T[r0 + e][1], T[r0 + e][2], T[r0 + e][3], bias2);
        ZJ_SCHED_BARRIER();
#pragma unroll
        for (int e = 0; e < 2; e++) {
            int32_t o[8];
            idct_1d_dot_add(h[e], o);
            out[2 * (r0 + e)] = sat_pk_u8_2(pack_sar<17>(o[0], o[1]), pack_sar<17>(o[2], o[3]));
            out[2 * (r0 + e) + 1] = sat_pk_u8_2(pack_sar<17>(o[4], o[5]), pack_sar<17>(o[6], o[7]));
        }
        ZJ_SCHED_BARRIER();
    }
}

// Q1: DC-only blocks take the shortcut value: i16 wrapping product, floor >> 3, + 128, NOT clamped
// (scalar.rs:48).  Returns the value replicated in both 16-bit lanes.
ZJ_DEV uint32_t dc_only_value(uint32_t w0, int32_t q0, const int clamp = 0)
{
    const int16_t dc = (int16_t)(uint16_t)((uint32_t)lo16s(w0) * (uint32_t)(int32_t)(int16_t)q0);
    int32_t v = ((int32_t)dc >> 3) + 128;
    if (clamp) v = v < 0 ? 0 : (v > 255 ? 255 : v); // ZJ_FLAG_CLAMP_DC (extension)
    return ((uint32_t)v & 0xffffu) | ((uint32_t)v << 16);
}

// ------------------------------------------------------------------------------------------------
// colour conversion of a packed pixel pair (color_convert/scalar.rs:66-85).  cb, cr are the RAW
// samples; the reference's `cb - 128`, `cr - 128` are folded into the multiply-adds, which is exact
// because every product/sum wraps mod 2^16 in the reference too (Q7):
//   45*(cr-128) == 45*cr - 5760,  11*(cb-128) + 23*(cr-128) == 11*cb + 23*cr - 4352,
//   113*(cb-128) == 113*cb - 14464                                   (mod 2^16)
// Returns R, G, B as UNCLAMPED i16 pairs.
// ------------------------------------------------------------------------------------------------
struct RGB2 { uint32_t r, g, b; };
ZJ_DEV RGB2 ycc_to_rgb_pair(uint32_t y_, uint32_t cb_, uint32_t cr_)
{
    const u16x2 y = as_u16x2(y_), cb = as_u16x2(cb_), cr = as_u16x2(cr_);
    RGB2 o;
    o.r = as_u32(y + (u16x2)sar(splat(45) * cr + splat(-5760), 5));
    // two multiply-adds: left alone the compiler forms 23*cr, 11*cb + that, + constant (three instructions; round 4 ledger)
    uint32_t gt = as_u32(splat(23) * cr + splat(-4352));
    ZJ_PIN(gt);
    o.g = as_u32(y - (u16x2)sar(splat(11) * cb + as_u16x2(gt), 5));
    o.b = as_u32(y + (u16x2)sar(splat(113) * cb + splat(-14464), 6));
    return o;
}

// triangle filter on packed pairs: (3*near + far + 2) >> 2   (upsampler/scalar.rs:35-41,124-126)
ZJ_DEV uint32_t tri(uint32_t near_, uint32_t far_)
{
    return as_u32(sar(splat(3) * as_u16x2(near_) + as_u16x2(far_) + splat(2), 2));
}
ZJ_DEV int tri1(int near_, int far_) { return (int)(int16_t)(uint16_t)(3 * near_ + far_ + 2) >> 2; }

// ------------------------------------------------------------------------------------------------
// Tile geometry per sampling mode.  A workgroup owns one "tile": the full height of one strip
// (the reference's unit of work, src/mcu.rs:225-226 / src/mcu_prog.rs:188-203) x TWY luma columns.
//   mode      strip = Y block rows x chroma block rows   (mcu.rs:139-218, mcu_prog.rs:138-144)
//   (1,1)     1 x 1      (2,1) 2 x 2 (two MCU rows)     (1,2) 2 x 1      (2,2) 4 x 2
// Horizontal modes need one extra chroma block column on each side (the triangle filter's taps);
// because the reference filters the strip as ONE flat array (Q4) the neighbour of the first/last
// column is the other end of the previous/next row, so the halo wraps around with a row shift.
// Tile width (TWC chroma block columns) is chosen so that the tile's blocks fill the workgroup's
// waves (one lane per block); the values below are measured choices, see the comments at each.
template <int HS, int VS, bool CHROMA> struct TileWidth;
// 4:2:0 -> RGB: 12*TWC + 8 blocks.  Round 1 ran TWC = 20 (248 of 256 lanes busy).  With the packed generation TWC = 16
// measures 2.5 % faster (tools/ab_libs.sh, profiles/r02_*): 256 pixels = exactly two colour rounds per wave, 4096-pixel
// rows split into whole tiles, and the blocks sort into pure waves (128 luma | 64 chroma | 8 halo).  12 and 24 are
// 10-30 % slower, 10 equals 20.
#ifndef ZJ_TWC_HV
#define ZJ_TWC_HV 16
#endif
#ifndef ZJ_NT_MIN_HV
#define ZJ_NT_MIN_HV 0 // 512-pixel tiles on 512 threads (-DZJ_TWC_HV=32 -DZJ_NT_MIN_HV=512, 3 workgroups per CU): +3.5 % (round 3)
#endif
template <> struct TileWidth<2, 2, true> { static constexpr int TWC = ZJ_TWC_HV; };
// 4:2:2 -> RGB: 8*TWC + 8 blocks.  TWC = 16 (256 pixels x 16 rows; 64 luma | 64 chroma | 8 halo blocks = pure waves, 192
// threads) is 8 % faster than round 1's 31 (tools/ab_libs_w.sh 422-rgb); 24: +5 %, 32: -14 %, 16 on 256 threads: +4 %.
#ifndef ZJ_TWC_H
#define ZJ_TWC_H 16
#endif
#ifndef ZJ_NT_MIN_H
#define ZJ_NT_MIN_H 0
#endif
template <> struct TileWidth<2, 1, true> { static constexpr int TWC = ZJ_TWC_H; };   // 8*TWC+8 blocks (31: 256, 496 px)
#ifndef ZJ_TWC_V
#define ZJ_TWC_V 32 // 256-pixel tiles on 128 threads: +2 % over 512 pixels on 256 (round 4, profiles/r04_workloads.txt); 64 is the round 1-3 shape
#endif
#ifndef ZJ_NT_MIN_V
#define ZJ_NT_MIN_V 0
#endif
template <> struct TileWidth<1, 2, true> { static constexpr int TWC = ZJ_TWC_V; };   // 4*TWC = 128 blocks, 256 px (128 threads)
// 4:4:4 with chroma: 3*TWC blocks.  Round 1 ran TWC = 84 (252 lanes busy, 672 pixels).  TWC = 64 with 256 threads is
// 15 % faster (tools/ab_libs_w.sh 444-rgb): 192 blocks on three waves (the fourth idles through the IDCT), but 512 pixels
// x 8 rows = 256 items = exactly one colour round for four waves, and 4096-pixel rows split into whole tiles (672 leaves a
// 10 % tile).  80: +3 %, 42: -2 %, 64 with 192 threads: +12 %.
#ifndef ZJ_TWC_444
#define ZJ_TWC_444 64
#endif
#ifndef ZJ_NT_MIN_444
#define ZJ_NT_MIN_444 256
#endif
template <> struct TileWidth<1, 1, true> { static constexpr int TWC = ZJ_TWC_444; };
template <> struct TileWidth<2, 2, false> { static constexpr int TWC = 32; };  // 8*TWC = 256 luma blocks
template <> struct TileWidth<2, 1, false> { static constexpr int TWC = 64; };
template <> struct TileWidth<1, 2, false> { static constexpr int TWC = 128; };
#ifndef ZJ_TWC_GRAY
#define ZJ_TWC_GRAY 128
#endif
#ifndef ZJ_NT_MIN_GRAY
#define ZJ_NT_MIN_GRAY 256 // 128 blocks = two waves of IDCT, but 512 items: four waves for the copy-out phase (+3.5 %)
#endif
template <> struct TileWidth<1, 1, false> { static constexpr int TWC = ZJ_TWC_GRAY; }; // 128 blocks, 2 waves

// OUT_RGBA (R G B 255 per pixel) and OUT_RGB_CHW (three u8 planes) are extensions beyond the reference
// (SURVEY 8f-3/4); both place every pixel at its own position (no Q5/Q6), like Params::plain does for OUT_RGB.
enum { OUT_RGB = 0, OUT_GRAY = 1, OUT_YCBCR = 2, OUT_RGBA = 3, OUT_RGB_CHW = 4 };
enum { GEN_WIDE = 0, GEN_PACKED = 1 };

// LDS layouts (byte offsets).
//   GEN_WIDE    Yp[SH][TWY] i16 | Cb[CROWS][CPITCH] i16 | Cr[..] | halo columns, raw + filtered | tables | vertical LUT
//   GEN_PACKED  Yb[SH][TWY] u8  | Cb[CROWS][CPITCH] i16 | Cr[..] | halo columns, raw + filtered | tables | vertical LUT | flag | store staging
// The packed kernel's allocation covers the wide layout too: a tile whose luma cannot be staged as bytes (an
// unclamped DC-only value outside 0..255, Q1) is redone by the wide code in the same workgroup.
template <int HS, int VS, int OUT>
struct Cfg {
    static constexpr bool CHROMA = OUT != OUT_GRAY;
    static constexpr int YBR = (HS == 2 && VS == 2) ? 4 : ((HS == 2 || VS == 2) ? 2 : 1);
    static constexpr int CBR = (HS == 2) ? 2 : 1;
    static constexpr int SH = YBR * 8;                 // luma rows per strip
    static constexpr int CROWS = CBR * 8;              // chroma rows per strip
    static constexpr int TWC = TileWidth<HS, VS, CHROMA>::TWC; // chroma block columns per tile
    static constexpr int HALO = (HS == 2) ? 1 : 0;
    static constexpr int TWYB = TWC * HS;              // luma block columns per tile
    static constexpr int TWY = TWYB * 8;               // luma pixels per tile row
    static constexpr int NYB = YBR * TWYB;             // luma blocks per tile
    static constexpr int CCOLS = TWC + 2 * HALO;       // chroma block columns incl. halo
    static constexpr int NCB = CBR * CCOLS;            // chroma blocks per tile and component
    // Chroma LDS rows hold the tile's own TWC * 8 samples and nothing else: 256 bytes per row for the 256-pixel tiles, so
    // the 16 lanes of a ds_read_b128 lane group -- which belong to two different luma rows -- hit 16 different bank quads
    // whatever chroma rows they read (rounds 1-2 kept the two halo columns inside the row, 288 bytes, and paid a 2-way
    // conflict on every such read; profiles/r03_lds_conflicts_by_phase.txt).  The halo blocks' single pixel columns live in
    // a side array [comp][side][chroma row], and -- packed generation -- vertically filtered per output row in
    // [comp][side][m] (halo_filter), where the colour phase's edge lanes pick them up.
    static constexpr int CPITCH = TWC * 8;             // i16 per chroma LDS row
    static constexpr int COFF = 0;                     // LDS column of chroma column 0
    static constexpr int YSZ = SH * TWY;               // luma samples per tile
    static constexpr int CSZ = CROWS * CPITCH;         // chroma samples per tile and component
    static constexpr int NGRP = TWY / 16;              // 16-pixel groups per tile row
    static constexpr int NITEMS = SH * NGRP;
    static constexpr int NBLK = NYB + (CHROMA ? 2 * NCB : 0);        // blocks per tile
    static constexpr int NT_BLK = (NBLK + 63) / 64 * 64;            // one lane per block
    static constexpr int NT_MIN = (HS == 1 && VS == 1) ? (CHROMA ? ZJ_NT_MIN_444 : ZJ_NT_MIN_GRAY) : ((HS == 2 && VS == 1 && CHROMA) ? ZJ_NT_MIN_H : ((HS == 2 && VS == 2 && CHROMA) ? ZJ_NT_MIN_HV : ((HS == 1 && VS == 2 && CHROMA) ? ZJ_NT_MIN_V : 0)));
    static constexpr int NT = NT_BLK > NT_MIN ? NT_BLK : NT_MIN;     // threads per workgroup
    static constexpr int NW = NT / 64;
    // Colour rounds that do not fill the workgroup (4:2:2: 256 items on 192 threads -- round 1 is one wave's worth) go to the
    // LAST wave instead of the first: the first waves carry the block transforms (~640 instructions per lane), the last one
    // the halo columns (~190), so the extra round lands where the workgroup's critical path is shortest.  ROUND_ROT = waves
    // by which the logical thread numbering is rotated per round (kernel: round_tid); 0 where the rounds are full.
    static constexpr int ROUND_ROT = (ZJ_ROUND_ROT && NITEMS % NT != 0 && NITEMS % NT <= 64 && NW > 1) ? 1 : 0;

    static constexpr int LUT_N = SH + 2;
    static constexpr int LUT_BYTES = ((2 * LUT_N * 2 + 15) / 16) * 16;
    static constexpr int HRAW = (CHROMA && HALO) ? 2 * 2 * CROWS : 0; // i16: halo pixel columns
    static constexpr int HFIL = (CHROMA && HALO) ? 2 * 2 * SH : 0;    // i16: the same per up-sampled row
    static constexpr int CBYTES = CHROMA ? (2 * CSZ + HRAW + HFIL) * 2 : 0;
    static_assert(CBYTES % 16 == 0, "tables stay 16-byte aligned");
    // interleaved outputs leave through staged stores: PPI 16-byte pieces per 16-pixel item (3 or 4 bytes per pixel)
    static constexpr bool TSCAP = OUT == OUT_RGB || OUT == OUT_YCBCR || OUT == OUT_RGBA;
    static constexpr int PPI = OUT == OUT_RGBA ? 4 : 3;
    // pieces of a wave's round that reuse the round's luma bytes (whole items only), and the staging for the rest
    static constexpr int INPL = 64 / PPI * PPI;                       // 63 | 64
    static constexpr int XSTAGE = ((64 * PPI - INPL) * 16 + 31) / 32 * 32; // 2080 | 3072 bytes per wave
    // the 8 halo blocks (one pixel column each is ever read) have the last wave to themselves: that wave works with
    // one lane per block COLUMN instead of one lane per block (halo_pass1 / halo_pass2)
    static constexpr bool HALO_PURE = CHROMA && HALO && (NYB + 2 * CBR * TWC) % 64 == 0 && 2 * CBR * 2 * 8 == 64;
    static constexpr int HALO_T0 = NYB + 2 * CBR * TWC; // first lane of that wave
    template <int GEN> struct L {
        static constexpr int YPX = GEN == GEN_PACKED ? 1 : 2;        // bytes per staged luma sample
        // Cb plane, then Cr.  Packed: a wave's round reuses the 16 luma bytes of each of its 64 items as store staging
        // (piece_addr), also for item numbers beyond NITEMS in a partly filled round: the luma area is padded to whole rounds
        static constexpr int C_OFF = GEN == GEN_PACKED ? (NITEMS + 63) / 64 * 64 * 16 : YSZ * 2;
        static constexpr int TAB_OFF = C_OFF + CBYTES;
        static constexpr int LUT_OFF = TAB_OFF + TAB_BYTES;
        static constexpr int FLAG_OFF = LUT_OFF + LUT_BYTES;         // GEN_PACKED: "redo this tile wide"
        static constexpr int X_OFF = FLAG_OFF + 16;                  // GEN_PACKED: per wave 2080 bytes of store staging (XSTAGE)
        // after X_OFF: the store staging of the 3-byte outputs, and (before the colour phase) the halo wave's scratch
        static constexpr int X_STAGE = TSCAP ? NW * XSTAGE : 0;
        static constexpr int X_HALO = HALO_PURE ? 8 * 80 * 4 : 0;
        static constexpr int BYTES = GEN == GEN_PACKED ? X_OFF + (X_STAGE > X_HALO ? X_STAGE : X_HALO) : FLAG_OFF;
        static_assert(C_OFF % 16 == 0 && TAB_OFF % 16 == 0, "16-byte alignment of the planes");
    };
    static constexpr int LDS_WIDE = L<GEN_WIDE>::BYTES;
    static constexpr int LDS_PACKED = L<GEN_PACKED>::BYTES > LDS_WIDE ? L<GEN_PACKED>::BYTES : LDS_WIDE;
    static constexpr int PIECES_PER_ROW = PPI * NGRP; // 16-byte pieces of a tile row
    // division of a piece index by PIECES_PER_ROW as multiply + shift, exact over every index color_copyout forms
    static constexpr int PPR_MAGIC = (1 << 20) / PIECES_PER_ROW + 1;
    static constexpr bool ppr_magic_ok()
    {
        for (int q = 0; q < PPI * ((NITEMS + NT - 1) / NT * NT) + 64 * PPI; q++)
            if ((int)(((unsigned)q * (unsigned)PPR_MAGIC) >> 20) != q / PIECES_PER_ROW) return false;
        return true;
    }
    static_assert(ppr_magic_ok(), "PPR_MAGIC");
    static_assert(NT <= 512 && NBLK <= NT, "one lane per block, at most 8 waves");
};

constexpr int SCATTER_MAX = 32;            // frames of one scattered launch (4 x 32 pointers = 1 KB of kernel arguments)

struct Params {
    const int16_t* y;
    const int16_t* cb;
    const int16_t* cr;
    uint8_t* out;
    long long y_frame_stride;     // i16 elements between frames
    long long c_frame_stride;
    long long out_frame_stride;   // bytes between frames
    int width, height;            // pixels
    int mcu_x;                    // MCUs per row (headers.rs:317)
    int n_strips;
    int tiles_per_row;
    int regular_px;               // ragged widths (RAG instantiations): pixels of a row made of ordinary 16-pixel groups
    int nframes;
    int zero_fill;                // 1: also write the bytes the reference leaves 0 (Q5/Q6)
    int total_tiles;
    uint32_t tpr_magic, tpr_shift; // magic_u31(tiles_per_row), magic_u31(n_strips): tile_from_id
    uint32_t ns_magic, ns_shift;
    int stagger_wgs, stagger_delay; // first-wave de-synchronisation (zj_kernels.hip: stagger_start); 0 = off
    uint32_t stagger_magic, stagger_shift; // magic_u31(CUs of the device): a workgroup's slot group = id / CUs
#if defined(ZJ_ABLATION)
    int debug;                    // diagnostic build only (tools/ablate.py): 1 skip IDCT, 2 skip colour math, 4 no loads, 8 no stores
#endif
    int plain;                    // OUT_RGB only: 1 = the last 16 samples of a row go to their own position (no Q5/Q6)
    int out_pitch;                // bytes between the starts of consecutive output rows (zj_frame_desc.out_pitch; tight = width x
                                  // components, CHW: width); what lies between a row's end and the next row is never written
    long long plane_stride;       // OUT_RGB_CHW: bytes between the R, G and B planes of a frame (out_pitch * height)
    int clamp_dc;                 // extension: DC-only shortcut value clamped to 0..255 (Q1 corrected)
    int edge_rep;                 // extension: horizontal chroma filter per row with replicated edges (Q4 corrected)
    uint32_t tab[3 * TAB_DW];     // the three quantisation tables + guard constants (build_table), by value
    // Scattered batch (zj_decode_frames_device): the frames of one launch are independent allocations, the way the
    // reference's callers own them (a fresh Vec per strip, src/mcu.rs:238-250; a Vec<u8> per decode, src/decoder.rs:178).
    // Their addresses travel by value like the tables -- nothing to stage, nothing to keep alive -- and a workgroup reads
    // the four of its frame with ONE scalar load (32 bytes, one cache line) indexed by the (uniform) frame number.  A
    // launch is scattered when its base pointer `y` is null: the flag costs the contiguous form no extra load.
    uint64_t fptr[SCATTER_MAX][4]; // per frame: y | cb | cr | out
};

// vertical schedule of upsample_vertical (upsampler/scalar.rs:84-144): pair k -> (near, far)
ZJ_DEV void vsched(int k, int& n, int& f) { n = k; f = (k == 0) ? 0 : (k < 7 ? k + 1 : 7); }

// chroma source rows (A weighted 3, B weighted 1) of up-sampled row m of a strip  (Q3)
template <int HS, int VS>
ZJ_DEV void vrows(int m, int& ra, int& rb)
{
    if (VS == 1) { ra = rb = m; return; }
    int n, f;
    if (HS == 1) { // (1,2): 8 real rows -> 16
        vsched(m >> 1, n, f);
        ra = (m & 1) ? f : n;
        rb = (m & 1) ? n : f;
    } else {       // (2,2): upsample_vertical sees 8 "rows" made of 2 real rows each
        vsched(m >> 2, n, f);
        const int half = m & 1, farw = (m >> 1) & 1;
        ra = 2 * (farw ? f : n) + half;
        rb = 2 * (farw ? n : f) + half;
    }
}

// a workgroup's tile, and where its frame lives (all workgroup-uniform: the compiler keeps them in scalar registers)
struct TileId {
    int frame, strip, tile;
    const int16_t* y; const int16_t* cb; const int16_t* cr; // the frame's planes
    uint8_t* out;                                           // the frame's pixels
};

// Division of a workgroup id by a launch constant (tiles per row, strips per frame) as multiply-high + shift with a
// host-computed multiplier: 5 scalar instructions where the compiler's expansion of `/` and `%` by a kernel argument is
// a float-reciprocal sequence of ~35 scalar + 4 vector instructions (v_rcp, v_readfirstlane) -- twice, in every workgroup,
// in front of the first load (profiles/r04_valu_ledger.txt).  Exact for every 0 <= n < 2^31 and 1 <= d < 2^30:
// with l = ceil(log2 d) and m = floor(2^(31+l) / d) + 1 the error m*d - 2^(31+l) is in (0, d], so n * error < 2^(31+l).
struct Magic { uint32_t m, s; };  // m == 0: d == 1
inline Magic magic_u31(const uint32_t d)
{
    Magic g = {0, 0};
    if (d <= 1) return g;
    uint32_t l = 0;
    while ((1ull << l) < d) l++;
    g.m = (uint32_t)((1ull << (31 + l)) / d) + 1u;
    g.s = l - 1;
    return g;
}
ZJ_DEV uint32_t magic_div(const uint32_t n, const Magic g)
{
#if defined(ZJ_EMU)
    return g.m ? (uint32_t)(((uint64_t)n * g.m) >> 32) >> g.s : n;
#else
    return g.m ? __umulhi(n, g.m) >> g.s : n;
#endif
}

ZJ_DEV TileId tile_from_id(const Params& p, const int id)
{
    TileId t;
    const Magic gt = {p.tpr_magic, p.tpr_shift}, gs = {p.ns_magic, p.ns_shift};
    const uint32_t r = magic_div((uint32_t)id, gt);
    t.tile = id - (int)r * p.tiles_per_row;
    const uint32_t f = magic_div(r, gs);
    t.strip = (int)r - (int)f * p.n_strips;
    t.frame = (int)f;
    return t;
}
// XCD-aware order: hardware sends workgroup b to XCD b % 8; give each XCD a contiguous run of `n` units
// so neighbouring tiles (which share halo blocks) meet in the same L2.
#ifndef ZJ_XCD_ORDER
#define ZJ_XCD_ORDER 1 // 0: workgroup b decodes tile b (A/B knob, round 4)
#endif
ZJ_DEV int xcd_order(const int bid, const int n) { return (ZJ_XCD_ORDER && (n & 7) == 0) ? (bid & 7) * (n >> 3) + (bid >> 3) : bid; }
ZJ_DEV TileId decode_tile(const Params& p, int bid)
{
    TileId t = tile_from_id(p, xcd_order(bid, p.total_tiles));
    // One scalar branch per workgroup; everything after it sees four scalar pointers, whichever form the launch has --
    // the per-lane code selects between t.cb and t.cr exactly as it selected between p.cb and p.cr before the table existed
    // (an index into the table by a per-lane component number would be a vector load from the kernel arguments).
    if (p.y == nullptr) {
        // (an integer turned into a pointer is a FLAT pointer to the compiler: every load and store of the kernel became a
        // flat_* instruction with a 64-bit per-lane address, +1.9 % kernel time, until the table's entries were declared
        // what they are -- addresses in global memory)
        t.y = ZJ_GLOBAL_PTR(const int16_t, p.fptr[t.frame][0]);
        t.cb = ZJ_GLOBAL_PTR(const int16_t, p.fptr[t.frame][1]);
        t.cr = ZJ_GLOBAL_PTR(const int16_t, p.fptr[t.frame][2]);
        t.out = ZJ_GLOBAL_PTR(uint8_t, p.fptr[t.frame][3]);
    } else {
        const unsigned long long f = (unsigned)t.frame;   // (never negative: spares the scalar unit the sign terms of a 32 x 64 product)
        t.y = p.y + f * (unsigned long long)p.y_frame_stride;
        t.cb = p.cb + f * (unsigned long long)p.c_frame_stride;
        t.cr = p.cr + f * (unsigned long long)p.c_frame_stride;
        t.out = p.out + f * (unsigned long long)p.out_frame_stride;
    }
    return t;
}

// ------------------------------------------------------------------------------------------------
// Block b of a tile: where its 64 coefficients live in HBM and where its pixels go in LDS.
//   b in [0, NYB): luma; then NCB Cb blocks; then NCB Cr blocks (halo columns first/last per row)
// ------------------------------------------------------------------------------------------------
// side arrays of the halo columns (Cfg::HRAW, Cfg::HFIL): comp 1 | 2, side 0 = left of the tile, 1 = right of it
template <class C, int GEN> ZJ_DEV int16_t* lds_halo_raw(char* lds, int comp, int side, int row)
{
    return reinterpret_cast<int16_t*>(lds + C::template L<GEN>::C_OFF + 2 * C::CSZ * 2) + ((comp - 1) * 2 + side) * C::CROWS + row;
}
template <class C, int GEN> ZJ_DEV int16_t* lds_halo_fil(char* lds, int comp, int side, int m)
{
    return reinterpret_cast<int16_t*>(lds + C::template L<GEN>::C_OFF + (2 * C::CSZ + C::HRAW) * 2) + ((comp - 1) * 2 + side) * C::SH + m;
}

struct BlockLoc {
    const U4* src;   // 8 x 16 bytes of coefficients
    char* dst;       // LDS address of the block's pixel (0,0) -- or of the single halo column
    int pitch;       // LDS row pitch in BYTES
    int comp;        // 0 Y, 1 Cb, 2 Cr
    int halo;        // 0: full block, 1: left halo (keep pixel column 7), 2: right halo (column 0)
    bool valid;
};

template <class C, int GEN>
ZJ_DEV BlockLoc locate(const Params& p, const TileId t, const int b, char* lds)
{
    using LL = typename C::template L<GEN>;
    BlockLoc L;
    L.valid = false; L.halo = 0; L.comp = 0; L.src = nullptr; L.dst = lds; L.pitch = C::TWY * LL::YPX;
    const int ybw = p.mcu_x * (C::TWYB / C::TWC); // luma blocks per plane row (= mcu_x * HS)
    const int cbw = p.mcu_x;                      // chroma blocks per plane row
    if (b < C::NYB) {
        const int brow = b / C::TWYB, bcol = b % C::TWYB;
        const int gcol = t.tile * C::TWYB + bcol;
        if (gcol >= ybw) return L;
        const long long blk = (long long)(t.strip * C::YBR + brow) * ybw + gcol;
        L.src = reinterpret_cast<const U4*>(t.y + blk * 64);
        L.dst = lds + ((brow * 8) * C::TWY + bcol * 8) * LL::YPX;
        L.valid = true;
        return L;
    }
    if (!C::CHROMA || b >= C::NBLK) return L;
    // chroma: the full blocks of Cb, the full blocks of Cr, then the halo blocks (both components) -- with TWC = 16
    // that fills waves 0-1 with luma, wave 2 with chroma and leaves the 8 halo blocks alone in the last wave, which
    // then runs the single-column transform (idct_block_packed_halo) as a wave-uniform branch
    const int cb_ = b - C::NYB;
    constexpr int NFULL = C::CBR * C::TWC; // full chroma blocks per component
    int comp, brow, j;
    if (cb_ < 2 * NFULL) {
        comp = cb_ < NFULL ? 1 : 2;
        const int bb = comp == 1 ? cb_ : cb_ - NFULL;
        brow = bb / C::TWC;
        j = bb % C::TWC + C::HALO;
    } else {
        const int hb = cb_ - 2 * NFULL;      // (comp, brow, side)
        comp = 1 + hb / (2 * C::CBR);
        brow = (hb / 2) % C::CBR;
        j = (hb & 1) ? C::CCOLS - 1 : 0;
    }
    const int cb0 = t.tile * C::TWC;
    const int nvalid = (cbw - cb0) < C::TWC ? (cbw - cb0) : C::TWC;
    int gcol, lcol = 0;
    if (C::HALO) {
        if (j == 0) { gcol = cb0 > 0 ? cb0 - 1 : cbw - 1; L.halo = 1; }
        else if (j == C::CCOLS - 1) { gcol = cb0 + nvalid < cbw ? cb0 + nvalid : 0; L.halo = 2; }
        else { if (j - 1 >= nvalid) return L; gcol = cb0 + j - 1; lcol = C::COFF + 8 * (j - 1); }
    } else {
        if (j >= nvalid) return L;
        gcol = cb0 + j; lcol = 8 * j;
    }
    const long long blk = (long long)(t.strip * C::CBR + brow) * cbw + gcol;
    const int16_t* plane = comp == 1 ? t.cb : t.cr;
    L.src = reinterpret_cast<const U4*>(plane + blk * 64);
    if (L.halo) { // the one pixel column of a halo block that is ever read: a column of the side array
        L.dst = reinterpret_cast<char*>(lds_halo_raw<C, GEN>(lds, comp, L.halo - 1, brow * 8));
        L.pitch = 2;
    } else {
        L.dst = lds + LL::C_OFF + ((comp - 1) * C::CSZ + (brow * 8) * C::CPITCH + lcol) * 2;
        L.pitch = C::CPITCH * 2;
    }
    L.comp = comp;
    L.valid = true;
    return L;
}

// a block's 8 pixel rows as packed i16 -> LDS (wide luma, chroma of both generations)
ZJ_DEV void store_block(const BlockLoc& L, const U4 px[8])
{
    if (L.halo == 0) {
#pragma unroll
        for (int r = 0; r < 8; r++) *reinterpret_cast<U4*>(L.dst + r * L.pitch) = px[r];
    } else {
        // only one pixel column of a halo block is ever read: its last (left) / first (right)
#pragma unroll
        for (int r = 0; r < 8; r++)
            *reinterpret_cast<int16_t*>(L.dst + r * L.pitch) = (int16_t)(L.halo == 1 ? (px[r].w >> 16) : (px[r].x & 0xffffu));
    }
}
// one splat row stored eight times (DC-only blocks): no 32-register fill for the shortcut lanes
ZJ_DEV void store_splat(const BlockLoc& L, const uint32_t v)
{
    const U4 row = {v, v, v, v};
    if (L.halo == 0) {
#pragma unroll
        for (int r = 0; r < 8; r++) *reinterpret_cast<U4*>(L.dst + r * L.pitch) = row;
    } else {
#pragma unroll
        for (int r = 0; r < 8; r++) *reinterpret_cast<int16_t*>(L.dst + r * L.pitch) = (int16_t)v;
    }
}

template <class C, int GEN> ZJ_DEV uint32_t* lds_tab(char* lds) { return reinterpret_cast<uint32_t*>(lds + C::template L<GEN>::TAB_OFF); }
template <class C, int GEN> ZJ_DEV int16_t* lds_lut(char* lds) { return reinterpret_cast<int16_t*>(lds + C::template L<GEN>::LUT_OFF); }
template <class C, int GEN> ZJ_DEV const int16_t* lds_lut(const char* lds) { return reinterpret_cast<const int16_t*>(lds + C::template L<GEN>::LUT_OFF); }
template <class C> ZJ_DEV uint32_t* lds_flag(char* lds) { return reinterpret_cast<uint32_t*>(lds + C::template L<GEN_PACKED>::FLAG_OFF); }

// ------------------------------------------------------------------------------------------------
// Phase 1 (one lane per block, every wave full whatever the component mix):
//   load_block   issues the lane's 8 x 16-byte coefficient loads                 (before the barrier
//   phase_setup  stages the three quantisation tables in LDS                      that publishes them)
//   finish_block dequantize + IDCT (or the DC-only shortcut, Q1) -> LDS planar staging
// ------------------------------------------------------------------------------------------------
ZJ_DEV void load_block(const BlockLoc& L, U4 raw[8], const int debug = 0)
{
    if (!L.valid) return;
    if (ZJ_ABL(debug, 4)) { // ablation (tools/ablate.py): no HBM reads, synthetic coefficients
        const uint32_t v = (uint32_t)(reinterpret_cast<uintptr_t>(L.src) >> 7) & 0x000f000fu;
#pragma unroll
        for (int i = 0; i < 8; i++) { raw[i].x = v + i; raw[i].y = v; raw[i].z = v >> 1; raw[i].w = 0; }
        return;
    }
#if !defined(ZJ_EMU)
    if (ZJ_NT & 2) {
#pragma unroll
        for (int i = 0; i < 8; i++) { const V4 t = __builtin_nontemporal_load(reinterpret_cast<const V4*>(L.src + i)); raw[i].x = t[0]; raw[i].y = t[1]; raw[i].z = t[2]; raw[i].w = t[3]; }
        return;
    }
#endif
#pragma unroll
    for (int i = 0; i < 8; i++) raw[i] = L.src[i];
}

// Stages the tables and, for vertically sub-sampled modes, the LDS offsets (bytes) of the two chroma rows
// (weights 3 and 1) behind every up-sampled row m = -1 .. SH (Q3): lutA[m+1], lutB[m+1].
template <class C, int HS, int VS, int GEN>
ZJ_DEV void phase_setup(const Params& p, const int tid, char* lds)
{
    for (int i = tid; i < 3 * TAB_DW; i += C::NT) lds_tab<C, GEN>(lds)[i] = p.tab[i];
    if (C::CHROMA && VS == 2) {
        for (int i = tid; i < C::LUT_N; i += C::NT) {
            int ra = 0, rb = 0;
            if (i >= 1 && i <= C::SH) vrows<HS, VS>(i - 1, ra, rb);
            lds_lut<C, GEN>(lds)[i] = (int16_t)(ra * C::CPITCH * 2);
            lds_lut<C, GEN>(lds)[C::LUT_N + i] = (int16_t)(rb * C::CPITCH * 2);
        }
    }
    if (GEN == GEN_PACKED && tid == 0) *lds_flag<C>(lds) = 0;
}

// i16 pixel rows (idct_block's output) -> bytes: values are 0..255 already
ZJ_DEV void rows_to_bytes(const U4 px[8], uint32_t b[16])
{
#pragma unroll
    for (int r = 0; r < 8; r++) {
        b[2 * r] = perm(px[r].y, px[r].x, 0x06040200u);
        b[2 * r + 1] = perm(px[r].w, px[r].z, 0x06040200u);
    }
}
// bytes -> i16 pixel rows
ZJ_DEV void bytes_to_rows(const uint32_t b[16], U4 px[8])
{
#pragma unroll
    for (int r = 0; r < 8; r++) {
        px[r].x = perm(0, b[2 * r], 0x0c010c00u); px[r].y = perm(0, b[2 * r], 0x0c030c02u);
        px[r].z = perm(0, b[2 * r + 1], 0x0c010c00u); px[r].w = perm(0, b[2 * r + 1], 0x0c030c02u);
    }
}

// ------------------------------------------------------------------------------------------------
// Halo blocks, one lane per block COLUMN (GEN_PACKED with Cfg::HALO_PURE).
// The colour phase reads ONE pixel column of a halo block (the last column of the block left of the tile, the first of
// the one right of it).  A lane per block would spend a whole wave pass (8 of 64 lanes busy) on the tile's 8 halo
// blocks -- 10 % of the tile's VALU work.  Here lane 8*h + j of the halo wave owns column j of halo block h:
//   halo_locate  where the block's coefficients are, where its pixel column goes
//   halo_load    the column's 8 coefficients (2-byte loads, in flight across the table barrier)
//   halo_pass1   dequantize, column transform (pass 1 of scalar.rs:79-167, the exact 24-bit form: no guard), results
//                to an LDS scratch (the store staging area, free until the colour phase), with the lane's "any AC" flag
//   halo_pass2   lane i of a block reads row i of the scratch, forms the one output of that row -- o0 = x0 + u3 or
//                o7 = x0 - u3 (scalar.rs:233-246), five multiply-adds per half -- or the DC-only shortcut value (Q1)
// About 110 instructions per lane instead of 515.  All 64 lanes of a wave: LDS operations execute in order, no barrier.
// Scratch per block (80 dwords): [k][j] pass-1 results (64), [64 + j] flags (8), [72] the block's DC coefficient.
// ------------------------------------------------------------------------------------------------
struct HaloLane { const int16_t* src; char* dst; int pitch; int comp; int hb; int j; bool last_col; };

template <class C>
ZJ_DEV HaloLane halo_locate(const Params& p, const TileId t, const int hl /* 0..63 */, char* lds)
{
    HaloLane H;
    H.hb = hl >> 3; H.j = hl & 7;
    H.comp = 1 + H.hb / (2 * C::CBR);
    const int brow = (H.hb / 2) % C::CBR, side = H.hb & 1;
    const int cbw = p.mcu_x, cb0 = t.tile * C::TWC;
    const int nvalid = (cbw - cb0) < C::TWC ? (cbw - cb0) : C::TWC;
    // the neighbour beyond the strip's first / last column is the other end of the row (Q4: one flat array)
    const int gcol = side == 0 ? (cb0 > 0 ? cb0 - 1 : cbw - 1) : (cb0 + nvalid < cbw ? cb0 + nvalid : 0);
    const long long blk = (long long)(t.strip * C::CBR + brow) * cbw + gcol;
    const int16_t* plane = H.comp == 1 ? t.cb : t.cr;
    H.src = plane + blk * 64 + H.j;
    H.dst = reinterpret_cast<char*>(lds_halo_raw<C, GEN_PACKED>(lds, H.comp, side, brow * 8));
    H.pitch = 2;
    H.last_col = side == 0; // left halo: the block's LAST pixel column
    return H;
}
ZJ_DEV void halo_load(const HaloLane& H, int32_t s[8])
{
#pragma unroll
    for (int k = 0; k < 8; k++) s[k] = H.src[8 * k];
}
template <class C> ZJ_DEV int32_t* lds_halo_scratch(char* lds, const int hb) { return reinterpret_cast<int32_t*>(lds + C::template L<GEN_PACKED>::X_OFF) + 80 * hb; }

template <class C>
ZJ_DEV void halo_pass1(const HaloLane& H, const int32_t s[8], char* lds)
{
    const uint16_t* q = reinterpret_cast<const uint16_t*>(lds_tab<C, GEN_PACKED>(lds) + TAB_DW * H.comp);
    int32_t* sc = lds_halo_scratch<C>(lds, H.hb);
    int32_t any = H.j == 0 ? 0 : s[0];
    int32_t d[8], o[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        if (k > 0) any |= s[k];
        d[k] = mul24(s[k], (int32_t)q[8 * k + H.j]); // dequantize (scalar.rs:308)
    }
    idct_1d(d, 512, o);
#pragma unroll
    for (int k = 0; k < 8; k++) sc[8 * k + H.j] = o[k] >> 10;
    sc[64 + H.j] = any;
    if (H.j == 0) sc[72] = s[0];
}

template <class C>
ZJ_DEV void halo_pass2(const HaloLane& H, char* lds, const int clamp_dc)
{
    const int32_t* sc = lds_halo_scratch<C>(lds, H.hb);
    const int i = H.j; // this lane finishes row i of its block
    const U4 f0 = *reinterpret_cast<const U4*>(sc + 64), f1 = *reinterpret_cast<const U4*>(sc + 68);
    const uint32_t any = f0.x | f0.y | f0.z | f0.w | f1.x | f1.y | f1.z | f1.w;
    int32_t v;
    if (any == 0) { // DC-only block (scalar.rs:45-74): the shortcut value, not clamped (Q1)
        const uint16_t* q = reinterpret_cast<const uint16_t*>(lds_tab<C, GEN_PACKED>(lds) + TAB_DW * H.comp);
        v = (int32_t)(int16_t)(dc_only_value((uint32_t)sc[72], (int32_t)q[0], clamp_dc) & 0xffffu);
    } else {
        const U4 a = *reinterpret_cast<const U4*>(sc + 8 * i), b = *reinterpret_cast<const U4*>(sc + 8 * i + 4);
        const int32_t t0 = (int32_t)a.x, t1 = (int32_t)a.y, t2 = (int32_t)a.z, t3 = (int32_t)a.w;
        const int32_t t4 = (int32_t)b.x, t5 = (int32_t)b.y, t6 = (int32_t)b.z, t7 = (int32_t)b.w;
        constexpr int32_t bias2 = 512 + 65536 + (128 << 17);
        // x0 = fsh(t0 + t4) + (t2 + t6) * 2217 + t2 * 3135 + bias; u3 = the first row of the odd 4x4 matrix (idct_1d_mul)
        int32_t x0 = mad24(t2, 2217 + 3135, wadd(wshl(wadd(t0, t4), 12), bias2));
        x0 = mad24(t6, 2217, x0);
        int32_t u3 = mul24(t1, 6149 + 4816 - 3685 - 1597);
        u3 = mad24(t7, 4816 - 3685, u3);
        u3 = mad24(t5, 4816 - 1597, u3);
        u3 = mad24(t3, 4816, u3);
        v = (H.last_col ? wsub(x0, u3) : wadd(x0, u3)) >> 17;
        v = v < 0 ? 0 : (v > 255 ? 255 : v);
    }
    *reinterpret_cast<int16_t*>(H.dst + i * H.pitch) = (int16_t)v;
}

// The halo sample beside row m of a tile (`side` 0: left of its first chroma sample, 1: right of its last): the chroma
// rows the vertical schedule gives row m (Q3) -- taken one up-sampled row up (left) / down (right) where the flat-array
// neighbour wraps to the other end of the strip (Q4; `wrap`: the tile is the first / last of its row).
template <class C, int HS, int VS, int GEN>
ZJ_DEV int halo_value(char* lds, const int comp, const int side, const int m, const bool wrap)
{
    const int16_t* raw = lds_halo_raw<C, GEN>(lds, comp, side, 0);
    if (VS == 2) {
        const int16_t* lut = lds_lut<C, GEN>(lds);
        const int i = m + 1 + (wrap ? (side == 0 ? -1 : 1) : 0);
        const int ra = lut[i] / (C::CPITCH * 2), rb = lut[C::LUT_N + i] / (C::CPITCH * 2);
        return tri1(raw[ra], raw[rb]);
    }
    const int r = side == 0 ? (wrap && m > 0 ? m - 1 : m) : (wrap && m < C::SH - 1 ? m + 1 : m);
    return raw[r];
}
// Packed generation, after halo_pass2 (same wave, LDS in order): every halo column filtered for every up-sampled row of
// the strip, 2 * 2 * SH values, so that the colour phase's edge lanes read one finished sample per side.
template <class C, int HS, int VS>
ZJ_DEV void halo_filter(const Params& p, const TileId t, const int hl, char* lds)
{
    const int cbw = p.mcu_x, cb0 = t.tile * C::TWC;
    const bool left_wrap = cb0 == 0, right_wrap = cb0 + C::TWC >= cbw;
    for (int q = hl; q < C::HFIL; q += 64) {
        const int comp = 1 + q / (2 * C::SH), side = (q / C::SH) & 1, m = q % C::SH;
        *lds_halo_fil<C, GEN_PACKED>(lds, comp, side, m) = (int16_t)halo_value<C, HS, VS, GEN_PACKED>(lds, comp, side, m, side == 0 ? left_wrap : right_wrap);
    }
}

// GEN_WIDE: the round-1 form.  GEN_PACKED: luma leaves as bytes, chroma as i16; NEED_Y16 = the output does
// arithmetic on luma (RGB family), so an unclamped DC-only luma value outside 0..255 (Q1) cannot be staged as
// a byte: the lane raises the tile's flag and the workgroup redoes the tile with the wide code.  (Gray and
// YCbCr outputs truncate luma to its low byte anyway, Q7.)
template <class C, int GEN, bool NEED_Y16>
ZJ_DEV void finish_block(const BlockLoc& L, const U4 raw[8], char* lds, const int debug = 0, const int clamp_dc = 0)
{
    if (!L.valid) return;
    const uint32_t* w = reinterpret_cast<const uint32_t*>(raw);
    const uint32_t* tab = lds_tab<C, GEN>(lds) + TAB_DW * L.comp;
    if (GEN == GEN_WIDE) {
        uint32_t any = w[0] & 0xffff0000u; // DC-only test (scalar.rs:45): all but coefficient 0 are zero
#pragma unroll
        for (int i = 1; i < 32; i++) any |= w[i];
        if (any != 0 && !ZJ_ABL(debug, 1)) {
            U4 px[8];
            idct_block(raw, reinterpret_cast<const uint16_t*>(tab), px);
            store_block(L, px);
        } else {
            store_splat(L, dc_only_value(w[0], (int32_t)(tab[0] & 0xffffu), clamp_dc));
        }
        return;
    }
    // With the halo blocks in a wave of their own (HALO_PURE) every block that comes here is a full one, and where a luma
    // row of bytes is as long as a chroma row of i16 (the 256-pixel tiles of the horizontally sub-sampled modes) the LDS
    // pitch is ONE compile-time constant: the eight row stores take immediate offsets instead of 64-bit multiply-adds on a
    // per-lane pitch (round 4 ledger: 4 v_mad_u64_u32 + 3 v_lshl_add per block store)
    using LLp = typename C::template L<GEN_PACKED>;
    constexpr bool CONST_PITCH = C::HALO_PURE && C::TWY * LLp::YPX == C::CPITCH * 2;
    BlockLoc Lc = L;
    if (CONST_PITCH) { Lc.pitch = C::CPITCH * 2; Lc.halo = 0; }
    int cls = classify_block(w, tab + 32);
    if (ZJ_ABL(debug, 1)) cls = 0;
    if (cls == 0) {
        const uint32_t v = dc_only_value(w[0], (int32_t)(tab[0] & 0xffffu), clamp_dc);
        if (L.comp != 0) { store_splat(Lc, v); return; }
        if (NEED_Y16 && (v & 0xffffu) > 255u) *lds_flag<C>(lds) = 1; // benign race: every writer stores 1
        const uint32_t b = (v & 0xffu) * 0x01010101u;
        const U2 row = {b, b};
#pragma unroll
        for (int r = 0; r < 8; r++) *reinterpret_cast<U2*>(Lc.dst + r * Lc.pitch) = row;
        return;
    }
    // the block's pixels (16 dwords of bytes) to LDS: luma rows of bytes, chroma rows of i16
    auto emit = [&](const uint32_t* b) {
        if (L.comp == 0) {
#pragma unroll
            for (int r = 0; r < 8; r++) { const U2 row = {b[2 * r], b[2 * r + 1]}; *reinterpret_cast<U2*>(Lc.dst + r * Lc.pitch) = row; }
        } else {
            U4 px[8];
            bytes_to_rows(b, px);
            store_block(Lc, px);
        }
    };
    if (cls == 1) {
        uint32_t b[16];
        idct_block_packed(raw, tab, b);
        emit(b);
        return;
    }
    {
        ZJ_NO_IF_CONVERT();
        uint32_t b[16];
        U4 px[8];
        idct_block(raw, reinterpret_cast<const uint16_t*>(tab), px);
        rows_to_bytes(px, b);
        emit(b);
    }
}

// ------------------------------------------------------------------------------------------------
// Phase 2: up-sample + colour-convert + store.  One item = 16 consecutive pixels of one row.
// ------------------------------------------------------------------------------------------------
ZJ_DEV void store16(uint8_t* p, const U4& v, const bool staged = false)
{
#if !defined(ZJ_EMU)
    if ((ZJ_NT & 1) || ((ZJ_NT & 4) && staged)) { const V4 t = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(t, reinterpret_cast<V4*>(p)); return; }
#endif
#if defined(ZJ_EMU)
    __builtin_memcpy(p, &v, 16); // rows of a ragged width start at any byte (the GPU's global stores take any alignment)
#else
    *reinterpret_cast<U4*>(p) = v;
#endif
}

// A staged 16-byte store whose cache policy is chosen per lane: write-back for a piece of a line shared with another
// tile, non-temporal otherwise.  Two instructions under complementary exec masks -- written as two C++ stores the compiler
// merges the arms into ONE store without the nt bit.
ZJ_DEV void store16_seam(uint8_t* p, const U4& v, const bool shared)
{
#if defined(ZJ_EMU)
    (void)shared;
    __builtin_memcpy(p, &v, 16);
#else
    const V4 t = {v.x, v.y, v.z, v.w};
    if (shared) asm volatile("global_store_dwordx4 %0, %1, off" : : "v"(p), "v"(t) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off nt" : : "v"(p), "v"(t) : "memory");
#endif
}

// 4 pixels -> 12 bytes from UNCLAMPED i16 pairs.  EO arrangement: (e) holds px 0,2  (o) px 1,3.
ZJ_DEV void pack_rgb4_eo(const RGB2& e, const RGB2& o, uint32_t& d0, uint32_t& d1, uint32_t& d2)
{
    const uint32_t x = sat_pk_u8_2(e.r, e.g); // R0 R2 G0 G2
    const uint32_t y = sat_pk_u8_2(e.b, o.r); // B0 B2 R1 R3
    const uint32_t z = sat_pk_u8_2(o.g, o.b); // G1 G3 B1 B3
    d0 = perm(y, x, 0x06040200u);                               // R0 G0 B0 R1
    d1 = perm(x, z, 0x07050200u);                               // G1 B1 R2 G2
    d2 = perm(z, y, 0x07050301u);                               // B2 R3 G3 B3
}
// natural arrangement: (a) holds px 0,1  (b) holds px 2,3
ZJ_DEV void pack_rgb4_nat(const RGB2& a, const RGB2& b, uint32_t& d0, uint32_t& d1, uint32_t& d2)
{
    const uint32_t x = sat_pk_u8_2(a.r, a.g); // R0 R1 G0 G1
    const uint32_t y = sat_pk_u8_2(a.b, b.r); // B0 B1 R2 R3
    const uint32_t z = sat_pk_u8_2(b.g, b.b); // G2 G3 B2 B3
    d0 = perm(y, x, 0x01040200u);                               // R0 G0 B0 R1
    const uint32_t m = perm(y, x, 0x00060503u);                 // G1 B1 R2 .
    d1 = perm(z, m, 0x04020100u);                               // G1 B1 R2 G2
    d2 = perm(z, y, 0x07050306u);                               // B2 R3 G3 B3
}
// 4 pixels -> 4 x (R G B 255).  p holds pixels (A, B), q holds (C, D); returns them in that order.
ZJ_DEV void pack_rgba4(const RGB2& p, const RGB2& q, uint32_t& a, uint32_t& b, uint32_t& c, uint32_t& d)
{
    const uint32_t x = sat_pk_u8_2(p.r, p.g); // Ra Rb Ga Gb
    const uint32_t y = sat_pk_u8_2(p.b, q.b); // Ba Bb Bc Bd
    const uint32_t z = sat_pk_u8_2(q.r, q.g); // Rc Rd Gc Gd
    a = perm(y, x, 0x0d040200u);              // selector 0x0d = constant 0xff
    b = perm(y, x, 0x0d050301u);
    c = perm(y, z, 0x0d060200u);
    d = perm(y, z, 0x0d070301u);
}
// one colour plane of 4 consecutive pixels; EO: p holds px (0, 2), q holds px (1, 3); else p (0, 1), q (2, 3)
template <bool EO> ZJ_DEV uint32_t pack_plane4(uint32_t p, uint32_t q)
{
    const uint32_t x = sat_pk_u8_2(p, q);
    return EO ? perm(x, x, 0x03010200u) : x;
}
// truncating variant (`as u8`, ycbcr_to_ycbcr, color_convert/scalar.rs:119-169): low bytes
ZJ_DEV RGB2 trunc3(uint32_t a, uint32_t b, uint32_t c)
{
    RGB2 o; o.r = a & 0x00ff00ffu; o.g = b & 0x00ff00ffu; o.b = c & 0x00ff00ffu; return o;
}

// Generic-width store: `ndw` dwords to orow[off ..), keeping only bytes below clip_end and outside
// [skip_lo, skip_hi) (the bytes the early-written RGB tail owns, Q5).  Dword stores when aligned.
ZJ_DEV void store_clip(uint8_t* orow, long long off, const uint32_t* w, int ndw, long long clip_end,
                       long long skip_lo, long long skip_hi)
{
    const bool aligned = ((reinterpret_cast<uintptr_t>(orow) + (uintptr_t)off) & 3) == 0;
    for (int i = 0; i < ndw; i++) {
        const long long o = off + 4 * i;
        const bool whole = o + 4 <= clip_end && (o + 4 <= skip_lo || o >= skip_hi);
        if (whole && aligned) {
            *reinterpret_cast<uint32_t*>(orow + o) = w[i];
        } else {
            for (int b = 0; b < 4; b++) {
                const long long ob = o + b;
                if (ob < clip_end && (ob < skip_lo || ob >= skip_hi)) orow[ob] = (uint8_t)(w[i] >> (8 * b));
            }
        }
    }
}

// One 8-pixel unit `u` of a row's 3-byte interleaved output (24 bytes in w6) by the reference's rules for ANY width
// (worker.rs:143-251 in full); `P` is the padded row length:
//   width < 16, YCbCr, plain : natural layout, clipped at 3W                      (:176-198)
//   u < 2*elements           : main groups at 24u                                  (:201-214)
//   the last two units       : the "last 16 samples", written at p' (Q5)           (:221-246)
//   units in between         : never converted (P % 16 == 8)
// main bytes inside [p', p'+48) belong to the tail, so every byte of the row has exactly one writer.
template <int OUT>
ZJ_DEV void store_unit_generic(const Params& p, uint8_t* orow, const int P, const int u, const uint32_t* w6)
{
    const int W = p.width;
    const long long stride = 3ll * W;
    const int units = P >> 3;
    if (u >= units) return;
    long long elems = P / 16 - 1; if (elems < 0) elems = 0;
    const long long position = 48 * elems;
    long long diff = 64 - (stride - position); if (diff < 0) diff = 0;
    const long long pp = position > diff ? position - diff : 0; // p'
    if (OUT == OUT_YCBCR || W < 16 || p.plain) {
        store_clip(orow, 24ll * u, w6, 6, stride, 0, 0);
    } else if (u >= units - 2) {
        const long long off = pp + 24ll * (u - (units - 2));
        store_clip(orow, off, w6, 6, off + 24, 0, 0);
        if (u == units - 1 && p.zero_fill) { // bytes the reference never writes (Q6)
            const long long z0 = position > pp + 48 ? position : pp + 48;
            for (long long o = z0; o < stride; o++) orow[o] = 0;
        }
    } else if (u < 2 * elems) {
        store_clip(orow, 24ll * u, w6, 6, stride, pp, pp + 48);
    }
}

// What a lane hands to the staged-store half of a round (TS): its item's 48 output bytes and where they go.
//   kind 0: nothing (no item)   1: pieces PPI*L ... PPI*L + PPI-1   2: (RGB) as 1 without the third piece
//        3: the row's last group under the early-tail quirk (Q5): pieces 3L-1, 3L, 3L+1, then zeros in 3L+2 (Q6)
struct ItemOut { U4 s0, s1, s2, s3; int kind; };

// The LOGICAL thread number a hardware thread plays in colour round `round` (see Cfg::ROUND_ROT): item = logical tid +
// round * NT everywhere in the colour phase; the staging area of a wave is the HARDWARE wave's (two hardware waves may be
// copying out different rounds at the same time: no barrier separates the rounds).
template <class C> ZJ_DEV int round_tid(const int tid, const int round)
{
    if (C::ROUND_ROT == 0) return tid;
    const int t = tid + 64 * C::ROUND_ROT * round;
    return t >= C::NT ? t - C::NT * (t / C::NT) : t;
}

// LDS address of piece q (0 .. 64*PPI-1) of a wave's round.  The first INPL pieces (whole items: lanes 0..20 for the
// 48-byte items, 0..15 for the 64-byte ones) reuse the luma bytes the wave's 64 items have just consumed (16 bytes each,
// contiguous because consecutive items are consecutive 16-pixel groups); the others live in the wave's XSTAGE bytes of
// staging.  The split falls on an item boundary, so a lane's pieces are contiguous and only ONE address per lane needs a
// select.
template <class C>
ZJ_DEV char* piece_addr(char* lds, const int item0, const int wave, const int q)
{
    using LL = typename C::template L<GEN_PACKED>;
    char* const ybase = lds + 16 * item0;
    char* const xbase = lds + LL::X_OFF + C::XSTAGE * wave - 16 * C::INPL;
    return (q < C::INPL ? ybase : xbase) + 16 * q;
}

// left / right neighbour pairs of a lane's eight chroma samples (phase_color, HS == 2), see there
// `pl`, `nr`: what the first / last lane of a full row gets instead (the halo samples, as aligned pairs): a DPP shift leaves
// the destination's old value in the lane that has no source, so the row's two end lanes need no select
template <class C, int VS>
ZJ_DEV void nb_pair(const uint32_t vm[4], const char* cp, const int oa, const int ob, const int lc, const int g, uint32_t& prev, uint32_t& next,
                    const uint32_t pl = 0, const uint32_t nr = 0)
{
#if !defined(ZJ_EMU)
    if (C::NGRP == 16) {
        prev = (uint32_t)__builtin_amdgcn_update_dpp((int)pl, (int)vm[3], 0x111, 0xf, 0xf, false); // row_shr:1: lane g-1's (v7, v8)
        next = (uint32_t)__builtin_amdgcn_update_dpp((int)nr, (int)vm[0], 0x101, 0xf, 0xf, false); // row_shl:1: lane g+1's (v1, v2)
        return;
    }
#endif
    // the emulation (lanes run one after the other), and tile widths whose rows are not one DPP row: the same samples
    // from LDS, wherever they exist (the ends of the row are the caller's)
    prev = next = 0;
    if (g > 0) {
        prev = *reinterpret_cast<const uint32_t*>(cp + oa + lc - 4);
        if (VS == 2) prev = tri(prev, *reinterpret_cast<const uint32_t*>(cp + ob + lc - 4));
    }
    if (g < C::NGRP - 1) {
        next = *reinterpret_cast<const uint32_t*>(cp + oa + lc + 16);
        if (VS == 2) next = tri(next, *reinterpret_cast<const uint32_t*>(cp + ob + lc + 16));
    }
}

template <class C>
ZJ_DEV void stage_item(const ItemOut& io, const int tid /* logical */, char* lds, const int round, const int hw_wave = -1)
{
    if (io.kind == 0) return;
    const int lwave = tid >> 6, lane = tid & 63;
    const int item0 = 64 * lwave + round * C::NT;
    const int wave = hw_wave < 0 ? lwave : hw_wave; // whose staging bytes
    if (C::PPI == 3 && io.kind == 3) { // rare: one lane per row of the tile that holds the row's end
        const U4 z = {0, 0, 0, 0};
        *reinterpret_cast<U4*>(piece_addr<C>(lds, item0, wave, 3 * lane - 1)) = io.s0; // launcher: lane > 0 here
        *reinterpret_cast<U4*>(piece_addr<C>(lds, item0, wave, 3 * lane)) = io.s1;
        *reinterpret_cast<U4*>(piece_addr<C>(lds, item0, wave, 3 * lane + 1)) = io.s2;
        *reinterpret_cast<U4*>(piece_addr<C>(lds, item0, wave, 3 * lane + 2)) = z;
        return;
    }
    U4* const dst = reinterpret_cast<U4*>(piece_addr<C>(lds, item0, wave, C::PPI * lane)); // PPI * 16 contiguous bytes
    dst[0] = io.s0;
    dst[1] = io.s1;
    if (C::PPI == 4) { dst[2] = io.s2; dst[3] = io.s3; }
    else if (io.kind == 1) dst[2] = io.s2;
}

// TS (staged stores, GEN_PACKED, FAST RGB / YCbCr only): processes ONE round (item = tid + round * NT) and, instead
// of storing its 48 bytes at a 48-byte lane stride, returns them in *io for stage_item; color_copyout then stores
// 16-byte pieces that are contiguous across the lanes of a wave.
// RAG (ragged width, FAST instantiations only): the row is irregular only at its end -- the last two 8-pixel units are
// written early (Q5), what lies between them and the padded width is never converted, everything is clipped at 3W
// (worker.rs:143-251).  16-pixel groups below Params::regular_px are ordinary ones (48 bytes at 48 * G, all inside the row):
// they take the fast stores, from rows that may start at any byte; the few groups beyond take the generic store path, by
// the lanes that hold them, in the same workgroup.
template <class C, int HS, int VS, int OUT, int GEN, bool FAST = true, bool TS = false, bool RAG = false>
ZJ_DEV void phase_color(const Params& p, const TileId t, const int tid, char* lds, const int round = 0, ItemOut* io = nullptr)
{
    using LL = typename C::template L<GEN>;
    const int P = p.mcu_x * 8 * HS;       // padded row length == luma width_stride (headers.rs:338)
    const int W = p.width;
    const int cbw = p.mcu_x;
    const int cb0 = t.tile * C::TWC;
    const int nvalid = (cbw - cb0) < C::TWC ? (cbw - cb0) : C::TWC;
    const bool left_wrap = cb0 == 0, right_wrap = cb0 + C::TWC >= cbw;
    const bool edge_tile = HS == 2 && (left_wrap || right_wrap); // workgroup-uniform
    const int x0 = t.tile * C::TWY;
    const long long row_bytes = p.out_pitch; // between rows (tight: W * ncomp; CHW: W, one plane's row); clip ends stay W-based
    uint8_t* const frame_out = t.out;
    const int elements = P / 16 - 1; // worker.rs:171 (P >= 32 on this path)
    if (TS) io->kind = 0;

    for (int item = TS ? tid + round * C::NT : tid; item < C::NITEMS; item += TS ? C::NITEMS : C::NT) {
        const int m = item / C::NGRP, g = item % C::NGRP;
        const int px0 = x0 + 16 * g;      // first pixel of the group in the padded row
        const int row = t.strip * C::SH + m;
        if (px0 >= P || row >= p.height) continue;
        const bool irregular = RAG && px0 >= p.regular_px; // a group at the row's end: generic stores (RAG only)
        uint8_t* const orow = frame_out + (long long)row * row_bytes;
        // ---- luma: yp[] = packed i16 pairs in the arrangement the chroma code produces -----------------
        //   HS == 2: [0..3] = (Y[4k], Y[4k+2]), [4..7] = (Y[4k+1], Y[4k+3]);  HS == 1: natural pairs (Y[2k], Y[2k+1])
        uint32_t yp[8];
        if (GEN == GEN_PACKED) {
            const U4 yv = *reinterpret_cast<const U4*>(lds + 16 * item);
            if (OUT == OUT_GRAY) {
                // ycbcr_to_grayscale (color_convert/scalar.rs:91-114): `as u8` truncation (Q7) == the staged byte
                if (FAST && !irregular) store16(orow + px0, yv, true); // lane-contiguous whole lines: streaming stores (+3.7 %)
                else { const uint32_t ow4[4] = {yv.x, yv.y, yv.z, yv.w}; store_clip(orow, px0, ow4, 4, W, 0, 0); }
                continue;
            }
            const uint32_t yb[4] = {yv.x, yv.y, yv.z, yv.w};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (HS == 2) { yp[k] = perm(0, yb[k], 0x0c020c00u); yp[4 + k] = perm(0, yb[k], 0x0c030c01u); }
                else { yp[2 * k] = perm(0, yb[k], 0x0c010c00u); yp[2 * k + 1] = perm(0, yb[k], 0x0c030c02u); }
            }
        } else {
            const U4* yrow = reinterpret_cast<const U4*>(lds + 32 * item);
            const U4 ya = yrow[0], yb = yrow[1];
            const uint32_t yw[8] = {ya.x, ya.y, ya.z, ya.w, yb.x, yb.y, yb.z, yb.w};
            if (OUT == OUT_GRAY) {
                U4 o;
                o.x = perm(yw[1], yw[0], 0x06040200u); o.y = perm(yw[3], yw[2], 0x06040200u);
                o.z = perm(yw[5], yw[4], 0x06040200u); o.w = perm(yw[7], yw[6], 0x06040200u);
                if (FAST && !irregular) store16(orow + px0, o);
                else { const uint32_t ow4[4] = {o.x, o.y, o.z, o.w}; store_clip(orow, px0, ow4, 4, W, 0, 0); }
                continue;
            }
            if (HS == 2) {
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    yp[k] = perm(yw[2 * k + 1], yw[2 * k], 0x05040100u);     // (Y[4k],   Y[4k+2])
                    yp[4 + k] = perm(yw[2 * k + 1], yw[2 * k], 0x07060302u); // (Y[4k+1], Y[4k+3])
                }
            } else {
#pragma unroll
                for (int k = 0; k < 8; k++) yp[k] = yw[k];
            }
        }

        // ---- chroma for the 16 pixels (raw samples), packed pairs ---------------------------------
        uint32_t cbp[8], crp[8]; // HS==2: [0..3] = E_k (px 4k, 4k+2), [4..7] = O_k (px 4k+1, 4k+3)
                                 // HS==1: natural pairs (px 2k, 2k+1)
        // LDS offsets (bytes) of the chroma rows behind up-sampled row m, and behind rows
        // m-1 / m+1 where the flat-array neighbour wraps to the other end of the strip (Q4)
        // Only the first / last tile of a row (a workgroup-uniform property) holds the strip's first / last chroma
        // column: every other tile skips the wrap logic and the patches below on a scalar branch.
        bool first = false, last = false;
        int oa, ob;
        if (VS == 2) {
            const int16_t* lut = lds_lut<C, GEN>(lds);
            oa = lut[m + 1]; ob = lut[C::LUT_N + m + 1];
        } else {
            oa = ob = m * C::CPITCH * 2;
        }
        if (edge_tile) {
            ZJ_NO_IF_CONVERT();
            first = (g == 0) && left_wrap;                  // chroma column 0 of the strip
            last = (8 * g + 8 == 8 * nvalid) && right_wrap; // last chroma column
        }
        // ZJ_FLAG_EDGE_REPLICATE (extension): the neighbour beyond a row's end is the end sample itself, in every row,
        // so neither the wrap to the previous / next row nor the strip-end special cases apply
        const bool rep_first = first && p.edge_rep, rep_last = last && p.edge_rep;
        const bool no_left = first && m == 0 && !p.edge_rep, no_right = last && m == C::SH - 1 && !p.edge_rep;
#pragma unroll
        for (int ch = 0; ch < 2; ch++) {
            const char* cp = lds + LL::C_OFF + ch * C::CSZ * 2;
            uint32_t* dst = ch ? crp : cbp;
            if (ZJ_ABL(ZJ_PDBG(p), 16)) { // ablation: no chroma reads from LDS, no filters (output is wrong)
#pragma unroll
                for (int k = 0; k < 8; k++) dst[k] = yp[k] + ch;
                continue;
            }
            if (HS == 1) {
                const U4* A = reinterpret_cast<const U4*>(cp + oa + 32 * g);
                const U4 a0 = A[0], a1 = A[1];
                uint32_t v[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
                if (VS == 2) {
                    const U4* B = reinterpret_cast<const U4*>(cp + ob + 32 * g);
                    const U4 b0 = B[0], b1 = B[1];
                    const uint32_t f[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
                    for (int k = 0; k < 8; k++) v[k] = tri(v[k], f[k]);
                }
#pragma unroll
                for (int k = 0; k < 8; k++) dst[k] = v[k];
            } else {
                const int lc = (C::COFF + 8 * g) * 2; // LDS byte column of this group's first chroma sample
                const U4 a = *reinterpret_cast<const U4*>(cp + oa + lc);
                uint32_t vm[4] = {a.x, a.y, a.z, a.w}; // (v1,v2) (v3,v4) (v5,v6) (v7,v8)
                // the neighbours come as aligned pairs: (x, v0) left of the group, (v9, x) right of it
                if (VS == 2) {
                    const U4 b = *reinterpret_cast<const U4*>(cp + ob + lc);
                    vm[0] = tri(vm[0], b.x); vm[1] = tri(vm[1], b.y);
                    vm[2] = tri(vm[2], b.z); vm[3] = tri(vm[3], b.w);
                }
                // The neighbours come as aligned pairs: (x, v0) left of the group, (v9, x) right of it.  Inside the tile's
                // row they are the neighbouring LANES' filtered samples -- a row of 16 groups is one DPP row of 16 lanes:
                // two v_mov_dpp where rounds 1-2 had four bank-conflicted 4-byte LDS reads and two more filters -- and at
                // the two ends of the row the halo columns (finished by halo_filter in the packed generation).
                uint32_t prev, next;
                {
                    int hvl, hvr;
                    if (GEN == GEN_PACKED && C::HALO_PURE) {
                        hvl = *lds_halo_fil<C, GEN>(lds, ch + 1, 0, m);
                        hvr = *lds_halo_fil<C, GEN>(lds, ch + 1, 1, m);
                    } else {
                        hvl = halo_value<C, HS, VS, GEN>(lds, ch + 1, 0, m, left_wrap);
                        hvr = halo_value<C, HS, VS, GEN>(lds, ch + 1, 1, m, right_wrap);
                    }
                    const uint32_t pl = (uint32_t)hvl << 16, nr = (uint32_t)hvr & 0xffffu;
                    nb_pair<C, VS>(vm, cp, oa, ob, lc, g, prev, next, pl, nr);
#if defined(ZJ_EMU)
                    constexpr bool DPP_ENDS = false;
#else
                    constexpr bool DPP_ENDS = C::NGRP == 16; // the DPP shifts have put pl / nr into lanes 0 / 15 of the row
#endif
                    if (!DPP_ENDS) {
                        if (g == 0) prev = pl;
                        if (g == nvalid - 1) next = nr;
                    } else if (nvalid != C::NGRP) { // a row's last, narrower tile (workgroup-uniform): its end is not lane 15
                        ZJ_NO_IF_CONVERT();
                        if (g == nvalid - 1) next = nr;
                    }
                }
                if (p.edge_rep) { // uniform branch (kernel argument): the default path pays no select for it
                    ZJ_NO_IF_CONVERT();
                    if (rep_first) prev = vm[0] << 16; // (x, v0) with v0 := v1
                    if (rep_last) next = vm[3] >> 16;  // (v9, x) with v9 := v8
                }
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const uint32_t L = align16(vm[k], k == 0 ? prev : vm[k - 1]); // (v_{2k},   v_{2k+1})
                    const uint32_t R = align16(k == 3 ? next : vm[k + 1], vm[k]); // (v_{2k+2}, v_{2k+3})
                    const u16x2 t3 = splat(3) * as_u16x2(vm[k]) + splat(2);        // shared 3*near + 2
                    dst[k] = as_u32(sar(t3 + as_u16x2(L), 2));     // even outputs: px 4k, 4k+2
                    dst[4 + k] = as_u32(sar(t3 + as_u16x2(R), 2)); // odd outputs:  px 4k+1, 4k+3
                }
                // the three unfiltered / mis-weighted samples of a strip (upsampler/scalar.rs:13,55,57)
                if (edge_tile) {
                    ZJ_NO_IF_CONVERT();
                    if (no_left) dst[0] = (dst[0] & 0xffff0000u) | (vm[0] & 0xffffu);       // out[0] = in[0]
                    if (no_right) {
                        const uint32_t o7 = dst[7];
                        // out[2n-2] = (3*in[n-2] + in[n-1] + 2) >> 2  == the odd output of column n-2
                        dst[3] = (dst[3] & 0x0000ffffu) | (o7 << 16);                        // px 14 <- O(px 13)
                        dst[7] = (o7 & 0x0000ffffu) | (vm[3] & 0xffff0000u);                 // px 15 = in[n-1]
                    }
                }
            }
        }

        uint32_t d[12];
        RGB2 c[8];
        ZJ_COLOR_SB();
#pragma unroll
        for (int k = 0; k < 8; k++)
            c[k] = (OUT != OUT_YCBCR && !ZJ_ABL(ZJ_PDBG(p), 2)) ? ycc_to_rgb_pair(yp[k], cbp[k], crp[k]) : trunc3(yp[k], cbp[k], crp[k]);
        if (OUT == OUT_RGBA) {
            // extension: 16 pixels -> 64 bytes at their own position, any width clipped at 4W
            uint32_t q[16];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (HS == 2) pack_rgba4(c[k], c[4 + k], q[4 * k], q[4 * k + 2], q[4 * k + 1], q[4 * k + 3]);
                else pack_rgba4(c[2 * k], c[2 * k + 1], q[4 * k], q[4 * k + 1], q[4 * k + 2], q[4 * k + 3]);
            }
            if (TS) { // (RAG: an irregular group is staged like any other; color_copyout stores its units clipped)
                io->s0 = U4{q[0], q[1], q[2], q[3]}; io->s1 = U4{q[4], q[5], q[6], q[7]};
                io->s2 = U4{q[8], q[9], q[10], q[11]}; io->s3 = U4{q[12], q[13], q[14], q[15]};
                io->kind = 1;
            } else if (FAST && !irregular) {
                uint8_t* o = orow + 4ll * px0;
#pragma unroll
                for (int k = 0; k < 4; k++) { const U4 v = {q[4 * k], q[4 * k + 1], q[4 * k + 2], q[4 * k + 3]}; store16(o + 16 * k, v); }
            } else {
#pragma unroll
                for (int h = 0; h < 2; h++) store_clip(orow, 4ll * (px0 + 8 * h), q + 8 * h, 8, 4ll * W, 0, 0);
            }
            continue;
        }
        if (OUT == OUT_RGB_CHW) {
            // extension: planar u8 (tensor layout C x H x W), 16 pixels -> 16 bytes in each plane
#pragma unroll
            for (int pl = 0; pl < 3; pl++) {
                uint32_t q[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const RGB2& a = HS == 2 ? c[k] : c[2 * k];
                    const RGB2& b = HS == 2 ? c[4 + k] : c[2 * k + 1];
                    q[k] = pl == 0 ? pack_plane4<HS == 2>(a.r, b.r) : (pl == 1 ? pack_plane4<HS == 2>(a.g, b.g) : pack_plane4<HS == 2>(a.b, b.b));
                }
                uint8_t* prow = orow + (long long)pl * p.plane_stride;
                if (FAST && !irregular) { const U4 v = {q[0], q[1], q[2], q[3]}; store16(prow + px0, v); }
                else store_clip(prow, px0, q, 4, W, 0, 0);
            }
            continue;
        }
        ZJ_COLOR_SB();
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (HS == 2) pack_rgb4_eo(c[k], c[4 + k], d[3 * k], d[3 * k + 1], d[3 * k + 2]);
            else pack_rgb4_nat(c[2 * k], c[2 * k + 1], d[3 * k], d[3 * k + 1], d[3 * k + 2]);
        }
        if (!FAST || (irregular && !TS)) { // (TS: an irregular group is staged like any other; color_copyout stores its units)
#pragma unroll
            for (int h = 0; h < 2; h++) store_unit_generic<OUT>(p, orow, P, (px0 >> 3) + h, d + 6 * h);
            continue;
        }
        const U4 s0 = {d[0], d[1], d[2], d[3]}, s1 = {d[4], d[5], d[6], d[7]}, s2 = {d[8], d[9], d[10], d[11]};
        if (ZJ_ABL(ZJ_PDBG(p), 8) && (d[0] ^ d[5] ^ d[11]) != 0x12345u) continue; // ablation: (practically) no HBM writes
        const int G = px0 >> 4; // 16-pixel group index in the row
        const bool quirk = OUT == OUT_RGB && !p.plain && !RAG; // (RAG: the tail groups went the generic way above)
        if (TS) {
            // The early RGB tail (Q5) is a shift by one piece: the last group of a row starts in the third slot of the
            // group before it, and the row's last piece is zero (Q6) or never stored (color_copyout)
            io->s0 = s0; io->s1 = s1; io->s2 = s2;
            io->kind = (quirk && G == elements) ? 3 : ((quirk && G == elements - 1) ? 2 : 1);
            continue;
        }
        if (!quirk) {
            uint8_t* o = orow + 48ll * G;
            store16(o, s0); store16(o + 16, s1); store16(o + 32, s2);
        } else {
            // color_convert_ycbcr (worker.rs:166-250), W % 16 == 0, W >= 32:
            //   groups 0..elements-1 at 48*G; the LAST 16 samples at 3W-64 (16 bytes early, Q5);
            //   bytes [3W-16, 3W) of every row are never written by the reference (Q6).
            if (G < elements - 1) {
                uint8_t* o = orow + 48ll * G;
                store16(o, s0); store16(o + 16, s1); store16(o + 32, s2);
            } else if (G == elements - 1) {
                uint8_t* o = orow + 48ll * G;
                store16(o, s0); store16(o + 16, s1); // last 16 bytes are overwritten by the tail
            } else {
                uint8_t* o = orow + 3ll * W - 64;
                store16(o, s0); store16(o + 16, s1); store16(o + 32, s2);
                if (p.zero_fill) { const U4 z = {0, 0, 0, 0}; store16(o + 48, z); }
            }
        }
    }
}

// Second half of a staged-store round: lane L of a wave stores pieces 64*j + L (j = 0 .. PPI-1) of the 64*PPI pieces
// its wave staged, i.e. every store instruction writes 1024 contiguous bytes of the tile's rows (row segments of
// PIECES_PER_ROW pieces), instead of 64 pieces 48 (64) bytes apart.
template <class C, int OUT, bool RAG = false, bool SEAM = false>
ZJ_DEV void color_copyout(const Params& p, const TileId t, const int tid /* logical */, char* lds, const int round, const int hw_wave = -1)
{
    using LL = typename C::template L<GEN_PACKED>;
    constexpr int PPI = C::PPI;
    const int P = RAG ? p.regular_px : p.mcu_x * 8 * (C::TWYB / C::TWC); // RAG: only the ordinary groups were staged
    const int x0 = t.tile * C::TWY;
    const int left = P - x0 > 0 ? (P - x0) / 16 : 0;
    const int nvg = left < C::NGRP ? left : C::NGRP; // valid 16-pixel groups of this tile
    const uint32_t row_bytes = (uint32_t)p.out_pitch;
    // everything up to here is uniform: a scalar base address, 32-bit per-lane offsets below
    uint8_t* const tile_out = t.out + (long long)t.strip * C::SH * row_bytes + (long long)(PPI == 4 ? 4 : 3) * x0;
    const int w = uniform(tid >> 6), L = tid & 63;
    const int item0 = 64 * w + round * C::NT;  // first item of this wave's round
    if (item0 >= C::NITEMS) return;            // (the last round of a tile is partly empty)
    // the piece the reference never writes: the last one of a row, in the tile that holds the row's end
    const bool row_end_here = !RAG && x0 + 16 * nvg == P;
    const int never = (OUT == OUT_RGB && !p.plain && !p.zero_fill && row_end_here) ? 3 * nvg - 1 : -1;
    const int rows_left = p.height - t.strip * C::SH; // > 0
    const char* const ybase = lds + 16 * item0;
    const char* const xbase = lds + LL::X_OFF + C::XSTAGE * (hw_wave < 0 ? w : hw_wave) - 16 * C::INPL;
    const char* src[PPI];
    uint32_t off[PPI];
    int mm[PPI], cc[PPI];
    // A round of NT items covers whole tile rows when NT is a multiple of the groups per row: the pieces of round r then
    // sit NT / NGRP rows below those of round 0, in the same columns -- the division below is written for round 0 only,
    // so that the unrolled rounds of a tile share it (round 3: -20 VALU instructions per wave and round after the first)
    constexpr bool WHOLE_ROWS = C::NT % C::NGRP == 0;
    const int item_q = WHOLE_ROWS ? 64 * w : item0;
    const int row_shift = WHOLE_ROWS ? round * (C::NT / C::NGRP) : 0;
#pragma unroll
    for (int j = 0; j < PPI; j++) {
        src[j] = (j == 0 && L < C::INPL ? ybase : xbase) + 16 * (64 * j + L);
        const int Q = PPI * item_q + 64 * j + L;  // piece index inside the tile (of round 0 when WHOLE_ROWS)
        const int m0 = (int)(((uint32_t)Q * (uint32_t)C::PPR_MAGIC) >> 20); // Q / PIECES_PER_ROW (3 instructions; checked in Cfg)
        cc[j] = Q - m0 * C::PIECES_PER_ROW;
        mm[j] = m0 + row_shift;
        off[j] = (uint32_t)mul24(m0, (int32_t)row_bytes) + 16u * (uint32_t)cc[j] + (uint32_t)row_shift * row_bytes; // m < 32, row_bytes < 2^18
    }
    // interior tiles (all but the last of a row, all but a clipped last strip), whole rounds: no per-piece test,
    // PPI LDS reads, one wait, PPI stores
    const bool plain_round = nvg == C::NGRP && rows_left >= C::SH && never < 0 && item0 + 64 <= C::NITEMS;
    // Rows that are not dword-aligned (ragged widths with width % 4 != 0, or a frame that starts at an odd address): a
    // 16-byte store per lane at an address that is not a multiple of 4 costs the memory pipeline dearly (+22 % kernel time
    // at 4090 pixels, where every second row is off by 2; dword-aligned but not 16-byte-aligned rows cost 3 %).  Such rows
    // are copied out SHIFTED: lane q stores bytes [16q + d, 16q + d + 16) of its row segment, d = the row's distance to the
    // next dword boundary, assembled from its own piece and the first dword of the next one with v_alignbyte_b32 -- every
    // 16-byte store is dword-aligned; the first d bytes of a row segment and the short last piece go out as bytes / dwords
    // from the two lanes at its ends.
    // Rows whose pitch or start is not a multiple of 128 bytes: a tile's row segment shares its first and last cache line
    // with the neighbouring tiles (tools/store_probe, profiles/r05_store_probe.txt).  x = byte offset of a piece from the
    // 128-byte boundary below its row segment's start.
    // (SEAM: an instantiation of its own, so that frames whose rows do start on line boundaries keep their code; rows of an
    // aligned width start on 16-byte boundaries, so a piece never lies across two lines)
    static_assert(!(SEAM && RAG), "the seam form serves aligned widths");
    uint32_t la0 = 0, lrho = 0, segb = 0;
    if (SEAM) { la0 = (uint32_t)reinterpret_cast<uintptr_t>(tile_out) & 127u; lrho = row_bytes & 127u; segb = 16u * (uint32_t)(PPI * nvg); }
    const bool seams = SEAM && (la0 | lrho) != 0;          // workgroup-uniform; segb = bytes of this tile's row segment
    auto shared_line = [&](const int j) {
        const uint32_t A = (la0 + (uint32_t)mm[j] * lrho) & 127u, ls = (A + 16u * (uint32_t)cc[j]) & ~127u;
        return ls < A || ls + 128u > A + segb;
    };
    if (RAG) {
        const uint32_t a0 = (uint32_t)reinterpret_cast<uintptr_t>(tile_out) & 3u, rho = row_bytes & 3u;
        if ((a0 | rho) != 0) { // workgroup-uniform
            ZJ_NO_IF_CONVERT();
#pragma unroll
            for (int j = 0; j < PPI; j++) {
                const bool ok = plain_round || (PPI * item0 + 64 * j + L < PPI * C::NITEMS && cc[j] < PPI * nvg && mm[j] < rows_left);
                const bool last = cc[j] == PPI * nvg - 1;   // the row segment's last ordinary piece: nothing of the next one
                const uint32_t delta = (4u - ((a0 + (uint32_t)mm[j] * rho) & 3u)) & 3u;
                const U4 v = *reinterpret_cast<const U4*>(src[j]);
                const int qn = 64 * j + L + 1;              // the next piece of the wave's round (same row unless `last`)
                const char* const src2 = last ? src[j] : (qn < C::INPL ? ybase : xbase) + 16 * qn;
                const uint32_t nx = *reinterpret_cast<const uint32_t*>(src2);
                U4 o;
                o.x = alignbyte(v.y, v.x, delta); o.y = alignbyte(v.z, v.y, delta);
                o.z = alignbyte(v.w, v.z, delta); o.w = alignbyte(nx, v.w, delta);
                uint8_t* const dst = tile_out + off[j];
                if (ok) {
                    if (!last) store16(dst + delta, o, true);
                    else {
                        uint32_t* const d32 = reinterpret_cast<uint32_t*>(dst + delta);
                        d32[0] = o.x; d32[1] = o.y; d32[2] = o.z;
                        if (delta == 0) d32[3] = o.w;
                        else for (uint32_t b = 0; b < 4u - delta; b++) dst[delta + 12 + b] = (uint8_t)(o.w >> (8 * b));
                    }
                    if (cc[j] == 0) for (uint32_t b = 0; b < delta; b++) dst[b] = (uint8_t)(v.x >> (8 * b));
                }
            }
            goto ragged_tail;
        }
    }
    if (plain_round) {
        U4 v[PPI];
#pragma unroll
        for (int j = 0; j < PPI; j++) v[j] = *reinterpret_cast<const U4*>(src[j]);
        if (SEAM && seams) {
#pragma unroll
            for (int j = 0; j < PPI; j++) store16_seam(tile_out + off[j], v[j], shared_line(j));
            return;
        }
#pragma unroll
        for (int j = 0; j < PPI; j++) store16(tile_out + off[j], v[j], true);
        return;
    }
    ZJ_NO_IF_CONVERT();
#pragma unroll
    for (int j = 0; j < PPI; j++) {
        const bool ok = PPI * item0 + 64 * j + L < PPI * C::NITEMS && cc[j] < PPI * nvg && mm[j] < rows_left && cc[j] != never;
        if (SEAM && ok && seams) { store16_seam(tile_out + off[j], *reinterpret_cast<const U4*>(src[j]), shared_line(j)); continue; }
        if (ok) store16(tile_out + off[j], *reinterpret_cast<const U4*>(src[j]), true);
    }
ragged_tail:
    if (RAG && nvg < C::NGRP) {
        // The row's end lies in this tile (workgroup-uniform): its groups beyond regular_px were staged like the others and
        // are stored here, one lane per (row, 8-pixel unit), by the generic rules -- early tail, gap, clip at the row's end.
        // At most four groups of a row are irregular (56 pixels), so eight units per row cover them.
        constexpr int RPR = 64 / C::NGRP > 0 ? 64 / C::NGRP : 1;  // tile rows of a wave's round (NGRP <= 64)
        const int Pfull = p.mcu_x * 8 * (C::TWYB / C::TWC);
        const int r = L >> 3, k = L & 7;
        const int u = (p.regular_px >> 3) + k;                    // unit index in the padded row
        const int g = (8 * u - x0) >> 4;                          // its group inside the tile
        const int item = r * C::NGRP + g;                         // ... inside the wave's round
        static_assert(!RAG || (64 % C::NGRP == 0 && C::NT % C::NGRP == 0), "a wave's round covers whole tile rows");
        const int m = item0 / C::NGRP + r;                        // tile row
        if (r < RPR && 8 * u >= x0 && g < C::NGRP && 8 * u < Pfull && item0 + item < C::NITEMS && m < rows_left) {
            const char* const sp = (PPI * item < C::INPL ? ybase : xbase) + 16 * PPI * item + (PPI * 4) * 2 * (u & 1);
            uint32_t wv[2 * PPI];
#pragma unroll
            for (int i = 0; i < 2 * PPI; i++) wv[i] = *reinterpret_cast<const uint32_t*>(sp + 4 * i);
            uint8_t* const orow = t.out + (long long)(t.strip * C::SH + m) * row_bytes;
            if (PPI == 4) store_clip(orow, 32ll * u, wv, 8, 4ll * p.width, 0, 0);
            else store_unit_generic<OUT>(p, orow, Pfull, u, wv);
        }
    }
}

// Are staged stores usable for this launch?  The early-tail shift writes into the piece before the lane's own, so
// the row's last group must never sit in lane 0 of a wave, and both tail groups must be in one tile.
template <class C>
inline bool ts_eligible(const Params& p, const int out, const bool fast, const bool rag = false)
{
    if (!fast || !(out == OUT_RGB || out == OUT_YCBCR || out == OUT_RGBA)) return false;
    if (out != OUT_RGB || p.plain || rag) return true; // (rag: the early tail never goes through the staging)
    const int P = p.mcu_x * 8 * (C::TWYB / C::TWC);
    const int tiles = (P + C::TWY - 1) / C::TWY;
    const int nvg_last = (P - (tiles - 1) * C::TWY) / 16;
    if (nvg_last < 2) return false;
    for (int m = 0; m < C::SH; m++)
        if ((m * C::NGRP + nvg_last - 1) % 64 == 0) return false;
    return true;
}

// Does this launch take the SEAM instantiation (color_copyout)?  4:2:0 (256-pixel tiles: the seams are densest there; the
// 512-pixel tiles of 4:4:4 and the 16-row tiles of 4:2:2 measured 0 .. -3 %), staged stores, and rows that do not start on
// 128-byte boundaries: the pitch, the first frame, the distance between frames, or any frame of a scattered launch.
template <class C>
inline bool seam_launch(const Params& p)
{
    if (!ZJ_SEAM_WB || C::YBR != 4) return false;
    if (p.out_pitch & 127) return true;
    if (p.y != nullptr) return (((uintptr_t)p.out | (uintptr_t)p.out_frame_stride) & 127) != 0;
    for (int f = 0; f < SCATTER_MAX; f++) if (p.fptr[f][3] & 127) return true;
    return false;
}

} // namespace zj
