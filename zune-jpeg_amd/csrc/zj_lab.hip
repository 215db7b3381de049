// zj_lab.hip -- "kernel lab": the IDCT and colour stages timed in isolation (registers/LDS only, no
// HBM traffic) in several formulations, to pick the cheapest exact one on gfx950.  tools/lab.py.
// Not part of the decode path.
#include <hip/hip_runtime.h>

#include "zj_device.h"
#include "zj_launch.h"

namespace zj {

// ---- 1-D pass variants (all compute the same integers mod 2^32) -------------------------------
// V0: odd part as direct 4x4 matrix, 24-bit mul/mad (the shipped form)
// V1: stb butterfly (9 muls + adds) with 24-bit mul/mad -- exact only when sums fit 24 bits (pass 2)
// V2: stb butterfly with full 32-bit multiplies (v_mul_lo_u32), exact everywhere
template <int V>
__device__ __forceinline__ void lab_1d(const int32_t s[8], const int32_t bias, int32_t o[8])
{
    if (V == 0) { idct_1d(s, bias, o); return; }
    if (V == 3) { // direct matrix with the multiply-add chains pinned (no re-association), 3-op even head
        auto K = [](int32_t& x) { asm volatile("" : "+v"(x)); };
        int32_t t3 = mul24(s[2], 2217 + 3135); K(t3); t3 = mad24(s[6], 2217, t3);
        int32_t t2 = mul24(s[2], 2217); K(t2); t2 = mad24(s[6], 2217 - 7567, t2);
        int32_t A = wadd(wshl(s[0], 12), bias); K(A);
        const int32_t t0 = mad24(s[4], 4096, A), t1 = mad24(s[4], -4096, A);
        const int32_t x0 = wadd(t0, t3), x3 = wsub(t0, t3), x1 = wadd(t1, t2), x2 = wsub(t1, t2);
        const int32_t a = s[7], b = s[5], c = s[3], d = s[1];
        int32_t u3 = mul24(d, 5683); K(u3); u3 = mad24(a, 1131, u3); K(u3); u3 = mad24(b, 3219, u3); K(u3); u3 = mad24(c, 4816, u3);
        int32_t u2 = mul24(c, -1129); K(u2); u2 = mad24(b, -5681, u2); K(u2); u2 = mad24(a, -3218, u2); K(u2); u2 = mad24(d, 4816, u2);
        int32_t u1 = mul24(b, 1132); K(u1); u1 = mad24(c, -5681, u1); K(u1); u1 = mad24(d, 3219, u1); K(u1); u1 = mad24(a, 4816, u1);
        int32_t u0 = mul24(a, -5680); K(u0); u0 = mad24(d, 1131, u0); K(u0); u0 = mad24(c, -3218, u0); K(u0); u0 = mad24(b, 4816, u0);
        o[0] = wadd(x0, u3); o[7] = wsub(x0, u3);
        o[1] = wadd(x1, u2); o[6] = wsub(x1, u2);
        o[2] = wadd(x2, u1); o[5] = wsub(x2, u1);
        o[3] = wadd(x3, u0); o[4] = wsub(x3, u0);
        return;
    }
    auto M = [](int32_t a, int32_t k) -> int32_t { return V == 1 ? mul24(a, k) : (int32_t)((uint32_t)a * (uint32_t)k); };
    const int32_t p1e = M(wadd(s[2], s[6]), 2217);
    const int32_t t2 = wadd(p1e, M(s[6], -7567));
    const int32_t t3 = wadd(p1e, M(s[2], 3135));
    const int32_t t0 = wadd(wshl(wadd(s[0], s[4]), 12), bias);
    const int32_t t1 = wadd(wshl(wsub(s[0], s[4]), 12), bias);
    const int32_t x0 = wadd(t0, t3), x3 = wsub(t0, t3), x1 = wadd(t1, t2), x2 = wsub(t1, t2);
    int32_t a = s[7], b = s[5], c = s[3], d = s[1];
    const int32_t p3 = wadd(a, c), p4 = wadd(b, d), p1 = wadd(a, d), p2 = wadd(b, c);
    const int32_t p5 = M(wadd(p3, p4), 4816);
    const int32_t P1 = wadd(p5, M(p1, -3685)), P2 = wadd(p5, M(p2, -10497));
    const int32_t P3 = M(p3, -8034), P4 = M(p4, -1597);
    const int32_t u3 = wadd(M(d, 6149), wadd(P1, P4));
    const int32_t u2 = wadd(M(c, 12586), wadd(P2, P3));
    const int32_t u1 = wadd(M(b, 8410), wadd(P2, P4));
    const int32_t u0 = wadd(M(a, 1223), wadd(P1, P3));
    o[0] = wadd(x0, u3); o[7] = wsub(x0, u3);
    o[1] = wadd(x1, u2); o[6] = wsub(x1, u2);
    o[2] = wadd(x2, u1); o[5] = wsub(x2, u1);
    o[3] = wadd(x3, u0); o[4] = wsub(x3, u0);
}

// ---- "grouped" formulation: the multiply half of G transforms, then their add/shift half, separated by
// scheduling barriers, so that simple VOP2 instructions (add/sub/shift: ~2.3 cycles in pure runs, ~3.5-4 when
// mixed with multiplies, profiles/r01_ubench_valu_issue_cost.txt) sit next to each other
struct LabHalf { int32_t t0, t1, t2, t3, u0, u1, u2, u3; };
__device__ __forceinline__ LabHalf lab_1d_mul(const int32_t s[8], const int32_t bias)
{
    auto K = [](int32_t& x) { asm volatile("" : "+v"(x)); };
    LabHalf h;
    h.t3 = mul24(s[2], 2217 + 3135); K(h.t3); h.t3 = mad24(s[6], 2217, h.t3);
    h.t2 = mul24(s[2], 2217); K(h.t2); h.t2 = mad24(s[6], 2217 - 7567, h.t2);
    int32_t A = wadd(wshl(s[0], 12), bias); K(A);
    h.t0 = wadd(wshl(s[4], 12), A); K(h.t0);
    h.t1 = mad24(s[4], -4096, A);
    const int32_t a = s[7], b = s[5], c = s[3], d = s[1];
    h.u3 = mul24(d, 5683); K(h.u3); h.u3 = mad24(a, 1131, h.u3); K(h.u3); h.u3 = mad24(b, 3219, h.u3); K(h.u3); h.u3 = mad24(c, 4816, h.u3);
    h.u2 = mul24(c, -1129); K(h.u2); h.u2 = mad24(b, -5681, h.u2); K(h.u2); h.u2 = mad24(a, -3218, h.u2); K(h.u2); h.u2 = mad24(d, 4816, h.u2);
    h.u1 = mul24(b, 1132); K(h.u1); h.u1 = mad24(c, -5681, h.u1); K(h.u1); h.u1 = mad24(d, 3219, h.u1); K(h.u1); h.u1 = mad24(a, 4816, h.u1);
    h.u0 = mul24(a, -5680); K(h.u0); h.u0 = mad24(d, 1131, h.u0); K(h.u0); h.u0 = mad24(c, -3218, h.u0); K(h.u0); h.u0 = mad24(b, 4816, h.u0);
    return h;
}
__device__ __forceinline__ void lab_1d_add(const LabHalf& h, int32_t o[8])
{
    const int32_t x0 = wadd(h.t0, h.t3), x3 = wsub(h.t0, h.t3), x1 = wadd(h.t1, h.t2), x2 = wsub(h.t1, h.t2);
    o[0] = wadd(x0, h.u3); o[7] = wsub(x0, h.u3);
    o[1] = wadd(x1, h.u2); o[6] = wsub(x1, h.u2);
    o[2] = wadd(x2, h.u1); o[5] = wsub(x2, h.u1);
    o[3] = wadd(x3, h.u0); o[4] = wsub(x3, h.u0);
}
template <int G>
__device__ __forceinline__ void lab_block_grouped(const U4 raw[8], const int32_t* qt, U4 out[8])
{
    const uint32_t* w = reinterpret_cast<const uint32_t*>(raw);
    int32_t tmp[64];
#pragma unroll
    for (int c0 = 0; c0 < 8; c0 += G) {
        LabHalf h[G];
#pragma unroll
        for (int g = 0; g < G; g++) {
            const int col = c0 + g;
            int32_t s[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const uint32_t pair = w[k * 4 + (col >> 1)];
                const int32_t cf = (col & 1) ? hi16s(pair) : lo16s(pair);
                s[k] = mul24(cf, qt[k * 8 + col]);
            }
            h[g] = lab_1d_mul(s, 512);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < G; g++) {
            int32_t o[8];
            lab_1d_add(h[g], o);
#pragma unroll
            for (int k = 0; k < 8; k++) tmp[k * 8 + c0 + g] = o[k] >> 10;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    constexpr int32_t bias2 = 512 + 65536 + (128 << 17);
    uint32_t* ow = reinterpret_cast<uint32_t*>(out);
#pragma unroll
    for (int r0 = 0; r0 < 8; r0 += G) {
        LabHalf h[G];
#pragma unroll
        for (int g = 0; g < G; g++) h[g] = lab_1d_mul(&tmp[(r0 + g) * 8], bias2);
        __builtin_amdgcn_sched_barrier(0);
        int32_t o[G][8];
#pragma unroll
        for (int g = 0; g < G; g++) lab_1d_add(h[g], o[g]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < G; g++)
#pragma unroll
            for (int k = 0; k < 8; k += 2) {
                const uint32_t hi = perm((uint32_t)o[g][k + 1], (uint32_t)o[g][k], 0x07060302u);
                const s16x2 z = {0, 0}, m = {255, 255};
                ow[(r0 + g) * 4 + (k >> 1)] = as_u32(pk_min(pk_max(sar(as_u16x2(hi), 1), z), m));
            }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// FIN: 0 = med3 + lshl_or (shipped), 1 = pack first then packed clamp
template <int V1, int V2, int FIN>
__device__ __forceinline__ void lab_block(const U4 raw[8], const int32_t* qt, U4 out[8])
{
    const uint32_t* w = reinterpret_cast<const uint32_t*>(raw);
    int32_t tmp[64];
#pragma unroll
    for (int col = 0; col < 8; col++) {
        int32_t s[8], o[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint32_t pair = w[k * 4 + (col >> 1)];
            const int32_t cf = (col & 1) ? hi16s(pair) : lo16s(pair);
            s[k] = mul24(cf, qt[k * 8 + col]);
        }
        lab_1d<V1>(s, 512, o);
#pragma unroll
        for (int k = 0; k < 8; k++) tmp[k * 8 + col] = o[k] >> 10;
    }
    constexpr int32_t bias2 = 512 + 65536 + (128 << 17);
    uint32_t* ow = reinterpret_cast<uint32_t*>(out);
#pragma unroll
    for (int r = 0; r < 8; r++) {
        int32_t o[8];
        lab_1d<V2>(&tmp[r * 8], bias2, o);
#pragma unroll
        for (int k = 0; k < 8; k += 2) {
            int32_t p0 = o[k] >> 17, p1 = o[k + 1] >> 17;
            if (FIN == 0) {
                p0 = p0 < 0 ? 0 : (p0 > 255 ? 255 : p0);
                p1 = p1 < 0 ? 0 : (p1 > 255 ? 255 : p1);
                ow[r * 4 + (k >> 1)] = (uint32_t)p0 | ((uint32_t)p1 << 16);
            } else {
                // (x >> 17) always fits 15 bits: pack, then clamp both lanes with packed min/max
                const uint32_t pk = ((uint32_t)p0 & 0xffffu) | ((uint32_t)p1 << 16);
                const s16x2 z = {0, 0}, m = {255, 255};
                ow[r * 4 + (k >> 1)] = as_u32(pk_min(pk_max(as_s16x2(pk), z), m));
            }
        }
    }
}

template <int V1, int V2, int FIN, int QSRC>
__global__ __launch_bounds__(256) void lab_idct(const int32_t* __restrict__ qt_g, int* out, int iters)
{
    __shared__ int32_t qt_l[192];
    if (threadIdx.x < 192) qt_l[threadIdx.x] = qt_g[threadIdx.x];
    __syncthreads();
    const int32_t* qt = QSRC == 0 ? (const int32_t*)(qt_l + 64 * (threadIdx.x % 3)) : qt_g;
    U4 raw[8];
    uint32_t seed = threadIdx.x * 2654435761u + blockIdx.x;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        seed = seed * 1664525u + 1013904223u; raw[i].x = seed & 0x003f003f;
        seed = seed * 1664525u + 1013904223u; raw[i].y = seed & 0x001f001f;
        seed = seed * 1664525u + 1013904223u; raw[i].z = seed & 0x000f000f;
        seed = seed * 1664525u + 1013904223u; raw[i].w = seed & 0x00070007;
    }
    for (int it = 0; it < iters; it++) {
        U4 px[8];
        if (V1 >= 10) lab_block_grouped<(V1 >= 10 ? V1 - 10 : 1)>(raw, qt, px);
        else if (FIN == 2) idct_block(raw, qt, px); // the shipped block function
        else lab_block<(V1 >= 10 ? 0 : V1), V2, (FIN == 2 ? 1 : FIN)>(raw, qt, px);
#pragma unroll
        for (int i = 0; i < 8; i++) raw[i] = px[i]; // dependent chain: nothing can be hoisted
    }
    uint32_t acc = 0;
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

typedef void (*lab_fn)(const int32_t*, int*, int);
static const struct { const char* name; lab_fn fn; } LAB[] = {
    {"idct direct/direct   med3  qt=LDS (shipped)", lab_idct<0, 0, 0, 0>},
    {"idct direct/direct   med3  qt=SGPR", lab_idct<0, 0, 0, 1>},
    {"idct direct/stb24    med3  qt=LDS", lab_idct<0, 1, 0, 0>},
    {"idct stb32/stb24     med3  qt=LDS", lab_idct<2, 1, 0, 0>},
    {"idct stb32/stb32     med3  qt=LDS", lab_idct<2, 2, 0, 0>},
    {"idct direct/direct   pkclamp qt=LDS", lab_idct<0, 0, 1, 0>},
    {"idct direct/stb24    pkclamp qt=LDS", lab_idct<0, 1, 1, 0>},
    {"idct pinned/pinned   med3  qt=LDS", lab_idct<3, 3, 0, 0>},
    {"idct pinned/pinned   pkclamp qt=LDS", lab_idct<3, 3, 1, 0>},
    {"idct pinned/stb24    pkclamp qt=LDS", lab_idct<3, 1, 1, 0>},
    {"idct stb32/stb24     pkclamp qt=LDS", lab_idct<2, 1, 1, 0>},
    {"idct pinned/pinned   pkclamp qt=SGPR", lab_idct<3, 3, 1, 1>},
    {"idct_block as shipped (perm epilogue)  qt=LDS", lab_idct<3, 3, 2, 0>},
    {"idct grouped x1 (mul half | add half) qt=LDS", lab_idct<11, 3, 1, 0>},
    {"idct grouped x2                       qt=LDS", lab_idct<12, 3, 1, 0>},
    {"idct grouped x4                       qt=LDS", lab_idct<14, 3, 1, 0>},
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

int lab_count() { return (int)(sizeof(LAB) / sizeof(LAB[0])) + 1; }
const char* lab_name(int i) { return i < lab_count() - 1 ? LAB[i].name : "colour 16px/lane (ycc->rgb + pack, EO)"; }
hipError_t launch_lab(int i, const int32_t* qt, int* out, int blocks, int iters, hipStream_t s)
{
    if (i < 0 || i >= lab_count()) return hipErrorInvalidValue;
    if (i == lab_count() - 1) hipLaunchKernelGGL(lab_color<0>, dim3(blocks), dim3(256), 0, s, out, iters);
    else hipLaunchKernelGGL(LAB[i].fn, dim3(blocks), dim3(256), 0, s, qt, out, iters);
    return hipGetLastError();
}

} // namespace zj
