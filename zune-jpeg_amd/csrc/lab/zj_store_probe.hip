// Store-pattern probe (tools only, never in the product): what does a row pitch that is not a multiple of 128 bytes
// cost the fused kernel's copy-out, and does a lane rotation that keeps every 8-lane group inside one 128-byte line
// recover it?  Each workgroup writes one 256 x 32 pixel RGB tile the way color_copyout does: a wave's store instruction
// covers 64 consecutive 16-byte pieces of the tile's 768-byte row segments.  No loads, no arithmetic: the store path alone.
//   usage: zj_store_probe [frames]        build: hipcc --offload-arch=gfx950 -O3 -o tools/store_probe <this file>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef unsigned int u4 __attribute__((ext_vector_type(4)));

// MODE 0: natural lane order; 1: lanes rotated so that every 8-lane group lies in one 128-byte line; 2: ... 4-lane group in 64 bytes;
// 3: natural order, the pieces at a row segment's two ends (the sectors shared with the neighbouring tiles) stored
//    write-back, everything else non-temporal (NT must be true)
// XCD: workgroup -> tile as the product does it (zj_device.h: xcd_order): each XCD takes a contiguous run of tiles, so the
//    tiles on both sides of a seam meet in ONE L2
// LD (second table): 0 no loads; 1 / 2: every workgroup first streams in as many bytes as it writes (24 KB, its tile's
//    coefficients in the fused kernel) with ordinary / non-temporal loads, folds them into what it stores, and waits a
//    pseudo-random 0 .. 8 us before its stores -- neighbouring tiles of the fused kernel do not reach their copy-out together
// PF (fourth table): after its own loads and its wait a workgroup touches one dword of each of the 192 lines that the
//    workgroup PF launch positions behind it will read (the one that takes over its slot, give or take): a prefetch into
//    the L2 by the only means gfx950 has, a load whose result nobody waits for
template <int MODE, bool NT, bool XCD, int LD = 0, int PF = 0>
__global__ __launch_bounds__(256) void probe(unsigned char* out, int tiles_per_row, int strips, unsigned pitch, long long frame_bytes,
                                             const u4* in = nullptr)
{
    int bid = blockIdx.x;
    if (XCD && (gridDim.x & 7) == 0) bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);
    u4 acc = {0, 0, 0, 0};
    if (LD) {
        const u4* src = in + (size_t)bid * 1536 + threadIdx.x;   // 24 KB per tile
#pragma unroll
        for (int k = 0; k < 6; k++) {
            const u4 x = LD == 2 ? __builtin_nontemporal_load(src + 256 * k) : src[256 * k];
            acc.x ^= x.x; acc.y ^= x.y; acc.z ^= x.z; acc.w ^= x.w;
        }
        const unsigned hsh = ((unsigned)bid * 2654435761u) >> 16;
        for (unsigned k = hsh % 38u; k > 0; k--) __builtin_amdgcn_s_sleep(8);   // 512 cycles a step
        if (PF) {
            int nb = (int)blockIdx.x + PF;
            if (nb < (int)gridDim.x && threadIdx.x < 192) {
                if (XCD && (gridDim.x & 7) == 0) nb = (nb & 7) * (gridDim.x >> 3) + (nb >> 3);
                const unsigned* line = reinterpret_cast<const unsigned*>(in + (size_t)nb * 1536) + 32 * threadIdx.x;
                unsigned dummy;
                asm volatile("global_load_dword %0, %1, off" : "=v"(dummy) : "v"(line) : "memory");
            }
        }
    }
    const int per_frame = tiles_per_row * strips;
    const int frame = bid / per_frame, rem = bid % per_frame;
    const int strip = rem / tiles_per_row, tile = rem % tiles_per_row;
    unsigned char* const tile_out = out + frame * frame_bytes + (long long)strip * 32 * pitch + 768ll * tile;
    const int w = threadIdx.x >> 6, L = threadIdx.x & 63;
    const unsigned a0 = (unsigned)((unsigned long long)tile_out >> 4);
    const unsigned rho = pitch >> 4;
    const u4 v = {(unsigned)bid ^ acc.x, (unsigned)threadIdx.x ^ acc.y, 0x01020304u ^ acc.z, 0x05060708u ^ acc.w};
#pragma unroll
    for (int round = 0; round < 2; round++) {
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const int Q = 3 * (64 * w + 256 * round) + 64 * j + L;
            const int m = Q / 48;
            int cc = Q - 48 * m;
            if (MODE == 1 || MODE == 2) {
                const unsigned s = (a0 + (unsigned)m * rho) & (MODE == 1 ? 7u : 3u);
                cc -= (int)s;
                if (cc < 0) cc += 48;
            }
            u4* const dst = reinterpret_cast<u4*>(tile_out + (unsigned)m * pitch + 16u * (unsigned)cc);
            if (MODE == 3) {
                // (written as two plain C++ stores the compiler merges the arms into one store WITHOUT the nt bit)
                if (cc == 0 || cc == 47) asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(dst), "v"(v) : "memory");
                else asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(dst), "v"(v) : "memory");
            } else if (MODE == 4 || MODE == 5) {
                // every piece of a sector (4: 64 bytes, 5: a 128-byte line) that the row segment shares with a neighbouring tile
                // is stored write-back, so that the L2 can put the two halves together; the sectors the tile owns stream out
                const unsigned long long G = MODE == 4 ? 64 : 128;
                const unsigned long long seg = (unsigned long long)(tile_out + (unsigned)m * pitch), a = (unsigned long long)dst;
                const unsigned long long sec = a & ~(G - 1);
                if (sec < seg || sec + G > seg + 768) asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(dst), "v"(v) : "memory");
                else asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(dst), "v"(v) : "memory");
            } else if (NT) __builtin_nontemporal_store(v, dst); else *dst = v;
        }
    }
}

// Third table: the SHAPE of what a workgroup writes, at equal bytes (24 KB read as whole lines, 24 KB written non-temporally,
// XCD order, aligned rows): SEG bytes of each of 24576 / SEG consecutive rows -- 768 x 32 is the 4:2:0 tile, 1536 x 16 a
// 512-pixel tile half as tall, 24576 x 1 a plain copy.
template <int SEG>
__global__ __launch_bounds__(256) void shape_probe(unsigned char* out, const u4* in, int tiles_per_row, unsigned pitch)
{
    int bid = blockIdx.x;
    if ((gridDim.x & 7) == 0) bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);
    constexpr int ROWS = 24576 / SEG, PPR = SEG / 16;
    const int strip = bid / tiles_per_row, tile = bid % tiles_per_row;
    unsigned char* const tile_out = out + (long long)strip * ROWS * pitch + (long long)SEG * tile;
    const u4* src = in + (size_t)bid * 1536 + threadIdx.x;
    u4 v[6];
#pragma unroll
    for (int k = 0; k < 6; k++) v[k] = src[256 * k];
#pragma unroll
    for (int k = 0; k < 6; k++) {
        const int Q = 256 * k + (int)threadIdx.x;          // piece of the tile, row-major
        const int m = Q / PPR, cc = Q - PPR * m;
        __builtin_nontemporal_store(v[k], reinterpret_cast<u4*>(tile_out + (long long)m * pitch + 16 * cc));
    }
}

template <int SEG>
static float run_shape(unsigned char* buf, const u4* in, long long total_bytes)
{
    // one "image" whose rows are 12288 bytes (4096 RGB pixels): tiles_per_row = 12288 / SEG (>= 1), rows = total / 12288
    const unsigned pitch = SEG > 12288 ? SEG : 12288;
    const int tiles = pitch / SEG;
    const int nwg = (int)(total_bytes / 24576);
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int i = 0; i < 20; i++) shape_probe<SEG><<<nwg, 256>>>(buf, in, tiles, pitch);
    CHECK(hipEventRecord(a));
    const int reps = 50;
    for (int i = 0; i < reps; i++) shape_probe<SEG><<<nwg, 256>>>(buf, in, tiles, pitch);
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    CHECK(hipEventDestroy(a)); CHECK(hipEventDestroy(b));
    return ms / reps;
}

template <int MODE, bool NT, bool XCD, int LD = 0, int PF = 0>
static float run(unsigned char* buf, int W, int H, int frames, size_t offset, const u4* in = nullptr)
{
    const int tiles = W / 256, strips = H / 32;
    const unsigned pitch = 3u * W;
    const long long fb = (long long)pitch * H;
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int i = 0; i < 20; i++) probe<MODE, NT, XCD, LD, PF><<<tiles * strips * frames, 256>>>(buf + offset, tiles, strips, pitch, fb, in);
    CHECK(hipEventRecord(a));
    const int reps = 50;
    for (int i = 0; i < reps; i++) probe<MODE, NT, XCD, LD, PF><<<tiles * strips * frames, 256>>>(buf + offset, tiles, strips, pitch, fb, in);
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    CHECK(hipEventDestroy(a)); CHECK(hipEventDestroy(b));
    return ms / reps;
}

int main(int argc, char** argv)
{
    const int frames = argc > 1 ? atoi(argv[1]) : 16;
    unsigned char* buf; CHECK(hipMalloc(&buf, (size_t)frames * 3 * 4352 * 4096 + 4096));
    CHECK(hipMemset(buf, 0, (size_t)frames * 3 * 4352 * 4096 + 4096));
    // the tile grid covers floor(W / 256) tiles per row: the widths differ in PITCH only (the bytes beyond the last tile are not written)
    const int widths[] = {4096, 4096 + 16, 4096 + 32, 4096 + 48, 4096 + 64, 4096 + 80, 4096 + 128, 4096 - 16 + 256};
    printf("%-30s %9s %9s | %9s %9s %9s %9s %9s %9s   (GB/s written; pure stores, %d frames of W x 4096, tiles 256 x 32)\n", "pitch",
           "wb", "NT", "xcd: wb", "xcd: NT", "rot128,NT", "ends wb", "64B mix", "128B mix", frames);
    for (int W : widths) {
        const int Wt = W / 256 * 256;   // pixels written per row
        const double bytes = (double)frames * 4096 * 3 * Wt;
        float t[8];
        t[0] = run<0, false, false>(buf, W, 4096, frames, 0); t[1] = run<0, true, false>(buf, W, 4096, frames, 0);
        t[2] = run<0, false, true>(buf, W, 4096, frames, 0);  t[3] = run<0, true, true>(buf, W, 4096, frames, 0);
        t[4] = run<1, true, true>(buf, W, 4096, frames, 0);   t[5] = run<3, true, true>(buf, W, 4096, frames, 0);
        t[6] = run<4, true, true>(buf, W, 4096, frames, 0);   t[7] = run<5, true, true>(buf, W, 4096, frames, 0);
        char name[64]; snprintf(name, sizeof name, "3 x %d = %u (%% 128 = %u)", W, 3u * W, 3u * W % 128u);
        printf("%-30s", name);
        for (int i = 0; i < 8; i++) printf(" %9.0f%s", bytes / (t[i] * 1e-3) / 1e9, i == 1 ? " |" : "");
        printf("\n");
    }
    // the same with the fused kernel's other half: as many bytes read as written, stores not in step (LD)
    u4* in; const size_t in_bytes = (size_t)frames * 16 * 128 * 24576;
    CHECK(hipMalloc(&in, in_bytes)); CHECK(hipMemset(in, 1, in_bytes));
    printf("\n%-30s %9s %9s %9s | %9s %9s %9s   (GB/s written, as much read; XCD order; loads ordinary | non-temporal; stores 0-8 us out of step)\n",
           "pitch", "wb", "NT", "128B mix", "wb", "NT", "128B mix");
    for (int W : widths) {
        const int Wt = W / 256 * 256;
        const double bytes = (double)frames * 4096 * 3 * Wt;
        float t[6];
        t[0] = run<0, false, true, 1>(buf, W, 4096, frames, 0, in); t[1] = run<0, true, true, 1>(buf, W, 4096, frames, 0, in); t[2] = run<5, true, true, 1>(buf, W, 4096, frames, 0, in);
        t[3] = run<0, false, true, 2>(buf, W, 4096, frames, 0, in); t[4] = run<0, true, true, 2>(buf, W, 4096, frames, 0, in); t[5] = run<5, true, true, 2>(buf, W, 4096, frames, 0, in);
        char name[64]; snprintf(name, sizeof name, "3 x %d = %u (%% 128 = %u)", W, 3u * W, 3u * W % 128u);
        printf("%-30s", name);
        for (int i = 0; i < 6; i++) printf(" %9.0f%s", bytes / (t[i] * 1e-3) / 1e9, i == 2 ? " |" : "");
        printf("\n");
    }
    {
        // fourth table: the desynchronised read + write pattern of the second table (aligned rows, ordinary loads, non-temporal
        // stores) with a prefetch of the slot's next occupant
        const int W = 4096;
        const double bytes = (double)frames * 4096 * 3 * W;
        const float t[5] = {run<0, true, true, 1, 0>(buf, W, 4096, frames, 0, in), run<0, true, true, 1, 1024>(buf, W, 4096, frames, 0, in),
                            run<0, true, true, 1, 2048>(buf, W, 4096, frames, 0, in), run<0, true, true, 1, 3072>(buf, W, 4096, frames, 0, in),
                            run<0, true, true, 1, 4096>(buf, W, 4096, frames, 0, in)};
        printf("\nprefetch of the slot's next occupant (GB/s read + written; loads, then 0-8 us, then stores; 8 workgroups per CU = 2048 slots):\n");
        const char* nm[5] = {"none", "1024 launch positions ahead", "2048", "3072", "4096"};
        for (int i = 0; i < 5; i++) printf("  %-32s %9.0f\n", nm[i], 2.0 * bytes / (t[i] * 1e-3) / 1e9);
    }
    {
        const long long total = (long long)frames * 4096 * 12288;
        const float t[6] = {run_shape<768>(buf, in, total), run_shape<1536>(buf, in, total), run_shape<3072>(buf, in, total),
                            run_shape<6144>(buf, in, total), run_shape<12288>(buf, in, total), run_shape<24576>(buf, in, total)};
        printf("\nshape of a workgroup's 24 KB of output (as much read; aligned rows; GB/s read + written):\n");
        const char* nm[6] = {"768 B x 32 rows (the 4:2:0 tile)", "1536 B x 16 rows", "3072 B x 8 rows", "6144 B x 4 rows", "12288 B x 2 rows", "24576 B contiguous (a copy)"};
        for (int i = 0; i < 6; i++) printf("  %-36s %9.0f\n", nm[i], 2.0 * total / (t[i] * 1e-3) / 1e9);
    }
    CHECK(hipFree(in));
    CHECK(hipFree(buf));
    return 0;
}
