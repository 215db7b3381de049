/* tools only (tools/feeder_ab.py): raw store / load rates of a host buffer and where its pages live.
   gcc -O2 -shared -fPIC -o tools/membench/libmembench.so tools/membench/membench.c */
#define _GNU_SOURCE
#include <immintrin.h>
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <time.h>
#include <sched.h>
#include <unistd.h>
#include <sys/syscall.h>

static double now(void)
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

/* seconds to write n bytes (a multiple of 128) at p (64-byte aligned) */
double mb_fill_nt16(void* p, size_t n)
{
    const __m128i v = _mm_set1_epi16(3);
    const double t0 = now();
    for (size_t o = 0; o < n; o += 16) _mm_stream_si128((__m128i*)((char*)p + o), v);
    _mm_sfence();
    return now() - t0;
}
__attribute__((target("avx"))) double mb_fill_nt32(void* p, size_t n)
{
    const __m256i v = _mm256_set1_epi16(3);
    const double t0 = now();
    for (size_t o = 0; o < n; o += 32) _mm256_stream_si256((__m256i*)((char*)p + o), v);
    _mm_sfence();
    return now() - t0;
}
__attribute__((target("avx512f"))) double mb_fill_nt64(void* p, size_t n)
{
    const __m512i v = _mm512_set1_epi16(3);
    const double t0 = now();
    for (size_t o = 0; o < n; o += 64) _mm512_stream_si512((__m512i*)((char*)p + o), v);
    _mm_sfence();
    return now() - t0;
}
double mb_fill_plain16(void* p, size_t n)
{
    const __m128i v = _mm_set1_epi16(3);
    const double t0 = now();
    for (size_t o = 0; o < n; o += 16) _mm_store_si128((__m128i*)((char*)p + o), v);
    _mm_sfence();
    return now() - t0;
}
/* the walker's pattern: a 128-byte block assembled in a stack buffer (memset + a few scalar stores), then moved out */
double mb_blocks_nt16(void* p, size_t n)
{
    int16_t blk[64] __attribute__((aligned(64)));
    const double t0 = now();
    for (size_t o = 0; o < n; o += 128) {
        memset(blk, 0, 128);
        blk[0] = (int16_t)o; blk[1] = 1; blk[8] = 2;
        for (int i = 0; i < 8; i++) _mm_stream_si128((__m128i*)((char*)p + o) + i, _mm_load_si128((const __m128i*)blk + i));
    }
    _mm_sfence();
    return now() - t0;
}
double mb_memset(void* p, size_t n)
{
    const double t0 = now();
    memset(p, 5, n);
    return now() - t0;
}
double mb_read(const void* p, size_t n, uint64_t* sink)
{
    const double t0 = now();
    __m128i a = _mm_setzero_si128();
    for (size_t o = 0; o < n; o += 16) a = _mm_add_epi64(a, _mm_load_si128((const __m128i*)((const char*)p + o)));
    *sink = (uint64_t)_mm_cvtsi128_si64(a);
    return now() - t0;
}
/* NUMA node of the page holding p, or a negative errno-style value */
int mb_page_node(void* p)
{
    void* pages[1] = {(void*)((uintptr_t)p & ~(uintptr_t)4095)};
    int status[1] = {-1000};
    const long rc = syscall(SYS_move_pages, 0, 1UL, pages, NULL, status, 0);
    return rc ? -999 : status[0];
}
/* MPOL_BIND (2) the calling thread's future allocations to `node`; node < 0: back to the default policy */
int mb_set_mempolicy(int node)
{
    if (node < 0) return (int)syscall(SYS_set_mempolicy, 0, NULL, 0UL);
    unsigned long mask[16];
    memset(mask, 0, sizeof mask);
    mask[node / 64] = 1UL << (node % 64);
    return (int)syscall(SYS_set_mempolicy, 2, mask, 1024UL);
}
int mb_cpu(void) { return sched_getcpu(); }
