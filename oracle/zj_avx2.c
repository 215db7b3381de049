/*
 * zj_avx2.c -- restatement of zune-jpeg's x86 AVX2 arms of the pixel path, multi-threaded the way the
 * reference is (one strip per pool job, src/mcu.rs:356-368).  THE TIMED CPU BASELINE of bench.py
 * ("cpu_baseline", kind "port").  TEST/BENCH INFRASTRUCTURE ONLY; never linked into libzjhip.so.
 *
 * What follows the reference instruction for instruction:
 *   idct      src/idct/avx2.rs:64-398   8 x __m256i rows of i32, dequantize, transpose, pass(512,>>10),
 *             transpose, pass(SCALE_BITS,>>17), packs/clamp/permute/store; DC-only shortcut CLAMPED
 *             (:163-167).  Row pass first, so it differs from the scalar arm by +-1:
 *             this file is NOT a parity target.
 *   colour    src/color_convert/avx.rs:81-192  16 pixels per call: sub 128, mullo_epi16, srai, add,
 *             clamp, scalar 3-byte interleave.
 *   worker    src/worker.rs:32-251 row/tail logic (shared semantics with the scalar arm).
 *   h2v2      src/upsampler/avx2.rs:29-342 upsample_hv_avx, literally (round 3): the fused vertical + horizontal filter on
 *             16 input samples per iteration, its scalar row tails and its copied edge samples (zja_upsample_hv_avx).
 *   h (SSE)   src/upsampler/sse.rs:24-134 upsample_horizontal_sse_u, literally (zja_upsample_h_sse; the reference's own
 *             SSE == scalar tests are replayed on it).
 * Kept beside them: zja_upsample_v / zja_upsample_h, the two filters vectorised with the SCALAR arm's edge rules (their
 * output equals the oracle's; rounds 1-2 timed these in place of upsample_hv_avx).
 */
#include <immintrin.h>
#include <pthread.h>
#include <stdatomic.h>
#include <stdlib.h>
#include <string.h>

#include "zj_oracle.h"

#define SCALE_BITS (512 + 65536 + (128 << 17))

/* ---- 8x8 i32 transpose, src/idct/avx2.rs:422-486 ------------------------------------------- */
static inline void transpose8(__m256i r[8])
{
    __m256 v0 = _mm256_castsi256_ps(r[0]), v1 = _mm256_castsi256_ps(r[1]), v2 = _mm256_castsi256_ps(r[2]),
           v3 = _mm256_castsi256_ps(r[3]), v4 = _mm256_castsi256_ps(r[4]), v5 = _mm256_castsi256_ps(r[5]),
           v6 = _mm256_castsi256_ps(r[6]), v7 = _mm256_castsi256_ps(r[7]);
    __m256 t0 = _mm256_unpacklo_ps(v0, v1), t1 = _mm256_unpackhi_ps(v0, v1);
    __m256 t2 = _mm256_unpacklo_ps(v2, v3), t3 = _mm256_unpackhi_ps(v2, v3);
    __m256 t4 = _mm256_unpacklo_ps(v4, v5), t5 = _mm256_unpackhi_ps(v4, v5);
    __m256 t6 = _mm256_unpacklo_ps(v6, v7), t7 = _mm256_unpackhi_ps(v6, v7);
    __m256 s0 = _mm256_shuffle_ps(t0, t2, 0x44), s1 = _mm256_shuffle_ps(t0, t2, 0xEE);
    __m256 s2 = _mm256_shuffle_ps(t1, t3, 0x44), s3 = _mm256_shuffle_ps(t1, t3, 0xEE);
    __m256 s4 = _mm256_shuffle_ps(t4, t6, 0x44), s5 = _mm256_shuffle_ps(t4, t6, 0xEE);
    __m256 s6 = _mm256_shuffle_ps(t5, t7, 0x44), s7 = _mm256_shuffle_ps(t5, t7, 0xEE);
    r[0] = _mm256_castps_si256(_mm256_permute2f128_ps(s0, s4, 0x20));
    r[1] = _mm256_castps_si256(_mm256_permute2f128_ps(s1, s5, 0x20));
    r[2] = _mm256_castps_si256(_mm256_permute2f128_ps(s2, s6, 0x20));
    r[3] = _mm256_castps_si256(_mm256_permute2f128_ps(s3, s7, 0x20));
    r[4] = _mm256_castps_si256(_mm256_permute2f128_ps(s0, s4, 0x31));
    r[5] = _mm256_castps_si256(_mm256_permute2f128_ps(s1, s5, 0x31));
    r[6] = _mm256_castps_si256(_mm256_permute2f128_ps(s2, s6, 0x31));
    r[7] = _mm256_castps_si256(_mm256_permute2f128_ps(s3, s7, 0x31));
}

#define MUL(a, k) _mm256_mullo_epi32((a), _mm256_set1_epi32(k))
#define ADD(a, b) _mm256_add_epi32((a), (b))
#define SUB(a, b) _mm256_sub_epi32((a), (b))

/* dct_pass!, src/idct/avx2.rs:251-331 */
static inline void dct_pass(__m256i r[8], int bias, int scale)
{
    const __m256i vb = _mm256_set1_epi32(bias);
    __m256i p1 = MUL(ADD(r[2], r[6]), 2217);
    __m256i t2 = ADD(p1, MUL(r[6], -7567));
    __m256i t3 = ADD(p1, MUL(r[2], 3135));
    __m256i t0 = _mm256_slli_epi32(ADD(r[0], r[4]), 12);
    __m256i t1 = _mm256_slli_epi32(SUB(r[0], r[4]), 12);
    __m256i x0 = ADD(ADD(t0, t3), vb), x3 = ADD(SUB(t0, t3), vb);
    __m256i x1 = ADD(ADD(t1, t2), vb), x2 = ADD(SUB(t1, t2), vb);
    __m256i p3 = ADD(r[7], r[3]), p4 = ADD(r[5], r[1]);
    __m256i q1 = ADD(r[7], r[1]), q2 = ADD(r[5], r[3]);
    __m256i p5 = MUL(ADD(p3, p4), 4816);
    t0 = MUL(r[7], 1223);
    t1 = MUL(r[5], 8410);
    t2 = MUL(r[3], 12586);
    t3 = MUL(r[1], 6149);
    q1 = ADD(p5, MUL(q1, -3685));
    q2 = ADD(p5, MUL(q2, -10497));
    p3 = MUL(p3, -8034);
    p4 = MUL(p4, -1597);
    t3 = ADD(t3, ADD(q1, p4));
    t2 = ADD(t2, ADD(q2, p3));
    t1 = ADD(t1, ADD(q2, p4));
    t0 = ADD(t0, ADD(q1, p3));
    const __m128i sc = _mm_cvtsi32_si128(scale);
    r[0] = _mm256_sra_epi32(ADD(x0, t3), sc);
    r[1] = _mm256_sra_epi32(ADD(x1, t2), sc);
    r[2] = _mm256_sra_epi32(ADD(x2, t1), sc);
    r[3] = _mm256_sra_epi32(ADD(x3, t0), sc);
    r[4] = _mm256_sra_epi32(SUB(x3, t0), sc);
    r[5] = _mm256_sra_epi32(SUB(x2, t1), sc);
    r[6] = _mm256_sra_epi32(SUB(x1, t2), sc);
    r[7] = _mm256_sra_epi32(SUB(x0, t3), sc);
}

static inline __m256i clamp_avx(__m256i v) /* src/idct/avx2.rs:402-413 */
{
    return _mm256_min_epi16(_mm256_max_epi16(v, _mm256_setzero_si256()), _mm256_set1_epi16(255));
}

/* src/idct/avx2.rs:64-398 */
int zja_idct_strip_avx2(const int16_t *coeff, size_t n, const int32_t qt[64], size_t stride,
                        size_t samp_factors, size_t v_samp, int16_t *out)
{
    if (samp_factors == 0) return ZJO_ERR_PANIC;
    memset(out, 0, n * 2);
    size_t chunks = n * v_samp / samp_factors;
    if (chunks == 0) return n == 0 ? ZJO_OK : ZJO_ERR_PANIC;
    __m256i q[8];
    for (int k = 0; k < 8; k++) q[k] = _mm256_loadu_si256((const __m256i *)(qt + 8 * k));
    for (size_t c0 = 0; c0 + chunks <= n; c0 += chunks) {
        const int16_t *in_chunk = coeff + c0;
        int16_t *out_chunk = out + c0;
        size_t pos = 0, x = 0;
        for (size_t b0 = 0; b0 + 64 <= chunks; b0 += 64) {
            const int16_t *v = in_chunk + b0;
            __m128i rw[8];
            for (int k = 0; k < 8; k++) rw[k] = _mm_loadu_si128((const __m128i *)(v + 8 * k));
            /* zero test of the 63 AC terms, :114-141 */
            __m128i z = _mm_loadu_si128((const __m128i *)(v + 1));
            z = _mm_insert_epi16(z, 0, 7); /* v[1..8] has 7 lanes; lane 7 (= v[8]) is covered by rw[1] */
            for (int k = 1; k < 8; k++) z = _mm_or_si128(z, rw[k]);
            if (_mm_test_all_zeros(z, z)) {
                int16_t val = (int16_t)((int16_t)((int16_t)(v[0] * (int16_t)qt[0]) >> 3) + 128);
                if (val < 0) val = 0;
                if (val > 255) val = 255; /* clamped here, unlike the scalar arm (:163-167) */
                const __m128i iv = _mm_set1_epi16(val);
                for (int r = 0; r < 8; r++) {
                    if (pos + 8 > chunks) return ZJO_ERR_PANIC;
                    _mm_storeu_si128((__m128i *)(out_chunk + pos), iv);
                    pos += stride;
                }
                x += 8;
                pos = x;
                continue;
            }
            __m256i r[8];
            for (int k = 0; k < 8; k++) r[k] = _mm256_mullo_epi32(_mm256_cvtepi16_epi32(rw[k]), q[k]);
            transpose8(r);
            dct_pass(r, 512, 10); /* rows first (:333-337) */
            transpose8(r);
            dct_pass(r, SCALE_BITS, 17);
            for (int k = 0; k < 8; k += 2) { /* permute_store!, :355-391 */
                __m256i a = _mm256_packs_epi32(r[k], r[k + 1]);
                __m256i b = clamp_avx(a);
                __m256i c = _mm256_permute4x64_epi64(b, 0xD8);
                if (pos + 8 > chunks) return ZJO_ERR_PANIC;
                _mm_storeu_si128((__m128i *)(out_chunk + pos), _mm256_castsi256_si128(c));
                pos += stride;
                if (pos + 8 > chunks) return ZJO_ERR_PANIC;
                _mm_storeu_si128((__m128i *)(out_chunk + pos), _mm256_extracti128_si256(c, 1));
                pos += stride;
            }
            x += 8;
            pos = x;
        }
    }
    return ZJO_OK;
}

/* ---- up-sampling: (3*near + far + 2) >> 2 on 16 lanes ---------------------------------------- */
static inline __m256i tri16(__m256i near_, __m256i far_)
{
    const __m256i three = _mm256_set1_epi16(3), two = _mm256_set1_epi16(2);
    return _mm256_srai_epi16(_mm256_add_epi16(_mm256_add_epi16(_mm256_mullo_epi16(near_, three), far_), two), 2);
}

/* vertical schedule of src/upsampler/scalar.rs:84-144 (the AVX2 arm keeps the same 8-row view, avx2.rs:316) */
int zja_upsample_v(const int16_t *in, size_t n, int16_t *out, size_t out_len)
{
    size_t stride = n >> 3;
    if (stride == 0 || 15 * stride > out_len) return ZJO_ERR_PANIC;
    static const int near_[8] = {0, 1, 2, 3, 4, 5, 6, 7}, far_[8] = {0, 2, 3, 4, 5, 6, 7, 7};
    for (int k = 0; k < 8; k++) {
        const int16_t *a = in + near_[k] * stride, *b = in + far_[k] * stride;
        int16_t *o0 = out + 2 * k * stride, *o1 = o0 + stride;
        size_t cnt = stride;
        if (2 * k * stride + 2 * stride > out_len) cnt = out_len - 2 * k * stride - stride;
        size_t i = 0;
        for (; i + 16 <= cnt; i += 16) {
            __m256i va = _mm256_loadu_si256((const __m256i *)(a + i)), vb = _mm256_loadu_si256((const __m256i *)(b + i));
            _mm256_storeu_si256((__m256i *)(o0 + i), tri16(va, vb));
            _mm256_storeu_si256((__m256i *)(o1 + i), tri16(vb, va));
        }
        for (; i < cnt; i++) {
            o0[i] = (int16_t)((int16_t)(3 * a[i] + b[i] + 2) >> 2);
            o1[i] = (int16_t)((int16_t)(3 * b[i] + a[i] + 2) >> 2);
        }
    }
    return ZJO_OK;
}

/* flat horizontal filter, scalar edge rules (src/upsampler/scalar.rs:5-60), 16 inputs -> 32 outputs per step */
int zja_upsample_h(const int16_t *in, size_t n, int16_t *out, size_t out_len)
{
    if (!(out_len > 4 && n > 2) || out_len < 2 * n) return ZJO_ERR_PANIC;
    size_t i = 1;
    for (; i + 17 <= n; i += 16) { /* needs in[i-1 .. i+16] */
        __m256i c = _mm256_loadu_si256((const __m256i *)(in + i));
        __m256i l = _mm256_loadu_si256((const __m256i *)(in + i - 1));
        __m256i r = _mm256_loadu_si256((const __m256i *)(in + i + 1));
        __m256i e = tri16(c, l), o = tri16(c, r);
        __m256i lo = _mm256_unpacklo_epi16(e, o), hi = _mm256_unpackhi_epi16(e, o);
        _mm256_storeu_si256((__m256i *)(out + 2 * i), _mm256_permute2x128_si256(lo, hi, 0x20));
        _mm256_storeu_si256((__m256i *)(out + 2 * i + 16), _mm256_permute2x128_si256(lo, hi, 0x31));
    }
    for (; i + 1 < n; i++) {
        int16_t s = (int16_t)(3 * in[i] + 2);
        out[2 * i] = (int16_t)((int16_t)(s + in[i - 1]) >> 2);
        out[2 * i + 1] = (int16_t)((int16_t)(s + in[i + 1]) >> 2);
    }
    out[0] = in[0];
    out[1] = (int16_t)((int16_t)(3 * in[0] + in[1] + 2) >> 2);
    out[out_len - 2] = (int16_t)((int16_t)(3 * in[n - 2] + in[n - 1] + 2) >> 2);
    out[out_len - 1] = in[n - 1];
    return ZJO_OK;
}

/* ---- upsample_hv_avx, src/upsampler/avx2.rs:29-342, statement by statement ----------------------------------------
 * The arm the reference selects for (2,2) sampling when use_unsafe is on (src/upsampler.rs:97-112, via upsample_hv_simd
 * :15-23: inputs shorter than 500 samples take the scalar arm).  It fuses the vertical and the horizontal triangle filter
 * over 16 input samples per iteration; the last 16 samples of every input row, and five more outputs per row pair, are
 * patched "manually".  NOT the scalar arm's function and not a parity target -- restated literally for the timed baseline:
 *   - the carried neighbours `prev` / `pixel_far` are (3 * (a + b + 2)) >> 2 after the first iteration (method-call
 *     precedence in :261-267: `3 * (*x).wrapping_add(y).wrapping_add(2)`), not (3a + b + 2) >> 2 as before the loop (:66-67);
 *   - the scalar tail (:282-304) filters the RAW input rows horizontally -- no vertical filter -- into the last v outputs of
 *     both output rows, and ends each with the raw sample input[pos] / input[pos + stride];
 *   - five outputs around the row ends are copies of their neighbours (:331-339).
 * Everywhere else (the interior of every row) its numbers equal upsample_hv's, which tests/test_avx2_baseline.py checks.
 * i16 arithmetic wraps (release build); every slice index / get_mut().unwrap() / assert! of the Rust is a ZJO_ERR_PANIC here. */
static inline __m256i hv_pack_shuffle(__m256i x, int hi_half)      /* pack_shuffle!, :72-81 */
{
    const __m256i v = hi_half ? _mm256_permute2x128_si256(x, x, 0x33) : _mm256_permute2x128_si256(x, x, 0x00);
    const __m256i rwn_hi = _mm256_unpackhi_epi16(v, v), rwn_lo = _mm256_unpacklo_epi16(v, v);
    return _mm256_permute2x128_si256(rwn_lo, rwn_hi, 0x30);
}
static inline int hv_horizontal(__m256i row, int16_t prev, int16_t pixel_far, int16_t *output, size_t out_len, size_t at)
{                                                                   /* upsample_horizontal!, :83-197 */
    const __m256i three = _mm256_set1_epi16(3), two = _mm256_set1_epi16(2);
    __m256i next_arr = _mm256_alignr_epi8(_mm256_permute2x128_si256(row, row, 0x81), row, 2);        /* shuffle(2,0,0,1) */
    __m256i prev_arr = _mm256_alignr_epi8(row, _mm256_permute2x128_si256(row, row, 0x08), 14);       /* shuffle(0,0,2,0) */
    prev_arr = _mm256_insert_epi16(prev_arr, prev, 0);
    next_arr = _mm256_insert_epi16(next_arr, pixel_far, 15);
    const __m256i near_lo = hv_pack_shuffle(row, 0), near_hi = hv_pack_shuffle(row, 1);
    const __m256i prev_lo = hv_pack_shuffle(prev_arr, 0), prev_hi = hv_pack_shuffle(prev_arr, 1);
    const __m256i next_lo = hv_pack_shuffle(next_arr, 0), next_hi = hv_pack_shuffle(next_arr, 1);
    if (at + 32 > out_len) return ZJO_ERR_PANIC;                    /* output.get_mut(..).unwrap(), :176,:192 */
    __m256i nn = _mm256_blend_epi16(prev_lo, next_lo, 0xAA);
    __m256i cn = _mm256_srai_epi16(_mm256_add_epi16(_mm256_mullo_epi16(near_lo, three), _mm256_add_epi16(nn, two)), 2);
    _mm256_storeu_si256((__m256i *)(output + at), cn);
    nn = _mm256_blend_epi16(prev_hi, next_hi, 0xAA);
    cn = _mm256_srai_epi16(_mm256_add_epi16(_mm256_mullo_epi16(near_hi, three), _mm256_add_epi16(nn, two)), 2);
    _mm256_storeu_si256((__m256i *)(output + at + 16), cn);
    return ZJO_OK;
}
int zja_upsample_hv_avx(const int16_t *input, size_t n, int16_t *output, size_t output_len)
{
#define W16(x) ((int16_t)(x))
    memset(output, 0, output_len * sizeof(int16_t));               /* vec![0; output_len], :55 */
    size_t stride = 0, pos = 0, output_position = 0;
    int modify_stride = 1;
    if (n <= 16) return ZJO_ERR_PANIC;                             /* input[pos + 16], :67 */
    int16_t prev = W16(W16(W16(3 * input[0]) + input[stride] + 2) >> 2);                              /* :66 */
    int16_t pixel_far = W16(W16(W16(3 * input[pos + 16]) + input[pos + stride + 16] + 2) >> 2);       /* :67 */
    const __m256i three = _mm256_set1_epi16(3), two = _mm256_set1_epi16(2);
    const size_t len = output_len / 16;                             /* :216 */
    const size_t end = (n >> 7) ? (n >> 7) - 1 : 0;                 /* saturating_sub, :217 */
    const size_t v = output_len / 16 > end * 32 ? output_len / 16 - end * 32 : 0;                     /* :219 */
    for (int j = 0; j < 8; j++) {
        for (size_t it = 0; it < end; it++) {
            if (pos > n || pos + stride > n) return ZJO_ERR_PANIC;  /* input[pos..], input[pos + stride..], :232-233 */
            if (pos + 16 > n || pos + stride + 16 > n) return ZJO_ERR_PANIC; /* (the unchecked 32-byte loads must stay inside) */
            const __m256i load_near = _mm256_loadu_si256((const __m256i *)(input + pos));
            const __m256i load_far = _mm256_loadu_si256((const __m256i *)(input + pos + stride));
            /* upsample_vertical!, :200-213 */
            const __m256i row_near = _mm256_srai_epi16(_mm256_add_epi16(_mm256_mullo_epi16(load_near, three), _mm256_add_epi16(load_far, two)), 2);
            const __m256i row_far = _mm256_srai_epi16(_mm256_add_epi16(_mm256_add_epi16(load_near, two), _mm256_mullo_epi16(load_far, three)), 2);
            int rc = hv_horizontal(row_near, prev, pixel_far, output, output_len, output_position);            /* :248 */
            if (rc == ZJO_OK) rc = hv_horizontal(row_far, prev, pixel_far, output, output_len, output_position + len); /* :250 */
            if (rc) return rc;
            output_position += 32;
            pos += 16;
            if (n < pos + stride + 16) return ZJO_ERR_PANIC;        /* assert!, :259 */
            {                                                       /* :261-267: 3 * (a + b + 2), see the header */
                const int16_t a = pos < n ? input[pos] : 0, b = pos + stride < n ? input[pos + stride] : 0;
                prev = W16(W16(3 * W16(W16(a + b) + 2)) >> 2);
                const int16_t c = pos + 16 < n ? input[pos + 16] : 0, d = pos + stride + 16 < n ? input[pos + stride + 16] : 0;
                pixel_far = W16(W16(3 * W16(W16(c + d) + 2)) >> 2);
            }
        }
        /* the part of the row the vector loop leaves: scalar, :274-304 */
        if (output_position + v > output_len) return ZJO_ERR_PANIC;  /* split_at_mut */
        if (v == 0) return ZJO_ERR_PANIC;                            /* unwritten.last_mut().unwrap() */
        const size_t blen = output_len - (output_position + v);
        if (len > blen || len < v) return ZJO_ERR_PANIC;             /* b[len - v..len] */
        int16_t *unwritten = output + output_position;               /* a[c..], c = a.len() - v */
        int16_t *unwritten_stride = output + output_position + v + (len - v);
        if (pos < v / 2 + 2 || pos > n) return ZJO_ERR_PANIC;        /* input[pos - v/2 - 2..pos] */
        {
            const int16_t *w = input + pos - v / 2 - 2;
            for (size_t k = 0; k + 1 < v; k += 2) { /* windows(3) over v/2 + 2 samples: v/2 windows; chunks_exact_mut(2): v/2 */
                const int16_t sample = W16(W16(3 * w[k / 2 + 1]) + 2);
                unwritten[k] = W16(W16(sample + w[k / 2]) >> 2);
                unwritten[k + 1] = W16(W16(sample + w[k / 2 + 2]) >> 2);
            }
        }
        if (pos >= n) return ZJO_ERR_PANIC;                          /* input[pos] */
        unwritten[v - 1] = input[pos];
        if (pos + stride < v / 2 + 2 || pos + stride > n) return ZJO_ERR_PANIC;
        {
            const int16_t *w = input + pos - v / 2 + stride - 2;
            for (size_t k = 0; k + 1 < v; k += 2) {
                const int16_t sample = W16(W16(3 * w[k / 2 + 1]) + 2);
                unwritten_stride[k] = W16(W16(sample + w[k / 2]) >> 2);
                unwritten_stride[k + 1] = W16(W16(sample + w[k / 2 + 2]) >> 2);
            }
        }
        if (pos + stride >= n) return ZJO_ERR_PANIC;                 /* input[pos + stride] */
        unwritten_stride[v - 1] = input[pos + stride];
        output_position += len + v;
        pos += v / 2;
        if (modify_stride) { stride = n / 8; modify_stride = 0; }   /* :312-317 */
        if (j == 6) stride = 0;                                      /* :318-326 */
        if (output_position > output_len || output_position < len + 4) return ZJO_ERR_PANIC;
        output[output_position - len] = output[output_position - len + 1];       /* :331 (index + 1 may be == output_len) */
        output[output_position - len - 2] = output[output_position - len - 4];
        output[output_position - len - 1] = output[output_position - len - 3];
        output[output_position - 2] = output[output_position - 4];
        output[output_position - 1] = output[output_position - 3];
    }
#undef W16
    return ZJO_OK;
}
/* upsample_hv_simd, src/upsampler/avx2.rs:15-23 */
int zja_upsample_hv_simd(const int16_t *input, size_t n, int16_t *output, size_t output_len)
{
    if (n < 500) return zjo_upsample_hv(input, n, output, output_len);
    return zja_upsample_hv_avx(input, n, output, output_len);
}

/* ---- upsample_horizontal_sse_u, src/upsampler/sse.rs:24-134, statement by statement --------------
 * NOT the same function as the scalar arm: its last eight outputs are written "manually" (sse.rs:113-131) and three of
 * them use other taps than src/upsampler/scalar.rs:5-60 -- out[2n-5] = (3 in[n-3] + in[n-3] + 2) >> 2, out[2n-4] =
 * (3 in[n-2] + in[n-2] + 2) >> 2, out[2n-3] = (3 in[n-2] + in[n-3] + 2) >> 2 where the scalar arm has taps in[n-2],
 * in[n-3], in[n-1].  On unit-step ramps both give the same numbers, which is all the reference's own tests
 * (src/upsampler.rs:126-151) assert; tests/test_oracle.py replays exactly those and shows the difference elsewhere.
 * i16 arithmetic wraps (release build).  Kept out of the timed baseline: the reference never selects it for (2,2). */
int zja_upsample_h_sse(const int16_t *in, size_t n, int16_t *out, size_t out_len)
{
    memset(out, 0, out_len * sizeof(int16_t));                      /* vec![0; output_len]            sse.rs:27 */
    if (!(out_len > 8 && n > 5)) return ZJO_ERR_PANIC;              /* assert!                        sse.rs:33 */
#define W16(x) ((int16_t)(x))
    out[0] = in[0];                                                 /* sse.rs:41-55 */
    out[1] = W16(W16(W16(in[0] * 3) + in[1] + 2) >> 2);
    out[2] = W16(W16(W16(in[1] * 3) + in[0] + 2) >> 2);
    out[3] = W16(W16(W16(in[1] * 3) + in[2] + 2) >> 2);
    out[4] = W16(W16(W16(in[2] * 3) + in[1] + 2) >> 2);
    out[5] = W16(W16(W16(in[2] * 3) + in[3] + 2) >> 2);
    out[6] = W16(W16(W16(in[3] * 3) + in[2] + 2) >> 2);
    out[7] = W16(W16(W16(in[3] * 3) + in[4] + 2) >> 2);
    for (size_t i = 1; i < (n >> 2) - 1; i++) {                     /* sse.rs:69 */
        const size_t pos = i << 2;
        __m128i yn = _mm_loadl_epi64((const __m128i *)(in + pos));
        yn = _mm_unpacklo_epi16(yn, yn);                            /* [a,a,b,b,c,c,d,d] */
        const __m128i v = _mm_loadl_epi64((const __m128i *)(in + pos - 1));
        const __m128i y = _mm_loadl_epi64((const __m128i *)(in + pos + 1));
        const __m128i even = _mm_unpacklo_epi16(v, v), odd = _mm_unpacklo_epi16(y, y);
        const __m128i nn = _mm_blend_epi16(even, odd, 0xAA);        /* sse.rs:85 */
        const __m128i an = _mm_add_epi16(_mm_slli_epi16(yn, 1), yn);/* input[x]*3                     sse.rs:95 */
        const __m128i bn = _mm_add_epi16(nn, _mm_set1_epi16(2));
        const __m128i cn = _mm_srai_epi16(_mm_add_epi16(an, bn), 2);
        if (i * 8 + 8 > out_len) return ZJO_ERR_PANIC;              /* out.get_mut(..).unwrap()       sse.rs:105 */
        _mm_storeu_si128((__m128i *)(out + i * 8), cn);
    }
    {                                                               /* sse.rs:112-131 */
        int16_t *l = out + (out_len - 8);
        const size_t il = n - 4;
        l[0] = W16(W16(W16(in[il] * 3) + in[il - 1] + 2) >> 2);
        l[1] = W16(W16(W16(in[il] * 3) + in[il + 1] + 2) >> 2);
        l[2] = W16(W16(W16(in[il + 1] * 3) + in[il] + 2) >> 2);
        l[3] = W16(W16(W16(in[il + 1] * 3) + in[il + 1] + 2) >> 2);
        l[4] = W16(W16(W16(in[il + 2] * 3) + in[il + 2] + 2) >> 2);
        l[5] = W16(W16(W16(in[il + 2] * 3) + in[il + 1] + 2) >> 2);
        l[6] = W16(W16(W16(in[il + 2] * 3) + in[il + 3] + 2) >> 2);
        l[7] = in[il + 3];
    }
#undef W16
    return ZJO_OK;
}

/* ---- colour: src/color_convert/avx.rs:81-192 --------------------------------------------------- */
static inline void ycbcr_to_rgb_avx2_16(const int16_t *y, const int16_t *cb, const int16_t *cr, uint8_t *o)
{
    const __m256i yc = _mm256_loadu_si256((const __m256i *)y);
    const __m256i cbr = _mm256_sub_epi16(_mm256_loadu_si256((const __m256i *)cb), _mm256_set1_epi16(128));
    const __m256i crr = _mm256_sub_epi16(_mm256_loadu_si256((const __m256i *)cr), _mm256_set1_epi16(128));
    const __m256i r = clamp_avx(_mm256_add_epi16(yc, _mm256_srai_epi16(_mm256_mullo_epi16(_mm256_set1_epi16(45), crr), 5)));
    const __m256i g = clamp_avx(_mm256_sub_epi16(yc, _mm256_srai_epi16(_mm256_add_epi16(_mm256_mullo_epi16(_mm256_set1_epi16(11), cbr),
                                                                                       _mm256_mullo_epi16(_mm256_set1_epi16(23), crr)), 5)));
    const __m256i b = clamp_avx(_mm256_add_epi16(_mm256_srai_epi16(_mm256_mullo_epi16(_mm256_set1_epi16(113), cbr), 6), yc));
    int16_t ra[16], ga[16], ba[16];
    _mm256_storeu_si256((__m256i *)ra, r);
    _mm256_storeu_si256((__m256i *)ga, g);
    _mm256_storeu_si256((__m256i *)ba, b);
    for (int j = 0; j < 16; j++) { /* scalar 3-byte interleave, avx.rs:96-105 */
        o[3 * j] = (uint8_t)ra[j];
        o[3 * j + 1] = (uint8_t)ga[j];
        o[3 * j + 2] = (uint8_t)ba[j];
    }
}

/* ycbcr_to_rgb_avx2 (src/color_convert/avx.rs:81-106) behind the ColorConvert16Ptr contract: 48 bytes at *pos, pos += 48 */
int zja_ycbcr_to_rgb16(const int16_t y[16], const int16_t cb[16], const int16_t cr[16], uint8_t *out, size_t out_len, size_t *pos)
{
    if (*pos + 48 > out_len) return ZJO_ERR_PANIC;                  /* output.get_mut(*pos..*pos + 48).expect  avx.rs:91 */
    ycbcr_to_rgb_avx2_16(y, cb, cr, out + *pos);
    *pos += 48;
    return ZJO_OK;
}

/* worker.rs:143-251 with the AVX2 16-pixel kernel; width >= 16 only (bench geometry) */
static int color_convert_ycbcr_avx2(const int16_t *const blk[3], size_t n, size_t width, size_t hv,
                                    uint8_t *output, size_t out_len)
{
    size_t width_chunk = (n / hv) >> 3, stride = width * 3, start = 0;
    if (width_chunk < 16 || width < 16) return ZJO_ERR_ARG;
    for (size_t c0 = 0; c0 + width_chunk <= n; c0 += width_chunk, start += stride) {
        if (start + stride > out_len) return ZJO_ERR_PANIC;
        uint8_t *out = output + start;
        size_t elements = width_chunk / 16 - 1, position = 0;
        for (size_t g = 0; g < elements; g++, position += 48) {
            if (position + 48 > stride) return ZJO_ERR_PANIC;
            ycbcr_to_rgb_avx2_16(blk[0] + c0 + 16 * g, blk[1] + c0 + 16 * g, blk[2] + c0 + 16 * g, out + position);
        }
        size_t rem = stride > position ? stride - position : 0;
        size_t diff = 64 > rem ? 64 - rem : 0;
        position = position > diff ? position - diff : 0;
        if (position + 48 > stride) return ZJO_ERR_PANIC;
        size_t t0 = c0 + width_chunk - 16;
        ycbcr_to_rgb_avx2_16(blk[0] + t0, blk[1] + t0, blk[2] + t0, out + position);
    }
    return ZJO_OK;
}

/* post_process (worker.rs:32-141), AVX2 arms, YCbCr -> RGB, modes (1,1) and (2,2).
 * The reference allocates a fresh Vec per stage per strip (worker.rs:73, scalar.rs:26/8/80); here each
 * worker thread reuses one scratch set, which only makes this baseline faster. */
typedef struct { int16_t *unp[3], *mid, *up[2]; size_t cap; } scratch_t;

static int scratch_init(scratch_t *s, size_t ylen)
{
    memset(s, 0, sizeof *s);
    s->cap = ylen;
    for (int i = 0; i < 3; i++) s->unp[i] = (int16_t *)aligned_alloc(64, ylen * 2 + 128);
    s->mid = (int16_t *)aligned_alloc(64, ylen * 2 + 128);
    s->up[0] = (int16_t *)aligned_alloc(64, ylen * 2 + 128);
    s->up[1] = (int16_t *)aligned_alloc(64, ylen * 2 + 128);
    return (s->unp[0] && s->unp[1] && s->unp[2] && s->mid && s->up[0] && s->up[1]) ? ZJO_OK : ZJO_ERR_NOMEM;
}
static void scratch_free(scratch_t *s)
{
    for (int i = 0; i < 3; i++) free(s->unp[i]);
    free(s->mid); free(s->up[0]); free(s->up[1]);
}

static int post_process_avx2(scratch_t *sc, const int16_t *const coeff[3], const size_t len[3],
                             const zjo_component comps[3], uint8_t *out, size_t out_len, size_t width)
{
    size_t hs = comps[0].horizontal_sample, vs = comps[0].vertical_sample;
    int rc = ZJO_OK;
    if (len[0] > sc->cap) return ZJO_ERR_ARG;
    for (int z = 0; z < 3 && rc == ZJO_OK; z++)
        rc = zja_idct_strip_avx2(coeff[z], len[z], comps[z].quantization_table, comps[z].width_stride, hs * vs,
                                 z == 0 ? 1 : vs, sc->unp[z]);
    size_t ylen = len[0];
    const int16_t *ch[3] = {sc->unp[0], sc->unp[1], sc->unp[2]};
    if (rc == ZJO_OK && hs == 2 && vs == 2) {
        for (int i = 1; i < 3 && rc == ZJO_OK; i++) { /* the arm choose_hv_samp_function picks, src/upsampler.rs:97-112 */
            rc = zja_upsample_hv_simd(sc->unp[i], len[i], sc->up[i - 1], ylen);
            ch[i] = sc->up[i - 1];
        }
    } else if (rc == ZJO_OK && !(hs == 1 && vs == 1)) {
        rc = ZJO_ERR_ARG;
    }
    if (rc == ZJO_OK) rc = color_convert_ycbcr_avx2(ch, ylen, width, hs * vs, out, out_len);
    return rc;
}

/* ---- strip-per-job thread pool, src/mcu.rs:135,356-368 / src/mcu_prog.rs:196-233 --------------- */
typedef struct {
    const zjo_frame *f;
    const int16_t *y, *cb, *cr;
    uint8_t *out;
    zjo_component comps[3];
    size_t n_strips, y_chunk, c_chunk, chunk_bytes, width, nframes, y_len, c_len, out_len;
    atomic_size_t next;
    atomic_int rc;
} job_t;

static void *worker(void *arg)
{
    job_t *j = (job_t *)arg;
    scratch_t sc;
    if (scratch_init(&sc, j->y_chunk)) { atomic_store(&j->rc, ZJO_ERR_NOMEM); scratch_free(&sc); return 0; }
    for (;;) {
        size_t s = atomic_fetch_add(&j->next, 1);
        if (s >= j->n_strips * j->nframes) break;
        size_t fr = s / j->n_strips, st = s % j->n_strips;
        const int16_t *coeff[3] = {j->y + fr * j->y_len + st * j->y_chunk, j->cb + fr * j->c_len + st * j->c_chunk,
                                   j->cr + fr * j->c_len + st * j->c_chunk};
        size_t len[3] = {j->y_chunk, j->c_chunk, j->c_chunk};
        int rc = post_process_avx2(&sc, coeff, len, j->comps, j->out + fr * j->out_len + st * j->chunk_bytes, j->chunk_bytes, j->width);
        if (rc) atomic_store(&j->rc, rc);
    }
    scratch_free(&sc);
    return 0;
}

/* nframes frames of YCbCr -> RGB, (1,1) or (2,2), height a multiple of the strip height. */
int zja_decode_planes_mt(const zjo_frame *f, size_t nframes, const int16_t *y, const int16_t *cb,
                         const int16_t *cr, uint8_t *out, int nthreads)
{
    if (!f || f->in_components != 3 || f->out_colorspace != ZJO_CS_RGB) return ZJO_ERR_ARG;
    if (!((f->h_max == 1 && f->v_max == 1) || (f->h_max == 2 && f->v_max == 2))) return ZJO_ERR_ARG;
    size_t rows = 8 * f->v_max * (f->h_max == 2 ? 2 : 1);
    if (f->height % rows || f->width % 16) return ZJO_ERR_ARG;
    job_t j;
    memset(&j, 0, sizeof j);
    size_t mcu_x = (f->width + 8 * f->h_max - 1) / (8 * f->h_max);
    for (int c = 0; c < 3; c++) {
        j.comps[c].horizontal_sample = c == 0 ? f->h_max : 1;
        j.comps[c].vertical_sample = c == 0 ? f->v_max : 1;
        j.comps[c].width_stride = j.comps[c].horizontal_sample * mcu_x * 8;
        memcpy(j.comps[c].quantization_table, f->qt[c], 256);
    }
    size_t bias = f->h_max == 2 ? 2 : 1;
    j.f = f; j.y = y; j.cb = cb; j.cr = cr; j.out = out; j.width = f->width; j.nframes = nframes;
    j.y_chunk = mcu_x * 64 * f->v_max * f->h_max * bias;
    j.c_chunk = mcu_x * 64 * bias;
    j.chunk_bytes = (size_t)f->width * 3 * rows;
    j.n_strips = f->height / rows;
    j.y_len = zjo_plane_len(f, 0);
    j.c_len = zjo_plane_len(f, 1);
    j.out_len = (size_t)f->width * f->height * 3;
    atomic_init(&j.next, 0);
    atomic_init(&j.rc, 0);
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    pthread_t th[256];
    for (int t = 0; t < nthreads; t++) pthread_create(&th[t], 0, worker, &j);
    for (int t = 0; t < nthreads; t++) pthread_join(th[t], 0);
    return atomic_load(&j.rc);
}
