/* libhalo_host.so -- host-side (CPU) helpers of the acquisition round's persistence step.  Plain C, no HIP.
 *
 * halo_png_gray8_*: the uint8 mode-L PNG the reference writes with PIL at the end of every image of a round
 * (core/active/build.py:67-68,162-164: Image.fromarray(np.uint8 mask).save(path_to_mask); read back by
 * core/datasets/cityscapes.py:231).  What must be identical is the DECODED image, not the compressed bytes.
 *
 * An acquisition mask is 255 ("unlabeled") almost everywhere with a few thousand 3x3 windows of class ids, i.e. long runs.
 * PIL's encoder takes ~12 ms per 1024x2048 mask, zlib level 1 with the run-length strategy 3-4.5 ms -- by far the largest
 * CPU item of retiring an image, on hosts where a rank has 2-16 usable cores.  This encoder is written for exactly that
 * data: filter type 0 on every scanline, ONE zlib stream holding ONE fixed-Huffman deflate block, every run of equal bytes
 * sent as its first byte + distance-1 matches of up to 258 bytes.  Runs are found 8 bytes at a time and the Adler-32 of a
 * run is a closed form, so the cost is proportional to the number of runs, not pixels: ~0.3 ms per mask.  Any decoder
 * (libpng, PIL) reads the result; an image without runs still encodes correctly (9 bits per pixel at worst).
 *
 * halo_retire_image: everything the round writes for one image, in ONE call that holds no interpreter lock -- the mask composed
 * from host data and the pick table (the low byte of origin_mask with the labels of the picks' windows, build.py:52-62,67-68),
 * its PNG, and the indicator file as a per-shape template of torch.save's own bytes with the two payloads and their CRC-32
 * fields replaced (halo_amd/core/active/build.py:_IndicatorTemplate).  Eight Python writer threads doing the same through
 * numpy / zlib / torch.save spend ~1 ms per image holding the GIL (torch.save's record writes), which bounded RegionSelection
 * at ~1 ms per image whatever else was improved (profiles/r04_region_selection_timing.txt).
 */
#define _GNU_SOURCE 1            /* writev, ftruncate, O_CLOEXEC under -std=c11 */
#include <errno.h>
#include <fcntl.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <sys/uio.h>
#include <unistd.h>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

#include "../../include/halo_host.h"

int halo_host_version(void) { return HALO_HOST_ABI_VERSION; }

/* ---- CRC-32 (IEEE 802.3: PNG chunks, zip entries).  Slicing-by-8 tables; on x86-64 with PCLMULQDQ the bulk of a long buffer
 * goes through the carry-less-multiplication folding of Gopal et al., "Fast CRC Computation for Generic Polynomials Using
 * PCLMULQDQ" (64 bytes per iteration, ~10 bytes per cycle), which tests/test_abi.py checks against zlib on random lengths. ---- */
static uint32_t crc_table[8][256];
/* The lazily built tables (these, the deflate code tables, the CPU feature flag) are initialised through pthread_once: up to 16
 * writer threads enter together in the first round, and "racing initialisers store the same values" -- what rounds 3-5 relied on
 * with a release / acquire flag -- is still a data race between one thread's stores and another's loads (ThreadSanitizer,
 * tests/test_sanitizers.py). */
static pthread_once_t crc_once = PTHREAD_ONCE_INIT;
static void crc_init(void)
{
    for (uint32_t n = 0; n < 256; ++n) {
        uint32_t c = n;
        for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xedb88320u ^ (c >> 1) : c >> 1;
        crc_table[0][n] = c;
    }
    for (uint32_t n = 0; n < 256; ++n)
        for (int t = 1; t < 8; ++t) crc_table[t][n] = crc_table[0][crc_table[t - 1][n] & 0xffu] ^ (crc_table[t - 1][n] >> 8);
}
/* raw register update (no pre/post inversion) */
static uint32_t crc_tables_raw(uint32_t crc, const uint8_t *p, size_t n)
{
    pthread_once(&crc_once, crc_init);
    while (n >= 8) {
        uint64_t v;
        memcpy(&v, p, 8);
        v ^= crc;
        crc = crc_table[7][v & 0xff] ^ crc_table[6][(v >> 8) & 0xff] ^ crc_table[5][(v >> 16) & 0xff] ^ crc_table[4][(v >> 24) & 0xff] ^
              crc_table[3][(v >> 32) & 0xff] ^ crc_table[2][(v >> 40) & 0xff] ^ crc_table[1][(v >> 48) & 0xff] ^ crc_table[0][v >> 56];
        p += 8; n -= 8;
    }
    while (n--) crc = crc_table[0][(crc ^ *p++) & 0xffu] ^ (crc >> 8);
    return crc;
}
#if defined(__x86_64__)
/* len >= 64 and a multiple of 16 */
__attribute__((target("pclmul,sse4.1"))) static uint32_t crc_clmul_raw(uint32_t crc, const uint8_t *buf, size_t len)
{
    static const uint64_t __attribute__((aligned(16))) k1k2[2] = {0x0154442bd4ull, 0x01c6e41596ull};
    static const uint64_t __attribute__((aligned(16))) k3k4[2] = {0x01751997d0ull, 0x00ccaa009eull};
    static const uint64_t __attribute__((aligned(16))) k5k0[2] = {0x0163cd6124ull, 0x0000000000ull};
    static const uint64_t __attribute__((aligned(16))) poly[2] = {0x01db710641ull, 0x01f7011641ull};
    __m128i x0, x1, x2, x3, x4, x5, x6, x7, x8, y5, y6, y7, y8;
    x1 = _mm_loadu_si128((const __m128i *)(buf + 0x00));
    x2 = _mm_loadu_si128((const __m128i *)(buf + 0x10));
    x3 = _mm_loadu_si128((const __m128i *)(buf + 0x20));
    x4 = _mm_loadu_si128((const __m128i *)(buf + 0x30));
    x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128((int)crc));
    x0 = _mm_load_si128((const __m128i *)k1k2);
    buf += 64; len -= 64;
    while (len >= 64) {                              /* fold four 128-bit lanes over the next 64 bytes */
        x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x6 = _mm_clmulepi64_si128(x2, x0, 0x00);
        x7 = _mm_clmulepi64_si128(x3, x0, 0x00); x8 = _mm_clmulepi64_si128(x4, x0, 0x00);
        x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x2 = _mm_clmulepi64_si128(x2, x0, 0x11);
        x3 = _mm_clmulepi64_si128(x3, x0, 0x11); x4 = _mm_clmulepi64_si128(x4, x0, 0x11);
        y5 = _mm_loadu_si128((const __m128i *)(buf + 0x00)); y6 = _mm_loadu_si128((const __m128i *)(buf + 0x10));
        y7 = _mm_loadu_si128((const __m128i *)(buf + 0x20)); y8 = _mm_loadu_si128((const __m128i *)(buf + 0x30));
        x1 = _mm_xor_si128(_mm_xor_si128(x1, x5), y5); x2 = _mm_xor_si128(_mm_xor_si128(x2, x6), y6);
        x3 = _mm_xor_si128(_mm_xor_si128(x3, x7), y7); x4 = _mm_xor_si128(_mm_xor_si128(x4, x8), y8);
        buf += 64; len -= 64;
    }
    x0 = _mm_load_si128((const __m128i *)k3k4);      /* fold the four lanes into one */
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x3), x5);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x4), x5);
    while (len >= 16) {                              /* remaining 16-byte blocks */
        x2 = _mm_loadu_si128((const __m128i *)buf);
        x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
        buf += 16; len -= 16;
    }
    x2 = _mm_clmulepi64_si128(x1, x0, 0x10);         /* 128 -> 64 bits */
    x3 = _mm_setr_epi32(~0, 0, ~0, 0);
    x1 = _mm_srli_si128(x1, 8);
    x1 = _mm_xor_si128(x1, x2);
    x0 = _mm_loadl_epi64((const __m128i *)k5k0);
    x2 = _mm_srli_si128(x1, 4);
    x1 = _mm_and_si128(x1, x3);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x00);
    x1 = _mm_xor_si128(x1, x2);
    x0 = _mm_load_si128((const __m128i *)poly);      /* Barrett reduction to 32 bits */
    x2 = _mm_and_si128(x1, x3);
    x2 = _mm_clmulepi64_si128(x2, x0, 0x10);
    x2 = _mm_and_si128(x2, x3);
    x2 = _mm_clmulepi64_si128(x2, x0, 0x00);
    x1 = _mm_xor_si128(x1, x2);
    return (uint32_t)_mm_extract_epi32(x1, 1);
}
static int clmul_state;
static pthread_once_t clmul_once = PTHREAD_ONCE_INIT;
static void clmul_probe(void) { clmul_state = (__builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1")) ? 1 : 0; }
static int have_clmul(void)
{
    pthread_once(&clmul_once, clmul_probe);
    return clmul_state;
}
#endif
static int crc_force_tables = 0;                     /* test switch: halo_crc32_mode(1) */
void halo_crc32_mode(int tables_only) { crc_force_tables = tables_only; }
static uint32_t crc32_update(uint32_t crc, const uint8_t *p, size_t n)
{
    crc = ~crc;
#if defined(__x86_64__)
    if (n >= 256 && !crc_force_tables && have_clmul()) {
        const size_t bulk = n & ~(size_t)15;
        crc = crc_clmul_raw(crc, p, bulk);
        p += bulk; n -= bulk;
    }
#endif
    crc = crc_tables_raw(crc, p, n);
    return ~crc;
}
/* zlib.crc32(buf, crc) */
uint32_t halo_crc32(uint32_t crc, const uint8_t *buf, size_t len) { return buf ? crc32_update(crc, buf, len) : crc; }

static void put_be32(uint8_t *p, uint32_t v) { p[0] = (uint8_t)(v >> 24); p[1] = (uint8_t)(v >> 16); p[2] = (uint8_t)(v >> 8); p[3] = (uint8_t)v; }

/* ---- deflate bit writer (LSB first) ---- */
typedef struct { uint8_t *p, *end; uint64_t acc; int nbits; int overflow; } bitw_t;

static inline void bw_flush_bytes(bitw_t *w)
{
    while (w->nbits >= 8) {
        if (w->p < w->end) *w->p++ = (uint8_t)w->acc; else w->overflow = 1;
        w->acc >>= 8;
        w->nbits -= 8;
    }
}
static inline void bw_put(bitw_t *w, uint32_t bits, int n)      /* n <= 24 */
{
    w->acc |= (uint64_t)bits << w->nbits;
    w->nbits += n;
    if (w->nbits >= 32) {                                        /* four whole bytes at once (little-endian hosts) */
        if (w->p + 4 <= w->end) { const uint32_t lo = (uint32_t)w->acc; memcpy(w->p, &lo, 4); w->p += 4; } else w->overflow = 1;
        w->acc >>= 32;
        w->nbits -= 32;
    }
}
static inline uint32_t rev_bits(uint32_t v, int n)
{
    uint32_t r = 0;
    for (int i = 0; i < n; ++i) { r = (r << 1) | (v & 1u); v >>= 1; }
    return r;
}

/* fixed Huffman code of RFC 1951 3.2.6, already bit-reversed for the LSB-first writer */
static uint16_t lit_code[288];
static uint8_t lit_len[288];
static uint16_t len_sym[259];        /* match length 3..258 -> length symbol */
static uint8_t len_xbits[259];
static uint16_t len_xval[259];
static uint32_t match_d1_code[259];  /* the whole "L bytes at distance 1" token: length code, extra bits, distance code 0 -- <= 18 bits */
static uint8_t match_d1_bits[259];
static pthread_once_t huff_once = PTHREAD_ONCE_INIT;
static void huff_init(void)
{
    for (int s = 0; s < 288; ++s) {
        uint32_t code; int n;
        if (s < 144) { code = 0x30u + (uint32_t)s; n = 8; }
        else if (s < 256) { code = 0x190u + (uint32_t)(s - 144); n = 9; }
        else if (s < 280) { code = (uint32_t)(s - 256); n = 7; }
        else { code = 0xc0u + (uint32_t)(s - 280); n = 8; }
        lit_code[s] = (uint16_t)rev_bits(code, n);
        lit_len[s] = (uint8_t)n;
    }
    static const int base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const int xb[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    for (int L = 3; L <= 258; ++L) {
        int k = 28;
        while (base[k] > L) --k;
        if (L == 258) k = 28;
        len_sym[L] = (uint16_t)(257 + k);
        len_xbits[L] = (uint8_t)xb[k];
        len_xval[L] = (uint16_t)(L - base[k]);
        const int s_ = 257 + k;
        uint32_t code = lit_code[s_];
        int nb = lit_len[s_];
        code |= (uint32_t)(L - base[k]) << nb; nb += xb[k];
        nb += 5;                                                 /* distance code 0 (5 zero bits) = distance 1, no extra bits */
        match_d1_code[L] = code;
        match_d1_bits[L] = (uint8_t)nb;
    }
}
static inline void put_literal(bitw_t *w, unsigned v) { bw_put(w, lit_code[v], lit_len[v]); }
static inline void put_match_d1(bitw_t *w, int L)              /* L bytes repeating the previous byte */
{
    bw_put(w, match_d1_code[L], match_d1_bits[L]);               /* length symbol + extra bits + distance code 0 in one token */
}
/* a run of n >= 1 equal bytes v whose first byte has NOT been sent yet */
static inline void put_run(bitw_t *w, unsigned v, size_t n)
{
    put_literal(w, v);
    --n;
    while (n >= 3) {
        size_t L = n > 258 ? 258 : n;
        if (n - L == 1 || n - L == 2) L = n - 3 >= 3 ? (n - 3 > 258 ? 258 : n - 3) : L;   /* leave a tail a match can still take */
        put_match_d1(w, (int)L);
        n -= L;
    }
    while (n--) put_literal(w, v);
}

#if defined(__x86_64__)
static int avx2_state;
static pthread_once_t avx2_once = PTHREAD_ONCE_INIT;
static void avx2_probe(void) { avx2_state = __builtin_cpu_supports("avx2") ? 1 : 0; }
static int have_avx2(void) { pthread_once(&avx2_once, avx2_probe); return avx2_state; }
/* number of leading bytes of p[0..n) equal to v, counted in whole 32-byte steps plus the position inside the first step that differs
 * (the caller finishes the last < 32 bytes) */
__attribute__((target("avx2"))) static size_t run_scan_avx2(const uint8_t *p, size_t n, uint8_t v)
{
    const __m256i pat = _mm256_set1_epi8((char)v);
    size_t j = 0;
    while (j + 32 <= n) {
        const unsigned m = (unsigned)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_loadu_si256((const __m256i *)(p + j)), pat));
        if (m != 0xffffffffu) return j + (size_t)__builtin_ctz(~m);
        j += 32;
    }
    return j;
}
#endif

#define ADLER_MOD 65521u
/* Adler-32 over a run of n bytes of value v, on UNREDUCED 64-bit sums: a <- a + n v, b <- b + n a + v n (n + 1) / 2.  The two
 * remainders are taken only when a sum nears 2^62 and once at the end -- three 64-bit divisions per run were a third of the encoder's
 * time on a mask with 20 000 labelled pixels (round 6). */
static inline void adler_run(uint64_t *a, uint64_t *b, unsigned v, size_t n)
{
    while (n) {
        const uint64_t m = n > (1u << 16) ? (1u << 16) : n;
        *b += m * *a + (uint64_t)v * (m * (m + 1) / 2);          /* a < 2^40, m <= 2^16: every term < 2^57 */
        *a += m * v;
        n -= (size_t)m;
        if ((*b >> 62) || (*a >> 40)) { *a %= ADLER_MOD; *b %= ADLER_MOD; }
    }
}

/* upper bound of the encoded size: 9 bits per byte of the filtered stream + framing */
size_t halo_png_gray8_bound(int64_t H, int64_t W)
{
    if (H <= 0 || W <= 0) return 0;
    const size_t raw = (size_t)H * ((size_t)W + 1);
    return raw + raw / 8 + 256;
}

/* img: H rows of W bytes, row_stride bytes apart.  Returns the number of bytes written to out (0: bad argument or cap too
 * small). */
size_t halo_png_gray8_encode(const uint8_t *img, int64_t H, int64_t W, int64_t row_stride, uint8_t *out, size_t cap)
{
    if (!img || !out || H <= 0 || W <= 0 || row_stride < W || H > 0x7fffffff || W > 0x7fffffff) return 0;
    if (cap < halo_png_gray8_bound(H, W)) return 0;
    pthread_once(&huff_once, huff_init);
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
    uint8_t *p = out;
    memcpy(p, sig, 8); p += 8;
    /* IHDR */
    put_be32(p, 13); memcpy(p + 4, "IHDR", 4);
    put_be32(p + 8, (uint32_t)W); put_be32(p + 12, (uint32_t)H);
    p[16] = 8; p[17] = 0; p[18] = 0; p[19] = 0; p[20] = 0;      /* bit depth 8, colour type 0 (greyscale), deflate, filter 0, no interlace */
    put_be32(p + 21, crc32_update(0, p + 4, 17));
    p += 25;
    /* IDAT: length patched afterwards */
    uint8_t *idat = p;
    memcpy(p + 4, "IDAT", 4);
    p += 8;
    *p++ = 0x78; *p++ = 0x01;                                    /* zlib header: deflate, 32 KiB window, no preset dictionary */
    bitw_t w = {p, out + cap - 32, 0, 0, 0};                /* room for the Adler-32, the chunk CRC and IEND */
    bw_put(&w, 1u, 1);                                           /* BFINAL */
    bw_put(&w, 1u, 2);                                           /* BTYPE = 01: fixed Huffman codes */
    uint64_t a = 1, b = 0;
#if defined(__x86_64__)
    const int avx2 = have_avx2();
#endif
    for (int64_t y = 0; y < H; ++y) {
        const uint8_t *row = img + (size_t)y * (size_t)row_stride;
        /* the scanline's filter byte (type 0) joins a leading run of zeros, if the row starts with one */
        size_t i = 0, n = (size_t)W;
        size_t run0 = 1;
        while (i < n && row[i] == 0) { ++i; ++run0; }
        put_run(&w, 0u, run0);
        adler_run(&a, &b, 0u, run0);
        while (i < n) {
            const unsigned v = row[i];
            size_t j = i + 1;
            const uint64_t pat = 0x0101010101010101ull * v;
            if (j < n && row[j] != v) goto found;          /* a single byte: most runs inside a labelled window */
#if defined(__x86_64__)
            if (j + 32 <= n && avx2) {                     /* long runs (an unlabelled mask is one value): 32 bytes per compare */
                const size_t adv = run_scan_avx2(row + j, n - j, (uint8_t)v);
                j += adv;
                if (j < n && row[j] != v) goto found;
            }
#endif
            while (j + 8 <= n) {
                uint64_t x;
                memcpy(&x, row + j, 8);
                x ^= pat;
                if (x) { j += (size_t)(__builtin_ctzll(x) >> 3); goto found; }
                j += 8;
            }
            while (j < n && row[j] == v) ++j;
        found:
            put_run(&w, v, j - i);
            adler_run(&a, &b, v, j - i);
            i = j;
        }
        if (w.overflow) return 0;
    }
    put_literal(&w, 256);                                        /* end of block */
    w.nbits = (w.nbits + 7) & ~7;                                /* pad to a byte */
    bw_flush_bytes(&w);
    if (w.overflow) return 0;
    p = w.p;
    put_be32(p, (uint32_t)(((b % ADLER_MOD) << 16) | (a % ADLER_MOD))); p += 4;   /* Adler-32 of the filtered stream */
    const uint32_t idat_len = (uint32_t)(p - (idat + 8));
    put_be32(idat, idat_len);
    put_be32(p, crc32_update(0, idat + 4, 4 + (size_t)idat_len)); p += 4;
    put_be32(p, 0); memcpy(p + 4, "IEND", 4); put_be32(p + 8, crc32_update(0, p + 4, 4)); p += 12;
    return (size_t)(p - out);
}

/* encode + write the file; 0 on success, -1 bad argument / out of memory, -2 I/O error */
/* A whole file from `cnt` pieces: the bytes of open(path, "wb") + write + close, produced WITHOUT truncating first.  The acquisition
 * rewrites the same mask and indicator files round after round (build.py:162-166; the indicator keeps its length, 4.3 MB per
 * 1024 x 2048 image): truncation hands the file's page-cache pages back only for the write to allocate them again -- 1.7 -> 1.0 ms on
 * tmpfs, 2.2 -> 0.6 ms on an overlay file system for 4.3 MB in the build container.  Pieces gathered by writev (no stdio copy); the
 * length is cut to `total` afterwards when the old file was longer.  0, or -2 on any I/O error. */
static int write_pieces(const char *path, struct iovec *iov, int cnt, size_t total)
{
    const int fd = open(path, O_WRONLY | O_CREAT | O_CLOEXEC, 0666);
    if (fd < 0) return -2;
    int i = 0, rc = 0;
    while (i < cnt) {
        if (iov[i].iov_len == 0) { ++i; continue; }
        const ssize_t w = writev(fd, iov + i, cnt - i);
        if (w < 0) {
            if (errno == EINTR) continue;
            rc = -2;
            break;
        }
        size_t adv = (size_t)w;
        while (i < cnt && adv >= iov[i].iov_len) { adv -= iov[i].iov_len; ++i; }
        if (i < cnt) { iov[i].iov_base = (char *)iov[i].iov_base + adv; iov[i].iov_len -= adv; }
    }
    struct stat st;
    if (rc == 0 && fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && (size_t)st.st_size != total && ftruncate(fd, (off_t)total) != 0) rc = -2;
    if (close(fd) != 0) rc = -2;
    return rc;
}
static int write_file(const char *path, const uint8_t *buf, size_t n)
{
    struct iovec v = {(void *)buf, n};
    return write_pieces(path, &v, 1, n);
}

int halo_png_gray8_write(const char *path, const uint8_t *img, int64_t H, int64_t W, int64_t row_stride)
{
    const size_t cap = halo_png_gray8_bound(H, W);
    if (!path || cap == 0) return -1;
    uint8_t *buf = (uint8_t *)malloc(cap);
    if (!buf) return -1;
    const size_t n = halo_png_gray8_encode(img, H, W, row_stride, buf, cap);
    int rc = -1;
    if (n) rc = write_file(path, buf, n);
    free(buf);
    return rc;
}

/* Per-thread scratch that is allocated once and kept: a retired image needs ~8 MB of temporaries (mask, PNG stream, two indicator
 * maps), and malloc serves blocks of that size by mmap / munmap -- every call then faults its pages in again and takes the process-wide
 * mapping lock, which is what eight writer threads spent most of their time waiting for (round 5: 6.7 ms per image alone, 38-64 ms
 * per image and thread with 4-8 threads in the build container; the pieces themselves add up to 3 ms). */
static __thread uint8_t *tl_scratch[2];
static __thread size_t tl_cap[2];
/* The blocks go back when the thread exits (RegionSelection's writer pool lives for one call: ~8 MB per writer thread and call
 * stayed allocated before -- ADVICE r5): a process-wide pthread key whose destructor frees the exiting thread's two blocks. */
static pthread_key_t tl_key;
static pthread_once_t tl_key_once = PTHREAD_ONCE_INIT;
static void thread_scratch_free(void *arg)
{
    (void)arg;
    for (int i = 0; i < 2; ++i) { free(tl_scratch[i]); tl_scratch[i] = 0; tl_cap[i] = 0; }
}
static void thread_scratch_key(void) { pthread_key_create(&tl_key, thread_scratch_free); }
static uint8_t *thread_scratch(int which, size_t n)
{
    if (tl_cap[which] < n) {
        pthread_once(&tl_key_once, thread_scratch_key);
        pthread_setspecific(tl_key, (void *)1);                 /* non-NULL: the destructor runs for this thread */
        free(tl_scratch[which]);
        tl_scratch[which] = (uint8_t *)malloc(n);
        tl_cap[which] = tl_scratch[which] ? n : 0;
    }
    return tl_scratch[which];
}
/* frees the calling thread's scratch now (long-lived threads that are done writing) */
void halo_host_thread_release(void) { thread_scratch_free(0); }

/* ---- one image's files from host data, the pick table and the device's indicator maps ---- */
static inline uint8_t low_byte_at(const void *src, int itemsize, size_t i)
{
    return ((const uint8_t *)src)[i * (size_t)itemsize];           /* little-endian hosts (x86-64, the GPU boxes) */
}
#if defined(__x86_64__)
/* int64 -> low bytes, the loader's 16.8 MB `origin_mask` per 1024 x 2048 image (core/datasets/cityscapes.py:234 hands the uint8 PNG
 * as int64): one vpmovqb per 8 elements where the host has AVX-512 (every GPU box of the pool: EPYC 9004 / Xeon), else two shuffles and
 * a permute per 8 under AVX2.  The loop is then bound by the one streaming read of the source (round 5's scalar loop ran at half of it). */
__attribute__((target("avx512f,avx512bw,avx512vl"))) static void low_bytes_q_avx512(uint8_t *dst, const uint64_t *s, size_t n)
{
    size_t i = 0;
    for (; i + 32 <= n; i += 32) {
        const __m128i a = _mm512_cvtepi64_epi8(_mm512_loadu_si512((const void *)(s + i)));
        const __m128i b = _mm512_cvtepi64_epi8(_mm512_loadu_si512((const void *)(s + i + 8)));
        const __m128i c = _mm512_cvtepi64_epi8(_mm512_loadu_si512((const void *)(s + i + 16)));
        const __m128i d = _mm512_cvtepi64_epi8(_mm512_loadu_si512((const void *)(s + i + 24)));
        _mm_storeu_si128((__m128i *)(dst + i), _mm_unpacklo_epi64(a, b));
        _mm_storeu_si128((__m128i *)(dst + i + 16), _mm_unpacklo_epi64(c, d));
    }
    for (; i < n; ++i) dst[i] = (uint8_t)s[i];
}
__attribute__((target("avx2"))) static void low_bytes_q_avx2(uint8_t *dst, const uint64_t *s, size_t n)
{
    /* byte 0 of each 64-bit lane to the low bytes of each 128-bit half, then the two halves side by side */
    const __m256i pick = _mm256_setr_epi8(0, 8, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, 0, 8, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1);
    size_t i = 0;
    for (; i + 16 <= n; i += 16) {
        uint16_t h[8];
        for (int j = 0; j < 4; ++j) {
            const __m256i v = _mm256_shuffle_epi8(_mm256_loadu_si256((const __m256i *)(s + i + 4 * j)), pick);
            h[2 * j] = (uint16_t)_mm256_extract_epi16(v, 0);
            h[2 * j + 1] = (uint16_t)_mm256_extract_epi16(v, 8);
        }
        memcpy(dst + i, h, 16);
    }
    for (; i < n; ++i) dst[i] = (uint8_t)s[i];
}
static int low_bytes_isa;                             /* 2: AVX-512, 1: AVX2, 0: scalar */
static pthread_once_t low_bytes_once = PTHREAD_ONCE_INIT;
static void low_bytes_probe(void)
{
    low_bytes_isa = (__builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("avx512vl")) ? 2
                    : (__builtin_cpu_supports("avx2") ? 1 : 0);
}
#endif
static int low_bytes_force = 0;                       /* test switch, halo_low_bytes_mode: 0 best available, 1 scalar, 2 at most AVX2 */
void halo_low_bytes_mode(int mode) { low_bytes_force = mode; }
static void low_bytes(uint8_t *dst, const void *src, int itemsize, size_t n)
{
    if (itemsize == 1) { memcpy(dst, src, n); return; }
#if defined(__x86_64__)
    if (itemsize == 8 && low_bytes_force != 1) {
        pthread_once(&low_bytes_once, low_bytes_probe);
        if (low_bytes_isa == 2 && low_bytes_force != 2) { low_bytes_q_avx512(dst, (const uint64_t *)src, n); return; }
        if (low_bytes_isa >= 1) { low_bytes_q_avx2(dst, (const uint64_t *)src, n); return; }
    }
#endif
    if (itemsize == 8) { const uint64_t *s = (const uint64_t *)src; for (size_t i = 0; i < n; ++i) dst[i] = (uint8_t)s[i]; return; }
    if (itemsize == 4) { const uint32_t *s = (const uint32_t *)src; for (size_t i = 0; i < n; ++i) dst[i] = (uint8_t)s[i]; return; }
    if (itemsize == 2) { const uint16_t *s = (const uint16_t *)src; for (size_t i = 0; i < n; ++i) dst[i] = (uint8_t)s[i]; return; }
    for (size_t i = 0; i < n; ++i) dst[i] = low_byte_at(src, itemsize, i);
}

/* mask (H, W) uint8 <- the low byte of every origin_mask element, then origin_label's low bytes over the
 * (2 radius + 1)^2 window of every pick (rows of `picks`: h, w, score as float64), windows clipped at the borders */
int halo_compose_mask(uint8_t *mask, const void *origin_mask, int mask_itemsize, const void *origin_label, int label_itemsize,
                      int64_t H, int64_t W, const double *picks, int64_t k, int64_t radius)
{
    if (!mask || !origin_mask || H <= 0 || W <= 0 || k < 0 || radius < 0 || (k > 0 && (!picks || !origin_label))) return -1;
    if ((mask_itemsize != 1 && mask_itemsize != 2 && mask_itemsize != 4 && mask_itemsize != 8) ||
        (k > 0 && label_itemsize != 1 && label_itemsize != 2 && label_itemsize != 4 && label_itemsize != 8)) return -1;
    low_bytes(mask, origin_mask, mask_itemsize, (size_t)H * (size_t)W);
    for (int64_t p = 0; p < k; ++p) {
        const int64_t h = (int64_t)picks[3 * p], w = (int64_t)picks[3 * p + 1];
        const int64_t y0 = h - radius < 0 ? 0 : h - radius, y1 = h + radius >= H ? H - 1 : h + radius;
        const int64_t x0 = w - radius < 0 ? 0 : w - radius, x1 = w + radius >= W ? W - 1 : w + radius;
        for (int64_t y = y0; y <= y1; ++y)
            for (int64_t x = x0; x <= x1; ++x) mask[y * W + x] = low_byte_at(origin_label, label_itemsize, (size_t)(y * W + x));
    }
    return 0;
}

/* active / selected (H, W) bool bytes after a round, from the maps the image ENTERED the round with and the round's pick table
 * (build.py:56-59: active[mask-radius window] = True, selected[radius window] = True around every pick; windows clipped at the
 * borders): the device's results without copying them back.  Out-of-place; prior_* may equal the outputs. */
int halo_compose_indicators(uint8_t *active, uint8_t *selected, const uint8_t *prior_active, const uint8_t *prior_selected, int64_t H,
                            int64_t W, const double *picks, int64_t k, int64_t radius, int64_t mask_radius)
{
    if (!active || !selected || !prior_active || !prior_selected || H <= 0 || W <= 0 || k < 0 || radius < 0 || mask_radius < 0 ||
        (k > 0 && !picks)) return -1;
    const size_t n = (size_t)H * (size_t)W;
    if (active != prior_active) memcpy(active, prior_active, n);
    if (selected != prior_selected) memcpy(selected, prior_selected, n);
    for (int64_t p = 0; p < k; ++p) {
        const int64_t h = (int64_t)picks[3 * p], w = (int64_t)picks[3 * p + 1];
        for (int which = 0; which < 2; ++which) {
            const int64_t r = which ? radius : mask_radius;
            uint8_t *dst = which ? selected : active;
            const int64_t y0 = h - r < 0 ? 0 : h - r, y1 = h + r >= H ? H - 1 : h + r;
            const int64_t x0 = w - r < 0 ? 0 : w - r, x1 = w + r >= W ? W - 1 : w + r;
            if (x1 < x0) continue;
            for (int64_t y = y0; y <= y1; ++y) memset(dst + y * W + x0, 1, (size_t)(x1 - x0 + 1));
        }
    }
    return 0;
}

/* The indicator file: `tpl` (tpl_len bytes: what torch.save wrote for two bool tensors of this shape) with the n payload bytes
 * of `active` at off_a and of `selected` at off_s, and each payload's CRC-32 stored (little-endian) at its two field offsets
 * (zip data descriptor / local header, and central directory). */
int halo_write_indicator(const char *path, const uint8_t *tpl, size_t tpl_len, const uint8_t *active, const uint8_t *selected, size_t n,
                         size_t off_a, size_t off_s, const uint64_t *crc_fields_a, const uint64_t *crc_fields_s)
{
    if (!path || !tpl || !active || !selected || !crc_fields_a || !crc_fields_s || off_a + n > tpl_len || off_s + n > tpl_len) return -1;
    if (!(off_a + n <= off_s || off_s + n <= off_a)) return -1;           /* the two payloads must not overlap */
    /* the file = template with two holes: only the ~1.5 KB around the payloads is copied (and patched), the payloads are
     * written straight from the caller's buffers */
    const int a_first = off_a < off_s;
    const size_t o1 = a_first ? off_a : off_s, o2 = a_first ? off_s : off_a;
    const uint8_t *p1 = a_first ? active : selected, *p2 = a_first ? selected : active;
    const size_t small_len = tpl_len - 2 * n;
    uint8_t *small = (uint8_t *)malloc(small_len ? small_len : 1);
    if (!small) return -1;
    memcpy(small, tpl, o1);                                               /* [0, o1) */
    memcpy(small + o1, tpl + o1 + n, o2 - (o1 + n));                      /* [o1 + n, o2) */
    memcpy(small + (o2 - n), tpl + o2 + n, tpl_len - (o2 + n));           /* [o2 + n, end) */
    const uint32_t ca = crc32_update(0, active, n), cs = crc32_update(0, selected, n);
    for (int i = 0; i < 2; ++i)
        for (int which = 0; which < 2; ++which) {
            const uint64_t f = which ? crc_fields_s[i] : crc_fields_a[i];
            const uint32_t c = which ? cs : ca;
            if (f + 4 > tpl_len || (f + 4 > o1 && f < o1 + n) || (f + 4 > o2 && f < o2 + n)) { free(small); return -1; }   /* inside a payload?! */
            const size_t g = f < o1 ? f : (f < o2 ? f - n : f - 2 * n);   /* template offset -> offset in `small` */
            for (int b = 0; b < 4; ++b) small[g + b] = (uint8_t)(c >> (8 * b));
        }
    struct iovec v[5] = {{small, o1}, {(void *)p1, n}, {small + o1, o2 - (o1 + n)}, {(void *)p2, n}, {small + (o2 - n), tpl_len - (o2 + n)}};
    const int rc = write_pieces(path, v, 5, tpl_len);
    free(small);
    return rc;
}

/* mask PNG (composed as halo_compose_mask) + indicator file (halo_write_indicator; skipped when tpl is NULL) of one image.
 * 0 on success, -1 bad argument / out of memory, -2 I/O error on the mask, -3 I/O error on the indicator. */
int halo_retire_image(const char *path_png, const char *path_indicator, const void *origin_mask, int mask_itemsize,
                      const void *origin_label, int label_itemsize, int64_t H, int64_t W, const double *picks, int64_t k, int64_t radius,
                      const uint8_t *active, const uint8_t *selected, int64_t compose_mask_radius, const uint8_t *tpl, size_t tpl_len,
                      size_t off_a, size_t off_s, const uint64_t *crc_fields_a, const uint64_t *crc_fields_s)
{
    if (!path_png || H <= 0 || W <= 0) return -1;
    if (compose_mask_radius >= 0 && tpl && path_indicator) {
        /* active / selected are the maps the image entered the round with: the round's windows are added here */
        const size_t n2 = (size_t)H * (size_t)W;
        uint8_t *ind = thread_scratch(1, 2 * n2);
        if (!ind) return -1;
        int rc2 = halo_compose_indicators(ind, ind + n2, active, selected, H, W, picks, k, radius, compose_mask_radius);
        if (rc2 == 0)
            rc2 = halo_retire_image(path_png, path_indicator, origin_mask, mask_itemsize, origin_label, label_itemsize, H, W, picks, k, radius,
                                    ind, ind + n2, -1, tpl, tpl_len, off_a, off_s, crc_fields_a, crc_fields_s);
        return rc2;
    }
    const size_t n = (size_t)H * (size_t)W, cap = halo_png_gray8_bound(H, W);
    uint8_t *mask = thread_scratch(0, n + cap);
    if (!mask) return -1;
    int rc = halo_compose_mask(mask, origin_mask, mask_itemsize, origin_label, label_itemsize, H, W, picks, k, radius);
    if (rc == 0) {
        const size_t m = halo_png_gray8_encode(mask, H, W, W, mask + n, cap);
        rc = m ? write_file(path_png, mask + n, m) : -1;
    }
    if (rc != 0) return rc;
    if (tpl && path_indicator) {
        rc = halo_write_indicator(path_indicator, tpl, tpl_len, active, selected, n, off_a, off_s, crc_fields_a, crc_fields_s);
        if (rc == -2) rc = -3;
    }
    return rc;
}
