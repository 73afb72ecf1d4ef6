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
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define HALO_HOST_ABI 1

int halo_host_version(void) { return HALO_HOST_ABI; }

/* ---- CRC-32 (IEEE 802.3, as PNG chunks use it), byte-wise table ---- */
static uint32_t crc_table[256];
static int crc_ready = 0;
static void crc_init(void)
{
    for (uint32_t n = 0; n < 256; ++n) {
        uint32_t c = n;
        for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xedb88320u ^ (c >> 1) : c >> 1;
        crc_table[n] = c;
    }
    crc_ready = 1;
}
static uint32_t crc32_update(uint32_t crc, const uint8_t *p, size_t n)
{
    if (!crc_ready) crc_init();                     /* idempotent: a race between threads writes the same values */
    crc = ~crc;
    for (size_t i = 0; i < n; ++i) crc = crc_table[(crc ^ p[i]) & 0xffu] ^ (crc >> 8);
    return ~crc;
}

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
    if (w->nbits >= 32) bw_flush_bytes(w);
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
static int huff_ready = 0;
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
    }
    huff_ready = 1;
}
static inline void put_literal(bitw_t *w, unsigned v) { bw_put(w, lit_code[v], lit_len[v]); }
static inline void put_match_d1(bitw_t *w, int L)              /* L bytes repeating the previous byte */
{
    const unsigned s = len_sym[L];
    bw_put(w, lit_code[s], lit_len[s]);
    if (len_xbits[L]) bw_put(w, len_xval[L], len_xbits[L]);
    bw_put(w, 0u, 5);                                            /* distance code 0 = distance 1, no extra bits */
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

#define ADLER_MOD 65521u
static inline void adler_run(uint32_t *a, uint32_t *b, unsigned v, size_t n)
{
    while (n) {
        const uint64_t m = n > (1u << 20) ? (1u << 20) : n;
        const uint64_t a0 = *a;
        *b = (uint32_t)((*b + m * a0 + (uint64_t)v * (m * (m + 1) / 2 % ADLER_MOD)) % ADLER_MOD);
        *a = (uint32_t)((a0 + m * v) % ADLER_MOD);
        n -= (size_t)m;
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
    if (!huff_ready) huff_init();
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
    uint32_t a = 1, b = 0;
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
    put_be32(p, (b << 16) | a); p += 4;                          /* Adler-32 of the filtered stream */
    const uint32_t idat_len = (uint32_t)(p - (idat + 8));
    put_be32(idat, idat_len);
    put_be32(p, crc32_update(0, idat + 4, 4 + (size_t)idat_len)); p += 4;
    put_be32(p, 0); memcpy(p + 4, "IEND", 4); put_be32(p + 8, crc32_update(0, p + 4, 4)); p += 12;
    return (size_t)(p - out);
}

/* encode + write the file; 0 on success, -1 bad argument / out of memory, -2 I/O error */
int halo_png_gray8_write(const char *path, const uint8_t *img, int64_t H, int64_t W, int64_t row_stride)
{
    const size_t cap = halo_png_gray8_bound(H, W);
    if (!path || cap == 0) return -1;
    uint8_t *buf = (uint8_t *)malloc(cap);
    if (!buf) return -1;
    const size_t n = halo_png_gray8_encode(img, H, W, row_stride, buf, cap);
    int rc = -1;
    if (n) {
        FILE *f = fopen(path, "wb");
        rc = -2;
        if (f) {
            const size_t wr = fwrite(buf, 1, n, f);
            if (fclose(f) == 0 && wr == n) rc = 0;
        }
    }
    free(buf);
    return rc;
}
