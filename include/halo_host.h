/* halo_host.h -- C ABI of libhalo_host.so: host-side (CPU, plain C, no HIP) helpers of the acquisition round's persistence
 * step.  The scoring and the selection run on the GPU behind include/halo_hip.h; what remains on the host in the reference too
 * is turning an image's results into its two files (paths relative to the reference repository):
 *
 *   core/active/build.py:58-62    active_mask[window] = ground_truth[window]   around every pick of the round
 *   core/active/build.py:67-68    to_np_array: np.array(tensor.cpu().numpy(), dtype=np.uint8)   (int64 -> uint8 wraps modulo 256)
 *   core/active/build.py:162-164  Image.fromarray(active_mask).save(path_to_mask)               (uint8 mode-L PNG)
 *   core/active/build.py:165-166  torch.save({'active': ..., 'selected': ...}, path_to_indicator)
 *
 * bound with ctypes in halo_amd/_hostlib.py and called by halo_amd.core.active.build.RegionSelection's writer threads; every
 * call releases the interpreter lock for its whole duration.  All pointers are HOST pointers; arrays are dense row-major.
 * Return values: 0 success, -1 bad argument / out of memory, -2 I/O error on the mask file, -3 I/O error on the indicator file
 * (size_t-returning functions: 0 = failure).
 */
#ifndef HALO_HOST_H
#define HALO_HOST_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HALO_HOST_ABI_VERSION 4
int halo_host_version(void);

/* ---- 8-bit greyscale PNG (build.py:162-164).  What must be identical to PIL's file is the DECODED image: filter type 0 on every
 * scanline, one zlib stream holding one fixed-Huffman deflate block whose runs of equal bytes are distance-1 matches (an
 * acquisition mask is 255 almost everywhere).  img: H rows of W bytes, row_stride >= W bytes apart. */
size_t halo_png_gray8_bound(int64_t H, int64_t W);                   /* capacity that halo_png_gray8_encode never exceeds */
size_t halo_png_gray8_encode(const uint8_t *img, int64_t H, int64_t W, int64_t row_stride, uint8_t *out, size_t cap);
int halo_png_gray8_write(const char *path, const uint8_t *img, int64_t H, int64_t W, int64_t row_stride);

/* ---- CRC-32 (IEEE 802.3; zlib.crc32(buf, crc)): PCLMULQDQ folding on x86-64 where available, slicing-by-8 tables otherwise;
 * halo_crc32_mode(1) forces the tables (test switch). */
uint32_t halo_crc32(uint32_t crc, const uint8_t *buf, size_t len);
void halo_crc32_mode(int tables_only);

/* ---- the mask file's pixels from host data and the pick table (build.py:58-62, 67-68; RegionSelection(mask_staging="table")):
 * mask (H, W) uint8 <- the low byte of every origin_mask element, then origin_label's low bytes over the
 * (2 radius + 1)^2 window of every pick, windows clipped at the image borders.  origin_mask / origin_label: integer arrays of
 * 1-, 2-, 4- or 8-byte little-endian elements; picks: k rows (h, w, score) of float64 as halo_greedy_select writes them. */
int halo_compose_mask(uint8_t *mask, const void *origin_mask, int mask_itemsize, const void *origin_label, int label_itemsize,
                      int64_t H, int64_t W, const double *picks, int64_t k, int64_t radius);

/* ---- the indicator file (build.py:165-166): `tpl` is what torch.save wrote ONCE for two bool tensors of this shape (produced and
 * checked with torch.load by halo_amd.core.active.build._IndicatorTemplate); the file is tpl with the n payload bytes of `active` at
 * off_a and of `selected` at off_s, and each payload's CRC-32 stored little-endian at its two field offsets (crc_fields_*[0..1]:
 * central directory, and data descriptor or local header). */
int halo_write_indicator(const char *path, const uint8_t *tpl, size_t tpl_len, const uint8_t *active, const uint8_t *selected, size_t n,
                         size_t off_a, size_t off_s, const uint64_t *crc_fields_a, const uint64_t *crc_fields_s);

/* ---- the round's indicator maps from the maps an image ENTERED the round with and its pick table (build.py:56-59:
 * active[h-R:h+R+1, w-R:w+R+1] = True with R = mask_radius, selected[...] = True with R = radius, around every pick; slices
 * clipped at the borders): what the selection kernel leaves on the device, recomputed on the host so that the 2 x H x W bytes need
 * not be copied back, and the files of a GLOBAL-budget round, written from the kept prefix of each table. */
int halo_compose_indicators(uint8_t *active, uint8_t *selected, const uint8_t *prior_active, const uint8_t *prior_selected, int64_t H,
                            int64_t W, const double *picks, int64_t k, int64_t radius, int64_t mask_radius);

/* ---- both files of one image in one call: halo_compose_mask -> halo_png_gray8_encode -> write, then halo_write_indicator
 * (skipped when tpl is NULL).  compose_mask_radius < 0: `active` / `selected` are the round's RESULTS (copied back from the device);
 * >= 0: they are the maps the image entered the round with and halo_compose_indicators (that mask radius) runs first.
 * Files (here, halo_png_gray8_write, halo_write_indicator): the bytes open(path, "wb") + write would leave, but an existing file is
 * rewritten IN PLACE and cut to length afterwards -- a round rewrites the files of the round before, and their page-cache pages stay. */
int halo_retire_image(const char *path_png, const char *path_indicator, const void *origin_mask, int mask_itemsize,
                      const void *origin_label, int label_itemsize, int64_t H, int64_t W, const double *picks, int64_t k, int64_t radius,
                      const uint8_t *active, const uint8_t *selected, int64_t compose_mask_radius, const uint8_t *tpl, size_t tpl_len,
                      size_t off_a, size_t off_s, const uint64_t *crc_fields_a, const uint64_t *crc_fields_s);

/* halo_compose_mask narrows 64-bit elements with AVX-512 / AVX2 where the host has them; halo_low_bytes_mode(1) forces the scalar
 * loop, (2) at most AVX2, (0) the best available (test switch, like halo_crc32_mode). */
void halo_low_bytes_mode(int mode);

/* The writer functions keep ~8 MB of scratch per calling thread between calls; it is freed when the thread exits, or now by this
 * call (long-lived writer threads that are done for the round). */
void halo_host_thread_release(void);

#ifdef __cplusplus
}
#endif
#endif /* HALO_HOST_H */
