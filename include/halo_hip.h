/* halo_hip.h -- C ABI of libhalo_hip.so: HALO's per-pixel hyperbolic acquisition-scoring
 * path as hand-written HIP kernels for gfx950 (MI355X / CDNA4).
 *
 * The reference (paolomandica/HALO) is pure Python on stock PyTorch + geoopt; it has no FFI.
 * Each entry point below replaces the reference call named in its comment (paths relative to
 * the reference repository).  The Python classes in halo_amd/core/ bind these through ctypes
 * (halo_amd/_lib.py); INTEGRATION.md shows the binding a maintainer adds on the reference side.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (tensor.data_ptr()); tensors are dense row-major,
 *     exactly the reference's layouts: logit (B,O,H,W) f32, decoder_out (B,C,H,W) f64|f32,
 *     maps (B,H,W); batch strides are passed in ELEMENTS so a caller may hand in views
 *   - stream is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream);
 *     one call = asynchronous enqueue on that stream, no allocation, no host sync, no global
 *     state => re-entrant and hipGraph-capturable
 *   - scratch comes from the caller: size it with the matching *_workspace_bytes()
 *   - return 0 on success, <0 on error (HALO_E_*); halo_last_error() gives the message of the
 *     calling thread's last failure
 *   - dtype codes: HALO_F32 = 0, HALO_F64 = 1
 */
#ifndef HALO_HIP_H
#define HALO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HALO_ABI_VERSION 8

enum { HALO_F32 = 0, HALO_F64 = 1 };

enum { HALO_OK = 0, HALO_E_ARG = -1, HALO_E_UNSUPPORTED = -2, HALO_E_WORKSPACE = -3, HALO_E_LAUNCH = -4 };

/* unc_type of FloatingRegionScore.forward (core/active/floating_region.py:158-163, 70-92).
 * Every other string the reference accepts ('none', 'hyperbolic', 'certainty', ...) yields a
 * zero map (floating_region.py:84-90) => HALO_UNC_ZEROS. */
enum { HALO_UNC_ENTROPY = 0, HALO_UNC_PIXEL_ENTROPY = 1, HALO_UNC_ORACLE_ACC = 2, HALO_UNC_ZEROS = 3 };

/* pur_type of FloatingRegionScore.forward (floating_region.py:165-202); anything else is the
 * reference's NotImplementedError and must be rejected by the caller. */
/* padding_mode of FloatingRegionScore's two box windows (floating_region.py:49,63: forwarded to nn.Conv2d): what a window tap
 * outside the image reads.  ZEROS (every caller in the reference's tree): nothing, and the purity window counts its in-image
 * taps only; REFLECT / REPLICATE / CIRCULAR: the image pixel torch's F.pad(mode) puts there, so every tap counts. */
enum { HALO_PAD_ZEROS = 0, HALO_PAD_REFLECT = 1, HALO_PAD_REPLICATE = 2, HALO_PAD_CIRCULAR = 3 };
/* The `normalize` argument of the halo_score_maps* calls is a flags word: bit 0 = normalise both maps (the reference's
 * `normalize`), bits 8-9 = HALO_PAD_* << 8.  Passing 0 / 1 keeps its old meaning (zero padding). */
enum { HALO_FLAG_NORMALIZE = 1 };
#define HALO_FLAG_PAD(mode) ((mode) << 8)

enum { HALO_PUR_RIPU = 0, HALO_PUR_ORACLE_RIPU = 1, HALO_PUR_HYPER = 2, HALO_PUR_NONE = 3,
       HALO_PUR_RADIUS = 4, HALO_PUR_EUC_NORM = 5 };

int halo_version(void);
const char *halo_last_error(void);

/* ---- hyperbolic ops: core/utils/hyperbolic.py (arithmetic = geoopt.manifolds.stereographic.math) ---- */

/* HyperMapper.expmap(x, dim) (hyperbolic.py:28-39): project(expmap0(x.double())).
 * x viewed as (outer, C, inner), reduced over C; x_dtype F32|F64; y is f64. */
int halo_expmap0_project(const void *x, int x_dtype, double *y, int64_t outer, int64_t C, int64_t inner,
                         double c, void *stream);

/* HyperMapper.logmap(x) (hyperbolic.py:51-60): project(logmap0(x.double())), f64 in/out. */
int halo_logmap0_project(const double *x, double *y, int64_t outer, int64_t C, int64_t inner, double c,
                         void *stream);

/* HyperMapper.poincare_distance_origin(x, dim) (hyperbolic.py:74-83): geoopt dist0; out dtype = dtype. */
int halo_dist0(const void *x, int dtype, void *out, int64_t outer, int64_t C, int64_t inner, double c,
               void *stream);

/* HyperMapper.poincare_distance(x, y) (hyperbolic.py:62-72): geoopt dist over the last dim, f64. */
int halo_pdist(const double *x, const double *y, double *out, int64_t n, int64_t d, double c, void *stream);

/* HyperMLR.forward / _hyper_logits (hyperbolic.py:120-188): x (B,C,hw) f64, P_MLR/A_MLR (O,C) f64,
 * out (B,O,hw) in out_dtype (F32 fuses the head's .float(), core/models/classifier.py:373,554).
 * workspace: halo_hypermlr_workspace_bytes(O, C). */
size_t halo_hypermlr_workspace_bytes(int64_t O, int64_t C);
int halo_hypermlr_logits(const double *x, const double *P, const double *A, void *out, int out_dtype,
                         int64_t B, int64_t C, int64_t O, int64_t hw, double c, void *workspace,
                         size_t workspace_bytes, void *stream);

/* The inference tail of both hyperbolic heads in one call (classifier.py:364-379: ASPP_Classifier_V2_Hyper.forward, :552-558:
 * DepthwiseSeparableASPP_Hyper.forward):  embed = mapper.expmap(feat, dim=1)  (B,C,hw) f64;  out = conv_seg(embed)[.float()]
 * (B,O,hw) in out_dtype.  feat (B,C,hw) f32.  Returns 1 and enqueues NOTHING when the shape is not served by the fused kernel
 * (C != 64, O > 32, odd hw, unaligned bases): the caller then makes the two calls above, whose results this call reproduces bit
 * for bit.  workspace: halo_hypermlr_workspace_bytes(O, C). */
int halo_head_tail(const float *feat, const double *P, const double *A, double *embed, void *out, int out_dtype, int64_t B,
                   int64_t C, int64_t O, int64_t hw, double c, void *workspace, size_t workspace_bytes, void *stream);

/* Backward of the head tail for training (classifier.py:553-554 under autograd).
 *  - halo_expmap0_project_bwd: gx = J^T gy for y = HyperMapper.expmap(x, dim); gy f64, gx in x's dtype.
 *  - halo_hypermlr_bwd_terms: reverse sweep through _hyper_logits' scalar algebra (hyperbolic.py:146-183):
 *    from gout = dL/dlogit (B,O,hw) f64 it writes dL/dpx, dL/dxa, and the per-element contributions to
 *    dL/dpp, dL/dpa, dL/d||A|| (B,O,hw each) plus dL/dxx summed over classes (B,hw).  The two remaining
 *    contractions (d x = W^T D + 2 x dxx, d W = D x^T) are plain GEMMs done by the caller's BLAS.
 *    workspace: halo_hypermlr_workspace_bytes(O, C). */
int halo_expmap0_project_bwd(const void *x, int x_dtype, const double *gy, void *gx, int64_t outer, int64_t C,
                             int64_t inner, double c, void *stream);
int halo_hypermlr_bwd_terms(const double *x, const double *P, const double *A, const double *gout, int64_t B, int64_t C,
                            int64_t O, int64_t hw, double c, double *dpx, double *dxa, double *dxx, double *dpp,
                            double *dpa, double *dan, void *workspace, size_t workspace_bytes, void *stream);

/* The whole backward of HyperMLR._hyper_logits in one call (ABI 7): from gout = dL/dlogit (B,O,hw), HALO_F64 or HALO_F32 (the head's
 * `.float()` under autograd hands a float32 gradient back: classifier.py:554), it writes
 * gx = dL/dx (B,C,hw), gP = dL/dP_MLR and gA = dL/dA_MLR (O,C), all f64 -- the reverse sweep above, d x = W^T D + 2 x dxx,
 * d W = D x^T and the parameter algebra (||P||^2, <-P,A^>, F.normalize, ||A||; hyperbolic.py:137-174) on the device, every sum
 * in a fixed order (run-to-run identical).  Serves O <= 20 classes and C a multiple of 64 up to 256 (the heads' 19 x 64):
 * halo_hypermlr_backward_workspace_bytes returns 0 for any other shape, and the caller then composes the backward from
 * halo_hypermlr_bwd_terms and its own GEMMs (halo_amd/core/utils/hyperbolic.py does). */
size_t halo_hypermlr_backward_workspace_bytes(int64_t B, int64_t C, int64_t O, int64_t hw);
int halo_hypermlr_backward(const double *x, const double *P, const double *A, const void *gout, int gout_dtype, int64_t B, int64_t C,
                           int64_t O, int64_t hw, double c, double *gx, double *gP, double *gA, void *workspace,
                           size_t workspace_bytes, void *stream);

/* F.interpolate(mode="bilinear", align_corners=True) (core/active/build.py:123-125,133-135;
 * classifier.py:375-377,556-557): planes x (h,w) -> planes x (H,W), dtype F32|F64.  ATen's order (columns first, rows second,
 * every p*q + r*s as fma(p, q, r*s)): the bits torch's CPU kernel returns at the path's shapes. */
int halo_bilinear_upsample(const void *src, void *dst, int dtype, int64_t planes, int64_t h, int64_t w,
                           int64_t H, int64_t W, void *stream);

/* ---- scoring: FloatingRegionScore.forward (core/active/floating_region.py:129-217) ----
 *
 * logit (B,O,H,W) f32; feat = decoder_out (B,C,H,W) f64|f32, needed for pur HYPER/RADIUS/EUC_NORM
 * (may be NULL otherwise); gt (B,H,W) i64, needed for UNC_ORACLE_ACC / PUR_ORACLE_RIPU.
 * ksize  = entropy_conv size (constructor `size`);  pksize = purity_conv size (3 if the module
 * was built with purity_type=='hyper', floating_region.py:54-55);  K = histogram bins for HYPER.
 * Outputs (B,H,W): score and impurity are f64 when pur is RADIUS|EUC_NORM and feat is f64,
 * otherwise f32 (the reference's type promotion, floating_region.py:210); uncertainty is f32.
 * impurity / uncertainty may be NULL when only the score is wanted.
 * active (B,H,W) u8, optional: where non-zero the score is written as -inf, fusing
 * `score[active] = -inf` (core/active/build.py:146); pass NULL for the plain forward.
 */
size_t halo_score_workspace_bytes(int64_t B, int64_t H, int64_t W);
int halo_score_maps(const float *logit, int64_t logit_bstride, const void *feat, int feat_dtype,
                    int64_t feat_bstride, const int64_t *gt, const uint8_t *active, int64_t B, int64_t O,
                    int64_t C, int64_t H, int64_t W, int unc_type, int pur_type, int normalize, int ksize,
                    int pksize, int64_t K, double c, void *score, void *impurity, float *uncertainty,
                    void *workspace, size_t workspace_bytes, void *stream);

/* Same call with two optional hipEvent_t (as void*, from halo_event_create) recorded on `stream`
 * immediately before and after the feature-reduction kernel (k_feat_reduce, the HBM-roofline
 * kernel) -- used by bench.py to time that kernel live inside the pipelined run -- and an optional
 * `score_range` output (a buffer of halo_score_range_bytes(B) bytes for these B maps, NULL = none): the value range of each
 * score map -- and, for normalised maps, the selector's coarse histogram of it -- in the form halo_greedy_select_ranged
 * accepts, so that the selector need not read the map once or twice more to find them.  Free when the maps are normalised
 * (a product of two values in [0, 1]); otherwise the range is reduced exactly and no histogram is handed over. */
int halo_score_maps_timed(const float *logit, int64_t logit_bstride, const void *feat, int feat_dtype,
                          int64_t feat_bstride, const int64_t *gt, const uint8_t *active, int64_t B, int64_t O,
                          int64_t C, int64_t H, int64_t W, int unc_type, int pur_type, int normalize, int ksize,
                          int pksize, int64_t K, double c, void *score, void *impurity, float *uncertainty,
                          void *workspace, size_t workspace_bytes, void *stream, void *ev_feat_start,
                          void *ev_feat_stop, void *score_range);

/* halo_score_maps_timed with its tail on a second stream of the caller: the passes over `logit` and `feat` run on `stream`,
 * ev_feat_stop (required) is recorded behind them, `tail_stream` waits for it and receives every launch after that (min / max,
 * normalize_map, the product, score[active] = -inf: floating_region.py:204-217 + build.py:146).  The small tail kernels of
 * one call then overlap the feature pass of the NEXT call on `stream` instead of standing between two of them.  The outputs are
 * complete on tail_stream; `workspace` belongs to the call until then (one workspace per call in flight).  Needs a purity type
 * that reads decoder_out.  tail_stream == stream or NULL: the same as halo_score_maps_timed. */
int halo_score_maps_split(const float *logit, int64_t logit_bstride, const void *feat, int feat_dtype,
                          int64_t feat_bstride, const int64_t *gt, const uint8_t *active, int64_t B, int64_t O,
                          int64_t C, int64_t H, int64_t W, int unc_type, int pur_type, int normalize, int ksize,
                          int pksize, int64_t K, double c, void *score, void *impurity, float *uncertainty,
                          void *workspace, size_t workspace_bytes, void *stream, void *tail_stream, void *ev_feat_start,
                          void *ev_feat_stop, void *score_range);

/* The same forward on LOW-RESOLUTION sources, fusing RegionSelection's two F.interpolate calls
 * (core/active/build.py:122-135) into the scorer: logit_lr (B,O,hl,wl) f32 and feat_lr (B,C,hf,wf)
 * f64|f32 are interpolated on the fly (bilinear, align_corners=True) to (H,W); the C x H x W tensor is
 * never materialised.  Results are bit-identical to halo_bilinear_upsample + halo_score_maps.
 * Returns HALO_E_UNSUPPORTED when a source window does not fit LDS (strong downsampling): the caller
 * then upsamples explicitly.  workspace: halo_score_lr_workspace_bytes(B, O, H, W). */
size_t halo_score_lr_workspace_bytes(int64_t B, int64_t O, int64_t H, int64_t W);
int halo_score_maps_lr(const float *logit_lr, int64_t logit_bstride, int64_t hl, int64_t wl, const void *feat_lr,
                       int feat_dtype, int64_t feat_bstride, int64_t hf, int64_t wf, const int64_t *gt,
                       const uint8_t *active, int64_t B, int64_t O, int64_t C, int64_t H, int64_t W, int unc_type,
                       int pur_type, int normalize, int ksize, int pksize, int64_t K, double c, void *score,
                       void *impurity, float *uncertainty, void *workspace, size_t workspace_bytes, void *stream);

/* halo_score_maps_lr with the embedding's radius / norm evaluated through the Gram form SURVEY 8f N1 describes:
 * ||sum_i w_i v_i||^2 = sum_{i<=j} (2 - [i==j]) w_i w_j <v_i, v_j> over the four corner vectors of an output pixel's low-res
 * cell -- the inner products are computed once (one pass over feat_lr; 5 maps over the low-res grid hold them), each
 * output pixel then costs 10 terms instead of C.  Same mathematics, different rounding (|difference| of the sum of squares: a few 1e-16 of the
 * largest corner norm^2), so this call is NOT bit-identical to upsample + halo_score_maps; float64 feat_lr only
 * (HALO_E_UNSUPPORTED otherwise).  Everything else (logits, normalisation, -inf masking) as halo_score_maps_lr.
 * workspace: halo_score_lr_gram_workspace_bytes(B, O, H, W, hf, wf). */
size_t halo_score_lr_gram_workspace_bytes(int64_t B, int64_t O, int64_t H, int64_t W, int64_t hf, int64_t wf);
int halo_score_maps_lr_gram(const float *logit_lr, int64_t logit_bstride, int64_t hl, int64_t wl, const void *feat_lr,
                            int feat_dtype, int64_t feat_bstride, int64_t hf, int64_t wf, const int64_t *gt,
                            const uint8_t *active, int64_t B, int64_t O, int64_t C, int64_t H, int64_t W, int unc_type,
                            int pur_type, int normalize, int ksize, int pksize, int64_t K, double c, void *score,
                            void *impurity, float *uncertainty, void *workspace, size_t workspace_bytes, void *stream);

/* Either of the two calls above (gram = 0 / 1) with optional hipEvent_t pairs (void*, from halo_event_create; NULL = none) recorded on
 * `stream` around the logit pass, around the embedding pass (ev_feat_mid: between the Gram pass and the radius pass of the
 * gram route) and behind the tail kernels (ev_tail_stop) -- bench.py times the kernels of the low-resolution boundary live
 * with them. */
int halo_score_maps_lr_timed(const float *logit_lr, int64_t logit_bstride, int64_t hl, int64_t wl, const void *feat_lr,
                             int feat_dtype, int64_t feat_bstride, int64_t hf, int64_t wf, const int64_t *gt,
                             const uint8_t *active, int64_t B, int64_t O, int64_t C, int64_t H, int64_t W, int unc_type,
                             int pur_type, int normalize, int ksize, int pksize, int64_t K, double c, void *score,
                             void *impurity, float *uncertainty, void *workspace, size_t workspace_bytes, void *stream,
                             int gram, void *ev_logit_start, void *ev_logit_stop, void *ev_feat_start, void *ev_feat_stop,
                             void *score_range, void *ev_feat_mid, void *ev_tail_stop);

/* Helper methods of FloatingRegionScore that are public by convention:
 *  - compute_region_uncertainty(unc_type, logit, p, ground_truth) / compute_pixel_entropy(p)
 *    (floating_region.py:70-92,123-127): x (B,O,H,W) f32 holds logits (is_prob=0) or softmax
 *    probabilities (is_prob=1); do_box bit 0 applies the k x k box sum, its bits 8-9 carry the padding mode
 *    (HALO_FLAG_PAD(HALO_PAD_*)); out (B,H,W) f32.  workspace: B*H*W*4 + 256 bytes.
 *  - compute_region_impurity(predict, K) (floating_region.py:112-121): pred (B,H,W) i64 ->
 *    impurity, count (B,H,W) f32 (count may be NULL); pad_mode = HALO_PAD_*.
 *  - quantize_uncert_map(decoder_out) (floating_region.py:94-110): -> pred (B,H,W) i64 in [0,K-1].
 *    workspace: halo_score_workspace_bytes(B,H,W). */
int halo_region_uncertainty(const float *x, int64_t bstride, int is_prob, const int64_t *gt, int64_t B, int64_t O,
                            int64_t H, int64_t W, int unc_type, int ksize, int do_box, float *out, void *workspace,
                            size_t workspace_bytes, void *stream);
int halo_region_impurity(const int64_t *pred, int64_t B, int64_t H, int64_t W, int ksize, int64_t K, float *impurity,
                         float *count, int pad_mode, void *stream);
int halo_quantize_radius(const void *feat, int feat_dtype, int64_t feat_bstride, int64_t B, int64_t C, int64_t H,
                         int64_t W, int64_t K, double c, int64_t *pred, void *workspace, size_t workspace_bytes,
                         void *stream);

/* ---- selection: select_pixels_to_label (core/active/build.py:27-64) ----
 *
 * score (B,H,W) f32|f64 is mutated (windows -> -inf) exactly like the reference; active, selected
 * (B,H,W) u8 (torch.bool storage) and active_mask (B,H,W) i64 are updated in place; gt (B,H,W) i64.
 * Up to n_regions picks per image: repeat { argmax with ties -> smallest w, then smallest h
 * (the reference's two-stage torch.max, build.py:38-43); stop when the max is -inf }.
 * picks (B,n_regions,3) f64 receives (h, w, value) in selection order (may be NULL);
 * n_picked (B) i32 receives the count (may be NULL).
 */
/* method: HALO_SELECT_AUTO runs the value-binned sweep (visit pixels in descending value order, test each
 * against the picks so far -- no per-pick pass over the map) and leaves what it cannot finish (NaN / +inf
 * in the map, large plateaus of exact ties, pick grid larger than LDS, mask radius 0 or above 14) to the serial
 * tile-table kernel on the same stream; HALO_SELECT_SERIAL runs only the latter; HALO_SELECT_BINNED
 * fails with HALO_E_UNSUPPORTED where the sweep does not serve the geometry.  All three give identical results.
 * workspace: halo_select_workspace_bytes(B, H, W, n_regions, mask_radius). */
enum { HALO_SELECT_AUTO = 0, HALO_SELECT_SERIAL = 1, HALO_SELECT_BINNED = 2 };
size_t halo_select_workspace_bytes(int64_t B, int64_t H, int64_t W, int64_t n_regions, int64_t mask_radius);
int halo_greedy_select(void *score, int dtype, int64_t B, int64_t H, int64_t W, int64_t n_regions,
                       int64_t active_radius, int64_t mask_radius, uint8_t *active, uint8_t *selected,
                       int64_t *active_mask, const int64_t *gt, double *picks, int32_t *n_picked,
                       void *workspace, size_t workspace_bytes, int method, void *stream);

/* halo_greedy_select given the value range of each score map (`score_range`: a buffer of halo_score_range_bytes(B) bytes filled
 * by halo_score_maps_timed / halo_score_maps_lr_timed or halo_score_range for the same B maps; NULL = find it here).  The range
 * only has to BOUND the finite values: the binning is monotone, so the picks do not depend on it.  The buffer also holds room
 * for the selector's coarse histogram of each map (2048 counters): for normalised maps the scorer counts it while it writes the
 * score and marks the record; the selector then skips its pass over the map and CLEARS the mark (the buffer is written through
 * the const pointer's storage: the counts describe the map as it was scored and are used once; should the caller have changed
 * the map in between, the sweep hands an exhausted image over to the serial kernel instead of trusting them -- results never
 * depend on the histogram).  halo_score_range computes the exact range records of existing maps (no histogram). */
size_t halo_score_range_bytes(int64_t B);
int halo_score_range(const void *score, int dtype, int64_t B, int64_t H, int64_t W, void *score_range, void *stream);
int halo_greedy_select_ranged(void *score, int dtype, int64_t B, int64_t H, int64_t W, int64_t n_regions,
                              int64_t active_radius, int64_t mask_radius, uint8_t *active, uint8_t *selected,
                              int64_t *active_mask, const int64_t *gt, double *picks, int32_t *n_picked,
                              void *workspace, size_t workspace_bytes, int method, const void *score_range, void *stream);

/* halo_greedy_select_ranged that also REPORTS what the value-binned sweep did with each image: `handover` (B,2) i32, NULL = do
 * not report; row b = {reason, picks the sweep made before it stopped}.  Reason 0 = the sweep finished the image; anything else =
 * the serial kernel continued it from that pick on (results are identical either way -- this is a cost counter: a handed-over
 * image costs milliseconds instead of tens of microseconds). */
enum { HALO_SWEEP_DONE = 0,
       HALO_SWEEP_BAD_VALUES = 1,     /* NaN / +inf in the map, a constant map, or nothing pickable: no value range to bin */
       HALO_SWEEP_BIN_OVERFLOW = 2,   /* a run of candidates too dense for its value bins (a plateau of ties) */
       HALO_SWEEP_SURVIVORS = 3,      /* more unsuppressed candidates in one step than the resolve stage holds */
       HALO_SWEEP_EXHAUSTED = 4,      /* candidates ran out behind a dropped threshold bin / a stale histogram */
       HALO_SWEEP_NOT_RUN = 5 };      /* serial method, or a geometry the sweep does not serve */
int halo_greedy_select_ex(void *score, int dtype, int64_t B, int64_t H, int64_t W, int64_t n_regions,
                          int64_t active_radius, int64_t mask_radius, uint8_t *active, uint8_t *selected,
                          int64_t *active_mask, const int64_t *gt, double *picks, int32_t *n_picked,
                          void *workspace, size_t workspace_bytes, int method, const void *score_range,
                          int32_t *handover, void *stream);

/* ---- pool side of the round: image-wise sharding, ONE all-gather of pick tables (the reference runs the
 * round on rank 0 only, core/train_learners.py:307-326) ----
 *  - halo_pack_pick_tables: picks (B,n_regions,3) f64 + n_picked (B) i32 (as halo_greedy_select writes them) -> B rows of
 *    the int32 wire block of halo_amd/pool.py: per pick (h << 16) | w and the two words of the float64 score, then the
 *    pick count at word 3*n_regions; wire_row_stride >= 3*n_regions + 1, in int32 elements.
 *  - halo_reset_round_state: the loader's round-1 state for n_pixels pixels (core/datasets/cityscapes.py:245-251):
 *    active = selected = False, active_mask = 255.
 *  - halo_undo_picks: the same state restored after a selection whose state WAS the round-1 state, from its pick
 *    table: only the windows select_pixels_to_label wrote (build.py:52-62) are rewritten.
 *  - halo_device_identity: "pci=<bus id> uuid=<32 hex digits>" of a HIP device ordinal, NUL-terminated, len >= 64
 *    (the ranks of a node must hold distinct devices).
 *  (Round 3 also exported HBM probes and a contiguous-range allocator here; ABI 5 moved those measurement aids to
 *  tools/halo_probe.hip -- this header keeps only entry points that replace a reference call, plus halo_event_*.) */
int halo_pack_pick_tables(const double *picks, const int32_t *n_picked, int64_t B, int64_t n_regions, int32_t *wire,
                          int64_t wire_row_stride, void *stream);
int halo_reset_round_state(uint8_t *active, uint8_t *selected, int64_t *active_mask, int64_t n_pixels, void *stream);
int halo_undo_picks(const double *picks, const int32_t *n_picked, int64_t B, int64_t H, int64_t W, int64_t n_regions,
                    int64_t active_radius, int64_t mask_radius, uint8_t *active, uint8_t *selected, int64_t *active_mask,
                    void *stream);
int halo_device_identity(int device, char *buf, size_t len);

/* ---- training-side window losses, float32 tensors, float64 sums on the device ----
 *  - NegativeLearningLoss (core/loss/negative_learning_loss.py:6-16): sums = {sum -mask*log(1-p+1e-6), sum mask},
 *    mask = p < threshold; loss = sums[0]/sums[1].  bwd: gp = gloss * mask / ((1-p+1e-6) * sums[1]).
 *  - LocalConsistentLoss (core/loss/local_consistent_loss.py:5-17 = LocalDiscrepancy + DetectSPBoundary,
 *    core/loss/boundary.py): x (B,O,h,w) logits, label (B,h,w) i64; writes p = softmax(x) and sums =
 *    {sum of the per-pixel discrepancy over boundary pixels with a valid label, their count}; kl: 0 'l1', 1 'kl'.
 *    coef_a/coef_b (B,O,h,w) and mask (B,h,w) bytes are for the backward call (all three NULL when no gradient is needed):
 *    mask = 1 at the selected pixels, and d l/d p, d l/d mean are written AT THOSE PIXELS ONLY (ABI 7: the rest of the two
 *    coefficient maps is left untouched -- allocate, do not clear).  bwd: gx = d (sums[0]/sums[1]) / d x * gloss  (zero when
 *    the selection is empty).
 *  workspace: halo_loss_workspace_bytes(number of pixels or elements). */
size_t halo_loss_workspace_bytes(int64_t n);
int halo_negative_learning_fwd(const float *p, int64_t n, double threshold, double *sums, void *workspace,
                               size_t workspace_bytes, void *stream);
int halo_negative_learning_bwd(const float *p, int64_t n, double threshold, const double *sums, const float *gloss,
                               float *gp, void *stream);
int halo_local_consistent_fwd(const float *x, const int64_t *label, int64_t B, int64_t O, int64_t h, int64_t w, int kl,
                              float *p, double *sums, float *coef_a, float *coef_b, uint8_t *mask, void *workspace,
                              size_t workspace_bytes, void *stream);
int halo_local_consistent_bwd(const float *p, const float *coef_a, const float *coef_b, const uint8_t *mask, int64_t B, int64_t O,
                              int64_t h, int64_t w, const double *sums, const float *gloss, float *gx, void *stream);

/* ---- measurement helpers (HIP events in the same runtime the kernels are launched through) ---- */
void *halo_event_create(void);
int halo_event_record(void *event, void *stream);
int halo_event_elapsed_ms(void *start, void *stop, float *ms);   /* synchronises on `stop` */
int halo_event_destroy(void *event);

#ifdef __cplusplus
}
#endif
#endif /* HALO_HIP_H */
