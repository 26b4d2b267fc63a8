/*
 * libssdbox -- C ABI of the MI355X (gfx950) SSD box pipeline.
 *
 * The reference (rao1140427950/face_detection) has no FFI layer: its boundary for this
 * path is the Python module surface of utils/anchor_utils.py and utils/box_utils.py.  Each
 * entry point below names the reference function (file:line) it stands in for.  All
 * pointers named d_* are DEVICE pointers on the context's GPU; everything else is host
 * memory.  All tensors are dense, row-major, float32 unless stated.  Every call returns 0
 * on success or a negative SSDBOX_ERR_* code and never throws or aborts across the ABI;
 * ssdbox_last_error() returns the message of the calling thread's most recent failure.
 *
 * Work is enqueued on the context's stream and is asynchronous with respect to the host
 * unless a function says otherwise.  A context (stream + scratch buffers) must not be used
 * by two threads at once; create one per thread (the reference calls this path from six
 * tf.data worker threads, utils/datasets.py:42,179-180).  Geometry handles are immutable
 * and may be shared between contexts of the same device.
 */
#ifndef SSDBOX_H
#define SSDBOX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SSDBOX_OK                 0
#define SSDBOX_ERR_INVALID       (-1)  /* bad argument (null pointer, size, threshold)      */
#define SSDBOX_ERR_HIP           (-2)  /* a HIP runtime call or kernel launch failed        */
#define SSDBOX_ERR_NOMEM         (-3)  /* scratch allocation failed                         */
#define SSDBOX_ERR_UNSUPPORTED   (-4)  /* valid request outside what the kernels implement  */

#define SSDBOX_MAX_LAYERS          8   /* predictor layers in one anchor geometry           */
#define SSDBOX_MAX_BOXES_PER_CELL 16   /* boxes per feature-map cell                        */
#define SSDBOX_MAX_NMS_LIMIT    4096   /* ranks one top-k + NMS launch pair of the general path takes */
#define SSDBOX_MAX_GT           4096   /* largest ground-truth count per image              */
#define SSDBOX_STATIC_SCRATCH_BYTES (4u << 20) /* fixed per-GPU scratch of ssdbox_match_encode_static */

typedef struct ssdbox_ctx ssdbox_ctx;
typedef struct ssdbox_geom ssdbox_geom;

/* Kernel stages whose device time a context can accumulate (ssdbox_ctx_set_timing). */
enum {
    SSDBOX_STAGE_MATCH = 0,      /* anchors x GT IoU, match, encode (one launch per batch)   */
    SSDBOX_STAGE_FORCE = 1,      /* forced best-anchor-per-GT fix-up                         */
    SSDBOX_STAGE_SELECT = 2,     /* score threshold + candidate compaction over y_pred       */
    SSDBOX_STAGE_TOPK = 3,       /* per-image top-k select, sort, decode of survivors        */
    SSDBOX_STAGE_NMS_MASK = 4,   /* K x K suppression bit matrix                             */
    SSDBOX_STAGE_NMS_SCAN = 5,   /* greedy scan + detection assembly (limit > 1024: whole NMS) */
    SSDBOX_STAGE_ELEMENTWISE = 6,
    SSDBOX_STAGE_SHAPES = 7,     /* unused (the per-box tables are built inside the match launch) */
    SSDBOX_STAGE_LOSS = 8,       /* SSD loss with hard-negative mining (all its launches)    */
    SSDBOX_STAGE_PRUNE = 9,      /* candidate pruning (+ histogram) when most anchors pass     */
    SSDBOX_NUM_STAGES = 10
};

int ssdbox_version(void);
const char *ssdbox_last_error(void);

/* ---- context ---------------------------------------------------------------------- */
/* stream: a hipStream_t owned by the caller (e.g. torch's current stream; NULL is the
 * device's default stream).  With own_stream != 0 the argument is ignored and the context
 * creates and owns a non-blocking stream instead, at the device's highest stream priority. */
int ssdbox_ctx_create(int device, void *stream, int own_stream, ssdbox_ctx **out);
int ssdbox_ctx_destroy(ssdbox_ctx *ctx);
int ssdbox_ctx_set_stream(ssdbox_ctx *ctx, void *stream, int own_stream);
int ssdbox_ctx_synchronize(ssdbox_ctx *ctx);            /* blocks the host */
/* The hipStream_t the context launches on (what a DLPack producer is given as `stream`). */
int ssdbox_ctx_get_stream(ssdbox_ctx *ctx, void **stream);
/* Per-stage device timing with HIP events on the context's stream.  collect() blocks
 * until the recorded launches finish, adds them up into ms[SSDBOX_NUM_STAGES] and
 * launches[SSDBOX_NUM_STAGES] and clears the records. */
int ssdbox_ctx_set_timing(ssdbox_ctx *ctx, int enabled);
int ssdbox_ctx_collect_timing(ssdbox_ctx *ctx, float *ms, int32_t *launches);
/* The same records launch by launch (a profiling aid; the reference has no counterpart): stage,
 * begin and end of up to `capacity` recorded launches in milliseconds after the first recorded
 * launch of `ref`, which may be `ctx` itself or another context of the same device, so that the
 * launches of several contexts in flight can be put on one time axis.  Blocks until both
 * contexts' streams are idle; the records stay until ssdbox_ctx_collect_timing clears them. */
int ssdbox_ctx_collect_timeline(ssdbox_ctx *ctx, ssdbox_ctx *ref, int32_t *stage, float *begin_ms,
                                float *end_ms, int capacity, int32_t *count);

/* ---- buffers: what a host binding needs to exchange arrays without any other GPU library ---- */
/* The reference's callers pass numpy arrays (utils/datasets.py:130-152) and tf.Tensors
 * (inference.py:38-60) and get tf.Tensors back; there is no counterpart of these calls in the
 * reference because TensorFlow owns its memory there.  With them the Python layer needs PyTorch
 * only when the caller passes torch tensors.
 * ssdbox_device_count: GPUs visible to the process (0 when there is none or HIP fails).
 * ssdbox_buf_alloc / _free: device memory on the context's GPU; a freed buffer may be handed out
 *   again by a later ssdbox_buf_alloc of the same context (safe in stream order), everything is
 *   released with the context.
 * ssdbox_copy_h2d: host memory (pageable is fine) -> device through the context's pinned staging
 *   ring, enqueued on the context's stream; returns once h_src may be reused.
 * ssdbox_copy_d2h: device -> host the same way; returns when h_dst holds the data (and with it
 *   everything enqueued on the stream before has finished).  A destination inside a block from
 *   ssdbox_host_alloc is written by the DMA directly, without the staging copy.
 * ssdbox_copy_d2d: device -> device on the context's stream.
 * ssdbox_host_alloc / _free: pinned host memory any GPU of the process can copy to (for result
 *   arrays a binding hands to its caller; the Python layer keeps a pool of them). */
int ssdbox_device_count(void);
int ssdbox_host_alloc(size_t bytes, void **h_out);
int ssdbox_host_free(void *h_ptr);
int ssdbox_buf_alloc(ssdbox_ctx *ctx, size_t bytes, void **d_out);
int ssdbox_buf_free(ssdbox_ctx *ctx, void *d_ptr);
int ssdbox_copy_h2d(ssdbox_ctx *ctx, void *d_dst, const void *h_src, size_t bytes);
int ssdbox_copy_d2h(ssdbox_ctx *ctx, void *h_dst, const void *d_src, size_t bytes);
int ssdbox_copy_d2d(ssdbox_ctx *ctx, void *d_dst, const void *d_src, size_t bytes);
/* DLPack exchange (what tf.experimental.dlpack.to_dlpack / from_dlpack, torch and numpy's
 * __dlpack__ produce and consume).  dl_managed_tensor is a DLManagedTensor* as dlpack.h defines it.
 * ssdbox_dlpack_import: a dense row-major tensor to a device pointer of the context's GPU.  A
 *   tensor in GPU memory (kDLROCM, same device) is used in place (*owned = 0: the caller keeps the
 *   producer alive until the work enqueued with it has finished); a tensor in host memory (kDLCPU,
 *   pinned host) is staged into a new buffer (*owned = 1: release it with ssdbox_buf_free).  The
 *   library never calls the tensor's deleter: inputs stay the caller's.  A borrowed GPU tensor
 *   is read on the context's stream: pass producer_ordered != 0 only when the producer has
 *   ordered its pending work before that stream (obj.__dlpack__(stream=<ssdbox_ctx_get_stream>),
 *   the DLPack stream protocol); with 0 (a capsule of unknown origin, such as
 *   tf.experimental.dlpack.to_dlpack returns) the call blocks until the device is idle first.
 * ssdbox_dlpack_export: wraps a buffer from ssdbox_buf_alloc as a kDLROCM tensor of the given
 *   shape and dtype (DLPack codes: 0 int, 1 uint, 2 float); ownership of the memory moves to the
 *   tensor, whose deleter frees it.  Blocks until the context's stream has finished, so the
 *   consumer may read on any stream.
 * ssdbox_dlpack_delete: calls the deleter of a tensor nobody consumed. */
int ssdbox_dlpack_import(ssdbox_ctx *ctx, void *dl_managed_tensor, int32_t producer_ordered, void **d_ptr,
                         int64_t *n_elems, int32_t *dtype_code, int32_t *dtype_bits, int32_t *owned);
int ssdbox_dlpack_export(ssdbox_ctx *ctx, void *d_ptr, int32_t ndim, const int64_t *shape,
                         int32_t dtype_code, int32_t dtype_bits, void **dl_managed_tensor);
int ssdbox_dlpack_delete(void *dl_managed_tensor);

/* ---- anchor geometry: utils/anchor_utils.py:6-62 ----------------------------------- */
/* Builds the feature-map geometry generate_anchors(config) iterates over, with the same
 * float64 arithmetic (centre = (index + offsets) / fm_size, sqrt of ratios and of
 * s_m * s_{m+1}) rounded once to float32 and clipped to [0,1].  Kernels given a geometry
 * generate anchors on the fly; nothing of size A is stored.
 *   steps[n_layers], scales[n_layers + 1], ratio_counts[n_layers],
 *   ratios_flat[sum(ratio_counts)] (the per-layer ratio lists back to back). */
int ssdbox_geom_create(int device, int img_width, int img_height, int n_layers,
                       const int32_t *steps, const double *scales,
                       const int32_t *ratio_counts, const double *ratios_flat,
                       int two_boxes_for_ar1, double offsets, ssdbox_geom **out);
int ssdbox_geom_destroy(ssdbox_geom *geom);
int64_t ssdbox_geom_num_anchors(const ssdbox_geom *geom);
/* Layer table for callers: fm_w/fm_h/n_box[n_layers] (any may be NULL); returns n_layers. */
int ssdbox_geom_layers(const ssdbox_geom *geom, int32_t *fm_w, int32_t *fm_h, int32_t *n_box);
/* Materialises what generate_anchors returns: d_out (A,4) [cx, cy, w, h]. */
int ssdbox_generate_anchors(ssdbox_ctx *ctx, const ssdbox_geom *geom, float *d_out);

/* ---- training side: utils/box_utils.py:47-98 (compute_target), batched -------------- */
/* For each of `batch` images: IoU of every anchor with the image's gt_count[b] boxes,
 * best GT per anchor (first maximum), best anchor per GT (lowest index) forced onto that
 * anchor in ascending GT order with IoU 1, conf = label of the matched GT where the IoU is
 * not < iou_threshold else 0, locs = encode(anchor, matched GT) for every anchor.
 * Anchors: pass geom (on the fly) or d_anchors (A,4) [cx,cy,w,h]; the other is NULL.
 *   d_gt_boxes (batch,gmax,4) corners, d_gt_labels (batch,gmax), d_gt_count (batch) int32,
 *   each count in [1, gmax].  Outputs: d_conf (batch,A), d_locs (batch,A,4); optional
 *   (may be NULL) d_match_idx (batch,A) int32 and d_match_iou (batch,A): best_gt_idx /
 *   best_gt_iou after the forced scatter (utils/box_utils.py:72-87).
 * A count outside [1, gmax] is a caller error the kernels do not detect (the reference
 * never calls this with zero boxes, utils/datasets.py:84-96); validate on the host. */
int ssdbox_match_encode(ssdbox_ctx *ctx, const ssdbox_geom *geom, const float *d_anchors,
                        int64_t num_anchors, const float *d_gt_boxes,
                        const float *d_gt_labels, const int32_t *d_gt_count, int batch,
                        int gmax, float max_size, float iou_threshold, float *d_conf,
                        float *d_locs, int32_t *d_match_idx, float *d_match_iou);

/* ssdbox_match_encode without any allocation.  No counterpart in the reference: it exists for
 * callers that cannot allocate device memory (code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc).  Same arguments, same results; the kernels' scratch
 * (one 8-byte best-anchor key per image and ground-truth slot) lives in a fixed array of
 * SSDBOX_STATIC_SCRATCH_BYTES that every GPU the library runs on holds until the process ends.
 * A call whose scratch does not fit (batch * gmax * 8 bytes; batch 32 x gmax 256 takes 64 KiB,
 * the array holds batch * gmax up to 524 288) returns SSDBOX_ERR_INVALID.  All calls on one GPU
 * share the array: calls that overlap on one GPU (other streams, other contexts, other threads)
 * race, so serialise them.  Stage timing (ssdbox_ctx_set_timing) is not recorded for this call. */
int ssdbox_match_encode_static(ssdbox_ctx *ctx, const ssdbox_geom *geom, const float *d_anchors,
                               int64_t num_anchors, const float *d_gt_boxes,
                               const float *d_gt_labels, const int32_t *d_gt_count, int batch,
                               int gmax, float max_size, float iou_threshold, float *d_conf,
                               float *d_locs, int32_t *d_match_idx, float *d_match_iou);

/* Same match, emitting the training-side consumer layout of utils/datasets.py:155-159
 * directly: d_y_true (batch,A,6) = [one_hot(int(conf), 2), locs]. */
int ssdbox_match_encode_ytrue(ssdbox_ctx *ctx, const ssdbox_geom *geom,
                              const float *d_anchors, int64_t num_anchors,
                              const float *d_gt_boxes, const float *d_gt_labels,
                              const int32_t *d_gt_count, int batch, int gmax,
                              float max_size, float iou_threshold, float *d_y_true);

/* ---- inference side: inference.py:33-83, batched ------------------------------------ */
/* d_y_pred (batch,A,6) = [softmax_bg, softmax_face, tx, ty, tw, th] (models/ssd.py:95-99).
 * Per image: keep score = y[:,1] > score_threshold (strict), rank by (score desc, anchor
 * index asc), take the first `limit`, decode those boxes (utils/box_utils.py:126-147),
 * greedy NMS with suppression !(IoU <= nms_threshold) (utils/box_utils.py:205-222), then
 * clip to [0,1] and scale x by img_wh[b][0], y by img_wh[b][1] (inference.py:73-77).
 * Outputs: d_dets (batch,limit,6) rows [1, score, xmin, ymin, xmax, ymax] in kept order,
 * zero rows after the count; d_keep_idx (batch,limit) int32 anchor index of each kept box
 * (-1 after the count); d_count (batch) int32.  d_img_wh (batch,2) may be NULL (scale 1).
 * limit must be >= 1 (any value: the outputs are sized by it); nms_threshold must be < 1.  Up to
 * 1024 ranked boxes per image (the reference's callers use 1000, inference.py:60) the suppression
 * matrix is built by the whole chip; above that one workgroup per image runs the reference's loop
 * as written, SSDBOX_MAX_NMS_LIMIT ranks per pair of launches (min(limit, A) ranks in all, the
 * boxes kept so far carried from pair to pair): same result, roughly ten to fifty times the time
 * per 4096 ranks. */
int ssdbox_decode_nms(ssdbox_ctx *ctx, const ssdbox_geom *geom, const float *d_anchors,
                      int64_t num_anchors, const float *d_y_pred, int batch,
                      float score_threshold, float nms_threshold, int limit,
                      const float *d_img_wh, float *d_dets, int32_t *d_keep_idx,
                      int32_t *d_count);

/* The same with the class columns of the head BEFORE the softmax of models/ssd.py:95 (what
 * inference.py:40 would compute with its commented-out tf.math.softmax): d_y_logits
 * (batch,A,6) = [logit_bg, logit_face, tx, ty, tw, th].  The score of an anchor is
 * exp(l1 - m) * (1 / (exp(l0 - m) + exp(l1 - m))) with m = max(l0, l1), each step rounded
 * to float32; everything after that is ssdbox_decode_nms. */
int ssdbox_decode_nms_logits(ssdbox_ctx *ctx, const ssdbox_geom *geom, const float *d_anchors,
                             int64_t num_anchors, const float *d_y_logits, int batch,
                             float score_threshold, float nms_threshold, int limit,
                             const float *d_img_wh, float *d_dets, int32_t *d_keep_idx,
                             int32_t *d_count);

/* utils/box_utils.py:184-222 (compute_nms) on caller-supplied boxes: d_boxes (n,4)
 * corners, d_scores (n).  Writes the kept indices into the input arrays, in kept order,
 * to d_keep (min(limit,n) int32) and their number to d_count (1 int32).  n == 0 yields
 * count 0.  Any limit >= 1 (fast up to min(limit, n) = 1024, the general path above, see
 * ssdbox_decode_nms). */
int ssdbox_compute_nms(ssdbox_ctx *ctx, const float *d_boxes, const float *d_scores,
                       int64_t n, float nms_threshold, int limit, int32_t *d_keep,
                       int32_t *d_count);

/* ---- elementwise box math ----------------------------------------------------------- */
/* utils/box_utils.py:4-18: d_top_left (n,2), d_bot_right (n,2) -> d_area (n). */
int ssdbox_compute_area(ssdbox_ctx *ctx, const float *d_top_left, const float *d_bot_right,
                        int64_t n, float max_size, float *d_area);
/* utils/box_utils.py:21-44: corners (na,4),(nb,4) -> d_iou (na,nb). */
int ssdbox_compute_iou(ssdbox_ctx *ctx, const float *d_boxes_a, int64_t na,
                       const float *d_boxes_b, int64_t nb, float max_size, float *d_iou);
/* utils/box_utils.py:101-123: d_boxes (rows,4) corners with rows a multiple of A (leading
 * batch dims flattened; row r pairs with anchor r % A) -> d_locs (rows,4). */
int ssdbox_encode(ssdbox_ctx *ctx, const ssdbox_geom *geom, const float *d_anchors,
                  int64_t num_anchors, const float *d_boxes, int64_t rows, float v_center,
                  float v_size, float *d_locs);
/* utils/box_utils.py:126-147: d_locs (rows,4) -> d_boxes (rows,4) corners, no clipping. */
int ssdbox_decode(ssdbox_ctx *ctx, const ssdbox_geom *geom, const float *d_anchors,
                  int64_t num_anchors, const float *d_locs, int64_t rows, float v_center,
                  float v_size, float *d_boxes);
/* utils/box_utils.py:150-164 and :167-181 on (n,4) rows. */
int ssdbox_corner_to_center(ssdbox_ctx *ctx, const float *d_in, int64_t n, float *d_out);
int ssdbox_center_to_corner(ssdbox_ctx *ctx, const float *d_in, int64_t n, float *d_out);

/* ---- the data formats either side of the path ------------------------------------------ */

/* Ground truth as the annotation files store it to what ssdbox_match_encode takes, for a
 * batch: utils/datasets.py:131-140 ((x, y, w, h) pixels -> corners, -1, divide by the image
 * size), then the box half of random_shrink (utils/transforms.py:60-63) where d_shrink
 * (batch,4) int32 rows [h2, w2, h_off, w_off] has h2 > 0 (d_shrink may be NULL: no image was
 * shrunk), then the box half of reize_with_pad (utils/transforms.py:28-33) for a
 * target_h x target_w network input.  d_raw_xywh (batch,gmax,4), d_gt_count (batch),
 * d_img_wh (batch,2) int32 original (width, height) -> d_boxes (batch,gmax,4) normalised
 * corners, zero after the count.  float32 operation for operation as the reference's numpy
 * code evaluates them. */
int ssdbox_prepare_gt(ssdbox_ctx *ctx, const float *d_raw_xywh, const int32_t *d_gt_count,
                      const int32_t *d_img_wh, const int32_t *d_shrink, int batch, int gmax,
                      int target_h, int target_w, float *d_boxes);

/* Detection rows of ssdbox_decode_nms (pixels of the original image) to the integer rows of a
 * WiderFace result file, evaluation.py:69-76: corners truncated toward zero, +1, then
 * (xmin, ymin, xmax - xmin, ymax - ymin).  d_dets (batch,limit,6), d_count (batch) ->
 * d_xywh (batch,limit,4) int32 and d_scores (batch,limit), zero after the count. */
int ssdbox_widerface_rows(ssdbox_ctx *ctx, const float *d_dets, const int32_t *d_count, int batch,
                          int limit, int32_t *d_xywh, float *d_scores);

/* SSD loss with hard-negative mining, utils/losses.py:97-181 (SSDLoss.call), for y_true and
 * y_pred of shape (batch, num_anchors, 2 + 4) = [background, face, 4 offsets]:
 * d_total (batch) the per-image loss; d_keep (batch, num_anchors) bytes, 1 where a negative was
 * mined by the batch-wide top-k (:149-157), or NULL; d_grad (batch, num_anchors, 6) the
 * gradient of mean(d_total) with respect to y_pred as TensorFlow's autodiff derives it (mask
 * and counts constant), or NULL.  tf.nn.top_k's order is frozen as: larger value first, equal
 * values by ascending flat index, NaN last. */
int ssdbox_ssd_loss(ssdbox_ctx *ctx, const float *d_y_true, const float *d_y_pred, int batch,
                    int64_t num_anchors, int neg_pos_ratio, int n_neg_min, float alpha,
                    float *d_total, uint8_t *d_keep, float *d_grad);

/* Device self-checks of the building blocks (wave reductions, exp/log against the values
 * in expect_log/expect_exp for the host inputs x_log/x_exp, n each).  Blocks the host.
 * Returns 0 when every check passes. */
int ssdbox_selftest(ssdbox_ctx *ctx, const float *x_log, const float *expect_log,
                    const float *x_exp, const float *expect_exp, int n);

#ifdef __cplusplus
}
#endif
#endif /* SSDBOX_H */
