// Inference side of the SSD box pipeline for a whole batch: score threshold, top-k,
// decode of the survivors, greedy NMS and detection assembly (reference inference.py:33-83
// around utils/box_utils.py:126-147 and :184-222).
//
// Four launches per batch, none of them per image:
//   1. score_select_kernel  streams the (B,A,6) head output once, keeps score > threshold
//      and appends a 64-bit rank key (score order bits, ~anchor index) per survivor to a
//      per-image candidate list (one atomic per 1024 anchors).
//   2. topk_sort_kernel     one workgroup per image: radix-select the `limit` largest keys
//      when there are more than 2048 candidates, bitonic-sort them in LDS (the key order is
//      "score descending, lower index first", tf.argsort/top_k's order), then decode only
//      those boxes -- threshold and top-k depend on scores alone.
//   3. nms_mask_kernel      lower-triangular K x K suppression bits, 64 x 64 per wave,
//      spread over the chip; bit (j, i) says box i (ranked earlier) suppresses box j, i.e.
//      !(IoU <= threshold).
//   4. nms_scan_kernel      one workgroup per image, one thread per ranked box: bands of 64
//      boxes are resolved in order, inside a band by rounds of ballots; kept boxes are
//      compacted in rank order into the detection rows.
#include "ssdbox_internal.h"

namespace {

constexpr int kSelBlock = 256;
constexpr int kSelPerThread = 4;
constexpr int kSelChunk = kSelBlock * kSelPerThread;
constexpr int kLimit = SSDBOX_MAX_NMS_LIMIT;
constexpr int kWords = kLimit / 64;
constexpr int kSortCap = 2048;
constexpr int kTopkBlock = 1024;
constexpr float kNmsMaxSize = 512.0f;      // compute_iou's default inside compute_nms (:205)

// ------------------------------------------------------------------------------ select
__global__ __launch_bounds__(kSelBlock) void score_select_kernel(
    const float *__restrict__ scores, long long img_stride, int elem_stride, int elem_off, int n,
    float thr, int use_thr, unsigned long long *__restrict__ cand, long long cand_stride,
    int *__restrict__ cand_count)
{
    __shared__ int s_cnt[(kSelBlock / 64) * kSelPerThread];
    __shared__ int s_base;
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = t >> 6;
    const int img = blockIdx.y;
    const float *sc = scores + (long long)img * img_stride;
    const int base = blockIdx.x * kSelChunk;

    float s[kSelPerThread];
    unsigned long long bal[kSelPerThread];
#pragma unroll
    for (int q = 0; q < kSelPerThread; ++q) {
        const int i = base + q * kSelBlock + t;
        s[q] = (i < n) ? sc[(long long)i * elem_stride + elem_off] : 0.0f;
    }
#pragma unroll
    for (int q = 0; q < kSelPerThread; ++q) {
        const int i = base + q * kSelBlock + t;
        const bool pass = (i < n) && (use_thr ? (s[q] > thr) : true);   // strict >, NaN fails
        bal[q] = __ballot(pass);
        if (lane == 0) s_cnt[wave * kSelPerThread + q] = __popcll(bal[q]);
    }
    __syncthreads();
    int total = 0;
    int before[kSelPerThread] = {0, 0, 0, 0};
#pragma unroll
    for (int e = 0; e < (kSelBlock / 64) * kSelPerThread; ++e) {
        const int c = s_cnt[e];
#pragma unroll
        for (int q = 0; q < kSelPerThread; ++q)
            if (e < wave * kSelPerThread + q) before[q] += c;
        total += c;
    }
    if (t == 0) s_base = total ? atomicAdd(&cand_count[img], total) : 0;
    __syncthreads();
    unsigned long long *out = cand + (long long)img * cand_stride + s_base;
    const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
    for (int q = 0; q < kSelPerThread; ++q) {
        if ((bal[q] >> lane) & 1ull) {
            const int i = base + q * kSelBlock + t;
            out[before[q] + __popcll(bal[q] & lt)] = rank_key(s[q], (uint32_t)i);
        }
    }
}

// ------------------------------------------------------------------------- top-k + sort
struct BoxSource {
    int mode;                     // 0: decode, anchors from geometry; 1: decode, anchor array; 2: gather
    const GeomDev *geom;
    const float *cx_tab;
    const float *cy_tab;
    const float4 *anchors;
    const float *y_pred;          // (batch, n, 6)
    const float4 *boxes;          // (n, 4), mode 2
    const float *scores;          // (n), mode 2
};

__global__ __launch_bounds__(kTopkBlock) void topk_sort_kernel(
    const unsigned long long *__restrict__ cand, long long cand_stride, int *__restrict__ cand_count,
    int n_per_image, int limit, BoxSource src, float4 *__restrict__ sboxes,
    float *__restrict__ sscores, int *__restrict__ sidx, int *__restrict__ keff)
{
    __shared__ unsigned long long s_keys[kSortCap];
    __shared__ unsigned int s_hist[2048];
    __shared__ unsigned int s_wtot[kTopkBlock / 64];
    __shared__ unsigned int s_sel[3];
    __shared__ int s_count;

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = t >> 6;
    const int img = blockIdx.x;
    const unsigned long long *c = cand + (long long)img * cand_stride;
    const int n = cand_count[img];
    __syncthreads();
    if (t == 0) {
        cand_count[img] = 0;       // the list is consumed: leave the counter ready for the next call
        keff[img] = min(n, limit);
    }
    if (n == 0) return;
    const int K = min(n, limit);

    int m;                         // keys staged in s_keys
    if (n <= kSortCap) {
        for (int i = t; i < n; i += kTopkBlock) s_keys[i] = c[i];
        m = n;
    } else {
        // MSB-first radix select on the unique 64-bit keys: narrow to the bucket holding the
        // K-th largest until everything at or above that bucket fits the sort buffer.
        unsigned long long pmask = 0ull, prefix = 0ull;
        int k_rem = K, n_above = 0;
        for (int pass = 0; pass < 6; ++pass) {
            const int shift = (pass < 5) ? 53 - 11 * pass : 0;
            const unsigned int dmask = (pass < 5) ? 0x7FFu : 0x1FFu;
            s_hist[t] = 0u;
            s_hist[t + kTopkBlock] = 0u;
            __syncthreads();
            for (int i = t; i < n; i += kTopkBlock) {
                const unsigned long long key = c[i];
                if ((key & pmask) == prefix) atomicAdd(&s_hist[(unsigned int)(key >> shift) & dmask], 1u);
            }
            __syncthreads();
            const unsigned int h0 = s_hist[2 * t], h1 = s_hist[2 * t + 1];
            const unsigned int pair = h0 + h1;
            unsigned int incl = pair;             // suffix sum over this wave's higher lanes
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned int v = __shfl_down(incl, o);
                if (lane + o < 64) incl += v;
            }
            if (lane == 0) s_wtot[wave] = incl;
            __syncthreads();
            unsigned int above1 = incl - pair;    // keys in bins above 2t+1
            for (int w = wave + 1; w < kTopkBlock / 64; ++w) above1 += s_wtot[w];
            const unsigned int above0 = above1 + h1;
            if ((int)above1 < k_rem && (int)(above1 + h1) >= k_rem) {
                s_sel[0] = 2 * t + 1;
                s_sel[1] = above1;
                s_sel[2] = h1;
            }
            if ((int)above0 < k_rem && (int)(above0 + h0) >= k_rem) {
                s_sel[0] = 2 * t;
                s_sel[1] = above0;
                s_sel[2] = h0;
            }
            __syncthreads();
            const unsigned int digit = s_sel[0];
            const int bucket = (int)s_sel[2];
            n_above += (int)s_sel[1];
            k_rem -= (int)s_sel[1];
            prefix |= (unsigned long long)digit << shift;
            pmask |= (unsigned long long)dmask << shift;
            if (n_above + bucket <= kSortCap) break;
        }
        if (t == 0) s_count = 0;
        __syncthreads();
        for (int i = t; i < n; i += kTopkBlock) {
            const unsigned long long key = c[i];
            if ((key & pmask) >= prefix) s_keys[atomicAdd(&s_count, 1)] = key;
        }
        __syncthreads();
        m = s_count;
    }

    int P = 2;
    while (P < m) P <<= 1;
    for (int i = m + t; i < P; i += kTopkBlock) s_keys[i] = 0ull;    // below every real key
    __syncthreads();
    // bitonic sort, descending
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = t; i < P; i += kTopkBlock) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const unsigned long long a = s_keys[i], b = s_keys[ixj];
                    const bool desc = ((i & k) == 0);
                    if ((a < b) == desc) {
                        s_keys[i] = b;
                        s_keys[ixj] = a;
                    }
                }
            }
            __syncthreads();
        }
    }

    for (int r = t; r < K; r += kTopkBlock) {
        const uint32_t idx = 0xFFFFFFFFu - (uint32_t)s_keys[r];
        float4 box;
        float score;
        if (src.mode == 2) {
            box = src.boxes[idx];
            score = src.scores[idx];
        } else {
            const float *row = src.y_pred + ((long long)img * n_per_image + idx) * 6;
            score = row[1];
            const float2 lxy = *reinterpret_cast<const float2 *>(row + 2);
            const float2 lwh = *reinterpret_cast<const float2 *>(row + 4);
            const float4 anc = (src.mode == 0) ? geom_anchor(src.geom, src.cx_tab, src.cy_tab, (int)idx)
                                               : src.anchors[idx];
            box = decode_one(anc, make_float4(lxy.x, lxy.y, lwh.x, lwh.y), 0.1f, 0.2f);
        }
        const long long o = (long long)img * kLimit + r;
        sboxes[o] = box;
        sscores[o] = score;
        sidx[o] = (int)idx;
    }
}

// ---------------------------------------------------------------------------- NMS mask
// clip_by_value(hw, 0, max) then product, reference utils/box_utils.py:15-16
SSD_DEV float box_area(float4 b, float max_size)
{
    return clip0(b.z - b.x, max_size) * clip0(b.w - b.y, max_size);
}

__global__ __launch_bounds__(64) void nms_mask_kernel(const float4 *__restrict__ sboxes,
                                                      const int *__restrict__ keff, float thr,
                                                      unsigned long long *__restrict__ mask)
{
    __shared__ float4 s_col[64];
    __shared__ float s_carea[64];
    const int cw = blockIdx.x;         // column word: earlier-ranked boxes i
    const int rb = blockIdx.y;         // row band: boxes j
    const int img = blockIdx.z;
    const int K = keff[img];
    if (cw > rb || rb * 64 >= K) return;
    const int lane = threadIdx.x;
    const float4 *boxes = sboxes + (long long)img * kLimit;
    const int j = rb * 64 + lane;
    const int ci = cw * 64 + lane;
    const float4 cb = (ci < K) ? boxes[ci] : make_float4(0.f, 0.f, 0.f, 0.f);
    s_col[lane] = cb;
    s_carea[lane] = box_area(cb, kNmsMaxSize);
    const float4 jb = (j < K) ? boxes[j] : make_float4(0.f, 0.f, 0.f, 0.f);
    const float jarea = box_area(jb, kNmsMaxSize);
    __syncthreads();
    const bool zero_suppresses = !(0.0f <= thr);
    unsigned long long bits = 0ull;
    if (j < K) {
        const int qn = (cw == rb) ? lane : 64;     // only boxes ranked before j
        for (int q = 0; q < qn; ++q) {
            const float4 ib = s_col[q];
            const float iw = fminf(ib.z, jb.z) - fmaxf(ib.x, jb.x);
            const float ih = fminf(ib.w, jb.w) - fmaxf(ib.y, jb.y);
            bool sup;
            if (iw > 0.0f && ih > 0.0f) {
                const float inter = fminf(iw, kNmsMaxSize) * fminf(ih, kNmsMaxSize);
                const float uni = (s_carea[q] + jarea) - inter;
                const float iou = inter / uni;
                sup = !(iou <= thr);
            } else {
                // intersection 0: IoU is +0 over a positive union, NaN over a zero one
                const float uni = s_carea[q] + jarea;
                sup = (uni > 0.0f) ? zero_suppresses : true;
            }
            bits |= (unsigned long long)sup << q;
        }
        mask[((long long)img * kLimit + j) * kWords + cw] = bits;
    }
}

// ---------------------------------------------------------------------------- NMS scan
__global__ __launch_bounds__(kLimit) void nms_scan_kernel(
    const unsigned long long *__restrict__ mask, const float4 *__restrict__ sboxes,
    const float *__restrict__ sscores, const int *__restrict__ sidx, const int *__restrict__ keff,
    int limit, const float *__restrict__ img_wh, float *__restrict__ dets,
    int *__restrict__ keep_idx, int *__restrict__ count)
{
    __shared__ unsigned long long s_kept[kWords];
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int w_me = t >> 6;
    const int img = blockIdx.x;
    const int K = keff[img];
    const int KW = (K + 63) >> 6;

    unsigned long long row[kWords];
    {
        const ulonglong2 *rp =
            reinterpret_cast<const ulonglong2 *>(mask + ((long long)img * kLimit + t) * kWords);
#pragma unroll
        for (int c2 = 0; c2 < kWords / 2; ++c2) {
            ulonglong2 v = make_ulonglong2(0ull, 0ull);
            if (t < K && 2 * c2 <= w_me) v = rp[c2];
            row[2 * c2] = v.x;
            row[2 * c2 + 1] = (2 * c2 + 1 <= w_me) ? v.y : 0ull;
        }
    }
    if (t < kWords) s_kept[t] = 0ull;
    bool removed = (t >= K);
    bool kept_me = false;
    __syncthreads();

#pragma unroll
    for (int w = 0; w < kWords; ++w) {
        if (w < KW) {
            if (w_me == w) {
                // Greedy order inside the band: a box with no undecided earlier suppressor is
                // kept; whatever a newly kept box suppresses is dropped; repeat.
                const unsigned long long D = row[w];
                unsigned long long U = __ballot(!removed);
                unsigned long long kept = 0ull;
                while (U != 0ull) {
                    const bool mine = (U >> lane) & 1ull;
                    const unsigned long long A = __ballot(mine && (D & U) == 0ull);
                    kept |= A;
                    const unsigned long long S = __ballot(mine && (D & A) != 0ull);
                    U &= ~(A | S);
                }
                kept_me = (kept >> lane) & 1ull;
                removed = !kept_me;
                if (lane == 0) s_kept[w] = kept;
            }
            __syncthreads();
            if (w_me > w && !removed && (row[w] & s_kept[w]) != 0ull) removed = true;
        }
    }
    __syncthreads();

    int before = 0, total = 0;
#pragma unroll
    for (int w = 0; w < kWords; ++w) {
        const int pc = __popcll(s_kept[w]);
        if (w < w_me) before += pc;
        total += pc;
    }
    if (kept_me) {
        const int pos = before + __popcll(s_kept[w_me] & ((1ull << lane) - 1ull));
        const long long o = (long long)img * kLimit + t;
        keep_idx[(long long)img * limit + pos] = sidx[o];
        if (dets != nullptr) {
            // clip to [0,1], scale to the original image, [class, score, box] rows
            // (reference inference.py:73-83)
            const float4 b = sboxes[o];
            const float W = img_wh ? img_wh[2 * img] : 1.0f;
            const float H = img_wh ? img_wh[2 * img + 1] : 1.0f;
            float *d = dets + ((long long)img * limit + pos) * 6;
            d[0] = 1.0f;
            d[1] = sscores[o];
            d[2] = clip0(b.x, 1.0f) * W;
            d[3] = clip0(b.y, 1.0f) * H;
            d[4] = clip0(b.z, 1.0f) * W;
            d[5] = clip0(b.w, 1.0f) * H;
        }
    }
    if (t >= total && t < limit) {
        keep_idx[(long long)img * limit + t] = -1;
        if (dets != nullptr) {
            float *d = dets + ((long long)img * limit + t) * 6;
#pragma unroll
            for (int e = 0; e < 6; ++e) d[e] = 0.0f;
        }
    }
    if (t == 0) count[img] = total;
}

// -------------------------------------------------------------------------------- host
int ensure_detect_scratch(ssdbox_ctx *ctx, int batch, long long n)
{
    int rc = ensure_buffer((void **)&ctx->cand, &ctx->cand_cap, (size_t)batch * (size_t)n,
                           sizeof(unsigned long long), false, ctx->stream);
    if (rc != SSDBOX_OK) return rc;
    rc = ensure_buffer((void **)&ctx->cand_count, &ctx->count_cap, (size_t)batch, sizeof(int), true,
                       ctx->stream);
    if (rc != SSDBOX_OK) return rc;
    if ((size_t)batch > ctx->det_cap) {
        SSD_CHECK_HIP(hipStreamSynchronize(ctx->stream));
        void *old[] = {ctx->sboxes, ctx->sscores, ctx->sidx, ctx->keff, ctx->mask};
        for (void *p : old)
            if (p) SSD_CHECK_HIP(hipFree(p));
        ctx->sboxes = nullptr;
        ctx->sscores = nullptr;
        ctx->sidx = nullptr;
        ctx->keff = nullptr;
        ctx->mask = nullptr;
        ctx->det_cap = 0;
        const size_t cap = (size_t)batch + (size_t)batch / 4;
        hipError_t e = hipMalloc((void **)&ctx->sboxes, cap * kLimit * sizeof(float4));
        if (e == hipSuccess) e = hipMalloc((void **)&ctx->sscores, cap * kLimit * sizeof(float));
        if (e == hipSuccess) e = hipMalloc((void **)&ctx->sidx, cap * kLimit * sizeof(int));
        if (e == hipSuccess) e = hipMalloc((void **)&ctx->keff, cap * sizeof(int));
        if (e == hipSuccess)
            e = hipMalloc((void **)&ctx->mask, cap * kLimit * kWords * sizeof(unsigned long long));
        if (e != hipSuccess) {
            set_error("detection scratch allocation failed: %s", hipGetErrorString(e));
            return SSDBOX_ERR_NOMEM;
        }
        ctx->det_cap = cap;
    }
    if (ctx->count_dirty) {
        SSD_CHECK_HIP(hipMemsetAsync(ctx->cand_count, 0, ctx->count_cap * sizeof(int), ctx->stream));
        ctx->count_dirty = false;
    }
    return SSDBOX_OK;
}

int run_detect(ssdbox_ctx *ctx, const float *scores, long long img_stride, int elem_stride,
               int elem_off, long long n, int batch, float thr, int use_thr, float nms_thr,
               int limit, const BoxSource &src, const float *d_img_wh, float *d_dets,
               int32_t *d_keep, int32_t *d_count)
{
    int rc = ensure_detect_scratch(ctx, batch, n);
    if (rc != SSDBOX_OK) return rc;
    ctx->count_dirty = true;
    {
        StageTimer timer(ctx, SSDBOX_STAGE_SELECT);
        dim3 grid((unsigned)((n + kSelChunk - 1) / kSelChunk), batch);
        hipLaunchKernelGGL(score_select_kernel, grid, dim3(kSelBlock), 0, ctx->stream, scores,
                           img_stride, elem_stride, elem_off, (int)n, thr, use_thr, ctx->cand,
                           (long long)n, ctx->cand_count);
    }
    SSD_CHECK_HIP(hipGetLastError());
    {
        StageTimer timer(ctx, SSDBOX_STAGE_TOPK);
        hipLaunchKernelGGL(topk_sort_kernel, dim3(batch), dim3(kTopkBlock), 0, ctx->stream,
                           ctx->cand, (long long)n, ctx->cand_count, (int)n, limit, src,
                           ctx->sboxes, ctx->sscores, ctx->sidx, ctx->keff);
    }
    SSD_CHECK_HIP(hipGetLastError());
    ctx->count_dirty = false;
    {
        StageTimer timer(ctx, SSDBOX_STAGE_NMS_MASK);
        const int kw = (limit + 63) / 64;
        hipLaunchKernelGGL(nms_mask_kernel, dim3(kw, kw, batch), dim3(64), 0, ctx->stream,
                           ctx->sboxes, ctx->keff, nms_thr, ctx->mask);
    }
    SSD_CHECK_HIP(hipGetLastError());
    {
        StageTimer timer(ctx, SSDBOX_STAGE_NMS_SCAN);
        hipLaunchKernelGGL(nms_scan_kernel, dim3(batch), dim3(kLimit), 0, ctx->stream, ctx->mask,
                           ctx->sboxes, ctx->sscores, ctx->sidx, ctx->keff, limit, d_img_wh, d_dets,
                           d_keep, d_count);
    }
    SSD_CHECK_HIP(hipGetLastError());
    return SSDBOX_OK;
}

}  // namespace

extern "C" int ssdbox_decode_nms(ssdbox_ctx *ctx, const ssdbox_geom *geom, const float *d_anchors,
                                 int64_t num_anchors, const float *d_y_pred, int batch,
                                 float score_threshold, float nms_threshold, int limit,
                                 const float *d_img_wh, float *d_dets, int32_t *d_keep_idx,
                                 int32_t *d_count)
{
    SSD_REQUIRE(ctx != nullptr, "decode_nms: ctx is NULL");
    SSD_REQUIRE((geom != nullptr) != (d_anchors != nullptr),
                "decode_nms: pass exactly one of geom and d_anchors");
    SSD_REQUIRE(d_y_pred && d_dets && d_keep_idx && d_count, "decode_nms: NULL buffer");
    SSD_REQUIRE(batch >= 1 && batch <= 65535, "decode_nms: batch %d outside [1, 65535]", batch);
    SSD_REQUIRE(limit >= 1, "decode_nms: limit %d < 1", limit);
    if (limit > kLimit) {
        set_error("decode_nms: limit %d above the supported %d", limit, kLimit);
        return SSDBOX_ERR_UNSUPPORTED;
    }
    // the reference loop never ends for thresholds >= 1 (a box's IoU with itself is <= thr)
    SSD_REQUIRE(nms_threshold < 1.0f, "decode_nms: nms_threshold must be < 1");
    SSD_REQUIRE(score_threshold == score_threshold, "decode_nms: score_threshold is NaN");
    SSD_REQUIRE(num_anchors >= 1 && num_anchors < (1ll << 30), "decode_nms: num_anchors %lld",
                (long long)num_anchors);
    if (geom != nullptr) {
        SSD_REQUIRE(geom->d_geom != nullptr, "decode_nms: geometry handle is host-only");
        SSD_REQUIRE(geom->device == ctx->device, "decode_nms: geometry on device %d, context on %d",
                    geom->device, ctx->device);
        SSD_REQUIRE(num_anchors == geom->host.num_anchors,
                    "decode_nms: num_anchors %lld does not match the geometry's %d",
                    (long long)num_anchors, geom->host.num_anchors);
    }
    DeviceGuard guard(ctx->device);
    BoxSource src;
    src.mode = geom ? 0 : 1;
    src.geom = geom ? geom->d_geom : nullptr;
    src.cx_tab = geom ? geom->d_cx : nullptr;
    src.cy_tab = geom ? geom->d_cy : nullptr;
    src.anchors = reinterpret_cast<const float4 *>(d_anchors);
    src.y_pred = d_y_pred;
    src.boxes = nullptr;
    src.scores = nullptr;
    return run_detect(ctx, d_y_pred, (long long)num_anchors * 6, 6, 1, num_anchors, batch,
                      score_threshold, 1, nms_threshold, limit, src, d_img_wh, d_dets, d_keep_idx,
                      d_count);
}

extern "C" int ssdbox_compute_nms(ssdbox_ctx *ctx, const float *d_boxes, const float *d_scores,
                                  int64_t n, float nms_threshold, int limit, int32_t *d_keep,
                                  int32_t *d_count)
{
    SSD_REQUIRE(ctx != nullptr, "compute_nms: ctx is NULL");
    SSD_REQUIRE(d_count != nullptr, "compute_nms: d_count is NULL");
    SSD_REQUIRE(n >= 0 && n < (1ll << 30), "compute_nms: n %lld", (long long)n);
    SSD_REQUIRE(limit >= 1, "compute_nms: limit %d < 1", limit);
    SSD_REQUIRE(nms_threshold < 1.0f, "compute_nms: nms_threshold must be < 1");
    DeviceGuard guard(ctx->device);
    if (n == 0) {                      // reference :198-199, empty result
        SSD_CHECK_HIP(hipMemsetAsync(d_count, 0, sizeof(int32_t), ctx->stream));
        return SSDBOX_OK;
    }
    SSD_REQUIRE(d_boxes && d_scores && d_keep, "compute_nms: NULL buffer");
    const int k = (int)((n < (int64_t)limit) ? n : (int64_t)limit);
    if (k > kLimit) {
        set_error("compute_nms: min(limit, n) = %d above the supported %d", k, kLimit);
        return SSDBOX_ERR_UNSUPPORTED;
    }
    BoxSource src;
    src.mode = 2;
    src.geom = nullptr;
    src.cx_tab = nullptr;
    src.cy_tab = nullptr;
    src.anchors = nullptr;
    src.y_pred = nullptr;
    src.boxes = reinterpret_cast<const float4 *>(d_boxes);
    src.scores = d_scores;
    return run_detect(ctx, d_scores, 0, 1, 0, n, 1, 0.0f, 0, nms_threshold, k, src, nullptr,
                      nullptr, d_keep, d_count);
}
