// The data formats either side of the box pipeline:
//   * ground truth as WiderFace stores it, (x, y, w, h) in pixels of the original image, to the
//     normalised corner boxes compute_target takes (reference utils/datasets.py:131-140, the box
//     half of random_shrink utils/transforms.py:60-63 and of reize_with_pad :28-33);
//   * detections [class, score, xmin, ymin, xmax, ymax] in pixels to the integer
//     (x, y, w, h) rows of a WiderFace result file (reference evaluation.py:69-76).
// Both are a few float32 operations per box, in the reference's order (numpy float32 arrays
// against Python / int32 scalars: every operation rounds to float32 once).
#include "ssdbox_internal.h"

namespace {

constexpr int kPB = 256;

// shrink: (B, 4) int32 rows [h2, w2, h_off, w_off] of random_shrink, h2 <= 0 = not shrunk
__global__ __launch_bounds__(kPB) void gt_prepare_kernel(const float4 *__restrict__ raw,
                                                         const int *__restrict__ count,
                                                         const int *__restrict__ img_wh,
                                                         const int *__restrict__ shrink, int gmax,
                                                         int target_h, int target_w,
                                                         float4 *__restrict__ out)
{
    const int img = blockIdx.y;
    const int g = blockIdx.x * kPB + threadIdx.x;
    if (g >= gmax) return;
    const long long o = (long long)img * gmax + g;
    if (g >= count[img]) {
        out[o] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        return;
    }
    const int w1 = img_wh[2 * img], h1 = img_wh[2 * img + 1];
    const float4 b = raw[o];
    // (xmin, ymin, w, h) to (xmin, ymin, xmax, ymax), then -1: utils/datasets.py:133-135
    float x0 = b.x, y0 = b.y;
    float x1 = b.x + b.z, y1 = b.y + b.w;
    x0 = x0 - 1.0f;
    y0 = y0 - 1.0f;
    x1 = x1 - 1.0f;
    y1 = y1 - 1.0f;
    // normalise: utils/datasets.py:137-140
    const float fw = (float)w1, fh = (float)h1;
    x0 = x0 / fw;
    x1 = x1 / fw;
    y0 = y0 / fh;
    y1 = y1 / fh;
    if (shrink != nullptr && shrink[4 * img] > 0) {
        // utils/transforms.py:60-63
        const float h2 = (float)shrink[4 * img], w2 = (float)shrink[4 * img + 1];
        const float h_off = (float)shrink[4 * img + 2], w_off = (float)shrink[4 * img + 3];
        x0 = (x0 * w2 + w_off - 1.0f) / fw;
        y0 = (y0 * h2 + h_off - 1.0f) / fh;
        x1 = (x1 * w2 + w_off - 1.0f) / fw;
        y1 = (y1 * h2 + h_off - 1.0f) / fh;
    }
    // utils/transforms.py:28-33: the comparison is Python float (double) arithmetic, the
    // integer products are exact before they become float32
    if ((double)h1 / (double)w1 < (double)target_h / (double)target_w) {
        const float den = (float)((long long)target_h * w1), num = (float)((long long)h1 * target_w);
        y0 = (y0 - 0.5f) / den * num + 0.5f;
        y1 = (y1 - 0.5f) / den * num + 0.5f;
    } else {
        const float den = (float)((long long)target_w * h1), num = (float)((long long)w1 * target_h);
        x0 = (x0 - 0.5f) / den * num + 0.5f;
        x1 = (x1 - 0.5f) / den * num + 0.5f;
    }
    out[o] = make_float4(x0, y0, x1, y1);
}

// evaluation.py:69-76: truncate the pixel corners, +1, corners to (x, y, w, h)
__global__ __launch_bounds__(kPB) void wider_rows_kernel(const float *__restrict__ dets,
                                                         const int *__restrict__ count, int limit,
                                                         int4 *__restrict__ xywh,
                                                         float *__restrict__ scores)
{
    const int img = blockIdx.y;
    const int r = blockIdx.x * kPB + threadIdx.x;
    if (r >= limit) return;
    const long long o = (long long)img * limit + r;
    if (r >= count[img]) {
        xywh[o] = make_int4(0, 0, 0, 0);
        scores[o] = 0.0f;
        return;
    }
    const float2 *d = reinterpret_cast<const float2 *>(dets + o * 6);
    const float2 cs = d[0], a = d[1], b = d[2];
    // astype(int) truncates toward zero; values outside the int32 range saturate here
    const float lim = 2147483520.0f;
    const int x0 = (int)fminf(fmaxf(a.x, -lim), lim) + 1, y0 = (int)fminf(fmaxf(a.y, -lim), lim) + 1;
    const int x1 = (int)fminf(fmaxf(b.x, -lim), lim) + 1, y1 = (int)fminf(fmaxf(b.y, -lim), lim) + 1;
    xywh[o] = make_int4(x0, y0, x1 - x0, y1 - y0);
    scores[o] = cs.y;
}

}  // namespace

extern "C" int ssdbox_prepare_gt(ssdbox_ctx *ctx, const float *d_raw_xywh, const int32_t *d_gt_count,
                                 const int32_t *d_img_wh, const int32_t *d_shrink, int batch, int gmax,
                                 int target_h, int target_w, float *d_boxes)
{
    SSD_REQUIRE(ctx != nullptr, "prepare_gt: ctx is NULL");
    SSD_REQUIRE(batch >= 0 && batch <= 65535, "prepare_gt: batch %d outside [0, 65535]", batch);
    SSD_REQUIRE(gmax >= 1 && gmax <= SSDBOX_MAX_GT, "prepare_gt: gmax %d outside [1, %d]", gmax,
                SSDBOX_MAX_GT);
    SSD_REQUIRE(target_h > 0 && target_w > 0, "prepare_gt: target size %dx%d", target_w, target_h);
    if (batch == 0) return SSDBOX_OK;
    SSD_REQUIRE(d_raw_xywh && d_gt_count && d_img_wh && d_boxes, "prepare_gt: NULL buffer");
    DeviceGuard guard(ctx->device);
    StageTimer timer(ctx, SSDBOX_STAGE_ELEMENTWISE);
    hipLaunchKernelGGL(gt_prepare_kernel, dim3((gmax + kPB - 1) / kPB, batch), dim3(kPB), 0, ctx->stream,
                       reinterpret_cast<const float4 *>(d_raw_xywh), d_gt_count, d_img_wh, d_shrink, gmax,
                       target_h, target_w, reinterpret_cast<float4 *>(d_boxes));
    SSD_CHECK_HIP(hipGetLastError());
    return SSDBOX_OK;
}

extern "C" int ssdbox_widerface_rows(ssdbox_ctx *ctx, const float *d_dets, const int32_t *d_count,
                                     int batch, int limit, int32_t *d_xywh, float *d_scores)
{
    SSD_REQUIRE(ctx != nullptr, "widerface_rows: ctx is NULL");
    SSD_REQUIRE(batch >= 0 && batch <= 65535, "widerface_rows: batch %d outside [0, 65535]", batch);
    SSD_REQUIRE(limit >= 1, "widerface_rows: limit %d", limit);
    if (batch == 0) return SSDBOX_OK;
    SSD_REQUIRE(d_dets && d_count && d_xywh && d_scores, "widerface_rows: NULL buffer");
    DeviceGuard guard(ctx->device);
    StageTimer timer(ctx, SSDBOX_STAGE_ELEMENTWISE);
    hipLaunchKernelGGL(wider_rows_kernel, dim3((limit + kPB - 1) / kPB, batch), dim3(kPB), 0, ctx->stream,
                       d_dets, d_count, limit, reinterpret_cast<int4 *>(d_xywh), d_scores);
    SSD_CHECK_HIP(hipGetLastError());
    return SSDBOX_OK;
}
