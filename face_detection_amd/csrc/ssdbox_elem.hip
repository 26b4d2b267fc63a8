// Elementwise box math and anchor materialisation: the small functions of reference
// utils/box_utils.py (compute_area :4-18, compute_iou :21-44, encode :101-123, decode
// :126-147, transform_* :150-181) and the array form of utils/anchor_utils.py:6-62.
// All of them are plain streaming kernels: one thread per output row, 16-byte accesses.
#include "ssdbox_internal.h"

namespace {

constexpr int kEB = 256;

inline unsigned blocks_for(long long n) { return (unsigned)((n + kEB - 1) / kEB); }

__global__ __launch_bounds__(kEB) void anchors_kernel(const GeomDev *__restrict__ g,
                                                      const float *__restrict__ cx,
                                                      const float *__restrict__ cy, int n,
                                                      float4 *__restrict__ out)
{
    const int a = blockIdx.x * kEB + threadIdx.x;
    if (a < n) out[a] = geom_anchor(g, cx, cy, a);
}

__global__ __launch_bounds__(kEB) void area_kernel(const float2 *__restrict__ tl,
                                                   const float2 *__restrict__ br, long long n,
                                                   float max_size, float *__restrict__ out)
{
    const long long i = (long long)blockIdx.x * kEB + threadIdx.x;
    if (i < n) {
        const float2 a = tl[i], b = br[i];
        out[i] = clip0(b.x - a.x, max_size) * clip0(b.y - a.y, max_size);
    }
}

// (na, nb) IoU matrix: each thread owns one row box and walks a 256-column slab staged in
// LDS, so stores of neighbouring rows do not interleave within a slab.
__global__ __launch_bounds__(kEB) void iou_kernel(const float4 *__restrict__ a, long long na,
                                                  const float4 *__restrict__ b, long long nb,
                                                  float max_size, float *__restrict__ out)
{
    __shared__ float4 s_b[kEB];
    __shared__ float s_area[kEB];
    const long long j0 = (long long)blockIdx.x * kEB;
    const long long i0 = (long long)blockIdx.y * kEB;
    const int t = threadIdx.x;
    const int nbj = (int)min((long long)kEB, nb - j0);
    if (t < nbj) {
        const float4 v = b[j0 + t];
        s_b[t] = v;
        s_area[t] = clip0(v.z - v.x, max_size) * clip0(v.w - v.y, max_size);
    }
    __syncthreads();
    const int nai = (int)min((long long)kEB, na - i0);
    // lanes sweep columns so that each store instruction writes consecutive floats of a row
    for (int r = 0; r < nai; ++r) {
        const float4 ra = a[i0 + r];
        const float area_a = clip0(ra.z - ra.x, max_size) * clip0(ra.w - ra.y, max_size);
        if (t < nbj) {
            const float4 cb = s_b[t];
            const float iw = clip0(fminf(ra.z, cb.z) - fmaxf(ra.x, cb.x), max_size);
            const float ih = clip0(fminf(ra.w, cb.w) - fmaxf(ra.y, cb.y), max_size);
            const float inter = iw * ih;
            out[(i0 + r) * nb + j0 + t] = inter / (area_a + s_area[t] - inter);
        }
    }
}

template <bool ENCODE, bool GEOM>
__global__ __launch_bounds__(kEB) void codec_kernel(const GeomDev *__restrict__ g,
                                                    const float *__restrict__ cx,
                                                    const float *__restrict__ cy,
                                                    const float4 *__restrict__ anchors,
                                                    long long num_anchors,
                                                    const float4 *__restrict__ in, long long rows,
                                                    float vc, float vs, float4 *__restrict__ out)
{
    const long long r = (long long)blockIdx.x * kEB + threadIdx.x;
    if (r >= rows) return;
    const long long ai = r % num_anchors;
    const float4 anc = GEOM ? geom_anchor(g, cx, cy, (int)ai) : anchors[ai];
    out[r] = ENCODE ? encode_one(anc, in[r], vc, vs) : decode_one(anc, in[r], vc, vs);
}

template <bool TO_CENTER>
__global__ __launch_bounds__(kEB) void transform_kernel(const float4 *__restrict__ in, long long n,
                                                        float4 *__restrict__ out)
{
    const long long i = (long long)blockIdx.x * kEB + threadIdx.x;
    if (i >= n) return;
    const float4 b = in[i];
    if (TO_CENTER) {
        out[i] = make_float4((b.x + b.z) / 2.0f, (b.y + b.w) / 2.0f, b.z - b.x, b.w - b.y);
    } else {
        const float hw = b.z / 2.0f, hh = b.w / 2.0f;
        out[i] = make_float4(b.x - hw, b.y - hh, b.x + hw, b.y + hh);
    }
}

__global__ void selftest_kernel(const float *x_log, const float *e_log, const float *x_exp,
                                const float *e_exp, int n, int *fail)
{
    const int t = threadIdx.x;
    // wave maximum against a shuffle reduction, several patterns
    for (int pat = 0; pat < 4; ++pat) {
        float v;
        if (pat == 0) v = (float)((t * 37) % 101);
        else if (pat == 1) v = (t == 63 - (int)(blockIdx.x % 64)) ? 5.0f : 0.0f;
        else if (pat == 2) v = 0.0f;
        else v = 1.0f / (float)(1 + ((t * 13 + blockIdx.x) % 64));
        float ref = v;
        for (int o = 32; o > 0; o >>= 1) ref = fmaxf(ref, __shfl_xor(ref, o));
        if (wave_max_nonneg(v) != ref) atomicOr(fail, 1);
        const uint32_t u = (pat == 2) ? 0xFFFFFFFFu : (uint32_t)((t * 2654435761u) >> (pat * 3)) + blockIdx.x;
        uint32_t uref = u;
        for (int o = 32; o > 0; o >>= 1) uref = min(uref, (uint32_t)__shfl_xor((int)uref, o));
        if (wave_min_u32(u) != uref) atomicOr(fail, 1);
    }
    if (lane_id() != (t & 63)) atomicOr(fail, 2);
    for (int i = blockIdx.x * blockDim.x + t; i < n; i += gridDim.x * blockDim.x) {
        if (f2u(ssd_logf(x_log[i])) != f2u(e_log[i]) && !(e_log[i] != e_log[i] && ssd_logf(x_log[i]) != ssd_logf(x_log[i])))
            atomicOr(fail, 4);
        if (f2u(ssd_expf(x_exp[i])) != f2u(e_exp[i]) && !(e_exp[i] != e_exp[i] && ssd_expf(x_exp[i]) != ssd_expf(x_exp[i])))
            atomicOr(fail, 8);
    }
}

int check_anchor_source(const char *who, ssdbox_ctx *ctx, const ssdbox_geom *geom,
                        const float *d_anchors, int64_t num_anchors)
{
    SSD_REQUIRE(ctx != nullptr, "%s: ctx is NULL", who);
    SSD_REQUIRE((geom != nullptr) != (d_anchors != nullptr),
                "%s: pass exactly one of geom and d_anchors", who);
    SSD_REQUIRE(num_anchors >= 1 && num_anchors < (1ll << 30), "%s: num_anchors %lld", who,
                (long long)num_anchors);
    if (geom != nullptr) {
        SSD_REQUIRE(geom->d_geom != nullptr, "%s: geometry handle is host-only", who);
        SSD_REQUIRE(geom->device == ctx->device, "%s: geometry on device %d, context on %d", who,
                    geom->device, ctx->device);
        SSD_REQUIRE(num_anchors == geom->host.num_anchors,
                    "%s: num_anchors %lld does not match the geometry's %d", who,
                    (long long)num_anchors, geom->host.num_anchors);
    }
    return SSDBOX_OK;
}

template <bool ENCODE>
int run_codec(const char *who, ssdbox_ctx *ctx, const ssdbox_geom *geom, const float *d_anchors,
              int64_t num_anchors, const float *d_in, int64_t rows, float vc, float vs, float *d_out)
{
    int rc = check_anchor_source(who, ctx, geom, d_anchors, num_anchors);
    if (rc != SSDBOX_OK) return rc;
    SSD_REQUIRE(rows >= 0 && rows % num_anchors == 0,
                "%s: rows %lld is not a multiple of num_anchors %lld", who, (long long)rows,
                (long long)num_anchors);
    if (rows == 0) return SSDBOX_OK;
    SSD_REQUIRE(d_in && d_out, "%s: NULL buffer", who);
    DeviceGuard guard(ctx->device);
    StageTimer timer(ctx, SSDBOX_STAGE_ELEMENTWISE);
    const float4 *in = reinterpret_cast<const float4 *>(d_in);
    float4 *out = reinterpret_cast<float4 *>(d_out);
    if (geom) {
        hipLaunchKernelGGL((codec_kernel<ENCODE, true>), dim3(blocks_for(rows)), dim3(kEB), 0,
                           ctx->stream, geom->d_geom, geom->d_cx, geom->d_cy, nullptr,
                           (long long)num_anchors, in, (long long)rows, vc, vs, out);
    } else {
        hipLaunchKernelGGL((codec_kernel<ENCODE, false>), dim3(blocks_for(rows)), dim3(kEB), 0,
                           ctx->stream, nullptr, nullptr, nullptr,
                           reinterpret_cast<const float4 *>(d_anchors), (long long)num_anchors, in,
                           (long long)rows, vc, vs, out);
    }
    SSD_CHECK_HIP(hipGetLastError());
    return SSDBOX_OK;
}

}  // namespace

extern "C" int ssdbox_generate_anchors(ssdbox_ctx *ctx, const ssdbox_geom *geom, float *d_out)
{
    SSD_REQUIRE(ctx && geom && d_out, "generate_anchors: NULL argument");
    SSD_REQUIRE(geom->d_geom != nullptr, "generate_anchors: geometry handle is host-only");
    SSD_REQUIRE(geom->device == ctx->device, "generate_anchors: geometry on device %d, context on %d",
                geom->device, ctx->device);
    const int n = geom->host.num_anchors;
    if (n == 0) return SSDBOX_OK;
    DeviceGuard guard(ctx->device);
    StageTimer timer(ctx, SSDBOX_STAGE_ELEMENTWISE);
    hipLaunchKernelGGL(anchors_kernel, dim3(blocks_for(n)), dim3(kEB), 0, ctx->stream, geom->d_geom,
                       geom->d_cx, geom->d_cy, n, reinterpret_cast<float4 *>(d_out));
    SSD_CHECK_HIP(hipGetLastError());
    return SSDBOX_OK;
}

extern "C" int ssdbox_compute_area(ssdbox_ctx *ctx, const float *d_top_left, const float *d_bot_right,
                                   int64_t n, float max_size, float *d_area)
{
    SSD_REQUIRE(ctx != nullptr && n >= 0, "compute_area: bad argument");
    if (n == 0) return SSDBOX_OK;
    SSD_REQUIRE(d_top_left && d_bot_right && d_area, "compute_area: NULL buffer");
    DeviceGuard guard(ctx->device);
    StageTimer timer(ctx, SSDBOX_STAGE_ELEMENTWISE);
    hipLaunchKernelGGL(area_kernel, dim3(blocks_for(n)), dim3(kEB), 0, ctx->stream,
                       reinterpret_cast<const float2 *>(d_top_left),
                       reinterpret_cast<const float2 *>(d_bot_right), (long long)n, max_size, d_area);
    SSD_CHECK_HIP(hipGetLastError());
    return SSDBOX_OK;
}

extern "C" int ssdbox_compute_iou(ssdbox_ctx *ctx, const float *d_boxes_a, int64_t na,
                                  const float *d_boxes_b, int64_t nb, float max_size, float *d_iou)
{
    SSD_REQUIRE(ctx != nullptr && na >= 0 && nb >= 0, "compute_iou: bad argument");
    if (na == 0 || nb == 0) return SSDBOX_OK;
    SSD_REQUIRE(d_boxes_a && d_boxes_b && d_iou, "compute_iou: NULL buffer");
    const long long gy = (na + kEB - 1) / kEB;
    SSD_REQUIRE(gy <= 65535, "compute_iou: more than %d rows", 65535 * kEB);
    DeviceGuard guard(ctx->device);
    StageTimer timer(ctx, SSDBOX_STAGE_ELEMENTWISE);
    hipLaunchKernelGGL(iou_kernel, dim3(blocks_for(nb), (unsigned)gy), dim3(kEB), 0, ctx->stream,
                       reinterpret_cast<const float4 *>(d_boxes_a), (long long)na,
                       reinterpret_cast<const float4 *>(d_boxes_b), (long long)nb, max_size, d_iou);
    SSD_CHECK_HIP(hipGetLastError());
    return SSDBOX_OK;
}

extern "C" int ssdbox_encode(ssdbox_ctx *ctx, const ssdbox_geom *geom, const float *d_anchors,
                             int64_t num_anchors, const float *d_boxes, int64_t rows, float v_center,
                             float v_size, float *d_locs)
{
    return run_codec<true>("encode", ctx, geom, d_anchors, num_anchors, d_boxes, rows, v_center,
                           v_size, d_locs);
}

extern "C" int ssdbox_decode(ssdbox_ctx *ctx, const ssdbox_geom *geom, const float *d_anchors,
                             int64_t num_anchors, const float *d_locs, int64_t rows, float v_center,
                             float v_size, float *d_boxes)
{
    return run_codec<false>("decode", ctx, geom, d_anchors, num_anchors, d_locs, rows, v_center,
                            v_size, d_boxes);
}

static int run_transform(ssdbox_ctx *ctx, const float *d_in, int64_t n, float *d_out, bool to_center)
{
    SSD_REQUIRE(ctx != nullptr && n >= 0, "transform: bad argument");
    if (n == 0) return SSDBOX_OK;
    SSD_REQUIRE(d_in && d_out, "transform: NULL buffer");
    DeviceGuard guard(ctx->device);
    StageTimer timer(ctx, SSDBOX_STAGE_ELEMENTWISE);
    const float4 *in = reinterpret_cast<const float4 *>(d_in);
    float4 *out = reinterpret_cast<float4 *>(d_out);
    if (to_center)
        hipLaunchKernelGGL(transform_kernel<true>, dim3(blocks_for(n)), dim3(kEB), 0, ctx->stream, in,
                           (long long)n, out);
    else
        hipLaunchKernelGGL(transform_kernel<false>, dim3(blocks_for(n)), dim3(kEB), 0, ctx->stream, in,
                           (long long)n, out);
    SSD_CHECK_HIP(hipGetLastError());
    return SSDBOX_OK;
}

extern "C" int ssdbox_corner_to_center(ssdbox_ctx *ctx, const float *d_in, int64_t n, float *d_out)
{
    return run_transform(ctx, d_in, n, d_out, true);
}

extern "C" int ssdbox_center_to_corner(ssdbox_ctx *ctx, const float *d_in, int64_t n, float *d_out)
{
    return run_transform(ctx, d_in, n, d_out, false);
}

extern "C" int ssdbox_selftest(ssdbox_ctx *ctx, const float *x_log, const float *expect_log,
                               const float *x_exp, const float *expect_exp, int n)
{
    SSD_REQUIRE(ctx != nullptr && n >= 0, "selftest: bad argument");
    SSD_REQUIRE(n == 0 || (x_log && expect_log && x_exp && expect_exp), "selftest: NULL table");
    DeviceGuard guard(ctx->device);
    float *d = nullptr;
    size_t bytes = (size_t)(n > 0 ? n : 1) * sizeof(float);
    SSD_CHECK_HIP(hipMalloc((void **)&d, 4 * bytes + sizeof(int)));
    int *d_fail = reinterpret_cast<int *>(reinterpret_cast<char *>(d) + 4 * bytes);
    int rc = SSDBOX_OK;
    int fail = -1;
    hipError_t e = hipMemsetAsync(d_fail, 0, sizeof(int), ctx->stream);
    const float *src[4] = {x_log, expect_log, x_exp, expect_exp};
    for (int i = 0; i < 4 && e == hipSuccess && n > 0; ++i)
        e = hipMemcpyAsync(reinterpret_cast<char *>(d) + i * bytes, src[i], (size_t)n * sizeof(float),
                           hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
        float *p = d;
        const size_t stride = bytes / sizeof(float);
        hipLaunchKernelGGL(selftest_kernel, dim3(64), dim3(64), 0, ctx->stream, p, p + stride,
                           p + 2 * stride, p + 3 * stride, n, d_fail);
        e = hipGetLastError();
    }
    if (e == hipSuccess)
        e = hipMemcpyAsync(&fail, d_fail, sizeof(int), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d);
    if (e != hipSuccess) {
        set_error("selftest: %s", hipGetErrorString(e));
        return SSDBOX_ERR_HIP;
    }
    if (fail != 0) {
        set_error("selftest failed, mask 0x%x (1 wave max, 2 lane id, 4 log, 8 exp)", fail);
        rc = SSDBOX_ERR_HIP;
    }
    return rc;
}
