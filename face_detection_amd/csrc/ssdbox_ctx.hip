// Context, error reporting, scratch management, stage timing and the anchor geometry
// (host side of reference utils/anchor_utils.py:6-62).
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <new>

#include "ssdbox_internal.h"

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int ssdbox_version(void) { return 100; }

extern "C" const char *ssdbox_last_error(void) { return g_err; }

// --------------------------------------------------------------------------- scratch
// Grows *ptr to hold `need` elements.  A grown (or first) buffer is zero-filled when
// `zero` is set; callers rely on that for the counters that kernels keep at zero.
int ensure_buffer(void **ptr, size_t *cap, size_t need, size_t elem_size, bool zero,
                  hipStream_t stream)
{
    if (need <= *cap && *ptr != nullptr) return SSDBOX_OK;
    if (*ptr != nullptr) {
        SSD_CHECK_HIP(hipStreamSynchronize(stream));
        SSD_CHECK_HIP(hipFree(*ptr));
        *ptr = nullptr;
        *cap = 0;
    }
    size_t want = need + need / 4 + 64;
    hipError_t e = hipMalloc(ptr, want * elem_size);
    if (e != hipSuccess) {
        *ptr = nullptr;
        set_error("hipMalloc of %zu bytes failed: %s", want * elem_size, hipGetErrorString(e));
        return SSDBOX_ERR_NOMEM;
    }
    *cap = want;
    if (zero) SSD_CHECK_HIP(hipMemsetAsync(*ptr, 0, want * elem_size, stream));
    return SSDBOX_OK;
}

// ----------------------------------------------------------------------------- timing
bool acquire_events(ssdbox_ctx *ctx, hipEvent_t ev[2])
{
    ev[0] = ev[1] = nullptr;
    for (int i = 0; i < 2; ++i) {
        if (!ctx->event_pool.empty()) {
            ev[i] = ctx->event_pool.back();
            ctx->event_pool.pop_back();
        } else if (hipEventCreate(&ev[i]) != hipSuccess) {
            ev[i] = nullptr;
        }
    }
    if (ev[0] != nullptr && ev[1] != nullptr) return true;
    for (int i = 0; i < 2; ++i)
        if (ev[i] != nullptr) ctx->event_pool.push_back(ev[i]);
    return false;
}

StageTimer::StageTimer(ssdbox_ctx *c, int stage) : ctx(c)
{
    if (!ctx->timing) return;
    hipEvent_t ev[2];
    if (!acquire_events(ctx, ev)) return;
    (void)hipEventRecord(ev[0], ctx->stream);
    stop = ev[1];
    ctx->records.push_back({stage, ev[0], ev[1]});
}

StageTimer::~StageTimer()
{
    if (stop != nullptr) (void)hipEventRecord(stop, ctx->stream);
}

// ---------------------------------------------------------------------------- context
// Fixed scratch for the entry point that allocates nothing (ssdbox_match_encode_static): one
// array per GPU, part of the loaded code object, held until the process ends.  Every context of
// a device gets the same address, so calls that overlap on one GPU race on it.
constexpr size_t kStaticScratchBytes = (size_t)SSDBOX_STATIC_SCRATCH_BYTES;
static __device__ __align__(256) unsigned char scratch[kStaticScratchBytes];

// A stream of the library's own: non-blocking, at the highest priority the device offers
// (several batches in flight on high-priority streams measured 9 % faster than on default ones).
static hipError_t create_own_stream(hipStream_t *stream)
{
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) greatest = 0;
    return hipStreamCreateWithPriority(stream, hipStreamNonBlocking, greatest);
}

extern "C" int ssdbox_ctx_create(int device, void *stream, int own_stream, ssdbox_ctx **out)
{
    SSD_REQUIRE(out != nullptr, "ssdbox_ctx_create: out is NULL");
    *out = nullptr;
    int n_dev = 0;
    SSD_CHECK_HIP(hipGetDeviceCount(&n_dev));
    SSD_REQUIRE(device >= 0 && device < n_dev, "ssdbox_ctx_create: device %d of %d", device, n_dev);
    DeviceGuard guard(device);
    hipDeviceProp_t prop;
    SSD_CHECK_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_error("libssdbox is built for gfx950 only; device %d is %s", device, prop.gcnArchName);
        return SSDBOX_ERR_UNSUPPORTED;
    }
    // The array's name is a host-side stand-in; its address on this device comes from the
    // runtime (the guard above made the device current).
    void *scratch_ptr = nullptr;
    SSD_CHECK_HIP(hipGetSymbolAddress(&scratch_ptr, HIP_SYMBOL(scratch)));
    ssdbox_ctx *ctx = new (std::nothrow) ssdbox_ctx();
    if (ctx == nullptr) {
        set_error("ssdbox_ctx_create: out of host memory");
        return SSDBOX_ERR_NOMEM;
    }
    ctx->device = device;
    ctx->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    ctx->static_scratch = static_cast<unsigned char *>(scratch_ptr);
    ctx->static_scratch_bytes = kStaticScratchBytes;
    ctx->buf_cached = 0;
    ctx->stage[0] = ctx->stage[1] = nullptr;
    ctx->stage_ev[0] = ctx->stage_ev[1] = nullptr;
    ctx->stage_busy[0] = ctx->stage_busy[1] = false;
    ctx->stage_next = 0;
    ctx->owns_stream = (own_stream != 0);
    ctx->timing = false;
    if (ctx->owns_stream) {
        hipError_t e = create_own_stream(&ctx->stream);
        if (e != hipSuccess) {
            delete ctx;
            set_error("hipStreamCreate failed: %s", hipGetErrorString(e));
            return SSDBOX_ERR_HIP;
        }
    } else {
        ctx->stream = (hipStream_t)stream;
    }
    *out = ctx;
    return SSDBOX_OK;
}

extern "C" int ssdbox_ctx_destroy(ssdbox_ctx *ctx)
{
    if (ctx == nullptr) return SSDBOX_OK;
    DeviceGuard guard(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (auto &r : ctx->records) {
        (void)hipEventDestroy(r.start);
        (void)hipEventDestroy(r.stop);
    }
    for (auto &e : ctx->event_pool) (void)hipEventDestroy(e);
    void *bufs[] = {ctx->gkey, ctx->loss_key, ctx->loss_small, ctx->cand, ctx->cand_count, ctx->cand_hist, ctx->sboxes, ctx->sscores,
                    ctx->sidx, ctx->keff, ctx->mask, ctx->selftest_out, ctx->kept_box, ctx->floor_key};
    for (void *b : bufs)
        if (b != nullptr) (void)hipFree(b);
    free_buffer_state(ctx);
    if (ctx->owns_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return SSDBOX_OK;
}

extern "C" int ssdbox_ctx_set_stream(ssdbox_ctx *ctx, void *stream, int own_stream)
{
    SSD_REQUIRE(ctx != nullptr, "ssdbox_ctx_set_stream: ctx is NULL");
    DeviceGuard guard(ctx->device);
    SSD_CHECK_HIP(hipStreamSynchronize(ctx->stream));     // scratch is tied to stream order
    // the new stream exists before the old one goes: a failure leaves the context as it was
    hipStream_t next = (hipStream_t)stream;
    if (own_stream != 0) SSD_CHECK_HIP(create_own_stream(&next));
    if (ctx->owns_stream) (void)hipStreamDestroy(ctx->stream);
    ctx->stream = next;
    ctx->owns_stream = (own_stream != 0);
    return SSDBOX_OK;
}

extern "C" int ssdbox_ctx_get_stream(ssdbox_ctx *ctx, void **stream)
{
    SSD_REQUIRE(ctx != nullptr && stream != nullptr, "ssdbox_ctx_get_stream: NULL argument");
    *stream = (void *)ctx->stream;
    return SSDBOX_OK;
}

extern "C" int ssdbox_ctx_synchronize(ssdbox_ctx *ctx)
{
    SSD_REQUIRE(ctx != nullptr, "ssdbox_ctx_synchronize: ctx is NULL");
    DeviceGuard guard(ctx->device);
    SSD_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    return SSDBOX_OK;
}

extern "C" int ssdbox_ctx_set_timing(ssdbox_ctx *ctx, int enabled)
{
    SSD_REQUIRE(ctx != nullptr, "ssdbox_ctx_set_timing: ctx is NULL");
    ctx->timing = (enabled != 0);
    return SSDBOX_OK;
}

extern "C" int ssdbox_ctx_collect_timing(ssdbox_ctx *ctx, float *ms, int32_t *launches)
{
    SSD_REQUIRE(ctx != nullptr && ms != nullptr && launches != nullptr,
                "ssdbox_ctx_collect_timing: NULL argument");
    DeviceGuard guard(ctx->device);
    for (int s = 0; s < SSDBOX_NUM_STAGES; ++s) {
        ms[s] = 0.0f;
        launches[s] = 0;
    }
    SSD_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    // Every record leaves the list and every event returns to the pool whatever happens; the
    // first error is reported once the bookkeeping is consistent.
    hipError_t first = hipSuccess;
    for (auto &r : ctx->records) {
        float t = 0.0f;
        const hipError_t e = hipEventElapsedTime(&t, r.start, r.stop);
        if (e == hipSuccess) {
            ms[r.stage] += t;
            launches[r.stage] += 1;
        } else if (first == hipSuccess) {
            first = e;
        }
        ctx->event_pool.push_back(r.start);
        ctx->event_pool.push_back(r.stop);
    }
    ctx->records.clear();
    if (first != hipSuccess) {
        set_error("ssdbox_ctx_collect_timing: hipEventElapsedTime failed: %s", hipGetErrorString(first));
        return SSDBOX_ERR_HIP;
    }
    return SSDBOX_OK;
}

// Like collect, but keeps every launch apart: stage, begin and end in milliseconds after the first
// recorded launch of `ref` (the same context, or another one of the same device, which keeps its
// records: the launches of several contexts on one time axis).  Events of one device share a
// clock, whichever stream they were recorded on.
extern "C" int ssdbox_ctx_collect_timeline(ssdbox_ctx *ctx, ssdbox_ctx *ref, int32_t *stage, float *begin_ms,
                                           float *end_ms, int capacity, int32_t *count)
{
    SSD_REQUIRE(ctx != nullptr && ref != nullptr && stage != nullptr && begin_ms != nullptr && end_ms != nullptr &&
                    count != nullptr && capacity >= 0,
                "ssdbox_ctx_collect_timeline: bad argument");
    SSD_REQUIRE(ctx->device == ref->device, "ssdbox_ctx_collect_timeline: contexts on different devices");
    *count = 0;
    DeviceGuard guard(ctx->device);
    SSD_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    SSD_CHECK_HIP(hipStreamSynchronize(ref->stream));
    SSD_REQUIRE(!ref->records.empty(), "ssdbox_ctx_collect_timeline: the reference context has no record");
    const hipEvent_t origin = ref->records.front().start;
    hipError_t first = hipSuccess;
    int n = 0;
    for (auto &r : ctx->records) {
        float t0 = 0.0f, t1 = 0.0f;
        hipError_t e = hipEventElapsedTime(&t0, origin, r.start);
        if (e == hipSuccess) e = hipEventElapsedTime(&t1, origin, r.stop);
        if (e == hipSuccess && n < capacity) {
            stage[n] = r.stage;
            begin_ms[n] = t0;
            end_ms[n] = t1;
            ++n;
        } else if (e != hipSuccess && first == hipSuccess) {
            first = e;
        }
    }
    *count = n;
    // (the records stay: ssdbox_ctx_collect_timing returns the events to the pool)
    if (first != hipSuccess) {
        set_error("ssdbox_ctx_collect_timeline: hipEventElapsedTime failed: %s", hipGetErrorString(first));
        return SSDBOX_ERR_HIP;
    }
    return SSDBOX_OK;
}

#ifdef SSD_EXP_TRACE
// ------------------------------------------------------------------- trace (experiment build)
namespace {
constexpr int kTraceCap = 1 << 18;
unsigned long long *g_trace_dev = nullptr;
int g_trace_n = 0;
struct TraceMeta {
    int stage;
    const void *ctx;
};
TraceMeta *g_trace_meta = nullptr;
}  // namespace

// One thread launches in the experiment (no lock).
unsigned long long *ssd_trace_slot(ssdbox_ctx *ctx, int stage)
{
    if (g_trace_dev == nullptr) {
        if (hipMalloc((void **)&g_trace_dev, (size_t)(kTraceCap + 1) * 2 * sizeof(unsigned long long)) != hipSuccess)
            abort();
        (void)hipMemset(g_trace_dev, 0, (size_t)(kTraceCap + 1) * 2 * sizeof(unsigned long long));
        g_trace_meta = new TraceMeta[kTraceCap];
    }
    if (g_trace_n >= kTraceCap) return g_trace_dev + (size_t)kTraceCap * 2;     // overflow slot
    g_trace_meta[g_trace_n] = {stage, ctx};
    return g_trace_dev + (size_t)(g_trace_n++) * 2;
}

extern "C" int ssdbox_debug_trace_reset()
{
    g_trace_n = 0;
    return 0;
}

// Call after the device is idle.  Returns the number of launches traced since the last reset.
extern "C" int ssdbox_debug_trace_dump(int32_t *stage, uint64_t *ctx_id, uint64_t *begin, uint64_t *end, int cap)
{
    const int n = g_trace_n < cap ? g_trace_n : cap;
    if (n == 0) return 0;
    std::vector<unsigned long long> h((size_t)n * 2);
    if (hipMemcpy(h.data(), g_trace_dev, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess)
        return -1;
    for (int i = 0; i < n; ++i) {
        stage[i] = g_trace_meta[i].stage;
        ctx_id[i] = (uint64_t)(uintptr_t)g_trace_meta[i].ctx;
        begin[i] = h[2 * i];
        end[i] = h[2 * i + 1];
    }
    return n;
}
#endif

// --------------------------------------------------------------------------- geometry
extern "C" int ssdbox_geom_create(int device, int img_width, int img_height, int n_layers,
                                  const int32_t *steps, const double *scales,
                                  const int32_t *ratio_counts, const double *ratios_flat,
                                  int two_boxes_for_ar1, double offsets, ssdbox_geom **out)
{
    SSD_REQUIRE(out != nullptr, "ssdbox_geom_create: out is NULL");
    *out = nullptr;
    SSD_REQUIRE(steps && scales && ratio_counts && ratios_flat, "ssdbox_geom_create: NULL table");
    SSD_REQUIRE(n_layers >= 1 && n_layers <= SSDBOX_MAX_LAYERS,
                "ssdbox_geom_create: n_layers %d outside [1, %d]", n_layers, SSDBOX_MAX_LAYERS);
    SSD_REQUIRE(img_width > 0 && img_height > 0, "ssdbox_geom_create: image %dx%d", img_width,
                img_height);
    try {                       // the tables grow in std::vector: nothing may throw across the ABI

    GeomDev g;
    memset(&g, 0, sizeof(g));
    g.n_layers = n_layers;
    std::vector<float> cx, cy;
    std::vector<float4> col;
    long long total = 0;
    int tiles = 0;
    const double *ratios = ratios_flat;
    for (int m = 0; m < n_layers; ++m) {
        SSD_REQUIRE(steps[m] > 0, "ssdbox_geom_create: steps[%d] = %d", m, steps[m]);
        LayerDev &L = g.layer[m];
        L.fm_w = img_width / steps[m];          // integer floor, anchor_utils.py:29-30
        L.fm_h = img_height / steps[m];
        int nb = 0;
        for (int r = 0; r < ratio_counts[m]; ++r) {
            double ratio = ratios[r];
            double w[2], h[2];
            int n = 0;
            if (ratio == 1.0) {                 // anchor_utils.py:36-49
                w[n] = scales[m];
                h[n] = scales[m];
                ++n;
                if (two_boxes_for_ar1) {
                    double s = sqrt(scales[m] * scales[m + 1]);
                    w[n] = s;
                    h[n] = s;
                    ++n;
                }
            } else {                            // anchor_utils.py:50-57
                double root = sqrt(ratio);
                w[n] = scales[m] * root;
                h[n] = scales[m] / root;
                ++n;
            }
            for (int q = 0; q < n; ++q) {
                SSD_REQUIRE(nb < SSDBOX_MAX_BOXES_PER_CELL,
                            "ssdbox_geom_create: layer %d has more than %d boxes per cell", m,
                            SSDBOX_MAX_BOXES_PER_CELL);
                // float64 -> float32 once (anchor_utils.py:59), then clip (:60)
                L.bw[nb] = fminf(fmaxf((float)w[q], 0.0f), 1.0f);
                L.bh[nb] = fminf(fmaxf((float)h[q], 0.0f), 1.0f);
                ++nb;
            }
        }
        ratios += ratio_counts[m];
        L.n_box = nb;
        L.k_off = g.k_total;
        g.k_total += nb;
        // one tile of padding per layer: the match kernel reads past the edge unclamped
        L.col_off = g.col_total;
        g.col_total += L.fm_w * nb + kTileCols;
        L.rowk_off = g.rowk_total;
        g.rowk_total += (L.fm_h + kTileRows) * nb;
        L.anchor_base = (int)total;
        L.row_anchors = L.fm_w * nb;
        total += (long long)L.fm_w * L.fm_h * nb;
        SSD_REQUIRE(total < (1ll << 30), "ssdbox_geom_create: more than 2^30 anchors");
        L.wmax = 0.0f;
        L.hmax = 0.0f;
        for (int k = 0; k < nb; ++k) {
            L.wmax = fmaxf(L.wmax, L.bw[k]);
            L.hmax = fmaxf(L.hmax, L.bh[k]);
        }
        // Match-kernel tiling.  A workgroup's ground-truth list grows with the share of the
        // image its anchors can touch; where that share is large (coarse maps, big boxes) the
        // tile shrinks from 32 rows (four anchors per thread) to 8 rows (one per thread).
        {
            const float fx = fminf(1.0f, (float)kTileCols / (float)(nb > 0 ? nb : 1) / (float)(L.fm_w > 0 ? L.fm_w : 1) + L.wmax);
            const float fy = fminf(1.0f, (float)kTileRows / (float)(L.fm_h > 0 ? L.fm_h : 1) + L.hmax);
            L.coarse = (fx * fy > 0.10f) ? 1 : 0;
        }
        L.tile_rows = L.coarse ? kCoarseTileRows : kTileRows;
        L.tiles_x = (L.row_anchors + kTileCols - 1) / kTileCols;
        L.tiles_y = (L.fm_h + L.tile_rows - 1) / L.tile_rows;
        // multiply-high division constants; each is used for n below the bound checked here
        auto magic = [](long long d) { return (unsigned)(((1ull << 32) + (unsigned long long)d - 1) / (unsigned long long)d); };
        L.nbox_magic = nb > 1 ? magic(nb) : 0u;
        L.rowanc_magic = L.row_anchors > 1 ? magic(L.row_anchors) : 0u;
        L.tilesx_magic = L.tiles_x > 1 ? magic(L.tiles_x) : 0u;
        SSD_REQUIRE((long long)L.fm_w * L.fm_h * nb * (long long)(L.row_anchors > 0 ? L.row_anchors : 1) < (1ll << 32) &&
                        (long long)L.tiles_x * L.tiles_y * (long long)(L.tiles_x > 0 ? L.tiles_x : 1) < (1ll << 32),
                    "ssdbox_geom_create: layer %d is too large for the index arithmetic", m);
        L.cx_off = (int)cx.size();
        L.cy_off = (int)cy.size();
        for (int j = 0; j < L.fm_w; ++j)        // anchor_utils.py:32
            cx.push_back(fminf(fmaxf((float)((j + offsets) / L.fm_w), 0.0f), 1.0f));
        for (int i = 0; i < L.fm_h; ++i)        // anchor_utils.py:33
            cy.push_back(fminf(fmaxf((float)((i + offsets) / L.fm_h), 0.0f), 1.0f));
        for (int i = 0; i < kTileRows; ++i) cy.push_back(L.fm_h > 0 ? cy.back() : 0.0f);
        for (int j = 0; j < L.fm_w; ++j)
            for (int k = 0; k < nb; ++k)
                col.push_back(make_float4(cx[L.cx_off + j], L.bw[k], L.bh[k], 0.0f));
        for (int c = 0; c < kTileCols; ++c)
            col.push_back(L.fm_w > 0 && nb > 0 ? col.back() : make_float4(0.0f, 1.0f, 1.0f, 0.0f));
    }
    // Tiles are numbered from the last (coarsest) layer to the first: workgroups start in
    // index order, and the coarse tiles carry the longest ground-truth lists.
    for (int m = n_layers - 1; m >= 0; --m) {
        g.layer[m].tile_base = tiles;
        tiles += g.layer[m].tiles_x * g.layer[m].tiles_y;
    }
    g.num_anchors = (int)total;
    g.n_tiles = tiles;

    ssdbox_geom *geom = new ssdbox_geom();     // (a failure is caught below)
    geom->device = device;
    geom->host = g;
    geom->d_geom = nullptr;
    geom->d_cx = nullptr;
    geom->d_cy = nullptr;
    geom->d_col = nullptr;
    if (device < 0) {           // host-only handle: layer table and counts, no kernels
        *out = geom;
        return SSDBOX_OK;
    }
    int n_dev = 0;
    hipError_t e0 = hipGetDeviceCount(&n_dev);
    if (e0 != hipSuccess || device >= n_dev) {
        delete geom;
        set_error("ssdbox_geom_create: device %d not available (%s)", device,
                  e0 != hipSuccess ? hipGetErrorString(e0) : "index out of range");
        return e0 != hipSuccess ? SSDBOX_ERR_HIP : SSDBOX_ERR_INVALID;
    }
    DeviceGuard guard(device);
    size_t ncx = cx.size() ? cx.size() : 1, ncy = cy.size() ? cy.size() : 1;
    hipError_t e = hipMalloc((void **)&geom->d_geom, sizeof(GeomDev));
    if (e == hipSuccess) e = hipMalloc((void **)&geom->d_cx, ncx * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void **)&geom->d_cy, ncy * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void **)&geom->d_col, (col.size() ? col.size() : 1) * sizeof(float4));
    if (e == hipSuccess && !col.empty())
        e = hipMemcpy(geom->d_col, col.data(), col.size() * sizeof(float4), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(geom->d_geom, &g, sizeof(GeomDev), hipMemcpyHostToDevice);
    if (e == hipSuccess && !cx.empty())
        e = hipMemcpy(geom->d_cx, cx.data(), cx.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess && !cy.empty())
        e = hipMemcpy(geom->d_cy, cy.data(), cy.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        set_error("ssdbox_geom_create: device upload failed: %s", hipGetErrorString(e));
        ssdbox_geom_destroy(geom);
        return SSDBOX_ERR_HIP;
    }
    *out = geom;
    return SSDBOX_OK;

    } catch (...) {
        set_error("ssdbox_geom_create: out of host memory");
        return SSDBOX_ERR_NOMEM;
    }
}

extern "C" int ssdbox_geom_destroy(ssdbox_geom *geom)
{
    if (geom == nullptr) return SSDBOX_OK;
    if (geom->device < 0) {
        delete geom;
        return SSDBOX_OK;
    }
    DeviceGuard guard(geom->device);
    if (geom->d_geom) (void)hipFree(geom->d_geom);
    if (geom->d_cx) (void)hipFree(geom->d_cx);
    if (geom->d_cy) (void)hipFree(geom->d_cy);
    if (geom->d_col) (void)hipFree(geom->d_col);
    delete geom;
    return SSDBOX_OK;
}

extern "C" int64_t ssdbox_geom_num_anchors(const ssdbox_geom *geom)
{
    return geom ? (int64_t)geom->host.num_anchors : -1;
}

extern "C" int ssdbox_geom_layers(const ssdbox_geom *geom, int32_t *fm_w, int32_t *fm_h,
                                  int32_t *n_box)
{
    if (geom == nullptr) return SSDBOX_ERR_INVALID;
    for (int m = 0; m < geom->host.n_layers; ++m) {
        if (fm_w) fm_w[m] = geom->host.layer[m].fm_w;
        if (fm_h) fm_h[m] = geom->host.layer[m].fm_h;
        if (n_box) n_box[m] = geom->host.layer[m].n_box;
    }
    return geom->host.n_layers;
}
