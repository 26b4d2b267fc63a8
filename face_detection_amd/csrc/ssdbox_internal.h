// Internal definitions shared by the libssdbox translation units (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include <mutex>
#include <string>
#include <vector>

#include "ssdbox.h"

// ---------------------------------------------------------------------------------------
// Anchor geometry as the kernels see it.  One LayerDev per predictor layer; the per-column
// and per-row centre tables (float32, already clipped) live in device memory because they
// are the only part whose size grows with the image (sum of fm_w + fm_h floats).
// ---------------------------------------------------------------------------------------
constexpr int kRowsPerThread = 4;  // match kernel: anchors (rows of one column) per thread
constexpr int kTileRows = 8 * kRowsPerThread;   // feature-map rows per workgroup tile
constexpr int kCoarseTileRows = 8;  // rows per tile on coarse maps (one anchor per thread)
constexpr int kTileCols = 32;      // match kernel: anchor columns (cell * n_box + k) per tile
constexpr int kBlock = 256;

struct LayerDev {
    int fm_w, fm_h, n_box;
    int anchor_base;               // index of the layer's first anchor
    int row_anchors;               // fm_w * n_box
    int tiles_x, tiles_y, tile_base;
    int cx_off, cy_off;            // offsets of this layer in the centre tables
    unsigned nbox_magic;           // ceil(2^32 / n_box), row_anchors, tiles_x: exact n / d by
    unsigned rowanc_magic;         // one multiply-high for every n these kernels divide
    unsigned tilesx_magic;
    int coarse;                    // 1: 8-row tiles, one anchor per thread (long box lists)
    int tile_rows;                 // feature-map rows per workgroup tile in this layer
    int k_off;                     // index of the layer's first box shape among all layers' shapes
    int col_off;                   // the layer's first entry in the anchor-column tables
    int rowk_off;                  // the layer's first entry in the (row, shape) table
    float wmax, hmax;              // largest clipped box width / height of the layer
    float bw[SSDBOX_MAX_BOXES_PER_CELL];
    float bh[SSDBOX_MAX_BOXES_PER_CELL];
};

struct GeomDev {
    int n_layers;
    int num_anchors;
    int n_tiles;
    int k_total;                   // box shapes of all layers together (sum of n_box)
    int col_total;                 // anchor columns of all layers (sum of fm_w * n_box)
    int rowk_total;                // (row, shape) pairs of all layers (sum of fm_h * n_box)
    LayerDev layer[SSDBOX_MAX_LAYERS];
};

struct ssdbox_geom {
    int device;
    GeomDev host;                  // host copy of what d_geom points to
    GeomDev *d_geom;
    float4 *d_col;                 // per anchor column of every layer (padded by a tile): cx, w, h, 0
    float *d_cx;                   // concatenated per-layer column centres
    float *d_cy;                   // concatenated per-layer row centres
};

// ---------------------------------------------------------------------------------------
// Context: stream, scratch, optional per-stage event timing.
// ---------------------------------------------------------------------------------------
constexpr size_t kMaxTimingRecords = 1u << 16;   // launches recorded between two collects

struct TimingRecord {
    int stage;
    hipEvent_t start, stop;
};

struct DeviceBuf {
    void *ptr;
    size_t bytes;
};

struct ssdbox_ctx {
    int device;
    int num_cus;                   // compute units of the device (sizes the persistent grids)
    hipStream_t stream;
    bool owns_stream;
    bool timing;
    std::vector<TimingRecord> records;
    std::vector<hipEvent_t> event_pool;

    // match scratch: packed (iou bits, ~anchor) best-anchor key per (image, GT); all zero
    // between calls (the fix-up kernel clears what it read).
    unsigned long long *gkey;
    size_t gkey_cap;               // entries
    bool gkey_dirty;

    // loss scratch: rank keys of the negative log-losses; histograms, state, sums, chunk counts
    unsigned *loss_key;
    size_t loss_key_cap;
    unsigned char *loss_small;
    size_t loss_small_cap;

    // detection scratch
    unsigned long long *cand;      // (batch, segments * 256) candidate keys, per-wave segments
    size_t cand_cap;
    int *cand_count;               // (batch, segments) candidates per segment
    size_t count_cap;
    unsigned int *cand_hist;       // (batch, 2048) score histogram for the pruning pass
    size_t hist_cap;
    bool hist_dirty;               // a call was enqueued only in part: the histogram is not all zero
    float4 *sboxes;                // (batch, LIMIT) decoded boxes in rank order
    float *sscores;
    int *sidx;
    int *keff;                     // (batch) effective K per image
    unsigned long long *mask;      // (batch, LIMIT, LIMIT/64) suppression bits
    size_t det_cap;                // images
    size_t det_rank_cap;           // ranked boxes per image the buffers above hold
    float4 *kept_box;              // (batch, limit) boxes kept so far, general path beyond 4096 ranks
    size_t kept_cap;
    unsigned long long *floor_key; // (batch) smallest key ranked so far, same path
    size_t floor_cap;
    int *selftest_out;

    // buffer surface (ssdbox_buf.hip): buffers handed out, freed buffers kept for reuse, and the
    // pinned staging ring of the host <-> device copies
    // (a buffer may be released from another thread than the one that runs the context, e.g. by
    // a garbage collector: the two lists are guarded by buf_mu)
    std::mutex buf_mu;
    std::vector<DeviceBuf> buf_live, buf_free;
    size_t buf_cached;
    void *stage[2];
    hipEvent_t stage_ev[2];
    bool stage_busy[2];
    int stage_next;

    // This device's copy of the library's fixed scratch array (ssdbox_ctx.hip), resolved when
    // the context is created: what the entry point that allocates nothing works in.
    unsigned char *static_scratch;
    size_t static_scratch_bytes;
};

void set_error(const char *fmt, ...);
void free_buffer_state(ssdbox_ctx *ctx);   // ssdbox_buf.hip

#define SSD_CHECK_HIP(expr)                                                              \
    do {                                                                                 \
        hipError_t e_ = (expr);                                                          \
        if (e_ != hipSuccess) {                                                          \
            set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__,   \
                      __LINE__);                                                         \
            return SSDBOX_ERR_HIP;                                                       \
        }                                                                                \
    } while (0)

#define SSD_REQUIRE(cond, ...)                                                           \
    do {                                                                                 \
        if (!(cond)) {                                                                   \
            set_error(__VA_ARGS__);                                                      \
            return SSDBOX_ERR_INVALID;                                                   \
        }                                                                                \
    } while (0)

// Makes ctx->device current for the scope and restores the caller's device afterwards.
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    explicit DeviceGuard(int dev)
    {
        if (hipGetDevice(&prev) == hipSuccess && prev != dev) {
            switched = (hipSetDevice(dev) == hipSuccess);
        }
    }
    ~DeviceGuard()
    {
        if (switched) (void)hipSetDevice(prev);
    }
};

// Brackets one kernel launch with events when timing is on.
struct StageTimer {
    ssdbox_ctx *ctx;
    hipEvent_t stop = nullptr;
    StageTimer(ssdbox_ctx *c, int stage);
    ~StageTimer();
};

// Two events for one timed launch (from the context's pool), or false.
bool acquire_events(ssdbox_ctx *ctx, hipEvent_t ev[2]);

// One kernel launch on the context's stream.  With timing on, the launch itself carries the
// two events, which then hold the kernel's own begin and end timestamps: the elapsed time is
// the kernel's execution time, without the gaps an event pair recorded around it includes.
template <typename... KArgs, typename... Args>
void launch_stage_lds(ssdbox_ctx *ctx, int stage, void (*kernel)(KArgs...), dim3 grid, dim3 block,
                      unsigned lds_bytes, Args &&...args)
{
    hipEvent_t ev[2];
    if (ctx->timing && ctx->records.size() < kMaxTimingRecords && acquire_events(ctx, ev)) {
        hipExtLaunchKernelGGL<KArgs...>(kernel, grid, block, lds_bytes, ctx->stream, ev[0], ev[1], 0u,
                                        static_cast<KArgs>(args)...);
        try {
            ctx->records.push_back({stage, ev[0], ev[1]});
        } catch (...) {               // the launch stays, its timing is dropped
        }
    } else {
        hipLaunchKernelGGL(kernel, grid, block, lds_bytes, ctx->stream, static_cast<KArgs>(args)...);
    }
}

template <typename... KArgs, typename... Args>
void launch_stage(ssdbox_ctx *ctx, int stage, void (*kernel)(KArgs...), dim3 grid, dim3 block,
                  Args &&...args)
{
    launch_stage_lds(ctx, stage, kernel, grid, block, 0u, static_cast<Args &&>(args)...);
}

int ensure_buffer(void **ptr, size_t *cap, size_t need, size_t elem_size, bool zero,
                  hipStream_t stream);

// Experiment build (-DSSD_EXP_TRACE, profiles/trace_inflight.py): the kernels of the two hot paths
// stamp their begin (first workgroup) and end (every workgroup; the last store stays) on the
// device's 100 MHz wall clock into a slot of a trace buffer the host hands out per launch, so the
// overlap of several batches in flight can be read without a profiler or events in the way.
// The product build carries none of it.
#ifdef SSD_EXP_TRACE
unsigned long long *ssd_trace_slot(ssdbox_ctx *ctx, int stage);
#define SSD_TRACE_PARAM , unsigned long long *tr
#define SSD_TRACE_ARG(ctx, stage) , ssd_trace_slot(ctx, stage)
#define SSD_TRACE_BEGIN(tr)                                                                  \
    do {                                                                                     \
        if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) (tr)[0] = wall_clock64(); \
    } while (0)
#define SSD_TRACE_END(tr)                                  \
    do {                                                   \
        if (threadIdx.x == 0) (tr)[1] = wall_clock64();    \
    } while (0)
#else
#define SSD_TRACE_PARAM
#define SSD_TRACE_ARG(ctx, stage)
#define SSD_TRACE_BEGIN(tr)
#define SSD_TRACE_END(tr)
#endif

// ---------------------------------------------------------------------------------------
// Device helpers.
// ---------------------------------------------------------------------------------------
#define SSD_DEV __device__ __forceinline__

SSD_DEV uint32_t f2u(float f) { return __builtin_bit_cast(uint32_t, f); }
SSD_DEV float u2f(uint32_t u) { return __builtin_bit_cast(float, u); }

// One v_min_f32 / v_max_f32 each: the same value as fminf / fmaxf for every quiet input (a
// NaN operand yields the other one), without the canonicalising copies hipcc puts in front
// of the library calls when an operand comes straight from memory.
SSD_DEV float vmin(float a, float b)
{
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
SSD_DEV float vmax(float a, float b)
{
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// clip_by_value(x, 0, hi) of reference utils/box_utils.py:15 for hi >= 0.
SSD_DEV float clip0(float x, float hi) { return fminf(fmaxf(x, 0.0f), hi); }

// Natural log and exp as fixed sequences of IEEE float32 operations (Cephes single
// precision): every + - * / below rounds once (the library is built with
// -ffp-contract=off), so any implementation of the same sequence returns the same bits.
SSD_DEV float ssd_logf(float x)
{
    if (x != x) return x;
    if (x < 0.0f) return u2f(0x7FC00000u);
    if (x == 0.0f) return u2f(0xFF800000u);
    uint32_t u = f2u(x);
    if (u == 0x7F800000u) return x;
    int e = 0;
    if ((u >> 23) == 0) {
        x = x * 8388608.0f;
        u = f2u(x);
        e = -23;
    }
    e += (int)(u >> 23) - 126;
    float m = u2f((u & 0x007FFFFFu) | 0x3F000000u);
    if (m < 0.707106781186547524f) {
        e -= 1;
        m = m + m - 1.0f;
    } else {
        m = m - 1.0f;
    }
    float z = m * m;
    float y = 7.0376836292E-2f;
    y = y * m + -1.1514610310E-1f;
    y = y * m + 1.1676998740E-1f;
    y = y * m + -1.2420140846E-1f;
    y = y * m + 1.4249322787E-1f;
    y = y * m + -1.6668057665E-1f;
    y = y * m + 2.0000714765E-1f;
    y = y * m + -2.4999993993E-1f;
    y = y * m + 3.3333331174E-1f;
    y = y * m * z;
    float fe = (float)e;
    if (e != 0) y = y + -2.12194440e-4f * fe;
    y = y + -0.5f * z;
    z = m + y;
    if (e != 0) z = z + 0.693359375f * fe;
    return z;
}

SSD_DEV float ssd_expf(float x)
{
    if (x != x) return x;
    if (x > 88.72283905206835f) return u2f(0x7F800000u);
    if (x < -103.278929903431851103f) return 0.0f;
    float z = floorf(1.44269504088896341f * x + 0.5f);
    x = x - z * 0.693359375f;
    x = x - z * -2.12194440e-4f;
    int n = (int)z;
    z = x * x;
    float p = 1.9875691500E-4f;
    p = p * x + 1.3981999507E-3f;
    p = p * x + 8.3334519073E-3f;
    p = p * x + 4.1665795894E-2f;
    p = p * x + 1.6666665459E-1f;
    p = p * x + 5.0000001201E-1f;
    z = p * z + x + 1.0f;
    if (n > 127) {
        z = z * u2f((uint32_t)(100 + 127) << 23);
        n -= 100;
    } else if (n < -126) {
        z = z * u2f((uint32_t)(-100 + 127) << 23);
        n += 100;
    }
    return z * u2f((uint32_t)(n + 127) << 23);
}

// n / d for d >= 1 with magic = ceil(2^32 / d); exact while n * d < 2^32 (checked when the
// geometry is built).
SSD_DEV int udiv_magic(int n, int d, unsigned magic)
{
    return d == 1 ? n : (int)__umulhi((unsigned)n, magic);
}

// Anchor a of a geometry: [cx, cy, w, h] (reference utils/anchor_utils.py:28-60 order:
// layer, row, column, box).
SSD_DEV float4 geom_anchor(const GeomDev &g, const float *__restrict__ cx_tab,
                           const float *__restrict__ cy_tab, int a)
{
    // anchor_base grows with the layer, so the layer is a count; every comparison reads a
    // fixed field (no chain of dependent loads)
    int m = 0;
#pragma unroll
    for (int l = 1; l < SSDBOX_MAX_LAYERS; ++l)
        m += (l < g.n_layers && a >= g.layer[l].anchor_base) ? 1 : 0;
    const LayerDev &L = g.layer[m];
    int r = a - L.anchor_base;
    int i = udiv_magic(r, L.row_anchors, L.rowanc_magic);
    int c = r - i * L.row_anchors;
    int j = udiv_magic(c, L.n_box, L.nbox_magic);
    int k = c - j * L.n_box;
    return make_float4(cx_tab[L.cx_off + j], cy_tab[L.cy_off + i], L.bw[k], L.bh[k]);
}

SSD_DEV float4 geom_anchor(const GeomDev *__restrict__ g, const float *__restrict__ cx_tab,
                           const float *__restrict__ cy_tab, int a)
{
    return geom_anchor(*g, cx_tab, cy_tab, a);
}

// encode of reference utils/box_utils.py:101-123 for one anchor [cx,cy,w,h] and one corner
// box; vc / vs are variance[0] / variance[1].
SSD_DEV float4 encode_one(float4 anc, float4 box, float vc, float vs)
{
    float cx = (box.x + box.z) / 2.0f;
    float cy = (box.y + box.w) / 2.0f;
    float w = box.z - box.x;
    float h = box.w - box.y;
    float4 t;
    t.x = (cx - anc.x) / (anc.z * vc);
    t.y = (cy - anc.y) / (anc.w * vc);
    t.z = ssd_logf(w / anc.z) / vs;
    t.w = ssd_logf(h / anc.w) / vs;
    return t;
}

// decode of reference utils/box_utils.py:126-147 for one anchor and one offset row.
SSD_DEV float4 decode_one(float4 anc, float4 l, float vc, float vs)
{
    float cx = l.x * vc * anc.z + anc.x;
    float cy = l.y * vc * anc.w + anc.y;
    float w = ssd_expf(l.z * vs) * anc.z;
    float h = ssd_expf(l.w * vs) * anc.w;
    float hw = w / 2.0f, hh = h / 2.0f;
    return make_float4(cx - hw, cy - hh, cx + hw, cy + hh);
}

// Maximum over the 64 lanes of a wave of a value >= 0 (identity 0), returned in every lane.
SSD_DEV float wave_max_nonneg(float v)
{
    int x = (int)f2u(v);
    // row_shr 1,2,4,8 inside each row of 16 lanes, then the two cross-row broadcasts; lanes
    // without a source keep the identity 0.  Non-negative floats order like their bits.
    x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false));
    x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false));
    x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false));
    x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false));
    x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false));
    x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false));
    return u2f((uint32_t)__builtin_amdgcn_readlane(x, 63));
}

// Minimum over the 64 lanes of a wave of an unsigned value (identity 0xFFFFFFFF), returned
// in every lane.  Same data movement as wave_max_nonneg.
SSD_DEV uint32_t wave_min_u32(uint32_t v)
{
    const int id = (int)0xFFFFFFFFu;
    int x = (int)v;
    x = (int)min((uint32_t)x, (uint32_t)__builtin_amdgcn_update_dpp(id, x, 0x111, 0xf, 0xf, false));
    x = (int)min((uint32_t)x, (uint32_t)__builtin_amdgcn_update_dpp(id, x, 0x112, 0xf, 0xf, false));
    x = (int)min((uint32_t)x, (uint32_t)__builtin_amdgcn_update_dpp(id, x, 0x114, 0xf, 0xf, false));
    x = (int)min((uint32_t)x, (uint32_t)__builtin_amdgcn_update_dpp(id, x, 0x118, 0xf, 0xf, false));
    x = (int)min((uint32_t)x, (uint32_t)__builtin_amdgcn_update_dpp(id, x, 0x142, 0xa, 0xf, false));
    x = (int)min((uint32_t)x, (uint32_t)__builtin_amdgcn_update_dpp(id, x, 0x143, 0xc, 0xf, false));
    return (uint32_t)__builtin_amdgcn_readlane(x, 63);
}

SSD_DEV int lane_id() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// Rank key of a score: larger key = ranks earlier.  -0 is folded into +0 (they compare
// equal, so order falls to the index), NaN ranks last.
SSD_DEV uint32_t score_key(float s)
{
    if (s != s) return 0u;
    s = s + 0.0f;
    uint32_t u = f2u(s);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

SSD_DEV unsigned long long rank_key(float s, uint32_t index)
{
    return ((unsigned long long)score_key(s) << 32) | (unsigned long long)(0xFFFFFFFFu - index);
}
