// Training side of the SSD box pipeline: anchors x ground-truth IoU, matching and offset
// encoding for a whole batch (reference utils/box_utils.py:47-98, called per image from
// utils/datasets.py:152).
//
// Kernel 1 (match_kernel): one thread per anchor.  With a geometry the workgroup owns an
// 8-row x 32-anchor-column tile of one feature map and builds its anchors from the centre
// tables; with a materialised anchor array it owns 256 consecutive anchors.  Ground-truth
// boxes whose bounding box cannot touch the tile are dropped by a ballot compaction into
// LDS: their IoU with every anchor of the tile is exactly +0, which can neither become a
// first maximum (the running best starts at +0 / index 0) nor a best anchor of that box
// (a zero key leaves the default "anchor 0").  The best anchor of each box is a wave-level
// maximum, folded through LDS and then global memory as a packed 64-bit
// (IoU bits, ~anchor index) atomic max, so ties resolve to the lowest anchor index.
//
// Kernel 2 (force_kernel): per image, the forced scatter of utils/box_utils.py:72-87 in
// ascending ground-truth order (the highest box index wins a shared anchor), rewriting the
// <= G affected anchors, and clearing the keys for the next call.
#include "ssdbox_internal.h"

namespace {

struct MatchArgs {
    const GeomDev *geom;
    const float *cx_tab;
    const float *cy_tab;
    const float4 *anchors;
    int num_anchors;
    const float4 *gt_boxes;
    const float *gt_labels;
    const int *gt_count;
    int gmax;
    float max_size;
    float thr;
    float *conf;
    float4 *locs;
    float *y_true;
    int *match_idx;
    float *match_iou;
    unsigned long long *gkey;
};

// One anchor's result in either output layout.  y_true rows are
// [one_hot(int(conf), 2), locs] (reference utils/datasets.py:155-159).
template <bool YTRUE>
SSD_DEV void store_target(const MatchArgs &p, long long row, float conf, float4 t, int idx, float iou)
{
    if (YTRUE) {
        int c = (int)conf;
        float2 *y = reinterpret_cast<float2 *>(p.y_true + row * 6);
        y[0] = make_float2(c == 0 ? 1.0f : 0.0f, c == 1 ? 1.0f : 0.0f);
        y[1] = make_float2(t.x, t.y);
        y[2] = make_float2(t.z, t.w);
    } else {
        p.conf[row] = conf;
        p.locs[row] = t;
        if (p.match_idx) p.match_idx[row] = idx;
        if (p.match_iou) p.match_iou[row] = iou;
    }
}

template <bool GEOM, bool YTRUE>
__global__ __launch_bounds__(kBlock) void match_kernel(MatchArgs p)
{
    __shared__ float4 s_box[kBlock];
    __shared__ float s_area[kBlock];
    __shared__ int s_gidx[kBlock];
    __shared__ unsigned long long s_key[kBlock];
    __shared__ int s_wcnt[kBlock / 64];
    __shared__ float s_ext[4][kBlock / 64];

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = t >> 6;
    const int img = blockIdx.y;

    int a;
    bool valid;
    float4 anc;
    float ex0, ey0, ex1, ey1;      // bounding box of every anchor of this workgroup

    if (GEOM) {
        const GeomDev *g = p.geom;
        const int tile = blockIdx.x;
        int m = 0;
        const int nl = g->n_layers;
        while (m + 1 < nl && tile >= g->layer[m + 1].tile_base) ++m;
        const LayerDev &L = g->layer[m];
        const int tt = tile - L.tile_base;
        const int ty = tt / L.tiles_x;
        const int tx = tt - ty * L.tiles_x;
        const int row = ty * kTileRows + (t >> 5);
        const int col = tx * kTileCols + (t & 31);
        valid = row < L.fm_h && col < L.row_anchors;
        const int rowc = min(row, L.fm_h - 1);
        const int colc = min(col, L.row_anchors - 1);
        const int j = colc / L.n_box;
        const int k = colc - j * L.n_box;
        a = L.anchor_base + rowc * L.row_anchors + colc;
        anc = make_float4(p.cx_tab[L.cx_off + j], p.cy_tab[L.cy_off + rowc], L.bw[k], L.bh[k]);
        // Corner rounding is monotonic, so the first/last centre with the widest box bound
        // every corner of the tile.
        const int r0 = ty * kTileRows;
        const int r1 = min(r0 + kTileRows, L.fm_h) - 1;
        const int c0 = tx * kTileCols;
        const int c1 = min(c0 + kTileCols, L.row_anchors) - 1;
        const float hw = L.wmax / 2.0f, hh = L.hmax / 2.0f;
        ex0 = p.cx_tab[L.cx_off + c0 / L.n_box] - hw;
        ex1 = p.cx_tab[L.cx_off + c1 / L.n_box] + hw;
        ey0 = p.cy_tab[L.cy_off + r0] - hh;
        ey1 = p.cy_tab[L.cy_off + r1] + hh;
    } else {
        a = blockIdx.x * kBlock + t;
        valid = a < p.num_anchors;
        a = min(a, p.num_anchors - 1);
        anc = p.anchors[a];
    }

    // transform_center_to_corner, utils/box_utils.py:177-179
    const float ahw = anc.z / 2.0f, ahh = anc.w / 2.0f;
    const float ax0 = anc.x - ahw, ay0 = anc.y - ahh, ax1 = anc.x + ahw, ay1 = anc.y + ahh;
    const float a_area = clip0(ax1 - ax0, p.max_size) * clip0(ay1 - ay0, p.max_size);

    if (!GEOM) {
        float lo_x = ax0, lo_y = ay0, hi_x = ax1, hi_y = ay1;
        for (int o = 32; o > 0; o >>= 1) {
            lo_x = fminf(lo_x, __shfl_xor(lo_x, o));
            lo_y = fminf(lo_y, __shfl_xor(lo_y, o));
            hi_x = fmaxf(hi_x, __shfl_xor(hi_x, o));
            hi_y = fmaxf(hi_y, __shfl_xor(hi_y, o));
        }
        if (lane == 0) {
            s_ext[0][wave] = lo_x;
            s_ext[1][wave] = lo_y;
            s_ext[2][wave] = hi_x;
            s_ext[3][wave] = hi_y;
        }
        __syncthreads();
        ex0 = fminf(fminf(s_ext[0][0], s_ext[0][1]), fminf(s_ext[0][2], s_ext[0][3]));
        ey0 = fminf(fminf(s_ext[1][0], s_ext[1][1]), fminf(s_ext[1][2], s_ext[1][3]));
        ex1 = fmaxf(fmaxf(s_ext[2][0], s_ext[2][1]), fmaxf(s_ext[2][2], s_ext[2][3]));
        ey1 = fmaxf(fmaxf(s_ext[3][0], s_ext[3][1]), fmaxf(s_ext[3][2], s_ext[3][3]));
    }

    const int G = min(p.gt_count[img], p.gmax);
    const float4 *gt = p.gt_boxes + (long long)img * p.gmax;
    unsigned long long *gkey = p.gkey + (long long)img * p.gmax;

    float best = 0.0f;     // reduce_max / argmax over GT, first maximum wins (:66-67)
    int bidx = 0;

    for (int g0 = 0; g0 < G; g0 += kBlock) {
        const int g = g0 + t;
        bool keep = false;
        float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
        if (g < G) {
            b = gt[g];
            // dropped only when provably disjoint from every anchor here (NaN is kept)
            keep = !(b.z <= ex0) && !(b.x >= ex1) && !(b.w <= ey0) && !(b.y >= ey1);
        }
        const unsigned long long bal = __ballot(keep);
        if (lane == 0) s_wcnt[wave] = __popcll(bal);
        s_key[t] = 0ull;
        __syncthreads();
        int off = __popcll(bal & ((1ull << lane) - 1ull));
        int n_list = 0;
#pragma unroll
        for (int w = 0; w < kBlock / 64; ++w) {
            if (w < wave) off += s_wcnt[w];
            n_list += s_wcnt[w];
        }
        if (keep) {
            s_box[off] = b;
            s_area[off] = clip0(b.z - b.x, p.max_size) * clip0(b.w - b.y, p.max_size);
            s_gidx[off] = g;
        }
        __syncthreads();

        for (int q = 0; q < n_list; ++q) {
            const float4 gb = s_box[q];
            // compute_iou, utils/box_utils.py:35-42
            const float iw = fminf(ax1, gb.z) - fmaxf(ax0, gb.x);
            const float ih = fminf(ay1, gb.w) - fmaxf(ay0, gb.y);
            float v = 0.0f;
            if (valid && iw > 0.0f && ih > 0.0f) {
                const float inter = fminf(iw, p.max_size) * fminf(ih, p.max_size);
                const float uni = (a_area + s_area[q]) - inter;
                const float iou = inter / uni;
                if (iou > best) {
                    best = iou;
                    bidx = s_gidx[q];
                }
                v = (iou > 0.0f) ? iou : 0.0f;
            }
            if (__ballot(v > 0.0f) != 0ull) {
                // argmax over anchors (:70): wave maximum, lowest lane = lowest anchor
                const float mx = wave_max_nonneg(v);
                const unsigned long long sel = __ballot(v == mx);
                const int src = __builtin_ctzll(sel);
                const int a_src = __builtin_amdgcn_readlane(a, src);
                if (lane == 0) {
                    const unsigned long long key =
                        ((unsigned long long)f2u(mx) << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)a_src);
                    if (key > s_key[q]) atomicMax(&s_key[q], key);
                }
            }
        }
        __syncthreads();
        if (t < n_list) {
            const unsigned long long key = s_key[t];
            if (key != 0ull) atomicMax(&gkey[s_gidx[t]], key);
        }
        __syncthreads();
    }

    if (valid) {
        // gather labels / boxes, threshold, encode every anchor (:89-96)
        const float4 gb = gt[bidx];
        const float label = p.gt_labels[(long long)img * p.gmax + bidx];
        const float conf = (best < p.thr) ? 0.0f : label;
        const float4 tgt = encode_one(anc, gb, 0.1f, 0.2f);
        store_target<YTRUE>(p, (long long)img * p.num_anchors + a, conf, tgt, bidx, best);
    }
}

template <bool GEOM, bool YTRUE>
__global__ __launch_bounds__(kBlock) void force_kernel(MatchArgs p)
{
    __shared__ int s_anc[SSDBOX_MAX_GT];
    const int t = threadIdx.x;
    const int img = blockIdx.x;
    const int G = min(p.gt_count[img], p.gmax);
    const float4 *gt = p.gt_boxes + (long long)img * p.gmax;
    unsigned long long *gkey = p.gkey + (long long)img * p.gmax;

    for (int g = t; g < G; g += kBlock) {
        const unsigned long long key = gkey[g];
        // no anchor overlaps this box: argmax of an all-zero column is anchor 0
        s_anc[g] = key ? (int)(0xFFFFFFFFu - (uint32_t)key) : 0;
        gkey[g] = 0ull;
    }
    __syncthreads();
    for (int g = t; g < G; g += kBlock) {
        const int a = s_anc[g];
        bool later = false;            // a later box claims the same anchor: it wins
        for (int g2 = g + 1; g2 < G; ++g2) {
            if (s_anc[g2] == a) {
                later = true;
                break;
            }
        }
        if (later) continue;
        const float4 anc = GEOM ? geom_anchor(p.geom, p.cx_tab, p.cy_tab, a) : p.anchors[a];
        const float label = p.gt_labels[(long long)img * p.gmax + g];
        const float conf = (1.0f < p.thr) ? 0.0f : label;
        const float4 tgt = encode_one(anc, gt[g], 0.1f, 0.2f);
        store_target<YTRUE>(p, (long long)img * p.num_anchors + a, conf, tgt, g, 1.0f);
    }
}

template <bool GEOM, bool YTRUE>
int launch_match(ssdbox_ctx *ctx, const MatchArgs &args, int batch, int n_blocks_x)
{
    {
        StageTimer timer(ctx, SSDBOX_STAGE_MATCH);
        hipLaunchKernelGGL((match_kernel<GEOM, YTRUE>), dim3(n_blocks_x, batch), dim3(kBlock), 0,
                           ctx->stream, args);
    }
    SSD_CHECK_HIP(hipGetLastError());
    {
        StageTimer timer(ctx, SSDBOX_STAGE_FORCE);
        hipLaunchKernelGGL((force_kernel<GEOM, YTRUE>), dim3(batch), dim3(kBlock), 0, ctx->stream,
                           args);
    }
    SSD_CHECK_HIP(hipGetLastError());
    return SSDBOX_OK;
}

int match_common(ssdbox_ctx *ctx, const ssdbox_geom *geom, const float *d_anchors,
                 int64_t num_anchors, const float *d_gt_boxes, const float *d_gt_labels,
                 const int32_t *d_gt_count, int batch, int gmax, float max_size,
                 float iou_threshold, float *d_conf, float *d_locs, float *d_y_true,
                 int32_t *d_match_idx, float *d_match_iou)
{
    SSD_REQUIRE(ctx != nullptr, "match_encode: ctx is NULL");
    SSD_REQUIRE((geom != nullptr) != (d_anchors != nullptr),
                "match_encode: pass exactly one of geom and d_anchors");
    SSD_REQUIRE(d_gt_boxes && d_gt_labels && d_gt_count, "match_encode: NULL ground-truth pointer");
    SSD_REQUIRE(d_y_true != nullptr || (d_conf != nullptr && d_locs != nullptr),
                "match_encode: NULL output pointer");
    SSD_REQUIRE(batch >= 1 && batch <= 65535, "match_encode: batch %d outside [1, 65535]", batch);
    SSD_REQUIRE(gmax >= 1 && gmax <= SSDBOX_MAX_GT, "match_encode: gmax %d outside [1, %d]", gmax,
                SSDBOX_MAX_GT);
    SSD_REQUIRE(max_size > 0.0f, "match_encode: max_size must be positive");
    SSD_REQUIRE(iou_threshold == iou_threshold, "match_encode: iou_threshold is NaN");
    if (geom != nullptr) {
        SSD_REQUIRE(geom->d_geom != nullptr, "match_encode: geometry handle is host-only");
        SSD_REQUIRE(geom->device == ctx->device, "match_encode: geometry lives on device %d, context on %d",
                    geom->device, ctx->device);
        SSD_REQUIRE(num_anchors == geom->host.num_anchors,
                    "match_encode: num_anchors %lld does not match the geometry's %d",
                    (long long)num_anchors, geom->host.num_anchors);
    }
    SSD_REQUIRE(num_anchors >= 1 && num_anchors < (1ll << 30), "match_encode: num_anchors %lld",
                (long long)num_anchors);

    DeviceGuard guard(ctx->device);
    const size_t need = (size_t)batch * (size_t)gmax;
    int rc = ensure_buffer((void **)&ctx->gkey, &ctx->gkey_cap, need, sizeof(unsigned long long),
                           true, ctx->stream);
    if (rc != SSDBOX_OK) return rc;
    if (ctx->gkey_dirty) {
        SSD_CHECK_HIP(hipMemsetAsync(ctx->gkey, 0, ctx->gkey_cap * sizeof(unsigned long long), ctx->stream));
        ctx->gkey_dirty = false;
    }

    MatchArgs args;
    args.geom = geom ? geom->d_geom : nullptr;
    args.cx_tab = geom ? geom->d_cx : nullptr;
    args.cy_tab = geom ? geom->d_cy : nullptr;
    args.anchors = reinterpret_cast<const float4 *>(d_anchors);
    args.num_anchors = (int)num_anchors;
    args.gt_boxes = reinterpret_cast<const float4 *>(d_gt_boxes);
    args.gt_labels = d_gt_labels;
    args.gt_count = d_gt_count;
    args.gmax = gmax;
    args.max_size = max_size;
    args.thr = iou_threshold;
    args.conf = d_conf;
    args.locs = reinterpret_cast<float4 *>(d_locs);
    args.y_true = d_y_true;
    args.match_idx = d_match_idx;
    args.match_iou = d_match_iou;
    args.gkey = ctx->gkey;

    ctx->gkey_dirty = true;    // cleared again only when both launches were enqueued
    if (geom != nullptr) {
        const int nx = geom->host.n_tiles;
        rc = d_y_true ? launch_match<true, true>(ctx, args, batch, nx)
                      : launch_match<true, false>(ctx, args, batch, nx);
    } else {
        const int nx = (int)((num_anchors + kBlock - 1) / kBlock);
        rc = d_y_true ? launch_match<false, true>(ctx, args, batch, nx)
                      : launch_match<false, false>(ctx, args, batch, nx);
    }
    if (rc == SSDBOX_OK) ctx->gkey_dirty = false;
    return rc;
}

}  // namespace

extern "C" int ssdbox_match_encode(ssdbox_ctx *ctx, const ssdbox_geom *geom, const float *d_anchors,
                                   int64_t num_anchors, const float *d_gt_boxes,
                                   const float *d_gt_labels, const int32_t *d_gt_count, int batch,
                                   int gmax, float max_size, float iou_threshold, float *d_conf,
                                   float *d_locs, int32_t *d_match_idx, float *d_match_iou)
{
    return match_common(ctx, geom, d_anchors, num_anchors, d_gt_boxes, d_gt_labels, d_gt_count,
                        batch, gmax, max_size, iou_threshold, d_conf, d_locs, nullptr, d_match_idx,
                        d_match_iou);
}

extern "C" int ssdbox_match_encode_ytrue(ssdbox_ctx *ctx, const ssdbox_geom *geom,
                                         const float *d_anchors, int64_t num_anchors,
                                         const float *d_gt_boxes, const float *d_gt_labels,
                                         const int32_t *d_gt_count, int batch, int gmax,
                                         float max_size, float iou_threshold, float *d_y_true)
{
    SSD_REQUIRE(d_y_true != nullptr, "match_encode_ytrue: d_y_true is NULL");
    return match_common(ctx, geom, d_anchors, num_anchors, d_gt_boxes, d_gt_labels, d_gt_count,
                        batch, gmax, max_size, iou_threshold, nullptr, nullptr, d_y_true, nullptr,
                        nullptr);
}
