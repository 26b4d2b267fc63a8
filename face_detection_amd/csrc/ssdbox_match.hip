// Training side of the SSD box pipeline: anchors x ground-truth IoU, matching and offset
// encoding for a whole batch (reference utils/box_utils.py:47-98, called per image from
// utils/datasets.py:152).
//
// Kernel 1 (match_kernel).  With a geometry a workgroup owns a tile of one feature map and
// builds its anchors from the centre tables; two tile shapes exist, chosen per layer when
// the geometry is created:
//   * 32 rows x 32 anchor columns, a thread owning one column and four rows of it (rows
//     share the column's x extent, so the x overlap with a box is computed once) -- fine
//     maps, where few boxes come near a tile;
//   * 8 rows x 32 anchor columns, one anchor per thread -- coarse maps, whose big anchors
//     overlap most boxes: the walk over the tile's box list costs a quarter per box, and
//     these tiles get the lowest block indices so that the longest ones start first.
// With a materialised anchor array a workgroup owns 256 consecutive anchors.
// Ground-truth boxes whose bounding box cannot touch the tile are dropped by a ballot
// compaction into LDS: their IoU with every anchor of the tile is exactly +0, which can
// neither become a first maximum (the running best starts at +0 / index 0) nor a best anchor
// of that box (a zero key leaves the default "anchor 0").  The best anchor of each box is a
// wave-level maximum, folded through LDS and then global memory as a packed 64-bit
// (IoU bits, ~anchor index) atomic max, so ties resolve to the lowest anchor index.
// Anchors that overlap nothing match box 0; their offsets depend only on the column (x), on
// row and box shape (y) and on the box shape (w, h), so they come from three small per-tile
// LDS tables instead of six divisions and two logarithms per anchor.
//
// Kernel 2 (force_kernel): per image, the forced scatter of utils/box_utils.py:72-87 in
// ascending ground-truth order (the highest box index wins a shared anchor), rewriting the
// <= G affected anchors, and clearing the keys for the next call.
#include <string.h>

#include "ssdbox_internal.h"

namespace {

typedef unsigned long long u64;

struct MatchArgs {
    GeomDev geom;                  // by value: layer constants arrive with the kernel arguments
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
    u64 *gkey;
};

struct MatchShared {
    float4 box[kBlock];            // compacted ground-truth list of the current chunk
    float area[kBlock];
    int gidx[kBlock];
    u64 key[kBlock];               // best (IoU, ~anchor) of each listed box within this tile
    int wcnt[2][kBlock / 64];
    float ext[4][kBlock / 64];
    // Offsets of every anchor of the tile against ground-truth box 0 (see the file header).
    float tx0[kTileCols];
    float ty0[kTileRows * SSDBOX_MAX_BOXES_PER_CELL];
    float2 twh0[SSDBOX_MAX_BOXES_PER_CELL];
};

// One anchor's result in either output layout.  y_true rows are
// [one_hot(int(conf), 2), locs] (reference utils/datasets.py:155-159).
template <bool YTRUE>
SSD_DEV void store_target(const MatchArgs &p, long long row, float conf, float4 t, int idx, float iou)
{
#ifdef SSD_EXP_NO_STORE      // timing experiment only: keep the values live, skip the stores
    if (conf != 12345.678f) return;
#endif
    if (YTRUE) {
        int c = (int)conf;
        float2 *y = reinterpret_cast<float2 *>(p.y_true + row * 6);
        y[0] = make_float2(c == 0 ? 1.0f : 0.0f, c == 1 ? 1.0f : 0.0f);
        y[1] = make_float2(t.x, t.y);
        y[2] = make_float2(t.z, t.w);
    } else {
#ifdef SSD_EXP_NO_CONF       // timing experiments only
        if (conf == 12345.678f)
#endif
        p.conf[row] = conf;
#ifdef SSD_EXP_NO_LOCS
        if (conf == 12345.678f)
#endif
        p.locs[row] = t;
#ifndef SSD_EXP_STAMPS
        if (p.match_idx) p.match_idx[row] = idx;
        if (p.match_iou) p.match_iou[row] = iou;
#endif
    }
}

// R anchors (rows of one anchor column) per thread; a tile is 8*R rows x 32 anchor columns.
template <bool GEOM, bool YTRUE, int R>
SSD_DEV void match_tile(const MatchArgs &p, MatchShared &sm, int m)
{
    constexpr int TR = 8 * R;      // rows per tile

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = t >> 6;
    const int slot = t;            // anchor slot inside the tile
    const int img = blockIdx.y;
    const float4 *gt = p.gt_boxes + (long long)img * p.gmax;
    // The count and the first 256 boxes are fetched together with everything else the
    // prologue reads, so a tile pays one memory round trip before it can start.
    const int G = max(min(p.gt_count[img], p.gmax), 1);
    const float4 b_first = gt[min(t, p.gmax - 1)];
#ifdef SSD_EXP_STAMPS        // timing experiment only: per-tile clock stamps into match_idx
    const u64 st_rt0 = __builtin_amdgcn_s_memrealtime(), st_t0 = __builtin_amdgcn_s_memtime();
    u64 st_t1 = 0, st_t2 = 0;
    int st_nlist = 0, st_evals = 0;
#endif

    int a[R];
    bool valid[R];
    float acy[R], ay0[R], ay1[R], a_area[R];
    float acx, aw, ah, ax0, ax1;
    float ex0, ey0, ex1, ey1;      // bounding box of every anchor of this workgroup
    float wy0 = -3.0e38f, wy1 = 3.0e38f;   // y extent of this wave's anchors (whole-tile modes)
    int k = 0, n_box = 1;
    // Row of anchor slot `s`, i-th anchor of the thread, inside the tile: a wave owns 2*R
    // consecutive rows, lanes 0-31 / 32-63 two neighbouring ones.
    auto row_in_tile = [](int s, int i) { return 2 * R * (s >> 6) + ((s >> 5) & 1) + 2 * i; };

    if (GEOM) {
        const LayerDev &L = p.geom.layer[m];
        n_box = L.n_box;
        const int tt = (int)blockIdx.x - L.tile_base;
        const int ty = udiv_magic(tt, L.tiles_x, L.tilesx_magic);
        const int tx = tt - ty * L.tiles_x;
        const int col = tx * kTileCols + (slot & 31);
        const int colc = min(col, L.row_anchors - 1);
        const int j = udiv_magic(colc, n_box, L.nbox_magic);
        k = colc - j * n_box;
        acx = p.cx_tab[L.cx_off + j];
        aw = L.bw[k];
        ah = L.bh[k];
        const int r0 = ty * TR;
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const int row = r0 + row_in_tile(slot, i);
            valid[i] = row < L.fm_h && col < L.row_anchors;
            const int rowc = min(row, L.fm_h - 1);
            a[i] = L.anchor_base + rowc * L.row_anchors + colc;
            acy[i] = p.cy_tab[L.cy_off + rowc];
        }
        // rows of this wave are contiguous: boxes above or below them are skipped whole
        const int wr0 = min(r0 + 2 * R * wave, L.fm_h - 1);
        const int wr1 = min(r0 + 2 * R * wave + 2 * R - 1, L.fm_h - 1);
        wy0 = p.cy_tab[L.cy_off + wr0] - L.hmax / 2.0f;
        wy1 = p.cy_tab[L.cy_off + wr1] + L.hmax / 2.0f;
        // Corner rounding is monotonic, so the first/last centre with the widest box bound
        // every corner of the tile.
        const int r1 = min(r0 + TR, L.fm_h) - 1;
        const int c0 = tx * kTileCols;
        const int c1 = min(c0 + kTileCols, L.row_anchors) - 1;
        const float hw = L.wmax / 2.0f, hh = L.hmax / 2.0f;
        ex0 = p.cx_tab[L.cx_off + udiv_magic(c0, n_box, L.nbox_magic)] - hw;
        ex1 = p.cx_tab[L.cx_off + udiv_magic(c1, n_box, L.nbox_magic)] + hw;
        ey0 = p.cy_tab[L.cy_off + r0] - hh;
        ey1 = p.cy_tab[L.cy_off + r1] + hh;

        // encode (utils/box_utils.py:115-121) of box 0 for the whole tile
        const float4 b0 = gt[0];
        if (t < kTileCols) sm.tx0[t] = ((b0.x + b0.z) / 2.0f - acx) / (aw * 0.1f);
        for (int u = t; u < TR * n_box; u += kBlock) {
            const int r = udiv_magic(u, n_box, L.nbox_magic);
            const int kk = u - r * n_box;
            const float cy = p.cy_tab[L.cy_off + min(r0 + r, L.fm_h - 1)];
            sm.ty0[u] = ((b0.y + b0.w) / 2.0f - cy) / (L.bh[kk] * 0.1f);
        }
        if (t >= 192 && t < 192 + n_box) {
            const int kk = t - 192;
            sm.twh0[kk] = make_float2(ssd_logf((b0.z - b0.x) / L.bw[kk]) / 0.2f,
                                      ssd_logf((b0.w - b0.y) / L.bh[kk]) / 0.2f);
        }
    } else {
        const int ai = blockIdx.x * kBlock + t;
        valid[0] = ai < p.num_anchors;
        a[0] = min(ai, p.num_anchors - 1);
        const float4 anc = p.anchors[a[0]];
        acx = anc.x;
        acy[0] = anc.y;
        aw = anc.z;
        ah = anc.w;
    }

    // transform_center_to_corner, utils/box_utils.py:177-179
    const float ahw = aw / 2.0f, ahh = ah / 2.0f;
    ax0 = acx - ahw;
    ax1 = acx + ahw;
    const float a_w = clip0(ax1 - ax0, p.max_size);
#pragma unroll
    for (int i = 0; i < R; ++i) {
        ay0[i] = acy[i] - ahh;
        ay1[i] = acy[i] + ahh;
        a_area[i] = a_w * clip0(ay1[i] - ay0[i], p.max_size);
    }

    if (!GEOM) {
        float lo_x = ax0, lo_y = ay0[0], hi_x = ax1, hi_y = ay1[0];
        for (int o = 32; o > 0; o >>= 1) {
            lo_x = fminf(lo_x, __shfl_xor(lo_x, o));
            lo_y = fminf(lo_y, __shfl_xor(lo_y, o));
            hi_x = fmaxf(hi_x, __shfl_xor(hi_x, o));
            hi_y = fmaxf(hi_y, __shfl_xor(hi_y, o));
        }
        if (lane == 0) {
            sm.ext[0][wave] = lo_x;
            sm.ext[1][wave] = lo_y;
            sm.ext[2][wave] = hi_x;
            sm.ext[3][wave] = hi_y;
        }
        __syncthreads();
        ex0 = fminf(fminf(sm.ext[0][0], sm.ext[0][1]), fminf(sm.ext[0][2], sm.ext[0][3]));
        ey0 = fminf(fminf(sm.ext[1][0], sm.ext[1][1]), fminf(sm.ext[1][2], sm.ext[1][3]));
        ex1 = fmaxf(fmaxf(sm.ext[2][0], sm.ext[2][1]), fmaxf(sm.ext[2][2], sm.ext[2][3]));
        ey1 = fmaxf(fmaxf(sm.ext[3][0], sm.ext[3][1]), fmaxf(sm.ext[3][2], sm.ext[3][3]));
    }

    u64 *gkey = p.gkey + (long long)img * p.gmax;
    float best[R];         // reduce_max / argmax over GT, first maximum wins (:66-67)
    int bidx[R];
#pragma unroll
    for (int i = 0; i < R; ++i) {
        best[i] = 0.0f;
        bidx[i] = 0;
    }

#ifdef SSD_EXP_STAMPS
    st_t1 = __builtin_amdgcn_s_memtime();
#endif
    int par = 0;
#ifdef SSD_EXP_NO_LOOP       // timing experiment only: no ground-truth loop at all
    for (int g0 = 0; g0 < G - 1000000; g0 += kBlock, par ^= 1) {
#else
    for (int g0 = 0; g0 < G; g0 += kBlock, par ^= 1) {
#endif
        const int g = g0 + t;
        bool keep = false;
        float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
        if (g < G) {
            b = (g0 == 0) ? b_first : gt[g];
            // dropped only when provably disjoint from every anchor here (NaN is kept)
            keep = !(b.z <= ex0) && !(b.x >= ex1) && !(b.w <= ey0) && !(b.y >= ey1);
        }
        const u64 bal = __ballot(keep);
        if (lane == 0) sm.wcnt[par][wave] = __popcll(bal);
        __syncthreads();
        int off = __popcll(bal & ((1ull << lane) - 1ull));
        int n_list = 0;
#pragma unroll
        for (int w = 0; w < kBlock / 64; ++w) {
            const int c = sm.wcnt[par][w];
            if (w < wave) off += c;
            n_list += c;
        }
#ifdef SSD_EXP_STAMPS
        st_nlist += n_list;
#endif
        if (n_list == 0) continue;      // nothing near this tile (the usual case on fine maps)
        if (keep) {
            sm.box[off] = b;
            sm.area[off] = clip0(b.z - b.x, p.max_size) * clip0(b.w - b.y, p.max_size);
            sm.gidx[off] = g;
        }
        if (t < n_list) sm.key[t] = 0ull;
        __syncthreads();

        for (int q = 0; q < n_list; ++q) {
            const float4 gb = sm.box[q];
            // compute_iou, utils/box_utils.py:35-42; the x overlap is shared by the rows
            const float iw = vmin(ax1, gb.z) - vmax(ax0, gb.x);
            // no lane of the wave can overlap this box: every IoU is +0, nothing to update
            if (__ballot(iw > 0.0f && !(gb.w <= wy0) && !(gb.y >= wy1)) == 0ull) continue;
#ifdef SSD_EXP_STAMPS
            st_evals += 1;
#endif
            // The rows are evaluated without branches (an IoU of a non-overlapping pair is
            // replaced by +0 afterwards) so that their divisions overlap.
            const float cw = vmin(iw, p.max_size);
            const float garea = sm.area[q];
            const int gidx = sm.gidx[q];
            float v = 0.0f;            // this lane's best IoU with box q, lowest row on ties
            int av = a[0];             // and the anchor it belongs to
#pragma unroll
            for (int i = 0; i < R; ++i) {
                const float ih = vmin(ay1[i], gb.w) - vmax(ay0[i], gb.y);
                const float inter = cw * vmin(ih, p.max_size);
                const float uni = (a_area[i] + garea) - inter;
                const float q_iou = inter / uni;
                const float iou = (valid[i] && iw > 0.0f && ih > 0.0f) ? q_iou : 0.0f;
                if (iou > best[i]) {
                    best[i] = iou;
                    bidx[i] = gidx;
                }
                if (iou > v) {
                    v = iou;
                    av = a[i];
                }
            }
            // Only a lane at or above the tile's best IoU so far for this box can change it.
            const float cur = u2f((uint32_t)(sm.key[q] >> 32));
            if (__ballot(v > 0.0f && v >= cur) != 0ull) {
                // argmax over anchors (:70): wave maximum, then the lowest anchor holding it
                const float mx = wave_max_nonneg(v);
                const int a_src = (int)wave_min_u32((v == mx) ? (uint32_t)av : 0xFFFFFFFFu);
                if (lane == 0) {
                    const u64 key = ((u64)f2u(mx) << 32) | (u64)(0xFFFFFFFFu - (uint32_t)a_src);
                    if (key > sm.key[q]) atomicMax(&sm.key[q], key);
                }
            }
        }
        __syncthreads();
        if (t < n_list) {
            const u64 key = sm.key[t];
            if (key != 0ull) atomicMax(&gkey[sm.gidx[t]], key);
        }
        __syncthreads();
    }

#ifdef SSD_EXP_STAMPS
    st_t2 = __builtin_amdgcn_s_memtime();
#endif
    if (GEOM) __syncthreads();         // the box-0 tables, whatever path the loop took

    // gather labels / boxes, threshold, encode every anchor (:89-96)
#pragma unroll
    for (int i = 0; i < R; ++i) {
        if (!valid[i]) continue;
        float conf = 0.0f;
        if (!(best[i] < p.thr)) conf = p.gt_labels[(long long)img * p.gmax + bidx[i]];
        float4 tgt;
        if (GEOM && bidx[i] == 0) {
            const float2 twh = sm.twh0[k];
            tgt = make_float4(sm.tx0[slot & (kTileCols - 1)], sm.ty0[row_in_tile(slot, i) * n_box + k],
                              twh.x, twh.y);
        } else {
            tgt = encode_one(make_float4(acx, acy[i], aw, ah), gt[bidx[i]], 0.1f, 0.2f);
        }
        store_target<YTRUE>(p, (long long)img * p.num_anchors + a[i], conf, tgt, bidx[i], best[i]);
    }
#ifdef SSD_EXP_STAMPS
    if (t == 0 && p.match_idx) {
        u64 *o = reinterpret_cast<u64 *>(p.match_idx) + ((long long)blockIdx.y * gridDim.x + blockIdx.x) * 8;
        o[0] = st_rt0;
        o[1] = __builtin_amdgcn_s_memrealtime();
        o[2] = st_t1 - st_t0;
        o[3] = st_t2 - st_t1;
        o[4] = __builtin_amdgcn_s_memtime() - st_t2;
        o[5] = (u64)m | ((u64)R << 8);
        o[6] = (u64)st_nlist;
        o[7] = (u64)st_evals;
    }
#endif
}

template <bool GEOM, bool YTRUE>
__global__ __launch_bounds__(kBlock) void match_kernel(const MatchArgs p)
{
    __shared__ MatchShared sm;
    if (GEOM) {
        const int tile = blockIdx.x;
        int m = p.geom.n_layers - 1;       // tile ranges run from the last layer to the first
        while (m > 0 && tile >= p.geom.layer[m - 1].tile_base) --m;
        if (p.geom.layer[m].coarse)
            match_tile<true, YTRUE, 1>(p, sm, m);
        else
            match_tile<true, YTRUE, kRowsPerThread>(p, sm, m);
    } else {
        match_tile<false, YTRUE, 1>(p, sm, 0);
    }
}

template <bool GEOM, bool YTRUE>
__global__ __launch_bounds__(kBlock) void force_kernel(const MatchArgs p, const GeomDev *__restrict__ d_geom)
{
    __shared__ int s_anc[SSDBOX_MAX_GT];
    const int t = threadIdx.x;
    const int img = blockIdx.x;
    const int G = max(min(p.gt_count[img], p.gmax), 1);
    const float4 *gt = p.gt_boxes + (long long)img * p.gmax;
    u64 *gkey = p.gkey + (long long)img * p.gmax;

    for (int g = t; g < G; g += kBlock) {
        const u64 key = gkey[g];
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
        const float4 anc = GEOM ? geom_anchor(d_geom, p.cx_tab, p.cy_tab, a) : p.anchors[a];
        const float label = p.gt_labels[(long long)img * p.gmax + g];
        const float conf = (1.0f < p.thr) ? 0.0f : label;
        const float4 tgt = encode_one(anc, gt[g], 0.1f, 0.2f);
        store_target<YTRUE>(p, (long long)img * p.num_anchors + a, conf, tgt, g, 1.0f);
    }
}

template <bool GEOM, bool YTRUE>
int launch_match(ssdbox_ctx *ctx, const MatchArgs &args, const GeomDev *d_geom, int batch, int n_blocks_x)
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
                           args, d_geom);
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
    int rc = ensure_buffer((void **)&ctx->gkey, &ctx->gkey_cap, need, sizeof(u64), true, ctx->stream);
    if (rc != SSDBOX_OK) return rc;
    if (ctx->gkey_dirty) {
        SSD_CHECK_HIP(hipMemsetAsync(ctx->gkey, 0, ctx->gkey_cap * sizeof(u64), ctx->stream));
        ctx->gkey_dirty = false;
    }

    MatchArgs args;
    memset(&args, 0, sizeof(args));
    if (geom != nullptr) args.geom = geom->host;
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
        rc = d_y_true ? launch_match<true, true>(ctx, args, geom->d_geom, batch, nx)
                      : launch_match<true, false>(ctx, args, geom->d_geom, batch, nx);
    } else {
        const int nx = (int)((num_anchors + kBlock - 1) / kBlock);
        rc = d_y_true ? launch_match<false, true>(ctx, args, nullptr, batch, nx)
                      : launch_match<false, false>(ctx, args, nullptr, batch, nx);
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
