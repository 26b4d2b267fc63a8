// Training side of the SSD box pipeline: anchors x ground-truth IoU, matching and offset
// encoding for a whole batch (reference utils/box_utils.py:47-98, called per image from
// utils/datasets.py:152).
//
// Kernel 0 (prep_kernel, geometry mode): everything in the encoding that does not depend on
// the single anchor, once per image instead of once per anchor --
//   * the box centre and log(box size / anchor size) / 0.2 for every (ground-truth box, anchor
//     shape) pair, one 16-byte entry: what an anchor matched to that box gathers;
//   * the x offset against box 0 for every anchor column and the y offset against box 0 for
//     every (feature-map row, anchor shape) pair: anchors that overlap nothing match box 0,
//     and that is most of them.
// The match kernel then needs no logarithm and, for the anchors matched to box 0, no division.
//
// Kernel 1 (match_kernel).  With a geometry a workgroup owns a tile of one feature map, 32
// anchor columns wide, and each of its four waves a band of 2*R rows of it (lanes 0-31 /
// 32-63 take two neighbouring rows, a lane owns R rows of one anchor column, so the x overlap
// with a box is computed once for its rows); anchors are built from the centre tables.
// R = 4 on fine maps; R = 1 on coarse maps, whose big anchors overlap most boxes (the walk
// over the tile's box list then costs a quarter per box, and these tiles get the lowest
// block indices so that the longest ones start first).  The tables are padded by a tile per
// layer, so rows and columns past the edge read (and never store) without clamping, and
// every row of a lane is reached from a per-wave scalar base plus one per-lane offset.
// With a materialised anchor array a workgroup owns 256 consecutive anchors.
// Ground-truth boxes whose bounding box cannot touch the tile are dropped by a ballot
// compaction into LDS, 256 boxes at a time: their IoU with every anchor of the tile is
// exactly +0, which can neither become a first maximum (the running best starts at +0 /
// index 0) nor a best anchor of that box (a zero key leaves the default "anchor 0").  The
// best anchor of each box is a wave-level maximum, folded through LDS and then global memory
// as a packed 64-bit (IoU bits, ~anchor index) atomic max, so ties resolve to the lowest
// anchor index.
//
// Kernel 2 (force_kernel): per image, the forced scatter of utils/box_utils.py:72-87 in
// ascending ground-truth order (the highest box index wins a shared anchor), rewriting the
// <= G affected anchors, and clearing the keys for the next call.
#include <string.h>

#include "ssdbox_internal.h"

namespace {

typedef unsigned long long u64;

// What the match kernel reads of one layer: sixteen dwords, fetched as one scalar load.
struct alignas(64) MatchLayer {
    int fm_h, row_anchors, n_box, anchor_base;
    int tiles_x, tile_base;
    unsigned tilesx_magic, nbox_magic;
    int col_off, rowk_off, cy_off, k_off;
    float half_wmax, half_hmax;
    int coarse, pad_;
};

struct MatchArgs {
    MatchLayer ml[SSDBOX_MAX_LAYERS];
    GeomDev geom;                  // by value: layer constants arrive with the kernel arguments
    const float4 *col_tab;         // geometry: cx, w, h of every anchor column (padded)
    const float *cx_tab;
    const float *cy_tab;
    const float4 *anchors;
    int num_anchors;
    int batch;
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
    float4 *shape_tab;             // (batch, gmax, geom.k_total), see prep_kernel
    float *b0_tab;                 // (batch, geom.col_total + geom.rowk_total), see prep_kernel
    int *order;                    // (batch) images by falling box count, see prep_kernel; or NULL
};

struct MatchShared {
    float4 box[kBlock];            // compacted ground-truth list of the current chunk
    float area[kBlock];
    int gidx[kBlock];
    u64 key[kBlock];               // best (IoU, ~anchor) of each listed box within this tile
    int wcnt[2][kBlock / 64];
    float ext[4][kBlock / 64];
};

// Loads and stores at a wave-uniform base plus a 32-bit per-lane byte offset: one address
// register per lane instead of a 64-bit add per access.
template <typename T>
SSD_DEV T ld_off(const void *base, unsigned byte_off)
{
    return *reinterpret_cast<const T *>(static_cast<const char *>(base) + byte_off);
}
// The targets are written once and not read again by this library: nontemporal stores keep
// them from lingering as dirty lines in the L2s (match+encode of a batch measured 18 % faster
// with several batches in flight, the kernel alone 6 %).
#ifndef SSD_NT_STORES
#define SSD_NT_STORES 1
#endif
typedef float vf2 __attribute__((ext_vector_type(2)));
typedef float vf4 __attribute__((ext_vector_type(4)));
template <typename T>
SSD_DEV void st_off(void *base, unsigned byte_off, T v)
{
    char *q = static_cast<char *>(base) + byte_off;
    if constexpr (!SSD_NT_STORES) {
        *reinterpret_cast<T *>(q) = v;
    } else if constexpr (sizeof(T) == 16) {
        __builtin_nontemporal_store(__builtin_bit_cast(vf4, v), reinterpret_cast<vf4 *>(q));
    } else if constexpr (sizeof(T) == 8) {
        __builtin_nontemporal_store(__builtin_bit_cast(vf2, v), reinterpret_cast<vf2 *>(q));
    } else {
        __builtin_nontemporal_store(__builtin_bit_cast(float, v), reinterpret_cast<float *>(q));
    }
}

// One anchor's result in either output layout: `row` is a wave-uniform first row, `off` the
// lane's row offset from it.  y_true rows are [one_hot(int(conf), 2), locs] (reference
// utils/datasets.py:155-159).  DETAILS: the caller asked for match_idx and/or match_iou.
template <bool YTRUE, bool DETAILS>
SSD_DEV void store_target(const MatchArgs &p, long long row, unsigned off, float conf, float4 t, int idx,
                          float iou)
{
    if (YTRUE) {
        int c = (int)conf;
        float *y = p.y_true + row * 6;
        st_off(y, off * 24u, make_float2(c == 0 ? 1.0f : 0.0f, c == 1 ? 1.0f : 0.0f));
        st_off(y, off * 24u + 8u, make_float2(t.x, t.y));
        st_off(y, off * 24u + 16u, make_float2(t.z, t.w));
    } else {
        st_off(p.conf + row, off * 4u, conf);
        st_off(p.locs + row, off * 16u, t);
        if (DETAILS) {
            if (p.match_idx) st_off(p.match_idx + row, off * 4u, idx);
            if (p.match_iou) st_off(p.match_iou + row, off * 4u, iou);
        }
    }
}

// Layer of entry `s` of a table whose per-layer offsets grow with the layer.
template <typename F>
SSD_DEV int layer_of(const GeomDev &g, int s, F offset_of)
{
    int m = 0;
#pragma unroll
    for (int l = 1; l < SSDBOX_MAX_LAYERS; ++l) m += (l < g.n_layers && s >= offset_of(g.layer[l])) ? 1 : 0;
    return m;
}

// encode (utils/box_utils.py:115-123) split by what each term depends on; the operations on
// each value are those of encode_one, so the bits are too.
__global__ __launch_bounds__(kBlock) void prep_kernel(const MatchArgs p)
{
    const int img = blockIdx.y;
    const GeomDev &g = p.geom;
    // The match kernel's workgroups start in grid order, image by image: images with many boxes
    // go first, so that the long tiles are not the last ones running (a fifth of the kernel
    // time between the best and the worst order of the bench batch).
    if (p.order != nullptr && blockIdx.x == 0) {
        __shared__ int s_rank;
        if (threadIdx.x == 0) s_rank = 0;
        __syncthreads();
        const int mine = p.gt_count[img];
        int before = 0;
        for (int b = threadIdx.x; b < p.batch; b += kBlock) {
            const int c = p.gt_count[b];
            before += (c > mine || (c == mine && b < img)) ? 1 : 0;
        }
        if (before != 0) atomicAdd(&s_rank, before);
        __syncthreads();
        if (threadIdx.x == 0) p.order[s_rank] = img;
    }
    const int n_b0 = g.col_total + g.rowk_total;
    const int u = blockIdx.x * kBlock + threadIdx.x;
    const float4 *gt = p.gt_boxes + (long long)img * p.gmax;
    if (u < n_b0) {
        const float4 b0 = gt[0];
        float v;
        if (u < g.col_total) {
            const float4 c = p.col_tab[u];
            v = ((b0.x + b0.z) / 2.0f - c.x) / (c.y * 0.1f);
        } else {
            const int s = u - g.col_total;
            const int m = layer_of(g, s, [](const LayerDev &L) { return L.rowk_off; });
            const LayerDev &L = g.layer[m];
            const int r = s - L.rowk_off;
            const int row = udiv_magic(r, L.n_box, L.nbox_magic);    // may lie in the padding
            const int k = r - row * L.n_box;
            v = ((b0.y + b0.w) / 2.0f - p.cy_tab[L.cy_off + row]) / (L.bh[k] * 0.1f);
        }
        p.b0_tab[(long long)img * n_b0 + u] = v;
        return;
    }
    const int kt = g.k_total;
    const int G = max(min(p.gt_count[img], p.gmax), 1);
    const int w = u - n_b0;
    if (w >= G * kt) return;
    const int gi = w / kt;
    const int s = w - gi * kt;
    const int m = layer_of(g, s, [](const LayerDev &L) { return L.k_off; });
    const int k = s - g.layer[m].k_off;
    const float4 b = gt[gi];
    p.shape_tab[((long long)img * p.gmax + gi) * kt + s] =
        make_float4((b.x + b.z) / 2.0f, (b.y + b.w) / 2.0f,
                    ssd_logf((b.z - b.x) / g.layer[m].bw[k]) / 0.2f,
                    ssd_logf((b.w - b.y) / g.layer[m].bh[k]) / 0.2f);
}

// R anchors (rows of one anchor column) per thread; a tile is 8*R rows x 32 anchor columns.
template <bool GEOM, bool YTRUE, bool DETAILS, int R>
SSD_DEV void match_tile(const MatchArgs &p, MatchShared &sm, int m)
{
    const unsigned t = threadIdx.x;
    const unsigned lane = t & 63u;
    const int wave = __builtin_amdgcn_readfirstlane((int)(t >> 6));
    const int tile = blockIdx.x;
    const int img = (GEOM && p.order != nullptr) ? p.order[blockIdx.y] : (int)blockIdx.y;
    const float4 *gt = p.gt_boxes + (long long)img * p.gmax;
    u64 *gkey = p.gkey + (long long)img * p.gmax;
    const float *labels = p.gt_labels + (long long)img * p.gmax;
    const float4 *shapes = p.shape_tab + (long long)img * p.gmax * p.geom.k_total;

    // The count and the first 256 boxes are fetched together with everything else the
    // prologue reads, so a tile pays one memory round trip before it can start.
    const int G = max(min(p.gt_count[img], p.gmax), 1);
    const float4 b_first = ld_off<float4>(gt, min(t, (unsigned)p.gmax - 1u) * 16u);

    bool valid[R];
    float acy[R], ay0[R], ay1[R], a_area[R];
    float ty0[R];                  // y offset against box 0
    float acx, aw, ah, ax0, ax1;
    float ex0, ey0, ex1, ey1;      // bounding box of every anchor of this workgroup
    float wy0 = -3.0e38f, wy1 = 3.0e38f;   // y extent of this wave's anchors (geometry mode)
    float tx0 = 0.0f;              // x offset against box 0
    float2 twh0 = make_float2(0.0f, 0.0f);
    unsigned shape = 0;            // the lane's anchor shape among the shapes of all layers
    int a0;                        // index of the lane's first anchor; row i adds i * a_step
    int a_step = 0;
    long long out_row = 0;         // output row of the wave's first anchor; row i adds i * a_step
    unsigned out_off;              // the lane's rows from there

    if (GEOM) {
        const MatchLayer L = p.ml[m];
        const int tt = tile - L.tile_base;
        const int ty = udiv_magic(tt, L.tiles_x, L.tilesx_magic);
        const int tx = tt - ty * L.tiles_x;
        const int t_r0 = ty * (8 * R);                   // first row of the tile
        const int r0 = t_r0 + 2 * R * wave;              // first row of this wave
        const float *b0 = p.b0_tab + (long long)img * (p.geom.col_total + p.geom.rowk_total);
        const unsigned c0 = (unsigned)tx * kTileCols;
        const unsigned c = c0 + (lane & 31u);            // may lie in the padding
        const unsigned k = L.n_box == 1 ? 0u : c - __umulhi(c, L.nbox_magic) * (unsigned)L.n_box;
        const unsigned ci = (unsigned)L.col_off + c;
        const float4 col = ld_off<float4>(p.col_tab, ci * 16u);
        acx = col.x;
        aw = col.y;
        ah = col.z;
        tx0 = ld_off<float>(b0, ci * 4u);
        shape = (unsigned)L.k_off + k;
        twh0 = ld_off<float2>(shapes, shape * 16u + 8u);
        const unsigned rowslot = lane >> 5;
        const bool col_ok = c < (unsigned)L.row_anchors;
        // a wave whose band starts below the map reads the padding of the last rows only
        const int r0c = min(r0, L.fm_h);
        const float *cy = p.cy_tab + L.cy_off + r0c;
        const float *ty_tab = b0 + p.geom.col_total + L.rowk_off + r0c * L.n_box;
        const unsigned ty_off = (rowslot * (unsigned)L.n_box + k) * 4u;
#pragma unroll
        for (int i = 0; i < R; ++i) {
            valid[i] = col_ok && r0 + (int)rowslot + 2 * i < L.fm_h;
            acy[i] = ld_off<float>(cy + 2 * i, rowslot * 4u);
            ty0[i] = ld_off<float>(ty_tab + 2 * i * L.n_box, ty_off);
        }
        out_off = rowslot * (unsigned)L.row_anchors + c;
        a_step = 2 * L.row_anchors;
        a0 = L.anchor_base + r0 * L.row_anchors + (int)out_off;
        out_row = (long long)img * p.num_anchors + L.anchor_base + r0 * L.row_anchors;
        // Corner rounding is monotonic, so the first/last centre with the widest box bound
        // every corner of the tile (and of the wave's band).
        const int t_r1 = min(t_r0 + 8 * R, L.fm_h) - 1;
        const int w_r0 = min(r0, L.fm_h - 1);
        const int w_r1 = min(r0 + 2 * R, L.fm_h) - 1;
        const int c1 = min((int)c0 + kTileCols, L.row_anchors) - 1;
        ex0 = p.col_tab[L.col_off + (int)c0].x - L.half_wmax;
        ex1 = p.col_tab[L.col_off + c1].x + L.half_wmax;
        ey0 = p.cy_tab[L.cy_off + t_r0] - L.half_hmax;
        ey1 = p.cy_tab[L.cy_off + t_r1] + L.half_hmax;
        wy0 = p.cy_tab[L.cy_off + w_r0] - L.half_hmax;
        wy1 = p.cy_tab[L.cy_off + max(w_r1, w_r0)] + L.half_hmax;
    } else {
        const int a_first = tile * kBlock + wave * 64;
        const int last = p.num_anchors - 1;
        valid[0] = a_first + (int)lane < p.num_anchors;
        a0 = min(a_first + (int)lane, last);
        const int base = min(a_first, last);             // a wave past the end stores nothing
        out_off = (unsigned)(a0 - base);
        out_row = (long long)img * p.num_anchors + base;
        const float4 anc = ld_off<float4>(p.anchors + base, out_off * 16u);
        acx = anc.x;
        acy[0] = anc.y;
        aw = anc.z;
        ah = anc.w;
    }

    const float ahw = aw / 2.0f, ahh = ah / 2.0f;
    if (!GEOM) {
        float lo_x = acx - ahw, lo_y = acy[0] - ahh, hi_x = acx + ahw, hi_y = acy[0] + ahh;
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
 
    float best[R];         // reduce_max / argmax over GT, first maximum wins (:66-67)
    int bidx[R];
#pragma unroll
    for (int i = 0; i < R; ++i) {
        best[i] = 0.0f;
        bidx[i] = 0;
    }

    bool touched = false;          // some box reached this wave's IoU evaluation
    int par = 0;
    for (int g0 = 0; g0 < G; g0 += kBlock, par ^= 1) {
        const int g = g0 + (int)t;
        bool keep = false;
        float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
        if (g < G) {
            b = (g0 == 0) ? b_first : ld_off<float4>(gt, (unsigned)g * 16u);
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
        if (n_list == 0) continue;      // nothing near this tile (the usual case on fine maps)
        if (keep) {
            sm.box[off] = b;
            sm.area[off] = clip0(b.z - b.x, p.max_size) * clip0(b.w - b.y, p.max_size);
            sm.gidx[off] = g;
        }
        if ((int)t < n_list) sm.key[t] = 0ull;
        __syncthreads();

        // Corners and areas only where a box came near: most tiles of a fine map never get
        // here.  transform_center_to_corner, utils/box_utils.py:177-179
        ax0 = acx - ahw;
        ax1 = acx + ahw;
        const float a_w = clip0(ax1 - ax0, p.max_size);
#pragma unroll
        for (int i = 0; i < R; ++i) {
            ay0[i] = acy[i] - ahh;
            ay1[i] = acy[i] + ahh;
            a_area[i] = a_w * clip0(ay1[i] - ay0[i], p.max_size);
            // an anchor past the edge overlaps nothing (and is never stored)
            if (!valid[i]) ay1[i] = -3.0e38f;
        }

        float4 gb_next = sm.box[0];
        for (int q = 0; q < n_list; ++q) {
            // the next box is on its way from LDS while this one is evaluated
            const float4 gb = gb_next;
            gb_next = sm.box[min(q + 1, n_list - 1)];
            // compute_iou, utils/box_utils.py:35-42; the x overlap is shared by the rows
            const float iw = vmin(ax1, gb.z) - vmax(ax0, gb.x);
            // no lane of the wave can overlap this box: every IoU is +0, nothing to update
            if (__ballot(iw > 0.0f && !(gb.w <= wy0) && !(gb.y >= wy1)) == 0ull) continue;
            touched = true;
            // The rows are evaluated without branches (an IoU of a non-overlapping pair is
            // replaced by +0 afterwards) so that their divisions overlap.
            const float cw = vmin(iw, p.max_size);
            const float garea = sm.area[q];
            const int gidx = sm.gidx[q];
            float iou[R];
#pragma unroll
            for (int i = 0; i < R; ++i) {
                const float ih = vmin(ay1[i], gb.w) - vmax(ay0[i], gb.y);
                const float inter = cw * vmin(ih, p.max_size);
                const float uni = (a_area[i] + garea) - inter;
                const float q_iou = inter / uni;
                iou[i] = (iw > 0.0f && ih > 0.0f) ? q_iou : 0.0f;
                if (iou[i] > best[i]) {
                    best[i] = iou[i];
                    bidx[i] = gidx;
                }
            }
            float v = iou[0];          // this lane's best IoU with the box
#pragma unroll
            for (int i = 1; i < R; ++i) v = fmaxf(v, iou[i]);
            // Only a lane at or above the tile's best IoU so far for this box can change it.
            const float cur = u2f((uint32_t)(sm.key[q] >> 32));
            if (__ballot(v > 0.0f && v >= cur) != 0ull) {
                // argmax over anchors (:70): wave maximum, then the lowest anchor holding it
                // (within a lane: its lowest row)
                int vi = 0;
#pragma unroll
                for (int i = R - 1; i > 0; --i)
                    if (iou[i] == v) vi = i;
                if (iou[0] == v) vi = 0;
                const float mx = wave_max_nonneg(v);
                const int av = a0 + vi * a_step;
                const int a_src = (int)wave_min_u32((v == mx) ? (uint32_t)av : 0xFFFFFFFFu);
                if (lane == 0) {
                    const u64 key = ((u64)f2u(mx) << 32) | (u64)(0xFFFFFFFFu - (uint32_t)a_src);
                    if (key > sm.key[q]) atomicMax(&sm.key[q], key);
                }
            }
        }
        __syncthreads();
        if ((int)t < n_list) {
            const u64 key = sm.key[t];
            if (key != 0ull) atomicMax(&gkey[sm.gidx[t]], key);
        }
        __syncthreads();
    }

    // No box came near and the threshold is positive: every anchor of the wave matches box 0
    // with IoU +0 and is a negative; its offsets are the table values.
    if (GEOM && !touched && 0.0f < p.thr) {
#pragma unroll
        for (int i = 0; i < R; ++i) {
            if (valid[i])
                store_target<YTRUE, DETAILS>(p, out_row + (long long)i * a_step, out_off, 0.0f,
                                             make_float4(tx0, ty0[i], twh0.x, twh0.y), 0, 0.0f);
        }
        return;
    }

    // gather labels / boxes, threshold, encode every anchor (:89-96).  Loads and stores share
    // one in-order counter per wave, so all the gathers go first and the stores last: a store
    // between two gathers would make the second one wait for the store to reach memory.
    // The decisions below are per wave and row (not per lane), which keeps the loads of all
    // rows in flight together.
    float lab[R];
    float4 gb[R];                  // the matched box (anchor array) / its table entry (geometry)
    bool far[R];                   // some lane of the wave matched row i to a box other than 0
#pragma unroll
    for (int i = 0; i < R; ++i) {
        lab[i] = 0.0f;
        if (__ballot(valid[i] && !(best[i] < p.thr)) != 0ull) lab[i] = ld_off<float>(labels, (unsigned)bidx[i] * 4u);
        far[i] = !GEOM || __ballot(valid[i] && bidx[i] != 0) != 0ull;
        if (far[i]) {
            gb[i] = GEOM ? ld_off<float4>(shapes, ((unsigned)bidx[i] * (unsigned)p.geom.k_total + shape) * 16u)
                         : ld_off<float4>(gt, (unsigned)bidx[i] * 16u);
        }
    }
    float conf[R];
    float4 tgt[R];
#pragma unroll
    for (int i = 0; i < R; ++i) {
        conf[i] = (best[i] < p.thr) ? 0.0f : lab[i];
        if (GEOM) {
            tgt[i] = make_float4(tx0, ty0[i], twh0.x, twh0.y);
            if (far[i]) {
                const float4 e = make_float4((gb[i].x - acx) / (aw * 0.1f), (gb[i].y - acy[i]) / (ah * 0.1f),
                                             gb[i].z, gb[i].w);
                if (bidx[i] != 0) tgt[i] = e;
            }
        } else {
            tgt[i] = encode_one(make_float4(acx, acy[i], aw, ah), gb[i], 0.1f, 0.2f);
        }
    }
    // every value is final (and every gather has landed) before the first store is issued
#pragma unroll
    for (int i = 0; i < R; ++i)
        asm volatile("" : "+v"(conf[i]), "+v"(tgt[i].x), "+v"(tgt[i].y), "+v"(tgt[i].z), "+v"(tgt[i].w));
#pragma unroll
    for (int i = 0; i < R; ++i) {
        if (valid[i])
            store_target<YTRUE, DETAILS>(p, out_row + (long long)i * a_step, out_off, conf[i], tgt[i], bidx[i],
                                         best[i]);
    }
}

template <bool GEOM, bool YTRUE, bool DETAILS>
__global__ __launch_bounds__(kBlock) void match_kernel(const MatchArgs p)
{
    __shared__ MatchShared sm;
    if (GEOM) {
        const int tile = blockIdx.x;
        int m = 0;                         // tile ranges run from the last layer to the first
#pragma unroll
        for (int l = 0; l + 1 < SSDBOX_MAX_LAYERS; ++l)
            m += (l + 1 < p.geom.n_layers && tile < p.ml[l].tile_base) ? 1 : 0;
        if (p.ml[m].coarse)
            match_tile<true, YTRUE, DETAILS, 1>(p, sm, m);
        else
            match_tile<true, YTRUE, DETAILS, kRowsPerThread>(p, sm, m);
    } else {
        match_tile<false, YTRUE, DETAILS, 1>(p, sm, 0);
    }
}

template <bool GEOM, bool YTRUE>
__global__ __launch_bounds__(kBlock) void force_kernel(const MatchArgs p)
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
        const float4 box = gt[g];
        const float label = p.gt_labels[(long long)img * p.gmax + g];
        bool later = false;            // a later box claims the same anchor: it wins
        for (int g2 = g + 1; g2 < G; ++g2) {
            if (s_anc[g2] == a) {
                later = true;
                break;
            }
        }
        if (later) continue;
        const float4 anc = GEOM ? geom_anchor(p.geom, p.cx_tab, p.cy_tab, a) : p.anchors[a];
        const float conf = (1.0f < p.thr) ? 0.0f : label;
        const float4 tgt = encode_one(anc, box, 0.1f, 0.2f);
        store_target<YTRUE, true>(p, (long long)img * p.num_anchors + a, 0u, conf, tgt, g, 1.0f);
    }
}

template <bool GEOM, bool YTRUE>
int launch_match(ssdbox_ctx *ctx, const MatchArgs &args, int batch, int n_tiles)
{
    if (GEOM) {
        const int items = args.geom.col_total + args.geom.rowk_total + args.gmax * args.geom.k_total;
        launch_stage(ctx, SSDBOX_STAGE_SHAPES, prep_kernel, dim3((items + kBlock - 1) / kBlock, batch),
                     dim3(kBlock), args);
    }
    SSD_CHECK_HIP(hipGetLastError());
    const dim3 grid(n_tiles, batch);
    if (args.match_idx != nullptr || args.match_iou != nullptr)
        launch_stage(ctx, SSDBOX_STAGE_MATCH, match_kernel<GEOM, YTRUE, true>, grid, dim3(kBlock), args);
    else
        launch_stage(ctx, SSDBOX_STAGE_MATCH, match_kernel<GEOM, YTRUE, false>, grid, dim3(kBlock), args);
    SSD_CHECK_HIP(hipGetLastError());
    launch_stage(ctx, SSDBOX_STAGE_FORCE, force_kernel<GEOM, YTRUE>, dim3(batch), dim3(kBlock), args);
    SSD_CHECK_HIP(hipGetLastError());
    return SSDBOX_OK;
}

int match_common(ssdbox_ctx *ctx, const ssdbox_geom *geom, const float *d_anchors,
                 int64_t num_anchors, const float *d_gt_boxes, const float *d_gt_labels,
                 const int32_t *d_gt_count, int batch, int gmax, float max_size,
                 float iou_threshold, float *d_conf, float *d_locs, float *d_y_true,
                 int32_t *d_match_idx, float *d_match_iou, bool static_scratch = false)
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
    SSD_REQUIRE(num_anchors >= 1 && num_anchors < (1ll << 28), "match_encode: num_anchors %lld",
                (long long)num_anchors);
    const long long n_tiles = geom ? geom->host.n_tiles : (num_anchors + kBlock - 1) / kBlock;

    DeviceGuard guard(ctx->device);
    const size_t need = (size_t)batch * (size_t)gmax;
    u64 *s_gkey = nullptr;
    float4 *s_shape = nullptr;
    float *s_b0 = nullptr;
    int *s_order = nullptr;
    if (static_scratch) {
        // Carve the call's scratch out of the device's fixed array: nothing is allocated.  The
        // array keeps the previous call's contents, so the keys are cleared on every call.
        SSD_REQUIRE(ctx->static_scratch != nullptr, "match_encode_static: the context has no fixed scratch");
        const size_t kt = geom ? (size_t)geom->host.k_total : 0;
        const size_t nb0 = geom ? (size_t)batch * (size_t)(geom->host.col_total + geom->host.rowk_total) : 0;
        const size_t off_shape = need * sizeof(u64);
        const size_t off_b0 = off_shape + need * kt * sizeof(float4);
        const size_t off_order = (off_b0 + nb0 * sizeof(float) + 15) / 16 * 16;
        const size_t total = off_order + (size_t)batch * sizeof(int);
        if (total > ctx->static_scratch_bytes) {
            set_error("match_encode_static: batch %d x gmax %d needs %zu bytes of scratch, the fixed array "
                      "holds %zu (use ssdbox_match_encode)", batch, gmax, total, ctx->static_scratch_bytes);
            return SSDBOX_ERR_INVALID;
        }
        unsigned char *base = ctx->static_scratch;
        s_gkey = reinterpret_cast<u64 *>(base);
        s_shape = reinterpret_cast<float4 *>(base + off_shape);
        s_b0 = reinterpret_cast<float *>(base + off_b0);
        s_order = reinterpret_cast<int *>(base + off_order);
        SSD_CHECK_HIP(hipMemsetAsync(s_gkey, 0, need * sizeof(u64), ctx->stream));
    } else {
        int rc = ensure_buffer((void **)&ctx->gkey, &ctx->gkey_cap, need, sizeof(u64), true, ctx->stream);
        if (rc != SSDBOX_OK) return rc;
        if (ctx->gkey_dirty) {
            SSD_CHECK_HIP(hipMemsetAsync(ctx->gkey, 0, ctx->gkey_cap * sizeof(u64), ctx->stream));
            ctx->gkey_dirty = false;
        }
        if (geom != nullptr) {
            rc = ensure_buffer((void **)&ctx->shape_tab, &ctx->shape_cap, need * (size_t)geom->host.k_total,
                               sizeof(float4), false, ctx->stream);
            if (rc != SSDBOX_OK) return rc;
            rc = ensure_buffer((void **)&ctx->b0_tab, &ctx->b0_cap,
                               (size_t)batch * (size_t)(geom->host.col_total + geom->host.rowk_total),
                               sizeof(float), false, ctx->stream);
            if (rc != SSDBOX_OK) return rc;
            rc = ensure_buffer((void **)&ctx->img_order, &ctx->order_cap, (size_t)batch, sizeof(int), false,
                               ctx->stream);
            if (rc != SSDBOX_OK) return rc;
        }
        s_gkey = ctx->gkey;
        s_shape = ctx->shape_tab;
        s_b0 = ctx->b0_tab;
        s_order = ctx->img_order;
    }

    MatchArgs args;
    memset(&args, 0, sizeof(args));
    if (geom != nullptr) {
        args.geom = geom->host;
        for (int m = 0; m < geom->host.n_layers; ++m) {
            const LayerDev &L = geom->host.layer[m];
            MatchLayer &M = args.ml[m];
            M.fm_h = L.fm_h;
            M.row_anchors = L.row_anchors;
            M.n_box = L.n_box;
            M.anchor_base = L.anchor_base;
            M.tiles_x = L.tiles_x;
            M.tile_base = L.tile_base;
            M.tilesx_magic = L.tilesx_magic;
            M.nbox_magic = L.nbox_magic;
            M.col_off = L.col_off;
            M.rowk_off = L.rowk_off;
            M.cy_off = L.cy_off;
            M.k_off = L.k_off;
            M.half_wmax = L.wmax / 2.0f;
            M.half_hmax = L.hmax / 2.0f;
            M.coarse = L.coarse;
        }
    }
    args.col_tab = geom ? geom->d_col : nullptr;
    args.cx_tab = geom ? geom->d_cx : nullptr;
    args.cy_tab = geom ? geom->d_cy : nullptr;
    args.anchors = reinterpret_cast<const float4 *>(d_anchors);
    args.num_anchors = (int)num_anchors;
    args.batch = batch;
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
    args.gkey = s_gkey;
    args.shape_tab = s_shape;
    args.b0_tab = s_b0;
    // ranking the images is quadratic in the batch: only where it is cheap
    args.order = (geom != nullptr && batch > 1 && batch <= 4096) ? s_order : nullptr;

    // Event timing takes events from a pool that may have to grow: not on the path that
    // allocates nothing.
    const bool timing = ctx->timing;
    if (static_scratch) ctx->timing = false;
    else ctx->gkey_dirty = true;    // cleared again only when every launch was enqueued
    int rc;
    if (geom != nullptr)
        rc = d_y_true ? launch_match<true, true>(ctx, args, batch, (int)n_tiles)
                      : launch_match<true, false>(ctx, args, batch, (int)n_tiles);
    else
        rc = d_y_true ? launch_match<false, true>(ctx, args, batch, (int)n_tiles)
                      : launch_match<false, false>(ctx, args, batch, (int)n_tiles);
    ctx->timing = timing;
    if (rc == SSDBOX_OK && !static_scratch) ctx->gkey_dirty = false;
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

// ssdbox_match_encode for callers that cannot allocate (HIP graph capture, serving loops): the
// same kernels, their scratch carved out of the device's fixed array instead of buffers the
// context grows.  No hipMalloc / hipFree and no host allocation on this path.
extern "C" int ssdbox_match_encode_static(ssdbox_ctx *ctx, const ssdbox_geom *geom, const float *d_anchors,
                                          int64_t num_anchors, const float *d_gt_boxes,
                                          const float *d_gt_labels, const int32_t *d_gt_count, int batch,
                                          int gmax, float max_size, float iou_threshold, float *d_conf,
                                          float *d_locs, int32_t *d_match_idx, float *d_match_iou)
{
    return match_common(ctx, geom, d_anchors, num_anchors, d_gt_boxes, d_gt_labels, d_gt_count,
                        batch, gmax, max_size, iou_threshold, d_conf, d_locs, nullptr, d_match_idx,
                        d_match_iou, true);
}
