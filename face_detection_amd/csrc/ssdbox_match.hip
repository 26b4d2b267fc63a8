// Training side of the SSD box pipeline: anchors x ground-truth IoU, matching and offset
// encoding for a whole batch (reference utils/box_utils.py:47-98, called per image from
// utils/datasets.py:152).
//
// Kernel 1 (match_geom_kernel, anchors from a geometry).  A workgroup owns a tile of one feature
// map, 32 anchor columns wide and several groups of 2*R rows tall; its four waves take the
// groups in turn (lanes 0-31 / 32-63 hold two neighbouring rows, a lane owns R rows of one anchor
// column, so the x overlap with a box is computed once for its rows).  R = 4 on fine maps, R = 1
// on coarse maps, whose big anchors overlap most boxes.
//   Prologue, one round trip to memory: the image's boxes and labels, the lane's anchor column,
// the tile's row centres.  Ground-truth boxes whose bounding box cannot touch the tile are
// dropped by a ballot compaction into LDS: their IoU with every anchor of the tile is exactly
// +0, which can neither become a first maximum (the running best starts at +0 / box 0) nor a
// best anchor of that box (a zero key leaves the default "anchor 0").  Then everything of
// encode (utils/box_utils.py:115-123) that does not depend on the single anchor goes into LDS
// tables: log(box size / shape size) / 0.2 for every (listed box, anchor shape) pair and for
// box 0, and the y offset against box 0 for every (row, shape) pair; the x offset against box 0
// is one division per lane.  Anchors that overlap nothing match box 0, and that is most of them:
// they need no division and no logarithm.
//   Main loop: a wave's groups read LDS only and issue the stores of their rows, so a wave never
// waits for memory between its prologue and its exit and one prologue is spread over many rows.
// The best anchor of each box is a wave-level maximum, folded through LDS and then global memory
// as a packed 64-bit (IoU bits, ~anchor index) atomic max, so ties resolve to the lowest anchor.
//   More than 256 boxes per image: the list is built 256 boxes at a time, every wave has one
// group, its running best lives in registers across the chunks and the final encode gathers the
// matched box from memory.
//
// Kernel 1' (match_array_kernel, materialised anchor array): a workgroup owns 256 consecutive
// anchors, same culling, encode from the gathered box.
//
// Kernel 2 (force_kernel): per image, the forced scatter of utils/box_utils.py:72-87 in
// ascending ground-truth order (the highest box index wins a shared anchor), rewriting the
// <= G affected anchors, and clearing the keys for the next call.
#include <stdlib.h>
#include <string.h>

#include "ssdbox_internal.h"

namespace {

typedef unsigned long long u64;

constexpr int kListCap = kBlock;       // ground-truth boxes examined (and at most listed) per chunk

// What the geometry kernel reads of one layer: sixteen dwords, fetched as one scalar load.
struct alignas(64) MatchLayer {
    int fm_h, row_anchors, n_box, anchor_base;
    int tiles_x, tile_base;
    unsigned tilesx_magic, nbox_magic;
    int col_off, cy_off;
    int n_groups;                  // groups of 2*R rows per tile
    int coarse;                    // R = 1 instead of 4
    float half_wmax, half_hmax;
    int pad_[2];
};

struct MatchArgs {
    MatchLayer ml[SSDBOX_MAX_LAYERS];
    int n_layers;
    int nbox_max;                  // largest n_box of the geometry: row length of the LDS tables
    int rows_max;                  // largest tile height in rows
    int stage_off;                 // byte offset of the y_true staging areas in the dynamic LDS
    const GeomDev *geom;           // device copy of the geometry (force kernel)
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
#ifdef SSD_EXP_TRACE
    unsigned long long *tr, *tr_force;
#endif
#ifdef SSD_EXP_WGTRACE             // timing experiment (profiles/match_waves.py): stamps per wave
    unsigned long long *wg;
#endif
};

struct MatchShared {
    float4 box[kListCap];          // compacted ground-truth list of the current chunk
    float area[kListCap];
    float label[kListCap];
    int gidx[kListCap];
    u64 key[kListCap];             // best (IoU, ~anchor) of each listed box within this tile
    int wcnt[2][kBlock / 64];
    float ext[4][kBlock / 64];
    float4 lane_const[kBlock];     // per thread: what only the final encode of a row needs
    float2 shape_wh[SSDBOX_MAX_BOXES_PER_CELL];   // the layer's box shapes
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
typedef float vf2 __attribute__((ext_vector_type(2)));
typedef float vf4 __attribute__((ext_vector_type(4)));
template <typename T>
SSD_DEV void st_off(void *base, unsigned byte_off, T v)
{
    char *q = static_cast<char *>(base) + byte_off;
    if constexpr (sizeof(T) == 16) {
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

// A run of consecutive y_true rows, one per lane of a group of LANES lanes (32: a half wave, 64:
// the wave), written as whole 16-byte pieces.  The rows are 24 bytes long, so a lane's own row
// would go out as three 8-byte pieces at a 24-byte stride and every 128-byte line would reach
// memory in several partial writes (twice the bytes on the memory side, a third of the store
// rate).  Instead the group lays its rows out in LDS as they lie in memory (shifted by the run's
// offset inside its first 16 bytes) and every lane stores aligned 16-byte pieces of the run; a
// run that starts or ends on an odd multiple of 8 bytes has one 8-byte piece at that end.
//   region   this group's staging area, 16-byte aligned, LANES * 24 + 16 bytes
//   gdst     address of the run's first row (the same in every lane of the group, 8-byte aligned)
//   n_rows   rows of the run (the same in every lane of the group; 0: nothing to store)
//   l        lane within the group; `have`: this lane holds row l of the run
// LDS instructions of one wave execute in order, so the group needs no barrier between its
// writes and its reads, nor before the next run overwrites the area.
constexpr unsigned kStageBytes32 = 32u * 24u + 16u;
constexpr unsigned kStageBytes64 = 64u * 24u + 16u;
template <int LANES>
SSD_DEV void store_ytrue_run(unsigned char *region, float *gdst, int n_rows, unsigned l, bool have, float conf,
                             float4 t)
{
    const unsigned shift = (unsigned)reinterpret_cast<uintptr_t>(gdst) & 8u;
    if (have) {
        const int c = (int)conf;
        float2 *w = reinterpret_cast<float2 *>(region + shift + 24u * l);
        w[0] = make_float2(c == 0 ? 1.0f : 0.0f, c == 1 ? 1.0f : 0.0f);
        w[1] = make_float2(t.x, t.y);
        w[2] = make_float2(t.z, t.w);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const unsigned end = shift + 24u * (unsigned)n_rows;         // the run is [shift, end) of the area
    unsigned char *g0 = reinterpret_cast<unsigned char *>(gdst) - shift;   // 16-byte aligned
    constexpr int kRounds = (LANES * 24 + 8 + 16 * LANES - 1) / (16 * LANES);
#pragma unroll
    for (int r = 0; r < kRounds; ++r) {
        const unsigned b = 16u * (l + (unsigned)(r * LANES));
        if (b >= shift && b + 16u <= end)
            __builtin_nontemporal_store(*reinterpret_cast<const vf4 *>(region + b), reinterpret_cast<vf4 *>(g0 + b));
    }
    if (n_rows > 0) {
        if (l == 0u && shift != 0u)
            __builtin_nontemporal_store(*reinterpret_cast<const vf2 *>(region + 8u), reinterpret_cast<vf2 *>(g0 + 8u));
        if (l == 1u && (end & 8u) != 0u)
            __builtin_nontemporal_store(*reinterpret_cast<const vf2 *>(region + end - 8u),
                                        reinterpret_cast<vf2 *>(g0 + end - 8u));
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// Ballot compaction of the chunk's boxes that may touch the extent [ex0, ex1] x [ey0, ey1] into
// the LDS list (dropped only when provably disjoint; NaN is kept).  Returns the list length
// (uniform).  Two barriers; `par` alternates the counter buffer between calls.
SSD_DEV int build_list(MatchShared &sm, int par, unsigned t, unsigned lane, int wave, int g, bool in_range,
                       float4 b, float lab, float ex0, float ey0, float ex1, float ey1, float max_size)
{
    const bool keep = in_range && !(b.z <= ex0) && !(b.x >= ex1) && !(b.w <= ey0) && !(b.y >= ey1);
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
    if (n_list == 0) return 0;          // nothing near this tile (the usual case on fine maps)
    if (keep) {
        sm.box[off] = b;
        sm.area[off] = clip0(b.z - b.x, max_size) * clip0(b.w - b.y, max_size);
        sm.label[off] = lab;
        sm.gidx[off] = g;
    }
    if ((int)t < n_list) sm.key[t] = 0ull;
    __syncthreads();
    return n_list;
}

// The walk of R rows of one anchor column over the listed boxes: compute_iou
// (utils/box_utils.py:35-42), running best per row (first maximum wins, :66-67) and the best
// anchor of every box within the tile (:70).  `sel` receives the list position (LISTPOS) or the
// ground-truth index of the best box.  Boxes outside the rows' y extent [wy0, wy1] are dropped 64
// at a time (a lane per box); of the others the next ones' box, area and key are on their way
// from LDS while the current ones are evaluated.  With one row per lane two boxes are evaluated
// per step, so that their divisions overlap as those of the rows do otherwise.  Returns whether
// any box reached the IoU evaluation.
template <int R, bool LISTPOS>
SSD_DEV bool walk_list(MatchShared &sm, int n_list, unsigned lane, float max_size, float ax0, float ax1,
                       const float (&ay0)[R], const float (&ay1)[R], const float (&a_area)[R], float wy0,
                       float wy1, int a0, int a_step, float (&best)[R], int (&sel)[R])
{
    constexpr int U = (R == 1) ? 2 : 1;        // boxes per step
    bool touched = false;
    for (int base = 0; base < n_list; base += 64) {
        const float4 mine = sm.box[min(base + (int)lane, n_list - 1)];
        u64 todo = __ballot(base + (int)lane < n_list && !(mine.w <= wy0) && !(mine.y >= wy1));
        if (todo == 0ull) continue;
        int qn[U];
        float4 gb_next[U];
        float area_next[U];
        u64 key_next[U];
        auto fetch = [&]() {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                // a step short of boxes repeats its last one: evaluated twice, same result
                if (todo != 0ull) {
                    qn[u] = base + __builtin_ctzll(todo);
                    todo &= todo - 1ull;
                } else {
                    qn[u] = qn[u > 0 ? u - 1 : 0];
                }
                gb_next[u] = sm.box[qn[u]];
                area_next[u] = sm.area[qn[u]];
                key_next[u] = sm.key[qn[u]];
            }
        };
        fetch();
        bool more = true;
        while (more) {
            int qc[U];
            float4 gb[U];
            float garea[U], cur[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                qc[u] = qn[u];
                gb[u] = gb_next[u];
                garea[u] = area_next[u];
                cur[u] = u2f((uint32_t)(key_next[u] >> 32));   // may be stale: it only filters
            }
            more = todo != 0ull;
            if (more) fetch();
            // the x overlap is shared by the rows
            float iw[U];
            bool any = false;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                iw[u] = vmin(ax1, gb[u].z) - vmax(ax0, gb[u].x);
                any = any || iw[u] > 0.0f;
            }
            // no lane of the wave can overlap these boxes: every IoU is +0, nothing to update
            if (__ballot(any) == 0ull) continue;
            touched = true;
            // Evaluated without branches so that the divisions overlap.  A lane without x overlap
            // gets width 0; a row without y overlap gets a height <= 0 (never NaN: the anchor's
            // corners are finite): the quotient is then <= 0 or NaN, where the reference has an
            // IoU of +0, and neither can exceed a running best (>= 0) or count as a best anchor.
            float iou[U][R];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float cw = iw[u] > 0.0f ? vmin(iw[u], max_size) : 0.0f;
#pragma unroll
                for (int i = 0; i < R; ++i) {
                    const float ih = vmin(ay1[i], gb[u].w) - vmax(ay0[i], gb[u].y);
                    const float inter = cw * vmin(ih, max_size);
                    const float uni = (a_area[i] + garea[u]) - inter;
                    iou[u][i] = inter / uni;
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (u > 0 && qc[u] == qc[u - 1]) continue;     // the repeated box of a short step
                const int gsel = LISTPOS ? qc[u] : sm.gidx[qc[u]];
                float v = 0.0f;        // this lane's best IoU with the box (a NaN operand is skipped)
#pragma unroll
                for (int i = 0; i < R; ++i) {
                    if (iou[u][i] > best[i]) {
                        best[i] = iou[u][i];
                        sel[i] = gsel;
                    }
                    v = vmax(v, iou[u][i]);
                }
                // argmax over anchors (:70): only a lane at or above the tile's best IoU so far
                // for this box can change it; wave maximum, then the lowest anchor holding it
                if (__ballot(v > 0.0f && v >= cur[u]) != 0ull) {
                    int vi = 0;        // the lane's lowest row holding v
#pragma unroll
                    for (int i = R - 1; i > 0; --i)
                        if (iou[u][i] == v) vi = i;
                    if (iou[u][0] == v) vi = 0;
                    const float mx = wave_max_nonneg(v);
                    const int av = a0 + vi * a_step;
                    const int a_src = (int)wave_min_u32((v == mx) ? (uint32_t)av : 0xFFFFFFFFu);
                    if (lane == 0) {
                        const u64 key = ((u64)f2u(mx) << 32) | (u64)(0xFFFFFFFFu - (uint32_t)a_src);
                        if (key > sm.key[qc[u]]) atomicMax(&sm.key[qc[u]], key);
                    }
                }
            }
        }
    }
    return touched;
}

// The tile's best anchor of every listed box joins the image-wide best in global memory.
SSD_DEV void flush_keys(MatchShared &sm, int n_list, unsigned t, u64 *gkey)
{
    __syncthreads();
    if ((int)t < n_list) {
        const u64 key = sm.key[t];
        if (key != 0ull) atomicMax(&gkey[sm.gidx[t]], key);
    }
}

extern __shared__ __align__(16) unsigned char match_dyn_lds[];

// MULTI: the image may hold more than kListCap boxes (one group per wave, see the file comment).
template <bool YTRUE, bool DETAILS, bool MULTI, int R>
SSD_DEV void geom_tile(const MatchArgs &p, MatchShared &sm, const MatchLayer &L, const int img, const int tile)
{
    const unsigned t = threadIdx.x;
    const unsigned lane = t & 63u;
    const int wave = __builtin_amdgcn_readfirstlane((int)(t >> 6));
    const float4 *gt = p.gt_boxes + (long long)img * p.gmax;
    const float *labels = p.gt_labels + (long long)img * p.gmax;
    u64 *gkey = p.gkey + (long long)img * p.gmax;
    const unsigned n_box = (unsigned)L.n_box;
#ifdef SSD_EXP_WGTRACE
    unsigned long long *wg = p.wg ? p.wg + (((long long)img * gridDim.x + tile) * 4 + wave) * 4 : nullptr;
    const unsigned long long wg_t0 = wall_clock64();
    unsigned long long wg_t1 = 0;
    int wg_list = 0;
#endif

    // LDS tables: log-size offsets per (listed box | box 0 in slot kListCap, shape), row centres,
    // y offsets against box 0 per (row, shape)
    float2 *s_lwh = reinterpret_cast<float2 *>(match_dyn_lds);
    float *s_cy = reinterpret_cast<float *>(s_lwh + (kListCap + 1) * p.nbox_max);
    float *s_ty0 = s_cy + p.rows_max;
    // y_true layout: every half wave's staging area for its run of rows (store_ytrue_run)
    unsigned char *stage = match_dyn_lds + p.stage_off + (t >> 5) * kStageBytes32;

    // ---- prologue: everything this workgroup reads from memory, requested together
    const int G = max(min(p.gt_count[img], p.gmax), 1);
    const unsigned g_first = min(t, (unsigned)p.gmax - 1u);
    const float4 b_first = ld_off<float4>(gt, g_first * 16u);
    const float l_first = ld_off<float>(labels, g_first * 4u);
    const float4 b0 = gt[0];
    const float label0 = labels[0];

    const int tt = tile - L.tile_base;
    const int ty = udiv_magic(tt, L.tiles_x, L.tilesx_magic);
    const int tx = tt - ty * L.tiles_x;
    const int tile_rows = L.n_groups * 2 * R;
    const int t_r0 = ty * tile_rows;                          // first row of the tile
    const int rows_here = min(tile_rows, L.fm_h - t_r0);      // >= 1
    const unsigned c0 = (unsigned)tx * kTileCols;
    const unsigned c = c0 + (lane & 31u);                     // may lie in the padding
    const unsigned k = n_box == 1u ? 0u : c - __umulhi(c, L.nbox_magic) * n_box;
    const float4 *cols = p.col_tab + L.col_off;               // padded by a tile per layer
    const float4 col = ld_off<float4>(cols, c * 16u);
    // the layer's box shapes are the (w, h) of its first n_box anchor columns: lane k of every
    // wave holds shape k
    const float4 shape_l = ld_off<float4>(cols, min(lane, n_box - 1u) * 16u);
    float cy_mine = 0.0f;
    if ((int)t < tile_rows) cy_mine = p.cy_tab[L.cy_off + min(t_r0 + (int)t, L.fm_h - 1)];
    const float acx = col.x, aw = col.y, ah = col.z;
    const bool col_ok = c < (unsigned)L.row_anchors;
    const unsigned rowslot = lane >> 5;

    // Corner rounding is monotonic, so the first/last centre with the widest box bound every
    // corner of the tile.
    const float ex0 = u2f((uint32_t)__builtin_amdgcn_readlane((int)f2u(acx), 0)) - L.half_wmax;
    const float ex1 = u2f((uint32_t)__builtin_amdgcn_readlane((int)f2u(acx), 31)) + L.half_wmax;
    const float ey0 = p.cy_tab[L.cy_off + t_r0] - L.half_hmax;
    const float ey1 = p.cy_tab[L.cy_off + t_r0 + rows_here - 1] + L.half_hmax;

    const float ahw = aw / 2.0f, ahh = ah / 2.0f;
    const float ax0 = acx - ahw, ax1 = acx + ahw;            // transform_center_to_corner, :177-179
    const float a_w = clip0(ax1 - ax0, p.max_size);
    if (t < n_box) sm.shape_wh[t] = make_float2(shape_l.y, shape_l.z);
    if (!MULTI) {
        // What only the final encode of a row needs waits in LDS (read back by the same thread),
        // not in registers that would stay occupied during the walk over the boxes: the
        // anchor's centre x, encode's divisors (:117-118) and the x offset against box 0.
        const float awv = aw * 0.1f, ahv = ah * 0.1f;
        sm.lane_const[t] = make_float4(acx, awv, ahv, ((b0.x + b0.z) / 2.0f - acx) / awv);
    }

    const int a_step = 2 * L.row_anchors;
    const unsigned out_off = rowslot * (unsigned)L.row_anchors + c;
    const int n_groups_here = (rows_here + 2 * R - 1) / (2 * R);
    const int n_cols = min((int)kTileCols, L.row_anchors - (int)c0);   // anchor columns of this tile
    // One anchor's result (called by every lane of the wave: the y_true layout stores by runs)
    auto emit = [&](long long row, bool ok, bool row_ok, float conf, float4 tgt, int idx, float iou) {
        if (YTRUE) {
            float *gdst = p.y_true + (row + (long long)(out_off - (lane & 31u))) * 6;
            store_ytrue_run<32>(stage, gdst, row_ok ? n_cols : 0, lane & 31u, ok, conf, tgt);
        } else if (ok) {
            store_target<false, DETAILS>(p, row, out_off, conf, tgt, idx, iou);
        }
    };

    float best[R];                 // reduce_max / argmax over GT, first maximum wins (:66-67)
    int sel[R];
    bool touched = false;

    // Tables that do not depend on the box list, written before its barriers: row centres, the
    // y offsets against box 0 per (row, shape), box 0's log-size offsets per shape.  A tile that
    // lists no box goes from the list's first barrier straight to its stores.
    if ((int)t < tile_rows) {
        s_cy[t] = cy_mine;
        if (!MULTI) {
            const float dy = (b0.y + b0.w) / 2.0f - cy_mine;
#pragma unroll 1
            for (unsigned kk = 0; kk < n_box; ++kk) {
                const float sh = u2f((uint32_t)__builtin_amdgcn_readlane((int)f2u(shape_l.z), (int)kk));
                s_ty0[t * n_box + kk] = dy / (sh * 0.1f);
            }
        }
    }
    if (!MULTI && t < n_box)
        s_lwh[(unsigned)kListCap * n_box + t] = make_float2(ssd_logf((b0.z - b0.x) / shape_l.y) / 0.2f,
                                                            ssd_logf((b0.w - b0.y) / shape_l.z) / 0.2f);

    int par = 0;
    for (int g0 = 0; g0 < G; g0 += kListCap, par ^= 1) {
        const int g = g0 + (int)t;
        float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
        float lab = 0.0f;
        if (g < G) {
            b = (g0 == 0) ? b_first : ld_off<float4>(gt, (unsigned)g * 16u);
            lab = (g0 == 0) ? l_first : ld_off<float>(labels, (unsigned)g * 4u);
        }
        const int n_list = build_list(sm, par, t, lane, wave, g, g < G, b, lab, ex0, ey0, ex1, ey1, p.max_size);

        if (!MULTI && n_list != 0) {
            // log-size offsets of the listed boxes, one (box, shape) item per thread
            const unsigned n_items = (unsigned)n_list * n_box;
#pragma unroll 1
            for (unsigned item = t; item < n_items; item += kBlock) {
                const unsigned q = n_box == 1u ? item : __umulhi(item, L.nbox_magic);
                const unsigned kk = item - q * n_box;
                const float2 sh = sm.shape_wh[kk];
                const float4 bx = sm.box[q];
                s_lwh[q * n_box + kk] =
                    make_float2(ssd_logf((bx.z - bx.x) / sh.x) / 0.2f, ssd_logf((bx.w - bx.y) / sh.y) / 0.2f);
            }
            __syncthreads();
        }

#ifdef SSD_EXP_WGTRACE
        if (g0 == 0) {
            wg_t1 = wall_clock64();
            wg_list = n_list;
        }
#endif
#pragma unroll 1
        for (int gi = wave; gi < n_groups_here; gi += kBlock / 64) {
            const int lr0 = gi * 2 * R;                       // the group's first row within the tile
            if (!MULTI || g0 == 0) {
#pragma unroll
                for (int i = 0; i < R; ++i) {
                    best[i] = 0.0f;
                    sel[i] = 0;
                }
                if (!MULTI) touched = false;
            }
            bool valid[R];
#pragma unroll
            for (int i = 0; i < R; ++i) valid[i] = col_ok && lr0 + (int)rowslot + 2 * i < rows_here;
            const long long out_row =
                (long long)img * p.num_anchors + L.anchor_base + (long long)(t_r0 + lr0) * L.row_anchors;
            if (n_list != 0) {
                // Corners and areas only where a box came near.
                float ay0[R], ay1[R], a_area[R];
#pragma unroll
                for (int i = 0; i < R; ++i) {
                    const float acy = s_cy[min(lr0 + (int)rowslot + 2 * i, tile_rows - 1)];
                    ay0[i] = acy - ahh;
                    ay1[i] = acy + ahh;
                    a_area[i] = a_w * clip0(ay1[i] - ay0[i], p.max_size);
                    // an anchor past the edge overlaps nothing (and is never stored)
                    if (!valid[i]) ay1[i] = -3.0e38f;
                }
                // y extent of the group's anchors
                const float wy0 = s_cy[lr0] - L.half_hmax;
                const float wy1 = s_cy[min(lr0 + 2 * R, rows_here) - 1] + L.half_hmax;
                const int a0 = L.anchor_base + (t_r0 + lr0) * L.row_anchors + (int)out_off;
                const bool hit = walk_list<R, !MULTI>(sm, n_list, lane, p.max_size, ax0, ax1, ay0, ay1, a_area,
                                                      wy0, wy1, a0, a_step, best, sel);
                touched = touched || hit;
            }
            if (MULTI && g0 + kListCap < G) continue;        // more boxes to come

            if (MULTI) {
                // gather labels / boxes, threshold, encode every anchor (:89-96)
                float labv[R];
                float4 gb[R];
#pragma unroll
                for (int i = 0; i < R; ++i) {
                    labv[i] = ld_off<float>(labels, (unsigned)sel[i] * 4u);
                    gb[i] = ld_off<float4>(gt, (unsigned)sel[i] * 16u);
                }
#pragma unroll
                for (int i = 0; i < R; ++i) {
                    const float acy = s_cy[min(lr0 + (int)rowslot + 2 * i, tile_rows - 1)];
                    const float conf = (best[i] < p.thr) ? 0.0f : labv[i];
                    const float4 tgt = encode_one(make_float4(acx, acy, aw, ah), gb[i], 0.1f, 0.2f);
                    emit(out_row + (long long)i * a_step, valid[i], lr0 + (int)rowslot + 2 * i < rows_here, conf, tgt,
                         sel[i], best[i]);
                }
                continue;
            }

            // Offsets against box 0: what every anchor that overlaps nothing gets.  Row by row,
            // stored at once: nothing below waits for memory.
            const float2 twh0 = s_lwh[(unsigned)kListCap * n_box + k];
            unsigned t_opaque = t;         // the compiler must not forward the stored registers
            asm volatile("" : "+v"(t_opaque));
            const float4 lc = sm.lane_const[t_opaque];                    // acx, awv, ahv, tx0
#pragma unroll
            for (int i = 0; i < R; ++i) {
                const int lr = min(lr0 + (int)rowslot + 2 * i, tile_rows - 1);
                float conf = (0.0f < p.thr) ? 0.0f : label0;
                float4 tgt = make_float4(lc.w, s_ty0[(unsigned)lr * n_box + k], twh0.x, twh0.y);
                int idx = 0;
                if (touched && __ballot(valid[i] && best[i] > 0.0f) != 0ull) {
                    // gather labels / boxes, threshold, encode (:89-96), from the LDS list
                    const unsigned q = (unsigned)sel[i];
                    const float4 bx = sm.box[q];
                    const float2 lwh = s_lwh[q * n_box + k];
                    const float lq = sm.label[q];
                    const int gq = sm.gidx[q];
                    const float ex = ((bx.x + bx.z) / 2.0f - lc.x) / lc.y;
                    const float ey = ((bx.y + bx.w) / 2.0f - s_cy[lr]) / lc.z;
                    if (best[i] > 0.0f) {
                        tgt = make_float4(ex, ey, lwh.x, lwh.y);
                        conf = (best[i] < p.thr) ? 0.0f : lq;
                        idx = gq;
                    }
                }
                emit(out_row + (long long)i * a_step, valid[i], lr0 + (int)rowslot + 2 * i < rows_here, conf, tgt, idx,
                     best[i]);
            }
        }
        if (n_list != 0) flush_keys(sm, n_list, t, gkey);
        if (g0 + kListCap < G) __syncthreads();
    }
#ifdef SSD_EXP_WGTRACE
    if (wg != nullptr && lane == 0) {
        __builtin_amdgcn_s_waitcnt(0);
        wg[0] = wg_t0;
        wg[1] = wg_t1;
        wg[2] = wall_clock64();
        wg[3] = ((unsigned long long)wg_list << 16) | ((unsigned long long)R << 8) | (unsigned long long)n_box;
    }
#endif
}

template <bool YTRUE, bool DETAILS, bool MULTI>
__global__ __launch_bounds__(kBlock, 5) void match_geom_kernel(const MatchArgs p)
{
    __shared__ MatchShared sm;
    SSD_TRACE_BEGIN(p.tr);
    const int tile = blockIdx.x;
    const int img = blockIdx.y;
    int m = 0;                             // tile_base grows with the layer
#pragma unroll
    for (int l = 1; l < SSDBOX_MAX_LAYERS; ++l) m += (l < p.n_layers && tile >= p.ml[l].tile_base) ? 1 : 0;
    const MatchLayer L = p.ml[m];
    if (L.coarse)
        geom_tile<YTRUE, DETAILS, MULTI, 1>(p, sm, L, img, tile);
    else
        geom_tile<YTRUE, DETAILS, MULTI, kRowsPerThread>(p, sm, L, img, tile);
    SSD_TRACE_END(p.tr);
}

// Materialised anchor array: a workgroup owns 256 consecutive anchors, one per thread.
template <bool YTRUE, bool DETAILS>
__global__ __launch_bounds__(kBlock) void match_array_kernel(const MatchArgs p)
{
    __shared__ MatchShared sm;
    const unsigned t = threadIdx.x;
    const unsigned lane = t & 63u;
    const int wave = __builtin_amdgcn_readfirstlane((int)(t >> 6));
    const int img = blockIdx.y;
    const float4 *gt = p.gt_boxes + (long long)img * p.gmax;
    const float *labels = p.gt_labels + (long long)img * p.gmax;
    u64 *gkey = p.gkey + (long long)img * p.gmax;

    const int G = max(min(p.gt_count[img], p.gmax), 1);
    const float4 b_first = ld_off<float4>(gt, min(t, (unsigned)p.gmax - 1u) * 16u);
    const float l_first = ld_off<float>(labels, min(t, (unsigned)p.gmax - 1u) * 4u);

    const int a_first = (int)blockIdx.x * kBlock + wave * 64;
    const int last = p.num_anchors - 1;
    const bool valid = a_first + (int)lane < p.num_anchors;
    const int a0 = min(a_first + (int)lane, last);
    const int base = min(a_first, last);                 // a wave past the end stores nothing
    const unsigned out_off = (unsigned)(a0 - base);
    const long long out_row = (long long)img * p.num_anchors + base;
    const float4 anc = ld_off<float4>(p.anchors + base, out_off * 16u);
    const float ahw = anc.z / 2.0f, ahh = anc.w / 2.0f;
    const float ax0 = anc.x - ahw, ax1 = anc.x + ahw;

    float lo_x = ax0, lo_y = anc.y - ahh, hi_x = ax1, hi_y = anc.y + ahh;
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
    const float ex0 = fminf(fminf(sm.ext[0][0], sm.ext[0][1]), fminf(sm.ext[0][2], sm.ext[0][3]));
    const float ey0 = fminf(fminf(sm.ext[1][0], sm.ext[1][1]), fminf(sm.ext[1][2], sm.ext[1][3]));
    const float ex1 = fmaxf(fmaxf(sm.ext[2][0], sm.ext[2][1]), fmaxf(sm.ext[2][2], sm.ext[2][3]));
    const float ey1 = fmaxf(fmaxf(sm.ext[3][0], sm.ext[3][1]), fmaxf(sm.ext[3][2], sm.ext[3][3]));

    float ay0[1] = {anc.y - ahh}, ay1[1] = {anc.y + ahh};
    float a_area[1] = {clip0(ax1 - ax0, p.max_size) * clip0(ay1[0] - ay0[0], p.max_size)};
    if (!valid) ay1[0] = -3.0e38f;
    float best[1] = {0.0f};
    int sel[1] = {0};
    int par = 0;
    for (int g0 = 0; g0 < G; g0 += kListCap, par ^= 1) {
        const int g = g0 + (int)t;
        float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
        float lab = 0.0f;
        if (g < G) {
            b = (g0 == 0) ? b_first : ld_off<float4>(gt, (unsigned)g * 16u);
            lab = (g0 == 0) ? l_first : ld_off<float>(labels, (unsigned)g * 4u);
        }
        const int n_list = build_list(sm, par, t, lane, wave, g, g < G, b, lab, ex0, ey0, ex1, ey1, p.max_size);
        if (n_list == 0) continue;
        walk_list<1, false>(sm, n_list, lane, p.max_size, ax0, ax1, ay0, ay1, a_area, -3.0e38f, 3.0e38f, a0, 0,
                            best, sel);
        flush_keys(sm, n_list, t, gkey);
        __syncthreads();
    }
    // gather labels / boxes, threshold, encode every anchor (:89-96)
    const float labv = ld_off<float>(labels, (unsigned)sel[0] * 4u);
    const float4 gb = ld_off<float4>(gt, (unsigned)sel[0] * 16u);
    const float conf = (best[0] < p.thr) ? 0.0f : labv;
    const float4 tgt = encode_one(anc, gb, 0.1f, 0.2f);
    if (YTRUE) {
        // the wave's anchors base .. are one run of rows
        const int n_rows = max(0, min(64, p.num_anchors - a_first));
        store_ytrue_run<64>(match_dyn_lds + wave * kStageBytes64, p.y_true + out_row * 6, n_rows, lane, valid, conf,
                            tgt);
    } else if (valid) {
        store_target<false, DETAILS>(p, out_row, out_off, conf, tgt, sel[0], best[0]);
    }
}

template <bool GEOM, bool YTRUE>
__global__ __launch_bounds__(kBlock) void force_kernel(const MatchArgs p)
{
    __shared__ int s_anc[SSDBOX_MAX_GT];
    SSD_TRACE_BEGIN(p.tr_force);
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
        // The anchor is requested before the scan below, from the launch's own layer table and the
        // anchor-column table: one round trip to memory.
        float4 anc;
        if (GEOM) {
            int m = 0;                         // anchor_base grows with the layer
#pragma unroll
            for (int l = 1; l < SSDBOX_MAX_LAYERS; ++l) m += (l < p.n_layers && a >= p.ml[l].anchor_base) ? 1 : 0;
            int base = 0, row_anchors = 1, col_off = 0, cy_off = 0;
#pragma unroll
            for (int l = 0; l < SSDBOX_MAX_LAYERS; ++l) {      // (scalar registers cannot be indexed by a lane)
                if (l == m) {
                    base = p.ml[l].anchor_base;
                    row_anchors = p.ml[l].row_anchors;
                    col_off = p.ml[l].col_off;
                    cy_off = p.ml[l].cy_off;
                }
            }
            const int r = a - base;
            const int i = r / row_anchors;
            const float4 col = p.col_tab[col_off + (r - i * row_anchors)];
            anc = make_float4(col.x, p.cy_tab[cy_off + i], col.y, col.z);
        } else {
            anc = p.anchors[a];
        }
        // A later box that claims the same anchor wins it.  No early exit: the loads of the scan
        // are independent of each other and stay in flight together.
        bool later = false;
        for (int g2 = 0; g2 < G; g2 += 8) {
            int other[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) other[u] = s_anc[min(g2 + u, G - 1)];
#pragma unroll
            for (int u = 0; u < 8; ++u) later = later || (g2 + u > g && g2 + u < G && other[u] == a);
        }
        if (later) continue;
        const float conf = (1.0f < p.thr) ? 0.0f : label;
        const float4 tgt = encode_one(anc, box, 0.1f, 0.2f);
        store_target<YTRUE, true>(p, (long long)img * p.num_anchors + a, 0u, conf, tgt, g, 1.0f);
    }
    SSD_TRACE_END(p.tr_force);
}

#ifdef SSD_EXP_WGTRACE
unsigned long long *g_wgtrace = nullptr;
#endif

template <bool GEOM, bool YTRUE>
int launch_match(ssdbox_ctx *ctx, const MatchArgs &args_in, int batch, int n_tiles, size_t lds_bytes)
{
#if defined(SSD_EXP_TRACE) || defined(SSD_EXP_WGTRACE)
    MatchArgs args = args_in;
#ifdef SSD_EXP_TRACE
    args.tr = ssd_trace_slot(ctx, SSDBOX_STAGE_MATCH);
    args.tr_force = ssd_trace_slot(ctx, SSDBOX_STAGE_FORCE);
#endif
#ifdef SSD_EXP_WGTRACE
    args.wg = g_wgtrace;
#endif
#else
    const MatchArgs &args = args_in;
#endif
    const dim3 grid(n_tiles, batch);
    const bool details = args.match_idx != nullptr || args.match_iou != nullptr;
    hipEvent_t ev[2] = {nullptr, nullptr};
    const bool timed = ctx->timing && ctx->records.size() < kMaxTimingRecords && acquire_events(ctx, ev);
    void (*kernel)(const MatchArgs);
    if (!GEOM)
        kernel = details ? match_array_kernel<YTRUE, true> : match_array_kernel<YTRUE, false>;
    else if (args.gmax > kListCap)
        kernel = details ? match_geom_kernel<YTRUE, true, true> : match_geom_kernel<YTRUE, false, true>;
    else
        kernel = details ? match_geom_kernel<YTRUE, true, false> : match_geom_kernel<YTRUE, false, false>;
    hipExtLaunchKernelGGL(kernel, grid, dim3(kBlock), (unsigned)lds_bytes, ctx->stream, ev[0], ev[1], 0u, args);
    if (timed) {
        try {
            ctx->records.push_back({SSDBOX_STAGE_MATCH, ev[0], ev[1]});
        } catch (...) {               // the launch stays, its timing is dropped
        }
    }
    SSD_CHECK_HIP(hipGetLastError());
    launch_stage(ctx, SSDBOX_STAGE_FORCE, force_kernel<GEOM, YTRUE>, dim3(batch), dim3(kBlock), args);
    SSD_CHECK_HIP(hipGetLastError());
    return SSDBOX_OK;
}

// Groups of rows a wave of a fine-map tile takes in turn (tile height = 4 waves x this x 8 rows).
// More groups spread one prologue over more rows: with several batches in flight, 5 groups gave
// 2.3e6 images/s for match+encode at batch 32 @640x640 against 2.0e6 with 2 and 1.7e6 with 1,
// while the kernel alone on the chip takes 25 us against 22 us (fewer, longer workgroups).
// SSDBOX_MATCH_GROUPS overrides the choice for experiments.
int groups_per_wave()
{
    static const int v = [] {
        const char *e = getenv("SSDBOX_MATCH_GROUPS");
        const int n = e ? atoi(e) : 0;
        return (n >= 1 && n <= 8) ? n : 5;
    }();
    return v;
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
    SSD_REQUIRE(reinterpret_cast<uintptr_t>(d_y_true) % 8 == 0, "match_encode: y_true must be 8-byte aligned");
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

    DeviceGuard guard(ctx->device);
    const size_t need = (size_t)batch * (size_t)gmax;
    u64 *s_gkey = nullptr;
    if (static_scratch) {
        // The call's scratch comes out of the device's fixed array: nothing is allocated.  The
        // array keeps the previous call's contents, so the keys are cleared on every call.
        SSD_REQUIRE(ctx->static_scratch != nullptr, "match_encode_static: the context has no fixed scratch");
        if (need * sizeof(u64) > ctx->static_scratch_bytes) {
            set_error("match_encode_static: batch %d x gmax %d needs %zu bytes of scratch, the fixed array "
                      "holds %zu (use ssdbox_match_encode)", batch, gmax, need * sizeof(u64),
                      ctx->static_scratch_bytes);
            return SSDBOX_ERR_INVALID;
        }
        s_gkey = reinterpret_cast<u64 *>(ctx->static_scratch);
        SSD_CHECK_HIP(hipMemsetAsync(s_gkey, 0, need * sizeof(u64), ctx->stream));
    } else {
        int rc = ensure_buffer((void **)&ctx->gkey, &ctx->gkey_cap, need, sizeof(u64), true, ctx->stream);
        if (rc != SSDBOX_OK) return rc;
        if (ctx->gkey_dirty) {
            SSD_CHECK_HIP(hipMemsetAsync(ctx->gkey, 0, ctx->gkey_cap * sizeof(u64), ctx->stream));
            ctx->gkey_dirty = false;
        }
        s_gkey = ctx->gkey;
    }

    MatchArgs args;
    memset(&args, 0, sizeof(args));
    long long n_tiles = (num_anchors + kBlock - 1) / kBlock;
    size_t lds_bytes = 0;
    if (geom != nullptr) {
        // Tiling of this launch: fine maps get tall tiles whose waves take several groups of
        // rows; coarse maps, and every map when an image may hold more boxes than one list,
        // one group per wave.  Tiles are numbered from the first (finest) layer to the last; the
        // order in which the layers' workgroups start measured within the run-to-run spread
        // (26-30 us for the bench batch, 118-123 us with 256 boxes in every image, for six
        // orders), only images taking turns tile by tile is slower (39 us: it interleaves the
        // output rows of all images).
        const bool multi = gmax > kListCap;
        const GeomDev &g = geom->host;
        args.n_layers = g.n_layers;
        for (int m = 0; m < g.n_layers; ++m) {
            const LayerDev &L = g.layer[m];
            MatchLayer &M = args.ml[m];
            const int rows_per_group = 2 * (L.coarse ? 1 : kRowsPerThread);
            const int groups_total = (L.fm_h + rows_per_group - 1) / rows_per_group;
            int n_groups = kBlock / 64;
            if (!multi && !L.coarse) {
                const int cap = (kBlock / 64) * groups_per_wave();
                const int tiles_y = (groups_total + cap - 1) / cap;
                n_groups = (groups_total + tiles_y - 1) / tiles_y;
            }
            const int tile_rows = n_groups * rows_per_group;
            const int tiles_y = (L.fm_h + tile_rows - 1) / tile_rows;
            M.fm_h = L.fm_h;
            M.row_anchors = L.row_anchors;
            M.n_box = L.n_box;
            M.anchor_base = L.anchor_base;
            M.tiles_x = L.tiles_x;
            M.tile_base = tiles_y;                          // count for now, base below
            M.tilesx_magic = L.tilesx_magic;
            M.nbox_magic = L.nbox_magic;
            M.col_off = L.col_off;
            M.cy_off = L.cy_off;
            M.n_groups = n_groups;
            M.coarse = L.coarse;
            M.half_wmax = L.wmax / 2.0f;
            M.half_hmax = L.hmax / 2.0f;
            args.nbox_max = L.n_box > args.nbox_max ? L.n_box : args.nbox_max;
            args.rows_max = tile_rows > args.rows_max ? tile_rows : args.rows_max;
        }
        n_tiles = 0;
        for (int m = 0; m < g.n_layers; ++m) {
            const long long count = (long long)args.ml[m].tiles_x * args.ml[m].tile_base;
            args.ml[m].tile_base = (int)n_tiles;
            n_tiles += count;
        }
        SSD_REQUIRE(args.rows_max <= kBlock, "match_encode: tile of %d rows", args.rows_max);

        lds_bytes = (size_t)(kListCap + 1) * args.nbox_max * sizeof(float2) +
                    (size_t)args.rows_max * (1 + args.nbox_max) * sizeof(float);
        if (d_y_true != nullptr) {
            lds_bytes = (lds_bytes + 15) & ~(size_t)15;
            args.stage_off = (int)lds_bytes;
            lds_bytes += (size_t)(kBlock / 32) * kStageBytes32;
        }
    } else if (d_y_true != nullptr) {
        lds_bytes = (size_t)(kBlock / 64) * kStageBytes64;
    }
    args.geom = geom ? geom->d_geom : nullptr;
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

    // Event timing takes events from a pool that may have to grow: not on the path that
    // allocates nothing.
    const bool timing = ctx->timing;
    if (static_scratch) ctx->timing = false;
    else ctx->gkey_dirty = true;    // cleared again only when every launch was enqueued
    int rc;
    if (geom != nullptr)
        rc = d_y_true ? launch_match<true, true>(ctx, args, batch, (int)n_tiles, lds_bytes)
                      : launch_match<true, false>(ctx, args, batch, (int)n_tiles, lds_bytes);
    else
        rc = d_y_true ? launch_match<false, true>(ctx, args, batch, (int)n_tiles, lds_bytes)
                      : launch_match<false, false>(ctx, args, batch, (int)n_tiles, 0);
    ctx->timing = timing;
    if (rc == SSDBOX_OK && !static_scratch) ctx->gkey_dirty = false;
    return rc;
}

}  // namespace

#ifdef SSD_EXP_WGTRACE
extern "C" int ssdbox_debug_set_wgtrace(void *dev_ptr)
{
    g_wgtrace = static_cast<unsigned long long *>(dev_ptr);
    return 0;
}
#endif

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
