// Inference side of the SSD box pipeline for a whole batch: score threshold, top-k,
// decode of the survivors, greedy NMS and detection assembly (reference inference.py:33-83
// around utils/box_utils.py:126-147 and :184-222).
//
// Four launches per batch, none of them per image:
//   1. select_*_kernel   streams the (B,A,6) head output once with 16-byte loads, keeps
//      score > threshold and writes a 64-bit rank key (score order bits, ~anchor index) per
//      survivor into the wave's own 256-slot segment of the candidate buffer, plus the
//      segment's count.  No atomics, no barriers: a wave never waits on another.
//   2. topk_sort_kernel  one workgroup per image: gathers the segments, cuts the candidates
//      down to the `limit` largest keys with a histogram select (linear score bins first,
//      key bit fields after), sorts them one key per thread (shuffles inside a wave, LDS
//      between waves) and decodes only those boxes -- threshold and top-k depend on the
//      scores alone.  Key order is "score descending, lower index first", which is
//      tf.argsort(direction='DESCENDING') / top_k's order.
//   3. nms_mask_kernel   lower-triangular K x K suppression bits, one wave per 64 x 64 tile:
//      the later box j is wave-uniform, lanes hold 64 earlier boxes i, and a ballot of
//      !(IoU <= threshold) is the whole 64-bit word (j, word of i).
//   4. nms_scan_kernel   one workgroup per image, one thread per ranked box: bands of 64
//      boxes are resolved in order, inside a band by rounds of ballots; kept boxes are
//      compacted in rank order into the detection rows.
#include <string.h>

#include <algorithm>

#include "ssdbox_internal.h"

namespace {

typedef unsigned long long u64;

constexpr int kSeg = 256;                  // candidate slots per segment = anchors per wave
constexpr int kSelBlock = 256;
constexpr int kLimit = 1024;                // ranked boxes per image on the fast path (mask + scan)
constexpr int kBigLimit = SSDBOX_MAX_NMS_LIMIT;   // ... on the general path (sequential greedy loop)
constexpr int kBigRanks = kBigLimit / 1024;
constexpr int kWords = kLimit / 64;
constexpr int kCap = 8192;                 // keys the top-k kernel stages in LDS
constexpr int kBucketMax = 256;            // bucket size resolved by direct ranking
constexpr int kBins = 2048;
constexpr int kTopkBlock = 1024;
constexpr float kNmsMaxSize = 512.0f;      // compute_iou's default inside compute_nms (:205)

// ------------------------------------------------------------------------------ select
// Face probability of a two-class softmax over raw head logits (reference models/ssd.py:95;
// the commented-out tf.math.softmax of inference.py:40): exp of the shifted logits times the
// reciprocal of their sum, every step rounded to float32.
SSD_DEV float softmax_face(float l0, float l1)
{
    const float m = fmaxf(l0, l1);
    const float e0 = ssd_expf(l0 - m), e1 = ssd_expf(l1 - m);
    return e1 * (1.0f / (e0 + e1));
}

// Head rows are 24 bytes [bg, face, tx, ty, tw, th]; read as a float4 stream, the face score
// of anchor 2q sits in .y of float4 3q and that of anchor 2q+1 in .w of float4 3q+1.  A wave
// owns 384 consecutive float4 (256 anchors): six fully coalesced 1-KiB loads.
// A wave takes (image, segment) items with a grid stride, the loads of its next item in flight
// while it works on the current one.  The grid is capped at kSelWgsPerCu workgroups per CU; at the
// benchmark size (12 768 items) that is one item per wave.  A small persistent grid (4 workgroups
// per CU, three items per wave) was measured and is slower, alone (15.2 against 13.3 us) and with
// four batches in flight (decode+NMS 1.38e6 against 1.48e6 images/s).
constexpr int kSelWgsPerCu = 64;
constexpr int kHistSegs = 8;               // segments per workgroup of the histogram kernels
SSD_DEV int prune_bin(u64 key, int has_range, float lo, float scale);

// One item (256 anchors = 384 float4 held by the wave): the keys of the anchors that pass, packed
// into `out`; returns their number.  on_key(key) runs on every lane that holds a passing anchor.
template <bool LOGITS, typename F>
SSD_DEV int compact_rows6(const float4 (&v)[6], unsigned f0, unsigned nf, float thr, u64 *out, u64 lt, F on_key)
{
    int off = 0;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const unsigned f = f0 + 64u * k;
        const unsigned q = __umulhi(f, 0xAAAAAAABu) >> 1;      // f / 3
        const unsigned r = f - 3u * q;
        const float s1 = (r == 0u) ? v[k].y : v[k].w;
        const float score = LOGITS ? softmax_face((r == 0u) ? v[k].x : v[k].z, s1) : s1;
        const bool pass = (f < nf) && (r != 2u) && (score > thr);   // strict >, NaN fails
        const u64 bal = __ballot(pass);
        if (pass) {
            const u64 key = rank_key(score, 2u * q + r);
            out[off + __popcll(bal & lt)] = key;
            on_key(key);
        }
        off += __popcll(bal);
    }
    return off;
}

template <bool LOGITS>
__global__ __launch_bounds__(kSelBlock) void select_rows6_kernel(const float4 *__restrict__ y, int n,
                                                                 float thr, u64 *__restrict__ cand,
                                                                 int *__restrict__ seg_count, int nseg,
                                                                 int n_items, unsigned nseg_magic SSD_TRACE_PARAM)
{
    SSD_TRACE_BEGIN(tr);
    const int lane = threadIdx.x & 63;
    const int stride = (int)gridDim.x * (kSelBlock / 64);
    int item = blockIdx.x * (kSelBlock / 64) + (threadIdx.x >> 6);
    if (item >= n_items) return;
    const unsigned nf = (unsigned)n / 2u * 3u;
    const u64 lt = (1ull << lane) - 1ull;

    auto fetch = [&](int it, float4 (&v)[6]) {
        const int img = udiv_magic(it, nseg, nseg_magic);
        const int seg = it - img * nseg;
        const float4 *src = y + (long long)img * nf;
        const unsigned f0 = (unsigned)seg * 384u + (unsigned)lane;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const unsigned f = f0 + 64u * k;
            v[k] = (f < nf) ? src[f] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };

    float4 v[6];
    fetch(item, v);
    while (true) {
        const int next = item + stride;
        float4 vn[6];
        if (next < n_items) fetch(next, vn);
        const int img = udiv_magic(item, nseg, nseg_magic);
        const int seg = item - img * nseg;
        const unsigned f0 = (unsigned)seg * 384u + (unsigned)lane;
        const int off = compact_rows6<LOGITS>(v, f0, nf, thr, cand + (long long)item * kSeg, lt, [](u64) {});
        if (lane == 0) seg_count[item] = off;
        if (next >= n_items) break;
        item = next;
#pragma unroll
        for (int k = 0; k < 6; ++k) v[k] = vn[k];
    }
    SSD_TRACE_END(tr);
}

// The same with the per-image histogram of the candidates' bins (what cand_prune_kernel cuts
// by), for calls where most anchors are candidates: a workgroup takes kHistSegs consecutive
// segments of one image, counts in LDS and adds its non-empty bins to the image's histogram,
// which saves the pass of cand_hist_kernel over the keys and its launch.
template <bool LOGITS>
__global__ __launch_bounds__(kSelBlock) void select_rows6_hist_kernel(const float4 *__restrict__ y, int n, float thr,
                                                                      u64 *__restrict__ cand,
                                                                      int *__restrict__ seg_count, int nseg,
                                                                      int has_range, float lo, float scale,
                                                                      unsigned int *__restrict__ hist)
{
    __shared__ unsigned int s_hist[kBins];
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int img = blockIdx.y;
    const unsigned nf = (unsigned)n / 2u * 3u;
    const u64 lt = (1ull << lane) - 1ull;
    const float4 *src = y + (long long)img * nf;
    const int s_end = min(((int)blockIdx.x + 1) * kHistSegs, nseg);
    int seg = blockIdx.x * kHistSegs + wave;

    auto fetch = [&](int sg, float4 (&v)[6]) {
        const unsigned f0 = (unsigned)sg * 384u + (unsigned)lane;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const unsigned f = f0 + 64u * k;
            v[k] = (f < nf) ? src[f] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    float4 v[6];
    if (seg < s_end) fetch(seg, v);
    for (int b = t; b < kBins; b += kSelBlock) s_hist[b] = 0u;
    __syncthreads();
    while (seg < s_end) {
        const int next = seg + kSelBlock / 64;
        float4 vn[6];
        if (next < s_end) fetch(next, vn);
        const unsigned f0 = (unsigned)seg * 384u + (unsigned)lane;
        const long long item = (long long)img * nseg + seg;
        const int off = compact_rows6<LOGITS>(v, f0, nf, thr, cand + item * kSeg, lt, [&](u64 key) {
            atomicAdd(&s_hist[prune_bin(key, has_range, lo, scale)], 1u);
        });
        if (lane == 0) seg_count[item] = off;
        seg = next;
#pragma unroll
        for (int k = 0; k < 6; ++k) v[k] = vn[k];
    }
    __syncthreads();
    for (int b = t; b < kBins; b += kSelBlock) {
        const unsigned int c = s_hist[b];
        if (c != 0u) atomicAdd(&hist[(long long)img * kBins + b], c);
    }
}

// Any score layout (element stride / offset), optional threshold: used for caller-supplied
// score arrays and for head outputs whose rows are not 16-byte aligned (odd anchor count).
__global__ __launch_bounds__(kSelBlock) void select_strided_kernel(
    const float *__restrict__ scores, long long img_stride, int elem_stride, int elem_off, int n,
    float thr, int use_thr, int logits, u64 *__restrict__ cand, int *__restrict__ seg_count, int nseg)
{
    const int lane = threadIdx.x & 63;
    const int seg = blockIdx.x * (kSelBlock / 64) + (threadIdx.x >> 6);
    if (seg >= nseg) return;
    const int img = blockIdx.y;
    const float *sc = scores + (long long)img * img_stride;
    float s[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = seg * kSeg + k * 64 + lane;
        s[k] = (i < n) ? sc[(long long)i * elem_stride + elem_off] : 0.0f;
        // the background logit sits right before the face logit
        if (logits) s[k] = softmax_face((i < n) ? sc[(long long)i * elem_stride + elem_off - 1] : 0.0f, s[k]);
    }
    u64 *out = cand + ((long long)img * nseg + seg) * kSeg;
    const u64 lt = (1ull << lane) - 1ull;
    int off = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = seg * kSeg + k * 64 + lane;
        const bool pass = (i < n) && (use_thr ? (s[k] > thr) : true);
        const u64 bal = __ballot(pass);
        if (pass) out[off + __popcll(bal & lt)] = rank_key(s[k], (uint32_t)i);
        off += __popcll(bal);
    }
    if (lane == 0) seg_count[(long long)img * nseg + seg] = off;
}

// -------------------------------------------------------------------------------- prune
// When most anchors are candidates (threshold 0 as in reference evaluation.py:67, or a large
// caller-supplied score array) one workgroup per image would spend its time reading hundreds
// of kilobytes of keys twice.  Two small grid-wide kernels cut the candidates down first:
//   cand_hist_kernel     per-image histogram of a monotone 2048-bin map of the keys
//   cand_prune_kernel    every workgroup finds the bin where the count from the top crosses
//                        `limit`; every wave rewrites its own segment with the keys at or above
//                        that bin (all of the `limit` largest keys are among them)
// The top-k kernel then runs unchanged on what is left.
// Monotone non-decreasing map from rank key to one of kBins bins: linear in the score when the
// scores are known to lie in [lo, lo + 2047 / scale], the leading key bits otherwise.
SSD_DEV int prune_bin(u64 key, int has_range, float lo, float scale)
{
    if (!has_range) return (int)(key >> 53);
    const uint32_t u = (uint32_t)(key >> 32);
    if (u == 0u) return 0;                                        // NaN scores rank last
    const uint32_t b = (u & 0x80000000u) ? (u ^ 0x80000000u) : ~u;  // undo score_key
    const float x = (u2f(b) - lo) * scale;
    return (int)fminf(fmaxf(x, 0.0f), (float)(kBins - 1));
}

__global__ __launch_bounds__(kSelBlock) void cand_hist_kernel(const u64 *__restrict__ cand,
                                                              const int *__restrict__ seg_count, int nseg,
                                                              int has_range, float lo, float scale,
                                                              unsigned int *__restrict__ hist)
{
    __shared__ unsigned int s_hist[kBins];
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int img = blockIdx.y;
    for (int b = t; b < kBins; b += kSelBlock) s_hist[b] = 0u;
    __syncthreads();
    const int s_end = min(((int)blockIdx.x + 1) * kHistSegs, nseg);
    for (int s = blockIdx.x * kHistSegs + wave; s < s_end; s += kSelBlock / 64) {
        const int cnt = seg_count[(long long)img * nseg + s];
        const u64 *src = cand + ((long long)img * nseg + s) * kSeg;
        u64 key[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) key[q] = src[min(q * 64 + lane, max(cnt - 1, 0))];
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (q * 64 + lane < cnt) atomicAdd(&s_hist[prune_bin(key[q], has_range, lo, scale)], 1u);
    }
    __syncthreads();
    for (int b = t; b < kBins; b += kSelBlock) {
        const unsigned int c = s_hist[b];
        if (c != 0u) atomicAdd(&hist[(long long)img * kBins + b], c);
    }
}

// grid (ceil(nseg / 4), batch), one wave per segment
__global__ __launch_bounds__(kSelBlock) void cand_prune_kernel(u64 *__restrict__ cand,
                                                               int *__restrict__ seg_count, int nseg,
                                                               int limit, int has_range, float lo, float scale,
                                                               const unsigned int *__restrict__ hist)
{
    static_assert(kBins == 8 * kSelBlock, "eight bins per thread");
    __shared__ unsigned int s_wtot[kSelBlock / 64];
    __shared__ int s_cut;
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int img = blockIdx.y;
    const int seg = blockIdx.x * (kSelBlock / 64) + wave;
    // the segment's keys are fetched first: the cut does not depend on them
    const int cnt = (seg < nseg) ? seg_count[(long long)img * nseg + seg] : 0;
    u64 *mine = cand + ((long long)img * nseg + min(seg, nseg - 1)) * kSeg;
    u64 key[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) key[q] = mine[min(q * 64 + lane, max(cnt - 1, 0))];

    // the bin where the count from the top reaches min(total, limit)
    const uint4 lo4 = reinterpret_cast<const uint4 *>(hist + (long long)img * kBins)[2 * t];
    const uint4 hi4 = reinterpret_cast<const uint4 *>(hist + (long long)img * kBins)[2 * t + 1];
    const unsigned int h[8] = {lo4.x, lo4.y, lo4.z, lo4.w, hi4.x, hi4.y, hi4.z, hi4.w};
    unsigned int sum = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) sum += h[j];
    unsigned int incl = sum;           // suffix sum over this wave's higher lanes
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned int v = __shfl_down(incl, o);
        if (lane + o < 64) incl += v;
    }
    if (lane == 0) s_wtot[wave] = incl;
    if (t == 0) s_cut = 0;             // fewer candidates than `limit`: everything stays
    __syncthreads();
    unsigned int above = incl - sum;
    for (int w = wave + 1; w < kSelBlock / 64; ++w) above += s_wtot[w];
#pragma unroll
    for (int j = 7; j >= 0; --j) {
        if (above < (unsigned int)limit && above + h[j] >= (unsigned int)limit) s_cut = 8 * t + j;
        above += h[j];
    }
    __syncthreads();
    const int cut = s_cut;
    if (seg >= nseg || cut == 0) return;

    // this wave's segment, rewritten in place with the keys at or above the cut
    const u64 lt = (1ull << lane) - 1ull;
    int off = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const bool pass = (q * 64 + lane < cnt) && prune_bin(key[q], has_range, lo, scale) >= cut;
        const u64 bal = __ballot(pass);
        if (pass) mine[off + __popcll(bal & lt)] = key[q];
        off += __popcll(bal);
    }
    if (lane == 0) seg_count[(long long)img * nseg + seg] = off;
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
    int logits;                   // y_pred columns 0 and 1 are raw logits
};

struct TopkArgs {
    const u64 *cand;              // (batch, nseg * kSeg)
    const int *seg_count;         // (batch, nseg)
    int nseg;
    int n_per_image;
    int limit;
    int has_range;                // scores are expected in [lo, lo + 2047 / scale]
    float lo, scale;
    BoxSource src;
    GeomDev geom;                 // by value (mode 0): layer constants come with the arguments
    float4 *sboxes;
    float *sscores;
    int *sidx;
    int *keff;
    unsigned int *hist_clear;     // not NULL: the image's pruning histogram, zeroed for the next call
    // General path, ranks taken kBigLimit at a time: this launch ranks the keys that follow the
    // first rank_offset ones.  floor_key holds, per image, the smallest key an earlier launch
    // ranked (read when rank_offset > 0: only smaller keys count) and takes this launch's.
    int rank_offset;
    u64 *floor_key;
#ifdef SSD_EXP_TRACE
    unsigned long long *tr;
#endif
};

// Monotone non-decreasing map from rank key to one of kBins linear score bins.
SSD_DEV int linear_bin(u64 key, float lo, float scale) { return prune_bin(key, 1, lo, scale); }

// Current bucket of the select: keys whose linear bin (if used) and leading key bits match.
struct Bucket {
    int use_lin;      // 1 once the linear pass has fixed a bin
    int lin;
    u64 pmask, prefix;
};

SSD_DEV bool in_bucket(const Bucket &b, u64 key, float lo, float scale)
{
    if (b.use_lin && linear_bin(key, lo, scale) != b.lin) return false;
    return (key & b.pmask) == b.prefix;
}

SSD_DEV bool above_bucket(const Bucket &b, u64 key, float lo, float scale)
{
    if (b.use_lin) {
        const int l = linear_bin(key, lo, scale);
        if (l != b.lin) return l > b.lin;
    }
    return (key & b.pmask) > b.prefix;
}

// Value of `x` in lane (lane ^ J), J a power of two below 64: quad permutes for 1 and 2, a
// swizzle (xor inside each half wave, no address register) for 4..16, a permute across the
// halves for 32.  (The swizzles run in the LDS pipeline next to the VALU, which is what bounds
// the network with 16 waves on one CU: row shifts for 4 and 8 and the gfx950 row / half-wave
// swaps for 16 and 32 instead measured 4.4 us against 3.9 us for the sort.)
template <int J>
SSD_DEV unsigned int xor_lane(unsigned int x)
{
    if constexpr (J == 1) {
        return (unsigned int)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xf, 0xf, true);   // [1,0,3,2]
    } else if constexpr (J == 2) {
        return (unsigned int)__builtin_amdgcn_update_dpp(0, (int)x, 0x4E, 0xf, 0xf, true);   // [2,3,0,1]
    } else if constexpr (J < 32) {
        return (unsigned int)__builtin_amdgcn_ds_swizzle((int)x, (J << 10) | 0x1F);
    } else {
        return (unsigned int)__shfl_xor((int)x, 32);
    }
}

// Bitonic network over kTopkBlock keys, one per thread, descending: compare-exchange
// distance J inside blocks of K keys, unrolled over the whole network.  Stages wider than
// the P keys actually present are skipped.  Distances below 64 stay inside a wave;
// larger ones go through LDS, alternating two buffers so that one barrier per step is enough.
template <int K, int J>
struct SortStep {
    static SSD_DEV void run(u64 &key, int t, u64 (*xchg)[kTopkBlock], int &phase, int P)
    {
        if (K <= P) {
            u64 other;
            if constexpr (J < 64) {
                const unsigned int olo = xor_lane<J>((unsigned int)key);
                const unsigned int ohi = xor_lane<J>((unsigned int)(key >> 32));
                other = ((u64)ohi << 32) | olo;
            } else {
                u64 *buf = xchg[phase & 1];
                ++phase;
                buf[t] = key;
                __syncthreads();
                other = buf[t ^ J];
            }
            const bool take_max = (((t & J) == 0) == ((t & K) == 0));
            const bool other_gt = other > key;
            key = (other_gt == take_max) ? other : key;
        }
        if constexpr (J > 1) {
            SortStep<K, J / 2>::run(key, t, xchg, phase, P);
        } else if constexpr (K < kTopkBlock) {
            SortStep<K * 2, K>::run(key, t, xchg, phase, P);
        }
    }
};

// Calls f(key) for every candidate key of an image still in global memory.  A wave takes
// four segments at a time: sixteen 512-byte loads in flight per wave, addresses clamped into
// the segment so that no load depends on a count.
template <typename F>
SSD_DEV void for_each_candidate(const u64 *__restrict__ cand, const int *__restrict__ segc, int nseg,
                                int wave, int lane, F f)
{
    constexpr int kWaves = kTopkBlock / 64;
    for (int s0 = wave * 4; s0 < nseg; s0 += kWaves * 4) {
        int cnt[4];
        u64 key[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) cnt[u] = (s0 + u < nseg) ? segc[s0 + u] : 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const u64 *src = cand + (long long)min(s0 + u, nseg - 1) * kSeg;
#pragma unroll
            for (int q = 0; q < 4; ++q) key[u][q] = src[min(q * 64 + lane, max(cnt[u] - 1, 0))];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (q * 64 + lane < cnt[u]) f(key[u][q]);
    }
}

// RANKS: ranked boxes per thread.  1: up to 1024 ranks, sorted in registers (the path every
// caller of the reference takes: limit 1000 at inference.py:60, 200 by default).  kBigRanks: up to
// kBigLimit ranks, sorted in LDS.
template <int RANKS>
__global__ __launch_bounds__(kTopkBlock) void topk_sort_kernel(const TopkArgs p)
{
    constexpr int kSortCap = kTopkBlock * RANKS;          // keys the final sort takes
    __shared__ u64 s_keys[kCap];
    __shared__ unsigned int s_hist[kBins];
    __shared__ u64 s_xchg[2][kTopkBlock];
    __shared__ u64 s_big[RANKS > 1 ? kSortCap : 1];
    __shared__ u64 s_bucket[kBucketMax];
    __shared__ unsigned int s_wtot[kTopkBlock / 64];
    __shared__ unsigned int s_sel[3];
    __shared__ int s_count, s_count2;

    SSD_TRACE_BEGIN(p.tr);
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = t >> 6;
    const int img = blockIdx.x;
    const int nseg = p.nseg;
    const u64 *cand = p.cand + (long long)img * nseg * kSeg;
    const int *segc = p.seg_count + (long long)img * nseg;
    const float lo = p.lo, scale = p.scale;
#ifdef SSD_EXP_STAMPS              // timing experiment (profiles/topk_stamps.py): phase stamps of image 0
    unsigned long long st[8];
    int sti = 0;
#define SSD_STAMP() st[sti++] = wall_clock64()
#else
#define SSD_STAMP()
#endif
    SSD_STAMP();

    // ---- total number of candidates, and where this thread's segments (t, t + 1024, ...) go
    // if everything is staged: the order of the staged keys does not matter
    const int cnt_first = (t < nseg) ? segc[t] : 0;
    // (requested together with the count: a segment holds 7 keys on average at a 2-3 % pass rate,
    // so staging every candidate rarely needs a second trip to memory; slots past the count hold
    // whatever an earlier call left and are not used)
    u64 first8[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) first8[u] = cand[(long long)min(t, nseg - 1) * kSeg + u];
    unsigned int mine = (unsigned int)cnt_first;
    for (int s = t + kTopkBlock; s < nseg; s += kTopkBlock) mine += (unsigned int)segc[s];
    unsigned int incl = mine;
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned int v = __shfl_up(incl, o);
        if (lane >= o) incl += v;
    }
    if (lane == 63) s_wtot[wave] = incl;
    if (t == 0) {
        s_count = 0;
        s_count2 = 0;
    }
    __syncthreads();
    int n = 0;
    int stage_pos = (int)(incl - mine);
#pragma unroll
    for (int w = 0; w < kTopkBlock / 64; ++w) {
        if (w < wave) stage_pos += (int)s_wtot[w];
        n += (int)s_wtot[w];
    }
    // (keys are unique, so the launches before this one ranked exactly the rank_offset largest)
    const bool later_ranks = RANKS > 1 && p.rank_offset > 0;
    const u64 ceil_key = later_ranks ? p.floor_key[img] : 0ull;
    if (later_ranks) n = max(n - p.rank_offset, 0);
    const int K = min(n, p.limit);
    if (t == 0) p.keff[img] = K;
    if (p.hist_clear != nullptr) {
        static_assert(kBins == 2 * kTopkBlock, "two bins per thread");
        reinterpret_cast<uint2 *>(p.hist_clear + (long long)img * kBins)[t] = make_uint2(0u, 0u);
    }
    if (n == 0) return;
    SSD_STAMP();

    Bucket bk;
    bk.use_lin = 0;
    bk.lin = 0;
    bk.pmask = 0ull;
    bk.prefix = 0ull;
    int k_rem = K, n_above = 0, bucket = n;
    int next_digit = p.has_range ? -1 : 0;      // -1: linear bins, 0..5: key bit fields
    bool in_lds = false;
    int m_lds = 0;

    // Every candidate key still in global memory: by whole segments when they are well filled
    // (threshold near 0), by a thread per segment when they hold a few keys each (more
    // candidates than LDS takes only because the image has many anchors).
    const bool dense = n >= nseg * 96;
    auto for_each_key = [&](auto f_all) {
        auto f = [&](u64 key) {
            if (!later_ranks || key < ceil_key) f_all(key);
        };
        if (dense) {
            for_each_candidate(cand, segc, nseg, wave, lane, f);
            return;
        }
        for (int s = t; s < nseg; s += kTopkBlock) {
            const int cnt = (s == t) ? cnt_first : segc[s];
            const u64 *src = cand + (long long)s * kSeg;
            for (int i0 = 0; i0 < cnt; i0 += 8) {
                u64 k8[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) k8[u] = src[min(i0 + u, cnt - 1)];
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (i0 + u < cnt) f(k8[u]);
            }
        }
    };

    while (true) {
        if (!in_lds && n_above + bucket <= kCap) {
            // ---- stage every key at or above the bucket in LDS
            if (bucket == n && !later_ranks) {
                // everything: a thread copies the keys of its own segments, eight loads in
                // flight at a time (a segment holds 7 keys on average at a 2-3 % pass rate), to
                // the place the scan above gave it
                int pos = stage_pos;
                for (int s = t; s < nseg; s += kTopkBlock) {
                    const int cnt = (s == t) ? cnt_first : segc[s];
                    const u64 *src = cand + (long long)s * kSeg;
                    for (int i0 = 0; i0 < cnt; i0 += 8) {
                        u64 key[8];
                        if (s == t && i0 == 0) {
#pragma unroll
                            for (int u = 0; u < 8; ++u) key[u] = first8[u];
                        } else {
#pragma unroll
                            for (int u = 0; u < 8; ++u) key[u] = src[min(i0 + u, cnt - 1)];
                        }
#pragma unroll
                        for (int u = 0; u < 8; ++u)
                            if (i0 + u < cnt) s_keys[pos + i0 + u] = key[u];
                    }
                    pos += cnt;
                }
                if (t == 0) s_count = n;
            } else {
                for_each_key([&](u64 key) {
                    if (above_bucket(bk, key, lo, scale) || in_bucket(bk, key, lo, scale))
                        s_keys[atomicAdd(&s_count, 1)] = key;
                });
            }
            __syncthreads();
            m_lds = s_count;           // == n_above + bucket
            in_lds = true;
        }
        if (in_lds && (m_lds <= kSortCap || bucket <= kBucketMax || bucket == k_rem)) break;

        // ---- one histogram pass over the current bucket
        const int dg = next_digit++;
        const int shift = (dg < 5) ? 53 - 11 * dg : 0;
        const unsigned int dmask = (dg < 5) ? 0x7FFu : 0x1FFu;
        s_hist[t] = 0u;
        s_hist[t + kTopkBlock] = 0u;
        __syncthreads();
        if (in_lds) {
            for (int i = t; i < m_lds; i += kTopkBlock) {
                const u64 key = s_keys[i];
                if (in_bucket(bk, key, lo, scale))
                    atomicAdd(&s_hist[dg < 0 ? linear_bin(key, lo, scale)
                                             : (int)((unsigned int)(key >> shift) & dmask)], 1u);
            }
        } else {
            for_each_key([&](u64 key) {
                if (in_bucket(bk, key, lo, scale))
                    atomicAdd(&s_hist[dg < 0 ? linear_bin(key, lo, scale)
                                             : (int)((unsigned int)(key >> shift) & dmask)], 1u);
            });
        }
        __syncthreads();
        // ---- the digit where the count from the top crosses k_rem
        const unsigned int h0 = s_hist[2 * t], h1 = s_hist[2 * t + 1];
        const unsigned int pair = h0 + h1;
        unsigned int incl = pair;                  // suffix sum over this wave's higher lanes
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned int v = __shfl_down(incl, o);
            if (lane + o < 64) incl += v;
        }
        if (lane == 0) s_wtot[wave] = incl;
        __syncthreads();
        unsigned int above1 = incl - pair;         // keys in digits above 2t+1
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
        n_above += (int)s_sel[1];
        k_rem -= (int)s_sel[1];
        bucket = (int)s_sel[2];
        if (dg < 0) {
            bk.use_lin = 1;
            bk.lin = (int)digit;
        } else {
            bk.prefix |= (u64)digit << shift;
            bk.pmask |= (u64)dmask << shift;
        }
        // after the last bit field the keys (unique) are fully resolved: bucket == 1 == k_rem
    }

    SSD_STAMP();
    // ---- the keys to sort: everything staged, or exactly the K largest
    u64 key = 0ull;                    // padding ranks below every real key
    int m_sort;
    u64 *s_out = RANKS > 1 ? s_big : s_xchg[0];
    if (m_lds <= kSortCap) {
        m_sort = m_lds;
        if (RANKS == 1) {
            if (t < m_lds) key = s_keys[t];
        } else {
            for (int i = t; i < m_lds; i += kTopkBlock) s_out[i] = s_keys[i];
        }
    } else {
        const bool whole_bucket = (bucket == k_rem);
        for (int i = t; i < m_lds; i += kTopkBlock) {
            const u64 kk = s_keys[i];
            if (above_bucket(bk, kk, lo, scale)) {
                s_out[atomicAdd(&s_count2, 1)] = kk;
            } else if (in_bucket(bk, kk, lo, scale)) {
                if (whole_bucket) s_out[atomicAdd(&s_count2, 1)] = kk;
                else s_bucket[atomicAdd(&s_count, 1) - m_lds] = kk;    // s_count restarts at m_lds
            }
        }
        __syncthreads();
        if (!whole_bucket && t < bucket) {
            const u64 me = s_bucket[t];
            int rank = 0;
            for (int i = 0; i < bucket; ++i) rank += (s_bucket[i] > me) ? 1 : 0;
            if (rank < k_rem) s_out[atomicAdd(&s_count2, 1)] = me;
        }
        __syncthreads();
        m_sort = K;                    // == s_count2
        if (RANKS == 1) {
            if (t < K) key = s_out[t];
            __syncthreads();           // s_xchg is reused by the sort
        }
    }

    SSD_STAMP();
    int P = 64;
    while (P < m_sort) P <<= 1;
    if (RANKS == 1) {
        // ---- bitonic sort, descending, one key per thread
        int phase = 0;
        SortStep<2, 1>::run(key, t, s_xchg, phase, P);
    } else {
        // ---- bitonic sort, descending, in LDS: every step one pass over the P / 2 pairs
        for (int i = m_sort + t; i < P; i += kTopkBlock) s_out[i] = 0ull;
        __syncthreads();
        for (int kb = 2; kb <= P; kb <<= 1) {
            for (int j = kb >> 1; j > 0; j >>= 1) {
                for (int i = t; i < P / 2; i += kTopkBlock) {
                    const int a = ((i & ~(j - 1)) << 1) | (i & (j - 1));
                    const int b = a | j;
                    const u64 ka = s_out[a], kbv = s_out[b];
                    const bool descending = (a & kb) == 0;
                    if (descending ? (ka < kbv) : (ka > kbv)) {
                        s_out[a] = kbv;
                        s_out[b] = ka;
                    }
                }
                __syncthreads();
            }
        }
    }

    SSD_STAMP();
    if (RANKS > 1 && p.floor_key != nullptr && t == 0) p.floor_key[img] = s_out[K - 1];
    // ---- ranks t, t + 1024, ... below K: decode / gather their boxes
#pragma unroll
    for (int q = 0; q < RANKS; ++q) {
        const int r = q * kTopkBlock + t;
        if (r >= K) {
            // Ranks past K hold a box of their own (finite, positive area): the mask kernel reads a
            // tile's boxes without waiting for K, and what it computes for such a rank is never used.
            if (RANKS == 1) p.sboxes[(long long)img * kSortCap + r] = make_float4(-2.0f, -2.0f, -1.0f, -1.0f);
            continue;
        }
        if (RANKS > 1) key = s_out[r];
        const uint32_t idx = 0xFFFFFFFFu - (uint32_t)key;
        float4 box;
        float score;
        if (p.src.mode == 2) {
            box = p.src.boxes[idx];
            score = p.src.scores[idx];
        } else {
            const float *row = p.src.y_pred + ((long long)img * p.n_per_image + idx) * 6;
            score = p.src.logits ? softmax_face(row[0], row[1]) : row[1];
            const float2 lxy = *reinterpret_cast<const float2 *>(row + 2);
            const float2 lwh = *reinterpret_cast<const float2 *>(row + 4);
            const float4 anc = (p.src.mode == 0)
                                   ? geom_anchor(p.geom, p.src.cx_tab, p.src.cy_tab, (int)idx)
                                   : p.src.anchors[idx];
            box = decode_one(anc, make_float4(lxy.x, lxy.y, lwh.x, lwh.y), 0.1f, 0.2f);
        }
        const long long o = (long long)img * kSortCap + r;
        p.sboxes[o] = box;
        p.sscores[o] = score;
        p.sidx[o] = (int)idx;
    }
    SSD_TRACE_END(p.tr);
#ifdef SSD_EXP_STAMPS
    __builtin_amdgcn_s_waitcnt(0);
    SSD_STAMP();
    if (img == 0 && t == 0)
        printf("topk img0 n=%d K=%d m_lds=%d  count %.2f  stage+select %.2f  extract %.2f  sort %.2f  gather+store %.2f us\n",
               n, K, m_lds, (st[1] - st[0]) * 0.01, (st[2] - st[1]) * 0.01, (st[3] - st[2]) * 0.01,
               (st[4] - st[3]) * 0.01, (st[5] - st[4]) * 0.01);
#endif
#undef SSD_STAMP
}

// ---------------------------------------------------------------------------- NMS mask
// clip_by_value(hw, 0, max) then product, reference utils/box_utils.py:15-16
SSD_DEV float box_area(float4 b, float max_size)
{
    return clip0(b.z - b.x, max_size) * clip0(b.w - b.y, max_size);
}

// mask layout: (image, word of earlier boxes, later box j) -> 64 bits, bit i set when box
// 64*word + i (ranked before j) suppresses j.
//
// One wave per 64 x 64 tile of (earlier, later) pairs.  Few pairs of a tile overlap at all,
// and a pair that does not overlap has IoU +0: with a threshold >= 0 and boxes of positive
// area it suppresses nothing.  So the tile is first swept for overlapping pairs only (four
// min/max and two compares per pair), the overlapping pairs are listed in LDS, and the IoU
// with its division is taken for 64 listed pairs at a time.  Tiles with a NaN coordinate, a
// box of zero area, a negative threshold or more listed pairs than the list holds take the
// plain pair-by-pair loop.
constexpr int kPairCap = 1536;     // listed pairs per tile on the fast path

__global__ __launch_bounds__(64) void nms_mask_kernel(const float4 *__restrict__ sboxes,
                                                      const int *__restrict__ keff, float thr,
                                                      u64 *__restrict__ mask SSD_TRACE_PARAM)
{
    SSD_TRACE_BEGIN(tr);
    __shared__ float4 s_jbox[64];
    __shared__ float s_jarea[64];
    __shared__ float4 s_ibox[64];
    __shared__ float s_iarea[64];
    __shared__ unsigned short s_pair[kPairCap];
    __shared__ u64 s_res[64];
    const int img = blockIdx.y;
    int rb = 0;                        // band of later boxes; word of earlier boxes cw <= rb
    while ((rb + 1) * (rb + 2) / 2 <= (int)blockIdx.x) ++rb;
    const int cw = (int)blockIdx.x - rb * (rb + 1) / 2;
    const int lane = threadIdx.x;
    const float4 *boxes = sboxes + (long long)img * kLimit;
    const int i = cw * 64 + lane;
    // one trip to memory: the boxes do not wait for K (ranks past K hold a box of their own)
    const float4 ib = boxes[i];
    const float4 jl = boxes[rb * 64 + lane];
    const int K = keff[img];
    if (rb * 64 >= K) return;
    const float iarea = box_area(ib, kNmsMaxSize);
    const float jarea = box_area(jl, kNmsMaxSize);
    s_jbox[lane] = jl;
    s_jarea[lane] = jarea;
    s_ibox[lane] = ib;
    s_iarea[lane] = iarea;
    s_res[lane] = 0ull;
    __syncthreads();
    const bool zero_suppresses = !(0.0f <= thr);
    const int jn = min(64, K - rb * 64);

    // finite coordinates and positive areas on both sides, threshold >= 0
    const bool plain = (ib.x - ib.x == 0.0f) && (ib.y - ib.y == 0.0f) && (ib.z - ib.z == 0.0f) &&
                       (ib.w - ib.w == 0.0f) && (jl.x - jl.x == 0.0f) && (jl.y - jl.y == 0.0f) &&
                       (jl.z - jl.z == 0.0f) && (jl.w - jl.w == 0.0f) && iarea > 0.0f && jarea > 0.0f;
    if (!zero_suppresses && __ballot(!plain) == 0ull) {
        const bool i_ok = i < K;
        // Lane j sweeps the earlier boxes for its later box.  With finite corners and positive
        // sizes on both sides, min(a.z, b.z) - max(a.x, b.x) > 0 is a.z > b.x && b.z > a.x.
        unsigned int cand_lo = 0u, cand_hi = 0u;       // earlier boxes overlapping later box `lane`
#pragma unroll
        for (int q = 0; q < 64; ++q) {
            const float4 a = s_ibox[q];                // broadcast read
            const bool ov = (a.z > jl.x) & (jl.z > a.x) & (a.w > jl.y) & (jl.w > a.y);   // no short circuit
            if (q < 32) cand_lo |= ov ? (1u << q) : 0u;
            else cand_hi |= ov ? (1u << (q - 32)) : 0u;
        }
        u64 cand = ((u64)cand_hi << 32) | cand_lo;
        cand &= __ballot(i_ok);
        // on the diagonal tile box q is itself lane q: bits at and above it are dropped
        if (cw == rb) cand &= (1ull << lane) - 1ull;
        if (lane >= jn) cand = 0ull;
        const int cnt = __popcll(cand);
        int incl = cnt;
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(incl, o);
            if (lane >= o) incl += v;
        }
        const int total = __builtin_amdgcn_readlane(incl, 63);
        if (total <= kPairCap) {
            int e = incl - cnt;
            u64 w = cand;
            while (w != 0ull) {
                const int pos = __builtin_ctzll(w);
                w &= w - 1ull;
                s_pair[e++] = (unsigned short)((lane << 6) | pos);
            }
            __syncthreads();
            for (int d = lane; d < total; d += 64) {
                const int pr = s_pair[d];
                const int j = pr >> 6, ii = pr & 63;
                const float4 a = s_ibox[ii], b = s_jbox[j];
                // compute_iou(boxes_a = earlier box, boxes_b = later box), utils/box_utils.py:35-42
                const float iw = vmin(a.z, b.z) - vmax(a.x, b.x);
                const float ih = vmin(a.w, b.w) - vmax(a.y, b.y);
                const float asum = s_iarea[ii] + s_jarea[j];
                const float inter = vmin(iw, kNmsMaxSize) * vmin(ih, kNmsMaxSize);
                const float iou = inter / (asum - inter);
                if (!(iou <= thr)) atomicOr(&s_res[j], 1ull << ii);
            }
            __syncthreads();
            if (lane < jn) mask[((long long)img * kWords + cw) * kLimit + rb * 64 + lane] = s_res[lane];
            SSD_TRACE_END(tr);
            return;
        }
    }

    u64 word = 0ull;                                   // lane q ends up holding the word of box q
#pragma unroll
    for (int q = 0; q < 64; ++q) {
        // (rows past jn repeat the last box; their words are never stored)
        const float4 jb = s_jbox[q];                   // broadcast read
        // compute_iou(boxes_a = earlier box, boxes_b = later box), utils/box_utils.py:35-42
        const float iw = vmin(ib.z, jb.z) - vmax(ib.x, jb.x);
        const float ih = vmin(ib.w, jb.w) - vmax(ib.y, jb.y);
        const float asum = iarea + s_jarea[q];
        bool sup;
        if (iw > 0.0f && ih > 0.0f) {
            const float inter = vmin(iw, kNmsMaxSize) * vmin(ih, kNmsMaxSize);
            const float iou = inter / (asum - inter);
            sup = !(iou <= thr);
        } else {
            // intersection 0: IoU is +0 over a positive union, NaN over a zero one
            sup = (asum > 0.0f) ? zero_suppresses : true;
        }
        // on the diagonal tile box q is itself lane q: bits at and above it are dropped
        const u64 w = __ballot(sup && (cw != rb || lane < q));
        if (lane == q) word = w;
    }
    if (lane < jn) mask[((long long)img * kWords + cw) * kLimit + rb * 64 + lane] = word;
    SSD_TRACE_END(tr);
}

// ---------------------------------------------------------------------------- NMS scan
__global__ __launch_bounds__(kLimit) void nms_scan_kernel(
    const u64 *__restrict__ mask, const float4 *__restrict__ sboxes,
    const float *__restrict__ sscores, const int *__restrict__ sidx, const int *__restrict__ keff,
    int limit, const float *__restrict__ img_wh, float *__restrict__ dets,
    int *__restrict__ keep_idx, int *__restrict__ count SSD_TRACE_PARAM)
{
    SSD_TRACE_BEGIN(tr);
    __shared__ u64 s_kept[kWords];
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int w_me = t >> 6;
    const int img = blockIdx.x;
    // One trip to memory: the mask rows, the box, the score and the index of rank t are all
    // requested before K arrives (ranks past K read what an earlier call left or the top-k
    // kernel's padding, and drop it below).
    u64 row[kWords];               // row[c]: earlier boxes of word c that suppress box t
#pragma unroll
    for (int c = 0; c < kWords; ++c)
        row[c] = (c <= w_me) ? mask[((long long)img * kWords + c) * kLimit + t] : 0ull;
    const long long o_me = (long long)img * kLimit + t;
    const float4 b_me = sboxes[o_me];
    const float score_me = sscores[o_me];
    const int idx_me = sidx[o_me];
    const int K = keff[img];
    const int KW = (K + 63) >> 6;
    if (t >= K) {
#pragma unroll
        for (int c = 0; c < kWords; ++c) row[c] = 0ull;
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
                const u64 D = row[w];
                u64 U = __ballot(!removed);
                u64 kept = 0ull;
                while (U != 0ull) {
                    const bool mine = (U >> lane) & 1ull;
                    const u64 A = __ballot(mine && (D & U) == 0ull);
                    kept |= A;
                    const u64 S = __ballot(mine && (D & A) != 0ull);
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
        keep_idx[(long long)img * limit + pos] = idx_me;
        if (dets != nullptr) {
            // clip to [0,1], scale to the original image, [class, score, box] rows
            // (reference inference.py:73-83)
            const float4 b = b_me;
            const float W = img_wh ? img_wh[2 * img] : 1.0f;
            const float H = img_wh ? img_wh[2 * img + 1] : 1.0f;
            float2 *d = reinterpret_cast<float2 *>(dets + ((long long)img * limit + pos) * 6);
            d[0] = make_float2(1.0f, score_me);
            d[1] = make_float2(clip0(b.x, 1.0f) * W, clip0(b.y, 1.0f) * H);
            d[2] = make_float2(clip0(b.z, 1.0f) * W, clip0(b.w, 1.0f) * H);
        }
    }
    if (t >= total && t < limit) {
        keep_idx[(long long)img * limit + t] = -1;
        if (dets != nullptr) {
            float2 *d = reinterpret_cast<float2 *>(dets + ((long long)img * limit + t) * 6);
            d[0] = make_float2(0.f, 0.f);
            d[1] = make_float2(0.f, 0.f);
            d[2] = make_float2(0.f, 0.f);
        }
    }
    if (t == 0) count[img] = total;
    SSD_TRACE_END(tr);
}

// ------------------------------------------------- NMS, general path (more than 1024 ranks)
// compute_nms's loop as it is written (utils/box_utils.py:205-222): the first box not yet
// suppressed is kept and suppresses every later box whose IoU with it is not <= thr; one
// workgroup per image keeps the ranked boxes in LDS and walks the ranks in order.  No caller of
// the reference reaches this path (limit is 1000 at inference.py:60 and 200 by default): it
// exists so that any limit gives the reference's result, not for speed (one barrier per kept
// box).  More than kBigLimit ranks are taken kBigLimit at a time, one launch each, in rank order:
// a launch first lets the boxes the earlier launches kept (unclipped, in kept_box) suppress its
// own, which is what the loop would have done by the time it reaches them, then runs the loop on
// what is left and appends what it keeps.

// compute_iou(boxes_a = earlier box, boxes_b = later box) > thr, utils/box_utils.py:35-42, :216
SSD_DEV bool nms_suppresses(float4 ib, float iarea, float4 jb, float jarea, float thr, bool zero_suppresses)
{
    const float iw = vmin(ib.z, jb.z) - vmax(ib.x, jb.x);
    const float ih = vmin(ib.w, jb.w) - vmax(ib.y, jb.y);
    const float asum = iarea + jarea;
    if (iw > 0.0f && ih > 0.0f) {
        const float inter = vmin(iw, kNmsMaxSize) * vmin(ih, kNmsMaxSize);
        const float iou = inter / (asum - inter);
        return !(iou <= thr);
    }
    // intersection 0: IoU is +0 over a positive union, NaN over a zero one
    return (asum > 0.0f) ? zero_suppresses : true;
}

__global__ __launch_bounds__(kTopkBlock) void nms_greedy_big_kernel(
    const float4 *__restrict__ sboxes, const float *__restrict__ sscores, const int *__restrict__ sidx,
    const int *__restrict__ keff, float thr, int limit, int rank_offset, int last,
    float4 *__restrict__ kept_box, const float *__restrict__ img_wh, float *__restrict__ dets,
    int *__restrict__ keep_idx, int *__restrict__ count)
{
    __shared__ float4 s_box[kBigLimit];
    __shared__ float s_area[kBigLimit];
    __shared__ u64 s_removed[kBigLimit / 64];
    __shared__ u64 s_kept[kBigLimit / 64];
    __shared__ float4 s_prev[kTopkBlock];
    __shared__ float s_prev_area[kTopkBlock];
    constexpr int kPer = kBigLimit / kTopkBlock;
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int img = blockIdx.x;
    const int K = keff[img];
    const int prev = rank_offset > 0 ? count[img] : 0;     // kept by the launches before this one
    const long long base = (long long)img * kBigLimit;
    for (int r = t; r < K; r += kTopkBlock) {
        const float4 b = sboxes[base + r];
        s_box[r] = b;
        s_area[r] = box_area(b, kNmsMaxSize);
    }
    if (t < kBigLimit / 64) {
        s_removed[t] = 0ull;
        s_kept[t] = 0ull;
    }
    __syncthreads();
    const bool zero_suppresses = !(0.0f <= thr);

    // ---- boxes kept from earlier ranks come first in the loop: they suppress before any box here
    for (int i0 = 0; i0 < prev; i0 += kTopkBlock) {
        const int ni = min(kTopkBlock, prev - i0);
        if (t < ni) {
            const float4 b = kept_box[(long long)img * limit + i0 + t];
            s_prev[t] = b;
            s_prev_area[t] = box_area(b, kNmsMaxSize);
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < kPer; ++q) {
            const int j = q * kTopkBlock + t;
            bool sup = false;
            if (j < K) {
                const float4 jb = s_box[j];
                const float jarea = s_area[j];
                for (int i = 0; i < ni && !sup; ++i) sup = nms_suppresses(s_prev[i], s_prev_area[i], jb, jarea, thr, zero_suppresses);
            }
            const u64 bal = __ballot(sup);
            if (bal != 0ull && lane == 0) atomicOr(&s_removed[j >> 6], bal);        // j - lane is a multiple of 64
        }
        __syncthreads();
    }

    for (int i = 0; i < K; ++i) {
        // (uniform: every thread reads the same word, written before the last barrier)
        if ((s_removed[i >> 6] >> (i & 63)) & 1ull) continue;
        if (t == 0) s_kept[i >> 6] |= 1ull << (i & 63);
        const float4 ib = s_box[i];
        const float iarea = s_area[i];
        for (int j0 = i + 1; j0 < K; j0 += kTopkBlock) {
            const int j = j0 + t;
            const bool sup = j < K && nms_suppresses(ib, iarea, s_box[j], s_area[j], thr, zero_suppresses);
            // a wave's 64 boxes j0 + 64 w .. lie in at most two words of the bit set
            const u64 bal = __ballot(sup);
            if (bal != 0ull && lane == 0) {
                const int first = j0 + (t & ~63);
                const int sh = first & 63;
                atomicOr(&s_removed[first >> 6], bal << sh);
                if (sh != 0 && (first >> 6) + 1 < kBigLimit / 64) atomicOr(&s_removed[(first >> 6) + 1], bal >> (64 - sh));
            }
        }
        __syncthreads();
    }
    __syncthreads();

    // kept boxes in rank order behind the earlier launches'; after the last launch the rest of
    // the `limit` slots cleared (as the scan kernel does)
    int total = prev;
    for (int w = 0; w < kBigLimit / 64; ++w) total += __popcll(s_kept[w]);
    const float W = img_wh ? img_wh[2 * img] : 1.0f;
    const float H = img_wh ? img_wh[2 * img + 1] : 1.0f;
    for (int r = t; r < K; r += kTopkBlock) {
        if ((s_kept[r >> 6] >> (r & 63)) & 1ull) {
            int pos = prev + __popcll(s_kept[r >> 6] & ((1ull << (r & 63)) - 1ull));
            for (int w = 0; w < (r >> 6); ++w) pos += __popcll(s_kept[w]);
            const long long o = (long long)img * limit + pos;
            keep_idx[o] = sidx[base + r];
            const float4 b = s_box[r];
            if (!last) kept_box[o] = b;
            if (dets != nullptr) {
                // clip to [0,1], scale to the original image, [class, score, box] rows
                // (reference inference.py:73-83)
                float2 *d = reinterpret_cast<float2 *>(dets + o * 6);
                d[0] = make_float2(1.0f, sscores[base + r]);
                d[1] = make_float2(clip0(b.x, 1.0f) * W, clip0(b.y, 1.0f) * H);
                d[2] = make_float2(clip0(b.z, 1.0f) * W, clip0(b.w, 1.0f) * H);
            }
        }
    }
    if (last) {
        for (int r = total + t; r < limit; r += kTopkBlock) {
            keep_idx[(long long)img * limit + r] = -1;
            if (dets != nullptr) {
                float2 *d = reinterpret_cast<float2 *>(dets + ((long long)img * limit + r) * 6);
                d[0] = make_float2(0.f, 0.f);
                d[1] = make_float2(0.f, 0.f);
                d[2] = make_float2(0.f, 0.f);
            }
        }
    }
    if (t == 0) count[img] = total;
}

// -------------------------------------------------------------------------------- host
int ensure_detect_scratch(ssdbox_ctx *ctx, int batch, long long nseg, int rank_cap)
{
    int rc = ensure_buffer((void **)&ctx->cand, &ctx->cand_cap, (size_t)batch * (size_t)nseg * kSeg,
                           sizeof(u64), false, ctx->stream);
    if (rc != SSDBOX_OK) return rc;
    rc = ensure_buffer((void **)&ctx->cand_count, &ctx->count_cap, (size_t)batch * (size_t)nseg,
                       sizeof(int), false, ctx->stream);
    if (rc != SSDBOX_OK) return rc;
    rc = ensure_buffer((void **)&ctx->cand_hist, &ctx->hist_cap, (size_t)batch * kBins, sizeof(unsigned int),
                       true, ctx->stream);
    if (rc != SSDBOX_OK) return rc;
    // ranked boxes: (images, rank_cap) of whichever path runs; the mask of the fast path
    if ((size_t)batch > ctx->det_cap || (size_t)rank_cap > ctx->det_rank_cap) {
        SSD_CHECK_HIP(hipStreamSynchronize(ctx->stream));
        void *old[] = {ctx->sboxes, ctx->sscores, ctx->sidx, ctx->keff, ctx->mask};
        for (void *p : old)
            if (p) SSD_CHECK_HIP(hipFree(p));
        ctx->sboxes = nullptr;
        ctx->sscores = nullptr;
        ctx->sidx = nullptr;
        ctx->keff = nullptr;
        ctx->mask = nullptr;
        const size_t cap = std::max((size_t)batch + (size_t)batch / 4, ctx->det_cap);
        const size_t ranks = std::max((size_t)rank_cap, ctx->det_rank_cap);
        ctx->det_cap = 0;
        ctx->det_rank_cap = 0;
        hipError_t e = hipMalloc((void **)&ctx->sboxes, cap * ranks * sizeof(float4));
        if (e == hipSuccess) e = hipMalloc((void **)&ctx->sscores, cap * ranks * sizeof(float));
        if (e == hipSuccess) e = hipMalloc((void **)&ctx->sidx, cap * ranks * sizeof(int));
        if (e == hipSuccess) e = hipMalloc((void **)&ctx->keff, cap * sizeof(int));
        if (e == hipSuccess) e = hipMalloc((void **)&ctx->mask, cap * kLimit * kWords * sizeof(u64));
        if (e != hipSuccess) {
            set_error("detection scratch allocation failed: %s", hipGetErrorString(e));
            return SSDBOX_ERR_NOMEM;
        }
        ctx->det_cap = cap;
        ctx->det_rank_cap = ranks;
    }
    return SSDBOX_OK;
}

struct ScoreLayout {
    const float *scores;
    long long img_stride;
    int elem_stride, elem_off;
    int use_thr;
    float thr;
    bool rows6;                    // (batch, n, 6) head rows, 16-byte aligned per image
    bool logits;                   // scores are the softmax of elements elem_off - 1 and elem_off
};

int run_detect(ssdbox_ctx *ctx, const ScoreLayout &sl, long long n, int batch, float nms_thr,
               int limit, const BoxSource &src, const GeomDev *host_geom, const float *d_img_wh,
               float *d_dets, int32_t *d_keep, int32_t *d_count)
{
    const int nseg = (int)((n + kSeg - 1) / kSeg);
    const bool big = limit > kLimit;
    int rc = ensure_detect_scratch(ctx, batch, nseg, big ? kBigLimit : kLimit);
    if (rc != SSDBOX_OK) return rc;
    // ranks beyond kBigLimit: one top-k + NMS launch pair per kBigLimit ranks (no more ranks than
    // scores exist), the kept boxes and the last ranked key carried from pair to pair
    const long long ranks_total = std::min<long long>(limit, n);
    const int n_chunks = big ? (int)((ranks_total + kBigLimit - 1) / kBigLimit) : 1;
    if (n_chunks > 1) {
        rc = ensure_buffer((void **)&ctx->kept_box, &ctx->kept_cap, (size_t)batch * (size_t)limit, sizeof(float4),
                           false, ctx->stream);
        if (rc != SSDBOX_OK) return rc;
        rc = ensure_buffer((void **)&ctx->floor_key, &ctx->floor_cap, (size_t)batch, sizeof(u64), false, ctx->stream);
        if (rc != SSDBOX_OK) return rc;
    }
    // head scores are softmax outputs: spread the linear bins over (threshold, 1]
    const int has_range = (sl.use_thr && sl.thr >= 0.0f && sl.thr < 1.0f) ? 1 : 0;
    const float range_lo = has_range ? sl.thr : 0.0f;
    const float range_scale = has_range ? (float)(kBins - 1) / (1.0f - sl.thr) : 0.0f;
    // Most anchors are candidates when there is no threshold or one below the usual detection
    // thresholds: prune before the one-workgroup-per-image kernel (a guess about speed only,
    // both ways give the same result).  The per-image histogram the pruning cuts by is all zero
    // between calls: the top-k kernel clears what this call counted.
    const bool dense = n > kCap && (!sl.use_thr || sl.thr < 0.02f);
    if (dense) {
        if (ctx->hist_dirty)
            SSD_CHECK_HIP(hipMemsetAsync(ctx->cand_hist, 0, ctx->hist_cap * sizeof(unsigned int), ctx->stream));
        ctx->hist_dirty = true;        // until every launch of this call is enqueued
    }
    const dim3 hist_grid((nseg + kHistSegs - 1) / kHistSegs, batch);
    bool counted = false;              // the select kernel filled the histogram itself
    {
        dim3 grid((unsigned)((nseg + kSelBlock / 64 - 1) / (kSelBlock / 64)), batch);
        const long long n_items = (long long)nseg * batch;
        // (the item -> image division by multiply-high is exact while items * segments < 2^32)
        const bool rows6 = sl.rows6 && n_items < (1ll << 31) / kSeg && n_items * nseg < (1ll << 32);
        const unsigned pgrid = (unsigned)std::min<long long>((n_items + kSelBlock / 64 - 1) / (kSelBlock / 64),
                                                             (long long)ctx->num_cus * kSelWgsPerCu);
        const unsigned nseg_magic = nseg > 1 ? (unsigned)(((1ull << 32) + (unsigned)nseg - 1) / (unsigned)nseg) : 0u;
        const float4 *rows = reinterpret_cast<const float4 *>(sl.scores);
        if (rows6 && dense) {
            counted = true;
            if (sl.logits)
                launch_stage(ctx, SSDBOX_STAGE_SELECT, select_rows6_hist_kernel<true>, hist_grid, dim3(kSelBlock), rows,
                             (int)n, sl.thr, ctx->cand, ctx->cand_count, nseg, has_range, range_lo, range_scale,
                             ctx->cand_hist);
            else
                launch_stage(ctx, SSDBOX_STAGE_SELECT, select_rows6_hist_kernel<false>, hist_grid, dim3(kSelBlock), rows,
                             (int)n, sl.thr, ctx->cand, ctx->cand_count, nseg, has_range, range_lo, range_scale,
                             ctx->cand_hist);
        } else if (rows6 && sl.logits) {
            launch_stage(ctx, SSDBOX_STAGE_SELECT, select_rows6_kernel<true>, dim3(pgrid), dim3(kSelBlock), rows,
                         (int)n, sl.thr, ctx->cand, ctx->cand_count, nseg, (int)n_items,
                         nseg_magic SSD_TRACE_ARG(ctx, SSDBOX_STAGE_SELECT));
        } else if (rows6) {
            launch_stage(ctx, SSDBOX_STAGE_SELECT, select_rows6_kernel<false>, dim3(pgrid), dim3(kSelBlock), rows,
                         (int)n, sl.thr, ctx->cand, ctx->cand_count, nseg, (int)n_items,
                         nseg_magic SSD_TRACE_ARG(ctx, SSDBOX_STAGE_SELECT));
        } else {
            launch_stage(ctx, SSDBOX_STAGE_SELECT, select_strided_kernel, grid, dim3(kSelBlock),
                         sl.scores, sl.img_stride, sl.elem_stride, sl.elem_off, (int)n, sl.thr,
                         sl.use_thr, sl.logits ? 1 : 0, ctx->cand, ctx->cand_count, nseg);
        }
    }
    SSD_CHECK_HIP(hipGetLastError());
    if (dense) {
        if (!counted)
            launch_stage(ctx, SSDBOX_STAGE_PRUNE, cand_hist_kernel, hist_grid, dim3(kSelBlock), ctx->cand,
                         ctx->cand_count, nseg, has_range, range_lo, range_scale, ctx->cand_hist);
        launch_stage(ctx, SSDBOX_STAGE_PRUNE, cand_prune_kernel,
                     dim3((nseg + kSelBlock / 64 - 1) / (kSelBlock / 64), batch), dim3(kSelBlock), ctx->cand,
                     ctx->cand_count, nseg, limit, has_range, range_lo, range_scale, ctx->cand_hist);
        SSD_CHECK_HIP(hipGetLastError());
    }
    {
        TopkArgs a;
        memset(&a, 0, sizeof(a));
        if (host_geom != nullptr) a.geom = *host_geom;
        a.cand = ctx->cand;
        a.seg_count = ctx->cand_count;
        a.nseg = nseg;
        a.n_per_image = (int)n;
        a.limit = limit;
        a.has_range = has_range;
        a.lo = range_lo;
        a.scale = range_scale;
        a.src = src;
        a.sboxes = ctx->sboxes;
        a.sscores = ctx->sscores;
        a.sidx = ctx->sidx;
        a.keff = ctx->keff;
        a.hist_clear = dense ? ctx->cand_hist : nullptr;
#ifdef SSD_EXP_TRACE
        a.tr = ssd_trace_slot(ctx, SSDBOX_STAGE_TOPK);
#endif
        if (big) {
            a.floor_key = n_chunks > 1 ? ctx->floor_key : nullptr;
            for (int c = 0; c < n_chunks; ++c) {
                a.rank_offset = c * kBigLimit;
                a.limit = (int)std::min<long long>(kBigLimit, ranks_total - a.rank_offset);
                launch_stage(ctx, SSDBOX_STAGE_TOPK, topk_sort_kernel<kBigRanks>, dim3(batch), dim3(kTopkBlock), a);
                a.hist_clear = nullptr;
                launch_stage(ctx, SSDBOX_STAGE_NMS_SCAN, nms_greedy_big_kernel, dim3(batch), dim3(kTopkBlock),
                             ctx->sboxes, ctx->sscores, ctx->sidx, ctx->keff, nms_thr, limit, a.rank_offset,
                             c == n_chunks - 1 ? 1 : 0, n_chunks > 1 ? ctx->kept_box : nullptr, d_img_wh, d_dets,
                             d_keep, d_count);
                SSD_CHECK_HIP(hipGetLastError());
            }
        } else {
            launch_stage(ctx, SSDBOX_STAGE_TOPK, topk_sort_kernel<1>, dim3(batch), dim3(kTopkBlock), a);
        }
    }
    SSD_CHECK_HIP(hipGetLastError());
    if (!big) {
        const int kw = (limit + 63) / 64;
        launch_stage(ctx, SSDBOX_STAGE_NMS_MASK, nms_mask_kernel, dim3(kw * (kw + 1) / 2, batch), dim3(64),
                     ctx->sboxes, ctx->keff, nms_thr, ctx->mask SSD_TRACE_ARG(ctx, SSDBOX_STAGE_NMS_MASK));
        SSD_CHECK_HIP(hipGetLastError());
        launch_stage(ctx, SSDBOX_STAGE_NMS_SCAN, nms_scan_kernel, dim3(batch), dim3(kLimit), ctx->mask,
                     ctx->sboxes, ctx->sscores, ctx->sidx, ctx->keff, limit, d_img_wh, d_dets,
                     d_keep, d_count SSD_TRACE_ARG(ctx, SSDBOX_STAGE_NMS_SCAN));
    }
    SSD_CHECK_HIP(hipGetLastError());
    if (dense) ctx->hist_dirty = false;
    return SSDBOX_OK;
}

}  // namespace

static int decode_nms_common(ssdbox_ctx *ctx, const ssdbox_geom *geom, const float *d_anchors,
                             int64_t num_anchors, const float *d_y_pred, int batch,
                             float score_threshold, float nms_threshold, int limit,
                             const float *d_img_wh, float *d_dets, int32_t *d_keep_idx,
                             int32_t *d_count, bool logits)
{
    SSD_REQUIRE(ctx != nullptr, "decode_nms: ctx is NULL");
    SSD_REQUIRE((geom != nullptr) != (d_anchors != nullptr),
                "decode_nms: pass exactly one of geom and d_anchors");
    SSD_REQUIRE(d_y_pred && d_dets && d_keep_idx && d_count, "decode_nms: NULL buffer");
    SSD_REQUIRE(batch >= 1 && batch <= 65535, "decode_nms: batch %d outside [1, 65535]", batch);
    SSD_REQUIRE(limit >= 1, "decode_nms: limit %d < 1", limit);
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
    src.logits = logits ? 1 : 0;
    ScoreLayout sl;
    sl.logits = logits;
    sl.scores = d_y_pred;
    sl.img_stride = (long long)num_anchors * 6;
    sl.elem_stride = 6;
    sl.elem_off = 1;
    sl.use_thr = 1;
    sl.thr = score_threshold;
    sl.rows6 = (num_anchors % 2 == 0) && (reinterpret_cast<uintptr_t>(d_y_pred) % 16 == 0);
    return run_detect(ctx, sl, num_anchors, batch, nms_threshold, limit, src,
                      geom ? &geom->host : nullptr, d_img_wh, d_dets,
                      d_keep_idx, d_count);
}

extern "C" int ssdbox_decode_nms(ssdbox_ctx *ctx, const ssdbox_geom *geom, const float *d_anchors,
                                 int64_t num_anchors, const float *d_y_pred, int batch,
                                 float score_threshold, float nms_threshold, int limit,
                                 const float *d_img_wh, float *d_dets, int32_t *d_keep_idx,
                                 int32_t *d_count)
{
    return decode_nms_common(ctx, geom, d_anchors, num_anchors, d_y_pred, batch, score_threshold,
                             nms_threshold, limit, d_img_wh, d_dets, d_keep_idx, d_count, false);
}

extern "C" int ssdbox_decode_nms_logits(ssdbox_ctx *ctx, const ssdbox_geom *geom, const float *d_anchors,
                                        int64_t num_anchors, const float *d_y_logits, int batch,
                                        float score_threshold, float nms_threshold, int limit,
                                        const float *d_img_wh, float *d_dets, int32_t *d_keep_idx,
                                        int32_t *d_count)
{
    return decode_nms_common(ctx, geom, d_anchors, num_anchors, d_y_logits, batch, score_threshold,
                             nms_threshold, limit, d_img_wh, d_dets, d_keep_idx, d_count, true);
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
    BoxSource src;
    src.mode = 2;
    src.geom = nullptr;
    src.cx_tab = nullptr;
    src.cy_tab = nullptr;
    src.anchors = nullptr;
    src.y_pred = nullptr;
    src.boxes = reinterpret_cast<const float4 *>(d_boxes);
    src.scores = d_scores;
    src.logits = 0;
    ScoreLayout sl;
    sl.logits = false;
    sl.scores = d_scores;
    sl.img_stride = 0;
    sl.elem_stride = 1;
    sl.elem_off = 0;
    sl.use_thr = 0;
    sl.thr = 0.0f;
    sl.rows6 = false;
    return run_detect(ctx, sl, n, 1, nms_threshold, k, src, nullptr, nullptr, nullptr, d_keep,
                      d_count);
}
