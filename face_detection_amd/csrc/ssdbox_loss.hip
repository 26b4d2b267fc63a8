// SSD loss with hard-negative mining for a batch (reference utils/losses.py:97-181), forward
// value per image, the mined-negative mask and the gradient of the batch mean with respect to
// y_pred (what TensorFlow's autodiff derives: the mask and the counts are constants).
// Rows are [background, face, 4 offsets] for y_true and y_pred alike.
//
// The top-k of the reference runs over all batch * anchors negative log-losses at once
// (utils/losses.py:149-153).  Here it is a radix select on the 32-bit rank key of the value
// (larger key = larger value, NaN last, -0 = +0): three histogram passes (11 + 11 + 10 bits)
// find the k-th largest key T; keys above T are kept, keys equal to T in ascending index order
// until k is reached (tf.nn.top_k ranks equal values by index), which takes an ordered count
// of the keys equal to T per workgroup chunk and a scan over the chunks.
//
//   pass1   per anchor: log-loss, smooth-L1, key -> scratch; per-image sums (double atomics);
//           positives and nonzero negative losses counted; histogram of the top 11 key bits
//   hist    (twice) k = min(max(ratio * n_pos, n_neg_min), n_nonzero) and the digit where the
//           count from the top of the previous histogram crosses what is still to be found,
//           recomputed by every workgroup (2048 bins) instead of a launch of its own; then the
//           histogram of the next digit among the keys that share the prefix
//   count   the last digit -> T, how many keys equal to T to keep; those keys per chunk and
//           per image
//   final   mask, mined-negative sums, gradient
//   finish  (pos + neg + alpha * loc) / max(1, n_pos) * batch per image
#include "ssdbox_internal.h"

namespace {

typedef unsigned long long u64;

constexpr int kLB = 256;               // threads per workgroup
constexpr int kPerThread = 10;         // anchors per thread and chunk
constexpr int kBins = 2048;

// Every field is written by one kernel (its first workgroup) and read by later ones only.
struct LossState {
    double n_pos;                      // sum of the positives column (utils/losses.py:112)
    u64 n_nonzero;                     // nonzero negative log-losses (:123)
    long long k;                       // negatives to keep (:135-136)
    long long k_rem[2];                // of which still to find under prefix[0] / prefix[1]
    unsigned prefix[2];                // key bits fixed after the first / second digit
    unsigned threshold;                // key of the k-th largest value
    unsigned need_eq, count_eq;        // keys equal to it: to keep / present
};

struct LossArgs {
    const float *y_true;
    const float *y_pred;
    int batch;
    int num_anchors;
    int blocks_per_image;
    int chunk;                         // anchors per workgroup, a multiple of kLB
    int neg_pos_ratio, n_neg_min;
    float alpha;
    unsigned *key;                     // (batch * anchors) rank keys of the negative log-losses
    unsigned *hist;                    // (3, kBins)
    LossState *state;
    double *sums;                      // (batch, 4): positive log-loss, smooth-L1, mined log-loss
    unsigned *chunk_eq;                // (batch * blocks_per_image) keys equal to the threshold
    unsigned *img_eq;                  // (batch) the same per image
    float *total;                      // (batch)
    unsigned char *keep;               // (batch, anchors) or NULL
    float *grad;                       // (batch, anchors, 6) or NULL
};

struct Row {
    float2 c, xy, wh;
};

SSD_DEV Row load_row(const float *base, long long i)
{
    const float2 *p = reinterpret_cast<const float2 *>(base + i * 6);
    Row r;
    r.c = p[0];
    r.xy = p[1];
    r.wh = p[2];
    return r;
}

// Two consecutive rows (48 bytes, 16-byte aligned when the first index is even and rows per
// image are even) as three 16-byte loads.
SSD_DEV void load_row_pair(const float *base, long long i, Row &a, Row &b)
{
    const float4 *p = reinterpret_cast<const float4 *>(base + i * 6);
    const float4 v0 = p[0], v1 = p[1], v2 = p[2];
    a.c = make_float2(v0.x, v0.y);
    a.xy = make_float2(v0.z, v0.w);
    a.wh = make_float2(v1.x, v1.y);
    b.c = make_float2(v1.z, v1.w);
    b.xy = make_float2(v2.x, v2.y);
    b.wh = make_float2(v2.z, v2.w);
}

// utils/losses.py:78-81
SSD_DEV float class_loss(const Row &t, const Row &p)
{
    const float l0 = ssd_logf(fmaxf(p.c.x, 1e-15f)), l1 = ssd_logf(fmaxf(p.c.y, 1e-15f));
    return -(t.c.x * l0 + t.c.y * l1);
}

SSD_DEV float smooth_l1_term(float t, float p)
{
    const float d = t - p;
    const float a = fabsf(d);
    return a < 1.0f ? 0.5f * (d * d) : a - 0.5f;
}

// utils/losses.py:58-61
SSD_DEV float loc_loss(const Row &t, const Row &p)
{
    float s = 0.0f;
    s = s + smooth_l1_term(t.xy.x, p.xy.x);
    s = s + smooth_l1_term(t.xy.y, p.xy.y);
    s = s + smooth_l1_term(t.wh.x, p.wh.x);
    s = s + smooth_l1_term(t.wh.y, p.wh.y);
    return s;
}

SSD_DEV double block_sum(double v, double *s_red)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    double r = 0.0;
#pragma unroll
    for (int w = 0; w < kLB / 64; ++w) r += s_red[w];
    return r;
}

SSD_DEV void flush_hist(const unsigned *s_hist, unsigned *g_hist)
{
    for (int b = threadIdx.x; b < kBins; b += kLB) {
        const unsigned c = s_hist[b];
        if (c != 0u) atomicAdd(&g_hist[b], c);
    }
}

// grid (blocks_per_image, batch); PAIR: a thread takes two consecutive anchors at a time
template <bool PAIR>
__global__ __launch_bounds__(kLB) void loss_pass1_kernel(const LossArgs p)
{
    constexpr int E = PAIR ? 2 : 1;
    __shared__ unsigned s_hist[kBins];
    __shared__ double s_red[kLB / 64];
    const int t = threadIdx.x;
    const int img = blockIdx.y;
    for (int b = t; b < kBins; b += kLB) s_hist[b] = 0u;
    __syncthreads();
    const int a0 = blockIdx.x * p.chunk;
    const int a1 = min(a0 + p.chunk, p.num_anchors);
    double pos_cls = 0.0, loc_s = 0.0, n_pos = 0.0;
    unsigned nonzero = 0;
    // The rows of the next round are requested before this round's keys are stored: loads and
    // stores of a wave complete in issue order, so a load behind the store would also wait for
    // the store to reach memory.
    auto load_rows = [&](int a, Row *yt, Row *yp) {
        const long long i = (long long)img * p.num_anchors + min(a, a1 - E);
        if (PAIR) {
            load_row_pair(p.y_true, i, yt[0], yt[E - 1]);
            load_row_pair(p.y_pred, i, yp[0], yp[E - 1]);
        } else {
            yt[0] = load_row(p.y_true, i);
            yp[0] = load_row(p.y_pred, i);
        }
    };
    Row yt[E], yp[E];
    load_rows(a0 + E * t, yt, yp);
    for (int a = a0 + E * t; a < a1; a += E * kLB) {
        const long long i = (long long)img * p.num_anchors + a;
        Row yt_next[E], yp_next[E];
        load_rows(a + E * kLB, yt_next, yp_next);          // clamped into the chunk past its end
        unsigned key[E];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const float cls = class_loss(yt[e], yp[e]);
            const float loc = loc_loss(yt[e], yp[e]);
            const float pos = yt[e].c.y;                   // :109 (one class besides background)
            pos_cls += (double)(cls * pos);                // :117
            loc_s += (double)(loc * pos);                  // :170
            n_pos += (double)pos;                          // :112
            const float neg_all = cls * yt[e].c.x;         // :122
            nonzero += (neg_all != 0.0f) ? 1u : 0u;        // :123 (NaN counts, -0 does not)
            key[e] = score_key(neg_all);
            atomicAdd(&s_hist[key[e] >> 21], 1u);
        }
        if (PAIR)
            *reinterpret_cast<uint2 *>(p.key + i) = make_uint2(key[0], key[E - 1]);
        else
            p.key[i] = key[0];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            yt[e] = yt_next[e];
            yp[e] = yp_next[e];
        }
    }
    pos_cls = block_sum(pos_cls, s_red);
    loc_s = block_sum(loc_s, s_red);
    n_pos = block_sum(n_pos, s_red);
    const double nz = block_sum((double)nonzero, s_red);
    if (t == 0) {
        if (pos_cls != 0.0) atomicAdd(&p.sums[4 * img + 0], pos_cls);
        if (loc_s != 0.0) atomicAdd(&p.sums[4 * img + 1], loc_s);
        if (n_pos != 0.0) atomicAdd(&p.state->n_pos, n_pos);
        if (nz != 0.0) atomicAdd(&p.state->n_nonzero, (u64)nz);
    }
    __syncthreads();
    flush_hist(s_hist, p.hist);
}

// The digit (bin) of a 2048-bin histogram where the count from the top crosses k_rem
// (1 <= k_rem <= total count), the count above it and in it.  Every thread of the workgroup
// calls it and gets the same answer.
struct Digit {
    unsigned digit, above, count;
};

SSD_DEV Digit find_digit(const unsigned *__restrict__ hist, long long k_rem, unsigned *s_wtot, unsigned *s_sel)
{
    static_assert(kBins == 8 * kLB, "eight bins per thread");
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const uint4 lo = reinterpret_cast<const uint4 *>(hist)[2 * t];
    const uint4 hi = reinterpret_cast<const uint4 *>(hist)[2 * t + 1];
    const unsigned h[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    unsigned sum = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) sum += h[j];
    unsigned incl = sum;               // suffix sum over this wave's higher lanes
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned v = __shfl_down(incl, o);
        if (lane + o < 64) incl += v;
    }
    if (lane == 0) s_wtot[wave] = incl;
    __syncthreads();
    unsigned above = incl - sum;       // keys in the bins above this thread's eight
    for (int w = wave + 1; w < kLB / 64; ++w) above += s_wtot[w];
#pragma unroll
    for (int j = 7; j >= 0; --j) {
        if ((long long)above < k_rem && (long long)above + h[j] >= k_rem) {
            s_sel[0] = 8u * t + j;
            s_sel[1] = above;
            s_sel[2] = h[j];
        }
        above += h[j];
    }
    __syncthreads();
    Digit d;
    d.digit = s_sel[0];
    d.above = s_sel[1];
    d.count = s_sel[2];
    __syncthreads();
    return d;
}

// grid (blocks_per_image, batch): the digit of pass PASS - 1, then the histogram of digit PASS
// among the keys under the prefix
template <int PASS>
__global__ __launch_bounds__(kLB) void loss_hist_kernel(const LossArgs p)
{
    __shared__ unsigned s_hist[kBins];
    __shared__ unsigned s_wtot[kLB / 64];
    __shared__ unsigned s_sel[3];
    const int t = threadIdx.x;
    const int img = blockIdx.y;
    const bool first = blockIdx.x == 0 && blockIdx.y == 0 && t == 0;
    LossState *st = p.state;
    long long k, k_rem;
    unsigned prefix = 0u;
    if (PASS == 1) {
        k = (long long)p.neg_pos_ratio * (long long)st->n_pos;                 // :135-136
        if (k < p.n_neg_min) k = p.n_neg_min;
        if (k > (long long)st->n_nonzero) k = (long long)st->n_nonzero;
        k_rem = k;
        if (first) st->k = k;
    } else {
        k = st->k;
        k_rem = st->k_rem[0];
        prefix = st->prefix[0];
    }
    if (k <= 0) return;                // nothing to keep (:163 and k == 0)
    const Digit d = find_digit(p.hist + (PASS - 1) * kBins, k_rem, s_wtot, s_sel);
    prefix |= d.digit << (PASS == 1 ? 21 : 10);
    const unsigned pmask = PASS == 1 ? 0xFFE00000u : 0xFFFFFC00u;
    if (first) {
        st->k_rem[PASS - 1] = k_rem - (long long)d.above;
        st->prefix[PASS - 1] = prefix;
    }
    for (int b = t; b < kBins; b += kLB) s_hist[b] = 0u;
    __syncthreads();
    const int shift = PASS == 1 ? 10 : 0;
    const unsigned dmask = PASS == 2 ? 0x3FFu : 0x7FFu;
    const int a0 = blockIdx.x * p.chunk;
    const int a1 = min(a0 + p.chunk, p.num_anchors);
    for (int a = a0 + t; a < a1; a += kLB) {
        const unsigned key = p.key[(long long)img * p.num_anchors + a];
        if ((key & pmask) == prefix) atomicAdd(&s_hist[(key >> shift) & dmask], 1u);
    }
    __syncthreads();
    flush_hist(s_hist, p.hist + PASS * kBins);
}

// grid (blocks_per_image, batch): the last digit, then the keys equal to the threshold in this
// chunk (only when not all of them are kept)
__global__ __launch_bounds__(kLB) void loss_count_eq_kernel(const LossArgs p)
{
    __shared__ double s_red[kLB / 64];
    __shared__ unsigned s_wtot[kLB / 64];
    __shared__ unsigned s_sel[3];
    const int t = threadIdx.x;
    const int img = blockIdx.y;
    const int blk = img * p.blocks_per_image + (int)blockIdx.x;
    const bool first = blockIdx.x == 0 && blockIdx.y == 0 && t == 0;
    LossState *st = p.state;
    if (st->k <= 0) {
        if (first) {
            st->threshold = 0xFFFFFFFFu;
            st->need_eq = 0u;
            st->count_eq = 0u;
        }
        return;
    }
    const long long k_rem = st->k_rem[1];
    const Digit d = find_digit(p.hist + 2 * kBins, k_rem, s_wtot, s_sel);
    const unsigned threshold = st->prefix[1] | d.digit;
    const unsigned need_eq = (unsigned)(k_rem - (long long)d.above);
    if (first) {
        st->threshold = threshold;
        st->need_eq = need_eq;
        st->count_eq = d.count;
    }
    if (need_eq >= d.count) return;    // every equal key is kept: no order needed
    unsigned n = 0;
    const int a0 = blockIdx.x * p.chunk;
    const int a1 = min(a0 + p.chunk, p.num_anchors);
    for (int a = a0 + t; a < a1; a += kLB)
        n += (p.key[(long long)img * p.num_anchors + a] == threshold) ? 1u : 0u;
    const unsigned tot = (unsigned)block_sum((double)n, s_red);
    if (t == 0 && tot != 0u) {
        p.chunk_eq[blk] = tot;
        atomicAdd(&p.img_eq[img], tot);
    }
}

SSD_DEV float smooth_l1_slope(float pv, float tv)
{
    const float d = pv - tv;
    return fabsf(d) < 1.0f ? d : (d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : d));
}

// grid (blocks_per_image, batch): mask, mined-negative sums, gradient
template <bool PAIR>
__global__ __launch_bounds__(kLB) void loss_final_kernel(const LossArgs p)
{
    constexpr int E = PAIR ? 2 : 1;
    __shared__ double s_red[kLB / 64];
    __shared__ unsigned s_weq[kLB / 64];
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int img = blockIdx.y;
    const LossState st = *p.state;
    const bool any = st.k > 0;
    const bool ordered = any && st.need_eq < st.count_eq;
    unsigned eq_before = 0u;           // equal keys in earlier images and earlier chunks of this one
    if (ordered) {
        unsigned part = 0u;
        for (int b = t; b < img; b += kLB) part += p.img_eq[b];
        for (int c = t; c < (int)blockIdx.x; c += kLB) part += p.chunk_eq[img * p.blocks_per_image + c];
        eq_before = (unsigned)block_sum((double)part, s_red);
    }
    const double denom = st.n_pos > 1.0 ? st.n_pos : 1.0;     // :174
    const float scale = (float)(1.0 / denom);
    const int a0 = blockIdx.x * p.chunk;
    const int a1 = min(a0 + p.chunk, p.num_anchors);
    double neg_cls = 0.0;
    for (int base = a0; base < a1; base += E * kLB) {        // tiles of consecutive anchors
        const int a = base + E * t;
        const bool in = a < a1;                              // PAIR: a1 - a0 is even
        const long long i = (long long)img * p.num_anchors + (in ? a : a1 - E);
        unsigned key[E];
        if (PAIR) {
            const uint2 kk = *reinterpret_cast<const uint2 *>(p.key + i);
            key[0] = kk.x;
            key[E - 1] = kk.y;
        } else {
            key[0] = p.key[i];
        }
        bool keep[E], eq[E];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            keep[e] = any && in && key[e] > st.threshold;
            eq[e] = any && in && key[e] == st.threshold;
        }
        if (ordered) {
            // rank of each key among the equal ones, in index order
            u64 bal[E];
            unsigned mine = 0;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                bal[e] = __ballot(eq[e]);
                mine += (unsigned)__popcll(bal[e]);
            }
            if (lane == 0) s_weq[wave] = mine;
            __syncthreads();
            unsigned rank = eq_before;
#pragma unroll
            for (int e = 0; e < E; ++e) rank += (unsigned)__popcll(bal[e] & ((1ull << lane) - 1ull));
            unsigned tile_total = 0;
#pragma unroll
            for (int w = 0; w < kLB / 64; ++w) {
                if (w < wave) rank += s_weq[w];
                tile_total += s_weq[w];
            }
#pragma unroll
            for (int e = 0; e < E; ++e) {
                keep[e] = keep[e] || (eq[e] && rank < st.need_eq);
                rank += eq[e] ? 1u : 0u;
            }
            eq_before += tile_total;
            __syncthreads();
        } else {
#pragma unroll
            for (int e = 0; e < E; ++e) keep[e] = keep[e] || eq[e];
        }
        if (in) {
            Row yt[E], yp[E];
            if (PAIR) {
                load_row_pair(p.y_true, i, yt[0], yt[E - 1]);
                load_row_pair(p.y_pred, i, yp[0], yp[E - 1]);
            } else {
                yt[0] = load_row(p.y_true, i);
                yp[0] = load_row(p.y_pred, i);
            }
            float g[E][6];
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const float cls = class_loss(yt[e], yp[e]);
                const float kf = keep[e] ? 1.0f : 0.0f;
                neg_cls += (double)(cls * kf);                                    // :160
                const float pos = yt[e].c.y;
                const float w = pos + kf;
                g[e][0] = yp[e].c.x >= 1e-15f ? (-(yt[e].c.x * w) / yp[e].c.x) * scale : 0.0f;
                g[e][1] = yp[e].c.y >= 1e-15f ? (-(yt[e].c.y * w) / yp[e].c.y) * scale : 0.0f;
                const float ap = p.alpha * pos;
                g[e][2] = (ap * smooth_l1_slope(yp[e].xy.x, yt[e].xy.x)) * scale;
                g[e][3] = (ap * smooth_l1_slope(yp[e].xy.y, yt[e].xy.y)) * scale;
                g[e][4] = (ap * smooth_l1_slope(yp[e].wh.x, yt[e].wh.x)) * scale;
                g[e][5] = (ap * smooth_l1_slope(yp[e].wh.y, yt[e].wh.y)) * scale;
            }
            if (p.keep) {
                if (PAIR)
                    *reinterpret_cast<unsigned short *>(p.keep + i) =
                        (unsigned short)((keep[0] ? 1u : 0u) | (keep[E - 1] ? 0x100u : 0u));
                else
                    p.keep[i] = keep[0] ? 1 : 0;
            }
            if (p.grad) {
                if (PAIR) {
                    float4 *o = reinterpret_cast<float4 *>(p.grad + i * 6);
                    o[0] = make_float4(g[0][0], g[0][1], g[0][2], g[0][3]);
                    o[1] = make_float4(g[0][4], g[0][5], g[E - 1][0], g[E - 1][1]);
                    o[2] = make_float4(g[E - 1][2], g[E - 1][3], g[E - 1][4], g[E - 1][5]);
                } else {
                    float2 *o = reinterpret_cast<float2 *>(p.grad + i * 6);
                    o[0] = make_float2(g[0][0], g[0][1]);
                    o[1] = make_float2(g[0][2], g[0][3]);
                    o[2] = make_float2(g[0][4], g[0][5]);
                }
            }
        }
    }
    neg_cls = block_sum(neg_cls, s_red);
    if (t == 0 && neg_cls != 0.0) atomicAdd(&p.sums[4 * img + 2], neg_cls);
}

// A launch of its own: a "last workgroup finishes" scheme inside the final kernel needs a
// device-scope fence per workgroup, which writes back the L2 of the workgroup's XCD each time
// (measured: 42 -> 116 us).
__global__ __launch_bounds__(kLB) void loss_finish_kernel(const LossArgs p)
{
    const int b = blockIdx.x * kLB + threadIdx.x;
    if (b >= p.batch) return;
    const double n_pos = p.state->n_pos;
    const double denom = n_pos > 1.0 ? n_pos : 1.0;
    const double *s = p.sums + 4 * b;
    p.total[b] = (float)((s[0] + s[2] + (double)p.alpha * s[1]) / denom * (double)p.batch);   // :174-179
}

}  // namespace

extern "C" int ssdbox_ssd_loss(ssdbox_ctx *ctx, const float *d_y_true, const float *d_y_pred, int batch,
                               int64_t num_anchors, int neg_pos_ratio, int n_neg_min, float alpha,
                               float *d_total, uint8_t *d_keep, float *d_grad)
{
    SSD_REQUIRE(ctx != nullptr, "ssd_loss: ctx is NULL");
    SSD_REQUIRE(batch >= 1 && batch <= 65535, "ssd_loss: batch %d outside [1, 65535]", batch);
    SSD_REQUIRE(num_anchors >= 1 && (long long)batch * num_anchors < (1ll << 31),
                "ssd_loss: batch %d x %lld anchors", batch, (long long)num_anchors);
    SSD_REQUIRE(neg_pos_ratio >= 0 && n_neg_min >= 0, "ssd_loss: negative mining parameter");
    SSD_REQUIRE(d_y_true && d_y_pred && d_total, "ssd_loss: NULL buffer");

    DeviceGuard guard(ctx->device);
    const long long n = (long long)batch * num_anchors;
    int bpi = (int)((num_anchors + kLB * kPerThread - 1) / (kLB * kPerThread));
    while ((long long)bpi * batch > (1 << 20)) bpi = (bpi + 1) / 2;      // bound the chunk scan
    long long chunk = (num_anchors + bpi - 1) / bpi;
    chunk = (chunk + kLB - 1) / kLB * kLB;
    bpi = (int)((num_anchors + chunk - 1) / chunk);

    int rc = ensure_buffer((void **)&ctx->loss_key, &ctx->loss_key_cap, (size_t)n, sizeof(unsigned), false,
                           ctx->stream);
    if (rc != SSDBOX_OK) return rc;
    // one block of small scratch: histograms, state, per-image sums, chunk and image counts
    const size_t small_bytes = 3 * kBins * sizeof(unsigned) + 256 + (size_t)batch * 4 * sizeof(double) +
                               (size_t)batch * bpi * sizeof(unsigned) + (size_t)batch * sizeof(unsigned);
    rc = ensure_buffer((void **)&ctx->loss_small, &ctx->loss_small_cap, small_bytes, 1, false, ctx->stream);
    if (rc != SSDBOX_OK) return rc;
    SSD_CHECK_HIP(hipMemsetAsync(ctx->loss_small, 0, small_bytes, ctx->stream));

    LossArgs a;
    a.y_true = d_y_true;
    a.y_pred = d_y_pred;
    a.batch = batch;
    a.num_anchors = (int)num_anchors;
    a.blocks_per_image = bpi;
    a.chunk = (int)chunk;
    a.neg_pos_ratio = neg_pos_ratio;
    a.n_neg_min = n_neg_min;
    a.alpha = alpha;
    a.key = ctx->loss_key;
    char *small = reinterpret_cast<char *>(ctx->loss_small);
    a.hist = reinterpret_cast<unsigned *>(small);
    a.state = reinterpret_cast<LossState *>(small + 3 * kBins * sizeof(unsigned));
    a.sums = reinterpret_cast<double *>(small + 3 * kBins * sizeof(unsigned) + 256);
    a.chunk_eq = reinterpret_cast<unsigned *>(small + 3 * kBins * sizeof(unsigned) + 256 +
                                              (size_t)batch * 4 * sizeof(double));
    a.img_eq = a.chunk_eq + (size_t)batch * bpi;
    a.total = d_total;
    a.keep = d_keep;
    a.grad = d_grad;
    static_assert(sizeof(LossState) <= 256, "state block");

    StageTimer timer(ctx, SSDBOX_STAGE_LOSS);
    const dim3 grid(bpi, batch);
    // two anchors per thread and load where every pair is 16-byte aligned
    const bool pair = num_anchors % 2 == 0 && ((uintptr_t)d_y_true % 16) == 0 && ((uintptr_t)d_y_pred % 16) == 0 &&
                      (d_grad == nullptr || ((uintptr_t)d_grad % 16) == 0) &&
                      (d_keep == nullptr || ((uintptr_t)d_keep % 2) == 0);
    if (pair)
        hipLaunchKernelGGL(loss_pass1_kernel<true>, grid, dim3(kLB), 0, ctx->stream, a);
    else
        hipLaunchKernelGGL(loss_pass1_kernel<false>, grid, dim3(kLB), 0, ctx->stream, a);
    hipLaunchKernelGGL(loss_hist_kernel<1>, grid, dim3(kLB), 0, ctx->stream, a);
    hipLaunchKernelGGL(loss_hist_kernel<2>, grid, dim3(kLB), 0, ctx->stream, a);
    hipLaunchKernelGGL(loss_count_eq_kernel, grid, dim3(kLB), 0, ctx->stream, a);
    if (pair)
        hipLaunchKernelGGL(loss_final_kernel<true>, grid, dim3(kLB), 0, ctx->stream, a);
    else
        hipLaunchKernelGGL(loss_final_kernel<false>, grid, dim3(kLB), 0, ctx->stream, a);
    hipLaunchKernelGGL(loss_finish_kernel, dim3((batch + kLB - 1) / kLB), dim3(kLB), 0, ctx->stream, a);
    SSD_CHECK_HIP(hipGetLastError());
    return SSDBOX_OK;
}
