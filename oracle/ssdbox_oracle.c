/*
 * Scalar C oracle for the SSD box pipeline -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load the
 * library built from this file; nothing under face_detection_amd/ links or calls it.
 *
 * It restates, as plain loops over float32 scalars, the reference's TensorFlow-eager box
 * math: utils/anchor_utils.py:6-62 (anchors), utils/box_utils.py:4-222 (area, IoU, match,
 * encode, decode, NMS) and the caller tails utils/datasets.py:152-159 and
 * inference.py:33-83.  It exists next to the numpy oracle because (a) it never
 * materialises the (A,G) IoU matrix, so full-size batches finish in seconds, and (b) its
 * exp/log are a fixed sequence of IEEE float32 operations (the Cephes single-precision
 * algorithms, the same family Eigen -- TensorFlow's CPU math -- derives from), so a second
 * implementation of the same sequence can be compared bit for bit.
 *
 * PARITY UNPINNED: the reference has no tests or golden vectors for this path and its
 * runtime (TensorFlow 2.4) is absent, so this file is pinned only by hand-derived
 * known-answer tests and by agreement with the numpy oracle (bit-exact on + - * / min max,
 * <= 2 ULP on exp/log).  Conventions frozen here are listed in oracle/box_oracle.py.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off, no fast-math).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* ------------------------------------------------------------------ exp / log (Cephes) */
float orc_logf(float x)
{
    if (x != x) return x;
    if (x < 0.0f) return u2f(0x7FC00000u);
    if (x == 0.0f) return u2f(0xFF800000u);
    if (f2u(x) == 0x7F800000u) return x;
    int e = 0;
    uint32_t u = f2u(x);
    if ((u >> 23) == 0) {            /* subnormal: renormalise exactly */
        x = x * 8388608.0f;
        u = f2u(x);
        e = -23;
    }
    e += (int)(u >> 23) - 126;       /* x = m * 2^e with m in [0.5, 1) */
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

float orc_expf(float x)
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
    /* z * 2^n with a single rounding even when the result is subnormal */
    if (n > 127) { z = z * u2f((uint32_t)(100 + 127) << 23); n -= 100; }
    else if (n < -126) { z = z * u2f((uint32_t)(-100 + 127) << 23); n += 100; }
    return z * u2f((uint32_t)(n + 127) << 23);
}

/* ---------------------------------------------------------------------------- anchors */
/* utils/anchor_utils.py:6-62.  ratios_flat holds the per-layer ratio lists back to back,
 * ratio_counts[m] entries for layer m.  Returns the number of anchors; writes
 * [cx, cy, w, h] float32 rows when out is not NULL. */
int64_t orc_generate_anchors(int img_w, int img_h, int n_layers, const int *steps,
                             const double *scales, const int *ratio_counts,
                             const double *ratios_flat, int two_boxes, double offsets,
                             float *out)
{
    int64_t n = 0;
    const double *ratios = ratios_flat;
    for (int m = 0; m < n_layers; ++m) {
        int fm_w = img_w / steps[m], fm_h = img_h / steps[m];
        double bw[64], bh[64];
        int nb = 0;
        for (int r = 0; r < ratio_counts[m]; ++r) {
            double ratio = ratios[r];
            if (ratio == 1.0) {
                bw[nb] = scales[m]; bh[nb] = scales[m]; ++nb;
                if (two_boxes) {
                    double s = sqrt(scales[m] * scales[m + 1]);
                    bw[nb] = s; bh[nb] = s; ++nb;
                }
            } else {
                double root = sqrt(ratio);
                bw[nb] = scales[m] * root; bh[nb] = scales[m] / root; ++nb;
            }
        }
        ratios += ratio_counts[m];
        for (int i = 0; i < fm_h; ++i) {
            double cy = (i + offsets) / fm_h;
            for (int j = 0; j < fm_w; ++j) {
                double cx = (j + offsets) / fm_w;
                for (int k = 0; k < nb; ++k) {
                    if (out) {
                        float v[4] = {(float)cx, (float)cy, (float)bw[k], (float)bh[k]};
                        for (int c = 0; c < 4; ++c)
                            out[n * 4 + c] = fminf(fmaxf(v[c], 0.0f), 1.0f);
                    }
                    ++n;
                }
            }
        }
    }
    return n;
}

/* ------------------------------------------------------------------------ box helpers */
static inline float clip0(float v, float hi) { return fminf(fmaxf(v, 0.0f), hi); }

/* utils/box_utils.py:4-18 on one pair of corners */
static inline float area1(float x0, float y0, float x1, float y1, float max_size)
{
    return clip0(x1 - x0, max_size) * clip0(y1 - y0, max_size);
}

/* utils/box_utils.py:21-44 on one pair of corner boxes */
static inline float iou1(const float *a, const float *b, float max_size)
{
    float inter = area1(fmaxf(a[0], b[0]), fmaxf(a[1], b[1]),
                        fminf(a[2], b[2]), fminf(a[3], b[3]), max_size);
    float area_a = area1(a[0], a[1], a[2], a[3], max_size);
    float area_b = area1(b[0], b[1], b[2], b[3], max_size);
    return inter / (area_a + area_b - inter);
}

void orc_compute_iou(const float *a, int64_t na, const float *b, int64_t nb,
                     float max_size, float *out)
{
    for (int64_t i = 0; i < na; ++i)
        for (int64_t j = 0; j < nb; ++j)
            out[i * nb + j] = iou1(a + 4 * i, b + 4 * j, max_size);
}

/* utils/box_utils.py:167-181 */
static inline void to_corner(const float *c, float *o)
{
    float hw = c[2] / 2.0f, hh = c[3] / 2.0f;
    o[0] = c[0] - hw; o[1] = c[1] - hh; o[2] = c[0] + hw; o[3] = c[1] + hh;
}

/* utils/box_utils.py:101-123 for one anchor / one corner box */
static inline void encode1(const float *anc, const float *box, float *t)
{
    float cx = (box[0] + box[2]) / 2.0f, cy = (box[1] + box[3]) / 2.0f;
    float w = box[2] - box[0], h = box[3] - box[1];
    t[0] = (cx - anc[0]) / (anc[2] * 0.1f);
    t[1] = (cy - anc[1]) / (anc[3] * 0.1f);
    t[2] = orc_logf(w / anc[2]) / 0.2f;
    t[3] = orc_logf(h / anc[3]) / 0.2f;
}

/* utils/box_utils.py:126-147 for one anchor / one offset row */
static inline void decode1(const float *anc, const float *l, float *box)
{
    float c[4];
    c[0] = l[0] * 0.1f * anc[2] + anc[0];
    c[1] = l[1] * 0.1f * anc[3] + anc[1];
    c[2] = orc_expf(l[2] * 0.2f) * anc[2];
    c[3] = orc_expf(l[3] * 0.2f) * anc[3];
    to_corner(c, box);
}

/* rows = leading batch dims flattened; anchor index = row % A (broadcast over batch) */
void orc_encode(const float *anchors, int64_t A, const float *boxes, int64_t rows, float *out)
{
    for (int64_t r = 0; r < rows; ++r)
        encode1(anchors + 4 * (r % A), boxes + 4 * r, out + 4 * r);
}

void orc_decode(const float *anchors, int64_t A, const float *locs, int64_t rows, float *out)
{
    for (int64_t r = 0; r < rows; ++r)
        decode1(anchors + 4 * (r % A), locs + 4 * r, out + 4 * r);
}

/* --------------------------------------------------------------------- compute_target */
/* utils/box_utils.py:47-98 for one image.  Optional outputs (may be NULL): match_idx
 * (best_gt_idx after the forced scatter), match_iou (best_gt_iou after it), best_anchor
 * (best_default_idx per GT). */
int orc_compute_target(const float *anchors, int64_t A, const float *gt, const float *labels,
                       int G, float max_size, float iou_thr, float *conf, float *locs,
                       int32_t *match_idx, float *match_iou, int32_t *best_anchor)
{
    if (G <= 0 || A <= 0) return -1;
    int32_t *idx = (int32_t *)malloc(sizeof(int32_t) * (size_t)A);
    float *best = (float *)malloc(sizeof(float) * (size_t)A);
    float *gbest = (float *)malloc(sizeof(float) * (size_t)G);
    int64_t *ganc = (int64_t *)malloc(sizeof(int64_t) * (size_t)G);
    for (int g = 0; g < G; ++g) { gbest[g] = 0.0f; ganc[g] = 0; }
    for (int64_t a = 0; a < A; ++a) {
        float c[4];
        to_corner(anchors + 4 * a, c);
        float b = 0.0f;
        int32_t bi = 0;
        for (int g = 0; g < G; ++g) {
            float v = iou1(c, gt + 4 * g, max_size);
            if (v > b) { b = v; bi = g; }                       /* first maximum, NaN never wins */
            if (v > gbest[g]) { gbest[g] = v; ganc[g] = a; }    /* lowest anchor among maxima */
        }
        idx[a] = bi;
        best[a] = b;
    }
    for (int g = 0; g < G; ++g) {              /* ascending: the last duplicate wins */
        idx[ganc[g]] = g;
        best[ganc[g]] = 1.0f;
    }
    for (int64_t a = 0; a < A; ++a) {
        conf[a] = (best[a] < iou_thr) ? 0.0f : labels[idx[a]];
        encode1(anchors + 4 * a, gt + 4 * idx[a], locs + 4 * a);
        if (match_idx) match_idx[a] = idx[a];
        if (match_iou) match_iou[a] = best[a];
    }
    if (best_anchor) for (int g = 0; g < G; ++g) best_anchor[g] = (int32_t)ganc[g];
    free(idx); free(best); free(gbest); free(ganc);
    return 0;
}

/* utils/datasets.py:155-159: one_hot(int32(conf), 2) ++ locs -> (A,6) */
void orc_y_true(const float *conf, const float *locs, int64_t A, float *y)
{
    for (int64_t a = 0; a < A; ++a) {
        int c = (int)conf[a];
        y[6 * a + 0] = (c == 0) ? 1.0f : 0.0f;
        y[6 * a + 1] = (c == 1) ? 1.0f : 0.0f;
        memcpy(y + 6 * a + 2, locs + 4 * a, 16);
    }
}

/* -------------------------------------------------------------------------------- NMS */
typedef struct { float score; int32_t idx; } ranked_t;

static int rank_cmp(const void *pa, const void *pb)
{
    const ranked_t *a = (const ranked_t *)pa, *b = (const ranked_t *)pb;
    int an = a->score != a->score, bn = b->score != b->score;
    if (an != bn) return an - bn;                        /* NaN scores rank last */
    if (!an) {
        if (a->score > b->score) return -1;
        if (a->score < b->score) return 1;
    }
    return (a->idx > b->idx) - (a->idx < b->idx);        /* stable: lower index first */
}

/* utils/box_utils.py:184-222.  keep must hold min(limit, n) entries; returns the count.
 * Suppression test is !(iou <= thr), the IoU row is that of the already kept box. */
int orc_nms(const float *boxes, const float *scores, int64_t n, float thr, int limit,
            int32_t *keep)
{
    if (n <= 0) return 0;
    ranked_t *r = (ranked_t *)malloc(sizeof(ranked_t) * (size_t)n);
    for (int64_t i = 0; i < n; ++i) { r[i].score = scores[i]; r[i].idx = (int32_t)i; }
    qsort(r, (size_t)n, sizeof(ranked_t), rank_cmp);
    int64_t k = n < limit ? n : limit;
    unsigned char *dead = (unsigned char *)calloc((size_t)k, 1);
    int nk = 0;
    for (int64_t i = 0; i < k; ++i) {
        if (dead[i] && i > 0) continue;                  /* rank 0 is always selected */
        keep[nk++] = r[i].idx;
        const float *bi = boxes + 4 * (int64_t)r[i].idx;
        for (int64_t j = i + 1; j < k; ++j) {
            if (dead[j]) continue;
            float v = iou1(bi, boxes + 4 * (int64_t)r[j].idx, 512.0f);
            if (!(v <= thr)) dead[j] = 1;
        }
    }
    free(r); free(dead);
    return nk;
}

/* inference.py:33-83 for one image: y_pred (A,6) [bg, face, tx, ty, tw, th].
 * dets rows are [1, score, xmin, ymin, xmax, ymax] clipped to [0,1] then scaled; idx are
 * the kept anchor indices.  Both must hold `limit` rows.  Returns the count. */
/* The head's class activation for two classes (reference models/ssd.py:95, Keras softmax over
 * the last axis; TF 2.4 evaluates it as exp(x - max) * (1 / sum exp(x - max)) in float32).
 * rows6 (rows, 6) in, out (rows, 6): columns 0 and 1 replaced by their softmax. */
void orc_softmax2(const float *rows6, int64_t rows, float *out)
{
    for (int64_t i = 0; i < rows; ++i) {
        const float *r = rows6 + i * 6;
        float *o = out + i * 6;
        const float m = fmaxf(r[0], r[1]);
        const float e0 = orc_expf(r[0] - m), e1 = orc_expf(r[1] - m);
        const float inv = 1.0f / (e0 + e1);
        o[0] = e0 * inv;
        o[1] = e1 * inv;
        for (int c = 2; c < 6; ++c) o[c] = r[c];
    }
}

int orc_postprocess(const float *y_pred, const float *anchors, int64_t A, float conf_thr,
                    float nms_thr, int limit, float img_w, float img_h, float *dets,
                    int32_t *idx)
{
    int64_t nc = 0;
    for (int64_t a = 0; a < A; ++a) nc += (y_pred[6 * a + 1] > conf_thr);
    if (nc == 0) return 0;
    float *cb = (float *)malloc(sizeof(float) * 4 * (size_t)nc);
    float *cs = (float *)malloc(sizeof(float) * (size_t)nc);
    int32_t *ca = (int32_t *)malloc(sizeof(int32_t) * (size_t)nc);
    int64_t c = 0;
    for (int64_t a = 0; a < A; ++a) {
        if (!(y_pred[6 * a + 1] > conf_thr)) continue;
        decode1(anchors + 4 * a, y_pred + 6 * a + 2, cb + 4 * c);
        cs[c] = y_pred[6 * a + 1];
        ca[c] = (int32_t)a;
        ++c;
    }
    int32_t *keep = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nc < limit ? nc : limit));
    int nk = orc_nms(cb, cs, nc, nms_thr, limit, keep);
    for (int i = 0; i < nk; ++i) {
        const float *b = cb + 4 * (int64_t)keep[i];
        dets[6 * i + 0] = 1.0f;
        dets[6 * i + 1] = cs[keep[i]];
        dets[6 * i + 2] = clip0(b[0], 1.0f) * img_w;
        dets[6 * i + 3] = clip0(b[1], 1.0f) * img_h;
        dets[6 * i + 4] = clip0(b[2], 1.0f) * img_w;
        dets[6 * i + 5] = clip0(b[3], 1.0f) * img_h;
        idx[i] = ca[keep[i]];
    }
    free(cb); free(cs); free(ca); free(keep);
    return nk;
}

/* ------------------------------------------------------------------------------- loss */
/* SSD loss with hard-negative mining, utils/losses.py:97-181, for rows of 2 + 4 values
 * [background, face, 4 offsets], and the gradient of its batch mean with respect to y_pred
 * (what TensorFlow's autodiff gives: the mask and the counts are constants).
 * Per-anchor values are float32 (logarithm = orc_logf), sums run in double.
 * keep (B*A) receives the mined-negative mask; cls and grad may be NULL.
 * tf.nn.top_k order: larger value first, equal values by ascending index, NaN last. */
int orc_ssd_loss(const float *yt, const float *yp, int B, int64_t A, int neg_pos_ratio,
                 int n_neg_min, float alpha, float *total, unsigned char *keep, float *cls_out,
                 float *grad, double *n_positive_out)
{
    const int64_t N = (int64_t)B * A;
    float *cls = (float *)malloc(sizeof(float) * (size_t)N);
    float *loc = (float *)malloc(sizeof(float) * (size_t)N);
    ranked_t *r = (ranked_t *)malloc(sizeof(ranked_t) * (size_t)N);
    if (!cls || !loc || !r || N >= (1ll << 31)) { free(cls); free(loc); free(r); return -1; }
    double n_pos = 0.0;
    int64_t n_neg_losses = 0;
    for (int64_t i = 0; i < N; ++i) {
        const float *t = yt + 6 * i, *p = yp + 6 * i;
        const float l0 = orc_logf(fmaxf(p[0], 1e-15f)), l1 = orc_logf(fmaxf(p[1], 1e-15f));
        cls[i] = -(t[0] * l0 + t[1] * l1);                      /* :78-81 */
        float s = 0.0f;
        for (int c = 0; c < 4; ++c) {                           /* :58-61 */
            const float d = t[2 + c] - p[2 + c];
            const float a = fabsf(d);
            s = s + (a < 1.0f ? 0.5f * (d * d) : a - 0.5f);
        }
        loc[i] = s;
        n_pos += (double)t[1];                                  /* :109-112 */
        const float neg_all = cls[i] * t[0];                    /* :122 */
        if (neg_all != 0.0f) ++n_neg_losses;                    /* :123, NaN counts */
        r[i].score = neg_all;
        r[i].idx = (int32_t)i;
        keep[i] = 0;
    }
    int64_t k = (int64_t)neg_pos_ratio * (int64_t)n_pos;        /* :135-136 */
    if (k < n_neg_min) k = n_neg_min;
    if (k > n_neg_losses) k = n_neg_losses;
    if (n_neg_losses != 0 && k > 0) {                           /* :163, :149-157 */
        qsort(r, (size_t)N, sizeof(ranked_t), rank_cmp);
        for (int64_t i = 0; i < k; ++i) keep[r[i].idx] = 1;
    }
    const double denom = n_pos > 1.0 ? n_pos : 1.0;             /* :174 */
    for (int b = 0; b < B; ++b) {
        double pos_c = 0.0, neg_c = 0.0, loc_s = 0.0;
        for (int64_t a = 0; a < A; ++a) {
            const int64_t i = (int64_t)b * A + a;
            const float pos = yt[6 * i + 1];
            pos_c += (double)(cls[i] * pos);                    /* :117 */
            neg_c += (double)(cls[i] * (float)keep[i]);         /* :160 */
            loc_s += (double)(loc[i] * pos);                    /* :170 */
        }
        total[b] = (float)((pos_c + neg_c + (double)alpha * loc_s) / denom * (double)B);   /* :174-179 */
    }
    if (grad) {
        const float scale = (float)(1.0 / denom);
        for (int64_t i = 0; i < N; ++i) {
            const float *t = yt + 6 * i, *p = yp + 6 * i;
            float *g = grad + 6 * i;
            const float w = t[1] + (float)keep[i];
            for (int c = 0; c < 2; ++c)
                g[c] = p[c] >= 1e-15f ? (-(t[c] * w) / p[c]) * scale : 0.0f;
            for (int c = 0; c < 4; ++c) {
                const float d = p[2 + c] - t[2 + c];
                const float dl = fabsf(d) < 1.0f ? d : (d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : d));
                g[2 + c] = ((alpha * t[1]) * dl) * scale;
            }
        }
    }
    if (cls_out) memcpy(cls_out, cls, sizeof(float) * (size_t)N);
    if (n_positive_out) *n_positive_out = n_pos;
    free(cls);
    free(loc);
    free(r);
    return 0;
}
