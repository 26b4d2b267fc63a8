/* Sanitizer run of the scalar C oracle (test infrastructure): every entry point on small seeded
 * inputs, degenerate ones included, built with -fsanitize=address,undefined by `make sanitize`.
 * Exit status 0 and no sanitizer report mean no out-of-bounds access, leak or undefined
 * behaviour on these inputs.  tests/test_oracle_sanitize.py runs it. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

float orc_logf(float x);
float orc_expf(float x);
int64_t orc_generate_anchors(int img_w, int img_h, int n_layers, const int *steps, const double *scales,
                             const int *ratio_counts, const double *ratios_flat, int two_boxes, double offsets,
                             float *out);
void orc_compute_iou(const float *a, int64_t na, const float *b, int64_t nb, float max_size, float *out);
void orc_encode(const float *anchors, int64_t A, const float *boxes, int64_t rows, float *out);
void orc_decode(const float *anchors, int64_t A, const float *locs, int64_t rows, float *out);
int orc_compute_target(const float *anchors, int64_t A, const float *gt, const float *labels, int G, float max_size,
                       float iou_thr, float *conf, float *locs, int32_t *match_idx, float *match_iou,
                       int32_t *best_anchor);
void orc_y_true(const float *conf, const float *locs, int64_t A, float *y);
int orc_nms(const float *boxes, const float *scores, int64_t n, float thr, int limit, int32_t *keep);
void orc_softmax2(const float *rows6, int64_t rows, float *out);
int orc_postprocess(const float *y_pred, const float *anchors, int64_t A, float conf_thr, float nms_thr, int limit,
                    float img_w, float img_h, float *dets, int32_t *idx);
int orc_ssd_loss(const float *yt, const float *yp, int B, int64_t A, int neg_pos_ratio, int n_neg_min, float alpha,
                 float *total, unsigned char *keep, float *cls_out, float *grad, double *n_positive_out);

static uint32_t state = 12345u;
static float rnd(void)
{
    state = state * 1664525u + 1013904223u;
    return (float)(state >> 8) / 16777216.0f;
}

int main(void)
{
    const int steps[3] = {8, 16, 30};
    const double scales[4] = {0.1, 0.35, 0.9, 1.4};
    const int counts[3] = {2, 1, 4};
    const double ratios[7] = {1.0, 3.0, 0.5, 1.0, 2.0, 0.5, 3.0};
    for (int two = 0; two < 2; ++two) {
        const int64_t A = orc_generate_anchors(75, 100, 3, steps, scales, counts, ratios, two, 0.3, NULL);
        float *anchors = malloc(sizeof(float) * 4 * (size_t)A);
        if (orc_generate_anchors(75, 100, 3, steps, scales, counts, ratios, two, 0.3, anchors) != A) return 2;

        for (int G = 1; G <= 9; G += 4) {
            float *gt = malloc(sizeof(float) * 4 * (size_t)G), *lab = malloc(sizeof(float) * (size_t)G);
            for (int g = 0; g < G; ++g) {
                const float x = rnd() * 0.8f, y = rnd() * 0.8f;
                gt[4 * g] = x; gt[4 * g + 1] = y; gt[4 * g + 2] = x + rnd() * 0.3f; gt[4 * g + 3] = y + rnd() * 0.3f;
                lab[g] = 1.0f;
            }
            if (G > 4) {                                   /* degenerate and non-finite boxes */
                gt[4] = gt[6];
                gt[9] = NAN;
                gt[14] = INFINITY;
            }
            float *conf = malloc(sizeof(float) * (size_t)A), *locs = malloc(sizeof(float) * 4 * (size_t)A);
            float *miou = malloc(sizeof(float) * (size_t)A), *y = malloc(sizeof(float) * 6 * (size_t)A);
            int32_t *midx = malloc(sizeof(int32_t) * (size_t)A), *banc = malloc(sizeof(int32_t) * (size_t)G);
            if (orc_compute_target(anchors, A, gt, lab, G, 100.0f, 0.5f, conf, locs, midx, miou, banc) != 0) return 3;
            if (orc_compute_target(anchors, A, gt, lab, G, 100.0f, 0.5f, conf, locs, NULL, NULL, NULL) != 0) return 3;
            orc_y_true(conf, locs, A, y);
            float *iou = malloc(sizeof(float) * (size_t)A * (size_t)G);
            orc_compute_iou(anchors, A, gt, G, 512.0f, iou);
            free(iou); free(conf); free(miou); free(midx); free(banc); free(gt); free(lab);

            /* head outputs: the targets as offsets, random two-class scores with ties and a NaN */
            float *yp = malloc(sizeof(float) * 6 * (size_t)A), *sm = malloc(sizeof(float) * 6 * (size_t)A);
            for (int64_t a = 0; a < A; ++a) {
                const float p = (a % 7 == 0) ? 0.75f : rnd();
                yp[6 * a] = 1.0f - p; yp[6 * a + 1] = p;
                for (int c = 0; c < 4; ++c) yp[6 * a + 2 + c] = (rnd() - 0.5f) * 4.0f;
            }
            yp[6 * 5 + 1] = NAN;
            yp[6 * 6 + 4] = 500.0f;                        /* exp overflows */
            orc_softmax2(yp, A, sm);
            float *dec = malloc(sizeof(float) * 4 * (size_t)A), *enc = malloc(sizeof(float) * 4 * (size_t)A);
            float *l4 = malloc(sizeof(float) * 4 * (size_t)A);
            for (int64_t a = 0; a < A; ++a) for (int c = 0; c < 4; ++c) l4[4 * a + c] = yp[6 * a + 2 + c];
            orc_decode(anchors, A, l4, A, dec);
            orc_encode(anchors, A, dec, A, enc);
            const int limits[3] = {1, 50, 1000};
            for (int li = 0; li < 3; ++li) {
                const int limit = limits[li];
                float *dets = malloc(sizeof(float) * 6 * (size_t)limit);
                int32_t *idx = malloc(sizeof(int32_t) * (size_t)limit);
                for (int t = 0; t < 3; ++t) {
                    const float thr = t == 0 ? 0.05f : t == 1 ? 0.0f : 2.0f;     /* 2.0: no candidate */
                    const int n = orc_postprocess(yp, anchors, A, thr, 0.45f, limit, 640.0f, 480.0f, dets, idx);
                    if (n < 0 || n > limit) return 4;
                }
                int32_t *keep = malloc(sizeof(int32_t) * (size_t)(A < limit ? A : limit));
                float *sc = malloc(sizeof(float) * (size_t)A);
                for (int64_t a = 0; a < A; ++a) sc[a] = yp[6 * a + 1];
                if (orc_nms(dec, sc, A, 0.45f, limit, keep) < 0) return 5;
                if (orc_nms(dec, sc, 0, 0.45f, limit, keep) != 0) return 5;
                if (orc_nms(dec, sc, A, -0.5f, limit, keep) < 0) return 5;
                free(keep); free(sc); free(dets); free(idx);
            }
            /* SSD loss on two copies of the image */
            float *yt2 = malloc(sizeof(float) * 12 * (size_t)A), *yp2 = malloc(sizeof(float) * 12 * (size_t)A);
            for (int64_t i = 0; i < 6 * A; ++i) { yt2[i] = yt2[6 * A + i] = y[i]; yp2[i] = yp2[6 * A + i] = yp[i]; }
            yp2[6 * 5 + 1] = 0.3f;                          /* the loss takes logarithms of the scores */
            yp2[6 * A + 6 * 5 + 1] = 0.3f;
            float total[2], *grad = malloc(sizeof(float) * 12 * (size_t)A), *cls = malloc(sizeof(float) * 2 * (size_t)A);
            unsigned char *keepm = malloc(2 * (size_t)A);
            double npos = 0.0;
            if (orc_ssd_loss(yt2, yp2, 2, A, 3, 0, 1.0f, total, keepm, cls, grad, &npos) != 0) return 6;
            if (orc_ssd_loss(yt2, yp2, 2, A, 3, 5, 1.0f, total, keepm, NULL, NULL, NULL) != 0) return 6;
            free(yt2); free(yp2); free(grad); free(cls); free(keepm);
            free(yp); free(sm); free(dec); free(enc); free(l4); free(locs); free(y);
        }
        free(anchors);
    }
    const float xs[8] = {0.0f, -1.0f, 1e-42f, 1.0f, 3.5f, INFINITY, NAN, 88.9f};
    float acc = 0.0f;
    for (int i = 0; i < 8; ++i) acc += (orc_logf(xs[i]) == 0.0f) + (orc_expf(xs[i]) == 0.0f) + (orc_expf(-xs[i]) == 0.0f);
    printf("sanitize ok (%g)\n", (double)acc);
    return 0;
}
