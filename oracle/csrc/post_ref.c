/* ORACLE (test infrastructure, not product code).
 *
 * Plain-C restatement of tf.raw_ops.NonMaxSuppressionV5 as the reference calls
 * it (src/postprocess.py:392-400).  TensorFlow 2.10 is a pinned, un-vendored
 * dependency (requirements.txt:6) that cannot be installed here, so this
 * restates the published algorithm of its CPU kernel
 * (tensorflow/core/kernels/image/non_max_suppression_op.cc, DoNonMaxSuppressionOp
 * with a float IOU similarity) — PARITY UNPINNED, anchored by the hand-derived
 * KATs in tests/test_oracle_kats.py:
 *
 *   - candidates: score > score_threshold, max-heap on (score, then SMALLER index first)
 *   - scale = soft_nms_sigma > 0 ? -0.5 / soft_nms_sigma : 0
 *   - pop c; for j = |selected|-1 .. c.suppress_begin: s = IOU(c, selected[j]);
 *       c.score *= (soft || s <= thr) ? exp(scale*s*s) : 0;
 *       hard mode and s > thr  -> drop c;      c.score <= score_threshold -> stop
 *     c.suppress_begin = |selected|
 *     unchanged score -> select (index, current score); else re-push if > score_threshold
 *   - pad_to_max_output_size: indices padded with 0, scores with 0.0
 *
 * exp(): TF evaluates std::exp(float).  The build DEFINES the weight as
 * (float)exp((double)x) — the correctly rounded value except with probability
 * ~2^-28 — so that the HIP kernel (which evaluates the same expression with the
 * device's double exp) reproduces it bit for bit.
 *
 * Build: oracle/Makefile  ->  oracle/_build/libpost_ref.so   (gcc -O2 -ffp-contract=off)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

typedef struct {
  float score;
  int32_t idx;
  int32_t begin;
} cand_t;

/* a outranks b: larger score, ties -> smaller index */
static int outranks(const cand_t* a, const cand_t* b) {
  return a->score > b->score || (a->score == b->score && a->idx < b->idx);
}

static void sift_down(cand_t* h, int n, int i) {
  for (;;) {
    int l = 2 * i + 1, r = l + 1, m = i;
    if (l < n && outranks(&h[l], &h[m])) m = l;
    if (r < n && outranks(&h[r], &h[m])) m = r;
    if (m == i) return;
    cand_t t = h[i]; h[i] = h[m]; h[m] = t;
    i = m;
  }
}

static void sift_up(cand_t* h, int i) {
  while (i > 0) {
    int p = (i - 1) / 2;
    if (!outranks(&h[i], &h[p])) return;
    cand_t t = h[i]; h[i] = h[p]; h[p] = t;
    i = p;
  }
}

float oracle_iou(const float* a, const float* b) {
  const float ymin_i = fminf(a[0], a[2]), xmin_i = fminf(a[1], a[3]);
  const float ymax_i = fmaxf(a[0], a[2]), xmax_i = fmaxf(a[1], a[3]);
  const float ymin_j = fminf(b[0], b[2]), xmin_j = fminf(b[1], b[3]);
  const float ymax_j = fmaxf(b[0], b[2]), xmax_j = fmaxf(b[1], b[3]);
  const float area_i = (ymax_i - ymin_i) * (xmax_i - xmin_i);
  const float area_j = (ymax_j - ymin_j) * (xmax_j - xmin_j);
  if (area_i <= 0 || area_j <= 0) return 0.0f;
  const float iy0 = fmaxf(ymin_i, ymin_j), ix0 = fmaxf(xmin_i, xmin_j);
  const float iy1 = fminf(ymax_i, ymax_j), ix1 = fminf(xmax_i, xmax_j);
  const float inter = fmaxf(iy1 - iy0, 0.0f) * fmaxf(ix1 - ix0, 0.0f);
  return inter / (area_i + area_j - inter);
}

float oracle_suppress_weight(float sim, float scale) {
  return (float)exp((double)(scale * sim * sim));
}

/* returns number of valid outputs; sel_idx / sel_scores hold max_out entries when
 * pad != 0 (padded with 0 / 0.0f), else only the first `valid` are written. */
int oracle_nms_v5(const float* boxes, const float* scores, int n, int max_out,
                  float iou_thr, float score_thr, float soft_sigma, int pad,
                  int32_t* sel_idx, float* sel_scores) {
  cand_t* heap = (cand_t*)malloc(sizeof(cand_t) * (size_t)(n > 0 ? n : 1));
  int hn = 0;
  for (int i = 0; i < n; ++i) {
    if (scores[i] > score_thr) {
      heap[hn].score = scores[i];
      heap[hn].idx = i;
      heap[hn].begin = 0;
      ++hn;
    }
  }
  for (int i = hn / 2 - 1; i >= 0; --i) sift_down(heap, hn, i);

  const int soft = soft_sigma > 0.0f;
  const float scale = soft ? -0.5f / soft_sigma : 0.0f;
  int nsel = 0;
  while (nsel < max_out && hn > 0) {
    cand_t c = heap[0];
    heap[0] = heap[--hn];
    if (hn > 0) sift_down(heap, hn, 0);
    const float original = c.score;
    int hard_suppressed = 0;
    for (int j = nsel - 1; j >= c.begin; --j) {
      const float sim = oracle_iou(boxes + 4 * (size_t)c.idx, boxes + 4 * (size_t)sel_idx[j]);
      const float wgt = (soft || sim <= iou_thr) ? oracle_suppress_weight(sim, scale) : 0.0f;
      c.score *= wgt;
      if (!soft && sim > iou_thr) { hard_suppressed = 1; break; }
      if (c.score <= score_thr) break;
    }
    c.begin = nsel;
    if (!hard_suppressed) {
      if (c.score == original) {
        sel_idx[nsel] = c.idx;
        sel_scores[nsel] = c.score;
        ++nsel;
        continue;
      }
      if (c.score > score_thr) {
        heap[hn] = c;
        sift_up(heap, hn);
        ++hn;
      }
    }
  }
  free(heap);
  if (pad) {
    for (int i = nsel; i < max_out; ++i) { sel_idx[i] = 0; sel_scores[i] = 0.0f; }
  }
  return nsel;
}

/* sigmoid as the build defines it: (float)(1 / (1 + exp(-(double)x))) */
void oracle_sigmoid(const float* x, float* y, int64_t n) {
  for (int64_t i = 0; i < n; ++i) y[i] = (float)(1.0 / (1.0 + exp(-(double)x[i])));
}
