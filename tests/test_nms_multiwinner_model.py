"""CPU model of the epoch-synchronous NMS of csrc/kernels_post.hip WITH several winners per grid-wide step
(nms_coop_kernel, round 5), checked bit for bit against the oracle's plain-C NonMaxSuppressionV5 twin.

The model states the rule the kernel implements, in the kernel's own terms (stale score / begin per candidate, cached
exact score = upper bound, blocks that each offer the exact score of their best candidate by upper bound), so that the
rule is pinned on the CPU before and independently of the HIP code:

  super-step at epoch k
    A  every block: its best candidate by upper bound takes its exact score (links begin .. k-1, newest first).
       bound = the W-th largest of those exact keys (fewer blocks alive: the smallest) -> at least W_eff candidates
       are known to have an exact key >= bound.
    B  every candidate whose upper bound reaches the bound takes its exact score; the W_eff best exact keys e_1 >= e_2
       >= ... are all >= bound, hence above every candidate that was not evaluated (upper bound < bound).
    C  e_1 is the winner of epoch k.  e_j is the winner of epoch k + j - 1 if its box does not strictly overlap e_1 ..
       e_{j-1}: its own score is then unchanged (IoU 0 -> weight exactly 1), every other score can only have decreased.
       The first e_j that overlaps an earlier one ends the super-step (its score will change).
    D  pops, per candidate, for t = 0 .. winners - 1 in order: while the candidate's STALE key outranks e_{t+1}'s key it
       is popped in epoch k + t: exact score over the links begin .. k + t - 1, newest first, begin = k + t.  Exactly the
       pops - hence exactly the float32 multiplication order - of one epoch per winner.
"""
import ctypes
import math

import numpy as np
import pytest

from oracle import post_ref

f32 = np.float32
NEG = f32(-np.inf)


def _key(s, i):
    return (float(s), -int(i))          # larger score first, ties -> smaller index


def _strict_overlap(a, b):
    y0, x0, y1, x1 = min(a[0], a[2]), min(a[1], a[3]), max(a[0], a[2]), max(a[1], a[3])
    sy0, sx0, sy1, sx1 = min(b[0], b[2]), min(b[1], b[3]), max(b[0], b[2]), max(b[1], b[3])
    return (min(y1, sy1) > max(y0, sy0)) and (min(x1, sx1) > max(x0, sx0))


class EpochModel:
    def __init__(self, boxes, scores, max_out, iou_thr, score_thr, soft_sigma, blocks=8, winners=4):
        self.lib = post_ref._lib()
        self.boxes = np.ascontiguousarray(boxes, f32)
        self.K, self.M = len(scores), max_out
        self.soft = soft_sigma > 0
        self.scale = f32(-0.5) / f32(soft_sigma) if self.soft else f32(0)
        self.iou_thr, self.score_thr = f32(iou_thr), f32(score_thr)
        s = np.asarray(scores, f32)
        self.stale = np.where(s > self.score_thr, s, NEG).astype(f32)
        self.ub = self.stale.copy()
        self.begin = np.zeros(self.K, np.int32)
        self.sel, self.sel_score = [], []
        self.blocks, self.W = blocks, winners
        self.steps = 0
        self.multi = 0

    def chain(self, i, score, begin, k):
        """links k-1 .. begin, newest first, with the reference's early exits (csrc chain_product)."""
        for j in range(k - 1, begin - 1, -1):
            if self.soft or self.iou_thr >= 0:
                if not _strict_overlap(self.boxes[i], self.boxes[self.sel[j]]):
                    continue                     # IoU 0, weight exactly 1: the kernel's bit mask skips it
            sim = f32(self.lib.oracle_iou(self.boxes[i].ctypes.data, self.boxes[self.sel[j]].ctypes.data))
            if self.soft or sim <= self.iou_thr:
                e = f32(f32(self.scale * sim) * sim)
                w = f32(1) if e == 0 else f32(math.exp(float(e)))
            else:
                w = f32(0)
            score = f32(score * w)
            if not self.soft and sim > self.iou_thr:
                return NEG
            if score <= self.score_thr:
                return NEG
        return score

    def run(self):
        K, per = self.K, -(-self.K // self.blocks)
        k = 0
        while k < self.M:
            self.steps += 1
            exact = {}

            def ex(i):
                if i not in exact:
                    exact[i] = self.chain(i, self.stale[i], self.begin[i], k)
                    self.ub[i] = exact[i]
                return exact[i]
            # A
            bkeys = []
            for b in range(self.blocks):
                idx = [i for i in range(b * per, min(K, (b + 1) * per)) if self.stale[i] != NEG and self.ub[i] != NEG]
                if not idx:
                    continue
                bi = max(idx, key=lambda i: _key(self.ub[i], i))
                v = ex(bi)
                if v != NEG:
                    bkeys.append(_key(v, bi))
            bkeys.sort(reverse=True)
            w_eff = min(self.W, len(bkeys), self.M - k)
            bound = bkeys[w_eff - 1] if w_eff else None
            # B
            ev = []
            for i in range(K):
                if self.stale[i] == NEG or self.ub[i] == NEG:
                    continue
                if bound is None or _key(self.ub[i], i) >= bound:
                    v = ex(i)
                    if v != NEG:
                        ev.append(_key(v, i))
            if not ev:
                break
            ev.sort(reverse=True)
            top = ev[:max(w_eff, 1)]
            # C
            wins = [top[0]]
            certifiable = self.soft or self.iou_thr >= 0
            for cand in top[1:]:
                ci = -cand[1]
                if not certifiable or cand < bound:
                    break
                if any(_strict_overlap(self.boxes[ci], self.boxes[-w[1]]) for w in wins):
                    break
                wins.append(cand)
            if len(wins) > 1:
                self.multi += 1
            # D
            for t, wkey in enumerate(wins):
                wi = -wkey[1]
                self.sel.append(wi)
                self.sel_score.append(f32(exact[wi]))
                self.stale[wi] = NEG
            widx = {-w[1] for w in wins}
            low = wins[-1]
            for i in range(K):
                s = self.stale[i]
                if s == NEG or i in widx or not _key(s, i) > low:
                    continue
                b = int(self.begin[i])
                for t, wkey in enumerate(wins):
                    if not _key(s, i) > wkey:
                        continue
                    if t == 0 and i in exact:
                        s = exact[i]            # (the kernel's "exact in this epoch" cache: same links, same order)
                    else:
                        s = self.chain(i, s, b, k + t)
                    b = k + t
                    if s == NEG:
                        break
                self.stale[i], self.begin[i], self.ub[i] = s, b, s
            k += len(wins)
        n = len(self.sel)
        idx = np.zeros(self.M, np.int32)
        sc = np.zeros(self.M, f32)
        idx[:n] = self.sel
        sc[:n] = self.sel_score
        return idx, sc, n


def _case(rng, n, regime, span=300.0):
    c = rng.uniform(0, span, (n, 2))
    wh = rng.uniform(4, 90, (n, 2))
    b = np.stack([c[:, 0] - wh[:, 0] / 2, c[:, 1] - wh[:, 1] / 2, c[:, 0] + wh[:, 0] / 2, c[:, 1] + wh[:, 1] / 2], 1).astype(f32)
    if regime == "tied":
        s = (0.01 + rng.normal(0, 1e-4, n)).astype(f32)
        s[rng.integers(0, n, n // 6)] = s[0]
    elif regime == "clusters":       # a few confident objects, many anchors each: consecutive winners overlap
        s = rng.uniform(0.0, 0.05, n).astype(f32)
        for o in range(6):
            m = rng.integers(0, n, 25)
            b[m] = b[m[0]] + rng.normal(0, 3.0, (25, 4)).astype(f32)
            s[m] = rng.uniform(0.5, 0.95, 25).astype(f32)
    else:
        s = rng.uniform(0, 1, n).astype(f32)
    return b, s


CASES = [(0, "spread", 0.25, 0.001, 20), (1, "spread", 0.25, 0.001, 20), (3, "spread", 0.25, 0.001, 20),
         (400, "spread", 0.25, 0.001, 60), (400, "tied", 0.25, 0.001, 60), (400, "clusters", 0.25, 0.001, 40),
         (300, "tied", 0.0, float("-inf"), 50), (300, "spread", 0.0, 0.3, 50), (300, "spread", 0.3, 0.2, 50),
         (500, "tied", 0.5, 0.001, 100)]


@pytest.mark.parametrize("n,regime,sigma,thr,m", CASES)
@pytest.mark.parametrize("winners,blocks", [(1, 8), (4, 8), (4, 3), (8, 12)])
def test_multi_winner_epochs_equal_the_heap(n, regime, sigma, thr, m, winners, blocks):
    rng = np.random.default_rng(n * 7 + len(regime) + int(sigma * 100))
    b, s = _case(rng, n, regime)
    want = post_ref.nms_v5(b, s, m, 0.5, thr, sigma, True)
    mod = EpochModel(b, s, m, 0.5, thr, sigma, blocks=blocks, winners=winners)
    idx, sc, valid = mod.run()
    assert valid == int(want[2])
    np.testing.assert_array_equal(idx, want[0])
    np.testing.assert_array_equal(sc.view(np.uint32), want[1].view(np.uint32))
    if winners == 1:
        assert mod.steps >= valid
    elif n >= 300 and regime != "clusters" and sigma > 0:
        assert mod.steps < valid, "scattered candidates: several winners per super-step"
