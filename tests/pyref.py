"""Independent pure-Python/numpy restatement of the host-side state machines,
used only to cross-check the C oracle in the CPU test-suite (small cases)."""
import math

import numpy as np
from scipy import ndimage


def process_frame(cur, ref, sigma):
    c, r, s6 = cur.astype(np.int32), ref.astype(np.int32), 6 * sigma.astype(np.int32)
    pos = np.clip(np.clip(c - r, 0, 255) - s6, 0, 255)
    neg = np.clip(np.clip(r - c, 0, 255) - s6, 0, 255)
    k = np.array([1, 4, 6, 4, 1])

    def g(a):
        t = ndimage.correlate1d(a, k, axis=1, mode="mirror")
        t = ndimage.correlate1d(t, k, axis=0, mode="mirror")
        return (t + 128) >> 8

    return np.abs(g(pos) - g(neg)).astype(np.uint8)


class Sig:
    """AnalyzerUnit::calculateSignificanceFrame (AnalyzerUnit.cpp:435-504) on histograms."""

    def __init__(self, P, training_set_size):
        self.P = P
        self.tss = training_set_size
        self.pix = [[] for _ in range(256)]
        self.loc_thres = 3

    def __call__(self, h, store):
        sig = 0.0
        rem = self.P
        first, maxadc = -1, 0
        for i in range(256):
            if rem <= 0:
                break
            b = float(np.float32(h[i]))
            if store:
                self.pix[i].append(int(b))
            if i > 1:
                n0 = len(self.pix[0])
                mean = float(sum(self.pix[i])) / n0
                ss = 0.0
                for v in self.pix[i]:
                    sq = (v * v) & 0xFFFFFFFF
                    if sq >= 2 ** 31:
                        sq -= 2 ** 32
                    ss += sq
                var = ss / n0 - mean * mean
                sd = math.sqrt(var) if var >= 0 else float("nan")
                if b != mean or sd > 0:
                    d = b - mean
                    if sd == 0:
                        sig += math.copysign(float("inf"), d) if d != 0 else float("nan")
                    else:
                        sig += d / sd
                if sig < 0:
                    sig = 0.0
            if sig > 3.5 and first < 0:
                first = i
            if i > maxadc:
                maxadc = i
            if store:
                lt = max(first - 1, maxadc - 1)
                if lt < 2:
                    lt = 2
                if lt > 3 or self.tss < 6:
                    lt = 3
                self.loc_thres = lt
            rem -= int(b)
        return sig


def find_trigger(frames, sigma, tss, start=1, sigobj=None, frame_ok=None):
    """AnalyzerUnit::FindTriggerFrame (AnalyzerUnit.cpp:119-324). -> (status, trig, ok, sigobj)"""
    n = len(frames)
    if n < 5:
        return -9, None, False, sigobj
    f32 = np.float32
    thr = f32(3.5)
    two = True
    if tss < 6:
        two = False
        thr = f32(float(thr) * (5 / 3.5))
    if start < 1:
        start = 1
    if start == 1 or sigobj is None:
        sigobj = Sig(frames[0].size, tss)
    prev = start - 1
    prevprev = prev if start < 2 else start - 2
    status, trig = -3, None

    def hist(i, r):
        return np.bincount(process_frame(frames[i], frames[r], sigma).ravel(), minlength=256)

    for i in range(start, n):
        if frame_ok is not None and not frame_ok[i]:
            return -9, None, False, sigobj
        s = f32(sigobj(hist(i, prevprev if two else prev), True))
        if s > thr and i >= 2 and i != n - 1:
            tpp, tp = prev, i
            mx = float(s)
            for ii in (1, 2):
                if i + ii >= n:
                    break
                pk = i + ii
                s = f32(sigobj(hist(pk, tpp if two else tp), False))
                with np.errstate(all="ignore"):
                    val = float(s) / (float(thr) / 3.5 * 5) + (float(s) / mx if mx != 0 else
                                                                (float("nan") if s == 0 else float("inf")))
                if val <= 3:
                    break
                elif ii == 2:
                    status, trig = 0, i
                if float(s) > mx:
                    mx = float(s)
                tpp, tp = tp, pk
            if status == 0:
                break
        prevprev, prev = prev, i
    return status, trig, status == 0, sigobj
