"""CPU oracle for the reference-clip ingest (test infrastructure only: tests/, smoke() and bench.py's cpu_baseline may
import this; the product never does).

Literal restatement, as plain Python loops, of the two stdlib routines pydub calls inside the reference's
AudioProcessor.load_audio (/root/reference/vietvoicetts/core/audio_processor.py:15-26 ->
AudioSegment.set_channels(1).set_frame_rate(sr)):

  tomono  -- CPython Modules/audioop.c audioop_tomono_impl + fbound(): val = l * lfactor + r * rfactor as double, clamped,
             floor()ed, cast to int
  ratecv  -- CPython Modules/audioop.c audioop_ratecv_impl with state None and weights (1, 0): the running counter d starts
             at -outrate, one input frame is consumed while d < 0 (d += outrate), one output frame is emitted while d >= 0
             (cur_o = (int)((prev * d + cur * (outrate - d)) / outrate) on samples shifted to 32 bit; d -= inrate)
  np.mean -- numpy's float32 add.reduce order: buffers of 8192 elements, each summed pairwise (8 interleaved accumulators on
             leaves of <= 128 elements, split at (n / 2) - (n / 2) % 8), buffer sums accumulated in order

Pinned: tests/test_ingest_cpu.py checks these loops against stdlib ``audioop`` / ``numpy.mean`` run live and against
tests/golden/ingest_golden.npz (produced by running the reference's own load_audio over stdlib audioop).  The product's
closed-form numpy path (vietvoice-tts_amd/core/audio_processor.py) and the gfx950 kernels (csrc/vv_ingest.hip) are checked
against the same fixtures; this file exists so that the counter-based state machine is written down once, independently of
the closed form the product evaluates.
"""
from math import floor, gcd

import numpy as np


def tomono(left, right, width, lfactor=0.5, rfactor=0.5):
    maxval = float((1 << (8 * width - 1)) - 1)
    minval = -float(1 << (8 * width - 1))
    out = []
    for l, r in zip(left, right):
        val = float(l) * lfactor + float(r) * rfactor
        if val > maxval:
            val = maxval
        elif val < minval + 1.0:
            val = minval
        out.append(int(floor(val)))
    return out


def ratecv(samples, width, inrate, outrate):
    """Mono; state None; weightA 1, weightB 0."""
    g = gcd(inrate, outrate)
    inrate, outrate = inrate // g, outrate // g
    shift = 32 - 8 * width
    d, prev_i, cur_i, out, pos, n = -outrate, 0, 0, [], 0, len(samples)
    while True:
        while d < 0:
            if pos == n:
                return out
            prev_i = cur_i
            cur_i = int(samples[pos]) << shift
            pos += 1
            cur_i = int((1.0 * float(cur_i) + 0.0 * float(prev_i)) / (1.0 + 0.0))
            d += outrate
        while d >= 0:
            cur_o = int((float(prev_i) * float(d) + float(cur_i) * float(outrate - d)) / float(outrate))      # C cast: truncation
            out.append(cur_o >> shift)
            d -= inrate


def _pairwise(a, lo, n):
    f = np.float32
    if n < 8:
        r = f(0.0)
        for i in range(n):
            r = f(r + a[lo + i])
        return r
    if n <= 128:
        r = [f(a[lo + j]) for j in range(8)]
        i = 8
        while i < n - n % 8:
            for j in range(8):
                r[j] = f(r[j] + a[lo + i + j])
            i += 8
        res = f(f(f(r[0] + r[1]) + f(r[2] + r[3])) + f(f(r[4] + r[5]) + f(r[6] + r[7])))
        while i < n:
            res = f(res + a[lo + i])
            i += 1
        return res
    n2 = n // 2
    n2 -= n2 % 8
    return f(_pairwise(a, lo, n2) + _pairwise(a, lo + n2, n - n2))


def float32_mean(a, bufsize=8192):
    a = np.asarray(a, dtype=np.float32)
    s = np.float32(0.0)
    for lo in range(0, len(a), bufsize):
        s = np.float32(s + _pairwise(a, lo, min(bufsize, len(a) - lo)))
    return np.float32(np.float64(s) / len(a))
