"""numpy model of the algorithm the HIP kernels implement (phaze_amd/csrc/pv_kernels.hip).

TEST INFRASTRUCTURE.  The GPU path does NOT walk the reference's loops; it restructures them:

  * forward real FFT via one packed N/2-point complex FFT (fp64) + split post-pass, instead of fft.js's
    in-place real radix-4 DIT (bundle:306-442);
  * the above-Nyquist "residue" that fft.js leaves in bins N/2+1..N-1 is rebuilt
    separately, only when a frame can read it, by re-running fft.js's stage structure on the upper half
    of the buffer only (quarters 2 and 3 of the top block), in fp32;
  * findPeaks as a purely local predicate, the region-of-influence walk of shiftPeaks (pv:119-173) as a
    per-source-bin owner rule (nearest peaks left/right + midpoint rule), the `break`s as predicates;
  * the phase rotation cos/sin(omega*t) as an exact N-th root of unity, index (delta*t) mod N;
  * the inverse as a packed N/2-point complex FFT (c2r), fp32;
  * overlap-add in reference order.

This model exists so the restructuring itself is proven equivalent to the reference on CPU
(tests/test_gpu_algo_model.py compares it with the golden vectors); the kernels then follow it.
"""
import numpy as np


def hann(N):
    i = np.arange(N, dtype=np.float64)
    return (0.5 * (1 - np.cos(2 * np.pi * i / N))).astype(np.float32)


def digit_reverse4(t, ndigits):
    r = 0
    for _ in range(ndigits):
        r = (r << 2) | (t & 3)
        t >>= 2
    return r


def forward_packed(xw):
    """fp64 spectrum X[0..N/2] of real xw[N] from one N/2-point complex FFT."""
    N = xw.shape[0]
    z = xw[0::2].astype(np.float64) + 1j * xw[1::2].astype(np.float64)
    Z = np.fft.fft(z)
    k = np.arange(N // 2 + 1)
    Zk = Z[k % (N // 2)]
    Zc = np.conj(Z[(N // 2 - k) % (N // 2)])
    W = np.exp(-2j * np.pi * k / N)
    X = 0.5 * ((Zk + Zc) - 1j * W * (Zk - Zc))
    X[0] = Z[0].real + Z[0].imag
    X[N // 2] = Z[0].real - Z[0].imag
    return X


def residue_upper(xw, dtype=np.complex64):
    """Content fft.js's realTransform leaves at positions N/2+1..N-1 (never the conjugate mirror).
    Re-runs the reference's stage structure on blocks inside [N/2, N) only.  Returns array[N] (lower half junk)."""
    N = xw.shape[0]
    power = int(np.log2(N))
    buf = np.zeros(N, dtype=dtype)
    x = xw.astype(np.float64 if dtype == np.complex128 else np.float32)
    if power % 2 == 0:
        base, nd = 4, (power - 2) // 2
        for t in range(N // 8, N // 4):                      # blocks with 4t >= N/2
            off = digit_reverse4(t, nd)
            a, b, c, d = x[off], x[off + N // 4], x[off + N // 2], x[off + 3 * N // 4]
            t0, t1, t2, t3 = a + c, a - c, b + d, b - d
            buf[4 * t:4 * t + 4] = [t0 + t2, t1 - 1j * t3, t0 - t2, t1 + 1j * t3]
    else:
        base, nd = 2, (power - 1) // 2
        for t in range(N // 4, N // 2):
            off = digit_reverse4(t, nd)
            a, b = x[off], x[off + N // 2]
            buf[2 * t:2 * t + 2] = [a + b, a - b]
    M = base * 4
    while M <= N // 4:
        q = M // 4
        w = np.exp(-2j * np.pi * np.arange(0, q // 2 + 1) / M).astype(dtype)
        for o in range(N // 2, N, M):
            i = np.arange(0, q // 2 + 1)
            A, B, C, D = buf[o + i], buf[o + q + i] * w, buf[o + 2 * q + i] * w * w, buf[o + 3 * q + i] * w * w * w
            T0, T1, T2, T3 = A + C, A - C, B + D, B - D
            FA, FB = T0 + T2, T1 - 1j * T3
            FC0 = T0[0] - T2[0]
            # mirrored outputs: out[q-i] = conj(T1 + j T3)  ,  out[2q-i] = conj(T0 - T2)      (bundle:412-440)
            SA, SB = np.conj(T1 + 1j * T3), np.conj(T0 - T2)
            buf[o + i] = FA
            buf[o + q + i] = FB
            buf[o + 2 * q] = FC0
            inner = i[1:-1] if len(i) > 2 else i[0:0]
            buf[o + q - inner] = SA[1:-1] if len(i) > 2 else SA[0:0]
            buf[o + 2 * q - inner] = SB[1:-1] if len(i) > 2 else SB[0:0]
        M *= 4
    return buf


def js_round(x):
    r = np.floor(x)
    return np.where(x - r >= 0.5, r + 1.0, r)


def shift_spectrum(Xlow, res_upper, pitch_f32, t, N):
    """findPeaks + shiftPeaks as per-bin parallel rules.  Xlow: complex128[N/2+1]; res_upper: complex[N] or None."""
    H = N // 2 + 1
    mag = (Xlow.real * Xlow.real + Xlow.imag * Xlow.imag).astype(np.float32)
    k = np.arange(H)
    flag = np.zeros(H, dtype=bool)
    kk = k[2:H - 2]
    flag[2:H - 2] = (mag[kk] > mag[kk - 1]) & (mag[kk] > mag[kk - 2]) & (mag[kk] > mag[kk + 1]) & (mag[kk] > mag[kk + 2])
    Y = np.zeros(H, dtype=np.complex64)
    if not flag.any():
        return Y, mag, np.nonzero(flag)[0]
    # nearest peak at or below / strictly above each source bin b in [0, N)
    b = np.arange(N)
    idx = np.where(flag, k, -1)
    prev_low = np.maximum.accumulate(idx)
    nxt_low = np.where(flag, k, 1 << 30)
    nxt_low = np.minimum.accumulate(nxt_low[::-1])[::-1]
    nxt_strict = np.concatenate([nxt_low[1:], [1 << 30]])           # smallest peak > b
    last_peak = prev_low[-1]
    prev = np.concatenate([prev_low, np.full(N - H, last_peak)])
    nxt = np.concatenate([nxt_strict, np.full(N - H, 1 << 30)])
    has_prev, has_next = prev >= 0, nxt < (1 << 30)
    boundary = prev + (nxt - prev + 1) // 2
    owner = np.where(~has_prev, nxt, np.where(~has_next, prev, np.where(b < boundary, prev, nxt)))
    pf = np.float64(np.float32(pitch_f32))
    with np.errstate(invalid="ignore", over="ignore"):
        psh = js_round(owner.astype(np.float64) * pf)
        active = ~(psh > H) & np.isfinite(psh)
        delta = np.where(active, psh - owner, 0.0)
        tgt = b + delta
        ok = active & (tgt >= 0) & (tgt < H)
    src = np.concatenate([Xlow, np.zeros(N - H, dtype=np.complex128)]).astype(np.complex64)
    if res_upper is not None:
        src[H:] = res_upper[H:]
    d = delta[ok].astype(np.int64)
    ridx = (d % N) * (int(t) % N) % N
    rot = np.exp(2j * np.pi * ridx / N)
    np.add.at(Y, tgt[ok].astype(np.int64), (src[ok] * rot).astype(np.complex64))
    return Y, mag, np.nonzero(flag)[0]


def inverse_packed(Y, N):
    """Real part of the N-point inverse DFT of the Hermitian completion of Y[0..N/2] (c2r), via N/2-point complex FFT."""
    h = N // 2
    k = np.arange(h)
    Yk = Y[k].astype(np.complex128)
    Yc = np.conj(Y[h - k].astype(np.complex128))
    Yk[0] = Y[0].real                       # Im(Y[0]) and Im(Y[N/2]) never reach the real part
    Yc[0] = Y[h].real
    Z = (Yk + Yc) + 1j * np.exp(2j * np.pi * k / N) * (Yk - Yc)
    z = np.fft.ifft(Z) * h / N              # (1/N) * sum
    out = np.empty(N, dtype=np.float64)
    out[0::2], out[1::2] = z.real, z.imag
    return out.astype(np.float32)


class Model:
    def __init__(self, N, hop):
        self.N, self.hop, self.R = N, hop, N // hop
        self.w = hann(N)
        self.hist = np.zeros(N, dtype=np.float32)
        self.acc = np.zeros(N, dtype=np.float32)
        self.m = 0
        self.last = None

    def process(self, block, pitch):
        N, h = self.N, self.hop
        self.hist = np.concatenate([self.hist[h:], np.asarray(block, np.float32)])
        xw = self.hist * self.w
        X = forward_packed(xw)
        pf = np.float32(pitch)
        res = residue_upper(xw) if (pf < 1 or not np.isfinite(pf)) else None
        Y, mag, peaks = shift_spectrum(X, res, pf, self.m * h, N)
        fr = inverse_packed(Y, N) * self.w
        self.acc = self.acc + fr / np.float32(self.R)
        out = self.acc[:h].copy()
        self.acc = np.concatenate([self.acc[h:], np.zeros(h, np.float32)])
        self.m += 1
        self.last = dict(X=X, res=res, Y=Y, mag=mag, peaks=peaks)
        return out
