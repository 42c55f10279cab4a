#!/usr/bin/env python
"""Offline study (CPU, numpy/scipy): how often would peak decisions taken on an fp32 forward FFT differ from the fp64 ones?
Input to DESIGN.md section 7 (fp32-first forward FFT with an amplitude-domain guard band and fp64 fallback).  Not part of the product."""
import numpy as np
import scipy.fft as sf

N, h = 1024, 256
rng = np.random.default_rng(0)


def frames(x):
    T = (len(x) - N) // h
    w = (0.5 * (1 - np.cos(2 * np.pi * np.arange(N) / N))).astype(np.float32)
    idx = np.arange(N)[None, :] + h * np.arange(T)[:, None]
    return (x[idx] * w).astype(np.float32)


def peaks(m):
    c = m[:, 2:-2]
    return (c > m[:, 1:-3]) & (c > m[:, :-4]) & (c > m[:, 3:-1]) & (c > m[:, 4:])


def study(x, name, kappa=16.0):
    F = frames(x.astype(np.float32))
    X64 = np.fft.rfft(F.astype(np.float64), axis=1)
    X32 = sf.rfft(F, axis=1)                                   # single precision
    m64 = (X64.real ** 2 + X64.imag ** 2).astype(np.float32)
    m32 = X32.real.astype(np.float32) ** 2 + X32.imag.astype(np.float32) ** 2
    diff = peaks(m64) != peaks(m32)
    rel = m64[:, 2:-2] > 1e-11 * m64.max(axis=1, keepdims=True)  # bins above -110 dB of the frame maximum
    # amplitude-domain guard band: |A_c - A_n| <= G = 2E, E = kappa * eps32 * rms(|X|)  (c - n = (A_c - A_n)(A_c + A_n))
    A = np.sqrt(m32.astype(np.float64))
    E = kappa * 6e-8 * np.sqrt((A ** 2).mean(axis=1, keepdims=True))
    c = A[:, 2:-2]
    amb = np.zeros_like(c, dtype=bool)
    for n in (A[:, 1:-3], A[:, :-4], A[:, 3:-1], A[:, 4:]):
        amb |= np.abs(c - n) <= 2 * E
    # a comparison only matters if the bin could be a peak: all other comparisons not clearly lost
    could = np.ones_like(c, dtype=bool)
    for n in (A[:, 1:-3], A[:, :-4], A[:, 3:-1], A[:, 4:]):
        could &= c - n > -2 * E
    fallback = (amb & could & rel).any(axis=1)
    missed = (diff & rel & ~(amb & could)).any(axis=1)
    err = np.abs(X32 - X64)
    print(f"{name:26s} frames {len(F)}  decisions differ (relevant bins) {100 * (diff & rel).any(axis=1).mean():6.2f} %   "
          f"guard band would fall back {100 * fallback.mean():6.2f} %   flips NOT caught {100 * missed.mean():.2f} %   "
          f"fp32 FFT error {np.sqrt((err ** 2).mean()) / np.sqrt((np.abs(X64) ** 2).mean()):.2e} of rms|X|")


n = 256 * 4000 + 1024
i = np.arange(n)
bench = 0.25 * np.sin(i * 2 * np.pi * 220 / 48000) + 0.125 * np.sin(i * 2 * np.pi * 1375 / 48000) + 0.0625 * np.sin(i * 2 * np.pi * 6857 / 48000) \
    + (rng.uniform(0, 1, n) - 0.5) * (2.0 / 64)
study(bench, "bench.py synth_input")
study(rng.uniform(-0.5, 0.5, n), "white noise")
for db in (60, 80, 100):
    study(0.5 * np.sin(2 * np.pi * i * 0.0123) + 0.3 * np.sin(2 * np.pi * i * 0.0931) + rng.uniform(-1, 1, n) * 10 ** (-db / 20), f"2 sines + noise -{db} dB")
