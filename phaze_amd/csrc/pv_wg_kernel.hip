// pv_wg_kernel.hip -- register-resident frame pipeline for N = 2048, 4096, 8192 with EIGHT elements per thread.
//
// Its role since round 4: the hops below N/8 that go through an overlap-add ring in LDS (the reference's 2048/128 when the one-wave kernel is switched off, 4096/256 ...),
// and -- under PV_FLAG_WORKGROUP_KERNEL -- the second implementation of the path the tests run every stream of N = 2048 / 4096 / 8192 through.  BASELINE's shapes run on
// pv_wave2k_kernel.hip (N = 2048) and pv_wg16_kernel.hip (N = 4096, 8192: sixteen elements per thread, two / four workgroups per CU); this kernel ran N = 8192 until then.
//
// Generalisation of pv_wave_kernel_1024: one frame is held by G = N/1024 wavefronts (T = 64 G threads = one workgroup = one
// frame chain), 8 packed complex elements per thread:
//
//   thread t, register r  <->  element t + T r   of   z[n] = xw[2n] + j xw[2n+1],   n in [0, M),  M = N/2 = 512 G = 8*8*8*G
//
// FFT = four in-register passes (radix 8, 8, 8, G) with three register<->thread transposes through LDS (row-padded layouts from
// tools/lds_layout_check.py):
//   pass A over r            -> digit kA (weight 1),   twiddle W_M^{t kA}
//   pass B over t_hi         -> digit kB (weight 8),   twiddle W_T^{t_lo kB}          t  = t_hi * 8G + t_lo
//   pass C over u_hi         -> digit kC (weight 64),  twiddle W_{8G}^{u_lo kC}       t_lo = u_hi * G + u_lo
//   pass D over u_lo (radix G) -> digit kD (weight 512)
// and the result lands again as thread t', register r' <-> bin t' + T r'.  Everything between the two FFTs (split pass, |X|^2, peak
// flags on 8 consecutive bins per thread, routes, scatter -- for f < 1: store / barrier / add on pairwise frames, else claim rounds --,
// per-quarter residue, c2r pre-pass) and the
// register-resident overlap-add follow the wave kernel; exchanges that were wave-local there (bpermute, ballot) go through LDS here.
// Reference citations are those of pv_kernels.hip / pv_wave_kernel.hip.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "pv_kernels.h"
#include "pv_device_common.h"
#include "pv_pk_math.h"
#ifdef PV_WG_STAMPS   // measurement build (make variant ... EXTRA=-DPV_WG_STAMPS CAPI_EXTRA=-DPV_STAMPS=1): s_memtime at the stations of a 1-hop launch, workgroup (0, ch)
#define WG_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); if (threadIdx.x == 0 && p.stamps) p.stamps[16 * blockIdx.y + (i)] = (unsigned)t_; } while (0)
#else
#define WG_STAMP(i)
#endif
#ifndef PV_PAIRWISE
#define PV_PAIRWISE 1                               // 0: every f < 1 frame goes through the claim rounds (A/B)
#endif

namespace {

// CT (compact twiddles, ring kernels): pass-A twiddle rows k = 1, 2, 4 only, the other four by products -- 8 KB per workgroup at G = 2, which is
// what lets the native 2048/128 configuration keep 4 workgroups per CU next to its overlap-add ring
#ifndef PV_WG_REG_T2
#define PV_WG_REG_T2 1        // G = 8: transpose 2 of both FFTs in registers (0 = the round-2 LDS form, for A/B builds)
#endif
#ifndef PV_WG_INV_SPLIT
#define PV_WG_INV_SPLIT 1     // G = 8: transposes 1 and 3 of the inverse FFT in separate halves of the scratch, 4 barriers less per frame (0 = round-3 form, A/B)
#endif
template <int G, bool CT = false>
struct WgCfg {
    static constexpr int T = 64 * G, M = 512 * G, N = 1024 * G, H = M + 1;
    // transpose 2: per-kA row pad, row length.  Row pad 4 (G = 2, 4) / 8 (G = 8) measured conflict-free for the writes AND the reads of both
    // element sizes (tools/lds_microbench.hip; the earlier row pad 8 cost the reads 6.4 instead of 3.4 LDS cycles)
    static constexpr int A2 = (G == 2) ? 2 : 0, P2 = 8 * (8 * G + A2) + (G == 8 ? 8 : 4);
    static constexpr int A3 = (G == 2) ? 4 : 1, P3 = 8 * (8 * G + A3);          // transpose 3
    static constexpr int P1 = T;                                                 // transpose 1 is conflict-free unpadded
    static constexpr int PMAX = (P2 > P3 ? (P2 > P1 ? P2 : P1) : (P3 > P1 ? P3 : P1));
    static constexpr int SCRATCH = 8 * PMAX * 16;                                // bytes (fp64 complex)
    // inside the scratch, between the two FFTs:
    static constexpr int OFF_Y = 0;                                              // float2[H]
    static constexpr int OFF_ROUTE = ((8 * H + 15) / 16) * 16;                   // u32[M + 16] routes | f32 mags | u32 claim words (aliases)
    static constexpr int OFF_RESQ = OFF_ROUTE + 4 * (M + 16);                    // float2[N / 4] one residue quarter
    static_assert(OFF_RESQ + 8 * (N / 4) <= SCRATCH, "scratch too small");
    // after the scratch:
    static constexpr int OFF_PSH = SCRATCH;                                      // i16[M]
    static constexpr int OFF_NEAR = OFF_PSH + 2 * M;                             // i32 LASTIN[T], FIRSTIN[T]
    static constexpr int OFF_OCC = OFF_NEAR + 8 * T;                             // u64[G]
    // twiddle rows k = 1..7 only (row 0 is all ones): keeps G = 2 at 4 workgroups per CU
    static constexpr int OFF_TWA = ((OFF_OCC + 8 * G + 15) / 16) * 16;           // double2[7][T]
    static constexpr int TWA_ROWS = CT ? 3 : 7;
    static constexpr int OFF_TWB = OFF_TWA + 16 * TWA_ROWS * T;                  // double2[7][8G]
    static constexpr int OFF_TWC = OFF_TWB + 16 * 7 * 8 * G;                     // double2[7][G]
    static constexpr int OFF_S2W = OFF_TWC + 16 * 7 * G;                         // float2[2][T] conj(W^{2k}) of the fast residue's bins k = 1 + t + T j
    static constexpr int OFF_XQ = OFF_S2W + 8 * 2 * T;                           // f32[N/4] windowed input samples xw[4n + 2] of the current frame (f < 0.75 only): base stage of the general residue
    static constexpr int LDS_BYTES = OFF_XQ + N;
    static constexpr int OFF_ACC = LDS_BYTES;                                    // f32[N - hop] overlap-add ring, only for hops below N/8 (S_ROWS = 0)
    static constexpr int LDS_BYTES_RING = OFF_ACC + 4 * N;
};

template <typename T_, bool INV>
__device__ __forceinline__ typename v2t<T_>::type twc(double2 w) { return typename v2t<T_>::type{(T_)w.x, INV ? (T_)(-w.y) : (T_)w.y}; }

// M-point complex FFT across the T threads of the workgroup: in/out layout thread t, reg r <-> element t + T r.
template <typename T_, bool INV, int G, bool CT>
__device__ __forceinline__ void fft_wg(typename v2t<T_>::type (&a)[8], typename v2t<T_>::type *S, const double2 *TWA, const double2 *TWB,
                                       const double2 *TWC, int t)
{
    using C = WgCfg<G>;
    using T2 = typename v2t<T_>::type;
    // ---- pass A ----
    radix8<T_, INV>(a);
#pragma unroll
    for (int k = 1; k < 8; k++) {
        if (!CT) {
            a[k] = cmul(a[k], twc<T_, INV>(TWA[(k - 1) * C::T + t]));
        } else {                                                           // rows W^1, W^2, W^4; the rest by products
            const T2 w1 = twc<T_, INV>(TWA[t]), w2 = twc<T_, INV>(TWA[C::T + t]), w4 = twc<T_, INV>(TWA[2 * C::T + t]);
            const T2 w = (k == 1) ? w1 : (k == 2) ? w2 : (k == 3) ? cmul(w1, w2) : (k == 4) ? w4 : (k == 5) ? cmul(w4, w1) : (k == 6) ? cmul(w4, w2) : cmul(w4, cmul(w1, w2));
            a[k] = cmul(a[k], w);
        }
    }
#pragma unroll
    for (int k = 0; k < 8; k++) S[k * C::P1 + t] = a[k];
    __syncthreads();
    const int kA1 = t / (8 * G), tlo = t % (8 * G);
#pragma unroll
    for (int n = 0; n < 8; n++) a[n] = S[kA1 * C::P1 + n * 8 * G + tlo];
    __syncthreads();
    // ---- pass B ----
    radix8<T_, INV>(a);
#pragma unroll
    for (int k = 1; k < 8; k++) a[k] = cmul(a[k], twc<T_, INV>(TWB[(k - 1) * 8 * G + tlo]));
    // transpose 2 stays inside the group of 8G <= 64 threads that share kA1, i.e. inside ONE wave.
    const int kB2 = tlo / G, ulo = tlo % G;               // destination role of this thread: (kA1, kB2, ulo)
    if (G == 8 && PV_WG_REG_T2 && sizeof(T_) == 8) {
        // G = 8: the exchange is register index <-> lane bits 5..3 of a full wave -- the register transpose of the wave FFTs (round 3):
        // no LDS traffic (64 KB written + read per transform before), and no barrier in front of transpose 3 (nothing of this one is in LDS)
        unsigned w[8][4];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint2 x = __builtin_bit_cast(uint2, (double)a[k].x), y = __builtin_bit_cast(uint2, (double)a[k].y);
            w[k][0] = x.x; w[k][1] = x.y; w[k][2] = y.x; w[k][3] = y.y;
        }
        transpose_hi3_regs<4>(w);
#pragma unroll
        for (int k = 0; k < 8; k++) {
            a[k].x = (T_)__builtin_bit_cast(double, uint2{w[k][0], w[k][1]});
            a[k].y = (T_)__builtin_bit_cast(double, uint2{w[k][2], w[k][3]});
        }
    } else {
#pragma unroll
        for (int k = 0; k < 8; k++) S[k * C::P2 + kA1 * (8 * G + C::A2) + tlo] = a[k];
        // the LDS traffic of a wave executes in order, so a compiler fence replaces the two workgroup barriers (4 of the ~21 barriers of a frame)
        wave_sync();
#pragma unroll
        for (int n = 0; n < 8; n++) a[n] = S[kB2 * C::P2 + kA1 * (8 * G + C::A2) + n * G + ulo];
        __syncthreads();                                  // the next transpose writes rows other waves still read here
    }
    // ---- pass C ----
    radix8<T_, INV>(a);
#pragma unroll
    for (int k = 1; k < 8; k++) a[k] = cmul(a[k], twc<T_, INV>(TWC[(k - 1) * G + ulo]));
#pragma unroll
    for (int k = 0; k < 8; k++) S[k * C::P3 + kA1 * (8 * G + C::A3) + kB2 * G + ulo] = a[k];
    __syncthreads();
    const int kA3 = t & 7, kB3 = (t >> 3) & 7, c3 = t >> 6;  // destination role: bin low part kA3 + 8 kB3 + 64 c3
#pragma unroll
    for (int q = 0; q < 8; q++) a[q] = S[(c3 + G * (q / G)) * C::P3 + kA3 * (8 * G + C::A3) + kB3 * G + (q % G)];
    __syncthreads();
    // ---- pass D: (8/G) independent radix-G DFTs over u_lo; output register j + (8/G) kD ----
    if (G == 8) {
        radix8<T_, INV>(a);
    } else if (G == 4) {
        T2 o[8];
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const T2 e0 = cadd(a[4 * j], a[4 * j + 2]), e1 = csub(a[4 * j], a[4 * j + 2]);
            const T2 e2 = cadd(a[4 * j + 1], a[4 * j + 3]), e3 = rot90<INV>(csub(a[4 * j + 1], a[4 * j + 3]));
            o[j] = cadd(e0, e2); o[j + 2] = cadd(e1, e3); o[j + 4] = csub(e0, e2); o[j + 6] = csub(e1, e3);
        }
#pragma unroll
        for (int q = 0; q < 8; q++) a[q] = o[q];
    } else {
        T2 o[8];
#pragma unroll
        for (int j = 0; j < 4; j++) { o[j] = cadd(a[2 * j], a[2 * j + 1]); o[j + 4] = csub(a[2 * j], a[2 * j + 1]); }
#pragma unroll
        for (int q = 0; q < 8; q++) a[q] = o[q];
    }
}

// The inverse instance in packed fp32 (pv_pk_math.h): same layouts as fft_wg<float, true, G>, twiddles conjugated and rounded from the fp64 tables.
__device__ __forceinline__ pk::c32 twc_inv_pk(double2 w) { return pk::c32{(float)w.x, -(float)w.y}; }

// G = 8 (round 4, PV_WG_INV_SPLIT): the packed-fp32 elements are half the size of the forward's, so transposes 1 and 3 each get their OWN half of the
// scratch (transpose 2 runs in registers) and the second barrier of both exchanges goes: a transpose only has to wait until its data has been
// WRITTEN, the readers of the other half are not in its way.  With the c2r hand-over in a third place (OFF_RESQ: read before transpose 1's
// barrier, overwritten by transpose 3 after it) an inverse costs 3 workgroup barriers + the end-of-frame one instead of 7.
template <int G, bool CT>
__device__ __forceinline__ void fft_wg_inv_pk(pk::c32 (&a)[8], pk::c32 *S, const double2 *TWA, const double2 *TWB, const double2 *TWC, int t)
{
    using C = WgCfg<G>;
    constexpr bool SPLIT = (G == 8) && PV_WG_REG_T2 && PV_WG_INV_SPLIT;
    static_assert(!SPLIT || (8 * C::P1 * 8 + 8 * C::P3 * 8 <= C::SCRATCH), "the two halves of the inverse's scratch do not fit");
    pk::radix8_inv(a);
#pragma unroll
    for (int k = 1; k < 8; k++) {
        if (!CT) {
            a[k] = pk::cmul(a[k], twc_inv_pk(TWA[(k - 1) * C::T + t]));
        } else {
            const pk::c32 w1 = twc_inv_pk(TWA[t]), w2 = twc_inv_pk(TWA[C::T + t]), w4 = twc_inv_pk(TWA[2 * C::T + t]);
            const pk::c32 w = (k == 1) ? w1 : (k == 2) ? w2 : (k == 3) ? pk::cmul(w1, w2) : (k == 4) ? w4 : (k == 5) ? pk::cmul(w4, w1) : (k == 6) ? pk::cmul(w4, w2)
                                                                                                                               : pk::cmul(w4, pk::cmul(w1, w2));
            a[k] = pk::cmul(a[k], w);
        }
    }
#pragma unroll
    for (int k = 0; k < 8; k++) S[k * C::P1 + t] = a[k];
    __syncthreads();
    const int kA1 = t / (8 * G), tlo = t % (8 * G);
#pragma unroll
    for (int n = 0; n < 8; n++) a[n] = S[kA1 * C::P1 + n * 8 * G + tlo];
    if (!SPLIT) __syncthreads();
    pk::radix8_inv(a);
#pragma unroll
    for (int k = 1; k < 8; k++) a[k] = pk::cmul(a[k], twc_inv_pk(TWB[(k - 1) * 8 * G + tlo]));
    const int kB2 = tlo / G, ulo = tlo % G;
    if (G == 8 && PV_WG_REG_T2) {                         // register transpose, no barrier (see fft_wg)
        unsigned w[8][2];
#pragma unroll
        for (int k = 0; k < 8; k++) { w[k][0] = __float_as_uint(a[k].x); w[k][1] = __float_as_uint(a[k].y); }
        transpose_hi3_regs<2>(w);
#pragma unroll
        for (int k = 0; k < 8; k++) a[k] = pk::c32{__uint_as_float(w[k][0]), __uint_as_float(w[k][1])};
    } else {
#pragma unroll
        for (int k = 0; k < 8; k++) S[k * C::P2 + kA1 * (8 * G + C::A2) + tlo] = a[k];
        wave_sync();                                      // wave-local exchange (see fft_wg)
#pragma unroll
        for (int n = 0; n < 8; n++) a[n] = S[kB2 * C::P2 + kA1 * (8 * G + C::A2) + n * G + ulo];
        __syncthreads();
    }
    pk::radix8_inv(a);
#pragma unroll
    for (int k = 1; k < 8; k++) a[k] = pk::cmul(a[k], twc_inv_pk(TWC[(k - 1) * G + ulo]));
    pk::c32 *S3 = SPLIT ? S + 8 * C::P1 : S;                // SPLIT: the half behind transpose 1's rows
#pragma unroll
    for (int k = 0; k < 8; k++) S3[k * C::P3 + kA1 * (8 * G + C::A3) + kB2 * G + ulo] = a[k];
    __syncthreads();
    const int kA3 = t & 7, kB3 = (t >> 3) & 7, c3 = t >> 6;
#pragma unroll
    for (int q = 0; q < 8; q++) a[q] = S3[(c3 + G * (q / G)) * C::P3 + kA3 * (8 * G + C::A3) + kB3 * G + (q % G)];
    if (!SPLIT) __syncthreads();                            // (SPLIT: the end-of-frame barrier stands between these reads and the next frame's writes)
    if (G == 8) {
        pk::radix8_inv(a);
    } else if (G == 4) {
        pk::c32 o[8];
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const pk::c32 e0 = pk::add(a[4 * j], a[4 * j + 2]), e1 = pk::sub(a[4 * j], a[4 * j + 2]);
            const pk::c32 e2 = pk::add(a[4 * j + 1], a[4 * j + 3]), d = pk::sub(a[4 * j + 1], a[4 * j + 3]);
            o[j] = pk::add(e0, e2); o[j + 2] = pk::add_j(e1, d); o[j + 4] = pk::sub(e0, e2); o[j + 6] = pk::sub_j(e1, d);
        }
#pragma unroll
        for (int q = 0; q < 8; q++) a[q] = o[q];
    } else {
        pk::c32 o[8];
#pragma unroll
        for (int j = 0; j < 4; j++) { o[j] = pk::add(a[2 * j], a[2 * j + 1]); o[j + 4] = pk::sub(a[2 * j], a[2 * j + 1]); }
#pragma unroll
        for (int q = 0; q < 8; q++) a[q] = o[q];
    }
}

// o * exp(+2 pi j r / 16), r = 0..3 (compile-time): the wave-uniform part of the c2r twiddle, packed
__device__ __forceinline__ pk::c32 mul_w16_inv_pk_wg(pk::c32 o, int r)
{
    const float c = 0.92387953251128675613f, sn = 0.38268343236508977173f, h = 0.70710678118654752440f;
    switch (r) {
    case 0: return o;
    case 1: return pk::cmul(o, pk::c32{c, sn});
    case 2: return pk::mul(pk::add_j(o, o), pk::c32{h, h});
    default: return pk::cmul(o, pk::c32{sn, c});
    }
}

// Workgroup-wide claim rounds (see claim_rounds in pv_wave_kernel.hip): the loop condition is reduced over the workgroup, and -- because
// several waves race for a claim word here -- a source posts its bin with an LDS atomic MIN: the smallest pending source bin wins the round,
// so every target accumulates its contributions in ascending source order (the order of the reference's loops, pv:122,146) whatever the
// timing of the waves; results are reproducible bit for bit for every f.  CLAIM[0..H) must be all-ones on entry and is all-ones on exit.
template <int NS, int H_>
__device__ __forceinline__ void claim_rounds_wg(const unsigned (&rt)[NS], const float2 (&ys)[NS], const int (&id)[NS], float2 *Y, unsigned *CLAIM)
{
    unsigned pend = 0;
    unsigned tg[NS];
#pragma unroll
    for (int r = 0; r < NS; r++) {
        const unsigned t = rt[r] & 0xFFFFu;
        const bool ok = t < (unsigned)H_;                                  // valid route <=> target field < H
        pend |= ok ? (1u << r) : 0u;
        tg[r] = ok ? t : 0u;
    }
    while (__syncthreads_or(pend != 0u)) {
#pragma unroll
        for (int r = 0; r < NS; r++) if (pend & (1u << r)) atomicMin(&CLAIM[tg[r]], (unsigned)id[r]);
        __syncthreads();
        unsigned c[NS];
        float2 o[NS];
#pragma unroll
        for (int r = 0; r < NS; r++) c[r] = CLAIM[tg[r]];                   // independent reads first, then the winners' stores (see pv_wave_kernel.hip)
#pragma unroll
        for (int r = 0; r < NS; r++) o[r] = Y[tg[r]];
#pragma unroll
        for (int r = 0; r < NS; r++) {
            if ((pend & (1u << r)) && c[r] == (unsigned)id[r]) {
                Y[tg[r]] = float2{o[r].x + ys[r].x, o[r].y + ys[r].y};
                CLAIM[tg[r]] = 0xFFFFFFFFu;                                // only the winner touches the word; losers re-post after the barrier
                pend &= ~(1u << r);
            }
        }
    }
}

__device__ __forceinline__ int digitrev4_(int v, int nd)
{
    if (nd == 0) return 0;
    const unsigned r = __brev((unsigned)v) >> (32 - 2 * nd);
    return (int)(((r & 0x55555555u) << 1) | ((r >> 1) & 0x55555555u));
}

// Rare path: above-Nyquist residue of fft.js's in-place real DIT, one quarter of the buffer at a time, then its sources
// are added into Y.  Same structure as residue_scatter_1024, any LOG2N (radix-2 base stage when log2 N is odd: bundle:447-463).
template <int LOG2N, int R_, bool CT>
__device__ __attribute__((noinline)) PV_NO_DS_MERGE void residue_scatter_wg(const float *in, const float *hist, int hist_len, bool sys, long s0, const float *__restrict__ hann,
                                                             const float2 *__restrict__ tw32, int t, int upper_end, int up_delta, unsigned up_ridx,
                                                             double *dbg_X, bool plain)
{
    // plain: the frame passed the pairwise test (see the peak search), so nothing but the residue itself lands on the residue's targets -- the
    // continuation of the last region, all distinct: plain stores instead of claim rounds (two barriers and an atomic per source and round)
    constexpr int G = 1 << (LOG2N - 10);
    using C = WgCfg<G, CT>;
    constexpr int N = C::N, H = C::H, T = C::T, QN = N / 4;
    constexpr bool BASE4 = (LOG2N % 2) == 0;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float2 *Y = reinterpret_cast<float2 *>(smem + C::OFF_Y);
    unsigned *CLAIM = reinterpret_cast<unsigned *>(smem + C::OFF_ROUTE);             // aliases ROUTE once the routes are in registers (f < 1)
    float2 *Q = reinterpret_cast<float2 *>(smem + C::OFF_RESQ);
    const double2 *TWA1 = reinterpret_cast<const double2 *>(smem + C::OFF_TWA);     // row 1 of the forward FFT's twiddle table: W_N^{2t}, t < T (same offset in both layouts)
    const WaveSrc src{in, hist, hist_len, sys};
    const float *XQ = reinterpret_cast<const float *>(smem + C::OFF_XQ);   // the frame's windowed samples xw[4n + 2] (stashed by the kernel): base stage of the first quarter
    for (int base = N / 2; base < N && base < upper_end; base += QN) {
        if (base == N / 2) {
            if (BASE4) {
                const int off = digitrev4_(N / 8 + t, (LOG2N - 2) / 2);      // = 2 (mod 4): sample off + q N/4 is XQ[(off - 2) / 4 + q N/16]
                const float a = XQ[(off - 2) >> 2], b = XQ[((off - 2) >> 2) + N / 16], c = XQ[((off - 2) >> 2) + N / 8], d = XQ[((off - 2) >> 2) + 3 * N / 16];
                const float t0 = a + c, t1 = a - c, t2 = b + d, t3 = b - d;
                Q[4 * t] = float2{t0 + t2, 0.f};
                Q[4 * t + 1] = float2{t1, -t3};
                Q[4 * t + 2] = float2{t0 - t2, 0.f};
                Q[4 * t + 3] = float2{t1, t3};
            } else {
#pragma unroll
                for (int i = 0; i < 2; i++) {
                    const int lb = t + T * i, off = digitrev4_(N / 4 + lb, (LOG2N - 1) / 2);
                    const float a = XQ[(off - 2) >> 2], b = XQ[((off - 2) >> 2) + N / 8];
                    Q[2 * lb] = float2{a + b, 0.f};
                    Q[2 * lb + 1] = float2{a - b, 0.f};
                }
            }
        } else if (BASE4) {
            constexpr int nd = (LOG2N - 2) / 2;
            const int blk = base / 4 + t;                                  // QN/4 = T blocks per quarter
            const int off = digitrev4_(blk, nd);
            const float a = mul_rounded(src.at(s0 + off), hann[off]), b = mul_rounded(src.at(s0 + off + N / 4), hann[off + N / 4]);
            const float c = mul_rounded(src.at(s0 + off + N / 2), hann[off + N / 2]), d = mul_rounded(src.at(s0 + off + 3 * N / 4), hann[off + 3 * N / 4]);
            const float t0 = a + c, t1 = a - c, t2 = b + d, t3 = b - d;
            Q[4 * t] = float2{t0 + t2, 0.f};
            Q[4 * t + 1] = float2{t1, -t3};
            Q[4 * t + 2] = float2{t0 - t2, 0.f};
            Q[4 * t + 3] = float2{t1, t3};
        } else {
            constexpr int nd = (LOG2N - 1) / 2;
#pragma unroll
            for (int i = 0; i < 2; i++) {                                  // QN/2 = 2T blocks per quarter
                const int lb = t + T * i, blk = base / 2 + lb;
                const int off = digitrev4_(blk, nd);
                const float a = mul_rounded(src.at(s0 + off), hann[off]), b = mul_rounded(src.at(s0 + off + N / 2), hann[off + N / 2]);
                Q[2 * lb] = float2{a + b, 0.f};
                Q[2 * lb + 1] = float2{a - b, 0.f};
            }
        }
        __syncthreads();
        constexpr int LOG2BASE = BASE4 ? 2 : 1;
        for (int log2m = LOG2BASE + 2; log2m <= LOG2N - 2; log2m += 2) {  // block sizes 4*base .. N/4 inside the quarter
            const int q = (1 << log2m) >> 2, hq = q >> 1;
            const int nblocks = QN >> log2m;
            const int tws = LOG2N - log2m;
            if (t < nblocks * (hq + 1)) {
                int blk, i;
                if (t < nblocks * hq) { blk = t / hq; i = t - blk * hq; } else { blk = t - nblocks * hq; i = hq; }
                const int o = blk << log2m;
                const float2 A = Q[o + i];
                // W^{e}, W^{2e}, W^{3e}, e = i << tws (even, <= N/8): W^e from the forward FFT's fp64 table in LDS (row 1 = W_N^{2t}), its square and
                // cube formed here.  Three loads from the global table per butterfly were three exposed round trips in each of the five stages (one
                // workgroup per CU has nothing to cover them with): 2500 of the 19 400 cycles of a call each (station clock, tools/read_wg_phases.py).
                float2 w1;
                if (i == hq) w1 = float2{0.70710678118654752440f, -0.70710678118654752440f};   // e = N/8: W_8 (one past the table row)
                else { const double2 wd = TWA1[i << (tws - 1)]; w1 = float2{(float)wd.x, (float)wd.y}; }
                const float2 w2 = cmul(w1, w1), w3 = cmul(w2, w1);
                const float2 Bv = cmul(Q[o + q + i], w1);
                const float2 Cc = cmul(Q[o + 2 * q + i], w2);
                const float2 D = cmul(Q[o + 3 * q + i], w3);
                const float2 T0 = cadd(A, Cc), T1 = csub(A, Cc), T2 = cadd(Bv, D), T3 = csub(Bv, D);
                Q[o + i] = cadd(T0, T2);
                Q[o + q + i] = float2{T1.x + T3.y, T1.y - T3.x};
                if (i == 0) {
                    Q[o + 2 * q] = csub(T0, T2);
                } else if (i != hq) {
                    Q[o + q - i] = float2{T1.x - T3.y, -(T1.y + T3.x)};
                    Q[o + 2 * q - i] = float2{T0.x - T2.x, -(T0.y - T2.y)};
                }
            }
            __syncthreads();
        }
        if (dbg_X)
            for (int i = t; i < QN; i += T) if (base + i >= H) { dbg_X[2 * (base + i)] = Q[i].x; dbg_X[2 * (base + i) + 1] = Q[i].y; }
        unsigned rt[4];
        float2 ys[4];
        int id[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int b = base + t + T * j, tgt = b + up_delta;
            rt[j] = (b >= H && b < upper_end && tgt >= 0 && tgt < H) ? ((up_ridx << 16) | (unsigned)tgt) : NOROUTE;
            ys[j] = rotate_route<R_, LOG2N>(rt[j], Q[t + T * j], tw32);
            id[j] = b - N / 2;                                             // ascending with the source bin; regular sources are done by now
        }
        if (plain) {
#pragma unroll
            for (int j = 0; j < 4; j++) if (rt[j] != NOROUTE) Y[rt[j] & 0xFFFFu] = ys[j];
        } else {
            claim_rounds_wg<4, (1 << (LOG2N - 1)) + 1>(rt, ys, id, Y, CLAIM);
        }
        __syncthreads();
    }
}

// RESIDENT = true (register-resident overlap-add only, S_ROWS >= 1): streaming instance that stays on the GPU (PV_FLAG_PERSISTENT_STREAM).  One workgroup per channel
// slot; each polls ITS OWN control word ctl[16 + ch] (same packing as the one-wave kernels' ctl[0]; channel count 0 = "carry your state, you are not in this quantum"),
// so that the host can hand the channels over one by one while it is still copying the next one's input (pv_capi.hip).
template <int LOG2N, int S_ROWS, bool AUX, bool RESIDENT = false>
__global__ __launch_bounds__(64 << (LOG2N - 10), 2) PV_NO_DS_MERGE void pv_wg_kernel(const PvKernelParams p)
{
    constexpr int G = 1 << (LOG2N - 10);
    using C = WgCfg<G, (S_ROWS == 0)>;
    constexpr int N = C::N, M = C::M, H = C::H, T = C::T;
    // S_ROWS = hop / (2T) in {1,2,4,8}: the frame advances by whole register rows -> overlap-add accumulator and input window live in
    // registers.  S_ROWS = 0: any other (even) hop dividing N, e.g. the reference's native 2048/128 (R = 16): accumulator ring in LDS
    // (reference order, ola:149-157), window re-loaded per frame (the overlap comes from L2).
    constexpr bool RING = (S_ROWS == 0);
    const int HOP = RING ? p.hop : 2 * T * S_ROWS;
    constexpr int R = RING ? 0 : N / (2 * T * (S_ROWS ? S_ROWS : 1));    // compile-time R (4 => exact j^q rotations); 0 = run time
    const int Rrt = N / HOP;
    constexpr int LROWS = RING ? 0 : 8 - S_ROWS;
    constexpr int NEGPD = -(1 << 30), POSPD = 1 << 30;                    // packed (bin << 16 | shift) sentinels: no peak on this side
    constexpr int DROP = 0x4000;                                        // shift sentinel: b + DROP >= H for every bin, above every real shift
    const int t = threadIdx.x;
    const int ch = blockIdx.y, chunk = blockIdx.x;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double2 *S64 = reinterpret_cast<double2 *>(smem);
    float2 *S32 = reinterpret_cast<float2 *>(smem);
    float2 *Y = reinterpret_cast<float2 *>(smem + C::OFF_Y);
    float *MAG = reinterpret_cast<float *>(smem + C::OFF_ROUTE);
    unsigned *ROUTE = reinterpret_cast<unsigned *>(smem + C::OFF_ROUTE);
    unsigned *CLAIM = reinterpret_cast<unsigned *>(smem + C::OFF_ROUTE);             // aliases ROUTE once the routes are in registers (f < 1)
    short *DSH = reinterpret_cast<short *>(smem + C::OFF_PSH);          // shift per candidate peak bin (DROP: peak dropped)
    int *LASTIN = reinterpret_cast<int *>(smem + C::OFF_NEAR), *FIRSTIN = LASTIN + T;
    unsigned long long *OCC = reinterpret_cast<unsigned long long *>(smem + C::OFF_OCC);
    double2 *TWA = reinterpret_cast<double2 *>(smem + C::OFF_TWA);
    double2 *TWB = reinterpret_cast<double2 *>(smem + C::OFF_TWB);
    double2 *TWC = reinterpret_cast<double2 *>(smem + C::OFF_TWC);

    WG_STAMP(0);
    // ---- tables: W_M^{t k} = tw[2 t k], W_T^{q k} = tw[16 q k], W_{8G}^{q k} = tw[128 q k]  (tw[i] = exp(-2 pi j i / N)) ----
#pragma unroll
    for (int k = 1; k < 8; k++) {
        if (!RING) TWA[(k - 1) * T + t] = p.tw64[(2 * t * k) & (N - 1)];
        else if (k == 1 || k == 2 || k == 4) TWA[(k == 4 ? 2 : k - 1) * T + t] = p.tw64[(2 * t * k) & (N - 1)];
    }
    for (int i = t; i < 7 * 8 * G; i += T) TWB[i] = p.tw64[(16 * (i % (8 * G)) * (i / (8 * G) + 1)) & (N - 1)];
    for (int i = t; i < 7 * G; i += T) TWC[i] = p.tw64[(128 * (i % G) * (i / G + 1)) & (N - 1)];
    // conj(W^{2k}) of the fast residue's two bins of a thread, k = 1 + t + T j: in LDS with the other tables (as a global load inside the frame it sat,
    // exposed, on the critical path of every f < 1 frame -- one workgroup per CU has nothing to cover a global round trip with)
    float2 *S2W = reinterpret_cast<float2 *>(smem + C::OFF_S2W);
    S2W[t] = cconj(p.tw32[2 * (1 + t)]);
    S2W[T + t] = cconj(p.tw32[2 * (1 + t + T)]);

    WG_STAMP(1);
    unsigned psh_key = 0u;                                 // bit pattern of the f the shift table was built for, valid once psh_valid
    bool psh_valid = false;
    // what changes from quantum to quantum in the resident form (constants of the launch otherwise)
    const float *hist_in = p.hist_in, *acc_in = p.acc_in;
    float *hist_out = p.hist_out, *acc_out = p.acc_out;
    int t0_mod_n = p.t0_mod_n;
    unsigned done_seq = p.done_seq;
    unsigned last_seq = p.done_seq;
resident_top:
    if (RESIDENT) {
        unsigned *BC = reinterpret_cast<unsigned *>(smem + C::OFF_OCC);  // broadcast slot (the peak search's wave masks: not live here)
        if (t == 0) {
            unsigned word;
            const unsigned long long idle0 = wall_clock64();
            for (;;) {
                word = __hip_atomic_load(p.ctl + 16 + ch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if ((word & 0xFFFFu) != (last_seq & 0xFFFFu)) break;
                if (__hip_atomic_load(p.ctl + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u || wall_clock64() - idle0 > (unsigned long long)p.idle_ticks) { word = 0u; break; }   // asked to leave / ~50 ms idle
                __builtin_amdgcn_s_sleep(2);
            }
            BC[0] = word;
        }
        __syncthreads();
        const unsigned word = BC[0];
        __syncthreads();
        if (word == 0u) return;                                          // (a sequence number is never 0)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
        const unsigned seq = word & 0xFFFFu, nch_now = (word >> 16) & 0x7Fu, cur = (word >> 23) & 1u;
        t0_mod_n = (int)(((word >> 24) & 0xFFu) * HOP) & (N - 1);
        hist_in = p.hist2[cur]; hist_out = p.hist2[cur ^ 1u];
        acc_in = p.acc2[cur]; acc_out = p.acc2[cur ^ 1u];
        done_seq = last_seq = seq;
        if (nch_now == 0u) {                                             // not part of this quantum: carry the state across the ping-pong flip
            for (int j = t; j < N - HOP; j += T) { hist_out[(long)ch * (N - HOP) + j] = hist_in[(long)ch * (N - HOP) + j]; acc_out[(long)ch * (N - HOP) + j] = acc_in[(long)ch * (N - HOP) + j]; }
            goto resident_top;
        }
    }
    const int first_out = chunk * p.frames_per_chunk;
    int last_out = first_out + p.frames_per_chunk;
    if (last_out > p.nhops) last_out = p.nhops;
    int first_frame = first_out - (Rrt - 1);
    const bool from_state = (first_frame <= 0);
    if (from_state) first_frame = 0;

    const long cbase = (long)ch * p.ch_stride;
    const WaveSrc src{p.in + cbase, hist_in + (long)ch * (N - HOP), N - HOP, RESIDENT && p.in_cached != 0};
    float *outp = p.out + cbase;
    const bool vec_out = (reinterpret_cast<uintptr_t>(outp) & 7u) == 0;
    const bool vec_in = ((reinterpret_cast<uintptr_t>(src.in) | reinterpret_cast<uintptr_t>(src.hist)) & 7u) == 0;
    const float *pitch_row = p.pitch + (p.pitch_stride ? (long)(ch / p.ch_per_stream) * p.pitch_stride : 0);
    const float invR = 1.0f / (float)Rrt;
    float *ACC = reinterpret_cast<float *>(smem + C::OFF_ACC);          // RING only
    const int Lr = N - HOP;
    int ring = 0;

    const double2 wl = p.tw64[t];                          // split pass: W_N^{t + T r} = wl * W_16^r  (N = 16 T)
    const float2 wlf = cconj(p.tw32[t]);
    const pk::c32 wlfs{wlf.x * (1.0f / (float)N), wlf.y * (1.0f / (float)N)};   // c2r twiddle with the 1/N of the inverse folded in (exact)
    float2 hw[8];                                          // Hann at samples 2(t + T r), +1
#pragma unroll
    for (int r = 0; r < 8; r++) hw[r] = float2{p.hann[2 * (t + T * r)], p.hann[2 * (t + T * r) + 1]};

    float2 acc[8];
#pragma unroll
    for (int r = 0; r < 8; r++) acc[r] = float2{0.f, 0.f};
    if (RING) {
        for (int j = t; j < Lr; j += T) ACC[j] = from_state ? acc_in[(long)ch * Lr + j] : 0.f;
    } else if (from_state) {
#pragma unroll
        for (int r = 0; r < LROWS; r++) {
            const float *a = acc_in + (long)ch * (N - HOP) + 2 * t + 2 * T * r;
            acc[r] = float2{a[0], a[1]};
        }
    }
    auto load_rows = [&](float2 *w, int nrows, int first_row, int frame) {
        const long s0 = (long)(frame + 1) * HOP - N + 2 * t;
#pragma unroll
        for (int r = 0; r < nrows; r++) {
            const long sx = s0 + 2 * T * (first_row + r);
            if (RESIDENT && src.sys && vec_in && sx >= 0) {                // the host's hop of this quantum: never from a cache
                const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(src.in + sx), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                w[r] = float2{__uint_as_float((unsigned)v), __uint_as_float((unsigned)(v >> 32))};
            } else if (vec_in) w[r] = *reinterpret_cast<const float2 *>(sx < 0 ? src.hist + sx + src.hist_len : src.in + sx);
            else w[r] = float2{src.at(sx), src.at(sx + 1)};
        }
    };
    float2 raw[8];
    load_rows(raw, 8, 0, first_frame);
    // all global accesses of a frame are issued one frame ahead in one place, and the stores are exec-masked straight-line code (emit_v is a
    // per-lane value on purpose): no s_waitcnt vmcnt(0) behind a just-issued load or store (see pv_wave_kernel.hip)
    float pf_next = (RESIDENT && src.sys) ? __hip_atomic_load(pitch_row + first_frame, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : pitch_row[first_frame];
    int emit_v = first_out;
    asm volatile("" : "+v"(emit_v));
    __syncthreads();
    WG_STAMP(2);
#ifdef PV_WG_STAMPS
    { float sink = 0.f; for (int r = 0; r < 8; r++) sink += raw[r].x; asm volatile("" :: "v"(sink)); }   // the input rows have arrived
    WG_STAMP(3);
#endif

    for (int m = first_frame; m < last_out; ++m) {
        // the thread id is made opaque once per frame: LDS addresses are recomputed from it instead of being hoisted into registers that
        // stay live across the whole frame (see pv_wave_kernel.hip experiments): the loop then fits its register budget without spills
        int tq = t;
        asm volatile("" : "+v"(tq));
        const int l = tq & 63, wv = __builtin_amdgcn_readfirstlane(tq >> 6);
        const float pfm = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(pf_next)));   // k-rate pitchFactor (pv:47), uniform
        const double pf = (double)pfm;
        const int tmod = (int)(((long)t0_mod_n + (long)m * HOP) & (N - 1));
        const bool dbg = AUX && (p.dbg_mag != nullptr) && ch == p.dbg_ch && m == p.dbg_frame;

        // ---- Hann (pv:55), pack, forward FFT in fp64 (the split pass's 1/2 is folded into the window, exact) ----
        double2 z[8];
#pragma unroll
        for (int r = 0; r < 8; r++) z[r] = double2{(double)(raw[r].x * (0.5f * hw[r].x)), (double)(raw[r].y * (0.5f * hw[r].y))};
        fft_wg<double, false, G, RING>(z, S64, TWA, TWB, TWC, tq);

        // ---- split pass in conjugate pairs (see pv_wave_kernel.hip): thread tq owns the pairs k = tq + T r, r < 4, i.e. bins XA[r] = X[k] and
        //      XB[r] = X[M - k]; thread 0 also the self-paired bin M/2.  The partner values Z[M - k] are rows 4..7 of other threads -> LDS ----
        float2 XA[4], XB[4], xHf{0.f, 0.f};
        {
#pragma unroll
            for (int r = 4; r < 8; r++) S64[tq + T * (r - 4)] = z[r];
            __syncthreads();
            double2 xa[4], xb[4], xH{0.0, 0.0};
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int k = tq + T * r;
                const double2 zm = (k == 0) ? z[0] : S64[4 * T - k];          // element M - k sits at (M - k) - 4T of the four stored rows
                const double2 E{z[r].x + zm.x, z[r].y - zm.y};
                const double2 O{z[r].x - zm.x, z[r].y + zm.y};
                const double2 WO = cmul(wl, mul_w16<double, false>(O, r));
                xa[r] = double2{E.x + WO.y, E.y - WO.x};
                xb[r] = double2{E.x - WO.y, -(E.y + WO.x)};
            }
            if (tq == 0) {
                xa[0] = double2{2.0 * (z[0].x + z[0].y), 0.0};                // X[0], X[M]: both real (bundle:447-508 keep Im = 0)
                xb[0] = double2{2.0 * (z[0].x - z[0].y), 0.0};
                xH = double2{2.0 * z[4].x, -2.0 * z[4].y};                    // k = M/2 pairs with itself: W^{N/4} = -j, X = 2 conj(Z)
            }
            // partner reads done: the scratch becomes MAG / Y / ROUTE.  MAG / ROUTE sit behind the four partner rows ([0, 4T) double2 = [0, OFF_ROUTE)),
            // only the f < 1 spectrum stash (in the Y region) overwrites them: an f >= 1 frame needs no barrier here (pf is uniform in the workgroup;
            // the barrier behind the shift table is the next one, Y is zeroed behind the one after that)
            static_assert(C::OFF_ROUTE >= 16 * 4 * T, "MAG must not alias the partner rows of the split pass");
            if (!(pf >= 1.0) || !PV_WG_INV_SPLIT) __syncthreads();
#pragma unroll
            for (int r = 0; r < 4; r++) {
                MAG[4 + tq + T * r] = (float)(xa[r].x * xa[r].x + xa[r].y * xa[r].y);
                MAG[4 + M - tq - T * r] = (float)(xb[r].x * xb[r].x + xb[r].y * xb[r].y);
                XA[r] = float2{(float)xa[r].x, (float)xa[r].y};
                XB[r] = float2{(float)xb[r].x, (float)xb[r].y};
            }
            if (tq == 0) MAG[4 + M / 2] = (float)(xH.x * xH.x + xH.y * xH.y);
            xHf = float2{(float)xH.x, (float)xH.y};
            if (pf < 1.0) {                                                // fp32 spectrum stash for the fast residue (Y is not live yet)
#pragma unroll
                for (int r = 0; r < 4; r++) { Y[tq + T * r] = XA[r]; Y[M - tq - T * r] = XB[r]; }
                if (tq == 0) Y[M / 2] = xHf;
                if (pf < 0.75 && (tq & 1)) {
                    // A frame that reads beyond position N/2 + N/8 above Nyquist (possible only for f < 0.75) rebuilds quarter 2 of fft.js's buffer from the
                    // windowed samples xw[4n + 2] (residue_scatter_wg).  They are still in registers here -- sample 2 (t + T r) of the odd threads -- and a
                    // digit-reversed gather from global memory later costs 5500-7500 cycles (profiles/r03_wg_phase_clock.md): stashed in natural order, 8 KB.
                    float *XQ = reinterpret_cast<float *>(smem + C::OFF_XQ);
#pragma unroll
                    for (int r = 0; r < 8; r++) XQ[(tq + T * r - 1) >> 1] = raw[r].x * hw[r].x;
                }
            }
            if (dbg) {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int ka = tq + T * r, kb = M - ka;
                    p.dbg_X[2 * ka] = xa[r].x; p.dbg_X[2 * ka + 1] = xa[r].y;
                    p.dbg_X[2 * kb] = xb[r].x; p.dbg_X[2 * kb + 1] = xb[r].y;
                }
                if (tq == 0) { p.dbg_X[M] = xH.x; p.dbg_X[M + 1] = xH.y; }
            }
        }
        // slide the raw window; the rows the next frame adds are issued here
        {
            const int mn = (m + 1 < last_out) ? m + 1 : m;                 // the last frame re-reads its own rows (unused): no branch
            if (RESIDENT) {
                // one frame per quantum: nothing to prefetch; the slide leaves the next call's history in raw[0 .. 8 - S_ROWS)
#pragma unroll
                for (int r = 0; r < 8 - (RING ? 8 : S_ROWS); r++) raw[r] = raw[r + S_ROWS];
            } else if (RING) {
                load_rows(raw, 8, 0, mn);
            } else {
#pragma unroll
                for (int r = 0; r < 8 - S_ROWS; r++) raw[r] = raw[r + S_ROWS];
                load_rows(&raw[8 - (RING ? 8 : S_ROWS)], S_ROWS, 8 - S_ROWS, mn);
            }
            if (!RESIDENT) pf_next = pitch_row[mn];
        }
        // ---- shift table Math.round(peak * f) - peak (pv:125,147), rebuilt only when f changes ----
        {
            const unsigned pfb = __float_as_uint(pfm);
            if (!psh_valid || pfb != psh_key) {
                psh_key = pfb;
                psh_valid = true;
#pragma unroll
                for (int r = 0; r < 8; r++) {
                    const int pk = tq + T * r;
                    const double ps = floor((double)pk * pf + 0.5);
                    const bool ok = (ps <= (double)H) && (ps >= -(double)(2 * N));
                    DSH[pk] = ok ? (short)((int)ps - pk) : (short)DROP;             // DROP pushes every target of the region out of range
                }
            }
        }
        __syncthreads();
        // ---- above-Nyquist residue, fast form (see residue_fast_1024 in pv_wave_kernel.hip): positions N/2+1 .. N/2+N/8 of fft.js's buffer are
        //      the clean first half of the N/4-point sub-DFT S2 of xw[4n+2], and W^{2k} S2[k] = (X[k] - X[k+N/4] + X[k+N/2] - X[k+3N/4]) / 4.
        //      Computed speculatively for every f < 1 frame (two bins per thread) while the stash is readable; used if the last region ends
        //      at or below N/2 + 1 + N/8, else the general path re-runs the reference's stage structure ----
        float2 s2v[2] = {float2{0.f, 0.f}, float2{0.f, 0.f}};
        if (pf < 1.0) {
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const int k = 1 + tq + T * j;                                // k in [1, N/8]
                const float2 x0 = Y[k], x1 = Y[k + M / 2], x2 = Y[M - k], x3 = Y[M / 2 - k];
                const float2 tsum{0.25f * ((x0.x - x1.x) + (x2.x - x3.x)), 0.25f * ((x0.y - x1.y) - (x2.y - x3.y))};
                s2v[j] = cmul(tsum, S2W[j * T + tq]);
            }
        }
        bool nonfinite = false;
        // ---- peak flags on bins 8t..8t+7 (pv:95-116) ----
        int lastown[8], firstown[8];                                      // last own peak <= bin i / first own peak > bin i
        int last_in, first_in;
        {
            // |X|^2 >= 0: fp32 order = order of the bit patterns as unsigned integers, so "strictly greater than all four neighbours" is
            // c > max(neighbours) with v_max3_u32 (see pv_wave_kernel.hip): two instructions per bin + eight shared pair maxima
            unsigned mg[12];
            // LDS-address-space vector loads: otherwise the optimizer re-pairs the 12 words into misaligned ds_read2_b32 (8 LDS cycles each)
            typedef const volatile __attribute__((address_space(3))) v2u *lds_v2u;
            typedef const volatile __attribute__((address_space(3))) v4u *lds_v4u;
            const v2u q0 = *(lds_v2u)(&MAG[4 + 8 * tq - 2]);
            const v4u q1 = *(lds_v4u)(&MAG[4 + 8 * tq]);
            const v4u q2 = *(lds_v4u)(&MAG[4 + 8 * tq + 4]);
            const v2u q3 = *(lds_v2u)(&MAG[4 + 8 * tq + 8]);
            mg[0] = q0.x; mg[1] = q0.y; mg[2] = q1.x; mg[3] = q1.y; mg[4] = q1.z; mg[5] = q1.w;
            mg[6] = q2.x; mg[7] = q2.y; mg[8] = q2.z; mg[9] = q2.w; mg[10] = q3.x; mg[11] = q3.y;
            unsigned pm[11];
#pragma unroll
            for (int j = 3; j < 11; j++) pm[j] = max(mg[j], mg[j + 1]);
            bool fl[8];
#pragma unroll
            for (int i = 0; i < 8; i++) {
                // bin k = 8t + i, candidates are 2 <= k < H - 2 (pv:97-100): thread 0 drops i < 2, the last thread drops i = 7
                const bool in_range = (i < 2) ? (tq != 0) : (i == 7) ? (tq != T - 1) : true;
                fl[i] = in_range & (max(max(mg[i], mg[i + 1]), pm[i + 3]) < mg[i + 2]);
            }
            nonfinite = __any(max(max(max(pm[3], pm[5]), max(pm[7], pm[9])), mg[2]) >= 0x7F800000u);   // Inf / NaN magnitude in this wave's bins (see pv_wave_kernel.hip)
            if (dbg) {
                for (int i = 0; i < 8; i++) { p.dbg_flags[8 * tq + i] = fl[i] ? 1 : 0; p.dbg_mag[8 * tq + i] = __uint_as_float(mg[i + 2]); }
                if (tq == T - 1) { p.dbg_flags[M] = 0; p.dbg_mag[M] = __uint_as_float(mg[10]); }
            }
            // candidate peaks travel as packed words (bin << 16 | shift & 0xFFFF), see pv_wave_kernel.hip: one 16-byte read of the shift table
            // per thread instead of a 4-way-conflicted DSH[owner] lookup per bin
            const v4u dq = *(lds_v4u)(&DSH[8 * tq]);
            int pd[8];
#pragma unroll
            for (int i = 0; i < 8; i++)
                pd[i] = (int)__builtin_amdgcn_perm((unsigned)(8 * tq + i), dq[i >> 1], (i & 1) ? 0x05040302u : 0x05040100u);
            int cur = NEGPD;
#pragma unroll
            for (int i = 0; i < 8; i++) { cur = fl[i] ? pd[i] : cur; lastown[i] = cur; }
            int nx = POSPD;
#pragma unroll
            for (int i = 7; i >= 0; i--) { firstown[i] = nx; nx = fl[i] ? pd[i] : nx; }
            last_in = cur; first_in = nx;
        }
        // ---- nearest peaks outside this thread's byte: per-wave occupancy ballots + per-thread last/first peak, through LDS ----
        {
            const unsigned long long occ = __ballot(last_in >= 0);
            LASTIN[tq] = last_in;
            FIRSTIN[tq] = first_in;
            if (l == 0) OCC[wv] = occ;
        }
        __syncthreads();                                                   // also: every MAG read is done -> ROUTE may overwrite MAG
        int cprev = NEGPD, cnext = POSPD, last_peak = -1, last_shift = 0;
        {
            const unsigned long long mine = OCC[wv];
            {
                int srcT = -1;
                const unsigned long long below = mine & ((1ull << l) - 1ull);
                if (below) srcT = wv * 64 + 63 - __clzll((long long)below);
                else
                    for (int w = wv - 1; w >= 0; --w) { const unsigned long long o = OCC[w]; if (o) { srcT = w * 64 + 63 - __clzll((long long)o); break; } }
                if (srcT >= 0) cprev = LASTIN[srcT];
            }
            {
                int srcT = -1;
                const unsigned long long above = (l == 63) ? 0ull : (mine >> (l + 1));
                if (above) srcT = wv * 64 + l + __ffsll((long long)above);
                else
                    for (int w = wv + 1; w < G; ++w) { const unsigned long long o = OCC[w]; if (o) { srcT = w * 64 + __ffsll((long long)o) - 1; break; } }
                if (srcT >= 0) cnext = FIRSTIN[srcT];
            }
            for (int w = G - 1; w >= 0; --w) {
                const unsigned long long o = OCC[w];
                if (o) { const int lp = LASTIN[w * 64 + 63 - __clzll((long long)o)]; last_peak = lp >> 16; last_shift = (int)(short)(lp & 0xFFFF); break; }
            }
        }
        {
            unsigned rt[8];
            unsigned rtM = NOROUTE;
            bool bad = false;                                               // f < 1: a gap next to this thread's bins overlaps by more than its rising side
            if (last_peak < 0) {                                            // no peak at all (workgroup-uniform): nothing moves
#pragma unroll
                for (int i = 0; i < 8; i++) rt[i] = NOROUTE;
            } else {
                // owner rule (pv:132-141) + shift (pv:147-152): ROUTE = ((delta * tq) mod N) << 16 | target; a route is valid iff its target
                // field is < H (pv:127-129 via DROP, pv:150-152, negative index); bits above the rotation index are don'tq-care
                auto route_of = [&](int b, int pp, int pn) -> unsigned {
                    const int own = (b - (pp >> 16) < (pn >> 16) - b) ? pp : pn;     // at least one side is a real peak here
                    const int delta = __builtin_amdgcn_sbfe(own, 0, 16);
                    return __builtin_amdgcn_perm((unsigned)__mul24(delta, tmod), (unsigned)(b + delta), 0x05040100u);
                };
#pragma unroll
                for (int i = 0; i < 8; i++) rt[i] = route_of(8 * tq + i, max(lastown[i], cprev), min(firstown[i], cnext));
                if (tq == T - 1) rtM = route_of(M, max(last_in, cprev), POSPD);
                if (!(pf >= 1.0)) {
                    // f < 1: bit 31 of a route = "rising side" (source owned by the peak on its right), and this thread's share of the test that lets
                    // the scatter run as store-then-add instead of claim rounds (pv_wave_kernel.hip, "pairwise"; tests/test_pairwise_rule.py)
                    rtM &= 0x7FFFFFFFu;
#pragma unroll
                    for (int i = 0; i < 8; i++) {
                        const int pp = max(lastown[i], cprev), pn = min(firstown[i], cnext), b = 8 * tq + i;
                        const bool rising = !(b - (pp >> 16) < (pn >> 16) - b);
                        rt[i] = (rt[i] & 0x7FFFFFFFu) | (rising ? 0x80000000u : 0u);
                    }
                    if (!(pfm >= PV_PAIRWISE_SURE)) {                       // (f >= 2/3: the test cannot fail, see pv_device_common.h)
#pragma unroll
                        for (int i = 0; i < 8; i++) {
                            const int pp = max(lastown[i], cprev), pn = min(firstown[i], cnext);
                            const int gap = (pn >> 16) - (pp >> 16), ov = __builtin_amdgcn_sbfe(pp, 0, 16) - __builtin_amdgcn_sbfe(pn, 0, 16);
                            bad |= ov > (gap >> 1);
                        }
                    }
                }
            }
            *reinterpret_cast<uint4 *>(&ROUTE[8 * tq]) = uint4{rt[0], rt[1], rt[2], rt[3]};
            *reinterpret_cast<uint4 *>(&ROUTE[8 * tq + 4]) = uint4{rt[4], rt[5], rt[6], rt[7]};
            if (tq == T - 1) ROUTE[M] = rtM;
            if (!(pf >= 1.0)) { const bool wbad = __any(bad); if (l == 0) ROUTE[M + 4 + wv] = wbad ? 1u : 0u; }   // (words M+1 .. M+15 of the route array are spare)
        }
        int upper_end = H;
        if (last_peak >= 0 && last_shift < 0) { upper_end = H - last_shift; if (upper_end > N) upper_end = N; }      // DROP is positive
        // ---- zero Y (pv:121) ----
#pragma unroll
        for (int r = 0; r < 4; r++) *reinterpret_cast<v4f *>(&Y[2 * tq + 2 * T * r]) = v4f{0.f, 0.f, 0.f, 0.f};      // four ds_write_b128 instead of eight ds_write_b64
        if (tq == 0) Y[M] = float2{0.f, 0.f};
        const bool need_res = upper_end > H;
        __syncthreads();
        // ---- shiftPeaks (pv:119-173) ----
        {
            const bool disjoint = (pf >= 1.0);
            if (disjoint) {
                // every rotation of this frame is exp(2 pi j delta tmod / N) and tmod is uniform: tmod = 0 moves the bins unrotated,
                // tmod = N/2 only flips signs (top bit of the rotation index = bit 15 + LOG2N of the route)
                auto scatter = [&](auto mode_tag) {
                    constexpr int MODE = decltype(mode_tag)::value;
                    auto rot = [&](unsigned rt, float2 v) -> float2 {
                        if (MODE == 0) return v;
                        if (MODE == 2) {
                            const unsigned sg = (rt << (16 - LOG2N)) & 0x80000000u;
                            return float2{__uint_as_float(__float_as_uint(v.x) ^ sg), __uint_as_float(__float_as_uint(v.y) ^ sg)};
                        }
                        return rotate_route<R, LOG2N>(rt, v, p.tw32);
                    };
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const unsigned ra = ROUTE[tq + T * r], ta = ra & 0xFFFFu;
                        const unsigned rb = ROUTE[M - tq - T * r], tb = rb & 0xFFFFu;
                        if (ta < (unsigned)H) Y[ta] = rot(ra, XA[r]);
                        if (tb < (unsigned)H) Y[tb] = rot(rb, XB[r]);
                    }
                    if (tq == 0) { const unsigned rt = ROUTE[M / 2], tg = rt & 0xFFFFu; if (tg < (unsigned)H) Y[tg] = rot(rt, xHf); }
                };
                if (tmod == 0) scatter(std::integral_constant<int, 0>{});
                else if (tmod == N / 2) scatter(std::integral_constant<int, 2>{});
                else scatter(std::integral_constant<int, 1>{});
            } else {
                bool pairwise;
                unsigned rt[9];
                float2 ys[9];
                int id[9];
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    id[r] = tq + T * r; rt[r] = ROUTE[id[r]]; ys[r] = rotate_route<R, LOG2N>(rt[r], XA[r], p.tw32);
                    id[4 + r] = M - tq - T * r; rt[4 + r] = ROUTE[id[4 + r]]; ys[4 + r] = rotate_route<R, LOG2N>(rt[4 + r], XB[r], p.tw32);
                }
                rt[8] = (tq == 0) ? ROUTE[M / 2] : NOROUTE;
                ys[8] = rotate_route<R, LOG2N>(rt[8], xHf, p.tw32);
                id[8] = M / 2;
                pairwise = PV_PAIRWISE != 0;
#pragma unroll
                for (int w = 0; w < G; w++) pairwise = pairwise && (ROUTE[M + 4 + w] == 0u);       // uniform in the workgroup
                if (pairwise) {
                    // every collision of this frame is one falling-side source against one rising-side source: the falling side and the residue (it
                    // continues the falling side of the last peak) store into the zeroed Y, one barrier, the rising side adds.  No claim words, and
                    // one barrier where every claim round has two.
                    unsigned key[9];
#pragma unroll
                    for (int r = 0; r < 9; r++) key[r] = rt[r] & 0x8000FFFFu;
#pragma unroll
                    for (int r = 0; r < 9; r++) if (key[r] < (unsigned)H) Y[key[r]] = ys[r];
                    const int up_delta = need_res ? (int)DSH[last_peak] : 0;
                    const unsigned up_ridx = (unsigned)((up_delta & (N - 1)) * tmod) & (N - 1);
                    if (need_res && upper_end <= H + N / 8) {
#pragma unroll
                        for (int j = 0; j < 2; j++) {
                            const int b = H + tq + T * j, tgt = b + up_delta;
                            const unsigned rtj = (b < upper_end && tgt >= 0 && tgt < H) ? ((up_ridx << 16) | (unsigned)tgt) : NOROUTE;
                            if (rtj != NOROUTE) Y[tgt] = rotate_route<R, LOG2N>(rtj, s2v[j], p.tw32);
                            if (dbg && b < upper_end) { p.dbg_X[2 * b] = s2v[j].x; p.dbg_X[2 * b + 1] = s2v[j].y; }
                        }
                    }
                    __syncthreads();
                    float2 o[9];
#pragma unroll
                    for (int r = 0; r < 9; r++) o[r] = Y[min(rt[r] & 0xFFFFu, (unsigned)M)];
#pragma unroll
                    for (int r = 0; r < 9; r++) if (key[r] - 0x80000000u < (unsigned)H) Y[key[r] - 0x80000000u] = float2{o[r].x + ys[r].x, o[r].y + ys[r].y};
                } else {
                __syncthreads();                                            // every ROUTE read is done: the region becomes the claim words
#pragma unroll
                for (int r = 0; r < 8; r++) CLAIM[tq + T * r] = 0xFFFFFFFFu;
                if (tq == 0) CLAIM[M] = 0xFFFFFFFFu;
                claim_rounds_wg<9, H>(rt, ys, id, Y, CLAIM);                  // (its first barrier orders the fill before the first claims)
                if (need_res) {
                    __syncthreads();
                    const int up_delta = (int)DSH[last_peak];
                    const unsigned up_ridx = (unsigned)((up_delta & (N - 1)) * tmod) & (N - 1);
                    if (upper_end <= H + N / 8) {                           // sources b = H + tq + T j, all owned by the last peak (pv:133)
                        unsigned rt2[2];
                        float2 ys2[2];
                        int id2[2];
#pragma unroll
                        for (int j = 0; j < 2; j++) {
                            const int b = H + tq + T * j, tgt = b + up_delta;
                            rt2[j] = (b < upper_end && tgt >= 0 && tgt < H) ? ((up_ridx << 16) | (unsigned)tgt) : NOROUTE;
                            ys2[j] = rotate_route<R, LOG2N>(rt2[j], s2v[j], p.tw32);
                            id2[j] = b - N / 2;
                            if (dbg && b < upper_end) { p.dbg_X[2 * b] = s2v[j].x; p.dbg_X[2 * b + 1] = s2v[j].y; }
                        }
                        claim_rounds_wg<2, H>(rt2, ys2, id2, Y, CLAIM);
                    }
                }
                }
                if (need_res && upper_end > H + N / 8) {                    // (one call site for both forms of the scatter: a second one spills the main loop)
                    __syncthreads();
                    const int up_delta = (int)DSH[last_peak];
                    residue_scatter_wg<LOG2N, R, RING>(src.in, src.hist, src.hist_len, src.sys, (long)(m + 1) * HOP - N, p.hann, p.tw32, tq, upper_end, up_delta,
                                                 (unsigned)((up_delta & (N - 1)) * tmod) & (N - 1), dbg ? p.dbg_X : nullptr, pairwise);
                }
            }
        }
        if (nonfinite && l == 0) Y[1 + wv] = float2{__uint_as_float(0x7FC00000u), __uint_as_float(0x7FC00000u)};   // the reference's frame is NaN: so is this one
        __syncthreads();
        if (dbg) {
#pragma unroll
            for (int r = 0; r < 8; r++) { const int k = tq + T * r; p.dbg_Y[2 * k] = Y[k].x; p.dbg_Y[2 * k + 1] = Y[k].y; }
            if (tq == 0) { p.dbg_Y[2 * M] = Y[M].x; p.dbg_Y[2 * M + 1] = Y[M].y; }
        }
        // ---- c2r pre-pass in conjugate pairs, packed fp32: with E = Yk + conj(Ym), O = Yk - conj(Ym), c = e^{+2 pi j k/N} O / N (m = M - k):
        //      Z[k] = E / N + j c and Z[m] = conj(E / N - j c); thread tq computes k = tq + T r, r < 4, and hands Z[m] over through LDS ----
        pk::c32 zi[8];
        {
            const float sc = 1.0f / (float)N;
            const pk::c32 scsc{sc, sc};
            const pk::c32 *Yc = reinterpret_cast<const pk::c32 *>(Y);
            pk::c32 zb[4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int k = tq + T * r;
                pk::c32 yk = Yc[k], ym = Yc[M - k];
                if (k == 0) { yk.y = 0.f; ym.y = 0.f; }
                const pk::c32 E = pk::add_conj(yk, ym), O = pk::sub_conj(yk, ym);
                const pk::c32 c = pk::cmul(mul_w16_inv_pk_wg(O, r), wlfs);
                zi[r] = pk::fma_addj(E, scsc, c);
                zb[r] = pk::fma_conj_subj(E, scsc, c);
            }
            const pk::c32 yH = Yc[M / 2];
            // hand-over buffer = the residue quarter buffer (free here, disjoint from Y): element m = M - k of the packed sequence, m in (4T, 8T)
            pk::c32 *XCH = reinterpret_cast<pk::c32 *>(smem + C::OFF_RESQ);
#pragma unroll
            for (int r = 0; r < 4; r++) if (r > 0 || tq > 0) XCH[4 * T - tq - T * r] = zb[r];   // index m - 4T; (tq = 0, r = 0) would be Z[M]: does not exist
            __syncthreads();
#pragma unroll
            for (int r = 4; r < 8; r++) zi[r] = XCH[tq + T * (r - 4)];
            if (tq == 0) zi[4] = pk::c32{2.0f * yH.x * sc, -2.0f * yH.y * sc};   // the self-paired bin M/2
        }
        // G = 8 with the split scratch: transpose 1 of the inverse writes [0, 32 KB) -- Y, whose reads all sit in front of the hand-over's barrier -- and
        // leaves the hand-over buffer alone (transpose 3 takes it over, behind transpose 1's barrier): no barrier here
        if (!(G == 8 && PV_WG_REG_T2 && PV_WG_INV_SPLIT)) __syncthreads();
        fft_wg_inv_pk<G, RING>(zi, reinterpret_cast<pk::c32 *>(S32), TWA, TWB, TWC, tq);
        // ---- Hann (pv:67), overlap-add in reference order, emit, shift ----
        {
            const bool emit_out = (m >= emit_v);
            float2 fr[8];
#pragma unroll
            for (int r = 0; r < 8; r++)                                    // rounded to fp32 BEFORE the accumulation like the reference's Float32Array (pv:67): no contraction into the adds
                fr[r] = float2{mul_rounded(zi[r].x, hw[r].x * invR), mul_rounded(zi[r].y, hw[r].y * invR)};
            if (RING) {
                // phase 1: samples below N - hop read their slot (emit the first hop, accumulate the rest); phase 2: the last hop samples
                // of the frame take over the slots the emitted hop freed (0 + x, ola:130-137)
#pragma unroll
                for (int r = 0; r < 8; r++) {
                    const int j = 2 * tq + 2 * T * r;
                    if (j < Lr) {
                        int slot = ring + j; if (slot >= Lr) slot -= Lr;
                        const float2 a = *reinterpret_cast<const float2 *>(&ACC[slot]);
                        const float2 o{a.x + fr[r].x, a.y + fr[r].y};
                        if (j < HOP) {
                            if (emit_out) { float *dst = outp + (long)m * HOP + j; dst[0] = o.x; dst[1] = o.y; }
                        } else {
                            *reinterpret_cast<float2 *>(&ACC[slot]) = o;
                        }
                    } else if (Lr == 0 && emit_out) {                       // hop == N cannot happen here (S_ROWS = 8 covers it)
                    }
                }
                __syncthreads();
#pragma unroll
                for (int r = 0; r < 8; r++) {
                    const int j = 2 * tq + 2 * T * r;
                    if (j >= Lr) {
                        int slot = ring + (j - Lr); if (slot >= Lr) slot -= Lr;
                        *reinterpret_cast<float2 *>(&ACC[slot]) = fr[r];
                    }
                }
                ring += HOP; if (ring >= Lr) ring -= Lr;
            } else {
#pragma unroll
            for (int r = 0; r < (RING ? 0 : S_ROWS); r++) {
                const float2 o{acc[r].x + fr[r].x, acc[r].y + fr[r].y};
                if (emit_out) {
                    float *dst = outp + (long)m * HOP + 2 * tq + 2 * T * r;
                    if (vec_out) __builtin_nontemporal_store(v2f{o.x, o.y}, reinterpret_cast<v2f *>(dst));
                    else { dst[0] = o.x; dst[1] = o.y; }
                }
            }
#pragma unroll
            for (int r = 0; r < LROWS; r++) {
                const int s = r + S_ROWS;
                acc[r] = (s < LROWS) ? float2{acc[s].x + fr[s].x, acc[s].y + fr[s].y} : fr[s];
            }
            }
        }
        __syncthreads();
    }

    WG_STAMP(4);
    if (RING && chunk == (int)gridDim.x - 1) {
        for (int j = t; j < Lr; j += T) {
            int slot = ring + j; if (slot >= Lr) slot -= Lr;
            acc_out[(long)ch * Lr + j] = ACC[slot];
            hist_out[(long)ch * Lr + j] = src.at((long)p.nhops * HOP - Lr + j);
        }
    }
    if (!RING && chunk == (int)gridDim.x - 1) {
#pragma unroll
        for (int r = 0; r < LROWS; r++) {
            float *a = acc_out + (long)ch * (N - HOP) + 2 * t + 2 * T * r;
            a[0] = acc[r].x; a[1] = acc[r].y;
            // the next call's history = rows S_ROWS..7 of the last frame's window = raw[0 .. 8 - S_ROWS) after the slide: from registers
            // (re-reading it costs a streaming quantum an exposed memory -- for 8-channel quanta PCIe -- round trip)
            float *hs = hist_out + (long)ch * (N - HOP) + 2 * t + 2 * T * r;
            hs[0] = raw[r].x; hs[1] = raw[r].y;
        }
    }
    WG_STAMP(5);
    pv_signal_done<true>(p.done, done_seq, (long)ch * gridDim.x + chunk);
    WG_STAMP(6);
    if (RESIDENT) goto resident_top;
}

template <int LOG2N, int S_ROWS, bool AUX>
hipError_t launch_wg(const PvKernelParams &p, int nch, int nchunks, hipStream_t st)
{
    constexpr int G = 1 << (LOG2N - 10);
    static std::atomic<bool> attr_done[16];
    auto k = pv_wg_kernel<LOG2N, S_ROWS, AUX>;
    using CR = WgCfg<G, true>;
    constexpr int lds_bytes = S_ROWS ? WgCfg<G>::LDS_BYTES : CR::LDS_BYTES_RING;
    {
        const hipError_t e = pv_set_dynamic_lds_once(attr_done, reinterpret_cast<const void *>(k), lds_bytes);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k, dim3(nchunks, nch, 1), dim3(64 * G, 1, 1), lds_bytes, st, p);
    return hipGetLastError();
}

template <int LOG2N, int S_ROWS>
hipError_t launch_wg_resident(const PvKernelParams &p, int nslots, hipStream_t st)
{
    constexpr int G = 1 << (LOG2N - 10);
    static std::atomic<bool> attr_done[16];
    auto k = pv_wg_kernel<LOG2N, S_ROWS, false, true>;
    constexpr int lds_bytes = WgCfg<G>::LDS_BYTES;
    {
        const hipError_t e = pv_set_dynamic_lds_once(attr_done, reinterpret_cast<const void *>(k), lds_bytes);
        if (e != hipSuccess) return e;
    }
    PvKernelParams q = p;
    q.nchunks = 1; q.nch = nslots; q.nhops = 1; q.frames_per_chunk = 1;
    hipLaunchKernelGGL(k, dim3(1, nslots, 1), dim3(64 * G, 1, 1), lds_bytes, st, q);
    return hipGetLastError();
}

template <int LOG2N>
hipError_t launch_wg_n(const PvKernelParams &p, int nch, int nchunks, hipStream_t st)
{
    const int N = 1 << LOG2N;
    const bool aux = (p.dbg_mag != nullptr);
    const int rows = (8 * p.hop % N == 0) ? 8 * p.hop / N : 0;
    switch (rows) {
    case 0: return aux ? launch_wg<LOG2N, 0, true>(p, nch, nchunks, st) : launch_wg<LOG2N, 0, false>(p, nch, nchunks, st);
    case 1: return aux ? launch_wg<LOG2N, 1, true>(p, nch, nchunks, st) : launch_wg<LOG2N, 1, false>(p, nch, nchunks, st);
    case 2: return aux ? launch_wg<LOG2N, 2, true>(p, nch, nchunks, st) : launch_wg<LOG2N, 2, false>(p, nch, nchunks, st);
    case 4: return aux ? launch_wg<LOG2N, 4, true>(p, nch, nchunks, st) : launch_wg<LOG2N, 4, false>(p, nch, nchunks, st);
    case 8: return aux ? launch_wg<LOG2N, 8, true>(p, nch, nchunks, st) : launch_wg<LOG2N, 8, false>(p, nch, nchunks, st);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace

bool pv_wg_supported(int log2n, int hop)
{
    if (log2n < 11 || log2n > 13) return false;
    const int N = 1 << log2n;
    if (hop == N / 8 || hop == N / 4 || hop == N / 2 || hop == N) return true;      // register-resident overlap-add
    return hop >= 2 && hop % 2 == 0 && N % hop == 0 && pv_wg_lds_bytes(log2n, hop, true) <= 160 * 1024;   // LDS ring (e.g. native 2048/128)
}

// N = 4096 / 8192 with the register-resident overlap-add run on pv_wg16_kernel.hip (sixteen elements per thread, N/32 threads per frame chain) unless the caller asks
// for the eight-element kernel of this file (wg8: PV_FLAG_WORKGROUP_KERNEL, the second implementation the tests compare with)
static bool use_wg16(int log2n, int hop, bool wg8) { return !wg8 && pv_wg16_supported(log2n, hop); }

size_t pv_wg_lds_bytes(int log2n, int hop, bool wg8)
{
    if (use_wg16(log2n, hop, wg8)) return pv_wg16_lds_bytes(log2n);
    const int N = 1 << log2n;
    const bool ring = !(hop == N / 8 || hop == N / 4 || hop == N / 2 || hop == N);
    switch (log2n) {
    case 11: return ring ? WgCfg<2, true>::LDS_BYTES_RING : WgCfg<2>::LDS_BYTES;
    case 12: return ring ? WgCfg<4, true>::LDS_BYTES_RING : WgCfg<4>::LDS_BYTES;
    case 13: return ring ? WgCfg<8, true>::LDS_BYTES_RING : WgCfg<8>::LDS_BYTES;
    default: return 0;
    }
}

int pv_wg_threads(int log2n, int hop, bool wg8) { return use_wg16(log2n, hop, wg8) ? pv_wg16_threads(log2n) : 64 << (log2n - 10); }

// resident streaming form: N = 8192 (and N = 4096 on pv_wg16_kernel) with the register-resident overlap-add (hop = N/8 .. N)
bool pv_wg_resident_supported(int log2n, int hop, bool wg8)
{
    const int N = 1 << log2n;
    return (log2n == 13 || (log2n == 12 && use_wg16(log2n, hop, wg8))) && (hop == N / 8 || hop == N / 4 || hop == N / 2 || hop == N);
}

hipError_t pv_launch_wg_resident(int log2n, const PvKernelParams &p, int nslots, hipStream_t st, bool wg8)
{
    if (use_wg16(log2n, p.hop, wg8)) return pv_launch_wg16_resident(log2n, p, nslots, st);
    if (log2n != 13) return hipErrorInvalidValue;
    switch (8 * p.hop / (1 << log2n)) {
    case 1: return launch_wg_resident<13, 1>(p, nslots, st);
    case 2: return launch_wg_resident<13, 2>(p, nslots, st);
    case 4: return launch_wg_resident<13, 4>(p, nslots, st);
    case 8: return launch_wg_resident<13, 8>(p, nslots, st);
    default: return hipErrorInvalidValue;
    }
}

hipError_t pv_launch_wg(int log2n, const PvKernelParams &p, int nch, int nchunks, hipStream_t st, bool wg8)
{
    if (use_wg16(log2n, p.hop, wg8)) return pv_launch_wg16(log2n, p, nch, nchunks, st);
    switch (log2n) {
    case 11: return launch_wg_n<11>(p, nch, nchunks, st);
    case 12: return launch_wg_n<12>(p, nch, nchunks, st);
    case 13: return launch_wg_n<13>(p, nch, nchunks, st);
    default: return hipErrorInvalidValue;
    }
}
