// pv_wg16_kernel.hip -- N = 8192 and N = 4096 (BASELINE configs[4] and [3]), hop in {N/8, N/4, N/2, N}: N/32 threads per frame chain (four / two wavefronts), 16 packed
// complex elements per thread, two / four workgroups per CU.
//
// pv_wg_kernel.hip holds an 8192-point frame in eight waves of eight elements: 157 KB of LDS and 250 VGPRs, i.e. ONE workgroup per CU whose waves move in lock-step
// between 13 barriers -- LDS phases idle the VALUs and arithmetic phases idle the LDS (VALU 53 % busy).  Here the same pipeline runs with
//   * 16 elements per thread, T = N/32 threads: M = N/2 = 16 * 16 * K2 (K2 = T/16 = 16 or 8) -- three in-register passes with ONE cross-wave exchange and one exchange
//     inside a wave per transform (the eight-wave kernel: four passes, two cross-wave exchanges and a register transpose), no LDS twiddle table for the per-thread pass
//     (W^{ts k} = products of four loaded powers), no shift table in LDS (a thread computes the shifts of its own 16 candidate bins);
//   * 80 KB (40 KB at N = 4096) of LDS and <= 256 VGPRs without spills: two (four) workgroups per CU, each SIMD holds one wave of either, and the parked time of one
//     workgroup (barriers, exchanges) is another's issue time -- given the phase priorities below.
//
// Layout.  z[n] = xw[2n] + j xw[2n+1], n = n0 + K2 n1 + 16 K2 n2.  Thread t = n1 + 16 n0 (n1 in the low four LANE bits, n0 = wave and lane bits 5..4), register r = n2:
// thread t holds the complex elements ts + T r with ts = (t >> 4) + K2 (t & 15) -- the float2 at samples 2 ts + 2 T r, i.e. the lanes of a load are 128 (64) bytes apart
// and the waves together use every byte of a line (plain loads and stores: L2 merges; measured traffic in profiles/hbm_traffic.json).  With k = k0 + 16 k1 + 256 k2:
//   pass A over r = n2 -> k0, twiddle W_M^{ts k0};     exchange inside groups of 16 lanes (register <-> low lane bits, XOR-swizzled, no barrier): registers n1
//   pass B over n1 -> k1, twiddle W_{M/16}^{n0 k1};    cross-wave exchange (register <-> wave and lane bits 5..4): thread t' = k0 + 16 (k1 mod K2), registers (k1 / K2, n0)
//   pass C over n0 -> k2 (radix 16, or two radix-8 DFTs per thread at N = 4096):  thread t', register r' <-> bin t' + T r'  -- the NATURAL bin order every stage
//   between the transforms wants.
// The inverse runs the same passes backwards (C, cross-wave, twiddle, B, in-wave, twiddle, A) in packed fp32 and lands in the sample layout it started from.
// Everything between the transforms is pv_wg_kernel's / pv_wave2k_kernel's pipeline (16 consecutive bins per thread in the padded magnitude / route layout);
// reference citations are those of pv_kernels.hip / pv_wave_kernel.hip.  DESIGN.md 3c has the measurements and the register story.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "pv_kernels.h"
#include "pv_device_common.h"
#include "pv_pk_math.h"
#ifndef PV_WG16_WPS
#define PV_WG16_WPS 2      // waves per SIMD the product instances are compiled for (measurement builds: 3)
#endif
#ifndef PV_WG16_WPS
#define PV_WG16_WPS 2      // waves per SIMD the product instances are compiled for (measurement builds: 3 -> 168 VGPRs, 99 spilled at N = 8192 / hop N/4: tools/experiments/README.md)
#endif
#ifndef PV_PAIRWISE
#define PV_PAIRWISE 1
#endif
// Reference-width flavour (never the product; `make fp64` -> build/exp/libphaze_fp64.so, DESIGN.md section 4, pv_wave_kernel.hip has the N = 1024 one): with
// -DPV_FP64_FLAVOUR=1 the shifted spectrum, the scatter (plain stores for f >= 1, claim rounds for f < 1), the above-Nyquist residue (always the re-run stage
// structure), the c2r pass and the inverse FFT run in fp64 like the reference (freqComplexBufferShifted / inverseTransform are JS doubles, bundle:102-114;
// phase-vocoder.js:37-39,161-170).  Y, the hand-over and the quarter buffer are twice as wide (135 KB / 68 KB of LDS): ONE workgroup per CU (two at N = 4096), 512 VGPRs.
// It prices the product's fp32 shift / inverse at N = 4096 / 8192 (tests/test_gpu_fp64_flavour.py, one line of bench.py).
#ifndef PV_FP64_FLAVOUR
#define PV_FP64_FLAVOUR 0
#endif

namespace {

constexpr bool FP64W = PV_FP64_FLAVOUR != 0;

// wave priority per phase (pv_wave_fft.h has the story): the SIMD's two waves belong to DIFFERENT workgroups; the latency chains between the transforms (short LDS
// round trips behind barriers) run above the arithmetic of the transforms, whose exchanges are lowest.  One table per size, phases in this order: forward arithmetic, forward
// exchanges, split pass, inverse arithmetic, inverse exchanges, overlap-add, peak search + routes, scatter, c2r pass; 9 = leave unchanged.  Measured (gpurun_out/r04wg16d, e, s, t:
// 25 tables): without priorities C5 at f = 1.5 takes 3.71 instead of 3.09 ms; peak search and scatter must stay on top (3.5 ms at level 2); the c2r pass at level 1 is worth
// 3 % on C4 and 1.6 % on C5's sweep, the split pass at level 2 another 2.5 % on the sweep (and costs C4 what the c2r level gave: N = 4096 keeps it at 3).
#ifndef PV_WG16_PT13
#define PV_WG16_PT13 2, 0, 2, 1, 0, 2, 3, 3, 1
#endif
#ifndef PV_WG16_PT12
#define PV_WG16_PT12 2, 0, 3, 1, 0, 2, 3, 3, 1
#endif
// Phase clock of a THROUGHPUT launch (measurement build only: make variant NAME=wg16ph FILE=pv_wg16_kernel EXTRA=-DPV_WG16_PH CAPI_EXTRA=-DPV_STAMPS=1; tools/read_wg16_phases.py;
// profiles/r05_wg16_phase_clock.md): s_memtime deltas accumulated per phase by every wave, written by wave 0 of a workgroup at the end of its chain.  A mark that
// follows a barrier books the wait at that barrier to the phase it closes.
#ifdef PV_WG16_PH
struct W16Clock {
    unsigned prev, acc[20];
    static __device__ __forceinline__ unsigned now() { __builtin_amdgcn_sched_barrier(0); const unsigned t = (unsigned)__builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); return t; }
    __device__ __forceinline__ void start() { for (int i = 0; i < 20; i++) acc[i] = 0; prev = now(); }
    __device__ __forceinline__ void operator()(int id) { const unsigned t = now(); acc[id] += t - prev; prev = t; }
};
#define W16_MARK(id) w16clk(id)
#define W16_ST , [&](int id) { w16clk(id); }
#define W16_STI , [&](int id) { w16clk(11 + id); }
#else
#define W16_MARK(id)
#define W16_ST
#define W16_STI
#endif
struct W16NoMark { __device__ __forceinline__ void operator()(int) const {} };

template <int PH, int T_> __device__ __forceinline__ void wg16_prio()
{
    constexpr int t13[] = {PV_WG16_PT13}, t12[] = {PV_WG16_PT12};
    constexpr int lvl = (T_ == 256) ? t13[PH] : t12[PH];
    if constexpr (lvl <= 3) __builtin_amdgcn_s_setprio(lvl);
}

template <int LOG2N_>
struct QC {
    static constexpr int LOG2N = LOG2N_, N = 1 << LOG2N_, M = N / 2, H = M + 1, T = M / 16;    // T = 256 (N = 8192) or 128 (N = 4096) threads, 16 elements each
    static constexpr int K2 = T / 16;                     // radix of pass C: M = 16 * 16 * K2
    static constexpr int NW = T / 64;                     // waves
    static constexpr int PR = 5 * T / 4, PM = 5 * M / 4;  // padded layout P(bin) = bin + 4 (bin >> 4): P(t + T r) = pl + PR r, P(M - t - T r) = PM - ql - PR r, P(M) = PM, P(M/2) = PM / 2
    // inside the scratch, between the two FFTs:
    static constexpr int OFF_Y = 0;                       // float2[H]                 | fp32 spectrum stash (f < 1) before Y is zeroed
    static constexpr int MAG0 = 8;                        // magnitudes / routes start 8 words in (bins -2, -1 of thread 0's window stay inside)
    static constexpr int YB = FP64W ? 16 : 8;             // bytes per bin of Y / per element of the quarter and hand-over buffers
    static constexpr int OFF_ROUTE = ((YB * H + 15) / 16) * 16;   // u32 routes | f32 mags, both padded (pv_wave2k_kernel.hip) | u32 claim words [H] (plain)
    static constexpr int ROUTE_WORDS = MAG0 + PM + 8;
    static constexpr int OFF_RESQ = ((OFF_ROUTE + 4 * H + 15) / 16) * 16;   // float2[N / 4] one residue quarter | c2r hand-over
    static constexpr int SCRATCH = OFF_RESQ + (YB / 4) * N;   // 65568 / 32800
    static_assert(OFF_ROUTE + 4 * ROUTE_WORDS <= SCRATCH && 16 * M <= SCRATCH, "routes / the fp64 exchange");
    static_assert(OFF_ROUTE >= 16 * 8 * T, "MAG must not alias the partner rows of the split pass");
    // i32 LASTIN[T], FIRSTIN[T].  N = 4096: inside the quarter buffer, behind the tail of the padded magnitudes (free between the end of the previous frame's inverse and
    // this frame's scatter / c2r hand-over) -- behind the scratch they are the 1 KB that costs the fourth workgroup of the CU; N = 8192 keeps them behind the scratch
    // (measured: the pitch sweep 2 % faster that way)
    static constexpr bool NEAR_IN = (LOG2N_ == 12) && !FP64W;
    static constexpr int OFF_NEAR = NEAR_IN ? OFF_RESQ + N : SCRATCH;
    static_assert(!NEAR_IN || (OFF_NEAR >= OFF_ROUTE + 4 * ROUTE_WORDS && OFF_NEAR + 8 * T <= SCRATCH), "LASTIN / FIRSTIN");
    // after the scratch:
    static constexpr int OFF_OCC = SCRATCH + (NEAR_IN ? 0 : 8 * T);               // u64[4] occupancy | u32[4] "a gap fails the pairwise test" | broadcast word (resident)
    static constexpr int OFF_TWB = OFF_OCC + 64;          // double2[15][K2]  W_{M/16}^{n0 k1}, k1 = 1..15
    static constexpr int OFF_TWBF = OFF_TWB + 16 * 15 * K2;   // float2[15][K2] conj, fp32
    static constexpr int OFF_XQ = OFF_TWBF + 8 * 15 * K2;   // f32[N / 4] windowed samples xw[4n + 2] of the frame (f < 0.75): base stage of the general residue
    static constexpr int LDS_BYTES = OFF_XQ + N;          // 81632 (two workgroups per CU) / 39840 (four)
    static_assert(LOG2N_ == 11 || (LDS_BYTES + 256 + 511) / 512 * 512 * (FP64W ? 256 / T : 512 / T) <= 160 * 1024, "workgroups per CU (256 static bytes: __syncthreads_or)");
};

// ---- radix-16 butterflies, natural order in and out: X[q + 4 p] = sum_j W4^{j p} ( W16^{j q} sum_m a[j + 4 m] W4^{m q} ) ----
__device__ __forceinline__ double2 dmul(double2 a, double2 w)          // a * w, the roundings spelled out (instances must agree bit for bit)
{
    return double2{__fma_rn(a.x, w.x, -__dmul_rn(a.y, w.y)), __fma_rn(a.x, w.y, __dmul_rn(a.y, w.x))};
}
__device__ __forceinline__ double2 dmulc(double2 a, double2 w) { return dmul(a, double2{w.x, -w.y}); }   // a * conj(w)

__device__ __forceinline__ void radix16_fwd(double2 (&a)[16])
{
    const double c = 0.92387953251128675613, s = 0.38268343236508977173, h = 0.70710678118654752440;
    double2 b[16];                                                      // b[4 j + q]
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const double2 x0 = a[j], x1 = a[j + 4], x2 = a[j + 8], x3 = a[j + 12];
        const double2 s02 = cadd(x0, x2), d02 = csub(x0, x2), s13 = cadd(x1, x3), d13 = csub(x1, x3);
        b[4 * j] = cadd(s02, s13);
        b[4 * j + 1] = double2{d02.x + d13.y, d02.y - d13.x};          // d02 - j d13
        b[4 * j + 2] = csub(s02, s13);
        b[4 * j + 3] = double2{d02.x - d13.y, d02.y + d13.x};          // d02 + j d13
    }
    // W16^{j q} = exp(-2 pi j (j q) / 16); the four eighth-root twiddles (e = 2, 6) keep their factor h for the second stage, which folds it into its additions (u = b W / h)
    b[5] = dmul(b[5], double2{c, -s});                                  // e = 1
    const double2 u6{b[6].x + b[6].y, b[6].y - b[6].x};                 // e = 2: (1 - j) b
    b[7] = dmul(b[7], double2{s, -c});                                  // e = 3
    const double2 u9{b[9].x + b[9].y, b[9].y - b[9].x};                 // e = 2
    b[10] = double2{b[10].y, -b[10].x};                                 // e = 4: -j
    const double2 u11{b[11].y - b[11].x, -(b[11].x + b[11].y)};         // e = 6: (-1 - j) b
    b[13] = dmul(b[13], double2{s, -c});                                // e = 3
    const double2 u14{b[14].y - b[14].x, -(b[14].x + b[14].y)};         // e = 6
    b[15] = dmul(b[15], double2{-c, s});                                // e = 9
    {   // q = 0: no twiddles
        const double2 s02 = cadd(b[0], b[8]), d02 = csub(b[0], b[8]), s13 = cadd(b[4], b[12]), d13 = csub(b[4], b[12]);
        a[0] = cadd(s02, s13); a[4] = double2{d02.x + d13.y, d02.y - d13.x}; a[8] = csub(s02, s13); a[12] = double2{d02.x - d13.y, d02.y + d13.x};
    }
    {   // q = 1: x2 = h u9
        const double2 x0 = b[1], x1 = b[5], x3 = b[13];
        const double2 s02{__fma_rn(h, u9.x, x0.x), __fma_rn(h, u9.y, x0.y)}, d02{__fma_rn(-h, u9.x, x0.x), __fma_rn(-h, u9.y, x0.y)};
        const double2 s13 = cadd(x1, x3), d13 = csub(x1, x3);
        a[1] = cadd(s02, s13); a[5] = double2{d02.x + d13.y, d02.y - d13.x}; a[9] = csub(s02, s13); a[13] = double2{d02.x - d13.y, d02.y + d13.x};
    }
    {   // q = 2: x1 = h u6, x3 = h u14
        const double2 s02 = cadd(b[2], b[10]), d02 = csub(b[2], b[10]);
        const double2 S{u6.x + u14.x, u6.y + u14.y}, D{u6.x - u14.x, u6.y - u14.y};
        a[2] = double2{__fma_rn(h, S.x, s02.x), __fma_rn(h, S.y, s02.y)};
        a[10] = double2{__fma_rn(-h, S.x, s02.x), __fma_rn(-h, S.y, s02.y)};
        a[6] = double2{__fma_rn(h, D.y, d02.x), __fma_rn(-h, D.x, d02.y)};    // d02 - j h D
        a[14] = double2{__fma_rn(-h, D.y, d02.x), __fma_rn(h, D.x, d02.y)};   // d02 + j h D
    }
    {   // q = 3: x2 = h u11
        const double2 x0 = b[3], x1 = b[7], x3 = b[15];
        const double2 s02{__fma_rn(h, u11.x, x0.x), __fma_rn(h, u11.y, x0.y)}, d02{__fma_rn(-h, u11.x, x0.x), __fma_rn(-h, u11.y, x0.y)};
        const double2 s13 = cadd(x1, x3), d13 = csub(x1, x3);
        a[3] = cadd(s02, s13); a[7] = double2{d02.x + d13.y, d02.y - d13.x}; a[11] = csub(s02, s13); a[15] = double2{d02.x - d13.y, d02.y + d13.x};
    }
}

// the inverse instance (exp(+2 pi j n k / 16)) in packed fp32
__device__ __forceinline__ void radix16_inv_pk(pk::c32 (&a)[16])
{
    const float c = 0.92387953251128675613f, s = 0.38268343236508977173f, h = 0.70710678118654752440f;
    const pk::c32 hh{h, h};
    pk::c32 b[16];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const pk::c32 x0 = a[j], x1 = a[j + 4], x2 = a[j + 8], x3 = a[j + 12];
        const pk::c32 s02 = pk::add(x0, x2), d02 = pk::sub(x0, x2), s13 = pk::add(x1, x3), d13 = pk::sub(x1, x3);
        b[4 * j] = pk::add(s02, s13);
        b[4 * j + 1] = pk::add_j(d02, d13);                             // d02 + j d13
        b[4 * j + 2] = pk::sub(s02, s13);
        b[4 * j + 3] = pk::sub_j(d02, d13);
    }
    b[5] = pk::cmul(b[5], pk::c32{c, s});                               // exp(+2 pi j / 16)
    b[6] = pk::mul(pk::add_j(b[6], b[6]), hh);                          // (1 + j) h
    b[7] = pk::cmul(b[7], pk::c32{s, c});
    b[9] = pk::mul(pk::add_j(b[9], b[9]), hh);
    b[10] = pk::c32{-b[10].y, b[10].x};                                 // + j
    b[11] = pk::mul(pk::neg_add_j(b[11], b[11]), hh);                   // (-1 + j) h
    b[13] = pk::cmul(b[13], pk::c32{s, c});
    b[14] = pk::mul(pk::neg_add_j(b[14], b[14]), hh);
    b[15] = pk::cmul(b[15], pk::c32{-c, -s});                           // exp(+2 pi j 9 / 16)
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const pk::c32 x0 = b[q], x1 = b[4 + q], x2 = b[8 + q], x3 = b[12 + q];
        const pk::c32 s02 = pk::add(x0, x2), d02 = pk::sub(x0, x2), s13 = pk::add(x1, x3), d13 = pk::sub(x1, x3);
        a[q] = pk::add(s02, s13);
        a[q + 4] = pk::add_j(d02, d13);
        a[q + 8] = pk::sub(s02, s13);
        a[q + 12] = pk::sub_j(d02, d13);
    }
}

// The fifteen per-thread twiddles of pass A from four of them: W^{ts k}, k = 1, 2, 4, 8 (exact table entries), the rest by products of at most three factors.
struct TwA { double2 w1, w2, w4, w8; };

// M = 4096-point complex FFT across the workgroup: in  thread t, reg r <-> element ts + 256 r (ts = (t >> 4) + 16 (t & 15)),
//                                                     out thread t, reg r <-> bin t + 256 r.  S: the 64 KB scratch.
template <int T_, typename ST = W16NoMark>
__device__ __forceinline__ void fft_wg16(double2 (&a)[16], double2 *S, const TwA &tw, const double2 *TWB, int t, ST st = ST{})
{
    radix16_fwd(a);
    {
        const double2 w3 = dmul(tw.w1, tw.w2), w5 = dmul(tw.w4, tw.w1), w6 = dmul(tw.w4, tw.w2), w7 = dmul(tw.w4, w3);
        a[1] = dmul(a[1], tw.w1); a[2] = dmul(a[2], tw.w2); a[3] = dmul(a[3], w3); a[4] = dmul(a[4], tw.w4);
        a[5] = dmul(a[5], w5); a[6] = dmul(a[6], w6); a[7] = dmul(a[7], w7); a[8] = dmul(a[8], tw.w8);
        a[9] = dmul(a[9], dmul(tw.w8, tw.w1)); a[10] = dmul(a[10], dmul(tw.w8, tw.w2)); a[11] = dmul(a[11], dmul(tw.w8, w3));
        a[12] = dmul(a[12], dmul(tw.w8, tw.w4)); a[13] = dmul(a[13], dmul(tw.w8, w5)); a[14] = dmul(a[14], dmul(tw.w8, w6)); a[15] = dmul(a[15], dmul(tw.w8, w7));
    }
    // exchange inside the groups of 16 lanes: [reg k0][lane n1] -> [reg n1][lane k0]; element (k0, n1) of group g at 256 g + 16 k0 + (n1 ^ k0)
    const int c = t & 15, g = t >> 4;
    double2 *Sg = S + 256 * g;
    st(0);
    wg16_prio<1, T_>();
#pragma unroll
    for (int k = 0; k < 16; k++) Sg[16 * k + (c ^ k)] = a[k];
    wave_sync();                                                        // the 16 lanes of a group sit in one wave: LDS traffic of a wave executes in order
#pragma unroll
    for (int n = 0; n < 16; n++) a[n] = Sg[16 * c + (n ^ c)];
    st(1);
    wg16_prio<0, T_>();
    radix16_fwd(a);
    constexpr int K2 = T_ / 16;
#pragma unroll
    for (int k = 1; k < 16; k++) a[k] = dmul(a[k], TWB[(k - 1) * K2 + g]);
    st(2);
    wg16_prio<1, T_>();
    __syncthreads();                                                    // every wave is done with its in-wave exchange: the rows below overwrite other waves' groups
#pragma unroll
    for (int k = 0; k < 16; k++) S[T_ * k + t] = a[k];                  // [reg k1][thread (n0, k0)]
    __syncthreads();
    // thread t = k0 + 16 (k1 mod K2) takes (k1 = g + K2 h, n0 = n, k0 = c) into register K2 h + n (K2 = 16: h = 0)
#pragma unroll
    for (int h = 0; h < 16 / K2; h++)
#pragma unroll
        for (int n = 0; n < K2; n++) a[K2 * h + n] = S[T_ * (g + K2 * h) + 16 * n + c];
    __syncthreads();                                                    // the scratch is free again
    st(3);
    wg16_prio<0, T_>();
    if (K2 == 16) {
        radix16_fwd(a);
    } else if (K2 == 8) {                                               // two radix-8 DFTs over n0; bin t + T (h + 2 k2) <- register 8 h + k2
        double2 lo[8], hi[8];
#pragma unroll
        for (int n = 0; n < 8; n++) { lo[n] = a[n]; hi[n] = a[8 + n]; }
        radix8<double, false>(lo);
        radix8<double, false>(hi);
#pragma unroll
        for (int k = 0; k < 8; k++) { a[2 * k] = lo[k]; a[2 * k + 1] = hi[k]; }
    } else {                                                            // K2 = 4 (N = 2048, the fp64 flavour only): four radix-4 DFTs over n0; bin t + T (h + 4 k2) <- register 4 h + k2
        double2 b[16];
#pragma unroll
        for (int h = 0; h < 4; h++) {
            const double2 s02 = cadd(a[4 * h], a[4 * h + 2]), d02 = csub(a[4 * h], a[4 * h + 2]), s13 = cadd(a[4 * h + 1], a[4 * h + 3]), d13 = csub(a[4 * h + 1], a[4 * h + 3]);
            b[h] = cadd(s02, s13); b[h + 4] = double2{d02.x + d13.y, d02.y - d13.x}; b[h + 8] = csub(s02, s13); b[h + 12] = double2{d02.x - d13.y, d02.y + d13.x};
        }
#pragma unroll
        for (int r = 0; r < 16; r++) a[r] = b[r];
    }
    st(4);
}

// The inverse in packed fp32, the same passes backwards: in thread t, reg r <-> bin t + 256 r, out thread t, reg r <-> element ts + 256 r.
// The caller guarantees that nobody still reads the first 32 KB of the scratch; the data of the cross-wave exchange is consumed before the function returns its
// in-wave exchange, which lives in the SECOND 32 KB (no barrier between the two).
struct TwAf { pk::c32 w1, w2, w4, w8; };
template <int T_, typename ST = W16NoMark>
__device__ __forceinline__ void fft_wg16_inv_pk(pk::c32 (&a)[16], pk::c32 *S, const TwAf &tw, const pk::c32 *TWBF, int t, ST st = ST{})
{
    const int c = t & 15, g = t >> 4;
    constexpr int K2 = T_ / 16;
    wg16_prio<3, T_>();
    if (K2 == 16) {
        radix16_inv_pk(a);
    } else if (K2 == 8) {                                               // register 8 h + k2 <- bin t + T (h + 2 k2); two radix-8 inverse DFTs over k2 -> n0
        pk::c32 lo[8], hi[8];
#pragma unroll
        for (int k = 0; k < 8; k++) { lo[k] = a[2 * k]; hi[k] = a[2 * k + 1]; }
        pk::radix8_inv(lo);
        pk::radix8_inv(hi);
#pragma unroll
        for (int n = 0; n < 8; n++) { a[n] = lo[n]; a[8 + n] = hi[n]; }
    } else {                                                            // K2 = 4: register 4 h + k2 <- bin t + T (h + 4 k2); four radix-4 inverse DFTs over k2 -> n0
        pk::c32 b[16];
#pragma unroll
        for (int h = 0; h < 4; h++) {
            const pk::c32 s02 = pk::add(a[h], a[h + 8]), d02 = pk::sub(a[h], a[h + 8]), s13 = pk::add(a[h + 4], a[h + 12]), d13 = pk::sub(a[h + 4], a[h + 12]);
            b[4 * h] = pk::add(s02, s13); b[4 * h + 1] = pk::add_j(d02, d13); b[4 * h + 2] = pk::sub(s02, s13); b[4 * h + 3] = pk::sub_j(d02, d13);
        }
#pragma unroll
        for (int r = 0; r < 16; r++) a[r] = b[r];
    }
    st(0);
    wg16_prio<4, T_>();
#pragma unroll
    for (int h = 0; h < 16 / K2; h++)
#pragma unroll
        for (int n = 0; n < K2; n++) S[T_ * (g + K2 * h) + 16 * n + c] = a[K2 * h + n];   // thread (k1 = g + K2 h, k0 = c), reg n0 = n
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; k++) a[k] = S[T_ * k + t];                  // thread (n0 = g, k0 = c), reg k1
    st(1);
    wg16_prio<3, T_>();
#pragma unroll
    for (int k = 1; k < 16; k++) a[k] = pk::cmul(a[k], TWBF[(k - 1) * K2 + g]);
    radix16_inv_pk(a);
    pk::c32 *Sg = S + 16 * T_ + 256 * g;                                // second half of the scratch (fp32 elements are half the size)
    st(2);
    wg16_prio<4, T_>();
#pragma unroll
    for (int n = 0; n < 16; n++) Sg[16 * c + (n ^ c)] = a[n];           // thread (n0, k0 = c), reg n1 = n
    wave_sync();
#pragma unroll
    for (int k = 0; k < 16; k++) a[k] = Sg[16 * k + (c ^ k)];           // thread (n0, n1 = c), reg k0
    st(3);
    wg16_prio<3, T_>();
    {
        const pk::c32 w3 = pk::cmul(tw.w1, tw.w2), w5 = pk::cmul(tw.w4, tw.w1), w6 = pk::cmul(tw.w4, tw.w2), w7 = pk::cmul(tw.w4, w3);
        a[1] = pk::cmul(a[1], tw.w1); a[2] = pk::cmul(a[2], tw.w2); a[3] = pk::cmul(a[3], w3); a[4] = pk::cmul(a[4], tw.w4);
        a[5] = pk::cmul(a[5], w5); a[6] = pk::cmul(a[6], w6); a[7] = pk::cmul(a[7], w7); a[8] = pk::cmul(a[8], tw.w8);
        a[9] = pk::cmul(a[9], pk::cmul(tw.w8, tw.w1)); a[10] = pk::cmul(a[10], pk::cmul(tw.w8, tw.w2)); a[11] = pk::cmul(a[11], pk::cmul(tw.w8, w3));
        a[12] = pk::cmul(a[12], pk::cmul(tw.w8, tw.w4)); a[13] = pk::cmul(a[13], pk::cmul(tw.w8, w5)); a[14] = pk::cmul(a[14], pk::cmul(tw.w8, w6));
        a[15] = pk::cmul(a[15], pk::cmul(tw.w8, w7));
    }
    radix16_inv_pk(a);
    st(4);
}

// ---- fp64 flavour: the inverse in doubles.  The radix-16 inverse DFT is the forward one between two conjugations (sign flips); the passes are fft_wg16_inv_pk's, the
//      twiddles the forward tables' conjugates, and the in-wave exchange reuses the scratch of the cross-wave exchange behind a barrier (the flavour does not race) ----
__device__ __forceinline__ void radix16_inv_d(double2 (&a)[16])
{
#pragma unroll
    for (int i = 0; i < 16; i++) a[i].y = -a[i].y;
    radix16_fwd(a);
#pragma unroll
    for (int i = 0; i < 16; i++) a[i].y = -a[i].y;
}
template <int T_>
__device__ __forceinline__ void fft_wg16_inv_d(double2 (&a)[16], double2 *S, const TwA &tw, const double2 *TWB, int t)
{
    const int c = t & 15, g = t >> 4;
    constexpr int K2 = T_ / 16;
    if (K2 == 16) {
        radix16_inv_d(a);
    } else if (K2 == 8) {
        double2 lo[8], hi[8];
#pragma unroll
        for (int k = 0; k < 8; k++) { lo[k] = a[2 * k]; hi[k] = a[2 * k + 1]; }
        radix8<double, true>(lo);
        radix8<double, true>(hi);
#pragma unroll
        for (int n = 0; n < 8; n++) { a[n] = lo[n]; a[8 + n] = hi[n]; }
    } else {
        double2 b[16];
#pragma unroll
        for (int h = 0; h < 4; h++) {
            const double2 s02 = cadd(a[h], a[h + 8]), d02 = csub(a[h], a[h + 8]), s13 = cadd(a[h + 4], a[h + 12]), d13 = csub(a[h + 4], a[h + 12]);
            b[4 * h] = cadd(s02, s13); b[4 * h + 1] = double2{d02.x - d13.y, d02.y + d13.x}; b[4 * h + 2] = csub(s02, s13); b[4 * h + 3] = double2{d02.x + d13.y, d02.y - d13.x};
        }
#pragma unroll
        for (int r = 0; r < 16; r++) a[r] = b[r];
    }
#pragma unroll
    for (int h = 0; h < 16 / K2; h++)
#pragma unroll
        for (int n = 0; n < K2; n++) S[T_ * (g + K2 * h) + 16 * n + c] = a[K2 * h + n];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; k++) a[k] = S[T_ * k + t];
#pragma unroll
    for (int k = 1; k < 16; k++) a[k] = dmulc(a[k], TWB[(k - 1) * K2 + g]);
    radix16_inv_d(a);
    __syncthreads();                                                    // every row of the cross-wave exchange has been read
    double2 *Sg = S + 256 * g;
#pragma unroll
    for (int n = 0; n < 16; n++) Sg[16 * c + (n ^ c)] = a[n];
    wave_sync();
#pragma unroll
    for (int k = 0; k < 16; k++) a[k] = Sg[16 * k + (c ^ k)];
    {
        const double2 w3 = dmul(tw.w1, tw.w2), w5 = dmul(tw.w4, tw.w1), w6 = dmul(tw.w4, tw.w2), w7 = dmul(tw.w4, w3);
        a[1] = dmulc(a[1], tw.w1); a[2] = dmulc(a[2], tw.w2); a[3] = dmulc(a[3], w3); a[4] = dmulc(a[4], tw.w4);
        a[5] = dmulc(a[5], w5); a[6] = dmulc(a[6], w6); a[7] = dmulc(a[7], w7); a[8] = dmulc(a[8], tw.w8);
        a[9] = dmulc(a[9], dmul(tw.w8, tw.w1)); a[10] = dmulc(a[10], dmul(tw.w8, tw.w2)); a[11] = dmulc(a[11], dmul(tw.w8, w3));
        a[12] = dmulc(a[12], dmul(tw.w8, tw.w4)); a[13] = dmulc(a[13], dmul(tw.w8, w5)); a[14] = dmulc(a[14], dmul(tw.w8, w6)); a[15] = dmulc(a[15], dmul(tw.w8, w7));
    }
    radix16_inv_d(a);
}

// v * exp(+2 pi j ridx / N) in doubles (rotate_route's fp64 twin)
template <int R_, int LOG2N_>
__device__ __forceinline__ double2 rotate_route_d(unsigned route, double2 v, const double2 *__restrict__ tw64)
{
    const unsigned ridx = (route >> 16) & ((1u << LOG2N_) - 1u);
    if (R_ == 4) { const unsigned q = ridx >> (LOG2N_ - 2); return q == 0 ? v : q == 1 ? double2{-v.y, v.x} : q == 2 ? double2{-v.x, -v.y} : double2{v.y, -v.x}; }   // j^q exactly
    const double2 w = tw64[ridx];
    return double2{__fma_rn(v.x, w.x, __dmul_rn(v.y, w.y)), __fma_rn(v.y, w.x, -__dmul_rn(v.x, w.y))};
}
template <int R_, int LOG2N_> __device__ __forceinline__ float2 rotate_any(unsigned route, float2 v, const float2 *tw32, const double2 *) { return rotate_route<R_, LOG2N_>(route, v, tw32); }
template <int R_, int LOG2N_> __device__ __forceinline__ double2 rotate_any(unsigned route, double2 v, const float2 *, const double2 *tw64) { return rotate_route_d<R_, LOG2N_>(route, v, tw64); }

// o * W_32^r (fp64, split pass) and o * exp(+2 pi j r / 32) (packed fp32, c2r pass), r = 0..7 compile-time: the row part of W_8192^{t + 256 r}
__device__ __forceinline__ double2 mul_w32_16(double2 o, int r)
{
    constexpr double c[9] = {1.0, 0.98078528040323044913, 0.92387953251128675613, 0.83146961230254523708, 0.70710678118654752440,
                             0.55557023301960222474, 0.38268343236508977173, 0.19509032201612826785, 0.0};
    if (r == 0) return o;
    return cmul(o, double2{c[r], -c[8 - r]});
}
__device__ __forceinline__ pk::c32 mul_w32_inv_pk16(pk::c32 o, int r)
{
    constexpr float c[9] = {1.0f, 0.98078528040323044913f, 0.92387953251128675613f, 0.83146961230254523708f, 0.70710678118654752440f,
                            0.55557023301960222474f, 0.38268343236508977173f, 0.19509032201612826785f, 0.0f};
    if (r == 0) return o;
    return pk::cmul(o, pk::c32{c[r], c[8 - r]});
}

// Workgroup-wide claim rounds (pv_wg_kernel.hip): atomic MIN on the claim word, the smallest pending source bin wins the round.  CLAIM[0..H) all-ones on entry and exit.
template <int NS, int H_, typename V2>
__device__ __forceinline__ void claim_rounds_wg16(const unsigned (&rt)[NS], const V2 (&ys)[NS], const int (&id)[NS], V2 *Y, unsigned *CLAIM)
{
    unsigned pend = 0;
    unsigned tg[NS];
#pragma unroll
    for (int r = 0; r < NS; r++) {
        const unsigned t = rt[r] & 0xFFFFu;
        const bool ok = t < (unsigned)H_;
        pend |= ok ? (1u << r) : 0u;
        tg[r] = ok ? t : 0u;
    }
    while (__syncthreads_or(pend != 0u)) {
#pragma unroll
        for (int r = 0; r < NS; r++) if (pend & (1u << r)) atomicMin(&CLAIM[tg[r]], (unsigned)id[r]);
        __syncthreads();
        unsigned c[NS];
        V2 o[NS];
#pragma unroll
        for (int r = 0; r < NS; r++) c[r] = CLAIM[tg[r]];
#pragma unroll
        for (int r = 0; r < NS; r++) o[r] = Y[tg[r]];
#pragma unroll
        for (int r = 0; r < NS; r++) {
            if ((pend & (1u << r)) && c[r] == (unsigned)id[r]) {
                Y[tg[r]] = V2{o[r].x + ys[r].x, o[r].y + ys[r].y};
                CLAIM[tg[r]] = 0xFFFFFFFFu;
                pend &= ~(1u << r);
            }
        }
    }
}

__device__ __forceinline__ int digitrev4_16(int v, int nd)
{
    const unsigned r = __brev((unsigned)v) >> (32 - 2 * nd);
    return (int)(((r & 0x55555555u) << 1) | ((r >> 1) & 0x55555555u));
}

// Rare path: above-Nyquist residue of fft.js's in-place real DIT, one quarter of the buffer at a time (log2 N odd: radix-2 base blocks, bundle:447-463,
// then the radix-4 stages with their predicated stores, bundle:329-441), then its sources are added into Y.  See residue_scatter_wg.
template <int LOG2N, int R_>
__device__ __attribute__((noinline)) PV_NO_DS_MERGE void residue_scatter_wg16(const float *in, const float *hist, int hist_len, bool sys, long s0, const float *__restrict__ hann,
                                                                               const float2 *__restrict__ tw32, int t, int upper_end, int up_delta, unsigned up_ridx,
                                                                               double *dbg_X, bool plain
#ifdef PV_WG16_PH
                                                                               , unsigned *ph_row
#endif
                                                                               )
{
    using C = QC<LOG2N>;
    constexpr int N = C::N, H = C::H, T = C::T, QN = N / 4;
    constexpr bool BASE4 = (LOG2N % 2) == 0;
#ifdef PV_WG16_PH
    unsigned ph_prev = W16Clock::now();
    auto ph_mark = [&](int k) { const unsigned n = W16Clock::now(); if (t == 0 && ph_row) atomicAdd(&ph_row[22 + k], n - ph_prev); ph_prev = n; };
#define RES_MARK(k) ph_mark(k)
#else
#define RES_MARK(k)
#endif
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float2 *Y = reinterpret_cast<float2 *>(smem + C::OFF_Y);
    unsigned *CLAIM = reinterpret_cast<unsigned *>(smem + C::OFF_ROUTE);
    float2 *Q = reinterpret_cast<float2 *>(smem + C::OFF_RESQ);
    const float *XQ = reinterpret_cast<const float *>(smem + C::OFF_XQ);
    const WaveSrc src{in, hist, hist_len, sys};
    for (int base = N / 2; base < N && base < upper_end; base += QN) {
        if (BASE4) {
            constexpr int nd = (LOG2N - 2) / 2;
#pragma unroll
            for (int i = 0; i < 2; i++) {                                  // QN / 4 = 2T radix-4 blocks per quarter (bundle:468-508)
                const int lb = t + T * i;
                float a, b, c, d;
                if (base == N / 2) {                                       // quarter 2: sub-FFT of xw[4n + 2], from the frame's stash
                    const int off = digitrev4_16(N / 8 + lb, nd);          // = 2 (mod 4): sample off + q N/4 is XQ[(off - 2) / 4 + q N/16]
                    a = XQ[(off - 2) >> 2]; b = XQ[((off - 2) >> 2) + N / 16]; c = XQ[((off - 2) >> 2) + N / 8]; d = XQ[((off - 2) >> 2) + 3 * N / 16];
                } else {
                    const int off = digitrev4_16(base / 4 + lb, nd);
                    a = mul_rounded(src.at(s0 + off), hann[off]); b = mul_rounded(src.at(s0 + off + N / 4), hann[off + N / 4]);
                    c = mul_rounded(src.at(s0 + off + N / 2), hann[off + N / 2]); d = mul_rounded(src.at(s0 + off + 3 * N / 4), hann[off + 3 * N / 4]);
                }
                const float t0 = a + c, t1 = a - c, t2 = b + d, t3 = b - d;
                Q[4 * lb] = float2{t0 + t2, 0.f};
                Q[4 * lb + 1] = float2{t1, -t3};
                Q[4 * lb + 2] = float2{t0 - t2, 0.f};
                Q[4 * lb + 3] = float2{t1, t3};
            }
        } else {
            constexpr int nd = (LOG2N - 1) / 2;
#pragma unroll
            for (int i = 0; i < 4; i++) {                                  // QN / 2 = 4T radix-2 blocks per quarter (bundle:447-463)
                const int lb = t + T * i;
                const int off = digitrev4_16((base == N / 2 ? N / 4 : base / 2) + lb, nd);
                float a, b;
                if (base == N / 2) { a = XQ[(off - 2) >> 2]; b = XQ[((off - 2) >> 2) + N / 8]; }     // quarter 2: sub-FFT of xw[4n + 2], from the frame's stash
                else { a = mul_rounded(src.at(s0 + off), hann[off]); b = mul_rounded(src.at(s0 + off + N / 2), hann[off + N / 2]); }
                Q[2 * lb] = float2{a + b, 0.f};
                Q[2 * lb + 1] = float2{a - b, 0.f};
            }
        }
        __syncthreads();
        RES_MARK(0);
        constexpr int LOG2BASE = BASE4 ? 2 : 1;
        // (round 5: the block / index split of an item is a shift and a mask instead of a division by a run-time power of two -- a stage is ~1-2 butterflies per thread
        //  and is bound by the instructions around them, profiles/r05_wg16_phase_clock.md.  Unrolling the stages makes their sizes constants, but the function then needs
        //  more registers than the kernel that calls it: the kernel's occupancy follows its callees')
#pragma unroll 1
        for (int log2m = LOG2BASE + 2; log2m <= LOG2N - 2; log2m += 2) {  // block sizes 4 * base .. N/4 inside the quarter
            const int q = (1 << log2m) >> 2, hq = q >> 1;
            const int nblocks = QN >> log2m;
            const int tws = LOG2N - log2m;
            for (int u = t; u < nblocks * (hq + 1); u += T) {
                int blk, i;
                if (u < nblocks * hq) { blk = u >> (log2m - 3); i = u & (hq - 1); } else { blk = u - nblocks * hq; i = hq; }
                const int o = blk << log2m;
                const float2 A = Q[o + i];
                const float2 w1 = tw32[i << tws];
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
            RES_MARK(1 + (log2m - LOG2BASE - 2) / 2);
        }
        if (dbg_X)
            for (int i = t; i < QN; i += T) if (base + i >= H) { dbg_X[2 * (base + i)] = Q[i].x; dbg_X[2 * (base + i) + 1] = Q[i].y; }
        unsigned rt[8];
        float2 ys[8];
        int id[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int b = base + t + T * j, tgt = b + up_delta;
            rt[j] = (b >= H && b < upper_end && tgt >= 0 && tgt < H) ? ((up_ridx << 16) | (unsigned)tgt) : NOROUTE;
            ys[j] = rotate_route<R_, LOG2N>(rt[j], Q[t + T * j], tw32);
            id[j] = b - N / 2;
        }
        if (plain) {
#pragma unroll
            for (int j = 0; j < 8; j++) if (rt[j] != NOROUTE) Y[rt[j] & 0xFFFFu] = ys[j];
        } else {
            claim_rounds_wg16<8, H>(rt, ys, id, Y, CLAIM);
        }
        __syncthreads();
        RES_MARK(7);
    }
}

// fp64 flavour: the same stage structure in doubles -- every quarter from the source samples (the window product rounded to fp32 as the reference's Float32Array does,
// pv:55), the three stage twiddles table entries like fft.js's (bundle:329-441), the sources added by claim rounds.
template <int LOG2N, int R_>
__device__ __attribute__((noinline)) PV_NO_DS_MERGE void residue_scatter_wg16_d(const float *in, const float *hist, int hist_len, bool sys, long s0, const float *__restrict__ hann,
                                                                                 const double2 *__restrict__ tw64, int t, int upper_end, int up_delta, unsigned up_ridx, double *dbg_X)
{
    using C = QC<LOG2N>;
    constexpr int N = C::N, H = C::H, T = C::T, QN = N / 4;
    constexpr bool BASE4 = (LOG2N % 2) == 0;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double2 *Y = reinterpret_cast<double2 *>(smem + C::OFF_Y);
    unsigned *CLAIM = reinterpret_cast<unsigned *>(smem + C::OFF_ROUTE);
    double2 *Q = reinterpret_cast<double2 *>(smem + C::OFF_RESQ);
    const WaveSrc src{in, hist, hist_len, sys};
    auto xw = [&](int off) -> double { return (double)mul_rounded(src.at(s0 + off), hann[off]); };
    for (int base = N / 2; base < N && base < upper_end; base += QN) {
        if (BASE4) {
            constexpr int nd = (LOG2N - 2) / 2;
#pragma unroll
            for (int i = 0; i < 2; i++) {
                const int lb = t + T * i;
                const int off = digitrev4_16(base / 4 + lb, nd);
                const double a = xw(off), b = xw(off + N / 4), c = xw(off + N / 2), d = xw(off + 3 * N / 4);
                const double t0 = a + c, t1 = a - c, t2 = b + d, t3 = b - d;
                Q[4 * lb] = double2{t0 + t2, 0.0};
                Q[4 * lb + 1] = double2{t1, -t3};
                Q[4 * lb + 2] = double2{t0 - t2, 0.0};
                Q[4 * lb + 3] = double2{t1, t3};
            }
        } else {
            constexpr int nd = (LOG2N - 1) / 2;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int lb = t + T * i;
                const int off = digitrev4_16(base / 2 + lb, nd);
                const double a = xw(off), b = xw(off + N / 2);
                Q[2 * lb] = double2{a + b, 0.0};
                Q[2 * lb + 1] = double2{a - b, 0.0};
            }
        }
        __syncthreads();
        constexpr int LOG2BASE = BASE4 ? 2 : 1;
        for (int log2m = LOG2BASE + 2; log2m <= LOG2N - 2; log2m += 2) {
            const int q = (1 << log2m) >> 2, hq = q >> 1;
            const int nblocks = QN >> log2m;
            const int tws = LOG2N - log2m;
            for (int u = t; u < nblocks * (hq + 1); u += T) {
                int blk, i;
                if (u < nblocks * hq) { blk = u / hq; i = u - blk * hq; } else { blk = u - nblocks * hq; i = hq; }
                const int o = blk << log2m;
                const double2 A = Q[o + i];
                const double2 Bv = cmul(Q[o + q + i], tw64[i << tws]);
                const double2 Cc = cmul(Q[o + 2 * q + i], tw64[2 * (i << tws)]);
                const double2 D = cmul(Q[o + 3 * q + i], tw64[3 * (i << tws)]);
                const double2 T0 = cadd(A, Cc), T1 = csub(A, Cc), T2 = cadd(Bv, D), T3 = csub(Bv, D);
                Q[o + i] = cadd(T0, T2);
                Q[o + q + i] = double2{T1.x + T3.y, T1.y - T3.x};
                if (i == 0) {
                    Q[o + 2 * q] = csub(T0, T2);
                } else if (i != hq) {
                    Q[o + q - i] = double2{T1.x - T3.y, -(T1.y + T3.x)};
                    Q[o + 2 * q - i] = double2{T0.x - T2.x, -(T0.y - T2.y)};
                }
            }
            __syncthreads();
        }
        if (dbg_X)
            for (int i = t; i < QN; i += T) if (base + i >= H) { dbg_X[2 * (base + i)] = Q[i].x; dbg_X[2 * (base + i) + 1] = Q[i].y; }
        unsigned rt[8];
        double2 ys[8];
        int id[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int b = base + t + T * j, tgt = b + up_delta;
            rt[j] = (b >= H && b < upper_end && tgt >= 0 && tgt < H) ? ((up_ridx << 16) | (unsigned)tgt) : NOROUTE;
            ys[j] = rotate_route_d<R_, LOG2N>(rt[j], Q[t + T * j], tw64);
            id[j] = b - N / 2;
        }
        claim_rounds_wg16<8, H>(rt, ys, id, Y, CLAIM);
        __syncthreads();
    }
}

// S_ROWS = hop / (N / 16) in {2, 4, 8, 16}: the frame advances by whole register rows (N / 16 samples), overlap-add accumulator and input window live in registers.
// AUX: test-tap instance.  RESIDENT: streaming instance that stays on the GPU (see pv_wg_kernel.hip).
template <int LOG2N, int S_ROWS, bool AUX, bool RESIDENT = false>
__global__ __launch_bounds__(QC<LOG2N>::T, (RESIDENT || FP64W) ? 1 : PV_WG16_WPS) PV_NO_DS_MERGE void pv_wg16_kernel(const PvKernelParams p)
{
    using C = QC<LOG2N>;
    constexpr int N = C::N, M = C::M, H = C::H, T = C::T, K2 = C::K2, NW = C::NW, PR = C::PR, PM = C::PM;
    constexpr int HOP = 2 * T * S_ROWS, R = N / HOP, LROWS = 16 - S_ROWS, L = N - HOP;
    constexpr int NEGPD = -(1 << 30), POSPD = 1 << 30;                    // packed (bin << 16 | shift) sentinels: no peak on this side
    constexpr int DROP = 0x4000;
    const int t = threadIdx.x;
    const int ch = blockIdx.y, chunk = blockIdx.x;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double2 *S64 = reinterpret_cast<double2 *>(smem);
    float2 *Y = reinterpret_cast<float2 *>(smem + C::OFF_Y);
    float *MAG = reinterpret_cast<float *>(smem + C::OFF_ROUTE);
    unsigned *ROUTE = reinterpret_cast<unsigned *>(smem + C::OFF_ROUTE);
    unsigned *CLAIM = reinterpret_cast<unsigned *>(smem + C::OFF_ROUTE);
    int *LASTIN = reinterpret_cast<int *>(smem + C::OFF_NEAR), *FIRSTIN = LASTIN + T;
    unsigned long long *OCC = reinterpret_cast<unsigned long long *>(smem + C::OFF_OCC);
    unsigned *BADW = reinterpret_cast<unsigned *>(smem + C::OFF_OCC + 32);
    double2 *TWB = reinterpret_cast<double2 *>(smem + C::OFF_TWB);
    pk::c32 *TWBF = reinterpret_cast<pk::c32 *>(smem + C::OFF_TWBF);

    // ---- table: W_256^{n0 k1} = tw[32 n0 k1] (tw[i] = exp(-2 pi j i / N)), and its conjugate in fp32 ----
    if (t < 15 * K2) {
        const int k1 = t / K2 + 1, n0 = t % K2;
        const double2 w = p.tw64[(32 * n0 * k1) & (N - 1)];
        TWB[t] = w;
        TWBF[t] = pk::c32{(float)w.x, -(float)w.y};
    }
    const int ts = (t >> 4) + K2 * (t & 15);                              // this thread's complex elements: ts + T r

    unsigned psh_key = 0u;
    bool psh_valid = false;
    v4u dq0{0u, 0u, 0u, 0u}, dq1{0u, 0u, 0u, 0u};                         // shifts of this thread's own 16 candidate bins (i16 each)
    const float *hist_in = p.hist_in, *acc_in = p.acc_in;
    float *hist_out = p.hist_out, *acc_out = p.acc_out;
    int t0_mod_n = p.t0_mod_n;
    unsigned done_seq = p.done_seq;
    unsigned last_seq = p.done_seq;
resident_top:
    if (RESIDENT) {
        unsigned *BC = reinterpret_cast<unsigned *>(smem + C::OFF_OCC + 48);
        if (t == 0) {
            unsigned word;
            const unsigned long long idle0 = wall_clock64();
            for (;;) {
                word = __hip_atomic_load(p.ctl + 16 + ch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if ((word & 0xFFFFu) != (last_seq & 0xFFFFu)) break;
                if (__hip_atomic_load(p.ctl + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u || wall_clock64() - idle0 > (unsigned long long)p.idle_ticks) { word = 0u; break; }
                __builtin_amdgcn_s_sleep(2);
            }
            BC[0] = word;
        }
        __syncthreads();
        const unsigned word = BC[0];
        __syncthreads();
        if (word == 0u) return;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
        const unsigned seq = word & 0xFFFFu, nch_now = (word >> 16) & 0x7Fu, cur = (word >> 23) & 1u;
        t0_mod_n = (int)(((word >> 24) & 0xFFu) * HOP) & (N - 1);
        hist_in = p.hist2[cur]; hist_out = p.hist2[cur ^ 1u];
        acc_in = p.acc2[cur]; acc_out = p.acc2[cur ^ 1u];
        done_seq = last_seq = seq;
        if (nch_now == 0u) {
            for (int j = t; j < L; j += T) { hist_out[(long)ch * L + j] = hist_in[(long)ch * L + j]; acc_out[(long)ch * L + j] = acc_in[(long)ch * L + j]; }
            goto resident_top;
        }
    }
    const int first_out = chunk * p.frames_per_chunk;
    int last_out = first_out + p.frames_per_chunk;
    if (last_out > p.nhops) last_out = p.nhops;
    int first_frame = first_out - (R - 1);
    const bool from_state = (first_frame <= 0);
    if (from_state) first_frame = 0;

    const long cbase = (long)ch * p.ch_stride;
    const WaveSrc src{p.in + cbase, hist_in + (long)ch * L, L, RESIDENT && p.in_cached != 0};
    float *outp = p.out + cbase;
    const bool vec_out = (reinterpret_cast<uintptr_t>(outp) & 7u) == 0;
    const bool vec_in = ((reinterpret_cast<uintptr_t>(src.in) | reinterpret_cast<uintptr_t>(src.hist)) & 7u) == 0;
    const float *pitch_row = p.pitch + (p.pitch_stride ? (long)(ch / p.ch_per_stream) * p.pitch_stride : 0);

    float2 hw[16];                                         // 0.5 * Hann (the second half of the table) at samples 2 (ts + 256 r), +1: the split pass's 1/2 folded into the window (exact)
#pragma unroll
    for (int r = 0; r < 16; r++) hw[r] = *reinterpret_cast<const float2 *>(p.hann + N + 2 * (ts + T * r));

    float2 acc[16];
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = float2{0.f, 0.f};
    if (from_state) {
#pragma unroll
        for (int r = 0; r < LROWS; r++) {
            const float *a = acc_in + (long)ch * L + 2 * ts + 2 * T * r;
            acc[r] = float2{a[0], a[1]};
        }
    }
    auto load_rows = [&](float2 *w, int nrows, int first_row, int frame) {
        const long s0 = (long)(frame + 1) * HOP - N + 2 * ts;
#pragma unroll
        for (int r = 0; r < nrows; r++) {
            const long sx = s0 + 2 * T * (first_row + r);
            if (RESIDENT && src.sys && vec_in && sx >= 0) {
                const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(src.in + sx), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                w[r] = float2{__uint_as_float((unsigned)v), __uint_as_float((unsigned)(v >> 32))};
            } else if (vec_in) w[r] = *reinterpret_cast<const float2 *>(sx < 0 ? src.hist + sx + src.hist_len : src.in + sx);
            else w[r] = float2{src.at(sx), src.at(sx + 1)};
        }
    };
    float2 raw[16];
    load_rows(raw, 16, 0, first_frame);
    float pf_next = (RESIDENT && src.sys) ? __hip_atomic_load(pitch_row + first_frame, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : pitch_row[first_frame];
    int emit_v = first_out;
    asm volatile("" : "+v"(emit_v));
    __syncthreads();
#ifdef PV_WG16_PH
    W16Clock w16clk;
    w16clk.start();
#endif
    TwA twa_next;
    { const int ts0 = (t >> 4) + K2 * (t & 15); twa_next = TwA{p.tw64[(2 * ts0) & (N - 1)], p.tw64[(4 * ts0) & (N - 1)], p.tw64[(8 * ts0) & (N - 1)], p.tw64[(16 * ts0) & (N - 1)]}; }

    for (int m = first_frame; m < last_out; ++m) {
        int tq = t;
        asm volatile("" : "+v"(tq));                                       // LDS addresses are recomputed per frame instead of hoisted (see pv_wg_kernel.hip)
        const int l = tq & 63, wv = __builtin_amdgcn_readfirstlane(tq >> 6);
        const float pfm = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(pf_next)));
        const double pf = (double)pfm;
        const int tmod = (int)(((long)t0_mod_n + (long)m * HOP) & (N - 1));
        const bool dbg = AUX && (p.dbg_mag != nullptr) && ch == p.dbg_ch && m == p.dbg_frame;
        const int pl = tq + 4 * (tq >> 4), ql = tq + 4 * ((tq + 15) >> 4);   // padded positions: P(tq + 256 r) = pl + 320 r, P(M - tq - 256 r) = 5120 - ql - 320 r

        // Loop-invariant per-thread values that are only needed in ONE phase of the frame are re-loaded per frame (L1 / L2 hits, issued ahead of their use) through
        // addresses derived from the opaque thread id, instead of occupying registers across the phases that need every one of them: the pass-A twiddles here,
        // the Hann values and the inverse's twiddles in front of the inverse FFT.
        const int tsq = (tq >> 4) + K2 * (tq & 15);
        // W_M^{ts k}, k = 1, 2, 4, 8: loaded behind the previous frame's inverse FFT (round 5: at the top of the frame the first use, ~150 instructions on, came before the
        // loads did) -- where the registers allow it: at hop = N / 2 sixteen more live values across the loop edge spill (22-28 VGPRs), there the loads stay here
        constexpr bool TWA_AHEAD = (S_ROWS != 8);
        if constexpr (!TWA_AHEAD) twa_next = TwA{p.tw64[(2 * tsq) & (N - 1)], p.tw64[(4 * tsq) & (N - 1)], p.tw64[(8 * tsq) & (N - 1)], p.tw64[(16 * tsq) & (N - 1)]};
        const TwA twa = twa_next;
        const double2 wl = p.tw64[tq];                                      // split pass: W_N^{tq + 256 r} = wl * W_32^r
        // ---- Hann (pv:55), pack, forward FFT in fp64 (the split pass's 1/2 is folded into the window, exact) ----
        wg16_prio<0, T>();
        double2 z[16];
#pragma unroll
        for (int r = 0; r < 16; r++) z[r] = double2{(double)(raw[r].x * hw[r].x), (double)(raw[r].y * hw[r].y)};
        if (!FP64W && pf < 0.75 && ((tq >> 4) & 1)) {
            // the general residue (f < 0.75 only) rebuilds quarter 2 of fft.js's buffer from the windowed samples xw[4n + 2] = sample 2 (ts + 256 r) of the threads
            // with an odd ts: stashed in natural order while they are in registers (pv_wg_kernel.hip)
            float *XQ = reinterpret_cast<float *>(smem + C::OFF_XQ);
#pragma unroll
            for (int r = 0; r < 16; r++) XQ[(tsq + T * r - 1) >> 1] = raw[r].x * (2.0f * hw[r].x);
        }
        fft_wg16<T>(z, S64, twa, TWB, tq W16_ST);

        // ---- split pass in conjugate pairs: thread tq owns the pairs k = tq + 256 r, r < 8: XA[r] = X[k], XB[r] = X[M - k]; thread 0 also the self-paired bin M/2.
        //      The partner values Z[M - k] are rows 8..15 of other threads -> LDS ----
        float2 XA[8], XB[8], xHf{0.f, 0.f};
        [[maybe_unused]] double2 XAd[8], XBd[8], xHd{0.0, 0.0};                // fp64 flavour: the source spectrum stays in doubles
        wg16_prio<2, T>();
        {
#pragma unroll
            for (int r = 8; r < 16; r++) S64[tq + T * (r - 8)] = z[r];
            __syncthreads();
            double2 xH{0.0, 0.0};
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const int k = tq + T * r;
                const double2 zm = (k == 0) ? z[0] : S64[8 * T - k];          // element M - k sits at (M - k) - 8T of the stored rows
                const double2 E{z[r].x + zm.x, z[r].y - zm.y};
                const double2 O{z[r].x - zm.x, z[r].y + zm.y};
                const double2 WO = cmul(wl, mul_w32_16(O, r));
                double2 xa{E.x + WO.y, E.y - WO.x};
                double2 xb{E.x - WO.y, -(E.y + WO.x)};
                if (r == 0 && tq == 0) {
                    xa = double2{2.0 * (z[0].x + z[0].y), 0.0};               // X[0], X[M]: both real
                    xb = double2{2.0 * (z[0].x - z[0].y), 0.0};
                }
                // (the magnitudes sit behind the partner rows: they may be written while other threads still read theirs; fp32 values at once: 16 doubles less to hold)
                MAG[C::MAG0 + pl + PR * r] = (float)(xa.x * xa.x + xa.y * xa.y);
                MAG[C::MAG0 + PM - ql - PR * r] = (float)(xb.x * xb.x + xb.y * xb.y);
                XA[r] = float2{(float)xa.x, (float)xa.y};
                XB[r] = float2{(float)xb.x, (float)xb.y};
                if constexpr (FP64W) { XAd[r] = xa; XBd[r] = xb; }
                if (dbg) {
                    const int ka = tq + T * r, kb = M - ka;
                    p.dbg_X[2 * ka] = xa.x; p.dbg_X[2 * ka + 1] = xa.y;
                    p.dbg_X[2 * kb] = xb.x; p.dbg_X[2 * kb + 1] = xb.y;
                }
            }
            if (tq == 0) {
                xH = double2{2.0 * z[8].x, -2.0 * z[8].y};                    // k = M/2 pairs with itself: X = 2 conj(Z)
                MAG[C::MAG0 + PM / 2] = (float)(xH.x * xH.x + xH.y * xH.y);
                if (dbg) { p.dbg_X[M] = xH.x; p.dbg_X[M + 1] = xH.y; }
            }
            xHf = float2{(float)xH.x, (float)xH.y};
            if constexpr (FP64W) xHd = xH;
            if (!FP64W && pf < 1.0) {                                                // fp32 spectrum stash for the fast residue (Y is not live yet; it aliases the partner rows)
                __syncthreads();
#pragma unroll
                for (int r = 0; r < 8; r++) { Y[tq + T * r] = XA[r]; Y[M - tq - T * r] = XB[r]; }
                if (tq == 0) Y[M / 2] = xHf;
            }
        }
        // conj(W^{2k}) of this thread's first bin k = 1 + tq of the fast residue (f < 1; k = 1 + tq + 256 j: times conj(W_16^j)): a load from the global table, issued HERE,
        // two barriers ahead of its use -- behind the "magnitudes complete" barrier it was an exposed round trip in every f < 1 frame (profiles/r05_wg16_phase_clock.md)
        // (the RAW table entry: a conjugation inside the branch would make the branch wait for its own load)
        float2 s2raw{1.f, 0.f};
        if (!FP64W && pf < 1.0) s2raw = p.tw32[2 * (1 + tq)];
        // slide the raw window; the rows the next frame adds are issued here
        {
            const int mn = (m + 1 < last_out) ? m + 1 : m;
#pragma unroll
            for (int r = 0; r < 16 - S_ROWS; r++) raw[r] = raw[r + S_ROWS];
            if (!RESIDENT) {
                load_rows(&raw[16 - S_ROWS], S_ROWS, 16 - S_ROWS, mn);
                pf_next = pitch_row[mn];
            }
        }
        // ---- shifts Math.round(peak * f) - peak (pv:125,147) of this thread's own 16 candidate bins, rebuilt only when f changes ----
        {
            const unsigned pfb = __float_as_uint(pfm);
            if (!psh_valid || pfb != psh_key) {
                psh_key = pfb;
                psh_valid = true;
                unsigned sh[16];
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    const int pk = 16 * tq + i;
                    const double ps = floor((double)pk * pf + 0.5);
                    const bool ok = (ps <= (double)H) && (ps >= -(double)(2 * N));
                    sh[i] = (unsigned)(ok ? ((int)ps - pk) : DROP) & 0xFFFFu;       // DROP pushes every target of the region out of range
                }
                dq0 = v4u{sh[0] | (sh[1] << 16), sh[2] | (sh[3] << 16), sh[4] | (sh[5] << 16), sh[6] | (sh[7] << 16)};
                dq1 = v4u{sh[8] | (sh[9] << 16), sh[10] | (sh[11] << 16), sh[12] | (sh[13] << 16), sh[14] | (sh[15] << 16)};
            }
        }
        W16_MARK(5);
        __syncthreads();                                                   // magnitudes (and the stash) complete
        W16_MARK(6);
        // ---- above-Nyquist residue, fast form: W^{2k} S2[k] = (X[k] - X[k+N/4] + X[k+N/2] - X[k+3N/4]) / 4, k = 1 + tq + 256 j (see pv_wg_kernel.hip) ----
        float2 s2v[4] = {float2{0.f, 0.f}, float2{0.f, 0.f}, float2{0.f, 0.f}, float2{0.f, 0.f}};
        if (!FP64W && pf < 1.0) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int k = 1 + tq + T * j;                                // k in [1, N/8]
                const float2 x0 = Y[k], x1 = Y[k + M / 2], x2 = Y[M - k], x3 = Y[M / 2 - k];
                const float2 tsum{0.25f * ((x0.x - x1.x) + (x2.x - x3.x)), 0.25f * ((x0.y - x1.y) - (x2.y - x3.y))};
                const float c8 = 0.92387953251128675613f, s8 = 0.38268343236508977173f, h8 = 0.70710678118654752440f;
                const float2 s2w0 = cconj(s2raw);
                const float2 wj = (j == 0) ? s2w0 : (j == 1) ? cmul(s2w0, float2{c8, s8}) : (j == 2) ? cmul(s2w0, float2{h8, h8}) : cmul(s2w0, float2{s8, c8});
                s2v[j] = cmul(tsum, wj);
            }
        }
        bool nonfinite = false;
        wg16_prio<6, T>();
        // ---- peak flags on bins 16 tq .. 16 tq + 15 (pv:95-116) ----
        int lastown[16], firstown[16];
        int last_in, first_in;
        {
            unsigned mg[20];
            typedef const volatile __attribute__((address_space(3))) v2u *lds_v2u;
            typedef const volatile __attribute__((address_space(3))) v4u *lds_v4u;
            const v2u q0 = *(lds_v2u)(&MAG[C::MAG0 + 20 * tq - 6]);        // bins 16 tq - 2, 16 tq - 1
            const v4u q1 = *(lds_v4u)(&MAG[C::MAG0 + 20 * tq]);
            const v4u q2 = *(lds_v4u)(&MAG[C::MAG0 + 20 * tq + 4]);
            const v4u q3 = *(lds_v4u)(&MAG[C::MAG0 + 20 * tq + 8]);
            const v4u q4 = *(lds_v4u)(&MAG[C::MAG0 + 20 * tq + 12]);
            const v2u q5 = *(lds_v2u)(&MAG[C::MAG0 + 20 * tq + 20]);       // bins 16 tq + 16, 16 tq + 17
            mg[0] = q0.x; mg[1] = q0.y;
            mg[2] = q1.x; mg[3] = q1.y; mg[4] = q1.z; mg[5] = q1.w; mg[6] = q2.x; mg[7] = q2.y; mg[8] = q2.z; mg[9] = q2.w;
            mg[10] = q3.x; mg[11] = q3.y; mg[12] = q3.z; mg[13] = q3.w; mg[14] = q4.x; mg[15] = q4.y; mg[16] = q4.z; mg[17] = q4.w;
            mg[18] = q5.x; mg[19] = q5.y;
            unsigned pm[19];
#pragma unroll
            for (int j = 3; j < 19; j++) pm[j] = max(mg[j], mg[j + 1]);
            bool fl[16];
#pragma unroll
            for (int i = 0; i < 16; i++) {
                // candidates are 2 <= k < H - 2 (pv:97-100): thread 0 drops i < 2, the last thread drops i = 15
                const bool in_range = (i < 2) ? (tq != 0) : (i == 15) ? (tq != T - 1) : true;
                fl[i] = in_range & (max(max(mg[i], mg[i + 1]), pm[i + 3]) < mg[i + 2]);
            }
            {
                unsigned mx = mg[2];
#pragma unroll
                for (int j = 3; j < 19; j += 2) mx = max(mx, pm[j]);
                nonfinite = __any(mx >= 0x7F800000u);                      // Inf / NaN magnitude in this wave's bins (see pv_wave_kernel.hip)
            }
            if (dbg) {
#pragma unroll
                for (int i = 0; i < 16; i++) { p.dbg_flags[16 * tq + i] = fl[i] ? 1 : 0; p.dbg_mag[16 * tq + i] = __uint_as_float(mg[i + 2]); }
                if (tq == T - 1) { p.dbg_flags[M] = 0; p.dbg_mag[M] = __uint_as_float(mg[18]); }
            }
            int pd[16];
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const unsigned w = (i < 8) ? dq0[i >> 1] : dq1[(i - 8) >> 1];
                pd[i] = (int)__builtin_amdgcn_perm((unsigned)(16 * tq + i), w, (i & 1) ? 0x05040302u : 0x05040100u);
            }
            int cur = NEGPD;
#pragma unroll
            for (int i = 0; i < 16; i++) { cur = fl[i] ? pd[i] : cur; lastown[i] = cur; }
            int nx = POSPD;
#pragma unroll
            for (int i = 15; i >= 0; i--) { firstown[i] = nx; nx = fl[i] ? pd[i] : nx; }
            last_in = cur; first_in = nx;
        }
        // ---- nearest peaks outside this thread's 16 bins: per-wave occupancy ballots + per-thread last / first peak, through LDS ----
        {
            const unsigned long long occ = __ballot(last_in >= 0);
            LASTIN[tq] = last_in;
            FIRSTIN[tq] = first_in;
            if (l == 0) OCC[wv] = occ;
        }
        __syncthreads();                                                   // also: every MAG read is done -> ROUTE may overwrite MAG
        W16_MARK(7);
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
                    for (int w = wv + 1; w < NW; ++w) { const unsigned long long o = OCC[w]; if (o) { srcT = w * 64 + __ffsll((long long)o) - 1; break; } }
                if (srcT >= 0) cnext = FIRSTIN[srcT];
            }
            for (int w = NW - 1; w >= 0; --w) {
                const unsigned long long o = OCC[w];
                if (o) { const int lp = LASTIN[w * 64 + 63 - __clzll((long long)o)]; last_peak = lp >> 16; last_shift = (int)(short)(lp & 0xFFFF); break; }
            }
        }
        {
            unsigned rt[16];
            unsigned rtM = NOROUTE;
            bool bad = false;
            if (last_peak < 0) {
#pragma unroll
                for (int i = 0; i < 16; i++) rt[i] = NOROUTE;
            } else {
                auto route_of = [&](int b, int pp, int pn) -> unsigned {
                    const int own = (b - (pp >> 16) < (pn >> 16) - b) ? pp : pn;     // owner rule (pv:132-141)
                    const int delta = __builtin_amdgcn_sbfe(own, 0, 16);
                    return __builtin_amdgcn_perm((unsigned)__mul24(delta, tmod), (unsigned)(b + delta), 0x05040100u);
                };
#pragma unroll
                for (int i = 0; i < 16; i++) rt[i] = route_of(16 * tq + i, max(lastown[i], cprev), min(firstown[i], cnext));
                if (tq == T - 1) rtM = route_of(M, max(last_in, cprev), POSPD);
                if (!(pf >= 1.0)) {
                    rtM &= 0x7FFFFFFFu;
#pragma unroll
                    for (int i = 0; i < 16; i++) {
                        const int pp = max(lastown[i], cprev), pn = min(firstown[i], cnext), b = 16 * tq + i;
                        const bool rising = !(b - (pp >> 16) < (pn >> 16) - b);
                        rt[i] = (rt[i] & 0x7FFFFFFFu) | (rising ? 0x80000000u : 0u);
                    }
                    if (!(pfm >= PV_PAIRWISE_SURE)) {                       // (f >= 2/3: the test cannot fail, see pv_device_common.h)
#pragma unroll
                        for (int i = 0; i < 16; i++) {
                            const int pp = max(lastown[i], cprev), pn = min(firstown[i], cnext);
                            const int gap = (pn >> 16) - (pp >> 16), ov = __builtin_amdgcn_sbfe(pp, 0, 16) - __builtin_amdgcn_sbfe(pn, 0, 16);
                            bad |= ov > (gap >> 1);
                        }
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < 4; j++) *reinterpret_cast<uint4 *>(&ROUTE[C::MAG0 + 20 * tq + 4 * j]) = uint4{rt[4 * j], rt[4 * j + 1], rt[4 * j + 2], rt[4 * j + 3]};
            if (tq == T - 1) ROUTE[C::MAG0 + PM] = rtM;
            if (!(pf >= 1.0)) { const bool wbad = __any(bad); if (l == 0) BADW[wv] = wbad ? 1u : 0u; }
        }
        int upper_end = H;
        if (last_peak >= 0 && last_shift < 0) { upper_end = H - last_shift; if (upper_end > N) upper_end = N; }      // DROP is positive
        const float2 wlf = cconj(p.tw32[tq]);                               // c2r twiddle e^{+2 pi j tq / N}: loaded a phase ahead of its use
        // ---- zero Y (pv:121) ----
        [[maybe_unused]] double2 *Yd = reinterpret_cast<double2 *>(smem + C::OFF_Y);
        if constexpr (FP64W) {
#pragma unroll
            for (int r = 0; r < 16; r++) *reinterpret_cast<v4f *>(&Yd[tq + T * r]) = v4f{0.f, 0.f, 0.f, 0.f};
            if (tq == 0) Yd[M] = double2{0.0, 0.0};
        } else {
#pragma unroll
            for (int r = 0; r < 8; r++) *reinterpret_cast<v4f *>(&Y[2 * tq + 2 * T * r]) = v4f{0.f, 0.f, 0.f, 0.f};
            if (tq == 0) Y[M] = float2{0.f, 0.f};
        }
        const bool need_res = upper_end > H;
        __syncthreads();
        W16_MARK(8);
        wg16_prio<7, T>();
        // ---- shiftPeaks (pv:119-173) ----
        if constexpr (FP64W) {
            // fp64 flavour: plain stores for f >= 1 (disjoint regions); f < 1 (and NaN): `+=` collisions (pv:169-170) by claim rounds in ascending batches, then the
            // sources above Nyquist (all owned by the last peak, pv:133) from the re-run stage structure -- neither the pairwise scatter nor the closed form of the residue
            if (pf >= 1.0) {
#pragma unroll
                for (int r = 0; r < 8; r++) {
                    const unsigned ra = ROUTE[C::MAG0 + pl + PR * r], ta = ra & 0xFFFFu;
                    const unsigned rb = ROUTE[C::MAG0 + PM - ql - PR * r], tb = rb & 0xFFFFu;
                    if (ta < (unsigned)H) Yd[ta] = rotate_route_d<R, LOG2N>(ra, XAd[r], p.tw64);
                    if (tb < (unsigned)H) Yd[tb] = rotate_route_d<R, LOG2N>(rb, XBd[r], p.tw64);
                }
                if (tq == 0) { const unsigned rt = ROUTE[C::MAG0 + PM / 2], tg = rt & 0xFFFFu; if (tg < (unsigned)H) Yd[tg] = rotate_route_d<R, LOG2N>(rt, xHd, p.tw64); }
            } else {
                unsigned rt0[9], rt1[9]; double2 ys0[9], ys1[9]; int id0[9], id1[9];
#pragma unroll
                for (int r = 0; r < 8; r++) {
                    id0[r] = tq + T * r; rt0[r] = ROUTE[C::MAG0 + pl + PR * r]; ys0[r] = rotate_route_d<R, LOG2N>(rt0[r], XAd[r], p.tw64);
                    id1[r] = M - tq - T * r; rt1[r] = ROUTE[C::MAG0 + PM - ql - PR * r]; ys1[r] = rotate_route_d<R, LOG2N>(rt1[r], XBd[r], p.tw64);
                }
                id0[8] = 0; rt0[8] = NOROUTE; ys0[8] = double2{0.0, 0.0};
                id1[8] = M / 2; rt1[8] = (tq == 0) ? ROUTE[C::MAG0 + PM / 2] : NOROUTE; ys1[8] = rotate_route_d<R, LOG2N>(rt1[8], xHd, p.tw64);
                __syncthreads();                                            // every ROUTE read is done: the region becomes the claim words
#pragma unroll
                for (int r = 0; r < 16; r++) CLAIM[tq + T * r] = 0xFFFFFFFFu;
                if (tq == 0) CLAIM[M] = 0xFFFFFFFFu;
                claim_rounds_wg16<9, H>(rt0, ys0, id0, Yd, CLAIM);
                claim_rounds_wg16<9, H>(rt1, ys1, id1, Yd, CLAIM);
                if (need_res) {
                    __syncthreads();
                    const int up_delta = last_shift;
                    residue_scatter_wg16_d<LOG2N, R>(src.in, src.hist, src.hist_len, src.sys, (long)(m + 1) * HOP - N, p.hann, p.tw64, tq, upper_end, up_delta,
                                                     (unsigned)((up_delta & (N - 1)) * tmod) & (N - 1), dbg ? p.dbg_X : nullptr);
                }
            }
        } else {
            if (pf >= 1.0) {
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
                    for (int r = 0; r < 8; r++) {
                        const unsigned ra = ROUTE[C::MAG0 + pl + PR * r], ta = ra & 0xFFFFu;
                        const unsigned rb = ROUTE[C::MAG0 + PM - ql - PR * r], tb = rb & 0xFFFFu;
                        if (ta < (unsigned)H) Y[ta] = rot(ra, XA[r]);
                        if (tb < (unsigned)H) Y[tb] = rot(rb, XB[r]);
                    }
                    if (tq == 0) { const unsigned rt = ROUTE[C::MAG0 + PM / 2], tg = rt & 0xFFFFu; if (tg < (unsigned)H) Y[tg] = rot(rt, xHf); }
                };
                if (tmod == 0) scatter(std::integral_constant<int, 0>{});
                else if (tmod == N / 2) scatter(std::integral_constant<int, 2>{});
                else scatter(std::integral_constant<int, 1>{});
            } else {
                // f < 1 (and NaN): regions compress, `+=` collisions (pv:169-170).  The sources go in TWO batches -- bins below M/2 (XA), then M/2 and the bins above
                // (XB) -- so that the route / value / key arrays are half as long (the 17-wide form spills the f >= 1 path of the loop); ascending source order, the
                // reference's order of accumulation (pv:122,146), is kept by finishing batch 0 before batch 1 starts.
                bool pairwise = PV_PAIRWISE != 0;
#pragma unroll
                for (int w = 0; w < NW; w++) pairwise = pairwise && (BADW[w] == 0u);       // uniform in the workgroup
                auto gather = [&](auto half_tag, unsigned (&rt)[9], float2 (&ys)[9], int (&id)[9]) {
                    constexpr int HALF = decltype(half_tag)::value;
#pragma unroll
                    for (int r = 0; r < 8; r++) {
                        if (HALF == 0) { id[r] = tq + T * r; rt[r] = ROUTE[C::MAG0 + pl + PR * r]; ys[r] = rotate_route<R, LOG2N>(rt[r], XA[r], p.tw32); }
                        else { id[r] = M - tq - T * r; rt[r] = ROUTE[C::MAG0 + PM - ql - PR * r]; ys[r] = rotate_route<R, LOG2N>(rt[r], XB[r], p.tw32); }
                    }
                    if (HALF == 0) { id[8] = 0; rt[8] = NOROUTE; ys[8] = float2{0.f, 0.f}; }
                    else { id[8] = M / 2; rt[8] = (tq == 0) ? ROUTE[C::MAG0 + PM / 2] : NOROUTE; ys[8] = rotate_route<R, LOG2N>(rt[8], xHf, p.tw32); }
                };
                if (pairwise) {
                    // every collision is one falling-side source against one rising-side source: the falling side (and the residue, which continues the falling side of
                    // the last peak) stores into the zeroed Y, one barrier, the rising side adds
                    unsigned rt0[9], rt1[9]; float2 ys0[9], ys1[9];
                    { int id[9]; gather(std::integral_constant<int, 0>{}, rt0, ys0, id); gather(std::integral_constant<int, 1>{}, rt1, ys1, id); }
#pragma unroll
                    for (int r = 0; r < 8; r++) { const unsigned key = rt0[r] & 0x8000FFFFu; if (key < (unsigned)H) Y[key] = ys0[r]; }
#pragma unroll
                    for (int r = 0; r < 9; r++) { const unsigned key = rt1[r] & 0x8000FFFFu; if (key < (unsigned)H) Y[key] = ys1[r]; }
                    const int up_delta = need_res ? last_shift : 0;
                    const unsigned up_ridx = (unsigned)((up_delta & (N - 1)) * tmod) & (N - 1);
                    if (need_res && upper_end <= H + N / 8) {
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            const int b = H + tq + T * j, tgt = b + up_delta;
                            const unsigned rtj = (b < upper_end && tgt >= 0 && tgt < H) ? ((up_ridx << 16) | (unsigned)tgt) : NOROUTE;
                            if (rtj != NOROUTE) Y[tgt] = rotate_route<R, LOG2N>(rtj, s2v[j], p.tw32);
                            if (dbg && b < upper_end) { p.dbg_X[2 * b] = s2v[j].x; p.dbg_X[2 * b + 1] = s2v[j].y; }
                        }
                    }
                    __syncthreads();
                    auto add_half = [&](const unsigned (&rt)[9], const float2 (&ys)[9]) {     // (two batches of reads: the 17-wide form spills the f >= 1 path of the loop)
                        float2 o[9];
#pragma unroll
                        for (int r = 0; r < 9; r++) o[r] = Y[min(rt[r] & 0xFFFFu, (unsigned)M)];
#pragma unroll
                        for (int r = 0; r < 9; r++) { const unsigned key = rt[r] & 0x8000FFFFu; if (key - 0x80000000u < (unsigned)H) Y[key - 0x80000000u] = float2{o[r].x + ys[r].x, o[r].y + ys[r].y}; }
                    };
                    add_half(rt0, ys0);                                  // (a target takes at most one rising-side source: the two batches do not meet)
                    add_half(rt1, ys1);
                } else {
                    unsigned rt0[9], rt1[9]; float2 ys0[9], ys1[9]; int id0[9], id1[9];
                    gather(std::integral_constant<int, 0>{}, rt0, ys0, id0);
                    gather(std::integral_constant<int, 1>{}, rt1, ys1, id1);
                    __syncthreads();                                        // every ROUTE read is done: the region becomes the claim words
#pragma unroll
                    for (int r = 0; r < 16; r++) CLAIM[tq + T * r] = 0xFFFFFFFFu;
                    if (tq == 0) CLAIM[M] = 0xFFFFFFFFu;
                    claim_rounds_wg16<9, H>(rt0, ys0, id0, Y, CLAIM);            // (its first barrier orders the fill before the first claims)
                    claim_rounds_wg16<9, H>(rt1, ys1, id1, Y, CLAIM);
                    if (need_res) {
                        __syncthreads();
                        const int up_delta = last_shift;
                        const unsigned up_ridx = (unsigned)((up_delta & (N - 1)) * tmod) & (N - 1);
                        if (upper_end <= H + N / 8) {
                            unsigned rt2[4];
                            float2 ys2[4];
                            int id2[4];
#pragma unroll
                            for (int j = 0; j < 4; j++) {
                                const int b = H + tq + T * j, tgt = b + up_delta;
                                rt2[j] = (b < upper_end && tgt >= 0 && tgt < H) ? ((up_ridx << 16) | (unsigned)tgt) : NOROUTE;
                                ys2[j] = rotate_route<R, LOG2N>(rt2[j], s2v[j], p.tw32);
                                id2[j] = b - N / 2;
                                if (dbg && b < upper_end) { p.dbg_X[2 * b] = s2v[j].x; p.dbg_X[2 * b + 1] = s2v[j].y; }
                            }
                            claim_rounds_wg16<4, H>(rt2, ys2, id2, Y, CLAIM);
                        }
                    }
                }
                if (need_res && upper_end > H + N / 8) {
                    __syncthreads();
                    const int up_delta = last_shift;
                    residue_scatter_wg16<LOG2N, R>(src.in, src.hist, src.hist_len, src.sys, (long)(m + 1) * HOP - N, p.hann, p.tw32, tq, upper_end, up_delta,
                                            (unsigned)((up_delta & (N - 1)) * tmod) & (N - 1), dbg ? p.dbg_X : nullptr, pairwise
#ifdef PV_WG16_PH
                                            , p.stamps ? p.stamps + 32 * ((long)ch * gridDim.x + chunk) : nullptr
#endif
                                            );
                }
            }
        }
        if (nonfinite && l == 0) {                                         // the reference's frame is NaN: so is this one
            if constexpr (FP64W) Yd[1 + wv] = double2{__longlong_as_double(0x7FF8000000000000ll), __longlong_as_double(0x7FF8000000000000ll)};
            else Y[1 + wv] = float2{__uint_as_float(0x7FC00000u), __uint_as_float(0x7FC00000u)};
        }
        __syncthreads();
        W16_MARK(9);
        if (dbg) {
            auto yat = [&](int k) -> float2 { if constexpr (FP64W) return float2{(float)Yd[k].x, (float)Yd[k].y}; else return Y[k]; };
#pragma unroll
            for (int r = 0; r < 16; r++) { const int k = tq + T * r; p.dbg_Y[2 * k] = yat(k).x; p.dbg_Y[2 * k + 1] = yat(k).y; }
            if (tq == 0) { p.dbg_Y[2 * M] = yat(M).x; p.dbg_Y[2 * M + 1] = yat(M).y; }
        }
        // the inverse's pass-A twiddles conj(W_M^{ts k}), k = 1, 2, 4, 8: issued HERE, in front of the c2r pass and its barrier.  Behind it, next to the sixteen window
        // loads, their place in the memory queue is the compiler's choice -- and vmcnt counts in order: when they come out BEHIND the window loads, the inverse FFT
        // waits for all twenty (round 5: 3.05 -> 3.53 ms on C5 when an unrelated change flipped that order)
        [[maybe_unused]] float2 f1{1.f, 0.f}, f2{1.f, 0.f}, f4{1.f, 0.f}, f8{1.f, 0.f};
        if constexpr (!FP64W) { f1 = cconj(p.tw32[(2 * tsq) & (N - 1)]); f2 = cconj(p.tw32[(4 * tsq) & (N - 1)]); f4 = cconj(p.tw32[(8 * tsq) & (N - 1)]); f8 = cconj(p.tw32[(16 * tsq) & (N - 1)]); }
        wg16_prio<8, T>();
        // ---- c2r pre-pass in conjugate pairs, packed fp32 (see pv_wg_kernel.hip): thread tq computes k = tq + 256 r, r < 8, and hands Z[M - k] over through LDS ----
        pk::c32 zi[16];
        [[maybe_unused]] double2 zd[16];
        if constexpr (FP64W) {
            // c2r pre-pass in fp64: Z[k] = SC ((Yk + Ym*) + j e^{+2 pi j k / N} (Yk - Ym*)), conjugate pairs as in the product
            constexpr double sc = 2.0 / ((double)N * (double)R);
            const double2 wlc{wl.x, -wl.y};
            double2 zb[8];
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const int k = tq + T * r;
                double2 yk = Yd[k], ym = Yd[M - k];
                if (k == 0) { yk.y = 0.0; ym.y = 0.0; }
                const double2 E{yk.x + ym.x, yk.y - ym.y}, O{yk.x - ym.x, yk.y + ym.y};
                const double2 c = cmul(cconj(mul_w32_16(cconj(O), r)), wlc);   // O * exp(+2 pi j r / 32) * exp(+2 pi j tq / N)
                zd[r] = double2{(E.x - c.y) * sc, (E.y + c.x) * sc};        // E + j c
                zb[r] = double2{(E.x + c.y) * sc, -(E.y - c.x) * sc};       // conj(E - j c)
            }
            const double2 yH = Yd[M / 2];
            double2 *XCHd = reinterpret_cast<double2 *>(smem + C::OFF_RESQ);
#pragma unroll
            for (int r = 0; r < 8; r++) if (r > 0 || tq > 0) XCHd[8 * T - tq - T * r] = zb[r];
            __syncthreads();
#pragma unroll
            for (int r = 8; r < 16; r++) zd[r] = XCHd[tq + T * (r - 8)];
            if (tq == 0) zd[8] = double2{2.0 * yH.x * sc, -2.0 * yH.y * sc};
        } else {
            constexpr float sc = 2.0f / ((float)N * (float)R);                // 1/N of the inverse, 1/R of the overlap-add, 2 for the 0.5 * Hann table (exact)
            const pk::c32 wlfs{wlf.x * sc, wlf.y * sc};                       // c2r twiddle with the scale folded in
            const pk::c32 scsc{sc, sc};
            const pk::c32 *Yc = reinterpret_cast<const pk::c32 *>(Y);
            pk::c32 zb[8];
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const int k = tq + T * r;
                pk::c32 yk = Yc[k], ym = Yc[M - k];
                if (k == 0) { yk.y = 0.f; ym.y = 0.f; }
                const pk::c32 E = pk::add_conj(yk, ym), O = pk::sub_conj(yk, ym);
                const pk::c32 c = pk::cmul(mul_w32_inv_pk16(O, r), wlfs);
                zi[r] = pk::fma_addj(E, scsc, c);
                zb[r] = pk::fma_conj_subj(E, scsc, c);
            }
            const pk::c32 yH = Yc[M / 2];
            pk::c32 *XCH = reinterpret_cast<pk::c32 *>(smem + C::OFF_RESQ);   // hand-over buffer = the residue quarter buffer: element m = M - k, m in (8T, 16T), at m - 8T
#pragma unroll
            for (int r = 0; r < 8; r++) if (r > 0 || tq > 0) XCH[8 * T - tq - T * r] = zb[r];   // (tq = 0, r = 0) would be Z[M]: does not exist
            __syncthreads();
#pragma unroll
            for (int r = 8; r < 16; r++) zi[r] = XCH[tq + T * (r - 8)];
            if (tq == 0) zi[8] = pk::c32{2.0f * yH.x * sc, -2.0f * yH.y * sc};   // the self-paired bin M/2
        }
        // the inverse's cross-wave exchange writes [0, 32 KB) -- Y, whose reads all sit in front of the hand-over's barrier --, its in-wave exchange [32 KB, 64 KB) -- dead
        // routes / claim words and the hand-over buffer, whose reads sit in front of the cross-wave exchange's barrier: no barrier here
        int tqi = tq;
        asm volatile("" : "+v"(tqi));                                      // a fresh copy: the addresses of this phase are not kept alive from the top of the frame
        const int tsi = (tqi >> 4) + K2 * (tqi & 15);
        {
            // the window of the overlap-add below AND of the next frame's analysis (see the top of the loop): live from here to the next frame's first lines only
#pragma unroll
            for (int r = 0; r < 16; r++) hw[r] = *reinterpret_cast<const float2 *>(p.hann + N + 2 * (tsi + T * r));
        }
        if constexpr (FP64W) {
            const TwA twi{p.tw64[(2 * tsi) & (N - 1)], p.tw64[(4 * tsi) & (N - 1)], p.tw64[(8 * tsi) & (N - 1)], p.tw64[(16 * tsi) & (N - 1)]};
            fft_wg16_inv_d<T>(zd, S64, twi, TWB, tqi);
#pragma unroll
            for (int r = 0; r < 16; r++) zi[r] = pk::c32{(float)zd[r].x, (float)zd[r].y};     // fromComplexArray -> Float32Array (bundle:46-51)
        } else {
        const TwAf twaf{pk::c32{f1.x, f1.y}, pk::c32{f2.x, f2.y}, pk::c32{f4.x, f4.y}, pk::c32{f8.x, f8.y}};
        W16_MARK(10);
        fft_wg16_inv_pk<T>(zi, reinterpret_cast<pk::c32 *>(smem), twaf, TWBF, tqi W16_STI);
        }
        if constexpr (S_ROWS != 8) twa_next = TwA{p.tw64[(2 * tsi) & (N - 1)], p.tw64[(4 * tsi) & (N - 1)], p.tw64[(8 * tsi) & (N - 1)], p.tw64[(16 * tsi) & (N - 1)]};   // the next frame's pass-A twiddles
        // ---- Hann (pv:67), overlap-add in reference order, emit, shift ----
        wg16_prio<5, T>();
        {
            const bool emit_out = (m >= emit_v);
            float2 fr[16];
#pragma unroll
            for (int r = 0; r < 16; r++)                                   // rounded to fp32 BEFORE the accumulation like the reference's Float32Array (pv:67)
                { const pk::c32 f = pk::mul(zi[r], pk::c32{hw[r].x, hw[r].y}); fr[r] = float2{f.x, f.y}; }   // an asm multiply: see mul_rounded (pv_device_common.h)
#pragma unroll
            for (int r = 0; r < S_ROWS; r++) {
                const float2 o{acc[r].x + fr[r].x, acc[r].y + fr[r].y};
                if (emit_out) {
                    float *dst = outp + (long)m * HOP + 2 * tsi + 2 * T * r;
                    if (vec_out) *reinterpret_cast<v2f *>(dst) = v2f{o.x, o.y};                // plain, not non-temporal: the waves' 32-byte pieces of a line merge in L2
                    else { dst[0] = o.x; dst[1] = o.y; }
                }
            }
#pragma unroll
            for (int r = 0; r < LROWS; r++) {
                const int s = r + S_ROWS;
                acc[r] = (s < LROWS) ? float2{acc[s].x + fr[s].x, acc[s].y + fr[s].y} : fr[s];
            }
        }
        W16_MARK(16);
        __syncthreads();
        W16_MARK(17);
    }
#ifdef PV_WG16_PH
    if (p.stamps && t == 0) {
        unsigned *o = p.stamps + 32 * ((long)ch * gridDim.x + chunk);
        for (int i = 0; i < 20; i++) o[i] = w16clk.acc[i];
        o[20] = (unsigned)(last_out - first_frame);
    }
#endif

    if (chunk == (int)gridDim.x - 1) {
#pragma unroll
        for (int r = 0; r < LROWS; r++) {
            float *a = acc_out + (long)ch * L + 2 * ts + 2 * T * r;
            a[0] = acc[r].x; a[1] = acc[r].y;
            // the next call's history = rows S_ROWS..15 of the last frame's window = raw[0 .. 16 - S_ROWS) after the slide: from registers
            float *hs = hist_out + (long)ch * L + 2 * ts + 2 * T * r;
            hs[0] = raw[r].x; hs[1] = raw[r].y;
        }
    }
    pv_signal_done<true>(p.done, done_seq, (long)ch * gridDim.x + chunk);
    if (RESIDENT) goto resident_top;
}

template <int LOG2N, int S_ROWS, bool AUX>
hipError_t launch_wg16(const PvKernelParams &p, int nch, int nchunks, hipStream_t st)
{
    using C = QC<LOG2N>;
    static std::atomic<bool> attr_done[16];
    auto k = pv_wg16_kernel<LOG2N, S_ROWS, AUX>;
    {
        const hipError_t e = pv_set_dynamic_lds_once(attr_done, reinterpret_cast<const void *>(k), C::LDS_BYTES);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k, dim3(nchunks, nch, 1), dim3(C::T, 1, 1), C::LDS_BYTES, st, p);
    return hipGetLastError();
}

template <int LOG2N, int S_ROWS>
hipError_t launch_wg16_resident(const PvKernelParams &p, int nslots, hipStream_t st)
{
    using C = QC<LOG2N>;
    static std::atomic<bool> attr_done[16];
    auto k = pv_wg16_kernel<LOG2N, S_ROWS, false, true>;
    {
        const hipError_t e = pv_set_dynamic_lds_once(attr_done, reinterpret_cast<const void *>(k), C::LDS_BYTES);
        if (e != hipSuccess) return e;
    }
    PvKernelParams q = p;
    q.nchunks = 1; q.nch = nslots; q.nhops = 1; q.frames_per_chunk = 1;
    hipLaunchKernelGGL(k, dim3(1, nslots, 1), dim3(C::T, 1, 1), C::LDS_BYTES, st, q);
    return hipGetLastError();
}

template <int LOG2N>
hipError_t launch_wg16_n(const PvKernelParams &p, int nch, int nchunks, hipStream_t st)
{
    const bool aux = (p.dbg_mag != nullptr);
    switch (16 * p.hop >> LOG2N) {
    case 2: return aux ? launch_wg16<LOG2N, 2, true>(p, nch, nchunks, st) : launch_wg16<LOG2N, 2, false>(p, nch, nchunks, st);
    case 4: return aux ? launch_wg16<LOG2N, 4, true>(p, nch, nchunks, st) : launch_wg16<LOG2N, 4, false>(p, nch, nchunks, st);
    case 8: return aux ? launch_wg16<LOG2N, 8, true>(p, nch, nchunks, st) : launch_wg16<LOG2N, 8, false>(p, nch, nchunks, st);
    case 16: return aux ? launch_wg16<LOG2N, 16, true>(p, nch, nchunks, st) : launch_wg16<LOG2N, 16, false>(p, nch, nchunks, st);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace

bool pv_wg16_supported(int log2n, int hop)
{
    // (the fp64 flavour also takes N = 2048 -- T = 64 threads, M = 16 * 16 * 4 --: C3's shape in reference-width arithmetic; the product runs pv_wave2k_kernel there, which is
    //  3 % faster than this kernel at that size, tools/experiments/README.md)
    if (log2n != 12 && log2n != 13 && !(FP64W && log2n == 11)) return false;
    const int N = 1 << log2n;
    return hop == N / 8 || hop == N / 4 || hop == N / 2 || hop == N;
}
size_t pv_wg16_lds_bytes(int log2n) { return log2n == 11 ? QC<11>::LDS_BYTES : log2n == 12 ? QC<12>::LDS_BYTES : QC<13>::LDS_BYTES; }
int pv_wg16_threads(int log2n) { return log2n == 11 ? QC<11>::T : log2n == 12 ? QC<12>::T : QC<13>::T; }

hipError_t pv_launch_wg16(int log2n, const PvKernelParams &p, int nch, int nchunks, hipStream_t st)
{
    if constexpr (FP64W) { if (log2n == 11) return launch_wg16_n<11>(p, nch, nchunks, st); }
    return log2n == 12 ? launch_wg16_n<12>(p, nch, nchunks, st) : launch_wg16_n<13>(p, nch, nchunks, st);
}

template <int LOG2N>
hipError_t launch_wg16_resident_n(const PvKernelParams &p, int nslots, hipStream_t st)
{
    switch (16 * p.hop >> LOG2N) {
    case 2: return launch_wg16_resident<LOG2N, 2>(p, nslots, st);
    case 4: return launch_wg16_resident<LOG2N, 4>(p, nslots, st);
    case 8: return launch_wg16_resident<LOG2N, 8>(p, nslots, st);
    case 16: return launch_wg16_resident<LOG2N, 16>(p, nslots, st);
    default: return hipErrorInvalidValue;
    }
}

hipError_t pv_launch_wg16_resident(int log2n, const PvKernelParams &p, int nslots, hipStream_t st)
{
    return log2n == 12 ? launch_wg16_resident_n<12>(p, nslots, st) : launch_wg16_resident_n<13>(p, nslots, st);
}
