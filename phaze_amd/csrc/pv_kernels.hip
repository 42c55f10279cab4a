// pv_kernels.hip -- CDNA4 (gfx950) kernels of the phaze hot path.
//
// One workgroup owns one channel and a chain of consecutive STFT frames; the whole frame lives in LDS:
//
//   HBM --coalesced float2--> Hann (fp32) --> packed N/2-pt complex FFT (fp64, LDS, radix-2^2 DIT)
//        --> split pass -> X[0..N/2] (fp64)             replaces fft.realTransform   bundle:90-100,306-442
//        --> |X|^2 -> f32, strict +-2 local maxima      replaces computeMagnitudes / findPeaks  pv:82-116
//        --> per-source-bin owner rule + LDS scatter    replaces shiftPeaks          pv:119-173
//            (rotation = exact N-th root of unity; above-Nyquist residue rebuilt in fp32 when read)
//        --> c2r via packed N/2-pt complex FFT (fp32)   replaces completeSpectrum + inverseTransform +
//                                                        fromComplexArray            bundle:46-51,69-76,102-114
//        --> Hann, overlap-add ring in LDS (reference summation order)   ola:149-157,130-137
//        --> coalesced store of the finished hop        ola:111-118
//
// HBM traffic per channel-frame is the algorithmic 2*hop*4 B (+ a (R-1)-frame halo per chunk); no MFMA:
// the path is LDS/VALU bound.  No compatibility layers: wave64, gfx950 only.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "pv_kernels.h"
#include "pv_signal.h"
#include "pv_mul_rounded.h"

namespace {

// ------------------------------------------------------------------------------------------------
// small complex helpers
// ------------------------------------------------------------------------------------------------
template <typename T> struct vec2;
template <> struct vec2<float> { using type = float2; };
template <> struct vec2<double> { using type = double2; };

template <typename T2> __device__ __forceinline__ T2 cadd(T2 a, T2 b) { return T2{a.x + b.x, a.y + b.y}; }
template <typename T2> __device__ __forceinline__ T2 csub(T2 a, T2 b) { return T2{a.x - b.x, a.y - b.y}; }
template <typename T2> __device__ __forceinline__ T2 cmul(T2 a, T2 b) { return T2{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
template <typename T2> __device__ __forceinline__ T2 cconj(T2 a) { return T2{a.x, -a.y}; }
// multiply by -j (forward) / +j (inverse)
template <bool INV, typename T2> __device__ __forceinline__ T2 cmul_mj(T2 a) { return INV ? T2{-a.y, a.x} : T2{a.y, -a.x}; }

// mul_rounded (a * b rounded to fp32 as an operation of its own, pv:55,67): pv_mul_rounded.h, shared with the register kernels
__device__ __forceinline__ int bitrev(int v, int bits) { return bits == 0 ? 0 : (int)(__brev((unsigned)v) >> (32 - bits)); }
// base-4 digit reversal over nd digits
__device__ __forceinline__ int digitrev4(int v, int nd)
{
    if (nd == 0) return 0;
    unsigned r = __brev((unsigned)v) >> (32 - 2 * nd);
    return (int)(((r & 0x55555555u) << 1) | ((r >> 1) & 0x55555555u));
}

// ------------------------------------------------------------------------------------------------
// In-place radix-2^2 DIT FFT of MP = 2^LOG2M complex points held in LDS in BIT-REVERSED order;
// natural-order output.  tw[k] = exp(-2 pi j k / (2*MP)) (the table of the full frame size N = 2*MP).
// ------------------------------------------------------------------------------------------------
template <typename T, int LOG2M, int THREADS, bool INV>
__device__ __forceinline__ void fft_dit_lds(typename vec2<T>::type *buf, const typename vec2<T>::type *__restrict__ tw, int tid)
{
    using T2 = typename vec2<T>::type;
    constexpr int MP = 1 << LOG2M;
    int log2s = 0;
    if (LOG2M & 1) {
        for (int i = tid; i < MP / 2; i += THREADS) {
            const T2 a = buf[2 * i], b = buf[2 * i + 1];
            buf[2 * i] = cadd(a, b);
            buf[2 * i + 1] = csub(a, b);
        }
        __syncthreads();
        log2s = 1;
    }
    for (; log2s < LOG2M; log2s += 2) {
        const int s = 1 << log2s;
        for (int bf = tid; bf < MP / 4; bf += THREADS) {
            const int j = bf & (s - 1);
            const int i0 = ((bf >> log2s) << (log2s + 2)) + j;
            // W_{4s}^j = tw[j * (2MP / 4s)],  W_{2s}^j = tw[j * (2MP / 2s)]
            T2 w2 = tw[j << (LOG2M - 1 - log2s)];
            T2 w1 = tw[j << (LOG2M - log2s)];
            if (INV) { w2 = cconj(w2); w1 = cconj(w1); }
            const T2 a0 = buf[i0], a1 = cmul(buf[i0 + s], w1), a2 = buf[i0 + 2 * s], a3 = cmul(buf[i0 + 3 * s], w1);
            const T2 b0 = cadd(a0, a1), b1 = csub(a0, a1);
            const T2 b2 = cmul(cadd(a2, a3), w2), b3 = cmul_mj<INV>(cmul(csub(a2, a3), w2));
            buf[i0] = cadd(b0, b2);
            buf[i0 + 2 * s] = csub(b0, b2);
            buf[i0 + s] = cadd(b1, b3);
            buf[i0 + 3 * s] = csub(b1, b3);
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// frame input accessor: sample s of the "extended" channel stream (s < 0 reads the carried history)
// ------------------------------------------------------------------------------------------------
struct FrameSrc {
    const float *in;     // channel base, sample 0 of this call
    const float *hist;   // (N - hop) samples preceding sample 0
    int hist_len;
    __device__ __forceinline__ float at(long s) const { return s < 0 ? hist[s + hist_len] : in[s]; }
};

// ------------------------------------------------------------------------------------------------
// Above-Nyquist residue: what fft.js's in-place real DIT leaves at positions N/2+1..N-1.
// Re-runs the reference's stage structure (bundle:306-442,447-508) on the blocks inside [N/2, N) only, fp32.
// ------------------------------------------------------------------------------------------------
template <int LOG2N, int THREADS>
__device__ __forceinline__ void residue_upper_half(float2 *B, const FrameSrc &src, long s0, const float *__restrict__ hann,
                                                   const float2 *__restrict__ tw32, int tid)
{
    constexpr int N = 1 << LOG2N;
    constexpr bool BASE4 = (LOG2N % 2) == 0;
    if (BASE4) {
        constexpr int nd = (LOG2N - 2) / 2;
        for (int t = N / 8 + tid; t < N / 4; t += THREADS) {          // bundle:468-508
            const int off = digitrev4(t, nd);
            const float a = mul_rounded(src.at(s0 + off), hann[off]);
            const float b = mul_rounded(src.at(s0 + off + N / 4), hann[off + N / 4]);
            const float c = mul_rounded(src.at(s0 + off + N / 2), hann[off + N / 2]);
            const float d = mul_rounded(src.at(s0 + off + 3 * N / 4), hann[off + 3 * N / 4]);
            const float t0 = a + c, t1 = a - c, t2 = b + d, t3 = b - d;
            B[4 * t] = float2{t0 + t2, 0.f};
            B[4 * t + 1] = float2{t1, -t3};
            B[4 * t + 2] = float2{t0 - t2, 0.f};
            B[4 * t + 3] = float2{t1, t3};
        }
    } else {
        constexpr int nd = (LOG2N - 1) / 2;
        for (int t = N / 4 + tid; t < N / 2; t += THREADS) {          // bundle:447-463
            const int off = digitrev4(t, nd);
            const float a = mul_rounded(src.at(s0 + off), hann[off]);
            const float b = mul_rounded(src.at(s0 + off + N / 2), hann[off + N / 2]);
            B[2 * t] = float2{a + b, 0.f};
            B[2 * t + 1] = float2{a - b, 0.f};
        }
    }
    __syncthreads();
    constexpr int LOG2BASE = BASE4 ? 2 : 1;
    for (int log2m = LOG2BASE + 2; log2m <= LOG2N - 2; log2m += 2) {  // block size Mb = 4*base .. N/4
        const int Mb = 1 << log2m, q = Mb >> 2, hq = q >> 1;          // butterflies i = 0..hq per block
        const int nblocks = (N / 2) >> log2m;
        const int tws = LOG2N - log2m;                                 // W_Mb^i = tw[i << tws]
        const int total = nblocks * (hq + 1);
        for (int it = tid; it < total; it += THREADS) {
            int blk, i;
            if (it < nblocks * hq) { blk = it / hq; i = it - blk * hq; } else { blk = it - nblocks * hq; i = hq; }
            const int o = N / 2 + (blk << log2m);
            const float2 A = B[o + i];
            const float2 Bv = cmul(B[o + q + i], tw32[i << tws]);
            const float2 C = cmul(B[o + 2 * q + i], tw32[(2 * i) << tws]);
            const float2 D = cmul(B[o + 3 * q + i], tw32[(3 * i) << tws]);
            const float2 T0 = cadd(A, C), T1 = csub(A, C), T2 = cadd(Bv, D), T3 = csub(Bv, D);
            B[o + i] = cadd(T0, T2);
            B[o + q + i] = float2{T1.x + T3.y, T1.y - T3.x};                     // T1 - j T3
            if (i == 0) {
                B[o + 2 * q] = csub(T0, T2);                                     // bundle:400-406
            } else if (i != hq) {                                                // bundle:409-440
                B[o + q - i] = float2{T1.x - T3.y, -(T1.y + T3.x)};              // conj(T1 + j T3)
                B[o + 2 * q - i] = float2{T0.x - T2.x, -(T0.y - T2.y)};          // conj(T0 - T2)
            }
        }
        __syncthreads();
    }
}

__device__ __forceinline__ double js_round(double x)   // Math.round (pv:125): nearest, ties toward +inf
{
    const double r = floor(x);
    return (x - r >= 0.5) ? r + 1.0 : r;
}

// ------------------------------------------------------------------------------------------------
// The chain kernel
// ------------------------------------------------------------------------------------------------
// GM (round 5; the sizes beyond the LDS of a CU, which the reference takes like any other power of two, bundle:4-8): 0 = every buffer in LDS (N <= 8192);
// 1 = the fp32 buffer B and the overlap-add ring in a global-memory scratch of the workgroup (N = 16384: the fp64 buffer alone is 128 KB); 2 = the fp64 buffer as well
// (N = 32768).  The workgroup's barriers order its global accesses as they order its LDS accesses (workgroup-scope fences).  Slow and complete: no BASELINE config.
template <int LOG2N, int THREADS, int GM = 0>
__global__ __launch_bounds__(THREADS) void pv_chain_kernel(const PvKernelParams p)
{
    constexpr int N = 1 << LOG2N, M = N / 2, H = M + 1, LOG2M = LOG2N - 1;
    constexpr int NWORDS = (H + 63) / 64;
    constexpr int NWAVES = THREADS / 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hop = p.hop, R = N / hop, L = N - hop;
    const int ch = blockIdx.y, chunk = blockIdx.x;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // GM != 0 (N >= 16384): B, the overlap-add ring, the claim words (and at GM == 2 the fp64 buffer) live in a per-workgroup slice of DEVICE memory and are ordered by
    // __syncthreads() alone, agent-scope atomicMin next to plain loads and stores.  That is sound because a workgroup's waves share ONE vector L1 (the default CU mode):
    // under -mtgsplit (threadgroup-split mode: the waves of a workgroup may sit on different CUs with different L1s) it would need workgroup-scope atomics / fences
    // around every exchange.  The build never uses that mode (the compiler defines no macro for it: the Makefile refuses the flag).
    unsigned char *gm = GM ? p.gscratch + ((size_t)ch * gridDim.x + chunk) * p.gscratch_stride : nullptr;
    unsigned char *abase = (GM == 2) ? gm : smem;                          // where the fp64 buffer lives
    unsigned char *bbase = (GM == 2) ? gm + 16 * (M + 1) : (GM == 1) ? gm : smem + 16 * (M + 1);
    double2 *A = reinterpret_cast<double2 *>(abase);                      // [M+1] fp64: packed FFT, then X[0..M]
    float2 *B = reinterpret_cast<float2 *>(bbase);                        // [N] fp32: mag (alias) / Y[0..M] / residue (M, N)
    float *acc = reinterpret_cast<float *>(bbase + 8 * N);                // [L] overlap-add ring
    unsigned long long *masks = reinterpret_cast<unsigned long long *>(GM == 2 ? smem : GM == 1 ? smem + 16 * (M + 1) : smem + 16 * (M + 1) + 8 * N + 4 * ((L + 1) & ~1));
    int *wprev = reinterpret_cast<int *>(masks + NWORDS);                 // largest peak in words < w (or -1)
    int *wnext = wprev + NWORDS;                                          // smallest peak in words > w (or BIG)
    float *magv = reinterpret_cast<float *>(B);                           // alias: mags die before Y is zeroed
    float2 *Zb = reinterpret_cast<float2 *>(A);                           // alias: inverse FFT runs where X lived
    const float *frame = reinterpret_cast<const float *>(A);              // real output of the c2r transform
    float2 *Af = reinterpret_cast<float2 *>(A);                           // X[0..M] rounded to fp32, in place over the first half of A (after the decisions)
    unsigned *CLAIM = reinterpret_cast<unsigned *>(abase + 8 * (M + 1) + 8);  // [H] claim words of the f < 1 scatter, in the half of A the rounding frees
    constexpr int BIG = 1 << 30;

    const int first_out = chunk * p.frames_per_chunk;
    int last_out = first_out + p.frames_per_chunk;
    if (last_out > p.nhops) last_out = p.nhops;
    int first_frame = first_out - (R - 1);                                // halo: frames that still overlap first_out
    const bool from_state = (first_frame <= 0);                           // the carried accumulator still reaches first_out
    if (from_state) first_frame = 0;

    const long cbase = (long)ch * p.ch_stride;
    FrameSrc src{p.in + cbase, p.hist_in + (long)ch * L, L};
    float *outp = p.out + cbase;
    const float *pitch_row = p.pitch + (p.pitch_stride ? (long)(ch / p.ch_per_stream) * p.pitch_stride : 0);
    const float invR = 1.0f / (float)R;                                   // exact: R is a power of two

    // accumulator ring: chunks whose halo reaches hop 0 resume from the carried state, later chunks rebuild it from the halo frames
    for (int j = tid; j < L; j += THREADS) acc[j] = from_state ? p.acc_in[(long)ch * L + j] : 0.f;
    int ring = 0;
    __syncthreads();

    for (int m = first_frame; m < last_out; ++m) {
        const long s0 = (long)(m + 1) * hop - N;                           // first sample of the window (ola:121-127)
        const double pf = (double)pitch_row[m];                            // f32 widened (pv:47)
        const int tmod = (int)(((long)p.t0_mod_n + (long)m * hop) & (N - 1));   // timeCursor mod N (pv:71)

        // ---- 1. load + Hann (pv:55,75-79) -> packed z[n] = xw[2n] + j xw[2n+1], bit-reversed into LDS ----
        for (int n = tid; n < M; n += THREADS) {
            const float x0 = src.at(s0 + 2 * n) * p.hann[2 * n];
            const float x1 = src.at(s0 + 2 * n + 1) * p.hann[2 * n + 1];
            A[bitrev(n, LOG2M)] = double2{(double)x0, (double)x1};
        }
        __syncthreads();
        // ---- 2. N/2-point complex FFT, fp64 ----
        fft_dit_lds<double, LOG2M, THREADS, false>(A, p.tw64, tid);
        // ---- 3. split pass: X[k], X[M-k] from Z[k], Z[M-k] (in place); X[0], X[M] real ----
        for (int k = tid; k <= M / 2; k += THREADS) {
            if (k == 0) {
                const double2 z0 = A[0];
                A[0] = double2{z0.x + z0.y, 0.0};
                A[M] = double2{z0.x - z0.y, 0.0};
            } else {
                const double2 zk = A[k], zm = A[M - k];
                const double2 E{0.5 * (zk.x + zm.x), 0.5 * (zk.y - zm.y)};
                const double2 O{0.5 * (zk.x - zm.x), 0.5 * (zk.y + zm.y)};
                const double2 jwo = cmul_mj<true>(cmul(p.tw64[k], O));     // j * W^k * O
                A[k] = csub(E, jwo);
                if (k != M - k) A[M - k] = cconj(cadd(E, jwo));
            }
        }
        __syncthreads();
        // ---- 4. |X|^2 in fp64, rounded once to f32 (pv:82-92).  X itself is then rounded to fp32 IN PLACE (the shift only moves fp32 values):
        //          the half of A this frees holds the claim words of the f < 1 scatter ----
        {
            constexpr int XPT = (H + THREADS - 1) / THREADS;
            float2 xf[XPT];
            const bool dbgf = p.dbg_mag && ch == p.dbg_ch && m == p.dbg_frame;
#pragma unroll
            for (int i = 0; i < XPT; i++) {
                const int k = tid + i * THREADS;
                if (k < H) {
                    const double2 v = A[k];
                    magv[k] = (float)(v.x * v.x + v.y * v.y);
                    xf[i] = float2{(float)v.x, (float)v.y};
                    if (dbgf) { p.dbg_X[2 * k] = v.x; p.dbg_X[2 * k + 1] = v.y; }
                }
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < XPT; i++) { const int k = tid + i * THREADS; if (k < H) Af[k] = xf[i]; }
        }
        __syncthreads();
        // ---- 5. strict +-2 local maxima (pv:95-116) as a local predicate -> bit masks via ballot ----
        // Non-finite magnitudes (NaN / Inf samples in the window): in the reference every comparison with a NaN fails to REJECT (pv:103,107),
        // peaks appear at every other bin and the NaNs they move reach every output sample: the frame is NaN.  The predicate below finds no
        // peak there, so such a frame is detected here and poisoned after the scatter (same rule as the register kernels).
        bool nonfinite = false;
        for (int w = wave; w < NWORDS; w += NWAVES) {
            const int k = w * 64 + lane;
            bool f = false, nf = false;
            if (k < H) nf = !(magv[k] < __builtin_huge_valf());
            if (k >= 2 && k < H - 2) {
                const float mg = magv[k];
                f = (magv[k - 1] < mg) && (magv[k - 2] < mg) && (magv[k + 1] < mg) && (magv[k + 2] < mg);
            }
            const unsigned long long bm = __ballot(f);
            nonfinite |= __any(nf);
            if (lane == 0) masks[w] = bm;
        }
        __syncthreads();
        // ---- 6. per-word nearest peaks on either side ----
        for (int w = tid; w < NWORDS; w += THREADS) {
            int pv = -1, nx = BIG;
            for (int v = w - 1; v >= 0; --v) { const unsigned long long mk = masks[v]; if (mk) { pv = v * 64 + 63 - __clzll(mk); break; } }
            for (int v = w + 1; v < NWORDS; ++v) { const unsigned long long mk = masks[v]; if (mk) { nx = v * 64 + __ffsll(mk) - 1; break; } }
            wprev[w] = pv;
            wnext[w] = nx;
        }
        __syncthreads();
        // last peak decides whether bins above Nyquist are ever read (pv:133: endIndex = fftSize)
        int last_peak;
        {
            const unsigned long long mk = masks[NWORDS - 1];
            last_peak = mk ? (NWORDS - 1) * 64 + 63 - __clzll(mk) : wprev[NWORDS - 1];
        }
        int upper_end = H;                                                 // sources b in [H, upper_end) contribute
        int last_delta = 0;
        if (last_peak >= 0) {
            const double psh = js_round((double)last_peak * pf);
            if (!(psh > (double)H) && psh >= -(double)(2 * N)) {
                last_delta = (int)psh - last_peak;
                if (last_delta < 0) { upper_end = H - last_delta; if (upper_end > N) upper_end = N; }
            }
        }
        if (p.dbg_mag && ch == p.dbg_ch && m == p.dbg_frame) {
            for (int k = tid; k < H; k += THREADS) {
                p.dbg_mag[k] = magv[k];
                p.dbg_flags[k] = (int)((masks[k >> 6] >> (k & 63)) & 1ull);
            }
        }
        // ---- 7. residue above Nyquist, only when the last region reads it ----
        if (upper_end > H) {
            residue_upper_half<LOG2N, THREADS>(B, src, s0, p.hann, p.tw32, tid);
            if (p.dbg_mag && ch == p.dbg_ch && m == p.dbg_frame)
                for (int k = H + tid; k < N; k += THREADS) { p.dbg_X[2 * k] = B[k].x; p.dbg_X[2 * k + 1] = B[k].y; }
        }
        // ---- 8. zero the shifted spectrum (pv:121) ----
        for (int k = tid; k < H; k += THREADS) B[k] = float2{0.f, 0.f};
        __syncthreads();
        // ---- 9. shiftPeaks (pv:119-173) as a per-source-bin rule + LDS scatter.  f >= 1: the shifted regions are disjoint -> plain stores.
        //          f < 1: regions compress and `+=` collisions happen (pv:169-170).  They are resolved in CLAIM ROUNDS: every pending source
        //          posts its bin with an LDS atomic MIN on the claim word of its target, the smallest bin wins the round and does a plain
        //          read-modify-write.  Each target therefore accumulates its contributions in ascending source order -- the order of the
        //          reference's loops (pv:122,146) -- whatever the timing of the waves: chunked, unchunked and call-split runs agree bit for
        //          bit for every f (float atomics, used here before, add in arrival order) ----
        const bool disjoint = (pf >= 1.0);
        // (the global-scratch instances take their sources in batches of 16 per thread, ascending: the order of accumulation is the same, the registers are not)
        constexpr int SPT_ALL = (N + THREADS - 1) / THREADS;              // sources per thread
        constexpr int NBATCH = GM ? (SPT_ALL + 15) / 16 : 1;
        constexpr int SPT = SPT_ALL / NBATCH;                             // ... per batch (<= 32)
        static_assert(SPT * NBATCH == SPT_ALL && SPT <= 32, "sources per thread");
        if (!disjoint) {
            for (int k = tid; k < H; k += THREADS) CLAIM[k] = 0xFFFFFFFFu;
            if (NBATCH > 1) __syncthreads();
        }
        for (int bt = 0; bt < NBATCH; bt++) {
        float2 ys[SPT];
        using tgt_t = typename std::conditional<(LOG2N > 16), unsigned, unsigned short>::type;   // target bin < H (N = 131072: 17 bits)
        tgt_t tg[SPT];
        unsigned pend = 0;
#pragma unroll
        for (int i = 0; i < SPT; i++) {
            const int b = tid + (bt * SPT + i) * THREADS;
            ys[i] = float2{0.f, 0.f};
            tg[i] = 0;
            if (b >= upper_end) continue;
            int prv, nxt;
            if (b < H) {
                const int w = b >> 6, bit = b & 63;
                const unsigned long long mk = masks[w];
                const unsigned long long lo = (bit == 63) ? mk : (mk & ((2ull << bit) - 1ull));
                const unsigned long long hi = (bit == 63) ? 0ull : (mk & ~((2ull << bit) - 1ull));
                prv = lo ? w * 64 + 63 - __clzll(lo) : wprev[w];
                nxt = hi ? w * 64 + __ffsll(hi) - 1 : wnext[w];
            } else {
                prv = last_peak;
                nxt = BIG;
            }
            int owner;
            if (prv < 0) owner = nxt;                                      // region of the first peak starts at 0 (pv:132)
            else if (nxt == BIG) owner = prv;                              // region of the last peak ends at N (pv:133)
            else owner = (b < prv + ((nxt - prv + 1) >> 1)) ? prv : nxt;   // pv:134-141
            if (owner == BIG || owner < 0) continue;                      // no peaks at all
            const double psh = js_round((double)owner * pf);               // pv:125
            if (!(psh <= (double)H) || psh < -(double)(2 * N)) continue;   // pv:127-129 break (NaN: no effect)
            const int delta = (int)psh - owner;
            const int tgt = b + delta;
            if (tgt < 0 || tgt >= H) continue;                             // pv:150-152; negative index: named property
            const int ridx = (int)(((unsigned)(delta & (N - 1)) * (unsigned)tmod) & (unsigned)(N - 1));   // (delta * t) mod N  (pv:155-157); unsigned: the product wraps at N >= 65536
            const float2 rot = cconj(p.tw32[ridx]);                        // exp(+2 pi j ridx / N)
            const float2 v = (b < H) ? Af[b] : B[b];
            ys[i] = cmul(v, rot);
            tg[i] = (tgt_t)tgt;
            if (disjoint) B[tgt] = ys[i];                                  // f >= 1: delta_i non-decreasing => shifted regions never overlap
            else pend |= 1u << i;
        }
        if (!disjoint) {
            while (__syncthreads_or(pend != 0u)) {
#pragma unroll
                for (int i = 0; i < SPT; i++) if (pend & (1u << i)) atomicMin(&CLAIM[tg[i]], (unsigned)(tid + (bt * SPT + i) * THREADS));
                __syncthreads();
#pragma unroll
                for (int i = 0; i < SPT; i++) {
                    if ((pend & (1u << i)) && CLAIM[tg[i]] == (unsigned)(tid + (bt * SPT + i) * THREADS)) {
                        const float2 o = B[tg[i]];
                        B[tg[i]] = float2{o.x + ys[i].x, o.y + ys[i].y};
                        CLAIM[tg[i]] = 0xFFFFFFFFu;                        // only the winner touches the word; losers re-post after the barrier
                        pend &= ~(1u << i);
                    }
                }
            }
        }
        }
        if (nonfinite && lane == 0) B[1 + wave] = float2{__uint_as_float(0x7FC00000u), __uint_as_float(0x7FC00000u)};
        __syncthreads();
        if (p.dbg_Y && ch == p.dbg_ch && m == p.dbg_frame)
            for (int k = tid; k < H; k += THREADS) { p.dbg_Y[2 * k] = B[k].x; p.dbg_Y[2 * k + 1] = B[k].y; }
        // ---- 10. c2r pre-pass: Z[k] = (Y[k] + conj(Y[M-k])) + j e^{+2 pi j k/N} (Y[k] - conj(Y[M-k])), scaled 1/N,
        //          written bit-reversed into the (now dead) fp64 buffer ----
        {
            const float sc = 1.0f / (float)N;
            for (int k = tid; k <= M / 2; k += THREADS) {
                float2 yk = B[k], ym = B[M - k];
                if (k == 0) { yk.y = 0.f; ym.y = 0.f; }                    // Im Y[0], Im Y[N/2] never reach Re (bundle:46-51)
                const float2 E{yk.x + ym.x, yk.y - ym.y};
                const float2 O{yk.x - ym.x, yk.y + ym.y};
                const float2 jc = cmul_mj<true>(cmul(cconj(p.tw32[k]), O));
                const float2 zk{(E.x + jc.x) * sc, (E.y + jc.y) * sc};
                const float2 zm{(E.x - jc.x) * sc, -(E.y - jc.y) * sc};
                if (k == 0) {
                    Zb[0] = zk;
                } else {
                    Zb[bitrev(k, LOG2M)] = zk;
                    if (k != M - k) Zb[bitrev(M - k, LOG2M)] = zm;
                }
            }
        }
        __syncthreads();
        // ---- 11. N/2-point inverse complex FFT, fp32: frame[2n] = Re z[n], frame[2n+1] = Im z[n] ----
        fft_dit_lds<float, LOG2M, THREADS, true>(Zb, p.tw32, tid);
        // ---- 12. Hann (pv:67), overlap-add in reference order (ola:149-157), emit hop (ola:111-118), shift (ola:130-137) ----
        const bool emit = (m >= first_out);
        for (int j = tid; j < hop; j += THREADS) {
            const float fr = mul_rounded(frame[j], p.hann[j]);               // rounded to fp32 before the accumulation (Float32Array, pv:67)
            float a = 0.f;
            int slot = 0;
            if (L > 0) { slot = ring + j; if (slot >= L) slot -= L; a = acc[slot]; }
            const float o = a + fr * invR;
            if (emit) outp[(long)m * hop + j] = o;
            if (L > 0) acc[slot] = mul_rounded(frame[j + L], p.hann[j + L]) * invR;   // freed slot receives the new tail (0 + x)
        }
        for (int j = hop + tid; j < L; j += THREADS) {
            int slot = ring + j; if (slot >= L) slot -= L;
            acc[slot] = acc[slot] + mul_rounded(frame[j], p.hann[j]) * invR;
        }
        if (L > 0) { ring += hop; if (ring >= L) ring -= L; }
        __syncthreads();
    }

    // ---- carry state out: accumulator tail + input history (ping-pong buffers, written by the last chunk) ----
    if (chunk == (int)gridDim.x - 1) {
        for (int j = tid; j < L; j += THREADS) {
            int slot = ring + j; if (slot >= L) slot -= L;
            p.acc_out[(long)ch * L + j] = acc[slot];
            p.hist_out[(long)ch * L + j] = src.at((long)p.nhops * hop - L + j);
        }
    }
    pv_signal_done<true>(p.done, p.done_seq, (long)ch * gridDim.x + chunk);
}

template <int LOG2N, int THREADS, int GM = 0>
hipError_t launch_one(const PvKernelParams &p, int nch, int nchunks, size_t lds, hipStream_t st)
{
    static std::atomic<bool> attr_done[16];
    auto k = pv_chain_kernel<LOG2N, THREADS, GM>;
    if (GM && (p.gscratch == nullptr || p.gscratch_stride < pv_kernel_gscratch_bytes(LOG2N, p.hop))) return hipErrorInvalidValue;
    {
        const hipError_t e = pv_set_dynamic_lds_once(attr_done, reinterpret_cast<const void *>(k), 160 * 1024 - 512);   // __syncthreads_or keeps a few static bytes
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k, dim3(nchunks, nch, 1), dim3(THREADS, 1, 1), lds, st, p);
    return hipGetLastError();
}

}  // namespace

int pv_kernel_threads(int log2n)
{
    if (log2n <= 10) return 64;
    if (log2n == 11) return 128;
    return log2n <= 13 ? 256 : 512;
}

size_t pv_kernel_lds_bytes(int log2n, int hop)
{
    const int N = 1 << log2n, M = N / 2, H = M + 1, L = N - hop;
    const int nwords = (H + 63) / 64;
    if (log2n >= 15) return (size_t)nwords * 16;                                       // masks + nearest-peak words only
    if (log2n == 14) return (size_t)16 * (M + 1) + (size_t)nwords * 16;                // + the fp64 buffer
    return (size_t)16 * (M + 1) + (size_t)8 * N + (size_t)4 * ((L + 1) & ~1) + (size_t)nwords * 16;
}

size_t pv_kernel_gscratch_bytes(int log2n, int hop)
{
    if (log2n < 14) return 0;
    const size_t N = (size_t)1 << log2n, M = N / 2, L = N - (size_t)hop;
    const size_t b = 8 * N + 4 * ((L + 1) & ~(size_t)1) + (log2n >= 15 ? 16 * (M + 1) : 0);
    return (b + 255) & ~(size_t)255;
}

hipError_t pv_launch_chain(int log2n, const PvKernelParams &p, int nch, int nchunks, hipStream_t st)
{
    const size_t lds = pv_kernel_lds_bytes(log2n, p.hop);
    switch (log2n) {
    case 1: return launch_one<1, 64>(p, nch, nchunks, lds, st);
    case 2: return launch_one<2, 64>(p, nch, nchunks, lds, st);
    case 3: return launch_one<3, 64>(p, nch, nchunks, lds, st);
    case 4: return launch_one<4, 64>(p, nch, nchunks, lds, st);
    case 5: return launch_one<5, 64>(p, nch, nchunks, lds, st);
    case 6: return launch_one<6, 64>(p, nch, nchunks, lds, st);
    case 7: return launch_one<7, 64>(p, nch, nchunks, lds, st);
    case 8: return launch_one<8, 64>(p, nch, nchunks, lds, st);
    case 9: return launch_one<9, 64>(p, nch, nchunks, lds, st);
    case 10: return launch_one<10, 64>(p, nch, nchunks, lds, st);
    case 11: return launch_one<11, 128>(p, nch, nchunks, lds, st);
    case 12: return launch_one<12, 256>(p, nch, nchunks, lds, st);
    case 13: return launch_one<13, 256>(p, nch, nchunks, lds, st);
    case 14: return launch_one<14, 512, 1>(p, nch, nchunks, lds, st);
    case 15: return launch_one<15, 512, 2>(p, nch, nchunks, lds, st);
    case 16: return launch_one<16, 512, 2>(p, nch, nchunks, lds, st);
    case 17: return launch_one<17, 512, 2>(p, nch, nchunks, lds, st);
    case 18: return launch_one<18, 512, 2>(p, nch, nchunks, lds, st);
    case 19: return launch_one<19, 512, 2>(p, nch, nchunks, lds, st);
    case 20: return launch_one<20, 512, 2>(p, nch, nchunks, lds, st);
    default: return hipErrorInvalidValue;
    }
}
