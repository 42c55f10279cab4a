// pv_wave_kernel.hip -- wave-per-frame kernel for N = 1024 (BASELINE configs[0..1], the bench workload).
//
// One 64-lane wavefront owns one channel and a chain of consecutive frames; 12 independent chains share a workgroup (and its LDS
// tables) = 3 waves per SIMD.  Everything that can stay in registers does; LDS carries only the two register<->lane transposes of
// each FFT, the magnitude / route exchange and the shifted spectrum Y:
//
//   lane l, register r  <->  packed complex element z[l + 64 r]   (z[n] = xw[2n] + j xw[2n+1], N/2 = 512 = 8*8*8)
//
//   * raw input samples slide in registers (hop = 128*S samples = S register rows): S new float2 loads per frame, issued a frame ahead
//   * forward 512-pt complex FFT in fp64 (pv:57; fp64 because the peak decisions are taken on the f32-rounded |X|^2 of an fp64
//     spectrum): three radix-8 butterflies per lane, two conflict-free LDS transposes (layouts from tools/lds_layout_check.py),
//     twiddles from LDS tables shared by the workgroup
//   * split pass in conjugate pairs (one twiddle product per pair k, 512-k); the partner values cross lanes through the freed transpose
//     scratch; |X|^2 -> f32 in registers (pv:82-92)
//   * peak flags on 8 consecutive bins per lane (pv:95-116); candidate peaks travel as packed (bin, shift) words: own shifts from one
//     16-byte read of the cached Math.round(p f) - p table, nearest peaks by select chains + one ballot + two bpermutes; owner rule ->
//     one route per source bin (pv:119-152)
//   * scatter of the register-resident source bins along their routes (pv:155-170): plain stores when f >= 1 (regions disjoint; frames
//     with t = 0 or N/2 (mod N) skip / simplify the rotation); f < 1 goes out of line: store-then-add when every collision of the frame is a
//     (falling side, rising side) pair -- the usual case down to f ~ 0.65 --, claim rounds otherwise; the above-Nyquist residue either
//     from the spectrum (decimation identity) or by re-running the reference's stage structure on one quarter
//   * c2r pre-pass (conjugate pairs) + 512-pt inverse FFT in packed fp32 (pv_pk_math.h), Hann, overlap-add accumulator in registers in
//     reference order (ola:149-157), finished hop stored coalesced, non-temporal
//
// Semantics are those of pv_chain_kernel (same reference citations); tests run every kernel against the oracle.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "pv_kernels.h"
#include "pv_device_common.h"
#include "pv_pk_math.h"
#include "pv_wave_fft.h"
#include "pv_guard.h"

namespace {

// Reference-width flavour (never the product; `make fp64` -> build/exp/libphaze_fp64.so): with -DPV_FP64_FLAVOUR=1 the shifted spectrum,
// the scatter (plain stores for f >= 1, claim rounds for f < 1), the above-Nyquist residue, the c2r pass and the inverse FFT run in fp64 like the
// reference (freqComplexBufferShifted / inverseTransform are JS doubles, bundle:102-114; phase-vocoder.js:37-39,161-170) -- for every pitchFactor.
// It is the SPREAD flow below (sources in registers, routes through LDS, Y a plain array) extended by the colliding scatter; 8 waves per workgroup
// (double-width Y, hand-over and quarter buffers), one kernel instance for all chains.  It exists to price the fp32 shift / inverse of the product
// (parity against the oracle tightens from ~6e-9 to ~1e-10 RMS) and is run by tests/test_gpu_fp64_flavour.py and one line of bench.py.
#ifndef PV_FP64_FLAVOUR
#define PV_FP64_FLAVOUR 0
#endif
#ifndef PV_WAVES
#define PV_WAVES (PV_FP64_FLAVOUR ? 8 : 12)
#endif
#ifndef PV_WAVES_PER_SIMD
#define PV_WAVES_PER_SIMD (PV_FP64_FLAVOUR ? 2 : 3)
#endif
constexpr bool FP64 = PV_FP64_FLAVOUR != 0;
constexpr int WAVES = PV_WAVES;                  // independent frame chains per workgroup (they only share the LDS tables)
constexpr int RES_WAVES = 4;                    // ... of the resident streaming instance: one wave per SIMD, i.e. the whole register file (no spills on the latency path)
constexpr int TAB_TW1 = 0;                       // double2[8*64]  W_512^{l k}
constexpr int TAB_TW2 = TAB_TW1 + 8 * 64 * 16;   // double2[8*8]   W_64^{n0 k}
// The fp32 tables are stored in PAIRS of rows (a ds_read costs the LDS pipe the same ~3 cycles whether it returns 8 or 16 bytes per lane,
// tools/valu_microbench2.hip): one ds_read_b128 fetches the entries of two consecutive k (twiddles) / two consecutive register rows (Hann).
constexpr int TAB_TW1F = TAB_TW2 + 8 * 8 * 16;   // float4[4*64]   conj(W_512^{l k}), fp32 (inverse): entry [j*64 + l] = (k = 2j, k = 2j+1)
constexpr int TAB_TW2F = TAB_TW1F + 4 * 64 * 16; // float4[4*8]    conj(W_64^{n0 k}): entry [j*8 + n0] = (k = 2j, k = 2j+1)
constexpr int TAB_HANN = TAB_TW2F + 4 * 8 * 16;  // float4[4*64]   0.5 * Hann at samples 2n, 2n+1 for n = l + 64 r: entry [j*64 + l] = (r = 2j, r = 2j+1).
                                                 //                ONE table serves both windows: the 1/2 of the split pass is folded in for the analysis
                                                 //                window, and the synthesis side folds 2/R into the scale of the c2r pass (all exact)
constexpr int TAB_TW1C = TAB_HANN + 4 * 64 * 16;   // fp64 flavour only: double2[8*64] conj(W_512^{l k}), double2[8*8] conj(W_64^{n0 k}) for its fp64 inverse FFT
constexpr int TAB_TW2C = TAB_TW1C + 8 * 64 * 16;
constexpr int TAB_BYTES = FP64 ? TAB_TW2C + 8 * 8 * 16 : TAB_TW1C;   // 17920 (27136)

// conj(W_512^{l k}) in fp32 from the pair-interleaved table (residue paths)
__device__ __forceinline__ float2 tw1f_at(int k, int l)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
    return reinterpret_cast<const float2 *>(smem_all + TAB_TW1F)[2 * ((k >> 1) * 64 + l) + (k & 1)];
}

// o * exp(+2 pi j r / 16), r = 0..3 (compile-time): the wave-uniform part of the c2r twiddle, packed
__device__ __forceinline__ pk::c32 mul_w16_inv_pk(pk::c32 o, int r)
{
    const float c = 0.92387953251128675613f, sn = 0.38268343236508977173f, h = 0.70710678118654752440f;
    switch (r) {
    case 0: return o;
    case 1: return pk::cmul(o, pk::c32{c, sn});
    case 2: return pk::mul(pk::add_j(o, o), pk::c32{h, h});
    default: return pk::cmul(o, pk::c32{sn, c});
    }
}


__device__ __forceinline__ double shfl_d(double v, int src) { return __shfl(v, src, 64); }

// Phase clock of the measurement builds (-DPV_STAMPS=1 coarse, =2 with the arithmetic / exchange split inside the FFTs): s_memtime deltas
// accumulated per phase in SGPRs, written once per chain.  Nothing of this exists in the product build.
#ifdef PV_STAMPS
struct Stamps {
    unsigned prev, acc[13];
    static __device__ __forceinline__ unsigned now()
    {
        __builtin_amdgcn_sched_barrier(0);
        const unsigned t = (unsigned)__builtin_amdgcn_s_memtime();
        __builtin_amdgcn_sched_barrier(0);
        return t;
    }
    __device__ __forceinline__ void start() { for (int i = 0; i < 13; i++) acc[i] = 0; prev = now(); }
    __device__ __forceinline__ void mark(int id, bool fine = false)
    {
        if (fine && PV_STAMPS < 2) return;
        const unsigned t = now();
        acc[id] += t - prev;
        prev = t;
    }
};
#define PV_STAMP(id) stamps.mark(id)
#define PV_STAMP_FINE(id) stamps.mark(id, true)
#else
#define PV_STAMP(id)
#define PV_STAMP_FINE(id)
#endif


// ---- per-wave LDS region (byte offsets) ----
// The shifted spectrum Y and the fp32 spectrum stash XS are stored TRANSPOSED, eight rows of 68 slots:
//     slot(t) = (t & 7) * 68 + (t >> 3),     byte address = 8 * slot(t) = t + (t & 7) * 543
// because both are touched in two lane orders.  "Strided": lane l <-> bins l + 64 r (how the FFTs leave and take the spectrum): (l & 7) * 68 + (l >> 3)
// + 8 r, the sixteen lanes of a ds_write_b64 group land on sixteen different bank pairs, r is an immediate offset.  "Natural": lane l <-> bins
// 8 l + i (the peak search, and since round 4 the scatter): slot = 68 i + l, consecutive lanes on consecutive slots, i an immediate offset.  A
// plain array is conflict-free in the first order and a 16-way bank conflict in the second (64 bytes between neighbouring lanes).
constexpr int YROW = 68;
constexpr int YSLOTS = 8 * YROW;    // 544 slots, 4352 bytes
__device__ __forceinline__ unsigned yslot_bytes(unsigned t) { return __umul24(t & 7u, 543u) + t; }

constexpr int OFF_Y = 0;            // float2[544]   shifted spectrum, transposed (aliases the fp64 transpose scratch: dead between the FFTs)
constexpr int OFF_MAG = 4112;       // f32[520]      |X|^2: MAG[4 + k], k in [-4, 516) -- behind the four partner rows of the split pass, which are still being read
                                    //               while the magnitudes are written.  Y's tail overlaps it: Y is zeroed after the magnitudes have been read
constexpr int OFF_CLAIM = 4352;     // u16[544]      claim words of the fallback scatter (inside the dead magnitudes, behind Y)
constexpr int OFF_ROUTE = 4352;     // u32[528]      f >= 1 frames: route of source bin b, from the lanes that compute it to the lanes that hold the bin (dead magnitudes +
                                    //               the head of the stash, which only f < 1 frames use)
constexpr int OFF_XS = 6224;        // float2[544]   fp32 spectrum stash, transposed: written by the split pass (strided), read by the scatter (natural) and the fast residue
constexpr int OFF_RESQ = 6224;      // float2[256]   one quarter of the above-Nyquist residue at a time (general path) | c2r hand-over -- alias XS: never live together
// fp64 flavour (SPREAD flow only): Y double2[513] at 0 | routes, later claim words at F64_ROUTE | one quarter of the residue / c2r hand-over, double2[256], at F64_Q
constexpr int F64_ROUTE = 8224, F64_Q = F64_ROUTE + 2112;
constexpr int OFF_PSH = FP64 ? F64_Q + 4096 : OFF_XS + 8 * YSLOTS;   // i16[512]  shift table Math.round(p * f) - p
constexpr int WAVE_LDS = OFF_PSH + 1024;       // 11600: 17920 + 12 * 11600 = 157120 B per workgroup (<= 160 KB); fp64 flavour 15456: 27136 + 8 * 15456 = 150784

// f < 1: regions compress and `+=` collisions happen (pv:169-170).  LDS float atomics serialise per lane (measured: half of the frame
// time), so collisions that are not simple pairs (see "pairwise" in the kernel) are resolved by CLAIM ROUNDS: every pending source writes its id
// to CLAIM[target], the id that sticks wins the round and does a plain read-modify-write on Y; losers retry.  Rounds = max multiplicity of a
// target (2-3 for f >= 0.3).
// YZERO: the caller guarantees that Y is still all zero (first scatter of a frame): the winners of round 1 then store their value instead
// of a read-modify-write (0 + v == v bit for bit, also for v = -0: Y is +0 and +0 + -0 = +0 -- so a -0 component is stored as +0 explicitly).
template <int NS, bool YZERO>
__device__ __forceinline__ void claim_rounds(const unsigned (&rt)[NS], const float2 (&ys)[NS], const int (&id)[NS], unsigned char *Yb, unsigned short *CLAIM)
{
    unsigned pend = 0;                                                     // (a bool per source would be carried through the loop as 0/1 VGPRs: slower)
    unsigned ta[NS];                                                       // byte address of the target's slot (both arrays are transposed: CLAIM[slot], Y[slot])
#pragma unroll
    for (int r = 0; r < NS; r++) {
        const unsigned t = rt[r] & 0xFFFFu;
        const bool ok = t < 513u;                                          // valid route <=> target field < H
        pend |= ok ? (1u << r) : 0u;
        ta[r] = yslot_bytes(ok ? t : 0u);                                  // in-range address for the unconditional reads below
    }
    auto claim = [&](int r) -> unsigned short & { return CLAIM[ta[r] >> 3]; };
    auto yat = [&](int r) -> float2 & { return *reinterpret_cast<float2 *>(Yb + ta[r]); };
    if (YZERO) {
#pragma unroll
        for (int r = 0; r < NS; r++) if (pend & (1u << r)) claim(r) = (unsigned short)id[r];
        wave_sync();
        unsigned short c[NS];
#pragma unroll
        for (int r = 0; r < NS; r++) c[r] = claim(r);
#pragma unroll
        for (int r = 0; r < NS; r++) {
            if ((pend & (1u << r)) && c[r] == (unsigned short)id[r]) {
                yat(r) = float2{0.f + ys[r].x, 0.f + ys[r].y};             // what the reference's += leaves in a zeroed bin (pv:121,169-170)
                pend &= ~(1u << r);
            }
        }
        wave_sync();
    }
    while (__any(pend != 0u)) {
#pragma unroll
        for (int r = 0; r < NS; r++) if (pend & (1u << r)) claim(r) = (unsigned short)id[r];
        wave_sync();
        // all claim words and all current Y values first (two batches of independent reads, one wait), then the winners' stores:
        // a per-source `if (CLAIM == id) { read Y; write Y }` costs two dependent LDS round trips per source instead
        unsigned short c[NS];
        float2 o[NS];
#pragma unroll
        for (int r = 0; r < NS; r++) c[r] = claim(r);
#pragma unroll
        for (int r = 0; r < NS; r++) o[r] = yat(r);
#pragma unroll
        for (int r = 0; r < NS; r++) {
            if ((pend & (1u << r)) && c[r] == (unsigned short)id[r]) {
                yat(r) = float2{o[r].x + ys[r].x, o[r].y + ys[r].y};
                pend &= ~(1u << r);
            }
        }
        wave_sync();
    }
}

// Above-Nyquist residue, fast path.  What fft.js's in-place real radix-4 DIT leaves at positions 512..640 is the clean first half
// of the 256-point sub-DFT S2 of xw[4n+2] (its last stage never touches quarter 2, bundle:329-441), and the decimation identity
//     W^{2k} S2[k] = (X[k] - X[k+256] + X[k+512] - X[k+768]) / 4,   W = exp(-2 pi j / 1024),   X[1024 - i] = conj(X[i])
// gives it from the spectrum the frame already has -- the stash XS every frame writes --: four LDS reads and a twiddle per bin instead of re-running a
// 256-point FFT.  Valid while the last region ends at or below position 641 (f >= 0.75 always; lower f when the last peak sits low enough); the
// general path below covers the rest.  Sources b = 513 + l + 64 j, j < 2, all owned by the last peak (pv:133): b -> b + up_delta.
template <int R_>
__device__ __forceinline__ void residue_fast_1024(const float2 *__restrict__ tw32, unsigned wave_off, int l, int upper_end, int up_delta,
                                                  unsigned up_ridx, double *dbg_X, unsigned (&rt)[2], float2 (&ys)[2])
{
    constexpr int H = 513;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
    const unsigned char *XSb = smem_all + wave_off + OFF_XS;
    // k = 1 + l + 64 j, k + 256, 512 - k, 256 - k: the two members of each pair sit 32 slots (256 bytes) apart, j moves every address by +-8 slots
    const unsigned a0 = yslot_bytes((unsigned)(1 + l)), a2 = yslot_bytes((unsigned)(192 - 1 - l));        // slot(k) at j = 0; slot(256 - k) at j = 1
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int k = 1 + l + 64 * j, b = 512 + k, tgt = b + up_delta;        // k in [1, 128]
        const float2 x0 = *reinterpret_cast<const float2 *>(XSb + a0 + 64 * j), x1 = *reinterpret_cast<const float2 *>(XSb + a0 + 64 * j + 256);
        const float2 x3 = *reinterpret_cast<const float2 *>(XSb + a2 + 64 * (1 - j)), x2 = *reinterpret_cast<const float2 *>(XSb + a2 + 64 * (1 - j) + 256);
        const float2 tsum{0.25f * ((x0.x - x1.x) + (x2.x - x3.x)), 0.25f * ((x0.y - x1.y) - (x2.y - x3.y))};
        // W^{-2k} = conj(W_512^k) = conj(W_512^{k & 63}) * conj(W_8^{k >> 6}) from the LDS table (a global table load would sit, exposed, on the
        // critical path of every f < 1 frame)
        float2 w = tw1f_at(1, k & 63);
        const int k6 = k >> 6;                                             // 0, 1 or 2
        const float hh = 0.70710678118654752440f;
        if (k6 == 1) w = float2{(w.x - w.y) * hh, (w.x + w.y) * hh};      // * (h + j h)
        else if (k6 == 2) w = float2{-w.y, w.x};                          // * j
        const float2 s2 = cmul(tsum, w);
        rt[j] = (b < upper_end && tgt >= 0 && tgt < H) ? ((up_ridx << 16) | (unsigned)tgt) : NOROUTE;
        ys[j] = rotate_route<R_, 10>(rt[j], s2, tw32);
        if (dbg_X && b < upper_end) { dbg_X[2 * b] = s2.x; dbg_X[2 * b + 1] = s2.y; }
    }
}

// Rare path (f < 1 frames whose last region reads above Nyquist beyond position 640): rebuild what fft.js's in-place real DIT leaves at
// positions N/2+1..N-1 -- one quarter of the buffer at a time (quarter 2 = sub-FFT of x[4n+2], positions 512..767; quarter 3 = x[4n+3], 768..1023),
// by re-running the reference's stage structure (bundle:306-442,468-508) on that quarter in fp32 -- and add those sources into Y.
// Kept out of line so that its registers do not count against the main pipeline (3 waves per SIMD need <= 168 VGPRs).
// plain: the frame passed the pairwise test, nothing but the residue itself lands on the residue's targets (the continuation of the last region):
// stores instead of claim rounds.
template <int R_>
__device__ __attribute__((noinline)) PV_NO_DS_MERGE void residue_scatter_1024(const float *in, const float *hist, int hist_len, bool sys, long s0,
                                                               const float2 *__restrict__ tw32, unsigned wave_off, int l, int upper_end, int up_delta,
                                                               unsigned up_ridx, double *dbg_X, bool plain)
{
    constexpr int N = 1024, H = 513;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
    unsigned char *Yb = smem_all + wave_off + OFF_Y;
    unsigned short *CLAIM = reinterpret_cast<unsigned short *>(smem_all + wave_off + OFF_CLAIM);
    float2 *Q = reinterpret_cast<float2 *>(smem_all + wave_off + OFF_RESQ);
    // 0.5 * Hann of sample s = 2n + c, n = ln + 64 r, from the pair-interleaved shared table (entry [(r >> 1) * 64 + ln] = rows 2j, 2j+1)
    const float *HWf = reinterpret_cast<const float *>(smem_all + TAB_HANN);
    auto hw_at = [&](int smp) { const int n = smp >> 1, ln = n & 63, r = n >> 6; return HWf[4 * ((r >> 1) * 64 + ln) + 2 * (r & 1) + (smp & 1)]; };
    const WaveSrc src{in, hist, hist_len, sys};
    for (int base = N / 2; base < N && base < upper_end; base += N / 4) {
        {   // base stage: radix-4 blocks t = base/4 + l (bundle:468-508), input index = base-4 digit reversal of t
            const int t = base / 4 + l;
            const unsigned rv = __brev((unsigned)t) >> (32 - 8);
            const int off = (int)(((rv & 0x55555555u) << 1) | ((rv >> 1) & 0x55555555u));
            const float a = src.at(s0 + off) * (2.0f * hw_at(off));                     // Hann from the LDS table (2 * 0.5 w: exact)
            const float b = src.at(s0 + off + N / 4) * (2.0f * hw_at(off + N / 4));
            const float c = src.at(s0 + off + N / 2) * (2.0f * hw_at(off + N / 2));
            const float d = src.at(s0 + off + 3 * N / 4) * (2.0f * hw_at(off + 3 * N / 4));
            const float t0 = a + c, t1 = a - c, t2 = b + d, t3 = b - d;
            Q[4 * l] = float2{t0 + t2, 0.f};
            Q[4 * l + 1] = float2{t1, -t3};
            Q[4 * l + 2] = float2{t0 - t2, 0.f};
            Q[4 * l + 3] = float2{t1, t3};
        }
        wave_sync();
#pragma unroll
        for (int log2m = 4; log2m <= 8; log2m += 2) {                     // block sizes 16, 64, 256 inside the quarter (bundle:329-441)
            const int q = (1 << log2m) >> 2, hq = q >> 1;                 // butterflies i = 0..hq per block
            const int nblocks = 256 >> log2m;
            const int step = 256 >> log2m;                                 // W_Mb^{i m} = W_1024^{1024 i m / Mb} = W_512^{(i step)(2m)}: table row 2m
            const int it = l;
            if (it < nblocks * (hq + 1)) {
                int blk, i;
                if (it < nblocks * hq) { blk = it / hq; i = it - blk * hq; } else { blk = it - nblocks * hq; i = hq; }
                const int o = blk << log2m;
                const float2 A = Q[o + i];
                const float2 Bv = cmul(Q[o + q + i], cconj(tw1f_at(2, i * step)));
                const float2 C = cmul(Q[o + 2 * q + i], cconj(tw1f_at(4, i * step)));
                const float2 D = cmul(Q[o + 3 * q + i], cconj(tw1f_at(6, i * step)));
                const float2 T0 = cadd(A, C), T1 = csub(A, C), T2 = cadd(Bv, D), T3 = csub(Bv, D);
                Q[o + i] = cadd(T0, T2);
                Q[o + q + i] = float2{T1.x + T3.y, T1.y - T3.x};          // T1 - j T3
                if (i == 0) {
                    Q[o + 2 * q] = csub(T0, T2);                          // bundle:400-406
                } else if (i != hq) {                                     // bundle:409-440
                    Q[o + q - i] = float2{T1.x - T3.y, -(T1.y + T3.x)};   // conj(T1 + j T3)
                    Q[o + 2 * q - i] = float2{T0.x - T2.x, -(T0.y - T2.y)};
                }
            }
            wave_sync();
        }
        if (dbg_X)
            for (int i = l; i < N / 4; i += 64) if (base + i >= H) { dbg_X[2 * (base + i)] = Q[i].x; dbg_X[2 * (base + i) + 1] = Q[i].y; }
        // the sources of this quarter, all owned by the last peak (pv:133): b -> b + up_delta
        unsigned rt[4];
        float2 ys[4];
        int id[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int b = base + l + 64 * j, tgt = b + up_delta;
            rt[j] = (b >= H && b < upper_end && tgt >= 0 && tgt < H) ? ((up_ridx << 16) | (unsigned)tgt) : NOROUTE;
            ys[j] = rotate_route<R_, 10>(rt[j], Q[l + 64 * j], tw32);
            id[j] = b;
        }
        if (plain) {
#pragma unroll
            for (int j = 0; j < 4; j++) if (rt[j] != NOROUTE) *reinterpret_cast<float2 *>(Yb + yslot_bytes(rt[j] & 0xFFFFu)) = ys[j];
            wave_sync();
        } else {
            claim_rounds<4, false>(rt, ys, id, Yb, CLAIM);
        }
    }
}


// ---- fp64 flavour (PV_FP64_FLAVOUR): the colliding scatter and the above-Nyquist residue in doubles, on a plain array Y[513] ----
__device__ __forceinline__ double2 rotate_route_d(int r_, unsigned route, double2 v, const double2 *__restrict__ tw64)
{
    const unsigned ridx = (route >> 16) & 1023u;
    if (r_ == 4) { const unsigned q = ridx >> 8; return q == 0 ? v : q == 1 ? double2{-v.y, v.x} : q == 2 ? double2{-v.x, -v.y} : double2{v.y, -v.x}; }   // j^q exactly
    const double2 w = tw64[ridx];                                          // exp(-2 pi j ridx / N): v * conj(w), roundings spelled out (see rotate_route)
    return double2{__fma_rn(v.x, w.x, __dmul_rn(v.y, w.y)), __fma_rn(v.y, w.x, -__dmul_rn(v.x, w.y))};
}

template <int NS, bool YZERO>
__device__ __forceinline__ void claim_rounds_d(const unsigned (&rt)[NS], const double2 (&ys)[NS], const int (&id)[NS], double2 *Y, unsigned short *CLAIM)
{
    unsigned pend = 0, tg[NS];
#pragma unroll
    for (int r = 0; r < NS; r++) {
        const unsigned t = rt[r] & 0xFFFFu;
        const bool ok = t < 513u;
        pend |= ok ? (1u << r) : 0u;
        tg[r] = ok ? t : 0u;
    }
    bool first = YZERO;
    while (__any(pend != 0u)) {
#pragma unroll
        for (int r = 0; r < NS; r++) if (pend & (1u << r)) CLAIM[tg[r]] = (unsigned short)id[r];
        wave_sync();
        unsigned short c[NS];
#pragma unroll
        for (int r = 0; r < NS; r++) c[r] = CLAIM[tg[r]];
#pragma unroll
        for (int r = 0; r < NS; r++) {
            if ((pend & (1u << r)) && c[r] == (unsigned short)id[r]) {
                const double2 o = first ? double2{0.0, 0.0} : Y[tg[r]];
                Y[tg[r]] = double2{o.x + ys[r].x, o.y + ys[r].y};              // (0 + v: what the reference's += leaves in a zeroed bin)
                pend &= ~(1u << r);
            }
        }
        first = false;
        wave_sync();
    }
}

// residue_scatter_1024 in doubles: the reference's stage structure (bundle:306-442,468-508) re-run on one quarter at a time, twiddles from the
// forward FFT's fp64 table (row 2m of W_512^{l k} = W_1024^{4 m l}), the window product rounded to fp32 as the reference's Float32Array does (pv:55).
template <int R_>
__device__ __attribute__((noinline)) PV_NO_DS_MERGE void residue_scatter_1024_d(const float *in, const float *hist, int hist_len, long s0, const float *__restrict__ hann,
                                                                               const double2 *__restrict__ tw64, unsigned wave_off, int l, int upper_end, int up_delta,
                                                                               unsigned up_ridx, double *dbg_X)
{
    constexpr int N = 1024, H = 513;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
    double2 *Y = reinterpret_cast<double2 *>(smem_all + wave_off + OFF_Y);
    unsigned short *CLAIM = reinterpret_cast<unsigned short *>(smem_all + wave_off + F64_ROUTE);
    double2 *Q = reinterpret_cast<double2 *>(smem_all + wave_off + F64_Q);
    const double2 *TW1 = reinterpret_cast<const double2 *>(smem_all + TAB_TW1);
    const WaveSrc src{in, hist, hist_len, false};
    for (int base = N / 2; base < N && base < upper_end; base += N / 4) {
        {
            const int t = base / 4 + l;
            const unsigned rv = __brev((unsigned)t) >> (32 - 8);
            const int off = (int)(((rv & 0x55555555u) << 1) | ((rv >> 1) & 0x55555555u));
            const double a = (double)__fmul_rn(src.at(s0 + off), hann[off]), b = (double)__fmul_rn(src.at(s0 + off + N / 4), hann[off + N / 4]);
            const double c = (double)__fmul_rn(src.at(s0 + off + N / 2), hann[off + N / 2]), d = (double)__fmul_rn(src.at(s0 + off + 3 * N / 4), hann[off + 3 * N / 4]);
            const double t0 = a + c, t1 = a - c, t2 = b + d, t3 = b - d;
            Q[4 * l] = double2{t0 + t2, 0.0};
            Q[4 * l + 1] = double2{t1, -t3};
            Q[4 * l + 2] = double2{t0 - t2, 0.0};
            Q[4 * l + 3] = double2{t1, t3};
        }
        wave_sync();
#pragma unroll
        for (int log2m = 4; log2m <= 8; log2m += 2) {
            const int q = (1 << log2m) >> 2, hq = q >> 1;
            const int nblocks = 256 >> log2m, step = 256 >> log2m;
            if (l < nblocks * (hq + 1)) {
                int blk, i;
                if (l < nblocks * hq) { blk = l / hq; i = l - blk * hq; } else { blk = l - nblocks * hq; i = hq; }
                const int o = blk << log2m;
                const double2 A = Q[o + i];
                const double2 Bv = cmul(Q[o + q + i], TW1[2 * 64 + i * step]);
                const double2 C = cmul(Q[o + 2 * q + i], TW1[4 * 64 + i * step]);
                const double2 D = cmul(Q[o + 3 * q + i], TW1[6 * 64 + i * step]);
                const double2 T0 = cadd(A, C), T1 = csub(A, C), T2 = cadd(Bv, D), T3 = csub(Bv, D);
                Q[o + i] = cadd(T0, T2);
                Q[o + q + i] = double2{T1.x + T3.y, T1.y - T3.x};
                if (i == 0) {
                    Q[o + 2 * q] = csub(T0, T2);
                } else if (i != hq) {
                    Q[o + q - i] = double2{T1.x - T3.y, -(T1.y + T3.x)};
                    Q[o + 2 * q - i] = double2{T0.x - T2.x, -(T0.y - T2.y)};
                }
            }
            wave_sync();
        }
        if (dbg_X)
            for (int i = l; i < N / 4; i += 64) if (base + i >= H) { dbg_X[2 * (base + i)] = Q[i].x; dbg_X[2 * (base + i) + 1] = Q[i].y; }
        unsigned rt[4];
        double2 ys[4];
        int id[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int b = base + l + 64 * j, tgt = b + up_delta;
            rt[j] = (b >= H && b < upper_end && tgt >= 0 && tgt < H) ? ((up_ridx << 16) | (unsigned)tgt) : NOROUTE;
            ys[j] = rotate_route_d(R_, rt[j], Q[l + 64 * j], tw64);
            id[j] = b;
        }
        claim_rounds_d<4, false>(rt, ys, id, Y, CLAIM);
    }
}

// Shift table DSH[p] = Math.round(p * f) - p (pv:125,147) for every candidate peak bin, DROP where the reference skips the peak (pv:127-129).
// Runs once per chain when f is constant; out of line so that its fp64 temporaries stay out of the main loop's register budget.
__device__ __attribute__((noinline)) void build_shift_table_1024(float f, unsigned wave_off, int l)
{
    constexpr int N = 1024, H = 513, DROP = 0x4000;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
    short *DSH = reinterpret_cast<short *>(smem_all + wave_off + OFF_PSH);
    const double pf = (double)f;
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const int pk = l + 64 * r;
        const double ps = floor((double)pk * pf + 0.5);                 // x + 0.5 is exact here (<= 37 significant bits)
        const bool ok = (ps <= (double)H) && (ps >= -(double)(2 * N));   // pv:127-129; NaN -> not ok
        DSH[pk] = ok ? (short)((int)ps - pk) : (short)DROP;             // DROP pushes every target of the region out of range
    }
}

// f < 1, frames that fail the pairwise test (f below ~0.65 with dense peaks): the colliding scatter with claim rounds, out of line.  It takes
// its nine routes (bins 8 l + i, and bin 512 in lane 63's ninth) as arguments and its sources from the stash the frame wrote anyway.
struct Routes9 { unsigned r[9]; };
template <int R_>
__device__ __attribute__((noinline)) PV_NO_DS_MERGE void scatter_claims_1024(Routes9 RT, unsigned wave_off, int l, int tmod, int last_peak, int upper_end,
                                                                            const float *in, const float *hist, int hist_len, bool sys, long s0,
                                                                            const float2 *__restrict__ tw32, double *dbg_X)
{
    constexpr int N = 1024, H = 513;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
    unsigned char *smem = smem_all + wave_off;
    unsigned char *Yb = smem + OFF_Y;
    unsigned short *CLAIM = reinterpret_cast<unsigned short *>(smem + OFF_CLAIM);
    const unsigned char *XSb = smem + OFF_XS;
    const short *DSH = reinterpret_cast<const short *>(smem + OFF_PSH);
    unsigned rt[9];
    float2 ys[9];
    int id[9];
#pragma unroll
    for (int i = 0; i < 9; i++) rt[i] = RT.r[i];
#pragma unroll
    for (int i = 0; i < 8; i++) { id[i] = 8 * l + i; ys[i] = *reinterpret_cast<const float2 *>(XSb + 8 * l + 544 * i); }
    id[8] = 512;
    ys[8] = *reinterpret_cast<const float2 *>(XSb + 512);                // slot(512) = 64: every lane reads it, only lane 63's ninth route is valid
#pragma unroll
    for (int i = 0; i < 9; i++) ys[i] = rotate_route<R_, 10>(rt[i], ys[i], tw32);
    const bool need_res = upper_end > H;
    const bool fast_res = need_res && (upper_end <= H + 128);
    // sources above Nyquist, all owned by the last peak (pv:133)
    const int up_delta = need_res ? (int)DSH[last_peak < 0 ? 0 : last_peak] : 0;
    const unsigned up_ridx = (unsigned)((up_delta & (N - 1)) * tmod) & (N - 1);
    // fill the claim words of this frame's targets?  No: a claim word is only ever compared with the ids posted in the SAME round, stale words are harmless.
    claim_rounds<9, true>(rt, ys, id, Yb, CLAIM);
    if (fast_res) {
        // The residue sources b = 513 .. upper_end - 1 land on the targets 513 + delta .. upper_end - 1 + delta, above every target of the last
        // region's ordinary sources (b <= 512).  Unless a source of an EARLIER region reaches up there too (only when the last region is shorter
        // than the overlap of its neighbour), nothing else touches those bins: they are still zero and the residue is stored, no claim rounds.
        bool clash = false;
#pragma unroll
        for (int r = 0; r < 9; r++) { const unsigned t = rt[r] & 0xFFFFu; clash |= (t < 513u) && ((int)t > 512 + up_delta); }
        const bool plain_res = !__any(clash);
        unsigned rt2[2];
        float2 ys2[2];
        residue_fast_1024<R_>(tw32, wave_off, l, upper_end, up_delta, up_ridx, dbg_X, rt2, ys2);
        if (plain_res) {
#pragma unroll
            for (int j = 0; j < 2; j++) if ((rt2[j] & 0xFFFFu) < 513u) *reinterpret_cast<float2 *>(Yb + yslot_bytes(rt2[j] & 0xFFFFu)) = float2{0.f + ys2[j].x, 0.f + ys2[j].y};
        } else {
            const int id2[2] = {513 + l, 513 + 64 + l};
            claim_rounds<2, false>(rt2, ys2, id2, Yb, CLAIM);
        }
    } else if (need_res) {
        residue_scatter_1024<R_>(in, hist, hist_len, sys, s0, tw32, wave_off, l, upper_end, up_delta, up_ridx, dbg_X, false);
    }
}

#ifndef PV_F32_REGT1
#define PV_F32_REGT1 true       // transpose 1 of the fp32 forward FFT in registers (false: through LDS; A/B builds)
#endif
// ---- the two forward transforms of one frame: one text for the kernel's inline code and for the out-of-line copy of the F32 instances' rare paths (forward_cold_1024) ----
// Reference width: Hann (pv:55), pack (the factor 1/2 of the split pass is folded into the table: exact), 512-point complex FFT in fp64, split pass in conjugate
// pairs: with E = Z[k] + conj(Z[512-k]), O = Z[k] - conj(Z[512-k]) (Z pre-halved), X[k] = E - j W^k O and X[512-k] = conj(E + j W^k O).  Lane l owns the pairs
// k = l + 64 r, r < 4 (partner value from lane 64-l, register 7-r): emit(r, X[l + 64 r], X[512 - l - 64 r]); lane 0 also the self-paired bin: emit256(X[256]).
template <typename EMIT, typename EMIT256, typename ST = NoStamp>
__device__ __forceinline__ void spectrum64_1024(const float2 (&raw)[8], const pk::c32 (&hw)[8], unsigned char *smem, int l, double2 wl, EMIT emit, EMIT256 emit256, ST st = ST{})
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
    const double2 *TW1 = reinterpret_cast<const double2 *>(smem_all + TAB_TW1);
    const double2 *TW2 = reinterpret_cast<const double2 *>(smem_all + TAB_TW2);
    double2 *S64 = reinterpret_cast<double2 *>(smem);
    pv_prio(PH_FA);
    double2 z[8];
#pragma unroll
    for (int r = 0; r < 8; r++) { const v2f xw = v2f{raw[r].x, raw[r].y} * hw[r]; z[r] = double2{(double)xw.x, (double)xw.y}; }
    fft512_wave<double, false>(z, S64, TW1, TW2, l, st);
    pv_prio(PH_SPLITX);
    // partner values through the (now free) transpose scratch: rows 4..7 written lane-contiguous, read back reversed.  Element 512 - k of the pair k = l + 64 r sits
    // at (3 - r) * 64 + (64 - l) for every lane (lane 0: 64 (8 - r), its own register 8 - r); half the LDS cycles of sixteen bpermutes.  (l = 0, r = 0) reads one
    // element past the rows: replaced below.
#pragma unroll
    for (int r = 4; r < 8; r++) S64[(r - 4) * 64 + l] = z[r];
    wave_sync();
    pv_prio(PH_SPLITM);
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const double2 zm = S64[(3 - r) * 64 + 64 - l];
        const double2 E{z[r].x + zm.x, z[r].y - zm.y};
        const double2 O{z[r].x - zm.x, z[r].y + zm.y};
        const double2 WO = cmul(wl, mul_w16<double, false>(O, r));         // W_1024^{l+64r} = W^l * W_16^r
        double2 xa{E.x + WO.y, E.y - WO.x};
        double2 xb{E.x - WO.y, -(E.y + WO.x)};
        if (r == 0 && l == 0) {
            // Z[0] is pre-halved: X[0] = 2(zr + zi), X[512] = 2(zr - zi), both real (bundle:447-508 keep Im = 0)
            xa = double2{2.0 * (z[0].x + z[0].y), 0.0};
            xb = double2{2.0 * (z[0].x - z[0].y), 0.0};
        }
        emit(r, xa, xb);
    }
    if (l == 0) emit256(double2{2.0 * z[4].x, -2.0 * z[4].y});             // k = 256 pairs with itself: W^256 = -j, X = 2 conj(Z)
}

// fp32 first (round 5): the same in PACKED fp32, half the issue cycles.  The forward FFT is the inverse instance on conjugated data, FFT(z) = conj(IFFT(conj z)), the
// first conjugation folded into the window product, the second into the split pass: with Zc = conj(Z), E' = Zc[k] + conj(Zc[512-k]), O' = Zc[k] - conj(Zc[512-k]),
// T = conj(W^k) O':  X[k] = conj(E' + j T), X[512-k] = E' - j T.  wlfs = SC conj(W^l) is the c2r twiddle of the kernel (SC a power of two), isc = 1 / SC takes the
// scale out again inside the FMAs: exact.
template <bool REGT1, typename EMIT, typename EMIT256, typename ST = NoStamp>
__device__ __forceinline__ void spectrum32_1024(const float2 (&raw)[8], const pk::c32 (&hw)[8], unsigned char *smem, int l, pk::c32 wlfs, float isc_, EMIT emit, EMIT256 emit256, ST st = ST{})
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
    const v4f *TW1F4 = reinterpret_cast<const v4f *>(smem_all + TAB_TW1F);
    const v4f *TW2F4 = reinterpret_cast<const v4f *>(smem_all + TAB_TW2F);
    pv_prio(PH_FA);
    pk::c32 zc[8];
#pragma unroll
    for (int r = 0; r < 8; r++) zc[r] = pk::mul_conj(pk::c32{raw[r].x, raw[r].y}, hw[r]);       // conj(z), z pre-halved like the fp64 form's
    fft512_wave_inv_pk<REGT1, ST, true>(zc, reinterpret_cast<pk::c32 *>(smem), TW1F4, TW2F4, l, st);
    pv_prio(PH_SPLITX);
    // partner rows 4..7, read back reversed.  (l = 0, r = 0) reads one element past the rows: Z[512] = Z[0] is put there, and the pair k = 0 goes through the same
    // arithmetic as the others (E' = 2 Re, O' = 2 j Im, W^0 = 1: X[0] = 2 (Re Z0 + Im Z0), X[512] = 2 (Re Z0 - Im Z0), with the roundings of the closed form)
    pk::c32 *XCH = reinterpret_cast<pk::c32 *>(smem);
#pragma unroll
    for (int r = 4; r < 8; r++) XCH[(r - 4) * 64 + l] = zc[r];
    if (l == 0) XCH[256] = zc[0];
    wave_sync();
    pv_prio(PH_SPLITM);
    const pk::c32 isc{isc_, isc_};
    pk::c32 zm[4];                                                          // all four partner values first: read one by one, each read queues behind the magnitude
                                                                            // stores of the pair before it (the compiler cannot tell the arrays apart)
#pragma unroll
    for (int r = 0; r < 4; r++) zm[r] = XCH[(3 - r) * 64 + 64 - l];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const pk::c32 E = pk::add_conj(zc[r], zm[r]), O = pk::sub_conj(zc[r], zm[r]);
        const pk::c32 T = pk::cmul(mul_w16_inv_pk(O, r), wlfs);             // SC conj(W^{l + 64 r}) O'
        const pk::c32 xa = pk::conj_fma_j(T, isc, E), xb = pk::fnma_j(T, isc, E);
        emit(r, xa, xb);
    }
    if (l == 0) emit256(pk::c32{2.0f * zc[4].x, 2.0f * zc[4].y});           // X[256] = 2 conj(Z[256]) = 2 Zc[256]
}

// The forward transforms of the F32 instances' RARE paths, out of line: a frame of class B (fp64), and the fp32 attempt of a chain that runs the fp64 transform first.
// Inline they quadruple what meets at the merge in front of the peak search, and the register allocator then parks values of the HOT path in scratch memory.
// Results travel through LDS: |X|^2 -> MAG; the fp32 spectrum -> the transposed stash (!SPREAD_: where the inline code puts it) or, SPREAD_, lane-private slots
// XV[r * 64 + l] (XA), XV[(4 + r) * 64 + l] (XB), XV[512] (X[256], lane 0) in the stash area that instance does not use.  Returns the fp32 transform's K (wide: 0).
struct Raw8 { float2 v[8]; };
template <bool SPREAD_>
__device__ __attribute__((noinline)) PV_NO_DS_MERGE float forward_cold_1024(Raw8 RW, int wide, double2 wl, pk::c32 wlfs, float isc, unsigned wave_off, int l)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
    unsigned char *smem = smem_all + wave_off;
    float *MAG = reinterpret_cast<float *>(smem + OFF_MAG);
    unsigned char *XSb = smem + OFF_XS;
    float2 *XV = reinterpret_cast<float2 *>(XSb);
    const v4f *HW4 = reinterpret_cast<const v4f *>(smem_all + TAB_HANN);
    pk::c32 hw[8];
#pragma unroll
    for (int j = 0; j < 4; j++) { const v4f h = HW4[j * 64 + l]; hw[2 * j] = pk::c32{h.x, h.y}; hw[2 * j + 1] = pk::c32{h.z, h.w}; }
    unsigned ystr = 0, ystr_m = 0;
    if constexpr (!SPREAD_) { ystr = yslot_bytes((unsigned)l); ystr_m = yslot_bytes((unsigned)(512 - 192 - l)); }
    auto put = [&](int r, float2 a, float2 b) {
        if constexpr (SPREAD_) { XV[r * 64 + l] = a; XV[(4 + r) * 64 + l] = b; }
        else { *reinterpret_cast<float2 *>(XSb + ystr + 64 * r) = a; *reinterpret_cast<float2 *>(XSb + ystr_m + 64 * (3 - r)) = b; }
    };
    auto put256 = [&](float2 a) { if constexpr (SPREAD_) XV[512] = a; else *reinterpret_cast<float2 *>(XSb + 256) = a; };
    float guard_k = 0.f;
    if (wide) {
        spectrum64_1024(RW.v, hw, smem, l, wl,
                        [&](int r, double2 xa, double2 xb) {
                            MAG[4 + l + 64 * r] = (float)(xa.x * xa.x + xa.y * xa.y);
                            MAG[4 + 512 - l - 64 * r] = (float)(xb.x * xb.x + xb.y * xb.y);
                            put(r, float2{(float)xa.x, (float)xa.y}, float2{(float)xb.x, (float)xb.y});
                        },
                        [&](double2 x256) { MAG[4 + 256] = (float)(x256.x * x256.x + x256.y * x256.y); put256(float2{(float)x256.x, (float)x256.y}); });
    } else {
        unsigned mmax = 0u;
        spectrum32_1024<PV_F32_REGT1>(RW.v, hw, smem, l, wlfs, isc,
                                      [&](int r, pk::c32 xa, pk::c32 xb) {
                                          const float ma = mag32(xa), mb = mag32(xb);
                                          MAG[4 + l + 64 * r] = ma;
                                          MAG[4 + 512 - l - 64 * r] = mb;
                                          mmax = max(max(mmax, __float_as_uint(ma)), __float_as_uint(mb));
                                          put(r, float2{xa.x, xa.y}, float2{xb.x, xb.y});
                                      },
                                      [&](pk::c32 x256) { const float m = mag32(x256); MAG[4 + 256] = m; mmax = max(mmax, __float_as_uint(m)); put256(float2{x256.x, x256.y}); });
        guard_k = guard_k_of(wave_max_u32(mmax));
    }
    return guard_k;
}

// S_ROWS = hop / 128 (rows of 128 samples a frame advances by): 1, 2, 4 or 8
// AUX = true: test-tap instance (pv_debug_frame); the production instance carries no tap code.
// RESIDENT = true: streaming instance that stays on the GPU (PV_FLAG_PERSISTENT_STREAM): after its quantum a wave polls the control block in pinned host memory
// for the next sequence number instead of ending (pv_capi.hip: persist_*); a quantum then costs no launch.  One wave per channel slot, 1 hop per quantum.
// SPREAD = true: the instance for frame chains whose pitchFactor is >= 1 on EVERY frame (chosen per chain: pv_classify_chains below / the host for a
// streaming quantum).  Its regions only spread, every target has one source: the spectrum stays in the registers the split pass leaves it in (lane l
// <-> bins l + 64 r, 512 - l - 64 r), the routes travel from the lanes that compute them (8 consecutive bins each) through a table in LDS, Y is a
// plain array, and none of the f < 1 machinery exists in the instance.  SPREAD = false handles every pitchFactor; there the SPECTRUM travels instead:
// rounded to fp32 into a transposed stash, read back by the lanes that compute the routes, which then scatter their own 8 bins into a transposed Y --
// no route table, and the stash is what the above-Nyquist residue is computed from anyway: an f < 1 frame makes three dependent LDS round trips in
// its scatter where round 3 made five.  (One instance that chooses per frame was tried first: the register allocator does not keep the two flows
// apart, at 168 VGPRs the f >= 1 flow then reloads its source bins from scratch memory.)
template <int S_ROWS, bool AUX, bool RESIDENT = false, bool SPREAD = false, bool F32 = false>
__global__ __launch_bounds__(64 * (RESIDENT ? RES_WAVES : WAVES), RESIDENT ? 1 : PV_WAVES_PER_SIMD) PV_NO_DS_MERGE void pv_wave_kernel_1024(const PvKernelParams p)
{
    static_assert(!F32 || (!FP64 && !AUX), "the fp32-first forward transform is a product path: neither the reference-width flavour nor the tap instance");
    constexpr int WGW = RESIDENT ? RES_WAVES : WAVES;                    // waves per workgroup of this instance
    constexpr int N = 1024, M = 512, H = 513;
    constexpr int HOP = 128 * S_ROWS, R = N / HOP, LROWS = 8 - S_ROWS;    // LROWS rows of carried accumulator
    const int l = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // chains are numbered channel-major over (channel, chunk) and packed 12 to a workgroup regardless of the channel they belong to: many short
    // streams (one chunk per channel) fill the workgroups exactly like one long stream does
    // batch launches: the chains of this instance's class come from the list pv_classify_chains compacted (p.chain_list: [0, nchains) the SPREAD class,
    // [nchains, 2 nchains) the other; p.chain_count[class] entries); a streaming quantum / the tap instance numbers its chains directly
    long chain = (long)blockIdx.x * WGW + wv;
#ifndef PV_CHAIN_BLOCKED
    if (!RESIDENT) chain = (long)wv * gridDim.x + blockIdx.x;              // (see below; with or without a list)
#endif
    bool listed = true;
    if (!RESIDENT && p.chain_list) {
        // The waves of a workgroup take list entries a whole grid apart (wave w of workgroup b: entry w * gridDim.x + b), not twelve neighbours.  Neighbouring
        // chains are neighbouring stretches of a stream and cost alike -- class-B frames come in clusters --, and a launch ends with its slowest chain: next to
        // average chains on its SIMD a slow one inherits their issue slots when they finish; next to its equally slow neighbours it does not
        // (profiles/r05_chain_times.md).
        const long total = (long)p.nch * p.nchunks;
        listed = chain < (long)p.chain_count[SPREAD ? 0 : 1];
        chain = listed ? (long)p.chain_list[(SPREAD ? 0 : total) + chain] : total;
    }
    const int ch = (int)(chain / p.nchunks), chunk = (int)(chain - (long)ch * p.nchunks);

#ifndef PV_CHAIN_BLOCKED
    if (!RESIDENT && p.chain_list && (long)blockIdx.x >= (long)p.chain_count[SPREAD ? 0 : 1]) return;           // no chain of this class left for the workgroup (uniform): not even the tables
#else
    if (!RESIDENT && p.chain_list && (long)blockIdx.x * WGW >= (long)p.chain_count[SPREAD ? 0 : 1]) return;     // no chain of this class left for the workgroup (uniform): not even the tables
#endif
    // ---- LDS carve (all dynamic, 16-byte aligned): shared tables, then one private region per wave ----
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
    const v4f *TW1F4 = reinterpret_cast<const v4f *>(smem_all + TAB_TW1F);
    const v4f *TW2F4 = reinterpret_cast<const v4f *>(smem_all + TAB_TW2F);
    const v4f *HW4 = reinterpret_cast<const v4f *>(smem_all + TAB_HANN);
    {
        double2 *t1 = reinterpret_cast<double2 *>(smem_all + TAB_TW1);
        double2 *t2 = reinterpret_cast<double2 *>(smem_all + TAB_TW2);
        float2 *t1f = reinterpret_cast<float2 *>(smem_all + TAB_TW1F);     // pair-interleaved: [((k >> 1) * 64 + l) * 2 + (k & 1)]
        float2 *t2f = reinterpret_cast<float2 *>(smem_all + TAB_TW2F);     // [((k >> 1) * 8 + n0) * 2 + (k & 1)]
        float2 *hh = reinterpret_cast<float2 *>(smem_all + TAB_HANN);      // [((r >> 1) * 64 + l) * 2 + (r & 1)]
        for (int i = threadIdx.x; i < 512; i += 64 * WGW) {
            const int k = i >> 6, ln = i & 63;
            const double2 w = p.tw64[(2 * ln * k) & (N - 1)];
            t1[i] = w;
            if (FP64) reinterpret_cast<double2 *>(smem_all + TAB_TW1C)[i] = double2{w.x, -w.y};
            t1f[2 * ((k >> 1) * 64 + ln) + (k & 1)] = float2{(float)w.x, -(float)w.y};
            hh[2 * ((k >> 1) * 64 + ln) + (k & 1)] = float2{0.5f * p.hann[2 * i], 0.5f * p.hann[2 * i + 1]};   // n = i = ln + 64 k: row k
            if (i < 64) {
                const int k2 = i >> 3, n0 = i & 7;
                const double2 w2 = p.tw64[(16 * n0 * k2) & (N - 1)];
                t2[i] = w2;
                if (FP64) reinterpret_cast<double2 *>(smem_all + TAB_TW2C)[i] = double2{w2.x, -w2.y};
                t2f[2 * ((k2 >> 1) * 8 + n0) + (k2 & 1)] = float2{(float)w2.x, -(float)w2.y};
            }
        }
    }
    __syncthreads();                                                     // the only workgroup-wide barrier
    if (!listed || ch >= p.nch) return;
    // what changes from quantum to quantum in the resident form (constants of the launch otherwise)
    const float *hist_in = p.hist_in, *acc_in = p.acc_in;
    float *hist_out = p.hist_out, *acc_out = p.acc_out;
    int t0_mod_n = p.t0_mod_n;
    unsigned done_seq = p.done_seq;
    unsigned last_seq = p.done_seq;                                      // resident: the last quantum completed before this launch
    [[maybe_unused]] unsigned pred = 0;                                  // F32: 0 .. PRED_MAX, >= PRED_WIDE: this chain runs the fp64 transform first (pv_guard.h).  Declared in front
                                                                         // of the resident form's loop: a resident wave's quanta are ONE chain of a stream
resident_top:
    if (RESIDENT) {
        // ctl[0] carries the whole quantum in ONE word -- sequence number (low 16 bits, never 0), channel count (7 bits), ping-pong half (1 bit),
        // timeCursor / hop mod R (8 bits) -- so that a successful poll needs no second round trip over PCIe before the input can be requested
        unsigned word;
        const unsigned long long idle0 = wall_clock64();
        for (;;) {
            word = __hip_atomic_load(p.ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if ((word & 0xFFFFu) != (last_seq & 0xFFFFu)) break;
            // leave when asked to, or after ~50 ms without work (the host relaunches on demand: a resident wave must never outlive its user)
            if (__hip_atomic_load(p.ctl + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u || wall_clock64() - idle0 > (unsigned long long)p.idle_ticks) return;
            __builtin_amdgcn_s_sleep(2);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");                    // system scope: what the host wrote before the word
        const unsigned seq = word & 0xFFFFu, nch_now = (word >> 16) & 0x7Fu, cur = (word >> 23) & 1u;
        t0_mod_n = (int)(((word >> 24) & 0xFFu) * HOP) & (N - 1);
        hist_in = p.hist2[cur]; hist_out = p.hist2[cur ^ 1u];
        acc_in = p.acc2[cur]; acc_out = p.acc2[cur ^ 1u];
        done_seq = last_seq = seq;
        if ((unsigned)ch >= nch_now) {
            // this channel slot is not part of the quantum: if it holds state (ctl[5] = slots in use), carry it across the ping-pong flip
            if ((unsigned)ch < __hip_atomic_load(p.ctl + 5, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM))
                for (int j = l; j < N - HOP; j += 64) {
                    hist_out[(long)ch * (N - HOP) + j] = hist_in[(long)ch * (N - HOP) + j];
                    acc_out[(long)ch * (N - HOP) + j] = acc_in[(long)ch * (N - HOP) + j];
                }
            goto resident_top;
        }
    }

    const unsigned wave_off = TAB_BYTES + wv * WAVE_LDS;
    unsigned char *smem = smem_all + wave_off;
    double2 *S64 = reinterpret_cast<double2 *>(smem);                    // 8*72*16 = 9216 B: fp64 transposes
    float2 *S32 = reinterpret_cast<float2 *>(smem);                      // fp32 transposes (first 4608 B)
    unsigned char *Yb = smem + OFF_Y;                                    // shifted spectrum Y[0..512], transposed (yslot_bytes), between the FFTs
    unsigned char *XSb = smem + OFF_XS;                                  // fp32 copy of the spectrum, transposed: split pass -> scatter
    float *MAG = reinterpret_cast<float *>(smem + OFF_MAG);              // MAG[4 + k], k in [-4, 516): |X|^2 exchange
    short *DSH = reinterpret_cast<short *>(smem + OFF_PSH);              // shift Math.round(p * f) - p per candidate peak bin p (DROP: peak dropped)
    unsigned psh_key = 0u;                                               // bit pattern of the f the table was built for
    bool psh_valid = false;                                              // ... once one has been built (any bit pattern, NaNs included, is a legal f)

    const int first_out = chunk * p.frames_per_chunk;
    int last_out = first_out + p.frames_per_chunk;
    if (last_out > p.nhops) last_out = p.nhops;
    int first_frame = first_out - (R - 1);
    const bool from_state = (first_frame <= 0);
    if (from_state) first_frame = 0;

    const long cbase = (long)ch * p.ch_stride;
    const WaveSrc src{p.in + cbase, hist_in + (long)ch * (N - HOP), N - HOP, RESIDENT && p.in_cached != 0};
    float *outp = p.out + cbase;
    const bool vec_out = (reinterpret_cast<uintptr_t>(outp) & 7u) == 0;           // 8-byte aligned channel base: float2 stores
    const bool vec_in = ((reinterpret_cast<uintptr_t>(src.in) | reinterpret_cast<uintptr_t>(src.hist)) & 7u) == 0;
    const float *pitch_row = p.pitch + (p.pitch_stride ? (long)(ch / p.ch_per_stream) * p.pitch_stride : 0);

    const double2 wl = p.tw64[l];          // split pass: W_1024^{l + 64 r} = wl * W_16^r (W_16^r is wave-uniform)
    const float2 wlf = cconj(p.tw32[l]);
    // c2r scale: the 1/N of the inverse (bundle:110-111) times 2/R -- the 1/R of the overlap-add (ola:149-157) and the factor 2 that turns the
    // shared 0.5 * Hann table into the synthesis window (pv:67).  All powers of two: every product below is the reference's, bit for bit.
    constexpr float SC = 2.0f / ((float)N * (float)R);
    const pk::c32 wlfs{wlf.x * SC, wlf.y * SC};                           // c2r twiddle with the scale folded in (exact)

    // ---- carried overlap-add accumulator in registers: row r <-> samples 2l + 128 r (+1) ----
    float2 acc[8];
#pragma unroll
    for (int r = 0; r < 8; r++) acc[r] = float2{0.f, 0.f};
    if (from_state) {
#pragma unroll
        for (int r = 0; r < LROWS; r++) {
            const float *a = acc_in + (long)ch * (N - HOP) + 2 * l + 128 * r;
            acc[r] = float2{a[0], a[1]};
        }
    }

    // ---- raw input window in registers: 8 rows of (2 samples x 64 lanes); a frame advances by S_ROWS rows, so only those rows are
    //      loaded per frame (prefetched one frame ahead) and every input sample is fetched from HBM once ----
    auto load_rows = [&](float2 *w, int nrows, int first_row, int frame) {
        const long s0 = (long)(frame + 1) * HOP - N + 2 * l;
#pragma unroll
        for (int r = 0; r < nrows; r++) {
            const long sx = s0 + 128 * (first_row + r);
            if (RESIDENT && src.sys && vec_in && sx >= 0) {                // the host's hop of this quantum: never from a cache
                const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(src.in + sx), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                w[r] = float2{__uint_as_float((unsigned)v), __uint_as_float((unsigned)(v >> 32))};
            } else if (vec_in) w[r] = *reinterpret_cast<const float2 *>(sx < 0 ? src.hist + sx + src.hist_len : src.in + sx);
            else w[r] = float2{src.at(sx), src.at(sx + 1)};
        }
    };
    float2 raw[8];
    load_rows(raw, 8, 0, first_frame);
    // Every global access of a frame is issued one frame ahead, in ONE place (after the split pass), and the stores of a frame are
    // exec-masked straight-line code (emit_v is a per-lane value on purpose): the only vector-memory wait of the loop is then an exact
    // s_waitcnt vmcnt(#stores) at the top of the next frame.  A load in mid-frame (the pitch row used to be read where it was needed) or a
    // wave-uniform branch around the stores makes the compiler wait with vmcnt(0) -- for the prefetch it has just issued and for the
    // acknowledgement of the stores -- and every frame of every wave then sits out two exposed HBM round trips.
    float pf_next = (RESIDENT && src.sys) ? __hip_atomic_load(pitch_row + first_frame, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : pitch_row[first_frame];
    int emit_v = first_out;
    asm volatile("" : "+v"(emit_v));                                     // opaque VGPR copy of first_out: keeps the store predicate divergent
    // 0.5 * Hann rows of this lane: read at the END of a frame (4 ds_read_b128) for the synthesis window and kept across the loop edge for
    // the analysis window of the next frame -- one table read per frame instead of two
    pk::c32 hw[8];
#pragma unroll
    for (int j = 0; j < 4; j++) { const v4f h = HW4[j * 64 + l]; hw[2 * j] = pk::c32{h.x, h.y}; hw[2 * j + 1] = pk::c32{h.z, h.w}; }

#ifdef PV_STAMPS
    Stamps stamps;
    stamps.start();
    const unsigned stamp_t0 = stamps.prev;
#endif
    [[maybe_unused]] unsigned n_fallback = 0;                            // F32: frames of this chain of class B (forward transform in fp64)
#ifdef PV_FLIP_COUNT
    unsigned n_flip = 0, n_uncaught = 0, n_sure = 0, n_incons = 0;
#endif
    for (int m = first_frame; m < last_out; ++m) {
        const float pfm = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(pf_next)));   // k-rate pitchFactor (pv:47), wave-uniform
        const double pf = (double)pfm;
        const int tmod = (int)(((long)t0_mod_n + (long)m * HOP) & (N - 1));
        const bool dbg = AUX && (p.dbg_mag != nullptr) && ch == p.dbg_ch && m == p.dbg_frame;

        float2 XA[4], XB[4], x256f{0.f, 0.f};                              // fp32 copy of the spectrum: the only thing the shift needs after the decisions
        [[maybe_unused]] double2 XAd[4], XBd[4], x256d{0.0, 0.0};         // (fp64 flavour: the spectrum itself)
        // ---- forward transform at the reference's width: Hann, pack, 512-point complex FFT in fp64, split pass, |X|^2 -> MAG, the spectrum rounded to fp32 -> XA / XB
        //      (and the transposed stash).  The only forward transform of the !F32 instances; what an F32 instance falls back to when a decision is in doubt ----
        auto forward64 = [&]() {
#ifdef PV_STAMPS
            auto st64 = [&](int id) { if (id & 1) stamps.mark(id); else stamps.mark(id, true); };
#else
            NoStamp st64;
#endif
            spectrum64_1024(raw, hw, smem, l, wl,
                            [&](int r, double2 xa, double2 xb) {
                                // ---- |X|^2 -> f32 (pv:82-92) for the neighbour tests, and the spectrum itself, rounded to fp32 (the only thing the shift needs after
                                //      the decisions) ----
                                MAG[4 + l + 64 * r] = (float)(xa.x * xa.x + xa.y * xa.y);
                                MAG[4 + 512 - l - 64 * r] = (float)(xb.x * xb.x + xb.y * xb.y);
                                XA[r] = float2{(float)xa.x, (float)xa.y};
                                XB[r] = float2{(float)xb.x, (float)xb.y};
                                if constexpr (FP64) { XAd[r] = xa; XBd[r] = xb; }
                                if (dbg) {
                                    const int ka = l + 64 * r, kb = 512 - ka;
                                    p.dbg_X[2 * ka] = xa.x; p.dbg_X[2 * ka + 1] = xa.y;
                                    p.dbg_X[2 * kb] = xb.x; p.dbg_X[2 * kb + 1] = xb.y;
                                }
                            },
                            [&](double2 x256) {
                                MAG[4 + 256] = (float)(x256.x * x256.x + x256.y * x256.y);
                                x256f = float2{(float)x256.x, (float)x256.y};
                                if constexpr (FP64) x256d = x256;
                                if (dbg) { p.dbg_X[2 * 256] = x256.x; p.dbg_X[2 * 256 + 1] = x256.y; }
                            }, st64);
            if constexpr (!SPREAD) {
                // into the transposed stash: the lanes that take the decisions (8 consecutive bins each) also move the bins.  Strided-order addresses: bin l + 64 r at
                // ystr + 64 r, bin 512 - l - 64 r at ystr_m + 64 (3 - r).  Formed HERE from an opaque copy of the lane id (four instructions per frame): as loop
                // invariants they would live in registers across the forward FFT, the register peak of the kernel, and push other addresses into scratch
                unsigned ystr, ystr_m;
                { int lq = l; asm volatile("" : "+v"(lq)); ystr = yslot_bytes((unsigned)lq); ystr_m = yslot_bytes((unsigned)(512 - 192 - lq)); }
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    *reinterpret_cast<float2 *>(XSb + ystr + 64 * r) = XA[r];
                    *reinterpret_cast<float2 *>(XSb + ystr_m + 64 * (3 - r)) = XB[r];
                }
                if (l == 0) *reinterpret_cast<float2 *>(XSb + 256) = x256f;                           // slot(256) = 32
            }
        };
        // ---- fp32-first forward transform (round 5; F32 instances; spectrum32_1024).  The peak decisions need the |X|^2 of an fp64 spectrum only where two magnitudes
        //      that are compared lie within the error of an fp32 transform of each other: a guard band around every comparison the decisions rest on (below, at the
        //      flags), and the fp64 transform only for frames with a comparison inside it.  Returns the absolute part K of the frame's guard band (0: the frame is
        //      outside the guarded range -- silent, below -117 dB, non-finite or absurdly large input -- and takes the fp64 transform unconditionally). ----
        [[maybe_unused]] auto forward32 = [&]() -> float {
#ifdef PV_STAMPS
            auto st32 = [&](int id) { if (id & 1) stamps.mark(id); else stamps.mark(id, true); };
#else
            NoStamp st32;
#endif
            spectrum32_1024<PV_F32_REGT1>(raw, hw, smem, l, wlfs, 1.0f / SC,
                                          [&](int r, pk::c32 xa, pk::c32 xb) {
                                              MAG[4 + l + 64 * r] = mag32(xa);
                                              MAG[4 + 512 - l - 64 * r] = mag32(xb);
                                              XA[r] = float2{xa.x, xa.y};
                                              XB[r] = float2{xb.x, xb.y};
                                          },
                                          [&](pk::c32 x256) { MAG[4 + 256] = mag32(x256); x256f = float2{x256.x, x256.y}; }, st32);
            if constexpr (!SPREAD) {
                unsigned ystr, ystr_m;
                { int lq = l; asm volatile("" : "+v"(lq)); ystr = yslot_bytes((unsigned)lq); ystr_m = yslot_bytes((unsigned)(512 - 192 - lq)); }
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    *reinterpret_cast<float2 *>(XSb + ystr + 64 * r) = XA[r];
                    *reinterpret_cast<float2 *>(XSb + ystr_m + 64 * (3 - r)) = XB[r];
                }
                if (l == 0) *reinterpret_cast<float2 *>(XSb + 256) = x256f;
            }
            return 0.f;                                                     // (the band's K comes from the magnitudes once they are read back in natural order: see the flags)
        };
        auto shift_table = [&]() {
            // ---- Math.round(peak * f) - peak (pv:125,147) for every possible peak bin, cached while f does not change.  BEFORE the prefetch is issued: the
            //      table build is a call, a function entry waits for vmcnt(0), and with the next frame's rows in flight that is an exposed HBM round trip in
            //      every frame whose f differs from the last one (a pitch sweep: -3 % per launch) ----
            {
                const unsigned pfb = __float_as_uint(pfm);
                if (!psh_valid || pfb != psh_key) { psh_key = pfb; psh_valid = true; build_shift_table_1024(pfm, wave_off, l); }
            }
        };
        auto slide_prefetch = [&]() {
            // slide the window: the rows the next frame adds are issued here and land behind the shift + inverse FFT
#pragma unroll
            for (int r = 0; r < 8 - S_ROWS; r++) raw[r] = raw[r + S_ROWS];
            {
                const int mn = (m + 1 < last_out) ? m + 1 : m;                 // the last frame of a chain re-reads its own rows (unused): no branch
                load_rows(&raw[8 - S_ROWS], S_ROWS, 8 - S_ROWS, mn);
                pf_next = pitch_row[mn];
            }
        };
        [[maybe_unused]] float guardK = 0.f;                               // F32: the absolute part of the guard band of this frame, 0 = frame out of the guarded range
        [[maybe_unused]] bool wide_first = false;                          // F32, wave-uniform: this chain's frames have been coming out as class B: the fp64 transform first
        if constexpr (!F32) { forward64(); shift_table(); slide_prefetch(); }
        else {
            shift_table();                                                  // (a call: before anything of this frame is in flight)
#ifndef PV_FLIP_COUNT
            wide_first = pred >= PRED_WIDE;
#endif
            if (wide_first) forward64(); else (void)forward32();
        }
        wave_sync();
        PV_STAMP(4);
        pv_prio(PH_PEAKS);
        // ---- peak flags (pv:95-116) for bins 8l..8l+7, nearest peaks by wave scans, then one route per source bin -- and the source bins themselves:
        //      lane l takes bins 8l .. 8l+7 (lane 63 also bin 512) out of the stash and moves them along its own routes ----
        int last_peak = -1, last_shift = 0;
        bool pairwise = false;                                              // wave-uniform, f < 1: every collision of this frame is a (falling side, rising side) pair
        bool nonfinite = false;                                             // wave-uniform: a magnitude of this frame is Inf or NaN
        unsigned rt[8];
        unsigned rt512 = NOROUTE;
        float2 xs[8], xs512;                                                // !SPREAD only: the lane's own source bins 8 l + i (and 512), from the stash
        {
            // |X|^2 >= 0, so the fp32 order of two magnitudes is the order of their bit patterns as unsigned integers: the strict test
            // "greater than all four neighbours" (pv:100-110, `>=` rejects) becomes c > max(neighbours) with v_max3_u32 -- two instructions per
            // bin plus eight shared pair maxima, instead of four compares and three mask ANDs.
            unsigned mg[12];
            unsigned pm[11], nm[8];                                         // pair maxima; nm[i] = the largest of the four neighbours of bin 8 l + i
            v4u dq;
            // volatile vector loads: otherwise the optimizer re-pairs the 12 words into five misaligned ds_read2_b32 (8 LDS cycles each)
            typedef const volatile __attribute__((address_space(3))) v2u *lds_v2u;
            typedef const volatile __attribute__((address_space(3))) v4u *lds_v4u;
            auto read_mags = [&]() {
                const v2u q0 = *(lds_v2u)(&MAG[4 + 8 * l - 2]);
                const v4u q1 = *(lds_v4u)(&MAG[4 + 8 * l]);
                const v4u q2 = *(lds_v4u)(&MAG[4 + 8 * l + 4]);
                const v2u q3 = *(lds_v2u)(&MAG[4 + 8 * l + 8]);
                dq = *(lds_v4u)(&DSH[8 * l]);
                if constexpr (!SPREAD) {
                    // the lane's source bins, natural order: bin 8 l + i sits in row i of the transposed stash, lane-contiguous (conflict-free ds_read_b64)
#pragma unroll
                    for (int i = 0; i < 8; i++) xs[i] = *reinterpret_cast<const float2 *>(XSb + 8 * l + 8 * YROW * i);
                    xs512 = *reinterpret_cast<const float2 *>(XSb + 512);   // slot(512) = 64; only lane 63 uses it
                }
                mg[0] = q0.x; mg[1] = q0.y; mg[2] = q1.x; mg[3] = q1.y; mg[4] = q1.z; mg[5] = q1.w;
                mg[6] = q2.x; mg[7] = q2.y; mg[8] = q2.z; mg[9] = q2.w; mg[10] = q3.x; mg[11] = q3.y;
#pragma unroll
                for (int j = 3; j < 11; j++) pm[j] = max(mg[j], mg[j + 1]);
#pragma unroll
                for (int i = 0; i < 8; i++) nm[i] = max(max(mg[i], mg[i + 1]), pm[i + 3]);
            };
            bool fl[8];
            auto take_flags = [&]() {
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    // bin k = 8l + i, candidates are 2 <= k < H - 2 = 511 (pv:97-100): lane 0 drops i < 2, lane 63 drops i = 7
                    const bool in_range = (i < 2) ? (l != 0) : (i == 7) ? (l != 63) : true;
                    fl[i] = in_range & (nm[i] < mg[i + 2]);
                }
            };
            // is any candidate bin of this lane within the band (K, R) of the largest of its four neighbours?  K = 0: the frame is out of the guarded range
            [[maybe_unused]] auto in_band = [&](float K, float Rr) -> bool {
                const pk::c32 KK{K, K}, RR{Rr, Rr};
                float tmin = 1.0f;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int i = 2 * j;
                    const pk::c32 c2{__uint_as_float(mg[i + 2]), __uint_as_float(mg[i + 3])}, n2{__uint_as_float(nm[i]), __uint_as_float(nm[i + 1])};
                    const pk::c32 d2 = pk::sub(c2, n2), s2 = pk::add(c2, n2);
                    pk::c32 t2 = pk::fms(d2, d2, pk::mul(pk::fma(s2, RR, KK), s2));      // (c - n)^2 - (c + n) (K + R (c + n)): <= 0 inside the band
                    if (j == 0 && l == 0) t2 = pk::c32{1.f, 1.f};            // bins 0, 1 and 511 are no candidates (pv:97-100)
                    if (j == 3 && l == 63) t2.y = 1.f;
                    tmin = fminf(fminf(tmin, t2.x), t2.y);
                }
                return !(K > 0.f) | (tmin <= 0.f);
            };
#ifdef PV_FLIP_COUNT
            // Validation build (tools/flip_count.py; never the product): EVERY frame runs the fp32 transform and then the fp64 one, and the two are compared --
            // frames whose flag sets differ at all; frames whose flags differ although the guard band did not ask for the fp64 transform (must be none); over all
            // candidate bins whose flag differs the largest q = (c - n)^2 / ((c + n) (K + R (c + n))) (the guard calls a bin ambiguous for q <= 1: sqrt(1 / q_max) is
            // the factor by which the band could shrink before a flip escapes); the error of the fp32 amplitudes beyond a relative allowance, in units of eps max|X|;
            // and for the class proof from the fp64 magnitudes: frames it calls class B for sure, and of those the ones the fp32 test calls class A (must be none).
            unsigned vb_f32bits = 0;
            float vb_q[8], vb_a32[8];
            bool vb_amb_any = false;
            float vb_K = 0.f;
            auto flip_first = [&](float gk) {
                vb_K = gk;
                vb_f32bits = 0;
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    vb_f32bits |= fl[i] ? (1u << i) : 0u;
                    const float c = __uint_as_float(mg[i + 2]), n = __uint_as_float(nm[i]), d = c - n, sm = c + n;
                    vb_q[i] = (gk > 0.f && sm > 0.f) ? (d * d) / (sm * (gk + GUARD_R * sm)) : 0.f;
                    vb_a32[i] = sqrtf(c);
                }
                vb_amb_any = __any(in_band(gk, GUARD_R));
            };
            auto flip_second = [&](bool sure_b) {
                if (vb_K > 0.f && p.fwd_stats) {
                    float e8 = 0.f, e32 = 0.f, amax = 0.f, a64[8];
#pragma unroll
                    for (int i = 0; i < 8; i++) { a64[i] = sqrtf(__uint_as_float(mg[i + 2])); amax = fmaxf(amax, a64[i]); }
#pragma unroll
                    for (int i = 0; i < 8; i++) {
                        const float e = fabsf(vb_a32[i] - a64[i]);
                        e8 = fmaxf(e8, e - 8.f * GUARD_EPS * a64[i]);
                        e32 = fmaxf(e32, e - 32.f * GUARD_EPS * a64[i]);
                    }
                    for (int o = 32; o; o >>= 1) { e8 = fmaxf(e8, __shfl_xor(e8, o, 64)); e32 = fmaxf(e32, __shfl_xor(e32, o, 64)); amax = fmaxf(amax, __shfl_xor(amax, o, 64)); }
                    if (l == 0) {
                        atomicMax(reinterpret_cast<unsigned *>(p.fwd_stats + 514), __float_as_uint(e8 / (GUARD_EPS * amax)));
                        atomicMax(reinterpret_cast<unsigned *>(p.fwd_stats + 516), __float_as_uint(e32 / (GUARD_EPS * amax)));
                    }
                }
                unsigned diff = 0;
                float qmax = 0.f;
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const bool dif = (fl[i] ? 1u : 0u) != ((vb_f32bits >> i) & 1u);
                    diff |= dif ? 1u : 0u;
                    if (dif) qmax = fmaxf(qmax, vb_q[i]);
                }
                const bool flip_any = __any(diff != 0u);
                n_flip += flip_any ? 1u : 0u;
                n_uncaught += (flip_any && !vb_amb_any) ? 1u : 0u;
                n_sure += sure_b ? 1u : 0u;
                n_incons += (sure_b && !vb_amb_any) ? 1u : 0u;
                if (diff && p.fwd_stats) atomicMax(reinterpret_cast<unsigned *>(p.fwd_stats + 512), __float_as_uint(qmax));
            };
#endif
            if constexpr (!F32) {
                read_mags();
                take_flags();
            } else {
                read_mags();
                take_flags();
                bool fall = true;
                if (!wide_first) {
                    // the frame's largest magnitude (|X|^2 >= 0: the order of the bit patterns; NaN / Inf come out on top) sets the absolute part of the band: the
                    // lane's own eight bins (and bin 512 behind lane 63's) are in registers here -- four maxima the non-finite test shares
                    guardK = guard_k_of(wave_max_u32(max(max(max(pm[3], pm[5]), max(pm[7], pm[9])), mg[2])));
                    fall = __any(in_band(guardK, GUARD_R));
#ifdef PV_FLIP_COUNT
                    flip_first(guardK);
                    fall = true;
#endif
                }
                if (fall) {
                    // Rare paths, out of line (forward_cold_1024): the transform hands |X|^2 and the fp32 spectrum over through LDS
                    auto cold = [&](int wide) -> float {
                        Raw8 rw;
#pragma unroll
                        for (int r = 0; r < 8; r++) rw.v[r] = raw[r];
                        const float k = forward_cold_1024<SPREAD>(rw, wide, wl, wlfs, 1.0f / SC, wave_off, l);
                        wave_sync();
                        pv_prio(PH_PEAKS);
                        read_mags();
                        take_flags();
                        if constexpr (SPREAD) {
                            const float2 *XV = reinterpret_cast<const float2 *>(XSb);
#pragma unroll
                            for (int r = 0; r < 4; r++) { XA[r] = XV[r * 64 + l]; XB[r] = XV[(4 + r) * 64 + l]; }
                            x256f = XV[512];                                // (only lane 0 uses it)
                        }
                        return k;
                    };
                    // class B (or, for a chain that runs it first, not known yet): the transform at the reference's width, its magnitudes, its flags -- and its
                    // spectrum, rounded to fp32, as the frame's sources: a class-B frame is exactly a frame of the !F32 instances
                    if (!wide_first) cold(1);
                    // what the fp64 magnitudes alone say about the class (see GUARD_GN)
                    const unsigned mb = wave_max_u32(max(max(max(pm[3], pm[5]), max(pm[7], pm[9])), mg[2]));       // the largest |X|^2 of bins 0 .. 512
                    const bool out_sure = mb < GUARD_M_MIN_BITS - GUARD_M_SLACK || mb >= GUARD_M_MAX_BITS + GUARD_M_SLACK;
                    const bool in_sure = mb >= GUARD_M_MIN_BITS + GUARD_M_SLACK && mb < GUARD_M_MAX_BITS - GUARD_M_SLACK;
                    const bool sure_b = out_sure || (in_sure && __any(in_band(GUARD_CKN * __uint_as_float(mb), GUARD_RN)));
#ifdef PV_FLIP_COUNT
                    flip_second(sure_b);
#endif
                    bool class_b = true;
                    if (wide_first && !sure_b) {
                        // the fp64 magnitudes do not settle this frame's class: the fp32 transform after all
                        const float k = cold(0);
                        class_b = __any(in_band(k, GUARD_R));
                        // (class B means the fp64 transform once more: keeping its flags and spectrum alive across the call above costs the hot paths their
                        //  registers -- loop invariants went to scratch memory --, and at the pitchFactor < 1 instance the stash, which the residue paths read,
                        //  holds the fp32 spectrum now)
                        if (class_b) cold(1);
                    }
                    // The counter: +1 for a frame that is class B and provable, -3 otherwise, fp64 first from PRED_WIDE up.  With the costs measured on the headline shape
                    // (a class-A frame 0.87, a class-B frame of an fp32-first chain 1.49, an unprovable frame of an fp64-first chain 1.45, resp. 2.3 when it turns out
                    // class B, of an fp64-only frame) the order pays off above three provable frames in four
                    pred = (class_b && sure_b) ? min(pred + 1u, (unsigned)PRED_MAX) : (pred > 3u ? pred - 3u : 0u);
#ifdef PV_FLIP_COUNT
                    n_fallback += vb_amb_any ? 1u : 0u;
#else
                    n_fallback += class_b ? 1u : 0u;
#endif
                } else {
                    pred = pred > 3u ? pred - 3u : 0u;
                }
                slide_prefetch();
            }
            // Non-finite magnitudes (NaN / Inf samples in the window): the order of bit patterns is not the order of floats any more.  In the
            // reference every comparison with NaN fails to reject (pv:103,107), peaks appear at every other bin and the NaNs they move reach
            // every output sample of the frame: the frame is NaN.  Detected here (two v_max3_u32 + a compare per frame), poisoned after the scatter.
            nonfinite = __any(max(max(max(pm[3], pm[5]), max(pm[7], pm[9])), mg[2]) >= 0x7F800000u);
            if (dbg) {
#pragma unroll
                for (int i = 0; i < 8; i++) { p.dbg_flags[8 * l + i] = fl[i] ? 1 : 0; p.dbg_mag[8 * l + i] = __uint_as_float(mg[i + 2]); }
                if (l == 63) { p.dbg_flags[512] = 0; p.dbg_mag[512] = __uint_as_float(mg[10]); }
            }
            // Every candidate peak travels as one packed word (bin << 16 | shift & 0xFFFF): the shifts of the lane's own 8 bins come from ONE
            // 16-byte read of the shift table (a per-bin DSH[owner] lookup is a 4-way bank conflict by construction: lanes l and l + 16 sit
            // 256 bytes apart), and the neighbour lanes' peaks bring their shift along in the same bpermute.  Packed words order like bins.
            constexpr int NEGPD = -(2048 << 16), POSPD = 4096 << 16;        // "no peak on this side"
            int pd[8];
#pragma unroll
            for (int i = 0; i < 8; i++)                                     // bytes {d.lo, d.hi, bin.lo, bin.hi}
                pd[i] = (int)__builtin_amdgcn_perm((unsigned)(8 * l + i), dq[i >> 1], (i & 1) ? 0x05040302u : 0x05040100u);
            int lastown[8], firstown[8];                                    // last own peak <= bin i / first own peak > bin i (packed)
            int cur = NEGPD;
#pragma unroll
            for (int i = 0; i < 8; i++) { cur = fl[i] ? pd[i] : cur; lastown[i] = cur; }
            int nx = POSPD;
#pragma unroll
            for (int i = 7; i >= 0; i--) { firstown[i] = nx; nx = fl[i] ? pd[i] : nx; }
            const int last_in = cur, first_in = nx;
            // nearest peak below / above this lane's byte: the 64-bit ballot of non-empty lanes locates the neighbour lane,
            // one bpermute each fetches its last / first peak (two independent LDS round trips instead of a 6-step scan)
            const unsigned long long occ = __ballot(cur >= 0);
            const unsigned long long below = occ & ((1ull << l) - 1ull);
            const unsigned long long above = (l == 63) ? 0ull : (occ >> (l + 1));
            const int src_lo = below ? 63 - __clzll((long long)below) : 0;
            const int src_hi = above ? l + __ffsll((long long)above) : 0;
            int cprev = __shfl(last_in, src_lo, 64), cnext = __shfl(first_in, src_hi, 64);
            if (!below) cprev = NEGPD;
            if (!above) cnext = POSPD;
            if (occ == 0ull) {                                              // no peak at all (wave-uniform): nothing moves (pv:122 loop is empty)
#pragma unroll
                for (int i = 0; i < 8; i++) rt[i] = NOROUTE;
            } else {
                const int lp = __shfl(last_in, 63 - __clzll((long long)occ), 64);
                last_peak = lp >> 16;
                last_shift = (int)(short)(lp & 0xFFFF);
                // owner rule (pv:132-141): regions tile [0, N); a bin belongs to the peak on its left iff it is strictly closer to it
                // (b < prv + ceil(gap/2)  <=>  b - prv < nxt - b; the midpoint of an even gap goes right).  Sentinels make the first
                // region start at 0 (pv:132) and the last one end at N (pv:133); at least one side is a real peak here.
                // shift (pv:147-152): route = ((delta * t) mod N) << 16 | target; a route is valid iff its target field is < H
                // (pv:127-129 via DROP, pv:150-152, negative index); bits above the 10 rotation bits are don't-care.
                auto route_of = [&](int b, int pp, int pn) -> unsigned {
                    const int own = (b - (pp >> 16) < (pn >> 16) - b) ? pp : pn;
                    const int delta = __builtin_amdgcn_sbfe(own, 0, 16);
                    return __builtin_amdgcn_perm((unsigned)__mul24(delta, tmod), (unsigned)(b + delta), 0x05040100u);
                };
#pragma unroll
                for (int i = 0; i < 8; i++) rt[i] = route_of(8 * l + i, max(lastown[i], cprev), min(firstown[i], cnext));
                if (l == 63) rt512 = route_of(512, max(last_in, cprev), POSPD);   // source bin N/2: owner is the last peak
                if (!SPREAD && !(pf >= 1.0)) {
                    // f < 1: regions compress and their targets overlap (pv:169-170).  The targets of a region are contiguous, so the overlap of two
                    // neighbours is the LAST ov = delta_i - delta_{i+1} targets of region i against the FIRST ov of region i+1.  While ov <= floor(gap / 2)
                    // -- the length of the rising side of peak i+1, the shorter of the two sides that meet -- every collision is ONE source from the falling
                    // side of a peak (owned by the peak on its left, the peak bin included) against ONE from the rising side of the next peak (owned by the
                    // peak on its right), never two of a kind.  The scatter then needs no claim rounds: falling-side sources store, rising-side sources add
                    // -- in the reference's order, region i before region i+1 (pv:122,146).  Bit 31 of a route = rising side (route_of leaves a stray bit
                    // of delta * t there; the rotation ignores it); one gap with a longer overlap (f below ~0.6, or very close peaks) sends the whole frame
                    // through the claim rounds instead.  tests/test_pairwise_rule.py checks the rule on random peak sets.
                    rt512 &= 0x7FFFFFFFu;                               // bin N/2: falling side of the last peak
                    bool bad = false;
#pragma unroll
                    for (int i = 0; i < 8; i++) {
                        const int pp = max(lastown[i], cprev), pn = min(firstown[i], cnext), b = 8 * l + i;
                        const bool rising = !(b - (pp >> 16) < (pn >> 16) - b);
                        rt[i] = (rt[i] & 0x7FFFFFFFu) | (rising ? 0x80000000u : 0u);
                    }
                    if (!(pfm >= PV_PAIRWISE_SURE)) {                       // (f >= 2/3: the test cannot fail, see pv_device_common.h)
#pragma unroll
                        for (int i = 0; i < 8; i++) {
                            const int pp = max(lastown[i], cprev), pn = min(firstown[i], cnext);
                            const int gap = (pn >> 16) - (pp >> 16), ov = __builtin_amdgcn_sbfe(pp, 0, 16) - __builtin_amdgcn_sbfe(pn, 0, 16);
                            bad |= ov > (gap >> 1);                         // (a missing neighbour is a sentinel thousands of bins away)
                        }
                    }
                    pairwise = !__any(bad);
                }
            }
        }
        PV_STAMP(5);
        pv_prio(PH_SCATTER);
        int upper_end = H;
        if (last_peak >= 0 && last_shift < 0) { upper_end = H - last_shift; if (upper_end > N) upper_end = N; }      // DROP is positive
        pk::c32 zi[8];
        if constexpr (SPREAD && FP64) {
            // ---- fp64 flavour: shift, c2r pass and inverse FFT in doubles (see PV_FP64_FLAVOUR at the top) ----
            wave_sync();
            double2 *Yd = reinterpret_cast<double2 *>(smem + OFF_Y);
            unsigned *ROUTE = reinterpret_cast<unsigned *>(smem + F64_ROUTE);
            unsigned short *CLAIM = reinterpret_cast<unsigned short *>(smem + F64_ROUTE);
            *reinterpret_cast<uint4 *>(&ROUTE[8 * l]) = uint4{rt[0], rt[1], rt[2], rt[3]};
            *reinterpret_cast<uint4 *>(&ROUTE[8 * l + 4]) = uint4{rt[4], rt[5], rt[6], rt[7]};
            if (l == 63) ROUTE[512] = rt512;
#pragma unroll
            for (int r = 0; r < 8; r++) *reinterpret_cast<v4f *>(smem + OFF_Y + 16 * l + 1024 * r) = v4f{0.f, 0.f, 0.f, 0.f};      // Y[0 .. 512)
            if (l == 0) Yd[512] = double2{0.0, 0.0};
            wave_sync();
            if (pf >= 1.0) {                                                // disjoint regions: plain stores
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const unsigned ra = ROUTE[l + 64 * r], ta = ra & 0xFFFFu;
                    const unsigned rb = ROUTE[512 - l - 64 * r], tb = rb & 0xFFFFu;
                    if (ta < (unsigned)H) Yd[ta] = rotate_route_d(R, ra, XAd[r], p.tw64);
                    if (tb < (unsigned)H) Yd[tb] = rotate_route_d(R, rb, XBd[r], p.tw64);
                }
                if (l == 0) { const unsigned r256 = ROUTE[256], tg = r256 & 0xFFFFu; if (tg < (unsigned)H) Yd[tg] = rotate_route_d(R, r256, x256d, p.tw64); }
            } else {
                // f < 1 (and NaN): `+=` collisions (pv:169-170) resolved by claim rounds, then the sources above Nyquist (all owned by the last peak, pv:133)
                // from the re-run stage structure -- this flavour has neither the pairwise scatter nor the closed form of the residue
                unsigned rc[9];
                double2 ys[9];
                int id[9];
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    id[r] = l + 64 * r; rc[r] = ROUTE[id[r]]; ys[r] = rotate_route_d(R, rc[r], XAd[r], p.tw64);
                    id[4 + r] = 512 - l - 64 * r; rc[4 + r] = ROUTE[id[4 + r]]; ys[4 + r] = rotate_route_d(R, rc[4 + r], XBd[r], p.tw64);
                }
                rc[8] = (l == 0) ? ROUTE[256] : NOROUTE;
                ys[8] = rotate_route_d(R, rc[8], x256d, p.tw64);
                id[8] = 256;
                wave_sync();                                               // routes are in registers: the claim words may overwrite them
                claim_rounds_d<9, true>(rc, ys, id, Yd, CLAIM);
                if (upper_end > H) {
                    const int up_delta = last_shift;
                    residue_scatter_1024_d<R>(src.in, src.hist, src.hist_len, (long)(m + 1) * HOP - N, p.hann, p.tw64, wave_off, l, upper_end, up_delta,
                                              (unsigned)((up_delta & (N - 1)) * tmod) & (N - 1), dbg ? p.dbg_X : nullptr);
                }
            }
            if (nonfinite && l == 0) Yd[1] = double2{__longlong_as_double(0x7FF8000000000000ll), __longlong_as_double(0x7FF8000000000000ll)};
            wave_sync();
            if (dbg) {
#pragma unroll
                for (int r = 0; r < 8; r++) { const int k = l + 64 * r; p.dbg_Y[2 * k] = (float)Yd[k].x; p.dbg_Y[2 * k + 1] = (float)Yd[k].y; }
                if (l == 0) { p.dbg_Y[1024] = (float)Yd[512].x; p.dbg_Y[1025] = (float)Yd[512].y; }
            }
            // c2r pre-pass in fp64: Z[k] = SC ((Yk + Ym*) + j e^{+2 pi j k/N} (Yk - Ym*)), conjugate pairs as in the product
            double2 zd[8], zb[4];
            const double sc = (double)SC;
            const double2 wlc{wl.x, -wl.y};                                 // e^{+2 pi j l / N}
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int k = l + 64 * r;
                double2 yk = Yd[k], ym = Yd[M - k];
                if (k == 0) { yk.y = 0.0; ym.y = 0.0; }
                const double2 E{yk.x + ym.x, yk.y - ym.y}, O{yk.x - ym.x, yk.y + ym.y};
                const double2 c = cmul(mul_w16<double, true>(O, r), wlc);
                zd[r] = double2{(E.x - c.y) * sc, (E.y + c.x) * sc};        // E + j c
                zb[r] = double2{(E.x + c.y) * sc, -(E.y - c.x) * sc};       // conj(E - j c)
            }
            const double2 y256 = Yd[256];
            double2 *XCHd = reinterpret_cast<double2 *>(smem + F64_Q);
#pragma unroll
            for (int r = 0; r < 4; r++) XCHd[r * 64 + l] = zb[r];
            wave_sync();
#pragma unroll
            for (int r = 0; r < 4; r++) zd[7 - r] = XCHd[r * 64 + 64 - l];
            if (l == 0) zd[4] = double2{2.0 * y256.x * sc, -2.0 * y256.y * sc};
            wave_sync();
            pv_prio(PH_IA);
            fft512_wave<double, true>(zd, S64, reinterpret_cast<const double2 *>(smem_all + TAB_TW1C), reinterpret_cast<const double2 *>(smem_all + TAB_TW2C), l);
#pragma unroll
            for (int r = 0; r < 8; r++) zi[r] = pk::c32{(float)zd[r].x, (float)zd[r].y};    // fromComplexArray -> Float32Array (bundle:46-51)
        } else {
        // ---- zero Y (pv:121).  The magnitudes have been read (LDS instructions of a wave execute in order): the routes (SPREAD) / Y's tail (!SPREAD) may overwrite them ----
        wave_sync();
        float2 *Yn = reinterpret_cast<float2 *>(smem + OFF_Y);               // SPREAD: Y as a plain array of 513 bins
        if constexpr (SPREAD) {
            unsigned *ROUTE = reinterpret_cast<unsigned *>(smem + OFF_ROUTE);
            *reinterpret_cast<uint4 *>(&ROUTE[8 * l]) = uint4{rt[0], rt[1], rt[2], rt[3]};
            *reinterpret_cast<uint4 *>(&ROUTE[8 * l + 4]) = uint4{rt[4], rt[5], rt[6], rt[7]};
            if (l == 63) ROUTE[512] = rt512;
#pragma unroll
            for (int r = 0; r < 4; r++) *reinterpret_cast<v4f *>(&Yn[2 * l + 128 * r]) = v4f{0.f, 0.f, 0.f, 0.f};     // 16 bytes per lane and store
            if (l == 0) Yn[512] = float2{0.f, 0.f};
        } else {
#pragma unroll
            for (int r = 0; r < 4; r++) *reinterpret_cast<v4f *>(Yb + 16 * l + 1024 * r) = v4f{0.f, 0.f, 0.f, 0.f};
            { int lq = l; asm volatile("" : "+v"(lq)); *reinterpret_cast<v4f *>(Yb + ((lq < 16) ? 4096 + 16 * lq : 16 * lq)) = v4f{0.f, 0.f, 0.f, 0.f}; }   // the last 256 of the 4352 bytes (the other lanes repeat their first store: no exec mask)
        }
        wave_sync();
        // ---- shiftPeaks (pv:119-173) ----
        {
            // every rotation of a frame is exp(2 pi j delta (m mod R) / R) and m mod R is wave-uniform: a frame with m = 0 (mod R) moves its bins unrotated,
            // m = R/2 (mod R) only flips signs (bit 9 of the rotation index = bit 25 of the route); tmod = (m mod R) * N/R
            auto rot_mode = [&](auto mode_tag, unsigned r_, float2 v) -> float2 {
                constexpr int MODE = decltype(mode_tag)::value;
                if (MODE == 0) return v;
                if (MODE == 2) {
                    const unsigned sg = (r_ << 6) & 0x80000000u;
                    return float2{__uint_as_float(__float_as_uint(v.x) ^ sg), __uint_as_float(__float_as_uint(v.y) ^ sg)};
                }
                return rotate_route<R, 10>(r_, v, p.tw32);
            };
            [[maybe_unused]] auto ystore = [&](unsigned t, float2 v) { *reinterpret_cast<float2 *>(Yb + yslot_bytes(t)) = v; };
            // a source without a valid target stores into slot 543, which is no bin's (row 7 ends at slot 539): straight-line code instead of a branch per store
            [[maybe_unused]] auto ystore_if = [&](bool ok, unsigned t, float2 v) { *reinterpret_cast<float2 *>(Yb + (ok ? yslot_bytes(t) : 8u * (YSLOTS - 1))) = v; };
            if constexpr (SPREAD) {
                // delta_i = round(p_i f) - p_i is non-decreasing in i for f >= 1, the shifted regions stay disjoint: plain stores of the register-resident
                // source bins along the routes of the table
                const unsigned *ROUTE = reinterpret_cast<const unsigned *>(smem + OFF_ROUTE);
                auto scatter = [&](auto mode_tag) {
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const unsigned ra = ROUTE[l + 64 * r], ta = ra & 0xFFFFu;
                        const unsigned rb = ROUTE[512 - l - 64 * r], tb = rb & 0xFFFFu;
                        if (ta < (unsigned)H) Yn[ta] = rot_mode(mode_tag, ra, XA[r]);
                        if (tb < (unsigned)H) Yn[tb] = rot_mode(mode_tag, rb, XB[r]);
                    }
                    if (l == 0) { const unsigned r256 = ROUTE[256], tg = r256 & 0xFFFFu; if (tg < (unsigned)H) Yn[tg] = rot_mode(mode_tag, r256, x256f); }
                };
                if (tmod == 0) scatter(std::integral_constant<int, 0>{});
                else if (tmod == N / 2) scatter(std::integral_constant<int, 2>{});
                else scatter(std::integral_constant<int, 1>{});
            } else if (pf >= 1.0) {
                // (a frame with f >= 1 in a chain that also has frames below 1: disjoint regions, plain stores, each lane its own 8 bins)
                auto scatter = [&](auto mode_tag) {
#pragma unroll
                    for (int i = 0; i < 8; i++) { const unsigned t = rt[i] & 0xFFFFu; if (t < (unsigned)H) ystore(t, rot_mode(mode_tag, rt[i], xs[i])); }
                    { const unsigned t = rt512 & 0xFFFFu; if (t < (unsigned)H) ystore(t, rot_mode(mode_tag, rt512, xs512)); }
                };
                if (tmod == 0) scatter(std::integral_constant<int, 0>{});
                else if (tmod == N / 2) scatter(std::integral_constant<int, 2>{});
                else scatter(std::integral_constant<int, 1>{});
            } else if (pairwise) {
                // f < 1, every collision is (falling side, rising side): pass A, the falling-side sources -- and the residue above Nyquist, which continues
                // the falling side of the last peak -- store into the zeroed Y; pass B, the rising-side sources read, add, store (the reference's order).
                const bool need_res = upper_end > H;
                const bool fast_res = need_res && (upper_end <= H + 128);
                const int up_delta = need_res ? last_shift : 0;                                        // sources above Nyquist: all owned by the last peak (pv:133), whose shift came with it
                const unsigned up_ridx = (unsigned)((up_delta & (N - 1)) * tmod) & (N - 1);
                float2 ys[8];
#pragma unroll
                for (int i = 0; i < 8; i++) ys[i] = rotate_route<R, 10>(rt[i], xs[i], p.tw32);
                const float2 ys512 = rotate_route<R, 10>(rt512, xs512, p.tw32);
                unsigned key[8];
#pragma unroll
                for (int i = 0; i < 8; i++) key[i] = rt[i] & 0x8000FFFFu;   // side bit | target: < 513 = a valid falling-side source
#pragma unroll
                for (int i = 0; i < 8; i++) ystore_if(key[i] < 513u, key[i], ys[i]);
                ystore_if((rt512 & 0xFFFFu) < 513u, rt512 & 0xFFFFu, ys512);
                if (fast_res) {
                    unsigned rt2[2];
                    float2 ys2[2];
                    residue_fast_1024<R>(p.tw32, wave_off, l, upper_end, up_delta, up_ridx, dbg ? p.dbg_X : nullptr, rt2, ys2);
#pragma unroll
                    for (int j = 0; j < 2; j++) ystore_if((rt2[j] & 0xFFFFu) < 513u, rt2[j] & 0xFFFFu, ys2[j]);
                }
                wave_sync();
                unsigned ab[8];
                float2 o[8];
#pragma unroll
                for (int i = 0; i < 8; i++) {                                   // rising-side sources: their target's slot; everybody else: the dummy slot
                    ab[i] = (key[i] - 0x80000000u < 513u) ? yslot_bytes(key[i] & 0xFFFFu) : 8u * (YSLOTS - 1);
                    o[i] = *reinterpret_cast<const float2 *>(Yb + ab[i]);
                }
#pragma unroll
                for (int i = 0; i < 8; i++) *reinterpret_cast<float2 *>(Yb + ab[i]) = float2{o[i].x + ys[i].x, o[i].y + ys[i].y};
                if (need_res && !fast_res) {
                    wave_sync();
                    residue_scatter_1024<R>(src.in, src.hist, src.hist_len, src.sys, (long)(m + 1) * HOP - N, p.tw32, wave_off, l, upper_end, up_delta, up_ridx,
                                            dbg ? p.dbg_X : nullptr, true);
                }
            } else {
                scatter_claims_1024<R>(Routes9{{rt[0], rt[1], rt[2], rt[3], rt[4], rt[5], rt[6], rt[7], rt512}}, wave_off, l, tmod, last_peak, upper_end,
                                       src.in, src.hist, src.hist_len, src.sys, (long)(m + 1) * HOP - N, p.tw32, dbg ? p.dbg_X : nullptr);
            }
        }
        if (nonfinite && l == 0) *reinterpret_cast<float2 *>(Yb + (SPREAD ? 8u : yslot_bytes(1u))) = float2{__uint_as_float(0x7FC00000u), __uint_as_float(0x7FC00000u)};   // bin 1; see "Non-finite magnitudes" above
        wave_sync();
        PV_STAMP(6);
        pv_prio(PH_C2R);
        // strided-order addresses of Y: bin l + 64 r at ystr + YR r, bin 512 - l - 64 r at ystr_m + YR (3 - r) -- plain array (SPREAD): 8 l, 8 (320 - l), YR = 512;
        // transposed: yslot_bytes, YR = 64, from an opaque copy of the lane id (four instructions per frame; see the split pass)
        constexpr int YR = SPREAD ? 512 : 64;
        unsigned ystr, ystr_m;
        if constexpr (SPREAD) { ystr = 8u * (unsigned)l; ystr_m = 8u * (unsigned)(512 - 192 - l); }
        else { int lq = l; asm volatile("" : "+v"(lq)); ystr = yslot_bytes((unsigned)lq); ystr_m = yslot_bytes((unsigned)(512 - 192 - lq)); }
        if (dbg) {
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const int k = l + 64 * r;
                const float2 y = *reinterpret_cast<const float2 *>(Yb + (SPREAD ? 8u * (unsigned)k : yslot_bytes((unsigned)k)));
                p.dbg_Y[2 * k] = y.x; p.dbg_Y[2 * k + 1] = y.y;
            }
            if (l == 0) { const float2 y = *reinterpret_cast<const float2 *>(Yb + (SPREAD ? 4096 : 512)); p.dbg_Y[1024] = y.x; p.dbg_Y[1025] = y.y; }
        }
        // ---- c2r pre-pass (bundle:69-76,102-114 folded): Z[k] = SC ((Yk + Ym*) + j e^{+2 pi j k/N} (Yk - Ym*)), packed fp32 ----
        {
            const float sc = SC;
            const pk::c32 scsc{sc, sc};
            {
                // conjugate pairs again: with E = Yk + conj(Ym), O = Yk - conj(Ym), c = e^{+2 pi j k/N} O / N (m = 512 - k):
                // Z[k] = E / N + j c and Z[m] = conj(E / N - j c); lane l computes k = l + 64 r, r < 4, and hands Z[m] to lane 64-l, register 7-r
                pk::c32 zb[4];
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    pk::c32 yk = *reinterpret_cast<const pk::c32 *>(Yb + ystr + YR * r), ym = *reinterpret_cast<const pk::c32 *>(Yb + ystr_m + YR * (3 - r));
                    if (r == 0 && l == 0) { yk.y = 0.f; ym.y = 0.f; }
                    const pk::c32 E = pk::add_conj(yk, ym), O = pk::sub_conj(yk, ym);
                    const pk::c32 c = pk::cmul(mul_w16_inv_pk(O, r), wlfs);
                    zi[r] = pk::fma_addj(E, scsc, c);
                    zb[r] = pk::fma_conj_subj(E, scsc, c);
                }
                const pk::c32 y256 = *reinterpret_cast<const pk::c32 *>(Yb + (SPREAD ? 2048 : 256));   // bin 256: byte 2048 of the plain array, slot 32 of the transposed one
                // hand-over through LDS (the stash is dead here): Z[m] of the pair (l', r') lands in lane 64 - l', register
                // 7 - r'; read address r * 64 + 64 - l for every lane (lane 0 pairs with itself one register higher, and its register 4
                // is the self-paired bin 256, replaced below)
                pk::c32 *XCH = reinterpret_cast<pk::c32 *>(smem + OFF_RESQ);
#pragma unroll
                for (int r = 0; r < 4; r++) XCH[r * 64 + l] = zb[r];
                wave_sync();
#pragma unroll
                for (int r = 0; r < 4; r++) zi[7 - r] = XCH[r * 64 + 64 - l];
                if (l == 0) zi[4] = pk::c32{2.0f * y256.x * sc, -2.0f * y256.y * sc};
            }
        }
        wave_sync();
#ifdef PV_STAMPS
        stamps.mark(7);
        fft512_wave_inv_pk<true>(zi, reinterpret_cast<pk::c32 *>(S32), TW1F4, TW2F4, l, [&](int id) { if (id & 1) stamps.mark(8 + id); else stamps.mark(8 + id, true); });
#else
        fft512_wave_inv_pk(zi, reinterpret_cast<pk::c32 *>(S32), TW1F4, TW2F4, l);
#endif
        }
        // The rows and the pitchFactor prefetched for the next frame are CONSUMED here, in front of this frame's output stores.  vmcnt counts loads and stores in ONE queue, in
        // order: left to the loop's back edge, the first use of the prefetched registers (the copies of the loop-carried window, the readfirstlane of f at the top) waits
        // for the stores issued after the loads as well -- `s_waitcnt vmcnt(0)` at the top of every frame, a store's full round trip to memory.  Here, a thousand
        // instructions after their issue, the loads have landed and the wait is free; the stores then stay in flight across the loop edge.
#ifndef PV_NO_CONSUME
#pragma unroll
        for (int r = 8 - S_ROWS; r < 8; r++) asm volatile("" :: "v"(raw[r]));
        asm volatile("" :: "v"(pf_next));
#endif
        // ---- Hann (pv:67), overlap-add in reference order (ola:149-157), emit (ola:111-118), shift (ola:130-137) ----
        pv_prio(PH_OLA);
        {
            const bool emit_out = (m >= emit_v);
            float2 fr[8];
#pragma unroll
            for (int j = 0; j < 4; j++) { const v4f h = HW4[j * 64 + l]; hw[2 * j] = pk::c32{h.x, h.y}; hw[2 * j + 1] = pk::c32{h.z, h.w}; }   // stays live for the next frame
#pragma unroll
            for (int r = 0; r < 8; r++) { const pk::c32 f = pk::mul(zi[r], hw[r]); fr[r] = float2{f.x, f.y}; }   // the windowed frame is ROUNDED to fp32 before it is accumulated
                                                                                                               // (Float32Array, pv:67): an asm multiply -- a plain one is contracted into the adds below (v_pk_fma_f32), one rounding less than the reference
#pragma unroll
            for (int r = 0; r < S_ROWS; r++) {
                const float2 o{acc[r].x + fr[r].x, acc[r].y + fr[r].y};
                if (emit_out) {
                    float *dst = outp + (long)m * HOP + 2 * l + 128 * r;
                    // streaming output: non-temporal, so the 1 GB/launch of results does not evict the input windows the next
                    // frames re-read from L2
                    if (vec_out) __builtin_nontemporal_store(v2f{o.x, o.y}, reinterpret_cast<v2f *>(dst));
                    else { __builtin_nontemporal_store(o.x, dst); __builtin_nontemporal_store(o.y, dst + 1); }
                }
            }
#pragma unroll
            for (int r = 0; r < LROWS; r++) {
                const int s = r + S_ROWS;
                acc[r] = (s < LROWS) ? float2{acc[s].x + fr[s].x, acc[s].y + fr[s].y} : fr[s];
            }
        }
        PV_STAMP(12);
    }
#ifdef PV_STAMPS
    if (p.stamps && l == 0) {
        unsigned *o = p.stamps + 16 * chain;
        for (int i = 0; i < 13; i++) o[i] = stamps.acc[i];
        o[13] = stamps.prev - stamp_t0;
        o[14] = (unsigned)(last_out - first_frame);
        o[15] = (unsigned)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_REG_HW_ID: SIMD / CU / SE of this wave
    }
#endif

    if constexpr (F32) {
        // forward-transform statistics (pv_forward_stats): {frames computed, frames that fell back to fp64}, spread over 128 slots so that the chains of a launch,
        // which all end together, do not queue on one address
        if (p.fwd_stats && l == 0) {
            unsigned long long *st = p.fwd_stats + 2 * (chain & 127);
            atomicAdd(st, (unsigned long long)(last_out - first_frame));
            if (n_fallback) atomicAdd(st + 1, (unsigned long long)n_fallback);
#ifdef PV_FLIP_COUNT
            if (n_flip) atomicAdd(st + 256, (unsigned long long)n_flip);
            if (n_uncaught) atomicAdd(st + 257, (unsigned long long)n_uncaught);
            if (n_sure) atomicAdd(p.fwd_stats + 600, (unsigned long long)n_sure);
            if (n_incons) atomicAdd(p.fwd_stats + 601, (unsigned long long)n_incons);
#endif
        }
    }
    if (chunk == p.nchunks - 1) {
#pragma unroll
        for (int r = 0; r < LROWS; r++) {
            float *a = acc_out + (long)ch * (N - HOP) + 2 * l + 128 * r;
            a[0] = acc[r].x; a[1] = acc[r].y;
            // the history of the next call = the last N - hop samples of the stream = rows S_ROWS..7 of the last frame's window, which the slide has
            // left in raw[0 .. 8 - S_ROWS): stored from registers (re-reading them costs a streaming quantum an exposed memory -- or PCIe -- round trip)
            float *hs = hist_out + (long)ch * (N - HOP) + 2 * l + 128 * r;
            hs[0] = raw[r].x; hs[1] = raw[r].y;
        }
    }
    pv_signal_done<false>(p.done, done_seq, chain);
    if (RESIDENT) goto resident_top;
}

// One wave per chain: are all its pitchFactors >= 1?  The chain is appended to the list of its class -- sixteen chains per workgroup, one atomic
// per class and workgroup (3000 waves bumping one counter each cost the headline launch 38 us).  The order of the workgroups' blocks within a class is
// whatever the atomics make it: chains are independent, the results do not depend on it.  list = {count[2], class 0 [nchains], class 1 [nchains]}.
constexpr int CLS_WAVES = 16;
template <int S_ROWS>
__global__ __launch_bounds__(64 * CLS_WAVES) void pv_classify_chains(const PvKernelParams p, unsigned *list, unsigned *list_next)
{
    if (list_next && blockIdx.x == 0 && threadIdx.x < 2) list_next[threadIdx.x] = 0u;     // the counters of the handle's NEXT launch (its previous user is behind us in stream order)
    constexpr int HOP = 128 * S_ROWS, R = 1024 / HOP;
    __shared__ unsigned cls_of[CLS_WAVES], base[2];
    const long nchains = (long)p.nch * p.nchunks;
    const int wv = threadIdx.x >> 6, l = threadIdx.x & 63;
    const long chain = (long)blockIdx.x * CLS_WAVES + wv;
    unsigned cls = 2u;                                                     // 2 = no chain
    if (chain < nchains) {
        const int ch = (int)(chain / p.nchunks), chunk = (int)(chain - (long)ch * p.nchunks);
        const int first_out = chunk * p.frames_per_chunk;
        int last_out = first_out + p.frames_per_chunk;
        if (last_out > p.nhops) last_out = p.nhops;
        int first_frame = first_out - (R - 1);
        if (first_frame < 0) first_frame = 0;
        const float *pitch_row = p.pitch + (p.pitch_stride ? (long)(ch / p.ch_per_stream) * p.pitch_stride : 0);
        bool low = false;
        // eight independent loads in flight per lane and pass (one after the other they were six dependent memory round trips: 7.7 -> 6.7 us for the headline launch)
        for (int m0 = first_frame + l; m0 < last_out; m0 += 512) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; j++) { const int m = m0 + 64 * j; v[j] = pitch_row[m < last_out ? m : last_out - 1]; }
#pragma unroll
            for (int j = 0; j < 8; j++) low |= !(v[j] >= 1.0f);                                 // NaN counts as "not >= 1", as in the kernel
        }
        cls = __any(low) ? 1u : 0u;
    }
    if (l == 0) cls_of[wv] = cls;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned n[2] = {0u, 0u};
        for (int w = 0; w < CLS_WAVES; w++) if (cls_of[w] < 2u) n[cls_of[w]]++;
        base[0] = n[0] ? atomicAdd(&list[0], n[0]) : 0u;
        base[1] = n[1] ? atomicAdd(&list[1], n[1]) : 0u;
    }
    __syncthreads();
    if (l == 0 && cls < 2u) {
        unsigned before = 0;
        for (int w = 0; w < wv; w++) before += (cls_of[w] == cls) ? 1u : 0u;
        list[2 + cls * nchains + base[cls] + before] = (unsigned)chain;
    }
}

template <int S_ROWS, bool AUX>
hipError_t launch_wave(const PvKernelParams &p, int nch, int nchunks, hipStream_t st, int spread, unsigned *list, unsigned *list_next)
{
    static std::atomic<bool> attr_done[16], attr_done_s[16];
    static std::atomic<bool> attr_done_f[16], attr_done_g[16];
    // F32FWD: the product's instances take the peak decisions on an fp32 forward transform behind a guard band (F32; p.fwd64 = 0); with p.fwd64 != 0, in the tap
    // instance and in the reference-width flavour every frame runs the fp64 forward transform (the round-4 kernels, bit for bit)
    constexpr bool F32OK = !AUX && !FP64;
    const bool f32 = F32OK && !p.fwd64;
    auto k = f32 ? pv_wave_kernel_1024<S_ROWS, AUX, false, false, F32OK> : pv_wave_kernel_1024<S_ROWS, AUX, false, false, false>;
    auto ks = f32 ? pv_wave_kernel_1024<S_ROWS, false, false, true, F32OK> : pv_wave_kernel_1024<S_ROWS, false, false, true>;
    {
        hipError_t e = pv_set_dynamic_lds_once(f32 ? attr_done_g : attr_done, reinterpret_cast<const void *>(k), (int)pv_wave_lds_bytes());
        if (e == hipSuccess && !AUX) e = pv_set_dynamic_lds_once(f32 ? attr_done_f : attr_done_s, reinterpret_cast<const void *>(ks), (int)pv_wave_lds_bytes());
        if (e != hipSuccess) return e;
    }
    PvKernelParams q = p;
    q.nchunks = nchunks;
    q.nch = nch;
    const long chains = (long)nch * nchunks;
    const dim3 grid((unsigned)((chains + WAVES - 1) / WAVES), 1, 1), block(64 * WAVES, 1, 1);
    if (FP64 && !AUX) {                                                    // reference-width flavour: ONE instance (the SPREAD flow with its own colliding scatter)
        hipLaunchKernelGGL(ks, grid, block, pv_wave_lds_bytes(), st, q);
    } else if (AUX || spread == 0 || (spread < 0 && !list)) {              // one instance that handles every pitchFactor
        hipLaunchKernelGGL(k, grid, block, pv_wave_lds_bytes(), st, q);
    } else if (spread > 0) {
        if (!AUX) hipLaunchKernelGGL(ks, grid, block, pv_wave_lds_bytes(), st, q);
    } else {
        // classify on the device, then both instances over the whole grid: a workgroup beyond its class's count leaves at once
        if (!list_next) {                                                  // (a caller without a second list: the counters are zeroed here, one more operation per launch)
            hipError_t e = hipMemsetAsync(list, 0, 2 * sizeof(unsigned), st);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(pv_classify_chains<S_ROWS>, dim3((unsigned)((chains + CLS_WAVES - 1) / CLS_WAVES), 1, 1), dim3(64 * CLS_WAVES, 1, 1), 0, st, q, list, list_next);
        q.chain_count = list;
        q.chain_list = list + 2;
        if (!AUX) hipLaunchKernelGGL(ks, grid, block, pv_wave_lds_bytes(), st, q);
        hipLaunchKernelGGL(k, grid, block, pv_wave_lds_bytes(), st, q);
    }
    return hipGetLastError();
}

template <int S_ROWS>
hipError_t launch_wave_resident(const PvKernelParams &p, int nslots, hipStream_t st)
{
    static std::atomic<bool> attr_done[16], attr_done_f[16];
    const bool f32 = !FP64 && !p.fwd64;
    auto k = f32 ? pv_wave_kernel_1024<S_ROWS, false, true, false, !FP64> : pv_wave_kernel_1024<S_ROWS, false, true, false, false>;
    {
        const hipError_t e = pv_set_dynamic_lds_once(f32 ? attr_done_f : attr_done, reinterpret_cast<const void *>(k), (int)pv_wave_lds_bytes());
        if (e != hipSuccess) return e;
    }
    PvKernelParams q = p;
    q.nchunks = 1;
    q.nch = nslots;                                                        // waves beyond the handle's channel slots leave at once
    q.nhops = 1;
    q.frames_per_chunk = 1;
    hipLaunchKernelGGL(k, dim3((unsigned)((nslots + RES_WAVES - 1) / RES_WAVES), 1, 1), dim3(64 * RES_WAVES, 1, 1), pv_wave_lds_bytes(), st, q);
    return hipGetLastError();
}

}  // namespace

size_t pv_wave_lds_bytes() { return TAB_BYTES + WAVES * WAVE_LDS; }

int pv_wave_threads() { return 64 * WAVES; }

bool pv_wave_supported(int log2n, int hop) { return log2n == 10 && (hop == 128 || hop == 256 || hop == 512 || hop == 1024); }

hipError_t pv_launch_wave_resident(const PvKernelParams &p, int nslots, hipStream_t st)
{
    switch (p.hop) {
    case 128: return launch_wave_resident<1>(p, nslots, st);
    case 256: return launch_wave_resident<2>(p, nslots, st);
    case 512: return launch_wave_resident<4>(p, nslots, st);
    case 1024: return launch_wave_resident<8>(p, nslots, st);
    default: return hipErrorInvalidValue;
    }
}

hipError_t pv_launch_wave(const PvKernelParams &p, int nch, int nchunks, hipStream_t st, int spread, unsigned *list, unsigned *list_next)
{
    const bool aux = (p.dbg_mag != nullptr);
    switch (p.hop) {
    case 128: return aux ? launch_wave<1, true>(p, nch, nchunks, st, 0, nullptr, nullptr) : launch_wave<1, false>(p, nch, nchunks, st, spread, list, list_next);
    case 256: return aux ? launch_wave<2, true>(p, nch, nchunks, st, 0, nullptr, nullptr) : launch_wave<2, false>(p, nch, nchunks, st, spread, list, list_next);
    case 512: return aux ? launch_wave<4, true>(p, nch, nchunks, st, 0, nullptr, nullptr) : launch_wave<4, false>(p, nch, nchunks, st, spread, list, list_next);
    case 1024: return aux ? launch_wave<8, true>(p, nch, nchunks, st, 0, nullptr, nullptr) : launch_wave<8, false>(p, nch, nchunks, st, spread, list, list_next);
    default: return hipErrorInvalidValue;
    }
}
