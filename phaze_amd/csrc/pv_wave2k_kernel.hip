// pv_wave2k_kernel.hip -- one wavefront per 2048-point frame chain: N = 2048, hop in {128, 256, 512, 1024, 2048} (BASELINE configs[2] and the
// reference's shipped 2048 / 128), every pitchFactor.
//
// The workgroup kernel spreads a 2048-point frame over two wavefronts and pays ~21 workgroup barriers per frame for it: its waves sit
// parked 40 % of their life (profiles/r02_wg_pmc.md).  Here ONE wave holds the frame, 16 packed complex elements per lane, and no
// barrier exists:
//
//   z[n] = xw[2n] + j xw[2n+1], n < 1024, split by parity:   ze[n'] = z[2n'],  zo[n'] = z[2n'+1]      lane l, register r <-> n' = l + 64 r
//
//   forward  : E = FFT512(ze), O = FFT512(zo)  (the two-transpose wave FFT of pv_wave_fft.h, one after the other through the same scratch),
//              Z[k] = E[k] + W_1024^k O[k], Z[k + 512] = E[k] - W_1024^k O[k]  in registers (decimation in time)
//   inverse  : A[k] = Z[k] + Z[k + 512], B[k] = (Z[k] - Z[k + 512]) e^{+2 pi j k / 1024}  in registers (decimation in frequency),
//              ze = IFFT512(A), zo = IFFT512(B)
//   a lane's raw / output samples of register row r are the 16 contiguous bytes 4 (l + 64 r) .. +3: float4 loads and stores
//
// Everything between the FFTs is the wave kernel's pipeline with 16 bins per lane (pv_wave_kernel.hip has the commentary and the reference
// citations: split pass in conjugate pairs, |X|^2 -> v_max3_u32 peak flags, packed (bin, shift) peak words, select chains + ballot + two
// bpermutes, one route per source bin, plain-store scatter for f >= 1, store-then-add (pairwise frames) or claim rounds for f < 1, fast above-Nyquist residue from the
// spectrum, c2r pre-pass, overlap-add accumulator in registers).
//
// f < 1 frames whose last region reads beyond N/2 + N/8 (possible below f = 0.75) rebuild the above-Nyquist residue quarter by quarter
// (residue_scatter_2k, out of line, rare).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "pv_kernels.h"
#include "pv_device_common.h"
#include "pv_pk_math.h"
#ifndef PV_PT
#define PV_PT 1, 0, 1, 2, 3, 2, 3, 3, 0, 0, 0, 2     // phase priorities of this kernel (pv_wave_fft.h; profiles/r03_priority_sweep.md: C3 3.76 -> 3.31 ms; re-swept in round 5 on the
                                                     // fp32-first kernel: split arithmetic, scatter and c2r pass at level 3 -- C3 3.024 -> 2.991 ms, 2048/128 at f = 0.8 3.189 -> 3.141)
#endif
#include "pv_wave_fft.h"
#include "pv_guard.h"
#ifdef PV_W2K_STAMPS  // measurement build (make variant ... EXTRA=-DPV_W2K_STAMPS CAPI_EXTRA=-DPV_STAMPS=1): s_memtime at the stations of a 1-hop launch, per chain
#define W2K_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); if ((threadIdx.x & 63) == 0 && p.stamps) p.stamps[16 * chain + (i)] = (unsigned)t_; } while (0)
#else
#define W2K_STAMP(i)
#endif
#ifndef PV_PAIRWISE
#define PV_PAIRWISE 1                               // 0: every f < 1 frame goes through the claim rounds (A/B)
#endif

namespace {

constexpr int WAVES2 = 8;                         // frame chains per workgroup = 2 waves per SIMD (<= 256 VGPRs)
constexpr int T2_TW1 = 0;                         // double2[8*64]  W_512^{l k}
constexpr int T2_TW2 = T2_TW1 + 8 * 64 * 16;      // double2[8*8]   W_64^{n0 k}
constexpr int T2_TW1F = T2_TW2 + 8 * 8 * 16;      // float4[4*64]   conj, fp32, rows k = 2j, 2j+1 interleaved
constexpr int T2_TW2F = T2_TW1F + 4 * 64 * 16;    // float4[4*8]
constexpr int T2_HANN = T2_TW2F + 4 * 8 * 16;     // float4[8*64]   0.5 * Hann at samples 4n'..4n'+3, n' = l + 64 r: entry [r*64 + l]
constexpr int T2_ROT = T2_HANN + 8 * 64 * 16;     // float2[16]     exp(+2 pi j q / 16): the rotations of pv:155-170 when the hop is N / 8 or N / 16 (R = 8, 16)
constexpr int T2_BYTES = T2_ROT + 16 * 8;         // 22144

constexpr int MAG0_WORDS = 8;                     // (= MAG0 below)
// per-wave LDS (byte offsets)
constexpr int O2_S = 0;                           // fp64 transpose scratch 9216 B | partner exchange | Y float2[1025] | fp32 transposes | spectrum stash (f < 1)
constexpr int O2_ROUTE = 9216;                    // routes u32 | f32 mags (alias) -- both in the padded layout below, running on into RESQ -- | u16 claim ids (alias) | c2r hand-over (alias)
constexpr int O2_RESQ = O2_ROUTE + 4160;          // float2[512] one quarter of the above-Nyquist residue (general form only)
constexpr int O2_CACHE = O2_RESQ + 1024;          // v4f[3 * 64] rows 0..2 of the NEXT frame's raw window (aliases the quarter buffer behind the KB the padded routes run into)
constexpr int WAVE2_LDS = O2_RESQ + 4096;         // 17472: 22144 + 8 * 17472 = 161920 B per workgroup (<= 160 KB)
static_assert(O2_ROUTE + 4 * (MAG0_WORDS + 1281) <= O2_CACHE, "the padded magnitudes / routes run into the row cache");

constexpr int N2 = 2048, M2 = 1024, H2 = 1025;

// Magnitudes and routes are written bin-major by the FFT's layout (lane l <-> bins l + 64 r) and read / written 16 consecutive bins per lane by the
// peak search: at a lane stride of 64 B the latter are 4- to 8-way bank conflicts.  Both arrays therefore live in a padded layout, every group
// of 16 bins followed by 4 unused words: P(bin) = bin + 4 (bin >> 4).  A lane's 16 bins start at word 20 l (80 B stride: the 16 lanes of a
// b128 pass cover the 64 banks exactly once), and the bin-major side only sees 2-way conflicts on 12 banks.
//   P(l + 64 r) = pl + 80 r, pl = l + 4 (l >> 4);   P(1024 - l - 64 r) = 1280 - ql - 80 r, ql = l + 4 ((l + 15) >> 4);   P(512) = 640, P(1024) = 1280
constexpr int MAG0 = 8;                           // magnitudes start 8 words in: bins -2, -1 of lane 0's window stay inside the region

// o * W_32^r = o * exp(-2 pi j r / 32), r = 0..7 (compile-time), fp64: the wave-uniform part of the split-pass twiddle
__device__ __forceinline__ double2 mul_w32(double2 o, int r)
{
    constexpr double c[8] = {1.0, 0.98078528040323044913, 0.92387953251128675613, 0.83146961230254523708, 0.70710678118654752440,
                             0.55557023301960222474, 0.38268343236508977173, 0.19509032201612826785};
    if (r == 0) return o;
    return cmul(o, double2{c[r], -c[8 - r]});      // sin(2 pi r / 32) = cos(2 pi (8 - r) / 32)
}
// o * exp(+2 pi j r / 32), packed fp32 (c2r twiddle)
__device__ __forceinline__ pk::c32 mul_w32_inv_pk(pk::c32 o, int r)
{
    constexpr float c[9] = {1.0f, 0.98078528040323044913f, 0.92387953251128675613f, 0.83146961230254523708f, 0.70710678118654752440f,
                            0.55557023301960222474f, 0.38268343236508977173f, 0.19509032201612826785f, 0.0f};
    if (r == 0) return o;
    return pk::cmul(o, pk::c32{c[r], c[8 - r]});
}
// o * exp(+2 pi j r / 16), r = 0..7, packed fp32 (first stage of the decimation-in-frequency inverse)
__device__ __forceinline__ pk::c32 mul_w16_inv_pk8(pk::c32 o, int r)
{
    constexpr float c = 0.92387953251128675613f, s = 0.38268343236508977173f, h = 0.70710678118654752440f;
    switch (r) {
    case 0: return o;
    case 1: return pk::cmul(o, pk::c32{c, s});
    case 2: return pk::cmul(o, pk::c32{h, h});
    case 3: return pk::cmul(o, pk::c32{s, c});
    case 4: return pk::c32{-o.y, o.x};            // * j
    case 5: return pk::cmul(o, pk::c32{-s, c});
    case 6: return pk::cmul(o, pk::c32{-h, h});
    default: return pk::cmul(o, pk::c32{-c, s});
    }
}

// claim rounds of one wave (see pv_wave_kernel.hip): every pending source posts its id on the claim word of its target, the id that sticks
// does a plain read-modify-write, losers retry.  LDS traffic of a wave executes in order, so the result is deterministic.
// (Measured and rejected: two ds_add_f32 per source instead of the rounds -- an LDS float atomic costs ~180 CU-cycles per wave instruction on
// gfx950, the f = 0.8 frame went from 3.8 to 9.1 ms per launch.)
// FRESH: Y is all zeros on entry (the frame's first scatter): the winners of the first round store their value without reading Y.
template <int NS, bool FRESH = false>
__device__ __forceinline__ void claim_rounds2(const unsigned (&rt)[NS], const float2 (&ys)[NS], const int (&id)[NS], float2 *Y, unsigned short *CLAIM)
{
    unsigned pend = 0;
    unsigned tg[NS];
#pragma unroll
    for (int r = 0; r < NS; r++) {
        const unsigned t = rt[r] & 0xFFFFu;
        const bool ok = t < (unsigned)H2;
        pend |= ok ? (1u << r) : 0u;
        tg[r] = ok ? t : 0u;
    }
    if (FRESH) {
#pragma unroll
        for (int r = 0; r < NS; r++) if (pend & (1u << r)) CLAIM[tg[r]] = (unsigned short)id[r];
        wave_sync();
        unsigned short c[NS];
#pragma unroll
        for (int r = 0; r < NS; r++) c[r] = CLAIM[tg[r]];
#pragma unroll
        for (int r = 0; r < NS; r++) {
            if ((pend & (1u << r)) && c[r] == (unsigned short)id[r]) {
                Y[tg[r]] = ys[r];
                pend &= ~(1u << r);
            }
        }
        wave_sync();
    }
    while (__any(pend != 0u)) {
#pragma unroll
        for (int r = 0; r < NS; r++) if (pend & (1u << r)) CLAIM[tg[r]] = (unsigned short)id[r];
        wave_sync();
        unsigned short c[NS];
        float2 o[NS];
#pragma unroll
        for (int r = 0; r < NS; r++) c[r] = CLAIM[tg[r]];
#pragma unroll
        for (int r = 0; r < NS; r++) o[r] = Y[tg[r]];
#pragma unroll
        for (int r = 0; r < NS; r++) {
            if ((pend & (1u << r)) && c[r] == (unsigned short)id[r]) {
                Y[tg[r]] = float2{o[r].x + ys[r].x, o[r].y + ys[r].y};
                pend &= ~(1u << r);
            }
        }
        wave_sync();
    }
}

// Rotation exp(+2 pi j ridx / N) of one source value (pv:155-170): ridx = (delta * t) mod N is a multiple of N / R.  R = 4 is rotate_route's
// swap-and-sign form; R = 8 and 16 read the root from a 16-entry LDS table (a global table load would sit on the critical path of every frame).
template <int R_>
__device__ __forceinline__ float2 rotate2k(unsigned route, float2 v, const float2 *ROT)
{
    if (R_ == 1) return v;
    if (R_ == 2) {
        const unsigned sg = (route << 5) & 0x80000000u;                     // top bit of the 11-bit rotation index = bit 26 of the route
        return float2{__uint_as_float(__float_as_uint(v.x) ^ sg), __uint_as_float(__float_as_uint(v.y) ^ sg)};
    }
    if (R_ == 4) return rotate_route<4, 11>(route, v, nullptr);
    return cmul(v, ROT[(route >> 23) & 15u]);                              // bits 23..26 of the route = top four bits of the rotation index
}

__device__ __forceinline__ int digitrev4_2k(int v, int nd)
{
    const unsigned r = __brev((unsigned)v) >> (32 - 2 * nd);
    return (int)(((r & 0x55555555u) << 1) | ((r >> 1) & 0x55555555u));
}

// Rare path (f < 1 frames whose last region reads beyond position N/2 + N/8): what fft.js's in-place real DIT leaves at positions
// N/2+1 .. N-1, one quarter of the buffer at a time (quarter 2 = sub-FFT of xw[4n+2], positions 1024..1535; quarter 3 = xw[4n+3], 1536..2047),
// by re-running the reference's stage structure on that quarter in fp32 (log2 N odd: radix-2 base blocks, bundle:447-463, then the radix-4
// stages with their predicated stores, bundle:329-441) -- and its sources, all owned by the last peak (pv:133), added into Y.  One wave, so
// the stages are separated by wave_sync only; out of line so that its registers and its global loads stay out of the main loop.
template <int R_>
__device__ __attribute__((noinline)) PV_NO_DS_MERGE void residue_scatter_2k(const float *in, const float *hist, int hist_len, bool sys, long s0, const float *__restrict__ hann,
                                                                             const float2 *__restrict__ tw32, unsigned wave_off, int l, int upper_end, int up_delta,
                                                                             unsigned up_ridx, double *dbg_X)
{
    constexpr int N = N2, H = H2, QN = N / 4, LOG2N = 11;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
    float2 *Y = reinterpret_cast<float2 *>(smem_all + wave_off + O2_S);
    unsigned short *CLAIM = reinterpret_cast<unsigned short *>(smem_all + wave_off + O2_ROUTE);
    float2 *Q = reinterpret_cast<float2 *>(smem_all + wave_off + O2_RESQ);
    const double2 *TW1R = reinterpret_cast<const double2 *>(smem_all + T2_TW1);   // shared table of the workgroup: entry 64 k + l = W_512^{l k}
    const WaveSrc src{in, hist, hist_len, sys};
    for (int base = N / 2; base < N && base < upper_end; base += QN) {
#pragma unroll
        for (int i = 0; i < 4; i++) {                                      // QN / 2 = 256 radix-2 blocks per quarter; input index = base-4 digit reversal of the block
            const int lb = l + 64 * i, blk = base / 2 + lb;
            const int off = digitrev4_2k(blk, (LOG2N - 1) / 2);
            const float a = mul_rounded(src.at(s0 + off), hann[off]), b = mul_rounded(src.at(s0 + off + N / 2), hann[off + N / 2]);
            Q[2 * lb] = float2{a + b, 0.f};
            Q[2 * lb + 1] = float2{a - b, 0.f};
        }
        wave_sync();
        for (int log2m = 3; log2m <= LOG2N - 2; log2m += 2) {              // block sizes 8, 32, 128, 512 inside the quarter
            const int q = (1 << log2m) >> 2, hq = q >> 1;
            const int nblocks = QN >> log2m;
            const int tws = LOG2N - log2m;
            for (int t = l; t < nblocks * (hq + 1); t += 64) {             // the butterflies of a stage touch disjoint elements: any order
                int blk, i;
                if (t < nblocks * hq) { blk = t / hq; i = t - blk * hq; } else { blk = t - nblocks * hq; i = hq; }
                const int o = blk << log2m;
                const float2 A = Q[o + i];
                // W^e, W^{2e}, W^{3e}, e = i << tws (a multiple of 4, <= N/8): W^e from row 1 of the forward FFT's fp64 table in LDS (W_512^l = W_N^{4l}), its
                // square and cube formed here, instead of three loads from the global table per butterfly (see pv_wg_kernel.hip)
                float2 w1;
                if (i == hq) w1 = float2{0.70710678118654752440f, -0.70710678118654752440f};   // e = N/8: W_8 (one past the table row)
                else { const double2 wd = TW1R[64 + (i << (tws - 2))]; w1 = float2{(float)wd.x, (float)wd.y}; }
                const float2 w2 = cmul(w1, w1), w3 = cmul(w2, w1);
                const float2 Bv = cmul(Q[o + q + i], w1);
                const float2 Cc = cmul(Q[o + 2 * q + i], w2);
                const float2 D = cmul(Q[o + 3 * q + i], w3);
                const float2 T0 = cadd(A, Cc), T1 = csub(A, Cc), T2 = cadd(Bv, D), T3 = csub(Bv, D);
                Q[o + i] = cadd(T0, T2);
                Q[o + q + i] = float2{T1.x + T3.y, T1.y - T3.x};
                if (i == 0) {
                    Q[o + 2 * q] = csub(T0, T2);                          // bundle:400-406
                } else if (i != hq) {                                     // bundle:409-440
                    Q[o + q - i] = float2{T1.x - T3.y, -(T1.y + T3.x)};
                    Q[o + 2 * q - i] = float2{T0.x - T2.x, -(T0.y - T2.y)};
                }
            }
            wave_sync();
        }
        if (dbg_X)
            for (int i = l; i < QN; i += 64) if (base + i >= H) { dbg_X[2 * (base + i)] = Q[i].x; dbg_X[2 * (base + i) + 1] = Q[i].y; }
        unsigned rt[8];
        float2 ys[8];
        int id[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int b = base + l + 64 * j, tgt = b + up_delta;
            rt[j] = (b >= H && b < upper_end && tgt >= 0 && tgt < H) ? ((up_ridx << 16) | (unsigned)tgt) : NOROUTE;
            ys[j] = rotate2k<R_>(rt[j], Q[l + 64 * j], reinterpret_cast<const float2 *>(smem_all + T2_ROT));
            id[j] = b;
        }
        claim_rounds2<8>(rt, ys, id, Y, CLAIM);
    }
}

// HOPQ = hop / 128.  HOPQ >= 2: the hop is S_ROWS whole rows of 256 samples, the accumulator slides by renaming registers.  HOPQ = 1 (the reference's
// shipped 2048 / 128, R = 16): the hop is HALF a row = 32 lanes.  Instead of moving the accumulator across lanes the kernel alternates two register
// layouts: on even frames of a chain lane L holds samples 4L.. of its row (as always), on odd frames samples 4(L ^ 32)..; the slide is then a
// lane-wise select between neighbouring rows, and the synthesis side of an odd frame (c2r, inverse FFTs, window) simply runs with the lane id
// L ^ 32 -- every exchange there goes through LDS addresses, so relabelling the lanes costs nothing.
// AUX = true: test-tap instance (pv_debug_frame: X / |X|^2 / peak flags / Y of one frame, incl. the above-Nyquist residue); the production
// instance carries no tap code.
// RESIDENT = true: streaming instance that stays on the GPU (PV_FLAG_PERSISTENT_STREAM; see pv_wave_kernel.hip): one wave per channel slot, 1 hop per quantum.
// ---- the two forward transforms of one 2048-point frame: one text for the kernel's inline code and for the out-of-line copy of the F32 instances' rare paths ----
// Reference width: Hann (pv:55), pack by parity, two 512-point fp64 FFTs, decimation-in-time stage, split pass in conjugate pairs: k = l + 64 r pairs with M - k =
// element 512 + (64 - l) + 64 (7 - r), i.e. zhi[7 - r] of lane 64 - l.  emit(r, X[l + 64 r], X[1024 - l - 64 r]), r < 8; lane 0 also emit512(X[512]).
template <typename EMIT, typename EMIT512>
__device__ __forceinline__ void spectrum64_2k(const v4f (&raw)[8], const v4f (&hw)[8], unsigned char *smem, int l, double2 w1024, double2 w2048, EMIT emit, EMIT512 emit512)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
    const double2 *TW1 = reinterpret_cast<const double2 *>(smem_all + T2_TW1);
    const double2 *TW2 = reinterpret_cast<const double2 *>(smem_all + T2_TW2);
    double2 *S64 = reinterpret_cast<double2 *>(smem + O2_S);
    pv_prio(PH_FA);
    double2 zlo[8], zhi[8];
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const v4f xw = raw[r] * hw[r];
        zlo[r] = double2{(double)xw.x, (double)xw.y};                      // ze[l + 64 r] = z[2 n']
        zhi[r] = double2{(double)xw.z, (double)xw.w};                      // zo[l + 64 r] = z[2 n' + 1]
    }
    fft512_wave<double, false>(zlo, S64, TW1, TW2, l);
    fft512_wave<double, false>(zhi, S64, TW1, TW2, l);
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const double2 t = cmul(w1024, mul_w16<double, false>(zhi[r], r));
        const double2 e = zlo[r];
        zlo[r] = cadd(e, t);                                               // Z[l + 64 r]
        zhi[r] = csub(e, t);                                               // Z[l + 64 r + 512]
    }
    pv_prio(PH_SPLITX);
#pragma unroll
    for (int r = 0; r < 8; r++) S64[r * 64 + l] = zhi[r];
    wave_sync();
    pv_prio(PH_SPLITM);
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const double2 zm = S64[(7 - r) * 64 + 64 - l];                     // (l = 0, r = 0) reads one element past the rows: replaced below
        const double2 E{zlo[r].x + zm.x, zlo[r].y - zm.y};
        const double2 O{zlo[r].x - zm.x, zlo[r].y + zm.y};
        const double2 WO = cmul(w2048, mul_w32(O, r));
        double2 xa{E.x + WO.y, E.y - WO.x};
        double2 xb{E.x - WO.y, -(E.y + WO.x)};
        if (r == 0 && l == 0) {                                            // Z[0] is pre-halved: X[0] = 2(zr + zi), X[1024] = 2(zr - zi), both real
            xa = double2{2.0 * (zlo[0].x + zlo[0].y), 0.0};
            xb = double2{2.0 * (zlo[0].x - zlo[0].y), 0.0};
        }
        emit(r, xa, xb);
    }
    if (l == 0) emit512(double2{2.0 * zhi[0].x, -2.0 * zhi[0].y});         // k = 512 pairs with itself: X = 2 conj(Z[512])
}

// fp32 first (round 5; pv_wave_kernel.hip, spectrum32_1024): the same in PACKED fp32 on conjugated data -- FFT(z) = conj(IFFT(conj z)), the first conjugation folded into
// the window product, the second into the split pass.  wl1024f = conj(W_1024^l) and wl2048s = SC conj(W_2048^l) are the twiddles of the kernel's inverse side (SC a
// power of two, taken out again by isc = 1 / SC inside the FMAs: exact).
template <bool REGT1, typename EMIT, typename EMIT512>
__device__ __forceinline__ void spectrum32_2k(const v4f (&raw)[8], const v4f (&hw)[8], unsigned char *smem, int l, pk::c32 wl1024f, pk::c32 wl2048s, float isc_, EMIT emit, EMIT512 emit512)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
    const v4f *TW1F4 = reinterpret_cast<const v4f *>(smem_all + T2_TW1F);
    const v4f *TW2F4 = reinterpret_cast<const v4f *>(smem_all + T2_TW2F);
    pk::c32 *S = reinterpret_cast<pk::c32 *>(smem + O2_S);
    pv_prio(PH_FA);
    pk::c32 zlo[8], zhi[8];
#pragma unroll
    for (int r = 0; r < 8; r++) {
        zlo[r] = pk::mul_conj(pk::c32{raw[r].x, raw[r].y}, pk::c32{hw[r].x, hw[r].y});      // conj(ze[l + 64 r])
        zhi[r] = pk::mul_conj(pk::c32{raw[r].z, raw[r].w}, pk::c32{hw[r].z, hw[r].w});      // conj(zo[l + 64 r])
    }
    fft512_wave_inv_pk<REGT1, NoStamp, true>(zlo, S, TW1F4, TW2F4, l);
    fft512_wave_inv_pk<REGT1, NoStamp, true>(zhi, S, TW1F4, TW2F4, l);
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const pk::c32 t = pk::cmul(mul_w16_inv_pk8(zhi[r], r), wl1024f);   // conj(W_1024^{l + 64 r}) conj(O[k])
        const pk::c32 e = zlo[r];
        zlo[r] = pk::add(e, t);                                            // conj(Z[l + 64 r])
        zhi[r] = pk::sub(e, t);                                            // conj(Z[l + 64 r + 512])
    }
    pv_prio(PH_SPLITX);
#pragma unroll
    for (int r = 0; r < 8; r++) S[r * 64 + l] = zhi[r];
    wave_sync();
    pv_prio(PH_SPLITM);
    const pk::c32 isc{isc_, isc_};
    pk::c32 zm[8];                                                         // every partner value first (pv_wave_kernel.hip: a read behind each magnitude store serialises the pairs)
#pragma unroll
    for (int r = 0; r < 8; r++) zm[r] = S[(7 - r) * 64 + 64 - l];
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const pk::c32 E = pk::add_conj(zlo[r], zm[r]), O = pk::sub_conj(zlo[r], zm[r]);
        const pk::c32 T = pk::cmul(mul_w32_inv_pk(O, r), wl2048s);         // SC conj(W_2048^{l + 64 r}) O'
        pk::c32 xa = pk::conj_fma_j(T, isc, E), xb = pk::fnma_j(T, isc, E);
        if (r == 0 && l == 0) {
            xa = pk::c32{2.0f * (zlo[0].x - zlo[0].y), 0.f};
            xb = pk::c32{2.0f * (zlo[0].x + zlo[0].y), 0.f};
        }
        emit(r, xa, xb);
    }
    if (l == 0) emit512(pk::c32{2.0f * zhi[0].x, 2.0f * zhi[0].y});        // X[512] = 2 conj(Z[512])
}

#ifndef PV_F32_REGT1_2K
#define PV_F32_REGT1_2K true
#endif
// The forward transforms of the F32 instances' rare paths, out of line (pv_wave_kernel.hip, forward_cold_1024): a frame of class B (fp64), and the fp32 attempt of a
// chain that runs the fp64 transform first.  The window is read again from memory; |X|^2 -> MAG (padded layout), the fp32 spectrum -> the stash XS in the scratch, where
// f < 1 frames put it anyway: the caller takes XA / XB / x512f back from there.  Returns the fp32 transform's K (wide: 0).
__device__ __attribute__((noinline)) PV_NO_DS_MERGE float forward_cold_2k(int wide, const float *in, const float *hist, int hist_len, int sys, int vec_in, long s0u,
                                                                         double2 w1024, double2 w2048, pk::c32 wl1024f, pk::c32 wl2048s, float isc, unsigned wave_off, int l)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
    unsigned char *smem = smem_all + wave_off;
    float *MAG = reinterpret_cast<float *>(smem + O2_ROUTE);
    float2 *XS = reinterpret_cast<float2 *>(smem + O2_S);
    const v4f *HW4 = reinterpret_cast<const v4f *>(smem_all + T2_HANN);
    const WaveSrc src{in, hist, hist_len, sys != 0};
    v4f raw[8], hw[8];
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const long sx = s0u + 4 * l + 256 * r;                             // a multiple of 4: the four samples never straddle history / input
        const float *q = sx < 0 ? hist + sx + hist_len : in + sx;
        if ((sys && sx >= 0) || !vec_in) raw[r] = v4f{src.at(sx), src.at(sx + 1), src.at(sx + 2), src.at(sx + 3)};
        else raw[r] = *reinterpret_cast<const v4f *>(q);
        hw[r] = HW4[r * 64 + l];
    }
    const int pl = l + 4 * (l >> 4), ql = l + 4 * ((l + 15) >> 4);
    float2 XA[8], XB[8], x512f{0.f, 0.f};
    float guard_k = 0.f;
    if (wide) {
        spectrum64_2k(raw, hw, smem, l, w1024, w2048,
                      [&](int r, double2 xa, double2 xb) {
                          MAG[MAG0 + pl + 80 * r] = (float)(xa.x * xa.x + xa.y * xa.y);
                          MAG[MAG0 + 1280 - ql - 80 * r] = (float)(xb.x * xb.x + xb.y * xb.y);
                          XA[r] = float2{(float)xa.x, (float)xa.y};
                          XB[r] = float2{(float)xb.x, (float)xb.y};
                      },
                      [&](double2 x512) { MAG[MAG0 + 640] = (float)(x512.x * x512.x + x512.y * x512.y); x512f = float2{(float)x512.x, (float)x512.y}; });
    } else {
        unsigned mmax = 0u;
        spectrum32_2k<PV_F32_REGT1_2K>(raw, hw, smem, l, wl1024f, wl2048s, isc,
                                       [&](int r, pk::c32 xa, pk::c32 xb) {
                                           const float ma = mag32(xa), mb = mag32(xb);
                                           MAG[MAG0 + pl + 80 * r] = ma;
                                           MAG[MAG0 + 1280 - ql - 80 * r] = mb;
                                           mmax = max(max(mmax, __float_as_uint(ma)), __float_as_uint(mb));
                                           XA[r] = float2{xa.x, xa.y};
                                           XB[r] = float2{xb.x, xb.y};
                                       },
                                       [&](pk::c32 x512) { const float m = mag32(x512); MAG[MAG0 + 640] = m; mmax = max(mmax, __float_as_uint(m)); x512f = float2{x512.x, x512.y}; });
        guard_k = guard_k_of(wave_max_u32(mmax));
    }
    wave_sync();                                                           // the partner rows of the split pass have been read: the stash may overwrite them
#pragma unroll
    for (int r = 0; r < 8; r++) { XS[l + 64 * r] = XA[r]; XS[1024 - l - 64 * r] = XB[r]; }
    if (l == 0) XS[512] = x512f;
    return guard_k;
}

template <int HOPQ, bool AUX, bool RESIDENT = false, bool F32 = false>
__global__ __launch_bounds__(RESIDENT ? 128 : 64 * WAVES2, RESIDENT ? 1 : 2) PV_NO_DS_MERGE void pv_wave2k_kernel(const PvKernelParams p)   // (resident: 2-wave workgroups, one wave per SIMD, the whole register file)
{
    static_assert(!F32 || !AUX, "the fp32-first forward transform is a product path, not the tap instance");
    constexpr int N = N2, M = M2, H = H2;
    constexpr bool HALF = (HOPQ == 1);
    constexpr int S_ROWS = HOPQ / 2, HOP = 128 * HOPQ, R = N / HOP, LROWS = HALF ? 8 : 8 - S_ROWS, L = N - HOP;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long chain = (long)blockIdx.x * (blockDim.x >> 6) + wv;        // a small launch runs fewer waves per workgroup (launch2k)
    const int ch = (int)(chain / p.nchunks), chunk = (int)(chain - (long)ch * p.nchunks);

    W2K_STAMP(0);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
    const v4f *TW1F4 = reinterpret_cast<const v4f *>(smem_all + T2_TW1F);
    const v4f *TW2F4 = reinterpret_cast<const v4f *>(smem_all + T2_TW2F);
    const v4f *HW4 = reinterpret_cast<const v4f *>(smem_all + T2_HANN);
    const float2 *ROT = reinterpret_cast<const float2 *>(smem_all + T2_ROT);
    {
        double2 *t1 = reinterpret_cast<double2 *>(smem_all + T2_TW1);
        double2 *t2 = reinterpret_cast<double2 *>(smem_all + T2_TW2);
        float2 *t1f = reinterpret_cast<float2 *>(smem_all + T2_TW1F);
        float2 *t2f = reinterpret_cast<float2 *>(smem_all + T2_TW2F);
        v4f *hh = reinterpret_cast<v4f *>(smem_all + T2_HANN);
        for (int i = threadIdx.x; i < 512; i += blockDim.x) {
            const int k = i >> 6, ln = i & 63;
            const double2 w = p.tw64[(4 * ln * k) & (N - 1)];              // W_512^{ln k} = exp(-2 pi j ln k 4 / 2048)
            t1[i] = w;
            t1f[2 * ((k >> 1) * 64 + ln) + (k & 1)] = float2{(float)w.x, -(float)w.y};
            hh[i] = v4f{0.5f * p.hann[4 * i], 0.5f * p.hann[4 * i + 1], 0.5f * p.hann[4 * i + 2], 0.5f * p.hann[4 * i + 3]};   // n' = i = ln + 64 k: row k
            if (i < 16) reinterpret_cast<float2 *>(smem_all + T2_ROT)[i] = cconj(p.tw32[(i * (N / 16)) & (N - 1)]);
            if (i < 64) {
                const int k2 = i >> 3, n0 = i & 7;
                const double2 w2 = p.tw64[(32 * n0 * k2) & (N - 1)];       // W_64^{n0 k2}
                t2[i] = w2;
                t2f[2 * ((k2 >> 1) * 8 + n0) + (k2 & 1)] = float2{(float)w2.x, -(float)w2.y};
            }
        }
    }
    __syncthreads();                                                     // the only workgroup-wide barrier
    if (ch >= p.nch) return;
    W2K_STAMP(1);
    // what changes from quantum to quantum in the resident form (constants of the launch otherwise)
    const float *hist_in = p.hist_in, *acc_in = p.acc_in;
    float *hist_out = p.hist_out, *acc_out = p.acc_out;
    int t0_mod_n = p.t0_mod_n;
    unsigned done_seq = p.done_seq;
    unsigned last_seq = p.done_seq;                                      // resident: the last quantum completed before this launch
    [[maybe_unused]] unsigned pred = 0;                                  // F32: the order counter of this chain (pv_guard.h); a resident wave's quanta are ONE chain of a stream
resident_top:
    if (RESIDENT) {
        // the control word of pv_wave_kernel_1024's resident form: sequence number (16 bits, never 0) | channel count (7) | ping-pong half (1) | timeCursor / hop mod R (8)
        unsigned word;
        const unsigned long long idle0 = wall_clock64();
        for (;;) {
            word = __hip_atomic_load(p.ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if ((word & 0xFFFFu) != (last_seq & 0xFFFFu)) break;
            if (__hip_atomic_load(p.ctl + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u || wall_clock64() - idle0 > (unsigned long long)p.idle_ticks) return;   // asked to leave, or ~50 ms without work
            __builtin_amdgcn_s_sleep(2);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
        const unsigned seq = word & 0xFFFFu, nch_now = (word >> 16) & 0x7Fu, cur = (word >> 23) & 1u;
        t0_mod_n = (int)(((word >> 24) & 0xFFu) * HOP) & (N - 1);
        hist_in = p.hist2[cur]; hist_out = p.hist2[cur ^ 1u];
        acc_in = p.acc2[cur]; acc_out = p.acc2[cur ^ 1u];
        done_seq = last_seq = seq;
        if ((unsigned)ch >= nch_now) {                                   // a slot outside this quantum: carry its state across the ping-pong flip
            if ((unsigned)ch < __hip_atomic_load(p.ctl + 5, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM))
                for (int j = lane; j < L; j += 64) { hist_out[(long)ch * L + j] = hist_in[(long)ch * L + j]; acc_out[(long)ch * L + j] = acc_in[(long)ch * L + j]; }
            goto resident_top;
        }
    }

    const unsigned wave_off = T2_BYTES + wv * WAVE2_LDS;
    unsigned char *smem = smem_all + wave_off;
    float2 *Y = reinterpret_cast<float2 *>(smem + O2_S);
    float *MAG = reinterpret_cast<float *>(smem + O2_ROUTE);
    unsigned *ROUTE = reinterpret_cast<unsigned *>(smem + O2_ROUTE);
    unsigned short *CLAIM = reinterpret_cast<unsigned short *>(smem + O2_ROUTE);
    v4u dq0{0u, 0u, 0u, 0u}, dq1{0u, 0u, 0u, 0u};                       // shifts of this lane's own 16 candidate bins (i16 each)
    unsigned psh_key = 0u;
    bool psh_valid = false;

    const int first_out = chunk * p.frames_per_chunk;
    int last_out = first_out + p.frames_per_chunk;
    if (last_out > p.nhops) last_out = p.nhops;
    int first_frame = first_out - (R - 1);
    const bool from_state = (first_frame <= 0);
    if (from_state) first_frame = 0;

    const long cbase = (long)ch * p.ch_stride;
    const WaveSrc src{p.in + cbase, hist_in + (long)ch * L, L, RESIDENT && p.in_cached != 0};
    float *outp = p.out + cbase;
    const bool vec_out = (reinterpret_cast<uintptr_t>(outp) & 15u) == 0;
    const bool vec_in = ((reinterpret_cast<uintptr_t>(src.in) | reinterpret_cast<uintptr_t>(src.hist)) & 15u) == 0;
    const float *pitch_row = p.pitch + (p.pitch_stride ? (long)(ch / p.ch_per_stream) * p.pitch_stride : 0);

    const double2 w1024 = p.tw64[2 * lane];                               // W_1024^l: decimation-in-time twiddle W_1024^{l + 64 r} = w1024 * W_16^r
    const double2 w2048 = p.tw64[lane];                                   // W_2048^l: split-pass twiddle W_2048^{l + 64 r} = w2048 * W_32^r
    constexpr float SC = 2.0f / ((float)N * (float)R);                    // 1/N of the inverse, 1/R of the overlap-add, 2 for the shared 0.5 * Hann table (exact)
    const float2 c2048 = cconj(p.tw32[lane]), c1024 = cconj(p.tw32[2 * lane]);
    const float2 s2w0 = cconj(p.tw32[2 * (1 + lane)]);                    // fast residue (f < 1): conj(W^{2k}) of this lane's first bin k = 1 + lane
    const pk::c32 wl2048s{c2048.x * SC, c2048.y * SC};                    // c2r twiddle e^{+2 pi j l / 2048}, scale folded in
    const pk::c32 wl1024f{c1024.x, c1024.y};                              // DIF twiddle e^{+2 pi j l / 1024}
    pk::c32 wl2048s_x = wl2048s, wl1024f_x = wl1024f;                     // the same for the lane id l ^ 32 (odd frames of the half-row hop)
    if (HALF) {
        const float2 d2048 = cconj(p.tw32[lane ^ 32]), d1024 = cconj(p.tw32[2 * (lane ^ 32)]);
        wl2048s_x = pk::c32{d2048.x * SC, d2048.y * SC};
        wl1024f_x = pk::c32{d1024.x, d1024.y};
    }

    v4f acc[8];
#pragma unroll
    for (int r = 0; r < 8; r++) acc[r] = v4f{0.f, 0.f, 0.f, 0.f};
    if (from_state) {
#pragma unroll
        for (int r = 0; r < LROWS; r++) {
            const float *a = acc_in + (long)ch * L + 4 * lane + 256 * r;
            if (4 * lane + 256 * r < L) acc[r] = v4f{a[0], a[1], a[2], a[3]};
        }
    }
    // A frame that lies inside the input takes wave-uniform row bases plus ONE lane offset (no per-row 64-bit address arithmetic); only the first
    // R - 1 frames of a stream reach back into the carried history and pick a pointer per row.
    auto ld_sys4 = [&](const float *q) -> v4f {                           // resident form: the host rewrites the hop between quanta of ONE launch: never from a cache
        const unsigned long long a = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(q), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const unsigned long long b = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(q) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return v4f{__uint_as_float((unsigned)a), __uint_as_float((unsigned)(a >> 32)), __uint_as_float((unsigned)b), __uint_as_float((unsigned)(b >> 32))};
    };
    auto load_rows = [&](v4f *w, int frame) {
        const long s0u = (long)(frame + 1) * HOP - N;                       // wave-uniform
        if (!(RESIDENT && src.sys) && HALF && s0u >= 0 && vec_in) {                                   // (hop 128 only: -3 % there; at hop 512 the f < 1 path lost 3 % to the changed register allocation)
            const unsigned ob = 16u * (unsigned)lane;
#pragma unroll
            for (int r = 0; r < 8; r++) w[r] = *reinterpret_cast<const v4f *>(reinterpret_cast<const char *>(src.in + s0u + 256 * r) + ob);
        } else {
            const long s0 = s0u + 4 * lane;
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const long sx = s0 + 256 * r;                               // a multiple of 4: the four samples never straddle history / input
                const float *q = sx < 0 ? src.hist + sx + src.hist_len : src.in + sx;
                if (RESIDENT && src.sys && sx >= 0) w[r] = vec_in ? ld_sys4(q) : v4f{src.at(sx), src.at(sx + 1), src.at(sx + 2), src.at(sx + 3)};
                else if (vec_in) w[r] = *reinterpret_cast<const v4f *>(q);
                else w[r] = v4f{q[0], q[1], q[2], q[3]};
            }
        }
    };
    // the same for a frame whose rows 0..2 sit in the LDS row cache: only the others are fetched
    constexpr bool ROWCACHE = !RESIDENT && HOP + 768 <= N;                // (hops of 2048 / 1024 leave less than three old rows: hop 1024 keeps rows 4..6)
    auto load_rows_except = [&](v4f *w, int frame) {
        const long s0 = (long)(frame + 1) * HOP - N + 4 * lane;
#pragma unroll
        for (int r = 3; r < 8; r++) {
            const long sx = s0 + 256 * r;
            const float *q = sx < 0 ? src.hist + sx + src.hist_len : src.in + sx;
            if (vec_in) w[r] = *reinterpret_cast<const v4f *>(q);
            else w[r] = v4f{q[0], q[1], q[2], q[3]};
        }
    };
    bool cache_ok = false;
    v4f raw[8];
    load_rows(raw, first_frame);
    float pf_next = (RESIDENT && src.sys) ? __hip_atomic_load(pitch_row + first_frame, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : pitch_row[first_frame];
    int emit_v = first_out;
    asm volatile("" : "+v"(emit_v));
    v4f hw[8];
#pragma unroll
    for (int r = 0; r < 8; r++) hw[r] = HW4[r * 64 + lane];
    W2K_STAMP(2);
#ifdef PV_W2K_STAMPS
    { float sink = 0.f; for (int r = 0; r < 8; r++) sink += raw[r].x + acc[r].x; asm volatile("" :: "v"(sink)); }
    W2K_STAMP(3);
#endif

    [[maybe_unused]] unsigned n_fallback = 0;                            // F32: class-B frames of this chain
#ifdef PV_FLIP_COUNT
    unsigned n_flip = 0, n_uncaught = 0, n_sure = 0, n_incons = 0;
#endif
    for (int m = first_frame; m < last_out; ++m) {
        const bool dbg = AUX && (p.dbg_mag != nullptr) && ch == p.dbg_ch && m == p.dbg_frame;
        int l = lane;
        asm volatile("" : "+v"(l));                                       // LDS addresses are recomputed per frame instead of hoisted (see pv_wg_kernel.hip)
        const float pfm = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(pf_next)));
        const double pf = (double)pfm;
        const int tmod = (int)(((long)t0_mod_n + (long)m * HOP) & (N - 1));
        const int pl = l + 4 * (l >> 4), ql = l + 4 * ((l + 15) >> 4);     // padded positions of the bins this lane's FFT registers hold (see P above)
        const int par = HALF ? ((m - first_frame) & 1) : 0;                // accumulator layout of this frame (wave-uniform)
        const int li = l ^ (par << 5);                                    // lane id of the synthesis side
        if (HALF) {
#pragma unroll
            for (int r = 0; r < 8; r++) hw[r] = HW4[r * 64 + l];            // the analysis window is always in the natural layout
        }

        float2 XA[8], XB[8];                                               // X[l + 64 r], X[1024 - l - 64 r] in fp32: what the shift needs after the decisions
        float2 x512f{0.f, 0.f};
        if (ROWCACHE) {
            // 768 samples of the window stay in LDS for the next frame (round 4) -- the OLDEST ones the next frame still needs, its rows 0..2: the registers
            // cannot carry them through the frame (the forward FFTs need every one), and re-reading the whole 8 KB window per frame left the oldest rows
            // to L2, where the streaming traffic of 512 chains per XCD (new rows in, results out) had evicted a third of them by the time of their last
            // use (HBM traffic 1.29x algorithmic on BASELINE configs[2], 1.33x on 2048 / 128).  The quarter buffer of the general residue (behind its
            // first KB, which the padded magnitudes / routes run into) is free in all but the rare frames that take that path: they invalidate the cache.
            // Sample s of this window (HOP <= s < HOP + 768) is sample s - HOP of the next one: byte 4 (s - HOP).
#pragma unroll
            for (int r = 0; r < 8; r++) {
                if (256 * r + 256 <= HOP || 256 * r >= HOP + 768) continue;
                const int s = 256 * r + 4 * l;
                if ((256 * r >= HOP && 256 * r + 256 <= HOP + 768) || (s >= HOP && s < HOP + 768))      // (whole rows: no lane test)
                    *reinterpret_cast<v4f *>(smem + O2_CACHE + 4 * (s - HOP)) = raw[r];
            }
            cache_ok = true;
        }
        // ---- forward transform at the reference's width (spectrum64_2k): |X|^2 -> MAG, the spectrum rounded to fp32 -> XA / XB.  The only forward transform of the
        //      !F32 instances; F32 instances run it out of line (forward_cold_2k) for the frames whose decisions the fp32 transform cannot carry ----
        [[maybe_unused]] auto forward64 = [&]() {
            spectrum64_2k(raw, hw, smem, l, w1024, w2048,
                          [&](int r, double2 xa, double2 xb) {
                              MAG[MAG0 + pl + 80 * r] = (float)(xa.x * xa.x + xa.y * xa.y);
                              MAG[MAG0 + 1280 - ql - 80 * r] = (float)(xb.x * xb.x + xb.y * xb.y);
                              XA[r] = float2{(float)xa.x, (float)xa.y};
                              XB[r] = float2{(float)xb.x, (float)xb.y};
                              if (dbg) {
                                  const int ka = l + 64 * r, kb = 1024 - ka;
                                  p.dbg_X[2 * ka] = xa.x; p.dbg_X[2 * ka + 1] = xa.y;
                                  p.dbg_X[2 * kb] = xb.x; p.dbg_X[2 * kb + 1] = xb.y;
                              }
                          },
                          [&](double2 x512) {
                              MAG[MAG0 + 640] = (float)(x512.x * x512.x + x512.y * x512.y);
                              x512f = float2{(float)x512.x, (float)x512.y};
                              if (dbg) { p.dbg_X[2 * 512] = x512.x; p.dbg_X[2 * 512 + 1] = x512.y; }
                          });
        };
        // ---- fp32 first (round 5; spectrum32_2k): returns the absolute part K of the frame's guard band (pv_guard.h), 0 = out of the guarded range ----
        [[maybe_unused]] auto forward32 = [&]() -> float {
            unsigned mmax = 0u;
            spectrum32_2k<PV_F32_REGT1_2K>(raw, hw, smem, l, wl1024f, wl2048s, 1.0f / SC,
                                           [&](int r, pk::c32 xa, pk::c32 xb) {
                                               const float ma = mag32(xa), mb = mag32(xb);
                                               MAG[MAG0 + pl + 80 * r] = ma;
                                               MAG[MAG0 + 1280 - ql - 80 * r] = mb;
                                               mmax = max(max(mmax, __float_as_uint(ma)), __float_as_uint(mb));
                                               XA[r] = float2{xa.x, xa.y};
                                               XB[r] = float2{xb.x, xb.y};
                                           },
                                           [&](pk::c32 x512) { const float m = mag32(x512); MAG[MAG0 + 640] = m; mmax = max(mmax, __float_as_uint(m)); x512f = float2{x512.x, x512.y}; });
            return guard_k_of(wave_max_u32(mmax));
        };
        // ---- shift table Math.round(p * f) - p (pv:125,147), rebuilt only when f changes ----
        auto shift_table = [&]() {
            const unsigned pfb = __float_as_uint(pfm);
            if (!psh_valid || pfb != psh_key) {
                psh_key = pfb; psh_valid = true;
                // Math.round(p * f) - p (pv:125,147) of every candidate bin, DROP where the reference skips the peak (pv:127-129): lane l computes
                // bins l + 64 r but needs bins 16l..16l+15, so the table is written as an image into the (free) scratch and 32 bytes are read back
                short *IMG = reinterpret_cast<short *>(smem + O2_S);
                const double pfd = (double)pfm;
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int pk = l + 64 * r;
                    const double ps = floor((double)pk * pfd + 0.5);
                    const bool ok = (ps <= (double)H) && (ps >= -(double)(2 * N));
                    IMG[pk] = ok ? (short)((int)ps - pk) : (short)0x4000;   // DROP pushes every target of the region out of range
                }
                wave_sync();
                typedef const volatile __attribute__((address_space(3))) v4u *lds_v4u_;
                dq0 = *(lds_v4u_)(smem + O2_S + 32 * l);
                dq1 = *(lds_v4u_)(smem + O2_S + 32 * l + 16);
                wave_sync();
            }
        };
        // ---- magnitudes -> peak flags (pv:95-116) for bins 16l..16l+15 ----
        bool nonfinite = false;                                             // a magnitude of this frame is Inf or NaN (see pv_wave_kernel.hip)
        unsigned mg[20], pm[19];
        bool fl[16];
        auto read_flags = [&]() {
            typedef const volatile __attribute__((address_space(3))) v2u *lds_v2u;
            typedef const volatile __attribute__((address_space(3))) v4u *lds_v4u;
            const v2u q0 = *(lds_v2u)(&MAG[MAG0 + 20 * l - 6]);            // bins 16 l - 2, 16 l - 1: the tail of the previous group
            const v4u q1 = *(lds_v4u)(&MAG[MAG0 + 20 * l]);
            const v4u q2 = *(lds_v4u)(&MAG[MAG0 + 20 * l + 4]);
            const v4u q3 = *(lds_v4u)(&MAG[MAG0 + 20 * l + 8]);
            const v4u q4 = *(lds_v4u)(&MAG[MAG0 + 20 * l + 12]);
            const v2u q5 = *(lds_v2u)(&MAG[MAG0 + 20 * l + 20]);           // bins 16 l + 16, 16 l + 17: the head of the next group
            mg[0] = q0.x; mg[1] = q0.y;
            mg[2] = q1.x; mg[3] = q1.y; mg[4] = q1.z; mg[5] = q1.w; mg[6] = q2.x; mg[7] = q2.y; mg[8] = q2.z; mg[9] = q2.w;
            mg[10] = q3.x; mg[11] = q3.y; mg[12] = q3.z; mg[13] = q3.w; mg[14] = q4.x; mg[15] = q4.y; mg[16] = q4.z; mg[17] = q4.w;
            mg[18] = q5.x; mg[19] = q5.y;
#pragma unroll
            for (int j = 3; j < 19; j++) pm[j] = max(mg[j], mg[j + 1]);
#pragma unroll
            for (int i = 0; i < 16; i++) {
                // candidates are 2 <= k < H - 2 = 1023 (pv:97-100): lane 0 drops i < 2, lane 63 drops i = 15
                const bool in_range = (i < 2) ? (l != 0) : (i == 15) ? (l != 63) : true;
                fl[i] = in_range & (max(max(mg[i], mg[i + 1]), pm[i + 3]) < mg[i + 2]);
            }
            {
                unsigned mx = mg[2];
#pragma unroll
                for (int j = 3; j < 19; j += 2) mx = max(mx, pm[j]);
                nonfinite = __any(mx >= 0x7F800000u);
            }
        };
        if constexpr (!F32) {
            forward64();
            wave_sync();
        } else {
            // ---- the guard band and the order of the two transforms: pv_wave_kernel.hip / pv_guard.h.  This kernel keeps the fp64 transform out of line altogether ----
            shift_table();                                                  // (its image lives in the scratch: before the transform)
            // is any candidate bin of this lane within the band (K, R) of the largest of its four neighbours?  K = 0: the frame is out of the guarded range
            auto in_band = [&](float K, float Rr) -> bool {
                const pk::c32 KK{K, K}, RR{Rr, Rr};
                float tmin = 1.0f;
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const int i = 2 * j;
                    const unsigned nm0 = max(max(mg[i], mg[i + 1]), pm[i + 3]), nm1 = max(max(mg[i + 1], mg[i + 2]), pm[i + 4]);
                    const pk::c32 c2{__uint_as_float(mg[i + 2]), __uint_as_float(mg[i + 3])}, n2{__uint_as_float(nm0), __uint_as_float(nm1)};
                    const pk::c32 d2 = pk::sub(c2, n2), s2 = pk::add(c2, n2);
                    pk::c32 t2 = pk::fms(d2, d2, pk::mul(pk::fma(s2, RR, KK), s2));      // (c - n)^2 - (c + n) (K + R (c + n)): <= 0 inside the band
                    if (j == 0 && l == 0) t2 = pk::c32{1.f, 1.f};            // bins 0, 1 and 1023 are no candidates (pv:97-100)
                    if (j == 7 && l == 63) t2.y = 1.f;
                    tmin = fminf(fminf(tmin, t2.x), t2.y);
                }
                return !(K > 0.f) | (tmin <= 0.f);
            };
            auto cold = [&](int wide) -> float {
                const float k = forward_cold_2k(wide, src.in, src.hist, src.hist_len, src.sys ? 1 : 0, vec_in ? 1 : 0, (long)(m + 1) * HOP - N, w1024, w2048, wl1024f, wl2048s,
                                                1.0f / SC, wave_off, l);
                wave_sync();
                pv_prio(PH_PEAKS);
                read_flags();
                const float2 *XS = reinterpret_cast<const float2 *>(smem + O2_S);
#pragma unroll
                for (int r = 0; r < 8; r++) { XA[r] = XS[l + 64 * r]; XB[r] = XS[1024 - l - 64 * r]; }
                x512f = XS[512];                                            // (only lane 0 uses it)
                return k;
            };
#ifdef PV_FLIP_COUNT
            const bool wide_first = false;
#else
            const bool wide_first = pred >= PRED_WIDE;                      // wave-uniform: this chain's frames have been coming out as class B and provably so
#endif
            bool fall = true;
#ifdef PV_FLIP_COUNT
            unsigned vb_bits = 0; float vb_q[16]; bool vb_amb = false; float vb_K = 0.f;
#endif
            if (!wide_first) {
                const float guardK = forward32();
                wave_sync();
                pv_prio(PH_PEAKS);
                read_flags();
                fall = __any(in_band(guardK, GUARD_R));
#ifdef PV_FLIP_COUNT
                vb_K = guardK; vb_amb = fall; fall = true;
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    vb_bits |= fl[i] ? (1u << i) : 0u;
                    const float c = __uint_as_float(mg[i + 2]), n = __uint_as_float(max(max(mg[i], mg[i + 1]), pm[i + 3])), d = c - n, sm = c + n;
                    vb_q[i] = (guardK > 0.f && sm > 0.f) ? (d * d) / (sm * (guardK + GUARD_R * sm)) : 0.f;
                }
#endif
            }
            if (__builtin_expect(fall, 0)) {
                cold(1);
                // what the fp64 magnitudes alone say about the class (pv_guard.h, GUARD_GN)
                unsigned mx = mg[2];
#pragma unroll
                for (int j = 3; j < 19; j += 2) mx = max(mx, pm[j]);
                const unsigned mb = wave_max_u32(mx);
                const bool out_sure = mb < GUARD_M_MIN_BITS - GUARD_M_SLACK || mb >= GUARD_M_MAX_BITS + GUARD_M_SLACK;
                const bool in_sure = mb >= GUARD_M_MIN_BITS + GUARD_M_SLACK && mb < GUARD_M_MAX_BITS - GUARD_M_SLACK;
                const bool sure_b = out_sure || (in_sure && __any(in_band(GUARD_CKN * __uint_as_float(mb), GUARD_RN)));
#ifdef PV_FLIP_COUNT
                {
                    unsigned diff = 0; float qmax = 0.f;
#pragma unroll
                    for (int i = 0; i < 16; i++) { const bool dif = (fl[i] ? 1u : 0u) != ((vb_bits >> i) & 1u); diff |= dif ? 1u : 0u; if (dif) qmax = fmaxf(qmax, vb_q[i]); }
                    const bool flip_any = __any(diff != 0u);
                    n_flip += flip_any ? 1u : 0u;
                    n_uncaught += (flip_any && !vb_amb) ? 1u : 0u;
                    n_sure += sure_b ? 1u : 0u;
                    n_incons += (sure_b && !vb_amb) ? 1u : 0u;
                    if (diff && p.fwd_stats) atomicMax(reinterpret_cast<unsigned *>(p.fwd_stats + 512), __float_as_uint(qmax));
                    (void)vb_K;
                }
#endif
                bool class_b = true;
                if (wide_first && !sure_b) {
                    // the fp64 magnitudes do not settle this frame's class: the fp32 transform after all; class B means the fp64 one once more
                    const float k = cold(0);
                    class_b = __any(in_band(k, GUARD_R));
                    if (class_b) cold(1);
                }
                pred = (class_b && sure_b) ? min(pred + 1u, (unsigned)PRED_MAX) : (pred > 3u ? pred - 3u : 0u);
#ifdef PV_FLIP_COUNT
                n_fallback += vb_amb ? 1u : 0u;
#else
                n_fallback += class_b ? 1u : 0u;
#endif
            } else {
                pred = pred > 3u ? pred - 3u : 0u;
            }
        }
        // ---- f < 1: above-Nyquist residue, fast form, computed while Y's space can hold a stash of the fp32 spectrum (positions N/2+1 .. N/2+N/8 of
        //      fft.js's buffer = clean first half of the N/4-point sub-DFT of xw[4n+2]: W^{2k} S2[k] = (X[k] - X[k+N/4] + X[k+N/2] - X[k+3N/4]) / 4) ----
        float2 s2v[4] = {float2{0.f, 0.f}, float2{0.f, 0.f}, float2{0.f, 0.f}, float2{0.f, 0.f}};
        if (pf < 1.0) {
            float2 *XS = reinterpret_cast<float2 *>(smem + O2_S);
#pragma unroll
            for (int r = 0; r < 8; r++) { XS[l + 64 * r] = XA[r]; XS[1024 - l - 64 * r] = XB[r]; }
            if (l == 0) XS[512] = x512f;
            wave_sync();
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int k = 1 + l + 64 * j;                                // k in [1, 256]
                const float2 x0 = XS[k], x1 = XS[k + 512], x2 = XS[1024 - k], x3 = XS[512 - k];
                const float2 tsum{0.25f * ((x0.x - x1.x) + (x2.x - x3.x)), 0.25f * ((x0.y - x1.y) - (x2.y - x3.y))};
                // conj(W^{2k}) = conj(W_1024^{1 + l}) * conj(W_16^j): one loop-invariant register pair and three constant rotations instead of four loads
                // from the global table in every f < 1 frame
                const float c8 = 0.92387953251128675613f, s8 = 0.38268343236508977173f, h8 = 0.70710678118654752440f;
                const float2 wj = (j == 0) ? s2w0 : (j == 1) ? cmul(s2w0, float2{c8, s8}) : (j == 2) ? cmul(s2w0, float2{h8, h8}) : cmul(s2w0, float2{s8, c8});
                s2v[j] = cmul(tsum, wj);
            }
            wave_sync();
        }
        if constexpr (!F32) {
            shift_table();
            pv_prio(PH_PEAKS);
            read_flags();
        }
        // ---- nearest peaks, one ROUTE word per source bin ----
        int last_peak = -1, last_shift = 0;
        bool pairwise = false;                                              // f < 1: every collision is a (falling side, rising side) pair (wave-uniform)
        {
            if (dbg) {
#pragma unroll
                for (int i = 0; i < 16; i++) { p.dbg_flags[16 * l + i] = fl[i] ? 1 : 0; p.dbg_mag[16 * l + i] = __uint_as_float(mg[i + 2]); }
                if (l == 63) { p.dbg_flags[1024] = 0; p.dbg_mag[1024] = __uint_as_float(mg[18]); }
            }
            constexpr int NEGPD = -(4096 << 16), POSPD = 8192 << 16;        // "no peak on this side"
            int pd[16];
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const unsigned w = (i < 8) ? dq0[i >> 1] : dq1[(i - 8) >> 1];
                pd[i] = (int)__builtin_amdgcn_perm((unsigned)(16 * l + i), w, (i & 1) ? 0x05040302u : 0x05040100u);
            }
            int lastown[16], firstown[16];
            int cur = NEGPD;
#pragma unroll
            for (int i = 0; i < 16; i++) { cur = fl[i] ? pd[i] : cur; lastown[i] = cur; }
            int nx = POSPD;
#pragma unroll
            for (int i = 15; i >= 0; i--) { firstown[i] = nx; nx = fl[i] ? pd[i] : nx; }
            const int last_in = cur, first_in = nx;
            const unsigned long long occ = __ballot(cur >= 0);
            const unsigned long long below = occ & ((1ull << l) - 1ull);
            const unsigned long long above = (l == 63) ? 0ull : (occ >> (l + 1));
            const int src_lo = below ? 63 - __clzll((long long)below) : 0;
            const int src_hi = above ? l + __ffsll((long long)above) : 0;
            int cprev = __shfl(last_in, src_lo, 64), cnext = __shfl(first_in, src_hi, 64);
            if (!below) cprev = NEGPD;
            if (!above) cnext = POSPD;
            unsigned rt[16];
            unsigned rt1024 = NOROUTE;
            if (occ == 0ull) {
#pragma unroll
                for (int i = 0; i < 16; i++) rt[i] = NOROUTE;
            } else {
                const int lp = __shfl(last_in, 63 - __clzll((long long)occ), 64);
                last_peak = lp >> 16;
                last_shift = (int)(short)(lp & 0xFFFF);
                auto route_of = [&](int b, int pp, int pn) -> unsigned {
                    const int own = (b - (pp >> 16) < (pn >> 16) - b) ? pp : pn;    // owner rule (pv:132-141)
                    const int delta = __builtin_amdgcn_sbfe(own, 0, 16);
                    return __builtin_amdgcn_perm((unsigned)__mul24(delta, tmod), (unsigned)(b + delta), 0x05040100u);
                };
#pragma unroll
                for (int i = 0; i < 16; i++) rt[i] = route_of(16 * l + i, max(lastown[i], cprev), min(firstown[i], cnext));
                if (l == 63) rt1024 = route_of(1024, max(last_in, cprev), POSPD);   // source bin N/2: owner is the last peak
                if (!(pf >= 1.0)) {
                    // f < 1: bit 31 of a route = "rising side" (source owned by the peak on its right), and the wave-uniform test that lets the
                    // scatter run as store-then-add instead of claim rounds (pv_wave_kernel.hip, "pairwise"; tests/test_pairwise_rule.py)
                    rt1024 &= 0x7FFFFFFFu;
                    bool bad = false;
#pragma unroll
                    for (int i = 0; i < 16; i++) {
                        const int pp = max(lastown[i], cprev), pn = min(firstown[i], cnext), b = 16 * l + i;
                        const bool rising = !(b - (pp >> 16) < (pn >> 16) - b);
                        rt[i] = (rt[i] & 0x7FFFFFFFu) | (rising ? 0x80000000u : 0u);
                    }
                    if (!(pfm >= PV_PAIRWISE_SURE)) {                      // (f >= 2/3: the test cannot fail, see pv_device_common.h)
#pragma unroll
                        for (int i = 0; i < 16; i++) {
                            const int pp = max(lastown[i], cprev), pn = min(firstown[i], cnext);
                            const int gap = (pn >> 16) - (pp >> 16), ov = __builtin_amdgcn_sbfe(pp, 0, 16) - __builtin_amdgcn_sbfe(pn, 0, 16);
                            bad |= ov > (gap >> 1);
                        }
                    }
                    pairwise = PV_PAIRWISE && !__any(bad);
                }
            }
            wave_sync();                                                   // MAG is dead (in registers): ROUTE aliases it
#pragma unroll
            for (int j = 0; j < 4; j++) *reinterpret_cast<uint4 *>(&ROUTE[20 * l + 4 * j]) = uint4{rt[4 * j], rt[4 * j + 1], rt[4 * j + 2], rt[4 * j + 3]};
            if (l == 63) ROUTE[1280] = rt1024;
        }
        (void)last_peak;
        int upper_end = H;
        if (last_shift < 0) { upper_end = H - last_shift; if (upper_end > N) upper_end = N; }
        pv_prio(PH_SCATTER);
        // ---- zero Y (pv:121) ----
#pragma unroll
        for (int r = 0; r < 8; r++) *reinterpret_cast<v4f *>(&Y[2 * l + 128 * r]) = v4f{0.f, 0.f, 0.f, 0.f};
        if (l == 0) Y[1024] = float2{0.f, 0.f};
        wave_sync();
        // ---- shiftPeaks (pv:119-173) ----
        if (pf >= 1.0) {
            auto scatter = [&](auto mode_tag) {
                constexpr int MODE = decltype(mode_tag)::value;
                auto rot = [&](unsigned rt, float2 v) -> float2 {
                    if (MODE == 0) return v;
                    if (MODE == 2) {
                        const unsigned sg = (rt << 5) & 0x80000000u;         // top bit of the 11-bit rotation index = bit 26 of the route
                        return float2{__uint_as_float(__float_as_uint(v.x) ^ sg), __uint_as_float(__float_as_uint(v.y) ^ sg)};
                    }
                    return rotate2k<R>(rt, v, ROT);
                };
#pragma unroll
                for (int r = 0; r < 8; r++) {
                    const unsigned ra = ROUTE[pl + 80 * r], ta = ra & 0xFFFFu;
                    const unsigned rb = ROUTE[1280 - ql - 80 * r], tb = rb & 0xFFFFu;
                    if (ta < (unsigned)H) Y[ta] = rot(ra, XA[r]);
                    if (tb < (unsigned)H) Y[tb] = rot(rb, XB[r]);
                }
                if (l == 0) { const unsigned rt = ROUTE[640], tg = rt & 0xFFFFu; if (tg < (unsigned)H) Y[tg] = rot(rt, x512f); }
            };
            if (tmod == 0) scatter(std::integral_constant<int, 0>{});
            else if (tmod == N / 2) scatter(std::integral_constant<int, 2>{});
            else scatter(std::integral_constant<int, 1>{});
        } else {
            // f < 1 (and NaN): regions compress, `+=` collisions (pv:169-170) -> claim rounds; then the residue sources b = N/2 + k, all owned by the last peak
            unsigned rt[17];
            float2 ys[17];
            int id[17];
#pragma unroll
            for (int r = 0; r < 8; r++) {
                id[r] = l + 64 * r; rt[r] = ROUTE[pl + 80 * r]; ys[r] = rotate2k<R>(rt[r], XA[r], ROT);
                id[8 + r] = 1024 - l - 64 * r; rt[8 + r] = ROUTE[1280 - ql - 80 * r]; ys[8 + r] = rotate2k<R>(rt[8 + r], XB[r], ROT);
            }
            rt[16] = (l == 0) ? ROUTE[640] : NOROUTE;
            ys[16] = rotate2k<R>(rt[16], x512f, ROT);
            id[16] = 512;
            wave_sync();                                                   // routes are in registers: CLAIM may overwrite ROUTE
            if (pairwise) {
                // every collision of this frame is one falling-side source against one rising-side source (see the peak search): the falling side
                // (and the residue, which continues the falling side of the last peak) stores into the zeroed Y, then the rising side adds
                unsigned key[17];
#pragma unroll
                for (int r = 0; r < 17; r++) key[r] = rt[r] & 0x8000FFFFu;
#pragma unroll
                for (int r = 0; r < 17; r++) if (key[r] < (unsigned)H) Y[key[r]] = ys[r];
                wave_sync();
#pragma unroll
                for (int h0 = 0; h0 < 17; h0 += 9) {                       // two batches of reads: 34 more live registers would spill the f >= 1 path
                    float2 o[9];
#pragma unroll
                    for (int r = h0; r < 17 && r < h0 + 9; r++) o[r - h0] = Y[min(rt[r] & 0xFFFFu, 1024u)];
#pragma unroll
                    for (int r = h0; r < 17 && r < h0 + 9; r++)
                        if (key[r] - 0x80000000u < (unsigned)H) Y[key[r] - 0x80000000u] = float2{o[r - h0].x + ys[r].x, o[r - h0].y + ys[r].y};
                }
                wave_sync();
            } else
            claim_rounds2<17>(rt, ys, id, Y, CLAIM);
            if (upper_end > H) {
                const int up_delta = last_shift;
                const unsigned up_ridx = (unsigned)((up_delta & (N - 1)) * tmod) & (N - 1);
                if (upper_end <= H + N / 8 && pairwise) {                  // the residue lands above every other target of the last region: plain stores
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const int b = 1024 + 1 + l + 64 * j, tgt = b + up_delta;
                        const unsigned rtj = (b < upper_end && tgt >= 0 && tgt < H) ? ((up_ridx << 16) | (unsigned)tgt) : NOROUTE;
                        if (rtj != NOROUTE) Y[tgt] = rotate2k<R>(rtj, s2v[j], ROT);
                        if (dbg && b < upper_end) { p.dbg_X[2 * b] = s2v[j].x; p.dbg_X[2 * b + 1] = s2v[j].y; }
                    }
                } else if (upper_end <= H + N / 8) {                       // always when f >= 0.75; the fast form of the residue (s2v above)
                    unsigned rt2[4];
                    float2 ys2[4];
                    int id2[4];
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const int b = 1024 + 1 + l + 64 * j, tgt = b + up_delta;
                        rt2[j] = (b < upper_end && tgt >= 0 && tgt < H) ? ((up_ridx << 16) | (unsigned)tgt) : NOROUTE;
                        ys2[j] = rotate2k<R>(rt2[j], s2v[j], ROT);
                        id2[j] = b;
                        if (dbg && b < upper_end) { p.dbg_X[2 * b] = s2v[j].x; p.dbg_X[2 * b + 1] = s2v[j].y; }
                    }
                    claim_rounds2<4>(rt2, ys2, id2, Y, CLAIM);
                } else {
                    residue_scatter_2k<R>(src.in, src.hist, src.hist_len, src.sys, (long)(m + 1) * HOP - N, p.hann, p.tw32, wave_off, l, upper_end, up_delta, up_ridx,
                                          dbg ? p.dbg_X : nullptr);
                    cache_ok = false;                                        // the quarter buffer held the cached rows
                }
            }
        }
        if (nonfinite && l == 0) Y[1] = float2{__uint_as_float(0x7FC00000u), __uint_as_float(0x7FC00000u)};   // the reference's frame is NaN: so is this one
        wave_sync();
        if (dbg) {
#pragma unroll
            for (int r = 0; r < 16; r++) { const int k = l + 64 * r; p.dbg_Y[2 * k] = Y[k].x; p.dbg_Y[2 * k + 1] = Y[k].y; }
            if (l == 0) { p.dbg_Y[2048] = Y[1024].x; p.dbg_Y[2049] = Y[1024].y; }
        }
        pv_prio(PH_C2R);
        // ---- c2r pre-pass in conjugate pairs (bundle:69-76,102-114 folded), packed fp32: Zc[k] = SC ((Yk + Ym*) + j e^{+2 pi j k/N} (Yk - Ym*)), m = M - k ----
        pk::c32 zA[8], zB[8];                                              // Zc[l + 64 r], Zc[l + 64 r + 512]
        {
            const pk::c32 scsc{SC, SC};
            const pk::c32 w2048i = par ? wl2048s_x : wl2048s;
            const pk::c32 *Yc = reinterpret_cast<const pk::c32 *>(Y);
            pk::c32 zb[8];
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const int k = li + 64 * r;
                pk::c32 yk = Yc[k], ym = Yc[M - k];
                if (k == 0) { yk.y = 0.f; ym.y = 0.f; }
                const pk::c32 E = pk::add_conj(yk, ym), O = pk::sub_conj(yk, ym);
                const pk::c32 c = pk::cmul(mul_w32_inv_pk(O, r), w2048i);
                zA[r] = pk::fma_addj(E, scsc, c);
                zb[r] = pk::fma_conj_subj(E, scsc, c);                        // Zc[M - k]
            }
            const pk::c32 y512 = Yc[512];
            // hand-over through LDS (the route region is free): Zc[M - k] of the pair (l', r') is element 512 + (64 - l') + 64 (7 - r')
            pk::c32 *XCH = reinterpret_cast<pk::c32 *>(smem + O2_ROUTE);
#pragma unroll
            for (int r = 0; r < 8; r++) XCH[r * 64 + li] = zb[r];
            wave_sync();
#pragma unroll
            for (int r = 0; r < 8; r++) zB[7 - r] = XCH[r * 64 + 64 - li];   // lane 0 pairs with itself one register higher; its zB[0] is the self-paired bin 512
            if (li == 0) zB[0] = pk::c32{2.0f * y512.x * SC, -2.0f * y512.y * SC};
        }
        wave_sync();
        pv_prio(PH_IA);
        // ---- decimation-in-frequency stage, then two 512-point packed-fp32 inverse FFTs ----
        const pk::c32 w1024i = par ? wl1024f_x : wl1024f;
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const pk::c32 a = pk::add(zA[r], zB[r]), d = pk::sub(zA[r], zB[r]);
            zA[r] = a;                                                     // -> z[2 n']
            zB[r] = pk::cmul(mul_w16_inv_pk8(d, r), w1024i);                // -> z[2 n' + 1]
        }
        {   // every global access of the next frame, issued here (no row is carried through the forward FFTs)
            const int mn = (m + 1 < last_out) ? m + 1 : m;
            if (ROWCACHE && cache_ok && mn == m + 1) {
                // rows 0..2 of the next frame come from LDS, the others from memory
                load_rows_except(raw, mn);
#pragma unroll
                for (int r = 0; r < 3; r++) raw[r] = *reinterpret_cast<const v4f *>(smem + O2_CACHE + 16 * l + 1024 * r);
            } else {
                load_rows(raw, mn);
            }
            pf_next = pitch_row[mn];
        }
        // hop 128 runs the synthesis side of odd frames under the lane id L ^ 32: an exchange through LDS addresses follows the relabelling for
        // free, a register transpose acts on physical lanes -- transpose 1 stays in LDS there
        fft512_wave_inv_pk<(HOPQ != 1)>(zA, reinterpret_cast<pk::c32 *>(smem + O2_S), TW1F4, TW2F4, li);
        fft512_wave_inv_pk<(HOPQ != 1)>(zB, reinterpret_cast<pk::c32 *>(smem + O2_S), TW1F4, TW2F4, li);
        pv_prio(PH_OLA);
        // ---- Hann (pv:67), overlap-add in reference order (ola:149-157), emit (ola:111-118), shift (ola:130-137) ----
        {
            const bool emit_out = (m >= emit_v);
#pragma unroll
            for (int r = 0; r < 8; r++) hw[r] = HW4[r * 64 + li];           // HOPQ >= 2: stays live for the next frame's analysis window
            v4f fr[8];
#pragma unroll
            for (int r = 0; r < 8; r++)                                    // rounded to fp32 BEFORE the accumulation like the reference's Float32Array (pv:67): no contraction into the adds
                fr[r] = v4f{mul_rounded(zA[r].x, hw[r].x), mul_rounded(zA[r].y, hw[r].y), mul_rounded(zB[r].x, hw[r].z), mul_rounded(zB[r].y, hw[r].w)};
            if (HALF) {
                // lane L holds samples 256 r + 4 (L ^ 32 par) ..: the half of the lanes with li < 32 holds the earlier half row
                const bool early = li < 32;
                v4f sum[9];
#pragma unroll
                for (int r = 0; r < 8; r++) sum[r] = acc[r] + fr[r];
                sum[8] = v4f{0.f, 0.f, 0.f, 0.f};
                if (emit_out && early) {
                    float *dst = outp + (long)m * HOP + 4 * li;
                    if (vec_out) __builtin_nontemporal_store(sum[0], reinterpret_cast<v4f *>(dst));
                    else { dst[0] = sum[0].x; dst[1] = sum[0].y; dst[2] = sum[0].z; dst[3] = sum[0].w; }
                }
#pragma unroll
                for (int r = 0; r < 8; r++) {                              // slide by half a row: the early half takes the next row, the layout flips
                    acc[r].x = early ? sum[r + 1].x : sum[r].x;
                    acc[r].y = early ? sum[r + 1].y : sum[r].y;
                    acc[r].z = early ? sum[r + 1].z : sum[r].z;
                    acc[r].w = early ? sum[r + 1].w : sum[r].w;
                }
            } else {
#pragma unroll
                for (int r = 0; r < S_ROWS; r++) {
                    const v4f o = acc[r] + fr[r];
                    if (emit_out) {
                        float *dst = outp + (long)m * HOP + 4 * l + 256 * r;
                        if (vec_out) __builtin_nontemporal_store(o, reinterpret_cast<v4f *>(dst));
                        else { dst[0] = o.x; dst[1] = o.y; dst[2] = o.z; dst[3] = o.w; }
                    }
                }
#pragma unroll
                for (int r = 0; r < LROWS; r++) {
                    const int s = r + S_ROWS;
                    acc[r] = (s < LROWS) ? acc[s] + fr[s] : fr[s];
                }
            }
        }
    }

    W2K_STAMP(4);
    if constexpr (F32) {
        if (p.fwd_stats && lane == 0) {                                     // forward-transform statistics (pv_forward_stats), spread over 128 slots
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
        const int lend = HALF ? (lane ^ (((last_out - first_frame) & 1) << 5)) : lane;     // layout the last frame left the accumulator in
        // The history of the next call, read in ONE batch before anything is stored: written as `hs[i] = src.at(..)` the loads and stores alias as far
        // as the compiler can tell, every load is waited for on its own, and a streaming quantum spends 5.5 us here (station clock, -DPV_W2K_STAMPS).
        v4f hrow[LROWS > 0 ? LROWS : 1];
#pragma unroll
        for (int r = 0; r < LROWS; r++) {
            const long s = (long)p.nhops * HOP - L + 4 * lane + 256 * r;
            hrow[r] = (4 * lane + 256 * r < L) ? v4f{src.at(s), src.at(s + 1), src.at(s + 2), src.at(s + 3)} : v4f{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int r = 0; r < LROWS; r++) asm volatile("" : "+v"(hrow[r]));              // (all loads issued and landed before the first store)
#pragma unroll
        for (int r = 0; r < LROWS; r++) {
            if (4 * lend + 256 * r < L) {
                float *a = acc_out + (long)ch * L + 4 * lend + 256 * r;
                a[0] = acc[r].x; a[1] = acc[r].y; a[2] = acc[r].z; a[3] = acc[r].w;
            }
            if (4 * lane + 256 * r < L) {
                float *hs = hist_out + (long)ch * L + 4 * lane + 256 * r;
                hs[0] = hrow[r].x; hs[1] = hrow[r].y; hs[2] = hrow[r].z; hs[3] = hrow[r].w;
            }
        }
    }
    W2K_STAMP(5);
    pv_signal_done<false>(p.done, done_seq, chain);
    W2K_STAMP(6);
    if (RESIDENT) goto resident_top;
}

#ifndef PV_W2K_WMIN
#define PV_W2K_WMIN 2
#endif
template <int HOPQ, bool AUX>
hipError_t launch2k(const PvKernelParams &p, int nch, int nchunks, hipStream_t st)
{
    static std::atomic<bool> attr_done[16], attr_done_f[16];
    // the product's instances run the forward transform in fp32 first (F32; pv_guard.h); with p.fwd64 != 0 (PV_FLAG_FP64_FORWARD) and in the tap instance every frame runs the fp64 one
    const bool f32 = !AUX && !p.fwd64;
    auto k = f32 ? pv_wave2k_kernel<HOPQ, AUX, false, !AUX> : pv_wave2k_kernel<HOPQ, AUX, false, false>;
    {
        const hipError_t e = pv_set_dynamic_lds_once(f32 ? attr_done_f : attr_done, reinterpret_cast<const void *>(k), (int)pv_wave2k_lds_bytes());
        if (e != hipSuccess) return e;
    }
    PvKernelParams q = p;
    q.nchunks = nchunks;
    q.nch = nch;
    const long chains = (long)nch * nchunks;
    // a streaming quantum has a handful of chains: spread them over the CUs instead of packing WAVES2 into one workgroup
    static std::atomic<int> cus_of[16];
    int dev = 0;
    (void)hipGetDevice(&dev);
    int cus = cus_of[dev & 15].load(std::memory_order_relaxed);
    if (!cus) { int c = 0; (void)hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev); cus = c > 0 ? c : 1; cus_of[dev & 15].store(cus, std::memory_order_relaxed); }
    long w = (chains + cus - 1) / cus;
    if (w < PV_W2K_WMIN) w = PV_W2K_WMIN;
    if (w > WAVES2) w = WAVES2;
    hipLaunchKernelGGL(k, dim3((unsigned)((chains + w - 1) / w), 1, 1), dim3(64 * (unsigned)w, 1, 1), T2_BYTES + (size_t)w * WAVE2_LDS, st, q);
    return hipGetLastError();
}

template <int HOPQ>
hipError_t launch2k_resident(const PvKernelParams &p, int nslots, hipStream_t st)
{
    static std::atomic<bool> attr_done[16], attr_done_f[16];
    const bool f32 = !p.fwd64;
    auto k = f32 ? pv_wave2k_kernel<HOPQ, false, true, true> : pv_wave2k_kernel<HOPQ, false, true, false>;
    {
        const hipError_t e = pv_set_dynamic_lds_once(f32 ? attr_done_f : attr_done, reinterpret_cast<const void *>(k), (int)pv_wave2k_lds_bytes());
        if (e != hipSuccess) return e;
    }
    PvKernelParams q = p;
    q.nchunks = 1; q.nch = nslots; q.nhops = 1; q.frames_per_chunk = 1;
    constexpr int w = 2;                                                   // two channel slots per workgroup: the slots spread over the CUs
    hipLaunchKernelGGL(k, dim3((unsigned)((nslots + w - 1) / w), 1, 1), dim3(64 * w, 1, 1), T2_BYTES + (size_t)w * WAVE2_LDS, st, q);
    return hipGetLastError();
}

}  // namespace

hipError_t pv_launch_wave2k_resident(const PvKernelParams &p, int nslots, hipStream_t st)
{
    switch (p.hop) {
    case 128: return launch2k_resident<1>(p, nslots, st);
    case 256: return launch2k_resident<2>(p, nslots, st);
    case 512: return launch2k_resident<4>(p, nslots, st);
    case 1024: return launch2k_resident<8>(p, nslots, st);
    case 2048: return launch2k_resident<16>(p, nslots, st);
    default: return hipErrorInvalidValue;
    }
}

size_t pv_wave2k_lds_bytes() { return T2_BYTES + WAVES2 * WAVE2_LDS; }
int pv_wave2k_threads() { return 64 * WAVES2; }
bool pv_wave2k_supported(int log2n, int hop) { return log2n == 11 && (hop == 128 || hop == 256 || hop == 512 || hop == 1024 || hop == 2048); }
hipError_t pv_launch_wave2k(const PvKernelParams &p, int nch, int nchunks, hipStream_t st)
{
    const bool aux = (p.dbg_mag != nullptr);                              // pv_debug_frame: the tap instance of the SAME kernel
    switch (p.hop) {
    case 128: return aux ? launch2k<1, true>(p, nch, nchunks, st) : launch2k<1, false>(p, nch, nchunks, st);
    case 256: return aux ? launch2k<2, true>(p, nch, nchunks, st) : launch2k<2, false>(p, nch, nchunks, st);
    case 512: return aux ? launch2k<4, true>(p, nch, nchunks, st) : launch2k<4, false>(p, nch, nchunks, st);
    case 1024: return aux ? launch2k<8, true>(p, nch, nchunks, st) : launch2k<8, false>(p, nch, nchunks, st);
    case 2048: return aux ? launch2k<16, true>(p, nch, nchunks, st) : launch2k<16, false>(p, nch, nchunks, st);
    default: return hipErrorInvalidValue;
    }
}
