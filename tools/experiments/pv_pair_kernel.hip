// pv_pair_kernel.hip -- a PAIR of wavefronts per 4096-point frame chain: N = 4096, hop in {512, 1024, 2048, 4096} (BASELINE configs[3]).
//
// One wave per frame (pv_wave2k_kernel.hip) stops at N = 2048: 32 packed elements per lane do not fit 256 registers.  The workgroup kernel
// (four waves per 4096-point frame, eight elements per lane) pays ~21 workgroup barriers and three transposes per FFT.  Here two waves share a
// frame, each running pv_wave2k's 1024-point complex FFT (two 512-point wave FFTs + one in-register radix-2 stage) on ITS half of the packed
// sequence, and the two halves meet in ONE exchange per transform:
//
//   z[n] = xw[2n] + j xw[2n+1], n < 2048;  wave g in {0, 1} owns z_g[n''] = z[2n'' + g];  E_g = FFT1024(z_g)       (no LDS traffic between the waves)
//   Z[k] = E_0[k] + W^k E_1[k], Z[k + 1024] = E_0[k] - W^k E_1[k], W = exp(-2 pi j / 2048), and the real-FFT split pairs Z[k] with Z[2048 - k]:
//   the four bins {k, 2048 - k, 1024 - k, 1024 + k} need exactly E_0[k], E_0[1024 - k], E_1[k], E_1[1024 - k].
//
// A wave first brings E_g[k] and E_g[1024 - k] into one lane (the wave-local partner exchange of pv_wave2k), then the waves split the groups:
// wave 0 finishes k = l + 64 r for r < 4, wave 1 for r in [4, 8) -- each publishes the 8 values the other one needs (8 ds_write_b128 into its
// OWN scratch: no write-after-read hazard with the other wave), one barrier, 8 reads, and every lane holds 16 finished bins.  The inverse
// mirrors it: the wave that owns a group computes the c2r pre-pass and the decimation-in-frequency stage for its four bins, keeps its own
// kind (wave 0 the even-sample spectrum A, wave 1 the odd-sample spectrum B), publishes the other kind, one barrier.  Six workgroup barriers
// per f >= 1 frame in all (forward exchange, magnitudes complete, peak hand-over across the wave boundary, routes + zeroed Y complete, scatter
// complete, inverse exchange), each between TWO waves; four such workgroups per CU keep every SIMD at two waves.
//
// Samples: lane l, register row r of wave g holds z[4n' + g] and z[4n' + 2 + g], n' = l + 64 r, i.e. the two float2 at samples 8n' + 2g and
// 8n' + 4 + 2g: the two waves interleave 8-byte pieces of every 32-byte group (L2 merges them; measured traffic in profiles/).  A register
// row is 512 samples; the overlap-add accumulator slides by register renaming (hop = 512 S_ROWS).
//
// Everything between the transforms is pv_wave2k's pipeline with the lane id L = 64 g + l (16 consecutive bins per lane, padded magnitude /
// route layout, packed peak words, select chains + ballot + bpermute inside a wave, two words handed across the wave boundary), the
// workgroup kernel's f < 1 scatter (store / barrier / add on pairwise frames, else atomic-MIN claim rounds: two waves post on the same target,
// the order must not depend on timing), the fast
// above-Nyquist residue from a spectrum stash, and the per-quarter rebuild (residue_scatter_pair) when the last region reads beyond N/2 + N/8.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "pv_kernels.h"
#include "pv_device_common.h"
#include "pv_pk_math.h"
#ifndef PV_PT
#define PV_PT 2, 0, 2, 3, 3, 3, 3, 3, 1, 0, 1, 3     // phase priorities of this kernel (pv_wave_fft.h; profiles/r03_priority_sweep.md: C4 0.92 -> 0.83 ms)
#endif
#include "pv_wave_fft.h"
#ifndef PV_PAIRWISE
#define PV_PAIRWISE 1                               // 0: every f < 1 frame goes through the claim rounds (A/B)
#endif

namespace {

constexpr int N4 = 4096, M4 = 2048, H4 = 2049, LOG2N4 = 12;

// workgroup LDS (byte offsets); one frame chain per workgroup of two waves
constexpr int P_TW1 = 0;                          // double2[8*64]  W_512^{l k}
constexpr int P_TW2 = P_TW1 + 8 * 64 * 16;        // double2[8*8]   W_64^{n0 k}
constexpr int P_ROT = P_TW2 + 8 * 8 * 16;         // float2[16]     exp(+2 pi j q / 16)  (hop = N/8: R = 8)
constexpr int P_BND = P_ROT + 16 * 8;             // int[4]         last / first peak word of each wave (peak search across the wave boundary)
constexpr int P_A = P_BND + 64;                   // 18432 B: fp64 transpose scratch of wave 0 | of wave 1 (9216 each: partner exchange, the forward publish,
                                                  //          shift-table image, fp32 transposes + the wave-local hand-over) | Y float2[2049] | spectrum stash (f < 1)
constexpr int P_B = P_A + 18432;                  // 10304 B: mags f32 / routes u32 in the padded layout (2569 words) | claim words u32[2049] | inverse publish
                                                  //          2 x 4608 | residue quarter float2[1024]
constexpr int P_BYTES = P_B + 10304;              // 38144: four workgroups per CU

// padded layout of magnitudes and routes (see pv_wave2k_kernel.hip): P(bin) = bin + 4 (bin >> 4); with k = l + 64 r, pl = l + 4 (l >> 4),
// ql = l + 4 ((l + 15) >> 4):  P(k) = pl + 80 r, P(1024 + k) = 1280 + pl + 80 r, P(1024 - k) = 1280 - ql - 80 r, P(2048 - k) = 2560 - ql - 80 r
constexpr int MAG0 = 8;

__device__ __forceinline__ double2 csq(double2 a) { return double2{(a.x - a.y) * (a.x + a.y), 2.0 * a.x * a.y}; }

// o * exp(-2 pi j r / 32), r = 0..7 (compile-time), fp64
__device__ __forceinline__ double2 mul_w32_f(double2 o, int r)
{
    constexpr double c[9] = {1.0, 0.98078528040323044913, 0.92387953251128675613, 0.83146961230254523708, 0.70710678118654752440,
                             0.55557023301960222474, 0.38268343236508977173, 0.19509032201612826785, 0.0};
    if (r == 0) return o;
    return cmul(o, double2{c[r], -c[8 - r]});
}
// o * exp(-2 pi j r / 64), r = 0..7, fp64
__device__ __forceinline__ double2 mul_w64_f(double2 o, int r)
{
    constexpr double c[8] = {1.0, 0.99518472667219688624, 0.98078528040323044913, 0.95694033573220886494, 0.92387953251128675613,
                             0.88192126434835502971, 0.83146961230254523708, 0.77301045336273696081};
    constexpr double s[8] = {0.0, 0.09801714032956060199, 0.19509032201612826785, 0.29028467725446236764, 0.38268343236508977173,
                             0.47139673682599764856, 0.55557023301960222474, 0.63439328416364549822};
    if (r == 0) return o;
    return cmul(o, double2{c[r], -s[r]});
}
// o * exp(+2 pi j r / 32), o * exp(+2 pi j r / 64), packed fp32
__device__ __forceinline__ pk::c32 mul_w32_i(pk::c32 o, int r)
{
    constexpr float c[9] = {1.0f, 0.98078528040323044913f, 0.92387953251128675613f, 0.83146961230254523708f, 0.70710678118654752440f,
                            0.55557023301960222474f, 0.38268343236508977173f, 0.19509032201612826785f, 0.0f};
    if (r == 0) return o;
    return pk::cmul(o, pk::c32{c[r], c[8 - r]});
}
__device__ __forceinline__ pk::c32 mul_w64_i(pk::c32 o, int r)
{
    constexpr float c[8] = {1.0f, 0.99518472667219688624f, 0.98078528040323044913f, 0.95694033573220886494f, 0.92387953251128675613f,
                            0.88192126434835502971f, 0.83146961230254523708f, 0.77301045336273696081f};
    constexpr float s[8] = {0.0f, 0.09801714032956060199f, 0.19509032201612826785f, 0.29028467725446236764f, 0.38268343236508977173f,
                            0.47139673682599764856f, 0.55557023301960222474f, 0.63439328416364549822f};
    if (r == 0) return o;
    return pk::cmul(o, pk::c32{c[r], s[r]});
}
// o * exp(+2 pi j r / 16), r = 0..7, packed fp32 (first stage of a wave's decimation-in-frequency inverse)
__device__ __forceinline__ pk::c32 mul_w16_i(pk::c32 o, int r)
{
    constexpr float c = 0.92387953251128675613f, s = 0.38268343236508977173f, h = 0.70710678118654752440f;
    switch (r) {
    case 0: return o;
    case 1: return pk::cmul(o, pk::c32{c, s});
    case 2: return pk::cmul(o, pk::c32{h, h});
    case 3: return pk::cmul(o, pk::c32{s, c});
    case 4: return pk::c32{-o.y, o.x};
    case 5: return pk::cmul(o, pk::c32{-s, c});
    case 6: return pk::cmul(o, pk::c32{-h, h});
    default: return pk::cmul(o, pk::c32{-c, s});
    }
}

// 512-point inverse wave FFT in packed fp32 (fft512_wave_inv_pk of pv_wave_fft.h) with its twiddles rounded on the fly from the fp64 tables:
// a workgroup of two waves cannot afford a second, fp32 copy of the tables in LDS (four workgroups share a CU).
__device__ __forceinline__ void fft512_wave_inv_pk64(pk::c32 (&a)[8], pk::c32 *S, const double2 *TW1, const double2 *TW2, int l)
{
    const int lh = l >> 3, ll = l & 7;
    v4f *S4 = reinterpret_cast<v4f *>(S);
    pk::radix8_inv(a);
#pragma unroll
    for (int k = 1; k < 8; k++) { const double2 w = TW1[k * 64 + l]; a[k] = pk::cmul(a[k], pk::c32{(float)w.x, -(float)w.y}); }
    if (PV_PERM_T1 & 2) {                                                  // transpose 1 in registers (pv_wave_fft.h)
        unsigned w[8][2];
#pragma unroll
        for (int k = 0; k < 8; k++) { w[k][0] = __float_as_uint(a[k].x); w[k][1] = __float_as_uint(a[k].y); }
        transpose_hi3_regs<2>(w);
#pragma unroll
        for (int k = 0; k < 8; k++) a[k] = pk::c32{__uint_as_float(w[k][0]), __uint_as_float(w[k][1])};
    } else {
        pv_prio(PH_IX);
#pragma unroll
        for (int j = 0; j < 4; j++) S4[j * TPP + l] = v4f{a[2 * j].x, a[2 * j].y, a[2 * j + 1].x, a[2 * j + 1].y};
        wave_sync();
#pragma unroll
        for (int n = 0; n < 8; n++) a[n] = S[2 * ((lh >> 1) * TPP + 8 * n + ll) + (lh & 1)];
        wave_sync();
        pv_prio(PH_IA);
    }
    pk::radix8_inv(a);
#pragma unroll
    for (int k = 1; k < 8; k++) { const double2 w = TW2[k * 8 + ll]; a[k] = pk::cmul(a[k], pk::c32{(float)w.x, -(float)w.y}); }
    pv_prio(PH_IX);
#pragma unroll
    for (int j = 0; j < 4; j++) S4[j * TPP + lh * 8 + ((ll + lh) & 7)] = v4f{a[2 * j].x, a[2 * j].y, a[2 * j + 1].x, a[2 * j + 1].y};
    wave_sync();
#pragma unroll
    for (int n = 0; n < 8; n++) a[n] = S[2 * ((lh >> 1) * TPP + ll * 8 + ((n + ll) & 7)) + (lh & 1)];
    wave_sync();
    pv_prio(PH_IP3);
    pk::radix8_inv(a);
}

// Rotation exp(+2 pi j ridx / N) of one source value (pv:155-170); ridx is a multiple of N / R
template <int R_>
__device__ __forceinline__ float2 rotate4k(unsigned route, float2 v, const float2 *ROT)
{
    if (R_ == 1) return v;
    if (R_ == 2) {
        const unsigned sg = (route << 4) & 0x80000000u;                     // top bit of the 12-bit rotation index = bit 27 of the route
        return float2{__uint_as_float(__float_as_uint(v.x) ^ sg), __uint_as_float(__float_as_uint(v.y) ^ sg)};
    }
    if (R_ == 4) return rotate_route<4, LOG2N4>(route, v, nullptr);
    return cmul(v, ROT[(route >> 24) & 15u]);
}

// Claim rounds of the workgroup (see claim_rounds_wg in pv_wg_kernel.hip): atomic-MIN on the claim word of the target, the smallest source id
// wins the round and does a plain read-modify-write, losers re-post.  Ascending source order whatever the timing of the two waves.
// CLAIM[0..H) is all-ones on entry and on exit.  Every thread of the workgroup calls it (barriers inside).
template <int NS>
__device__ __forceinline__ void claim_rounds_pair(const unsigned (&rt)[NS], const float2 (&ys)[NS], const int (&id)[NS], float2 *Y, unsigned *CLAIM)
{
    unsigned pend = 0;
    unsigned tg[NS];
#pragma unroll
    for (int r = 0; r < NS; r++) {
        const unsigned t = rt[r] & 0xFFFFu;
        const bool ok = t < (unsigned)H4;
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
        for (int r = 0; r < NS; r++) c[r] = CLAIM[tg[r]];
#pragma unroll
        for (int r = 0; r < NS; r++) o[r] = Y[tg[r]];
#pragma unroll
        for (int r = 0; r < NS; r++) {
            if ((pend & (1u << r)) && c[r] == (unsigned)id[r]) {
                Y[tg[r]] = float2{o[r].x + ys[r].x, o[r].y + ys[r].y};
                CLAIM[tg[r]] = 0xFFFFFFFFu;
                pend &= ~(1u << r);
            }
        }
    }
}

__device__ __forceinline__ int digitrev4_4k(int v, int nd)
{
    const unsigned r = __brev((unsigned)v) >> (32 - 2 * nd);
    return (int)(((r & 0x55555555u) << 1) | ((r >> 1) & 0x55555555u));
}

// Rare path (f < 1 frames whose last region reads beyond position N/2 + N/8): what fft.js's in-place real DIT leaves at positions N/2+1 .. N-1, one
// quarter of the buffer at a time, by re-running the reference's stage structure on that quarter in fp32 (log2 N even: radix-4 base blocks,
// bundle:468-508, then the radix-4 stages with their predicated stores, bundle:329-441); its sources, all owned by the last peak (pv:133), are
// added into Y.  Both waves of the workgroup run it (barriers inside).  The quarter buffer aliases the claim words: they are refilled afterwards.
template <int R_>
__device__ __attribute__((noinline)) PV_NO_DS_MERGE void residue_scatter_pair(const float *in, const float *hist, int hist_len, long s0, const float *__restrict__ hann,
                                                                               const float2 *__restrict__ tw32, int t, int upper_end, int up_delta, unsigned up_ridx,
                                                                               double *dbg_X, bool plain)
{
    // plain: the frame passed the pairwise test, nothing but the residue lands on the residue's targets (all distinct): stores, no claim rounds
    constexpr int N = N4, H = H4, QN = N / 4, T = 128;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float2 *Y = reinterpret_cast<float2 *>(smem + P_A);
    unsigned *CLAIM = reinterpret_cast<unsigned *>(smem + P_B);
    float2 *Q = reinterpret_cast<float2 *>(smem + P_B);
    const float2 *ROT = reinterpret_cast<const float2 *>(smem + P_ROT);
    const WaveSrc src{in, hist, hist_len};
    for (int base = N / 2; base < N && base < upper_end; base += QN) {
        __syncthreads();                                                   // the claim words of the previous scatter are done with
#pragma unroll
        for (int it = 0; it < 2; it++) {                                   // QN / 4 = 256 radix-4 blocks per quarter; input index = base-4 digit reversal of the block
            const int lb = t + T * it, blk = base / 4 + lb;
            const int off = digitrev4_4k(blk, (LOG2N4 - 2) / 2);
            const float a = src.at(s0 + off) * hann[off], b = src.at(s0 + off + N / 4) * hann[off + N / 4];
            const float c = src.at(s0 + off + N / 2) * hann[off + N / 2], d = src.at(s0 + off + 3 * N / 4) * hann[off + 3 * N / 4];
            const float t0 = a + c, t1 = a - c, t2 = b + d, t3 = b - d;
            Q[4 * lb] = float2{t0 + t2, 0.f};
            Q[4 * lb + 1] = float2{t1, -t3};
            Q[4 * lb + 2] = float2{t0 - t2, 0.f};
            Q[4 * lb + 3] = float2{t1, t3};
        }
        __syncthreads();
        for (int log2m = 4; log2m <= LOG2N4 - 2; log2m += 2) {             // block sizes 16, 64, 256, 1024 inside the quarter
            const int q = (1 << log2m) >> 2, hq = q >> 1;
            const int nblocks = QN >> log2m;
            const int tws = LOG2N4 - log2m;
            for (int u = t; u < nblocks * (hq + 1); u += T) {              // the butterflies of a stage touch disjoint elements: any order
                int blk, i;
                if (u < nblocks * hq) { blk = u / hq; i = u - blk * hq; } else { blk = u - nblocks * hq; i = hq; }
                const int o = blk << log2m;
                const float2 Av = Q[o + i];
                const float2 Bv = cmul(Q[o + q + i], tw32[i << tws]);
                const float2 Cc = cmul(Q[o + 2 * q + i], tw32[(2 * i) << tws]);
                const float2 D = cmul(Q[o + 3 * q + i], tw32[(3 * i) << tws]);
                const float2 T0 = cadd(Av, Cc), T1 = csub(Av, Cc), T2 = cadd(Bv, D), T3 = csub(Bv, D);
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
        unsigned rt[8];
        float2 ys[8];
        int id[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int b = base + t + T * j, tgt = b + up_delta;
            rt[j] = (b >= H && b < upper_end && tgt >= 0 && tgt < H) ? ((up_ridx << 16) | (unsigned)tgt) : NOROUTE;
            ys[j] = rotate4k<R_>(rt[j], Q[t + T * j], ROT);
            id[j] = b - N / 2;                                             // ascending with the source bin; the regular sources are done by now
        }
        if (plain) {
#pragma unroll
            for (int j = 0; j < 8; j++) if (rt[j] != NOROUTE) Y[rt[j] & 0xFFFFu] = ys[j];
            continue;                                                      // (the barrier at the top of the next quarter orders the quarter buffer)
        }
        __syncthreads();                                                   // the quarter is in registers: its space becomes the claim words again
#pragma unroll
        for (int j = 0; j < 16; j++) CLAIM[t + T * j] = 0xFFFFFFFFu;
        if (t == 0) CLAIM[M4] = 0xFFFFFFFFu;
        claim_rounds_pair<8>(rt, ys, id, Y, CLAIM);                         // (its first barrier orders the fill before the first claims)
    }
}

// S_ROWS = hop / 512
// AUX = true: test-tap instance (pv_debug_frame); the production instance carries no tap code.
template <int S_ROWS, bool AUX>
__global__ __launch_bounds__(128, 2) PV_NO_DS_MERGE void pv_pair_kernel(const PvKernelParams p)
{
    constexpr int N = N4, M = M4, H = H4;
    constexpr int HOP = 512 * S_ROWS, R = N / HOP, LROWS = 8 - S_ROWS, L = N - HOP;
    const int lane = threadIdx.x & 63;
    const int g = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);        // which half of the packed sequence this wave transforms
    const long chain = blockIdx.x;
    const int ch = (int)(chain / p.nchunks), chunk = (int)(chain - (long)ch * p.nchunks);

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const double2 *TW1 = reinterpret_cast<const double2 *>(smem + P_TW1);
    const double2 *TW2 = reinterpret_cast<const double2 *>(smem + P_TW2);
    const float2 *ROT = reinterpret_cast<const float2 *>(smem + P_ROT);
    volatile int *BND = reinterpret_cast<volatile int *>(smem + P_BND);
    {
        double2 *t1 = reinterpret_cast<double2 *>(smem + P_TW1);
        double2 *t2 = reinterpret_cast<double2 *>(smem + P_TW2);
        for (int i = threadIdx.x; i < 512; i += 128) {
            const int k = i >> 6, ln = i & 63;
            t1[i] = p.tw64[(8 * ln * k) & (N - 1)];                         // W_512^{ln k} = exp(-2 pi j ln k 8 / 4096)
            if (i < 64) { const int k2 = i >> 3, n0 = i & 7; t2[i] = p.tw64[(64 * n0 * k2) & (N - 1)]; }   // W_64^{n0 k2}
            if (i < 16) reinterpret_cast<float2 *>(smem + P_ROT)[i] = cconj(p.tw32[(i * (N / 16)) & (N - 1)]);
        }
    }
    __syncthreads();

    unsigned char *SA = smem + P_A + 9216 * g;                              // this wave's scratch
    const unsigned char *SO = smem + P_A + 9216 * (1 - g);                  // the other wave's (read-only here, and only what it published)
    double2 *S64 = reinterpret_cast<double2 *>(SA);
    float2 *Y = reinterpret_cast<float2 *>(smem + P_A);
    float *MAG = reinterpret_cast<float *>(smem + P_B);
    unsigned *ROUTE = reinterpret_cast<unsigned *>(smem + P_B);
    unsigned *CLAIM = reinterpret_cast<unsigned *>(smem + P_B);
    v4u dq0{0u, 0u, 0u, 0u}, dq1{0u, 0u, 0u, 0u};                           // shifts of this lane's own 16 candidate bins (i16 each)
    unsigned psh_key = 0u;
    bool psh_valid = false;

    const int first_out = chunk * p.frames_per_chunk;
    int last_out = first_out + p.frames_per_chunk;
    if (last_out > p.nhops) last_out = p.nhops;
    int first_frame = first_out - (R - 1);
    const bool from_state = (first_frame <= 0);
    if (from_state) first_frame = 0;

    const long cbase = (long)ch * p.ch_stride;
    const WaveSrc src{p.in + cbase, p.hist_in + (long)ch * L, L};
    float *outp = p.out + cbase;
    const bool vec_out = (reinterpret_cast<uintptr_t>(outp) & 7u) == 0;
    const bool vec_in = ((reinterpret_cast<uintptr_t>(src.in) | reinterpret_cast<uintptr_t>(src.hist)) & 7u) == 0;
    const float *pitch_row = p.pitch + (p.pitch_stride ? (long)(ch / p.ch_per_stream) * p.pitch_stride : 0);

    const double2 wN = p.tw64[lane];                                       // W_4096^l; W_2048^l and W_1024^l are its squares
    constexpr float SC = 2.0f / ((float)N * (float)R);                      // 1/N of the inverse, 1/R of the overlap-add, 2 for the halved Hann (exact)
    const int so = 8 * lane + 2 * g;                                        // this lane's first sample inside a register row of 512

    v4f acc[8];
#pragma unroll
    for (int r = 0; r < 8; r++) acc[r] = v4f{0.f, 0.f, 0.f, 0.f};
    if (from_state) {
#pragma unroll
        for (int r = 0; r < LROWS; r++) {
            const float *a = p.acc_in + (long)ch * L + so + 512 * r;
            acc[r] = v4f{a[0], a[1], a[4], a[5]};
        }
    }
    // Global accesses of a frame: one VGPR byte offset (this lane's position inside a register row) on top of wave-uniform row bases, so that the
    // 32 loads and 4 stores need no per-access 64-bit address arithmetic.  A frame that reaches back into the carried history takes the slow form.
    auto load_rows = [&](v4f *w, int frame, int so) {
        const long s0 = (long)(frame + 1) * HOP - N;                        // wave-uniform
        if (s0 >= 0 && vec_in) {
            const unsigned ob = 4u * (unsigned)so;
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const char *rb = reinterpret_cast<const char *>(src.in + s0 + 512 * r);
                const v2f a = *reinterpret_cast<const v2f *>(rb + ob), b = *reinterpret_cast<const v2f *>(rb + ob + 16);
                w[r] = v4f{a.x, a.y, b.x, b.y};
            }
        } else {
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const long sx = s0 + so + 512 * r;                          // even: neither float2 straddles history / input (both lengths are even)
                const float *q0 = sx < 0 ? src.hist + sx + src.hist_len : src.in + sx;
                const float *q1 = sx + 4 < 0 ? src.hist + sx + 4 + src.hist_len : src.in + sx + 4;
                if (vec_in) { const v2f a = *reinterpret_cast<const v2f *>(q0), b = *reinterpret_cast<const v2f *>(q1); w[r] = v4f{a.x, a.y, b.x, b.y}; }
                else w[r] = v4f{q0[0], q0[1], q1[0], q1[1]};
            }
        }
    };
    auto load_hann = [&](v4f *w, int so) {                                  // 0.5 * Hann at this lane's samples: the second half of the table (pv_kernels.h)
        const unsigned ob = 4u * (unsigned)so;
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const char *rb = reinterpret_cast<const char *>(p.hann + N + 512 * r);
            const v2f a = *reinterpret_cast<const v2f *>(rb + ob), b = *reinterpret_cast<const v2f *>(rb + ob + 16);
            w[r] = v4f{a.x, a.y, b.x, b.y};
        }
    };
    v4f raw[8], hw[8];
    load_rows(raw, first_frame, so);
    load_hann(hw, so);
    float pf_next = pitch_row[first_frame];
    int emit_v = first_out;
    asm volatile("" : "+v"(emit_v));
    // conj(W^{2 kk}) of the fast residue's four bins of this lane, kk = 1 + (64 g + lane) + 128 j: loop-invariant (as a load inside the frame it sat, exposed,
    // on the critical path of every f < 1 frame)
    float2 s2w[4];
#pragma unroll
    for (int j = 0; j < 4; j++) s2w[j] = cconj(p.tw32[2 * (1 + 64 * g + lane + 128 * j)]);

    for (int m = first_frame; m < last_out; ++m) {
        const bool dbg = AUX && (p.dbg_mag != nullptr) && ch == p.dbg_ch && m == p.dbg_frame;
        int l = lane;
        asm volatile("" : "+v"(l));                                         // LDS addresses are recomputed per frame instead of hoisted (see pv_wg_kernel.hip)
        const int LL = 64 * g + l;                                          // lane id of the 16-bins-per-lane side
        const float pfm = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(pf_next)));
        const double pf = (double)pfm;
        const int tmod = (int)(((long)p.t0_mod_n + (long)m * HOP) & (N - 1));
        const int pl = l + 4 * (l >> 4), ql = l + 4 * ((l + 15) >> 4);
        // W_2048^l and W_1024^l are squares of W_4096^l, formed from an opaque copy wherever they are needed instead of living in registers
        auto lane_twiddle = [&]() { double2 w = wN; asm volatile("" : "+v"(w.x), "+v"(w.y)); return w; };

        // ---- shift table of this wave's 1024 candidate bins, rebuilt only when f changes (image in the wave's own scratch, 32 bytes read back) ----
        {
            const unsigned pfb = __float_as_uint(pfm);
            if (!psh_valid || pfb != psh_key) {
                psh_key = pfb; psh_valid = true;
                short *IMG = reinterpret_cast<short *>(SA);
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int pk = 1024 * g + l + 64 * r;
                    const double ps = floor((double)pk * pf + 0.5);
                    const bool ok = (ps <= (double)H) && (ps >= -(double)(2 * N));
                    IMG[l + 64 * r] = ok ? (short)((int)ps - pk) : (short)0x4000;   // DROP pushes every target of the region out of range
                }
                wave_sync();
                typedef const volatile __attribute__((address_space(3))) v4u *lds_v4u_;
                dq0 = *(lds_v4u_)(SA + 32 * l);
                dq1 = *(lds_v4u_)(SA + 32 * l + 16);
                wave_sync();
            }
        }

        pv_prio(PH_FA);
        // ---- Hann (pv:55), this wave's half of the packed sequence split by parity, two 512-point fp64 FFTs, decimation-in-time stage ----
        double2 zlo[8], zhi[8];
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const v4f xw = raw[r] * hw[r];
            zlo[r] = double2{(double)xw.x, (double)xw.y};
            zhi[r] = double2{(double)xw.z, (double)xw.w};
        }
        fft512_wave<double, false>(zlo, S64, TW1, TW2, l);
        fft512_wave<double, false>(zhi, S64, TW1, TW2, l);
        {
            const double2 w1024 = csq(csq(lane_twiddle()));
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const double2 t = cmul(w1024, mul_w16<double, false>(zhi[r], r));
                const double2 e = zlo[r];
                zlo[r] = cadd(e, t);                                        // E_g[l + 64 r]
                zhi[r] = csub(e, t);                                        // E_g[l + 64 r + 512]
            }
        }
        pv_prio(PH_SPLITX);
        // ---- wave-local partner exchange: E_g[1024 - k] of k = l + 64 r is element 512 + (64 - l) + 64 (7 - r) ----
        double2 zm[8];
#pragma unroll
        for (int r = 0; r < 8; r++) S64[r * 64 + l] = zhi[r];
        wave_sync();
#pragma unroll
        for (int r = 0; r < 8; r++) zm[r] = S64[(7 - r) * 64 + 64 - l];     // (l = 0, r = 0) reads one element past the rows: replaced below
        const double2 e512 = zhi[0];                                        // lane 0: E_g[512], self-paired inside the wave
        wave_sync();
        // ---- forward exchange: the four groups the OTHER wave finishes, (E_g[k], E_g[1024 - k]) each, into this wave's own scratch ----
        if (g == 0) {
#pragma unroll
            for (int i = 0; i < 4; i++) { S64[(2 * i) * 64 + l] = zlo[4 + i]; S64[(2 * i + 1) * 64 + l] = zm[4 + i]; }
            if (l == 0) S64[512] = e512;
        } else {
#pragma unroll
            for (int i = 0; i < 4; i++) { S64[(2 * i) * 64 + l] = zlo[i]; S64[(2 * i + 1) * 64 + l] = zm[i]; }
        }
        __syncthreads();                                                   // barrier 1
        pv_prio(PH_SPLITM);
        float2 XA[4], XB[4], XC[4], XD[4];                                  // X[k], X[2048 - k], X[1024 - k], X[1024 + k] of this lane's four groups, rounded to fp32
        float2 xm0{0.f, 0.f}, xm1{0.f, 0.f};                                // wave 1, lane 0: X[512], X[1536]
        auto finish_groups = [&](auto gtag) {
            constexpr int G = decltype(gtag)::value;
            const double2 *SO64 = reinterpret_cast<const double2 *>(SO);
            const double2 wNf = lane_twiddle();
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int r = 4 * G + i;
                const double2 oa = SO64[(2 * i) * 64 + l], ob = SO64[(2 * i + 1) * 64 + l];
                const double2 a0 = G ? oa : zlo[r], b0 = G ? ob : zm[r], a1 = G ? zlo[r] : oa, b1 = G ? zm[r] : ob;
                // Z[k] = a0 + W^k a1, Z[1024 + k] = a0 - W^k a1, Z[1024 - k] = b0 - conj(W^k) b1, Z[2048 - k] = b0 + conj(W^k) b1.
                // The group's twiddles are formed once: v = W_4096^k = wN W_64^r (split pass), W^k = W_2048^k = v^2 (radix-2 stage)
                const double2 vk = mul_w64_f(wNf, r), wk = csq(vk);
                const double2 t = cmul(wk, a1);
                const double2 uc = cmul(wk, cconj(b1));                      // conj(conj(W^k) b1)
                const double2 Zk = cadd(a0, t), Ze = csub(a0, t);
                const double2 Zc{b0.x - uc.x, b0.y + uc.y}, Zd{b0.x + uc.x, b0.y - uc.y};
                // pair (k, 2048 - k), twiddle W_4096^k = wN W_64^r
                const double2 E{Zk.x + Zd.x, Zk.y - Zd.y}, O{Zk.x - Zd.x, Zk.y + Zd.y};
                const double2 WO = cmul(vk, O);
                double2 xa{E.x + WO.y, E.y - WO.x}, xb{E.x - WO.y, -(E.y + WO.x)};
                // pair (1024 - k, 1024 + k), twiddle W_4096^{1024 - k} = -j conj(W_4096^k): with q = W_4096^k conj(O2), -j conj(q) O2-term = (-q.y, -q.x)
                const double2 E2{Zc.x + Ze.x, Zc.y - Ze.y}, O2{Zc.x - Ze.x, Zc.y + Ze.y};
                const double2 q = cmul(vk, cconj(O2));
                double2 xc{E2.x - q.x, E2.y + q.y}, xd{E2.x + q.x, -(E2.y - q.y)};
                if (G == 0 && i == 0 && l == 0) {                           // k = 0: Z[0] = a0 + a1, Z[1024] = a0 - a1 (pre-halved)
                    const double2 Z0 = cadd(a0, a1), Z1 = csub(a0, a1);
                    xa = double2{2.0 * (Z0.x + Z0.y), 0.0};                 // X[0]
                    xb = double2{2.0 * (Z0.x - Z0.y), 0.0};                 // X[2048]
                    xc = double2{2.0 * Z1.x, -2.0 * Z1.y};                  // X[1024] = 2 conj(Z[1024]) (self-paired)
                    xd = xc;                                                // no fourth bin: its route is dropped below
                }
                MAG[MAG0 + pl + 80 * r] = (float)(xa.x * xa.x + xa.y * xa.y);
                MAG[MAG0 + 2560 - ql - 80 * r] = (float)(xb.x * xb.x + xb.y * xb.y);
                MAG[MAG0 + 1280 - ql - 80 * r] = (float)(xc.x * xc.x + xc.y * xc.y);
                if (!(G == 0 && i == 0) || l != 0) MAG[MAG0 + 1280 + pl + 80 * r] = (float)(xd.x * xd.x + xd.y * xd.y);
                XA[i] = float2{(float)xa.x, (float)xa.y};
                XB[i] = float2{(float)xb.x, (float)xb.y};
                XC[i] = float2{(float)xc.x, (float)xc.y};
                XD[i] = float2{(float)xd.x, (float)xd.y};
                if (dbg) {
                    const int k = l + 64 * r;
                    p.dbg_X[2 * k] = xa.x; p.dbg_X[2 * k + 1] = xa.y;
                    p.dbg_X[2 * (2048 - k)] = xb.x; p.dbg_X[2 * (2048 - k) + 1] = xb.y;
                    p.dbg_X[2 * (1024 - k)] = xc.x; p.dbg_X[2 * (1024 - k) + 1] = xc.y;
                    if (k != 0) { p.dbg_X[2 * (1024 + k)] = xd.x; p.dbg_X[2 * (1024 + k) + 1] = xd.y; }
                }
            }
            if (G == 1 && l == 0) {                                         // k = 512: Z[512] = a0 - j a1, Z[1536] = a0 + j a1, the pair (512, 1536), W_4096^512 = e^{-j pi/4}
                const double2 a0 = SO64[512], a1 = e512;
                const double2 Z5{a0.x + a1.y, a0.y - a1.x}, Z15{a0.x - a1.y, a0.y + a1.x};
                const double2 E{Z5.x + Z15.x, Z5.y - Z15.y}, O{Z5.x - Z15.x, Z5.y + Z15.y};
                const double hh = 0.70710678118654752440;
                const double2 WO{(O.x + O.y) * hh, (O.y - O.x) * hh};
                const double2 x5{E.x + WO.y, E.y - WO.x}, x15{E.x - WO.y, -(E.y + WO.x)};
                MAG[MAG0 + 640] = (float)(x5.x * x5.x + x5.y * x5.y);
                MAG[MAG0 + 1920] = (float)(x15.x * x15.x + x15.y * x15.y);
                xm0 = float2{(float)x5.x, (float)x5.y};
                xm1 = float2{(float)x15.x, (float)x15.y};
                if (dbg) { p.dbg_X[2 * 512] = x5.x; p.dbg_X[2 * 512 + 1] = x5.y; p.dbg_X[2 * 1536] = x15.x; p.dbg_X[2 * 1536 + 1] = x15.y; }
            }
        };
        if (g == 0) finish_groups(std::integral_constant<int, 0>{});
        else finish_groups(std::integral_constant<int, 1>{});
        const int rg = 4 * g;                                               // first register row of this wave's groups
        __syncthreads();                                                   // barrier 2: magnitudes complete; the published values have been read

        // ---- f < 1: stash the fp32 spectrum (region A is free until Y is zeroed) for the fast form of the above-Nyquist residue ----
        const bool collide = !(pf >= 1.0);
        if (collide) {
            float2 *XS = reinterpret_cast<float2 *>(smem + P_A);
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int k = l + 64 * (rg + i);
                XS[k] = XA[i]; XS[2048 - k] = XB[i]; XS[1024 - k] = XC[i];
                if (k != 0) XS[1024 + k] = XD[i];
            }
            if (g == 1 && l == 0) { XS[512] = xm0; XS[1536] = xm1; }
        }

        pv_prio(PH_PEAKS);
        // ---- peak flags (pv:95-116) for bins 16 LL .. 16 LL + 15, nearest peaks inside the wave ----
        int last_shift = 0;
        bool nonfinite = false;                                             // a magnitude of this wave's bins is Inf or NaN (see pv_wave_kernel.hip)
        unsigned rt[16];
        unsigned rtM = NOROUTE;
        int lastown[16], firstown[16], last_in, first_in, cprev, cnext;
        bool below_any, above_any;
        {
            unsigned mg[20];
            typedef const volatile __attribute__((address_space(3))) v2u *lds_v2u;
            typedef const volatile __attribute__((address_space(3))) v4u *lds_v4u;
            const v2u q0 = *(lds_v2u)(&MAG[MAG0 + 20 * LL - 6]);
            const v4u q1 = *(lds_v4u)(&MAG[MAG0 + 20 * LL]);
            const v4u q2 = *(lds_v4u)(&MAG[MAG0 + 20 * LL + 4]);
            const v4u q3 = *(lds_v4u)(&MAG[MAG0 + 20 * LL + 8]);
            const v4u q4 = *(lds_v4u)(&MAG[MAG0 + 20 * LL + 12]);
            const v2u q5 = *(lds_v2u)(&MAG[MAG0 + 20 * LL + 20]);
            mg[0] = q0.x; mg[1] = q0.y;
            mg[2] = q1.x; mg[3] = q1.y; mg[4] = q1.z; mg[5] = q1.w; mg[6] = q2.x; mg[7] = q2.y; mg[8] = q2.z; mg[9] = q2.w;
            mg[10] = q3.x; mg[11] = q3.y; mg[12] = q3.z; mg[13] = q3.w; mg[14] = q4.x; mg[15] = q4.y; mg[16] = q4.z; mg[17] = q4.w;
            mg[18] = q5.x; mg[19] = q5.y;
            unsigned pm[19];
#pragma unroll
            for (int j = 3; j < 19; j++) pm[j] = max(mg[j], mg[j + 1]);
            constexpr int NEGPD = -(8192 << 16), POSPD = 16384 << 16;       // "no peak on this side"
            int pd[16];
            int cur = NEGPD;
#pragma unroll
            for (int i = 0; i < 16; i++) {
                // candidates are 2 <= k < H - 2 = 2047 (pv:97-100): lane 0 drops i < 2, lane 127 drops i = 15
                const bool in_range = (i < 2) ? (LL != 0) : (i == 15) ? (LL != 127) : true;
                const bool fl = in_range & (max(max(mg[i], mg[i + 1]), pm[i + 3]) < mg[i + 2]);
                const unsigned w = (i < 8) ? dq0[i >> 1] : dq1[(i - 8) >> 1];
                pd[i] = (int)__builtin_amdgcn_perm((unsigned)(16 * LL + i), w, (i & 1) ? 0x05040302u : 0x05040100u);
                cur = fl ? pd[i] : cur;
                lastown[i] = cur;
                firstown[i] = fl ? 1 : 0;                                   // flag, turned into the next-peak word below
                if (dbg) { p.dbg_flags[16 * LL + i] = fl ? 1 : 0; p.dbg_mag[16 * LL + i] = __uint_as_float(mg[i + 2]); }
            }
            {
                unsigned mx = mg[2];
#pragma unroll
                for (int j = 3; j < 19; j += 2) mx = max(mx, pm[j]);
                nonfinite = __any(mx >= 0x7F800000u);
            }
            if (dbg && LL == 127) { p.dbg_flags[2048] = 0; p.dbg_mag[2048] = __uint_as_float(mg[18]); }
            int nx = POSPD;
#pragma unroll
            for (int i = 15; i >= 0; i--) { const bool fl = firstown[i] != 0; firstown[i] = nx; nx = fl ? pd[i] : nx; }
            last_in = cur; first_in = nx;
            const unsigned long long occ = __ballot(cur >= 0);
            const unsigned long long below = occ & ((1ull << l) - 1ull);
            const unsigned long long above = (l == 63) ? 0ull : (occ >> (l + 1));
            const int src_lo = below ? 63 - __clzll((long long)below) : 0;
            const int src_hi = above ? l + __ffsll((long long)above) : 0;
            cprev = __shfl(last_in, src_lo, 64); cnext = __shfl(first_in, src_hi, 64);
            below_any = below != 0ull; above_any = above != 0ull;
            // what the other wave needs: this wave's last peak (for the lanes of wave 1 with nothing below) and first peak (for wave 0, nothing above)
            const int wlast = occ ? __shfl(last_in, 63 - __clzll((long long)occ), 64) : NEGPD;
            const int wfirst = occ ? __shfl(first_in, __ffsll((long long)occ) - 1, 64) : POSPD;
            if (l == 0) { BND[2 * g] = wlast; BND[2 * g + 1] = wfirst; }
        }
        __syncthreads();                                                   // barrier 3: peak words across the wave boundary; every magnitude read is done; the stash is complete
        float2 s2v[4] = {float2{0.f, 0.f}, float2{0.f, 0.f}, float2{0.f, 0.f}, float2{0.f, 0.f}};
        bool bad = false;                                                  // f < 1: a gap of this wave's bins overlaps by more than its rising side
        {
            constexpr int NEGPD = -(8192 << 16), POSPD = 16384 << 16;
            const int o_last = BND[2 * (1 - g)], o_first = BND[2 * (1 - g) + 1], my_last = BND[2 * g];
            if (!below_any) cprev = (g == 1) ? o_last : NEGPD;
            if (!above_any) cnext = (g == 0) ? o_first : POSPD;
            const int lp = (g == 1) ? (my_last >= 0 ? my_last : o_last) : (o_last >= 0 ? o_last : my_last);   // the last peak of the frame
            const bool any_peak = lp >= 0;
            if (any_peak) last_shift = (int)(short)(lp & 0xFFFF);
            if (!any_peak) {
#pragma unroll
                for (int i = 0; i < 16; i++) rt[i] = NOROUTE;
            } else {
                auto route_of = [&](int b, int pp, int pn) -> unsigned {
                    const int own = (b - (pp >> 16) < (pn >> 16) - b) ? pp : pn;    // owner rule (pv:132-141)
                    const int delta = __builtin_amdgcn_sbfe(own, 0, 16);
                    return __builtin_amdgcn_perm((unsigned)__mul24(delta, tmod), (unsigned)(b + delta), 0x05040100u);
                };
#pragma unroll
                for (int i = 0; i < 16; i++) rt[i] = route_of(16 * LL + i, max(lastown[i], cprev), min(firstown[i], cnext));
                if (LL == 127) rtM = route_of(M, max(last_in, cprev), POSPD);       // source bin N/2: owner is the last peak
                if (collide) {
                    // f < 1: bit 31 of a route = "rising side" (source owned by the peak on its right), and this wave's share of the test that lets
                    // the scatter run as store-then-add instead of claim rounds (pv_wave_kernel.hip, "pairwise"; tests/test_pairwise_rule.py)
                    rtM &= 0x7FFFFFFFu;
#pragma unroll
                    for (int i = 0; i < 16; i++) {
                        const int pp = max(lastown[i], cprev), pn = min(firstown[i], cnext), b = 16 * LL + i;
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
            if (collide) {                                                  // fast form of the residue: positions N/2 + kk, kk = 1 + LL + 128 j (see pv_wave_kernel.hip)
                const float2 *XS = reinterpret_cast<const float2 *>(smem + P_A);
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int kk = 1 + LL + 128 * j;                         // kk in [1, 512]
                    const float2 x0 = XS[kk], x1 = XS[kk + 1024], x2 = XS[2048 - kk], x3 = XS[1024 - kk];
                    const float2 tsum{0.25f * ((x0.x - x1.x) + (x2.x - x3.x)), 0.25f * ((x0.y - x1.y) - (x2.y - x3.y))};
                    s2v[j] = cmul(tsum, s2w[j]);
                }
                __syncthreads();                                           // (f < 1 only, uniform in the workgroup) the stash is dead: Y may be zeroed
            }
        }
        pv_prio(PH_SCATTER);
        // ---- routes (aliasing the magnitudes) and the zeroed Y (pv:121) ----
#pragma unroll
        for (int j = 0; j < 4; j++) *reinterpret_cast<uint4 *>(&ROUTE[20 * LL + 4 * j]) = uint4{rt[4 * j], rt[4 * j + 1], rt[4 * j + 2], rt[4 * j + 3]};
        if (LL == 127) ROUTE[2560] = rtM;
#pragma unroll
        for (int r = 0; r < 8; r++) *reinterpret_cast<v4f *>(&Y[1024 * g + 2 * l + 128 * r]) = v4f{0.f, 0.f, 0.f, 0.f};
        if (LL == 127) Y[M] = float2{0.f, 0.f};
        int upper_end = H;
        if (last_shift < 0) { upper_end = H - last_shift; if (upper_end > N) upper_end = N; }      // DROP is positive
        if (collide) { const bool wbad = __any(bad); if (l == 0) BND[4 + g] = wbad ? 1 : 0; }
        __syncthreads();                                                   // barrier 4
        const bool pairwise = PV_PAIRWISE && collide && !(BND[4] | BND[5]);  // uniform in the workgroup
        // ---- shiftPeaks (pv:119-173): this lane's sources are the bins its groups produced ----
        if (!collide) {
            auto scatter = [&](auto mode_tag) {
                constexpr int MODE = decltype(mode_tag)::value;
                auto put = [&](unsigned rtv, float2 v) {
                    const unsigned tg = rtv & 0xFFFFu;
                    if (tg < (unsigned)H) {
                        float2 o = v;
                        if (MODE == 2) {
                            const unsigned sg = (rtv << 4) & 0x80000000u;
                            o = float2{__uint_as_float(__float_as_uint(v.x) ^ sg), __uint_as_float(__float_as_uint(v.y) ^ sg)};
                        } else if (MODE == 1) o = rotate4k<R>(rtv, v, ROT);
                        Y[tg] = o;
                    }
                };
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int r = rg + i;
                    put(ROUTE[pl + 80 * r], XA[i]);
                    put(ROUTE[2560 - ql - 80 * r], XB[i]);
                    put(ROUTE[1280 - ql - 80 * r], XC[i]);
                    const unsigned rd = ROUTE[1280 + pl + 80 * r];
                    put((g == 0 && i == 0 && l == 0) ? NOROUTE : rd, XD[i]);
                }
                if (g == 1 && l == 0) { put(ROUTE[640], xm0); put(ROUTE[1920], xm1); }
            };
            if (tmod == 0) scatter(std::integral_constant<int, 0>{});
            else if (tmod == N / 2) scatter(std::integral_constant<int, 2>{});
            else scatter(std::integral_constant<int, 1>{});
        } else {
            unsigned rs[18];
            float2 ys[18];
            int id[18];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int r = rg + i, k = l + 64 * r;
                id[4 * i] = k; rs[4 * i] = ROUTE[pl + 80 * r]; ys[4 * i] = XA[i];
                id[4 * i + 1] = 2048 - k; rs[4 * i + 1] = ROUTE[2560 - ql - 80 * r]; ys[4 * i + 1] = XB[i];
                id[4 * i + 2] = 1024 - k; rs[4 * i + 2] = ROUTE[1280 - ql - 80 * r]; ys[4 * i + 2] = XC[i];
                id[4 * i + 3] = 1024 + k; rs[4 * i + 3] = (k == 0) ? NOROUTE : ROUTE[1280 + pl + 80 * r]; ys[4 * i + 3] = XD[i];
            }
            const bool mid = (g == 1 && l == 0);
            id[16] = 512; rs[16] = mid ? ROUTE[640] : NOROUTE; ys[16] = xm0;
            id[17] = 1536; rs[17] = mid ? ROUTE[1920] : NOROUTE; ys[17] = xm1;
#pragma unroll
            for (int i = 0; i < 18; i++) ys[i] = rotate4k<R>(rs[i], ys[i], ROT);
            if (pairwise) {
                // every collision of this frame is one falling-side source against one rising-side source: the falling side and the residue (it
                // continues the falling side of the last peak) store into the zeroed Y, one barrier, the rising side adds.  No claim words, one
                // barrier instead of two per round.
                unsigned key[18];
#pragma unroll
                for (int i = 0; i < 18; i++) key[i] = rs[i] & 0x8000FFFFu;
#pragma unroll
                for (int i = 0; i < 18; i++) if (key[i] < (unsigned)H) Y[key[i]] = ys[i];
                if (upper_end > H && upper_end <= H + N / 8) {
                    const int up_delta = last_shift;
                    const unsigned up_ridx = (unsigned)((up_delta & (N - 1)) * tmod) & (N - 1);
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const int b = M + 1 + LL + 128 * j, tgt = b + up_delta;
                        const unsigned rtj = (b < upper_end && tgt >= 0 && tgt < H) ? ((up_ridx << 16) | (unsigned)tgt) : NOROUTE;
                        if (rtj != NOROUTE) Y[tgt] = rotate4k<R>(rtj, s2v[j], ROT);
                        if (dbg && b < upper_end) { p.dbg_X[2 * b] = s2v[j].x; p.dbg_X[2 * b + 1] = s2v[j].y; }
                    }
                }
                __syncthreads();
#pragma unroll
                for (int h0 = 0; h0 < 18; h0 += 9) {
                    float2 o[9];
#pragma unroll
                    for (int i = h0; i < h0 + 9; i++) o[i - h0] = Y[min(rs[i] & 0xFFFFu, (unsigned)M)];
#pragma unroll
                    for (int i = h0; i < h0 + 9; i++)
                        if (key[i] - 0x80000000u < (unsigned)H) Y[key[i] - 0x80000000u] = float2{o[i - h0].x + ys[i].x, o[i - h0].y + ys[i].y};
                }
                if (upper_end > H + N / 8)
                    residue_scatter_pair<R>(src.in, src.hist, src.hist_len, (long)(m + 1) * HOP - N, p.hann, p.tw32, (int)threadIdx.x, upper_end, last_shift,
                                            (unsigned)((last_shift & (N - 1)) * tmod) & (N - 1), dbg ? p.dbg_X : nullptr, true);
            } else {
            __syncthreads();                                               // every route read is done: the region becomes the claim words
#pragma unroll
            for (int j = 0; j < 16; j++) CLAIM[threadIdx.x + 128 * j] = 0xFFFFFFFFu;
            if (threadIdx.x == 0) CLAIM[M] = 0xFFFFFFFFu;
            claim_rounds_pair<18>(rs, ys, id, Y, CLAIM);                    // (its first barrier orders the fill before the first claims)
            if (upper_end > H) {                                            // uniform in the workgroup
                const int up_delta = last_shift;
                const unsigned up_ridx = (unsigned)((up_delta & (N - 1)) * tmod) & (N - 1);
                if (upper_end <= H + N / 8) {
                    unsigned rt2[4];
                    float2 ys2[4];
                    int id2[4];
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const int b = M + 1 + LL + 128 * j, tgt = b + up_delta;
                        rt2[j] = (b < upper_end && tgt >= 0 && tgt < H) ? ((up_ridx << 16) | (unsigned)tgt) : NOROUTE;
                        ys2[j] = rotate4k<R>(rt2[j], s2v[j], ROT);
                        id2[j] = b - N / 2;
                        if (dbg && b < upper_end) { p.dbg_X[2 * b] = s2v[j].x; p.dbg_X[2 * b + 1] = s2v[j].y; }
                    }
                    claim_rounds_pair<4>(rt2, ys2, id2, Y, CLAIM);
                } else {
                    residue_scatter_pair<R>(src.in, src.hist, src.hist_len, (long)(m + 1) * HOP - N, p.hann, p.tw32, (int)threadIdx.x, upper_end, up_delta, up_ridx,
                                            dbg ? p.dbg_X : nullptr, false);
                }
            }
            }
        }
        if (nonfinite && l == 0) Y[1 + g] = float2{__uint_as_float(0x7FC00000u), __uint_as_float(0x7FC00000u)};   // the reference's frame is NaN: so is this one
        __syncthreads();                                                   // barrier 5: Y complete
        if (dbg) {
#pragma unroll
            for (int r = 0; r < 16; r++) { const int k = (int)threadIdx.x + 128 * r; p.dbg_Y[2 * k] = Y[k].x; p.dbg_Y[2 * k + 1] = Y[k].y; }
            if (threadIdx.x == 0) { p.dbg_Y[2 * M] = Y[M].x; p.dbg_Y[2 * M + 1] = Y[M].y; }
        }
        pv_prio(PH_C2R);
        // ---- c2r pre-pass (bundle:69-76,102-114 folded) and the decimation-in-frequency stage for this wave's four groups, packed fp32 ----
        //      Zc[k] = SC ((Yk + Ym*) + j e^{+2 pi j k/N} (Yk - Ym*)), m = M - k;  A[q] = Zc[q] + Zc[q + 1024], B[q] = (Zc[q] - Zc[q + 1024]) e^{+2 pi j q / 2048}
        pk::c32 vkO[4], vcO[4];                                            // V[k], V[1024 - k] of this wave's kind (A for wave 0, B for wave 1) for its own four groups
        pk::c32 v512{0.f, 0.f};
        {
            const pk::c32 scsc{SC, SC};
            const double2 wNf = lane_twiddle(), wM = csq(wNf);
            const pk::c32 cNs{(float)wNf.x * SC, -(float)wNf.y * SC};       // e^{+2 pi j l / 4096} SC
            const pk::c32 cM{(float)wM.x, -(float)wM.y};                    // e^{+2 pi j l / 2048}
            const pk::c32 *Yc = reinterpret_cast<const pk::c32 *>(Y);
            pk::c32 *PUB = reinterpret_cast<pk::c32 *>(smem + P_B + 4608 * g);
            auto inverse_groups = [&](auto gtag) {
                constexpr int G = decltype(gtag)::value;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int r = 4 * G + i, k = l + 64 * r;
                    pk::c32 yk = Yc[k], yd = Yc[2048 - k];
                    const pk::c32 yc = Yc[1024 - k], ye = Yc[1024 + k];
                    if (G == 0 && i == 0 && l == 0) { yk.y = 0.f; yd.y = 0.f; }
                    const pk::c32 E = pk::add_conj(yk, yd), O = pk::sub_conj(yk, yd);
                    const pk::c32 vks = mul_w64_i(cNs, r);                    // e^{+2 pi j k / 4096} SC, formed once for both pairs of the group
                    const pk::c32 wki = mul_w32_i(cM, r);                     // e^{+2 pi j k / 2048}, once for both differences
                    const pk::c32 c = pk::cmul(O, vks);
                    const pk::c32 Zk = pk::fma_addj(E, scsc, c);             // Zc[k]
                    const pk::c32 Zd = pk::fma_conj_subj(E, scsc, c);        // Zc[2048 - k]
                    // pair (1024 - k, 1024 + k): e^{+2 pi j (1024 - k)/N} = j conj(e^{+2 pi j k/N}); with pp = e^{+2 pi j k/N} SC (Y[1024+k] - conj(Y[1024-k])):
                    // Zc[1024 - k] = E2 SC + conj(pp), Zc[1024 + k] = conj(E2 SC) - pp
                    const pk::c32 E2 = pk::add_conj(yc, ye), O2c = pk::sub_conj(ye, yc);
                    const pk::c32 pp = pk::cmul(O2c, vks);
                    pk::c32 Zc{E2.x * SC + pp.x, E2.y * SC - pp.y};
                    pk::c32 Ze{E2.x * SC - pp.x, -(E2.y * SC) - pp.y};
                    if (G == 0 && i == 0 && l == 0) { Ze = pk::c32{2.0f * yc.x * SC, -2.0f * yc.y * SC}; Zc = Ze; }   // bin 1024 pairs with itself: 2 conj(Y[1024]) SC
                    const pk::c32 Ak = pk::add(Zk, Ze), Dk = pk::sub(Zk, Ze);
                    const pk::c32 Bk = pk::cmul(Dk, wki);
                    const pk::c32 Ac = pk::add(Zc, Zd), Dc = pk::sub(Zc, Zd);
                    const pk::c32 sB = pk::cmul(pk::c32{Dc.x, -Dc.y}, wki);
                    const pk::c32 Bc{-sB.x, sB.y};                           // (Zc[1024-k] - Zc[2048-k]) e^{+2 pi j (1024 - k)/2048}
                    if (G == 0) {
                        vkO[i] = Ak; vcO[i] = Ac;
                        PUB[(2 * i) * 64 + l] = Bk; PUB[(2 * i + 1) * 64 + l] = Bc;
                    } else {
                        vkO[i] = Bk; vcO[i] = Bc;
                        PUB[(2 * i) * 64 + l] = Ak; PUB[(2 * i + 1) * 64 + l] = Ac;
                    }
                }
                if (G == 1 && l == 0) {                                     // pair (512, 1536): twiddle e^{+j pi/4} SC; A[512] = sum, B[512] = j (difference)
                    const pk::c32 y5 = Yc[512], y15 = Yc[1536];
                    const pk::c32 E = pk::add_conj(y5, y15), O = pk::sub_conj(y5, y15);
                    const float hs = 0.70710678118654752440f * SC;
                    const pk::c32 c{(O.x - O.y) * hs, (O.x + O.y) * hs};
                    const pk::c32 Z5 = pk::fma_addj(E, scsc, c), Z15 = pk::fma_conj_subj(E, scsc, c);
                    const pk::c32 d = pk::sub(Z5, Z15);
                    PUB[512] = pk::add(Z5, Z15);
                    v512 = pk::c32{-d.y, d.x};
                }
            };
            if (g == 0) inverse_groups(std::integral_constant<int, 0>{});
            else inverse_groups(std::integral_constant<int, 1>{});
        }
        __syncthreads();                                                   // barrier 6: inverse exchange
        pk::c32 zA[8], zB[8];
        {
            const pk::c32 *OP = reinterpret_cast<const pk::c32 *>(smem + P_B + 4608 * (1 - g));
            pk::c32 vkR[4], vcR[4];                                         // the same for the four groups the other wave finished
#pragma unroll
            for (int i = 0; i < 4; i++) { vkR[i] = OP[(2 * i) * 64 + l]; vcR[i] = OP[(2 * i + 1) * 64 + l]; }
            if (g == 0 && l == 0) v512 = OP[512];
            // wave-local hand-over: V[1024 - k] of the pair (l', r') is element 512 + (64 - l') + 64 (7 - r'); register rows only enter the addresses
            pk::c32 *XCH = reinterpret_cast<pk::c32 *>(SA + 4608);
#pragma unroll
            for (int i = 0; i < 4; i++) { XCH[(rg + i) * 64 + l] = vcO[i]; XCH[(4 - rg + i) * 64 + l] = vcR[i]; }
            wave_sync();
#pragma unroll
            for (int r = 0; r < 8; r++) zB[7 - r] = XCH[r * 64 + 64 - l];    // lane 0 pairs with itself one register higher
            if (l == 0) zB[0] = v512;
            wave_sync();
            const bool hi = (g != 0);                                       // wave 1 owns rows 4..7, wave 0 rows 0..3: a select, not a branch (no merge of register arrays)
#pragma unroll
            for (int i = 0; i < 4; i++) {
                zA[i] = pk::c32{hi ? vkR[i].x : vkO[i].x, hi ? vkR[i].y : vkO[i].y};
                zA[4 + i] = pk::c32{hi ? vkO[i].x : vkR[i].x, hi ? vkO[i].y : vkR[i].y};
            }
        }
        pv_prio(PH_IA);
        // ---- this wave's 1024-point inverse: decimation-in-frequency stage, two 512-point packed-fp32 inverse FFTs ----
        {
            const double2 w1024 = csq(csq(lane_twiddle()));
            const pk::c32 c1024{(float)w1024.x, -(float)w1024.y};
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const pk::c32 a = pk::add(zA[r], zB[r]), d = pk::sub(zA[r], zB[r]);
                zA[r] = a;
                zB[r] = pk::cmul(mul_w16_i(d, r), c1024);
            }
        }
        {   // every global access of the next frame, issued here
            const int mn = (m + 1 < last_out) ? m + 1 : m;
            int sof = so;
            asm volatile("" : "+v"(sof));
            load_rows(raw, mn, sof);
            load_hann(hw, sof);
            pf_next = pitch_row[mn];
        }
        fft512_wave_inv_pk64(zA, reinterpret_cast<pk::c32 *>(SA), TW1, TW2, l);
        fft512_wave_inv_pk64(zB, reinterpret_cast<pk::c32 *>(SA), TW1, TW2, l);
        pv_prio(PH_OLA);
        // ---- Hann (pv:67), overlap-add in reference order (ola:149-157), emit (ola:111-118), shift (ola:130-137) ----
        {
            const bool emit_out = (m >= emit_v);
            v4f fr[8];
#pragma unroll
            for (int r = 0; r < 8; r++)                                    // rounded to fp32 BEFORE the accumulation like the reference's Float32Array (pv:67): no contraction into the adds
                fr[r] = v4f{__fmul_rn(zA[r].x, hw[r].x), __fmul_rn(zA[r].y, hw[r].y), __fmul_rn(zB[r].x, hw[r].z), __fmul_rn(zB[r].y, hw[r].w)};
#pragma unroll
            for (int r = 0; r < S_ROWS; r++) {
                const v4f o = acc[r] + fr[r];
                if (emit_out) {
                    char *rb = reinterpret_cast<char *>(outp + (long)m * HOP + 512 * r);      // wave-uniform row base + this lane's byte offset
                    const unsigned ob = 4u * (unsigned)(8 * l + 2 * g);
                    if (vec_out) {                                              // plain stores: the two waves fill alternate 8-byte pieces of every 32-byte sector and
                        *reinterpret_cast<v2f *>(rb + ob) = v2f{o.x, o.y};              // L2 merges them (non-temporal stores reached HBM as two partial writes per sector:
                        *reinterpret_cast<v2f *>(rb + ob + 16) = v2f{o.z, o.w};         // WRITE_SIZE 2.0 x the output)
                    } else { float *dst = reinterpret_cast<float *>(rb + ob); dst[0] = o.x; dst[1] = o.y; dst[4] = o.z; dst[5] = o.w; }
                }
            }
#pragma unroll
            for (int r = 0; r < LROWS; r++) {
                const int s = r + S_ROWS;
                acc[r] = (s < LROWS) ? acc[s] + fr[s] : fr[s];
            }
        }
    }

    if (chunk == p.nchunks - 1) {
        // the next call's history is read in one batch before anything is stored (interleaved, every load is waited for on its own: pv_wave2k_kernel.hip)
        v4f hrow[LROWS > 0 ? LROWS : 1];
#pragma unroll
        for (int r = 0; r < LROWS; r++) {
            const long s = (long)p.nhops * HOP - L + so + 512 * r;
            hrow[r] = v4f{src.at(s), src.at(s + 1), src.at(s + 4), src.at(s + 5)};
        }
#pragma unroll
        for (int r = 0; r < LROWS; r++) asm volatile("" : "+v"(hrow[r]));
#pragma unroll
        for (int r = 0; r < LROWS; r++) {
            float *a = p.acc_out + (long)ch * L + so + 512 * r;
            a[0] = acc[r].x; a[1] = acc[r].y; a[4] = acc[r].z; a[5] = acc[r].w;
            float *hs = p.hist_out + (long)ch * L + so + 512 * r;
            hs[0] = hrow[r].x; hs[1] = hrow[r].y; hs[4] = hrow[r].z; hs[5] = hrow[r].w;
        }
    }
    pv_signal_done<true>(p.done, p.done_seq, chain);
}

template <int S_ROWS, bool AUX>
hipError_t launch_pair(const PvKernelParams &p, int nch, int nchunks, hipStream_t st)
{
    static std::atomic<bool> attr_done[16];
    auto k = pv_pair_kernel<S_ROWS, AUX>;
    {
        const hipError_t e = pv_set_dynamic_lds_once(attr_done, reinterpret_cast<const void *>(k), (int)pv_pair_lds_bytes());
        if (e != hipSuccess) return e;
    }
    PvKernelParams q = p;
    q.nchunks = nchunks;
    q.nch = nch;
    const long chains = (long)nch * nchunks;
    hipLaunchKernelGGL(k, dim3((unsigned)chains, 1, 1), dim3(128, 1, 1), pv_pair_lds_bytes(), st, q);
    return hipGetLastError();
}

}  // namespace

size_t pv_pair_lds_bytes() { return P_BYTES; }
int pv_pair_threads() { return 128; }
bool pv_pair_supported(int log2n, int hop) { return log2n == 12 && (hop == 512 || hop == 1024 || hop == 2048 || hop == 4096); }

hipError_t pv_launch_pair(const PvKernelParams &p, int nch, int nchunks, hipStream_t st)
{
    const bool aux = (p.dbg_mag != nullptr);                              // pv_debug_frame: the tap instance of the SAME kernel
    switch (p.hop) {
    case 512: return aux ? launch_pair<1, true>(p, nch, nchunks, st) : launch_pair<1, false>(p, nch, nchunks, st);
    case 1024: return aux ? launch_pair<2, true>(p, nch, nchunks, st) : launch_pair<2, false>(p, nch, nchunks, st);
    case 2048: return aux ? launch_pair<4, true>(p, nch, nchunks, st) : launch_pair<4, false>(p, nch, nchunks, st);
    case 4096: return aux ? launch_pair<8, true>(p, nch, nchunks, st) : launch_pair<8, false>(p, nch, nchunks, st);
    default: return hipErrorInvalidValue;
    }
}
