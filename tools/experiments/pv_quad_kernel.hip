// pv_quad_kernel.hip -- FOUR wavefronts per 8192-point frame chain: N = 8192, hop in {1024, 2048, 4096, 8192} (BASELINE configs[4]).
//
// pv_pair_kernel.hip's recipe with a radix-4 meeting point.  Wave g in {0..3} owns z_g[n''] = z[4n'' + g] of the packed sequence
// z[n] = xw[2n] + j xw[2n+1], n < 4096, and runs pv_wave2k's 1024-point complex FFT on it: E_g.  With T_h[k] = W_4096^{h k} E_h[k]:
//
//   Z[k + 1024 q] = sum_h (-j)^{h q} T_h[k],   Z[1024 - k + 1024 q] = sum_h (-j)^{h q} U_h[k],  U_h[k] = (-j)^h conj(W_4096^{h k}) E_h[1024 - k]
//
// and the real-FFT split pairs bin k + 1024 q with bin 1024 - k + 1024 (3 - q): the EIGHT bins {k + 1024 q} u {1024 - k + 1024 q} need exactly
// E_h[k], E_h[1024 - k] of the four waves.  Every wave brings E_g[k], E_g[1024 - k] into one lane (wave-local partner exchange), PUBLISHES all
// eight (k, 1024 - k) pairs of its registers, and FINISHES two of the eight register groups (r = 2g, 2g + 1; k = l + 64 r): it gathers the 16
// values of a group from LDS -- its own included, so that nothing in the code depends on the wave id except addresses -- and every lane holds 16
// finished bins.  The publish takes two rounds (48 KB of fp64 do not fit one buffer): rows 0..3 into the waves' own scratches, rows 4..7 into the
// (still free) magnitude region after the first barrier.  The inverse mirrors it in packed fp32 and one round: the owner of a group runs the c2r
// pre-pass for its eight bins and the radix-4 decimation-in-frequency stage, and publishes V_h[k], V_h[1024 - k] for all four waves; wave h reads
// its 1024-point spectrum straight into the register layout of its inverse FFT (the conjugate-pair hand-over of the smaller kernels is folded
// into the read addresses).  Eight workgroup barriers per f >= 1 frame; two workgroups = 8 waves per CU.
//
// Samples: lane l, register row r of wave g holds z[8n' + g] and z[8n' + 4 + g], n' = l + 64 r, i.e. the float2 at samples 16n' + 2g and
// 16n' + 8 + 2g; a register row is 1024 samples; plain stores (the four waves' 8-byte pieces merge in L2).
// Everything between the transforms is pv_pair_kernel's pipeline with the lane id L = 64 g + l (see there and pv_wave2k_kernel.hip).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "pv_kernels.h"
#include "pv_device_common.h"
#include "pv_pk_math.h"
#include "pv_wave_fft.h"

namespace {

constexpr int N8 = 8192, M8 = 4096, H8 = 4097, LOG2N8 = 13;

// workgroup LDS (byte offsets); one frame chain per workgroup of four waves
constexpr int Q_TW1 = 0;                          // double2[8*64]  W_512^{l k}
constexpr int Q_TW2 = Q_TW1 + 8 * 64 * 16;        // double2[8*8]   W_64^{n0 k}
constexpr int Q_ROT = Q_TW2 + 8 * 8 * 16;         // float2[16]     exp(+2 pi j q / 16)  (hop = N/8: R = 8)
constexpr int Q_BND = Q_ROT + 16 * 8;             // int[8] last / first peak word of each wave | double2[4] E_h[512] | c32[4] V_h[512]
constexpr int Q_A = Q_BND + 192;                  // 36864 B: four wave scratches (9216 each) | Y float2[4097] | spectrum stash (f < 1)
constexpr int Q_B = Q_A + 36864;                  // 32768 B: forward publish round 2 (4 x 8192) | mags / routes, padded layout (5129 words) | claim words u32[4097]
                                                  //          | inverse publish 4 x 8192 | residue quarter float2[2048]
constexpr int Q_BYTES = Q_B + 32768;              // 79168: two workgroups per CU

constexpr int MAG0Q = 8;                          // padded layout P(bin) = bin + 4 (bin >> 4), see pv_wave2k_kernel.hip

__device__ __forceinline__ double2 csq8(double2 a) { return double2{(a.x - a.y) * (a.x + a.y), 2.0 * a.x * a.y}; }

// exp(-2 pi j n / 64), n < 22, and exp(-2 pi j n / 128), n < 8: the register-row parts of the group twiddles.  The row r = 2g + i is wave-uniform
// but only known at run time, so they come from constant memory (scalar loads) instead of eight specialised copies of the group code.
__device__ __constant__ double QW64[22][2] = {
    {1.00000000000000000000, -0.00000000000000000000},
    {0.99518472667219692873, -0.09801714032956060363},
    {0.98078528040323043058, -0.19509032201612824808},
    {0.95694033573220882438, -0.29028467725446233105},
    {0.92387953251128673848, -0.38268343236508978178},
    {0.88192126434835504956, -0.47139673682599764204},
    {0.83146961230254523567, -0.55557023301960217765},
    {0.77301045336273699338, -0.63439328416364548779},
    {0.70710678118654757274, -0.70710678118654746172},
    {0.63439328416364548779, -0.77301045336273699338},
    {0.55557023301960228867, -0.83146961230254523567},
    {0.47139673682599780857, -0.88192126434835493853},
    {0.38268343236508983729, -0.92387953251128673848},
    {0.29028467725446233105, -0.95694033573220893540},
    {0.19509032201612833135, -0.98078528040323043058},
    {0.09801714032956077016, -0.99518472667219681771},
    {0.00000000000000006123, -1.00000000000000000000},
    {-0.09801714032956064526, -0.99518472667219692873},
    {-0.19509032201612819257, -0.98078528040323043058},
    {-0.29028467725446216452, -0.95694033573220893540},
    {-0.38268343236508972627, -0.92387953251128673848},
    {-0.47139673682599769755, -0.88192126434835504956}};
__device__ __constant__ double QW128[8][2] = {
    {1.00000000000000000000, -0.00000000000000000000},
    {0.99879545620517240501, -0.04906767432741801493},
    {0.99518472667219692873, -0.09801714032956060363},
    {0.98917650996478101444, -0.14673047445536174793},
    {0.98078528040323043058, -0.19509032201612824808},
    {0.97003125319454397424, -0.24298017990326387094},
    {0.95694033573220882438, -0.29028467725446233105},
    {0.94154406518302080631, -0.33688985339222005111}};
__device__ __forceinline__ double2 qw64(int n) { return double2{QW64[n][0], QW64[n][1]}; }
__device__ __forceinline__ double2 qw128(int n) { return double2{QW128[n][0], QW128[n][1]}; }
__device__ __forceinline__ pk::c32 qw64i(int n) { return pk::c32{(float)QW64[n][0], -(float)QW64[n][1]}; }      // conjugated, fp32
__device__ __forceinline__ pk::c32 qw128i(int n) { return pk::c32{(float)QW128[n][0], -(float)QW128[n][1]}; }

__device__ __forceinline__ pk::c32 mul_w16_qi(pk::c32 o, int r)         // o * exp(+2 pi j r / 16), r = 0..7
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
// o * W_8^q (forward, fp64) and o * conj(W_8^q) (inverse, fp32), q = 0..3
__device__ __forceinline__ double2 mul_w8_f(double2 o, int q)
{
    const double h = 0.70710678118654752440;
    switch (q) {
    case 0: return o;
    case 1: return double2{(o.x + o.y) * h, (o.y - o.x) * h};
    case 2: return double2{o.y, -o.x};
    default: return double2{(o.y - o.x) * h, -(o.x + o.y) * h};
    }
}
__device__ __forceinline__ pk::c32 mul_w8_i(pk::c32 o, int q)
{
    const float h = 0.70710678118654752440f;
    switch (q) {
    case 0: return o;
    case 1: return pk::c32{(o.x - o.y) * h, (o.x + o.y) * h};
    case 2: return pk::c32{-o.y, o.x};
    default: return pk::c32{-(o.x + o.y) * h, (o.x - o.y) * h};
    }
}

// 512-point inverse wave FFT in packed fp32 with twiddles rounded on the fly from the fp64 tables (see pv_pair_kernel.hip)
__device__ __forceinline__ void fft512_wave_inv_pk64q(pk::c32 (&a)[8], pk::c32 *S, const double2 *TW1, const double2 *TW2, int l)
{
    const int lh = l >> 3, ll = l & 7;
    v4f *S4 = reinterpret_cast<v4f *>(S);
    pk::radix8_inv(a);
#pragma unroll
    for (int k = 1; k < 8; k++) { const double2 w = TW1[k * 64 + l]; a[k] = pk::cmul(a[k], pk::c32{(float)w.x, -(float)w.y}); }
#pragma unroll
    for (int j = 0; j < 4; j++) S4[j * TPP + l] = v4f{a[2 * j].x, a[2 * j].y, a[2 * j + 1].x, a[2 * j + 1].y};
    wave_sync();
#pragma unroll
    for (int n = 0; n < 8; n++) a[n] = S[2 * ((lh >> 1) * TPP + 8 * n + ll) + (lh & 1)];
    wave_sync();
    pk::radix8_inv(a);
#pragma unroll
    for (int k = 1; k < 8; k++) { const double2 w = TW2[k * 8 + ll]; a[k] = pk::cmul(a[k], pk::c32{(float)w.x, -(float)w.y}); }
#pragma unroll
    for (int j = 0; j < 4; j++) S4[j * TPP + lh * 8 + ((ll + lh) & 7)] = v4f{a[2 * j].x, a[2 * j].y, a[2 * j + 1].x, a[2 * j + 1].y};
    wave_sync();
#pragma unroll
    for (int n = 0; n < 8; n++) a[n] = S[2 * ((lh >> 1) * TPP + ll * 8 + ((n + ll) & 7)) + (lh & 1)];
    wave_sync();
    pk::radix8_inv(a);
}

template <int R_>
__device__ __forceinline__ float2 rotate8k(unsigned route, float2 v, const float2 *ROT)
{
    if (R_ == 1) return v;
    if (R_ == 2) {
        const unsigned sg = (route << 3) & 0x80000000u;                     // top bit of the 13-bit rotation index = bit 28 of the route
        return float2{__uint_as_float(__float_as_uint(v.x) ^ sg), __uint_as_float(__float_as_uint(v.y) ^ sg)};
    }
    if (R_ == 4) return rotate_route<4, LOG2N8>(route, v, nullptr);
    return cmul(v, ROT[(route >> 25) & 15u]);
}

// atomic-MIN claim rounds of the workgroup (see pv_pair_kernel.hip / pv_wg_kernel.hip); CLAIM[0..H) all-ones on entry and on exit
template <int NS>
__device__ __forceinline__ void claim_rounds_quad(const unsigned (&rt)[NS], const float2 (&ys)[NS], const int (&id)[NS], float2 *Y, unsigned *CLAIM)
{
    unsigned pend = 0;
    unsigned tg[NS];
#pragma unroll
    for (int r = 0; r < NS; r++) {
        const unsigned t = rt[r] & 0xFFFFu;
        const bool ok = t < (unsigned)H8;
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

__device__ __forceinline__ int digitrev4_8k(int v, int nd)
{
    const unsigned r = __brev((unsigned)v) >> (32 - 2 * nd);
    return (int)(((r & 0x55555555u) << 1) | ((r >> 1) & 0x55555555u));
}

// Rare path: the above-Nyquist residue rebuilt one quarter of the buffer at a time (log2 N odd: radix-2 base blocks, bundle:447-463, then the radix-4
// stages with their predicated stores, bundle:329-441); see residue_scatter_pair.  All four waves run it (barriers inside).
template <int R_>
__device__ __attribute__((noinline)) PV_NO_DS_MERGE void residue_scatter_quad(const float *in, const float *hist, int hist_len, long s0, const float *__restrict__ hann,
                                                                               const float2 *__restrict__ tw32, int t, int upper_end, int up_delta, unsigned up_ridx)
{
    constexpr int N = N8, H = H8, QN = N / 4, T = 256;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float2 *Y = reinterpret_cast<float2 *>(smem + Q_A);
    unsigned *CLAIM = reinterpret_cast<unsigned *>(smem + Q_B);
    float2 *Q = reinterpret_cast<float2 *>(smem + Q_B);
    const float2 *ROT = reinterpret_cast<const float2 *>(smem + Q_ROT);
    const WaveSrc src{in, hist, hist_len};
    for (int base = N / 2; base < N && base < upper_end; base += QN) {
        __syncthreads();                                                   // the claim words of the previous scatter are done with
#pragma unroll
        for (int it = 0; it < 4; it++) {                                   // QN / 2 = 1024 radix-2 blocks per quarter
            const int lb = t + T * it, blk = base / 2 + lb;
            const int off = digitrev4_8k(blk, (LOG2N8 - 1) / 2);
            const float a = src.at(s0 + off) * hann[off], b = src.at(s0 + off + N / 2) * hann[off + N / 2];
            Q[2 * lb] = float2{a + b, 0.f};
            Q[2 * lb + 1] = float2{a - b, 0.f};
        }
        __syncthreads();
        for (int log2m = 3; log2m <= LOG2N8 - 2; log2m += 2) {             // block sizes 8, 32, 128, 512, 2048 inside the quarter
            const int q = (1 << log2m) >> 2, hq = q >> 1;
            const int nblocks = QN >> log2m;
            const int tws = LOG2N8 - log2m;
            for (int u = t; u < nblocks * (hq + 1); u += T) {
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
        unsigned rt[8];
        float2 ys[8];
        int id[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int b = base + t + T * j, tgt = b + up_delta;
            rt[j] = (b >= H && b < upper_end && tgt >= 0 && tgt < H) ? ((up_ridx << 16) | (unsigned)tgt) : NOROUTE;
            ys[j] = rotate8k<R_>(rt[j], Q[t + T * j], ROT);
            id[j] = b - N / 2;
        }
        __syncthreads();                                                   // the quarter is in registers: its space becomes the claim words again
#pragma unroll
        for (int j = 0; j < 16; j++) CLAIM[t + T * j] = 0xFFFFFFFFu;
        if (t == 0) CLAIM[M8] = 0xFFFFFFFFu;
        claim_rounds_quad<8>(rt, ys, id, Y, CLAIM);
    }
}

// S_ROWS = hop / 1024
template <int S_ROWS>
__global__ __launch_bounds__(256, 2) PV_NO_DS_MERGE void pv_quad_kernel(const PvKernelParams p)
{
    constexpr int N = N8, M = M8, H = H8;
    constexpr int HOP = 1024 * S_ROWS, R = N / HOP, LROWS = 8 - S_ROWS, L = N - HOP;
    const int lane = threadIdx.x & 63;
    const int g = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long chain = blockIdx.x;
    const int ch = (int)(chain / p.nchunks), chunk = (int)(chain - (long)ch * p.nchunks);

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const double2 *TW1 = reinterpret_cast<const double2 *>(smem + Q_TW1);
    const double2 *TW2 = reinterpret_cast<const double2 *>(smem + Q_TW2);
    const float2 *ROT = reinterpret_cast<const float2 *>(smem + Q_ROT);
    volatile int *BND = reinterpret_cast<volatile int *>(smem + Q_BND);
    double2 *E512 = reinterpret_cast<double2 *>(smem + Q_BND + 32);        // E_h[512], h < 4
    pk::c32 *V512 = reinterpret_cast<pk::c32 *>(smem + Q_BND + 96);        // V_h[512], h < 4
    {
        double2 *t1 = reinterpret_cast<double2 *>(smem + Q_TW1);
        double2 *t2 = reinterpret_cast<double2 *>(smem + Q_TW2);
        for (int i = threadIdx.x; i < 512; i += 256) {
            const int k = i >> 6, ln = i & 63;
            t1[i] = p.tw64[(16 * ln * k) & (N - 1)];                        // W_512^{ln k}
            if (i < 64) { const int k2 = i >> 3, n0 = i & 7; t2[i] = p.tw64[(128 * n0 * k2) & (N - 1)]; }   // W_64^{n0 k2}
            if (i < 16) reinterpret_cast<float2 *>(smem + Q_ROT)[i] = cconj(p.tw32[(i * (N / 16)) & (N - 1)]);
        }
    }
    __syncthreads();

    unsigned char *SA = smem + Q_A + 9216 * g;                              // this wave's scratch
    double2 *S64 = reinterpret_cast<double2 *>(SA);
    float2 *Y = reinterpret_cast<float2 *>(smem + Q_A);
    float *MAG = reinterpret_cast<float *>(smem + Q_B);
    unsigned *ROUTE = reinterpret_cast<unsigned *>(smem + Q_B);
    unsigned *CLAIM = reinterpret_cast<unsigned *>(smem + Q_B);
    v4u dq0{0u, 0u, 0u, 0u}, dq1{0u, 0u, 0u, 0u};
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

    const double2 wN = p.tw64[lane];                                       // W_8192^l
    constexpr float SC = 2.0f / ((float)N * (float)R);
    const int so = 16 * lane + 2 * g;                                       // this lane's first sample inside a register row of 1024

    v4f acc[8];
#pragma unroll
    for (int r = 0; r < 8; r++) acc[r] = v4f{0.f, 0.f, 0.f, 0.f};
    if (from_state) {
#pragma unroll
        for (int r = 0; r < LROWS; r++) {
            const float *a = p.acc_in + (long)ch * L + so + 1024 * r;
            acc[r] = v4f{a[0], a[1], a[8], a[9]};
        }
    }
    auto load_rows = [&](v4f *w, int frame, int so) {
        const long s0 = (long)(frame + 1) * HOP - N;                        // wave-uniform
        if (s0 >= 0 && vec_in) {
            const unsigned ob = 4u * (unsigned)so;
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const char *rb = reinterpret_cast<const char *>(src.in + s0 + 1024 * r);
                const v2f a = *reinterpret_cast<const v2f *>(rb + ob), b = *reinterpret_cast<const v2f *>(rb + ob + 32);
                w[r] = v4f{a.x, a.y, b.x, b.y};
            }
        } else {
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const long sx = s0 + so + 1024 * r;
                const float *q0 = sx < 0 ? src.hist + sx + src.hist_len : src.in + sx;
                const float *q1 = sx + 8 < 0 ? src.hist + sx + 8 + src.hist_len : src.in + sx + 8;
                if (vec_in) { const v2f a = *reinterpret_cast<const v2f *>(q0), b = *reinterpret_cast<const v2f *>(q1); w[r] = v4f{a.x, a.y, b.x, b.y}; }
                else w[r] = v4f{q0[0], q0[1], q1[0], q1[1]};
            }
        }
    };
    auto load_hann = [&](v4f *w, int so) {                                  // 0.5 * Hann: the second half of the table (pv_kernels.h)
        const unsigned ob = 4u * (unsigned)so;
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const char *rb = reinterpret_cast<const char *>(p.hann + N + 1024 * r);
            const v2f a = *reinterpret_cast<const v2f *>(rb + ob), b = *reinterpret_cast<const v2f *>(rb + ob + 32);
            w[r] = v4f{a.x, a.y, b.x, b.y};
        }
    };
    v4f raw[8], hw[8];
    load_rows(raw, first_frame, so);
    load_hann(hw, so);
    float pf_next = pitch_row[first_frame];
    int emit_v = first_out;
    asm volatile("" : "+v"(emit_v));

    for (int m = first_frame; m < last_out; ++m) {
        int l = lane;
        asm volatile("" : "+v"(l));
        const int LL = 64 * g + l;
        const float pfm = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(pf_next)));
        const double pf = (double)pfm;
        const int tmod = (int)(((long)p.t0_mod_n + (long)m * HOP) & (N - 1));
        const int pl = l + 4 * (l >> 4), ql = l + 4 * ((l + 15) >> 4);
        auto lane_twiddle = [&]() { double2 w = wN; asm volatile("" : "+v"(w.x), "+v"(w.y)); return w; };

        // ---- shift table of this wave's 1024 candidate bins ----
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
                    IMG[l + 64 * r] = ok ? (short)((int)ps - pk) : (short)0x4000;
                }
                wave_sync();
                typedef const volatile __attribute__((address_space(3))) v4u *lds_v4u_;
                dq0 = *(lds_v4u_)(SA + 32 * l);
                dq1 = *(lds_v4u_)(SA + 32 * l + 16);
                wave_sync();
            }
        }

        // ---- Hann, this wave's quarter of the packed sequence split by parity, two 512-point fp64 FFTs, decimation-in-time stage ----
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
            const double2 w1024 = csq8(csq8(csq8(lane_twiddle())));
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const double2 t = cmul(w1024, mul_w16<double, false>(zhi[r], r));
                const double2 e = zlo[r];
                zlo[r] = cadd(e, t);
                zhi[r] = csub(e, t);
            }
        }
        // ---- wave-local partner exchange, then the publish: slot (2 (r & 3) + {0: E_g[k], 1: E_g[1024 - k]}) of round r >> 2 ----
        double2 zmA[4], zmB[4];
        {
#pragma unroll
            for (int r = 0; r < 8; r++) S64[r * 64 + l] = zhi[r];
            wave_sync();
#pragma unroll
            for (int r = 0; r < 4; r++) { zmA[r] = S64[(7 - r) * 64 + 64 - l]; zmB[r] = S64[(3 - r) * 64 + 64 - l]; }
            if (l == 0) E512[g] = zhi[0];
            wave_sync();
#pragma unroll
            for (int r = 0; r < 4; r++) { S64[(2 * r) * 64 + l] = zlo[r]; S64[(2 * r + 1) * 64 + l] = zmA[r]; }
            __syncthreads();                                               // barrier 1: round 1 (rows 0..3) readable; every wave is done with the previous frame's inverse publish
            double2 *P2 = reinterpret_cast<double2 *>(smem + Q_B + 8192 * g);
#pragma unroll
            for (int r = 0; r < 4; r++) { P2[(2 * r) * 64 + l] = zlo[4 + r]; P2[(2 * r + 1) * 64 + l] = zmB[r]; }
        }
        __syncthreads();                                                   // barrier 2: round 2 (rows 4..7) readable
        // ---- finish this wave's two groups (register rows 2g, 2g + 1): 8 bins each ----
        float2 XK[2][4], XC[2][4];                                          // X[k + 1024 q], X[1024 - k + 1024 q], q < 4, rounded to fp32
        float2 xm[4] = {float2{0.f, 0.f}, float2{0.f, 0.f}, float2{0.f, 0.f}, float2{0.f, 0.f}};   // wave 3, lane 0: X[512 + 1024 q]
        {
            const double2 wNf = lane_twiddle();
            const double2 w1 = csq8(wNf), w2 = csq8(w1), w3 = cmul(w2, w1);  // W_4096^{h l}, h = 1, 2, 3
            // where the published rows of this wave's groups live: rows 0..3 in the scratches (region A), rows 4..7 in region B
            const unsigned char *pbase = (g < 2) ? smem + Q_A : smem + Q_B;
            const int pstride = (g < 2) ? 9216 : 8192;
            double2 ga[2][4], gb[2][4];
#pragma unroll
            for (int i = 0; i < 2; i++) {
                const int rr = 2 * (g & 1) + i;                             // row inside its round
#pragma unroll
                for (int h = 0; h < 4; h++) {
                    const double2 *ph = reinterpret_cast<const double2 *>(pbase + pstride * h);
                    ga[i][h] = ph[(2 * rr) * 64 + l];
                    gb[i][h] = ph[(2 * rr + 1) * 64 + l];
                }
            }
            __syncthreads();                                               // barrier 3: every published value has been read -- the magnitudes may overwrite round 2
#pragma unroll
            for (int i = 0; i < 2; i++) {
                const double2 a[4] = {ga[i][0], ga[i][1], ga[i][2], ga[i][3]};
                const double2 b[4] = {gb[i][0], gb[i][1], gb[i][2], gb[i][3]};
                {
                    const int r = 2 * g + i;                                // wave-uniform, run time: its twiddle parts are scalar loads
                    const double2 cw1 = qw64(r), cw2 = qw64(2 * r), cw3 = qw64(3 * r), cv = qw128(r);
                    const bool k0 = (r == 0) && (l == 0);
                    // T_h = W_4096^{h k} a_h,  conj(W_4096^{h k}) b_h = conj(W_4096^{h k} conj(b_h)),  U_h = (-j)^h of that
                    double2 T[4], U[4];
                    T[0] = a[0]; U[0] = b[0];
                    {
                        const double2 t1 = cmul(w1, cmul(a[1], cw1)), u1 = cmul(w1, cmul(cconj(b[1]), cw1));
                        const double2 t2 = cmul(w2, cmul(a[2], cw2)), u2 = cmul(w2, cmul(cconj(b[2]), cw2));
                        const double2 t3 = cmul(w3, cmul(a[3], cw3)), u3 = cmul(w3, cmul(cconj(b[3]), cw3));
                        T[1] = t1; T[2] = t2; T[3] = t3;
                        U[1] = double2{-u1.y, -u1.x};                       // -j conj(u1)
                        U[2] = double2{-u2.x, u2.y};                        // -conj(u2)
                        U[3] = double2{u3.y, u3.x};                         // +j conj(u3)
                    }
                    double2 Zk[4], Zc[4];
                    {
                        const double2 s02 = cadd(T[0], T[2]), d02 = csub(T[0], T[2]), s13 = cadd(T[1], T[3]), d13 = csub(T[1], T[3]);
                        Zk[0] = cadd(s02, s13); Zk[2] = csub(s02, s13);
                        Zk[1] = double2{d02.x + d13.y, d02.y - d13.x};      // d02 - j d13
                        Zk[3] = double2{d02.x - d13.y, d02.y + d13.x};      // d02 + j d13
                    }
                    {
                        const double2 s02 = cadd(U[0], U[2]), d02 = csub(U[0], U[2]), s13 = cadd(U[1], U[3]), d13 = csub(U[1], U[3]);
                        Zc[0] = cadd(s02, s13); Zc[2] = csub(s02, s13);
                        Zc[1] = double2{d02.x + d13.y, d02.y - d13.x};
                        Zc[3] = double2{d02.x - d13.y, d02.y + d13.x};
                    }
                    if (k0) { Zc[2] = Zk[3]; }                              // k = 0: bin 1024 pairs with bin 3072 = Z[3072] itself
#pragma unroll
                    for (int q = 0; q < 4; q++) {                           // pair (k + 1024 q, 1024 - k + 1024 (3 - q)), twiddle W_8192^{k + 1024 q} = wN W_128^r W_8^q
                        const double2 zk = Zk[q], zm = Zc[3 - q];
                        const double2 E{zk.x + zm.x, zk.y - zm.y}, O{zk.x - zm.x, zk.y + zm.y};
                        const double2 WO = cmul(wNf, cmul(mul_w8_f(O, q), cv));
                        double2 xa{E.x + WO.y, E.y - WO.x}, xb{E.x - WO.y, -(E.y + WO.x)};
                        if (k0 && q == 0) { xa = double2{2.0 * (zk.x + zk.y), 0.0}; xb = double2{2.0 * (zk.x - zk.y), 0.0}; }   // X[0], X[4096]
                        if (k0 && q == 2) { xa = double2{2.0 * zk.x, -2.0 * zk.y}; xb = xa; }                                     // X[2048] = 2 conj(Z[2048])
                        const bool keep_a = !(k0 && q == 3);                // k = 0: bin 3072 is produced by the pair q = 1
                        const bool keep_b = !(k0 && (q == 2 || q == 3));    //        bins 2048 (q = 2) and 1024 (q = 3) by XK
                        if (keep_a) MAG[MAG0Q + 1280 * q + pl + 80 * r] = (float)(xa.x * xa.x + xa.y * xa.y);
                        if (keep_b) MAG[MAG0Q + 1280 * (4 - q) - ql - 80 * r] = (float)(xb.x * xb.x + xb.y * xb.y);
                        XK[i][q] = float2{(float)xa.x, (float)xa.y};
                        XC[i][3 - q] = float2{(float)xb.x, (float)xb.y};
                    }
                }
            }
            if (g == 3 && l == 0) {                                         // k = 512: T_h = W_8^h E_h[512]; pairs (512, 3584), (1536, 2560), twiddles W_16^1, W_16^3
                const double2 T0 = E512[0], T1 = mul_w8_f(E512[1], 1), T2 = mul_w8_f(E512[2], 2), T3 = mul_w8_f(E512[3], 3);
                const double2 s02 = cadd(T0, T2), d02 = csub(T0, T2), s13 = cadd(T1, T3), d13 = csub(T1, T3);
                const double2 Z0 = cadd(s02, s13), Z2 = csub(s02, s13), Z1{d02.x + d13.y, d02.y - d13.x}, Z3{d02.x - d13.y, d02.y + d13.x};
                const double c1 = 0.92387953251128675613, s1 = 0.38268343236508977173;
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    const double2 zk = q ? Z1 : Z0, zm = q ? Z2 : Z3;
                    const double2 E{zk.x + zm.x, zk.y - zm.y}, O{zk.x - zm.x, zk.y + zm.y};
                    const double2 w = q ? double2{s1, -c1} : double2{c1, -s1};  // W_16^{1 + 2q}
                    const double2 WO = cmul(O, w);
                    const double2 xa{E.x + WO.y, E.y - WO.x}, xb{E.x - WO.y, -(E.y + WO.x)};
                    const int ba = 512 + 1024 * q, bb = 4096 - ba;
                    MAG[MAG0Q + ba + 4 * (ba >> 4)] = (float)(xa.x * xa.x + xa.y * xa.y);
                    MAG[MAG0Q + bb + 4 * (bb >> 4)] = (float)(xb.x * xb.x + xb.y * xb.y);
                    xm[q] = float2{(float)xa.x, (float)xa.y};                // X[512], X[1536]
                    xm[3 - q] = float2{(float)xb.x, (float)xb.y};            // X[3584], X[2560]
                }
            }
        }
        __syncthreads();                                                   // barrier 4: magnitudes complete
        const int r0 = 2 * g;                                               // first register row of this wave's groups
        const bool collide = !(pf >= 1.0);
        // bin and padded position of source slot (i, q): XK -> k + 1024 q, XC -> 1024 - k + 1024 q, k = l + 64 (r0 + i)
        auto for_each_source = [&](auto fn) {
#pragma unroll
            for (int i = 0; i < 2; i++) {
                const int r = r0 + i, k = l + 64 * r;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const bool okk = !(k == 0 && q == 3);
                    const bool okc = !(k == 0 && q < 2);
                    fn(4 * i + q, k + 1024 * q, 1280 * q + pl + 80 * r, XK[i][q], okk);
                    fn(8 + 4 * i + q, 1024 - k + 1024 * q, 1280 * (q + 1) - ql - 80 * r, XC[i][q], okc);
                }
            }
        };
        if (collide) {                                                      // f < 1: stash the fp32 spectrum for the fast form of the above-Nyquist residue
            float2 *XS = reinterpret_cast<float2 *>(smem + Q_A);
            for_each_source([&](int, int bin, int, float2 v, bool ok) { if (ok) XS[bin] = v; });
            if (g == 3 && l == 0) { XS[512] = xm[0]; XS[1536] = xm[1]; XS[2560] = xm[2]; XS[3584] = xm[3]; }
        }

        // ---- peak flags for bins 16 LL .. 16 LL + 15, nearest peaks inside the wave ----
        int last_shift = 0;
        unsigned rt[16];
        unsigned rtM = NOROUTE;
        int lastown[16], firstown[16], last_in, first_in, cprev, cnext;
        bool below_any, above_any;
        constexpr int NEGPD = -(16384 << 16), POSPD = 16384 << 16;          // "no peak on this side"
        {
            unsigned mg[20];
            typedef const volatile __attribute__((address_space(3))) v2u *lds_v2u;
            typedef const volatile __attribute__((address_space(3))) v4u *lds_v4u;
            const v2u q0 = *(lds_v2u)(&MAG[MAG0Q + 20 * LL - 6]);
            const v4u q1 = *(lds_v4u)(&MAG[MAG0Q + 20 * LL]);
            const v4u q2 = *(lds_v4u)(&MAG[MAG0Q + 20 * LL + 4]);
            const v4u q3 = *(lds_v4u)(&MAG[MAG0Q + 20 * LL + 8]);
            const v4u q4 = *(lds_v4u)(&MAG[MAG0Q + 20 * LL + 12]);
            const v2u q5 = *(lds_v2u)(&MAG[MAG0Q + 20 * LL + 20]);
            mg[0] = q0.x; mg[1] = q0.y;
            mg[2] = q1.x; mg[3] = q1.y; mg[4] = q1.z; mg[5] = q1.w; mg[6] = q2.x; mg[7] = q2.y; mg[8] = q2.z; mg[9] = q2.w;
            mg[10] = q3.x; mg[11] = q3.y; mg[12] = q3.z; mg[13] = q3.w; mg[14] = q4.x; mg[15] = q4.y; mg[16] = q4.z; mg[17] = q4.w;
            mg[18] = q5.x; mg[19] = q5.y;
            unsigned pm[19];
#pragma unroll
            for (int j = 3; j < 19; j++) pm[j] = max(mg[j], mg[j + 1]);
            int pd[16];
            int cur = NEGPD;
#pragma unroll
            for (int i = 0; i < 16; i++) {
                // candidates are 2 <= k < H - 2 = 4095 (pv:97-100): lane 0 drops i < 2, lane 255 drops i = 15
                const bool in_range = (i < 2) ? (LL != 0) : (i == 15) ? (LL != 255) : true;
                const bool fl = in_range & (max(max(mg[i], mg[i + 1]), pm[i + 3]) < mg[i + 2]);
                const unsigned w = (i < 8) ? dq0[i >> 1] : dq1[(i - 8) >> 1];
                pd[i] = (int)__builtin_amdgcn_perm((unsigned)(16 * LL + i), w, (i & 1) ? 0x05040302u : 0x05040100u);
                cur = fl ? pd[i] : cur;
                lastown[i] = cur;
                firstown[i] = fl ? 1 : 0;
            }
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
            const int wlast = occ ? __shfl(last_in, 63 - __clzll((long long)occ), 64) : NEGPD;
            const int wfirst = occ ? __shfl(first_in, __ffsll((long long)occ) - 1, 64) : POSPD;
            if (l == 0) { BND[2 * g] = wlast; BND[2 * g + 1] = wfirst; }
        }
        __syncthreads();                                                   // barrier 5: peak words across the wave boundaries; every magnitude read is done; the stash is complete
        float2 s2v[4] = {float2{0.f, 0.f}, float2{0.f, 0.f}, float2{0.f, 0.f}, float2{0.f, 0.f}};
        {
            // nearest peak in the waves below / above, the last peak of the frame
            int lo = NEGPD, hi = POSPD, lp = NEGPD;
#pragma unroll
            for (int h = 0; h < 4; h++) {
                const int wl = BND[2 * h], wf = BND[2 * h + 1];
                if (h < g && wl >= 0) lo = wl;                              // ascending h: the highest wave below that has a peak wins
                if (h > g && wf != POSPD && hi == POSPD) hi = wf;           // the lowest wave above that has one
                if (wl >= 0) lp = wl;
            }
            if (!below_any) cprev = lo;
            if (!above_any) cnext = hi;
            const bool any_peak = lp >= 0;
            if (any_peak) last_shift = (int)(short)(lp & 0xFFFF);
            if (!any_peak) {
#pragma unroll
                for (int i = 0; i < 16; i++) rt[i] = NOROUTE;
            } else {
                auto route_of = [&](int b, int pp, int pn) -> unsigned {
                    const int own = (b - (pp >> 16) < (pn >> 16) - b) ? pp : pn;
                    const int delta = __builtin_amdgcn_sbfe(own, 0, 16);
                    return __builtin_amdgcn_perm((unsigned)__mul24(delta, tmod), (unsigned)(b + delta), 0x05040100u);
                };
#pragma unroll
                for (int i = 0; i < 16; i++) rt[i] = route_of(16 * LL + i, max(lastown[i], cprev), min(firstown[i], cnext));
                if (LL == 255) rtM = route_of(M, max(last_in, cprev), POSPD);
            }
            if (collide) {                                                  // fast form of the residue: positions N/2 + kk, kk = 1 + LL + 256 j
                const float2 *XS = reinterpret_cast<const float2 *>(smem + Q_A);
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int kk = 1 + LL + 256 * j;                         // kk in [1, 1024]
                    const float2 x0 = XS[kk], x1 = XS[kk + 2048], x2 = XS[4096 - kk], x3 = XS[2048 - kk];
                    const float2 tsum{0.25f * ((x0.x - x1.x) + (x2.x - x3.x)), 0.25f * ((x0.y - x1.y) - (x2.y - x3.y))};
                    s2v[j] = cmul(tsum, cconj(p.tw32[2 * kk]));
                }
                __syncthreads();                                           // (f < 1 only) the stash is dead: Y may be zeroed
            }
        }
        // ---- routes (aliasing the magnitudes) and the zeroed Y ----
#pragma unroll
        for (int j = 0; j < 4; j++) *reinterpret_cast<uint4 *>(&ROUTE[20 * LL + 4 * j]) = uint4{rt[4 * j], rt[4 * j + 1], rt[4 * j + 2], rt[4 * j + 3]};
        if (LL == 255) ROUTE[5120] = rtM;
#pragma unroll
        for (int r = 0; r < 8; r++) *reinterpret_cast<v4f *>(&Y[1024 * g + 2 * l + 128 * r]) = v4f{0.f, 0.f, 0.f, 0.f};
        if (LL == 255) Y[M] = float2{0.f, 0.f};
        int upper_end = H;
        if (last_shift < 0) { upper_end = H - last_shift; if (upper_end > N) upper_end = N; }
        __syncthreads();                                                   // barrier 6
        // ---- shiftPeaks (pv:119-173) ----
        if (!collide) {
            auto scatter = [&](auto mode_tag) {
                constexpr int MODE = decltype(mode_tag)::value;
                auto put = [&](unsigned rtv, float2 v) {
                    const unsigned tg = rtv & 0xFFFFu;
                    if (tg < (unsigned)H) {
                        float2 o = v;
                        if (MODE == 2) {
                            const unsigned sg = (rtv << 3) & 0x80000000u;
                            o = float2{__uint_as_float(__float_as_uint(v.x) ^ sg), __uint_as_float(__float_as_uint(v.y) ^ sg)};
                        } else if (MODE == 1) o = rotate8k<R>(rtv, v, ROT);
                        Y[tg] = o;
                    }
                };
                for_each_source([&](int, int, int pos, float2 v, bool ok) { const unsigned rv = ROUTE[pos]; put(ok ? rv : NOROUTE, v); });
                if (g == 3 && l == 0) { put(ROUTE[640], xm[0]); put(ROUTE[1920], xm[1]); put(ROUTE[3200], xm[2]); put(ROUTE[4480], xm[3]); }
            };
            if (tmod == 0) scatter(std::integral_constant<int, 0>{});
            else if (tmod == N / 2) scatter(std::integral_constant<int, 2>{});
            else scatter(std::integral_constant<int, 1>{});
        } else {
            unsigned rs[20];
            float2 ys[20];
            int id[20];
            for_each_source([&](int slot, int bin, int pos, float2 v, bool ok) { id[slot] = bin; rs[slot] = ok ? ROUTE[pos] : NOROUTE; ys[slot] = v; });
            const bool mid = (g == 3 && l == 0);
#pragma unroll
            for (int q = 0; q < 4; q++) { id[16 + q] = 512 + 1024 * q; rs[16 + q] = mid ? ROUTE[640 + 1280 * q] : NOROUTE; ys[16 + q] = xm[q]; }
#pragma unroll
            for (int i = 0; i < 20; i++) ys[i] = rotate8k<R>(rs[i], ys[i], ROT);
            __syncthreads();                                               // every route read is done: the region becomes the claim words
#pragma unroll
            for (int j = 0; j < 16; j++) CLAIM[threadIdx.x + 256 * j] = 0xFFFFFFFFu;
            if (threadIdx.x == 0) CLAIM[M] = 0xFFFFFFFFu;
            claim_rounds_quad<20>(rs, ys, id, Y, CLAIM);
            if (upper_end > H) {
                const int up_delta = last_shift;
                const unsigned up_ridx = (unsigned)((up_delta & (N - 1)) * tmod) & (N - 1);
                if (upper_end <= H + N / 8) {
                    unsigned rt2[4];
                    float2 ys2[4];
                    int id2[4];
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const int b = M + 1 + LL + 256 * j, tgt = b + up_delta;
                        rt2[j] = (b < upper_end && tgt >= 0 && tgt < H) ? ((up_ridx << 16) | (unsigned)tgt) : NOROUTE;
                        ys2[j] = rotate8k<R>(rt2[j], s2v[j], ROT);
                        id2[j] = b - N / 2;
                    }
                    claim_rounds_quad<4>(rt2, ys2, id2, Y, CLAIM);
                } else {
                    residue_scatter_quad<R>(src.in, src.hist, src.hist_len, (long)(m + 1) * HOP - N, p.hann, p.tw32, (int)threadIdx.x, upper_end, up_delta, up_ridx);
                }
            }
        }
        __syncthreads();                                                   // barrier 7: Y complete
        // ---- c2r pre-pass and the radix-4 decimation-in-frequency stage for this wave's two groups, packed fp32 ----
        //      V_h[q'] = conj(W_4096^{h q'}) sum_q (+j)^{q h} Zc[q' + 1024 q]; published for every wave h: slot 4 h + 2 i + {0: q' = k, 1: q' = 1024 - k}
        {
            const pk::c32 scsc{SC, SC};
            const double2 wNf = lane_twiddle();
            const double2 w1 = csq8(wNf), w2 = csq8(w1), w3 = cmul(w2, w1);
            const pk::c32 cNs{(float)wNf.x * SC, -(float)wNf.y * SC};       // e^{+2 pi j l / 8192} SC
            const pk::c32 c1{(float)w1.x, -(float)w1.y}, c2{(float)w2.x, -(float)w2.y}, c3{(float)w3.x, -(float)w3.y};   // e^{+2 pi j h l / 4096}
            const pk::c32 *Yc = reinterpret_cast<const pk::c32 *>(Y);
            pk::c32 *PUB = reinterpret_cast<pk::c32 *>(smem + Q_B + 8192 * g);
#pragma unroll
            for (int i = 0; i < 2; i++) {
                {
                    const int r = 2 * g + i;
                    const pk::c32 cw1 = qw64i(r), cw2 = qw64i(2 * r), cw3 = qw64i(3 * r), cv = qw128i(r);
                    const int k = l + 64 * r;
                    const bool k0 = (r == 0) && (l == 0);
                    pk::c32 ZK[4], ZC[4];
#pragma unroll
                    for (int q = 0; q < 4; q++) {                           // pair (b = k + 1024 q, m = 1024 - k + 1024 (3 - q)), twiddle e^{+2 pi j b / 8192} SC
                        pk::c32 yb = Yc[k + 1024 * q], ym = Yc[1024 - k + 1024 * (3 - q)];
                        if (k0 && q == 0) { yb.y = 0.f; ym.y = 0.f; }
                        if (k0 && q == 1) ym = Yc[3072];                    // k = 0: bin 1024 pairs with bin 3072
                        const pk::c32 E = pk::add_conj(yb, ym), O = pk::sub_conj(yb, ym);
                        const pk::c32 c = pk::cmul(pk::cmul(mul_w8_i(O, q), cv), cNs);
                        ZK[q] = pk::fma_addj(E, scsc, c);
                        ZC[3 - q] = pk::fma_conj_subj(E, scsc, c);
                    }
                    if (k0) {                                               // k = 0: Zc[2048] = 2 conj(Y[2048]) SC (self-paired), Zc[3072] from the pair q = 1
                        const pk::c32 y2 = Yc[2048];
                        ZK[2] = pk::c32{2.0f * y2.x * SC, -2.0f * y2.y * SC};
                        ZK[3] = ZC[2];
                    }
                    pk::c32 Sk[4], Sc[4];
                    {
                        const pk::c32 s02 = pk::add(ZK[0], ZK[2]), d02 = pk::sub(ZK[0], ZK[2]), s13 = pk::add(ZK[1], ZK[3]), d13 = pk::sub(ZK[1], ZK[3]);
                        Sk[0] = pk::add(s02, s13); Sk[2] = pk::sub(s02, s13); Sk[1] = pk::add_j(d02, d13); Sk[3] = pk::sub_j(d02, d13);
                    }
                    {
                        const pk::c32 s02 = pk::add(ZC[0], ZC[2]), d02 = pk::sub(ZC[0], ZC[2]), s13 = pk::add(ZC[1], ZC[3]), d13 = pk::sub(ZC[1], ZC[3]);
                        Sc[0] = pk::add(s02, s13); Sc[2] = pk::sub(s02, s13); Sc[1] = pk::add_j(d02, d13); Sc[3] = pk::sub_j(d02, d13);
                    }
                    // V_h[k] = e^{+2 pi j h k / 4096} Sk[h];  V_h[1024 - k] = (+j)^h e^{-2 pi j h k / 4096} Sc[h] = (+j)^h conj(e^{+2 pi j h k/4096} conj(Sc[h]))
                    pk::c32 Vk[4], Vc[4];
                    Vk[0] = Sk[0]; Vc[0] = Sc[0];
                    Vk[1] = pk::cmul(pk::cmul(Sk[1], cw1), c1);
                    Vk[2] = pk::cmul(pk::cmul(Sk[2], cw2), c2);
                    Vk[3] = pk::cmul(pk::cmul(Sk[3], cw3), c3);
                    {
                        const pk::c32 u1 = pk::cmul(pk::cmul(pk::c32{Sc[1].x, -Sc[1].y}, cw1), c1);
                        const pk::c32 u2 = pk::cmul(pk::cmul(pk::c32{Sc[2].x, -Sc[2].y}, cw2), c2);
                        const pk::c32 u3 = pk::cmul(pk::cmul(pk::c32{Sc[3].x, -Sc[3].y}, cw3), c3);
                        Vc[1] = pk::c32{u1.y, u1.x};                        // +j conj(u1)
                        Vc[2] = pk::c32{-u2.x, u2.y};                       // -conj(u2)
                        Vc[3] = pk::c32{-u3.y, -u3.x};                      // -j conj(u3)
                    }
#pragma unroll
                    for (int h = 0; h < 4; h++) { PUB[(4 * h + 2 * i) * 64 + l] = Vk[h]; PUB[(4 * h + 2 * i + 1) * 64 + l] = Vc[h]; }
                }
            }
            if (g == 3 && l == 0) {                                         // k = 512: pairs (512, 3584) and (1536, 2560); V_h[512] = conj(W_8^h) S_h
                pk::c32 ZK[4];
                const float c1f = 0.92387953251128675613f, s1f = 0.38268343236508977173f;
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    const pk::c32 yb = Yc[512 + 1024 * q], ym = Yc[3584 - 1024 * q];
                    const pk::c32 E = pk::add_conj(yb, ym), O = pk::sub_conj(yb, ym);
                    const pk::c32 tw = q ? pk::c32{s1f * SC, c1f * SC} : pk::c32{c1f * SC, s1f * SC};   // e^{+2 pi j (512 + 1024 q) / 8192} SC
                    const pk::c32 c = pk::cmul(O, tw);
                    ZK[q] = pk::fma_addj(E, scsc, c);
                    ZK[3 - q] = pk::fma_conj_subj(E, scsc, c);
                }
                const pk::c32 s02 = pk::add(ZK[0], ZK[2]), d02 = pk::sub(ZK[0], ZK[2]), s13 = pk::add(ZK[1], ZK[3]), d13 = pk::sub(ZK[1], ZK[3]);
                V512[0] = pk::add(s02, s13);
                V512[1] = mul_w8_i(pk::add_j(d02, d13), 1);
                V512[2] = mul_w8_i(pk::sub(s02, s13), 2);
                V512[3] = mul_w8_i(pk::sub_j(d02, d13), 3);
            }
        }
        __syncthreads();                                                   // barrier 8: inverse exchange
        pk::c32 zA[8], zB[8];
        {
            // zA[r] = V_g[l + 64 r] from the owner of row r (wave r >> 1, group r & 1); zB[7 - r] = V_g[1024 - k'] of the pair (lane 64 - l, row r) --
            // lane 0 pairs with itself one register higher -- read straight from the publish, no hand-over
            const pk::c32 *PB = reinterpret_cast<const pk::c32 *>(smem + Q_B);
#pragma unroll
            for (int r = 0; r < 8; r++) {
                zA[r] = PB[1024 * (r >> 1) + (4 * g + 2 * (r & 1)) * 64 + l];
                const int rn = (r + 1) & 7;
                const int a_self = 1024 * (rn >> 1) + (4 * g + 2 * (rn & 1) + 1) * 64;            // lane 0: row r + 1, lane 0
                const int a_pair = 1024 * (r >> 1) + (4 * g + 2 * (r & 1) + 1) * 64 + 64 - l;      // others: row r, lane 64 - l
                zB[7 - r] = PB[(l == 0) ? a_self : a_pair];
            }
            if (l == 0) zB[0] = V512[g];
        }
        // ---- this wave's 1024-point inverse ----
        {
            const double2 w1024 = csq8(csq8(csq8(lane_twiddle())));
            const pk::c32 c1024{(float)w1024.x, -(float)w1024.y};
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const pk::c32 a = pk::add(zA[r], zB[r]), d = pk::sub(zA[r], zB[r]);
                zA[r] = a;
                zB[r] = pk::cmul(mul_w16_qi(d, r), c1024);
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
        fft512_wave_inv_pk64q(zA, reinterpret_cast<pk::c32 *>(SA), TW1, TW2, l);
        fft512_wave_inv_pk64q(zB, reinterpret_cast<pk::c32 *>(SA), TW1, TW2, l);
        // ---- Hann, overlap-add, emit, shift ----
        {
            const bool emit_out = (m >= emit_v);
            v4f fr[8];
#pragma unroll
            for (int r = 0; r < 8; r++) fr[r] = v4f{zA[r].x, zA[r].y, zB[r].x, zB[r].y} * hw[r];
#pragma unroll
            for (int r = 0; r < S_ROWS; r++) {
                const v4f o = acc[r] + fr[r];
                if (emit_out) {
                    char *rb = reinterpret_cast<char *>(outp + (long)m * HOP + 1024 * r);
                    const unsigned ob = 4u * (unsigned)(16 * l + 2 * g);
                    if (vec_out) {
                        *reinterpret_cast<v2f *>(rb + ob) = v2f{o.x, o.y};
                        *reinterpret_cast<v2f *>(rb + ob + 32) = v2f{o.z, o.w};
                    } else { float *dst = reinterpret_cast<float *>(rb + ob); dst[0] = o.x; dst[1] = o.y; dst[8] = o.z; dst[9] = o.w; }
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
#pragma unroll
        for (int r = 0; r < LROWS; r++) {
            float *a = p.acc_out + (long)ch * L + so + 1024 * r;
            a[0] = acc[r].x; a[1] = acc[r].y; a[8] = acc[r].z; a[9] = acc[r].w;
            float *hs = p.hist_out + (long)ch * L + so + 1024 * r;
            const long s = (long)p.nhops * HOP - L + so + 1024 * r;
            hs[0] = src.at(s); hs[1] = src.at(s + 1); hs[8] = src.at(s + 8); hs[9] = src.at(s + 9);
        }
    }
}

template <int S_ROWS>
hipError_t launch_quad(const PvKernelParams &p, int nch, int nchunks, hipStream_t st)
{
    static bool attr_done[16] = {};
    auto k = pv_quad_kernel<S_ROWS>;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!attr_done[dev & 15]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)pv_quad_lds_bytes());
        if (e != hipSuccess) return e;
        attr_done[dev & 15] = true;
    }
    PvKernelParams q = p;
    q.nchunks = nchunks;
    q.nch = nch;
    const long chains = (long)nch * nchunks;
    hipLaunchKernelGGL(k, dim3((unsigned)chains, 1, 1), dim3(256, 1, 1), pv_quad_lds_bytes(), st, q);
    return hipGetLastError();
}

}  // namespace

size_t pv_quad_lds_bytes() { return Q_BYTES; }
int pv_quad_threads() { return 256; }
bool pv_quad_supported(int log2n, int hop) { return log2n == 13 && (hop == 1024 || hop == 2048 || hop == 4096 || hop == 8192); }

hipError_t pv_launch_quad(const PvKernelParams &p, int nch, int nchunks, hipStream_t st)
{
    switch (p.hop) {
    case 1024: return launch_quad<1>(p, nch, nchunks, st);
    case 2048: return launch_quad<2>(p, nch, nchunks, st);
    case 4096: return launch_quad<4>(p, nch, nchunks, st);
    case 8192: return launch_quad<8>(p, nch, nchunks, st);
    default: return hipErrorInvalidValue;
    }
}
