// pv_wave_fft.h -- the 512-point complex FFTs of ONE wavefront (lane l, register r <-> element l + 64 r), shared by the wave kernels
// (pv_wave_kernel.hip: N = 1024; pv_wave2k_kernel.hip: N = 2048 = two of these + one in-register radix-2 stage).
#pragma once
#include <hip/hip_runtime.h>
#include "pv_device_common.h"
#include "pv_pk_math.h"

namespace {

#ifndef PV_TP
#define PV_TP 72
#endif
constexpr int TP = PV_TP;   // padded row of the transpose scratch (elements); 72 is conflict-free with the skew below

// Measurement hooks (tools/exp_headline.sh; never defined in the product build): PV_ABL is a bit mask of pipeline parts left out of a
// timing-only build (1 forward FFT arithmetic, 2 forward transposes, 4 inverse arithmetic, 8 inverse transposes, 16 peak search), and
// the FFTs call st(id) at their phase boundaries (PV_STAMPS builds accumulate s_memtime deltas there; the default functor is empty).
#ifndef PV_ABL
#define PV_ABL 0
#endif
struct NoStamp { __device__ __forceinline__ void operator()(int) const {} };

// ---- wave priority per pipeline phase (round 3; profiles/r03_priority_sweep.md) ----
// The waves of a SIMD are arbitrated by priority, then age.  With every wave at priority 0 the three frame chains of a SIMD interleave
// instruction by instruction, drift into the same phase and then queue at the LDS together; the phases that are chains of short dependent LDS
// round trips (split exchange, peak search, scatter, c2r hand-over: half of a frame's wall time for a quarter of its instructions) pay that
// queue on every hop.  Raising those phases above the bulk phases (FFT arithmetic, and lowest of all the FFT transposes, which are long
// anyway) lets a wave run through its latency chain while the other two fill the VALU: measured -13 % on the headline launch (2.35 -> 2.03 ms),
// -12 % on N = 2048, -10 % on N = 4096, results bit-identical.  A static priority per wave (by rank on the SIMD) does nothing.
// PV_PT is the table: one s_setprio level (0..3) per phase, in the order of the enum below; 9 = leave the priority unchanged (all 9 = the
// round-2 behaviour).  A kernel file may define its own table before including this header.
#ifndef PV_PT
#define PV_PT 1, 0, 1, 2, 2, 2, 3, 3, 0, 0, 0, 2
#endif
enum { PH_FA = 0,      // forward FFT: window, pack, passes 1 and 2 (arithmetic)
       PH_FX,          // forward FFT: LDS transposes
       PH_FP3,         // forward FFT: pass 3
       PH_SPLITX,      // split pass: partner exchange
       PH_SPLITM,      // split pass: arithmetic, |X|^2, prefetch of the next frame's rows
       PH_PEAKS, PH_SCATTER, PH_C2R,
       PH_IA,          // inverse FFT: passes 1 and 2
       PH_IX,          // inverse FFT: LDS transposes
       PH_IP3,         // inverse FFT: pass 3
       PH_OLA,         // window, overlap-add, stores
       PH_COUNT };
template <int PHASE> __device__ __forceinline__ void pv_prio_t()
{
    constexpr int t[] = {PV_PT};
    static_assert(sizeof(t) / sizeof(t[0]) == PH_COUNT, "PV_PT needs one level per phase");
    if constexpr (t[PHASE] <= 3) __builtin_amdgcn_s_setprio(t[PHASE]);
}
#define pv_prio(PHASE) pv_prio_t<PHASE>()

// ---- transpose 1 of the wave FFTs in registers (round 3) ----
// [reg k0][lane (n1, n0)] -> [reg n1][lane (k0, n0)] exchanges the register index with the HIGH three lane bits, one bit per stage: bit 5 with
// v_permlane32_swap (upper half of A <-> lower half of B), bit 4 with v_permlane16_swap (odd rows of A <-> even rows of B), both new in CDNA4,
// bit 3 with two bank-masked DPP moves (row_ror:8 = lane ^ 8 inside a row of 16).  NDW dwords per element (4: double2, 2: packed fp32 complex).
// It takes 8 KB (fp64) / 4 KB (fp32) per frame off the LDS store path, the busiest pipe of the CU, for 80 / 40 VALU instructions: -1.3 % each
// alone, -2.8 % together on the headline launch, -2.5 % on top of the priorities (profiles/r03_priority_sweep.md).  Transpose 2 moves the LOW lane
// bits, where only DPP quad permutes reach: 4 instructions per dword pair, slower than the LDS round trip it would replace -- it stays in LDS.
// PV_PERM_T1: bit 0 forward, bit 1 inverse (0 = the round-2 LDS transposes, for A/B builds).
#ifndef PV_PERM_T1
#define PV_PERM_T1 3
#endif
// 512-point complex FFT across one wave: in/out layout lane l, reg r <-> element l + 64 r.
// TW1[k*64 + l] = W_512^{l k}, TW2[k*8 + n0] = W_64^{n0 k} (k = 1..7) live in LDS, shared by the waves of the workgroup,
// already conjugated / rounded for the inverse fp32 instance.
template <typename T, bool INV, typename ST = NoStamp>
__device__ __forceinline__ void fft512_wave(typename v2t<T>::type (&a)[8], typename v2t<T>::type *S, const typename v2t<T>::type *TW1,
                                            const typename v2t<T>::type *TW2, int l, ST st = ST{})
{
    constexpr bool MATH = !(PV_ABL & 1), XPOSE = !(PV_ABL & 2);
    const int lh = l >> 3, ll = l & 7;
    // pass 1: DFT over n2 (register index); twiddle W_512^{l*k0}
    if (MATH) {
        radix8<T, INV>(a);
#pragma unroll
        for (int k = 1; k < 8; k++) a[k] = cmul(a[k], TW1[k * 64 + l]);
    }
    st(0);
    // transpose 1: [reg k0][lane (n1,n0)] -> [reg n1][lane (k0,n0)]
    if (XPOSE && (PV_PERM_T1 & 1) && sizeof(T) == 8) {
        unsigned w[8][4];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint2 x = __builtin_bit_cast(uint2, (double)a[k].x), y = __builtin_bit_cast(uint2, (double)a[k].y);
            w[k][0] = x.x; w[k][1] = x.y; w[k][2] = y.x; w[k][3] = y.y;
        }
        transpose_hi3_regs<4>(w);
#pragma unroll
        for (int k = 0; k < 8; k++) {
            a[k].x = (T)__builtin_bit_cast(double, uint2{w[k][0], w[k][1]});
            a[k].y = (T)__builtin_bit_cast(double, uint2{w[k][2], w[k][3]});
        }
    } else if (XPOSE) {
        pv_prio(PH_FX);
#pragma unroll
        for (int k = 0; k < 8; k++) S[k * TP + l] = a[k];
        wave_sync();
#pragma unroll
        for (int n = 0; n < 8; n++) a[n] = S[lh * TP + 8 * n + ll];
        wave_sync();
        pv_prio(PH_FA);
    }
    st(1);
    // pass 2: DFT over n1; twiddle W_64^{n0*k1}
    if (MATH) {
        radix8<T, INV>(a);
#pragma unroll
        for (int k = 1; k < 8; k++) a[k] = cmul(a[k], TW2[k * 8 + ll]);
    }
    st(2);
    // transpose 2: [reg k1][lane (k0,n0)] -> [reg n0][lane (k1,k0)], skewed rows
    if (XPOSE) {
        pv_prio(PH_FX);
#pragma unroll
        for (int k = 0; k < 8; k++) S[k * TP + lh * 8 + ((ll + lh) & 7)] = a[k];
        wave_sync();
#pragma unroll
        for (int n = 0; n < 8; n++) a[n] = S[lh * TP + ll * 8 + ((n + ll) & 7)];
        wave_sync();
        pv_prio(PH_FP3);
    }
    st(3);
    // pass 3: DFT over n0 -> k2; lane l now holds X[l + 64 k2]
    if (MATH) radix8<T, INV>(a);
}


// The inverse instance in packed fp32 (pv_pk_math.h): 106 packed instructions instead of ~210.  Its LDS traffic is organised around the
// measured costs of the LDS pipe (tools/valu_microbench2.hip: a read costs ~3 cycles per wave-instruction whatever its width, ds_write_b64
// ~5.8, ds_write_b128 ~9.1): the twiddles of two consecutive k come from ONE ds_read_b128 of a pair-interleaved table, and the transposes
// write register PAIRS (4 ds_write_b128 instead of 8 ds_write_b64); a reader picks the half it needs with the address.  Rows of 64 pair slots
// are padded to TPP = 72 slots (1152 B = 128 mod 256): the four rows a read touches fall on alternating halves of the 64 banks -> 2 passes,
// the minimum for 512 bytes (checked with tools/lds_layout_check.py pairs).
constexpr int TPP = 72;
// REGT1 = false keeps transpose 1 in LDS: needed when `l` is NOT the physical lane id (pv_wave2k_kernel relabels the lanes of odd frames at hop 128).
// FWDPH = true: the instance that serves as the fp32 FORWARD transform (round 5: FFT(z) = conj(IFFT(conj z)), the conjugations folded into the window product
// and the split pass) -- same arithmetic, the forward phases' priorities.
template <bool REGT1 = true, typename ST = NoStamp, bool FWDPH = false>
__device__ __forceinline__ void fft512_wave_inv_pk(pk::c32 (&a)[8], pk::c32 *S, const v4f *TW1F4, const v4f *TW2F4, int l, ST st = ST{})
{
    constexpr int PHX = FWDPH ? PH_FX : PH_IX, PHA = FWDPH ? PH_FA : PH_IA, PHP3 = FWDPH ? PH_FP3 : PH_IP3;
    constexpr bool MATH = !(PV_ABL & 4), XPOSE = !(PV_ABL & 8);
    const int lh = l >> 3, ll = l & 7;
    v4f *S4 = reinterpret_cast<v4f *>(S);
    if (MATH) {
        pk::radix8_inv(a);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const v4f t = TW1F4[j * 64 + l];
            if (j) a[2 * j] = pk::cmul(a[2 * j], pk::c32{t.x, t.y});
            a[2 * j + 1] = pk::cmul(a[2 * j + 1], pk::c32{t.z, t.w});
        }
    }
    st(0);
    // transpose 1: [reg k0][lane (n1,n0)] -> [reg n1][lane (k0,n0)]; pair row k0 >> 1, half k0 & 1
    if (XPOSE && REGT1 && (PV_PERM_T1 & 2)) {
        unsigned w[8][2];
#pragma unroll
        for (int k = 0; k < 8; k++) { w[k][0] = __float_as_uint(a[k].x); w[k][1] = __float_as_uint(a[k].y); }
        transpose_hi3_regs<2>(w);
#pragma unroll
        for (int k = 0; k < 8; k++) a[k] = pk::c32{__uint_as_float(w[k][0]), __uint_as_float(w[k][1])};
    } else if (XPOSE) {
        pv_prio(PHX);
#pragma unroll
        for (int j = 0; j < 4; j++) S4[j * TPP + l] = v4f{a[2 * j].x, a[2 * j].y, a[2 * j + 1].x, a[2 * j + 1].y};
        wave_sync();
#pragma unroll
        for (int n = 0; n < 8; n++) a[n] = S[2 * ((lh >> 1) * TPP + 8 * n + ll) + (lh & 1)];
        wave_sync();
        pv_prio(PHA);
    }
    st(1);
    if (MATH) {
        pk::radix8_inv(a);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const v4f t = TW2F4[j * 8 + ll];
            if (j) a[2 * j] = pk::cmul(a[2 * j], pk::c32{t.x, t.y});
            a[2 * j + 1] = pk::cmul(a[2 * j + 1], pk::c32{t.z, t.w});
        }
    }
    st(2);
    // transpose 2: [reg k1][lane (k0,n0)] -> [reg n0][lane (k1,k0)], skewed columns
    if (XPOSE) {
        pv_prio(PHX);
#pragma unroll
        for (int j = 0; j < 4; j++) S4[j * TPP + lh * 8 + ((ll + lh) & 7)] = v4f{a[2 * j].x, a[2 * j].y, a[2 * j + 1].x, a[2 * j + 1].y};
        wave_sync();
#pragma unroll
        for (int n = 0; n < 8; n++) a[n] = S[2 * ((lh >> 1) * TPP + ll * 8 + ((n + ll) & 7)) + (lh & 1)];
        wave_sync();
        pv_prio(PHP3);
    }
    st(3);
    if (MATH) pk::radix8_inv(a);
}

}  // namespace
