// pv_device_common.h -- device helpers shared by the wave kernel (N = 1024) and the workgroup kernel (N = 2048..8192).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "pv_signal.h"
#include "pv_mul_rounded.h"

// Measured on gfx950 (tools/lds_microbench.hip): ds_read2_b64 / ds_read2st64_b64 / ds_read2_b32 occupy the LDS pipe for 8 cycles, twice
// the cost of the two single reads they replace (2 + 2); ds_write2_b64 is neutral.  The SI load/store optimizer forms them wherever
// two reads share a base register (Hann tables, the fp32 transposes), so the kernels opt out of it per function.  The attribute only
// means something to the device pass.
#if defined(__HIP_DEVICE_COMPILE__)
#define PV_NO_DS_MERGE __attribute__((target("no-load-store-opt")))
#else
#define PV_NO_DS_MERGE
#endif

namespace {

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));   // clang vector type (the nontemporal builtins need one)

template <typename T> struct v2t;
template <> struct v2t<float> { using type = float2; };
template <> struct v2t<double> { using type = double2; };

template <typename T2> __device__ __forceinline__ T2 cadd(T2 a, T2 b) { return T2{a.x + b.x, a.y + b.y}; }
template <typename T2> __device__ __forceinline__ T2 csub(T2 a, T2 b) { return T2{a.x - b.x, a.y - b.y}; }
template <typename T2> __device__ __forceinline__ T2 cmul(T2 a, T2 b) { return T2{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
template <typename T2> __device__ __forceinline__ T2 cconj(T2 a) { return T2{a.x, -a.y}; }
// multiply by -j (forward) or +j (inverse)
template <bool INV, typename T2> __device__ __forceinline__ T2 rot90(T2 a) { return INV ? T2{-a.y, a.x} : T2{a.y, -a.x}; }

// In-register 8-point DFT, natural order in and out.  INV selects exp(+2 pi j nk/8).
template <typename T, bool INV>
__device__ __forceinline__ void radix8(typename v2t<T>::type (&a)[8])
{
    using T2 = typename v2t<T>::type;
    const T h = (T)0.70710678118654752440;
    const T2 b0 = cadd(a[0], a[4]), b4 = csub(a[0], a[4]);
    const T2 b1 = cadd(a[1], a[5]), b5 = csub(a[1], a[5]);
    const T2 b2 = cadd(a[2], a[6]), b6 = csub(a[2], a[6]);
    const T2 b3 = cadd(a[3], a[7]), b7 = csub(a[3], a[7]);
    // odd half: c_n = b_{n+4} * W8^n
    T2 c1, c3;
    if (!INV) { c1 = T2{(b5.x + b5.y) * h, (b5.y - b5.x) * h}; c3 = T2{(b7.y - b7.x) * h, -(b7.x + b7.y) * h}; }
    else      { c1 = T2{(b5.x - b5.y) * h, (b5.x + b5.y) * h}; c3 = T2{-(b7.x + b7.y) * h, (b7.x - b7.y) * h}; }
    const T2 c0 = b4, c2 = rot90<INV>(b6);
    // even outputs: 4-pt DFT of b0..b3
    {
        const T2 e0 = cadd(b0, b2), e1 = csub(b0, b2), e2 = cadd(b1, b3), e3 = rot90<INV>(csub(b1, b3));
        a[0] = cadd(e0, e2); a[4] = csub(e0, e2); a[2] = cadd(e1, e3); a[6] = csub(e1, e3);
    }
    // odd outputs: 4-pt DFT of c0..c3
    {
        const T2 e0 = cadd(c0, c2), e1 = csub(c0, c2), e2 = cadd(c1, c3), e3 = rot90<INV>(csub(c1, c3));
        a[1] = cadd(e0, e2); a[5] = csub(e0, e2); a[3] = cadd(e1, e3); a[7] = csub(e1, e3);
    }
}

// o * exp(-+2 pi j r / 16): the wave-uniform part of the split-pass twiddle (INV = conjugate)
template <typename T, bool INV>
__device__ __forceinline__ typename v2t<T>::type mul_w16(typename v2t<T>::type o, int r)
{
    using T2 = typename v2t<T>::type;
    const T c = (T)0.92387953251128675613, s = (T)0.38268343236508977173, h = (T)0.70710678118654752440;
    T2 w;
    switch (r) {               // r is a compile-time constant at every call site (unrolled loops)
    case 0: return o;
    case 4: return rot90<INV>(o);
    case 1: w = T2{c, -s}; break;
    case 2: w = T2{h, -h}; break;
    case 3: w = T2{s, -c}; break;
    case 5: w = T2{-s, -c}; break;
    case 6: w = T2{-h, -h}; break;
    default: w = T2{-c, -s}; break;
    }
    if (INV) w.y = -w.y;
    return cmul(o, w);
}

// mul_rounded (a * b rounded to fp32 as an operation of its own, pv:55,67): pv_mul_rounded.h

struct WaveSrc {
    const float *in;
    const float *hist;
    int hist_len;
    bool sys = false;      // resident streaming kernel with its input in DEVICE memory (cached in L2) that the host rewrites through the BAR between quanta WITHIN one launch -> system-scope loads
    __device__ __forceinline__ float at(long s) const
    {
        if (s < 0) return hist[s + hist_len];
        return sys ? __hip_atomic_load(in + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : in[s];
    }
};

constexpr unsigned NOROUTE = 0xFFFFFFFFu;        // route = (rotation index << 16) | target bin

// The pairwise test of the f < 1 scatter (pv_wave_kernel.hip: ov = delta_i - delta_{i+1} <= floor(gap / 2) for every pair of neighbouring peaks) cannot
// fail when f >= 2/3: Math.round(x) lies in (x - 1/2, x + 1/2], so ov = gap - (round(p_{i+1} f) - round(p_i f)) < gap (1 - f) + 1 <= gap / 3 + 1, i.e.
// ov <= ceil(gap / 3), and ceil(gap / 3) <= floor(gap / 2) for every gap >= 3 -- which is the least distance of two local maxima over +-2 bins
// (pv:101-110); no peak is dropped in that range either (pv:127-129).  The kernels skip the test for such f (tests/test_pairwise_rule.py proves the
// bound by exhaustion over peak positions and gaps for N <= 8192).
constexpr float PV_PAIRWISE_SURE = 0.6667f;

// Rotation exp(+2 pi j ridx / N) of one source value (pv:155-170).  R = 4: (delta * t) mod N is a multiple of N/4, so the rotation is
// j^qd exactly: a swap and two sign-bit XORs (j^1 = (-y, x), j^2 = (-x, -y), j^3 = (y, -x)).
template <int R_, int LOG2N_>
__device__ __forceinline__ float2 rotate_route(unsigned route, float2 v, const float2 *__restrict__ tw32)
{
    const unsigned ridx = route >> 16;
    if (R_ == 4) {
        const unsigned qd = (ridx >> (LOG2N_ - 2)) & 3u;                  // bits above the rotation index are don't-care
        const bool sw = (qd & 1u) != 0u;
        const float a = sw ? v.y : v.x, b = sw ? v.x : v.y;
        return float2{__uint_as_float(__float_as_uint(a) ^ ((((qd + 1u) >> 1) & 1u) << 31)), __uint_as_float(__float_as_uint(b) ^ ((qd >> 1) << 31))};
    }
    // v * conj(w), the order of the roundings spelled out: the instances of a kernel (streaming / batch, f >= 1 only / every f) must agree bit for bit,
    // and a contraction left to the compiler comes out as fma(v.x, w.x, v.y * w.y) in one and fma(v.y, w.y, v.x * w.x) in another
    const float2 w = tw32[ridx & ((1u << LOG2N_) - 1u)];                  // masked: NOROUTE carries ridx = 0xFFFF
    return float2{__fmaf_rn(v.x, w.x, __fmul_rn(v.y, w.y)), __fmaf_rn(v.y, w.x, -__fmul_rn(v.x, w.y))};
}


// Register <-> lane transpose WITHOUT LDS: exchanges the register index (8 registers) with the HIGH three lane bits, one bit per stage (see
// pv_wave_fft.h, "transpose 1 of the wave FFTs in registers"); used by the wave FFTs and by transpose 2 of the 8192-point workgroup FFT.
template <int NDW>
__device__ __forceinline__ void transpose_hi3_regs(unsigned (&w)[8][NDW])
{
#pragma unroll
    for (int k = 0; k < 4; k++)                       // reg bit 2 <-> lane bit 5
#pragma unroll
        for (int d = 0; d < NDW; d++) {
            const auto r = __builtin_amdgcn_permlane32_swap(w[k][d], w[k + 4][d], false, false);
            w[k][d] = r[0]; w[k + 4][d] = r[1];
        }
#pragma unroll
    for (int q = 0; q < 4; q++) {                     // reg bit 1 <-> lane bit 4
        const int k = (q & 1) | ((q & 2) << 1);       // 0, 1, 4, 5
#pragma unroll
        for (int d = 0; d < NDW; d++) {
            const auto r = __builtin_amdgcn_permlane16_swap(w[k][d], w[k + 2][d], false, false);
            w[k][d] = r[0]; w[k + 2][d] = r[1];
        }
    }
#pragma unroll
    for (int k = 0; k < 8; k += 2)                    // reg bit 0 <-> lane bit 3
#pragma unroll
        for (int d = 0; d < NDW; d++) {
            const unsigned a = w[k][d], b = w[k + 1][d];
            w[k][d] = __builtin_amdgcn_update_dpp(a, b, 0x128, 0xF, 0xC, false);       // lanes 8..15 of every row take B[l ^ 8]
            w[k + 1][d] = __builtin_amdgcn_update_dpp(b, a, 0x128, 0xF, 0x3, false);   // lanes 0..7 take A[l ^ 8]
        }
}

// Wave-local ordering of LDS traffic: LDS instructions of one wave execute in order, so only the compiler must be fenced.
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

}  // namespace
