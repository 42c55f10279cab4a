// pv_guard.h -- the guard band of the fp32-first forward transform (round 5), shared by the one-wave kernels (pv_wave_kernel.hip: N = 1024, pv_wave2k_kernel.hip: N = 2048).
#pragma once
#include <hip/hip_runtime.h>
#include "pv_pk_math.h"

namespace {

// ---- fp32-first forward transform: the guard band (round 5) ----
// An F32 instance takes the peak decisions (pv:95-116) on the |X|^2 of a PACKED-fp32 forward transform wherever that is provably the same decision the
// reference's fp64 transform gives.  Error law of the fp32 path, MEASURED on the GPU against the fp64 path of the same kernel (tools/flip_count.py, validation build
// -DPV_FLIP_COUNT: 4.7e7 frames of eleven signal classes, 2.4e10 bins): in the amplitude A = |X| of a bin
//     |A32 - A64|  <=  3.3 eps max|X|  +  (a few eps) A            (largest value seen; 1.0 ... 3.3 across the classes)
// i.e. the absolute part scales with the LARGEST bin of the frame, not with the frame's rms: a weak bin is a cancellation of terms the size of the strong partial's,
// and what the early passes round off at that size reaches every bin (the rms model of tools/study_fp32_decisions.py was refuted by that build: bins outside its band flipped).
// A comparison c > n of two magnitudes is therefore safe while |A_c - A_n| > G + rho eps (A_c + A_n) / 2, G = g eps max|X| -- evaluated without square roots:
//     ambiguous  <=>  (c - n)^2 <= (c + n) (K + R (c + n)),     K = 2 G^2 = 2 g^2 eps^2 max|X|^2,   R = rho^2 eps^2
// which is the exact condition when A_c = A_n (the only place it matters) and errs on the ambiguous side elsewhere.  Only the comparison of a bin with the LARGEST
// of its four neighbours decides whether it is a peak, so one test per bin.  A frame with one ambiguous bin re-runs its forward transform in fp64.
// g = 10: three times the largest single-bin error seen, and three times the largest discrepancy that flipped a decision in that build (2.4 ... 3.2 eps max|X|
// at g = 8, 12, 16: q_max of profiles/r05_flip_count.json); rho = 32.
#ifndef PV_GUARD_G
#define PV_GUARD_G 10.0f
#endif
#ifndef PV_GUARD_RHO
#define PV_GUARD_RHO 32.0f
#endif
constexpr float GUARD_EPS = 5.9604644775390625e-8f;                       // 2^-24
constexpr float GUARD_CK = 2.0f * PV_GUARD_G * PV_GUARD_G * GUARD_EPS * GUARD_EPS;
constexpr float GUARD_R = PV_GUARD_RHO * PV_GUARD_RHO * GUARD_EPS * GUARD_EPS;
// The guarded range of the frame's largest magnitude M = max|X|^2.  The test works on SQUARED magnitudes -- fourth powers of amplitudes --, so a quiet frame's
// products reach the denormal range: with d = c - n, s = c + n the last operation, fma(d, d, -(s (K + R s))), has the sign of the exact difference of its operands,
// and for every bin with A >= G / 30 the subtrahend is >= G^4 / 225, which keeps 13 significant bits (>= 1.4e-41) once G >= 2.4e-10, i.e. max|X| >= 3.4e-4 (a
// sine of amplitude 1.3e-6: -117 dB re full scale); bins below G / 30 come out ambiguous whatever is lost (d^2 <= s^2 <= s G^2 / 450 against s K = 2 s G^2).
// Quieter frames, digital silence and non-finite input (whose bit pattern is the largest of all) take the fp64 transform; above 1e15 the squares could overflow.
constexpr unsigned GUARD_M_MIN_BITS = 0x33F00000u /* 1.1e-7 */, GUARD_M_MAX_BITS = 0x58635FA9u /* 1e15 */;

// largest of v over the 64 lanes (v_max_u32 with DPP: 0 is the identity), the same bits in every lane
__device__ __forceinline__ unsigned wave_max_u32(unsigned v)
{
#define PV_DPP_MAX(ctrl, rowmask) v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, rowmask, 0xF, false))
    PV_DPP_MAX(0x111, 0xF);       // row_shr:1
    PV_DPP_MAX(0x112, 0xF);       // row_shr:2
    PV_DPP_MAX(0x114, 0xF);       // row_shr:4
    PV_DPP_MAX(0x118, 0xF);       // row_shr:8: lane 15 of every row holds the row's maximum
    PV_DPP_MAX(0x142, 0xA);       // row_bcast:15 into rows 1 and 3
    PV_DPP_MAX(0x143, 0xC);       // row_bcast:31 into rows 2 and 3: lane 63 holds the total
#undef PV_DPP_MAX
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

// ---- which transform FIRST: a frame's class is fixed by its samples, the order of evaluation is not ----
// A frame is of class B (fp64 spectrum and decisions) iff the guard test above, taken on the fp32 magnitudes, finds an ambiguous bin or the frame out of range;
// else of class A (fp32 spectrum, fp32 decisions).  A chain whose frames keep coming out as class B (clean tonal material, 16-bit material, silence: every frame)
// wastes the fp32 transform on each of them.  For such chains the kernel runs the fp64 transform FIRST and proves the class from the fp64 magnitudes alone where it
// can: with |A32 - A64| <= E_b = 4 eps max|X| per bin (largest seen: 3.3) a pair that is within G_n = G - 2 E_b = 2 eps max|X| (relative part rho_n = rho - 16)
// of each other in the fp64 magnitudes is within G in the fp32 ones -- class B for sure, the fp32 transform is never run.  A frame that the narrow test does not
// decide takes the fp32 transform after all (and, if that says B, the fp64 one again).  The predictor is a counter per chain -- +1 for a frame that is class B
// and provable, -3 otherwise, fp64 first from 4 up: the order pays off above three provable frames in four -- and only ever changes the ORDER: chunked / call-split / resident runs, whose chains meet a frame with different counters, still agree bit for bit.  The same margin (2^12 ulps) keeps a
// largest magnitude near an end of the guarded range from being judged differently by the two transforms.
constexpr float GUARD_GN = PV_GUARD_G - 8.0f, GUARD_RHON = PV_GUARD_RHO - 16.0f;
static_assert(GUARD_GN > 0.f && GUARD_RHON > 0.f, "the guard band is too narrow to prove a frame's class from its fp64 magnitudes");
constexpr float GUARD_CKN = 2.0f * GUARD_GN * GUARD_GN * GUARD_EPS * GUARD_EPS;
constexpr float GUARD_RN = GUARD_RHON * GUARD_RHON * GUARD_EPS * GUARD_EPS;
constexpr unsigned GUARD_M_SLACK = 1u << 12;
constexpr int PRED_MAX = 7;                  // the per-chain counter saturates here ...
constexpr unsigned PRED_WIDE = 4u;           // ... and from here on the chain runs the fp64 transform first

// |X|^2 of an fp32 bin, roundings spelled out (every instance must form the same bits: the class of a frame rests on them)
__device__ __forceinline__ float mag32(pk::c32 x) { return __fmaf_rn(x.y, x.y, __fmul_rn(x.x, x.x)); }
// the absolute part K of a frame's guard band from the bit pattern of its largest fp32 magnitude; 0: out of the guarded range
__device__ __forceinline__ float guard_k_of(unsigned mbits) { return (mbits >= GUARD_M_MIN_BITS && mbits < GUARD_M_MAX_BITS) ? GUARD_CK * __uint_as_float(mbits) : 0.f; }

}  // namespace
