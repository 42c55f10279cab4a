// pv_pk_math.h -- packed-fp32 complex arithmetic for gfx950 (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 with op_sel / neg modifiers).
//
// A complex number lives in one 64-bit VGPR pair (x = low dword, y = high dword).  The VOP3P source modifiers make conjugation,
// multiplication by +-j and the real/imaginary cross terms of a complex product free: op_sel[i] picks the dword of source i that feeds
// the LOW result, op_sel_hi[i] the one that feeds the HIGH result, neg_lo / neg_hi negate source i per result half.  The compiler does
// not fold these swaps (it emits v_mov pairs instead: 66 moves per 512-point FFT), hence the one-instruction asm helpers below.
// They are plain (non-volatile) asm: pure functions of their operands, free to be scheduled, CSE'd or dropped.
#pragma once
#include <hip/hip_runtime.h>

namespace pk {

typedef float c32 __attribute__((ext_vector_type(2)));

#define PV_PK2(name, text)                                                                          \
    __device__ __forceinline__ c32 name(c32 a, c32 b) { c32 d; asm(text : "=v"(d) : "v"(a), "v"(b)); return d; }
#define PV_PK3(name, text)                                                                          \
    __device__ __forceinline__ c32 name(c32 a, c32 b, c32 c) { c32 d; asm(text : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }

PV_PK2(add, "v_pk_add_f32 %0, %1, %2")                                                               // a + b
PV_PK2(sub, "v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]")                                     // a - b
PV_PK2(add_j, "v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]")                   // a + j b = (ax - by, ay + bx)
PV_PK2(sub_j, "v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]")                   // a - j b = (ax + by, ay - bx)
PV_PK2(add_conj, "v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]")                                             // a + conj(b)
PV_PK2(sub_conj, "v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]")                                             // a - conj(b)
PV_PK2(neg_add_j, "v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[1,1] neg_hi:[1,0]")  // -a + j b = (-ax - by, -ay + bx)
PV_PK2(mul, "v_pk_mul_f32 %0, %1, %2")                                                               // component-wise
PV_PK2(mul_conj, "v_pk_mul_f32 %0, %1, %2 neg_hi:[0,1]")                                             // conj of the component-wise product: (ax bx, -ay by)
PV_PK2(conj_add_j, "v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[1,1]") // conj(a + j b) = (ax - by, -ay - bx)
PV_PK2(mul_ay, "v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[1,0]")                  // (-ay by, ay bx)
PV_PK3(fma_ax, "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]")                       // (ax bx + cx, ax by + cy)
PV_PK3(fma, "v_pk_fma_f32 %0, %1, %2, %3")                                                           // a * b + c component-wise
PV_PK3(fnma, "v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[1,0,0] neg_hi:[1,0,0]")                            // -a * b + c
PV_PK3(fma_j, "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]")         // c + j (a * b)  (b = (s, s))
PV_PK3(fnma_j, "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]")        // c - j (a * b)  (b = (s, s))
PV_PK3(conj_fma_j, "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,1]")   // conj(c + j (a * b)) = (cx - ay s, -cy - ax s)  (b = (s, s))
PV_PK3(fms, "v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[0,0,1] neg_hi:[0,0,1]")                             // a * b - c component-wise
PV_PK3(fma_addj, "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,1,0] neg_lo:[0,0,1]")      // a * b + j c
PV_PK3(fma_conj_subj, "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,1,0] neg_hi:[1,0,0]") // conj(a * b - j c) = (ax bx + cy, -ay by + cx)  (b = (s, s))

#undef PV_PK2
#undef PV_PK3

// a * w (complex): two instructions
__device__ __forceinline__ c32 cmul(c32 a, c32 w) { return fma_ax(a, w, mul_ay(a, w)); }

// In-register 8-point inverse DFT (exp(+2 pi j nk/8)), natural order in and out: 26 packed instructions.
__device__ __forceinline__ void radix8_inv(c32 (&a)[8])
{
    const c32 hh{0.70710678118654752440f, 0.70710678118654752440f};
    const c32 b0 = add(a[0], a[4]), b4 = sub(a[0], a[4]);
    const c32 b1 = add(a[1], a[5]), b5 = sub(a[1], a[5]);
    const c32 b2 = add(a[2], a[6]), b6 = sub(a[2], a[6]);
    const c32 b3 = add(a[3], a[7]), b7 = sub(a[3], a[7]);
    {   // even outputs: 4-point DFT of b0..b3
        const c32 e0 = add(b0, b2), e1 = sub(b0, b2), e2 = add(b1, b3), t = sub(b1, b3);
        a[0] = add(e0, e2); a[4] = sub(e0, e2); a[2] = add_j(e1, t); a[6] = sub_j(e1, t);
    }
    {   // odd outputs: 4-point DFT of c_n = b_{n+4} W8^{-n}; c1 = h u1, c3 = h u3 with the factor h folded into the last stage
        const c32 u1 = add_j(b5, b5);          // b5 (1 + j)
        const c32 u3 = neg_add_j(b7, b7);      // b7 (-1 + j)
        const c32 e0 = add_j(b4, b6), e1 = sub_j(b4, b6);
        const c32 e2 = add(u1, u3), t = sub(u1, u3);
        a[1] = fma(e2, hh, e0); a[5] = fnma(e2, hh, e0); a[3] = fma_j(t, hh, e1); a[7] = fnma_j(t, hh, e1);
    }
}

}  // namespace pk
