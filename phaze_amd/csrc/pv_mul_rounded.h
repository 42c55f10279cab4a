// pv_mul_rounded.h -- the one definition of mul_rounded, shared by the generic kernel (pv_kernels.hip) and the register kernels (pv_device_common.h).
#pragma once
#include <hip/hip_runtime.h>

namespace {

// a * b ROUNDED to fp32, as an operation of its own.  hipcc contracts a plain product into a following add or subtract (v_fma_f32 / v_pk_fma_f32) -- also `__fmul_rn`, which
// this toolchain defines as `x * y` --: one rounding less than the reference, whose windowed samples and windowed frames are Float32Array elements (pv:55,67).  An asm
// multiply is opaque to that (pure, not volatile: free to be scheduled or dropped).  Found by the reference-width flavour of pv_wg16_kernel in round 5: the fused form differs
// from the reference by one ulp in ~40 % of the output samples (4e-9 RMS), which the product's own fp32 inverse (6e-9) had covered.
__device__ __forceinline__ float mul_rounded(float a, float b)
{
    float d;
    asm("v_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

}  // namespace
