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

// 512-point complex FFT across one wave: in/out layout lane l, reg r <-> element l + 64 r.
// TW1[k*64 + l] = W_512^{l k}, TW2[k*8 + n0] = W_64^{n0 k} (k = 1..7) live in LDS, shared by the waves of the workgroup,
// already conjugated / rounded for the inverse fp32 instance.
template <typename T, bool INV>
__device__ __forceinline__ void fft512_wave(typename v2t<T>::type (&a)[8], typename v2t<T>::type *S, const typename v2t<T>::type *TW1,
                                            const typename v2t<T>::type *TW2, int l)
{
    const int lh = l >> 3, ll = l & 7;
    // pass 1: DFT over n2 (register index); twiddle W_512^{l*k0}
    radix8<T, INV>(a);
#pragma unroll
    for (int k = 1; k < 8; k++) a[k] = cmul(a[k], TW1[k * 64 + l]);
    // transpose 1: [reg k0][lane (n1,n0)] -> [reg n1][lane (k0,n0)]
#pragma unroll
    for (int k = 0; k < 8; k++) S[k * TP + l] = a[k];
    wave_sync();
#pragma unroll
    for (int n = 0; n < 8; n++) a[n] = S[lh * TP + 8 * n + ll];
    wave_sync();
    // pass 2: DFT over n1; twiddle W_64^{n0*k1}
    radix8<T, INV>(a);
#pragma unroll
    for (int k = 1; k < 8; k++) a[k] = cmul(a[k], TW2[k * 8 + ll]);
    // transpose 2: [reg k1][lane (k0,n0)] -> [reg n0][lane (k1,k0)], skewed rows
#pragma unroll
    for (int k = 0; k < 8; k++) S[k * TP + lh * 8 + ((ll + lh) & 7)] = a[k];
    wave_sync();
#pragma unroll
    for (int n = 0; n < 8; n++) a[n] = S[lh * TP + ll * 8 + ((n + ll) & 7)];
    wave_sync();
    // pass 3: DFT over n0 -> k2; lane l now holds X[l + 64 k2]
    radix8<T, INV>(a);
}


// The inverse instance in packed fp32 (pv_pk_math.h): 106 packed instructions instead of ~210.  Its LDS traffic is organised around the
// measured costs of the LDS pipe (tools/valu_microbench2.hip: a read costs ~3 cycles per wave-instruction whatever its width, ds_write_b64
// ~5.8, ds_write_b128 ~9.1): the twiddles of two consecutive k come from ONE ds_read_b128 of a pair-interleaved table, and the transposes
// write register PAIRS (4 ds_write_b128 instead of 8 ds_write_b64); a reader picks the half it needs with the address.  Rows of 64 pair slots
// are padded to TPP = 72 slots (1152 B = 128 mod 256): the four rows a read touches fall on alternating halves of the 64 banks -> 2 passes,
// the minimum for 512 bytes (checked with tools/lds_layout_check.py pairs).
constexpr int TPP = 72;
__device__ __forceinline__ void fft512_wave_inv_pk(pk::c32 (&a)[8], pk::c32 *S, const v4f *TW1F4, const v4f *TW2F4, int l)
{
    const int lh = l >> 3, ll = l & 7;
    v4f *S4 = reinterpret_cast<v4f *>(S);
    pk::radix8_inv(a);
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const v4f t = TW1F4[j * 64 + l];
        if (j) a[2 * j] = pk::cmul(a[2 * j], pk::c32{t.x, t.y});
        a[2 * j + 1] = pk::cmul(a[2 * j + 1], pk::c32{t.z, t.w});
    }
    // transpose 1: [reg k0][lane (n1,n0)] -> [reg n1][lane (k0,n0)]; pair row k0 >> 1, half k0 & 1
#pragma unroll
    for (int j = 0; j < 4; j++) S4[j * TPP + l] = v4f{a[2 * j].x, a[2 * j].y, a[2 * j + 1].x, a[2 * j + 1].y};
    wave_sync();
#pragma unroll
    for (int n = 0; n < 8; n++) a[n] = S[2 * ((lh >> 1) * TPP + 8 * n + ll) + (lh & 1)];
    wave_sync();
    pk::radix8_inv(a);
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const v4f t = TW2F4[j * 8 + ll];
        if (j) a[2 * j] = pk::cmul(a[2 * j], pk::c32{t.x, t.y});
        a[2 * j + 1] = pk::cmul(a[2 * j + 1], pk::c32{t.z, t.w});
    }
    // transpose 2: [reg k1][lane (k0,n0)] -> [reg n0][lane (k1,k0)], skewed columns
#pragma unroll
    for (int j = 0; j < 4; j++) S4[j * TPP + lh * 8 + ((ll + lh) & 7)] = v4f{a[2 * j].x, a[2 * j].y, a[2 * j + 1].x, a[2 * j + 1].y};
    wave_sync();
#pragma unroll
    for (int n = 0; n < 8; n++) a[n] = S[2 * ((lh >> 1) * TPP + ll * 8 + ((n + ll) & 7)) + (lh & 1)];
    wave_sync();
    pk::radix8_inv(a);
}

}  // namespace
