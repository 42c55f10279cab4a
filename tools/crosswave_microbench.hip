// crosswave_microbench.hip -- can a CU overlap the LDS traffic of SOME waves with the VALU work of OTHER waves?
// The kernels' frame time equals VALU issue time + LDS pipe time.  If that were a property of the dependency structure
// inside a wave, waves that do nothing but LDS exchanges next to waves that do nothing but fp64 arithmetic would overlap perfectly
// (time = max); if it is how the CU issues, the mix costs the sum.  12 waves per CU (3 per SIMD), NL of them run the LDS loop
// (8 ds_write_b128 + 8 ds_read_b128 per iteration), the rest the VALU loop (64 dependent-chain-free v_fma_f64 per iteration); every wave runs a
// fixed number of iterations and reports its own clock count.
// build: hipcc -O3 --offload-arch=gfx950 -o crosswave_microbench crosswave_microbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v2d __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(768) void kern(long long *out, int iters, int nl_mask, double seed)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int l = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    v2d *S = reinterpret_cast<v2d *>(smem + wv * 9216);
    for (int i = threadIdx.x; i < 12 * 9216 / 4; i += blockDim.x) reinterpret_cast<float *>(smem)[i] = 0.f;
    __syncthreads();
    v2d a[8];
#pragma unroll
    for (int r = 0; r < 8; r++) a[r] = v2d{seed + r, seed - r};
    const double c = 1.0 + 1e-12 * seed;
    const int lh = l >> 3, ll = l & 7;
    const bool lds_wave = (nl_mask >> wv) & 1;
    const bool idle = (nl_mask >> (16 + wv)) & 1;                          // waves that do nothing (to run one kind alone at the same occupancy)
    __syncthreads();
    const long long t0 = clock64();
    if (idle) {
    } else if (lds_wave) {
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int k = 0; k < 8; k++) S[k * 72 + l] = a[k];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int n = 0; n < 8; n++) a[n] = S[lh * 72 + 8 * n + ll];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    } else {
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int rep = 0; rep < 4; rep++)
#pragma unroll
                for (int r = 0; r < 8; r++) { a[r].x = a[r].x * c + a[(r + 1) & 7].y; a[r].y = a[r].y * c - a[(r + 3) & 7].x; }
        }
    }
    const long long t1 = clock64();
    if (l == 0) out[blockIdx.x * 12 + wv] = t1 - t0;
    double acc = 0;
#pragma unroll
    for (int r = 0; r < 8; r++) acc += a[r].x + a[r].y;
    if (acc == 1.2345) out[0] = 0;
}

static double run(int nl_mask, int iters, int kind /*0: mean of LDS waves, 1: mean of VALU waves*/)
{
    long long *d;
    const int blocks = 256;
    hipMalloc(&d, sizeof(long long) * blocks * 12);
    hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 12 * 9216);
    for (int w = 0; w < 2; w++) hipLaunchKernelGGL(kern, dim3(blocks), dim3(768), 12 * 9216, 0, d, iters, nl_mask, 1.0);
    hipDeviceSynchronize();
    std::vector<long long> h(blocks * 12);
    hipMemcpy(h.data(), d, sizeof(long long) * blocks * 12, hipMemcpyDeviceToHost);
    hipFree(d);
    double s = 0; int n = 0;
    for (int b = 0; b < blocks; b++)
        for (int w = 0; w < 12; w++) {
            const bool lw = (nl_mask >> w) & 1, idle = (nl_mask >> (16 + w)) & 1;
            if (idle) continue;
            if ((kind == 0) == lw) { s += (double)h[b * 12 + w]; n++; }
        }
    return n ? s / n / iters : 0.0;
}

int main()
{
    const int iters = 2000;
    // wave w sits on SIMD w % 4.  Masks: bit w = LDS wave, bit 16 + w = idle.
    const int one_per_simd = 0x00F;          // waves 0..3 (one per SIMD) run the LDS loop
    const int two_per_simd = 0x0FF;          // waves 0..7
    const int all = 0xFFF;
    printf("cycles per iteration of a wave (LDS loop: 8 ds_write_b128 + 8 ds_read_b128; VALU loop: 64 v_fma_f64), 12 wave slots per CU\n");
    printf("LDS waves alone    : 4 waves %.0f   8 waves %.0f   12 waves %.0f\n", run(one_per_simd | (0xFF0 << 16), iters, 0), run(two_per_simd | (0xF00 << 16), iters, 0), run(all, iters, 0));
    printf("VALU waves alone   : 4 waves %.0f   8 waves %.0f   12 waves %.0f\n", run(0 | (0xFF0 << 16), iters, 1), run(0 | (0xF00 << 16), iters, 1), run(0, iters, 1));
    printf("4 LDS + 8 VALU     : LDS wave %.0f   VALU wave %.0f\n", run(one_per_simd, iters, 0), run(one_per_simd, iters, 1));
    printf("8 LDS + 4 VALU     : LDS wave %.0f   VALU wave %.0f\n", run(two_per_simd, iters, 0), run(two_per_simd, iters, 1));
    printf("4 LDS + 4 VALU (4 idle): LDS wave %.0f   VALU wave %.0f\n", run(one_per_simd | (0xF00 << 16), iters, 0), run(one_per_simd | (0xF00 << 16), iters, 1));
    return 0;
}
