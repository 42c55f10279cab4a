// overlap_microbench.hip -- does a wave that keeps LDS traffic in flight while it issues VALU work of an independent frame finish sooner than
// one that alternates (the software-pipelining question of DESIGN.md section 7)?  One "phase pair" = a transpose-like LDS exchange
// (8 ds_write_b128, fence, 8 ds_read_b128) + 60 fp64 instructions.
//   mode 0 (alternating, what the kernels do): exchange; wait; VALU that depends on the loaded data
//   mode 1 (software-pipelined): exchange issued; VALU of the OTHER frame (independent registers); wait; consume the loads (8 adds)
// Reports shader clocks per phase pair per CU at 4, 8, 12, 16 waves per CU.  Design aid, not product.
// build: hipcc -O3 --offload-arch=gfx950 -o overlap_microbench overlap_microbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v2d __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(1024) void kern(long long *out, int iters, double seed)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int l = threadIdx.x & 63, wv = threadIdx.x >> 6;
    v2d *S = reinterpret_cast<v2d *>(smem + wv * 9216);
    for (int i = threadIdx.x; i < 36864; i += blockDim.x) reinterpret_cast<float *>(smem)[i] = 0.f;
    __syncthreads();
    v2d a[8], b[8];
#pragma unroll
    for (int r = 0; r < 8; r++) { a[r] = v2d{seed + r, seed - r}; b[r] = v2d{seed * r, seed + 2 * r}; }
    const double c = 1.0 + 1e-12 * seed;
    const int lh = l >> 3, ll = l & 7;
    const long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
        // exchange of frame A
#pragma unroll
        for (int k = 0; k < 8; k++) S[k * 72 + l] = a[k];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        v2d t[8];
#pragma unroll
        for (int n = 0; n < 8; n++) t[n] = S[lh * 72 + 8 * n + ll];
        if (MODE == 1) {
            // 60 fp64 instructions on frame B's registers (independent of the loads)
#pragma unroll
            for (int rep = 0; rep < 4; rep++)
#pragma unroll
                for (int r = 0; r < 8; r++) { b[r].x = b[r].x * c + b[(r + 1) & 7].y; if (rep < 3) b[r].y = b[r].y * c - b[(r + 3) & 7].x; }
#pragma unroll
            for (int r = 0; r < 8; r++) a[r] = t[r] + a[r];                    // consume (8 x 2 adds)
        } else {
#pragma unroll
            for (int r = 0; r < 8; r++) a[r] = t[r] + a[r];                    // consume first: the wave waits for the loads here
            asm volatile("" ::: "memory");
#pragma unroll
            for (int rep = 0; rep < 4; rep++)
#pragma unroll
                for (int r = 0; r < 8; r++) { b[r].x = b[r].x * c + a[(r + 1) & 7].y * 1e-30; if (rep < 3) b[r].y = b[r].y * c - b[(r + 3) & 7].x; }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    const long long t1 = clock64();
    if (l == 0) out[blockIdx.x * (blockDim.x >> 6) + wv] = t1 - t0;
    double acc = 0;
#pragma unroll
    for (int r = 0; r < 8; r++) acc += a[r].x + a[r].y + b[r].x + b[r].y;
    if (acc == 1.2345) out[0] = 0;
}

template <int MODE> static double run(int waves)
{
    const int blocks = 256, iters = 2000;
    long long *c; hipMalloc(&c, sizeof(long long) * blocks * waves);
    for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL((kern<MODE>), dim3(blocks), dim3(64 * waves), 16 * 9216, 0, c, iters, 1.5);
    hipDeviceSynchronize();
    std::vector<long long> h(blocks * waves);
    hipMemcpy(h.data(), c, sizeof(long long) * h.size(), hipMemcpyDeviceToHost); hipFree(c);
    double s = 0; for (auto v : h) s += (double)v;
    return s / h.size() / iters;          // cycles per phase pair per wave
}

int main()
{
    printf("phase pair = 8 ds_write_b128 + fence + 8 ds_read_b128 + ~60 fp64 instructions + 16 fp64 adds; cycles per pair per WAVE, and per CU (= per wave / waves)\n");
    for (int w : {4, 8, 12, 16}) {
        const double a = run<0>(w), b = run<1>(w);
        printf("waves/CU %2d: alternating %7.1f (%6.1f per CU)   software-pipelined %7.1f (%6.1f per CU)   ratio %.3f\n", w, a, a / w, b, b / w, a / b);
    }
    return 0;
}
