// lds_microbench.hip -- measures the LDS-pipe cost (shader clocks per wave-instruction, 8 waves of one CU contending) of the access
// patterns the wave kernel uses, to choose conflict-free layouts from measurements rather than from the bank model alone.
// Design aid, not part of the product.  build: hipcc -O3 --offload-arch=gfx950 -o lds_microbench lds_microbench.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>
#include <functional>

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
enum Op { RD32, RD64, RD128, WR32, WR64, WR128, RD2ST64_B64, RD2_B64_72, WR2_B64_72, RD2_B32_1, BPERM };

template <int OP>
__global__ __launch_bounds__(1024) void kern(const int *offs, long long *cycles, int iters)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int l = threadIdx.x & 63;
    const unsigned a = (unsigned)offs[l] + 0u;
    v4f v{1.f, 2.f, 3.f, 4.f}; v2f v2{1.f, 2.f};
    float4 acc{0.f, 0.f, 0.f, 0.f};
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) reinterpret_cast<float *>(smem)[i] = 0.f;
    __syncthreads();
    const long long t0 = clock64();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (OP == RD32) { float x; asm volatile("ds_read_b32 %0, %1" : "=v"(x) : "v"(a)); (void)x; }
            if (OP == RD64) { v2f x; asm volatile("ds_read_b64 %0, %1" : "=v"(x) : "v"(a)); (void)x; }
            if (OP == RD128) { v4f x; asm volatile("ds_read_b128 %0, %1" : "=v"(x) : "v"(a)); (void)x; }
            if (OP == WR32) asm volatile("ds_write_b32 %0, %1" ::"v"(a), "v"(v.x) : "memory");
            if (OP == WR64) asm volatile("ds_write_b64 %0, %1" ::"v"(a), "v"(v2) : "memory");
            if (OP == RD2ST64_B64) { v4f x; asm volatile("ds_read2st64_b64 %0, %1 offset1:1" : "=v"(x) : "v"(a)); (void)x; }
            if (OP == RD2_B64_72) { v4f x; asm volatile("ds_read2_b64 %0, %1 offset1:72" : "=v"(x) : "v"(a)); (void)x; }
            if (OP == WR2_B64_72) asm volatile("ds_write2_b64 %0, %1, %2 offset1:72" ::"v"(a), "v"(v2), "v"(v2) : "memory");
            if (OP == RD2_B32_1) { v2f x; asm volatile("ds_read2_b32 %0, %1 offset1:1" : "=v"(x) : "v"(a)); (void)x; }
            if (OP == BPERM) { float x; asm volatile("ds_bpermute_b32 %0, %1, %2" : "=v"(x) : "v"(a), "v"(v.x)); (void)x; }
            if (OP == WR128) asm volatile("ds_write_b128 %0, %1" ::"v"(a), "v"(v) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const long long t1 = clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) cycles[0] = t1 - t0;
    if (acc.x == 123.f) cycles[1] = 1;
}

static double run(int op, const std::vector<int> &offs, int waves)
{
    int *d; long long *c;
    hipMalloc(&d, 64 * 4); hipMalloc(&c, 16);
    hipMemcpy(d, offs.data(), 64 * 4, hipMemcpyHostToDevice);
    const int iters = 2000;
    auto launch = [&](auto k) { hipLaunchKernelGGL(k, dim3(256), dim3(64 * waves), 65536, 0, d, c, iters); };
    for (int rep = 0; rep < 1; rep++) {
        switch (op) {
        case RD32: launch(kern<RD32>); break; case RD64: launch(kern<RD64>); break; case RD128: launch(kern<RD128>); break;
        case WR32: launch(kern<WR32>); break; case WR64: launch(kern<WR64>); break; case WR128: launch(kern<WR128>); break; case RD2ST64_B64: launch(kern<RD2ST64_B64>); break; case RD2_B64_72: launch(kern<RD2_B64_72>); break;
        case WR2_B64_72: launch(kern<WR2_B64_72>); break; case RD2_B32_1: launch(kern<RD2_B32_1>); break; case BPERM: launch(kern<BPERM>); break;
        }
        hipDeviceSynchronize();
    }
    long long h[2];
    hipMemcpy(h, c, 16, hipMemcpyDeviceToHost);
    hipFree(d); hipFree(c);
    return (double)h[0] / ((double)iters * 8 * waves);
}

int main()
{
    struct Pat { std::string name; int op; std::function<int(int)> f; };
    std::vector<Pat> pats;
    auto add = [&](std::string n, int op, std::function<int(int)> f) { pats.push_back({n, op, f}); };
    // references: contiguous
    add("rd32 contiguous", RD32, [](int l) { return 4 * l; });
    add("rd64 contiguous", RD64, [](int l) { return 8 * l; });
    add("rd128 contiguous", RD128, [](int l) { return 16 * l; });
    add("wr32 contiguous", WR32, [](int l) { return 4 * l; });
    add("wr64 contiguous", WR64, [](int l) { return 8 * l; });
    add("wr128 contiguous", WR128, [](int l) { return 16 * l; });
    // search: pv_wg_kernel transpose 3 with a rotation of the 8G-element block by SG * kA
    for (int G : {2, 4}) for (int E : {16, 8}) for (int A : {0, 1, 2, 4}) for (int SG = 0; SG < 8 * G; SG++) {
        const int P = 8 * (8 * G + A);
        std::string tag = " G" + std::to_string(G) + " E" + std::to_string(E) + " A" + std::to_string(A) + " RP" + std::to_string(SG);
        const int WR = (E == 16) ? WR128 : WR64, RD = (E == 16) ? RD128 : RD64;
        add("T3wr" + tag, WR, [=](int l) { int t = l, kA1 = t / (8 * G), tlo = t % (8 * G); return E * (3 * P + kA1 * (8 * G + A) + ((tlo + SG * kA1) % (8 * G))) % 60000; });
        add("T3rd" + tag, RD, [=](int l) { int t = l, kA3 = t & 7, kB3 = (t >> 3) & 7; int q = 3; return E * ((0 + G * (q / G)) * P + kA3 * (8 * G + A) + ((kB3 * G + (q % G) + SG * kA3) % (8 * G))) % 60000; });
    }
    for (auto &p : pats) {
        std::vector<int> offs(64);
        for (int l = 0; l < 64; l++) offs[l] = p.f(l);
        printf("%-40s  16 waves: %6.1f clk/instr\n", p.name.c_str(), run(p.op, offs, 16));
    }
    return 0;
}
