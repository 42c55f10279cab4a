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
    add("rd2st64_b64 contiguous (Hann pair)", RD2ST64_B64, [](int l) { return 8 * l; });
    add("rd2_b64 +72 contiguous", RD2_B64_72, [](int l) { return 8 * l; });
    add("wr2_b64 +72 contiguous (T1f pair)", WR2_B64_72, [](int l) { return 8 * l; });
    add("rd2_b64 +72 T1f read", RD2_B64_72, [](int l) { return 8 * ((l >> 3) * 72 + (l & 7)); });
    add("rd2_b32 +1 stride 8B", RD2_B32_1, [](int l) { return 8 * l; });
    add("bpermute lane 64-l", BPERM, [](int l) { return 4 * ((64 - l) & 63); });
    add("bpermute identity", BPERM, [](int l) { return 4 * l; });
    add("bpermute broadcast 5", BPERM, [](int l) { return 4 * 5; });
    add("rd u16-ish random (b32 stride 2B*k)", RD32, [](int l) { return 4 * ((l * 37) & 255); });
    add("wr64 random-ish", WR64, [](int l) { return 8 * ((l * 37) & 511); });
    add("rd128 8l rowpad 0", RD128, [](int l) { return 4 * (8 * l); });
    add("wr128 8l rowpad 0", WR128, [](int l) { return 4 * (8 * l); });
    add("wr128 8l pad4per4lanes", WR128, [](int l) { return 4 * (8 * l + 4 * (l >> 2)); });
    add("rd128 8l pad4per4lanes", RD128, [](int l) { return 4 * (8 * l + 4 * (l >> 2)); });
    // fp64 transposes of fft512_wave (TP = 72 double2 per row)
    add("T1 wr128 S[k*72+l]", WR128, [](int l) { return 16 * (3 * 72 + l); });
    add("T1 rd128 S[lh*72+8n+ll]", RD128, [](int l) { return 16 * ((l >> 3) * 72 + 8 * 3 + (l & 7)); });
    add("T2 wr128 skew", WR128, [](int l) { return 16 * (3 * 72 + (l >> 3) * 8 + (((l & 7) + (l >> 3)) & 7)); });
    add("T2 rd128 skew", RD128, [](int l) { return 16 * ((l >> 3) * 72 + (l & 7) * 8 + ((3 + (l & 7)) & 7)); });
    // fp32 versions (float2, TP = 72)
    add("T1f wr64 S[k*72+l]", WR64, [](int l) { return 8 * (3 * 72 + l); });
    add("T1f rd64 S[lh*72+8n+ll]", RD64, [](int l) { return 8 * ((l >> 3) * 72 + 8 * 3 + (l & 7)); });
    add("T2f wr64 skew", WR64, [](int l) { return 8 * (3 * 72 + (l >> 3) * 8 + (((l & 7) + (l >> 3)) & 7)); });
    add("T2f rd64 skew", RD64, [](int l) { return 8 * ((l >> 3) * 72 + (l & 7) * 8 + ((3 + (l & 7)) & 7)); });
    // twiddle tables
    add("TW2 rd128 [k*8+ll]", RD128, [](int l) { return 16 * (3 * 8 + (l & 7)); });
    add("TW2F rd64 [k*8+ll]", RD64, [](int l) { return 8 * (3 * 8 + (l & 7)); });
    add("Y rd64 reversed", RD64, [](int l) { return 8 * (512 - l); });
    for (auto &p : pats) {
        std::vector<int> offs(64);
        for (int l = 0; l < 64; l++) offs[l] = p.f(l);
        printf("%-40s  1 wave: %6.1f   8 waves: %6.1f  16 waves: %6.1f clk/instr\n", p.name.c_str(), run(p.op, offs, 1), run(p.op, offs, 8), run(p.op, offs, 16));
    }
    return 0;
}
