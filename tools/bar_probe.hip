// Probe for the streaming form: how fast can the host hand ONE render quantum to a kernel that is already running?
//   variant H: control word + input in pinned HOST memory (the resident kernel of tools/experiments/resident_stream_kernel.patch): every poll is a PCIe read
//   variant D: control word + input in DEVICE memory, written by the host through the large BAR (posted writes); the wave polls its own HBM / L2
// Both: the wave answers by storing the sequence number into pinned host memory (posted write), the host spins on its own memory.
// Prints the round-trip percentiles of N quanta for both variants and for payloads of 0 / 1 KB / 8 KB.
//   hipcc --offload-arch=gfx950 -O2 -o tools/bar_probe tools/bar_probe.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
#include <emmintrin.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void resident(volatile unsigned *ctrl, const float *in, float *out_host, unsigned *done_host, int nwords, unsigned last)
{
    const int l = threadIdx.x;
    unsigned seq = 1;
    for (;;) {
        unsigned c;
        do { c = __hip_atomic_load(const_cast<unsigned *>(ctrl), __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM); } while (c < seq);
        float acc = 0.f;
        for (int i = l; i < nwords; i += 64) acc += __builtin_nontemporal_load(in + i);     // the quantum's input
        for (int i = l; i < 256; i += 64) out_host[i] = acc + (float)i;                       // 1 KB of output
        __threadfence_system();
        if (l == 0) __hip_atomic_store(done_host, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        if (seq == last) return;
        seq++;
    }
}

static double pct(std::vector<double> &v, double p) { std::sort(v.begin(), v.end()); return v[(size_t)(p * (v.size() - 1))]; }

int main()
{
    int large_bar = -1;
    CK(hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, 0));
    printf("hipDeviceAttributeIsLargeBar = %d\n", large_bar);
    const unsigned N = 3000;
    float *out_h; unsigned *done_h; float *in_h; unsigned *ctrl_h;
    CK(hipHostMalloc(&out_h, 4096, hipHostMallocMapped)); CK(hipHostMalloc(&done_h, 4096, hipHostMallocMapped));
    CK(hipHostMalloc(&in_h, 65536, hipHostMallocMapped)); CK(hipHostMalloc(&ctrl_h, 4096, hipHostMallocMapped));
    float *in_d; unsigned *ctrl_d;
    CK(hipMalloc(&in_d, 65536)); CK(hipMalloc(&ctrl_d, 4096));
    std::vector<float> src(16384, 1.0f);
    for (int variant = 0; variant < 2; variant++) {
        if (variant == 1 && large_bar != 1) { printf("variant D skipped: no large BAR\n"); break; }
        for (int nwords : {0, 256, 2048}) {
            volatile unsigned *ctrl = variant ? ctrl_d : ctrl_h;
            float *in = variant ? in_d : in_h;
            if (variant) { CK(hipMemset(ctrl_d, 0, 4096)); CK(hipDeviceSynchronize()); } else ctrl_h[0] = 0;
            done_h[0] = 0;
            hipLaunchKernelGGL(resident, dim3(1), dim3(64), 0, 0, ctrl, in, out_h, done_h, nwords, N);
            CK(hipGetLastError());
            std::vector<double> us;
            for (unsigned s = 1; s <= N; s++) {
                const auto t0 = std::chrono::steady_clock::now();
                if (nwords) memcpy(in, src.data(), (size_t)nwords * 4);              // host -> (pinned | BAR)
                _mm_sfence();
                *ctrl = s;
                _mm_sfence();
                while (*(volatile unsigned *)done_h < s) { }
                const auto t1 = std::chrono::steady_clock::now();
                if (s > 200) us.push_back(std::chrono::duration<double, std::micro>(t1 - t0).count());
                for (volatile int k = 0; k < 2000; k++) { }                           // a little idle time between quanta
            }
            CK(hipDeviceSynchronize());
            printf("variant %s  payload %5d B: round trip p50 %.2f us  p90 %.2f  p99 %.2f  (n = %zu)\n", variant ? "D (BAR writes, device polls HBM)" : "H (pinned host, device polls PCIe)",
                   nwords * 4, pct(us, 0.5), pct(us, 0.9), pct(us, 0.99), us.size());
        }
    }
    return 0;
}
