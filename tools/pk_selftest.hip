// pk_selftest.hip -- checks every helper of pv_pk_math.h against scalar arithmetic on the device (design aid; run on the GPU box).
#include "../phaze_amd/csrc/pv_pk_math.h"
#include <cstdio>
#include <cmath>
#include <complex>
#include <vector>
using pk::c32;
__global__ void k(const float *in, float *out)
{
    const int t = threadIdx.x;
    c32 a{in[6 * t], in[6 * t + 1]}, b{in[6 * t + 2], in[6 * t + 3]}, c{in[6 * t + 4], in[6 * t + 5]};
    c32 r[17];
    r[0] = pk::add(a, b); r[1] = pk::sub(a, b); r[2] = pk::add_j(a, b); r[3] = pk::sub_j(a, b); r[4] = pk::add_conj(a, b); r[5] = pk::sub_conj(a, b);
    r[6] = pk::neg_add_j(a, b); r[7] = pk::mul(a, b); r[8] = pk::cmul(a, b); r[9] = pk::fma(a, b, c); r[10] = pk::fnma(a, b, c);
    const c32 s{b.x, b.x};
    r[11] = pk::fma_j(a, s, c); r[12] = pk::fnma_j(a, s, c); r[13] = pk::fma_addj(a, b, c); r[14] = pk::fma_conj_subj(a, s, c);
    r[15] = pk::mul_ay(a, b); r[16] = pk::fma_ax(a, b, c);
    for (int i = 0; i < 17; i++) { out[(t * 17 + i) * 2] = r[i].x; out[(t * 17 + i) * 2 + 1] = r[i].y; }
}
__global__ void kfft(const float *in, float *out)
{
    c32 a[8];
    for (int i = 0; i < 8; i++) a[i] = c32{in[2 * i], in[2 * i + 1]};
    pk::radix8_inv(a);
    for (int i = 0; i < 8; i++) { out[2 * i] = a[i].x; out[2 * i + 1] = a[i].y; }
}
int main()
{
    const int T = 64;
    std::vector<float> h(6 * T), o(T * 17 * 2);
    for (auto &v : h) v = (float)rand() / RAND_MAX - 0.5f;
    float *di, *dout;
    hipMalloc(&di, h.size() * 4); hipMalloc(&dout, o.size() * 4);
    hipMemcpy(di, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(T), 0, 0, di, dout);
    hipMemcpy(o.data(), dout, o.size() * 4, hipMemcpyDeviceToHost);
    const char *names[17] = {"add", "sub", "add_j", "sub_j", "add_conj", "sub_conj", "neg_add_j", "mul", "cmul", "fma", "fnma", "fma_j", "fnma_j", "fma_addj", "fma_conj_subj", "mul_ay", "fma_ax"};
    typedef std::complex<double> C;
    const C J(0, 1);
    double worst[17] = {};
    for (int t = 0; t < T; t++) {
        C a(h[6 * t], h[6 * t + 1]), b(h[6 * t + 2], h[6 * t + 3]), c(h[6 * t + 4], h[6 * t + 5]);
        const double s = b.real();
        C e[17] = {a + b, a - b, a + J * b, a - J * b, a + std::conj(b), a - std::conj(b), -a + J * b, C(a.real() * b.real(), a.imag() * b.imag()), a * b,
                   C(a.real() * b.real() + c.real(), a.imag() * b.imag() + c.imag()), C(-a.real() * b.real() + c.real(), -a.imag() * b.imag() + c.imag()),
                   c + J * (a * s), c - J * (a * s), C(a.real() * b.real(), a.imag() * b.imag()) + J * c, std::conj(a * s - J * c),
                   C(-a.imag() * b.imag(), a.imag() * b.real()), C(a.real() * b.real() + c.real(), a.real() * b.imag() + c.imag())};
        for (int i = 0; i < 17; i++) {
            const double d = std::abs(e[i] - C(o[(t * 17 + i) * 2], o[(t * 17 + i) * 2 + 1]));
            if (d > worst[i]) worst[i] = d;
        }
    }
    int bad = 0;
    for (int i = 0; i < 17; i++) { printf("%-14s max err %.3g %s\n", names[i], worst[i], worst[i] < 1e-6 ? "ok" : "WRONG"); bad += worst[i] >= 1e-6; }
    // radix-8 inverse
    std::vector<float> fi(16), fo(16);
    for (auto &v : fi) v = (float)rand() / RAND_MAX - 0.5f;
    hipMemcpy(di, fi.data(), 64, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(kfft, dim3(1), dim3(1), 0, 0, di, dout);
    hipMemcpy(fo.data(), dout, 64, hipMemcpyDeviceToHost);
    double w8 = 0;
    for (int kx = 0; kx < 8; kx++) {
        C acc = 0;
        for (int n = 0; n < 8; n++) acc += C(fi[2 * n], fi[2 * n + 1]) * std::polar(1.0, 2 * M_PI * n * kx / 8);
        w8 = std::max(w8, std::abs(acc - C(fo[2 * kx], fo[2 * kx + 1])));
    }
    printf("radix8_inv     max err %.3g %s\n", w8, w8 < 1e-5 ? "ok" : "WRONG");
    bad += w8 >= 1e-5;
    return bad;
}
