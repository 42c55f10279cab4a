// pv_kernels.h -- host-visible launch interface of the CDNA4 kernels (internal; the public ABI is include/phaze_amd.h)
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <atomic>

struct PvKernelParams {
    const float *in;          // planar input, channel c at in + c*ch_stride, nhops*hop samples
    float *out;               // planar output, same layout
    long ch_stride;
    int nhops;                // hops (= process() calls) in this launch
    int hop;
    int frames_per_chunk;     // output hops per workgroup (per wave in the wave kernel)
    int nchunks;              // number of chunks per channel (set by the launcher)
    int nch;                  // number of channel slots of this launch (set by the launcher; wave kernel: chains are numbered channel-major)
    const float *pitch;       // pitchFactor per hop (k-rate, phase-vocoder.js:47)
    int pitch_stride;         // 0: one row shared by all channels; else row = (c / ch_per_stream)
    int ch_per_stream;
    const float *hist_in;     // [ch][N-hop] input history carried in   (ola-processor.js:59,121-127)
    float *hist_out;          // [ch][N-hop] carried out (other half of the ping-pong)
    const float *acc_in;      // [ch][N-hop] overlap-add accumulator tail carried in (ola-processor.js:77,130-137)
    float *acc_out;
    int t0_mod_n;             // timeCursor at the first hop of the launch, mod N (phase-vocoder.js:31,71)
    const double2 *tw64;      // exp(-2 pi j k / N), k in [0, N), fp64
    const float2 *tw32;       // same, rounded to fp32
    const float *hann;        // periodic Hann, fp32 (phase-vocoder.js:8-14), N values, followed by N values of 0.5 * Hann
    // test taps (all null in production)
    double *dbg_X;
    float *dbg_mag;
    int *dbg_flags;
    float *dbg_Y;
    int dbg_ch, dbg_frame;
    // streaming quantum (pv_process): when non-null, every frame chain stores done_seq into done[chain] once its output (and state) is written,
    // and the host spins on those words in pinned memory instead of going through hipStreamSynchronize (null in batch launches)
    unsigned *done;
    unsigned done_seq;
    // resident streaming kernel (PV_FLAG_PERSISTENT_STREAM): the kernel does not end after its quantum but polls ctl[0] (pinned host memory) for the
    // next sequence number.  ctl = {seq, nch, t0_mod_n, cur, stop}; state2[] = both halves of the state ping-pong (cur selects hist_in / acc_in).
    const unsigned *ctl;
    float *hist2[2], *acc2[2];
    unsigned idle_ticks;      // resident form: wall_clock64() ticks without work after which the waves leave (~50 ms; the host restarts them on demand)
    int in_cached;            // resident form: `in` / `pitch` are DEVICE memory the host rewrites through the BAR (cached in L2: system-scope loads);
                              // 0 = pinned host memory (uncached on the device: plain, coalesced loads behind the acquire fence of the poll)
    // N = 1024 batch launches: the chains of the launch sorted into two classes by pv_classify_chains (pv_wave_kernel.hip) -- chain_list[0 .. nchains) holds the
    // chains whose pitchFactor is >= 1 on every frame (count chain_count[0]), chain_list[nchains .. 2 nchains) the others (chain_count[1]); each class runs on
    // its own instance of the kernel.  Null: chains are numbered directly (streaming quantum, test tap)
    const unsigned *chain_list;
    const unsigned *chain_count;
    // round 5: the one-wave kernels take the peak decisions on a packed-fp32 forward transform behind a guard band and re-run a frame's forward transform in fp64
    // only when a decision is in doubt (pv_wave_kernel.hip: F32).  fwd64 != 0: every frame at the reference's width (PV_FLAG_FP64_FORWARD: the round-4 kernels).
    int fwd64;
    unsigned long long *fwd_stats;   // 128 x {frames computed by an F32 instance, frames of those that fell back}; null: not counted
    unsigned char *gscratch;  // N >= 16384 (pv_chain_kernel's global-scratch instances): pv_kernel_gscratch_bytes() per workgroup, workgroup (ch, chunk) at (ch * nchunks + chunk) * stride
    unsigned long long gscratch_stride;
    unsigned *stamps;         // measurement builds only (-DPV_STAMPS, tools/exp_headline.sh): [chain][16] accumulated s_memtime deltas per phase
};

// One-time per-device raise of a kernel's dynamic-LDS limit, shared by the launchers.  Handles may be driven from different host threads: the
// flags are atomics; two threads racing through a first launch both set the same attribute value, which is harmless.
inline hipError_t pv_set_dynamic_lds_once(std::atomic<bool> (&done)[16], const void *kernel, int bytes)
{
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::atomic<bool> &f = done[dev & 15];
    if (f.load(std::memory_order_acquire)) return hipSuccess;
    const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) f.store(true, std::memory_order_release);
    return e;
}

int pv_kernel_threads(int log2n);
size_t pv_kernel_lds_bytes(int log2n, int hop);
size_t pv_kernel_gscratch_bytes(int log2n, int hop);      // per workgroup; 0 for N <= 8192 (everything in LDS)
hipError_t pv_launch_chain(int log2n, const PvKernelParams &p, int nch, int nchunks, hipStream_t st);

// wave-per-frame kernel for N = 1024 (pv_wave_kernel.hip)
size_t pv_wave_lds_bytes();
int pv_wave_threads();
bool pv_wave_supported(int log2n, int hop);
// spread: > 0 every frame of the launch has pitchFactor >= 1 (the host knows: streaming quantum), 0 it does not, < 0 unknown: classify the chains on the device
// (list = 2 * nch * nchunks + 2 words of device memory owned by the caller) and run each class on its instance
// list_next: the list the NEXT classified launch of this handle will use; its two counters are zeroed by this launch's classification (no memset per launch)
hipError_t pv_launch_wave(const PvKernelParams &p, int nch, int nchunks, hipStream_t st, int spread, unsigned *list, unsigned *list_next = nullptr);
// resident form of the same kernel for streaming quanta (p.ctl != null): one wave per channel slot, nslots of them, polling p.ctl until ctl[4] (stop) or ~50 ms idle
hipError_t pv_launch_wave_resident(const PvKernelParams &p, int nslots, hipStream_t st);
hipError_t pv_launch_wave2k_resident(const PvKernelParams &p, int nslots, hipStream_t st);
bool pv_wg_resident_supported(int log2n, int hop, bool wg8);
hipError_t pv_launch_wg_resident(int log2n, const PvKernelParams &p, int nslots, hipStream_t st, bool wg8);

// one wavefront per 2048-point frame (pv_wave2k_kernel.hip): N = 2048, hop in {128, 256, 512, 1024, 2048}, every pitchFactor
bool pv_wave2k_supported(int log2n, int hop);
size_t pv_wave2k_lds_bytes();
int pv_wave2k_threads();
hipError_t pv_launch_wave2k(const PvKernelParams &p, int nch, int nchunks, hipStream_t st);

// register-resident workgroup kernel for N = 2048..8192, hop in {N/8, N/4, N/2, N} (pv_wg_kernel.hip)
bool pv_wg_supported(int log2n, int hop);
size_t pv_wg_lds_bytes(int log2n, int hop, bool wg8);
int pv_wg_threads(int log2n, int hop, bool wg8);
hipError_t pv_launch_wg(int log2n, const PvKernelParams &p, int nch, int nchunks, hipStream_t st, bool wg8);   // wg8: the eight-element kernel also where pv_wg16_kernel exists

// sixteen elements per thread, N / 32 threads: N = 8192 (four waves, two workgroups per CU) and N = 4096 (two waves, four workgroups per CU), hop in {N/8, N/4, N/2, N}
// (pv_wg16_kernel.hip); dispatched by pv_launch_wg / pv_launch_wg_resident
bool pv_wg16_supported(int log2n, int hop);
size_t pv_wg16_lds_bytes(int log2n);
int pv_wg16_threads(int log2n);
hipError_t pv_launch_wg16(int log2n, const PvKernelParams &p, int nch, int nchunks, hipStream_t st);
hipError_t pv_launch_wg16_resident(int log2n, const PvKernelParams &p, int nslots, hipStream_t st);

