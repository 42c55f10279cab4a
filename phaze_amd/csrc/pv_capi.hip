// pv_capi.hip -- host side of the C ABI declared in include/phaze_amd.h.
//
// Owns the device state of one processor instance (what OLAProcessor / PhaseVocoderProcessor keep in
// typed arrays: ola-processor.js:20-33,54-88 and phase-vocoder.js:30-42) and turns process() calls into
// launches of the chain kernel (pv_kernels.hip).  No CPU compute path exists here: without a HIP device
// pv_create fails with PV_ERR_DEVICE.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <vector>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

#include "../../include/phaze_amd.h"
#include "pv_kernels.h"

namespace {
constexpr uint32_t kMagic = 0x50564d49u;   // 'PVMI'
constexpr int kHdrFloats = 16;             // pinned staging header: [0] = pitchFactor
constexpr int kMaxPieces = 16;             // pieces a pipelined host-buffer batch is cut into at most
constexpr int kFwdStatWords = 1024;        // forward-transform statistics: [0, 256) 128 x {frames, fallbacks} (pv_forward_stats); the rest belongs to the validation
                                           // build -DPV_FLIP_COUNT: [256, 512) 128 x {frames whose flags differ, of those not caught by the guard}, [512] largest q
thread_local char g_create_err[256] = "";
}  // namespace

struct pv_handle {
    uint32_t magic;
    int N, hop, L, R, log2n;
    int max_channels, max_hops, device, cus;
    int frames_per_chunk_cfg;
    int last_frames_per_chunk;
    hipStream_t own_stream, stream;
    double2 *d_tw64;
    float2 *d_tw32;
    float *d_hann;
    float *d_hist[2], *d_acc[2];
    int cur;
    float *d_stage_in, *d_stage_out, *d_pitch;   // host-buffer batch staging
    unsigned *d_chain_list;                      // N = 1024 batch launches: chain classes (pv_launch_wave), 2 + 2 * chain_list_cap words
    long chain_list_cap;
    int chain_list_flip = 0;                     // which of the two lists the next launch fills (the other one's counters are zeroed by that launch's classification)
    unsigned char *d_gscratch;                   // N >= 16384: the generic kernel's per-workgroup scratch in device memory (pv_kernel_gscratch_bytes), grown on demand
    size_t gscratch_cap;
    float *d_snap;                               // pipelined host-buffer batch cut into spans of hops: copy of the live half of the channel state (hist | acc), taken
    size_t snap_floats;                          //   before the first piece so that a failure in a later piece can put the handle back (allocated on first use)
    unsigned idle_ticks;                         // resident kernels: constant-rate clock ticks (wall_clock64) after which waves without work leave (~50 ms)
    bool fwd64;                                  // PV_FLAG_FP64_FORWARD: every frame's forward transform in fp64 (the round-4 kernels)
    unsigned long long *d_fwd_stats;             // 128 x {frames computed by an fp32-first instance, frames of those that fell back to fp64} (pv_forward_stats)
    hipStream_t s_in, s_out;                     // pipelined host-buffer batch: H2D of piece k+1 || kernel of piece k || D2H of piece k-1 (created on first use)
    hipEvent_t ev_in[kMaxPieces], ev_k[kMaxPieces];
    bool pipe_ready;
    float *h_pin;                                // pinned: [hdr | max_channels*hop in | max_channels*hop out]
    float *d_quantum;                            // device twin of h_pin
    float *d_pin_mapped;                         // device view of h_pin (zero-copy streaming quantum); null = stage through d_quantum
    bool bar_input;                              // large-BAR device: the host writes small quanta straight into d_quantum (PV_FLAG_STREAM_PINNED_INPUT: off)
    volatile unsigned *hdp_flush;                // HDP_MEM_COHERENCY_FLUSH_CNTL of the device, mapped by the runtime (null: not exposed): written after host stores through the BAR
    volatile unsigned *h_done;                   // pinned: completion word of every frame chain of a streaming quantum (PvKernelParams::done)
    unsigned *d_done;                            // device view of h_done; null = wait through hipStreamSynchronize
    unsigned quantum_seq;                        // sequence number the chains of the pending quantum store (never 0)
    volatile unsigned *h_ctl;                    // pinned control block of the resident streaming kernel {seq, nch, t0 mod N, cur, stop, slots in use}; null = not used
    unsigned *d_ctl;
    bool resident_bar;                           // the resident kernel's control block lives in DEVICE memory, written by the host through the BAR
    bool resident_in_bar;                        // ... and so does its input (largest quantum <= 16 KB; otherwise the waves read it from pinned host memory)
    bool resident_wg;                            // the resident kernel is pv_wg_kernel: one control word per channel slot (ctl[16 + c]), channels handed over one by one
    bool copied[64];                             // pv_process_end: channels whose output has been copied out already
    bool resident_on;                            // a resident kernel has been launched on the stream and not been stopped since
    std::chrono::steady_clock::time_point last_quantum;   // ... and when it was last given work (its waves leave after ~50 ms without)
    double *d_dbgX; float *d_dbgMag; int *d_dbgFlags; float *d_dbgY;
#ifdef PV_STAMPS
    unsigned *d_stamps;                          // measurement builds only: [chain][16] phase clocks of the wave kernel
#endif
    int64_t time_cursor;
    int active_nch;
    int used_channels;                           // channel slots [used_channels, max_channels) have not been processed since they were last zeroed:
                                                 // they are zero in BOTH ping-pong halves and need no copy across a flip
    int pending_nch;                             // > 0: a quantum launched by pv_process_begin waits for pv_process_end
    int pending_cur, pending_active_nch;         // ... and what to restore if that wait fails
    int64_t pending_time_cursor;
    bool host_channels;                          // PV_FLAG_HOST_CHANNEL_BOOKKEEPING: a changed nch resets nothing here
    bool test_fail_piece = false;      // PV_FLAG_TEST_FAIL_SECOND_PIECE: the second piece of a pipelined host-buffer batch reports a device error (exercises the roll-back)
    bool use_wave;                               // N = 1024: wave-per-frame kernel (pv_wave_kernel.hip)
    bool use_wg;                                 // N = 2048..8192, R <= 8: register-resident workgroup kernel (pv_wg_kernel.hip)
    bool use_wave2k;                             // N = 2048, hop 128..2048: one wave per frame (pv_wave2k_kernel.hip)
    bool use_wg16;                               // N = 4096 / 8192, hop N/8..N: sixteen elements per thread, N/32 threads per frame chain (pv_wg16_kernel.hip); implies use_wg
    char devname[64];
    char err[256];
};

namespace {

// What run_chain() commits (ping-pong half of the channel state, timeCursor, channel count).  The synchronous entry points take a snapshot
// first and restore it when a later step -- the stream sync, a copy -- fails, so that a failed call leaves the handle as it was
// (pv_get_time_cursor and the next output unchanged); the asynchronous pv_process_batch_device cannot know, its errors surface in pv_synchronize.
struct Commit { int cur; int64_t time_cursor; int active_nch; };

int fail(pv_handle *h, int code, const char *msg)
{
    if (h) snprintf(h->err, sizeof h->err, "%s", msg);
    else snprintf(g_create_err, sizeof g_create_err, "%s", msg);
    return code;
}

int fail_hip(pv_handle *h, hipError_t e, const char *what)
{
    char buf[256];
    snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
    return fail(h, PV_ERR_DEVICE, buf);
}

#define HIPCHK(h, call)                                            \
    do {                                                           \
        hipError_t e_ = (call);                                    \
        if (e_ != hipSuccess) return fail_hip((h), e_, #call);     \
    } while (0)

bool live(const pv_handle *h) { return h && h->magic == kMagic; }

int pick_frames_per_chunk(const pv_handle *h, int nch, int nhops)
{
    if (h->frames_per_chunk_cfg > 0) return h->frames_per_chunk_cfg;
    // Trade-off: long chains amortise the (R-1)-frame halo, but the last partial round of workgroups idles the chip.
    // resident = chains the GPU runs concurrently (wave kernels: one per wave; others: LDS-limited workgroups per CU).
    long per_cu;
    if (h->use_wave2k) per_cu = pv_wave2k_threads() / 64;
    else if (h->use_wave) per_cu = pv_wave_threads() / 64;
    else if (h->use_wg) { per_cu = (160 * 1024) / (long)(pv_wg_lds_bytes(h->log2n, h->hop, !h->use_wg16) + 256); if (per_cu < 1) per_cu = 1; }   // (+ 256 static bytes: __syncthreads_or)
    else { per_cu = (160 * 1024) / (long)pv_kernel_lds_bytes(h->log2n, h->hop); if (per_cu > 8) per_cu = 8; if (per_cu < 1) per_cu = 1; }
    const long resident = per_cu * h->cus;
    const int R = h->R;
    int best = R > 1 ? R : 1;
    double best_eff = -1.0;
    auto consider = [&](int F) {
        if (F < 1) return;
        const long chunks = (nhops + F - 1) / F;
        const long chains = (long)nch * chunks;
        const long rounds = (chains + resident - 1) / resident;
        const double fill = (double)chains / (double)(rounds * resident);                        // tail effect
        const double halo = (double)nhops / (double)(nhops + (chunks - 1) * (R - 1));            // recomputed frames: none for a channel's first chunk
        const double eff = fill * halo;
        if (eff > best_eff + 1e-9) { best_eff = eff; best = F; }
    };
    consider(nhops);                                                                             // one chain per channel: no halo at all
    // chain lengths that fill exactly r rounds of the resident chains (r = 1: every CU starts once, the halo is the smallest a full chip allows)
    for (long r = 1; r <= 8; r++) {
        const long chunks = (r * resident) / (nch > 0 ? nch : 1);
        if (chunks >= 1) { consider((int)((nhops + chunks - 1) / chunks)); consider((int)((nhops + chunks - 1) / chunks) + 1); }
    }
    for (int F = 48 * R; F >= (R > 1 ? R : 1); F -= (F > 8 * R ? R : 1)) {
        if (F > nhops && F > R) continue;
        consider(F);
    }
    if (best > nhops) best = nhops;
    return best < 1 ? 1 : best;
}

// One launch over the channel slots [ch0, ch0 + nch) x [nhops] hops, reading the current half of the state ping-pong and writing the other one.
// d_in / d_out / d_pitch are the pointers of slot ch0 (the caller has applied the offsets); ch0 only places the state.  Nothing is committed.
// spread: what the HOST knows about the pitchFactors of the launch (> 0: all >= 1, 0: not all, < 0: nothing -- they live in device memory)
int launch_chain(pv_handle *h, const float *d_in, float *d_out, int ch0, int nch, int nhops, long ch_stride, const float *d_pitch,
                 int pitch_stride, int ch_per_stream, bool chunked, int dbg_ch, unsigned done_seq, int spread = -1)
{
    PvKernelParams p;
    memset(&p, 0, sizeof p);
    p.in = d_in; p.out = d_out; p.ch_stride = ch_stride;
    p.nhops = nhops; p.hop = h->hop;
    p.frames_per_chunk = chunked ? pick_frames_per_chunk(h, nch, nhops) : nhops;
    p.pitch = d_pitch; p.pitch_stride = pitch_stride; p.ch_per_stream = ch_per_stream > 0 ? ch_per_stream : 1;
    const size_t soff = (size_t)ch0 * h->L;
    p.hist_in = h->d_hist[h->cur] + soff; p.hist_out = h->d_hist[h->cur ^ 1] + soff;
    p.acc_in = h->d_acc[h->cur] + soff;   p.acc_out = h->d_acc[h->cur ^ 1] + soff;
    p.t0_mod_n = (int)(h->time_cursor & (int64_t)(h->N - 1));
    p.tw64 = h->d_tw64; p.tw32 = h->d_tw32; p.hann = h->d_hann;
    p.dbg_ch = -1; p.dbg_frame = -1;
    p.fwd64 = h->fwd64 ? 1 : 0; p.fwd_stats = h->d_fwd_stats;
    if (done_seq) { p.done = h->d_done; p.done_seq = done_seq; }
#ifdef PV_STAMPS
    p.stamps = h->d_stamps;
#endif
    if (dbg_ch >= 0) { p.dbg_X = h->d_dbgX; p.dbg_mag = h->d_dbgMag; p.dbg_flags = h->d_dbgFlags; p.dbg_Y = h->d_dbgY; p.dbg_ch = dbg_ch; p.dbg_frame = 0; }
    const int nchunks = (nhops + p.frames_per_chunk - 1) / p.frames_per_chunk;
    h->last_frames_per_chunk = p.frames_per_chunk;
    hipError_t e = hipSuccess;
    if (h->use_wave2k) {                                         // (pv_debug_frame runs the tap instance of the kernel the handle uses)
        e = pv_launch_wave2k(p, nch, nchunks, h->stream);
    } else {
        if (!h->use_wave && nch > 65535) return fail(h, PV_ERR_CAPACITY, "more than 65535 channel slots in one launch (grid.y limit): split the call");
        unsigned *list = nullptr, *list_next = nullptr;
        if (h->use_wave && spread < 0 && dbg_ch < 0) {
            // chain classes are sorted on the device: room for two lists of nch * nchunks chains (grown on demand; a launch in flight may still read the old one)
            const long chains = (long)nch * nchunks;
            if (chains > h->chain_list_cap) {
                HIPCHK(h, hipStreamSynchronize(h->stream));
                if (h->d_chain_list) (void)hipFree(h->d_chain_list);
                h->d_chain_list = nullptr; h->chain_list_cap = 0;
                const long cap = chains + chains / 2 + 64;
                // TWO lists, used alternately: the classification of launch k zeroes the counters launch k + 1 will fill (the kernels of launch k - 1, which read them, are
                // behind it in stream order), so no memset sits in front of every launch (round 6: 4.6 us of a 1.7 ms step)
                HIPCHK(h, hipMalloc(&h->d_chain_list, 2 * sizeof(unsigned) * (size_t)(2 + 2 * cap)));
                HIPCHK(h, hipMemsetAsync(h->d_chain_list, 0, 2 * sizeof(unsigned) * (size_t)(2 + 2 * cap), h->stream));
                h->chain_list_cap = cap;
                h->chain_list_flip = 0;
            }
            list = h->d_chain_list + (size_t)h->chain_list_flip * (size_t)(2 + 2 * h->chain_list_cap);
            list_next = h->d_chain_list + (size_t)(h->chain_list_flip ^ 1) * (size_t)(2 + 2 * h->chain_list_cap);
            h->chain_list_flip ^= 1;
        }
        if (!h->use_wave && !h->use_wg) {
            // N >= 16384: the generic kernel keeps its fp32 buffer and its overlap-add ring (N = 32768: its fp64 buffer too) in device memory, one slice per workgroup
            const size_t stride = pv_kernel_gscratch_bytes(h->log2n, h->hop);
            if (stride) {
                const size_t need = stride * (size_t)nch * (size_t)nchunks;
                if (need > ((size_t)16 << 30)) return fail(h, PV_ERR_CAPACITY, "fft_size >= 16384: the launch needs more than 16 GiB of scratch (fewer channels or hops per call)");
                if (need > h->gscratch_cap) {
                    HIPCHK(h, hipStreamSynchronize(h->stream));
                    if (h->d_gscratch) (void)hipFree(h->d_gscratch);
                    h->d_gscratch = nullptr; h->gscratch_cap = 0;
                    HIPCHK(h, hipMalloc(&h->d_gscratch, need));
                    h->gscratch_cap = need;
                }
                p.gscratch = h->d_gscratch; p.gscratch_stride = stride;
            }
        }
        e = h->use_wave ? pv_launch_wave(p, nch, nchunks, h->stream, spread, list, list_next)
          : h->use_wg ? pv_launch_wg(h->log2n, p, nch, nchunks, h->stream, !h->use_wg16)
                      : pv_launch_chain(h->log2n, p, nch, nchunks, h->stream);
    }
    if (e != hipSuccess) return fail_hip(h, e, "kernel launch");
    return PV_OK;
}

// What a completed pass over the slots [0, nch) x [nhops] hops commits: the ping-pong flip (slots outside the pass keep their state: copied
// across), timeCursor.
int commit_chain(pv_handle *h, int nch, int nhops)
{
    if (nch > h->used_channels) h->used_channels = nch;
    if (nch < h->used_channels && h->L > 0) {
        const size_t off = (size_t)nch * h->L, cnt = (size_t)(h->used_channels - nch) * h->L * sizeof(float);
        HIPCHK(h, hipMemcpyAsync(h->d_hist[h->cur ^ 1] + off, h->d_hist[h->cur] + off, cnt, hipMemcpyDeviceToDevice, h->stream));
        HIPCHK(h, hipMemcpyAsync(h->d_acc[h->cur ^ 1] + off, h->d_acc[h->cur] + off, cnt, hipMemcpyDeviceToDevice, h->stream));
    }
    h->cur ^= 1;
    h->time_cursor += (int64_t)nhops * h->hop;
    return PV_OK;
}

// One launch over [nch] channel slots x [nhops] hops; flips the state ping-pong and advances timeCursor.
int run_chain(pv_handle *h, const float *d_in, float *d_out, int nch, int nhops, long ch_stride, const float *d_pitch,
              int pitch_stride, int ch_per_stream, bool commit, int dbg_ch, unsigned done_seq = 0, int spread = -1)
{
    const int rc = launch_chain(h, d_in, d_out, 0, nch, nhops, ch_stride, d_pitch, pitch_stride, ch_per_stream, commit, dbg_ch, done_seq, spread);
    if (rc != PV_OK || !commit) return rc;
    return commit_chain(h, nch, nhops);
}

// Is this host pointer page-locked memory the HIP runtime knows (pv_host_alloc, hipHostMalloc, hipHostRegister)?  Only such memory can be the
// end point of an asynchronous copy; pageable memory is staged by the runtime synchronously.
bool host_pinned(const void *ptr)
{
    hipPointerAttribute_t a;
    memset(&a, 0, sizeof a);
    if (hipPointerGetAttributes(&a, ptr) != hipSuccess) { (void)hipGetLastError(); return false; }
    return a.type == hipMemoryTypeHost;
}
// ... the whole range [ptr, ptr + bytes): a buffer that is only partly registered (hipHostRegister on a sub-range, a view that runs past its registration, two
// registrations with a pageable gap between them) must not take the asynchronous DMA path.  Both ends must be page-locked AND belong to the SAME allocation / registration
// (hipMemGetAddressRange on the device-side alias of the two ends: same base); where the runtime cannot name the allocation, the span is probed at up to 2048 evenly
// spaced points in between (every page of a span of up to 8 MB; coarser above -- a gap smaller than the probe stride inside two separately registered regions is then
// not seen, which costs such a caller a synchronous staging copy inside the runtime, never a wrong result).
bool host_pinned_range(const void *ptr, size_t bytes)
{
    if (bytes == 0) return host_pinned(ptr);
    const char *lo = static_cast<const char *>(ptr), *hi = lo + bytes - 1;
    hipPointerAttribute_t a0, a1;
    memset(&a0, 0, sizeof a0); memset(&a1, 0, sizeof a1);
    if (hipPointerGetAttributes(&a0, lo) != hipSuccess || hipPointerGetAttributes(&a1, hi) != hipSuccess) { (void)hipGetLastError(); return false; }
    if (a0.type != hipMemoryTypeHost || a1.type != hipMemoryTypeHost) return false;
    if (a0.devicePointer && a1.devicePointer) {
        hipDeviceptr_t b0 = nullptr, b1 = nullptr;
        size_t s0 = 0, s1 = 0;
        if (hipMemGetAddressRange(&b0, &s0, a0.devicePointer) == hipSuccess && hipMemGetAddressRange(&b1, &s1, a1.devicePointer) == hipSuccess)
            return b0 == b1 && s0 == s1;                                     // one allocation / one registration covers both ends, hence everything between them
        (void)hipGetLastError();
    }
    const size_t page = 4096, pages = (bytes + page - 1) / page;
    const size_t probes = pages < 2048 ? pages : 2048;
    for (size_t i = 1; i + 1 < probes; i++)
        if (!host_pinned(lo + (size_t)((double)i / (double)(probes - 1) * (double)(bytes - 1)))) return false;
    return true;
}

// ---- resident streaming kernel (PV_FLAG_PERSISTENT_STREAM) ----
// One word handed to the resident waves: everything written before it (input, pitchFactor, parameters) is in memory first, and the word itself leaves
// the write-combining buffer and the device's host data path now, not when something else happens to drain them (the control block may live in
// DEVICE memory behind the BAR).  Used for the sequence words of a quantum AND for the stop word.
void resident_publish(pv_handle *h, volatile unsigned *word, unsigned value)
{
    std::atomic_thread_fence(std::memory_order_seq_cst);
#if defined(__x86_64__)
    _mm_sfence();
#endif
    *word = value;
#if defined(__x86_64__)
    _mm_sfence();
#endif
    if (h->resident_bar && h->hdp_flush) *h->hdp_flush = 1u;
}

// Asks the resident waves to leave and waits for them: required before anything else is put on the handle's stream (it would queue behind them).
int resident_stop(pv_handle *h)
{
    if (!h->resident_on) return PV_OK;
    resident_publish(h, h->h_ctl + 4, 1u);
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    resident_publish(h, h->h_ctl + 4, 0u);
    h->resident_on = false;
    return PV_OK;
}

// (Re)starts the resident waves: `last_seq` is the last sequence number that has been completed (they react to the next one).
int resident_start(pv_handle *h, unsigned last_seq)
{
    PvKernelParams p;
    memset(&p, 0, sizeof p);
    const int hop = h->hop;
    float *stage = h->resident_in_bar ? h->d_quantum : h->d_pin_mapped;      // where the host puts pitchFactor + input (see pv_process_begin)
    p.in = stage + kHdrFloats; p.out = h->d_pin_mapped + kHdrFloats + (size_t)h->max_channels * hop; p.ch_stride = hop;
    p.nhops = 1; p.hop = hop; p.frames_per_chunk = 1;
    p.pitch = stage; p.pitch_stride = 0; p.ch_per_stream = 1;
    p.hist_in = h->d_hist[0]; p.hist_out = h->d_hist[1]; p.acc_in = h->d_acc[0]; p.acc_out = h->d_acc[1];
    for (int i = 0; i < 2; i++) { p.hist2[i] = h->d_hist[i]; p.acc2[i] = h->d_acc[i]; }
    p.tw64 = h->d_tw64; p.tw32 = h->d_tw32; p.hann = h->d_hann;
    p.dbg_ch = -1; p.dbg_frame = -1;
    p.done = h->d_done; p.done_seq = last_seq;
    p.ctl = h->d_ctl;
    p.idle_ticks = h->idle_ticks;
    p.fwd64 = h->fwd64 ? 1 : 0; p.fwd_stats = h->d_fwd_stats;
    p.in_cached = h->resident_in_bar ? 1 : 0;
    resident_publish(h, h->h_ctl + 4, 0u);
    // stale completion words must not match a future 16-bit sequence number (a slot unused for exactly 65535 quanta)
    for (int c = 0; c < h->max_channels; c++) h->h_done[c] = 0u;
    const hipError_t e = h->resident_wg ? pv_launch_wg_resident(h->log2n, p, h->max_channels, h->stream, !h->use_wg16)
                       : h->use_wave2k ? pv_launch_wave2k_resident(p, h->max_channels, h->stream) : pv_launch_wave_resident(p, h->max_channels, h->stream);
    if (e != hipSuccess) return fail_hip(h, e, "resident kernel launch");
    h->resident_on = true;
    h->last_frames_per_chunk = 1;
    return PV_OK;
}

}  // namespace

extern "C" {

const char *pv_status_string(int status)
{
    switch (status) {
    case PV_OK: return "ok";
    case PV_ERR_FFT_SIZE: return "FFT size must be a power of two and bigger than 1";
    case PV_ERR_ARGUMENT: return "invalid argument";
    case PV_ERR_UNSUPPORTED: return "configuration outside the supported kernel range";
    case PV_ERR_CAPACITY: return "channel or hop count exceeds the handle's capacity";
    case PV_ERR_DEVICE: return "HIP device error";
    case PV_ERR_DESTROYED: return "handle destroyed";
    default: return "unknown status";
    }
}

const char *pv_last_error(const pv_handle *h) { return live(h) ? h->err : g_create_err; }

int pv_abi_version(void) { return PV_ABI_VERSION; }

int pv_device_count(int32_t *out)
{
    if (!out) return PV_ERR_ARGUMENT;
    int n = 0;
    const hipError_t e = hipGetDeviceCount(&n);
    (void)hipGetLastError();
    *out = (e == hipSuccess && n > 0) ? n : 0;
    return PV_OK;
}

int pv_create(const pv_config *cfg, pv_handle **out)
{
    if (!cfg || !out) return fail(nullptr, PV_ERR_ARGUMENT, "pv_create: null argument");
    *out = nullptr;
    if (cfg->struct_size != (int32_t)sizeof(pv_config))
        return fail(nullptr, PV_ERR_ARGUMENT, "pv_create: pv_config.struct_size does not match this library (start from PV_CONFIG_INIT; PV_ABI_VERSION mismatch?)");
    if (cfg->flags & ~(int32_t)PV_FLAG_ALL) return fail(nullptr, PV_ERR_ARGUMENT, "pv_create: unknown bits in pv_config.flags");
    const int N = cfg->fft_size;                                 // the host passes 2048 for the reference default (phase-vocoder.js:6)
    const int hop = cfg->hop_size;                               // ... and 128 (ola-processor.js:3)
    if (N <= 1 || (N & (N - 1)) != 0)                            // bundle:6-7
        return fail(nullptr, PV_ERR_FFT_SIZE, "FFT size must be a power of two and bigger than 1");
    if (hop <= 0 || N % hop != 0) return fail(nullptr, PV_ERR_ARGUMENT, "hop_size must be positive and divide fft_size");
    int log2n = 0;
    while ((1 << log2n) < N) log2n++;
    if (log2n > 20) return fail(nullptr, PV_ERR_UNSUPPORTED, "fft_size must be within 2..1048576 for the gfx950 kernels");
    if (hop < 2) return fail(nullptr, PV_ERR_UNSUPPORTED, "hop_size must be >= 2");
    {
        const bool generic = (cfg->flags & PV_FLAG_GENERIC_KERNEL) != 0;
        const bool reg_kernel = !generic && (pv_wave_supported(log2n, hop) || pv_wg_supported(log2n, hop));
        if (!reg_kernel && pv_kernel_lds_bytes(log2n, hop) > 160 * 1024 - 512)
            return fail(nullptr, PV_ERR_UNSUPPORTED, "fft_size/hop_size combination exceeds the 160 KiB LDS of a CU");
    }
    const int maxch = cfg->max_channels > 0 ? cfg->max_channels : 2;
    const int maxhops = cfg->max_hops > 0 ? cfg->max_hops : 1;

    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) return fail(nullptr, PV_ERR_DEVICE, "no HIP device available (this library has no CPU path)");
    if (cfg->device_id < 0 || cfg->device_id >= ndev) return fail(nullptr, PV_ERR_ARGUMENT, "device_id out of range");

    pv_handle *h = (pv_handle *)calloc(1, sizeof(pv_handle));
    if (!h) return fail(nullptr, PV_ERR_DEVICE, "pv_create: out of host memory");
    h->magic = kMagic;
    h->N = N; h->hop = hop; h->L = N - hop; h->R = N / hop; h->log2n = log2n;
    h->max_channels = maxch; h->max_hops = maxhops; h->device = cfg->device_id;
    h->frames_per_chunk_cfg = cfg->frames_per_chunk;
    h->active_nch = -1;
    h->host_channels = (cfg->flags & PV_FLAG_HOST_CHANNEL_BOOKKEEPING) != 0;
    h->fwd64 = (cfg->flags & PV_FLAG_FP64_FORWARD) != 0;
    h->test_fail_piece = (cfg->flags & PV_FLAG_TEST_FAIL_SECOND_PIECE) != 0;
    {
        // the resident waves leave after ~50 ms without work, measured on the device's constant-rate clock (wall_clock64): the host stops and restarts them by ITS
        // wall clock after 20 ms (pv_process_begin), so the device-side figure must never come out below that -- a poll COUNT did, depending on where the control block lives
        int khz = 0;
        if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, cfg->device_id) != hipSuccess || khz <= 0) { (void)hipGetLastError(); khz = 100000; }
        h->idle_ticks = (unsigned)khz * 50u;
    }
    {
        const bool generic = (cfg->flags & PV_FLAG_GENERIC_KERNEL) != 0;      // explicit A/B switch (tests, measurements); no environment is read
        h->use_wave = pv_wave_supported(log2n, hop) && !generic;
        h->use_wg = pv_wg_supported(log2n, hop) && !generic;
        const bool wg_only = (cfg->flags & PV_FLAG_WORKGROUP_KERNEL) != 0;    // A/B: the eight-element workgroup kernel where a one-wave / sixteen-element kernel exists
        h->use_wg16 = h->use_wg && !wg_only && pv_wg16_supported(log2n, hop);
        h->use_wave2k = h->use_wg && !wg_only && !h->use_wg16 && pv_wave2k_supported(log2n, hop);   // (the two never meet in the product; the reference-width flavour's sixteen-element kernel also takes N = 2048)
    }

#define CHK(call)                                                          \
    do {                                                                   \
        hipError_t e2_ = (call);                                           \
        if (e2_ != hipSuccess) {                                           \
            int rc_ = fail_hip(nullptr, e2_, #call);                       \
            pv_destroy(h);                                                 \
            return rc_;                                                    \
        }                                                                  \
    } while (0)
    CHK(hipSetDevice(h->device));
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, h->device));
    h->cus = prop.multiProcessorCount;
    snprintf(h->devname, sizeof h->devname, "%s (%s)", prop.name[0] ? prop.name : "AMD GPU", prop.gcnArchName);
    CHK(hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking));
    h->stream = h->own_stream;

    // tables: twiddles exp(-2 pi j k/N) (role of bundle:12-18) and the periodic Hann window (pv:8-14), fp64 on host
    std::vector<double2> tw64(N);
    std::vector<float2> tw32(N);
    std::vector<float> hann(N);
    for (int k = 0; k < N; k++) {
        const double ang = 2.0 * M_PI * (double)k / (double)N;
        tw64[k] = double2{cos(ang), -sin(ang)};
        tw32[k] = float2{(float)tw64[k].x, (float)tw64[k].y};
        hann[k] = (float)(0.5 * (1.0 - cos(ang)));
    }
    // exact values on the axes (libm returns ~1e-16 residues)
    tw64[0] = double2{1, 0}; tw32[0] = float2{1, 0};
    if (N >= 4) { tw64[N / 4] = double2{0, -1}; tw32[N / 4] = float2{0, -1}; tw64[3 * N / 4] = double2{0, 1}; tw32[3 * N / 4] = float2{0, 1}; }
    tw64[N / 2] = double2{-1, 0}; tw32[N / 2] = float2{-1, 0};
    CHK(hipMalloc(&h->d_tw64, sizeof(double2) * N));
    CHK(hipMalloc(&h->d_tw32, sizeof(float2) * N));
    CHK(hipMalloc(&h->d_hann, sizeof(float) * 2 * N));                       // [0, N): the window; [N, 2N): half of it (exact), for kernels that fold the 1/2 of the split pass into it
    CHK(hipMemcpy(h->d_tw64, tw64.data(), sizeof(double2) * N, hipMemcpyHostToDevice));
    CHK(hipMemcpy(h->d_tw32, tw32.data(), sizeof(float2) * N, hipMemcpyHostToDevice));
    CHK(hipMemcpy(h->d_hann, hann.data(), sizeof(float) * N, hipMemcpyHostToDevice));
    for (int k = 0; k < N; k++) hann[k] *= 0.5f;
    CHK(hipMemcpy(h->d_hann + N, hann.data(), sizeof(float) * N, hipMemcpyHostToDevice));

    CHK(hipMalloc(&h->d_fwd_stats, sizeof(unsigned long long) * kFwdStatWords));
    CHK(hipMemset(h->d_fwd_stats, 0, sizeof(unsigned long long) * kFwdStatWords));
    const size_t state = sizeof(float) * (size_t)maxch * (size_t)(h->L > 0 ? h->L : 1);
    for (int i = 0; i < 2; i++) {
        CHK(hipMalloc(&h->d_hist[i], state));
        CHK(hipMalloc(&h->d_acc[i], state));
        CHK(hipMemset(h->d_hist[i], 0, state));
        CHK(hipMemset(h->d_acc[i], 0, state));
    }
    const size_t stage = sizeof(float) * (size_t)maxch * (size_t)maxhops * (size_t)hop;
    CHK(hipMalloc(&h->d_stage_in, stage));
    CHK(hipMalloc(&h->d_stage_out, stage));
    CHK(hipMalloc(&h->d_pitch, sizeof(float) * (size_t)maxch * (size_t)maxhops));
    const size_t quantum = sizeof(float) * (kHdrFloats + 2 * (size_t)maxch * hop);
    CHK(hipHostMalloc((void **)&h->h_pin, quantum, hipHostMallocMapped));
    {
        // streaming quantum: the kernel reads the hop straight from / writes it straight to pinned host memory (one launch + one sync,
        // no copy nodes).  PV_FLAG_STREAM_COPY restores H2D + kernel + D2H staging.
        void *dp = nullptr;
        if (!(cfg->flags & PV_FLAG_STREAM_COPY) && hipHostGetDevicePointer(&dp, h->h_pin, 0) == hipSuccess) h->d_pin_mapped = (float *)dp;
        (void)hipGetLastError();
    }
    CHK(hipMalloc(&h->d_quantum, quantum));
    {
        int large_bar = 0;
        if (hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, h->device) != hipSuccess) { large_bar = 0; (void)hipGetLastError(); }
        h->bar_input = large_bar == 1 && h->d_pin_mapped != nullptr && !(cfg->flags & PV_FLAG_STREAM_PINNED_INPUT);
        if (h->bar_input) {
            // Host stores through the BAR pass the device's HDP block; the runtime maps its flush register for exactly this use.  The attribute
            // call stores a POINTER (8 bytes) through its int* argument -- that is how ROCm defines hipDeviceAttributeHdpMemFlushCntl; `reg` is a
            // full pointer-sized object, so nothing is written out of bounds.  One more posted write per quantum, behind the data in PCIe order.
            unsigned *reg = nullptr;
            const bool have = !(cfg->flags & PV_FLAG_TEST_NO_HDP_FLUSH) &&
                              hipDeviceGetAttribute(reinterpret_cast<int *>(&reg), hipDeviceAttributeHdpMemFlushCntl, h->device) == hipSuccess && reg != nullptr;
            (void)hipGetLastError();
            // Without the flush register a host store through the BAR may still sit in the HDP when the kernel reads: no BAR hand-over then, the quantum
            // is read from pinned host memory (the PV_FLAG_STREAM_PINNED_INPUT form) and the resident kernel keeps its control block there as well.
            if (have) h->hdp_flush = reg; else h->bar_input = false;
        }
    }
    if (h->d_pin_mapped && !(cfg->flags & PV_FLAG_STREAM_EVENT_WAIT)) {
        // completion words of the streaming quantum, one per channel slot, in pinned host memory the kernels store to
        unsigned *hd = nullptr;
        void *dd = nullptr;
        CHK(hipHostMalloc((void **)&hd, sizeof(unsigned) * (size_t)maxch, hipHostMallocMapped));
        memset(hd, 0, sizeof(unsigned) * (size_t)maxch);
        h->h_done = hd;
        if (hipHostGetDevicePointer(&dd, hd, 0) == hipSuccess) h->d_done = (unsigned *)dd;
        (void)hipGetLastError();
        if (h->d_done && (cfg->flags & PV_FLAG_PERSISTENT_STREAM) && maxch <= 64 &&
            (h->use_wave || h->use_wave2k || (h->use_wg && pv_wg_resident_supported(log2n, hop, !h->use_wg16)))) {
            // Control block: in DEVICE memory when the host can write it through the BAR and the largest quantum is small enough to travel the same
            // way -- the waves then poll their own HBM and find the input there too, the only PCIe traffic of a quantum being posted writes in both
            // directions (tools/bar_probe.hip: 1 KB handed over and acknowledged in 3.5 us, 7.2 us with the block and the input in pinned host memory)
            h->resident_wg = !(h->use_wave || h->use_wave2k);
            h->resident_bar = h->bar_input;
            h->resident_in_bar = h->bar_input && sizeof(float) * (size_t)maxch * hop <= 16384;
            constexpr size_t kCtlBytes = sizeof(unsigned) * (16 + 64);       // {seq word, -, -, -, stop, slots in use, ...} + one word per channel slot (resident_wg)
            if (h->resident_bar) {
                void *dc = nullptr;
                CHK(hipMalloc(&dc, kCtlBytes));
                CHK(hipMemset(dc, 0, kCtlBytes));
                h->d_ctl = (unsigned *)dc;
                h->h_ctl = (volatile unsigned *)dc;                          // (written, never read, by the host)
            } else {
                unsigned *hc = nullptr;
                void *dc = nullptr;
                CHK(hipHostMalloc((void **)&hc, kCtlBytes, hipHostMallocMapped));
                memset(hc, 0, kCtlBytes);
                h->h_ctl = hc;
                if (hipHostGetDevicePointer(&dc, hc, 0) == hipSuccess) h->d_ctl = (unsigned *)dc; else h->h_ctl = nullptr;
                (void)hipGetLastError();
            }
        }
    }
    CHK(hipMalloc(&h->d_dbgX, sizeof(double) * 2 * N));
    CHK(hipMalloc(&h->d_dbgMag, sizeof(float) * (N / 2 + 1)));
    CHK(hipMalloc(&h->d_dbgFlags, sizeof(int) * (N / 2 + 1)));
    CHK(hipMalloc(&h->d_dbgY, sizeof(float) * 2 * (N / 2 + 1)));
#ifdef PV_STAMPS
    CHK(hipMalloc(&h->d_stamps, sizeof(unsigned) * 16 * 65536));
    CHK(hipMemset(h->d_stamps, 0, sizeof(unsigned) * 16 * 65536));
#endif
#undef CHK
    *out = h;
    return PV_OK;
}

int pv_destroy(pv_handle *h)
{
    if (!h) return PV_OK;
    if (h->magic != kMagic) return PV_ERR_DESTROYED;
    (void)hipSetDevice(h->device);
    (void)resident_stop(h);
    if (h->own_stream) (void)hipStreamSynchronize(h->own_stream);
    if (h->pipe_ready) {
        (void)hipStreamSynchronize(h->s_in); (void)hipStreamSynchronize(h->s_out);
        for (int i = 0; i < kMaxPieces; i++) { (void)hipEventDestroy(h->ev_in[i]); (void)hipEventDestroy(h->ev_k[i]); }
        (void)hipStreamDestroy(h->s_in); (void)hipStreamDestroy(h->s_out);
    }
    (void)hipFree(h->d_tw64); (void)hipFree(h->d_tw32); (void)hipFree(h->d_hann);
    for (int i = 0; i < 2; i++) { (void)hipFree(h->d_hist[i]); (void)hipFree(h->d_acc[i]); }
    (void)hipFree(h->d_stage_in); (void)hipFree(h->d_stage_out); (void)hipFree(h->d_pitch);
    if (h->d_chain_list) (void)hipFree(h->d_chain_list);
    if (h->d_gscratch) (void)hipFree(h->d_gscratch);
    (void)hipFree(h->d_fwd_stats);
    if (h->d_snap) (void)hipFree(h->d_snap);
    if (h->h_pin) (void)hipHostFree(h->h_pin);
    if (h->h_done) (void)hipHostFree((void *)h->h_done);
    if (h->h_ctl) { if (h->resident_bar) (void)hipFree((void *)h->h_ctl); else (void)hipHostFree((void *)h->h_ctl); }
    (void)hipFree(h->d_quantum);
    (void)hipFree(h->d_dbgX); (void)hipFree(h->d_dbgMag); (void)hipFree(h->d_dbgFlags); (void)hipFree(h->d_dbgY);
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    h->magic = 0;
    free(h);
    return PV_OK;
}

int pv_get_info(const pv_handle *h, pv_info *out)
{
    if (!live(h)) return PV_ERR_DESTROYED;
    if (!out) return PV_ERR_ARGUMENT;
    memset(out, 0, sizeof *out);
    out->fft_size = h->N; out->hop_size = h->hop; out->overlaps = h->R;
    out->max_channels = h->max_channels; out->max_hops = h->max_hops;
    out->threads_per_workgroup = h->use_wave ? pv_wave_threads() : h->use_wave2k ? pv_wave2k_threads() : h->use_wg ? pv_wg_threads(h->log2n, h->hop, !h->use_wg16) : pv_kernel_threads(h->log2n);
    snprintf(out->kernel_name, sizeof out->kernel_name, "%s",
             h->use_wave ? "pv_wave_kernel_1024" : h->use_wave2k ? "pv_wave2k_kernel" : h->use_wg16 ? "pv_wg16_kernel" : h->use_wg ? "pv_wg_kernel" : "pv_chain_kernel");
    out->lds_bytes_per_workgroup = (int32_t)(h->use_wave ? pv_wave_lds_bytes() : h->use_wave2k ? pv_wave2k_lds_bytes()
                                             : h->use_wg ? pv_wg_lds_bytes(h->log2n, h->hop, !h->use_wg16) : pv_kernel_lds_bytes(h->log2n, h->hop));
    out->frames_per_chunk = h->last_frames_per_chunk;
    out->compute_units = h->cus; out->device_id = h->device;
    snprintf(out->device_name, sizeof out->device_name, "%s", h->devname);
    return PV_OK;
}

int pv_reset_channels_part(pv_handle *h, int32_t first, int32_t count, int32_t parts)
{
    if (!live(h)) return PV_ERR_DESTROYED;
    { const int rs_ = resident_stop(h); if (rs_ != PV_OK) return rs_; }
    if (first < 0 || count < 0 || first + count > h->max_channels) return fail(h, PV_ERR_ARGUMENT, "pv_reset_channels: range out of bounds");
    if (parts <= 0 || (parts & ~(PV_STATE_HISTORY | PV_STATE_ACCUMULATOR))) return fail(h, PV_ERR_ARGUMENT, "pv_reset_channels_part: parts must be PV_STATE_HISTORY and / or PV_STATE_ACCUMULATOR");
    if (count == 0 || h->L == 0) return PV_OK;
    HIPCHK(h, hipSetDevice(h->device));
    const size_t off = (size_t)first * h->L, bytes = sizeof(float) * (size_t)count * h->L;
    for (int i = 0; i < 2; i++) {                                  // both ping-pong halves: a zeroed slot stays zero across flips without a copy
        if (parts & PV_STATE_HISTORY) HIPCHK(h, hipMemsetAsync(h->d_hist[i] + off, 0, bytes, h->stream));
        if (parts & PV_STATE_ACCUMULATOR) HIPCHK(h, hipMemsetAsync(h->d_acc[i] + off, 0, bytes, h->stream));
    }
    // (a slot is only "unused" -- zero in both halves, no copy across a flip -- when BOTH sides were zeroed)
    if (parts == (PV_STATE_HISTORY | PV_STATE_ACCUMULATOR) && first + count >= h->used_channels && first < h->used_channels) h->used_channels = first;
    return PV_OK;
}

int pv_reset_channels(pv_handle *h, int32_t first, int32_t count)
{
    return pv_reset_channels_part(h, first, count, PV_STATE_HISTORY | PV_STATE_ACCUMULATOR);
}

int pv_reset(pv_handle *h)
{
    if (!live(h)) return PV_ERR_DESTROYED;
    const int rc = pv_reset_channels(h, 0, h->max_channels);
    if (rc != PV_OK) return rc;
    h->time_cursor = 0;
    h->active_nch = -1;
    h->pending_nch = 0;                                          // a quantum still in flight is abandoned (the memsets above queue behind it)
    return PV_OK;
}

int pv_get_time_cursor(const pv_handle *h, int64_t *out)
{
    if (!live(h)) return PV_ERR_DESTROYED;
    if (!out) return PV_ERR_ARGUMENT;
    *out = h->time_cursor;
    return PV_OK;
}

int pv_set_time_cursor(pv_handle *h, int64_t value)
{
    if (!live(h)) return PV_ERR_DESTROYED;
    if (value < 0) return fail(h, PV_ERR_ARGUMENT, "pv_set_time_cursor: negative");
    if (value % h->hop != 0) return fail(h, PV_ERR_ARGUMENT, "pv_set_time_cursor: not a multiple of hop_size (the reference only advances it by hop_size, pv:71)");
    h->time_cursor = value;
    return PV_OK;
}

int pv_export_state(pv_handle *h, int32_t ch, float *hist, float *acc, int64_t *time_cursor)
{
    if (!live(h)) return PV_ERR_DESTROYED;
    { const int rs_ = resident_stop(h); if (rs_ != PV_OK) return rs_; }
    if (ch < 0 || ch >= h->max_channels) return fail(h, PV_ERR_ARGUMENT, "pv_export_state: channel out of range");
    if (h->pending_nch > 0) return fail(h, PV_ERR_ARGUMENT, "pv_export_state: a quantum is pending (pv_process_end)");
    HIPCHK(h, hipSetDevice(h->device));
    const size_t bytes = sizeof(float) * (size_t)h->L;
    if (hist && h->L) HIPCHK(h, hipMemcpyAsync(hist, h->d_hist[h->cur] + (size_t)ch * h->L, bytes, hipMemcpyDeviceToHost, h->stream));
    if (acc && h->L) HIPCHK(h, hipMemcpyAsync(acc, h->d_acc[h->cur] + (size_t)ch * h->L, bytes, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (time_cursor) *time_cursor = h->time_cursor;
    return PV_OK;
}

int pv_import_state(pv_handle *h, int32_t ch, const float *hist, const float *acc, int64_t time_cursor)
{
    if (!live(h)) return PV_ERR_DESTROYED;
    { const int rs_ = resident_stop(h); if (rs_ != PV_OK) return rs_; }
    if (ch < 0 || ch >= h->max_channels) return fail(h, PV_ERR_ARGUMENT, "pv_import_state: channel out of range");
    if (h->pending_nch > 0) return fail(h, PV_ERR_ARGUMENT, "pv_import_state: a quantum is pending (pv_process_end)");
    if (time_cursor >= 0 && time_cursor % h->hop != 0) return fail(h, PV_ERR_ARGUMENT, "pv_import_state: time_cursor is not a multiple of hop_size");
    HIPCHK(h, hipSetDevice(h->device));
    const size_t bytes = sizeof(float) * (size_t)h->L;
    if (hist && h->L) HIPCHK(h, hipMemcpyAsync(h->d_hist[h->cur] + (size_t)ch * h->L, hist, bytes, hipMemcpyHostToDevice, h->stream));
    if (acc && h->L) HIPCHK(h, hipMemcpyAsync(h->d_acc[h->cur] + (size_t)ch * h->L, acc, bytes, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));              // the host buffers are the caller's again on return
    if (ch + 1 > h->used_channels) h->used_channels = ch + 1;
    if (time_cursor >= 0) h->time_cursor = time_cursor;
    return PV_OK;
}

int pv_set_stream(pv_handle *h, void *hip_stream)
{
    if (!live(h)) return PV_ERR_DESTROYED;
    { const int rs_ = resident_stop(h); if (rs_ != PV_OK) return rs_; }
    hipStream_t ns = hip_stream ? (hipStream_t)hip_stream : h->own_stream;
    if (ns != h->stream) {                                               // launches on the old stream finish before anything is queued on the new one (the chain lists of
        HIPCHK(h, hipSetDevice(h->device));                              // consecutive launches, the state ping-pong and the staging buffers are ordered by the stream alone)
        HIPCHK(h, hipStreamSynchronize(h->stream));
    }
    h->stream = ns;
    return PV_OK;
}

int pv_synchronize(pv_handle *h)
{
    if (!live(h)) return PV_ERR_DESTROYED;
    { const int rs_ = resident_stop(h); if (rs_ != PV_OK) return rs_; }
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return PV_OK;
}

int pv_forward_stats(pv_handle *h, uint64_t *frames, uint64_t *fallbacks, int32_t reset)
{
    if (!live(h)) return PV_ERR_DESTROYED;
    { const int rs_ = resident_stop(h); if (rs_ != PV_OK) return rs_; }
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    unsigned long long st[256];
    HIPCHK(h, hipMemcpy(st, h->d_fwd_stats, sizeof st, hipMemcpyDeviceToHost));
    uint64_t a = 0, b = 0;
    for (int i = 0; i < 128; i++) { a += st[2 * i]; b += st[2 * i + 1]; }
    if (frames) *frames = a;
    if (fallbacks) *fallbacks = b;
    if (reset) HIPCHK(h, hipMemset(h->d_fwd_stats, 0, sizeof(unsigned long long) * kFwdStatWords));
    return PV_OK;
}

#ifdef PV_FLIP_COUNT
// validation build only (tools/flip_count.py): out = {frames, guard-band fallbacks, frames whose fp32 and fp64 flags differ, of those NOT caught, largest q as float bits}
PV_API int pv_exp_flip_stats(pv_handle *h, uint64_t *out)
{
    if (!live(h) || !out) return PV_ERR_ARGUMENT;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    unsigned long long st[kFwdStatWords];
    HIPCHK(h, hipMemcpy(st, h->d_fwd_stats, sizeof st, hipMemcpyDeviceToHost));
    for (int k = 0; k < 12; k++) out[k] = 0;
    for (int i = 0; i < 128; i++) { out[0] += st[2 * i]; out[1] += st[2 * i + 1]; out[2] += st[256 + 2 * i]; out[3] += st[257 + 2 * i]; }
    for (int k = 0; k < 6; k++) out[4 + k] = st[512 + k] & 0xFFFFFFFFull;     // float bits: q_max; max over bins of (err - r eps A) / (eps rms|X|) and / (eps max|X|) for r = 8, 32; max of max|X| / rms|X|
    out[10] = st[600]; out[11] = st[601];       // frames the fp64 magnitudes prove to be class B; of those, frames the fp32 test calls class A (must be 0)
    return PV_OK;
}
#endif

int pv_process_begin(pv_handle *h, const float *const *in, int32_t nch, int32_t nsamples, float pitch_factor)
{
    if (!live(h)) return PV_ERR_DESTROYED;
    if (h->pending_nch > 0) return fail(h, PV_ERR_ARGUMENT, "pv_process_begin: the previous quantum has not been collected (pv_process_end)");
    if (nch < 0) return fail(h, PV_ERR_ARGUMENT, "pv_process: bad channel arguments");
    if (nch > h->max_channels) return fail(h, PV_ERR_CAPACITY, "pv_process: nch exceeds max_channels");
    const bool paused = (nsamples == 0 || in == nullptr);                       // ola-processor.js:93-100
    if (!paused && nsamples != h->hop) return fail(h, PV_ERR_ARGUMENT, "pv_process: nsamples must equal hop_size (or 0 when paused)");
    HIPCHK(h, hipSetDevice(h->device));
    const Commit before{h->cur, h->time_cursor, h->active_nch};
    if (h->active_nch >= 0 && nch != h->active_nch && !h->host_channels) {      // ola-processor.js:38-52
        const int rc = pv_reset_channels(h, 0, h->max_channels);
        if (rc != PV_OK) return rc;
    }
    h->active_nch = nch;
    if (nch == 0) { h->time_cursor += h->hop; return PV_OK; }                   // pv:71 still advances
    const int hop = h->hop;
    // Small quanta on a large-BAR device: the host writes pitchFactor + input straight into DEVICE memory (hipMalloc'ed memory is host-visible
    // through the BAR; the stores are posted and PCIe keeps them ahead of the launch doorbell), so the kernel starts on local HBM instead of a
    // PCIe read round trip of its own (tools/bar_probe.hip: 1 KB handed to a running wave in 3.5 us this way against 7.2 us read from pinned memory).
    // Above kBarInputMax the BAR write itself (~1.4 GB/s from one core) costs more than the waves' parallel reads.  The previous quantum's kernel
    // has finished (pv_process_end) before this buffer is written again.
    constexpr size_t kBarInputMax = 16384;
    const bool bar = h->h_ctl ? h->resident_in_bar : (h->bar_input && sizeof(float) * (size_t)nch * hop <= kBarInputMax);   // (a resident kernel's pointers are fixed)
    float *stage = bar ? h->d_quantum : h->h_pin;
    float *pin_in = stage + kHdrFloats;
    stage[0] = pitch_factor;
    auto stage_channel = [&](int c) {
        if (paused || !in[c]) memset(pin_in + (size_t)c * hop, 0, sizeof(float) * hop);
        else memcpy(pin_in + (size_t)c * hop, in[c], sizeof(float) * hop);       // host block is only valid during the call (ola:64)
    };
    const bool piecewise = h->h_ctl && h->resident_wg && h->d_pin_mapped;       // resident workgroup kernel: every channel is handed over as soon as it is staged
    if (!piecewise) for (int c = 0; c < nch; c++) stage_channel(c);
#if defined(__x86_64__)
    if (bar) _mm_sfence();                                                       // write-combining buffers drained before the doorbell
#endif
    if (bar && h->hdp_flush) *h->hdp_flush = 1u;                                 // ... and the device's host data path flushed behind them
    auto launch = [&]() -> int {
        if (h->d_pin_mapped && h->h_ctl) {
            // resident kernel: no launch -- start the waves if none are there (first quantum, or they left after their idle time-out), then publish the
            // quantum: its parameters first, the sequence number last
            // Long pause: the waves count their idle polls one by one and leave after ~50 ms, NOT at the same instant -- a quantum published while some
            // have left and others still poll would be picked up by the survivors only (they reset their count, the kernel never ends, the departed
            // channels never complete).  So after any pause that comes near the time-out the waves are told to leave, waited for, and started afresh.
            const auto now = std::chrono::steady_clock::now();
            if (h->resident_on && now - h->last_quantum > std::chrono::milliseconds(20)) {
                const int rc = resident_stop(h);
                if (rc != PV_OK) return rc;
            }
            h->last_quantum = now;
            if (!h->resident_on) { const int rc = resident_start(h, h->quantum_seq); if (rc != PV_OK) return rc; }
            unsigned seq = (h->quantum_seq + 1u) & 0xFFFFu;                  // the resident protocol carries 16 bits of it (never 0)
            if (seq == 0) seq = 1;
            h->quantum_seq = seq;
            if (nch > h->used_channels) h->used_channels = nch;
            const unsigned common = seq | ((unsigned)h->cur << 23) | ((unsigned)((h->time_cursor / hop) % h->R) << 24);
            auto publish = [&](volatile unsigned *word, unsigned value) { resident_publish(h, word, value); };
            if (piecewise) {
                // one word per channel slot: the workgroup of channel 0 is on its frame while the host still copies channel 1's input, and so on
                // (pv_process_end collects the outputs in the same order).  Slots that hold state but are not in this quantum carry it (count 0).
                for (int c = 0; c < nch; c++) { stage_channel(c); publish(h->h_ctl + 16 + c, common | (1u << 16)); h->copied[c] = false; }
                for (int c = nch; c < h->used_channels; c++) publish(h->h_ctl + 16 + c, common);
            } else {
                h->h_ctl[5] = (unsigned)h->used_channels;
                publish(h->h_ctl, common | ((unsigned)nch << 16));
            }
            h->cur ^= 1;                                                     // what run_chain() commits for a launched quantum
            h->time_cursor += hop;
            return PV_OK;
        }
        if (h->d_pin_mapped) {
            float *m_in = (bar ? h->d_quantum : h->d_pin_mapped) + kHdrFloats, *m_out = h->d_pin_mapped + kHdrFloats + (size_t)h->max_channels * hop;
            unsigned seq = 0;
            if (h->d_done) { seq = ++h->quantum_seq; if (seq == 0) seq = h->quantum_seq = 1; }
            return run_chain(h, m_in, m_out, nch, 1, hop, bar ? h->d_quantum : h->d_pin_mapped, 0, 1, true, -1, seq, pitch_factor >= 1.0f ? 1 : 0);
        }
        float *dq_in = h->d_quantum + kHdrFloats, *dq_out = dq_in + (size_t)h->max_channels * hop;
        float *pin_out = pin_in + (size_t)h->max_channels * hop;
        HIPCHK(h, hipMemcpyAsync(h->d_quantum, h->h_pin, sizeof(float) * (kHdrFloats + (size_t)nch * hop), hipMemcpyHostToDevice, h->stream));
        const int rc = run_chain(h, dq_in, dq_out, nch, 1, hop, h->d_quantum, 0, 1, true, -1, 0, pitch_factor >= 1.0f ? 1 : 0);
        if (rc != PV_OK) return rc;
        HIPCHK(h, hipMemcpyAsync(pin_out, dq_out, sizeof(float) * (size_t)nch * hop, hipMemcpyDeviceToHost, h->stream));
        return PV_OK;
    };
    const int rc = launch();
    if (rc != PV_OK) { h->cur = before.cur; h->time_cursor = before.time_cursor; h->active_nch = before.active_nch; return rc; }
    h->pending_nch = nch;
    h->pending_cur = before.cur; h->pending_time_cursor = before.time_cursor; h->pending_active_nch = before.active_nch;
    return PV_OK;
}

int pv_process_end(pv_handle *h, float *const *out)
{
    if (!live(h)) return PV_ERR_DESTROYED;
    const int nch = h->pending_nch;
    if (nch == 0 && h->active_nch == 0) return PV_OK;                            // a quantum without channels launched nothing
    if (nch <= 0) return fail(h, PV_ERR_ARGUMENT, "pv_process_end: no quantum is pending (pv_process_begin)");
    if (!out) return fail(h, PV_ERR_ARGUMENT, "pv_process: bad channel arguments");
    h->pending_nch = 0;
    const int hop = h->hop;
    const float *pin_out = h->h_pin + kHdrFloats + (size_t)h->max_channels * hop;
    hipError_t e = hipSuccess;
    bool done = false;
    if (h->d_done && h->d_pin_mapped) {
        // every chain of the quantum stores the sequence number into its word once its output sits in the pinned buffer: spin on them (a
        // quantum is a few microseconds of kernel; the runtime's own completion path -- signal, interrupt, wake-up -- costs more than that).
        // Bounded: after 200 ms without completion the stream wait below reports whatever went wrong.
        const unsigned seq = h->quantum_seq;
        const auto t0 = std::chrono::steady_clock::now();
        int c = 0;                                                                // channels are collected in order, each copied out as soon as it is complete
        for (int k = 0; k < nch && k < 64; k++) h->copied[k] = false;
        for (unsigned spins = 0;; spins++) {
            while (c < nch && h->h_done[c] == seq) {
                if (c < 64) {
                    std::atomic_thread_fence(std::memory_order_acquire);
                    if (out[c]) memcpy(out[c], pin_out + (size_t)c * hop, sizeof(float) * hop);
                    h->copied[c] = true;
                }
                c++;
            }
            if (c == nch) { done = true; break; }
#if defined(__x86_64__)
            _mm_pause();
#endif
            if ((spins & 0x3FFu) == 0x3FFu) {
                const auto waited = std::chrono::steady_clock::now() - t0;
                if (h->h_ctl && h->resident_on && waited > std::chrono::milliseconds(2) && hipStreamQuery(h->stream) == hipSuccess) {
                    // the resident waves left (idle time-out) just as this quantum was published: start new ones, they pick it up
                    // (a PARTIAL leave cannot happen here: pv_process_begin restarts the waves after every pause of 20 ms or more)
                    h->resident_on = false;
                    if (resident_start(h, seq - 1u) != PV_OK) break;
                    for (int k = 0; k < c && k < 64; k++) if (h->copied[k]) h->h_done[k] = seq;   // resident_start cleared the words of channels already collected
                }
                (void)hipGetLastError();
                if (waited > std::chrono::milliseconds(200)) break;
            }
        }
        std::atomic_thread_fence(std::memory_order_acquire);
    }
    if (!done) {
        if (h->resident_on) resident_publish(h, h->h_ctl + 4, 1u);             // ask the resident waves to leave: the wait below must end
        e = hipSetDevice(h->device);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        if (h->resident_on) { resident_publish(h, h->h_ctl + 4, 0u); h->resident_on = false; if (e == hipSuccess) e = hipErrorLaunchFailure; }   // the quantum did not complete
    }
    if (e != hipSuccess) {                                                       // the state is committed only when the whole quantum succeeded
        h->cur = h->pending_cur; h->time_cursor = h->pending_time_cursor; h->active_nch = h->pending_active_nch;
        return fail_hip(h, e, "pv_process_end: stream synchronize");
    }
    for (int c = 0; c < nch; c++)
        if (out[c] && !(done && c < 64 && h->copied[c])) memcpy(out[c], pin_out + (size_t)c * hop, sizeof(float) * hop);
    return PV_OK;                                                                // ola-processor.js:170: return true
}

int pv_process(pv_handle *h, const float *const *in, float *const *out, int32_t nch, int32_t nsamples, float pitch_factor)
{
    if (!live(h)) return PV_ERR_DESTROYED;
    if (nch > 0 && !out) return fail(h, PV_ERR_ARGUMENT, "pv_process: bad channel arguments");
    const int rc = pv_process_begin(h, in, nch, nsamples, pitch_factor);
    if (rc != PV_OK || nch == 0) return rc;
    return pv_process_end(h, out);
}

int pv_process_batch_device(pv_handle *h, const float *d_in, float *d_out, int32_t nch, int32_t nhops, int64_t ch_stride,
                            const float *d_pitch, int32_t pitch_stride, int32_t channels_per_stream)
{
    if (!live(h)) return PV_ERR_DESTROYED;
    { const int rs_ = resident_stop(h); if (rs_ != PV_OK) return rs_; }
    if (!d_in || !d_out || !d_pitch || nch <= 0 || nhops <= 0) return fail(h, PV_ERR_ARGUMENT, "pv_process_batch_device: bad arguments");
    if (h->pending_nch > 0) return fail(h, PV_ERR_ARGUMENT, "pv_process_batch_device: a quantum is pending (pv_process_end)");
    if (nch > h->max_channels) return fail(h, PV_ERR_CAPACITY, "pv_process_batch_device: nch exceeds max_channels");
    if (ch_stride < (int64_t)nhops * h->hop) return fail(h, PV_ERR_ARGUMENT, "pv_process_batch_device: ch_stride smaller than nhops*hop");
    if (pitch_stride != 0 && pitch_stride < nhops) return fail(h, PV_ERR_ARGUMENT, "pv_process_batch_device: pitch_stride smaller than nhops");
    if (pitch_stride < 0 || channels_per_stream < 0) return fail(h, PV_ERR_ARGUMENT, "pv_process_batch_device: negative stride");
    HIPCHK(h, hipSetDevice(h->device));
    return run_chain(h, d_in, d_out, nch, nhops, (long)ch_stride, d_pitch, pitch_stride, channels_per_stream, true, -1);
}

int pv_process_batch(pv_handle *h, const float *in, float *out, int32_t nch, int32_t nhops, int64_t ch_stride, const float *pitch,
                     int32_t pitch_stride, int32_t channels_per_stream)
{
    if (!live(h)) return PV_ERR_DESTROYED;
    { const int rs_ = resident_stop(h); if (rs_ != PV_OK) return rs_; }
    if (!in || !out || !pitch || nch <= 0 || nhops <= 0) return fail(h, PV_ERR_ARGUMENT, "pv_process_batch: bad arguments");
    if (h->pending_nch > 0) return fail(h, PV_ERR_ARGUMENT, "pv_process_batch: a quantum is pending (pv_process_end)");
    if (nch > h->max_channels || nhops > h->max_hops) return fail(h, PV_ERR_CAPACITY, "pv_process_batch: nch/nhops exceed the handle's capacity");
    const size_t row = (size_t)nhops * h->hop;
    if (ch_stride < (int64_t)row) return fail(h, PV_ERR_ARGUMENT, "pv_process_batch: ch_stride smaller than nhops*hop");
    if (pitch_stride < 0 || channels_per_stream < 0) return fail(h, PV_ERR_ARGUMENT, "pv_process_batch: negative stride");
    HIPCHK(h, hipSetDevice(h->device));
    const int cps = channels_per_stream > 0 ? channels_per_stream : 1;
    const int nrows = pitch_stride ? (nch + cps - 1) / cps : 1;
    if (pitch_stride && pitch_stride < nhops) return fail(h, PV_ERR_ARGUMENT, "pv_process_batch: pitch_stride smaller than nhops");
    const Commit before{h->cur, h->time_cursor, h->active_nch};
    // ---- plan: how the batch is cut into pieces (PINNED host buffers only: pageable memory is staged by the runtime, synchronously) ----
    // With in / out in page-locked memory (pv_host_alloc) the copies are asynchronous DMA: the batch is cut into up to kMaxPieces pieces and piece k+1
    // is on its way to the device while piece k is in the kernel and piece k-1 on its way back (three streams, two events per piece).  Pieces are
    //   * groups of whole streams (contiguous channel rows: the largest DMA segments, no extra halo) when the batch has enough of them, else
    //   * spans of hops: consecutive calls on the carried state, bit-identical to one call (tests: call-splitting invariance).
    // The kernel rate is 10-30x the PCIe rate, so the pieces only have to be large enough for the DMA engines.
    const size_t row_bytes = row * sizeof(float), total = row_bytes * (size_t)nch;
    const size_t span_bytes = ((size_t)(nch - 1) * (size_t)ch_stride + row) * sizeof(float);       // what the 2D copies touch
    const bool pinned = host_pinned_range(in, span_bytes) && host_pinned_range(out, span_bytes);
    int pieces = 1;
    bool by_channel = false;
    if (pinned && total >= ((size_t)4 << 20)) {
        const int groups = nch / cps;                                           // whole streams
        if (nch % cps == 0 && groups >= 4 && total / 4 >= ((size_t)1 << 20)) {
            by_channel = true;
            pieces = (int)(total / ((size_t)2 << 20));
            if (pieces > groups) pieces = groups;
        } else {
            const int min_hops = (int)((65536 + (size_t)h->hop * 4 - 1) / ((size_t)h->hop * 4));      // >= 64 KB per row segment
            pieces = (int)(total / ((size_t)2 << 20));
            if (pieces > nhops / (min_hops > 0 ? min_hops : 1)) pieces = nhops / (min_hops > 0 ? min_hops : 1);
        }
        if (pieces > kMaxPieces) pieces = kMaxPieces;
        if (pieces < 1) pieces = 1;
    }
    size_t snap_fl = 0;                                                      // > 0: the state snapshot of a hop-span pipeline has been taken
    auto batch = [&]() -> int {
        if (pieces > 1 && !h->pipe_ready) {
            HIPCHK(h, hipStreamCreateWithFlags(&h->s_in, hipStreamNonBlocking));
            HIPCHK(h, hipStreamCreateWithFlags(&h->s_out, hipStreamNonBlocking));
            for (int i = 0; i < kMaxPieces; i++) {
                HIPCHK(h, hipEventCreateWithFlags(&h->ev_in[i], hipEventDisableTiming));
                HIPCHK(h, hipEventCreateWithFlags(&h->ev_k[i], hipEventDisableTiming));
            }
            h->pipe_ready = true;
        }
        HIPCHK(h, hipMemcpy2DAsync(h->d_pitch, (size_t)nhops * sizeof(float), pitch, (size_t)(pitch_stride ? pitch_stride : nhops) * sizeof(float),
                                   (size_t)nhops * sizeof(float), nrows, hipMemcpyHostToDevice, h->stream));
        if (pieces == 1) {
            HIPCHK(h, hipMemcpy2DAsync(h->d_stage_in, row_bytes, in, (size_t)ch_stride * sizeof(float), row_bytes, nch, hipMemcpyHostToDevice, h->stream));
            const int rc = run_chain(h, h->d_stage_in, h->d_stage_out, nch, nhops, (long)row, h->d_pitch, pitch_stride ? nhops : 0, cps, true, -1);
            if (rc != PV_OK) return rc;
            HIPCHK(h, hipMemcpy2DAsync(out, (size_t)ch_stride * sizeof(float), h->d_stage_out, row_bytes, row_bytes, nch, hipMemcpyDeviceToHost, h->stream));
            HIPCHK(h, hipStreamSynchronize(h->stream));
            return PV_OK;
        }
        // the device staging buffers hold the WHOLE batch (planar, row = nhops * hop floats per channel): pieces never share bytes, no buffer is
        // reused inside a call, so the only ordering is H2D(k) -> kernel(k) -> D2H(k)
        const int groups = nch / cps;
        if (!by_channel) {
            // spans of hops commit the state ping-pong piece by piece: after two pieces both halves are overwritten.  A copy of the live half (every slot in use),
            // taken before the first piece, is what a failure in a later piece is rolled back to
            const int used = nch > h->used_channels ? nch : h->used_channels;
            const size_t fl = (size_t)used * (size_t)(h->L > 0 ? h->L : 1);
            if (2 * fl > h->snap_floats) {
                if (h->d_snap) (void)hipFree(h->d_snap);
                h->d_snap = nullptr; h->snap_floats = 0;
                HIPCHK(h, hipMalloc(&h->d_snap, 2 * fl * sizeof(float)));
                h->snap_floats = 2 * fl;
            }
            HIPCHK(h, hipMemcpyAsync(h->d_snap, h->d_hist[h->cur], fl * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
            HIPCHK(h, hipMemcpyAsync(h->d_snap + fl, h->d_acc[h->cur], fl * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
            snap_fl = fl;
        }
        for (int k = 0; k < pieces; k++) {
            int c0 = 0, cn = nch, m0 = 0, mn = nhops;
            if (by_channel) { c0 = (int)((long)groups * k / pieces) * cps; cn = (int)((long)groups * (k + 1) / pieces) * cps - c0; }
            else { m0 = (int)((long)nhops * k / pieces); mn = (int)((long)nhops * (k + 1) / pieces) - m0; }
            const size_t doff = (size_t)c0 * row + (size_t)m0 * h->hop, hoff = (size_t)c0 * (size_t)ch_stride + (size_t)m0 * h->hop;
            const size_t w = (size_t)mn * h->hop * sizeof(float);
            HIPCHK(h, hipMemcpy2DAsync(h->d_stage_in + doff, row_bytes, in + hoff, (size_t)ch_stride * sizeof(float), w, cn, hipMemcpyHostToDevice, h->s_in));
            HIPCHK(h, hipEventRecord(h->ev_in[k], h->s_in));
            HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_in[k], 0));
            const float *pp = h->d_pitch + (pitch_stride ? (size_t)(c0 / cps) * nhops : 0) + m0;
            const int rc = by_channel ? launch_chain(h, h->d_stage_in + doff, h->d_stage_out + doff, c0, cn, mn, (long)row, pp, pitch_stride ? nhops : 0, cps, true, -1, 0)
                                      : run_chain(h, h->d_stage_in + doff, h->d_stage_out + doff, nch, mn, (long)row, pp, pitch_stride ? nhops : 0, cps, true, -1);
            if (rc != PV_OK) return rc;
            if (k == 1 && h->test_fail_piece) return fail(h, PV_ERR_DEVICE, "pv_process_batch: injected failure behind the second piece (PV_FLAG_TEST_FAIL_SECOND_PIECE)");
            HIPCHK(h, hipEventRecord(h->ev_k[k], h->stream));
            HIPCHK(h, hipStreamWaitEvent(h->s_out, h->ev_k[k], 0));
            HIPCHK(h, hipMemcpy2DAsync(out + hoff, (size_t)ch_stride * sizeof(float), h->d_stage_out + doff, row_bytes, w, cn, hipMemcpyDeviceToHost, h->s_out));
        }
        if (by_channel) { const int rc = commit_chain(h, nch, nhops); if (rc != PV_OK) return rc; }
        HIPCHK(h, hipStreamSynchronize(h->s_out));
        HIPCHK(h, hipStreamSynchronize(h->stream));                          // (the state copies of the commit)
        return PV_OK;
    };
    const int rc = batch();
    if (rc != PV_OK) {                                                       // a failed call leaves the handle as it was (and nothing in flight)
        if (h->pipe_ready) { (void)hipStreamSynchronize(h->s_in); (void)hipStreamSynchronize(h->s_out); }
        (void)hipStreamSynchronize(h->stream);
        (void)hipGetLastError();
        h->cur = before.cur; h->time_cursor = before.time_cursor; h->active_nch = before.active_nch;
        if (snap_fl) {                                                       // pieces of a hop-span pipeline may have overwritten the half that is live again: put it back
            const bool ok = hipMemcpy(h->d_hist[h->cur], h->d_snap, snap_fl * sizeof(float), hipMemcpyDeviceToDevice) == hipSuccess &&
                            hipMemcpy(h->d_acc[h->cur], h->d_snap + snap_fl, snap_fl * sizeof(float), hipMemcpyDeviceToDevice) == hipSuccess;
            if (!ok) { (void)hipGetLastError(); (void)pv_reset(h); return fail(h, PV_ERR_DEVICE, "pv_process_batch: failed, and the channel state could not be restored: the handle has been reset"); }
        }
    }
    return rc;
}

int pv_host_alloc(size_t bytes, void **out)
{
    if (!out) return PV_ERR_ARGUMENT;
    *out = nullptr;
    if (bytes == 0) return PV_ERR_ARGUMENT;
    void *p = nullptr;
    const hipError_t e = hipHostMalloc(&p, bytes, hipHostMallocPortable);        // visible to every device of the process
    if (e != hipSuccess) { (void)hipGetLastError(); return fail_hip(nullptr, e, "pv_host_alloc: hipHostMalloc"); }
    *out = p;
    return PV_OK;
}

int pv_host_free(void *p)
{
    if (!p) return PV_OK;
    const hipError_t e = hipHostFree(p);
    if (e != hipSuccess) { (void)hipGetLastError(); return fail_hip(nullptr, e, "pv_host_free: hipHostFree"); }
    return PV_OK;
}

int pv_debug_frame(pv_handle *h, int32_t ch, const float *block, float pitch_factor, double *X, float *mag, int32_t *peak_flags, float *Y)
{
    if (!live(h)) return PV_ERR_DESTROYED;
    { const int rs_ = resident_stop(h); if (rs_ != PV_OK) return rs_; }
    if (ch < 0 || ch >= h->max_channels || !block) return fail(h, PV_ERR_ARGUMENT, "pv_debug_frame: bad arguments");
    HIPCHK(h, hipSetDevice(h->device));
    const int hop = h->hop, N = h->N, H = N / 2 + 1;
    // place the block at slot `ch` of the quantum buffer so that channel indexing matches the state arrays
    float *pin_in = h->h_pin + kHdrFloats;
    h->h_pin[0] = pitch_factor;
    memset(pin_in, 0, sizeof(float) * (size_t)(ch + 1) * hop);
    memcpy(pin_in + (size_t)ch * hop, block, sizeof(float) * hop);
    float *dq_in = h->d_quantum + kHdrFloats, *dq_out = dq_in + (size_t)h->max_channels * hop;
    HIPCHK(h, hipMemcpyAsync(h->d_quantum, h->h_pin, sizeof(float) * (kHdrFloats + (size_t)(ch + 1) * hop), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemsetAsync(h->d_dbgX, 0, sizeof(double) * 2 * N, h->stream));
    const int rc = run_chain(h, dq_in, dq_out, ch + 1, 1, hop, h->d_quantum, 0, 1, false, ch);   // commit=false: state untouched
    if (rc != PV_OK) return rc;
    // ... except that the kernel has written the frame's history / accumulator of slots [0, ch] into the OTHER ping-pong half.  Slots below
    // used_channels are rewritten there by the next committed launch (by the kernel or by run_chain's copy); slots [used_channels, ch] are
    // covered by neither -- they count as "zero in both halves" -- so they are zeroed again here.
    if (ch + 1 > h->used_channels && h->L > 0) {
        const size_t off = (size_t)h->used_channels * h->L, bytes = sizeof(float) * (size_t)(ch + 1 - h->used_channels) * h->L;
        HIPCHK(h, hipMemsetAsync(h->d_hist[h->cur ^ 1] + off, 0, bytes, h->stream));
        HIPCHK(h, hipMemsetAsync(h->d_acc[h->cur ^ 1] + off, 0, bytes, h->stream));
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (X) HIPCHK(h, hipMemcpy(X, h->d_dbgX, sizeof(double) * 2 * N, hipMemcpyDeviceToHost));
    if (mag) HIPCHK(h, hipMemcpy(mag, h->d_dbgMag, sizeof(float) * H, hipMemcpyDeviceToHost));
    if (peak_flags) HIPCHK(h, hipMemcpy(peak_flags, h->d_dbgFlags, sizeof(int) * H, hipMemcpyDeviceToHost));
    if (Y) HIPCHK(h, hipMemcpy(Y, h->d_dbgY, sizeof(float) * 2 * H, hipMemcpyDeviceToHost));
    return PV_OK;
}

#ifdef PV_STAMPS
/* measurement builds only (never in the product): phase clocks of the last launch, 16 words per chain */
PV_API int pv_exp_read_stamps(pv_handle *h, unsigned *dst, int nchains)
{
    if (!live(h) || !dst || nchains < 0 || nchains > 65536) return PV_ERR_ARGUMENT;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpy(dst, h->d_stamps, sizeof(unsigned) * 16 * (size_t)nchains, hipMemcpyDeviceToHost));
    return PV_OK;
}
#endif

}  // extern "C"
