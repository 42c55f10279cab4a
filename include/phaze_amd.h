/*
 * phaze_amd.h -- C ABI of the MI355X-native phase-vocoder pitch shifter (drop-in boundary).
 *
 * This library replaces ONE path of olvb/phaze: the AudioWorkletProcessor call
 *     process(inputs, outputs, {pitchFactor}) -> true
 * of OLAProcessor.process                (/root/reference/src/ola-processor.js:159-171) with
 *    PhaseVocoderProcessor.processOLA    (/root/reference/src/phase-vocoder.js:45-72) and the fft.js
 *    arithmetic it calls                 (/root/reference/www/phase-vocoder.js:2-508).
 * The reference has no FFI (it is JavaScript in a browser audio thread); the functions below are what a
 * Node.js N-API addon (phaze_amd/node/phaze_napi.c) or any other FFI (ctypes, cgo, JNI) binds.
 * See INTEGRATION.md for the reference-side binding.
 *
 * Conventions: plain C types only; the caller owns every I/O buffer; the library owns all device state;
 * every function returns an int status (PV_OK == 0) and never throws or aborts across the boundary.
 * A handle is NOT thread-safe (the reference runs on one audio thread: one caller per handle).
 */
#ifndef PHAZE_AMD_H
#define PHAZE_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define PV_API __attribute__((visibility("default")))
#else
#define PV_API
#endif

/* ---- status codes ---- */
enum {
    PV_OK = 0,
    PV_ERR_FFT_SIZE = 1,     /* 'FFT size must be a power of two and bigger than 1' (bundle:6-7)        */
    PV_ERR_ARGUMENT = 2,     /* NULL pointer, negative count, hop does not divide fft_size, ...          */
    PV_ERR_UNSUPPORTED = 3,  /* valid for the reference but outside this build's kernel range (2..1048576)  */
    PV_ERR_CAPACITY = 4,     /* more channels / hops than the handle was created for                     */
    PV_ERR_DEVICE = 5,       /* HIP runtime error (no GPU, launch failure, out of memory)                */
    PV_ERR_DESTROYED = 6     /* handle already destroyed                                                  */
};

typedef struct pv_handle pv_handle;

/* Version of this header's binary interface (struct layouts + semantics).  pv_abi_version() returns the value the LIBRARY was built with;
 * 2 = round 3: pv_config carries its own size, unknown pv_config.flags bits are rejected, PV_FLAG_PERSISTENT_STREAM;
 * 3 = round 4: pv_host_alloc / pv_host_free (page-locked host buffers: pv_process_batch pipelines them), PV_FLAG_TEST_NO_HDP_FLUSH,
 *     pv_reset_channels_part + PV_FLAG_HOST_CHANNEL_BOOKKEEPING;
 * 4 = round 5: PV_FLAG_FP64_FORWARD, pv_forward_stats;
 * 5 = round 6: PV_FLAG_TEST_FAIL_SECOND_PIECE (a test hook); no layout or semantic change of anything that existed. */
#define PV_ABI_VERSION 5

/* Construction options.  Replaces `new PhaseVocoderProcessor(options)` (phase-vocoder.js:24-43,
 * ola-processor.js:7-34).  The reference hard-codes fft_size 2048 (phase-vocoder.js:6) and hop 128
 * (ola-processor.js:3); both are options here; the host layer (phaze_amd/node/phase-vocoder.js) supplies the reference's values when omitted. */
typedef struct pv_config {
    int32_t struct_size;     /* sizeof(pv_config) as the CALLER compiled it (PV_CONFIG_INIT sets it).  pv_create rejects any other value with
                              * PV_ERR_ARGUMENT: a caller built against another layout would otherwise have trailing fields (flags!) read as garbage */
    int32_t fft_size;        /* N, power of two > 1 (else PV_ERR_FFT_SIZE); kernels cover 2..1048576      */
    int32_t hop_size;        /* h >= 2, divides N.  nbOverlaps R = N / h (ola-processor.js:17)           */
    int32_t max_channels;    /* channel slots owned by this handle (streams x channels); 0 => 2.  One launch
                              * takes any count for N = 1024 (hops 128..1024) and up to 65535 otherwise (PV_ERR_CAPACITY) */
    int32_t max_hops;        /* largest nhops of a host-buffer batch call (staging size); 0 => 1        */
    int32_t device_id;       /* HIP device ordinal                                                       */
    int32_t frames_per_chunk;/* batch kernel: output hops per workgroup (0 => auto)                      */
    int32_t flags;           /* PV_FLAG_* bits, 0 for production use                                     */
} pv_config;
/* pv_config cfg = PV_CONFIG_INIT; cfg.fft_size = ...;  (every other field 0 = its default) */
#define PV_CONFIG_INIT { (int32_t)sizeof(pv_config), 0, 0, 0, 0, 0, 0, 0 }

/* pv_config.flags: explicit A/B switches for tests and measurements (the library reads NO environment
 * variables).  All select complete, parity-tested implementations of the same path. */
enum {
    PV_FLAG_GENERIC_KERNEL = 1, /* always launch the LDS-staged fallback kernel (pv_chain_kernel)        */
    PV_FLAG_STREAM_COPY = 2,    /* streaming quantum through H2D + kernel + D2H copies instead of the
                                 * zero-copy mapping of the pinned staging buffer                         */
    PV_FLAG_WORKGROUP_KERNEL = 4, /* N = 2048 / 4096 / 8192: the eight-element workgroup kernel (pv_wg_kernel) instead of the
                                  * one-wave (pv_wave2k_kernel) / sixteen-element (pv_wg16_kernel) kernels */
    PV_FLAG_STREAM_EVENT_WAIT = 8, /* streaming quantum: wait through hipStreamSynchronize (round-2 behaviour).  Default since round 3: every
                                  * frame chain stores a sequence number into pinned host memory when its output is written and pv_process /
                                  * pv_process_end spin on those words (bounded; falls back to the stream wait) -- the runtime's completion
                                  * path costs more than the kernel of a quantum */
    PV_FLAG_STREAM_PINNED_INPUT = 16, /* streaming quantum: the kernel READS its hop from pinned host memory over PCIe (the round-2/3 form).  Default
                                  * since round 3 on a large-BAR device (hipDeviceAttributeIsLargeBar; every MI355X): a quantum of up to 16 KB of input
                                  * is written by the HOST into device memory through the BAR (posted writes, ordered before the launch doorbell),
                                  * so the kernel starts on local HBM instead of a PCIe read round trip; larger quanta and other devices keep the
                                  * pinned-memory read.  The output always goes to pinned host memory (posted device writes) */
    PV_FLAG_PERSISTENT_STREAM = 32, /* streaming quanta on a RESIDENT kernel (N = 1024 and 2048 with the hops their one-wave kernels cover, N = 8192 with
                                  * hop >= N/8; at most 64 channel slots; ignored elsewhere): the kernel of the first pv_process stays on the GPU, one wave
                                  * (N = 8192: one workgroup) per channel slot polling a control word -- in device memory, written by the host through the
                                  * large BAR together with quanta of up to 16 KB of input; in pinned host memory on other devices / with
                                  * PV_FLAG_STREAM_PINNED_INPUT.  A quantum then costs no launch.  The waves leave by themselves after ~50 ms without work
                                  * (and are relaunched on demand) and before any other use of the handle's stream (batch calls, state export / import,
                                  * reset, pv_synchronize).  Same kernel code, same bits as the launch-per-quantum form.  Off by default: while resident, a
                                  * device-wide synchronize of another user of the GPU waits for that idle time-out, and the handle should keep its own
                                  * stream (pv_set_stream to a stream shared with other work would queue that work behind the resident kernel) */
    PV_FLAG_TEST_NO_HDP_FLUSH = 64, /* TEST HOOK, never needed in production: behave as if the device did not expose its HDP flush register
                                  * (hipDeviceAttributeHdpMemFlushCntl).  The library then must not hand quanta over through the BAR -- a host store
                                  * could still sit in the device's host data path when the kernel reads -- and falls back to the pinned-memory
                                  * form on its own; tests/test_gpu_stream_forms.py runs every hand-over form under this bit */
    PV_FLAG_HOST_CHANNEL_BOOKKEEPING = 128, /* pv_process / pv_process_begin do NOT reset the channel state when nch differs from the previous call: the host does the
                                  * reference's bookkeeping itself with pv_reset_channels_part -- input and output buffers separately, ola-processor.js:38-52 -- as
                                  * phaze_amd/node/phase-vocoder.js does for hosts whose outputs do not mirror their inputs */
    PV_FLAG_FP64_FORWARD = 256,  /* every frame's forward transform in fp64, as the reference computes it (realTransform on JS doubles, bundle:306-508) and as every
                                  * round-4 kernel did.  Default since round 5 (N = 1024 and N = 2048): the forward transform runs in packed fp32 FIRST and the peak decisions
                                  * (phase-vocoder.js:95-116) are taken on its magnitudes wherever every comparison they rest on lies outside a guard band around the fp32
                                  * transform's error; a frame with a comparison inside the band re-runs its forward transform in fp64 (pv_forward_stats counts them).
                                  * The band is an EMPIRICALLY VALIDATED law, not an analytic bound: g = 10 eps max|X| against a largest observed discrepancy of 3.3
                                  * (a rigorous FFT error bound in max|X| terms is ~20x wider).  What backs "the decisions are the fp64 transform's" is a validation
                                  * build in which every frame computes both transforms and compares the two sets of peak flags: 0 frames whose flags differ without
                                  * the band asking for the fp64 transform over 4.7e7 + 1.2e7 frames (profiles/r05_flip_count*.json), re-measured on every GPU test run
                                  * over fifteen signal classes incl. four adversarial ones (tests/test_gpu_flip_count.py, >= 2.9e6 frames, band-shrink margin >= 2
                                  * asserted, 4.5 measured).  An uncaught flip would misplace one region of one frame (far inside the 1e-4 RMS bar) -- set this flag where
                                  * decision parity must hold by construction.  The source spectrum of a guarded frame carries the fp32 transform's rounding
                                  * (~1e-7 of the frame's rms instead of a correctly rounded fp64 value).  Which frames fall back depends on their own samples only:
                                  * chunked, call-split, streaming and batch runs of one stream still agree bit for bit */
    PV_FLAG_TEST_FAIL_SECOND_PIECE = 512, /* TEST HOOK, never needed in production: a pipelined host-buffer batch (pv_process_batch on page-locked memory, >= 4 MB) reports
                                  * PV_ERR_DEVICE behind its second piece, as a failed copy or launch would -- tests/test_gpu_batch_pipeline.py checks that the handle is
                                  * rolled back to its state before the call (timeCursor, ping-pong half, the state snapshot of a hop-span pipeline) */
    PV_FLAG_ALL = 1023           /* every bit this build knows: pv_create rejects anything else (PV_ERR_ARGUMENT) */
};

typedef struct pv_info {
    int32_t fft_size, hop_size, overlaps, max_channels, max_hops;
    int32_t threads_per_workgroup, lds_bytes_per_workgroup, frames_per_chunk;
    int32_t compute_units, device_id;
    char device_name[64];
    char kernel_name[32];    /* "pv_wave_kernel_1024" (N = 1024, hop in {128,256,512,1024}), "pv_wave2k_kernel" (N = 2048, hop in
                              * {128,256,512,1024,2048}: one wave per frame), "pv_wg16_kernel" (N = 4096 / 8192, hop in {N/8, N/4, N/2,
                              * N}: two / four waves per frame, sixteen elements per thread), "pv_wg_kernel" (N >= 2048 with smaller even
                              * hops that fit LDS, and PV_FLAG_WORKGROUP_KERNEL: a workgroup per frame, eight elements per thread) or
                              * "pv_chain_kernel" (everything else / PV_FLAG_GENERIC_KERNEL) */
} pv_info;

/* ---- lifetime ---------------------------------------------------------------------------------- */
/* Allocates device state (zeroed input history + overlap-add accumulators, timeCursor = 0) and the
 * FFT/Hann tables.  Replaces the constructor chain phase-vocoder.js:24-43 -> ola-processor.js:7-34.  */
PV_API int pv_create(const pv_config *cfg, pv_handle **out);
/* PV_ABI_VERSION of the loaded library (a binding checks it against the header it was generated from). */
PV_API int pv_abi_version(void);
PV_API int pv_destroy(pv_handle *h);

/* Human-readable text of the last failure on this handle (h == NULL: of the last failed pv_create). */
PV_API const char *pv_last_error(const pv_handle *h);
PV_API const char *pv_status_string(int status);
PV_API int pv_get_info(const pv_handle *h, pv_info *out);
/* Number of HIP devices this process can create handles on (pv_config.device_id in [0, count)): what a host needs to shard streams over the
 * GPUs of a node (stream s -> device s mod count; streams are independent processors, nothing is exchanged). */
PV_API int pv_device_count(int32_t *out);

/* Forward-transform statistics since the handle was created (or last reset): frames whose forward transform an fp32-first kernel instance computed, and how many
 * of them re-ran it in fp64 because a peak decision (phase-vocoder.js:95-116) was within the fp32 transform's error (PV_FLAG_FP64_FORWARD).  Frames that run on
 * instances without the fp32-first path are not counted.  Synchronizes the handle's stream -- and, like every call that puts work on that stream, first asks the resident
 * waves of a PV_FLAG_PERSISTENT_STREAM handle to leave (the next quantum relaunches them: ~20 us once): poll it between streams, not between quanta.  Either pointer may be
 * NULL; reset != 0 zeroes the counters. */
PV_API int pv_forward_stats(pv_handle *h, uint64_t *frames, uint64_t *fallbacks, int32_t reset);

/* ---- state ------------------------------------------------------------------------------------- */
/* Zero history + accumulators of ALL channels and set timeCursor = 0 (a freshly constructed processor). */
PV_API int pv_reset(pv_handle *h);
/* Zero history + accumulator of channel slots [first, first+count): what allocateInputChannels /
 * allocateOutputChannels do when a channel count changes (ola-processor.js:38-52,54-88).  timeCursor kept. */
PV_API int pv_reset_channels(pv_handle *h, int32_t first, int32_t count);
/* The same for ONE side of the state: the reference reallocates inputBuffers when inputs[i].length changes (ola-processor.js:40-44,54-71: the input history
 * restarts from zeros) and outputBuffers when outputs[i].length changes (:46-51,73-88: the pending overlap-add sums restart from zeros) -- two separate
 * events for a host whose outputs do not mirror its inputs.  parts = PV_STATE_HISTORY | PV_STATE_ACCUMULATOR bits. */
enum { PV_STATE_HISTORY = 1, PV_STATE_ACCUMULATOR = 2 };
PV_API int pv_reset_channels_part(pv_handle *h, int32_t first, int32_t count, int32_t parts);
/* timeCursor (phase-vocoder.js:31,71): samples consumed so far = hops * hop_size.  The reference only ever
 * advances it by hop_size (pv:71), so a value that is negative or not a multiple of hop_size is rejected
 * with PV_ERR_ARGUMENT (the register kernels rely on t = m * hop for their exact rotations). */
PV_API int pv_get_time_cursor(const pv_handle *h, int64_t *out);
PV_API int pv_set_time_cursor(pv_handle *h, int64_t value);

/* State export / import of ONE channel slot: everything the reference keeps between process() calls for a channel --
 * hist[N - hop] = the newest N - hop input samples (inputBuffers, ola-processor.js:59,121-127), acc[N - hop] = the pending
 * overlap-add sums (outputBuffers, ola-processor.js:77,130-137) -- plus the processor-wide timeCursor (phase-vocoder.js:31).
 * Checkpoint / resume, moving a stream to another handle or GPU, and splitting ONE stream along the time axis all reduce to
 * this: a handle that imports the state another one exported continues bit for bit.  (Time-sharding needs no hand-over at all:
 * hist is plain input, and the accumulator only depends on the last R - 1 frames, so a span can import {input tail, acc = 0,
 * cursor} R - 1 hops early and recompute its halo; see bench.py --time-shard.)  Synchronous.  hist / acc may be NULL (skipped);
 * pv_import_state sets the handle's timeCursor when time_cursor >= 0 (multiple of hop_size) and leaves it when negative. */
PV_API int pv_export_state(pv_handle *h, int32_t ch, float *hist, float *acc, int64_t *time_cursor);
PV_API int pv_import_state(pv_handle *h, int32_t ch, const float *hist, const float *acc, int64_t time_cursor);

/* ---- the hot call, streaming form (one render quantum) ------------------------------------------- */
/* Replaces OLAProcessor.process(inputs, outputs, parameters) (ola-processor.js:159-171) for ONE input /
 * ONE output with nch channels:  in[c] -> nsamples (== hop_size) host floats, valid only during the call;
 * out[c] <- hop_size floats.  pitch_factor = parameters.pitchFactor[last] (phase-vocoder.js:47).
 * nsamples == 0 (or in == NULL) reproduces the paused branch (ola-processor.js:93-100): the newest hop
 * is treated as zeros, timeCursor still advances.  A change of nch against the previous call resets all
 * channel state first (ola-processor.js:38-52).  Synchronous: outputs are valid on return.  PV_OK <=> the
 * reference's `return true`. */
PV_API int pv_process(pv_handle *h, const float *const *in, float *const *out, int32_t nch,
                      int32_t nsamples, float pitch_factor);

/* The same quantum split into its launch and its wait, for a host that drives SEVERAL handles (the inputs of one processor with
 * numberOfInputs > 1, phase-vocoder.js:49-50, or handles on different GPUs): call pv_process_begin on every handle -- each copies its
 * blocks to pinned memory and launches on its own stream, nothing waits -- then pv_process_end on every handle.  All launches are in
 * flight before the first wait.  pv_process(h, in, out, ...) == pv_process_begin(h, in, ...) + pv_process_end(h, out).  Between the two calls
 * the handle accepts no other call; pv_process_end without a pending quantum returns PV_ERR_ARGUMENT.  When the wait fails the handle is
 * rolled back to its state before pv_process_begin. */
PV_API int pv_process_begin(pv_handle *h, const float *const *in, int32_t nch, int32_t nsamples, float pitch_factor);
PV_API int pv_process_end(pv_handle *h, float *const *out);

/* ---- the hot call, batch (throughput) form ------------------------------------------------------- */
/* nhops consecutive process() calls for nch channel slots in one launch.  Planar layout: channel c
 * occupies in[c*ch_stride .. c*ch_stride + nhops*hop_size), same for out.  pitch[m] is the k-rate
 * pitchFactor of hop m; with pitch_stride != 0 channel c uses the row of its stream,
 * pitch[(c / channels_per_stream) * pitch_stride + m] (independent processors batched together).
 * State (history, accumulator tail, timeCursor) carries across calls exactly as if process() had been
 * called hop by hop.  Host-pointer variant: synchronous, stages through device memory.  When `in` and `out` are page-locked memory
 * (pv_host_alloc below, or anything hipHostMalloc / hipHostRegister produced) a batch of 4 MB or more is PIPELINED: cut into up to 16 pieces
 * -- groups of whole streams when there are enough, else spans of hops -- with piece k+1 on its way to the device while piece k is in the
 * kernel and piece k-1 on its way back (DMA both ways at once; the results are bit-identical to the unpipelined call).  Pageable buffers
 * work as before (the runtime stages them synchronously: several times slower). */
PV_API int pv_process_batch(pv_handle *h, const float *in, float *out, int32_t nch, int32_t nhops,
                            int64_t ch_stride, const float *pitch, int32_t pitch_stride,
                            int32_t channels_per_stream);

/* Page-locked host memory for the batch call above (hipHostMalloc, visible to every device of the process): what a host that owns its audio
 * buffers should put them in -- the N-API addon hands it out as external ArrayBuffers (native.allocPinned), so that a Node host writes its
 * streams in place.  Replaces the `new Float32Array(...)` a caller of OLAProcessor.process owns (ola-processor.js:159-171 receives host
 * arrays).  No handle needed; pv_host_free(NULL) is a no-op. */
PV_API int pv_host_alloc(size_t bytes, void **out);
PV_API int pv_host_free(void *p);

/* Device-pointer variant: in/out/pitch are DEVICE pointers (HBM-resident), the launch is asynchronous on
 * the handle's stream (pv_set_stream / pv_synchronize).  This is the form bench.py times. */
PV_API int pv_process_batch_device(pv_handle *h, const float *d_in, float *d_out, int32_t nch,
                                   int32_t nhops, int64_t ch_stride, const float *d_pitch,
                                   int32_t pitch_stride, int32_t channels_per_stream);

/* Use an externally owned hipStream_t (e.g. torch.cuda.current_stream().cuda_stream); NULL => the
 * handle's own stream. */
PV_API int pv_set_stream(pv_handle *h, void *hip_stream);
PV_API int pv_synchronize(pv_handle *h);

/* ---- test taps ----------------------------------------------------------------------------------- */
/* Runs ONE frame of channel `ch` through the kernels from the CURRENT state without changing it and
 * returns the intermediates the reference keeps in freqComplexBuffer / magnitudes / peakIndexes /
 * freqComplexBufferShifted (phase-vocoder.js:37-42): X[2N] doubles (bins 0..N/2 and, when the frame reads
 * it, the above-Nyquist residue), mag[N/2+1], peak_flags[N/2+1] (0/1), Y[2*(N/2+1)] floats.
 * block: hop_size host floats (the newest hop).  Any output pointer may be NULL. */
PV_API int pv_debug_frame(pv_handle *h, int32_t ch, const float *block, float pitch_factor, double *X,
                          float *mag, int32_t *peak_flags, float *Y);

#ifdef __cplusplus
}
#endif
#endif /* PHAZE_AMD_H */
