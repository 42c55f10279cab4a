#define _POSIX_C_SOURCE 200809L
/*
 * phaze_napi.c -- thin N-API (C) addon over the C ABI of include/phaze_amd.h.
 *
 * Host language of the reference is JavaScript (an AudioWorkletProcessor, /root/reference/src/phase-vocoder.js);
 * this addon is what lets a Node.js host keep that surface while the inner loop runs as HIP kernels.
 * It holds no algorithm: every export forwards to one pv_* entry point.  JS typed-array memory is only
 * touched during the call (the host may reuse its blocks afterwards, ola-processor.js:64).
 *
 * Build: plain gcc against /usr/include/node (no node-gyp, no network) -- see Makefile.
 */
#define NAPI_VERSION 6
#include <node_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "../../include/phaze_amd.h"

#define MAX_CH 64

/* what a JS "handle" external points to: the C handle plus the constants process() needs per quantum (no pv_get_info on the hot call) */
typedef struct pv_slot {
    pv_handle *h;
    int32_t hop;
    int32_t fft;
    int busy;          /* an asynchronous batch runs on a worker thread: the handle accepts no other call until its promise settles */
    int doomed;        /* destroy() arrived while busy: the completion callback destroys the handle */
    int orphan;        /* the JS handle object was collected while busy: the completion callback also frees this slot */
    double t_begin_ms, t_end_ms;   /* CLOCK_MONOTONIC window of the last asynchronous batch on its worker thread (batchWindow(): are the shards really side by side?) */
    int begun;         /* channel count of the quantum pv_process_begin launched (pv_process_end indexes exactly that many output pointers) */
} pv_slot;

#define NAPI_OK_OR_THROW(env, call, msg)                         \
    do {                                                         \
        if ((call) != napi_ok) {                                 \
            napi_throw_error((env), NULL, (msg));                \
            return NULL;                                         \
        }                                                        \
    } while (0)

static napi_value throw_status(napi_env env, pv_handle *h, int rc)
{
    const char *msg = pv_last_error(h);
    if (!msg || !msg[0]) msg = pv_status_string(rc);
    char code[16];
    snprintf(code, sizeof code, "PV_%d", rc);
    napi_throw_error(env, code, msg);   /* e.g. Error('FFT size must be a power of two and bigger than 1') (bundle:6-7) */
    return NULL;
}

static void finalize_handle(napi_env env, void *data, void *hint)
{
    (void)env; (void)hint;
    pv_slot *slot = (pv_slot *)data;
    if (slot) {
        if (slot->busy) { slot->doomed = 1; slot->orphan = 1; return; }      /* the worker thread still owns it: batch_complete cleans up */
        if (slot->h) pv_destroy(slot->h);
        free(slot);
    }
}

static int get_i32_prop(napi_env env, napi_value obj, const char *name, int32_t dflt)
{
    napi_value v;
    bool has = false;
    if (napi_has_named_property(env, obj, name, &has) != napi_ok || !has) return dflt;
    if (napi_get_named_property(env, obj, name, &v) != napi_ok) return dflt;
    napi_valuetype t;
    if (napi_typeof(env, v, &t) != napi_ok || t != napi_number) return dflt;
    int32_t out = dflt;
    napi_get_value_int32(env, v, &out);
    return out;
}

static pv_slot *unwrap(napi_env env, napi_value v)
{
    void *p = NULL;
    if (napi_get_value_external(env, v, &p) != napi_ok || !p) {
        napi_throw_type_error(env, NULL, "expected a phaze handle");
        return NULL;
    }
    pv_slot *slot = (pv_slot *)p;
    if (!slot->h) {
        napi_throw_error(env, "PV_6", pv_status_string(PV_ERR_DESTROYED));
        return NULL;
    }
    if (slot->busy) {
        napi_throw_error(env, "PV_BUSY", "handle is busy: an asynchronous batch has not settled yet (one caller per handle, ola-processor.js runs on one thread)");
        return NULL;
    }
    return slot;
}

/* create({fftSize, hopSize, maxChannels, maxHops, deviceId, framesPerChunk, flags}) -> external */
static napi_value js_create(napi_env env, napi_callback_info info)
{
    size_t argc = 1;
    napi_value argv[1];
    NAPI_OK_OR_THROW(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL), "bad arguments");
    if (argc < 1) { napi_throw_type_error(env, NULL, "create(options) needs an options object"); return NULL; }
    pv_config cfg = PV_CONFIG_INIT;
    cfg.fft_size = get_i32_prop(env, argv[0], "fftSize", 2048);          /* phase-vocoder.js:6  */
    cfg.hop_size = get_i32_prop(env, argv[0], "hopSize", 128);           /* ola-processor.js:3  */
    cfg.max_channels = get_i32_prop(env, argv[0], "maxChannels", 2);
    cfg.max_hops = get_i32_prop(env, argv[0], "maxHops", 1);
    cfg.device_id = get_i32_prop(env, argv[0], "deviceId", 0);
    cfg.frames_per_chunk = get_i32_prop(env, argv[0], "framesPerChunk", 0);
    cfg.flags = get_i32_prop(env, argv[0], "flags", 0);                  /* PV_FLAG_* (tests / measurements only) */
    pv_handle *h = NULL;
    const int rc = pv_create(&cfg, &h);
    if (rc != PV_OK) return throw_status(env, NULL, rc);
    pv_slot *slot = (pv_slot *)calloc(1, sizeof(pv_slot));
    if (!slot) { pv_destroy(h); napi_throw_error(env, NULL, "out of memory"); return NULL; }
    slot->h = h;
    slot->hop = cfg.hop_size;
    slot->fft = cfg.fft_size;
    slot->busy = 0;
    slot->doomed = 0;
    slot->orphan = 0;
    napi_value ext;
    NAPI_OK_OR_THROW(env, napi_create_external(env, slot, finalize_handle, NULL, &ext), "napi_create_external failed");
    return ext;
}

static napi_value js_destroy(napi_env env, napi_callback_info info)
{
    size_t argc = 1;
    napi_value argv[1];
    NAPI_OK_OR_THROW(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL), "bad arguments");
    void *p = NULL;
    if (argc >= 1 && napi_get_value_external(env, argv[0], &p) == napi_ok && p) {
        pv_slot *slot = (pv_slot *)p;
        if (slot->busy) slot->doomed = 1;                   /* the worker thread still uses the handle: destroyed when its batch completes */
        else if (slot->h) { pv_destroy(slot->h); slot->h = NULL; }
    }
    return NULL;
}

/* gather float* of each Float32Array in a JS array; returns channel count, lengths in len[] */
static int gather_channels(napi_env env, napi_value arr, float **ptr, size_t *len, int maxn)
{
    uint32_t n = 0;
    bool is_arr = false;
    if (napi_is_array(env, arr, &is_arr) != napi_ok || !is_arr) return -1;
    napi_get_array_length(env, arr, &n);
    if ((int)n > maxn) return -2;
    for (uint32_t c = 0; c < n; c++) {
        napi_value el;
        napi_typedarray_type ty;
        size_t l = 0, off = 0;
        void *data = NULL;
        napi_value ab;
        if (napi_get_element(env, arr, c, &el) != napi_ok) return -1;
        if (napi_get_typedarray_info(env, el, &ty, &l, &data, &ab, &off) != napi_ok || ty != napi_float32_array) return -1;
        ptr[c] = (float *)data;
        len[c] = l;
    }
    return (int)n;
}

/* process(handle, inputChannels: Float32Array[], outputChannels: Float32Array[], pitchFactor: number) -> true
 * One render quantum of ONE input/output pair: OLAProcessor.process (ola-processor.js:159-171). */
static napi_value js_process(napi_env env, napi_callback_info info)
{
    size_t argc = 4;
    napi_value argv[4];
    NAPI_OK_OR_THROW(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL), "bad arguments");
    if (argc < 4) { napi_throw_type_error(env, NULL, "process(handle, inputs, outputs, pitchFactor)"); return NULL; }
    pv_slot *slot = unwrap(env, argv[0]);
    if (!slot) return NULL;
    float *in[MAX_CH], *out[MAX_CH];
    size_t inlen[MAX_CH], outlen[MAX_CH];
    const int nin = gather_channels(env, argv[1], in, inlen, MAX_CH);
    const int nout = gather_channels(env, argv[2], out, outlen, MAX_CH);
    if (nin < 0 || nout < 0) { napi_throw_type_error(env, NULL, "inputs/outputs must be arrays of Float32Array (<= 64 channels)"); return NULL; }
    double pf = 1.0;
    napi_get_value_double(env, argv[3], &pf);
    const int hop = slot->hop;
    /* paused: inputs[0][0].length == 0 (ola-processor.js:93) */
    int nsamples = hop;
    if (nin > 0 && inlen[0] == 0) nsamples = 0;
    float *outp[MAX_CH];
    for (int c = 0; c < nin; c++) {
        if (nsamples && inlen[c] != (size_t)hop) { napi_throw_range_error(env, NULL, "input block length must equal hopSize"); return NULL; }
        outp[c] = (c < nout && outlen[c] >= (size_t)hop) ? out[c] : NULL;   /* "output is symetric to input" (phase-vocoder.js:51) */
    }
    const int rc = pv_process(slot->h, (const float *const *)in, outp, nin, nsamples, (float)pf);
    if (rc != PV_OK) return throw_status(env, slot->h, rc);
    napi_value t;
    napi_get_boolean(env, true, &t);
    return t;                                                               /* ola-processor.js:170 */
}

/* processBatch(handle, in: Float32Array [nch*nhops*hop], out: Float32Array, nch, nhops, pitch: Float32Array, pitchStride, channelsPerStream) */
static napi_value js_process_batch(napi_env env, napi_callback_info info)
{
    size_t argc = 9;
    napi_value argv[9];
    NAPI_OK_OR_THROW(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL), "bad arguments");
    if (argc < 6) { napi_throw_type_error(env, NULL, "processBatch(handle, in, out, nch, nhops, pitch[, pitchStride, channelsPerStream])"); return NULL; }
    pv_slot *slot = unwrap(env, argv[0]);
    if (!slot) return NULL;
    napi_typedarray_type ty;
    size_t nin = 0, nout = 0, npitch = 0, off;
    void *din = NULL, *dout = NULL, *dp = NULL;
    napi_value ab;
    if (napi_get_typedarray_info(env, argv[1], &ty, &nin, &din, &ab, &off) != napi_ok || ty != napi_float32_array ||
        napi_get_typedarray_info(env, argv[2], &ty, &nout, &dout, &ab, &off) != napi_ok || ty != napi_float32_array ||
        napi_get_typedarray_info(env, argv[5], &ty, &npitch, &dp, &ab, &off) != napi_ok || ty != napi_float32_array) {
        napi_throw_type_error(env, NULL, "in, out and pitch must be Float32Array");
        return NULL;
    }
    int32_t nch = 0, nhops = 0, pstride = 0, cps = 1;
    int64_t chs = 0;
    napi_get_value_int32(env, argv[3], &nch);
    napi_get_value_int32(env, argv[4], &nhops);
    if (argc > 6) napi_get_value_int32(env, argv[6], &pstride);
    if (argc > 7) napi_get_value_int32(env, argv[7], &cps);
    if (argc > 8) napi_get_value_int64(env, argv[8], &chs);                  /* channel stride in floats; 0 / omitted: nhops * hopSize (packed rows) */
    if (chs <= 0) chs = (int64_t)nhops * slot->hop;
    if (nch <= 0 || nhops <= 0 || chs < (int64_t)nhops * slot->hop) { napi_throw_range_error(env, NULL, "buffer sizes do not match nch*nhops*hopSize"); return NULL; }
    const size_t need = (size_t)(nch - 1) * (size_t)chs + (size_t)nhops * (size_t)slot->hop;
    const size_t rows = pstride ? (size_t)((nch + (cps > 0 ? cps : 1) - 1) / (cps > 0 ? cps : 1)) : 1;
    if (nch <= 0 || nhops <= 0 || nin < need || nout < need || npitch < (pstride ? (rows - 1) * (size_t)pstride + (size_t)nhops : (size_t)nhops)) {
        napi_throw_range_error(env, NULL, "buffer sizes do not match nch*nhops*hopSize");
        return NULL;
    }
    const int rc = pv_process_batch(slot->h, (const float *)din, (float *)dout, nch, nhops, chs, (const float *)dp, pstride, cps);
    if (rc != PV_OK) return throw_status(env, slot->h, rc);
    napi_value t;
    napi_get_boolean(env, true, &t);
    return t;
}


/* processBegin(handle, inputChannels, pitchFactor) / processEnd(handle, outputChannels) -> true: the quantum split into launch and wait
 * (pv_process_begin / pv_process_end) so that a processor with several inputs -- one handle each, phase-vocoder.js:49-50 -- has every launch
 * in flight before it waits for the first. */
static napi_value js_process_begin(napi_env env, napi_callback_info info)
{
    size_t argc = 3;
    napi_value argv[3];
    NAPI_OK_OR_THROW(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL), "bad arguments");
    if (argc < 3) { napi_throw_type_error(env, NULL, "processBegin(handle, inputs, pitchFactor)"); return NULL; }
    pv_slot *slot = unwrap(env, argv[0]);
    if (!slot) return NULL;
    float *in[MAX_CH];
    size_t inlen[MAX_CH];
    const int nin = gather_channels(env, argv[1], in, inlen, MAX_CH);
    if (nin < 0) { napi_throw_type_error(env, NULL, "inputs must be an array of Float32Array (<= 64 channels)"); return NULL; }
    double pf = 1.0;
    napi_get_value_double(env, argv[2], &pf);
    int nsamples = slot->hop;
    if (nin > 0 && inlen[0] == 0) nsamples = 0;                               /* paused (ola-processor.js:93) */
    for (int c = 0; c < nin; c++)
        if (nsamples && inlen[c] != (size_t)slot->hop) { napi_throw_range_error(env, NULL, "input block length must equal hopSize"); return NULL; }
    const int rc = pv_process_begin(slot->h, (const float *const *)in, nin, nsamples, (float)pf);
    if (rc != PV_OK) return throw_status(env, slot->h, rc);
    slot->begun = nin;
    napi_value n;
    napi_create_int32(env, nin, &n);
    return n;
}

static napi_value js_process_end(napi_env env, napi_callback_info info)
{
    size_t argc = 3;
    napi_value argv[3];
    NAPI_OK_OR_THROW(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL), "bad arguments");
    if (argc < 2) { napi_throw_type_error(env, NULL, "processEnd(handle, outputs[, nInputChannels])"); return NULL; }
    pv_slot *slot = unwrap(env, argv[0]);
    if (!slot) return NULL;
    float *out[MAX_CH], *outp[MAX_CH];
    size_t outlen[MAX_CH];
    const int nout = gather_channels(env, argv[1], out, outlen, MAX_CH);
    if (nout < 0) { napi_throw_type_error(env, NULL, "outputs must be an array of Float32Array (<= 64 channels)"); return NULL; }
    /* pv_process_end reads one pointer per channel of the quantum that was BEGUN, whatever the caller passes here: missing or short outputs are
     * NULL (skipped), as the reference skips outputs that do not mirror the inputs (phase-vocoder.js:51).  The optional third argument is accepted
     * for compatibility and ignored. */
    for (int c = 0; c < MAX_CH; c++) outp[c] = (c < slot->begun && c < nout && outlen[c] >= (size_t)slot->hop) ? out[c] : NULL;
    slot->begun = 0;
    const int rc = pv_process_end(slot->h, outp);
    if (rc != PV_OK) return throw_status(env, slot->h, rc);
    napi_value t;
    napi_get_boolean(env, true, &t);
    return t;
}

/* processBatchAsync(handle, in, out, nch, nhops, pitch[, pitchStride, channelsPerStream]) -> Promise<true>
 * The batch of processBatch on a libuv worker thread (napi_async_work): the JS thread returns at once, so ONE Node process keeps a batch in
 * flight on every GPU of the node (sharded.js: stream s -> handle s mod N).  The typed arrays are referenced until the promise settles. */
typedef struct batch_job {
    napi_async_work work;
    napi_deferred deferred;
    napi_ref refs[3];
    pv_slot *slot;
    const float *in, *pitch;
    float *out;
    int32_t nch, nhops, pstride, cps;
    int64_t chs;
    int rc;
    char err[256];
} batch_job;

static void batch_execute(napi_env env, void *data)
{
    (void)env;
    batch_job *j = (batch_job *)data;
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    j->slot->t_begin_ms = (double)ts.tv_sec * 1e3 + (double)ts.tv_nsec * 1e-6;
    j->rc = pv_process_batch(j->slot->h, j->in, j->out, j->nch, j->nhops, j->chs, j->pitch, j->pstride, j->cps);
    if (j->rc != PV_OK) {
        const char *m = pv_last_error(j->slot->h);
        snprintf(j->err, sizeof j->err, "%s", (m && m[0]) ? m : pv_status_string(j->rc));
    }
    clock_gettime(CLOCK_MONOTONIC, &ts);
    j->slot->t_end_ms = (double)ts.tv_sec * 1e3 + (double)ts.tv_nsec * 1e-6;
}

static void batch_complete(napi_env env, napi_status status, void *data)
{
    batch_job *j = (batch_job *)data;
    pv_slot *slot = j->slot;
    slot->busy = 0;
    if (slot->doomed && slot->h) { pv_destroy(slot->h); slot->h = NULL; }
    for (int i = 0; i < 3; i++) napi_delete_reference(env, j->refs[i]);
    if (status == napi_ok && j->rc == PV_OK) {
        napi_value t;
        napi_get_boolean(env, true, &t);
        napi_resolve_deferred(env, j->deferred, t);
    } else {
        napi_value msg, err;
        napi_create_string_utf8(env, status == napi_ok ? j->err : "asynchronous work cancelled", NAPI_AUTO_LENGTH, &msg);
        napi_create_error(env, NULL, msg, &err);
        napi_reject_deferred(env, j->deferred, err);
    }
    napi_delete_async_work(env, j->work);
    free(j);
    if (slot->orphan) free(slot);
}

static napi_value js_process_batch_async(napi_env env, napi_callback_info info)
{
    size_t argc = 9;
    napi_value argv[9];
    NAPI_OK_OR_THROW(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL), "bad arguments");
    if (argc < 6) { napi_throw_type_error(env, NULL, "processBatchAsync(handle, in, out, nch, nhops, pitch[, pitchStride, channelsPerStream])"); return NULL; }
    pv_slot *slot = unwrap(env, argv[0]);
    if (!slot) return NULL;
    napi_typedarray_type ty;
    size_t nin = 0, nout = 0, npitch = 0, off;
    void *din = NULL, *dout = NULL, *dp = NULL;
    napi_value ab;
    if (napi_get_typedarray_info(env, argv[1], &ty, &nin, &din, &ab, &off) != napi_ok || ty != napi_float32_array ||
        napi_get_typedarray_info(env, argv[2], &ty, &nout, &dout, &ab, &off) != napi_ok || ty != napi_float32_array ||
        napi_get_typedarray_info(env, argv[5], &ty, &npitch, &dp, &ab, &off) != napi_ok || ty != napi_float32_array) {
        napi_throw_type_error(env, NULL, "in, out and pitch must be Float32Array");
        return NULL;
    }
    int32_t nch = 0, nhops = 0, pstride = 0, cps = 1;
    int64_t chs = 0;
    napi_get_value_int32(env, argv[3], &nch);
    napi_get_value_int32(env, argv[4], &nhops);
    if (argc > 6) napi_get_value_int32(env, argv[6], &pstride);
    if (argc > 7) napi_get_value_int32(env, argv[7], &cps);
    if (argc > 8) napi_get_value_int64(env, argv[8], &chs);                  /* channel stride in floats; 0 / omitted: nhops * hopSize */
    if (chs <= 0) chs = (int64_t)nhops * slot->hop;
    if (nch <= 0 || nhops <= 0 || chs < (int64_t)nhops * slot->hop) { napi_throw_range_error(env, NULL, "buffer sizes do not match nch*nhops*hopSize"); return NULL; }
    const size_t need = (size_t)(nch - 1) * (size_t)chs + (size_t)nhops * (size_t)slot->hop;
    const size_t rows = pstride ? (size_t)((nch + (cps > 0 ? cps : 1) - 1) / (cps > 0 ? cps : 1)) : 1;
    if (nch <= 0 || nhops <= 0 || nin < need || nout < need || npitch < (pstride ? (rows - 1) * (size_t)pstride + (size_t)nhops : (size_t)nhops)) {
        napi_throw_range_error(env, NULL, "buffer sizes do not match nch*nhops*hopSize");
        return NULL;
    }
    batch_job *j = (batch_job *)calloc(1, sizeof(batch_job));
    if (!j) { napi_throw_error(env, NULL, "out of memory"); return NULL; }
    j->slot = slot; j->in = (const float *)din; j->out = (float *)dout; j->pitch = (const float *)dp;
    j->nch = nch; j->nhops = nhops; j->pstride = pstride; j->cps = cps; j->chs = chs;
    napi_value promise, name;
    if (napi_create_promise(env, &j->deferred, &promise) != napi_ok) { free(j); napi_throw_error(env, NULL, "napi_create_promise failed"); return NULL; }
    napi_create_reference(env, argv[1], 1, &j->refs[0]);
    napi_create_reference(env, argv[2], 1, &j->refs[1]);
    napi_create_reference(env, argv[5], 1, &j->refs[2]);
    napi_create_string_utf8(env, "phaze.processBatchAsync", NAPI_AUTO_LENGTH, &name);
    if (napi_create_async_work(env, NULL, name, batch_execute, batch_complete, j, &j->work) != napi_ok || napi_queue_async_work(env, j->work) != napi_ok) {
        for (int i = 0; i < 3; i++) napi_delete_reference(env, j->refs[i]);
        free(j);
        napi_throw_error(env, NULL, "could not queue the asynchronous batch");
        return NULL;
    }
    slot->busy = 1;
    return promise;
}

/* batchWindow(handle) -> [beginMs, endMs]: when the last processBatchAsync of this handle ran on its libuv worker thread (CLOCK_MONOTONIC).  A host
 * that drives G shards checks with it that G batches were in flight together (UV_THREADPOOL_SIZE >= G took effect), tools/bench_node_sharded.js. */
static napi_value js_batch_window(napi_env env, napi_callback_info info)
{
    size_t argc = 1;
    napi_value argv[1];
    NAPI_OK_OR_THROW(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL), "bad arguments");
    if (argc < 1) { napi_throw_type_error(env, NULL, "batchWindow(handle)"); return NULL; }
    pv_slot *slot = unwrap(env, argv[0]);
    if (!slot) return NULL;
    napi_value arr, a, b;
    napi_create_array_with_length(env, 2, &arr);
    napi_create_double(env, slot->t_begin_ms, &a);
    napi_create_double(env, slot->t_end_ms, &b);
    napi_set_element(env, arr, 0, a);
    napi_set_element(env, arr, 1, b);
    return arr;
}

/* allocPinned(nFloats) -> Float32Array in page-locked host memory (pv_host_alloc; freed when the ArrayBuffer is collected).  processBatch /
 * processBatchAsync on such arrays are pipelined (DMA to and from the device at the same time, include/phaze_amd.h); a host that owns its audio
 * buffers (what a caller of OLAProcessor.process does, ola-processor.js:159-171) allocates them here and writes its streams in place. */
static void finalize_pinned(napi_env env, void *data, void *hint)
{
    /* hint = the block's size: V8 is told that the external memory is gone (it was told about it in allocPinned: page-locked memory is scarce, and an engine that
     * does not see the bytes behind an external ArrayBuffer has no reason to collect it -- create / close cycles of sharded hosts pinned 2 x 268 MB each) */
    int64_t after = 0;
    if (env) (void)napi_adjust_external_memory(env, -(int64_t)(uintptr_t)hint, &after);
    pv_host_free(data);
}

static napi_value js_alloc_pinned(napi_env env, napi_callback_info info)
{
    size_t argc = 1;
    napi_value argv[1];
    NAPI_OK_OR_THROW(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL), "bad arguments");
    double nf = 0;
    if (argc < 1 || napi_get_value_double(env, argv[0], &nf) != napi_ok || !(nf >= 1) || nf > 4e12) { napi_throw_range_error(env, NULL, "allocPinned(nFloats): nFloats >= 1"); return NULL; }
    const size_t n = (size_t)nf;
    void *p = NULL;
    const int rc = pv_host_alloc(n * sizeof(float), &p);
    if (rc != PV_OK) return throw_status(env, NULL, rc);
    memset(p, 0, n * sizeof(float));
    napi_value ab, ta;
    if (napi_create_external_arraybuffer(env, p, n * sizeof(float), finalize_pinned, (void *)(uintptr_t)(n * sizeof(float)), &ab) != napi_ok) { pv_host_free(p); napi_throw_error(env, NULL, "napi_create_external_arraybuffer failed"); return NULL; }
    { int64_t after = 0; (void)napi_adjust_external_memory(env, (int64_t)(n * sizeof(float)), &after); }
    NAPI_OK_OR_THROW(env, napi_create_typedarray(env, napi_float32_array, n, ab, 0, &ta), "napi_create_typedarray failed");
    return ta;
}

/* exportState(handle, channel) -> {hist: Float32Array(N - hop), acc: Float32Array(N - hop), timeCursor}
 * importState(handle, channel, hist | null, acc | null[, timeCursor])            (pv_export_state / pv_import_state:
 * what the reference keeps per channel between process() calls, ola-processor.js:59,77, phase-vocoder.js:31) */
static napi_value js_export_state(napi_env env, napi_callback_info info)
{
    size_t argc = 2;
    napi_value argv[2];
    NAPI_OK_OR_THROW(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL), "bad arguments");
    if (argc < 2) { napi_throw_type_error(env, NULL, "exportState(handle, channel)"); return NULL; }
    pv_slot *slot = unwrap(env, argv[0]);
    if (!slot) return NULL;
    int32_t ch = 0;
    napi_get_value_int32(env, argv[1], &ch);
    const size_t L = (size_t)(slot->fft - slot->hop);
    napi_value abh, aba, hist, acc, o, tc;
    void *ph = NULL, *pa = NULL;
    NAPI_OK_OR_THROW(env, napi_create_arraybuffer(env, L * sizeof(float), &ph, &abh), "allocation failed");
    NAPI_OK_OR_THROW(env, napi_create_arraybuffer(env, L * sizeof(float), &pa, &aba), "allocation failed");
    int64_t t = 0;
    const int rc = pv_export_state(slot->h, ch, (float *)ph, (float *)pa, &t);
    if (rc != PV_OK) return throw_status(env, slot->h, rc);
    napi_create_typedarray(env, napi_float32_array, L, abh, 0, &hist);
    napi_create_typedarray(env, napi_float32_array, L, aba, 0, &acc);
    napi_create_object(env, &o);
    napi_set_named_property(env, o, "hist", hist);
    napi_set_named_property(env, o, "acc", acc);
    napi_create_int64(env, t, &tc);
    napi_set_named_property(env, o, "timeCursor", tc);
    return o;
}

static napi_value js_import_state(napi_env env, napi_callback_info info)
{
    size_t argc = 5;
    napi_value argv[5];
    NAPI_OK_OR_THROW(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL), "bad arguments");
    if (argc < 4) { napi_throw_type_error(env, NULL, "importState(handle, channel, hist, acc[, timeCursor])"); return NULL; }
    pv_slot *slot = unwrap(env, argv[0]);
    if (!slot) return NULL;
    int32_t ch = 0;
    napi_get_value_int32(env, argv[1], &ch);
    const size_t L = (size_t)(slot->fft - slot->hop);
    const float *ptr[2] = {NULL, NULL};
    for (int i = 0; i < 2; i++) {
        napi_valuetype vt;
        napi_typeof(env, argv[2 + i], &vt);
        if (vt == napi_null || vt == napi_undefined) continue;
        napi_typedarray_type ty;
        size_t n = 0, off;
        void *d = NULL;
        napi_value ab;
        if (napi_get_typedarray_info(env, argv[2 + i], &ty, &n, &d, &ab, &off) != napi_ok || ty != napi_float32_array || n != L) {
            napi_throw_range_error(env, NULL, "hist / acc must be Float32Array(fftSize - hopSize) or null");
            return NULL;
        }
        ptr[i] = (const float *)d;
    }
    int64_t t = -1;
    if (argc > 4) {
        napi_valuetype vt;
        napi_typeof(env, argv[4], &vt);
        if (vt == napi_number) napi_get_value_int64(env, argv[4], &t);
    }
    const int rc = pv_import_state(slot->h, ch, ptr[0], ptr[1], t);
    if (rc != PV_OK) return throw_status(env, slot->h, rc);
    return NULL;
}

static napi_value js_device_count(napi_env env, napi_callback_info info)
{
    (void)info;
    int32_t n = 0;
    pv_device_count(&n);
    napi_value v;
    napi_create_int32(env, n, &v);
    return v;
}

static napi_value js_reset(napi_env env, napi_callback_info info)
{
    size_t argc = 4;
    napi_value argv[4];
    NAPI_OK_OR_THROW(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL), "bad arguments");
    pv_slot *slot = unwrap(env, argv[0]);
    if (!slot) return NULL;
    int rc;
    if (argc >= 3) {
        int32_t first = 0, count = 0, parts = PV_STATE_HISTORY | PV_STATE_ACCUMULATOR;
        napi_get_value_int32(env, argv[1], &first);
        napi_get_value_int32(env, argv[2], &count);
        if (argc >= 4) napi_get_value_int32(env, argv[3], &parts);           /* 1: input history (ola-processor.js:54-71), 2: overlap-add sums (:73-88) */
        rc = pv_reset_channels_part(slot->h, first, count, parts);
    } else {
        rc = pv_reset(slot->h);
    }
    if (rc != PV_OK) return throw_status(env, slot->h, rc);
    return NULL;
}

static napi_value js_time_cursor(napi_env env, napi_callback_info info)
{
    size_t argc = 2;
    napi_value argv[2];
    NAPI_OK_OR_THROW(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL), "bad arguments");
    pv_slot *slot = unwrap(env, argv[0]);
    if (!slot) return NULL;
    if (argc >= 2) {
        int64_t v = 0;
        napi_get_value_int64(env, argv[1], &v);
        const int rc = pv_set_time_cursor(slot->h, v);
        if (rc != PV_OK) return throw_status(env, slot->h, rc);
    }
    int64_t t = 0;
    pv_get_time_cursor(slot->h, &t);
    napi_value out;
    napi_create_int64(env, t, &out);
    return out;
}

static napi_value js_info(napi_env env, napi_callback_info info)
{
    size_t argc = 1;
    napi_value argv[1];
    NAPI_OK_OR_THROW(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL), "bad arguments");
    pv_slot *slot = unwrap(env, argv[0]);
    if (!slot) return NULL;
    pv_info inf;
    pv_get_info(slot->h, &inf);
    napi_value o, v;
    napi_create_object(env, &o);
#define SETI(name, val) do { napi_create_int32(env, (val), &v); napi_set_named_property(env, o, name, v); } while (0)
    SETI("fftSize", inf.fft_size); SETI("hopSize", inf.hop_size); SETI("overlaps", inf.overlaps);
    SETI("maxChannels", inf.max_channels); SETI("maxHops", inf.max_hops);
    SETI("threadsPerWorkgroup", inf.threads_per_workgroup); SETI("ldsBytesPerWorkgroup", inf.lds_bytes_per_workgroup);
    SETI("framesPerChunk", inf.frames_per_chunk); SETI("computeUnits", inf.compute_units); SETI("deviceId", inf.device_id);
#undef SETI
    napi_create_string_utf8(env, inf.device_name, NAPI_AUTO_LENGTH, &v);
    napi_set_named_property(env, o, "deviceName", v);
    napi_create_string_utf8(env, inf.kernel_name, NAPI_AUTO_LENGTH, &v);
    napi_set_named_property(env, o, "kernelName", v);
    return o;
}

/* forwardStats(handle[, reset]) -> {frames, fallbacks}   (pv_forward_stats, round 5: frames whose forward transform ran fp32-first, and how many of them re-ran it in
 * fp64 because a peak decision -- phase-vocoder.js:95-116 -- lay within the fp32 transform's error; what a host looks at before it chooses PV_FLAG_FP64_FORWARD) */
static napi_value js_forward_stats(napi_env env, napi_callback_info info)
{
    size_t argc = 2;
    napi_value argv[2];
    NAPI_OK_OR_THROW(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL), "bad arguments");
    pv_slot *slot = unwrap(env, argv[0]);
    if (!slot) return NULL;
    bool reset = false;
    if (argc >= 2) napi_get_value_bool(env, argv[1], &reset);
    uint64_t frames = 0, fallbacks = 0;
    const int rc = pv_forward_stats(slot->h, &frames, &fallbacks, reset ? 1 : 0);
    if (rc != PV_OK) return throw_status(env, slot->h, rc);
    napi_value o, a, b;
    NAPI_OK_OR_THROW(env, napi_create_object(env, &o), "napi_create_object failed");
    napi_create_double(env, (double)frames, &a);
    napi_create_double(env, (double)fallbacks, &b);
    napi_set_named_property(env, o, "frames", a);
    napi_set_named_property(env, o, "fallbacks", b);
    return o;
}

static napi_value init(napi_env env, napi_value exports)
{
    const napi_property_descriptor props[] = {
        {"create", NULL, js_create, NULL, NULL, NULL, napi_enumerable, NULL},
        {"destroy", NULL, js_destroy, NULL, NULL, NULL, napi_enumerable, NULL},
        {"process", NULL, js_process, NULL, NULL, NULL, napi_enumerable, NULL},
        {"processBatch", NULL, js_process_batch, NULL, NULL, NULL, napi_enumerable, NULL},
        {"processBegin", NULL, js_process_begin, NULL, NULL, NULL, napi_enumerable, NULL},
        {"processEnd", NULL, js_process_end, NULL, NULL, NULL, napi_enumerable, NULL},
        {"processBatchAsync", NULL, js_process_batch_async, NULL, NULL, NULL, napi_enumerable, NULL},
        {"allocPinned", NULL, js_alloc_pinned, NULL, NULL, NULL, napi_enumerable, NULL},
        {"batchWindow", NULL, js_batch_window, NULL, NULL, NULL, napi_enumerable, NULL},
        {"exportState", NULL, js_export_state, NULL, NULL, NULL, napi_enumerable, NULL},
        {"importState", NULL, js_import_state, NULL, NULL, NULL, napi_enumerable, NULL},
        {"deviceCount", NULL, js_device_count, NULL, NULL, NULL, napi_enumerable, NULL},
        {"reset", NULL, js_reset, NULL, NULL, NULL, napi_enumerable, NULL},
        {"timeCursor", NULL, js_time_cursor, NULL, NULL, NULL, napi_enumerable, NULL},
        {"info", NULL, js_info, NULL, NULL, NULL, napi_enumerable, NULL},
        {"forwardStats", NULL, js_forward_stats, NULL, NULL, NULL, napi_enumerable, NULL},
    };
    napi_define_properties(env, exports, sizeof props / sizeof props[0], props);
    return exports;
}

NAPI_MODULE(NODE_GYP_MODULE_NAME, init)
