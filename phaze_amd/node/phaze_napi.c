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

#include "../../include/phaze_amd.h"

#define MAX_CH 64

/* what a JS "handle" external points to: the C handle plus the constants process() needs per quantum (no pv_get_info on the hot call) */
typedef struct pv_slot {
    pv_handle *h;
    int32_t hop;
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
    return slot;
}

/* create({fftSize, hopSize, maxChannels, maxHops, deviceId, framesPerChunk, flags}) -> external */
static napi_value js_create(napi_env env, napi_callback_info info)
{
    size_t argc = 1;
    napi_value argv[1];
    NAPI_OK_OR_THROW(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL), "bad arguments");
    if (argc < 1) { napi_throw_type_error(env, NULL, "create(options) needs an options object"); return NULL; }
    pv_config cfg;
    memset(&cfg, 0, sizeof cfg);
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
    pv_slot *slot = (pv_slot *)malloc(sizeof(pv_slot));
    slot->h = h;
    slot->hop = cfg.hop_size;
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
        if (slot->h) { pv_destroy(slot->h); slot->h = NULL; }
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
    size_t argc = 8;
    napi_value argv[8];
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
    napi_get_value_int32(env, argv[3], &nch);
    napi_get_value_int32(env, argv[4], &nhops);
    if (argc > 6) napi_get_value_int32(env, argv[6], &pstride);
    if (argc > 7) napi_get_value_int32(env, argv[7], &cps);
    const size_t need = (size_t)nch * (size_t)nhops * (size_t)slot->hop;
    const size_t rows = pstride ? (size_t)((nch + (cps > 0 ? cps : 1) - 1) / (cps > 0 ? cps : 1)) : 1;
    if (nch <= 0 || nhops <= 0 || nin < need || nout < need || npitch < (pstride ? (rows - 1) * (size_t)pstride + (size_t)nhops : (size_t)nhops)) {
        napi_throw_range_error(env, NULL, "buffer sizes do not match nch*nhops*hopSize");
        return NULL;
    }
    const int rc = pv_process_batch(slot->h, (const float *)din, (float *)dout, nch, nhops, (int64_t)nhops * slot->hop, (const float *)dp, pstride, cps);
    if (rc != PV_OK) return throw_status(env, slot->h, rc);
    napi_value t;
    napi_get_boolean(env, true, &t);
    return t;
}

static napi_value js_reset(napi_env env, napi_callback_info info)
{
    size_t argc = 3;
    napi_value argv[3];
    NAPI_OK_OR_THROW(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL), "bad arguments");
    pv_slot *slot = unwrap(env, argv[0]);
    if (!slot) return NULL;
    int rc;
    if (argc >= 3) {
        int32_t first = 0, count = 0;
        napi_get_value_int32(env, argv[1], &first);
        napi_get_value_int32(env, argv[2], &count);
        rc = pv_reset_channels(slot->h, first, count);     /* ola-processor.js:54-88 */
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

static napi_value init(napi_env env, napi_value exports)
{
    const napi_property_descriptor props[] = {
        {"create", NULL, js_create, NULL, NULL, NULL, napi_enumerable, NULL},
        {"destroy", NULL, js_destroy, NULL, NULL, NULL, napi_enumerable, NULL},
        {"process", NULL, js_process, NULL, NULL, NULL, napi_enumerable, NULL},
        {"processBatch", NULL, js_process_batch, NULL, NULL, NULL, napi_enumerable, NULL},
        {"reset", NULL, js_reset, NULL, NULL, NULL, napi_enumerable, NULL},
        {"timeCursor", NULL, js_time_cursor, NULL, NULL, NULL, napi_enumerable, NULL},
        {"info", NULL, js_info, NULL, NULL, NULL, napi_enumerable, NULL},
    };
    napi_define_properties(env, exports, sizeof props / sizeof props[0], props);
    return exports;
}

NAPI_MODULE(NODE_GYP_MODULE_NAME, init)
