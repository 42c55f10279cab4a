/* pv_stream.c -- the C ABI from plain C99: what a host that is neither Python nor Node has to write.
 *
 * One stereo stream is pitch-shifted twice: render quantum by render quantum through pv_process() -- the reference's calling pattern,
 * /root/reference/src/ola-processor.js:159-171 -- and in one pv_process_batch() call; the two results must agree bit for bit (the library runs
 * one kernel per configuration).  Prints one line of JSON.
 *
 *   gcc -std=c99 -O2 -Wall -Wextra -pedantic -I include examples/pv_stream.c -o build/pv_stream \
 *       -L phaze_amd/lib -lphaze_amd -Wl,-rpath,$PWD/phaze_amd/lib -Wl,-rpath,/opt/rocm/lib -lm
 *   build/pv_stream [fftSize hopSize pitchFactor hops]
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "phaze_amd.h"

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static int die(pv_handle *h, const char *what, int rc)
{
    fprintf(stderr, "%s: %s (%s)\n", what, pv_status_string(rc), h ? pv_last_error(h) : "no handle");
    if (h) pv_destroy(h);
    return 1;
}

int main(int argc, char **argv)
{
    const int fft = argc > 1 ? atoi(argv[1]) : 2048, hop = argc > 2 ? atoi(argv[2]) : 128;
    const float pitch = argc > 3 ? (float)atof(argv[3]) : 1.5f;
    const int hops = argc > 4 ? atoi(argv[4]) : 512, nch = 2;
    const size_t n = (size_t)hops * (size_t)hop;
    float *x = (float *)malloc(sizeof(float) * n * nch), *ys = (float *)calloc(n * nch, sizeof(float)), *yb = (float *)calloc(n * nch, sizeof(float));
    float *pf = (float *)malloc(sizeof(float) * (size_t)hops);
    unsigned s = 12345u;
    int c, m, rc;
    size_t i;
    pv_config cfg = PV_CONFIG_INIT;
    pv_handle *h = NULL;
    pv_info info;
    double t0, t_stream, t_batch;

    if (!x || !ys || !yb || !pf) return 1;
    for (c = 0; c < nch; c++)
        for (i = 0; i < n; i++) {                                       /* a tone per channel over a noise floor (the reference needs one: SURVEY K12) */
            s = s * 1664525u + 1013904223u;
            x[(size_t)c * n + i] = 0.25f * (float)sin(0.02 * (double)(c + 1) * (double)i) + ((float)(s >> 8) / 8388608.0f - 1.0f) / 256.0f;
        }
    for (m = 0; m < hops; m++) pf[m] = pitch;

    cfg.fft_size = fft; cfg.hop_size = hop; cfg.max_channels = nch; cfg.max_hops = hops; cfg.device_id = 0;
    rc = pv_create(&cfg, &h);
    if (rc != PV_OK) return die(h, "pv_create", rc);

    /* the reference's pattern: one process() per render quantum, host-owned blocks that are only valid during the call */
    t0 = now_s();
    for (m = 0; m < hops; m++) {
        const float *in[2];
        float *out[2];
        for (c = 0; c < nch; c++) { in[c] = x + (size_t)c * n + (size_t)m * (size_t)hop; out[c] = ys + (size_t)c * n + (size_t)m * (size_t)hop; }
        rc = pv_process(h, in, out, nch, hop, pitch);
        if (rc != PV_OK) return die(h, "pv_process", rc);
    }
    t_stream = now_s() - t0;

    rc = pv_reset(h);
    if (rc != PV_OK) return die(h, "pv_reset", rc);
    t0 = now_s();
    rc = pv_process_batch(h, x, yb, nch, hops, (int64_t)n, pf, 0, 1);
    if (rc != PV_OK) return die(h, "pv_process_batch", rc);
    t_batch = now_s() - t0;

    rc = pv_get_info(h, &info);
    if (rc != PV_OK) return die(h, "pv_get_info", rc);
    {
        const int same = memcmp(ys, yb, sizeof(float) * n * nch) == 0;
        double e = 0.0;
        for (i = 0; i < n * nch; i++) e += (double)yb[i] * (double)yb[i];
        printf("{\"fft\": %d, \"hop\": %d, \"channels\": %d, \"hops\": %d, \"pitchFactor\": %g, \"kernel\": \"%s\", \"device\": \"%s\", "
               "\"stream_equals_batch\": %s, \"output_rms\": %.6g, \"stream_us_per_quantum\": %.2f, \"batch_frames_per_s\": %.4g}\n",
               fft, hop, nch, hops, (double)pitch, info.kernel_name, info.device_name, same ? "true" : "false", sqrt(e / (double)(n * nch)),
               1e6 * t_stream / hops, (double)hops * nch / t_batch);
        pv_destroy(h);
        free(x); free(ys); free(yb); free(pf);
        return same ? 0 : 2;
    }
}
