/* Sanitizer driver for the CPU oracle (test infrastructure, SURVEY section 5: ASan / UBSan on the host side and the restatement).
 * Built by `make -C oracle asan` with -fsanitize=address,undefined and run by tests/test_sanitizers.py: every fft size class, f < 1 (the
 * above-Nyquist reads, the colliding +=), pathological pitch factors, channel-count changes and the paused branch -- any out-of-bounds
 * access, signed overflow or misaligned access aborts the run. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct pvo pvo_t;
pvo_t *pvo_create(int fft_size, int hop, int nch);
void pvo_destroy(pvo_t *s);
int pvo_process(pvo_t *s, const float *const *in, float *const *out, int nch, int paused, float pitch);
int pvo_process_planar(pvo_t *s, const float *in, float *out, int nch, int nhops, long ch_stride, const float *pitch);

static unsigned lcg = 4242u;
static float rnd(void) { lcg = lcg * 1664525u + 1013904223u; return ((float)(lcg >> 8) - 8388608.0f) / 8388608.0f; }

int main(void)
{
    static const int sizes[][2] = {{64, 16}, {128, 128}, {256, 64}, {1024, 256}, {2048, 128}, {2048, 512}, {4096, 1024}, {8192, 2048}};
    static const float pitches[] = {1.0f, 1.5f, 0.8f, 0.5f, 0.05f, 3.0f, 100.0f, 0.0f, -1.0f, NAN, INFINITY, -INFINITY};
    double checksum = 0.0;
    for (unsigned si = 0; si < sizeof sizes / sizeof sizes[0]; si++) {
        const int N = sizes[si][0], hop = sizes[si][1], T = 24, nch = 3;
        float *x = (float *)malloc(sizeof(float) * (size_t)nch * T * hop), *y = (float *)malloc(sizeof(float) * (size_t)nch * T * hop);
        float *p = (float *)malloc(sizeof(float) * T);
        if (!x || !y || !p) return 2;
        for (int i = 0; i < nch * T * hop; i++) x[i] = 0.3f * sinf(0.05f * (float)i) + 0.02f * rnd();
        for (unsigned pi = 0; pi < sizeof pitches / sizeof pitches[0]; pi++) {
            pvo_t *s = pvo_create(N, hop, nch);
            if (!s) return 3;
            for (int m = 0; m < T; m++) p[m] = (m % 5 == 4) ? pitches[(pi + 1) % (sizeof pitches / sizeof pitches[0])] : pitches[pi];
            if (pvo_process_planar(s, x, y, nch, T, (long)T * hop, p) != 1) return 4;   /* 1 = the reference's `return true` (ola:170) */
            for (int i = 0; i < nch * T * hop; i += 97) if (isfinite(y[i])) checksum += y[i];
            pvo_destroy(s);
        }
        {   /* streaming form: channel-count changes (ola:38-52) and the paused branch (ola:93-100) */
            pvo_t *s = pvo_create(N, hop, 1);
            const float *in[3];
            float *out[3];
            for (int m = 0; m < 12; m++) {
                const int c = 1 + (m / 4) % 3;
                for (int k = 0; k < c; k++) { in[k] = x + (size_t)k * T * hop + (size_t)m * hop; out[k] = y + (size_t)k * T * hop; }
                if (pvo_process(s, in, out, c, m % 6 == 5, 0.9f) != 1) return 5;
            }
            pvo_destroy(s);
        }
        free(x); free(y); free(p);
    }
    if (pvo_create(1000, 250, 1) != NULL || pvo_create(1, 1, 1) != NULL) return 6;     /* bundle:6-7 */
    printf("oracle sanitizer driver ok, checksum %.6f\n", checksum);
    return 0;
}
