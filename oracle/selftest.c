/* selftest.c -- drives every oracle function on small inputs; built with
 * -fsanitize=address,undefined by tests/test_oracle_sanitize.py so that the
 * checker itself is known to stay in bounds (the reference's own FastFIR reads
 * and writes one element past its buffers, SURVEY A.3 Q2 -- the restatement must
 * not).  TEST INFRASTRUCTURE ONLY. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "fm_oracle.h"

int main(void)
{
    for (int mode = 0; mode < 4; mode++)
        for (int ch = 1; ch <= 2; ch++) {
            fmo_params p;
            if (fmo_mode_params(mode, 101, 13, 13, &p)) return 1;   /* 13 audio taps keeps modes 2/3 quick */
            fmo_pipeline *pl = fmo_pipeline_create(&p, ch);
            const size_t nb = (size_t)p.block_bytes;
            uint8_t *iq = (uint8_t *)malloc(nb);                      /* exact size: any overrun trips ASan */
            float *al = (float *)malloc(sizeof(float) * fmo_pipeline_n_audio(pl, nb));
            float *ar = (float *)malloc(sizeof(float) * fmo_pipeline_n_audio(pl, nb));
            float *dm = (float *)malloc(sizeof(float) * fmo_pipeline_n_if(pl, nb));
            for (int b = 0; b < 2; b++) {
                fmo_synth_fm_u8(iq, nb / 2, p.rf_Fs, 7, (uint64_t)b * nb / 2);
                size_t na = fmo_pipeline_process(pl, iq, nb, NULL, NULL, dm, al, ch == 2 ? ar : NULL);
                if (na != fmo_pipeline_n_audio(pl, nb)) return 2;
            }
            int16_t *pcm = (int16_t *)malloc(sizeof(int16_t) * fmo_pipeline_n_audio(pl, nb));
            fmo_pcm16(al, fmo_pipeline_n_audio(pl, nb), pcm, 1);
            fmo_pcm16(al, fmo_pipeline_n_audio(pl, nb), pcm, 0);
            free(pcm); free(dm); free(ar); free(al); free(iq);
            fmo_pipeline_destroy(pl);
        }
    /* function level, minimum legal sizes */
    {
        enum { T = 101, N = 100 };
        float h[T], x[N], st[T - 1], y[N + T - 1];
        fmo_impulse_response_lpf(240e3f, 16e3f, T, h);
        fmo_band_pass(240e3f, 22e3f, 54e3f, T, h);
        for (int i = 0; i < N; i++) x[i] = (float)i / N;
        memset(st, 0, sizeof(st));
        fmo_convolve_block_fast_fir(y, x, N, h, T, st, 5);   /* n == taps-1 */
        fmo_convolve_block_fir(y, x, N, h, T, st);
        fmo_convolve_fir(y, x, N, h, T);
        float xu[N * 3], xd[N];
        fmo_upsample(x, N, xu, 3);
        if (fmo_downsample(xd, x, N, 7) != 15) return 3;
        float pi = 0, pq = 0, d[N];
        fmo_fm_demod(d, x, x, N, &pi, &pq);
        float aps[50] = {0}, apo[N];
        fmo_all_pass(x, N, aps, 50, apo);
        float pst[6] = {0, 0, 1, 0, 1, 0}, nco[N + 1];
        fmo_fm_pll(x, N, nco, pst, 19e3f, 240e3f, 2.0f, 0.0f, 0.01f);
        enum { U = 4, TR = 101 * U };
        static float hr[TR], sr[TR - 1], yr[125 * U / 5];
        fmo_impulse_response_lpf(240e3f * U, 16e3f, TR, hr);
        memset(sr, 0, sizeof(sr));
        float xr[125];
        for (int i = 0; i < 125; i++) xr[i] = x[i % N];
        fmo_convolve_block_resample_fir(yr, xr, 125, hr, TR, sr, 5, U);   /* 125*4 = 500 >= taps-1 = 403, 500 % 5 == 0 */
    }
    puts("oracle selftest ok");
    return 0;
}
