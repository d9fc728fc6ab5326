// Host-side construction of the CELT kernels' constant tables (uploaded once per device).
#include <cmath>
#include "celt_tables.hpp"
#include "../rfc6716_tables.h"

void og_fill_celt_tables(OgCeltTables *t)
{
    const double PI = 3.14159265358979323846;
    for (int k = 0; k < 480; k++) {
        double a = -2.0 * PI * k / 480.0;
        t->fft_w[k].r = (float)std::cos(a);
        t->fft_w[k].i = (float)std::sin(a);
    }
    for (int n4 = 60; n4 <= 480; n4 *= 2) {            // block size n2 = 2*n4, transform length N = 4*n4
        for (int i = 0; i < n4; i++) {
            double th = 2.0 * PI * (i + 0.125) / (4.0 * n4);
            t->tw[n4 - 60 + i].r = (float)std::cos(th);
            t->tw[n4 - 60 + i].i = (float)std::sin(th);
        }
    }
    for (int i = 0; i < 120; i++) {                    // RFC 6716 4.3.7: sin(pi/2 sin^2(pi (i+1/2) / 2L)), L = 120
        double s = std::sin(0.5 * PI * (i + 0.5) / 120);
        t->window[i] = (float)std::sin(0.5 * PI * s * s);
    }
    t->lcg_a[0] = 1; t->lcg_c[0] = 0;                  // celt/decoder.rs:1595-1599 composed j times
    for (int j = 0; j + 1 < 184; j++) {
        t->lcg_a[j + 1] = 1664525u * t->lcg_a[j];
        t->lcg_c[j + 1] = 1664525u * t->lcg_c[j] + 1013904223u;
    }
    static const float emeans[21] = {   // RFC 6716 4.3.2 (quant_bands.c eMeans); not in the reference
        6.437500f, 6.250000f, 5.750000f, 5.312500f, 5.062500f, 4.812500f, 4.500000f, 4.375000f, 4.875000f,
        4.687500f, 4.562500f, 4.437500f, 4.875000f, 4.625000f, 4.312500f, 4.500000f, 4.375000f, 4.625000f,
        4.750000f, 4.437500f, 3.750000f};
    OgCeltConst &k = t->k;
    for (int i = 0; i < 24; i++) { k.emeans[i] = i < 21 ? emeans[i] : 0.f; k.ebands[i] = i < 22 ? CELT_EBANDS[i] : 0; k.width[i] = i < 21 ? CELT_BAND_WIDTH[i] : 0; }
    for (int i = 0; i < 3; i++) for (int j = 0; j < 4; j++) k.pf_taps[i][j] = j < 3 ? CELT_POSTFILTER_TAPS[i][j] : 0.f;
    for (int i = 0; i < 16; i++) { k.bit_interleave[i] = CELT_BIT_INTERLEAVE[i]; k.bit_deinterleave[i] = CELT_BIT_DEINTERLEAVE[i]; }
    for (int i = 0; i < 32; i++) k.hadamard[i] = i < 30 ? CELT_HADAMARD_ORDERY[i] : 0;
    for (int i = 0; i < 1272; i++) t->pvq_u[i] = CELT_PVQ_U[i];
    for (int i = 0; i < 16; i++) t->pvq_row[i] = i < 15 ? CELT_PVQ_U_ROW[i] : 0;
    for (int lo = 0; lo < 15; lo++)
        for (int hi = 0; hi < 177; hi++) {
            const int idx = CELT_PVQ_U_ROW[lo] + hi, end = lo < 14 ? CELT_PVQ_U_ROW[lo + 1] + lo + 1 : 1272;
            t->pvq_u2[lo * 177 + hi] = (hi >= lo && idx < end) ? CELT_PVQ_U[idx] : 0xFFFFFFFFu;
        }
    t->pvq_u2[15 * 177] = 0;                             // (padding: the kernel stages the table in 16-byte pieces)
    for (int lm = 0; lm < 4; lm++)
        for (int j = 0; j < 960; j++) {
            int b = 255;
            for (int i = 0; i < 21; i++)
                if (j >= (CELT_EBANDS[i] << lm) && j < (CELT_EBANDS[i + 1] << lm)) b = i;
            t->bin2band[lm][j] = (uint8_t)b;
            t->bin2sq[lm][j] = b < 21 ? sqrtf((float)(CELT_BAND_WIDTH[b] << lm)) : 0.f;
        }
}
