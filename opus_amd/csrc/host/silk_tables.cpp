// Host-side construction of the SILK output stage's tables.
//
// Resampler (absent from the reference, non-normative in RFC 6716; the build's own design):
// zero-stuff by L = 48 / rate_kHz, then a 73-tap Kaiser (beta 8) windowed-sinc low-pass at 0.46 * fs_in with
// gain L; group delay 36 samples @ 48 kHz for every input rate.
#include <cmath>
#include "../hip/silk_synth.h"

static double bessel_i0(double x)
{
    double s = 1, t = 1;
    for (int k = 1; k < 40; k++) { t *= (x / (2 * k)) * (x / (2 * k)); s += t; }
    return s;
}

void og_fill_silk_tables(OgSilkTables *t)
{
    const double PI = 3.14159265358979323846;
    static const int rates[3] = {8, 12, 16};
    for (int r = 0; r < 3; r++) {
        const int L = 48 / rates[r];
        const double fc = 0.46 * rates[r] / 48.0;      // cycles per 48 kHz sample
        for (int n = 0; n < OG_RS_TAPS + 3; n++) {
            if (n >= OG_RS_TAPS) { t->rs_taps[r][n] = 0.f; continue; }
            const double x = n - OG_RS_HALF;
            const double sinc = x == 0 ? 2 * fc : std::sin(2 * PI * fc * x) / (PI * x);
            const double w = bessel_i0(8.0 * std::sqrt(1 - (x / OG_RS_HALF) * (x / OG_RS_HALF))) / bessel_i0(8.0);
            t->rs_taps[r][n] = (float)(L * sinc * w);
        }
    }
}
