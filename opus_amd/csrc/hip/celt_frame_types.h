// Constants and table layout shared by the CELT kernel and the host code that fills its tables.
#pragma once
#include <stdint.h>

#ifndef OG_WG
#define OG_WG 256                     // 4 wavefronts per frame
#endif
#define OG_WGT 128                    // tail kernel: one wavefront per channel
#define OG_HIST 1024                  // comb filter reaches back period + 2 <= 1024 samples
#define OG_OUTLEN 2048                // per channel: OG_HIST history + 960 frame + 60 tail (+ 4)
#define OG_MAXLEAF 704                // 21 bands x 2 channels x 16 partitions, rounded up

struct OgCf { float r, i; };

struct OgCeltConst {                  // small RFC 6716 tables the band walk indexes with uniform indices: kept in LDS
    float emeans[24];                 // 4.3.2 mean band energies (log2)
    float pf_taps[3][4];              // celt/decoder.rs:90-94
    uint8_t ebands[24];               // celt/decoder.rs:281-283 band edges (units of 2.5 ms bins)
    uint8_t width[24];
    uint8_t bit_interleave[16], bit_deinterleave[16];   // celt/decoder.rs:1700-1706
    uint8_t hadamard[32];             // celt/decoder.rs:468-470 ordery tables for strides 2,4,8,16
};

struct OgCeltTables {                 // constant tables, built once per device by og_fill_celt_tables()
    OgCf fft_w[480];                  // e^{-2 pi i t / 480}
    OgCf tw[900];                     // IMDCT pre/post twiddles for block sizes 120/240/480/960 at 0/60/180/420
    float window[120];                // RFC 6716 4.3.7 power-complementary window
    uint32_t lcg_a[184], lcg_c[184];  // seed after j LCG steps = lcg_a[j] * seed + lcg_c[j]
    uint32_t pvq_u[1272];             // U(N,K) rows (celt/decoder.rs:293-409)
    uint16_t pvq_row[16];             // row offsets (celt/decoder.rs:410-413)
    alignas(16) uint32_t pvq_u2[15 * 177 + 1];   // the same table unpacked: U(lo, hi) at [lo * 177 + hi] (no dependent row look-up); 16-byte pieces
    uint8_t bin2band[4][960];         // per LM: band of MDCT bin j (255 above band 20)
    float bin2sq[4][960];             // per LM: sqrt(width of bin j's band) -- lowband_out = sqrt(N0) x, celt/decoder.rs:2021-2028 (0 above band 20)
    OgCeltConst k;
};

