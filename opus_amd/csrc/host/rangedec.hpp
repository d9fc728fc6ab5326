// Host range decoder of the opus_amd engine (product code).
//
// Replaces the reference's RangeDecoder + CeltOnly trait (lu-zero/opus src/entropy.rs:264-543).  Written
// from RFC 6716 section 4.1 in the "val/rng/rem" form (difference-to-top coding with a one-byte
// carry), i.e. deliberately NOT the same formulation as the CPU oracle, so that side-info parity tests
// compare two independent decoders.
#pragma once
#include <cstdint>
#include <cstring>
#include "../rfc6716_tables.h"

namespace og {

static inline int ilog(uint32_t v) { return v ? 32 - __builtin_clz(v) : 0; }   // maths.rs:8-18

class RangeDec {
public:
    RangeDec(const uint8_t *buf, uint32_t len) { init(buf, len); }

    void init(const uint8_t *buf, uint32_t len)   // entropy.rs:301-316
    {
        buf_ = buf; storage_ = len; offs_ = 0; end_offs_ = 0; end_window_ = 0; nend_bits_ = 0;
        nbits_total_ = 9;
        rng_ = 128;
        rem_ = read_byte();
        val_ = rng_ - 1 - (rem_ >> 1);
        normalize();
    }

    uint32_t rng() const { return rng_; }
    int storage_bits() const { return (int)storage_ * 8; }
    void set_storage(uint32_t bytes) { storage_ = bytes; }               // decoder.rs:107 (_size), RFC 4.5.1
    int tell() const { return nbits_total_ - ilog(rng_); }               // entropy.rs:375
    int tell_frac() const                                                // entropy.rs:380-392
    {
        // The reference squares the top 16 bits of rng three times and collects one bit each time.  That recurrence is monotonic
        // in r = rng >> (l - 16), so its three bits are the number of thresholds at or below r (the seven values of r in
        // [2^15, 2^16) where the recurrence's result steps; tests/test_host_parse.py derives them from the loop again).
        const int l = ilog(rng_);
        const uint32_t r = rng_ >> (l - 16);
        const int frac = (r >= 35734u) + (r >= 38968u) + (r >= 42496u) + (r >= 46341u) + (r >= 50536u) + (r >= 55110u) + (r >= 60098u);
        return nbits_total_ * 8 - (l * 8 + frac);
    }
    void to_end() { nbits_total_ += storage_bits() - tell(); }           // entropy.rs:540

    // cumulative-frequency interface (RFC 4.1.2 / 4.1.3)
    uint32_t decode(uint32_t ft)
    {
        ext_ = rng_ / ft;
        uint32_t s = val_ / ext_;
        return ft - (s + 1 < ft ? s + 1 : ft);
    }
    void update(uint32_t fl, uint32_t fh, uint32_t ft)                   // entropy.rs:318-330
    {
        uint32_t s = ext_ * (ft - fh);
        val_ -= s;
        rng_ = fl > 0 ? ext_ * (fh - fl) : rng_ - s;
        normalize();
    }
    int bit_logp(unsigned logp)                                          // entropy.rs:339-355
    {
        uint32_t s = rng_ >> logp;
        int ret = val_ < s;
        if (!ret) val_ -= s;
        rng_ = ret ? s : rng_ - s;
        normalize();
        return ret;
    }
    int icdf(int pool_off) { return icdf_table(RFC_ICDF_POOL + pool_off); }   // entropy.rs:357-372
    int icdf_table(const uint16_t *t)                                    // the same on a table outside the pool: {total, count, cumulative bounds ...}
    {
        uint32_t ft = t[0];
        uint32_t fs = decode(ft);
        int k = 0;
        while (t[2 + k] <= fs) k++;
        update(k ? t[1 + k] : 0, t[2 + k], ft);
        return k;
    }
    uint32_t bits(unsigned n)                                            // entropy.rs:422-425
    {
        if (n == 0) return 0;
        while (nend_bits_ < (int)n) {
            uint64_t b = end_offs_ < storage_ ? buf_[storage_ - ++end_offs_] : 0;
            end_window_ |= b << nend_bits_;
            nend_bits_ += 8;
        }
        uint32_t r = (uint32_t)(end_window_ & ((1ull << n) - 1));
        end_window_ >>= n;
        nend_bits_ -= n;
        nbits_total_ += n;
        return r;
    }
    uint32_t uint(uint32_t ft)                                           // entropy.rs:427-445 (+ RFC clamp)
    {
        ft--;
        int ftb = ilog(ft);
        if (ftb > 8) {
            ftb -= 8;
            uint32_t top = (ft >> ftb) + 1;
            uint32_t s = decode(top);
            update(s, s + 1, top);
            uint32_t t = (s << ftb) | bits(ftb);
            return t <= ft ? t : ft;
        }
        ft++;
        uint32_t s = decode(ft);
        update(s, s + 1, ft);
        return s;
    }
    int laplace(unsigned fs, int decay)                                  // entropy.rs:448-485, RFC 4.3.2.1
    {
        int val = 0;
        uint32_t fm = decode_bin15();
        uint32_t fl = 0;
        if (fm >= fs) {
            val++;
            fl = fs;
            fs = ((32768 - 32 - fs) * (uint32_t)(16384 - decay) >> 15) + 1;   // first value's frequency, minp = 1
            while (fs > 1 && fm >= fl + 2 * fs) {
                fs *= 2;
                fl += fs;
                fs = ((fs - 2) * (uint32_t)decay) >> 15;
                fs += 1;
                val++;
            }
            if (fs <= 1) {
                int di = (int)((fm - fl) >> 1);
                val += di;
                fl += 2 * di;
            }
            if (fm < fl + fs) val = -val;
            else fl += fs;
        }
        uint32_t fh = fl + fs < 32768 ? fl + fs : 32768;
        update(fl, fh, 32768);
        return val;
    }

private:
    uint32_t decode_bin15()
    {
        ext_ = rng_ >> 15;
        uint32_t s = val_ / ext_;
        return 32768u - (s + 1 < 32768u ? s + 1 : 32768u);
    }
    uint32_t read_byte() { return offs_ < storage_ ? buf_[offs_++] : 0; }
    void normalize()                                                     // entropy.rs:290-299
    {
        while (rng_ <= (1u << 23)) {
            nbits_total_ += 8;
            rng_ <<= 8;
            uint32_t sym = rem_;
            rem_ = read_byte();
            sym = ((sym << 8) | rem_) >> 1;
            val_ = ((val_ << 8) + (255 & ~sym)) & 0x7FFFFFFFu;
        }
    }

    const uint8_t *buf_;
    uint32_t storage_, offs_, end_offs_;
    uint64_t end_window_;
    int nend_bits_, nbits_total_;
    uint32_t rng_, val_, ext_, rem_;
};

}  // namespace og
