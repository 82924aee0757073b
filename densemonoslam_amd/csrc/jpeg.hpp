// Baseline JPEG -> 8-bit RGB, host side.  The colour images of compressed .klg logs and compressed eflcm::Frame messages
// are JPEG (Logger2 / the LCM publishers encode with OpenCV = libjpeg); the reference decodes them with libjpeg's defaults
// through GUI/src/Tools/JPEGLoader.h:44-95 (jpeg_read_header / jpeg_start_decompress / jpeg_read_scanlines, then an R<->B
// swap per pixel).  libjpeg's headers are not in this image, so this is a from-scratch decoder of the same published
// algorithms, written to give libjpeg's BYTES for the streams those encoders produce (sequential DCT, Huffman, 8 bit,
// Y Cb Cr with 4:4:4, 4:2:2 or 4:2:0 sampling, optional restart intervals):
//   * dequantisation + the "slow" integer inverse DCT (Loeffler-Ligtenberg-Moschytz, 13-bit constants, 2 extra bits after
//     the column pass: libjpeg's JDCT_ISLOW, its default),
//   * "fancy" chroma upsampling, libjpeg's default (triangle filter: 3/4 nearer + 1/4 further sample, the two roundings
//     alternating; 9/16, 3/16, 3/16, 1/16 for 4:2:0; edge rows / columns replicated),
//   * the 16-bit fixed-point Y Cb Cr -> RGB tables (1.402, 1.772, 0.71414, 0.34414).
// Pinned by tests/golden/jpeg_cases.npz: streams encoded and decoded by Pillow's bundled libjpeg-turbo in this image
// (tests/golden/make_jpeg_golden.py); every pixel must match.  Progressive, arithmetic-coded, 12-bit, CMYK / greyscale
// streams are reported as unsupported (the reference's loader would mis-handle the last two itself: it assumes 3 bytes
// per pixel).
#pragma once
#include <stdint.h>
#include <string.h>

#include <vector>

namespace dms {
namespace jpeg {

struct Huff {
  // canonical Huffman table: codes of length l lie in [mincode[l], maxcode[l]]; valptr[l] indexes the first of them
  int mincode[17], maxcode[18], valptr[17];
  unsigned char vals[256];
  bool present = false;
  // 9-bit lookahead: (length << 8) | symbol, 0 = longer code
  unsigned short look[512];
};

struct Component {
  int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
  int dw = 0, dh = 0;          // downsampled size in samples (the "real" rows / columns)
  int bw = 0, bh = 0;          // size in blocks, padded to whole MCUs
  int pred = 0;
  std::vector<unsigned char> plane;  // bw * 8 x bh * 8 samples after the inverse DCT
};

struct Decoder {
  const unsigned char* p = nullptr;
  const unsigned char* end = nullptr;
  unsigned long long bitbuf = 0;
  int bits = 0;
  bool hit_marker = false;
  const char* err = nullptr;
  int unsupported = 0;

  unsigned short qt[4][64] = {};
  bool qt_present[4] = {};
  Huff dc[4], ac[4];
  Component comp[3];
  int ncomp = 0, width = 0, height = 0, hmax = 1, vmax = 1, restart_interval = 0;

  bool fail(const char* what) {
    if (!err) err = what;
    return false;
  }
  bool nosupport(const char* what) {
    unsupported = 1;
    return fail(what);
  }

  // ---- entropy-coded segment: bit reader with byte unstuffing; a marker ends the data (zero bits are supplied, as libjpeg does)
  void fill() {
    while (bits <= 56) {
      unsigned c = 0;
      if (!hit_marker && p < end) {
        c = *p++;
        if (c == 0xFF) {
          if (p < end && *p == 0x00) {
            ++p;
          } else {  // a marker: leave it for the caller
            --p;
            hit_marker = true;
            c = 0;
          }
        }
      } else {
        hit_marker = true;
      }
      bitbuf |= (unsigned long long)c << (56 - bits);
      bits += 8;
    }
  }
  int peek(int n) {
    if (bits < n) fill();
    return (int)(bitbuf >> (64 - n));
  }
  void skip(int n) {
    bitbuf <<= n;
    bits -= n;
  }
  int get(int n) {
    if (n == 0) return 0;
    const int v = peek(n);
    skip(n);
    return v;
  }
  static int extend(int v, int s) { return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }

  bool build(Huff& h, const unsigned char* counts, const unsigned char* vals, int nvals) {
    int code = 0, k = 0;
    memset(h.look, 0, sizeof(h.look));
    for (int l = 1; l <= 16; ++l) {
      h.valptr[l] = k;
      h.mincode[l] = code;
      for (int i = 0; i < counts[l - 1]; ++i) {
        if (k >= nvals || code >= (1 << l)) return fail("bad Huffman table");
        if (l <= 9) {
          const int base = code << (9 - l);
          for (int f = 0; f < (1 << (9 - l)); ++f) h.look[base + f] = (unsigned short)((l << 8) | vals[k]);
        }
        ++code;
        ++k;
      }
      h.maxcode[l] = counts[l - 1] ? code - 1 : -1;
      if (code > (1 << l)) return fail("bad Huffman table");
      code <<= 1;
    }
    h.maxcode[17] = 0x7fffffff;
    memcpy(h.vals, vals, (size_t)nvals);
    h.present = true;
    return true;
  }
  int decode_symbol(const Huff& h) {
    const int look = peek(9);
    const unsigned short e = h.look[look];
    if (e) {
      skip(e >> 8);
      return e & 0xff;
    }
    int code = peek(16);
    for (int l = 10; l <= 16; ++l) {
      const int c = code >> (16 - l);
      if (h.maxcode[l] >= 0 && c <= h.maxcode[l] && c >= h.mincode[l]) {
        skip(l);
        return h.vals[h.valptr[l] + c - h.mincode[l]];
      }
    }
    fail("bad Huffman code");
    skip(16);
    return 0;
  }

  // ---- one 8x8 block: coefficients (natural order) -> samples.  jidctint.c's arithmetic.
  static inline int64_t descale(int64_t x, int n) { return (x + ((int64_t)1 << (n - 1))) >> n; }
  static inline unsigned char range_limit(int64_t x) {  // libjpeg's table lookup with its 10-bit index mask
    const int i = (int)(x & 1023);
    if (i < 128) return (unsigned char)(128 + i);
    if (i < 512) return 255;
    if (i < 896) return 0;
    return (unsigned char)(i - 896);
  }
  static void idct(const short* coef, const unsigned short* q, unsigned char* out, int stride) {
    constexpr int CB = 13, P1 = 2;
    constexpr int64_t F0_298 = 2446, F0_390 = 3196, F0_541 = 4433, F0_765 = 6270, F0_899 = 7373, F1_175 = 9633, F1_501 = 12299,
                      F1_847 = 15137, F1_961 = 16069, F2_053 = 16819, F2_562 = 20995, F3_072 = 25172;
    int64_t ws[64];
    for (int c = 0; c < 8; ++c) {
      const short* in = coef + c;
      const unsigned short* qq = q + c;
      if (!in[8] && !in[16] && !in[24] && !in[32] && !in[40] && !in[48] && !in[56]) {
        const int64_t d = (int64_t)(in[0] * (int)qq[0]) * (1 << P1);
        for (int r = 0; r < 8; ++r) ws[r * 8 + c] = d;
        continue;
      }
      int64_t z2 = in[16] * (int)qq[16], z3 = in[48] * (int)qq[48];
      int64_t z1 = (z2 + z3) * F0_541;
      int64_t tmp2 = z1 + z3 * (-F1_847), tmp3 = z1 + z2 * F0_765;
      z2 = in[0] * (int)qq[0];
      z3 = in[32] * (int)qq[32];
      int64_t tmp0 = (z2 + z3) * (1 << CB), tmp1 = (z2 - z3) * (1 << CB);
      const int64_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
      tmp0 = in[56] * (int)qq[56];
      tmp1 = in[40] * (int)qq[40];
      tmp2 = in[24] * (int)qq[24];
      tmp3 = in[8] * (int)qq[8];
      z1 = tmp0 + tmp3;
      z2 = tmp1 + tmp2;
      z3 = tmp0 + tmp2;
      int64_t z4 = tmp1 + tmp3;
      const int64_t z5 = (z3 + z4) * F1_175;
      tmp0 *= F0_298;
      tmp1 *= F2_053;
      tmp2 *= F3_072;
      tmp3 *= F1_501;
      z1 *= -F0_899;
      z2 *= -F2_562;
      z3 *= -F1_961;
      z4 *= -F0_390;
      z3 += z5;
      z4 += z5;
      tmp0 += z1 + z3;
      tmp1 += z2 + z4;
      tmp2 += z2 + z3;
      tmp3 += z1 + z4;
      ws[0 * 8 + c] = descale(tmp10 + tmp3, CB - P1);
      ws[7 * 8 + c] = descale(tmp10 - tmp3, CB - P1);
      ws[1 * 8 + c] = descale(tmp11 + tmp2, CB - P1);
      ws[6 * 8 + c] = descale(tmp11 - tmp2, CB - P1);
      ws[2 * 8 + c] = descale(tmp12 + tmp1, CB - P1);
      ws[5 * 8 + c] = descale(tmp12 - tmp1, CB - P1);
      ws[3 * 8 + c] = descale(tmp13 + tmp0, CB - P1);
      ws[4 * 8 + c] = descale(tmp13 - tmp0, CB - P1);
    }
    for (int r = 0; r < 8; ++r) {
      const int64_t* w = ws + r * 8;
      unsigned char* o = out + (size_t)r * stride;
      int64_t z2 = w[2], z3 = w[6];
      int64_t z1 = (z2 + z3) * F0_541;
      int64_t tmp2 = z1 + z3 * (-F1_847), tmp3 = z1 + z2 * F0_765;
      int64_t tmp0 = (w[0] + w[4]) * (1 << CB), tmp1 = (w[0] - w[4]) * (1 << CB);
      const int64_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
      tmp0 = w[7];
      tmp1 = w[5];
      tmp2 = w[3];
      tmp3 = w[1];
      z1 = tmp0 + tmp3;
      z2 = tmp1 + tmp2;
      z3 = tmp0 + tmp2;
      int64_t z4 = tmp1 + tmp3;
      const int64_t z5 = (z3 + z4) * F1_175;
      tmp0 *= F0_298;
      tmp1 *= F2_053;
      tmp2 *= F3_072;
      tmp3 *= F1_501;
      z1 *= -F0_899;
      z2 *= -F2_562;
      z3 *= -F1_961;
      z4 *= -F0_390;
      z3 += z5;
      z4 += z5;
      tmp0 += z1 + z3;
      tmp1 += z2 + z4;
      tmp2 += z2 + z3;
      tmp3 += z1 + z4;
      constexpr int S = CB + P1 + 3;
      o[0] = range_limit(descale(tmp10 + tmp3, S));
      o[7] = range_limit(descale(tmp10 - tmp3, S));
      o[1] = range_limit(descale(tmp11 + tmp2, S));
      o[6] = range_limit(descale(tmp11 - tmp2, S));
      o[2] = range_limit(descale(tmp12 + tmp1, S));
      o[5] = range_limit(descale(tmp12 - tmp1, S));
      o[3] = range_limit(descale(tmp13 + tmp0, S));
      o[4] = range_limit(descale(tmp13 - tmp0, S));
    }
  }

  bool decode_block(Component& c, int bx, int by) {
    static const unsigned char zz[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                         41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                         30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
    short coef[64];
    memset(coef, 0, sizeof(coef));
    const int s = decode_symbol(dc[c.td]);
    if (s > 15) return fail("bad DC size");
    if (s) c.pred += extend(get(s), s);
    coef[0] = (short)c.pred;
    for (int k = 1; k < 64;) {
      const int rs = decode_symbol(ac[c.ta]);
      const int r = rs >> 4, sz = rs & 15;
      if (sz == 0) {
        if (r != 15) break;
        k += 16;
        continue;
      }
      k += r;
      if (k > 63) return fail("AC run past the block");
      coef[zz[k]] = (short)extend(get(sz), sz);
      ++k;
    }
    if (err) return false;
    idct(coef, qt[c.tq], c.plane.data() + (size_t)by * 8 * (c.bw * 8) + (size_t)bx * 8, c.bw * 8);
    return true;
  }

  // ---- markers
  static unsigned be16(const unsigned char* q) { return (unsigned)q[0] << 8 | q[1]; }

  bool parse_tables_and_scan(int& got_scan) {
    got_scan = 0;
    for (;;) {
      if (end - p < 4) return fail("truncated stream");
      if (p[0] != 0xFF) return fail("marker expected");
      while (p < end && *p == 0xFF) ++p;
      if (p >= end) return fail("truncated stream");
      const unsigned m = *p++;
      if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) continue;
      if (m == 0xD9) return fail("end of image before a scan");
      if (end - p < 2) return fail("truncated segment");
      const unsigned len = be16(p);
      if (len < 2 || (size_t)(end - p) < len) return fail("truncated segment");
      const unsigned char* s = p + 2;
      const unsigned char* se = p + len;
      p += len;
      if (m == 0xDB) {  // DQT
        while (s < se) {
          const int pq = s[0] >> 4, tq = s[0] & 15;
          ++s;
          if (tq > 3 || pq > 1 || se - s < 64 * (pq + 1)) return fail("bad quantisation table");
          static const unsigned char zz[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                               41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                               30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
          for (int i = 0; i < 64; ++i) {
            qt[tq][zz[i]] = (unsigned short)(pq ? be16(s) : s[0]);
            s += pq + 1;
          }
          qt_present[tq] = true;
        }
      } else if (m == 0xC4) {  // DHT
        while (s < se) {
          if (se - s < 17) return fail("bad Huffman segment");
          const int tc = s[0] >> 4, th = s[0] & 15;
          int n = 0;
          for (int i = 0; i < 16; ++i) n += s[1 + i];
          if (tc > 1 || th > 3 || n > 256 || se - s < 17 + n) return fail("bad Huffman segment");
          if (!build(tc ? ac[th] : dc[th], s + 1, s + 17, n)) return false;
          s += 17 + n;
        }
      } else if (m == 0xC0 || m == 0xC1) {  // SOF0 / SOF1: sequential, Huffman
        if (se - s < 6) return fail("bad frame header");
        if (s[0] != 8) return nosupport("sample precision other than 8 bits");
        height = (int)be16(s + 1);
        width = (int)be16(s + 3);
        ncomp = s[5];
        if (ncomp != 3) return nosupport("not a three-component (Y Cb Cr) image");
        if (se - s < 6 + 3 * ncomp) return fail("bad frame header");
        for (int i = 0; i < ncomp; ++i) {
          comp[i].id = s[6 + 3 * i];
          comp[i].h = s[7 + 3 * i] >> 4;
          comp[i].v = s[7 + 3 * i] & 15;
          comp[i].tq = s[8 + 3 * i];
          if (comp[i].tq > 3) return fail("bad frame header");
        }
      } else if (m == 0xC2 || (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC)) {
        return nosupport("progressive / lossless / arithmetic-coded JPEG");
      } else if (m == 0xDD) {  // DRI
        if (se - s < 2) return fail("bad restart interval");
        restart_interval = (int)be16(s);
      } else if (m == 0xDA) {  // SOS
        if (!ncomp) return fail("scan before the frame header");
        if (se - s < 1 || s[0] != ncomp || se - s < 1 + 2 * ncomp + 3) return nosupport("non-interleaved scan");
        for (int i = 0; i < ncomp; ++i) {
          if (s[1 + 2 * i] != comp[i].id) return nosupport("scan component order");
          comp[i].td = s[2 + 2 * i] >> 4;
          comp[i].ta = s[2 + 2 * i] & 15;
          if (comp[i].td > 3 || comp[i].ta > 3 || !dc[comp[i].td].present || !ac[comp[i].ta].present || !qt_present[comp[i].tq])
            return fail("scan refers to a missing table");
        }
        got_scan = 1;
        return true;
      }
      // every other segment (APPn, COM, ...) is skipped
    }
  }

  bool decode_scan() {
    hmax = vmax = 1;
    for (int i = 0; i < ncomp; ++i) {
      if (comp[i].h < 1 || comp[i].v < 1 || comp[i].h > 2 || comp[i].v > 2) return nosupport("sampling factors above 2");
      hmax = comp[i].h > hmax ? comp[i].h : hmax;
      vmax = comp[i].v > vmax ? comp[i].v : vmax;
    }
    if (comp[0].h != hmax || comp[0].v != vmax || comp[1].h != 1 || comp[1].v != 1 || comp[2].h != 1 || comp[2].v != 1 || (hmax == 1 && vmax == 2))
      return nosupport("chroma sampling other than 4:4:4, 4:2:2 or 4:2:0");
    if (width <= 0 || height <= 0) return fail("empty image");
    const int mcux = (width + 8 * hmax - 1) / (8 * hmax), mcuy = (height + 8 * vmax - 1) / (8 * vmax);
    for (int i = 0; i < ncomp; ++i) {
      Component& c = comp[i];
      c.dw = (width * c.h + hmax - 1) / hmax;
      c.dh = (height * c.v + vmax - 1) / vmax;
      c.bw = mcux * c.h;
      c.bh = mcuy * c.v;
      c.plane.assign((size_t)c.bw * 8 * c.bh * 8, 0);
      c.pred = 0;
    }
    bitbuf = 0;
    bits = 0;
    hit_marker = false;
    int until_restart = restart_interval, next_rst = 0;
    for (int my = 0; my < mcuy; ++my)
      for (int mx = 0; mx < mcux; ++mx) {
        if (restart_interval && until_restart == 0) {
          // byte-align, expect RSTn
          bitbuf = 0;
          bits = 0;
          hit_marker = false;
          while (p < end && !(p[0] == 0xFF && p + 1 < end && p[1] >= 0xD0 && p[1] <= 0xD7)) ++p;  // (skips stuffing left in the stream)
          if (end - p < 2) return fail("missing restart marker");
          if (p[1] != 0xD0 + next_rst) return fail("restart markers out of sequence");
          p += 2;
          next_rst = (next_rst + 1) & 7;
          until_restart = restart_interval;
          for (int i = 0; i < ncomp; ++i) comp[i].pred = 0;
        }
        for (int i = 0; i < ncomp; ++i)
          for (int v = 0; v < comp[i].v; ++v)
            for (int h = 0; h < comp[i].h; ++h)
              if (!decode_block(comp[i], mx * comp[i].h + h, my * comp[i].v + v)) return false;
        if (restart_interval) --until_restart;
      }
    return true;
  }

  // ---- chroma to full resolution (jdsample.c: fullsize copy, h2v1_fancy_upsample, h2v2_fancy_upsample)
  static void upsample(const Component& c, int hmax, int vmax, int width, int height, std::vector<unsigned char>& out) {
    out.assign((size_t)width * height, 0);
    const int stride = c.bw * 8, dw = c.dw, dh = c.dh;
    const unsigned char* src = c.plane.data();
    const int hx = hmax / c.h, vx = vmax / c.v;
    if (hx == 1 && vx == 1) {
      for (int y = 0; y < height; ++y) memcpy(&out[(size_t)y * width], src + (size_t)y * stride, (size_t)width);
      return;
    }
    std::vector<unsigned char> row((size_t)dw * 2 + 2);
    if (dw <= 2) {  // libjpeg only smooths planes more than two samples wide (jdsample.c jinit_upsampler): plain replication
      for (int y = 0; y < height; ++y)
        for (int x = 0; x < width; ++x) out[(size_t)y * width + x] = src[(size_t)(vx == 2 ? y >> 1 : y) * stride + (x >> 1)];
      return;
    }
    for (int y = 0; y < height; ++y) {
      const int r = vx == 2 ? y >> 1 : y;
      const unsigned char* in0 = src + (size_t)r * stride;
      if (vx == 1) {  // h2v1
        unsigned char* o = row.data();
        {
          int v = in0[0];
          *o++ = (unsigned char)v;
          *o++ = (unsigned char)((v * 3 + in0[1] + 2) >> 2);
          for (int x = 1; x < dw - 1; ++x) {
            v = in0[x] * 3;
            *o++ = (unsigned char)((v + in0[x - 1] + 1) >> 2);
            *o++ = (unsigned char)((v + in0[x + 1] + 2) >> 2);
          }
          if (dw >= 2) {
            v = in0[dw - 1];
            *o++ = (unsigned char)((v * 3 + in0[dw - 2] + 1) >> 2);
            *o++ = (unsigned char)v;
          }
        }
      } else {  // h2v2: the nearer row 3/4, the further one 1/4; rows beyond the real ones are copies of the edge row
        int r1 = (y & 1) ? r + 1 : r - 1;
        if (r1 < 0) r1 = 0;
        if (r1 > dh - 1) r1 = dh - 1;
        const unsigned char* in1 = src + (size_t)r1 * stride;
        unsigned char* o = row.data();
        {
          int thiscol = in0[0] * 3 + in1[0], nextcol = in0[1] * 3 + in1[1], lastcol;
          *o++ = (unsigned char)((thiscol * 4 + 8) >> 4);
          *o++ = (unsigned char)((thiscol * 3 + nextcol + 7) >> 4);
          lastcol = thiscol;
          thiscol = nextcol;
          for (int x = 2; x < dw; ++x) {
            nextcol = in0[x] * 3 + in1[x];
            *o++ = (unsigned char)((thiscol * 3 + lastcol + 8) >> 4);
            *o++ = (unsigned char)((thiscol * 3 + nextcol + 7) >> 4);
            lastcol = thiscol;
            thiscol = nextcol;
          }
          if (dw >= 2) {
            *o++ = (unsigned char)((thiscol * 3 + lastcol + 8) >> 4);
            *o++ = (unsigned char)((thiscol * 4 + 7) >> 4);
          }
        }
      }
      memcpy(&out[(size_t)y * width], row.data(), (size_t)width);
    }
  }
};

// Decodes `data` into width x height RGB8 (R, G, B byte order, rows top to bottom — libjpeg's JCS_RGB scanlines).
// Returns 0, -1 (malformed / size mismatch) or -2 (a JPEG flavour this decoder does not handle); *err names the reason.
inline int decode_rgb(const unsigned char* data, size_t len, int width, int height, unsigned char* rgb, const char** err) {
  Decoder d;
  d.p = data;
  d.end = data + len;
  auto out = [&](int rc) {
    if (err) *err = d.err ? d.err : "";
    return rc;
  };
  if (len < 4 || data[0] != 0xFF || data[1] != 0xD8) {
    d.fail("not a JPEG stream (no SOI)");
    return out(-1);
  }
  int got = 0;
  if (!d.parse_tables_and_scan(got) || !got) return out(d.unsupported ? -2 : -1);
  if (d.width != width || d.height != height) {
    d.fail("image size differs from the log's resolution");
    return out(-1);
  }
  if (!d.decode_scan()) return out(d.unsupported ? -2 : -1);
  std::vector<unsigned char> cb, cr;
  Decoder::upsample(d.comp[1], d.hmax, d.vmax, width, height, cb);
  Decoder::upsample(d.comp[2], d.hmax, d.vmax, width, height, cr);
  // jdcolor.c build_ycc_rgb_table / ycc_rgb_convert
  int cr_r[256], cb_b[256];
  int64_t cr_g[256], cb_g[256];
  for (int i = 0; i < 256; ++i) {
    const int64_t x = i - 128;
    cr_r[i] = (int)((91881 * x + 32768) >> 16);
    cb_b[i] = (int)((116130 * x + 32768) >> 16);
    cr_g[i] = -46802 * x;
    cb_g[i] = -22554 * x + 32768;
  }
  auto clamp = [](int v) { return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v)); };
  const int ystride = d.comp[0].bw * 8;
  for (int y = 0; y < height; ++y) {
    const unsigned char* yr = d.comp[0].plane.data() + (size_t)y * ystride;
    const unsigned char* br = &cb[(size_t)y * width];
    const unsigned char* rr = &cr[(size_t)y * width];
    unsigned char* o = rgb + (size_t)y * width * 3;
    for (int x = 0; x < width; ++x) {
      const int Y = yr[x], B = br[x], R = rr[x];
      o[3 * x + 0] = clamp(Y + cr_r[R]);
      o[3 * x + 1] = clamp(Y + (int)((cb_g[B] + cr_g[R]) >> 16));
      o[3 * x + 2] = clamp(Y + cb_b[B]);
    }
  }
  return out(0);
}

}  // namespace jpeg
}  // namespace dms
