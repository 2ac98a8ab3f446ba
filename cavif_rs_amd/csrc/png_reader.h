// png_reader.h -- PNG -> RGBA8, the input side of the cavif CLI (reference: load_rgba, src/main.rs:265-283, which maps
// every load_image pixel kind to RGBA8: RGB -> alpha 255, 16-bit -> high byte, gray -> r=g=b).
// Host C++ over zlib's inflate; written from the PNG specification (chunks IHDR / PLTE / tRNS / IDAT / IEND, the five
// scanline filters, Adam7).  JPEG input (the reference also accepts it through load_image) is not handled: callers get
// MI_UNSUPPORTED.
#pragma once
#include <algorithm>
#include <zlib.h>
#include <cstdint>
#include <cstring>
#include <vector>

namespace mi {

inline uint32_t png_be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

// un-filters `rows` scanlines of `rowbytes` bytes (each preceded by its filter byte) in place; bpp = bytes per complete pixel (>= 1)
inline bool png_unfilter(uint8_t *data, size_t rows, size_t rowbytes, int bpp) {
  std::vector<uint8_t> zero(rowbytes, 0);
  const uint8_t *prev = zero.data();
  for (size_t y = 0; y < rows; y++) {
    uint8_t *line = data + y * (rowbytes + 1);
    const int ft = line[0];
    uint8_t *cur = line + 1;
    for (size_t i = 0; i < rowbytes; i++) {
      const int a = i >= (size_t)bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= (size_t)bpp ? prev[i - bpp] : 0;
      int add;
      switch (ft) {
        case 0: add = 0; break;
        case 1: add = a; break;
        case 2: add = b; break;
        case 3: add = (a + b) >> 1; break;
        case 4: { const int p = a + b - c, pa = p > a ? p - a : a - p, pb = p > b ? p - b : b - p, pc = p > c ? p - c : c - p;
                  add = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); break; }
        default: return false;
      }
      cur[i] = (uint8_t)(cur[i] + add);
    }
    prev = cur;
  }
  return true;
}

// Returns 0 on success, 2 (MI_UNSUPPORTED) for non-PNG / unsupported data, 3 (MI_ENCODING_ERROR) for corrupt streams.
inline int png_decode_rgba(const uint8_t *d, size_t len, std::vector<uint8_t> &rgba, uint32_t &w, uint32_t &h) {
  static const uint8_t sig[8] = { 0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A };
  if (len < 8 + 25 || memcmp(d, sig, 8) != 0) return 2;
  size_t pos = 8;
  int depth = 0, ctype = 0, interlace = 0; bool have_ihdr = false;
  std::vector<uint8_t> idat, plte, trns;
  while (pos + 12 <= len) {
    const uint32_t n = png_be32(d + pos); const uint8_t *type = d + pos + 4, *body = d + pos + 8;
    if (n > len || pos + 12 + (size_t)n > len) return 3;
    if (!memcmp(type, "IHDR", 4)) {
      if (n != 13) return 3;
      w = png_be32(body); h = png_be32(body + 4); depth = body[8]; ctype = body[9]; interlace = body[12];
      if (body[10] != 0 || body[11] != 0 || interlace > 1 || w == 0 || h == 0 || w > (1u << 16) || h > (1u << 16)) return 2;
      have_ihdr = true;
    } else if (!memcmp(type, "PLTE", 4)) plte.assign(body, body + n);
    else if (!memcmp(type, "tRNS", 4)) trns.assign(body, body + n);
    else if (!memcmp(type, "IDAT", 4)) idat.insert(idat.end(), body, body + n);
    else if (!memcmp(type, "IEND", 4)) break;
    pos += 12 + (size_t)n;
  }
  if (!have_ihdr || idat.empty()) return 3;
  const int channels = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
  if (!channels) return 2;
  if (!(depth == 8 || depth == 16 || ((ctype == 0 || ctype == 3) && (depth == 1 || depth == 2 || depth == 4))) || (ctype == 3 && depth == 16)) return 2;
  if (ctype == 3 && plte.size() < 3) return 3;
  const int bits_pp = channels * depth, bpp = bits_pp >= 8 ? bits_pp / 8 : 1;
  // pass geometry: one pass for non-interlaced, seven for Adam7
  struct Pass { uint32_t x0, y0, dx, dy; };
  static const Pass adam7[7] = { { 0, 0, 8, 8 }, { 4, 0, 8, 8 }, { 0, 4, 4, 8 }, { 2, 0, 4, 4 }, { 0, 2, 2, 4 }, { 1, 0, 2, 2 }, { 0, 1, 1, 2 } };
  const Pass whole = { 0, 0, 1, 1 };
  const int npass = interlace ? 7 : 1;
  size_t total = 0;
  for (int p = 0; p < npass; p++) {
    const Pass &ps = interlace ? adam7[p] : whole;
    const uint32_t pw = (w - ps.x0 + ps.dx - 1) / ps.dx, ph = (h - ps.y0 + ps.dy - 1) / ps.dy;
    if (w <= ps.x0 || h <= ps.y0 || !pw || !ph) continue;
    total += (size_t)ph * (1 + ((size_t)pw * bits_pp + 7) / 8);
  }
  // a deflate stream expands at most ~1032x: a tiny file that claims a huge canvas is refused before anything is allocated
  if (total > idat.size() * 1040 + 65536) return 3;
  std::vector<uint8_t> raw(total);
  {
    z_stream zs; memset(&zs, 0, sizeof(zs));
    if (inflateInit(&zs) != Z_OK) return 3;
    // zlib counts in uInt: feed and drain in chunks so that streams and canvases beyond 4 GiB work
    size_t in_pos = 0, out_pos = 0; int zr = Z_OK;
    const size_t chunk = (size_t)1 << 30;
    while (zr == Z_OK || zr == Z_BUF_ERROR) {
      if (zs.avail_in == 0 && in_pos < idat.size()) { const size_t k = std::min(chunk, idat.size() - in_pos); zs.next_in = idat.data() + in_pos; zs.avail_in = (uInt)k; in_pos += k; }
      if (zs.avail_out == 0 && out_pos < total) { const size_t k = std::min(chunk, total - out_pos); zs.next_out = raw.data() + out_pos; zs.avail_out = (uInt)k; out_pos += k; }
      const uInt in_before = zs.avail_in, out_before = zs.avail_out;
      zr = inflate(&zs, Z_NO_FLUSH);
      if (zr == Z_BUF_ERROR && in_before == zs.avail_in && out_before == zs.avail_out && (in_pos >= idat.size() || out_pos >= total)) break;   // no progress possible
    }
    const size_t got = out_pos - zs.avail_out;
    inflateEnd(&zs);
    if ((zr != Z_STREAM_END && zr != Z_OK && zr != Z_BUF_ERROR) || got != total) return 3;
  }
  rgba.assign((size_t)w * h * 4, 0);
  const int mx = (1 << (depth > 8 ? 8 : depth)) - 1;
  size_t off = 0;
  for (int p = 0; p < npass; p++) {
    const Pass &ps = interlace ? adam7[p] : whole;
    if (w <= ps.x0 || h <= ps.y0) continue;
    const uint32_t pw = (w - ps.x0 + ps.dx - 1) / ps.dx, ph = (h - ps.y0 + ps.dy - 1) / ps.dy;
    if (!pw || !ph) continue;
    const size_t rowbytes = ((size_t)pw * bits_pp + 7) / 8;
    if (!png_unfilter(raw.data() + off, ph, rowbytes, bpp)) return 3;
    for (uint32_t yy = 0; yy < ph; yy++) {
      const uint8_t *row = raw.data() + off + (size_t)yy * (rowbytes + 1) + 1;
      for (uint32_t xx = 0; xx < pw; xx++) {
        // sample fetch: `depth`-bit big-endian samples, MSB first inside a byte; 16-bit samples keep their high byte
        // (px.map(|c| (c >> 8) as u8), src/main.rs:272-273), sub-byte gray is scaled to 0..255
        int s[4] = { 0, 0, 0, 0 }; int key16[4] = { 0, 0, 0, 0 };
        for (int c = 0; c < channels; c++) {
          const size_t bit = ((size_t)xx * channels + c) * depth;
          if (depth == 8) s[c] = row[bit >> 3];
          else if (depth == 16) { s[c] = row[bit >> 3]; key16[c] = (row[bit >> 3] << 8) | row[(bit >> 3) + 1]; }
          else s[c] = (row[bit >> 3] >> (8 - depth - (bit & 7))) & mx;
        }
        uint8_t r, g, b, a = 255;
        if (ctype == 3) {
          const size_t idx = (size_t)s[0];
          if (idx * 3 + 2 >= plte.size()) return 3;
          r = plte[idx * 3]; g = plte[idx * 3 + 1]; b = plte[idx * 3 + 2];
          if (idx < trns.size()) a = trns[idx];
        } else if (ctype == 0 || ctype == 4) {
          const int v = depth < 8 ? s[0] * 255 / mx : s[0];
          r = g = b = (uint8_t)v;
          if (ctype == 4) a = (uint8_t)s[1];
          else if (trns.size() >= 2) { const int key = (trns[0] << 8) | trns[1]; if ((depth == 16 ? key16[0] : s[0]) == key) a = 0; }
        } else {
          r = (uint8_t)s[0]; g = (uint8_t)s[1]; b = (uint8_t)s[2];
          if (ctype == 6) a = (uint8_t)s[3];
          else if (trns.size() >= 6) {
            bool eq = true;
            for (int c = 0; c < 3; c++) { const int key = (trns[2 * c] << 8) | trns[2 * c + 1]; eq = eq && ((depth == 16 ? key16[c] : s[c]) == key); }
            if (eq) a = 0;
          }
        }
        uint8_t *o = &rgba[(((size_t)ps.y0 + (size_t)yy * ps.dy) * w + ps.x0 + (size_t)xx * ps.dx) * 4];
        o[0] = r; o[1] = g; o[2] = b; o[3] = a;
      }
    }
    off += (size_t)ph * (rowbytes + 1);
  }
  return 0;
}

}  // namespace mi
