// rtx_images.inl — image files for textures and environment maps. Included by rtx_host.cpp.
//
// read_image (rc/imageio.rs:16-33) picks the decoder by file extension: tga / TGA / png / PNG -> the `image` crate
// (0.24.2, png 0.17.5; absent from /root/reference, pinned by Cargo.lock) followed by `to_rgb8()` and c / 255
// (imageio.rs:94-112); hdr -> image's HdrDecoder (non-strict) and Rgbe8Pixel::to_hdr (imageio.rs:114-132); pfm ->
// read_image_pfm (rtxh_pfm_read above); exr -> the `exr` crate (1.4.2), scan-line files below. The formats' published
// definitions are restated below: RFC 1950 / 1951 (zlib, DEFLATE), the PNG specification (filters, Adam7, sample
// expansion as png's EXPAND transformation does it), Truevision TGA 2.0 (types 1 2 3 9 10 11) and the Radiance RGBE
// picture format (flat and new-style run-length scanlines). Row 0 of every result is the top of the image.

namespace {

struct Bytes {
  const uint8_t* p; size_t n, pos = 0;
  bool need(size_t k) const { return pos + k <= n; }
  uint8_t u8() { return p[pos++]; }
  uint32_t be32() { uint32_t v = (uint32_t)p[pos] << 24 | (uint32_t)p[pos + 1] << 16 | (uint32_t)p[pos + 2] << 8 | p[pos + 3]; pos += 4; return v; }
  uint16_t le16() { uint16_t v = (uint16_t)(p[pos] | p[pos + 1] << 8); pos += 2; return v; }
};

bool read_whole_file(const char* path, std::vector<uint8_t>& out) {
  FILE* f = fopen(path, "rb"); if (!f) return false;
  uint8_t buf[65536]; size_t n; out.clear();
  while ((n = fread(buf, 1, sizeof buf, f)) > 0) out.insert(out.end(), buf, buf + n);
  fclose(f); return true;
}

// ------------------------------------------------------------------ DEFLATE (RFC 1951) inside a zlib stream (RFC 1950)
struct Inflater {
  const uint8_t* in; size_t n, pos = 0; uint32_t bitbuf = 0; int bitcnt = 0; std::vector<uint8_t>& out; const char* err = nullptr;
  Inflater(const uint8_t* i, size_t nn, std::vector<uint8_t>& o) : in(i), n(nn), out(o) {}
  int bits(int need) {
    uint32_t v = bitbuf;
    while (bitcnt < need) { if (pos >= n) { err = "deflate: out of input"; return 0; } v |= (uint32_t)in[pos++] << bitcnt; bitcnt += 8; }
    bitbuf = need == 32 ? 0 : v >> need; bitcnt -= need;
    return (int)(v & ((need == 32 ? 0 : (1u << need)) - 1u));
  }
  struct Huff { uint16_t count[16]; uint16_t symbol[288]; };
  static bool build(Huff& h, const uint8_t* len, int n) {  // canonical code from code lengths (RFC 1951 §3.2.2)
    memset(h.count, 0, sizeof h.count);
    for (int s = 0; s < n; ++s) h.count[len[s]]++;
    if (h.count[0] == n) return true;
    int left = 1;
    for (int l = 1; l < 16; ++l) { left <<= 1; left -= h.count[l]; if (left < 0) return false; }
    uint16_t offs[16]; offs[1] = 0;
    for (int l = 1; l < 15; ++l) offs[l + 1] = (uint16_t)(offs[l] + h.count[l]);
    for (int s = 0; s < n; ++s) if (len[s]) h.symbol[offs[len[s]]++] = (uint16_t)s;
    return true;
  }
  int decode(const Huff& h) {
    int code = 0, first = 0, index = 0;
    for (int l = 1; l < 16; ++l) {
      code |= bits(1); if (err) return -1;
      int c = h.count[l];
      if (code - c < first) return h.symbol[index + (code - first)];
      index += c; first += c; first <<= 1; code <<= 1;
    }
    err = "deflate: bad code"; return -1;
  }
  bool codes(const Huff& lit, const Huff& dist) {
    static const uint16_t lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint16_t lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    static const uint16_t dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const uint16_t dext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    for (;;) {
      int sym = decode(lit); if (err) return false;
      if (sym < 256) out.push_back((uint8_t)sym);
      else if (sym == 256) return true;
      else {
        sym -= 257; if (sym >= 29) { err = "deflate: bad length symbol"; return false; }
        int len = lbase[sym] + bits(lext[sym]);
        int ds = decode(dist); if (err) return false;
        if (ds >= 30) { err = "deflate: bad distance symbol"; return false; }
        size_t d = (size_t)dbase[ds] + (size_t)bits(dext[ds]); if (err) return false;
        if (d > out.size()) { err = "deflate: distance too far back"; return false; }
        size_t from = out.size() - d;
        for (int k = 0; k < len; ++k) out.push_back(out[from + k]);
      }
    }
  }
  bool run() {
    for (;;) {
      int last = bits(1), type = bits(2); if (err) return false;
      if (type == 0) {
        bitbuf = 0; bitcnt = 0;
        if (pos + 4 > n) { err = "deflate: out of input"; return false; }
        unsigned len = in[pos] | in[pos + 1] << 8, nlen = in[pos + 2] | in[pos + 3] << 8; pos += 4;
        if (len != (~nlen & 0xffffu)) { err = "deflate: stored block length mismatch"; return false; }
        if (pos + len > n) { err = "deflate: out of input"; return false; }
        out.insert(out.end(), in + pos, in + pos + len); pos += len;
      } else if (type == 1) {
        uint8_t l[320]; Huff lit, dist;
        for (int s = 0; s < 144; ++s) l[s] = 8; for (int s = 144; s < 256; ++s) l[s] = 9; for (int s = 256; s < 280; ++s) l[s] = 7; for (int s = 280; s < 288; ++s) l[s] = 8;
        build(lit, l, 288);
        for (int s = 0; s < 30; ++s) l[s] = 5;
        build(dist, l, 30);
        if (!codes(lit, dist)) return false;
      } else if (type == 2) {
        static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
        int nlen = bits(5) + 257, ndist = bits(5) + 1, ncode = bits(4) + 4; if (err) return false;
        if (nlen > 286 || ndist > 30) { err = "deflate: too many codes"; return false; }
        uint8_t l[320]; memset(l, 0, sizeof l);
        for (int i = 0; i < ncode; ++i) l[order[i]] = (uint8_t)bits(3);
        if (err) return false;
        Huff lencode; if (!build(lencode, l, 19)) { err = "deflate: bad code lengths"; return false; }
        uint8_t ll[320]; int idx = 0;
        while (idx < nlen + ndist) {
          int sym = decode(lencode); if (err) return false;
          if (sym < 16) ll[idx++] = (uint8_t)sym;
          else {
            int prev = 0, rep;
            if (sym == 16) { if (idx == 0) { err = "deflate: repeat without a previous length"; return false; } prev = ll[idx - 1]; rep = 3 + bits(2); }
            else if (sym == 17) rep = 3 + bits(3); else rep = 11 + bits(7);
            if (err) return false;
            if (idx + rep > nlen + ndist) { err = "deflate: too many lengths"; return false; }
            while (rep--) ll[idx++] = (uint8_t)prev;
          }
        }
        if (ll[256] == 0) { err = "deflate: no end-of-block code"; return false; }
        Huff lit, dist;
        if (!build(lit, ll, nlen) || !build(dist, ll + nlen, ndist)) { err = "deflate: over-subscribed code"; return false; }
        if (!codes(lit, dist)) return false;
      } else { err = "deflate: reserved block type"; return false; }
      if (last) return true;
    }
  }
};

bool zlib_inflate(const uint8_t* src, size_t n, std::vector<uint8_t>& out, std::string& err) {
  if (n < 6) { err = "zlib: stream too short"; return false; }
  if ((src[0] & 0x0f) != 8 || ((src[0] << 8 | src[1]) % 31) != 0 || (src[1] & 0x20)) { err = "zlib: bad header"; return false; }
  Inflater z(src + 2, n - 2, out);
  if (!z.run()) { err = z.err ? z.err : "deflate: error"; return false; }
  size_t end = 2 + z.pos;
  if (end + 4 > n) { err = "zlib: missing checksum"; return false; }
  uint32_t a = 1, b = 0;
  for (uint8_t c : out) { a = (a + c) % 65521u; b = (b + a) % 65521u; }
  const uint32_t want = (uint32_t)src[end] << 24 | (uint32_t)src[end + 1] << 16 | (uint32_t)src[end + 2] << 8 | src[end + 3];
  if (((b << 16) | a) != want) { err = "zlib: Adler-32 mismatch"; return false; }
  return true;
}

uint32_t crc32_of(const uint8_t* p, size_t n) {
  static uint32_t table[256]; static bool init = false;
  if (!init) { for (uint32_t i = 0; i < 256; ++i) { uint32_t c = i; for (int k = 0; k < 8; ++k) c = c & 1 ? 0xedb88320u ^ (c >> 1) : c >> 1; table[i] = c; } init = true; }
  uint32_t c = 0xffffffffu;
  for (size_t i = 0; i < n; ++i) c = table[(c ^ p[i]) & 0xff] ^ (c >> 8);
  return c ^ 0xffffffffu;
}

// ------------------------------------------------------------------ PNG -> RGB8 (png's EXPAND + image's to_rgb8)
inline uint8_t sample16_to_8(unsigned c) { return (uint8_t)((c + 128u) / 257u); }  // image 0.24 FromPrimitive<u16> for u8
bool png_decode(const std::vector<uint8_t>& file, int& W, int& H, std::vector<uint8_t>& rgb, std::string& err) {
  static const uint8_t sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
  if (file.size() < 8 || memcmp(file.data(), sig, 8) != 0) { err = "PNG: bad signature"; return false; }
  Bytes b{file.data(), file.size(), 8};
  uint32_t w = 0, h = 0; int depth = 0, ctype = -1, interlace = 0; bool have_ihdr = false, have_end = false;
  std::vector<uint8_t> idat, plte;
  while (!have_end) {
    if (!b.need(8)) { err = "PNG: truncated chunk header"; return false; }
    uint32_t len = b.be32(); const uint8_t* type = b.p + b.pos; b.pos += 4;
    if (!b.need((size_t)len + 4)) { err = "PNG: truncated chunk"; return false; }
    const uint8_t* data = b.p + b.pos; b.pos += len;
    const uint32_t crc = b.be32();
    if (crc32_of(type, (size_t)len + 4) != crc) { err = "PNG: chunk CRC mismatch"; return false; }
    if (!memcmp(type, "IHDR", 4)) {
      if (len != 13) { err = "PNG: bad IHDR"; return false; }
      Bytes d{data, 13}; w = d.be32(); h = d.be32(); depth = d.u8(); ctype = d.u8();
      if (d.u8() != 0 || d.u8() != 0) { err = "PNG: unknown compression / filter method"; return false; }
      interlace = d.u8(); have_ihdr = true;
      if (w == 0 || h == 0 || w > 65536 || h > 65536 || (uint64_t)w * h > (1ull << 28) || interlace > 1) { err = "PNG: bad dimensions"; return false; }
      const bool ok = (ctype == 0 && (depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)) || (ctype == 3 && (depth == 1 || depth == 2 || depth == 4 || depth == 8)) ||
                      ((ctype == 2 || ctype == 4 || ctype == 6) && (depth == 8 || depth == 16));
      if (!ok) { err = "PNG: invalid colour type / bit depth"; return false; }
    } else if (!memcmp(type, "PLTE", 4)) plte.assign(data, data + len);
    else if (!memcmp(type, "IDAT", 4)) idat.insert(idat.end(), data, data + len);
    else if (!memcmp(type, "IEND", 4)) have_end = true;
    else if (!(type[0] & 0x20)) { err = "PNG: unknown critical chunk"; return false; }
  }
  if (!have_ihdr || idat.empty()) { err = "PNG: missing IHDR / IDAT"; return false; }
  if (ctype == 3 && (plte.empty() || plte.size() % 3)) { err = "PNG: missing palette"; return false; }
  std::vector<uint8_t> raw; raw.reserve((size_t)w * h * 4 + h);
  if (!zlib_inflate(idat.data(), idat.size(), raw, err)) return false;
  const int channels = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : 4;
  const int bits_pp = channels * depth, bpp = bits_pp >= 8 ? bits_pp / 8 : 1;
  W = (int)w; H = (int)h; rgb.assign((size_t)w * h * 3, 0);
  auto put = [&](uint32_t x, uint32_t y, const uint8_t* line, uint32_t i) {  // i-th pixel of an unfiltered scanline -> rgb[x, y]
    uint8_t* o = &rgb[((size_t)y * w + x) * 3];
    auto s8 = [&](int c) -> uint8_t { return depth == 16 ? sample16_to_8((unsigned)line[(i * channels + c) * 2] << 8 | line[(i * channels + c) * 2 + 1]) : line[i * channels + c]; };
    if (ctype == 2 || ctype == 6) { o[0] = s8(0); o[1] = s8(1); o[2] = s8(2); return; }
    if (ctype == 4) { o[0] = o[1] = o[2] = s8(0); return; }
    unsigned v;
    if (depth >= 8) v = s8(0);
    else { const unsigned per = 8 / depth, byte = line[i / per], shift = (per - 1 - i % per) * depth; v = (byte >> shift) & ((1u << depth) - 1u); }
    if (ctype == 3) { if (3 * v + 2 >= plte.size()) { o[0] = o[1] = o[2] = 0; return; } o[0] = plte[3 * v]; o[1] = plte[3 * v + 1]; o[2] = plte[3 * v + 2]; return; }
    if (depth < 8) v = v * (255u / ((1u << depth) - 1u));  // png expand_gray_u8
    o[0] = o[1] = o[2] = (uint8_t)v;
  };
  static const int xs[7] = {0, 4, 0, 2, 0, 1, 0}, ys[7] = {0, 0, 4, 0, 2, 0, 1}, dx[7] = {8, 8, 4, 4, 2, 2, 1}, dy[7] = {8, 8, 8, 4, 4, 2, 2};
  size_t pos = 0;
  for (int pass = 0; pass < (interlace ? 7 : 1); ++pass) {
    const uint32_t pw = interlace ? (w > (uint32_t)xs[pass] ? (w - xs[pass] + dx[pass] - 1) / dx[pass] : 0) : w;
    const uint32_t ph = interlace ? (h > (uint32_t)ys[pass] ? (h - ys[pass] + dy[pass] - 1) / dy[pass] : 0) : h;
    if (pw == 0 || ph == 0) continue;
    const size_t stride = ((size_t)pw * bits_pp + 7) / 8;
    std::vector<uint8_t> prev(stride, 0), cur(stride);
    for (uint32_t r = 0; r < ph; ++r) {
      if (pos + 1 + stride > raw.size()) { err = "PNG: not enough image data"; return false; }
      const int ft = raw[pos++];
      memcpy(cur.data(), &raw[pos], stride); pos += stride;
      for (size_t i = 0; i < stride; ++i) {  // PNG specification, 9.2 filter types
        const int a = i >= (size_t)bpp ? cur[i - bpp] : 0, up = prev[i], c = i >= (size_t)bpp ? prev[i - bpp] : 0;
        int add;
        switch (ft) {
          case 0: add = 0; break;
          case 1: add = a; break;
          case 2: add = up; break;
          case 3: add = (a + up) >> 1; break;
          case 4: { int p = a + up - c, pa = std::abs(p - a), pb = std::abs(p - up), pc = std::abs(p - c); add = (pa <= pb && pa <= pc) ? a : (pb <= pc ? up : c); break; }
          default: err = "PNG: bad filter type"; return false;
        }
        cur[i] = (uint8_t)(cur[i] + add);
      }
      for (uint32_t i = 0; i < pw; ++i) put(interlace ? xs[pass] + i * dx[pass] : i, interlace ? ys[pass] + r * dy[pass] : r, cur.data(), i);
      prev.swap(cur);
    }
  }
  return true;
}

// ------------------------------------------------------------------ TGA -> RGB8 (image's TgaDecoder + to_rgb8)
bool tga_decode(const std::vector<uint8_t>& file, int& W, int& H, std::vector<uint8_t>& rgb, std::string& err) {
  if (file.size() < 18) { err = "TGA: truncated header"; return false; }
  Bytes b{file.data(), file.size()};
  const int id_len = b.u8(), map_type = b.u8(), type = b.u8();
  const int map_first = b.le16(), map_len = b.le16(), map_bits = b.u8();
  b.le16(); b.le16();
  const int w = b.le16(), h = b.le16(), bits = b.u8(), desc = b.u8();
  const bool rle = type & 8; const int base = type & 7;
  if (w == 0 || h == 0 || (uint64_t)w * h > (1ull << 28)) { err = "TGA: bad dimensions"; return false; }
  if (base < 1 || base > 3 || type > 11) { err = "TGA: unsupported image type"; return false; }
  if (base == 1 && (map_type != 1 || bits != 8 || (map_bits != 24 && map_bits != 32))) { err = "TGA: unsupported colour map"; return false; }
  if (base == 2 && bits != 24 && bits != 32) { err = "TGA: unsupported pixel depth"; return false; }
  if (base == 3 && bits != 8 && bits != 16) { err = "TGA: unsupported pixel depth"; return false; }
  b.pos += id_len;
  const int map_bytes = map_type == 1 ? map_bits / 8 : 0;
  if (!b.need((size_t)map_len * map_bytes)) { err = "TGA: truncated colour map"; return false; }
  const uint8_t* cmap = b.p + b.pos; b.pos += (size_t)map_len * map_bytes;
  const int bytes = bits / 8; const size_t npx = (size_t)w * h;
  std::vector<uint8_t> px(npx * bytes);
  if (!rle) {
    if (!b.need(px.size())) { err = "TGA: truncated image data"; return false; }
    memcpy(px.data(), b.p + b.pos, px.size());
  } else {
    size_t o = 0;
    while (o < npx) {
      if (!b.need(1)) { err = "TGA: truncated run-length data"; return false; }
      const int hd = b.u8(); const size_t cnt = (size_t)(hd & 0x7f) + 1;
      if (o + cnt > npx) { err = "TGA: run past the end of the image"; return false; }
      if (hd & 0x80) { if (!b.need(bytes)) { err = "TGA: truncated run-length data"; return false; } for (size_t k = 0; k < cnt; ++k) memcpy(&px[(o + k) * bytes], b.p + b.pos, bytes); b.pos += bytes; }
      else { if (!b.need(cnt * bytes)) { err = "TGA: truncated run-length data"; return false; } memcpy(&px[o * bytes], b.p + b.pos, cnt * bytes); b.pos += cnt * bytes; }
      o += cnt;
    }
  }
  W = w; H = h; rgb.resize(npx * 3);
  const bool top_origin = desc & 0x20, right_origin = desc & 0x10;
  for (int y = 0; y < h; ++y) for (int x = 0; x < w; ++x) {
    const uint8_t* s = &px[((size_t)y * w + x) * bytes];
    uint8_t* o = &rgb[((size_t)(top_origin ? y : h - 1 - y) * w + (right_origin ? w - 1 - x : x)) * 3];
    if (base == 3) { o[0] = o[1] = o[2] = s[0]; }
    else if (base == 2) { o[0] = s[2]; o[1] = s[1]; o[2] = s[0]; }  // stored B G R (A)
    else { const int i = (int)s[0] - map_first; if (i < 0 || i >= map_len) { o[0] = o[1] = o[2] = 0; } else { const uint8_t* c = cmap + (size_t)i * map_bytes; o[0] = c[2]; o[1] = c[1]; o[2] = c[0]; } }
  }
  return true;
}

// ------------------------------------------------------------------ Radiance .hdr -> float RGB (image's HdrDecoder, non-strict)
bool hdr_decode(const std::vector<uint8_t>& file, int& W, int& H, std::vector<float>& rgb, std::string& err) {
  size_t pos = 0;
  auto line = [&](std::string& out) -> bool { out.clear(); if (pos >= file.size()) return false; while (pos < file.size() && file[pos] != '\n') out += (char)file[pos++]; if (pos < file.size()) ++pos; return true; };
  std::string l;
  if (!line(l) || (l.compare(0, 10, "#?RADIANCE") != 0 && l.compare(0, 6, "#?RGBE") != 0)) { err = "HDR: missing #?RADIANCE signature"; return false; }
  bool have_format = false;
  for (;;) { if (!line(l)) { err = "HDR: truncated header"; return false; } if (l.empty()) break; if (l.compare(0, 7, "FORMAT=") == 0) { have_format = true; if (l != "FORMAT=32-bit_rle_rgbe") { err = "HDR: unsupported FORMAT"; return false; } } }
  (void)have_format;
  if (!line(l)) { err = "HDR: missing resolution line"; return false; }
  int h = 0, w = 0;
  if (sscanf(l.c_str(), "-Y %d +X %d", &h, &w) != 2 || w <= 0 || h <= 0 || (uint64_t)w * (uint64_t)h > (1ull << 28)) { err = "HDR: unsupported orientation / resolution \"" + l + "\""; return false; }
  W = w; H = h; rgb.assign((size_t)w * h * 3, 0.0f);
  std::vector<uint8_t> scan((size_t)w * 4);
  for (int y = 0; y < h; ++y) {
    if (pos + 4 > file.size()) { err = "HDR: truncated scanline"; return false; }
    const uint8_t* s = &file[pos];
    if (w >= 8 && w < 32768 && s[0] == 2 && s[1] == 2 && (s[2] << 8 | s[3]) == w) {  // new-style RLE: the four components separately
      pos += 4;
      for (int c = 0; c < 4; ++c) {
        int x = 0;
        while (x < w) {
          if (pos >= file.size()) { err = "HDR: truncated run-length data"; return false; }
          int cnt = file[pos++];
          if (cnt > 128) { cnt -= 128; if (pos >= file.size() || x + cnt > w) { err = "HDR: bad run"; return false; } const uint8_t v = file[pos++]; for (int k = 0; k < cnt; ++k) scan[(size_t)(x++) * 4 + c] = v; }
          else { if (cnt == 0 || pos + cnt > file.size() || x + cnt > w) { err = "HDR: bad run"; return false; } for (int k = 0; k < cnt; ++k) scan[(size_t)(x++) * 4 + c] = file[pos++]; }
        }
      }
    } else {  // flat (or old-style run-length: a pixel (1,1,1,n) repeats the previous one n << shift times)
      int x = 0, shift = 0;
      while (x < w) {
        if (pos + 4 > file.size()) { err = "HDR: truncated scanline"; return false; }
        const uint8_t* q = &file[pos]; pos += 4;
        if (q[0] == 1 && q[1] == 1 && q[2] == 1 && x > 0) { if (shift > 24) { err = "HDR: bad run"; return false; } const uint64_t cnt = (uint64_t)q[3] << shift; if ((uint64_t)x + cnt > (uint64_t)w) { err = "HDR: bad run"; return false; } for (uint64_t k = 0; k < cnt; ++k, ++x) memcpy(&scan[(size_t)x * 4], &scan[(size_t)(x - 1) * 4], 4); shift += 8; }
        else { memcpy(&scan[(size_t)x * 4], q, 4); ++x; shift = 0; }
      }
    }
    for (int x = 0; x < w; ++x) {  // Rgbe8Pixel::to_hdr: 0 for e == 0, else c * 2^(e - 136)
      const uint8_t* q = &scan[(size_t)x * 4]; float* o = &rgb[((size_t)y * w + x) * 3];
      if (q[3] == 0) { o[0] = o[1] = o[2] = 0.0f; continue; }
      const float e = std::exp2((float)q[3] - (128.0f + 8.0f));
      o[0] = e * (float)q[0]; o[1] = e * (float)q[1]; o[2] = e * (float)q[2];
    }
  }
  return true;
}

}  // namespace

// ------------------------------------------------------------------ OpenEXR (scan-line, single part) -> float RGB
// read_image_exr (rc/imageio.rs:134-160) calls exr 1.4.2's read_first_rgba_layer_from_file: the first layer that has R, G and B
// channels, samples converted to f32, stored at their position inside the layer. Restated from the OpenEXR file layout
// specification: magic + version, attribute list (channels, compression, dataWindow, displayWindow, lineOrder), offset table,
// scan-line blocks of 1 (NONE, RLE, ZIPS), 16 (ZIP, PXR24) or 32 (PIZ) lines - or tiles - holding each line's channels in alphabetical order. ZIP / ZIPS / RLE
// payloads are byte-delta coded and split into even / odd halves before compression; PIZ and PXR24 are described at their decoders. Tiled files give their
// full-resolution level (what exr 1.4.2 calls the largest resolution level), multi-part files their first flat part with R, G and B. B44 / B44A are decoded (below); the lossy
// DWAA / DWAB compressions and deep data are refused with a message. The reference takes the resolution from displayWindow and the pixels from the layer
// (imageio.rs:153-159), which only agree when the two windows do: files where they differ are refused.
namespace {

inline float half_to_float(uint16_t h) {
  const uint32_t sign = (uint32_t)(h & 0x8000u) << 16; uint32_t e = (h >> 10) & 0x1f, m = h & 0x3ff, bits;
  if (e == 0) {
    if (m == 0) bits = sign;
    else { int sh = 0; while (!(m & 0x400)) { m <<= 1; ++sh; } m &= 0x3ff; bits = sign | (uint32_t)(127 - 15 - sh + 1) << 23 | m << 13; }
  } else if (e == 31) bits = sign | 0x7f800000u | m << 13;
  else bits = sign | (e + 112) << 23 | m << 13;
  float f; memcpy(&f, &bits, 4); return f;
}

bool exr_unpredict(std::vector<uint8_t>& raw, size_t want, std::string& err) {  // undo the delta predictor, then re-interleave the two halves
  if (raw.size() != want) { err = "EXR: block decompressed to the wrong size"; return false; }
  for (size_t i = 1; i < raw.size(); ++i) raw[i] = (uint8_t)(raw[i - 1] + raw[i] - 128);
  std::vector<uint8_t> out(raw.size());
  const size_t half = (raw.size() + 1) / 2;
  for (size_t i = 0; i < raw.size(); ++i) out[i] = (i & 1) ? raw[half + i / 2] : raw[i / 2];
  raw.swap(out);
  return true;
}

// ---- PIZ (compression 4): per block a bitmap of the 16-bit values in use (-> a lookup table that packs them), a Haar-like 2-D wavelet transform of each
// channel's 16-bit planes, and a canonical Huffman code over the result with a run-length symbol. Restated from the format's published description
// (OpenEXR "ImfPizCompressor / ImfHuf / ImfWav": technical introduction + file layout documents).
// Attribution: the wavelet decode below - piz_wdec14, piz_wdec16, piz_wav_decode - follows OpenEXR's ImfWav.cpp (wdec14 / wdec16 / wav2Decode)
// step for step, loop nest and variable roles included; the lifting steps ARE the format, so a decoder cannot do anything else, but the shape of the code is
// theirs. OpenEXR is Copyright (c) Contributors to the OpenEXR Project / Industrial Light & Magic, distributed under the BSD-3-Clause licence: redistribution
// and use in source and binary forms, with or without modification, are permitted provided that the copyright notice, the list of conditions and the
// disclaimer of the licence are retained ("THIS SOFTWARE IS PROVIDED BY THE COPYRIGHT HOLDERS AND CONTRIBUTORS "AS IS" AND ANY EXPRESS OR IMPLIED
// WARRANTIES ... ARE DISCLAIMED"). Not part of /root/reference (which calls the `exr` crate); off the hot path.
struct ExrBits {  // MSB-first bit reader over [p, end)
  const uint8_t* p; const uint8_t* end; uint64_t c = 0; int lc = 0; bool ok = true;
  uint32_t get(int n) {
    while (lc < n) { if (p >= end) { ok = false; return 0; } c = (c << 8) | *p++; lc += 8; }
    lc -= n; return (uint32_t)((c >> lc) & ((1ull << n) - 1ull));
  }
};
bool piz_huf_uncompress(const uint8_t* src, size_t n_src, std::vector<uint16_t>& out, size_t n_out, std::string& err) {
  out.assign(n_out, 0);
  if (n_src == 0) { if (n_out) { err = "EXR: empty PIZ Huffman block"; return false; } return true; }
  if (n_src < 20) { err = "EXR: truncated PIZ Huffman header"; return false; }
  auto u32 = [&](size_t o) { return (uint32_t)src[o] | (uint32_t)src[o + 1] << 8 | (uint32_t)src[o + 2] << 16 | (uint32_t)src[o + 3] << 24; };
  const uint32_t im = u32(0), iM = u32(4), n_bits = u32(12);
  constexpr uint32_t ENC = 65537;  // 2^16 values + the run-length symbol
  if (im >= ENC || iM >= ENC || im > iM) { err = "EXR: bad PIZ Huffman table range"; return false; }
  // packed code lengths: 6 bits each; 59..62 = a run of 2..5 zero lengths, 63 = a run of 6 + (8 more bits) zeros
  std::vector<uint8_t> len(ENC, 0);
  ExrBits tb{src + 20, src + n_src};
  for (uint32_t i = im; i <= iM; ++i) {
    const uint32_t l = tb.get(6);
    if (!tb.ok) { err = "EXR: truncated PIZ Huffman table"; return false; }
    if (l == 63u || l >= 59u) {
      uint32_t run = l == 63u ? tb.get(8) + 6u : l - 59u + 2u;
      if (!tb.ok || i + run > iM + 1u) { err = "EXR: bad zero run in PIZ Huffman table"; return false; }
      i += run - 1u;  // (the lengths stay 0)
    } else len[i] = (uint8_t)l;
  }
  // canonical codes: within a length consecutive in symbol order, the longest lengths take the smallest code values
  uint64_t count[59] = {0}, first[59] = {0};
  for (uint32_t i = 0; i < ENC; ++i) count[len[i]] += 1;
  { uint64_t c = 0; for (int l = 58; l > 0; --l) { const uint64_t nc = (c + count[l]) >> 1; first[l] = c; c = nc; } }
  std::vector<uint32_t> sorted; sorted.reserve(ENC); size_t base[59] = {0};
  for (int l = 1; l < 59; ++l) { base[l] = sorted.size(); for (uint32_t i = 0; i < ENC; ++i) if (len[i] == l) sorted.push_back(i); }
  // the code data starts at the next byte after the table
  const uint8_t* data = tb.p;
  if ((uint64_t)(src + n_src - data) * 8ull < n_bits) { err = "EXR: truncated PIZ Huffman data"; return false; }
  ExrBits db{data, src + n_src};
  uint64_t used = 0; size_t o = 0;
  while (used < n_bits && o < n_out) {
    uint64_t v = 0; int l = 0; uint32_t sym = 0; bool hit = false;
    while (l < 58 && used < n_bits) {
      v = (v << 1) | db.get(1); ++l; ++used;
      if (count[l] && v >= first[l] && v - first[l] < count[l]) { sym = sorted[base[l] + (size_t)(v - first[l])]; hit = true; break; }
    }
    if (!hit || !db.ok) { err = "EXR: bad PIZ Huffman code"; return false; }
    if (sym == iM) {  // run-length symbol: repeat the previous value
      if (used + 8 > n_bits || o == 0) { err = "EXR: bad PIZ run"; return false; }
      const uint32_t run = db.get(8); used += 8;
      if (o + run > n_out) { err = "EXR: PIZ run past the end of the block"; return false; }
      const uint16_t prev = out[o - 1];
      for (uint32_t k = 0; k < run; ++k) out[o++] = prev;
    } else out[o++] = (uint16_t)sym;
  }
  if (o != n_out) { err = "EXR: PIZ block decodes to the wrong size"; return false; }
  return true;
}
inline void piz_wdec14(uint16_t l, uint16_t h, uint16_t& a, uint16_t& b) {
  const int ls = (int16_t)l, hs = (int16_t)h;
  const int ai = ls + (hs & 1) + (hs >> 1);
  a = (uint16_t)(int16_t)ai; b = (uint16_t)(int16_t)(ai - hs);
}
inline void piz_wdec16(uint16_t l, uint16_t h, uint16_t& a, uint16_t& b) {
  const int m = l, d = h;
  const int bb = (m - (d >> 1)) & 0xffff;
  const int aa = (d + bb - 0x8000) & 0xffff;
  b = (uint16_t)bb; a = (uint16_t)aa;
}
void piz_wav_decode(uint16_t* in, int nx, int ox, int ny, int oy, uint16_t mx) {
  const bool w14 = mx < (1u << 14);
  const int n = nx > ny ? ny : nx;
  int p = 1; while (p <= n) p <<= 1;
  p >>= 1; int p2 = p; p >>= 1;
  auto dec = [&](uint16_t l, uint16_t h, uint16_t& a, uint16_t& b) { if (w14) piz_wdec14(l, h, a, b); else piz_wdec16(l, h, a, b); };
  while (p >= 1) {
    uint16_t* py = in; uint16_t* const ey = in + (ptrdiff_t)oy * (ny - p2);
    const ptrdiff_t oy1 = (ptrdiff_t)oy * p, oy2 = (ptrdiff_t)oy * p2, ox1 = (ptrdiff_t)ox * p, ox2 = (ptrdiff_t)ox * p2;
    for (; py <= ey; py += oy2) {
      uint16_t* px = py; uint16_t* const ex = py + (ptrdiff_t)ox * (nx - p2);
      for (; px <= ex; px += ox2) {
        uint16_t* p01 = px + ox1; uint16_t* p10 = px + oy1; uint16_t* p11 = p10 + ox1;
        uint16_t i00, i01, i10, i11;
        dec(*px, *p10, i00, i10); dec(*p01, *p11, i01, i11);
        dec(i00, i01, *px, *p01); dec(i10, i11, *p10, *p11);
      }
      if (nx & p) { uint16_t* p10 = px + oy1; uint16_t i00; dec(*px, *p10, i00, *p10); *px = i00; }
    }
    if (ny & p) {
      uint16_t* px = py; uint16_t* const ex = py + (ptrdiff_t)ox * (nx - p2);
      for (; px <= ex; px += ox2) { uint16_t* p01 = px + ox1; uint16_t i00; dec(*px, *p01, i00, *p01); *px = i00; }
    }
    p2 = p; p >>= 1;
  }
}
struct ExrChan { std::string name; int type, xs, ys; int plinear = 0; };
// one block (rows x w pixels, every channel) of PIZ data -> the uncompressed block layout (per row, per channel, little-endian samples)
bool piz_uncompress(const uint8_t* src, size_t size, const std::vector<ExrChan>& chans, size_t w, size_t rows, std::vector<uint8_t>& raw, std::string& err) {
  if (size < 4) { err = "EXR: truncated PIZ block"; return false; }
  const unsigned min_nz = src[0] | src[1] << 8, max_nz = src[2] | src[3] << 8;
  size_t pos = 4;
  std::vector<uint8_t> bitmap(8192, 0);
  if (min_nz <= max_nz) {
    if (max_nz >= 8192 || pos + (max_nz - min_nz + 1) > size) { err = "EXR: bad PIZ bitmap"; return false; }
    memcpy(bitmap.data() + min_nz, src + pos, max_nz - min_nz + 1); pos += max_nz - min_nz + 1;
  }
  std::vector<uint16_t> lut(65536, 0); unsigned k = 0;
  for (unsigned i = 0; i < 65536; ++i) if (i == 0 || (bitmap[i >> 3] & (1u << (i & 7)))) lut[k++] = (uint16_t)i;
  const uint16_t max_value = (uint16_t)(k - 1);
  if (pos + 4 > size) { err = "EXR: truncated PIZ block"; return false; }
  const uint32_t hlen = (uint32_t)src[pos] | (uint32_t)src[pos + 1] << 8 | (uint32_t)src[pos + 2] << 16 | (uint32_t)src[pos + 3] << 24; pos += 4;
  if (hlen > size - pos) { err = "EXR: PIZ Huffman data longer than the block"; return false; }
  size_t total = 0; std::vector<size_t> start(chans.size());
  for (size_t c = 0; c < chans.size(); ++c) { start[c] = total; total += w * rows * (chans[c].type == 1 ? 1 : 2); }
  std::vector<uint16_t> tmp;
  if (!piz_huf_uncompress(src + pos, hlen, tmp, total, err)) return false;
  for (size_t c = 0; c < chans.size(); ++c) {
    const int sz = chans[c].type == 1 ? 1 : 2;
    for (int j = 0; j < sz; ++j) piz_wav_decode(tmp.data() + start[c] + j, (int)w, sz, (int)rows, (int)w * sz, max_value);
  }
  for (uint16_t& v : tmp) v = lut[v];
  raw.clear(); raw.reserve(total * 2);
  std::vector<size_t> cur = start;
  for (size_t r = 0; r < rows; ++r)
    for (size_t c = 0; c < chans.size(); ++c) {
      const size_t n = w * (chans[c].type == 1 ? 1 : 2);
      for (size_t i = 0; i < n; ++i) { const uint16_t v = tmp[cur[c] + i]; raw.push_back((uint8_t)(v & 0xff)); raw.push_back((uint8_t)(v >> 8)); }
      cur[c] += n;
    }
  return true;
}
// PXR24 (compression 5): zlib over, per row and channel, the byte planes (most significant first) of the differences between neighbouring samples;
// 32-bit floats keep their upper 24 bits
bool pxr24_uncompress(const uint8_t* src, size_t size, const std::vector<ExrChan>& chans, size_t w, size_t rows, std::vector<uint8_t>& raw, std::string& err) {
  std::vector<uint8_t> z;
  if (!zlib_inflate(src, size, z, err)) return false;
  size_t want = 0; for (const ExrChan& c : chans) want += w * rows * (c.type == 1 ? 2 : (c.type == 2 ? 3 : 4));
  if (z.size() != want) { err = "EXR: PXR24 block decompressed to the wrong size"; return false; }
  raw.clear(); size_t p = 0;
  for (size_t r = 0; r < rows; ++r)
    for (const ExrChan& c : chans) {
      const int planes = c.type == 1 ? 2 : (c.type == 2 ? 3 : 4);
      uint32_t pixel = 0;
      for (size_t x = 0; x < w; ++x) {
        uint32_t diff = 0;
        for (int k = 0; k < planes; ++k) diff = (diff << 8) | z[p + (size_t)k * w + x];
        if (c.type == 2) diff <<= 8;
        pixel += diff;
        if (c.type == 1) { raw.push_back((uint8_t)(pixel & 0xff)); raw.push_back((uint8_t)((pixel >> 8) & 0xff)); }
        else { raw.push_back((uint8_t)(pixel & 0xff)); raw.push_back((uint8_t)((pixel >> 8) & 0xff)); raw.push_back((uint8_t)((pixel >> 16) & 0xff)); raw.push_back((uint8_t)(pixel >> 24)); }
      }
      p += (size_t)planes * w;
    }
  return true;
}

// B44 / B44A (compression 6 / 7): lossy, fixed rate. The block's channels are stored one after the other (not interleaved by row). A HALF channel is cut into
// 4 x 4 pixel cells (rows and columns past the edge repeat the last one when packing; only the part inside is kept here), each cell packed into 14 bytes: the
// first value as 16 bits, a 6-bit shift, and fifteen 6-bit running differences - down the first column, then along each row - in units of 2^shift, biased by 32;
// B44A writes a cell whose sixteen values are equal as 3 bytes (value, then 0xfc where the shift would be). Values travel in an order-preserving 16-bit form
// (sign bit flipped for positives, all bits for negatives). Channels flagged pLinear are stored as exp(x / 8) and come back through 8 ln(x). FLOAT and UINT
// channels are stored raw. After the OpenEXR library's ImfB44Compressor, which exr 1.4.2's b44 module restates.
inline uint16_t float_to_half_rne(float f) {
  uint32_t x; memcpy(&x, &f, 4);
  const uint32_t sign = (x >> 16) & 0x8000u; x &= 0x7fffffffu;
  if (x >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | (x > 0x7f800000u ? 0x200u | ((x >> 13) & 0x3ffu) : 0u));
  if (x >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);                 // rounds to infinity
  if (x < 0x33000001u) return (uint16_t)sign;                               // rounds to zero
  if (x < 0x38800000u) {                                                    // subnormal half
    const int e = (int)(x >> 23); const uint32_t m = (x & 0x7fffffu) | 0x800000u; const int sh = 126 - e;   // value = m * 2^(e - 150); half ulp 2^-24
    const uint32_t q = m >> sh, rem = m & ((1u << sh) - 1u), half = 1u << (sh - 1);
    return (uint16_t)(sign | (q + ((rem > half || (rem == half && (q & 1u))) ? 1u : 0u)));
  }
  const uint32_t m = x & 0x1fffu, base = (x - 0x38000000u) >> 13;
  return (uint16_t)(sign | (base + ((m > 0x1000u || (m == 0x1000u && (base & 1u))) ? 1u : 0u)));
}
inline void b44_unpack14(const uint8_t* b, uint16_t* s) {
  s[0] = (uint16_t)((b[0] << 8) | b[1]);
  const unsigned shift = b[2] >> 2; const unsigned bias = 0x20u << shift;
  s[4] = (uint16_t)(s[0] + ((((b[2] << 4) | (b[3] >> 4)) & 0x3fu) << shift) - bias);
  s[8] = (uint16_t)(s[4] + ((((b[3] << 2) | (b[4] >> 6)) & 0x3fu) << shift) - bias);
  s[12] = (uint16_t)(s[8] + ((b[4] & 0x3fu) << shift) - bias);
  s[1] = (uint16_t)(s[0] + ((unsigned)(b[5] >> 2) << shift) - bias);
  s[5] = (uint16_t)(s[4] + ((((b[5] << 4) | (b[6] >> 4)) & 0x3fu) << shift) - bias);
  s[9] = (uint16_t)(s[8] + ((((b[6] << 2) | (b[7] >> 6)) & 0x3fu) << shift) - bias);
  s[13] = (uint16_t)(s[12] + ((b[7] & 0x3fu) << shift) - bias);
  s[2] = (uint16_t)(s[1] + ((unsigned)(b[8] >> 2) << shift) - bias);
  s[6] = (uint16_t)(s[5] + ((((b[8] << 4) | (b[9] >> 4)) & 0x3fu) << shift) - bias);
  s[10] = (uint16_t)(s[9] + ((((b[9] << 2) | (b[10] >> 6)) & 0x3fu) << shift) - bias);
  s[14] = (uint16_t)(s[13] + ((b[10] & 0x3fu) << shift) - bias);
  s[3] = (uint16_t)(s[2] + ((unsigned)(b[11] >> 2) << shift) - bias);
  s[7] = (uint16_t)(s[6] + ((((b[11] << 4) | (b[12] >> 4)) & 0x3fu) << shift) - bias);
  s[11] = (uint16_t)(s[10] + ((((b[12] << 2) | (b[13] >> 6)) & 0x3fu) << shift) - bias);
  s[15] = (uint16_t)(s[14] + ((b[13] & 0x3fu) << shift) - bias);
  for (int i = 0; i < 16; ++i) s[i] = (s[i] & 0x8000u) ? (uint16_t)(s[i] & 0x7fffu) : (uint16_t)~s[i];
}
// 8 ln(x) per half bit pattern (negative / non-finite -> 0): what a pLinear channel comes back through. Built on first use (a function-local static: once, thread-safe)
inline const std::vector<uint16_t>& b44_log_table() {
  static const std::vector<uint16_t> table = [] {
    std::vector<uint16_t> t(65536);
    for (uint32_t i = 0; i < 65536; ++i) {
      const bool finite = (i & 0x7c00u) != 0x7c00u; const float h = half_to_float((uint16_t)i);
      t[i] = (!finite || h < 0.0f) ? (uint16_t)0 : float_to_half_rne((float)(8.0 * std::log((double)h)));
    }
    return t;
  }();
  return table;
}
bool b44_uncompress(const uint8_t* src, size_t size, const std::vector<ExrChan>& chans, size_t w, size_t rows, std::vector<uint8_t>& raw, std::string& err) {
  size_t line_bytes = 0; std::vector<size_t> chan_off(chans.size());
  for (size_t k = 0; k < chans.size(); ++k) { chan_off[k] = line_bytes; line_bytes += w * (chans[k].type == 1 ? 2 : 4); }
  raw.assign(rows * line_bytes, 0);
  size_t p = 0;
  for (size_t k = 0; k < chans.size(); ++k) {
    const ExrChan& c = chans[k];
    if (c.type != 1) {  // FLOAT / UINT: the channel's rows, raw
      const size_t n = w * 4;
      if (p + rows * n > size) { err = "EXR: truncated B44 block"; return false; }
      for (size_t r = 0; r < rows; ++r) { memcpy(&raw[r * line_bytes + chan_off[k]], src + p, n); p += n; }
      continue;
    }
    for (size_t y = 0; y < rows; y += 4)
      for (size_t x = 0; x < w; x += 4) {
        uint16_t s[16];
        if (p + 3 > size) { err = "EXR: truncated B44 block"; return false; }
        if (src[p + 2] >= (13 << 2)) {  // a shift field >= 13 marks a 3-byte flat cell (ImfB44Compressor: the encoder writes 0xfc, any such value decodes as flat)
          uint16_t v = (uint16_t)((src[p] << 8) | src[p + 1]); v = (v & 0x8000u) ? (uint16_t)(v & 0x7fffu) : (uint16_t)~v;
          for (int i = 0; i < 16; ++i) s[i] = v;
          p += 3;
        } else {
          if (p + 14 > size) { err = "EXR: truncated B44 block"; return false; }
          b44_unpack14(src + p, s); p += 14;
        }
        if (c.plinear) { const std::vector<uint16_t>& log_table = b44_log_table(); for (int i = 0; i < 16; ++i) s[i] = log_table[s[i]]; }
        for (size_t dy = 0; dy < 4 && y + dy < rows; ++dy)
          for (size_t dx = 0; dx < 4 && x + dx < w; ++dx) {
            uint8_t* q = &raw[(y + dy) * line_bytes + chan_off[k] + 2 * (x + dx)];
            q[0] = (uint8_t)(s[4 * dy + dx] & 0xff); q[1] = (uint8_t)(s[4 * dy + dx] >> 8);
          }
      }
  }
  if (p != size) { err = "EXR: B44 block longer than its cells"; return false; }
  return true;
}

struct ExrPart {
  std::vector<ExrChan> chans; int compression = -1, line_order = 0; int32_t dw[4] = {0, 0, -1, -1}, disp[4] = {0, 0, -1, -1}; bool have_dw = false, have_disp = false;
  bool tiled = false, deep = false; uint32_t tile_w = 0, tile_h = 0; int level_mode = 0, rounding = 0; long long chunk_count = -1; std::string type;
};
bool exr_decode(const std::vector<uint8_t>& file, int& W, int& H, std::vector<float>& rgb, std::string& err) {
  Bytes b{file.data(), file.size()};
  auto le32 = [&]() -> uint32_t { uint32_t v = (uint32_t)b.p[b.pos] | (uint32_t)b.p[b.pos + 1] << 8 | (uint32_t)b.p[b.pos + 2] << 16 | (uint32_t)b.p[b.pos + 3] << 24; b.pos += 4; return v; };
  auto cstr = [&](std::string& out) -> bool { out.clear(); while (b.pos < b.n && b.p[b.pos]) out += (char)b.p[b.pos++]; if (b.pos >= b.n) return false; ++b.pos; return true; };
  if (!b.need(8) || le32() != 20000630u) { err = "EXR: bad magic number"; return false; }
  const uint32_t version = le32();
  if ((version & 0xff) != 2) { err = "EXR: unsupported file version"; return false; }
  const bool multipart = (version & 0x1000) != 0;
  if (version & 0x800) { err = "EXR: deep images are not supported"; return false; }
  // headers: one, or (multi-part) several, each ended by an empty attribute name, the list by an empty header
  std::vector<ExrPart> parts;
  for (;;) {
    ExrPart pt; pt.tiled = !multipart && (version & 0x200) != 0;
    bool any = false;
    for (;;) {
      std::string name, type;
      if (!cstr(name)) { err = "EXR: truncated header"; return false; }
      if (name.empty()) break;
      any = true;
      if (!cstr(type) || !b.need(4)) { err = "EXR: truncated header"; return false; }
      const uint32_t size = le32();
      if (!b.need(size)) { err = "EXR: truncated attribute"; return false; }
      const size_t end = b.pos + size;
      if (name == "channels") {
        while (b.pos < end && b.p[b.pos]) {
          ExrChan c; if (!cstr(c.name) || b.pos + 16 > end) { err = "EXR: bad channel list"; return false; }
          c.type = (int)le32(); c.plinear = b.p[b.pos] != 0; b.pos += 4; c.xs = (int)le32(); c.ys = (int)le32();
          if (c.type < 0 || c.type > 2) { err = "EXR: unknown pixel type"; return false; }
          if (c.xs != 1 || c.ys != 1) { err = "EXR: sub-sampled channels are not supported"; return false; }
          pt.chans.push_back(c);
        }
      } else if (name == "compression" && size >= 1) pt.compression = b.p[b.pos];
      else if (name == "lineOrder" && size >= 1) pt.line_order = b.p[b.pos];
      else if ((name == "dataWindow" || name == "displayWindow") && size == 16) {
        int32_t* w = name == "dataWindow" ? pt.dw : pt.disp; (name == "dataWindow" ? pt.have_dw : pt.have_disp) = true;
        for (int k = 0; k < 4; ++k) { uint32_t v = le32(); memcpy(&w[k], &v, 4); }
      } else if (name == "tiles" && size == 9) { pt.tile_w = le32(); pt.tile_h = le32(); pt.level_mode = b.p[b.pos] & 0xf; pt.rounding = b.p[b.pos] >> 4; }
      else if (name == "type" && type == "string") { pt.type.assign((const char*)b.p + b.pos, size); pt.tiled = pt.type == "tiledimage"; pt.deep = pt.type.compare(0, 4, "deep") == 0; }
      else if (name == "chunkCount" && size == 4) { pt.chunk_count = (int32_t)le32(); }
      b.pos = end;
    }
    if (!any) { if (!multipart || parts.empty()) { err = "EXR: empty header"; return false; } break; }
    parts.push_back(pt);
    if (!multipart) break;
  }
  // chunks per part: what its offset table holds
  auto levels = [](long long n, int rounding) { int l = 1; while (n > 1) { n = rounding ? (n + 1) / 2 : n / 2; ++l; } return l; };
  auto level_size = [](long long n, int l, int rounding) { for (int k = 0; k < l; ++k) n = std::max<long long>(1, rounding ? (n + 1) / 2 : n / 2); return n; };
  auto lines_of = [](int c) { return c == 3 || c == 5 ? 16 : (c == 4 || c == 6 || c == 7 ? 32 : (c == 8 ? 32 : (c == 9 ? 256 : 1))); };
  std::vector<long long> n_chunks(parts.size(), 0);
  for (size_t k = 0; k < parts.size(); ++k) {
    const ExrPart& pt = parts[k];
    if (pt.chans.empty() || pt.compression < 0 || !pt.have_dw || !pt.have_disp) { err = "EXR: missing required attribute"; return false; }
    const long long w = (long long)pt.dw[2] - pt.dw[0] + 1, h = (long long)pt.dw[3] - pt.dw[1] + 1;
    if (w <= 0 || h <= 0 || w > 65536 || h > 65536 || w * h > (1ll << 28)) { err = "EXR: bad data window"; return false; }
    if (pt.tiled) {
      if (pt.tile_w == 0 || pt.tile_h == 0 || pt.tile_w > 65536 || pt.tile_h > 65536) { err = "EXR: bad tile description"; return false; }
      if (pt.level_mode > 2) { err = "EXR: unknown level mode"; return false; }
      long long total = 0;
      const int nlx = pt.level_mode == 0 ? 1 : levels(pt.level_mode == 1 ? std::max(w, h) : w, pt.rounding), nly = pt.level_mode == 2 ? levels(h, pt.rounding) : nlx;
      if (pt.level_mode == 2) { for (int ly = 0; ly < nly; ++ly) for (int lx = 0; lx < nlx; ++lx) total += ((level_size(w, lx, pt.rounding) + pt.tile_w - 1) / pt.tile_w) * ((level_size(h, ly, pt.rounding) + pt.tile_h - 1) / pt.tile_h); }
      else for (int l = 0; l < nlx; ++l) total += ((level_size(w, l, pt.rounding) + pt.tile_w - 1) / pt.tile_w) * ((level_size(h, l, pt.rounding) + pt.tile_h - 1) / pt.tile_h);
      n_chunks[k] = total;
    } else n_chunks[k] = (h + lines_of(pt.compression) - 1) / lines_of(pt.compression);
    if (multipart) { if (pt.chunk_count < 0) { err = "EXR: a part of a multi-part file lacks chunkCount"; return false; } n_chunks[k] = pt.chunk_count; }
  }
  // the first part (exr 1.4.2 read_first_rgba_layer_from_file: the first flat layer) that has R, G and B channels
  int ci[3] = {-1, -1, -1}; int part = -1;
  for (size_t k = 0; k < parts.size() && part < 0; ++k) {
    if (parts[k].deep) continue;
    const std::vector<ExrChan>& chans = parts[k].chans; bool found = false;
    auto triple = [&](const std::string& pre) {
      int r = -1, g = -1, bl = -1;
      for (size_t q = 0; q < chans.size(); ++q) { if (chans[q].name == pre + "R") r = (int)q; if (chans[q].name == pre + "G") g = (int)q; if (chans[q].name == pre + "B") bl = (int)q; }
      if (r >= 0 && g >= 0 && bl >= 0) { ci[0] = r; ci[1] = g; ci[2] = bl; found = true; }
    };
    triple("");
    for (size_t q = 0; q < chans.size() && !found; ++q) {
      const std::string& nm = chans[q].name;
      if (nm.size() >= 2 && nm.compare(nm.size() - 2, 2, ".R") == 0) triple(nm.substr(0, nm.size() - 1));
    }
    if (found) part = (int)k;
  }
  if (part < 0) { err = "EXR: no layer with R, G and B channels"; return false; }
  const ExrPart& pt = parts[part];
  const std::vector<ExrChan>& chans = pt.chans; const int compression = pt.compression; const int32_t* dw = pt.dw; const int32_t* disp = pt.disp;
  if (compression > 7) { err = "EXR: DWAA / DWAB compression is not supported"; return false; }
  if (pt.line_order > 2) { err = "EXR: unsupported line order"; return false; }
  const long long w = (long long)dw[2] - dw[0] + 1, h = (long long)dw[3] - dw[1] + 1;
  if (disp[2] - disp[0] != dw[2] - dw[0] || disp[3] - disp[1] != dw[3] - dw[1]) { err = "EXR: data window and display window differ in size"; return false; }
  // offset tables: one per part, in header order
  size_t table = b.pos, all = 0;
  for (size_t k = 0; k < parts.size(); ++k) {
    if (n_chunks[k] < 0 || (unsigned long long)n_chunks[k] > file.size() / 8) { err = "EXR: truncated offset table"; return false; }
    all += (size_t)n_chunks[k] * 8;
    if ((int)k < part) table += (size_t)n_chunks[k] * 8;
  }
  if (b.pos + all > file.size()) { err = "EXR: truncated offset table"; return false; }
  W = (int)w; H = (int)h; rgb.assign((size_t)w * h * 3, 0.0f);
  // chunks that hold full-resolution pixels: every scan-line block, or the tiles of level (0, 0) - which the offset table lists first
  const long long tiles_x = pt.tiled ? (w + pt.tile_w - 1) / pt.tile_w : 1, tiles_y = pt.tiled ? (h + pt.tile_h - 1) / pt.tile_h : 0;
  const size_t n_read = pt.tiled ? (size_t)(tiles_x * tiles_y) : (size_t)n_chunks[part];
  // the part's own table must hold every chunk that is read (a multi-part file states its chunkCount; a patched one could name fewer entries than the data
  // window needs and the loop below would index past the table - and past the file); a scan-line part has exactly ceil(h / lines per block) blocks
  if ((unsigned long long)n_chunks[part] < (unsigned long long)n_read ||
      (!pt.tiled && (long long)n_chunks[part] != (h + lines_of(compression) - 1) / lines_of(compression))) { err = "EXR: truncated offset table"; return false; }
  const int lines_per_block = lines_of(compression);
  std::vector<uint8_t> raw;
  for (size_t blk = 0; blk < n_read; ++blk) {
    uint64_t off = 0; for (int k = 7; k >= 0; --k) off = off << 8 | file[table + blk * 8 + k];
    const size_t head = (multipart ? 4u : 0u) + (pt.tiled ? 20u : 8u);
    if (file.size() < head || off > file.size() - head) { err = "EXR: block offset outside the file"; return false; }  // (off + head would wrap for offsets near 2^64)
    b.pos = (size_t)off;
    if (multipart) { const uint32_t pn = le32(); if ((int)pn != part) { err = "EXR: chunk of another part in this part's offset table"; return false; } }
    long long x0 = 0, y0, cols = w, rows;
    if (pt.tiled) {
      const uint32_t tx = le32(), ty = le32(), lx = le32(), ly = le32();
      if (lx != 0 || ly != 0 || (long long)tx >= tiles_x || (long long)ty >= tiles_y) { err = "EXR: unexpected tile in the full-resolution level"; return false; }
      x0 = (long long)tx * pt.tile_w; y0 = (long long)ty * pt.tile_h;
      cols = std::min<long long>(pt.tile_w, w - x0); rows = std::min<long long>(pt.tile_h, h - y0);
    } else {
      int32_t yy; { uint32_t v = le32(); memcpy(&yy, &v, 4); }
      if (yy < dw[1] || yy > dw[3]) { err = "EXR: block outside the data window"; return false; }
      y0 = (long long)yy - dw[1]; rows = std::min<long long>(lines_per_block, h - y0);
    }
    const uint32_t size = le32();
    if (!b.need(size)) { err = "EXR: truncated block"; return false; }
    std::vector<size_t> chan_off(chans.size()); size_t line_bytes = 0;
    for (size_t k = 0; k < chans.size(); ++k) { chan_off[k] = line_bytes; line_bytes += (size_t)cols * (chans[k].type == 1 ? 2 : 4); }
    const size_t want = (size_t)rows * line_bytes;
    const uint8_t* src = b.p + b.pos;
    if (size == want) raw.assign(src, src + size);   // stored uncompressed (always for NONE; for the others when compression did not help)
    else if (compression == 2 || compression == 3) { raw.clear(); if (!zlib_inflate(src, size, raw, err) || !exr_unpredict(raw, want, err)) return false; }
    else if (compression == 1) {
      raw.clear(); size_t i = 0;
      while (i < size) {
        const int c = (int8_t)src[i++];
        if (c < 0) { const size_t n = (size_t)(-c); if (i + n > size) { err = "EXR: bad RLE run"; return false; } raw.insert(raw.end(), src + i, src + i + n); i += n; }
        else { if (i >= size) { err = "EXR: bad RLE run"; return false; } raw.insert(raw.end(), (size_t)c + 1, src[i]); ++i; }
      }
      if (!exr_unpredict(raw, want, err)) return false;
    } else if (compression == 4) { if (!piz_uncompress(src, size, chans, (size_t)cols, (size_t)rows, raw, err)) return false; }
    else if (compression == 5) { if (!pxr24_uncompress(src, size, chans, (size_t)cols, (size_t)rows, raw, err)) return false; }
    else if (compression == 6 || compression == 7) { if (!b44_uncompress(src, size, chans, (size_t)cols, (size_t)rows, raw, err)) return false; }
    else { err = "EXR: compressed block in an uncompressed file"; return false; }
    if (raw.size() != want) { err = "EXR: block decompressed to the wrong size"; return false; }
    for (size_t r = 0; r < (size_t)rows; ++r) {
      const size_t y = (size_t)y0 + r;
      for (int c = 0; c < 3; ++c) {
        const ExrChan& ch = chans[ci[c]]; const uint8_t* q = raw.data() + r * line_bytes + chan_off[ci[c]];
        for (size_t x = 0; x < (size_t)cols; ++x) {
          float v;
          if (ch.type == 1) v = half_to_float((uint16_t)(q[2 * x] | q[2 * x + 1] << 8));
          else if (ch.type == 2) memcpy(&v, q + 4 * x, 4);
          else { uint32_t u; memcpy(&u, q + 4 * x, 4); v = (float)u; }
          rgb[(y * (size_t)w + (size_t)x0 + x) * 3 + c] = v;
        }
      }
    }
  }
  return true;
}

}  // namespace

static int image_read_unguarded(const char* path, int32_t* width, int32_t* height, float** rgb);
extern "C" int rtxh_image_read(const char* path, int32_t* width, int32_t* height, float** rgb) {
  if (!path || !width || !height || !rgb) return fail(RT_ERR_INVALID, "null argument");
  g_err.clear();
  try { return image_read_unguarded(path, width, height, rgb); }  // no C++ exception may cross the C ABI
  catch (const std::bad_alloc&) { return fail(RT_ERR_OOM, std::string("out of memory while decoding ") + path); }
  catch (const std::exception& e) { return fail(RT_ERR_INVALID, std::string(e.what()) + " (" + path + ")"); }
}
static int image_read_unguarded(const char* path, int32_t* width, int32_t* height, float** rgb) {
  const std::string p = path; const size_t dot = p.find_last_of('.'), slash = p.find_last_of('/');
  if (dot == std::string::npos || (slash != std::string::npos && dot < slash)) return fail(RT_ERR_INVALID, "Texture filename doesn't have an extension");  // imageio.rs:19-21
  const std::string ext = p.substr(dot + 1);
  if (ext == "pfm") return rtxh_pfm_read(path, width, height, rgb);
  const bool ldr = ext == "tga" || ext == "TGA" || ext == "png" || ext == "PNG", exr = ext == "exr" || ext == "EXR";
  if (!ldr && !exr && ext != "hdr") return fail(RT_ERR_INVALID, "Unsupported file format");  // imageio.rs:30-32
  std::vector<uint8_t> file;
  if (!read_whole_file(path, file)) return fail(RT_ERR_INVALID, std::string("cannot open ") + path);
  int w = 0, h = 0; std::string err; std::vector<float> f;
  if (ldr) {
    std::vector<uint8_t> px;
    // image::open sniffs the content first and falls back on the extension (TGA has no signature)
    static const uint8_t png_sig[4] = {137, 80, 78, 71};
    const bool is_png = file.size() >= 4 && memcmp(file.data(), png_sig, 4) == 0;
    if (!(is_png ? png_decode(file, w, h, px, err) : tga_decode(file, w, h, px, err))) return fail(RT_ERR_INVALID, err + " (" + p + ")");
    f.resize(px.size());
    for (size_t i = 0; i < px.size(); ++i) f[i] = (float)px[i] / 255.0f;  // imageio.rs:104-106
  } else if (exr) {
    if (!exr_decode(file, w, h, f, err)) return fail(err.find("not supported") != std::string::npos ? RT_ERR_UNSUPPORTED : RT_ERR_INVALID, err + " (" + p + ")");
  } else if (!hdr_decode(file, w, h, f, err)) return fail(RT_ERR_INVALID, err + " (" + p + ")");
  float* out = (float*)malloc(f.size() * sizeof(float));
  if (!out) return fail(RT_ERR_INVALID, "out of memory");
  memcpy(out, f.data(), f.size() * sizeof(float));
  *width = w; *height = h; *rgb = out;
  return RT_OK;
}
