// png_read.hip -- host-side staging for sweeps: EuRoC camera frames (8-bit greyscale PNG, one file per frame) decoded on
// host threads straight into the caller's image batch (reference: src/streaming/dataset.py:93-158 reads them with
// cv2.imread(path, -1) on a reader thread per sensor; SURVEY 8f.1 "PNG decode on host threads").  Host code only: inflate
// + the five PNG row filters.  No device, no context; thread-safe (each file has its own state).
//
// A 64-stream sweep is bound by this file (round 4: 4.3 ms per 752 x 480 frame on one core, 2.2 of them zlib's inflate and 1.8 the
// Paeth filter), so both halves are done the fast way:
//   * inflate + Adler-32 + chunk CRCs by libdeflate when the system has it (libdeflate.so.0, loaded at run time -- the image
//     ships the library but not its header, so the four entry points used are declared here); zlib otherwise.  Same standard,
//     same bytes out, the same failures reported.
//   * Paeth / Average / Sub rows carry a dependency from pixel to pixel (one byte per pixel: nothing for SIMD to do along a
//     row), but row r + 1 at column x only needs row r up to x: eight consecutive Paeth rows are walked together, one column
//     apart, as the lanes of one SSE2 register (unfilter_paeth8; four rows in scalar registers where fewer than eight line up).
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <dlfcn.h>
#include <emmintrin.h>
#include <zlib.h>
#include <vector>
#include <omp.h>
#include "av_common.h"

namespace {

inline uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

// libdeflate's public C API (libdeflate.h, v1.x): opaque decompressor, one-shot zlib-wrapper decompression, CRC-32
struct LibDeflate {
    void* (*alloc)(void) = nullptr;
    int (*zlib_decompress)(void*, const void*, size_t, void*, size_t, size_t*) = nullptr;   // 0 = LIBDEFLATE_SUCCESS
    void (*free_)(void*) = nullptr;
    uint32_t (*crc32)(uint32_t, const void*, size_t) = nullptr;
    bool ok = false;
    LibDeflate()
    {
        if (getenv("AV_PNG_ZLIB")) return;                    // A/B and test switch: force the zlib path
        void* h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
        if (!h) return;
        alloc = reinterpret_cast<void* (*)(void)>(dlsym(h, "libdeflate_alloc_decompressor"));
        zlib_decompress = reinterpret_cast<int (*)(void*, const void*, size_t, void*, size_t, size_t*)>(dlsym(h, "libdeflate_zlib_decompress"));
        free_ = reinterpret_cast<void (*)(void*)>(dlsym(h, "libdeflate_free_decompressor"));
        crc32 = reinterpret_cast<uint32_t (*)(uint32_t, const void*, size_t)>(dlsym(h, "libdeflate_crc32"));
        ok = alloc && zlib_decompress && free_ && crc32;
    }
};
const LibDeflate& libdeflate() { static const LibDeflate L; return L; }

inline uint32_t chunk_crc(const uint8_t* p, size_t n)
{
    const LibDeflate& L = libdeflate();
    return L.ok ? L.crc32(0, p, n) : (uint32_t)crc32(crc32(0L, Z_NULL, 0), p, (uInt)n);
}

// the whole zlib stream (all IDAT payloads, in order) -> exactly `want` bytes; false: corrupt, truncated, or longer than the image
bool inflate_all(const uint8_t* z, size_t zn, uint8_t* raw, size_t want)
{
    const LibDeflate& L = libdeflate();
    if (L.ok) {
        static thread_local void* d = nullptr;               // one decompressor per decode thread, kept (libdeflate: not shareable, reusable)
        if (!d) d = L.alloc();
        if (!d) return false;
        size_t got = 0;
        const int rc = L.zlib_decompress(d, z, zn, raw, want, &got);    // checks the Adler-32; a stream that holds more than `want` fails (no space)
        return rc == 0 && got == want;
    }
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    if (inflateInit(&zs) != Z_OK) return false;
    uint8_t slack[8];
    zs.next_in = const_cast<Bytef*>(z); zs.avail_in = (uInt)zn;
    zs.next_out = raw; zs.avail_out = (uInt)want;
    int zrc = inflate(&zs, Z_NO_FLUSH);
    if (zrc == Z_OK && zs.avail_out == 0) {                  // the image is complete: only the end marker / Adler-32 may follow
        zs.next_out = slack; zs.avail_out = sizeof(slack);
        zrc = inflate(&zs, Z_NO_FLUSH);
        if (zs.avail_out != sizeof(slack)) zrc = Z_DATA_ERROR;
    }
    const bool good = zrc == Z_STREAM_END && zs.total_out == want;
    inflateEnd(&zs);
    return good;
}

inline int paeth(int a, int b, int c)
{
    const int p = b - c, q = a - c;
    const int pa = p < 0 ? -p : p, pb = q < 0 ? -q : q, pc = p + q < 0 ? -(p + q) : p + q;
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

inline void paeth_span(const uint8_t* in, const uint8_t* up, uint8_t* o, int x0, int x1)
{
    int a = x0 > 0 ? o[x0 - 1] : 0, c = x0 > 0 ? up[x0 - 1] : 0;
    for (int x = x0; x < x1; ++x) { const int b = up[x]; a = (uint8_t)(in[x] + paeth(a, b, c)); o[x] = (uint8_t)a; c = b; }
}
// Eight consecutive Paeth rows (not the image's first; `in` = filter byte of the first of them, o = its output row, width >= 16) in
// the 16-bit lanes of one SSE2 register: lane k = row k, at step t at column t - k, so lane k's `b` is what lane k - 1 held one step
// earlier (a lane shift) and its `c` is its own previous `b`.  Inputs and outputs go through 8 x 8 byte transposes (eight steps at a
// time); the triangles at both ends of the rows are done by the scalar code.  0.27 ms per 752 x 480 image (one row at a time: 1.56)
void unfilter_paeth8(const uint8_t* in, size_t pitch, uint8_t* o, int width)
{
    const uint8_t* ink[8]; uint8_t* ok[8];
    for (int k = 0; k < 8; ++k) { ink[k] = in + (size_t)k * pitch + 1; ok[k] = o + (size_t)k * width; }
    const uint8_t* up0 = o - width;
    // head: steps t < 7 (row k: columns 0 .. 6 - k)
    for (int k = 0; k < 7; ++k) paeth_span(ink[k], ok[k] - width, ok[k], 0, 7 - k);
    const int nblk = (width - 7) / 8, t_end = 7 + 8 * nblk;
    // lane k of the vectors = row k; at step t it is at column t - k
    uint16_t av[8], cv[8];
    for (int k = 0; k < 8; ++k) { av[k] = k < 7 ? ok[k][6 - k] : 0; cv[k] = k == 0 ? up0[6] : (k < 7 ? ok[k - 1][6 - k] : 0); }
    __m128i a = _mm_loadu_si128((const __m128i*)av), c = _mm_loadu_si128((const __m128i*)cv);
    const __m128i zero = _mm_setzero_si128(), ff = _mm_set1_epi16(0xFF);
    for (int blk = 0; blk < nblk; ++blk) {
        const int t0 = 7 + 8 * blk;
        __m128i r[8];
        for (int k = 0; k < 8; ++k) r[k] = _mm_loadl_epi64((const __m128i*)(ink[k] + t0 - k));
        const __m128i t_0 = _mm_unpacklo_epi8(r[0], r[1]), t_1 = _mm_unpacklo_epi8(r[2], r[3]), t_2 = _mm_unpacklo_epi8(r[4], r[5]), t_3 = _mm_unpacklo_epi8(r[6], r[7]);
        const __m128i u0 = _mm_unpacklo_epi16(t_0, t_1), u1 = _mm_unpackhi_epi16(t_0, t_1), u2 = _mm_unpacklo_epi16(t_2, t_3), u3 = _mm_unpackhi_epi16(t_2, t_3);
        __m128i v[4] = {_mm_unpacklo_epi32(u0, u2), _mm_unpackhi_epi32(u0, u2), _mm_unpacklo_epi32(u1, u3), _mm_unpackhi_epi32(u1, u3)};
        __m128i outv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const __m128i x = (j & 1) ? _mm_unpackhi_epi8(v[j >> 1], zero) : _mm_unpacklo_epi8(v[j >> 1], zero);
            const __m128i b = _mm_insert_epi16(_mm_slli_si128(a, 2), up0[t0 + j], 0);
            const __m128i p = _mm_sub_epi16(b, c), q = _mm_sub_epi16(a, c), pq = _mm_add_epi16(p, q);
            const __m128i pa = _mm_max_epi16(p, _mm_sub_epi16(zero, p)), pb = _mm_max_epi16(q, _mm_sub_epi16(zero, q)), pc = _mm_max_epi16(pq, _mm_sub_epi16(zero, pq));
            const __m128i not_a = _mm_or_si128(_mm_cmpgt_epi16(pa, pb), _mm_cmpgt_epi16(pa, pc));
            const __m128i use_c = _mm_cmpgt_epi16(pb, pc);
            const __m128i bc = _mm_or_si128(_mm_and_si128(use_c, c), _mm_andnot_si128(use_c, b));
            const __m128i pred = _mm_or_si128(_mm_and_si128(not_a, bc), _mm_andnot_si128(not_a, a));
            a = _mm_and_si128(_mm_add_epi16(x, pred), ff);
            c = b;
            outv[j] = a;
        }
        // back: 8 steps x 8 rows of 16-bit values -> row k's 8 bytes at columns t0 - k ..
        const __m128i p0 = _mm_packus_epi16(outv[0], outv[1]), p1 = _mm_packus_epi16(outv[2], outv[3]), p2 = _mm_packus_epi16(outv[4], outv[5]), p3 = _mm_packus_epi16(outv[6], outv[7]);
        const __m128i s0 = _mm_unpacklo_epi8(p0, _mm_srli_si128(p0, 8)), s1 = _mm_unpacklo_epi8(p1, _mm_srli_si128(p1, 8)), s2 = _mm_unpacklo_epi8(p2, _mm_srli_si128(p2, 8)), s3 = _mm_unpacklo_epi8(p3, _mm_srli_si128(p3, 8));
        const __m128i w0 = _mm_unpacklo_epi16(s0, s1), w1 = _mm_unpackhi_epi16(s0, s1), w2 = _mm_unpacklo_epi16(s2, s3), w3 = _mm_unpackhi_epi16(s2, s3);
        const __m128i z0 = _mm_unpacklo_epi32(w0, w2), z1 = _mm_unpackhi_epi32(w0, w2), z2 = _mm_unpacklo_epi32(w1, w3), z3 = _mm_unpackhi_epi32(w1, w3);
        _mm_storel_epi64((__m128i*)(ok[0] + t0), z0);     _mm_storel_epi64((__m128i*)(ok[1] + t0 - 1), _mm_srli_si128(z0, 8));
        _mm_storel_epi64((__m128i*)(ok[2] + t0 - 2), z1); _mm_storel_epi64((__m128i*)(ok[3] + t0 - 3), _mm_srli_si128(z1, 8));
        _mm_storel_epi64((__m128i*)(ok[4] + t0 - 4), z2); _mm_storel_epi64((__m128i*)(ok[5] + t0 - 5), _mm_srli_si128(z2, 8));
        _mm_storel_epi64((__m128i*)(ok[6] + t0 - 6), z3); _mm_storel_epi64((__m128i*)(ok[7] + t0 - 7), _mm_srli_si128(z3, 8));
    }
    // tail: row k from column t_end - k
    for (int k = 0; k < 8; ++k) paeth_span(ink[k], ok[k] - width, ok[k], t_end - k, width);
}

// four consecutive Paeth rows (none of them the first row of the image): at step t row k does column t - k, and what row k needs
// of the row above (b at this column; c is last step's b) is the value row k - 1 produced one step ago -- it never leaves its register
#define AV_PAETH_PX(k, bv, x) { const int b_ = (bv); a##k = (uint8_t)(in##k[x] + paeth(a##k, b_, c##k)); o##k[x] = (uint8_t)a##k; c##k = b_; }
void unfilter_paeth4(const uint8_t* in, size_t pitch, uint8_t* o, int width)
{
    const uint8_t* __restrict in0 = in + 1, * __restrict in1 = in0 + pitch, * __restrict in2 = in1 + pitch, * __restrict in3 = in2 + pitch;
    uint8_t* __restrict o0 = o, * __restrict o1 = o0 + width, * __restrict o2 = o1 + width, * __restrict o3 = o2 + width;
    const uint8_t* __restrict up0 = o0 - width;
    int a0 = 0, a1 = 0, a2 = 0, a3 = 0, c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    for (int t = 0; t < width + 3; ++t) {
        const int n0 = a0, n1 = a1, n2 = a2;
        if (t < width) AV_PAETH_PX(0, up0[t], t);
        if (t >= 1 && t - 1 < width) AV_PAETH_PX(1, n0, t - 1);
        if (t >= 2 && t - 2 < width) AV_PAETH_PX(2, n1, t - 2);
        if (t >= 3 && t - 3 < width) AV_PAETH_PX(3, n2, t - 3);
    }
}
#undef AV_PAETH_PX

// 0 ok, 1 unsupported flavour (not 8-bit greyscale, interlaced, wrong size), 2 unreadable / corrupt
int decode_one(const char* path, int width, int height, uint8_t* out, char* why, size_t why_cap)
{
    FILE* f = fopen(path, "rb");
    if (!f) { snprintf(why, why_cap, "cannot open %s", path); return 2; }
    // per-thread buffers that only grow: a fresh 200 KB + 360 KB vector per frame is an mmap, ~140 page faults and a munmap each time
    static thread_local std::vector<uint8_t> file, raw;
    {
        fseek(f, 0, SEEK_END);
        const long sz = ftell(f);
        fseek(f, 0, SEEK_SET);
        if (sz < 57) { fclose(f); snprintf(why, why_cap, "%s: not a PNG file", path); return 2; }
        file.resize((size_t)sz);
        const size_t got = fread(file.data(), 1, (size_t)sz, f);
        fclose(f);
        if (got != (size_t)sz) { snprintf(why, why_cap, "%s: short read", path); return 2; }
    }
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (memcmp(file.data(), sig, 8) != 0) { snprintf(why, why_cap, "%s: not a PNG file", path); return 2; }
    size_t pos = 8;
    bool have_hdr = false, crc_bad = false, order_bad = false;
    const uint8_t* z = nullptr; size_t zn = 0;                  // the zlib stream: the single IDAT in place, or the IDATs joined in `joined`
    std::vector<uint8_t> joined;
    int n_idat = 0;
    while (pos + 12 <= file.size()) {
        const uint32_t len = be32(&file[pos]);
        const uint8_t* type = &file[pos + 4];
        if (pos + 12 + (size_t)len > file.size()) break;
        const uint8_t* data = &file[pos + 8];
        // chunk CRC (PNG spec 5.3: over type + data): a bit flip in a frame must not decode to plausible pixels (cv2.imread returns None)
        if ((!memcmp(type, "IHDR", 4) || !memcmp(type, "IDAT", 4)) && chunk_crc(type, 4 + (size_t)len) != be32(data + len)) { crc_bad = true; break; }
        if (!memcmp(type, "IHDR", 4)) {
            if (len < 13) break;
            const uint32_t w = be32(data), h = be32(data + 4);
            const int depth = data[8], colour = data[9], interlace = data[12];
            if ((int)w != width || (int)h != height || depth != 8 || colour != 0 || interlace != 0) {
                snprintf(why, why_cap, "%s: %ux%u depth %d colour type %d interlace %d (expected %dx%d 8-bit greyscale)", path, w, h, depth, colour, interlace, width, height);
                return 1;
            }
            have_hdr = true;
        } else if (!memcmp(type, "IDAT", 4)) {
            if (!have_hdr) { order_bad = true; break; }
            if (n_idat == 0) { z = data; zn = len; }
            else {
                if (n_idat == 1) joined.assign(z, z + zn);
                joined.insert(joined.end(), data, data + len);
            }
            ++n_idat;
        } else if (!memcmp(type, "IEND", 4)) {
            break;
        }
        pos += 12 + (size_t)len;
    }
    if (n_idat > 1) { z = joined.data(); zn = joined.size(); }
    const size_t pitch = (size_t)width + 1;                      // filter byte + one row of 8-bit grey samples
    if (raw.size() < pitch * (size_t)height) raw.resize(pitch * (size_t)height);
    const size_t raw_bytes = pitch * (size_t)height;
    // complete = the deflate stream ends where it says it does, holds exactly the image and its Adler-32 matches
    const bool complete = have_hdr && !crc_bad && !order_bad && n_idat > 0 && inflate_all(z, zn, raw.data(), raw_bytes);
    if (!complete) { snprintf(why, why_cap, "%s: corrupt or truncated PNG%s", path, crc_bad ? " (chunk CRC mismatch)" : ""); return 2; }
    // un-filter (PNG spec 9.2; bpp = 1): row r of the image lands in out + r * width
    for (int r = 0; r < height; ++r) if (raw[(size_t)r * pitch] > 4) { snprintf(why, why_cap, "%s: bad filter type %d", path, raw[(size_t)r * pitch]); return 2; }
    const uint8_t* prev = nullptr;
    for (int r = 0; r < height; ++r) {
        const uint8_t* in = &raw[(size_t)r * pitch];
        uint8_t* o = out + (size_t)r * width;
        const int ft = in[0];
        if (ft == 4 && r > 0 && r + 7 < height && width >= 16) {
            int n4 = 1;
            while (n4 < 8 && in[(size_t)n4 * pitch] == 4) ++n4;
            if (n4 == 8) {
                unfilter_paeth8(in, pitch, o, width);
                r += 7; prev = o + (size_t)7 * width;
                continue;
            }
        }
        if (ft == 4 && r > 0 && r + 3 < height && in[pitch] == 4 && in[2 * pitch] == 4 && in[3 * pitch] == 4) {
            unfilter_paeth4(in, pitch, o, width);
            r += 3; prev = o + (size_t)3 * width;
            continue;
        }
        ++in;
        switch (ft) {
        case 0: memcpy(o, in, (size_t)width); break;
        case 1: { uint8_t a = 0; for (int x = 0; x < width; ++x) { a = (uint8_t)(in[x] + a); o[x] = a; } break; }
        case 2: if (prev) { for (int x = 0; x < width; ++x) o[x] = (uint8_t)(in[x] + prev[x]); } else memcpy(o, in, (size_t)width); break;
        case 3: { uint8_t a = 0; for (int x = 0; x < width; ++x) { const int b = prev ? prev[x] : 0; a = (uint8_t)(in[x] + ((a + b) >> 1)); o[x] = a; } break; }
        default: {                                            // 4: Paeth
            int a = 0, c = 0;
            for (int x = 0; x < width; ++x) {
                const int b = prev ? prev[x] : 0;
                a = (uint8_t)(in[x] + paeth(a, b, c)); o[x] = (uint8_t)a; c = b;
            }
            break;
        }
        }
        prev = o;
    }
    return 0;
}

}  // namespace

AV_EXPORT int av_png_decode_gray8(const char* const* paths, int n, int width, int height, uint8_t* out, int64_t out_stride, int threads, int32_t* status)
{
    if (!paths || n < 0 || width <= 0 || height <= 0 || !out || out_stride < (int64_t)width * height) { av_set_error("av_png_decode_gray8: bad arguments"); return AV_E_INVALID; }
    if (threads < 1) threads = 1;
    if (threads > 64) threads = 64;
    int worst = 0;
    char msg[400] = "";
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads < n ? threads : (n > 0 ? n : 1))
    for (int i = 0; i < n; ++i) {
        char why[400] = "";
        const int rc = paths[i] ? decode_one(paths[i], width, height, out + (size_t)i * out_stride, why, sizeof(why)) : 2;
        if (status) status[i] = rc;
        if (rc) {
#pragma omp critical(av_png_err)
            { if (rc > worst) { worst = rc; memcpy(msg, why, sizeof(msg)); } }
        }
    }
    if (worst) { av_set_error("av_png_decode_gray8: %s", msg[0] ? msg : "null path"); return worst == 1 ? AV_E_CAPACITY : AV_E_INVALID; }
    return AV_OK;
}
