// png_read.hip -- host-side staging for sweeps: EuRoC camera frames (8-bit greyscale PNG, one file per frame) decoded on
// host threads straight into the caller's image batch (reference: src/streaming/dataset.py:93-158 reads them with
// cv2.imread(path, -1) on a reader thread per sensor; SURVEY 8f.1 "PNG decode on host threads").  Host code only: zlib
// inflate + the five PNG row filters.  No device, no context; thread-safe (each file has its own state).
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <zlib.h>
#include <vector>
#include <omp.h>
#include "av_common.h"

namespace {

inline uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

// 0 ok, 1 unsupported flavour (not 8-bit greyscale, interlaced, wrong size), 2 unreadable / corrupt
int decode_one(const char* path, int width, int height, uint8_t* out, char* why, size_t why_cap)
{
    FILE* f = fopen(path, "rb");
    if (!f) { snprintf(why, why_cap, "cannot open %s", path); return 2; }
    std::vector<uint8_t> file;
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
    bool have_hdr = false;
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    if (inflateInit(&zs) != Z_OK) { snprintf(why, why_cap, "zlib init failed"); return 2; }
    const size_t pitch = (size_t)width + 1;                      // filter byte + one row of 8-bit grey samples
    std::vector<uint8_t> raw(pitch * (size_t)height + 8);       // + slack: the stream's end marker and Adler-32 may sit in a later IDAT chunk, and
                                                                // inflate must be able to go on (and must not be able to write more than the image)
    zs.next_out = raw.data(); zs.avail_out = (uInt)raw.size();
    int zrc = Z_OK;
    bool crc_bad = false;
    while (pos + 12 <= file.size()) {
        const uint32_t len = be32(&file[pos]);
        const uint8_t* type = &file[pos + 4];
        if (pos + 12 + (size_t)len > file.size()) break;
        const uint8_t* data = &file[pos + 8];
        // chunk CRC (PNG spec 5.3: over type + data): a bit flip in a frame must not decode to plausible pixels (cv2.imread returns None)
        if ((!memcmp(type, "IHDR", 4) || !memcmp(type, "IDAT", 4)) && (uint32_t)crc32(crc32(0L, Z_NULL, 0), type, 4 + len) != be32(data + len)) { crc_bad = true; break; }
        if (!memcmp(type, "IHDR", 4)) {
            if (len < 13) break;
            const uint32_t w = be32(data), h = be32(data + 4);
            const int depth = data[8], colour = data[9], interlace = data[12];
            if ((int)w != width || (int)h != height || depth != 8 || colour != 0 || interlace != 0) {
                inflateEnd(&zs);
                snprintf(why, why_cap, "%s: %ux%u depth %d colour type %d interlace %d (expected %dx%d 8-bit greyscale)", path, w, h, depth, colour, interlace, width, height);
                return 1;
            }
            have_hdr = true;
        } else if (!memcmp(type, "IDAT", 4)) {
            if (!have_hdr) break;
            zs.next_in = const_cast<Bytef*>(data); zs.avail_in = len;
            zrc = inflate(&zs, Z_NO_FLUSH);
            if (zrc != Z_OK && zrc != Z_STREAM_END) break;
        } else if (!memcmp(type, "IEND", 4)) {
            break;
        }
        pos += 12 + (size_t)len;
    }
    // Z_STREAM_END: the deflate stream ended where it says it does and its Adler-32 matched (inflate checks it); a stream that merely
    // filled the output buffer (Z_OK, avail_out == 0) may be truncated
    const bool complete = have_hdr && !crc_bad && zs.total_out == pitch * (size_t)height && zrc == Z_STREAM_END;
    inflateEnd(&zs);
    if (!complete) { snprintf(why, why_cap, "%s: corrupt or truncated PNG%s", path, crc_bad ? " (chunk CRC mismatch)" : ""); return 2; }
    // un-filter (PNG spec 9.2; bpp = 1): row r of the image lands in out + r * width
    const uint8_t* prev = nullptr;
    for (int r = 0; r < height; ++r) {
        const uint8_t* in = &raw[(size_t)r * pitch];
        uint8_t* o = out + (size_t)r * width;
        const int ft = in[0];
        ++in;
        switch (ft) {
        case 0: memcpy(o, in, (size_t)width); break;
        case 1: { uint8_t a = 0; for (int x = 0; x < width; ++x) { a = (uint8_t)(in[x] + a); o[x] = a; } break; }
        case 2: if (prev) { for (int x = 0; x < width; ++x) o[x] = (uint8_t)(in[x] + prev[x]); } else memcpy(o, in, (size_t)width); break;
        case 3: { uint8_t a = 0; for (int x = 0; x < width; ++x) { const int b = prev ? prev[x] : 0; a = (uint8_t)(in[x] + ((a + b) >> 1)); o[x] = a; } break; }
        case 4: {
            int a = 0, c = 0;
            for (int x = 0; x < width; ++x) {
                const int b = prev ? prev[x] : 0;
                const int p = a + b - c, pa = p > a ? p - a : a - p, pb = p > b ? p - b : b - p, pc = p > c ? p - c : c - p;
                const int pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
                a = (uint8_t)(in[x] + pred); o[x] = (uint8_t)a; c = b;
            }
            break;
        }
        default: snprintf(why, why_cap, "%s: bad filter type %d", path, ft); return 2;
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
