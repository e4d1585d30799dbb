// pyramid.hip -- padded u8 Gaussian pyramids for LK (gfx950).
//
// Replaces the pyrDown chain that cv2.calcOpticalFlowPyrLK rebuilds inside every call
// (reference: src/image_processing/pyramid_builder.py:22-48 is a pass-through, SURVEY.md F2;
// call sites feature_tracker.py:102, stereo_matcher.py:64,70).  Built ONCE per image here and
// kept resident; the prev-cam0 pyramid is reused by the next frame.
//
// Layout: every level is stored with a 16-pixel BORDER_REFLECT_101 frame (AV_PYR_BORDER), so the
// LK kernel never clamps or reflects a coordinate.  Both kernels are written as pure gathers over
// the PADDED destination: a destination pixel in the frame recomputes the value of the interior
// pixel it mirrors, so there is no scatter and no separate border pass.
//
// Roofline: HBM-bound streaming (level 0: w*h read + padded write; levels 1..3: 1/4, 1/16, 1/64
// of that).  Integer arithmetic only: separable [1 4 6 4 1], (sum + 128) >> 8.
#include <limits.h>
#include <stdlib.h>

#include "av_common.h"

namespace {

struct PyrArgs {
    const uint8_t* img0;
    const uint8_t* img1;
    int64_t img_stride;
    int imgs_per_stream;
    uint8_t* pyr_base;
    int64_t stream_stride, slot_stride;
    int slot0, slot1;
    PyrGeom g;
    int n_img, tiles_x, tiles_y;       // pyr_l0l1_kernel: XCD-aware 1-D launch when tiles_x > 0
    const int* index;                  // optional: image group i (a "stream") is storage entry index[i] of img0 / img1 / pyr_base (shared frame store)
};

// image number of the launch -> (storage entry, camera)
__device__ __forceinline__ void pyr_entry(const PyrArgs& a, int img, int& s, int& cam)
{
    s = img / a.imgs_per_stream; cam = img - s * a.imgs_per_stream;
    if (a.index) s = a.index[s];
}

__device__ __forceinline__ uint8_t* pyr_of(const PyrArgs& a, int img)
{
    int s, cam;
    pyr_entry(a, img, s, cam);
    return a.pyr_base + s * a.stream_stride + (cam == 0 ? a.slot0 : a.slot1) * a.slot_stride;
}

// level 0: copy the tightly packed input image into the padded level, frame included.  One thread writes 16 bytes;
// chunks that lie inside the image (all but the 16-pixel frame columns) are one 16-byte load when rows are 16-aligned.
__global__ __launch_bounds__(256) void pad_level0_kernel(PyrArgs a)
{
    const int w = a.g.w[0], h = a.g.h[0], pitch = a.g.pitch[0];
    const int chunks_per_row = pitch >> 4;                   // pitch is a multiple of 16
    const int ph = h + 2 * AV_PYR_BORDER;
    int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= chunks_per_row * ph) return;
    int yp = q / chunks_per_row, xc = q - yp * chunks_per_row;
    int img = blockIdx.y;
    int s, cam;
    pyr_entry(a, img, s, cam);
    const uint8_t* src = (cam == 0 ? a.img0 : a.img1) + s * a.img_stride;
    uint8_t* dst = pyr_of(a, img) + a.g.off[0];
    int y = av_reflect101(yp - AV_PYR_BORDER, h);
    const uint8_t* row = src + (size_t)y * w;
    const int x0 = xc * 16 - AV_PYR_BORDER;
    uint4 out;
    if (x0 >= 0 && x0 + 16 <= w && ((w | (int)(a.img_stride & 15) | (int)(reinterpret_cast<uintptr_t>(src) & 15)) & 15) == 0) {
        out = *reinterpret_cast<const uint4*>(row + x0);
    } else {
        uint32_t o[4] = {0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            int x = av_reflect101(x0 + i, w);
            x = min(max(x, 0), w - 1);      // slack columns beyond w+32 (pitch rounding): any valid pixel
            o[i >> 2] |= (uint32_t)row[x] << (8 * (i & 3));
        }
        out = make_uint4(o[0], o[1], o[2], o[3]);
    }
    *reinterpret_cast<uint4*>(dst + (size_t)yp * pitch + xc * 16) = out;
}

// Levels 0 AND 1 in one pass over the input image: a 128 x 32 tile of the image (+ a 2-pixel halo, reflect-101 at the image
// border) is staged in LDS once; from it the workgroup writes its part of the padded level 0 (16-byte stores, the frame
// parts of border tiles gathered through the reflection), filters the 64 x 16 tile of level 1 (row filter by v_dot4 into
// u16 sums, column filter, (sum + 128) >> 8) and writes it, plus the mirror images of those level-1 pixels that lie within
// 16 pixels of the image border (the frame of level 1).  The separate pad and first pyrDown kernels read the image and
// re-read the padded level 0 (0.8 MB per image through L2); here every input byte is read once.
constexpr int FT_W = 128, FT_LP = FT_W + 8;        // LDS row: columns x0-4 .. x0+131
// tile height: 96 rows when the image height allows (1,000 sixteen-byte items per workgroup: four loads in flight per thread,
// 2.1 % halo rows), else 32
constexpr int FT_H_TALL = 96, FT_H_BASE = 32;
template <bool WRITE_L0, int FT_H>
__global__ __launch_bounds__(256) void pyr_l0l1_kernel(PyrArgs a)
{
    __shared__ __attribute__((aligned(16))) uint8_t src[(FT_H + 4) * FT_LP];
    __shared__ __attribute__((aligned(16))) uint16_t hs[(FT_H + 4) * (FT_W / 2)];
    // the level-1 tile lives where the staged source rows were: their last reader is the row filter, a barrier ahead of the first write
    // (26.4 instead of 29.4 KB of LDS: six workgroups per CU instead of five -- the kernel is bound by the bytes its CU has in flight)
    static_assert((FT_H / 2) * (FT_W / 2) <= (FT_H + 4) * FT_LP, "level-1 tile fits the source tile");
    uint8_t* const l1t = src;
    const int w = a.g.w[0], h = a.g.h[0], pitch0 = a.g.pitch[0];
    const int w1 = a.g.w[1], h1 = a.g.h[1], pitch1 = a.g.pitch[1];
    const int tid = threadIdx.x;
    // workgroup -> (image, tile): with the 1-D launch all tiles of an image run on ONE XCD (round-robin dispatch over 8 XCDs, an
    // L2 each): horizontally adjacent tiles share every 128-byte line a 752-byte image row lays across their boundary, and
    // vertically adjacent ones their 4 halo rows -- on eight L2s each of those lines is fetched once per XCD that meets it
    // (PMC: 2.3x the image bytes), on one L2 once.
    int img, bx, by;
    if (a.tiles_x > 0) {
        const int L = blockIdx.x, j = L >> 3, per = a.tiles_x * a.tiles_y;
        img = (L & 7) + 8 * (j / per);
        if (img >= a.n_img) return;
        const int t = j % per;
        by = t / a.tiles_x; bx = t - by * a.tiles_x;
    } else { img = blockIdx.z; bx = blockIdx.x; by = blockIdx.y; }
    const int x0 = bx * FT_W, y0 = by * FT_H;
    int s, cam;
    pyr_entry(a, img, s, cam);
    const uint8_t* in = (cam == 0 ? a.img0 : a.img1) + s * a.img_stride;
    uint8_t* base = pyr_of(a, img);
    uint8_t* d0 = base + a.g.off[0];
    uint8_t* d1 = base + a.g.off[1];
    uint32_t* srcw = reinterpret_cast<uint32_t*>(src);

    // ---- stage rows y0-2 .. y0+FT_H+1, columns x0-4 .. x0+131 (reflect-101 outside the image): per row eight 16-byte chunks
    //      of the tile's own columns plus the two halo dwords ----
    const bool al16 = ((a.img_stride & 15) == 0) && ((reinterpret_cast<uintptr_t>(in) & 15) == 0);
    // NIT items per thread ((FT_H + 4) x 10 items, 256 threads).  The loads of ALL items are issued before any is waited for: as a plain
    // loop the compiler waited (vmcnt(0)) inside every iteration, i.e. one 16-byte load in flight per thread -- 4 KB per workgroup,
    // far below what HBM latency x bandwidth needs per CU once the level-0 store stream is gone.
    {
        constexpr int NIT = ((FT_H + 4) * 10 + 255) / 256;
        uint4 q4[NIT]; uint32_t hv[NIT]; int kind[NIT], rr[NIT], jj[NIT];          // kind: 0 none, 1 sixteen bytes, 2 halo dword, 3 gather (image border)
#pragma unroll
        for (int u = 0; u < NIT; ++u) {
            const int i = tid + 256 * u;
            const bool valid = i < (FT_H + 4) * 10;
            const int r = valid ? i / 10 : 0, j = valid ? i - r * 10 : 0;
            rr[u] = r; jj[u] = j;
            kind[u] = valid ? 3 : 0; hv[u] = 0; q4[u] = make_uint4(0, 0, 0, 0);
        }
        if (al16) {                                      // wave-uniform; the four loads below sit in ONE basic block: no wait between them
#pragma unroll
            for (int u = 0; u < NIT; ++u) {
                const int r = rr[u], j = jj[u];
                const uint8_t* row = in + (size_t)av_reflect101(y0 - 2 + r, h) * w;
                const int gx = x0 + 16 * j, gh = j == 8 ? x0 - 4 : x0 + FT_W;
                const bool ok16 = kind[u] != 0 && j < 8 && gx + 16 <= w;
                const bool okh = kind[u] != 0 && j >= 8 && gh >= 0 && gh + 4 <= w;
                q4[u] = *reinterpret_cast<const uint4*>(ok16 ? row + gx : in);          // (a lane without a 16-byte item re-reads the image's first line)
                hv[u] = *reinterpret_cast<const uint32_t*>(okh ? row + gh : in);
                kind[u] = ok16 ? 1 : (okh ? 2 : kind[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < NIT; ++u) {
            const int r = rr[u], j = jj[u];
            if (kind[u] == 1) {
                uint32_t* dst = srcw + r * (FT_LP / 4) + 1 + 4 * j;
                dst[0] = q4[u].x; dst[1] = q4[u].y; dst[2] = q4[u].z; dst[3] = q4[u].w;
            } else if (kind[u] == 2) {
                srcw[r * (FT_LP / 4) + (j == 8 ? 0 : FT_LP / 4 - 1)] = hv[u];
            } else if (kind[u] == 3) {
                const uint8_t* row = in + (size_t)av_reflect101(y0 - 2 + r, h) * w;
                if (j < 8) {
                    const int gx = x0 + 16 * j;
                    uint32_t v[4];
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        if (gx + 4 * d + 4 <= w) v[d] = *reinterpret_cast<const uint32_t*>(row + gx + 4 * d);
                        else {
                            v[d] = 0;
#pragma unroll
                            for (int b = 0; b < 4; ++b) { int x = av_reflect101(gx + 4 * d + b, w); x = min(max(x, 0), w - 1); v[d] |= (uint32_t)row[x] << (8 * b); }
                        }
                    }
                    uint32_t* dst = srcw + r * (FT_LP / 4) + 1 + 4 * j;
                    dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2]; dst[3] = v[3];
                } else {
                    const int gx = j == 8 ? x0 - 4 : x0 + FT_W;
                    uint32_t v = 0;
#pragma unroll
                    for (int b = 0; b < 4; ++b) { int x = av_reflect101(gx + b, w); x = min(max(x, 0), w - 1); v |= (uint32_t)row[x] << (8 * b); }
                    srcw[r * (FT_LP / 4) + (j == 8 ? 0 : FT_LP / 4 - 1)] = v;
                }
            }
        }
    }
    __syncthreads();

    // ---- padded level 0: destination rows / 16-byte chunks owned by this tile (interior + its share of the frame) ----
    //      (WRITE_L0 = false: level 0 stays the caller's image -- LK and FAST read it in place, lk.hip LKArgs::imgI -- which
    //       removes 401 KB of stores and the re-read of that copy per image: 44 % of this kernel's bytes)
    if (WRITE_L0) {
        const bool top = y0 == 0, bot = y0 + FT_H >= h, left = x0 == 0, right = x0 + FT_W >= w;
        const int tw = min(FT_W, w - x0);                                 // valid interior columns (multiple of 16 or the image's tail)
        const int nrow = FT_H + (top ? AV_PYR_BORDER : 0) + (bot ? AV_PYR_BORDER : 0);
        const int nch_in = (tw + 15) >> 4;
        const int nch = nch_in + (left ? 1 : 0) + (right ? (pitch0 - AV_PYR_BORDER - w + 15) / 16 : 0);
        for (int i = tid; i < nrow * nch; i += 256) {
            const int ri = i / nch, ci = i - ri * nch;
            // destination row (image coordinates): the tile's own rows first, then the frame rows above / below
            int yp;
            if (ri < FT_H) yp = y0 + ri;
            else if (top && ri < FT_H + AV_PYR_BORDER) yp = -(ri - FT_H + 1);
            else yp = h + (ri - FT_H - (top ? AV_PYR_BORDER : 0));
            if (yp >= h + AV_PYR_BORDER || (ri < FT_H && yp >= h)) continue;
            int xd;                                                        // first destination column of the chunk
            bool interior;
            if (ci < nch_in) { xd = x0 + 16 * ci; interior = true; }
            else if (left && ci == nch_in) { xd = -AV_PYR_BORDER; interior = false; }
            else { xd = w + 16 * (ci - nch_in - (left ? 1 : 0)); interior = false; }
            const int sy = av_reflect101(yp, h) - (y0 - 2);                // staged row of the source pixel row
            uint32_t o[4];
            if (interior && xd + 16 <= w) {
                const uint32_t* p = srcw + sy * (FT_LP / 4) + ((xd - x0 + 4) >> 2);
                o[0] = p[0]; o[1] = p[1]; o[2] = p[2]; o[3] = p[3];
            } else {
                o[0] = o[1] = o[2] = o[3] = 0;
#pragma unroll
                for (int b = 0; b < 16; ++b) {
                    int x = av_reflect101(xd + b, w); x = min(max(x, 0), w - 1);
                    o[b >> 2] |= (uint32_t)src[sy * FT_LP + (x - x0 + 4)] << (8 * (b & 3));
                }
            }
            *reinterpret_cast<uint4*>(d0 + (size_t)(yp + AV_PYR_BORDER) * pitch0 + (xd + AV_PYR_BORDER)) = make_uint4(o[0], o[1], o[2], o[3]);
        }
    }

    // ---- level 1: row filter [1 4 6 4 1] of the staged rows into u16 sums: 4 outputs per item from 16 staged bytes ----
    for (int i = tid; i < (FT_H + 4) * (FT_W / 8); i += 256) {
        const int r = i / (FT_W / 8), xq = i - r * (FT_W / 8);
        const uint2 lo = *reinterpret_cast<const uint2*>(src + r * FT_LP + 8 * xq);
        const uint2 hi = *reinterpret_cast<const uint2*>(src + r * FT_LP + 8 * xq + 8);
        const uint32_t dx_ = lo.x, dy_ = lo.y, dz_ = hi.x, dw_ = hi.y;
        const uint32_t t0 = __builtin_amdgcn_alignbyte(dy_, dx_, 2), t1 = dy_, t2 = __builtin_amdgcn_alignbyte(dz_, dy_, 2), t3 = dz_;
        const uint32_t W4 = 0x04060401u;
        const uint32_t h0 = __builtin_amdgcn_udot4(t0, W4, (dy_ >> 16) & 0xFF, false);
        const uint32_t h1_ = __builtin_amdgcn_udot4(t1, W4, dz_ & 0xFF, false);
        const uint32_t h2 = __builtin_amdgcn_udot4(t2, W4, (dz_ >> 16) & 0xFF, false);
        const uint32_t h3 = __builtin_amdgcn_udot4(t3, W4, dw_ & 0xFF, false);
        *reinterpret_cast<uint2*>(hs + r * (FT_W / 2) + 4 * xq) = make_uint2(h0 | (h1_ << 16), h2 | (h3 << 16));
    }
    __syncthreads();
    // column filter: thread t -> level-1 row yo = t / 16, columns 4 xq .. 4 xq + 3 of the tile
    for (int t_ = tid; t_ < (FT_H / 2) * (FT_W / 8); t_ += 256) {
        const int yo = t_ >> 4, xq = t_ & 15;
        uint32_t out = 0;
        const uint16_t* c = hs + (2 * yo) * (FT_W / 2) + 4 * xq;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int v = (int)c[k] + (int)c[4 * (FT_W / 2) + k] + 4 * ((int)c[(FT_W / 2) + k] + (int)c[3 * (FT_W / 2) + k]) + 6 * (int)c[2 * (FT_W / 2) + k];
            out |= (uint32_t)((v + 128) >> 8) << (8 * k);
        }
        *reinterpret_cast<uint32_t*>(l1t + yo * (FT_W / 2) + 4 * xq) = out;
        const int x1 = x0 / 2 + 4 * xq, y1 = y0 / 2 + yo;
        if (x1 < w1 && y1 < h1) *reinterpret_cast<uint32_t*>(d1 + (size_t)(y1 + AV_PYR_BORDER) * pitch1 + (x1 + AV_PYR_BORDER)) = out;
    }
    // frame of level 1: the level-1 pixels of this tile that lie within 16 pixels of the border are mirrored outwards
    const int X1 = x0 / 2, Y1 = y0 / 2;
    const bool band = X1 <= AV_PYR_BORDER || X1 + FT_W / 2 >= w1 - AV_PYR_BORDER - 1 || Y1 <= AV_PYR_BORDER || Y1 + FT_H / 2 >= h1 - AV_PYR_BORDER - 1;
    if (band) {
        __syncthreads();
        for (int i = tid; i < (FT_H / 2) * (FT_W / 2); i += 256) {
            const int yo = i / (FT_W / 2), xo = i - yo * (FT_W / 2);
            const int x1 = X1 + xo, y1 = Y1 + yo;
            if (x1 >= w1 || y1 >= h1) continue;
            const uint8_t v = l1t[i];
            const int xm = (x1 >= 1 && x1 <= AV_PYR_BORDER) ? -x1 : ((x1 >= w1 - 1 - AV_PYR_BORDER && x1 <= w1 - 2) ? 2 * (w1 - 1) - x1 : INT_MIN);
            const int ym = (y1 >= 1 && y1 <= AV_PYR_BORDER) ? -y1 : ((y1 >= h1 - 1 - AV_PYR_BORDER && y1 <= h1 - 2) ? 2 * (h1 - 1) - y1 : INT_MIN);
            if (xm != INT_MIN) d1[(size_t)(y1 + AV_PYR_BORDER) * pitch1 + (xm + AV_PYR_BORDER)] = v;
            if (ym != INT_MIN) d1[(size_t)(ym + AV_PYR_BORDER) * pitch1 + (x1 + AV_PYR_BORDER)] = v;
            if (xm != INT_MIN && ym != INT_MIN) d1[(size_t)(ym + AV_PYR_BORDER) * pitch1 + (xm + AV_PYR_BORDER)] = v;
        }
    }
}

// level l (>= 1) from padded level l-1.
__global__ __launch_bounds__(256) void pyr_down_kernel(PyrArgs a, int level)
{
    const int w = a.g.w[level], h = a.g.h[level], pitch = a.g.pitch[level];
    const int spitch = a.g.pitch[level - 1];
    const int quads_per_row = pitch >> 2;
    const int ph = h + 2 * AV_PYR_BORDER;
    int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= quads_per_row * ph) return;
    int yp = q / quads_per_row, xq = q - yp * quads_per_row;
    uint8_t* base = pyr_of(a, blockIdx.y);
    const uint8_t* src = base + a.g.off[level - 1];
    uint8_t* dst = base + a.g.off[level];
    int y = av_reflect101(yp - AV_PYR_BORDER, h);
    // padded source rows 2y-2 .. 2y+2
    const uint8_t* r0 = src + (size_t)(2 * y - 2 + AV_PYR_BORDER) * spitch + AV_PYR_BORDER;
    uint32_t out = 0;
    const int x0 = xq * 4 - AV_PYR_BORDER;
    if (x0 >= 0 && x0 + 4 <= w) {
        // interior quad: source bytes 2*x0-2 .. 2*x0+8 of each row sit inside the 16 bytes at 2*x0-4 (dword aligned:
        // x0 is a multiple of 4, the padded pitch a multiple of 16).  Row filter [1 4 6 4] as one v_dot4 plus the fifth tap.
        int hs[4][5];
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const uint4 d = *reinterpret_cast<const uint4*>(r0 + (size_t)k * spitch + 2 * x0 - 4);
            const uint32_t t0 = __builtin_amdgcn_alignbyte(d.y, d.x, 2), t1 = d.y, t2 = __builtin_amdgcn_alignbyte(d.z, d.y, 2), t3 = d.z;
            const uint32_t W4 = 0x04060401u;
            hs[0][k] = (int)__builtin_amdgcn_udot4(t0, W4, (d.y >> 16) & 0xFF, false);
            hs[1][k] = (int)__builtin_amdgcn_udot4(t1, W4, d.z & 0xFF, false);
            hs[2][k] = (int)__builtin_amdgcn_udot4(t2, W4, (d.z >> 16) & 0xFF, false);
            hs[3][k] = (int)__builtin_amdgcn_udot4(t3, W4, d.w & 0xFF, false);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int v = hs[i][0] + hs[i][4] + 4 * (hs[i][1] + hs[i][3]) + 6 * hs[i][2];
            out |= (uint32_t)((v + 128) >> 8) << (8 * i);
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int xp = xq * 4 + i;
            int x = av_reflect101(xp - AV_PYR_BORDER, w);
            x = min(max(x, 0), w - 1);
            const uint8_t* p = r0 + 2 * x;
            int hsum[5];
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                const uint8_t* r = p + (size_t)k * spitch;
                hsum[k] = (int)r[-2] + (int)r[2] + 4 * ((int)r[-1] + (int)r[1]) + 6 * (int)r[0];
            }
            int v = hsum[0] + hsum[4] + 4 * (hsum[1] + hsum[3]) + 6 * hsum[2];
            out |= (uint32_t)((v + 128) >> 8) << (8 * i);
        }
    }
    *reinterpret_cast<uint32_t*>(dst + (size_t)yp * pitch + xq * 4) = out;
}

// Levels 2 AND 3 of one image by one workgroup: level 2 is filtered from the padded level 1 (global) into LDS, level 3 from
// that LDS copy, and both 16-pixel frames are mirrored out of LDS.  The per-level gather kernel spends most of its time in the
// frame at these sizes (52 % of the padded level 3 is frame, every frame pixel re-filters 25 source bytes through the
// reflection); here a frame pixel is one LDS byte read.
template <int L23_T>
__global__ __launch_bounds__(L23_T) void pyr_l2l3_kernel(PyrArgs a)
{
    extern __shared__ uint8_t lds23[];
    const int w1 = a.g.w[1], p1 = a.g.pitch[1];
    const int w2 = a.g.w[2], h2 = a.g.h[2], p2 = a.g.pitch[2];
    const int w3 = a.g.w[3], h3 = a.g.h[3], p3 = a.g.pitch[3];
    const int lp2 = (w2 + 3) & ~3, lp3 = (w3 + 3) & ~3;             // LDS pitches (bytes)
    uint8_t* l2 = lds23;                                            // [h2][lp2]
    uint8_t* l3 = lds23 + (((size_t)h2 * lp2 + 15) & ~(size_t)15);  // [h3][lp3]
    const int tid = threadIdx.x;
    uint8_t* base = pyr_of(a, blockIdx.x);
    const uint8_t* s1 = base + a.g.off[1] + (size_t)AV_PYR_BORDER * p1 + AV_PYR_BORDER;      // level-1 pixel (0,0); its frame is valid around it
    uint8_t* d2 = base + a.g.off[2];
    uint8_t* d3 = base + a.g.off[3];
    (void)w1;
    // ---- level 2 interior: quads of 4 pixels from five 16-byte source segments (as pyr_down_kernel) ----
    const int q2 = lp2 >> 2;
    for (int i = tid; i < q2 * h2; i += L23_T) {
        const int y = i / q2, x0 = (i - y * q2) * 4;
        const uint8_t* r0 = s1 + (ptrdiff_t)(2 * y - 2) * p1;
        int hs[4][5];
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const uint4 d = *reinterpret_cast<const uint4*>(r0 + (size_t)k * p1 + 2 * x0 - 4);
            const uint32_t t0 = __builtin_amdgcn_alignbyte(d.y, d.x, 2), t1 = d.y, t2 = __builtin_amdgcn_alignbyte(d.z, d.y, 2), t3 = d.z;
            const uint32_t W4 = 0x04060401u;
            hs[0][k] = (int)__builtin_amdgcn_udot4(t0, W4, (d.y >> 16) & 0xFF, false);
            hs[1][k] = (int)__builtin_amdgcn_udot4(t1, W4, d.z & 0xFF, false);
            hs[2][k] = (int)__builtin_amdgcn_udot4(t2, W4, (d.z >> 16) & 0xFF, false);
            hs[3][k] = (int)__builtin_amdgcn_udot4(t3, W4, d.w & 0xFF, false);
        }
        uint32_t out = 0;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int v = hs[c][0] + hs[c][4] + 4 * (hs[c][1] + hs[c][3]) + 6 * hs[c][2];
            out |= (uint32_t)((v + 128) >> 8) << (8 * c);
        }
        *reinterpret_cast<uint32_t*>(l2 + y * lp2 + x0) = out;       // columns >= w2 of the last quad are never read back
    }
    __syncthreads();
    // ---- padded level 2 out of LDS (interior and frame alike: a destination quad gathers 4 bytes through the reflection) ----
    const int ph2 = h2 + 2 * AV_PYR_BORDER, pq2 = p2 >> 2;
    for (int i = tid; i < pq2 * ph2; i += L23_T) {
        const int yp = i / pq2, xq = i - yp * pq2;
        const int y = av_reflect101(yp - AV_PYR_BORDER, h2);
        uint32_t out = 0;
        const int xs = xq * 4 - AV_PYR_BORDER;
        if (xs >= 0 && xs + 4 <= w2) out = *reinterpret_cast<const uint32_t*>(l2 + y * lp2 + xs);
        else {
#pragma unroll
            for (int c = 0; c < 4; ++c) { int x = av_reflect101(xs + c, w2); x = min(max(x, 0), w2 - 1); out |= (uint32_t)l2[y * lp2 + x] << (8 * c); }
        }
        *reinterpret_cast<uint32_t*>(d2 + (size_t)yp * p2 + xq * 4) = out;
    }
    // ---- level 3 interior from the LDS copy of level 2 (reflect-101 at its border) ----
    const int q3 = lp3 >> 2;
    for (int i = tid; i < q3 * h3; i += L23_T) {
        const int y = i / q3, x0 = (i - y * q3) * 4;
        uint32_t out = 0;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int x = x0 + c;
            int v = 0;
            if (x < w3) {
                int hsum[5];
#pragma unroll
                for (int k = 0; k < 5; ++k) {
                    const uint8_t* r = l2 + av_reflect101(2 * y - 2 + k, h2) * lp2;
                    const int xm2 = av_reflect101(2 * x - 2, w2), xm1 = av_reflect101(2 * x - 1, w2), xp1 = av_reflect101(2 * x + 1, w2), xp2 = av_reflect101(2 * x + 2, w2);
                    hsum[k] = (int)r[xm2] + (int)r[xp2] + 4 * ((int)r[xm1] + (int)r[xp1]) + 6 * (int)r[2 * x];
                }
                v = (hsum[0] + hsum[4] + 4 * (hsum[1] + hsum[3]) + 6 * hsum[2] + 128) >> 8;
            }
            out |= (uint32_t)v << (8 * c);
        }
        *reinterpret_cast<uint32_t*>(l3 + y * lp3 + x0) = out;
    }
    __syncthreads();
    const int ph3 = h3 + 2 * AV_PYR_BORDER, pq3 = p3 >> 2;
    for (int i = tid; i < pq3 * ph3; i += L23_T) {
        const int yp = i / pq3, xq = i - yp * pq3;
        const int y = av_reflect101(yp - AV_PYR_BORDER, h3);
        uint32_t out = 0;
        const int xs = xq * 4 - AV_PYR_BORDER;
        if (xs >= 0 && xs + 4 <= w3) out = *reinterpret_cast<const uint32_t*>(l3 + y * lp3 + xs);
        else {
#pragma unroll
            for (int c = 0; c < 4; ++c) { int x = av_reflect101(xs + c, w3); x = min(max(x, 0), w3 - 1); out |= (uint32_t)l3[y * lp3 + x] << (8 * c); }
        }
        *reinterpret_cast<uint32_t*>(d3 + (size_t)yp * p3 + xq * 4) = out;
    }
}

}  // namespace

PyrGeom av_make_geom(const av_pyr_layout& l)
{
    PyrGeom g;
    memset(&g, 0, sizeof(g));
    g.levels = l.levels;
    for (int i = 0; i < l.levels; ++i) {
        g.w[i] = l.w[i]; g.h[i] = l.h[i]; g.pitch[i] = l.pitch[i]; g.off[i] = (int)l.offset[i];
    }
    return g;
}

int av_launch_pyramid(const uint8_t* img0, const uint8_t* img1, int64_t img_stride, int n_streams, int imgs_per_stream,
                      const PyrGeom& g, uint8_t* pyr_base, int64_t stream_stride, int64_t slot_stride, int slot0, int slot1,
                      hipStream_t st, bool write_level0, bool* wrote_level0, const int* index)
{
    if (wrote_level0) *wrote_level0 = true;
    if (n_streams <= 0) return AV_OK;
    PyrArgs a;
    a.img0 = img0; a.img1 = img1; a.img_stride = img_stride; a.imgs_per_stream = imgs_per_stream;
    a.pyr_base = pyr_base; a.stream_stride = stream_stride; a.slot_stride = slot_stride;
    a.slot0 = slot0; a.slot1 = slot1; a.g = g;
    a.n_img = 0; a.tiles_x = 0; a.tiles_y = 0; a.index = index;
    const int n_img = n_streams * imgs_per_stream;
    const int w = g.w[0], h = g.h[0];
    // the fused level-0 + level-1 kernel needs: dword-aligned rows, whole 32-row tiles, a last tile column that still holds the
    // 17 pixels its frame mirrors, images large enough that a pixel is never in two mirror bands
    static const bool tall_ok = [] { const char* e = getenv("AV_PYR_TALL"); return !(e && atoi(e) == 0); }();      // A/B switch
    const int FT_H = (tall_ok && (h % FT_H_TALL) == 0 && h >= 2 * FT_H_TALL) ? FT_H_TALL : FT_H_BASE;
    const bool fused = g.levels >= 2 && (w & 15) == 0 && (h % FT_H) == 0 && h >= 64 && w >= 64 && (w % FT_W == 0 || w % FT_W >= 20) &&
                       (img_stride & 3) == 0 && (reinterpret_cast<uintptr_t>(img0) & 3) == 0 && (!img1 || (reinterpret_cast<uintptr_t>(img1) & 3) == 0) &&
                       !getenv("AV_PYR_UNFUSED");
    if (fused) {
        static const bool xcd_map = [] { const char* e = getenv("AV_PYR_XCD"); return !(e && atoi(e) == 0); }();      // A/B switch
        const int tx = (w + FT_W - 1) / FT_W, ty = h / FT_H;
        a.n_img = n_img; a.tiles_x = xcd_map ? tx : 0; a.tiles_y = ty;
        dim3 grid = xcd_map ? dim3((unsigned)(tx * ty) * 8u * (unsigned)((n_img + 7) / 8)) : dim3(tx, ty, n_img);
        if (FT_H == FT_H_TALL) {
            if (write_level0) hipLaunchKernelGGL((pyr_l0l1_kernel<true, FT_H_TALL>), grid, dim3(256), 0, st, a);
            else { hipLaunchKernelGGL((pyr_l0l1_kernel<false, FT_H_TALL>), grid, dim3(256), 0, st, a); if (wrote_level0) *wrote_level0 = false; }
        } else {
            if (write_level0) hipLaunchKernelGGL((pyr_l0l1_kernel<true, FT_H_BASE>), grid, dim3(256), 0, st, a);
            else { hipLaunchKernelGGL((pyr_l0l1_kernel<false, FT_H_BASE>), grid, dim3(256), 0, st, a); if (wrote_level0) *wrote_level0 = false; }
        }
        AV_LAUNCH_CHECK();
    } else {
        int chunks = (g.pitch[0] >> 4) * (g.h[0] + 2 * AV_PYR_BORDER);
        dim3 grid((chunks + 255) / 256, n_img);
        hipLaunchKernelGGL(pad_level0_kernel, grid, dim3(256), 0, st, a);
        AV_LAUNCH_CHECK();
    }
    // levels 2 + 3 by one workgroup per image when the level-2 image fits in LDS
    const size_t lds23 = g.levels == 4 ? ((((size_t)g.h[2] * ((g.w[2] + 3) & ~3) + 15) & ~(size_t)15) + (size_t)g.h[3] * ((g.w[3] + 3) & ~3) + 16) : 0;
    const bool fused23 = g.levels == 4 && lds23 <= 60 * 1024 && (g.pitch[1] & 3) == 0 && !getenv("AV_PYR_UNFUSED");
    int lbeg = fused ? 2 : 1;
    if (fused23) {
        if (!fused) {
            int quads = (g.pitch[1] >> 2) * (g.h[1] + 2 * AV_PYR_BORDER);
            dim3 grid((quads + 255) / 256, n_img);
            hipLaunchKernelGGL(pyr_down_kernel, grid, dim3(256), 0, st, a, 1);
            AV_LAUNCH_CHECK();
        }
        // (512- and 256-thread workgroups -- a 1,024-thread workgroup is sixteen waves, two of them fill a CU's wave slots -- measured
        //  0.961 / 0.969 ms of pyramid time per step against 0.973: within the noise, round 5)
        hipLaunchKernelGGL(pyr_l2l3_kernel<1024>, dim3(n_img), dim3(1024), lds23, st, a);
        AV_LAUNCH_CHECK();
        lbeg = g.levels;
    }
    for (int l = lbeg; l < g.levels; ++l) {
        int quads = (g.pitch[l] >> 2) * (g.h[l] + 2 * AV_PYR_BORDER);
        dim3 grid((quads + 255) / 256, n_img);
        hipLaunchKernelGGL(pyr_down_kernel, grid, dim3(256), 0, st, a, l);
        AV_LAUNCH_CHECK();
    }
    return AV_OK;
}

AV_EXPORT int av_pyramid_layout(int w, int h, int levels, av_pyr_layout* out)
{
    if (!out || levels < 1 || levels > AV_MAX_LEVELS || w <= 0 || h <= 0) {
        av_set_error("av_pyramid_layout: bad arguments (w=%d h=%d levels=%d)", w, h, levels);
        return AV_E_INVALID;
    }
    memset(out, 0, sizeof(*out));
    out->levels = levels;
    int64_t off = 0;
    int lw = w, lh = h;
    for (int l = 0; l < levels; ++l) {
        if (lw <= AV_PYR_BORDER || lh <= AV_PYR_BORDER) {
            av_set_error("av_pyramid_layout: level %d is %dx%d, too small for a %d-pixel reflect-101 frame", l, lw, lh, AV_PYR_BORDER);
            return AV_E_INVALID;
        }
        out->w[l] = lw; out->h[l] = lh;
        out->pitch[l] = ((lw + 2 * AV_PYR_BORDER) + 15) & ~15;
        out->offset[l] = off;
        off += (int64_t)out->pitch[l] * (lh + 2 * AV_PYR_BORDER);
        lw = (lw + 1) / 2; lh = (lh + 1) / 2;
    }
    out->bytes = (off + 255) & ~(int64_t)255;
    if (out->bytes > 0x7fffffff) { av_set_error("av_pyramid_layout: image too large"); return AV_E_INVALID; }
    return AV_OK;
}

AV_EXPORT int av_pyramid_build(const uint8_t* img_dev, int64_t img_stride, int n_img, int w, int h, int levels,
                               uint8_t* pyr_dev, int64_t pyr_stride, void* stream)
{
    av_pyr_layout lay;
    int rc = av_pyramid_layout(w, h, levels, &lay);
    if (rc) return rc;
    if (!img_dev || !pyr_dev || n_img < 0 || pyr_stride < lay.bytes || img_stride < (int64_t)w * h) {
        av_set_error("av_pyramid_build: bad arguments");
        return AV_E_INVALID;
    }
    return av_launch_pyramid(img_dev, nullptr, img_stride, n_img, 1, av_make_geom(lay), pyr_dev, pyr_stride, 0, 0, 0,
                             (hipStream_t)stream);
}
