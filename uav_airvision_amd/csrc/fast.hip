// fast.hip -- FAST-9/16 corner detector with corner score and 3x3 non-max suppression (gfx950).
//
// Replaces cv2.FastFeatureDetector_create(threshold).detect(img, mask) (TYPE_9_16, NMS on):
//   src/image_processing/pipeline.py:23-25 (ctor), feature_initializer.py:52, feature_adder.py:64.
// Semantics follow OpenCV 4.x features2d/fast.cpp (FAST_t<16>) and fast_score.cpp
// (cornerScore<16>): a pixel is a corner when 9 contiguous pixels of the 16-pixel Bresenham
// circle are all brighter than v+t or all darker than v-t; score = the largest threshold for
// which it stays a corner = max over the 16 arcs of min|diff| - 1; keypoints are strict maxima
// of the score over their 8 neighbours; a 3-pixel border is never a corner; detect(img, mask)
// drops keypoints whose mask pixel is 0 AFTER detection.  Integer-exact.
//
// MI355X mapping: one 256-thread workgroup per 64x48 tile; the 72x56 pixel tile and the 72x50 score tile live in
// LDS (every image byte is read from HBM once + halo, scores never go to HBM).  Three phases per tile, all of them
// working on DWORDS of four horizontally adjacent pixels (the kernel is LDS-latency bound, not VALU bound: byte-wide
// LDS reads were 4x the instructions):
//  (1) the 4-pixel necessary test: a lane reads the five dwords that hold the compass pixels of its four positions,
//      survivors are compacted into a wave-private LDS list with ballots (no atomics, no barrier before phase 2);
//  (2) the ~200-op corner score for the compacted candidates only (dense lanes instead of a divergent early-out);
//  (3) 3x3 non-max suppression, again a dword of four scores per lane with its eight neighbouring dwords; quads with no
//      positive score (most) leave after one read.
// Survivors go to an LDS list first.  The front-end engine takes them as PER-TILE lists (tile_kp / tile_count: plain
// coalesced stores, every tile writes its count, no atomics and nothing to wait for -- the returning global atomics of a
// per-cell list were a quarter of the kernel's time); its select kernel bins them into grid cells.  The stand-alone
// detector (av_fast_detect) appends them to one flat list per image with one returning atomic per tile.
// Bound: HBM read of the image (w*h bytes) -- the score arithmetic is ~200 VALU ops per candidate.
#include <stdlib.h>

#include "av_common.h"

namespace {

constexpr int TW = 64, TH = 48;
constexpr int PW = TW + 8, PH = TH + 8;     // pixel tile: columns x0-4 .. x0+67 (pc = x - x0 + 4), rows y0-4 .. y0+51
constexpr int PWD = PW / 4;                 // 18 dwords per pixel row
constexpr int SP = PW, SH = TH + 2;         // score tile: column index = pc (same dword phase as the pixels), row sr = y - y0 + 1
constexpr int SPD = SP / 4;
constexpr int TCAP = TW * TH / 4;           // strict 3x3 maxima in one tile: at most one per 2x2 block

struct FastArgs {
    const uint8_t* img;
    int64_t img_stride;
    int img_pitch;
    int border;                 // valid pixels around the w x h image in every direction (padded pyramid level: AV_PYR_BORDER; raw image: 0)
    const uint8_t* mask;
    int64_t mask_stride;
    int w, h, threshold;
    uint32_t* kp; int* count; int cap;
    uint32_t* tile_kp; int* tile_count;          // [n_img][tiles][TCAP], [n_img][tiles] (tiles row-major: blockIdx.y * gridDim.x + blockIdx.x)
    int* overflow; int stat_stride;
    int dbg;
    int n_img, tiles_x, tiles_y;                 // XCD-aware 1-D launch when tiles_x > 0 (fast_kernel)
    const int* index;                            // optional: image i of the launch is storage entry index[i] (shared frame store)
};

// Corner score of one candidate, the 16 ring differences two to a register (packed 16-bit lanes: ring position j in the low
// half, its antipode j + 8 in the high half, j = 0..7).  The circular sliding minima / maxima of cornerScore<16>
// (windows of 2, 4, then 9 = 4 + 4 + 1) become 8 packed operations per stage instead of 16: "the next position" of the pair
// (j, j + 8) is the pair (j + 1, j + 9), and past the end of the array it is an earlier pair with its halves swapped, which
// the packed instructions take for free through their operand half selects.
typedef short av_s2 __attribute__((ext_vector_type(2)));
// Three-input packed minima / maxima (round 5).  gfx950 has no three-input packed INTEGER min / max, but it has v_pk_minimum3_f16 /
// v_pk_maximum3_f16, and the order of non-negative half-precision bit patterns is the order of the integers they spell: the ring
// differences are biased into 1 .. 511 (subnormal patterns, kept as they are: the kernel runs with fp16 denormals on, the default),
// compared as halves, and un-biased at the end.  A window minimum of nine = min3(min4, min4, far) is then ONE instruction instead of
// two, and two windows fold into the running best at once: 56 packed operations per candidate instead of 80 (all of them half-rate
// instructions, profiles/r05/valu_issue_microbench.json).  Same integers, same scores: tests/test_gpu_ops.py.
typedef _Float16 av_h2 __attribute__((ext_vector_type(2)));
typedef unsigned short av_us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ av_h2 h2_swap(av_h2 x) { return __builtin_shufflevector(x, x, 1, 0); }
__device__ __forceinline__ av_h2 h2_min(av_h2 a, av_h2 b) { return __builtin_elementwise_minimum(a, b); }
__device__ __forceinline__ av_h2 h2_max(av_h2 a, av_h2 b) { return __builtin_elementwise_maximum(a, b); }
__device__ __forceinline__ av_h2 h2_min3(av_h2 a, av_h2 b, av_h2 c) { return __builtin_elementwise_minimum(__builtin_elementwise_minimum(a, b), c); }
__device__ __forceinline__ av_h2 h2_max3(av_h2 a, av_h2 b, av_h2 c) { return __builtin_elementwise_maximum(__builtin_elementwise_maximum(a, b), c); }
__device__ __forceinline__ int fast_score(const uint8_t* c, int t)
{
    // c points at the centre pixel inside the LDS pixel tile (row stride PW)
    constexpr int DX[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
    constexpr int DY[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};
    constexpr int BIAS = 256;
    const int v = c[0];
    const av_us2 vb = {(unsigned short)(v + BIAS), (unsigned short)(v + BIAS)};
    av_h2 D[8];                                     // BIAS + v - ring[j], BIAS + v - ring[j + 8]: 1 .. 511
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const uint32_t pk = (uint32_t)c[DY[j] * PW + DX[j]] | ((uint32_t)c[DY[j + 8] * PW + DX[j + 8]] << 16);
        D[j] = __builtin_bit_cast(av_h2, vb - __builtin_bit_cast(av_us2, pk));
    }
    av_h2 L2[8], H2[8], L4[8], H4[8];               // min / max over 2 and over 4 consecutive ring positions
#pragma unroll
    for (int j = 0; j < 8; ++j) { const av_h2 nx = j < 7 ? D[j + 1] : h2_swap(D[0]); L2[j] = h2_min(D[j], nx); H2[j] = h2_max(D[j], nx); }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        L4[j] = h2_min(L2[j], j < 6 ? L2[j + 2] : h2_swap(L2[j - 6]));
        H4[j] = h2_max(H2[j], j < 6 ? H2[j + 2] : h2_swap(H2[j - 6]));
    }
    av_h2 lo9[8], hi9[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {                   // windows of 9: positions i..i+3, i+4..i+7, i+8
        const av_h2 far = h2_swap(D[j]);
        lo9[j] = h2_min3(L4[j], j < 4 ? L4[j + 4] : h2_swap(L4[j - 4]), far);
        hi9[j] = h2_max3(H4[j], j < 4 ? H4[j + 4] : h2_swap(H4[j - 4]), far);
    }
    av_h2 A = h2_max(lo9[0], lo9[1]), Bm = h2_min(hi9[0], hi9[1]);      // best arc of "centre brighter than ring by at least" / "darker" (negated)
#pragma unroll
    for (int j = 2; j < 8; j += 2) { A = h2_max3(A, lo9[j], lo9[j + 1]); Bm = h2_min3(Bm, hi9[j], hi9[j + 1]); }
    const av_us2 Au = __builtin_bit_cast(av_us2, A), Bu = __builtin_bit_cast(av_us2, Bm);
    const int m = max(max((int)Au.x, (int)Au.y) - BIAS, BIAS - min((int)Bu.x, (int)Bu.y));
    return m > t ? m - 1 : 0;
}

__global__ __launch_bounds__(256) void fast_kernel(FastArgs a)
{
    __shared__ __attribute__((aligned(16))) uint8_t pix[PH * PW + 16];     // +16: the last quad of the last row reads one dword past it
    __shared__ __attribute__((aligned(16))) uint8_t sc[SH * SP];
    // candidate lists, one private segment per wavefront (a wavefront visits every fourth chunk of 64 quads, 4 positions each)
    constexpr int NQ = SH * PWD;                                  // 900 quad positions
    constexpr int SEG = ((NQ + 255) / 256) * 64 * 4;
    __shared__ uint16_t cand[4 * SEG];
    // workgroup -> (image, tile).  1-D launch (a.tiles_x > 0): all tiles of an image on ONE XCD (workgroups are dealt round-robin
    // over the 8 XCDs, each with its own L2): neighbouring tiles share their halo rows and, with 752-byte image rows, most of
    // their 128-byte lines -- spread over eight L2s every line was fetched ~3.7x (PMC, profiles/r03), on one L2 once.
    int img_i, bx, by;
    if (a.tiles_x > 0) {
        const int L = blockIdx.x, j = L >> 3, per = a.tiles_x * a.tiles_y;
        img_i = (L & 7) + 8 * (j / per);
        if (img_i >= a.n_img) return;
        const int t = j % per;
        by = t / a.tiles_x; bx = t - by * a.tiles_x;
    } else { img_i = blockIdx.z; bx = blockIdx.x; by = blockIdx.y; }
    if (a.index) img_i = a.index[img_i];                     // storage entry of this image (shared frame store): image, mask and lists
    const uint8_t* img = a.img + img_i * a.img_stride;
    const int x0 = bx * TW, y0 = by * TH;
    const int tid = threadIdx.x;

    // pixel tile: 56 rows x 72 bytes from (x0-4, y0-4) (x0 is a multiple of 64, so x0-4 is dword aligned when pitch and base are)
    uint32_t* pixw = reinterpret_cast<uint32_t*>(pix);
    // The mask bytes of the four pixels each lane will judge in phase 3 are fetched NOW, with the pixel tile: a mask read
    // behind the non-max suppression was a dependent HBM round trip in the middle of every tile.
    constexpr int NIT3 = (TW / 4) * TH / 256;
    uint32_t mask4[NIT3];
    const bool mask_dw = a.mask && (((a.w | (int)(a.mask_stride & 3) | (int)(reinterpret_cast<uintptr_t>(a.mask) & 3)) & 3) == 0);
#pragma unroll
    for (int it = 0; it < NIT3; ++it) {
        const int i = tid + 256 * it;
        const int r = i / (TW / 4), x = x0 + 4 * (i - r * (TW / 4)), y = y0 + r;
        mask4[it] = 0xFFFFFFFFu;
        if (mask_dw) mask4[it] = (x + 4 <= a.w && y < a.h) ? *reinterpret_cast<const uint32_t*>(a.mask + img_i * a.mask_stride + (size_t)y * a.w + x) : 0u;
    }
    // Tile rows as dwords whenever rows, stride and base are dword aligned (then a dword lies wholly inside or wholly outside the
    // valid columns): all four loads of a thread are issued in one basic block -- as a loop the compiler waited for every load
    // before issuing the next (one dword in flight per thread); rows above / below the valid rows are clamped, dwords left /
    // right of the valid columns read as zero.  Pixels outside the image never reach a valid output (corners need x, y in
    // [3, dim - 3)), so clamp vs reflect vs zero is immaterial.
    const bool dw_rows = ((a.img_pitch | a.w | (int)(a.img_stride & 3) | (int)(reinterpret_cast<uintptr_t>(a.img) & 3)) & 3) == 0;
    if (dw_rows) {
        constexpr int NLD = (PH * PWD + 255) / 256;
        uint32_t v[NLD]; bool in[NLD];
#pragma unroll
        for (int it = 0; it < NLD; ++it) {
            const int i = min(tid + 256 * it, PH * PWD - 1);
            const int r = i / PWD, c = i - r * PWD;
            const int xb = x0 - 4 + 4 * c;
            const int y = min(max(y0 - 4 + r, -a.border), a.h + a.border - 1);
            in[it] = xb >= -a.border && xb + 4 <= a.w + a.border;
            v[it] = *reinterpret_cast<const uint32_t*>(in[it] ? img + (ptrdiff_t)y * a.img_pitch + xb : img);
        }
#pragma unroll
        for (int it = 0; it < NLD; ++it) {
            const int i = tid + 256 * it;
            if (i < PH * PWD) pixw[i] = in[it] ? v[it] : 0u;
        }
    } else {
        for (int i = tid; i < PH * PW; i += 256) {
            int r = i / PW, c = i - r * PW;
            int y = min(max(y0 - 4 + r, 0), a.h - 1), x = min(max(x0 - 4 + c, 0), a.w - 1);
            pix[i] = img[(size_t)y * a.img_pitch + x];
        }
    }
    if (tid < 4) pixw[PH * PWD + tid] = 0;
    __syncthreads();
    // Phase 1: cheap necessary condition on the 4 compass pixels of the ring (any 9-arc contains at least two of them) for the
    // four positions pc = 4q .. 4q+3 of score row sr (pixel row sr+3): dwords U (3 rows up), D (3 rows down), L, C, R of the row.
    uint32_t* scw = reinterpret_cast<uint32_t*>(sc);
    const int lane = tid & 63, wave = tid >> 6;
    int wcnt = 0;                                   // candidates found by this wavefront so far (wavefront-uniform)
    const int t = a.threshold;
    for (int i0 = 0; i0 < NQ; i0 += 256) {
        const int i = i0 + tid;
        bool cf[4] = {false, false, false, false};
        int sr = 0, q = 0;
        if (i < NQ) {
            sr = i / PWD; q = i - sr * PWD;
            scw[i] = 0;
            const int y = y0 - 1 + sr;
            if (y >= 3 && y < a.h - 3 && !(a.dbg & 2)) {
                const uint32_t* rowc = pixw + (sr + 3) * PWD + q;
                const uint32_t C = rowc[0], L = q > 0 ? rowc[-1] : 0u, R = rowc[1];
                const uint32_t U = rowc[-3 * PWD], D = rowc[3 * PWD];
                const uint32_t R3 = __builtin_amdgcn_alignbyte(R, C, 3), L3 = __builtin_amdgcn_alignbyte(C, L, 1);      // p[+3], p[-3] of the four positions
                // Packed 16-bit lanes, two pixels per register.  Any 9-arc of the 16-ring holds at least one pixel of EVERY
                // antipodal pair, so "centre brighter than one of (p0, p8) AND than one of (p4, p12)" -- or the same for darker --
                // is necessary for a corner (OpenCV's own first rejection tests):
                //   bright: min(max(d0, d8), max(d4, d12)) > t     dark: max(min(d0, d8), min(d4, d12)) < -t
                auto half = [](uint32_t x, int hi) -> av_s2 { return __builtin_bit_cast(av_s2, __builtin_amdgcn_perm(0, x, hi ? 0x0C030C02u : 0x0C010C00u)); };
                const av_s2 tt = {(short)t, (short)t}, zz = {0, 0};
                uint32_t fl[2];
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    const av_s2 c = half(C, hf);
                    const av_s2 d0 = c - half(D, hf), d8 = c - half(U, hf), d4 = c - half(R3, hf), d12 = c - half(L3, hf);
                    const av_s2 mb = __builtin_elementwise_min(__builtin_elementwise_max(d0, d8), __builtin_elementwise_max(d4, d12));
                    const av_s2 md = __builtin_elementwise_max(__builtin_elementwise_min(d0, d8), __builtin_elementwise_min(d4, d12));
                    const av_s2 val = __builtin_elementwise_max(mb, zz - md);
                    fl[hf] = __builtin_bit_cast(uint32_t, tt - val);             // negative (bit 15 of its half set) iff val > t
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int pc = 4 * q + k, x = x0 - 4 + pc;
                    const bool hit = (fl[k >> 1] >> (15 + 16 * (k & 1))) & 1u;
                    cf[k] = hit && pc >= 3 && pc <= TW + 4 && x >= 3 && x < a.w - 3;
                }
            }
        }
        // ordered compaction into this wavefront's own segment: no LDS atomic, no cross-lane broadcast
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned long long b = __ballot(cf[k]);
            if (cf[k]) cand[wave * SEG + wcnt + __builtin_amdgcn_mbcnt_hi((unsigned)(b >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b, 0))] = (uint16_t)(sr * SP + 4 * q + k);
            wcnt += __popcll(b);
        }
    }
    // Phase 2: full corner score only for the candidates -- each wavefront scores its own list (the LDS traffic of one
    // wavefront is ordered, so no workgroup barrier is needed between the two phases)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (!(a.dbg & 1)) for (int k = lane; k < wcnt; k += 64) {
        const int i = cand[wave * SEG + k];
        const int sr = i / SP, pc = i - sr * SP;
        sc[i] = (uint8_t)fast_score(&pix[(sr + 3) * PW + pc], t);
    }
    __syncthreads();
    // Phase 3: 3x3 non-max suppression + mask; survivors go to an LDS list first so that the tile issues
    // ONE returning global atomic per output list it touches (a 64x16 tile overlaps <= 4 grid cells)
    // instead of one per keypoint.
    __shared__ uint32_t surv_word[TCAP];
    __shared__ int nsurv, flat_base;
    if (tid == 0) nsurv = 0;
    __syncthreads();
    // quads of the tile interior: columns pc = 4q .. 4q+3, q = 1..16 (x = x0 .. x0+63), score rows sr = 1..48 (y = y0 .. y0+47)
#pragma unroll
    for (int it = 0; it < (TW / 4) * TH / 256; ++it) {
        const int i = tid + 256 * it;
        const int r = i / (TW / 4), q = 1 + (i - r * (TW / 4));
        const uint32_t* rowc = scw + (r + 1) * SPD + q;
        const uint32_t B = rowc[0];
        if (B == 0 || (a.dbg & 4)) continue;                                 // no positive score among the four pixels
        // per score row the 6-byte window [pc-1 .. pc+4] as (lo, hi): pixel k sees bytes k, k+1, k+2
        uint32_t lo[3], hi[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const uint32_t* rp = rowc + (d - 1) * SPD;
            const uint32_t A = rp[-1], Bm = rp[0], Cn = rp[1];
            lo[d] = __builtin_amdgcn_alignbyte(Bm, A, 3); hi[d] = __builtin_amdgcn_alignbyte(Cn, Bm, 3);
        }
        // strict 3x3 maximum test for the four pixels at once, two pixels per register (packed u16): byte w of the 8-byte
        // window (lo, hi) of a row is column pc - 1 + w; pixel k sees w = k (left), k + 1 (centre), k + 2 (right)
        typedef unsigned short av_u2 __attribute__((ext_vector_type(2)));
        auto pair = [&](int d, uint32_t sel) -> av_u2 { return __builtin_bit_cast(av_u2, __builtin_amdgcn_perm(hi[d], lo[d], sel)); };
        constexpr uint32_t P01 = 0x0C010C00u, P12 = 0x0C020C01u, P23 = 0x0C030C02u, P34 = 0x0C040C03u, P45 = 0x0C050C04u;
        // (scores are 0 .. 254: as half-precision bit patterns they order like the integers -- one v_pk_maximum3_f16 per three, see fast_score)
        auto mx3 = [](av_u2 x, av_u2 y, av_u2 z) -> av_u2 { return __builtin_bit_cast(av_u2, h2_max3(__builtin_bit_cast(av_h2, x), __builtin_bit_cast(av_h2, y), __builtin_bit_cast(av_h2, z))); };
        const av_u2 n01 = mx3(mx3(pair(0, P01), pair(0, P12), pair(0, P23)), mx3(pair(2, P01), pair(2, P12), pair(2, P23)), mx3(pair(1, P01), pair(1, P23), pair(1, P23)));
        const av_u2 n23 = mx3(mx3(pair(0, P23), pair(0, P34), pair(0, P45)), mx3(pair(2, P23), pair(2, P34), pair(2, P45)), mx3(pair(1, P23), pair(1, P45), pair(1, P45)));
        const av_u2 g01 = __builtin_elementwise_sub_sat(pair(1, P12), n01), g23 = __builtin_elementwise_sub_sat(pair(1, P34), n23);
        const uint32_t gt[2] = {__builtin_bit_cast(uint32_t, g01), __builtin_bit_cast(uint32_t, g23)};      // half k & 1 of gt[k >> 1] != 0: score > all 8 neighbours
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int s = (int)((B >> (8 * k)) & 0xFF);
            const int x = x0 + 4 * (q - 1) + k, y = y0 + r;
            bool keep = ((gt[k >> 1] >> (16 * (k & 1))) & 0xFFFFu) != 0 && x < a.w && y < a.h;
            if (keep && a.mask) keep = mask_dw ? ((mask4[it] >> (8 * k)) & 0xFF) != 0 : a.mask[img_i * a.mask_stride + (size_t)y * a.w + x] != 0;
            if (keep && !(a.dbg & 8)) {
                const int slot = atomicAdd(&nsurv, 1);            // LDS atomic; a tile has <= TCAP strict maxima
                surv_word[slot] = ((uint32_t)s << AV_KP_RASTER_BITS) | (AV_KP_RASTER_MASK - (uint32_t)(y * a.w + x));
            }
        }
    }
    __syncthreads();
    const int ns = nsurv;
    if (a.tile_kp) {
        const size_t tile = a.tiles_x > 0 ? (size_t)img_i * (a.tiles_x * a.tiles_y) + by * a.tiles_x + bx
                                          : (size_t)img_i * (gridDim.x * gridDim.y) + blockIdx.y * gridDim.x + blockIdx.x;
        if (tid == 0) a.tile_count[tile] = ns;
        for (int q = tid; q < ns; q += 256) a.tile_kp[tile * TCAP + q] = surv_word[q];
        return;
    }
    if (ns == 0) return;
    if (tid == 0) flat_base = atomicAdd(&a.count[img_i], ns);
    __syncthreads();
    for (int q = tid; q < ns; q += 256) {
        const int idx = flat_base + q;
        if (idx < a.cap) a.kp[(size_t)img_i * a.cap + idx] = surv_word[q];
        else if (a.overflow) atomicOr(&a.overflow[img_i * a.stat_stride], 1);
    }
}

}  // namespace

void av_fast_tiles(int w, int h, int* tiles, int* tile_cap)
{
    *tiles = ((w + TW - 1) / TW) * ((h + TH - 1) / TH);
    *tile_cap = TCAP;
}

// Exactly one of (kp, count, cap) -- a flat list per image -- and (tile_kp, tile_count) -- per-tile lists, see av_fast_tiles -- is given.
int av_launch_fast(const uint8_t* img, int64_t img_stride, int img_pitch, int border, const uint8_t* mask, int64_t mask_stride,
                   int n_img, int w, int h, int threshold,
                   uint32_t* kp, int* count, int cap, uint32_t* tile_kp, int* tile_count,
                   int* overflow, int stat_stride, hipStream_t st, const int* index)
{
    if (n_img <= 0) return AV_OK;
    if ((int64_t)w * h > (int64_t)(AV_KP_RASTER_MASK + 1)) {
        av_set_error("av_fast_detect: image %dx%d exceeds 2^19 pixels", w, h);
        return AV_E_INVALID;
    }
    FastArgs a;
    a.img = img; a.img_stride = img_stride; a.img_pitch = img_pitch; a.border = border; a.mask = mask; a.mask_stride = mask_stride;
    a.w = w; a.h = h; a.threshold = threshold;
    a.kp = kp; a.count = count; a.cap = cap;
    a.tile_kp = tile_kp; a.tile_count = tile_count; a.overflow = overflow; a.stat_stride = stat_stride;
    { const char* e = getenv("AV_FAST_DBG"); a.dbg = e ? atoi(e) : 0; }
    static const bool xcd_map = [] { const char* e = getenv("AV_FAST_XCD"); return !(e && atoi(e) == 0); }();      // A/B switch
    const int tx = (w + TW - 1) / TW, ty = (h + TH - 1) / TH;
    a.n_img = n_img; a.tiles_x = xcd_map ? tx : 0; a.tiles_y = ty; a.index = index;
    dim3 grid = xcd_map ? dim3((unsigned)(tx * ty) * 8u * (unsigned)((n_img + 7) / 8)) : dim3(tx, ty, n_img);
    // (An occupancy throttle -- unused dynamic LDS holding the detector to 6 / 5 / 4 workgroups per CU so that the filter's kernels find
    //  room beside it -- was measured in round 5: 156.9 / 154.7 / 149.7 k against 157.5 k frames/s.  profiles/r05/README.md)
    // (The same 64 x 48 tile worked by two / one wavefront instead of four -- a workgroup of fewer wavefronts is easier to place beside the
    //  filter's single-wavefront tasks -- measured in round 5: detector 1.71 / 2.43 against 1.51 ms alone, 2.60 / 3.38 against 2.21 ms in
    //  the shared run, 170.2 / 159.5 against 173.4 k frames/s: with 19 KB of LDS per tile the CU then holds half / a quarter of the
    //  wavefronts.  What the detector needs is a wavefront that walks a column of small tiles.  profiles/r05/README.md)
    hipLaunchKernelGGL(fast_kernel, grid, dim3(256), 0, st, a);
    AV_LAUNCH_CHECK();
    return AV_OK;
}

AV_EXPORT int av_fast_detect(const uint8_t* img_dev, int64_t img_stride, const uint8_t* mask_dev, int64_t mask_stride,
                             int n_img, int w, int h, int threshold, uint32_t* kp_dev, int32_t* count_dev, int cap,
                             void* stream)
{
    if (!img_dev || !kp_dev || !count_dev || cap <= 0 || n_img < 0 || w < 7 || h < 7 || img_stride < (int64_t)w * h) {
        av_set_error("av_fast_detect: bad arguments");
        return AV_E_INVALID;
    }
    hipStream_t st = (hipStream_t)stream;
    AV_HIP(hipMemsetAsync(count_dev, 0, sizeof(int) * (size_t)n_img, st));
    return av_launch_fast(img_dev, img_stride, w, 0, mask_dev, mask_stride, n_img, w, h, threshold, kp_dev, count_dev, cap,
                          nullptr, nullptr, nullptr, 0, st);
}
