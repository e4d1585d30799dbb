// frontend.hip -- the image front-end as a device-resident engine over S independent stereo streams.
//
// Reference being replaced (all under src/image_processing/):
//   pipeline.py:14-150            ImageProcessingPipeline.{__init__, imu_callback, stereo_callback}
//   imu_processor.py:22-67        IMU buffer, mean-gyro rotation prediction
//   feature_initializer.py:45-85  first frame: FAST -> stereo match all -> top-3/cell -> ids
//   feature_tracker.py:74-177     temporal LK, bounds mask, stereo re-match, re-bin, lifetime+1
//   stereo_matcher.py:33-115      predict -> LK fwd -> LK bwd -> gates (err<3, |dy|<20, bounds, epipolar)
//   feature_adder.py:52-108       7x7 mask, FAST(mask), top-5/cell, stereo match, top-3/cell, ids
//   feature_pruner.py:8-19        per cell keep top-5 by lifetime (stable)
//   feature_publisher.py:90-121   undistort to normalised coords -> feature_msg
//
// The reference runs one stream, one frame at a time, with Python lists of FeatureMetaData.  Here
// the per-stream state (feature grid, ids, three padded pyramids, FAST mask) lives in HBM for all S
// streams; one call to av_frontend_step enqueues ~15 batched kernels (grid.y or block = stream) and
// never synchronises with the host.  Frames of one stream are a dependent chain (LK at t needs the
// points of t-1), streams are independent, so S is the parallel axis that fills 256 CUs.
//
// Order-sensitive list semantics of the reference are reproduced exactly:
//   * `select(...)` compactions keep order  -> ballot/popcount ordered compaction per stream
//   * Python's stable `sorted(key=response/lifetime, reverse=True)` -> explicit (key, insertion index)
//     total orders: FAST candidates by (score desc, raster asc), pruning by (lifetime desc, insertion)
//   * ids are handed out cell-major, response-descending inside a cell
//   * the 7x7 FAST mask follows numpy slice semantics (no masking when x<3 or y<3)
#include <math.h>

#include <deque>
#include <mutex>
#include <new>
#include <vector>

#include "av_common.h"

namespace {

constexpr int CNT_BEFORE = 0, CNT_TRACKED = 1, CNT_MATCHED = 2, CNT_FAST = 3, CNT_CAND = 4, CNT_NEW = 5, CNT_PUB = 6, CNT_OVF = 7;
constexpr int NCNT = 8;
constexpr unsigned long long LIFE_MAX = 0xFFFFFFull;

// everything the glue kernels need, by value
struct FeDev {
    int S, w, h, C, grid_row, grid_col, gh, gw, gmin, gmax;
    int MAXF;        // capacity of a published grid  = C * gmax
    int NT;          // capacity of tracker work lists = MAXF
    int CC;          // capacity of the candidate list = max(max_corners, C * gmax)
    int cell_cap;    // capacity of one cell's FAST list
    int NSORT;       // pow2 >= C * (gmax + gmin)
    CamModel cam0, cam1;
    double R0to1[9], E[9], I3[9];
    double epi_thr;
    // published grid, double buffered (index = parity)
    long long* feat_id[2]; int* feat_life[2]; float* feat_p0[2]; float* feat_p1[2]; int* feat_cell[2]; int* n_feat[2];
    long long* next_id; int* first_frame;
    const double* Hmat;                                   // [S][9]
    // temporal tracking
    float* trk_prev; float* trk_next; uint8_t* trk_status; int* trk_count;
    // survivors of temporal tracking, stereo-matched
    int* sv_src; float* sv_p0; float* sv_init; float* sv_p1; uint8_t* sv_st; float* sv_back; uint8_t* sv_st2; int* sv_count;
    // curr_features after the tracker (insertion order)
    long long* cur_id; int* cur_life; float* cur_p0; float* cur_p1; int* cur_cell; int* cur_count;
    // FAST per-cell lists + mask
    uint32_t* cell_kp; int* cell_count; uint8_t* mask;
    uint32_t* tile_kp; int* tile_count; int n_tiles, tile_cap;      // FAST survivors per detector tile (fast.hip), binned into cells by select_kernel
    // candidates for new features
    uint32_t* cand_key; float* cand_p0; float* cand_init; float* cand_p1; uint8_t* cand_st; float* cand_back; uint8_t* cand_st2;
    uint8_t* cand_inl; int* cand_off; int* cand_count;
    int* r1_list; int* r2_list; int* r1_count; int* r2_count;      // lazy candidate matching: indices matched in round 1 / round 2
    // output
    long long* out_ids; double* out_uv; int* out_n;
    int* counters;
    // Shared frame store (av_frontend_frames_*): stream s reads storage entry slot_cur[s] of the FAST lists (and, through the maps
    // handed to the LK launches, of the images and pyramids); < 0 = the stream has no frame in this step: its workgroups return at
    // once and its state stays as it is.  Null = every stream owns entry s (av_frontend_step / _step_host).
    const int* slot_cur;
};

__device__ __forceinline__ bool fe_idle(const FeDev& d, int s) { return d.slot_cur != nullptr && d.slot_cur[s] < 0; }

__device__ __forceinline__ int cell_of(const FeDev& d, float x, float y)
{
    // int(pt[1] / grid_h) * grid_col + int(pt[0] / grid_w)  (feature_tracker.py:144-146); float32 division
    return (int)(y / (float)d.gh) * d.grid_col + (int)(x / (float)d.gw);
}

// workgroup size of the per-stream glue kernels below: 256 threads, or 64 (one wavefront per stream: step_impl picks by batch size)
#define NTB ((int)blockDim.x)
#define NWB ((int)(blockDim.x >> 6))
// ordered compaction position inside the workgroup; `base` is a running total in registers
__device__ __forceinline__ int block_ordered_pos(bool flag, int& base, int* lds4)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned long long b = __ballot(flag);
    int lane_prefix = __popcll(b & ((1ull << lane) - 1ull));
    if (lane == 0) lds4[wv] = __popcll(b);
    __syncthreads();
    int off = 0, tot = 0;
    for (int i = 0; i < NWB; ++i) { int c = lds4[i]; if (i < wv) off += c; tot += c; }
    int pos = base + off + lane_prefix;
    base += tot;
    __syncthreads();
    return pos;
}

// stereo_matcher.py:47-61: initial guess in cam1 = distort(undistort(p, R0to1)) with the cam0 model
__device__ __forceinline__ void stereo_init(const FeDev& d, float x, float y, float& ox, float& oy)
{
    double ux, uy;
    av_undistort(d.cam0, d.R0to1, (double)x, (double)y, ux, uy);
    float fx = (float)ux, fy = (float)uy;                      // undistortPoints output is float32
    double px, py;
    av_distort(d.cam0, (double)fx, (double)fy, px, py);
    ox = (float)px; oy = (float)py;                            // projectPoints output is float32
}

// stereo_matcher.py:75-113
constexpr int CAND_R1 = 5;      // candidates of a cell matched in round 1 (grid_min = 3 in the reference: + 2 spares)

__device__ __forceinline__ bool stereo_gate(const FeDev& d, float p0x, float p0y, float inx, float iny, float p1x, float p1y,
                                            uint8_t st, float bx, float by)
{
    (void)inx;
    if (!st) return false;
    float ex = p0x - bx, ey = p0y - by;
    float err = sqrtf(ex * ex + ey * ey);
    if (!(err < 3.f)) return false;
    float disp = fabsf(iny - p1y);
    if (!(disp < 20.f)) return false;
    if (p1x < 0 || p1x >= (float)d.w || p1y < 0 || p1y >= (float)d.h) return false;
    double ax, ay, bxx, byy;
    av_undistort(d.cam0, d.I3, (double)p0x, (double)p0y, ax, ay);
    av_undistort(d.cam0, d.I3, (double)p1x, (double)p1y, bxx, byy);
    double u0x = (double)(float)ax, u0y = (double)(float)ay, u1x = (double)(float)bxx;
    double l0 = (d.E[0] * u0x + d.E[1] * u0y) + d.E[2] * 1.0;
    double l1 = (d.E[3] * u0x + d.E[4] * u0y) + d.E[5] * 1.0;
    double err_epi = fabs(u1x * l0) / sqrt(l0 * l0 + l1 * l1);
    if (err_epi > d.epi_thr) return false;
    return true;
}

// ---- G1: predicted positions for temporal tracking (feature_tracker.py:86-101,159-177) --------
__global__ __launch_bounds__(256) void track_prepare_kernel(FeDev d, int par)
{
    const int s = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (fe_idle(d, s)) return;
    const int n = d.n_feat[par][s];
    if (i == 0) { d.trk_count[s] = n; d.counters[s * NCNT + CNT_BEFORE] = n; }
    if (i >= n) return;
    const size_t k = (size_t)s * d.MAXF + i, t = (size_t)s * d.NT + i;
    float x = d.feat_p0[par][2 * k], y = d.feat_p0[par][2 * k + 1];
    const double* H = d.Hmat + s * 9;
    double h0 = (H[0] * (double)x + H[1] * (double)y) + H[2] * 1.0;
    double h1 = (H[3] * (double)x + H[4] * (double)y) + H[5] * 1.0;
    double h2 = (H[6] * (double)x + H[7] * (double)y) + H[8] * 1.0;
    d.trk_prev[2 * t] = x; d.trk_prev[2 * t + 1] = y;
    d.trk_next[2 * t] = (float)(h0 / h2); d.trk_next[2 * t + 1] = (float)(h1 / h2);
}

// ---- G2: bounds mask + ordered compaction + stereo initial guess (feature_tracker.py:110-126) --
__global__ __launch_bounds__(256) void track_gate_kernel(FeDev d)
{
    __shared__ int lds4[4];
    const int s = blockIdx.x;
    if (fe_idle(d, s)) return;
    const int n = d.trk_count[s];
    int base = 0;
    for (int i0 = 0; i0 < n; i0 += NTB) {
        const int i = i0 + threadIdx.x;
        const size_t t = (size_t)s * d.NT + i;
        bool keep = false; float x = 0, y = 0;
        if (i < n) {
            x = d.trk_next[2 * t]; y = d.trk_next[2 * t + 1];
            keep = d.trk_status[t] != 0 && !(x < 0 || x > (float)(d.w - 1) || y < 0 || y > (float)(d.h - 1));
        }
        int pos = block_ordered_pos(keep, base, lds4);
        if (keep) {
            const size_t o = (size_t)s * d.NT + pos;
            d.sv_src[o] = i;
            d.sv_p0[2 * o] = x; d.sv_p0[2 * o + 1] = y;
            float ix, iy;
            stereo_init(d, x, y, ix, iy);
            d.sv_init[2 * o] = ix; d.sv_init[2 * o + 1] = iy;
            d.sv_p1[2 * o] = ix; d.sv_p1[2 * o + 1] = iy;
            d.sv_back[2 * o] = x; d.sv_back[2 * o + 1] = y;
        }
    }
    if (threadIdx.x == 0) { d.sv_count[s] = base; d.counters[s * NCNT + CNT_TRACKED] = base; }
}

// 7x7 box of the FAST mask with numpy slice semantics (feature_adder.py:59-62)
__device__ __forceinline__ void mask_box(const FeDev& d, uint8_t* m, float px, float py, int j, uint8_t val)
{
    int fx = (int)px, fy = (int)py;
    if (fx < 3 || fy < 3) return;                 // negative slice start => empty slice
    int yy = fy - 3 + j / 7, xx = fx - 3 + j % 7;
    if (yy < d.h && xx < d.w) m[(size_t)yy * d.w + xx] = val;
}

// ---- G3: stereo gate of the survivors, re-bin into curr_features, set FAST mask ----------------
__global__ __launch_bounds__(256) void rebin_kernel(FeDev d, int par)
{
    __shared__ int lds4[4];
    const int s = blockIdx.x;
    if (fe_idle(d, s)) return;
    const int n = d.sv_count[s];
    int base = 0;
    for (int i0 = 0; i0 < n; i0 += NTB) {
        const int i = i0 + threadIdx.x;
        const size_t t = (size_t)s * d.NT + i;
        bool keep = false;
        if (i < n)
            keep = stereo_gate(d, d.sv_p0[2 * t], d.sv_p0[2 * t + 1], d.sv_init[2 * t], d.sv_init[2 * t + 1],
                               d.sv_p1[2 * t], d.sv_p1[2 * t + 1], d.sv_st[t], d.sv_back[2 * t], d.sv_back[2 * t + 1]);
        int pos = block_ordered_pos(keep, base, lds4);
        if (keep) {
            const size_t o = (size_t)s * d.NT + pos;
            const size_t src = (size_t)s * d.MAXF + d.sv_src[t];
            d.cur_id[o] = d.feat_id[par][src];
            d.cur_life[o] = d.feat_life[par][src] + 1;
            float x = d.sv_p0[2 * t], y = d.sv_p0[2 * t + 1];
            d.cur_p0[2 * o] = x; d.cur_p0[2 * o + 1] = y;
            d.cur_p1[2 * o] = d.sv_p1[2 * t]; d.cur_p1[2 * o + 1] = d.sv_p1[2 * t + 1];
            d.cur_cell[o] = cell_of(d, x, y);
        }
    }
    if (threadIdx.x == 0) { d.cur_count[s] = base; d.counters[s * NCNT + CNT_MATCHED] = base; }
    __syncthreads();
    uint8_t* m = d.mask + (size_t)s * d.w * d.h;
    for (int q = threadIdx.x; q < base * 49; q += NTB) {
        int f = q / 49, j = q - f * 49;
        const size_t o = (size_t)s * d.NT + f;
        mask_box(d, m, d.cur_p0[2 * o], d.cur_p0[2 * o + 1], j, 0);
    }
}

__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        unsigned long long o = __shfl_xor(v, m, 64);
        v = o > v ? o : v;
    }
    return v;
}

// ---- G5: per-cell top-grid_max FAST candidates (feature_adder.py:66-80) or all of them on the
//          first frame (feature_initializer.py:52-55), plus their stereo initial guess ----------
__global__ __launch_bounds__(256) void select_kernel(FeDev d)
{
    extern __shared__ int sm[];
    int* cnt = sm;               // [C]
    int* off = sm + d.C;         // [C+1]
    int* ccnt = off + d.C + 1;   // [C]   FAST keypoints per cell
    int* tpre = ccnt + d.C;      // [n_tiles + 1] prefix of the detector's per-tile counts
    const int s = blockIdx.x;
    if (fe_idle(d, s)) return;
    const int ts = d.slot_cur ? d.slot_cur[s] : s;              // storage entry of this frame's FAST lists
    const bool first = d.first_frame[s] != 0;
    // bin the detector's per-tile survivor lists into the per-cell lists (feature_adder.py:66-71: row = y / grid_height,
    // col = x / grid_width).  The cell lists are unordered; everything below orders by (response, raster) key.
    // Dense mapping, no search: slot j of a chunk is entry (j % BK) + kb of tile j / BK; a round issues BU independent loads
    // per thread (list word, then its mask byte) -- this loop is a chain of memory round trips, not work.
    {
        __shared__ int tmax;
        const int nt = d.n_tiles;
        constexpr int BK = 32, BU = 5;
        if (threadIdx.x == 0) tmax = 0;
        for (int c = threadIdx.x; c < d.C; c += NTB) ccnt[c] = 0;
        __syncthreads();
        int mx = 0;
        for (int t = threadIdx.x; t < nt; t += NTB) { const int n = d.tile_count[(size_t)ts * nt + t]; tpre[t] = n; mx = max(mx, n); }
        if (mx > 0) atomicMax(&tmax, mx);
        __syncthreads();
        const int maxcount = tmax;
        for (int kb = 0; kb < maxcount; kb += BK)
            for (int j0 = threadIdx.x; j0 < nt * BK; j0 += NTB * BU) {
                uint32_t word[BU]; int x[BU], y[BU]; bool ok[BU]; uint8_t mk[BU];
#pragma unroll
                for (int u = 0; u < BU; ++u) {
                    const int j = j0 + NTB * u, t = j / BK, k = kb + (j % BK);
                    ok[u] = j < nt * BK && k < tpre[min(t, nt - 1)];
                    word[u] = ok[u] ? d.tile_kp[((size_t)ts * nt + t) * d.tile_cap + k] : 0u;
                }
#pragma unroll
                for (int u = 0; u < BU; ++u) {
                    const uint32_t raster = AV_KP_RASTER_MASK - (word[u] & AV_KP_RASTER_MASK);
                    y[u] = (int)(raster / (uint32_t)d.w); x[u] = (int)(raster % (uint32_t)d.w);
                    // detect(img, mask) drops keypoints on masked pixels AFTER the non-max suppression (fast.hip header): the mask
                    // byte is read here, for the few thousand survivors, instead of for every pixel inside the detector
                    mk[u] = ok[u] ? d.mask[(size_t)s * d.w * d.h + (size_t)y[u] * d.w + x[u]] : (uint8_t)0;
                }
#pragma unroll
                for (int u = 0; u < BU; ++u) {
                    if (!ok[u] || mk[u] == 0) continue;
                    const int cell = (y[u] / d.gh) * d.grid_col + x[u] / d.gw;
                    const int idx = atomicAdd(&ccnt[cell], 1);
                    if (idx < d.cell_cap) d.cell_kp[((size_t)s * d.C + cell) * d.cell_cap + idx] = word[u];
                    else atomicOr(&d.counters[s * NCNT + CNT_OVF], 2);
                }
            }
        __threadfence_block();
        __syncthreads();
        if (threadIdx.x == 0) { int tot = 0; for (int c = 0; c < d.C; ++c) tot += ccnt[c]; d.counters[s * NCNT + CNT_FAST] = tot; }
    }
    for (int c = threadIdx.x; c < d.C; c += NTB) {
        int n = min(ccnt[c], d.cell_cap);
        cnt[c] = first ? n : min(n, d.gmax);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int acc = 0;
        for (int c = 0; c < d.C; ++c) {
            off[c] = acc;
            int take = cnt[c];
            if (acc + take > d.CC) { take = d.CC - acc; atomicOr(&d.counters[s * NCNT + CNT_OVF], 4); }
            cnt[c] = take;
            acc += take;
        }
        off[d.C] = acc;
        d.cand_count[s] = acc;
        d.counters[s * NCNT + CNT_CAND] = acc;
    }
    __syncthreads();
    for (int c = threadIdx.x; c <= d.C; c += NTB) d.cand_off[s * (d.C + 1) + c] = off[c];
    // Round 1 of the stereo matching (see cand_round2_kernel): the first R1 candidates of every cell, all of them on the first frame
    {
        __shared__ int r1n;
        if (threadIdx.x == 0) r1n = 0;
        __syncthreads();
        for (int c = threadIdx.x; c < d.C; c += NTB) {
            const int take = first ? cnt[c] : min(cnt[c], CAND_R1);
            const int base = atomicAdd(&r1n, take);                     // order of the list is irrelevant: results are written by index
            for (int r = 0; r < take; ++r) d.r1_list[(size_t)s * d.CC + base + r] = off[c] + r;
        }
        __syncthreads();
        if (threadIdx.x == 0) d.r1_count[s] = r1n;
    }

    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int c = wv; c < d.C; c += NWB) {
        const uint32_t* list = d.cell_kp + ((size_t)s * d.C + c) * d.cell_cap;
        const int n = min(ccnt[c], d.cell_cap);
        const int take = cnt[c];
        uint32_t* dst = d.cand_key + (size_t)s * d.CC + off[c];
        if (first) {
            for (int i = lane; i < take; i += 64) dst[i] = list[i];
        } else {
            // the first 4 x 64 keys of the list live in registers for all `take` selection rounds (re-reading the list from HBM
            // every round made this kernel a chain of ~75 memory round trips per wavefront); longer lists read the rest per round
            constexpr int KR = 4;
            uint32_t kreg[KR];
#pragma unroll
            for (int u = 0; u < KR; ++u) kreg[u] = lane + 64 * u < n ? list[lane + 64 * u] : 0u;
            uint32_t last = 0xFFFFFFFFu;
            for (int r = 0; r < take; ++r) {
                uint32_t best = 0;
#pragma unroll
                for (int u = 0; u < KR; ++u) if (kreg[u] < last && kreg[u] > best) best = kreg[u];
                for (int i = lane + 64 * KR; i < n; i += 64) { uint32_t k = list[i]; if (k < last && k > best) best = k; }
                best = (uint32_t)wave_max_u64(best);
                if (lane == 0) dst[r] = best;
                last = best;
            }
        }
    }
    __syncthreads();
    const int total = off[d.C];
    for (int i = threadIdx.x; i < total; i += NTB) {
        const size_t o = (size_t)s * d.CC + i;
        uint32_t raster = AV_KP_RASTER_MASK - (d.cand_key[o] & AV_KP_RASTER_MASK);
        float y = (float)(raster / (uint32_t)d.w), x = (float)(raster % (uint32_t)d.w);
        d.cand_p0[2 * o] = x; d.cand_p0[2 * o + 1] = y;
        float ix, iy;
        stereo_init(d, x, y, ix, iy);
        d.cand_init[2 * o] = ix; d.cand_init[2 * o + 1] = iy;
        d.cand_p1[2 * o] = ix; d.cand_p1[2 * o + 1] = iy;
        d.cand_back[2 * o] = x; d.cand_back[2 * o + 1] = y;
    }
}

// ---- G6: lazy stereo matching of the candidates.  feature_adder.py:80-108 matches every candidate (up to grid_max per
// cell) and then keeps, per cell, the grid_min inliers of highest response.  Candidates are sorted by response inside their
// cell, so once the first R1 of a cell hold grid_min inliers the rest of the cell cannot change the result: round 1 matches the
// first R1 = grid_min + 2 candidates of every cell, this kernel gates them and lists for round 2 the remaining candidates of
// the (rare) cells that are still short.  Same ids, same points, ~3x fewer candidate LK passes at grid_max = 15.
__global__ __launch_bounds__(256) void cand_round2_kernel(FeDev d)
{
    __shared__ int r2n;
    const int s = blockIdx.x;
    if (fe_idle(d, s)) return;
    const int* coff = d.cand_off + s * (d.C + 1);
    const bool first = d.first_frame[s] != 0;
    if (threadIdx.x == 0) r2n = 0;
    __syncthreads();
    for (int c = threadIdx.x; c < d.C; c += NTB) {
        const int b = coff[c], e = coff[c + 1];
        const int r1 = first ? (e - b) : min(e - b, CAND_R1);
        int inl = 0;
        for (int i = b; i < b + r1; ++i) {
            const size_t o = (size_t)s * d.CC + i;
            const bool ok = stereo_gate(d, d.cand_p0[2 * o], d.cand_p0[2 * o + 1], d.cand_init[2 * o], d.cand_init[2 * o + 1],
                                        d.cand_p1[2 * o], d.cand_p1[2 * o + 1], d.cand_st[o], d.cand_back[2 * o], d.cand_back[2 * o + 1]);
            d.cand_inl[o] = ok ? 1 : 0;
            inl += ok;
        }
        const bool more = inl < d.gmin && b + r1 < e;
        for (int i = b + r1; i < e; ++i) d.cand_inl[(size_t)s * d.CC + i] = more ? 2 : 0;      // 2 = matched in round 2, gated by finalize
        if (more) {
            const int base = atomicAdd(&r2n, e - b - r1);
            for (int i = b + r1; i < e; ++i) d.r2_list[(size_t)s * d.CC + base + (i - b - r1)] = i;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) d.r2_count[s] = r2n;
}

__device__ __forceinline__ void bitonic_sort_u64(unsigned long long* k, int n)
{
    for (int size = 2; size <= n; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (int t = threadIdx.x; t < (n >> 1); t += NTB) {
                int lo = 2 * t - (t & (stride - 1));
                int hi = lo + stride;
                bool up = (lo & size) == 0;
                unsigned long long a = k[lo], b = k[hi];
                if ((a > b) == up) { k[lo] = b; k[hi] = a; }
            }
        }
    __syncthreads();
}

// ---- G7: new-feature selection + ids, prune, publish, mask restore ------------------------------
//  feature_adder.py:82-108 / feature_initializer.py:57-85, feature_pruner.py:8-19,
//  feature_publisher.py:90-121, pipeline.py:145-148
__global__ __launch_bounds__(256) void finalize_kernel(FeDev d, int par)
{
    extern __shared__ unsigned long long smem64[];
    unsigned long long* keys = smem64;                               // [NSORT]
    int* ism = reinterpret_cast<int*>(keys + d.NSORT);
    int* sel = ism;                   ism += d.C * d.gmin;           // candidate index of selection (c, r)
    int* sel_n = ism;                 ism += d.C;
    int* sel_pre = ism;               ism += d.C + 1;
    int* trk_n = ism;                 ism += d.C;                    // tracked features per cell
    int* cell_start = ism;            ism += d.C + 1;                // start of each cell in the sorted order
    int* out_start = ism;             ism += d.C + 1;                // start of each cell in the pruned grid
    int* new_cand = ism;              ism += d.C * d.gmin;           // insertion rank -> candidate index
    int* flags = ism;                 ism += 4;                      // [0] has_new among published

    const int s = blockIdx.x;
    if (fe_idle(d, s)) { if (threadIdx.x == 0) d.out_n[s] = 0; return; }       // nothing published, grid untouched
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int T = d.cur_count[s];
    const int ncand = d.cand_count[s];
    const int* coff = d.cand_off + s * (d.C + 1);
    const int nxt = par ^ 1;

    // A. stereo gate of the candidates matched in round 2 (round 1 was gated by cand_round2_kernel; 0 = never needed)
    for (int i = threadIdx.x; i < ncand; i += NTB) {
        const size_t o = (size_t)s * d.CC + i;
        if (d.cand_inl[o] == 2) d.cand_inl[o] = stereo_gate(d, d.cand_p0[2 * o], d.cand_p0[2 * o + 1], d.cand_init[2 * o], d.cand_init[2 * o + 1],
                                    d.cand_p1[2 * o], d.cand_p1[2 * o + 1], d.cand_st[o], d.cand_back[2 * o], d.cand_back[2 * o + 1]) ? 1 : 0;
    }
    for (int c = threadIdx.x; c < d.C; c += NTB) trk_n[c] = 0;
    if (threadIdx.x == 0) flags[0] = 0;
    __syncthreads();

    // B. per cell: top grid_min inliers by (response desc, raster asc)
    for (int c = wv; c < d.C; c += NWB) {
        const int b = coff[c], e = coff[c + 1];
        uint32_t last = 0xFFFFFFFFu;
        int r = 0;
        for (; r < d.gmin; ++r) {
            unsigned long long best = 0;
            for (int i = b + lane; i < e; i += 64) {
                const size_t o = (size_t)s * d.CC + i;
                uint32_t k = d.cand_key[o];
                if (d.cand_inl[o] && k < last) {
                    unsigned long long v = ((unsigned long long)k << 32) | (uint32_t)i;
                    if (v > best) best = v;
                }
            }
            best = wave_max_u64(best);
            if (best == 0) break;
            if (lane == 0) sel[c * d.gmin + r] = (int)(best & 0xFFFFFFFFu);
            last = (uint32_t)(best >> 32);
        }
        if (lane == 0) sel_n[c] = r;
    }
    // D. tracked features per cell
    for (int i = threadIdx.x; i < T; i += NTB) atomicAdd(&trk_n[d.cur_cell[(size_t)s * d.NT + i]], 1);
    __syncthreads();

    // C. prefixes (C <= 256 cells: serial is fine)
    if (threadIdx.x == 0) {
        int a = 0, b = 0, o = 0;
        for (int c = 0; c < d.C; ++c) {
            sel_pre[c] = a; cell_start[c] = b; out_start[c] = o;
            int nc = trk_n[c] + sel_n[c];
            a += sel_n[c]; b += nc; o += min(nc, d.gmax);
        }
        sel_pre[d.C] = a; cell_start[d.C] = b; out_start[d.C] = o;
    }
    __syncthreads();
    const int n_new = sel_pre[d.C];
    const int M = T + n_new;

    // E. sort keys: cell | (lifetime key when the cell will be pruned) | insertion index
    for (int j = threadIdx.x; j < d.NSORT; j += NTB) keys[j] = ~0ull;
    __syncthreads();
    for (int i = threadIdx.x; i < T; i += NTB) {
        const size_t o = (size_t)s * d.NT + i;
        const int c = d.cur_cell[o];
        const bool pruned = trk_n[c] + sel_n[c] > d.gmax;
        unsigned long long lk = pruned ? (LIFE_MAX - (unsigned long long)d.cur_life[o]) : 0ull;
        keys[i] = ((unsigned long long)c << 48) | (lk << 24) | (unsigned long long)i;
    }
    for (int q = threadIdx.x; q < d.C * d.gmin; q += NTB) {
        const int c = q / d.gmin, r = q - c * d.gmin;
        if (r < sel_n[c]) {
            const int ins = T + sel_pre[c] + r;
            new_cand[ins - T] = sel[q];
            const bool pruned = trk_n[c] + sel_n[c] > d.gmax;
            unsigned long long lk = pruned ? (LIFE_MAX - 1ull) : 0ull;
            keys[ins] = ((unsigned long long)c << 48) | (lk << 24) | (unsigned long long)ins;
        }
    }
    bitonic_sort_u64(keys, d.NSORT);

    // G. pruned grid of this frame, cell-major
    const long long id0 = d.next_id[s];
    for (int j = threadIdx.x; j < M; j += NTB) {
        const unsigned long long k = keys[j];
        const int c = (int)(k >> 48), ins = (int)(k & 0xFFFFFFull);
        const int rank = j - cell_start[c];
        if (rank >= d.gmax) continue;
        const size_t o = (size_t)s * d.MAXF + out_start[c] + rank;
        if (ins < T) {
            const size_t t = (size_t)s * d.NT + ins;
            d.feat_id[nxt][o] = d.cur_id[t];
            d.feat_life[nxt][o] = d.cur_life[t];
            d.feat_p0[nxt][2 * o] = d.cur_p0[2 * t]; d.feat_p0[nxt][2 * o + 1] = d.cur_p0[2 * t + 1];
            d.feat_p1[nxt][2 * o] = d.cur_p1[2 * t]; d.feat_p1[nxt][2 * o + 1] = d.cur_p1[2 * t + 1];
        } else {
            const size_t q = (size_t)s * d.CC + new_cand[ins - T];
            d.feat_id[nxt][o] = id0 + (ins - T);
            d.feat_life[nxt][o] = 1;
            d.feat_p0[nxt][2 * o] = d.cand_p0[2 * q]; d.feat_p0[nxt][2 * o + 1] = d.cand_p0[2 * q + 1];
            d.feat_p1[nxt][2 * o] = d.cand_p1[2 * q]; d.feat_p1[nxt][2 * o + 1] = d.cand_p1[2 * q + 1];
            flags[0] = 1;                                   // benign race: all writers store 1
        }
        d.feat_cell[nxt][o] = c;
    }
    __syncthreads();
    const int n_out = out_start[d.C];
    // feature_publisher.py:97-107 dtype rule: cam0 coordinates stay float64 iff any
    // published cam0 point is a FAST tuple (= a feature created this frame); cam1 is always float32.
    const bool f64 = flags[0] != 0;
    for (int j = threadIdx.x; j < n_out; j += NTB) {
        const size_t o = (size_t)s * d.MAXF + j;
        double u0, v0, u1, v1;
        av_undistort(d.cam0, d.I3, (double)d.feat_p0[nxt][2 * o], (double)d.feat_p0[nxt][2 * o + 1], u0, v0);
        av_undistort(d.cam1, d.I3, (double)d.feat_p1[nxt][2 * o], (double)d.feat_p1[nxt][2 * o + 1], u1, v1);
        if (!f64) { u0 = (double)(float)u0; v0 = (double)(float)v0; }
        d.out_ids[o] = d.feat_id[nxt][o];
        d.out_uv[4 * o] = u0; d.out_uv[4 * o + 1] = v0;
        d.out_uv[4 * o + 2] = (double)(float)u1; d.out_uv[4 * o + 3] = (double)(float)v1;
    }
    // restore the FAST mask to all ones
    uint8_t* m = d.mask + (size_t)s * d.w * d.h;
    for (int q = threadIdx.x; q < T * 49; q += NTB) {
        int f = q / 49, j = q - f * 49;
        const size_t o = (size_t)s * d.NT + f;
        mask_box(d, m, d.cur_p0[2 * o], d.cur_p0[2 * o + 1], j, 1);
    }
    if (threadIdx.x == 0) {
        d.n_feat[nxt][s] = n_out;
        d.out_n[s] = n_out;
        d.next_id[s] = id0 + n_new;
        d.first_frame[s] = 0;
        d.counters[s * NCNT + CNT_NEW] = n_new;
        d.counters[s * NCNT + CNT_PUB] = n_out;
    }
}

// ---- host-side engine ---------------------------------------------------------------------------
struct ImuSample { double t; double w[3]; };

struct StreamHost {
    std::deque<ImuSample> imu;
    std::mutex mu;
    double t_prev = 0.0;
    bool first = true;
};

inline void mat3_mul(const double* A, const double* B, double* o)
{
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) o[i * 3 + j] = (A[i * 3 + 0] * B[0 * 3 + j] + A[i * 3 + 1] * B[1 * 3 + j]) + A[i * 3 + 2] * B[2 * 3 + j];
}

// cv::Rodrigues (vector -> matrix), as imu_processor.py:63-64 uses it
inline void rodrigues(const double* r, double* R)
{
    double theta = sqrt((r[0] * r[0] + r[1] * r[1]) + r[2] * r[2]);
    if (theta < 2.220446049250313e-16) { for (int i = 0; i < 9; ++i) R[i] = (i % 4 == 0) ? 1.0 : 0.0; return; }
    double c = cos(theta), s = sin(theta), c1 = 1. - c, it = 1. / theta;
    double x = r[0] * it, y = r[1] * it, z = r[2] * it;
    double rrt[9] = {x * x, x * y, x * z, x * y, y * y, y * z, x * z, y * z, z * z};
    double rx[9] = {0, -z, y, z, 0, -x, -y, x, 0};
    for (int i = 0; i < 9; ++i) R[i] = (c * ((i % 4 == 0) ? 1.0 : 0.0) + c1 * rrt[i]) + s * rx[i];
}

}  // namespace

struct av_frontend {
    av_frontend_config cfg;
    int device = 0;
    FeDev d;
    av_pyr_layout lay;
    PyrGeom geom;
    LKParams lk;
    uint8_t* pyr = nullptr;            // [S][3][lay.bytes]
    // step_host staging: two slots of {pinned host, device} image pairs, a copy stream and events, so that the H2D copy of
    // frame k+1 runs behind the kernels of frame k (pinned double-buffered H2D of frame pairs)
    struct HostSlot { uint8_t* pin = nullptr; uint8_t* dev = nullptr; hipEvent_t copied = nullptr, consumed = nullptr; bool used = false; };
    HostSlot hs[3]; hipStream_t copy_stream = nullptr; int hs_next = 0;      // three: the slot of frame k is still read (as the previous cam0 image) by step k+1
    // Level 0 of a pyramid slot (0 / 1: cam0 of alternating frames, 2: cam1) is either the padded copy inside the slot
    // (l0_img = nullptr) or the caller's image itself, read in place by LK and FAST (zero copy).  In place is possible when the
    // image outlives the step that gets it: always for av_frontend_step_host (the library's own staging slots), and for
    // av_frontend_step when the engine was created with AV_FE_INPUTS_PERSIST (include/airvision.h).
    const uint8_t* l0_img[3] = {nullptr, nullptr, nullptr}; int64_t l0_stride[3] = {0, 0, 0};
    // av_frontend_prestage: the pyramids of the NEXT step's images are already built (same slots, same launch): that step skips its own launch
    bool pre_on = false, pre_wrote_l0 = true; const uint8_t* pre_img0 = nullptr; const uint8_t* pre_img1 = nullptr; int64_t pre_stride = 0;
    void* zero_region = nullptr; size_t zero_bytes = 0;
    // Shared frame store (av_frontend_frames_reserve / _upload / av_frontend_step_frames): every DISTINCT stereo frame of a sweep is
    // uploaded, pyramided and FAST-scanned once and stays resident; the streams that replay it -- the offset streams of one
    // sequence, run.bat:4-12 / dataset.py:206-214 -- only carry an index into the store.
    struct FrameStore {
        int n_slots = 0;
        uint8_t* img = nullptr;                  // [n_slots][2][w * h]: cam0, cam1 (level 0 is read in place)
        uint8_t* pyr = nullptr;                  // [n_slots][2][lay.bytes]
        uint32_t* tile_kp = nullptr; int* tile_count = nullptr;      // FAST survivor lists of cam0, per slot
        bool l0_in_place = true;                 // false if the pyramid launcher had to write padded level-0 copies (unaligned geometry)
        std::vector<int> prev;                   // host: slot of every stream's previous frame (-1: none yet)
        struct Up { uint8_t* pin = nullptr; int* idx_h = nullptr; int* idx_d = nullptr; size_t cap = 0; hipEvent_t done = nullptr; bool used = false; };
        Up up[4]; int up_next = 0;               // upload staging ring (pinned frames + the slot list of the upload's kernels)
        hipEvent_t uploaded = nullptr; bool any_upload = false;      // copy stream: the newest upload's kernels have finished
        hipEvent_t stepped = nullptr; bool any_step = false;          // step stream: the newest step has finished
        long long frames_uploaded = 0;
    } fs;
    std::vector<void*> allocs;
    double* dH = nullptr;
    double* hH[8] = {nullptr}; hipEvent_t hH_ev[8]; int hH_slot = 0;
    int parity = 0;                    // index of the "prev" grid buffer; cam0 prev pyramid slot
    std::vector<StreamHost> streams;
    bool any_first = true;
    // feature read-back: two pinned staging slots (n | ids | uv | counters) so that the copy of frame k can be in flight
    // behind the kernels of frame k+1 (av_frontend_read_features_begin / _end)
    struct ReadSlot { int* n = nullptr; long long* ids = nullptr; double* uv = nullptr; int* cnt = nullptr; hipEvent_t ev = nullptr; bool pending = false; };
    ReadSlot rd[2];
    // optional HIP-event timing per kernel class (bench.py's roofline leg)
    bool timing = false;
    std::vector<hipEvent_t> ev; std::vector<int> ev_cls; size_t ev_used = 0;

    explicit av_frontend(int S) : streams(S) {}
};

namespace {

template <typename T>
int dev_alloc(av_frontend* fe, T** p, size_t count, int fill = 0)
{
    void* q = nullptr;
    AV_HIP(hipMalloc(&q, count * sizeof(T) + 256));
    AV_HIP(hipMemset(q, fill, count * sizeof(T) + 256));
    fe->allocs.push_back(q);
    *p = reinterpret_cast<T*>(q);
    return AV_OK;
}

int integrate_imu(av_frontend* fe, int s, double t_curr, double* H)
{
    // imu_processor.py:28-67 + feature_tracker.py:166-171
    StreamHost& sh = fe->streams[s];
    const av_frontend_config& c = fe->cfg;
    double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    {
        std::lock_guard<std::mutex> g(sh.mu);
        int ib = -1, ie = -1;
        const int n = (int)sh.imu.size();
        for (int i = 0; i < n; ++i) if (sh.imu[i].t >= sh.t_prev - 0.01) { ib = i; break; }
        for (int i = 0; i < n; ++i) if (sh.imu[i].t >= t_curr - 0.004) { ie = i; break; }
        if (ib >= 0 && ie >= 0) {
            double m[3] = {0, 0, 0};
            for (int i = ib; i < ie; ++i) { m[0] += sh.imu[i].w[0]; m[1] += sh.imu[i].w[1]; m[2] += sh.imu[i].w[2]; }
            int cnt = ie - ib;
            if (cnt > 0) { m[0] /= cnt; m[1] /= cnt; m[2] /= cnt; }
            // cam0_mean = R_cam0_imu.T @ mean
            const double* Rc = c.R_cam0_imu;
            double cm[3];
            for (int i = 0; i < 3; ++i) cm[i] = (Rc[0 * 3 + i] * m[0] + Rc[1 * 3 + i] * m[1]) + Rc[2 * 3 + i] * m[2];
            double dt = t_curr - sh.t_prev;
            double rv[3] = {cm[0] * dt, cm[1] * dt, cm[2] * dt};
            double Rr[9];
            rodrigues(rv, Rr);
            for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R[i * 3 + j] = Rr[j * 3 + i];      // .T
            sh.imu.erase(sh.imu.begin(), sh.imu.begin() + ie);
        }
    }
    const double fx = c.cam0_intrinsics[0], fy = c.cam0_intrinsics[1], cx = c.cam0_intrinsics[2], cy = c.cam0_intrinsics[3];
    double K[9] = {fx, 0, cx, 0, fy, cy, 0, 0, 1};
    double Ki[9] = {1. / fx, 0, -cx / fx, 0, 1. / fy, -cy / fy, 0, 0, 1};
    double KR[9];
    mat3_mul(K, R, KR);
    mat3_mul(KR, Ki, H);
    return AV_OK;
}

// event pair around a launch group; classes: 0 pyramid, 1 LK, 2 FAST, 3 glue
struct Span {
    av_frontend* fe; hipStream_t st; bool on;
    Span(av_frontend* f, int cls, hipStream_t s) : fe(f), st(s), on(false)
    {
        if (fe->timing && fe->ev_used + 2 <= fe->ev.size()) {
            on = true;
            fe->ev_cls[fe->ev_used / 2] = cls;
            (void)hipEventRecord(fe->ev[fe->ev_used], st);
        }
    }
    ~Span()
    {
        if (on) { (void)hipEventRecord(fe->ev[fe->ev_used + 1], st); fe->ev_used += 2; }
    }
};

// slots != nullptr: the step reads the shared frame store (stream s: entry slots[s], < 0 = no frame in this step); img0 / img1 unused.
int step_impl(av_frontend* fe, const uint8_t* img0, const uint8_t* img1, int64_t img_stride, const double* ts, hipStream_t st, bool inputs_persist,
              const int32_t* slots = nullptr)
{
    FeDev d = fe->d;                         // by value: the kernels take it by value, the frame-store step patches a few fields
    const int S = d.S;
    const bool frames = slots != nullptr;
    AV_HIP(hipSetDevice(fe->device));
    // host: IMU rotation prediction -> homographies
    const int slot = fe->hH_slot;
    fe->hH_slot = (slot + 1) & 7;
    AV_HIP(hipEventSynchronize(fe->hH_ev[slot]));
    double* hH = fe->hH[slot];
    bool any_first = false;
    int* hmap = reinterpret_cast<int*>(hH + (size_t)9 * S);      // [2][S] behind the homographies: this frame's / the previous frame's store entry
    for (int s = 0; s < S; ++s) {
        StreamHost& sh = fe->streams[s];
        if (frames) {
            hmap[s] = slots[s] < 0 ? -1 : slots[s]; hmap[S + s] = fe->fs.prev[s];
            if (slots[s] < 0) {                  // no frame for this stream in this step: its IMU buffer, t_prev and grid stay as they are
                for (int i = 0; i < 9; ++i) hH[s * 9 + i] = (i % 4 == 0) ? 1.0 : 0.0;
                continue;
            }
            fe->fs.prev[s] = slots[s];
        }
        if (sh.first) {
            any_first = true;
            for (int i = 0; i < 9; ++i) hH[s * 9 + i] = (i % 4 == 0) ? 1.0 : 0.0;
        } else {
            integrate_imu(fe, s, ts[s], hH + s * 9);
        }
        sh.t_prev = ts[s];
        sh.first = false;
    }
    AV_HIP(hipMemcpyAsync(fe->dH, hH, sizeof(double) * 9 * S + (frames ? sizeof(int) * 2 * S : 0), hipMemcpyHostToDevice, st));
    AV_HIP(hipEventRecord(fe->hH_ev[slot], st));
    AV_HIP(hipMemsetAsync(fe->zero_region, 0, fe->zero_bytes, st));
    const int* map_cur = nullptr; const int* map_prev = nullptr;
    if (frames) {
        map_cur = reinterpret_cast<const int*>(fe->dH + (size_t)9 * S); map_prev = map_cur + S;
        d.slot_cur = map_cur; d.tile_kp = fe->fs.tile_kp; d.tile_count = fe->fs.tile_count;
        if (fe->fs.any_upload) AV_HIP(hipStreamWaitEvent(st, fe->fs.uploaded, 0));      // the frames this step reads are in the store, pyramids and FAST lists with them
    }

    const int par = fe->parity;              // prev grid buffer / prev cam0 pyramid slot
    const int cur0 = par ^ 1;                // curr cam0 pyramid slot (0/1), cam1 pyramid is slot 2
    int64_t sstride = 3 * fe->lay.bytes; const int64_t slotb = fe->lay.bytes;
    int rc;
    static const bool zc_off = [] { const char* e = getenv("AV_FE_ZERO_COPY"); return e && atoi(e) == 0; }();      // A/B switch
    const uint8_t *I_prev0, *I_cur0, *I_cur1, *P_prev0, *P_cur0, *P_cur1; int64_t st_prev0;
    if (frames) {
        // pyramids and level-0 images of both cameras lie in the store, built when the frame was uploaded: entry e of camera c at
        // pyr + (2 e + c) * bytes / img + (2 e + c) * w * h; the LK launches index it through map_prev / map_cur
        const size_t hw = (size_t)d.w * d.h;
        sstride = 2 * slotb; img_stride = (int64_t)(2 * hw); st_prev0 = img_stride;
        P_prev0 = P_cur0 = fe->fs.pyr; P_cur1 = fe->fs.pyr + slotb;
        I_prev0 = I_cur0 = fe->fs.l0_in_place ? fe->fs.img : nullptr; I_cur1 = fe->fs.l0_in_place ? fe->fs.img + hw : nullptr;
    } else {
        bool wrote_l0 = true;
        if (fe->pre_on && fe->pre_img0 == img0 && fe->pre_img1 == img1 && fe->pre_stride == img_stride) wrote_l0 = fe->pre_wrote_l0;      // built by av_frontend_prestage
        else { Span sp(fe, 0, st);
          if ((rc = av_launch_pyramid(img0, img1, img_stride, S, 2, fe->geom, fe->pyr, sstride, slotb, cur0, 2, st, !(inputs_persist && !zc_off), &wrote_l0))) return rc; }
        fe->pre_on = false;
        fe->l0_img[cur0] = wrote_l0 ? nullptr : img0; fe->l0_img[2] = wrote_l0 ? nullptr : img1;
        fe->l0_stride[cur0] = fe->l0_stride[2] = img_stride;
        I_prev0 = fe->l0_img[par]; st_prev0 = fe->l0_stride[par];      // (first frame: nothing is tracked from it)
        I_cur0 = fe->l0_img[cur0]; I_cur1 = fe->l0_img[2];
        P_prev0 = fe->pyr + par * slotb;
        P_cur0 = fe->pyr + cur0 * slotb;
        P_cur1 = fe->pyr + 2 * slotb;
    }

    // FAST reads level 0 of the cam0 pyramid built above (same pixels as the input image, with a 16-pixel frame: every
    // tile but the right-most column copies whole dwords without clamping) or the caller's image in place
    auto launch_fast = [&](hipStream_t fs) -> int {
        Span sp(fe, 2, fs);
        const uint8_t* fast_img = P_cur0 + fe->geom.off[0] + (size_t)AV_PYR_BORDER * fe->geom.pitch[0] + AV_PYR_BORDER;
        if (I_cur0) return av_launch_fast(I_cur0, img_stride, d.w, 0, nullptr, 0, S, d.w, d.h, fe->cfg.fast_threshold,
                                          nullptr, nullptr, 0, d.tile_kp, d.tile_count, d.counters + CNT_OVF, NCNT, fs);
        return av_launch_fast(fast_img, sstride, fe->geom.pitch[0], AV_PYR_BORDER, nullptr, 0, S, d.w, d.h, fe->cfg.fast_threshold,
                              nullptr, nullptr, 0, d.tile_kp, d.tile_count, d.counters + CNT_OVF, NCNT, fs);
    };
    // per-stream glue kernels: 256-thread workgroups.  (AV_FE_GLUE_WG=64, one wavefront per stream -- bit-exact, the kernels take any
    // workgroup size -- measured in round 5: glue 0.78 against 0.66 ms alone, 0.94 against 1.07 beside the filter, and the detector next
    // to them 2.46 against 2.31: front-end alone 216.9 against 220.3 k, complete path 173.3 against 174.3 k.  profiles/r05/README.md)
    static const int gwg = [] { const char* e = getenv("AV_FE_GLUE_WG"); return (e && atoi(e) == 64) ? 64 : 256; }();
    { Span sp(fe, 3, st);
      hipLaunchKernelGGL(track_prepare_kernel, dim3((d.NT + 255) / 256, S), dim3(256), 0, st, d, par);
      AV_LAUNCH_CHECK(); }
    { Span sp(fe, 1, st);
      if ((rc = av_launch_lk(P_prev0, P_cur0, sstride, S, fe->geom, d.trk_prev, d.trk_next, d.trk_status, d.trk_count, d.NT, d.NT, fe->lk, st, nullptr, I_prev0, st_prev0, I_cur0, img_stride, map_prev, map_cur))) return rc; }
    { Span sp(fe, 3, st);
      hipLaunchKernelGGL(track_gate_kernel, dim3(S), dim3(gwg), 0, st, d);
      AV_LAUNCH_CHECK(); }
    { Span sp(fe, 1, st);
      if ((rc = av_launch_lk(P_cur0, P_cur1, sstride, S, fe->geom, d.sv_p0, d.sv_p1, d.sv_st, d.sv_count, d.NT, d.NT, fe->lk, st, nullptr, I_cur0, img_stride, I_cur1, img_stride, map_cur, map_cur))) return rc; }
    { Span sp(fe, 1, st);
      if ((rc = av_launch_lk(P_cur1, P_cur0, sstride, S, fe->geom, d.sv_p1, d.sv_back, d.sv_st2, d.sv_count, d.NT, d.NT, fe->lk, st, nullptr, I_cur1, img_stride, I_cur0, img_stride, map_cur, map_cur))) return rc; }
    { Span sp(fe, 3, st);
      hipLaunchKernelGGL(rebin_kernel, dim3(S), dim3(gwg), 0, st, d, par);
      AV_LAUNCH_CHECK(); }
    // (FAST on a second HIP stream beside the temporal / stereo LK launches -- it reads only the new cam0 image and is first needed by
    //  select_kernel -- was measured in round 5: front-end alone 196.6 k against 205.1 k frames/s, complete path 157.7 against 157.3 k:
    //  the LK launches slow down by more than the detector's time; profiles/r05/README.md)
    if (!frames && (rc = launch_fast(st))) return rc;                           // (frame store: FAST ran when the frame was uploaded)
    { Span sp(fe, 3, st);
      hipLaunchKernelGGL(select_kernel, dim3(S), dim3(gwg), sizeof(int) * (3 * d.C + 1 + d.n_tiles + 1), st, d);
      AV_LAUNCH_CHECK(); }
    const int r1_launch = any_first ? d.CC : d.C * (d.gmax < CAND_R1 ? d.gmax : CAND_R1);
    { Span sp(fe, 1, st);
      if ((rc = av_launch_lk(P_cur0, P_cur1, sstride, S, fe->geom, d.cand_p0, d.cand_p1, d.cand_st, d.r1_count, d.CC, r1_launch, fe->lk, st, d.r1_list, I_cur0, img_stride, I_cur1, img_stride, map_cur, map_cur))) return rc; }
    { Span sp(fe, 1, st);
      if ((rc = av_launch_lk(P_cur1, P_cur0, sstride, S, fe->geom, d.cand_p1, d.cand_back, d.cand_st2, d.r1_count, d.CC, r1_launch, fe->lk, st, d.r1_list, I_cur1, img_stride, I_cur0, img_stride, map_cur, map_cur))) return rc; }
    { Span sp(fe, 3, st);
      hipLaunchKernelGGL(cand_round2_kernel, dim3(S), dim3(gwg), 0, st, d);
      AV_LAUNCH_CHECK(); }
    if (d.gmax > CAND_R1) {                       // round 2: the rest of the cells that are still short of inliers (usually none)
        const int r2_launch = d.C * (d.gmax - CAND_R1);
        { Span sp(fe, 1, st);
          if ((rc = av_launch_lk(P_cur0, P_cur1, sstride, S, fe->geom, d.cand_p0, d.cand_p1, d.cand_st, d.r2_count, d.CC, r2_launch, fe->lk, st, d.r2_list, I_cur0, img_stride, I_cur1, img_stride, map_cur, map_cur))) return rc; }
        { Span sp(fe, 1, st);
          if ((rc = av_launch_lk(P_cur1, P_cur0, sstride, S, fe->geom, d.cand_p1, d.cand_back, d.cand_st2, d.r2_count, d.CC, r2_launch, fe->lk, st, d.r2_list, I_cur1, img_stride, I_cur0, img_stride, map_cur, map_cur))) return rc; }
    }
    size_t fin_lds = sizeof(unsigned long long) * d.NSORT + sizeof(int) * (2 * d.C * d.gmin + 2 * d.C + 3 * (d.C + 1) + 4);
    { Span sp(fe, 3, st);
      hipLaunchKernelGGL(finalize_kernel, dim3(S), dim3(gwg), fin_lds, st, d, par);
      AV_LAUNCH_CHECK(); }
    fe->parity = par ^ 1;
    if (frames) { AV_HIP(hipEventRecord(fe->fs.stepped, st)); fe->fs.any_step = true; }
    return AV_OK;
}

}  // namespace

AV_EXPORT int av_frontend_create(const av_frontend_config* cfg, int n_streams, int device, av_frontend** out)
{
    if (!cfg || !out || n_streams <= 0) { av_set_error("av_frontend_create: bad arguments"); return AV_E_INVALID; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { av_set_error("av_frontend_create: no HIP device visible"); return AV_E_NODEVICE; }
    if (device < 0 || device >= ndev) { av_set_error("av_frontend_create: device %d out of range (%d visible)", device, ndev); return AV_E_INVALID; }
    const int C = cfg->grid_row * cfg->grid_col;
    if (cfg->grid_row <= 0 || cfg->grid_col <= 0 || C > 256 || cfg->grid_min_feature_num <= 0 || cfg->grid_max_feature_num <= 0 ||
        cfg->lk_levels < 1 || cfg->lk_levels > AV_MAX_LEVELS || cfg->max_corners <= 0) {
        av_set_error("av_frontend_create: unsupported configuration (grid %dx%d, levels %d)", cfg->grid_row, cfg->grid_col, cfg->lk_levels);
        return AV_E_INVALID;
    }
    AV_HIP(hipSetDevice(device));
    av_frontend* fe = new (std::nothrow) av_frontend(n_streams);
    if (!fe) { av_set_error("out of host memory"); return AV_E_INVALID; }
    fe->cfg = *cfg; fe->device = device;
    int rc = av_pyramid_layout(cfg->width, cfg->height, cfg->lk_levels, &fe->lay);
    if (rc) { delete fe; return rc; }
    fe->geom = av_make_geom(fe->lay);
    fe->lk.win = cfg->lk_win;
    fe->lk.max_iter = cfg->lk_max_iter < 0 ? 0 : (cfg->lk_max_iter > 100 ? 100 : cfg->lk_max_iter);
    double e = cfg->lk_eps < 0 ? 0. : (cfg->lk_eps > 10. ? 10. : cfg->lk_eps);
    fe->lk.eps2 = e * e;
    fe->lk.min_eig = cfg->lk_min_eig;
    if (fe->lk.win < 3 || fe->lk.win > 31) { av_set_error("av_frontend_create: lk_win %d outside 3 .. 31", fe->lk.win); delete fe; return AV_E_INVALID; }      // 15: the 16-lane kernel; others: the general one

    FeDev& d = fe->d;
    memset(&d, 0, sizeof(d));
    const int S = n_streams, w = cfg->width, h = cfg->height;
    d.S = S; d.w = w; d.h = h; d.C = C; d.grid_row = cfg->grid_row; d.grid_col = cfg->grid_col;
    d.gh = (h + cfg->grid_row - 1) / cfg->grid_row;          // int(np.ceil(h / grid_row)), feature_tracker.py:70-71
    d.gw = (w + cfg->grid_col - 1) / cfg->grid_col;
    d.gmin = cfg->grid_min_feature_num; d.gmax = cfg->grid_max_feature_num;
    d.MAXF = C * d.gmax; d.NT = d.MAXF;
    d.CC = cfg->max_corners > C * d.gmax ? cfg->max_corners : C * d.gmax;
    d.cell_cap = (d.gh * d.gw) / 4 + 64;
    if (d.cell_cap > d.CC) d.cell_cap = d.CC;
    int ns = 2; while (ns < C * (d.gmax + d.gmin)) ns <<= 1;
    d.NSORT = ns;
    if (ns > 4096) { av_set_error("av_frontend_create: grid_num*(grid_max+grid_min) = %d exceeds 4096", C * (d.gmax + d.gmin)); delete fe; return AV_E_INVALID; }
    d.cam0 = CamModel{cfg->cam0_intrinsics[0], cfg->cam0_intrinsics[1], cfg->cam0_intrinsics[2], cfg->cam0_intrinsics[3],
                      cfg->cam0_distortion[0], cfg->cam0_distortion[1], cfg->cam0_distortion[2], cfg->cam0_distortion[3], cfg->cam0_distortion_model};
    d.cam1 = CamModel{cfg->cam1_intrinsics[0], cfg->cam1_intrinsics[1], cfg->cam1_intrinsics[2], cfg->cam1_intrinsics[3],
                      cfg->cam1_distortion[0], cfg->cam1_distortion[1], cfg->cam1_distortion[2], cfg->cam1_distortion[3], cfg->cam1_distortion_model};
    for (int i = 0; i < 9; ++i) { d.R0to1[i] = cfg->R0to1[i]; d.E[i] = cfg->E[i]; d.I3[i] = (i % 4 == 0) ? 1.0 : 0.0; }
    d.epi_thr = cfg->stereo_threshold * cfg->norm_unit;

#define A(ptr, n) if ((rc = dev_alloc(fe, &(ptr), (size_t)(n)))) { av_frontend_destroy(fe); return rc; }
    for (int b = 0; b < 2; ++b) {
        A(d.feat_id[b], S * d.MAXF) A(d.feat_life[b], S * d.MAXF) A(d.feat_p0[b], 2 * S * d.MAXF) A(d.feat_p1[b], 2 * S * d.MAXF)
        A(d.feat_cell[b], S * d.MAXF) A(d.n_feat[b], S)
    }
    A(d.next_id, S)
    if ((rc = dev_alloc(fe, &d.first_frame, (size_t)S, 1))) { av_frontend_destroy(fe); return rc; }   // bytes 0x01 -> non-zero ints
    A(fe->dH, 9 * S + S)                       // + [2][S] ints: the frame-store maps of the step (step_impl)
    d.Hmat = fe->dH;
    A(d.trk_prev, 2 * S * d.NT) A(d.trk_next, 2 * S * d.NT) A(d.trk_status, S * d.NT)
    A(d.sv_src, S * d.NT) A(d.sv_p0, 2 * S * d.NT) A(d.sv_init, 2 * S * d.NT) A(d.sv_p1, 2 * S * d.NT) A(d.sv_st, S * d.NT)
    A(d.sv_back, 2 * S * d.NT) A(d.sv_st2, S * d.NT)
    A(d.cur_id, S * d.NT) A(d.cur_life, S * d.NT) A(d.cur_p0, 2 * S * d.NT) A(d.cur_p1, 2 * S * d.NT) A(d.cur_cell, S * d.NT)
    A(d.cell_kp, (size_t)S * C * d.cell_cap)
    av_fast_tiles(w, h, &d.n_tiles, &d.tile_cap);
    A(d.tile_kp, (size_t)S * d.n_tiles * d.tile_cap) A(d.tile_count, (size_t)S * d.n_tiles)
    if ((rc = dev_alloc(fe, &d.mask, (size_t)S * w * h, 1))) { av_frontend_destroy(fe); return rc; }
    A(d.cand_key, (size_t)S * d.CC) A(d.cand_p0, 2 * (size_t)S * d.CC) A(d.cand_init, 2 * (size_t)S * d.CC) A(d.cand_p1, 2 * (size_t)S * d.CC)
    A(d.cand_st, (size_t)S * d.CC) A(d.cand_back, 2 * (size_t)S * d.CC) A(d.cand_st2, (size_t)S * d.CC) A(d.cand_inl, (size_t)S * d.CC)
    A(d.cand_off, S * (C + 1))
    A(d.r1_list, (size_t)S * d.CC) A(d.r2_list, (size_t)S * d.CC) A(d.r1_count, S) A(d.r2_count, S)
    A(d.out_ids, S * d.MAXF) A(d.out_uv, 4 * S * d.MAXF) A(d.out_n, S)
    // per-step zeroed counters in one region: trk_count, sv_count, cur_count, cand_count, cell_count, counters
    {
        size_t n_int = (size_t)S * 4 + (size_t)S * C + (size_t)S * NCNT;
        int* z = nullptr;
        A(z, n_int)
        fe->zero_region = z; fe->zero_bytes = n_int * sizeof(int);
        d.trk_count = z; d.sv_count = z + S; d.cur_count = z + 2 * S; d.cand_count = z + 3 * S;
        d.cell_count = z + 4 * S; d.counters = z + 4 * S + (size_t)S * C;
    }
    A(fe->pyr, (size_t)S * 3 * fe->lay.bytes)
#undef A
    for (int i = 0; i < 8; ++i) {
        if (hipHostMalloc((void**)&fe->hH[i], sizeof(double) * (9 * S + S), hipHostMallocDefault) != hipSuccess ||
            hipEventCreateWithFlags(&fe->hH_ev[i], hipEventDisableTiming) != hipSuccess) {
            av_set_error("av_frontend_create: pinned staging allocation failed");
            av_frontend_destroy(fe);
            return AV_E_HIP;
        }
    }
    for (int i = 0; i < 2; ++i) {
        av_frontend::ReadSlot& r = fe->rd[i];
        if (hipHostMalloc((void**)&r.n, sizeof(int) * S, hipHostMallocDefault) != hipSuccess ||
            hipHostMalloc((void**)&r.ids, sizeof(long long) * (size_t)S * d.MAXF, hipHostMallocDefault) != hipSuccess ||
            hipHostMalloc((void**)&r.uv, sizeof(double) * 4 * (size_t)S * d.MAXF, hipHostMallocDefault) != hipSuccess ||
            hipHostMalloc((void**)&r.cnt, sizeof(int) * (size_t)S * NCNT, hipHostMallocDefault) != hipSuccess ||
            hipEventCreateWithFlags(&r.ev, hipEventDisableTiming) != hipSuccess) {
            av_set_error("av_frontend_create: pinned read-back staging allocation failed");
            av_frontend_destroy(fe);
            return AV_E_HIP;
        }
    }
    *out = fe;
    return AV_OK;
}

AV_EXPORT void av_frontend_destroy(av_frontend* fe)
{
    if (!fe) return;
    (void)hipSetDevice(fe->device);
    (void)hipDeviceSynchronize();
    for (void* p : fe->allocs) (void)hipFree(p);
    for (hipEvent_t e : fe->ev) (void)hipEventDestroy(e);
    for (int i = 0; i < 3; ++i) {
        av_frontend::HostSlot& h = fe->hs[i];
        if (h.pin) (void)hipHostFree(h.pin);
        if (h.dev) (void)hipFree(h.dev);
        if (h.copied) (void)hipEventDestroy(h.copied);
        if (h.consumed) (void)hipEventDestroy(h.consumed);
    }
    for (av_frontend::FrameStore::Up& u : fe->fs.up) {
        if (u.pin) (void)hipHostFree(u.pin);
        if (u.idx_h) (void)hipHostFree(u.idx_h);
        if (u.idx_d) (void)hipFree(u.idx_d);
        if (u.done) (void)hipEventDestroy(u.done);
    }
    if (fe->fs.uploaded) (void)hipEventDestroy(fe->fs.uploaded);
    if (fe->fs.stepped) (void)hipEventDestroy(fe->fs.stepped);
    if (fe->copy_stream) (void)hipStreamDestroy(fe->copy_stream);
    for (int i = 0; i < 8; ++i) {
        if (fe->hH[i]) { (void)hipHostFree(fe->hH[i]); (void)hipEventDestroy(fe->hH_ev[i]); }
    }
    for (int i = 0; i < 2; ++i) {
        av_frontend::ReadSlot& r = fe->rd[i];
        if (r.n) (void)hipHostFree(r.n);
        if (r.ids) (void)hipHostFree(r.ids);
        if (r.uv) (void)hipHostFree(r.uv);
        if (r.cnt) (void)hipHostFree(r.cnt);
        if (r.ev) (void)hipEventDestroy(r.ev);
    }
    delete fe;
}

AV_EXPORT int av_frontend_push_imu(av_frontend* fe, int stream, double timestamp, const double gyro[3])
{
    if (!fe || stream < 0 || stream >= fe->d.S || !gyro) { av_set_error("av_frontend_push_imu: bad arguments"); return AV_E_INVALID; }
    StreamHost& sh = fe->streams[stream];
    std::lock_guard<std::mutex> g(sh.mu);
    sh.imu.push_back(ImuSample{timestamp, {gyro[0], gyro[1], gyro[2]}});
    return AV_OK;
}

AV_EXPORT int av_frontend_push_imu_batch(av_frontend* fe, const int32_t* stream_idx, const double* timestamps,
                                         const double* gyro, int n)
{
    if (!fe || !stream_idx || !timestamps || !gyro || n < 0) { av_set_error("av_frontend_push_imu_batch: bad arguments"); return AV_E_INVALID; }
    for (int i = 0; i < n; ++i) {
        int rc = av_frontend_push_imu(fe, stream_idx[i], timestamps[i], gyro + 3 * (size_t)i);
        if (rc) return rc;
    }
    return AV_OK;
}

// The pyramids of the images the NEXT av_frontend_step will get, enqueued now (behind the step just enqueued): that step then starts
// with its tracking launch.  Same kernels, same slots, same results -- only their place in the stream moves.  A caller that hands a
// step's feature message to the batched filter AFTER this call (av_msckf_batch_submit_dev copies it on this stream) starts the
// filter's chain behind the pyramid kernels instead of beside them: the filter's first kernel (39 KB of LDS, 4 x 128 registers per
// stream) and the pyramid kernel (all 160 KB of LDS at six workgroups per CU) otherwise halve each other (profiles/r05/README.md).
// Needs AV_FE_INPUTS_PERSIST (the images are read again by the step itself); a step that comes with other images builds its own.
AV_EXPORT int av_frontend_prestage(av_frontend* fe, const uint8_t* img0_dev, const uint8_t* img1_dev, int64_t img_stride, void* stream)
{
    if (!fe || !img0_dev || !img1_dev || img_stride < (int64_t)fe->d.w * fe->d.h) { av_set_error("av_frontend_prestage: bad arguments"); return AV_E_INVALID; }
    if (!(fe->cfg.flags & AV_FE_INPUTS_PERSIST)) { av_set_error("av_frontend_prestage: the engine was created without AV_FE_INPUTS_PERSIST"); return AV_E_INVALID; }
    hipStream_t st = (hipStream_t)stream;
    AV_HIP(hipSetDevice(fe->device));
    static const bool zc_off = [] { const char* e = getenv("AV_FE_ZERO_COPY"); return e && atoi(e) == 0; }();
    const int cur0 = fe->parity ^ 1;         // the slots the next step will call cur0 / 2
    int rc;
    bool wrote_l0 = true;
    { Span sp(fe, 0, st);
      if ((rc = av_launch_pyramid(img0_dev, img1_dev, img_stride, fe->d.S, 2, fe->geom, fe->pyr, 3 * fe->lay.bytes, fe->lay.bytes, cur0, 2, st, zc_off, &wrote_l0))) return rc; }
    // (The detector's pass over the new cam0 image -- it reads nothing but the image -- enqueued here as well ran at its exclusive speed,
    //  1.55 ms against 2.3 beside the filter's back end, and the LK launches took what it gave back: 173.4-173.9 against 174.1-175.2 k
    //  frames/s.  profiles/r05/README.md)
    fe->pre_on = true; fe->pre_wrote_l0 = wrote_l0; fe->pre_img0 = img0_dev; fe->pre_img1 = img1_dev; fe->pre_stride = img_stride;
    return AV_OK;
}

AV_EXPORT int av_frontend_step(av_frontend* fe, const uint8_t* img0_dev, const uint8_t* img1_dev, int64_t img_stride,
                               const double* timestamps, void* stream)
{
    if (!fe || !img0_dev || !img1_dev || !timestamps || img_stride < (int64_t)fe->d.w * fe->d.h) {
        av_set_error("av_frontend_step: bad arguments");
        return AV_E_INVALID;
    }
    return step_impl(fe, img0_dev, img1_dev, img_stride, timestamps, (hipStream_t)stream, (fe->cfg.flags & AV_FE_INPUTS_PERSIST) != 0);
}

AV_EXPORT int av_frontend_step_host(av_frontend* fe, const uint8_t* img0_host, const uint8_t* img1_host, int64_t img_stride,
                                    const double* timestamps, void* stream)
{
    if (!fe || !img0_host || !img1_host || !timestamps || img_stride < (int64_t)fe->d.w * fe->d.h) {
        av_set_error("av_frontend_step_host: bad arguments");
        return AV_E_INVALID;
    }
    hipStream_t st = (hipStream_t)stream;
    const size_t img_bytes = (size_t)fe->d.w * fe->d.h;
    const int S = fe->d.S;
    AV_HIP(hipSetDevice(fe->device));
    av_frontend::HostSlot& h = fe->hs[fe->hs_next];
    fe->hs_next = (fe->hs_next + 1) % 3;
    if (!fe->copy_stream) AV_HIP(hipStreamCreateWithFlags(&fe->copy_stream, hipStreamNonBlocking));
    if (!h.pin) {
        AV_HIP(hipHostMalloc((void**)&h.pin, 2 * img_bytes * S, hipHostMallocDefault));
        AV_HIP(hipMalloc((void**)&h.dev, 2 * img_bytes * S));
        AV_HIP(hipEventCreateWithFlags(&h.copied, hipEventDisableTiming));
        AV_HIP(hipEventCreateWithFlags(&h.consumed, hipEventDisableTiming));
    }
    // the slot was filled three calls ago and last read two calls ago (as that step's previous cam0 image): wait for the
    // step recorded after it -- every earlier step on the stream has then retired too
    if (h.used) AV_HIP(hipEventSynchronize(h.consumed));
    // caller memory -> pinned slot (the caller's buffers are free again when this returns), all cores
#pragma omp parallel for schedule(static) num_threads(S >= 16 ? 8 : 1)
    for (int i = 0; i < 2 * S; ++i) {
        const int s2 = i >> 1, cam = i & 1;
        memcpy(h.pin + ((size_t)cam * S + s2) * img_bytes, (cam ? img1_host : img0_host) + (size_t)s2 * img_stride, img_bytes);
    }
    AV_HIP(hipMemcpyAsync(h.dev, h.pin, 2 * img_bytes * S, hipMemcpyHostToDevice, fe->copy_stream));
    AV_HIP(hipEventRecord(h.copied, fe->copy_stream));
    AV_HIP(hipStreamWaitEvent(st, h.copied, 0));
    const int rc = step_impl(fe, h.dev, h.dev + img_bytes * S, (int64_t)img_bytes, timestamps, st, true);      // staging slots outlive the next step
    if (rc) return rc;
    // `consumed` of the slot filled ONE call ago is recorded now: this step was the last reader of its cam0 image
    av_frontend::HostSlot& hp = fe->hs[(fe->hs_next + 1) % 3];
    if (hp.pin) { AV_HIP(hipEventRecord(hp.consumed, st)); hp.used = true; }
    AV_HIP(hipEventRecord(h.consumed, st));
    h.used = true;
    return AV_OK;
}

// ---- shared frame store -------------------------------------------------------------------------------------------------
AV_EXPORT int av_frontend_frames_reserve(av_frontend* fe, int n_slots)
{
    if (!fe || n_slots <= 0) { av_set_error("av_frontend_frames_reserve: bad arguments"); return AV_E_INVALID; }
    av_frontend::FrameStore& fs = fe->fs;
    if (fs.n_slots) {
        if (n_slots <= fs.n_slots) return AV_OK;
        av_set_error("av_frontend_frames_reserve: the store holds %d entries and cannot grow (asked for %d)", fs.n_slots, n_slots);
        return AV_E_INVALID;
    }
    AV_HIP(hipSetDevice(fe->device));
    const FeDev& d = fe->d;
    const size_t hw = (size_t)d.w * d.h;
    int rc;
    if ((rc = dev_alloc(fe, &fs.img, (size_t)n_slots * 2 * hw)) || (rc = dev_alloc(fe, &fs.pyr, (size_t)n_slots * 2 * fe->lay.bytes)) ||
        (rc = dev_alloc(fe, &fs.tile_kp, (size_t)n_slots * d.n_tiles * d.tile_cap)) || (rc = dev_alloc(fe, &fs.tile_count, (size_t)n_slots * d.n_tiles))) return rc;
    AV_HIP(hipEventCreateWithFlags(&fs.uploaded, hipEventDisableTiming));
    AV_HIP(hipEventCreateWithFlags(&fs.stepped, hipEventDisableTiming));
    if (!fe->copy_stream) AV_HIP(hipStreamCreateWithFlags(&fe->copy_stream, hipStreamNonBlocking));
    fs.prev.assign((size_t)d.S, -1);
    fs.n_slots = n_slots;
    return AV_OK;
}

AV_EXPORT int av_frontend_frames_upload(av_frontend* fe, const int32_t* slots, int n, const uint8_t* img0_host, const uint8_t* img1_host,
                                        int64_t img_stride, void* stream)
{
    (void)stream;
    if (!fe || n < 0 || (n > 0 && (!slots || !img0_host || !img1_host)) || img_stride < (int64_t)fe->d.w * fe->d.h) {
        av_set_error("av_frontend_frames_upload: bad arguments");
        return AV_E_INVALID;
    }
    av_frontend::FrameStore& fs = fe->fs;
    if (!fs.n_slots) { av_set_error("av_frontend_frames_upload: no frame store (av_frontend_frames_reserve first)"); return AV_E_INVALID; }
    if (n == 0) return AV_OK;
    for (int i = 0; i < n; ++i)
        if (slots[i] < 0 || slots[i] >= fs.n_slots) { av_set_error("av_frontend_frames_upload: entry %d outside the store (%d entries)", slots[i], fs.n_slots); return AV_E_INVALID; }
    AV_HIP(hipSetDevice(fe->device));
    const FeDev& d = fe->d;
    const size_t hw = (size_t)d.w * d.h;
    av_frontend::FrameStore::Up& u = fs.up[fs.up_next];
    fs.up_next = (fs.up_next + 1) % 4;
    if (u.used) AV_HIP(hipEventSynchronize(u.done));                // the staging area's previous upload has left it
    if ((size_t)n > u.cap) {
        if (u.pin) { (void)hipHostFree(u.pin); (void)hipHostFree(u.idx_h); (void)hipFree(u.idx_d); u.pin = nullptr; u.idx_h = nullptr; u.idx_d = nullptr; }
        const size_t cap = (size_t)n + 16;
        AV_HIP(hipHostMalloc((void**)&u.pin, cap * 2 * hw, hipHostMallocDefault));
        AV_HIP(hipHostMalloc((void**)&u.idx_h, cap * sizeof(int), hipHostMallocDefault));
        AV_HIP(hipMalloc((void**)&u.idx_d, cap * sizeof(int)));
        if (!u.done) AV_HIP(hipEventCreateWithFlags(&u.done, hipEventDisableTiming));
        u.cap = cap;
    }
#pragma omp parallel for schedule(static) num_threads(n >= 8 ? 8 : 1)
    for (int i = 0; i < 2 * n; ++i) {
        const int f = i >> 1, cam = i & 1;
        memcpy(u.pin + ((size_t)2 * f + cam) * hw, (cam ? img1_host : img0_host) + (size_t)f * img_stride, hw);
    }
    for (int i = 0; i < n; ++i) u.idx_h[i] = slots[i];
    hipStream_t cs = fe->copy_stream;
    // The entries handed to an upload are free as of the newest ENQUEUED step (the caller's promise): the copies wait for that step
    // and run beside whatever the caller enqueues next -- upload the frames of step k + 1 before enqueueing step k and the two overlap.
    if (fs.any_step) AV_HIP(hipStreamWaitEvent(cs, fs.stepped, 0));
    AV_HIP(hipMemcpyAsync(u.idx_d, u.idx_h, sizeof(int) * n, hipMemcpyHostToDevice, cs));
    for (int i = 0; i < n;) {                                        // runs of consecutive entries go down in one copy
        int j = i + 1;
        while (j < n && slots[j] == slots[j - 1] + 1) ++j;
        AV_HIP(hipMemcpyAsync(fs.img + (size_t)slots[i] * 2 * hw, u.pin + (size_t)i * 2 * hw, (size_t)(j - i) * 2 * hw, hipMemcpyHostToDevice, cs));
        i = j;
    }
    int rc;
    bool wrote_l0 = true;
    if ((rc = av_launch_pyramid(fs.img, fs.img + hw, (int64_t)(2 * hw), n, 2, fe->geom, fs.pyr, 2 * fe->lay.bytes, fe->lay.bytes, 0, 1, cs, false, &wrote_l0, u.idx_d))) return rc;
    fs.l0_in_place = !wrote_l0;
    if (fs.l0_in_place) rc = av_launch_fast(fs.img, (int64_t)(2 * hw), d.w, 0, nullptr, 0, n, d.w, d.h, fe->cfg.fast_threshold,
                                            nullptr, nullptr, 0, fs.tile_kp, fs.tile_count, nullptr, 0, cs, u.idx_d);
    else rc = av_launch_fast(fs.pyr + fe->geom.off[0] + (size_t)AV_PYR_BORDER * fe->geom.pitch[0] + AV_PYR_BORDER, 2 * fe->lay.bytes, fe->geom.pitch[0], AV_PYR_BORDER,
                             nullptr, 0, n, d.w, d.h, fe->cfg.fast_threshold, nullptr, nullptr, 0, fs.tile_kp, fs.tile_count, nullptr, 0, cs, u.idx_d);
    if (rc) return rc;
    AV_HIP(hipEventRecord(u.done, cs));
    AV_HIP(hipEventRecord(fs.uploaded, cs));
    u.used = true; fs.any_upload = true;
    fs.frames_uploaded += n;
    return AV_OK;
}

AV_EXPORT int av_frontend_step_frames(av_frontend* fe, const int32_t* slot_of_stream, const double* timestamps, void* stream)
{
    if (!fe || !slot_of_stream || !timestamps) { av_set_error("av_frontend_step_frames: bad arguments"); return AV_E_INVALID; }
    if (!fe->fs.n_slots) { av_set_error("av_frontend_step_frames: no frame store (av_frontend_frames_reserve first)"); return AV_E_INVALID; }
    for (int s = 0; s < fe->d.S; ++s)
        if (slot_of_stream[s] >= fe->fs.n_slots) { av_set_error("av_frontend_step_frames: stream %d: entry %d outside the store (%d entries)", s, slot_of_stream[s], fe->fs.n_slots); return AV_E_INVALID; }
    return step_impl(fe, nullptr, nullptr, 0, timestamps, (hipStream_t)stream, true, slot_of_stream);
}

AV_EXPORT int av_frontend_max_features(const av_frontend* fe) { return fe ? fe->d.MAXF : AV_E_INVALID; }

// The feature_msg of the last step where it lies on the device (consumed by av_msckf_batch_submit_dev)
AV_EXPORT int av_frontend_features_dev(av_frontend* fe, const int64_t** ids_dev, const double** uv_dev, const int32_t** n_dev, int* cap)
{
    if (!fe || !ids_dev || !uv_dev || !n_dev || !cap) { av_set_error("av_frontend_features_dev: bad arguments"); return AV_E_INVALID; }
    static_assert(sizeof(long long) == sizeof(int64_t), "feature ids");
    *ids_dev = reinterpret_cast<const int64_t*>(fe->d.out_ids); *uv_dev = fe->d.out_uv; *n_dev = fe->d.out_n; *cap = fe->d.MAXF;
    return AV_OK;
}

// Read-back in two halves.  _begin enqueues the device-to-host copies of the features published by the last step into
// pinned slot `slot` (0/1) behind everything already on `stream` and returns; _end waits for exactly those copies and
// unpacks them into the caller's arrays.  A caller that enqueues the NEXT step between the two keeps the GPU busy while
// it consumes this frame's features.
AV_EXPORT int av_frontend_read_features_begin(av_frontend* fe, int slot, void* stream)
{
    if (!fe || slot < 0 || slot > 1) { av_set_error("av_frontend_read_features_begin: bad arguments"); return AV_E_INVALID; }
    hipStream_t st = (hipStream_t)stream;
    const FeDev& d = fe->d;
    av_frontend::ReadSlot& r = fe->rd[slot];
    AV_HIP(hipSetDevice(fe->device));
    AV_HIP(hipMemcpyAsync(r.n, d.out_n, sizeof(int) * d.S, hipMemcpyDeviceToHost, st));
    AV_HIP(hipMemcpyAsync(r.ids, d.out_ids, sizeof(long long) * d.S * d.MAXF, hipMemcpyDeviceToHost, st));
    AV_HIP(hipMemcpyAsync(r.uv, d.out_uv, sizeof(double) * 4 * d.S * d.MAXF, hipMemcpyDeviceToHost, st));
    AV_HIP(hipMemcpyAsync(r.cnt, d.counters, sizeof(int) * d.S * NCNT, hipMemcpyDeviceToHost, st));
    AV_HIP(hipEventRecord(r.ev, st));
    r.pending = true;
    return AV_OK;
}

AV_EXPORT int av_frontend_read_features_end(av_frontend* fe, int slot, int64_t* ids_out, double* uv_out, int32_t* n_out, int cap)
{
    if (!fe || slot < 0 || slot > 1 || !ids_out || !uv_out || !n_out || cap < fe->d.MAXF) { av_set_error("av_frontend_read_features_end: bad arguments"); return AV_E_INVALID; }
    const FeDev& d = fe->d;
    av_frontend::ReadSlot& r = fe->rd[slot];
    if (!r.pending) { av_set_error("av_frontend_read_features_end: no read-back was begun on slot %d", slot); return AV_E_INVALID; }
    AV_HIP(hipSetDevice(fe->device));
    AV_HIP(hipEventSynchronize(r.ev));
    r.pending = false;
    int ovf = 0;
#pragma omp parallel for schedule(static) num_threads(d.S >= 64 ? 4 : 1) reduction(|:ovf)
    for (int s = 0; s < d.S; ++s) {
        const int n = r.n[s];
        n_out[s] = n;
        memcpy(ids_out + (size_t)s * cap, r.ids + (size_t)s * d.MAXF, sizeof(long long) * (size_t)n);
        memcpy(uv_out + (size_t)s * cap * 4, r.uv + (size_t)s * d.MAXF * 4, sizeof(double) * 4 * (size_t)n);
        ovf |= r.cnt[(size_t)s * NCNT + CNT_OVF];
    }
    if (ovf) { av_set_error("front-end device buffer overflow (flags 0x%x): raise max_corners", ovf); return AV_E_CAPACITY; }
    return AV_OK;
}

AV_EXPORT int av_frontend_read_features(av_frontend* fe, int64_t* ids_out, double* uv_out, int32_t* n_out, int cap, void* stream)
{
    int rc = av_frontend_read_features_begin(fe, 0, stream);
    if (rc) return rc;
    return av_frontend_read_features_end(fe, 0, ids_out, uv_out, n_out, cap);
}

AV_EXPORT int av_frontend_read_grid(av_frontend* fe, int stream_idx, int64_t* ids, int32_t* lifetime, int32_t* cell,
                                    float* pts, int cap, int32_t* n_out, int64_t* next_feature_id, void* stream)
{
    if (!fe || stream_idx < 0 || stream_idx >= fe->d.S || !ids || !lifetime || !cell || !pts || !n_out || cap < fe->d.MAXF) {
        av_set_error("av_frontend_read_grid: bad arguments");
        return AV_E_INVALID;
    }
    hipStream_t st = (hipStream_t)stream;
    const FeDev& d = fe->d;
    const int b = fe->parity;        // after a step, parity indexes the grid just published
    const size_t o = (size_t)stream_idx * d.MAXF;
    AV_HIP(hipSetDevice(fe->device));
    AV_HIP(hipStreamSynchronize(st));
    int n = 0; long long nid = 0;
    AV_HIP(hipMemcpy(&n, d.n_feat[b] + stream_idx, sizeof(int), hipMemcpyDeviceToHost));
    AV_HIP(hipMemcpy(&nid, d.next_id + stream_idx, sizeof(long long), hipMemcpyDeviceToHost));
    std::vector<float> p0(2 * (size_t)d.MAXF), p1(2 * (size_t)d.MAXF);
    std::vector<long long> idv(d.MAXF);
    AV_HIP(hipMemcpy(idv.data(), d.feat_id[b] + o, sizeof(long long) * d.MAXF, hipMemcpyDeviceToHost));
    AV_HIP(hipMemcpy(lifetime, d.feat_life[b] + o, sizeof(int) * d.MAXF, hipMemcpyDeviceToHost));
    AV_HIP(hipMemcpy(cell, d.feat_cell[b] + o, sizeof(int) * d.MAXF, hipMemcpyDeviceToHost));
    AV_HIP(hipMemcpy(p0.data(), d.feat_p0[b] + 2 * o, sizeof(float) * 2 * d.MAXF, hipMemcpyDeviceToHost));
    AV_HIP(hipMemcpy(p1.data(), d.feat_p1[b] + 2 * o, sizeof(float) * 2 * d.MAXF, hipMemcpyDeviceToHost));
    for (int k = 0; k < n; ++k) {
        ids[k] = idv[k];
        pts[4 * k] = p0[2 * k]; pts[4 * k + 1] = p0[2 * k + 1]; pts[4 * k + 2] = p1[2 * k]; pts[4 * k + 3] = p1[2 * k + 1];
    }
    *n_out = n;
    if (next_feature_id) *next_feature_id = nid;
    return AV_OK;
}

// [candidates matched in round 1, in round 2] of the last step for one stream (the lazy stereo matching of
// cand_round2_kernel): the LK point passes actually run for new-feature candidates.  Synchronises.
AV_EXPORT int av_frontend_read_match_counts(av_frontend* fe, int stream_idx, int32_t out[2], void* stream)
{
    if (!fe || !out || stream_idx < 0 || stream_idx >= fe->d.S) { av_set_error("av_frontend_read_match_counts: bad arguments"); return AV_E_INVALID; }
    AV_HIP(hipSetDevice(fe->device));
    AV_HIP(hipStreamSynchronize((hipStream_t)stream));
    AV_HIP(hipMemcpy(out, fe->d.r1_count + stream_idx, sizeof(int), hipMemcpyDeviceToHost));
    AV_HIP(hipMemcpy(out + 1, fe->d.r2_count + stream_idx, sizeof(int), hipMemcpyDeviceToHost));
    return AV_OK;
}

AV_EXPORT int av_frontend_read_counters(av_frontend* fe, int stream_idx, int32_t out[8], void* stream)
{
    if (!fe || stream_idx < 0 || stream_idx >= fe->d.S || !out) { av_set_error("av_frontend_read_counters: bad arguments"); return AV_E_INVALID; }
    AV_HIP(hipSetDevice(fe->device));
    AV_HIP(hipStreamSynchronize((hipStream_t)stream));
    AV_HIP(hipMemcpy(out, fe->d.counters + (size_t)stream_idx * NCNT, sizeof(int) * NCNT, hipMemcpyDeviceToHost));
    return AV_OK;
}

AV_EXPORT int av_frontend_enable_timing(av_frontend* fe, int max_spans)
{
    if (!fe || max_spans < 0) { av_set_error("av_frontend_enable_timing: bad arguments"); return AV_E_INVALID; }
    AV_HIP(hipSetDevice(fe->device));
    AV_HIP(hipDeviceSynchronize());
    for (hipEvent_t e : fe->ev) (void)hipEventDestroy(e);
    fe->ev.clear(); fe->ev_cls.clear(); fe->ev_used = 0;
    fe->timing = max_spans > 0;
    for (int i = 0; i < 2 * max_spans; ++i) {
        hipEvent_t e;
        AV_HIP(hipEventCreate(&e));
        fe->ev.push_back(e);
    }
    fe->ev_cls.assign((size_t)max_spans, 0);
    return AV_OK;
}

AV_EXPORT int av_frontend_read_timing(av_frontend* fe, double ms_out[4], int32_t spans_out[4])
{
    if (!fe || !ms_out || !spans_out) { av_set_error("av_frontend_read_timing: bad arguments"); return AV_E_INVALID; }
    AV_HIP(hipSetDevice(fe->device));
    AV_HIP(hipDeviceSynchronize());
    for (int i = 0; i < 4; ++i) { ms_out[i] = 0.0; spans_out[i] = 0; }
    for (size_t k = 0; k + 1 < fe->ev_used; k += 2) {
        float ms = 0.f;
        AV_HIP(hipEventElapsedTime(&ms, fe->ev[k], fe->ev[k + 1]));
        int c = fe->ev_cls[k / 2];
        ms_out[c] += (double)ms; spans_out[c] += 1;
    }
    fe->ev_used = 0;
    return AV_OK;
}
