// msckf_store.h -- the filter's observation map (map_server, src/msckf.py:120) RESIDENT ON THE DEVICE.
//
// Reference semantics kept: map_server is a dict id -> Feature whose observations are a dict
// camera-state id -> (u0, v0, u1, v1); dict order = insertion (birth) order defines the order in which lost features and
// camera-pruning candidates are stacked (msckf.py:425-441 add_feature_observations, 614-676 remove_lost_features,
// 712-786 prune_cam_state_buffer).
//
// Layout (per stream; the batched filter holds S of these side by side): FRAME-MAJOR.  One frame per camera state of the
// window: the message as it arrived (ids, measurements) plus, per entry, links to the same feature's entry in the previous
// and the next frame it was seen in, and the feature's slot in a small table (birth number, position, initialised flag,
// observation count, newest entry).  Appending a frame = a copy of the front-end's published arrays + one probe per
// feature into a hash of the previous frame's ids; lost features = entries of the previous frame without a successor;
// a feature's observations = a walk along its links; pruning candidates = links from one removed frame that reach the
// other; removing a camera state removes its frame and joins the neighbours' links.
//
// Every operation below is written for ONE TEAM (a 256-thread workgroup on the device) working on ONE stream: loops are
// team-strided, ordered compactions use the team's exclusive scan, phases are separated by team barriers.  The same text
// compiles for the host with a one-thread team (AVS_CPU_MODEL: tests/native/devstore_harness.cpp runs it against a dict model
// of the reference under ASan/UBSan) -- that build is test infrastructure only; the product path is the device build.
#pragma once
#include <stdint.h>

#ifdef AVS_CPU_MODEL
#define AVS_FN inline
struct AvsTeam {
    int tid() const { return 0; }
    int nt() const { return 1; }
    void sync() const {}
    int excl_scan(int v, int* total) const { *total = v; return 0; }
    static int atomic_cas(int* p, int cmp, int val) { const int o = *p; if (o == cmp) *p = val; return o; }
    static int atomic_min(int* p, int v) { const int o = *p; if (v < o) *p = v; return o; }
    static int atomic_max(int* p, int v) { const int o = *p; if (v > o) *p = v; return o; }
};
#else
#define AVS_FN __device__ __forceinline__
// 256-thread workgroup; `red` = 8 ints of LDS owned by the team object
struct AvsTeam {
    int* red;
    __device__ __forceinline__ int tid() const { return (int)threadIdx.x; }
    __device__ __forceinline__ int nt() const { return (int)blockDim.x; }
    __device__ __forceinline__ void sync() const { __syncthreads(); }
    // exclusive prefix of v over the team's threads (thread order), total to everybody; two barriers
    __device__ __forceinline__ int excl_scan(int v, int* total) const
    {
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
        int incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(incl, d, 64); if (lane >= d) incl += t; }
        __syncthreads();                                  // `red` may still be read by the previous call
        if (lane == 63) red[w] = incl;
        __syncthreads();
        int base = 0, tot = 0;
        for (int i = 0; i < nw; ++i) { const int r = red[i]; if (i < w) base += r; tot += r; }
        *total = tot;
        return base + incl - v;
    }
    __device__ static __forceinline__ int atomic_cas(int* p, int cmp, int val) { return atomicCAS(p, cmp, val); }
    __device__ static __forceinline__ int atomic_min(int* p, int v) { return atomicMin(p, v); }
    __device__ static __forceinline__ int atomic_max(int* p, int v) { return atomicMax(p, v); }
};
#endif

// header words of a stream's store
enum { AVS_N_ORDER = 0, AVS_N_FREE = 1, AVS_N_USED = 2, AVS_LIVE = 3, AVS_OVERFLOW = 4, AVS_HDR_WORDS = 8 };

struct AvsStore {
    int cap;                 // entries per frame (= capacity of a feature message)
    int mcap;                // feature table slots (>= 2 cap: a stream holds at most cap features after a step, 2 cap inside one)
    int ns;                  // frame slots (> camera states of a full window)
    int hcells;              // hash cells, power of two >= 2 cap + 16
    long long* fr_id; double* fr_z; int* fr_fslot; int* fr_prev; int* fr_next;      // [ns][cap] (z: [ns][cap][4])
    int* fr_n; int* pos_of; int* order;                                              // [ns]: entries; slot -> camera index (-1 free); camera index -> slot
    int* hdr;                // [AVS_HDR_WORDS]
    long long* births;       // [1] insertion counter of the dict
    long long* m_id; long long* m_birth; double* m_pos; int* m_init; int* m_nobs; int* m_tail;   // [mcap] (pos: [mcap][3])
    int* free_stack;         // [mcap]
    // scratch of the operations (contents do not survive a call)
    int* hash_a; int* hash_b; int* hash_c;      // [hcells] each
    int* tmp_a; int* tmp_b; int* tmp_c;         // [cap] each
    unsigned long long* keys;                   // [cap]
};

AVS_FN int avs_link(int slot, int idx) { return (slot << 24) | idx; }
AVS_FN int avs_lslot(int l) { return l >> 24; }
AVS_FN int avs_lidx(int l) { return l & 0xFFFFFF; }
AVS_FN unsigned avs_mix(long long id) { unsigned long long x = (unsigned long long)id * 0x9E3779B97F4A7C15ull; return (unsigned)(x ^ (x >> 29)); }

// reset_state / online_reset (msckf.py:800-843): empty map, empty window
AVS_FN void avs_clear(const AvsTeam& T, const AvsStore& V)
{
    for (int i = T.tid(); i < V.ns; i += T.nt()) { V.fr_n[i] = 0; V.pos_of[i] = -1; V.order[i] = -1; }
    if (T.tid() == 0) { V.hdr[AVS_N_ORDER] = 0; V.hdr[AVS_N_FREE] = 0; V.hdr[AVS_N_USED] = 0; V.hdr[AVS_LIVE] = 0; }
    T.sync();
}

// add_feature_observations for one message (msckf.py:425-441).  A duplicate id inside one message re-assigns the same dict
// key: the entry keeps the position of the first occurrence and takes the measurement of the last.  Returns the frame slot
// (to every thread), or -1 if no slot is free / -2 if the feature table is full (nothing is changed then); *tracked = the
// reference's count of messages entries that hit an existing feature (tracking_rate = tracked / len(map) before).
AVS_FN int avs_add_frame(const AvsTeam& T, const AvsStore& V, const long long* ids, const double* uv, int nk, int* tracked_out)
{
    const int n_order = V.hdr[AVS_N_ORDER];
    int slot = -1;
    for (int i = 0; i < V.ns; ++i) if (V.pos_of[i] < 0) { slot = i; break; }          // uniform: every thread reads the same table
    if (slot < 0 || nk > V.cap) { *tracked_out = 0; return -1; }
    const int pslot = n_order > 0 ? V.order[n_order - 1] : -1;
    const int pn = pslot >= 0 ? V.fr_n[pslot] : 0;
    const long long* pid = pslot >= 0 ? V.fr_id + (size_t)pslot * V.cap : nullptr;
    const unsigned hm = (unsigned)V.hcells - 1u;
    int* hp = V.hash_a;          // previous frame: cell -> entry index of the previous frame (-1 empty)
    int* hf = V.hash_b;          // this message: cell -> FIRST position k of its id
    int* hl = V.hash_c;          //               cell -> LAST position k of its id
    for (int i = T.tid(); i < V.hcells; i += T.nt()) { hp[i] = -1; hf[i] = -1; hl[i] = -1; }
    T.sync();
    for (int i = T.tid(); i < pn; i += T.nt()) {
        const long long id = pid[i];
        for (unsigned h = avs_mix(id) & hm;; h = (h + 1) & hm) if (AvsTeam::atomic_cas(&hp[h], -1, i) == -1) break;      // ids of a stored frame are distinct
    }
    for (int k = T.tid(); k < nk; k += T.nt()) {
        const long long id = ids[k];
        for (unsigned h = avs_mix(id) & hm;; h = (h + 1) & hm) {
            int v = hf[h];
            if (v < 0) { v = AvsTeam::atomic_cas(&hf[h], -1, k); if (v < 0) { AvsTeam::atomic_max(&hl[h], k); break; } }
            if (ids[v] == id) { AvsTeam::atomic_min(&hf[h], k); AvsTeam::atomic_max(&hl[h], k); break; }      // (a cell only ever holds positions of ONE id)
        }
    }
    T.sync();
    // ordered pass over the message: representative entries (first occurrence) in message order, new features numbered in message order
    const int n_free0 = V.hdr[AVS_N_FREE], n_used0 = V.hdr[AVS_N_USED];
    const long long births0 = *V.births;
    long long* fid = V.fr_id + (size_t)slot * V.cap; double* fz = V.fr_z + (size_t)slot * V.cap * 4;
    int* ffs = V.fr_fslot + (size_t)slot * V.cap; int* fprev = V.fr_prev + (size_t)slot * V.cap; int* fnext = V.fr_next + (size_t)slot * V.cap;
    int n_ent = 0, n_new = 0, n_trk = 0;
    // pass 1: how many new features does the message bring?  (the table must hold them before anything is written)
    {
        int cnt = 0;
        for (int base = 0; base < nk; base += T.nt()) {
            const int k = base + T.tid();
            int is_new = 0;
            if (k < nk) {
                const long long id = ids[k];
                unsigned h = avs_mix(id) & hm;
                while (ids[hf[h]] != id) h = (h + 1) & hm;
                if (hf[h] == k) {
                    int pi = -1;
                    for (unsigned g = avs_mix(id) & hm;; g = (g + 1) & hm) { const int v = hp[g]; if (v < 0) break; if (pid[v] == id) { pi = v; break; } }
                    is_new = !(pi >= 0 && V.fr_fslot[(size_t)pslot * V.cap + pi] >= 0);
                }
            }
            int tot; (void)T.excl_scan(is_new, &tot);
            cnt += tot;
        }
        if (cnt > n_free0 + (V.mcap - n_used0)) { *tracked_out = 0; return -2; }
    }
    for (int base = 0; base < nk; base += T.nt()) {
        const int k = base + T.tid();
        int rep = 0, is_new = 0, trk = 0, pi = -1, last = k;
        long long id = 0;
        if (k < nk) {
            id = ids[k];
            unsigned h = avs_mix(id) & hm;
            while (ids[hf[h]] != id) h = (h + 1) & hm;
            rep = hf[h] == k; last = hl[h];
            if (rep) {
                for (unsigned g = avs_mix(id) & hm;; g = (g + 1) & hm) { const int v = hp[g]; if (v < 0) break; if (pid[v] == id) { pi = v; break; } }
                const bool hit = pi >= 0 && V.fr_fslot[(size_t)pslot * V.cap + pi] >= 0;
                is_new = !hit; trk = hit;
            } else trk = 1;                                  // a repeated id hits the feature its first occurrence made or found
        }
        int tot_e, tot_n, tot_t;
        const int i = n_ent + T.excl_scan(rep, &tot_e);
        const int r = n_new + T.excl_scan(is_new, &tot_n);
        (void)T.excl_scan(trk, &tot_t);
        if (rep) {
            fid[i] = id;
            for (int e = 0; e < 4; ++e) fz[(size_t)4 * i + e] = uv[(size_t)4 * last + e];
            fnext[i] = -1;
            if (!is_new) {
                const int fs = V.fr_fslot[(size_t)pslot * V.cap + pi];
                ffs[i] = fs; fprev[i] = avs_link(pslot, pi); V.fr_next[(size_t)pslot * V.cap + pi] = avs_link(slot, i);
                V.m_nobs[fs] += 1; V.m_tail[fs] = avs_link(slot, i);
            } else {
                const int fs = r < n_free0 ? V.free_stack[n_free0 - 1 - r] : n_used0 + (r - n_free0);
                V.m_id[fs] = id; V.m_birth[fs] = births0 + r; V.m_init[fs] = 0; V.m_nobs[fs] = 1; V.m_tail[fs] = avs_link(slot, i);
                V.m_pos[(size_t)3 * fs] = 0; V.m_pos[(size_t)3 * fs + 1] = 0; V.m_pos[(size_t)3 * fs + 2] = 0;
                ffs[i] = fs; fprev[i] = -1;
            }
        }
        n_ent += tot_e; n_new += tot_n; n_trk += tot_t;
    }
    T.sync();
    if (T.tid() == 0) {
        V.fr_n[slot] = n_ent;
        V.hdr[AVS_N_FREE] = n_new < n_free0 ? n_free0 - n_new : 0;
        V.hdr[AVS_N_USED] = n_new > n_free0 ? n_used0 + (n_new - n_free0) : n_used0;
        V.hdr[AVS_LIVE] += n_new;
        *V.births = births0 + n_new;
        V.pos_of[slot] = n_order; V.order[n_order] = slot; V.hdr[AVS_N_ORDER] = n_order + 1;
    }
    T.sync();
    *tracked_out = n_trk;
    return slot;
}

// Ordering of `n` feature slots by birth (dict order): out[rank] = in[i], and the same permutation for up to two companion arrays.
// Births are distinct, so the rank of an item is the number of items born before it -- no barriers inside the count, any memory
// for the keys (V.keys: the kernels point it at LDS when the list fits, which makes the n reads per item broadcasts).
AVS_FN void avs_sort_by_birth(const AvsTeam& T, const AvsStore& V, const int* in, int n, int* out,
                              const int* a0_in = nullptr, int* a0_out = nullptr, const int* a1_in = nullptr, int* a1_out = nullptr)
{
    for (int i = T.tid(); i < n; i += T.nt()) V.keys[i] = (unsigned long long)V.m_birth[in[i]];
    T.sync();
    for (int i = T.tid(); i < n; i += T.nt()) {
        const unsigned long long k = V.keys[i];
        int r = 0;
        for (int j = 0; j < n; ++j) r += V.keys[j] < k;
        out[r] = in[i];
        if (a0_in) a0_out[r] = a0_in[i];
        if (a1_in) a1_out[r] = a1_in[i];
    }
    T.sync();
}

// remove_lost_features, selection (msckf.py:614-640): features not observed in the newest frame = entries of the frame before
// it without a successor (every live feature was seen in that frame: features that are not are deleted in the same step).
// cand[0 .. *n_cand): feature slots with >= 3 observations in dict order; the others (invalid[0 .. *n_inv)) are only deleted.
// cand / invalid: caller's arrays of V.cap ints (not V.tmp_a / V.tmp_b).
AVS_FN void avs_select_lost(const AvsTeam& T, const AvsStore& V, int* cand, int* n_cand, int* invalid, int* n_inv)
{
    const int n_order = V.hdr[AVS_N_ORDER];
    if (n_order < 2) { *n_cand = 0; *n_inv = 0; return; }
    const int ps = V.order[n_order - 2], pn = V.fr_n[ps];
    const int* pfs = V.fr_fslot + (size_t)ps * V.cap; const int* pnext = V.fr_next + (size_t)ps * V.cap;
    int nc = 0, ni = 0;
    for (int base = 0; base < pn; base += T.nt()) {
        const int i = base + T.tid();
        int fs = -1, is_c = 0, is_i = 0;
        if (i < pn) { fs = pfs[i]; if (fs >= 0 && pnext[i] < 0) { if (V.m_nobs[fs] < 3) is_i = 1; else is_c = 1; } }
        int tc, ti;
        const int oc = nc + T.excl_scan(is_c, &tc), oi = ni + T.excl_scan(is_i, &ti);
        if (is_c) V.tmp_a[oc] = fs;
        if (is_i) invalid[oi] = fs;
        nc += tc; ni += ti;
    }
    T.sync();
    avs_sort_by_birth(T, V, V.tmp_a, nc, cand);
    *n_cand = nc; *n_inv = ni;
}

// observations of feature slot fs in ascending camera order, written backwards from its newest entry:
// cam[q] = camera index (position in the window), z[4 q ..] = measurement, q < m_nobs[fs].  One thread.
AVS_FN int avs_collect(const AvsStore& V, int fs, int* cam, double* z)
{
    const int n = V.m_nobs[fs];
    int q = n - 1;
    for (int l = V.m_tail[fs]; l >= 0 && q >= 0; --q) {
        const int s = avs_lslot(l), i = avs_lidx(l);
        cam[q] = V.pos_of[s];
        const double* zz = V.fr_z + ((size_t)s * V.cap + i) * 4;
        z[(size_t)4 * q] = zz[0]; z[(size_t)4 * q + 1] = zz[1]; z[(size_t)4 * q + 2] = zz[2]; z[(size_t)4 * q + 3] = zz[3];
        l = V.fr_prev[(size_t)s * V.cap + i];
    }
    return n;
}

// delete the features list[0 .. n): their entries stay in their frames as dead entries until the frames go
AVS_FN void avs_erase(const AvsTeam& T, const AvsStore& V, const int* list, int n)
{
    const int n_free0 = V.hdr[AVS_N_FREE];
    T.sync();
    for (int k = T.tid(); k < n; k += T.nt()) {
        const int fs = list[k];
        for (int l = V.m_tail[fs]; l >= 0; ) { const int s = avs_lslot(l), i = avs_lidx(l); V.fr_fslot[(size_t)s * V.cap + i] = -1; l = V.fr_prev[(size_t)s * V.cap + i]; }
        V.m_tail[fs] = -1; V.m_nobs[fs] = 0;
        V.free_stack[n_free0 + k] = fs;
    }
    T.sync();
    if (T.tid() == 0) { V.hdr[AVS_N_FREE] = n_free0 + n; V.hdr[AVS_LIVE] -= n; }
    T.sync();
}

// prune_cam_state_buffer, candidates (msckf.py:736-757): features seen from BOTH cameras i0 < i1 (window positions) -- follow
// the links of the entries of frame i0 to frame i1.  cand[k] = feature slot (dict order), e0[k] / e1[k] = its entry in the two
// frames (links).  cand / e0 / e1: caller's arrays of V.cap ints.
AVS_FN void avs_select_prune(const AvsTeam& T, const AvsStore& V, int i0, int i1, int* cand, int* e0, int* e1, int* n_cand)
{
    const int s0 = V.order[i0], an = V.fr_n[s0];
    const int* afs = V.fr_fslot + (size_t)s0 * V.cap; const int* anext = V.fr_next + (size_t)s0 * V.cap;
    int nc = 0;
    for (int base = 0; base < an; base += T.nt()) {
        const int i = base + T.tid();
        int fs = -1, hit = 0, l = -1;
        if (i < an) {
            fs = afs[i];
            if (fs >= 0) {
                l = anext[i];
                while (l >= 0 && V.pos_of[avs_lslot(l)] < i1) l = V.fr_next[(size_t)avs_lslot(l) * V.cap + avs_lidx(l)];
                hit = l >= 0 && V.pos_of[avs_lslot(l)] == i1;
            }
        }
        int tc;
        const int o = nc + T.excl_scan(hit, &tc);
        if (hit) { V.tmp_a[o] = fs; V.tmp_b[o] = l; V.tmp_c[o] = avs_link(s0, i); }
        nc += tc;
    }
    T.sync();
    avs_sort_by_birth(T, V, V.tmp_a, nc, cand, V.tmp_b, e1, V.tmp_c, e0);
    *n_cand = nc;
}

// the camera state at window position ci leaves the window: its frame goes, the links of its neighbours are joined
AVS_FN void avs_remove_frame(const AvsTeam& T, const AvsStore& V, int ci)
{
    const int n_order = V.hdr[AVS_N_ORDER];
    const int slot = V.order[ci], n = V.fr_n[slot];
    T.sync();
    for (int i = T.tid(); i < n; i += T.nt()) {
        const int fs = V.fr_fslot[(size_t)slot * V.cap + i];
        if (fs < 0) continue;
        const int pl = V.fr_prev[(size_t)slot * V.cap + i], nl = V.fr_next[(size_t)slot * V.cap + i];
        if (pl >= 0) V.fr_next[(size_t)avs_lslot(pl) * V.cap + avs_lidx(pl)] = nl;
        if (nl >= 0) V.fr_prev[(size_t)avs_lslot(nl) * V.cap + avs_lidx(nl)] = pl;
        V.m_nobs[fs] -= 1;
        if (nl < 0) V.m_tail[fs] = pl;
    }
    T.sync();
    if (T.tid() == 0) {
        V.fr_n[slot] = 0; V.pos_of[slot] = -1;
        for (int k = ci; k + 1 < n_order; ++k) { V.order[k] = V.order[k + 1]; V.pos_of[V.order[k]] = k; }
        V.order[n_order - 1] = -1;
        V.hdr[AVS_N_ORDER] = n_order - 1;
    }
    T.sync();
}
