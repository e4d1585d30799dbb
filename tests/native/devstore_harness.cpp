// CPU harness for the device-resident observation store (uav_airvision_amd/csrc/msckf_store.h, compiled here with a one-thread
// team: AVS_CPU_MODEL): drives it and a plain dict-of-dicts model of the reference's map_server (msckf.py:120, 425-441,
// 614-676, 712-786) through the same random message sequences and compares, every frame: live features, tracked counts, the
// lost features in dict order with their observation lists, the camera-pruning candidates in dict order with their two
// entries, and every live feature's observation list after the two frames are removed.  Run under ASan / UBSan.
#define AVS_CPU_MODEL 1
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <random>
#include <vector>
#include "msckf_store.h"

struct Owned {
    std::vector<long long> fr_id, births, m_id, m_birth; std::vector<double> fr_z, m_pos;
    std::vector<int> fr_fslot, fr_prev, fr_next, fr_n, pos_of, order, hdr, m_init, m_nobs, m_tail, free_stack, ha, hb, hc, ta, tb, tc;
    std::vector<unsigned long long> keys;
    AvsStore V;
    Owned(int cap, int ns)
    {
        V.cap = cap; V.mcap = 2 * cap + 64; V.ns = ns;
        int hc_ = 64; while (hc_ < 2 * cap + 16) hc_ *= 2;
        V.hcells = hc_;
        fr_id.resize((size_t)ns * cap); fr_z.resize((size_t)ns * cap * 4); fr_fslot.resize((size_t)ns * cap); fr_prev.resize((size_t)ns * cap); fr_next.resize((size_t)ns * cap);
        fr_n.assign(ns, 0); pos_of.assign(ns, -1); order.assign(ns, -1); hdr.assign(AVS_HDR_WORDS, 0); births.assign(1, 0);
        m_id.resize(V.mcap); m_birth.resize(V.mcap); m_pos.resize((size_t)V.mcap * 3); m_init.resize(V.mcap); m_nobs.resize(V.mcap); m_tail.resize(V.mcap); free_stack.resize(V.mcap);
        ha.resize(hc_); hb.resize(hc_); hc.resize(hc_); ta.resize(cap); tb.resize(cap); tc.resize(cap); keys.resize(cap);
        V.fr_id = fr_id.data(); V.fr_z = fr_z.data(); V.fr_fslot = fr_fslot.data(); V.fr_prev = fr_prev.data(); V.fr_next = fr_next.data();
        V.fr_n = fr_n.data(); V.pos_of = pos_of.data(); V.order = order.data(); V.hdr = hdr.data(); V.births = births.data();
        V.m_id = m_id.data(); V.m_birth = m_birth.data(); V.m_pos = m_pos.data(); V.m_init = m_init.data(); V.m_nobs = m_nobs.data(); V.m_tail = m_tail.data();
        V.free_stack = free_stack.data(); V.hash_a = ha.data(); V.hash_b = hb.data(); V.hash_c = hc.data(); V.tmp_a = ta.data(); V.tmp_b = tb.data(); V.tmp_c = tc.data(); V.keys = keys.data();
    }
};
struct ModelFeat { long long birth; std::vector<std::pair<long long, std::vector<double>>> obs; };
int main(int argc, char** argv)
{
    const unsigned seed = argc > 1 ? (unsigned)atoi(argv[1]) : 1;
    const int frames = argc > 2 ? atoi(argv[2]) : 200, per = argc > 3 ? atoi(argv[3]) : 60, max_cam = 20;
    std::mt19937 rng(seed);
    const int cap = per + 8;
    Owned O(cap, max_cam + 2);
    AvsStore& S = O.V;
    AvsTeam T;
    avs_clear(T, S);
    std::map<long long, ModelFeat> M;                       // id -> feature (dict order = birth)
    std::vector<long long> cams;                            // camera ids in the window (parallel to S.order)
    long long births = 0, next_id = 1;
    std::vector<long long> alive;
    long long checks = 0;
    std::vector<int> cand(cap), inval(cap), e0(cap), e1(cap), ocam(64);
    std::vector<double> oz(64 * 4);
    for (int t = 0; t < frames; ++t) {
        std::vector<long long> ids;
        for (long long id : alive) if (rng() % 100 < 85) ids.push_back(id);
        std::shuffle(ids.begin(), ids.end(), rng);
        const int n_new = per - (int)ids.size() > 0 ? (int)(rng() % (unsigned)(per - (int)ids.size() + 1)) : 0;
        for (int k = 0; k < n_new; ++k) ids.push_back(next_id++);
        if (!ids.empty() && rng() % 7 == 0) ids.push_back(ids[rng() % ids.size()]);          // a duplicate id now and then
        if (!ids.empty() && rng() % 11 == 0) { const long long d = ids[rng() % ids.size()]; ids.insert(ids.begin(), d); ids.push_back(d); }   // ... and a triple
        if (rng() % 31 == 0) ids.clear();                   // a blank frame loses every track
        if ((int)ids.size() > cap) ids.resize(cap);
        std::vector<double> uv(4 * ids.size());
        for (double& v : uv) v = (double)(rng() % 100000) / 1000.0;
        const long long cam = t;
        // ---- model
        long long tracked_m = 0;
        const size_t before = M.size();
        for (size_t k = 0; k < ids.size(); ++k) {
            std::vector<double> z(uv.begin() + 4 * k, uv.begin() + 4 * k + 4);
            auto it = M.find(ids[k]);
            if (it == M.end()) { ModelFeat f; f.birth = births++; f.obs.push_back({cam, z}); M[ids[k]] = f; }
            else {
                ++tracked_m;                                 // msckf.py:437-440: the id is in map_server
                if (!it->second.obs.empty() && it->second.obs.back().first == cam) it->second.obs.back().second = z;
                else it->second.obs.push_back({cam, z});
            }
        }
        (void)before;
        cams.push_back(cam);
        // ---- store
        int tracked = -1;
        const int slot = avs_add_frame(T, S, ids.data(), uv.data(), (int)ids.size(), &tracked);
        if (slot < 0) { printf("FAIL add_frame %d at frame %d\n", slot, t); return 1; }
        if (S.hdr[AVS_LIVE] != (int)M.size()) { printf("FAIL live %d vs %zu at frame %d\n", S.hdr[AVS_LIVE], M.size(), t); return 1; }
        if (tracked != tracked_m) { printf("FAIL tracked %d vs %lld at frame %d\n", tracked, tracked_m, t); return 1; }
        if (S.hdr[AVS_N_ORDER] != (int)cams.size()) { printf("FAIL window size\n"); return 1; }
        // ---- lost features: not observed in this frame, in dict (birth) order
        std::vector<std::pair<long long, long long>> lost_m, inv_m;          // (birth, id)
        for (auto& kv : M) if (kv.second.obs.back().first != cam) (kv.second.obs.size() < 3 ? inv_m : lost_m).push_back({kv.second.birth, kv.first});
        std::sort(lost_m.begin(), lost_m.end());
        int nc = 0, ni = 0;
        avs_select_lost(T, S, cand.data(), &nc, inval.data(), &ni);
        if (nc != (int)lost_m.size() || ni != (int)inv_m.size()) { printf("FAIL lost count %d/%zu invalid %d/%zu at frame %d\n", nc, lost_m.size(), ni, inv_m.size(), t); return 1; }
        for (int k = 0; k < nc; ++k) {
            const int fs = cand[k];
            const ModelFeat& f = M[lost_m[k].second];
            if (S.m_id[fs] != lost_m[k].second || S.m_nobs[fs] != (int)f.obs.size() || S.m_birth[fs] != f.birth) { printf("FAIL lost[%d] id %lld/%lld nobs %d/%zu\n", k, S.m_id[fs], lost_m[k].second, S.m_nobs[fs], f.obs.size()); return 1; }
            const int no = avs_collect(S, fs, ocam.data(), oz.data());
            if (no != (int)f.obs.size()) { printf("FAIL collect count\n"); return 1; }
            for (int q = 0; q < no; ++q) {
                if (cams[ocam[q]] != f.obs[q].first) { printf("FAIL obs camera\n"); return 1; }
                for (int e = 0; e < 4; ++e) if (oz[4 * q + e] != f.obs[q].second[e]) { printf("FAIL obs value\n"); return 1; }
                ++checks;
            }
        }
        {
            std::vector<long long> a, b;
            for (int k = 0; k < ni; ++k) a.push_back(S.m_id[inval[k]]);
            for (auto& p : inv_m) b.push_back(p.second);
            std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end());
            if (a != b) { printf("FAIL invalid set at frame %d\n", t); return 1; }
        }
        avs_erase(T, S, cand.data(), nc);
        avs_erase(T, S, inval.data(), ni);
        for (auto& p : lost_m) M.erase(p.second);
        for (auto& p : inv_m) M.erase(p.second);
        if (S.hdr[AVS_LIVE] != (int)M.size()) { printf("FAIL live after erase\n"); return 1; }
        // ---- camera pruning once the window is full: two cameras (random, never the newest), candidates = seen from both
        if ((int)cams.size() >= max_cam) {
            int i0 = (int)(rng() % (cams.size() - 1)), i1 = (int)(rng() % (cams.size() - 1));
            if (i0 == i1) i1 = (i0 + 1) % (int)(cams.size() - 1);
            if (i0 > i1) std::swap(i0, i1);
            const long long r0 = cams[i0], r1 = cams[i1];
            std::vector<std::pair<long long, long long>> cand_m;
            for (auto& kv : M) {
                bool a = false, b = false;
                for (auto& o : kv.second.obs) { a |= o.first == r0; b |= o.first == r1; }
                if (a && b) cand_m.push_back({kv.second.birth, kv.first});
            }
            std::sort(cand_m.begin(), cand_m.end());
            int np = 0;
            avs_select_prune(T, S, i0, i1, cand.data(), e0.data(), e1.data(), &np);
            if (np != (int)cand_m.size()) { printf("FAIL prune candidates %d vs %zu at frame %d\n", np, cand_m.size(), t); return 1; }
            for (int k = 0; k < np; ++k) {
                const ModelFeat& f = M[cand_m[k].second];
                if (S.m_id[cand[k]] != cand_m[k].second) { printf("FAIL prune candidate order\n"); return 1; }
                if (S.pos_of[avs_lslot(e0[k])] != i0 || S.pos_of[avs_lslot(e1[k])] != i1) { printf("FAIL prune entries\n"); return 1; }
                for (auto& o : f.obs) {
                    const int l = o.first == r0 ? e0[k] : (o.first == r1 ? e1[k] : -1);
                    if (l < 0) continue;
                    const double* z = S.fr_z + ((size_t)avs_lslot(l) * S.cap + avs_lidx(l)) * 4;
                    for (int e = 0; e < 4; ++e) if (z[e] != o.second[e]) { printf("FAIL prune entry value\n"); return 1; }
                    ++checks;
                }
            }
            for (auto& kv : M) {
                auto& ob = kv.second.obs;
                ob.erase(std::remove_if(ob.begin(), ob.end(), [&](const std::pair<long long, std::vector<double>>& o) { return o.first == r0 || o.first == r1; }), ob.end());
            }
            avs_remove_frame(T, S, i1); avs_remove_frame(T, S, i0);
            cams.erase(cams.begin() + i1); cams.erase(cams.begin() + i0);
            if (S.hdr[AVS_N_ORDER] != (int)cams.size()) { printf("FAIL window after prune\n"); return 1; }
            const int ns = S.order[S.hdr[AVS_N_ORDER] - 1];
            for (int i = 0; i < S.fr_n[ns]; ++i) {
                const int fs = S.fr_fslot[(size_t)ns * S.cap + i];
                if (fs < 0) continue;
                const ModelFeat& f = M[S.m_id[fs]];
                const int no = avs_collect(S, fs, ocam.data(), oz.data());
                if (no != (int)f.obs.size()) { printf("FAIL nobs after prune %d %zu\n", no, f.obs.size()); return 1; }
                for (int q = 0; q < no; ++q) if (cams[ocam[q]] != f.obs[q].first || oz[4 * q] != f.obs[q].second[0]) { printf("FAIL obs after prune\n"); return 1; }
                checks += no;
            }
        }
        alive.clear();
        for (auto& kv : M) alive.push_back(kv.first);
        if (rng() % 97 == 0) { avs_clear(T, S); M.clear(); cams.clear(); alive.clear(); }       // online reset
    }
    printf("OK %lld checks\n", checks);
    return 0;
}
