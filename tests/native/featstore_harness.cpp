// CPU harness for FeatStore (uav_airvision_amd/csrc/msckf_batch.inc): drives the frame-major observation store and a plain
// dict-of-dicts model of the reference's map_server (msckf.py:120, 425-441, 614-676, 712-786) through the same random
// message sequences and compares, every frame: live features, the lost features in dict order with their observation
// lists, the camera-pruning candidates in dict order, and the observation counts after the two frames are removed.
// The struct text is spliced in by tests/test_featstore.py at the marker below, so the test always runs the shipped code.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <random>
#include <vector>
FEATSTORE_SRC
struct ModelFeat { long long birth; std::vector<std::pair<long long, std::vector<double>>> obs; };
int main(int argc, char** argv)
{
    const unsigned seed = argc > 1 ? (unsigned)atoi(argv[1]) : 1;
    const int frames = argc > 2 ? atoi(argv[2]) : 200, per = argc > 3 ? atoi(argv[3]) : 60, max_cam = 20;
    std::mt19937 rng(seed);
    FeatStore S;
    std::map<long long, ModelFeat> M;                       // id -> feature (dict order = birth)
    std::vector<long long> cams;                            // camera ids in the window (parallel to S.order)
    long long births = 0, next_id = 1;
    std::vector<long long> alive;                           // ids observed in the previous frame
    long long checks = 0;
    for (int t = 0; t < frames; ++t) {
        // message: most of the previous features survive (shuffled), some die, new ones appear, now and then a duplicate id
        std::vector<long long> ids;
        for (long long id : alive) if (rng() % 100 < 85) ids.push_back(id);
        std::shuffle(ids.begin(), ids.end(), rng);
        const int n_new = per - (int)ids.size() > 0 ? (int)(rng() % (unsigned)(per - (int)ids.size() + 1)) : 0;
        for (int k = 0; k < n_new; ++k) ids.push_back(next_id++);
        if (!ids.empty() && rng() % 7 == 0) ids.push_back(ids[rng() % ids.size()]);
        if (rng() % 31 == 0) ids.clear();                   // a blank frame loses every track
        std::vector<double> uv(4 * ids.size());
        for (double& v : uv) v = (double)(rng() % 100000) / 1000.0;
        const long long cam = t;
        // ---- model
        for (size_t k = 0; k < ids.size(); ++k) {
            std::vector<double> z(uv.begin() + 4 * k, uv.begin() + 4 * k + 4);
            auto it = M.find(ids[k]);
            if (it == M.end()) { ModelFeat f; f.birth = births++; f.obs.push_back({cam, z}); M[ids[k]] = f; }
            else if (!it->second.obs.empty() && it->second.obs.back().first == cam) it->second.obs.back().second = z;
            else it->second.obs.push_back({cam, z});
        }
        cams.push_back(cam);
        // ---- store
        if (S.add_frame(cam, ids.data(), uv.data(), (int)ids.size()) < 0) { printf("FAIL add_frame\n"); return 1; }
        if (S.live != (int)M.size()) { printf("FAIL live %d vs %zu at frame %d\n", S.live, M.size(), t); return 1; }
        // ---- lost features: not observed in this frame, in dict (birth) order
        std::vector<std::pair<long long, long long>> lost_m;          // (birth, id)
        for (auto& kv : M) if (kv.second.obs.back().first != cam) lost_m.push_back({kv.second.birth, kv.first});
        std::sort(lost_m.begin(), lost_m.end());
        std::vector<int> lost_s;
        if (S.order.size() >= 2) {
            const FeatStore::Frame& P = S.fr[S.order[S.order.size() - 2]];
            for (int i = 0; i < P.n; ++i) if (P.fslot[i] >= 0 && P.next[i] < 0) lost_s.push_back(P.fslot[i]);
            std::sort(lost_s.begin(), lost_s.end(), [&](int a, int b) { return S.meta[a].birth < S.meta[b].birth; });
        }
        if (lost_s.size() != lost_m.size()) { printf("FAIL lost count %zu vs %zu at frame %d\n", lost_s.size(), lost_m.size(), t); return 1; }
        for (size_t k = 0; k < lost_s.size(); ++k) {
            const FMeta& m = S.meta[lost_s[k]];
            const ModelFeat& f = M[lost_m[k].second];
            if (m.id != lost_m[k].second || m.nobs != (int)f.obs.size()) { printf("FAIL lost[%zu] id %lld/%lld nobs %d/%zu\n", k, m.id, lost_m[k].second, m.nobs, f.obs.size()); return 1; }
            int oc[FeatStore::NSLOT]; const double* oz[FeatStore::NSLOT];
            const int no = S.collect(m, oc, oz, FeatStore::NSLOT);
            if (no != (int)f.obs.size()) { printf("FAIL collect count\n"); return 1; }
            for (int q = 0; q < no; ++q) {
                if (cams[oc[q]] != f.obs[q].first) { printf("FAIL obs camera\n"); return 1; }
                for (int e = 0; e < 4; ++e) if (oz[q][e] != f.obs[q].second[e]) { printf("FAIL obs value\n"); return 1; }
                ++checks;
            }
        }
        for (int fs : lost_s) S.erase(fs);
        for (auto& p : lost_m) M.erase(p.second);
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
            std::vector<int> cand_s;
            const FeatStore::Frame& A = S.fr[S.order[i0]];
            for (int i = 0; i < A.n; ++i) {
                if (A.fslot[i] < 0) continue;
                int l = A.next[i];
                while (l >= 0 && S.pos_of[FeatStore::lslot(l)] < i1) l = S.fr[FeatStore::lslot(l)].next[FeatStore::lidx(l)];
                if (l >= 0 && S.pos_of[FeatStore::lslot(l)] == i1) cand_s.push_back(A.fslot[i]);
            }
            std::sort(cand_s.begin(), cand_s.end(), [&](int a, int b) { return S.meta[a].birth < S.meta[b].birth; });
            if (cand_s.size() != cand_m.size()) { printf("FAIL prune candidates %zu vs %zu at frame %d\n", cand_s.size(), cand_m.size(), t); return 1; }
            for (size_t k = 0; k < cand_s.size(); ++k) if (S.meta[cand_s[k]].id != cand_m[k].second) { printf("FAIL prune candidate order\n"); return 1; }
            for (auto& kv : M) {
                auto& ob = kv.second.obs;
                ob.erase(std::remove_if(ob.begin(), ob.end(), [&](const std::pair<long long, std::vector<double>>& o) { return o.first == r0 || o.first == r1; }), ob.end());
            }
            S.remove_frame(i1); S.remove_frame(i0);
            cams.erase(cams.begin() + i1); cams.erase(cams.begin() + i0);
            // every live feature: observation list after the removal
            const FeatStore::Frame& N = S.fr[S.newest()];
            for (int i = 0; i < N.n; ++i) {
                const int fs = N.fslot[i];
                if (fs < 0) continue;
                const FMeta& m = S.meta[fs];
                const ModelFeat& f = M[m.id];
                int oc[FeatStore::NSLOT]; const double* oz[FeatStore::NSLOT];
                const int no = S.collect(m, oc, oz, FeatStore::NSLOT);
                if (no != (int)f.obs.size() || m.nobs != no) { printf("FAIL nobs after prune %d %d %zu\n", no, m.nobs, f.obs.size()); return 1; }
                for (int q = 0; q < no; ++q) if (cams[oc[q]] != f.obs[q].first || oz[q][0] != f.obs[q].second[0]) { printf("FAIL obs after prune\n"); return 1; }
                checks += no;
            }
        }
        alive.clear();
        for (auto& kv : M) alive.push_back(kv.first);
        if (rng() % 97 == 0) { S.clear(); M.clear(); cams.clear(); alive.clear(); }       // online reset
    }
    printf("OK %lld checks\n", checks);
    return 0;
}
