// CPU ORACLE for the CifCaf decode path -- TEST INFRASTRUCTURE ONLY.
//
// A plain C++17 restatement (no libtorch, raw pointers, C ABI) of the algorithm
// of the reference decoder under /root/reference/src/openpifpaf/csrc.  It exists
// to CHECK the HIP path; nothing in openpifpaf_amd/ may link, load or call it.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
//
// Parity pinning: the reference's own tests hold no golden vectors for this path
// (SURVEY.md section 8c), so this restatement is pinned against the reference
// itself: oracle/_ref/openpifpaf_ref.so (the unmodified reference sources built
// by oracle/build_ref.py) on seeded synthetic fields, see
// tests/test_oracle_vs_reference.py and the committed fixtures in tests/golden/.
//
// Semantics are those of a FRESH reference decoder instance (CifHr revision 1.0,
// Occupancy revision 1): the reference's results drift with the number of calls
// made on one instance (cif_hr.cpp:84-86,115-120), so "fresh" is the definition.
//
// Every function cites the reference file:line it follows.  Float/double
// promotion is reproduced operation by operation; build WITHOUT -march=native /
// -ffast-math so that no FMA contraction happens (the reference wheels have none).

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <numeric>
#include <queue>
#include <unordered_set>
#include <utility>
#include <vector>

namespace {

struct Params {                      // defaults = the reference's static members
    double cif_threshold = 0.3;      // cif_hr.cpp:14
    int64_t cifhr_neighbors = 16;    // cif_hr.cpp:13
    double seed_threshold = 0.2;     // cif_seeds.cpp:11
    double caf_threshold = 0.3;      // caf_scored.cpp:11
    double cif_floor = 0.1;          // cifcaf.cpp:153
    double keypoint_threshold = 0.15;      // cifcaf.cpp:20
    double keypoint_threshold_rel = 0.5;   // cifcaf.cpp:21
    double nms_suppression = 0.00001;      // nms_keypoints.cpp:12
    double nms_instance_threshold = 0.15;  // nms_keypoints.cpp:13
    double nms_keypoint_threshold = 0.15;  // nms_keypoints.cpp:14
    double force_complete_caf_th = 0.001;  // cifcaf.cpp:24
    double occupancy_reduction = 2.0;      // cifcaf.hpp:103
    double occupancy_min_scale = 4.0;      // cifcaf.hpp:103
    int32_t greedy = 0;              // cifcaf.cpp:19
    int32_t reverse_match = 1;       // cifcaf.cpp:22
    int32_t force_complete = 0;      // cifcaf.cpp:23
    int32_t block_joints = 0;        // cifcaf.cpp:18
    int32_t ablation_cifseeds_nms = 0;        // cif_seeds.cpp:13
    int32_t ablation_cifseeds_no_rescore = 0; // cif_seeds.cpp:14
    int32_t ablation_caf_no_rescore = 0;      // caf_scored.cpp:12
    int32_t ablation_cifhr_skip = 0;          // cif_hr.cpp:15
};

inline int64_t clamp64(int64_t v, int64_t lo, int64_t hi) { return std::min(std::max(v, lo), hi); }

// Order of seeds with EXACTLY equal scores.  0 = whatever libstdc++'s unstable std::sort leaves (the
// reference, cif_seeds.cpp:93-99: unspecified by the language, deterministic for one library build);
// 1 = cell index ascending (the total order the HIP path sorts by); 2 = cell index descending.
// Test-only knob (tools/tie_study.py, tests/test_tie_rule.py): it lets the CPU suite measure how often
// the tie order changes a decode on quantised (bf16) fields.
int g_seed_tie_rule = 0;

// ---------------------------------------------------------------- CifHr
// High-resolution confidence map, stored exactly like the reference buffer of a
// fresh instance: 0.0 where never touched, otherwise revision(=1) + accumulated.
struct HiRes {
    float* data; int64_t F, H, W;    // H, W are high-res sizes
    double revision = 1.0;           // cif_hr.cpp:114 after the first reset()
    float& at(int64_t f, int64_t y, int64_t x) { return data[(f * H + y) * W + x]; }
};

// cif_hr.cpp:18-25
inline float approx_exp(float x) {
    if (x > 2.0 || x < -2.0) return 0.0f;
    x = 1.0 + x / 8.0;               // double arithmetic, rounded to float once
    x *= x; x *= x; x *= x;
    return x;
}

// cif_hr.cpp:58-89
void add_gauss(HiRes& hr, int64_t f, float v, float x, float y, float sigma, float truncate) {
    const int64_t minx = clamp64(int64_t(x - truncate * sigma), 0, hr.W - 1);
    const int64_t miny = clamp64(int64_t(y - truncate * sigma), 0, hr.H - 1);
    const int64_t maxx = clamp64(int64_t(x + truncate * sigma + 1), minx + 1, hr.W);
    const int64_t maxy = clamp64(int64_t(y + truncate * sigma + 1), miny + 1, hr.H);
    const float sigma2 = sigma * sigma;
    const float trunc2_sigma2 = truncate * truncate * sigma2;
    for (int64_t xx = minx; xx < maxx; xx++) {
        const float dx2 = (xx - x) * (xx - x);
        for (int64_t yy = miny; yy < maxy; yy++) {
            const float dy2 = (yy - y) * (yy - y);
            if (dx2 + dy2 > trunc2_sigma2) continue;
            float vv;
            if (dx2 < 0.25 && dy2 < 0.25) vv = v;
            else vv = v * approx_exp(-0.5 * (dx2 + dy2) / sigma2);
            float& e = hr.at(f, yy, xx);
            e = fmaxf(e, hr.revision) + vv;
            e = fminf(e, hr.revision + 1.0);
        }
    }
}

// cif_hr.cpp:28-55 ; field layout [F,5,H,W]
void cifhr_accumulate(HiRes& hr, const float* cif, int64_t F, int64_t H, int64_t W, int64_t stride,
                      double min_scale, double factor, const Params& p) {
    if (p.ablation_cifhr_skip) return;
    const int64_t HW = H * W;
    const float min_scale_f = min_scale / stride;
    for (int64_t f = 0; f < F; f++) {
        const float* plane = cif + f * 5 * HW;
        for (int64_t j = 0; j < H; j++) for (int64_t i = 0; i < W; i++) {
            const int64_t o = j * W + i;
            const float v = plane[1 * HW + o];
            if (v < p.cif_threshold) continue;
            const float scale = plane[4 * HW + o];
            if (scale < min_scale_f) continue;
            const float x = plane[2 * HW + o] * stride;
            const float y = plane[3 * HW + o] * stride;
            const float sigma = fmaxf(1.0, 0.5 * scale * stride);
            add_gauss(hr, f, v / p.cifhr_neighbors * factor, x, y, sigma, 1.0);
        }
    }
}

// cif_seeds.cpp:17-30 and caf_scored.cpp:15-26 (identical bodies)
inline float cifhr_value(const HiRes& hr, int64_t f, float x, float y, float default_value) {
    const float max_x = static_cast<float>(hr.W) - 0.51;
    const float max_y = static_cast<float>(hr.H) - 0.51;
    if (f >= hr.F || x < -0.49 || y < -0.49 || x > max_x || y > max_y) return default_value;
    const float value = hr.data[(f * hr.H + int64_t(y + 0.5)) * hr.W + int64_t(x + 0.5)] - hr.revision;
    if (value < 0.0) return default_value;
    return value;
}

// ---------------------------------------------------------------- CifSeeds
struct Seed { int64_t f; float v, x, y, s; };   // cif_seeds.hpp:17-23

// cif_seeds.cpp:33-66 (+ the 3x3 max-pool gate of :35-40,49-51) and :93-99
std::vector<Seed> cif_seeds(const HiRes& hr, const float* cif, int64_t F, int64_t H, int64_t W,
                            int64_t stride, const Params& p) {
    std::vector<Seed> seeds;
    const int64_t HW = H * W;
    for (int64_t f = 0; f < F; f++) {
        const float* plane = cif + f * 5 * HW;
        for (int64_t j = 0; j < H; j++) for (int64_t i = 0; i < W; i++) {
            const int64_t o = j * W + i;
            float c = plane[1 * HW + o];
            if (c < p.seed_threshold) continue;
            if (p.ablation_cifseeds_nms) {   // max_pool2d(conf, 3, 1, 1), -inf padding
                float m = c;
                for (int64_t dj = -1; dj <= 1; dj++) for (int64_t di = -1; di <= 1; di++) {
                    const int64_t jj = j + dj, ii = i + di;
                    if (jj < 0 || jj >= H || ii < 0 || ii >= W) continue;
                    m = std::max(m, plane[1 * HW + jj * W + ii]);
                }
                if (c < m) continue;
            }
            const float x = plane[2 * HW + o] * stride;
            const float y = plane[3 * HW + o] * stride;
            if (!p.ablation_cifseeds_no_rescore)
                c = 0.9 * cifhr_value(hr, f, x, y, -1.0) + 0.1 * c;
            if (c < p.seed_threshold) continue;
            const float s = plane[4 * HW + o] * stride;
            seeds.push_back(Seed{f, c, x, y, s});
        }
    }
    if (g_seed_tie_rule == 0) {
        std::sort(seeds.begin(), seeds.end(), [](const Seed& a, const Seed& b) { return a.v > b.v; });
    } else {                         // seeds were pushed in cell-index order: a stable sort keeps it among equals
        if (g_seed_tie_rule == 2) std::reverse(seeds.begin(), seeds.end());
        std::stable_sort(seeds.begin(), seeds.end(), [](const Seed& a, const Seed& b) { return a.v > b.v; });
    }
    return seeds;
}

// ---------------------------------------------------------------- CifDet pieces
// cif_hr.cpp:124-150 ; field layout [F,6,H,W]: 0 unused, 1 conf, 2 x, 3 y, 4 w, 5 h
void cifdethr_accumulate(HiRes& hr, const float* field, int64_t F, int64_t H, int64_t W, int64_t stride,
                         double min_scale, double factor, const Params& p) {
    const int64_t HW = H * W;
    const float min_scale_f = min_scale / stride;
    for (int64_t f = 0; f < F; f++) {
        const float* plane = field + f * 6 * HW;
        for (int64_t j = 0; j < H; j++) for (int64_t i = 0; i < W; i++) {
            const int64_t o = j * W + i;
            const float v = plane[1 * HW + o];
            if (v < p.cif_threshold) continue;
            const float w = plane[4 * HW + o], h = plane[5 * HW + o];
            if (w < min_scale_f || h < min_scale_f) continue;
            const float x = plane[2 * HW + o] * stride;
            const float y = plane[3 * HW + o] * stride;
            const float sigma = fmaxf(1.0, 0.1 * fmin(w, h) * stride);
            add_gauss(hr, f, v / p.cifhr_neighbors * factor, x, y, sigma, 1.0);
        }
    }
}

struct DetSeed { int64_t c; float v, x, y, w, h; };   // cif_seeds.hpp:26-32

// cif_seeds.cpp:69-90,117-123
std::vector<DetSeed> cifdet_seeds(const HiRes& hr, const float* field, int64_t F, int64_t H, int64_t W,
                                  int64_t stride, const Params& p) {
    std::vector<DetSeed> seeds;
    const int64_t HW = H * W;
    for (int64_t f = 0; f < F; f++) {
        const float* plane = field + f * 6 * HW;
        for (int64_t j = 0; j < H; j++) for (int64_t i = 0; i < W; i++) {
            const int64_t o = j * W + i;
            const float c = plane[1 * HW + o];
            if (c < p.seed_threshold) continue;
            const float x = plane[2 * HW + o] * stride;
            const float y = plane[3 * HW + o] * stride;
            const float v = 0.9 * cifhr_value(hr, f, x, y, -1.0) + 0.1 * c;
            if (v < p.seed_threshold) continue;
            const float w = plane[4 * HW + o] * stride;
            const float h = plane[5 * HW + o] * stride;
            seeds.push_back(DetSeed{f, v, x, y, w, h});
        }
    }
    std::sort(seeds.begin(), seeds.end(), [](const DetSeed& a, const DetSeed& b) { return a.v > b.v; });
    return seeds;
}

// ---------------------------------------------------------------- CafScored
struct Assoc { float c, x1, y1, x2, y2, s1, s2; };   // caf_scored.hpp:17-33
typedef std::vector<std::vector<Assoc>> AssocLists;

// caf_scored.cpp:29-83 ; field layout [A,8,H,W]; skeleton 0-based [A,2]
void caf_scored(const HiRes& hr, const float* caf, int64_t A, int64_t H, int64_t W, int64_t stride,
                const int64_t* skeleton, double score_th, double cif_floor, bool no_rescore,
                AssocLists& forward, AssocLists& backward) {
    forward.assign(A, {});
    backward.assign(A, {});
    const int64_t HW = H * W;
    for (int64_t a = 0; a < A; a++) {
        const float* plane = caf + a * 8 * HW;
        for (int64_t j = 0; j < H; j++) for (int64_t i = 0; i < W; i++) {
            const int64_t o = j * W + i;
            const float c = plane[1 * HW + o];
            if (c < score_th) continue;
            Assoc fw{c, plane[2 * HW + o] * stride, plane[3 * HW + o] * stride,
                        plane[4 * HW + o] * stride, plane[5 * HW + o] * stride,
                        plane[6 * HW + o] * stride, plane[7 * HW + o] * stride};
            Assoc bw{c, fw.x2, fw.y2, fw.x1, fw.y1, fw.s2, fw.s1};
            if (!no_rescore) {
                const float fhr = cifhr_value(hr, skeleton[2 * a + 1], fw.x2, fw.y2, 0.0);
                const float bhr = cifhr_value(hr, skeleton[2 * a + 0], bw.x2, bw.y2, 0.0);
                fw.c = fw.c * (cif_floor + (1.0 - cif_floor) * fhr);
                bw.c = bw.c * (cif_floor + (1.0 - cif_floor) * bhr);
            }
            if (fw.c > score_th) forward[a].push_back(fw);
            if (bw.c > score_th) backward[a].push_back(bw);
        }
    }
}

// ---------------------------------------------------------------- Occupancy
// occupancy.cpp:13-79, as a plain byte map (the reference's int16 + revision is
// a lazy-clear device; a fresh map per phase has the same meaning).
struct Occupancy {
    std::vector<uint8_t> map; int64_t F = 0, H = 0, W = 0;
    double reduction, min_scale_reduced;
    Occupancy(double reduction_, double min_scale) : reduction(reduction_), min_scale_reduced(min_scale / reduction_) {}
    void reset(int64_t F_, int64_t hr_h, int64_t hr_w) {        // occupancy.cpp:46-68
        F = F_; H = static_cast<int64_t>(hr_h / reduction) + 1; W = static_cast<int64_t>(hr_w / reduction) + 1;
        map.assign(F * H * W, 0);
    }
    void clear() { std::fill(map.begin(), map.end(), 0); }      // occupancy.cpp:71-77
    void set(int64_t f, double x, double y, double sigma) {     // occupancy.cpp:13-29
        if (reduction != 1.0) { x /= reduction; y /= reduction; sigma = fmax(min_scale_reduced, sigma / reduction); }
        const int64_t minx = clamp64(int64_t(x - sigma), 0, W - 1);
        const int64_t miny = clamp64(int64_t(y - sigma), 0, H - 1);
        const int64_t maxx = clamp64(int64_t(x + sigma), minx + 1, W);
        const int64_t maxy = clamp64(int64_t(y + sigma), miny + 1, H);
        for (int64_t yy = miny; yy < maxy; yy++)
            std::memset(&map[(f * H + yy) * W + minx], 1, maxx - minx);
    }
    bool get(int64_t f, double x, double y) const {             // occupancy.cpp:32-43
        if (f >= F) return true;
        if (reduction != 1.0) { x /= reduction; y /= reduction; }
        const int64_t xi = clamp64(int64_t(x), 0, W - 1);
        const int64_t yi = clamp64(int64_t(y), 0, H - 1);
        return map[(f * H + yi) * W + xi] != 0;
    }
};

// ---------------------------------------------------------------- CifCaf
struct Joint { double v = 0, x = 0, y = 0, s = 0; };           // cifcaf.hpp:21-28
struct Ann { std::vector<Joint> joints; int64_t id = -1; };    // cifcaf.hpp:31-40

// cifcaf.cpp:32-103 ; `rows` is an [n,7] (c,x1,y1,x2,y2,s1,s2) list
Joint connection_blend(const float* rows, int64_t n, double x, double y, double xy_scale,
                       double filter_sigmas, bool only_max) {
    xy_scale = fmax(xy_scale, 0.5);
    const float sigma_filter = filter_sigmas * xy_scale / 2.0;
    const float sigma2 = 0.25 * xy_scale * xy_scale;
    int64_t i1 = 0, i2 = 0;
    float score_1 = 0.0f, score_2 = 0.0f;
    for (int64_t i = 0; i < n; i++) {
        const float* r = rows + 7 * i;
        if (r[1] < x - sigma_filter) continue;
        if (r[1] > x + sigma_filter) continue;
        if (r[2] < y - sigma_filter) continue;
        if (r[2] > y + sigma_filter) continue;
        const float d2 = std::pow(r[1] - x, 2) + std::pow(r[2] - y, 2);
        const float score = std::exp(-0.5 * d2 / sigma2) * r[0];
        if (score >= score_1) { i2 = i1; score_2 = score_1; i1 = i; score_1 = score; }
        else if (score > score_2) { i2 = i; score_2 = score; }
    }
    Joint out;
    if (score_1 == 0.0) return out;
    const float e1[3] = {rows[7 * i1 + 3], rows[7 * i1 + 4], fmaxf(0.0f, rows[7 * i1 + 6])};
    if (only_max) { out.v = score_1; out.x = e1[0]; out.y = e1[1]; out.s = e1[2]; return out; }
    if (score_2 < 0.01 || score_2 < 0.5 * score_1) { out.v = 0.5 * score_1; out.x = e1[0]; out.y = e1[1]; out.s = e1[2]; return out; }
    const float e2[3] = {rows[7 * i2 + 3], rows[7 * i2 + 4], fmaxf(0.0f, rows[7 * i2 + 6])};
    const float blend_d2 = std::pow(e1[0] - e2[0], 2) + std::pow(e1[1] - e2[1], 2);
    if (blend_d2 > std::pow(e1[2], 2) / 4.0) { out.v = 0.5 * score_1; out.x = e1[0]; out.y = e1[1]; out.s = e1[2]; return out; }
    out.v = 0.5 * (score_1 + score_2);
    out.x = (score_1 * e1[0] + score_2 * e2[0]) / (score_1 + score_2);
    out.y = (score_1 * e1[1] + score_2 * e2[1]) / (score_1 + score_2);
    out.s = (score_1 * e1[2] + score_2 * e2[2]) / (score_1 + score_2);
    return out;
}

struct FrontierEntry { float max_score; Joint joint; int64_t start_i, end_i; };   // cifcaf.hpp:51-61
struct FrontierLess { bool operator()(const FrontierEntry& a, const FrontierEntry& b) const { return a.max_score < b.max_score; } };  // cifcaf.cpp:27-29
struct PairHash {   // cifcaf.hpp:70-76
    size_t operator()(const std::pair<int64_t, int64_t>& p) const noexcept {
        return std::hash<int64_t>{}(p.first) ^ (std::hash<int64_t>{}(p.second) << 1);
    }
};

struct Decoder {
    int64_t K, A; const int64_t* skeleton; Params p; int64_t occ_fields = 0;
    std::priority_queue<FrontierEntry, std::vector<FrontierEntry>, FrontierLess> frontier;
    std::unordered_set<std::pair<int64_t, int64_t>, PairHash> in_frontier;

    static const float* rows(const std::vector<Assoc>& l) { return reinterpret_cast<const float*>(l.data()); }

    void frontier_add_from(const Ann& ann, int64_t start_i) {   // cifcaf.cpp:316-346
        const float max_score = sqrt(ann.joints[start_i].v);
        for (int64_t a = 0; a < A; a++) {
            const int64_t p0 = skeleton[2 * a], p1 = skeleton[2 * a + 1];
            int64_t other;
            if (p0 == start_i) other = p1; else if (p1 == start_i) other = p0; else continue;
            if (ann.joints[other].v > 0.0) continue;
            if (in_frontier.count({start_i, other})) continue;
            frontier.push(FrontierEntry{max_score, Joint(), start_i, other});
            in_frontier.emplace(start_i, other);
        }
    }

    Joint connection_value(const Ann& ann, const AssocLists& fwd, const AssocLists& bwd,   // cifcaf.cpp:349-411
                           int64_t start_i, int64_t end_i, bool reverse_match_, double filter_sigmas) {
        int64_t a = 0; bool forward = true;
        for (; a < A; a++) {
            if (skeleton[2 * a] == start_i && skeleton[2 * a + 1] == end_i) { forward = true; break; }
            if (skeleton[2 * a + 1] == start_i && skeleton[2 * a] == end_i) { forward = false; break; }
        }
        const std::vector<Assoc>& caf_f = forward ? fwd[a] : bwd[a];
        const std::vector<Assoc>& caf_b = forward ? bwd[a] : fwd[a];
        const Joint& st = ann.joints[start_i];
        Joint nj = connection_blend(rows(caf_f), caf_f.size(), st.x, st.y, st.s, filter_sigmas, false);
        if (nj.v == 0.0) return nj;
        nj.v = sqrt(nj.v * st.v);
        if (nj.v < p.keypoint_threshold || nj.v < st.v * p.keypoint_threshold_rel) { nj.v = 0.0; return nj; }
        if (p.reverse_match && reverse_match_ && start_i < occ_fields) {
            Joint rj = connection_blend(rows(caf_b), caf_b.size(), nj.x, nj.y, nj.s, filter_sigmas, false);
            if (rj.v == 0.0) { nj.v = 0.0; return nj; }
            if (fabs(st.x - rj.x) + fabs(st.y - rj.y) > st.s) { nj.v = 0.0; return nj; }
        }
        return nj;
    }

    void grow(Ann* ann, const AssocLists& fwd, const AssocLists& bwd, bool reverse_match_, double filter_sigmas) {  // cifcaf.cpp:265-313
        while (!frontier.empty()) frontier.pop();
        in_frontier.clear();
        for (int64_t j = 0; j < K; j++) if (ann->joints[j].v != 0.0) frontier_add_from(*ann, j);
        while (!frontier.empty()) {
            FrontierEntry e = frontier.top();
            frontier.pop();
            if (ann->joints[e.end_i].v > 0.0) continue;
            if (e.joint.v == 0.0) {
                Joint nj = connection_value(*ann, fwd, bwd, e.start_i, e.end_i, reverse_match_, filter_sigmas);
                if (nj.v == 0.0) continue;     // block_joints branch is a no-op, cifcaf.cpp:290-296
                if (!p.greedy) { frontier.push(FrontierEntry{float(nj.v), nj, e.start_i, e.end_i}); continue; }
                e.max_score = nj.v; e.joint = nj;
            }
            ann->joints[e.end_i] = e.joint;
            frontier_add_from(*ann, e.end_i);
        }
    }

    void flood_fill(Ann* ann) {   // cifcaf.cpp:429-449
        while (!frontier.empty()) frontier.pop();
        in_frontier.clear();
        for (int64_t j = 0; j < K; j++) if (ann->joints[j].v != 0.0) frontier_add_from(*ann, j);
        while (!frontier.empty()) {
            FrontierEntry e = frontier.top();
            frontier.pop();
            if (ann->joints[e.end_i].v > 0.0) continue;
            ann->joints[e.end_i] = ann->joints[e.start_i];
            ann->joints[e.end_i].v = 0.00001;
            frontier_add_from(*ann, e.end_i);
        }
    }
};

// nms_keypoints.hpp:25-32 : double accumulator narrowed to float at every step
double uniform_score(const Ann& ann) {
    double acc = 0.0;
    for (const Joint& j : ann.joints) { float i = acc; acc = i + j.v; }
    return acc / ann.joints.size();
}

// nms_keypoints.cpp:17-70
void nms_keypoints(Occupancy* occ, std::vector<Ann>* anns, const Params& p) {
    occ->clear();
    auto by_score = [](const Ann& a, const Ann& b) { return uniform_score(a) > uniform_score(b); };
    std::sort(anns->begin(), anns->end(), by_score);
    for (Ann& ann : *anns) {
        int64_t f = -1;
        for (Joint& j : ann.joints) {
            f++;
            if (f >= occ->F) break;
            if (j.v == 0.0) continue;
            if (occ->get(f, j.x, j.y)) j.v *= p.nms_suppression;
            else occ->set(f, j.x, j.y, j.s);
        }
    }
    for (Ann& ann : *anns) for (Joint& j : ann.joints) if (!(j.v > p.nms_keypoint_threshold)) j.v = 0.0;
    anns->erase(std::remove_if(anns->begin(), anns->end(),
                               [&](const Ann& a) { return uniform_score(a) < p.nms_instance_threshold; }),
                anns->end());
    std::sort(anns->begin(), anns->end(), by_score);
}

void flatten(const AssocLists& lists, int64_t cap, float* out, int32_t* counts) {
    for (size_t a = 0; a < lists.size(); a++) {
        counts[a] = int32_t(lists[a].size());
        if (out) std::memcpy(out + a * cap * 7, lists[a].data(), lists[a].size() * sizeof(Assoc));
    }
}

}  // namespace

extern "C" {

typedef Params oracle_params;

void oracle_default_params(oracle_params* p) { *p = Params(); }
void oracle_set_seed_tie_rule(int rule) { g_seed_tie_rule = rule; }

// The permutation std::sort leaves a sequence of seed scores in (cif_seeds.cpp:94: Seed structs, comparator a.v > b.v):
// perm[k] = original position of the element that ends up at rank k.  tests/test_tie_order_model.py checks the
// formulation the HIP tie pass uses (stop pairing, per-segment stable placement) against it without a GPU.
void oracle_sorted_seed_order(const float* v, int64_t n, int64_t* perm) {
    std::vector<Seed> seeds((size_t)n);
    for (int64_t i = 0; i < n; i++) seeds[(size_t)i] = Seed{i, v[i], 0.f, 0.f, 0.f};
    std::sort(seeds.begin(), seeds.end(), [](const Seed& a, const Seed& b) { return a.v > b.v; });
    for (int64_t i = 0; i < n; i++) perm[i] = seeds[(size_t)i].f;
}
int oracle_get_seed_tie_rule(void) { return g_seed_tie_rule; }

// cifhr [F,Hhr,Whr] must be zero-filled by the caller; on return it holds the
// raw reference buffer content at revision 1 (0 = untouched, else 1 + value).
void oracle_cifhr_accumulate(const float* cif, int64_t F, int64_t H, int64_t W, int64_t stride,
                             double min_scale, double factor, const oracle_params* p, float* cifhr) {
    HiRes hr{cifhr, F, (H - 1) * stride + 1, (W - 1) * stride + 1};
    cifhr_accumulate(hr, cif, F, H, W, stride, min_scale, factor, *p);
}

// returns the number of seeds; writes at most `cap`
int64_t oracle_cifseeds(const float* cif, int64_t F, int64_t H, int64_t W, int64_t stride,
                        const float* cifhr, const oracle_params* p,
                        int64_t* out_f, float* out_vxys, int64_t cap) {
    HiRes hr{const_cast<float*>(cifhr), F, (H - 1) * stride + 1, (W - 1) * stride + 1};
    std::vector<Seed> s = cif_seeds(hr, cif, F, H, W, stride, *p);
    for (int64_t i = 0; i < int64_t(s.size()) && i < cap; i++) {
        out_f[i] = s[i].f;
        out_vxys[4 * i + 0] = s[i].v; out_vxys[4 * i + 1] = s[i].x;
        out_vxys[4 * i + 2] = s[i].y; out_vxys[4 * i + 3] = s[i].s;
    }
    return int64_t(s.size());
}

// CifDetSeeds(cifhr, revision).fill + get, cif_seeds.cpp:69-90,117-139; out_vxywh [cap,5]
int64_t oracle_cifdetseeds(const float* field, int64_t F, int64_t H, int64_t W, int64_t stride,
                           const float* cifhr, const oracle_params* p,
                           int64_t* out_f, float* out_vxywh, int64_t cap) {
    HiRes hr{const_cast<float*>(cifhr), F, (H - 1) * stride + 1, (W - 1) * stride + 1};
    std::vector<DetSeed> s = cifdet_seeds(hr, field, F, H, W, stride, *p);
    for (int64_t i = 0; i < int64_t(s.size()) && i < cap; i++) {
        out_f[i] = s[i].c;
        out_vxywh[5 * i + 0] = s[i].v; out_vxywh[5 * i + 1] = s[i].x; out_vxywh[5 * i + 2] = s[i].y;
        out_vxywh[5 * i + 3] = s[i].w; out_vxywh[5 * i + 4] = s[i].h;
    }
    return int64_t(s.size());
}

// fwd/bwd: [A, cap, 7] row lists; n_fwd/n_bwd: [A]
void oracle_cafscored(const float* caf, int64_t A, int64_t H, int64_t W, int64_t stride,
                      const float* cifhr, int64_t F, int64_t cif_H, int64_t cif_W, int64_t cif_stride,
                      const int64_t* skeleton, double score_th, double cif_floor, const oracle_params* p,
                      int64_t cap, float* fwd, int32_t* n_fwd, float* bwd, int32_t* n_bwd) {
    HiRes hr{const_cast<float*>(cifhr), F, (cif_H - 1) * cif_stride + 1, (cif_W - 1) * cif_stride + 1};
    AssocLists f, b;
    caf_scored(hr, caf, A, H, W, stride, skeleton, score_th >= 0.0 ? score_th : p->caf_threshold, cif_floor,
               p->ablation_caf_no_rescore, f, b);
    flatten(f, cap, fwd, n_fwd);
    flatten(b, cap, bwd, n_bwd);
}

// out: (x, y, s, v) like the exported op, cifcaf.cpp:105-113
void oracle_grow_connection_blend(const float* rows, int64_t n, double x, double y, double s,
                                  double filter_sigmas, int32_t only_max, double* out) {
    Joint j = connection_blend(rows, n, x, y, s, filter_sigmas, only_max != 0);
    out[0] = j.x; out[1] = j.y; out[2] = j.s; out[3] = j.v;
}

// The whole decode, cifcaf.cpp:126-262.  Returns the number of annotations
// (which may exceed `cap`; only the first `cap` are written).
// out: [cap, K, 4] (v,x,y,s) float32 ; out_ids: [cap]
// initial: optional [n_initial, K, 4] (v,x,y,s) + ids (may be NULL / 0)
// cifhr_out: optional [F,Hhr,Whr] raw buffer copy (may be NULL)
// K = n_keypoints of the decoder object (cifcaf.hpp:99-107), F = fields of the CIF tensor.  K > F is the
// reference's tracking setup (decoder/tracking_pose.py:47-80): occupancy, seeds and the high-res map cover
// the F current-frame joints only (cifcaf.cpp:173), annotations carry K joints.
int64_t oracle_cifcaf_decode_k(const float* cif, int64_t F, int64_t H, int64_t W, int64_t cif_stride,
                               const float* caf, int64_t A, int64_t caf_H, int64_t caf_W, int64_t caf_stride,
                               const int64_t* skeleton, const oracle_params* params,
                               const float* initial, const int64_t* initial_ids, int64_t n_initial,
                               int64_t cap, float* out, int64_t* out_ids, float* cifhr_out, int64_t K) {
    const Params& p = *params;
    const int64_t hh = (H - 1) * cif_stride + 1, hw = (W - 1) * cif_stride + 1;
    std::vector<float> buffer(size_t(F) * hh * hw, 0.0f);
    HiRes hr{buffer.data(), F, hh, hw};
    cifhr_accumulate(hr, cif, F, H, W, cif_stride, 0.0, 1.0, p);              // cifcaf.cpp:140-142
    if (cifhr_out) std::memcpy(cifhr_out, buffer.data(), buffer.size() * sizeof(float));

    std::vector<Seed> seeds = cif_seeds(hr, cif, F, H, W, cif_stride, p);     // :144-146
    AssocLists fwd, bwd;
    caf_scored(hr, caf, A, caf_H, caf_W, caf_stride, skeleton, p.caf_threshold, p.cif_floor,
               p.ablation_caf_no_rescore, fwd, bwd);                           // :153-161

    Occupancy occ(p.occupancy_reduction, p.occupancy_min_scale);
    occ.reset(F, hh, hw);                                                      // :173
    Decoder dec{K, A, skeleton, p, occ.F};
    std::vector<Ann> anns;

    auto mark = [&](const Ann& ann) {
        for (int64_t of = 0; of < occ.F; of++) {
            const Joint& j = ann.joints[of];
            if (j.v == 0.0) continue;
            occ.set(of, j.x, j.y, j.s);
        }
    };

    for (int64_t n = 0; n < n_initial; n++) {                                  // :177-202
        Ann ann; ann.joints.resize(K); ann.id = initial_ids ? initial_ids[n] : -1;
        for (int64_t k = 0; k < K; k++) {
            const float* r = initial + (n * K + k) * 4;
            ann.joints[k].v = r[0]; ann.joints[k].x = r[1]; ann.joints[k].y = r[2]; ann.joints[k].s = r[3];
        }
        dec.grow(&ann, fwd, bwd, true, 1.0);
        mark(ann);
        anns.push_back(ann);
    }

    for (const Seed& sd : seeds) {                                             // :206-231
        if (occ.get(sd.f, sd.x, sd.y)) continue;
        Ann ann; ann.joints.resize(K);
        Joint& j = ann.joints[sd.f];
        j.v = sd.v; j.x = sd.x; j.y = sd.y; j.s = sd.s;
        dec.grow(&ann, fwd, bwd, true, 1.0);
        mark(ann);
        anns.push_back(ann);
    }

    if (p.force_complete) {                                                    // :233-236, 414-426
        AssocLists f2, b2;
        caf_scored(hr, caf, A, caf_H, caf_W, caf_stride, skeleton, p.force_complete_caf_th, 0.1,
                   p.ablation_caf_no_rescore, f2, b2);
        for (Ann& ann : anns) dec.grow(&ann, f2, b2, false, 4.0);
        for (Ann& ann : anns) dec.flood_fill(&ann);
    }

    nms_keypoints(&occ, &anns, p);                                             // :241

    for (int64_t n = 0; n < int64_t(anns.size()) && n < cap; n++) {            // :246-260
        for (int64_t k = 0; k < K; k++) {
            const Joint& j = anns[n].joints[k];
            float* r = out + (n * K + k) * 4;
            r[0] = j.v; r[1] = j.x; r[2] = j.y; r[3] = j.s;
        }
        out_ids[n] = anns[n].id;
    }
    return int64_t(anns.size());
}

int64_t oracle_cifcaf_decode(const float* cif, int64_t F, int64_t H, int64_t W, int64_t cif_stride,
                             const float* caf, int64_t A, int64_t caf_H, int64_t caf_W, int64_t caf_stride,
                             const int64_t* skeleton, const oracle_params* params,
                             const float* initial, const int64_t* initial_ids, int64_t n_initial,
                             int64_t cap, float* out, int64_t* out_ids, float* cifhr_out) {
    return oracle_cifcaf_decode_k(cif, F, H, W, cif_stride, caf, A, caf_H, caf_W, caf_stride, skeleton, params,
                                  initial, initial_ids, n_initial, cap, out, out_ids, cifhr_out, F);
}

// CifDet::call, cifdet.cpp:24-80.  Returns the number of detections (<= max_detections).
// categories int64 [max], scores float [max], boxes float [max,4] (x0,y0,x1,y1); cifhr_out optional.
int64_t oracle_cifdet_decode(const float* field, int64_t F, int64_t H, int64_t W, int64_t stride,
                             const oracle_params* params, int64_t max_detections,
                             int64_t* categories, float* scores, float* boxes, float* cifhr_out) {
    const Params& p = *params;
    const int64_t hh = (H - 1) * stride + 1, hw = (W - 1) * stride + 1;
    std::vector<float> buffer(size_t(F) * hh * hw, 0.0f);
    HiRes hr{buffer.data(), F, hh, hw};
    cifdethr_accumulate(hr, field, F, H, W, stride, 0.0, 1.0, p);            // cifdet.cpp:30-32
    if (cifhr_out) std::memcpy(cifhr_out, buffer.data(), buffer.size() * sizeof(float));
    std::vector<DetSeed> seeds = cifdet_seeds(hr, field, F, H, W, stride, p);   // :34-38
    Occupancy occ(p.occupancy_reduction, p.occupancy_min_scale);              // cifdet.hpp:37
    occ.reset(F, hh, hw);                                                     // :44
    int64_t n = 0;
    for (const DetSeed& s : seeds) {                                          // :50-67
        if (occ.get(s.c, s.x, s.y)) continue;
        occ.set(s.c, s.x, s.y, 0.1 * fmin(s.w, s.h));
        categories[n] = s.c + 1;
        scores[n] = s.v;
        boxes[4 * n + 0] = s.x - 0.5f * s.w; boxes[4 * n + 1] = s.y - 0.5f * s.h;
        boxes[4 * n + 2] = s.x + 0.5f * s.w; boxes[4 * n + 3] = s.y + 0.5f * s.h;
        n++;
        if (n >= max_detections) break;
    }
    return n;
}

}  // extern "C"
