// CifCaf greedy keypoint association + keypoint NMS on gfx950.
//
// Replaces reference CifCaf::call_with_initial_annotations, _grow,
// _frontier_add_from, _connection_value, grow_connection_blend, _force_complete,
// _flood_fill (csrc/src/cifcaf.cpp:32-449), Occupancy (occupancy.cpp:13-79) and
// NMSKeypoints::call (nms_keypoints.cpp:17-70).
//
// One wavefront per image (images are the data-parallel unit; a batch fills the
// chip).  The reference's control flow is a serial dependency chain per image
// (seed k is skipped iff an earlier pose occupies its cell; growth is a
// best-first search), so the wave runs that control flow wave-uniformly and uses
// its 64 lanes where the reference has inner loops:
//   * 64 sorted seeds are tested against the occupancy map per step (ballot + ctz
//     picks the next live seed);
//   * grow_connection_blend scans a CAF candidate list 64 entries per step
//     (coalesced SoA planes, L2 resident) and reduces top-1 / top-2 with
//     cross-lane shuffles, reproducing the reference's ">=" / ">" tie rules by
//     list position;
//   * occupancy boxes are filled one row of lanes at a time; NMS tests all joints
//     of a pose in one step.
// The frontier is an exact re-implementation of the binary max-heap behind
// std::priority_queue (sift-up on push, sift-down-to-leaf + sift-up on pop), so
// that equal-priority entries -- the norm: all edges leaving one joint share the
// bound sqrt(v) -- pop in the reference's order.  Joint confidences are kept in
// double like the reference's Joint struct; every float/double promotion follows
// the reference operation by operation and the library is built with
// -ffp-contract=off.
#include "common.hpp"

namespace opa {

// ------------------------------------------------------------------ helpers
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

struct ListView { const float* base; int cap; int n; };   // 7 SoA planes: c,x1,y1,x2,y2,s1,s2

struct ImageCtx {
    int K, A, F;                         // F = occupancy fields = n_cif
    const int32_t *adj_off, *adj_other, *adj_bone, *adj_fwd, *adj_first;
    const float* lists; const int32_t* list_counts; int list_cap;
    unsigned char* occ; int occ_h, occ_w;
    // LDS
    double* jv; float *jx, *jy, *js;     // current pose [K]
    float* e_score; double* e_v; float *e_x, *e_y, *e_s; int* e_se;   // frontier entry pool [4A]
    int* heap;                           // [4A] entry ids
    unsigned char* in_frontier;          // [2A]
    int heap_n, n_entries;
};

__device__ __forceinline__ ListView list_view(const ImageCtx& c, int bone, int dir) {
    ListView v;
    v.base = c.lists + ((size_t)bone * 2 + dir) * 7 * c.list_cap;
    v.cap = c.list_cap;
    v.n = c.list_counts[bone * 2 + dir];
    return v;
}

// -------------------------------------------------------- grow_connection_blend
// cifcaf.cpp:32-103.  Returns false for the all-zero joint.
struct BlendQuery { double x, y, xlo, xhi, ylo, yhi; float sigma2; };

__device__ __forceinline__ bool entry_score(const ListView& L, int i, const BlendQuery& q, float* score) {
    const float x1 = L.base[1 * L.cap + i], y1 = L.base[2 * L.cap + i];
    if ((double)x1 < q.xlo) return false;                      // cifcaf.cpp:54-57
    if ((double)x1 > q.xhi) return false;
    if ((double)y1 < q.ylo) return false;
    if ((double)y1 > q.yhi) return false;
    const double dx = (double)x1 - q.x, dy = (double)y1 - q.y;
    const float d2 = (float)(dx * dx + dy * dy);               // :60
    *score = (float)(exp(-0.5 * (double)d2 / (double)q.sigma2) * (double)L.base[i]);   // :63
    return true;
}

__device__ bool blend(const ListView& L, double x, double y, double xy_scale, double filter_sigmas,
                      bool only_max, double* ov, float* ox, float* oy, float* os) {
    const int lane = lane_id();
    xy_scale = fmax(xy_scale, 0.5);                            // :44
    const float sigma_filter = (float)(filter_sigmas * xy_scale / 2.0);   // :47
    BlendQuery q;
    q.x = x; q.y = y;
    q.sigma2 = (float)(0.25 * xy_scale * xy_scale);            // :48
    q.xlo = x - (double)sigma_filter; q.xhi = x + (double)sigma_filter;
    q.ylo = y - (double)sigma_filter; q.yhi = y + (double)sigma_filter;

    // pass 1: first place = max score, LAST list position among equals (">=", :65)
    float s1 = 0.0f; int i1 = -1;
    for (int i = lane; i < L.n; i += kWave) {
        float sc;
        if (entry_score(L, i, q, &sc) && sc >= s1) { s1 = sc; i1 = i; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float so = __shfl_xor(s1, off); const int io = __shfl_xor(i1, off);
        if (so > s1 || (so == s1 && io > i1)) { s1 = so; i1 = io; }
    }
    if (s1 == 0.0f || i1 < 0) return false;                    // :76

    const float e1x = L.base[3 * L.cap + i1], e1y = L.base[4 * L.cap + i1];
    const float e1s = fmaxf(0.0f, L.base[6 * L.cap + i1]);    // :78-81
    if (only_max) { *ov = (double)s1; *ox = e1x; *oy = e1y; *os = e1s; return true; }

    // pass 2: second place.  Sequential rule (:65-73) == max score among the rest;
    // among equals: the last position before i1 if any, else the first after i1.
    float s2 = 0.0f; int r2 = -1;
    for (int i = lane; i < L.n; i += kWave) {
        float sc;
        if (i == i1 || !entry_score(L, i, q, &sc) || !(sc > 0.0f)) continue;
        const int rank = i < i1 ? L.n + i : L.n - i;
        if (sc > s2 || (sc == s2 && rank > r2)) { s2 = sc; r2 = rank; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float so = __shfl_xor(s2, off); const int ro = __shfl_xor(r2, off);
        if (so > s2 || (so == s2 && ro > r2)) { s2 = so; r2 = ro; }
    }
    if (r2 < 0 || (double)s2 < 0.01 || (double)s2 < 0.5 * (double)s1) {     // :84-85
        *ov = 0.5 * (double)s1; *ox = e1x; *oy = e1y; *os = e1s; return true;
    }
    const int i2 = r2 >= L.n ? r2 - L.n : L.n - r2;
    const float e2x = L.base[3 * L.cap + i2], e2y = L.base[4 * L.cap + i2];
    const float e2s = fmaxf(0.0f, L.base[6 * L.cap + i2]);    // :88-91
    const double ddx = (double)(e1x - e2x), ddy = (double)(e1y - e2y);
    const float blend_d2 = (float)(ddx * ddx + ddy * ddy);     // :93
    if ((double)blend_d2 > ((double)e1s * (double)e1s) / 4.0) {             // :94-95
        *ov = 0.5 * (double)s1; *ox = e1x; *oy = e1y; *os = e1s; return true;
    }
    const float ssum = s1 + s2;                                // :97-102
    *ov = 0.5 * (double)ssum;
    *ox = (s1 * e1x + s2 * e2x) / ssum;
    *oy = (s1 * e1y + s2 * e2y) / ssum;
    *os = (s1 * e1s + s2 * e2s) / ssum;
    return true;
}

// -------------------------------------------------------------- frontier heap
// Exact behaviour of std::priority_queue<FrontierEntry, vector, FrontierCompare>
// (cifcaf.hpp:93, cifcaf.cpp:27-29): comp(a,b) = a.max_score < b.max_score.
__device__ __forceinline__ bool heap_less(const ImageCtx& c, int a, int b) { return c.e_score[a] < c.e_score[b]; }

__device__ void heap_sift_up(ImageCtx& c, int hole, int top, int value) {
    int parent = (hole - 1) / 2;
    while (hole > top && heap_less(c, c.heap[parent], value)) {
        c.heap[hole] = c.heap[parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    c.heap[hole] = value;
}

__device__ void heap_push(ImageCtx& c, int entry) {
    c.heap_n++;
    heap_sift_up(c, c.heap_n - 1, 0, entry);
}

__device__ int heap_pop(ImageCtx& c) {          // returns the top entry id
    const int top = c.heap[0];
    const int len = c.heap_n - 1;               // heap length after removing the back
    if (len > 0) {
        const int value = c.heap[len];
        int hole = 0, child = 0;
        while (child < (len - 1) / 2) {
            child = 2 * (child + 1);
            if (heap_less(c, c.heap[child], c.heap[child - 1])) child--;
            c.heap[hole] = c.heap[child];
            hole = child;
        }
        if ((len & 1) == 0 && child == (len - 2) / 2) {
            child = 2 * (child + 1);
            c.heap[hole] = c.heap[child - 1];
            hole = child - 1;
        }
        heap_sift_up(c, hole, 0, value);
    }
    c.heap_n = len;
    return top;
}

__device__ int new_entry(ImageCtx& c, float score, double v, float x, float y, float s, int start, int end) {
    const int e = c.n_entries++;
    c.e_score[e] = score; c.e_v[e] = v; c.e_x[e] = x; c.e_y[e] = y; c.e_s[e] = s;
    c.e_se[e] = (start << 16) | end;
    return e;
}

// cifcaf.cpp:316-346
__device__ void frontier_add_from(ImageCtx& c, int start) {
    const float max_score = (float)sqrt(c.jv[start]);
    for (int t = c.adj_off[start]; t < c.adj_off[start + 1]; t++) {
        const int other = c.adj_other[t];
        if (c.jv[other] > 0.0) continue;
        const int first = c.adj_first[t];
        if (c.in_frontier[first]) continue;
        heap_push(c, new_entry(c, max_score, 0.0, 0.f, 0.f, 0.f, start, other));
        c.in_frontier[first] = 1;
    }
}

// cifcaf.cpp:349-411 ; t = adjacency slot of (start -> end)
__device__ bool connection_value(ImageCtx& c, const DevParams& p, int start, int t,
                                 bool reverse_match_, double filter_sigmas,
                                 double* nv, float* nx, float* ny, float* ns) {
    const int bone = c.adj_bone[t], fwd = c.adj_fwd[t];
    const ListView caf_f = list_view(c, bone, fwd ? 0 : 1);
    const ListView caf_b = list_view(c, bone, fwd ? 1 : 0);
    const double sv = c.jv[start], sx = (double)c.jx[start], sy = (double)c.jy[start], ss = (double)c.js[start];
    if (!blend(caf_f, sx, sy, ss, filter_sigmas, false, nv, nx, ny, ns)) return false;
    *nv = sqrt(*nv * sv);                                                       // :386
    if (*nv < p.keypoint_threshold || *nv < sv * p.keypoint_threshold_rel) return false;   // :387-390
    if (p.reverse_match && reverse_match_ && start < c.F) {                     // :397
        double rv; float rx, ry, rs;
        if (!blend(caf_b, (double)*nx, (double)*ny, (double)*ns, filter_sigmas, false, &rv, &rx, &ry, &rs))
            return false;
        if (fabs(sx - (double)rx) + fabs(sy - (double)ry) > ss) return false;   // :404
    }
    return true;
}

__device__ void frontier_reset(ImageCtx& c) {
    const int lane = lane_id();
    for (int t = lane; t < 2 * c.A; t += kWave) c.in_frontier[t] = 0;
    c.heap_n = 0; c.n_entries = 0;
    wave_sync();
}

__device__ int find_adj_slot(const ImageCtx& c, int start, int end) {
    for (int t = c.adj_off[start]; t < c.adj_off[start + 1]; t++)
        if (c.adj_other[t] == end) return c.adj_first[t];
    return -1;
}

// cifcaf.cpp:265-313
__device__ void grow(ImageCtx& c, const DevParams& p, int greedy, bool reverse_match_, double filter_sigmas) {
    frontier_reset(c);
    for (int j = 0; j < c.K; j++) if (c.jv[j] != 0.0) frontier_add_from(c, j);
    while (c.heap_n > 0) {
        const int e = heap_pop(c);
        const int start = c.e_se[e] >> 16, end = c.e_se[e] & 0xffff;
        if (c.jv[end] > 0.0) continue;                                   // :284
        double v = c.e_v[e]; float x = c.e_x[e], y = c.e_y[e], s = c.e_s[e];
        if (v == 0.0) {                                                  // :287
            const int t = find_adj_slot(c, start, end);
            if (!connection_value(c, p, start, t, reverse_match_, filter_sigmas, &v, &x, &y, &s)) continue;
            if (!greedy) {                                               // :298-303
                heap_push(c, new_entry(c, (float)v, v, x, y, s, start, end));
                continue;
            }
        }
        c.jv[end] = v; c.jx[end] = x; c.jy[end] = y; c.js[end] = s;     // :310
        frontier_add_from(c, end);
    }
}

// cifcaf.cpp:429-449
__device__ void flood_fill(ImageCtx& c) {
    frontier_reset(c);
    for (int j = 0; j < c.K; j++) if (c.jv[j] != 0.0) frontier_add_from(c, j);
    while (c.heap_n > 0) {
        const int e = heap_pop(c);
        const int start = c.e_se[e] >> 16, end = c.e_se[e] & 0xffff;
        if (c.jv[end] > 0.0) continue;
        c.jv[end] = 0.00001; c.jx[end] = c.jx[start]; c.jy[end] = c.jy[start]; c.js[end] = c.js[start];
        frontier_add_from(c, end);
    }
}

// ---------------------------------------------------------------- occupancy
// occupancy.cpp:32-43 (byte map; `level`: 1 = association phase, 2 = NMS phase)
__device__ __forceinline__ size_t occ_cell(const ImageCtx& c, const DevParams& p, int f, double x, double y) {
    if (p.occupancy_reduction != 1.0) { x /= p.occupancy_reduction; y /= p.occupancy_reduction; }
    const long long xi = clampll(trunc_ll(x), 0, c.occ_w - 1);
    const long long yi = clampll(trunc_ll(y), 0, c.occ_h - 1);
    return ((size_t)f * c.occ_h + yi) * c.occ_w + xi;
}

// occupancy.cpp:13-29, all lanes cooperate on one box
__device__ void occ_set(const ImageCtx& c, const DevParams& p, int f, double x, double y, double sigma,
                        unsigned char level) {
    if (p.occupancy_reduction != 1.0) {
        x /= p.occupancy_reduction; y /= p.occupancy_reduction;
        sigma = fmax(p.occupancy_min_scale_reduced, sigma / p.occupancy_reduction);
    }
    const int minx = (int)clampll(trunc_ll(x - sigma), 0, c.occ_w - 1);
    const int miny = (int)clampll(trunc_ll(y - sigma), 0, c.occ_h - 1);
    const int maxx = (int)clampll(trunc_ll(x + sigma), minx + 1, c.occ_w);
    const int maxy = (int)clampll(trunc_ll(y + sigma), miny + 1, c.occ_h);
    const int bw = maxx - minx;
    const int lane = lane_id();
    unsigned char* plane = c.occ + (size_t)f * c.occ_h * c.occ_w;
    if (bw <= 16) {                       // 4 rows x 16 columns per step
        const int lx = lane & 15, ly = lane >> 4;
        for (int yy = miny + ly; yy < maxy; yy += 4)
            if (lx < bw) plane[(size_t)yy * c.occ_w + minx + lx] = level;
    } else if (bw <= 32) {                // 2 rows x 32 columns per step
        const int lx = lane & 31, ly = lane >> 5;
        for (int yy = miny + ly; yy < maxy; yy += 2)
            if (lx < bw) plane[(size_t)yy * c.occ_w + minx + lx] = level;
    } else {
        for (int yy = miny; yy < maxy; yy++)
            for (int xx = minx + lane; xx < maxx; xx += kWave) plane[(size_t)yy * c.occ_w + xx] = level;
    }
}

// mark every filled joint of the current pose, cifcaf.cpp:225-229
__device__ void mark_pose(const ImageCtx& c, const DevParams& p) {
    for (int f = 0; f < c.F; f++) {
        if (c.jv[f] == 0.0) continue;
        occ_set(c, p, f, (double)c.jx[f], (double)c.jy[f], (double)c.js[f], 1);
    }
    __threadfence_block();
}

// nms_keypoints.hpp:25-32 on the LDS pose
__device__ double pose_score_lds(const ImageCtx& c) {
    double acc = 0.0;
    for (int k = 0; k < c.K; k++) { const float i = (float)acc; acc = (double)i + c.jv[k]; }
    return acc / (double)c.K;
}

// ------------------------------------------------------------------- kernel
__global__ __launch_bounds__(64) void cifcaf_assoc_kernel(AssocArgs a, DevSkeleton sk, DevParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int b = blockIdx.x, lane = lane_id();
    const int K = a.K, A = a.A, E = 2 * A, P4 = 4 * A;

    ImageCtx c;
    c.K = K; c.A = A; c.F = K;
    c.adj_off = sk.adj_off; c.adj_other = sk.adj_other; c.adj_bone = sk.adj_bone; c.adj_fwd = sk.adj_fwd;
    c.adj_first = sk.adj_first;
    const int greedy = p.greedy;
    c.lists = a.lists + (size_t)b * A * 2 * 7 * a.list_cap;
    c.list_counts = a.list_counts + (size_t)b * A * 2;
    c.list_cap = a.list_cap;
    c.occ = a.occ + (size_t)b * K * a.occ_h * a.occ_w; c.occ_h = a.occ_h; c.occ_w = a.occ_w;
    // LDS carve (8-byte items first)
    unsigned char* sp = smem;
    c.jv = (double*)sp; sp += sizeof(double) * K;
    c.e_v = (double*)sp; sp += sizeof(double) * P4;
    double* nms_score = (double*)sp; sp += sizeof(double) * a.max_ann;
    c.jx = (float*)sp; sp += sizeof(float) * K;
    c.jy = (float*)sp; sp += sizeof(float) * K;
    c.js = (float*)sp; sp += sizeof(float) * K;
    c.e_score = (float*)sp; sp += sizeof(float) * P4;
    c.e_x = (float*)sp; sp += sizeof(float) * P4;
    c.e_y = (float*)sp; sp += sizeof(float) * P4;
    c.e_s = (float*)sp; sp += sizeof(float) * P4;
    c.e_se = (int*)sp; sp += sizeof(int) * P4;
    c.heap = (int*)sp; sp += sizeof(int) * P4;
    int* nms_order = (int*)sp; sp += sizeof(int) * a.max_ann;
    c.in_frontier = sp; sp += E;
    c.heap_n = 0; c.n_entries = 0;

    double* anns = a.anns + (size_t)b * a.max_ann * K * 4;
    int64_t* ann_ids = a.ann_ids + (size_t)b * a.max_ann;
    int n_kept = 0, n_dropped = 0;
    const bool prune = !p.force_complete;     // a pose scoring below the instance threshold before NMS cannot survive it

    auto store_pose = [&](long long id) {
        if (prune && pose_score_lds(c) < p.nms_instance_threshold) return;
        if (n_kept >= a.max_ann) { n_dropped++; return; }
        double* dst = anns + (size_t)n_kept * K * 4;
        for (int k = lane; k < K; k += kWave) {
            dst[4 * k + 0] = c.jv[k]; dst[4 * k + 1] = (double)c.jx[k];
            dst[4 * k + 2] = (double)c.jy[k]; dst[4 * k + 3] = (double)c.js[k];
        }
        if (lane == 0) ann_ids[n_kept] = id;
        n_kept++;
    };

    // ---- initial annotations (tracking API), cifcaf.cpp:177-202
    for (int n = 0; n < a.n_initial; n++) {
        const float* src = a.initial + ((size_t)b * a.n_initial + n) * K * 4;
        for (int k = lane; k < K; k += kWave) {
            c.jv[k] = (double)src[4 * k + 0]; c.jx[k] = src[4 * k + 1];
            c.jy[k] = src[4 * k + 2]; c.js[k] = src[4 * k + 3];
        }
        wave_sync();
        grow(c, p, greedy, true, 1.0);
        mark_pose(c, p);
        store_pose(a.initial_ids ? a.initial_ids[(size_t)b * a.n_initial + n] : -1);
    }

    // ---- seeds in score order, cifcaf.cpp:206-231
    int n_seeds = a.seed_count[b];
    if (n_seeds > a.seed_cap) n_seeds = a.seed_cap;
    const int32_t* seed_f = a.seed_f + (size_t)b * a.seed_cap;
    const float4* seed_vxys = reinterpret_cast<const float4*>(a.seed_vxys) + (size_t)b * a.seed_cap;
    int pos = 0;
    while (pos < n_seeds) {
        const int i = pos + lane;
        bool live = false; int f = 0; float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < n_seeds) {
            f = seed_f[i]; s = seed_vxys[i];
            live = c.occ[occ_cell(c, p, f, (double)s.y, (double)s.z)] == 0;     // :211
        }
        const unsigned long long mask = __ballot(live);
        if (mask == 0) { pos += kWave; continue; }
        const int l = __builtin_ctzll(mask);
        const int sf = __shfl(f, l);
        const float sv = __shfl(s.x, l), sx = __shfl(s.y, l), sy = __shfl(s.z, l), ss = __shfl(s.w, l);
        for (int k = lane; k < K; k += kWave) { c.jv[k] = 0.0; c.jx[k] = 0.f; c.jy[k] = 0.f; c.js[k] = 0.f; }
        wave_sync();
        c.jv[sf] = (double)sv; c.jx[sf] = sx; c.jy[sf] = sy; c.js[sf] = ss;   // :213-218
        wave_sync();
        grow(c, p, greedy, true, 1.0);
        mark_pose(c, p);
        store_pose(-1);
        pos += l + 1;
    }
    __threadfence_block();

    // ---- force complete, cifcaf.cpp:233-236,414-449
    if (p.force_complete) {
        c.lists = a.lists_fc + (size_t)b * A * 2 * 7 * a.list_cap;
        c.list_counts = a.list_counts_fc + (size_t)b * A * 2;
        for (int pass = 0; pass < 2; pass++) {          // all grows first, then all flood fills
            for (int n = 0; n < n_kept; n++) {
                double* src = anns + (size_t)n * K * 4;
                for (int k = lane; k < K; k += kWave) {
                    c.jv[k] = src[4 * k + 0]; c.jx[k] = (float)src[4 * k + 1];
                    c.jy[k] = (float)src[4 * k + 2]; c.js[k] = (float)src[4 * k + 3];
                }
                wave_sync();
                if (pass == 0) grow(c, p, greedy, false, 4.0); else flood_fill(c);
                wave_sync();
                for (int k = lane; k < K; k += kWave) {
                    src[4 * k + 0] = c.jv[k]; src[4 * k + 1] = (double)c.jx[k];
                    src[4 * k + 2] = (double)c.jy[k]; src[4 * k + 3] = (double)c.js[k];
                }
                __threadfence_block();
            }
        }
    }

    // ---- keypoint NMS, nms_keypoints.cpp:17-70
    auto global_score = [&](int n) {                    // UniformScore on a stored pose
        const double* src = anns + (size_t)n * K * 4;
        double acc = 0.0;
        for (int k = 0; k < K; k++) { const float i = (float)acc; acc = (double)i + src[4 * k]; }
        return acc / (double)K;
    };
    for (int n = lane; n < n_kept; n += kWave) nms_score[n] = global_score(n);
    wave_sync();
    for (int n = lane; n < n_kept; n += kWave) {        // rank by score desc (ties: creation order)
        const double sn = nms_score[n];
        int rank = 0;
        for (int m = 0; m < n_kept; m++) { const double sm = nms_score[m]; rank += (sm > sn || (sm == sn && m < n)) ? 1 : 0; }
        nms_order[rank] = n;
    }
    wave_sync();
    for (int r = 0; r < n_kept; r++) {
        double* pose = anns + (size_t)nms_order[r] * K * 4;
        for (int k0 = 0; k0 < K; k0 += kWave) {         // K is also the number of occupancy fields
            const int k = k0 + lane;
            bool need_set = false; double v = 0.0, x = 0.0, y = 0.0, s = 0.0;
            if (k < K) {
                v = pose[4 * k]; x = pose[4 * k + 1]; y = pose[4 * k + 2]; s = pose[4 * k + 3];
                if (v != 0.0) {
                    if (c.occ[occ_cell(c, p, k, x, y)] >= 2) pose[4 * k] = v * p.nms_suppression;   // :50-51
                    else need_set = true;
                }
            }
            unsigned long long m = __ballot(need_set);
            while (m) {
                const int l = __builtin_ctzll(m); m &= m - 1;
                occ_set(c, p, k0 + l, __shfl(x, l), __shfl(y, l), __shfl(s, l), 2);               // :53
            }
        }
        __threadfence_block();
    }
    // keypoint threshold, instance threshold, final order (:58-69)
    for (int r = lane; r < n_kept; r += kWave) {
        double* pose = anns + (size_t)nms_order[r] * K * 4;
        double acc = 0.0;
        for (int k = 0; k < K; k++) {
            double v = pose[4 * k];
            if (!(v > p.nms_keypoint_threshold)) { v = 0.0; pose[4 * k] = 0.0; }
            const float i = (float)acc; acc = (double)i + v;
        }
        nms_score[r] = acc / (double)K;                 // indexed by sorted position r now
    }
    wave_sync();
    __threadfence_block();
    int n_out = 0;
    float* out = a.out + (size_t)b * a.max_ann * K * 4;
    int64_t* out_ids = a.out_ids + (size_t)b * a.max_ann;
    for (int r0 = 0; r0 < n_kept; r0 += kWave) {
        const int r = r0 + lane;
        bool keep = false; int rank = 0;
        if (r < n_kept) {
            const double sr = nms_score[r];
            keep = !(sr < p.nms_instance_threshold);
            if (keep) for (int m = 0; m < n_kept; m++) {
                const double sm = nms_score[m];
                if (sm < p.nms_instance_threshold) continue;
                rank += (sm > sr || (sm == sr && m < r)) ? 1 : 0;
            }
        }
        unsigned long long m = __ballot(keep);
        n_out += __popcll(m);
        while (m) {                                     // one pose per step, lanes over joints
            const int l = __builtin_ctzll(m); m &= m - 1;
            const int src_r = r0 + l, dst = __shfl(rank, l);
            const int src_n = nms_order[src_r];
            const double* pose = anns + (size_t)src_n * K * 4;
            for (int k = lane; k < K; k += kWave) {     // cifcaf.cpp:250-258
                float4 o;
                o.x = (float)pose[4 * k]; o.y = (float)pose[4 * k + 1];
                o.z = (float)pose[4 * k + 2]; o.w = (float)pose[4 * k + 3];
                reinterpret_cast<float4*>(out)[(size_t)dst * K + k] = o;
            }
            if (lane == 0) out_ids[dst] = ann_ids[src_n];
        }
    }
    if (lane == 0) {
        a.out_count[b] = n_dropped > 0 ? a.max_ann + n_dropped : n_out;
        a.status[b] = n_dropped;
    }
}

hipError_t launch_assoc(const AssocArgs& a, const DevSkeleton& sk, const DevParams& p, hipStream_t st) {
    const int K = a.K, A = a.A, P4 = 4 * A, E = 2 * A;
    size_t lds = sizeof(double) * (K + P4 + a.max_ann) + sizeof(float) * (3 * K + 4 * P4)
               + sizeof(int) * (2 * P4 + a.max_ann) + E + 16;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)cifcaf_assoc_kernel,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    cifcaf_assoc_kernel<<<a.B, 64, lds, st>>>(a, sk, p);
    prof_mark(st, "cifcaf_assoc_kernel");
    return hipGetLastError();
}

// ------------------------------------------- exported grow_connection_blend op
// cifcaf.cpp:105-113 : rows [n,7] (AoS, as the reference's op takes it)
__global__ __launch_bounds__(64) void blend_rows_kernel(const float* rows, int n, double x, double y, double s,
                                                        double filter_sigmas, int only_max, float* soa, double* out4) {
    const int lane = lane_id();
    for (int i = lane; i < n; i += kWave)
        for (int k = 0; k < 7; k++) soa[(size_t)k * n + i] = rows[(size_t)i * 7 + k];
    __threadfence_block();
    ListView L; L.base = soa; L.cap = n; L.n = n;
    double v = 0.0; float ox = 0.f, oy = 0.f, os = 0.f;
    const bool ok = blend(L, x, y, s, filter_sigmas, only_max != 0, &v, &ox, &oy, &os);
    if (lane == 0) {
        if (ok) { out4[0] = (double)ox; out4[1] = (double)oy; out4[2] = (double)os; out4[3] = v; }
        else { out4[0] = 0.0; out4[1] = 0.0; out4[2] = 0.0; out4[3] = 0.0; }
    }
}

hipError_t launch_blend(const float* rows, int n, double x, double y, double s, double filter_sigmas,
                        int only_max, double* out4_dev, hipStream_t st) {
    // scratch for the SoA copy sits behind the 4 result doubles
    float* soa = reinterpret_cast<float*>(out4_dev + 4);
    blend_rows_kernel<<<1, 64, 0, st>>>(rows, n, x, y, s, filter_sigmas, only_max, soa, out4_dev);
    return hipGetLastError();
}

}  // namespace opa
