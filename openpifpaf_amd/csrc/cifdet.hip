// CifDet: detection decoding on gfx950.
//
// Replaces reference CifDet::call (csrc/src/cifdet.cpp:24-80).  CifDetHr accumulation and
// CifDetSeeds are the CifHr / CifSeeds kernels in their DET variants (cifhr.hip,
// cifseeds.hip); this file holds the last step: walk the score-sorted seeds, skip seeds
// whose cell is occupied, mark a box of 0.1*min(w,h) around accepted ones, emit
// (category, score, box) until max_detections_before_nms.  One wavefront per image: 64
// seeds are tested against the occupancy map per step (ballot + ctz picks the next live
// one); the occupancy box is filled by the wave's lanes.
#include "common.hpp"

namespace opa {

__global__ __launch_bounds__(64) void cifdet_collect_kernel(DetArgs a, DevParams p) {
    const int b = blockIdx.x, lane = lane_id();
    const int occ_h = a.occ_h, occ_w = a.occ_w;
    unsigned char* occ = a.occ + (size_t)b * a.F * occ_h * occ_w;
    int n_seeds = a.seed_count[b];
    if (n_seeds > a.seed_cap) n_seeds = a.seed_cap;
    const int32_t* seed_f = a.seed_f + (size_t)b * a.seed_cap;
    const float* seed_v = a.seed_vxywh + (size_t)b * a.seed_cap * 5;
    int64_t* cat = a.categories + (size_t)b * a.max_det;
    float* sc = a.scores + (size_t)b * a.max_det;
    float* bx = a.boxes + (size_t)b * a.max_det * 4;
    const double red = p.occupancy_reduction;

    int n = 0, pos = 0;
    while (pos < n_seeds && n < a.max_det) {
        const int i = pos + lane;
        bool live = false; int f = 0; float v = 0.f, x = 0.f, y = 0.f, w = 0.f, h = 0.f;
        if (i < n_seeds) {
            f = seed_f[i];
            const float* r = seed_v + (size_t)i * 5;
            v = r[0]; x = r[1]; y = r[2]; w = r[3]; h = r[4];
            double xd = (double)x, yd = (double)y;                       // occupancy.cpp:32-43
            if (red != 1.0) { xd /= red; yd /= red; }
            const long long xi = clampll(trunc_ll(xd), 0, occ_w - 1);
            const long long yi = clampll(trunc_ll(yd), 0, occ_h - 1);
            live = occ[((size_t)f * occ_h + yi) * occ_w + xi] == 0;      // cifdet.cpp:58
        }
        const unsigned long long mask = __ballot(live);
        if (mask == 0) { pos += kWave; continue; }
        const int l = __builtin_ctzll(mask);
        const int sf = __shfl(f, l);
        const float sv = __shfl(v, l), sx = __shfl(x, l), sy = __shfl(y, l), sw = __shfl(w, l), sh = __shfl(h, l);
        // occupancy.set(f, x, y, 0.1 * fmin(w, h)), cifdet.cpp:60 / occupancy.cpp:13-29
        double xd = (double)sx, yd = (double)sy, sigma = 0.1 * (double)fminf(sw, sh);
        if (red != 1.0) { xd /= red; yd /= red; sigma = fmax(p.occupancy_min_scale_reduced, sigma / red); }
        const int minx = (int)clampll(trunc_ll(xd - sigma), 0, occ_w - 1);
        const int miny = (int)clampll(trunc_ll(yd - sigma), 0, occ_h - 1);
        const int maxx = (int)clampll(trunc_ll(xd + sigma), minx + 1, occ_w);
        const int maxy = (int)clampll(trunc_ll(yd + sigma), miny + 1, occ_h);
        unsigned char* plane = occ + (size_t)sf * occ_h * occ_w;
        for (int yy = miny; yy < maxy; yy++)
            for (int xx = minx + lane; xx < maxx; xx += kWave) plane[(size_t)yy * occ_w + xx] = 1;
        if (lane == 0) {                                                  // cifdet.cpp:61-63
            cat[n] = sf + 1;
            sc[n] = sv;
            bx[4 * n + 0] = sx - 0.5f * sw; bx[4 * n + 1] = sy - 0.5f * sh;
            bx[4 * n + 2] = sx + 0.5f * sw; bx[4 * n + 3] = sy + 0.5f * sh;
        }
        n++;
        __threadfence_block();
        pos += l + 1;
    }
    if (lane == 0) a.counts[b] = n;
}

hipError_t launch_cifdet_collect(const DetArgs& a, const DevParams& p, hipStream_t st) {
    cifdet_collect_kernel<<<a.B, 64, 0, st>>>(a, p);
    prof_mark(st, "cifdet_collect_kernel");
    return hipGetLastError();
}

}  // namespace opa
