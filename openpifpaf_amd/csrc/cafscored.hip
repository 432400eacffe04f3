// CafScored: association-field thresholding, rescoring and list building on gfx950.
//
// Replaces reference CafScored::fill / get (csrc/src/caf_scored.cpp:29-104): every
// CAF cell with confidence >= th yields a forward tuple (c,x1,y1,x2,y2,s1,s2)*stride
// and the mirrored backward tuple; each is rescored with the CifHr value at its
// TARGET joint, c * (floor + (1-floor) * hr), and kept when the result is > th.
//
// One 512-thread workgroup per (image, CAF field) walks the 7 used component
// planes in raster order with coalesced loads (this is the bandwidth-bound stage
// of the decode), gathers the two CifHr values from the L2-resident map, and
// stream-compacts survivors IN RASTER ORDER (wave ballot + cross-wave prefix in
// LDS) into structure-of-arrays lists:
//     lists[b][a][dir][component 0..6][cap]      counts[b][a][dir]
// Raster order is kept because grow_connection_blend's top-2 selection breaks score
// ties by list position (cifcaf.cpp:65-73).  SoA planes make the association
// kernel's list scans coalesced.
#include "common.hpp"

namespace opa {

// One workgroup walks a field.  8 waves at 76 VGPRs: three workgroups per CU, so the 608 planes of a bench batch are
// all resident at once; with 1024 threads a CU held one workgroup and the batch took three rounds (50 -> 37 us).
constexpr int kScoredThreads = 512;

__global__ __launch_bounds__(kScoredThreads, 2) void cafscored_kernel(
        const float* __restrict__ caf, int A, int HW, int stride,
        const float* __restrict__ cifhr, int F, int hr_rows, int hr_cols, int hr_pitch,
        const int64_t* __restrict__ skeleton, double score_th, double cif_floor, int no_rescore,
        float* __restrict__ lists, int32_t* __restrict__ counts, float* __restrict__ chunk_bbox) {
    __shared__ int wave_tot[2][kScoredThreads / 64];
    __shared__ float bb[2][kListBboxChunks][4];      // (xmin, xmax, ymin, ymax) of the (x1, y1) columns per list chunk
    if (threadIdx.x < 2 * kListBboxChunks * 4) (&bb[0][0][0])[threadIdx.x] = (threadIdx.x & 1) ? -__builtin_inff() : __builtin_inff();
    __syncthreads();
    const int plane = blockIdx.x;                  // b*A + a
    const int b = plane / A, a = plane - b * A;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const float* P = caf + (size_t)plane * 8 * HW;
    const float* hr = cifhr + (size_t)b * F * hr_rows * hr_pitch;
    float* Lf = lists + ((size_t)plane * 2 + 0) * 7 * HW;
    float* Lb = lists + ((size_t)plane * 2 + 1) * 7 * HW;
    const long long j1 = skeleton[2 * a + 0], j2 = skeleton[2 * a + 1];
    const float stride_f = (float)stride;
    int base_f = 0, base_b = 0, parity = 0;

    for (int c0 = 0; c0 < HW; c0 += kScoredThreads, parity ^= 1) {
        const int o = c0 + tid;
        bool keep_f = false, keep_b = false;
        float c = 0.f, cf = 0.f, cb = 0.f, x1 = 0.f, y1 = 0.f, x2 = 0.f, y2 = 0.f, s1 = 0.f, s2 = 0.f;
        if (o < HW) {
            // all seven planes of the cell are requested at once (the stage's compulsory bytes): one memory
            // round trip before the two CifHr gathers instead of two
            c = P[1 * HW + o];
            const float r2 = P[2 * HW + o], r3 = P[3 * HW + o], r4 = P[4 * HW + o], r5 = P[5 * HW + o],
                        r6 = P[6 * HW + o], r7 = P[7 * HW + o];
            if (!((double)c < score_th)) {                               // caf_scored.cpp:44
                x1 = r2 * stride_f; y1 = r3 * stride_f;                  // :46-54
                x2 = r4 * stride_f; y2 = r5 * stride_f;
                s1 = r6 * stride_f; s2 = r7 * stride_f;
                cf = c; cb = c;
                if (!no_rescore) {                                       // :66-71
                    const float fhr = cifhr_value(hr, F, hr_rows, hr_cols, hr_pitch, j2, x2, y2, 0.0f);
                    const float bhr = cifhr_value(hr, F, hr_rows, hr_cols, hr_pitch, j1, x1, y1, 0.0f);
                    cf = (float)((double)c * (cif_floor + (1.0 - cif_floor) * (double)fhr));
                    cb = (float)((double)c * (cif_floor + (1.0 - cif_floor) * (double)bhr));
                }
                keep_f = (double)cf > score_th;                          // :74
                keep_b = (double)cb > score_th;                          // :77
            }
        }
        const unsigned long long mf = __ballot(keep_f), mb = __ballot(keep_b);
        const unsigned long long lt = (1ull << lane) - 1ull;
        if (lane == 0) wave_tot[parity][w] = __popcll(mf) | (__popcll(mb) << 16);
        __syncthreads();                              // double-buffered totals: one barrier per step
        int off_f = base_f + __popcll(mf & lt), off_b = base_b + __popcll(mb & lt);
        int tot_f = 0, tot_b = 0;
#pragma unroll
        for (int k = 0; k < kScoredThreads / 64; k++) {
            const int t = wave_tot[parity][k];
            if (k < w) { off_f += t & 0xffff; off_b += t >> 16; }
            tot_f += t & 0xffff; tot_b += t >> 16;
        }
        if (chunk_bbox) {
            // LDS float min/max (ds_min_f32 / ds_max_f32); a NaN coordinate never passes the window test and must
            // not poison the box
            auto widen = [](float* q, float x, float y) {
                if (x == x) { __hip_atomic_fetch_min(q + 0, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                              __hip_atomic_fetch_max(q + 1, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
                if (y == y) { __hip_atomic_fetch_min(q + 2, y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                              __hip_atomic_fetch_max(q + 3, y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
            };
            if (keep_f && off_f < kListBboxChunks * 64) widen(bb[0][off_f >> 6], x1, y1);
            if (keep_b && off_b < kListBboxChunks * 64) widen(bb[1][off_b >> 6], x2, y2);
        }
        if (keep_f) {
            Lf[0 * HW + off_f] = cf; Lf[1 * HW + off_f] = x1; Lf[2 * HW + off_f] = y1;
            Lf[3 * HW + off_f] = x2; Lf[4 * HW + off_f] = y2; Lf[5 * HW + off_f] = s1; Lf[6 * HW + off_f] = s2;
        }
        if (keep_b) {                                                    // mirrored tuple, :55-63
            Lb[0 * HW + off_b] = cb; Lb[1 * HW + off_b] = x2; Lb[2 * HW + off_b] = y2;
            Lb[3 * HW + off_b] = x1; Lb[4 * HW + off_b] = y1; Lb[5 * HW + off_b] = s2; Lb[6 * HW + off_b] = s1;
        }
        base_f += tot_f; base_b += tot_b;
    }
    if (tid == 0) { counts[plane * 2 + 0] = base_f; counts[plane * 2 + 1] = base_b; }
    if (chunk_bbox) {                                 // the chunk boxes (common.hpp), gathered in LDS while the lists were built
        __syncthreads();
        if (tid < 2 * kListBboxChunks * 4)
            chunk_bbox[(size_t)plane * 2 * kListBboxChunks * 4 + tid] = (&bb[0][0][0])[tid];
    }
}

hipError_t launch_cafscored(const float* caf, int B, int A, int cH, int cW, int cstride,
                            const float* cifhr, int F, int hr_rows, int hr_cols, int hr_pitch,
                            const int64_t* skeleton, double score_th, double cif_floor, int no_rescore,
                            float* lists, int32_t* counts, hipStream_t st, float* chunk_bbox) {
    cafscored_kernel<<<B * A, kScoredThreads, 0, st>>>(caf, A, cH * cW, cstride, cifhr, F, hr_rows, hr_cols, hr_pitch,
                                            skeleton, score_th, cif_floor, no_rescore, lists, counts, chunk_bbox);
    prof_mark(st, "cafscored_kernel");
    return hipGetLastError();
}

}  // namespace opa
