// CafScored::fill for one (image, CAF field) plane by one group of kScoredThreads threads: shared by cafscored_kernel
// (cafscored.hip: one group per workgroup) and the fused sort + cafscored launch of the decode (cifseeds.hip: two
// groups per 1024-thread workgroup, beside the seed sort's workgroups).  See cafscored.hip for the description.
#pragma once
#include "common.hpp"

namespace opa {

// One workgroup walks a field.  8 waves at 76 VGPRs: three workgroups per CU, so the 608 planes of a bench batch are
// all resident at once; with 1024 threads a CU held one workgroup and the batch took three rounds (50 -> 37 us).
constexpr int kScoredThreads = 512;

// Wave-wide min / max of a float with DPP row operations (register only; result broadcast from lane 63).
// (A NaN coordinate never passes the window test and must not poison a box: callers feed the identity for it.)
template <bool MAX>
__device__ __forceinline__ float wave_minmax_f32(float v) {
    auto step = [](float x, int o) { const float y = __int_as_float(o); return MAX ? fmaxf(x, y) : fminf(x, y); };
    v = step(v, __builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0xB1, 0xF, 0xF, false));
    v = step(v, __builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x4E, 0xF, 0xF, false));
    v = step(v, __builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x141, 0xF, 0xF, false));
    v = step(v, __builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x140, 0xF, 0xF, false));
    v = step(v, __builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x142, 0xA, 0xF, false));
    v = step(v, __builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x143, 0xC, 0xF, false));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// The entries a wave keeps in one step land on consecutive list positions, i.e. in at most two chunks: their
// (x, y) are reduced inside the wave and ONE lane widens the two boxes in LDS (64 lanes hammering one LDS word with
// ds_min_f32 serialise; the force-complete lists keep most cells of a field).
__device__ __forceinline__ void widen_boxes(float* bb, int nb, bool keep, int off, float x, float y) {
    const unsigned long long m = __ballot(keep);
    if (m == 0ull) return;
    const int lane = threadIdx.x & 63;
    const int c0 = __builtin_amdgcn_readlane(off, __builtin_ctzll(m)) >> 6;
    if (c0 >= nb) return;
    const float inf = __builtin_inff();
    const bool in0 = keep && (off >> 6) == c0, in1 = keep && !in0;
    const bool xok = x == x, yok = y == y;
    const float x0lo = wave_minmax_f32<false>(in0 && xok ? x : inf), x0hi = wave_minmax_f32<true>(in0 && xok ? x : -inf);
    const float y0lo = wave_minmax_f32<false>(in0 && yok ? y : inf), y0hi = wave_minmax_f32<true>(in0 && yok ? y : -inf);
    const bool any1 = __ballot(in1) != 0ull && c0 + 1 < nb;
    float x1lo = inf, x1hi = -inf, y1lo = inf, y1hi = -inf;
    if (any1) {
        x1lo = wave_minmax_f32<false>(in1 && xok ? x : inf); x1hi = wave_minmax_f32<true>(in1 && xok ? x : -inf);
        y1lo = wave_minmax_f32<false>(in1 && yok ? y : inf); y1hi = wave_minmax_f32<true>(in1 && yok ? y : -inf);
    }
    if (lane == 0) {
        float* q = bb + c0 * 4;
        __hip_atomic_fetch_min(q + 0, x0lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_max(q + 1, x0hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_min(q + 2, y0lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_max(q + 3, y0hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (any1) {
            __hip_atomic_fetch_min(q + 4, x1lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_max(q + 5, x1hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_min(q + 6, y1lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_max(q + 7, y1hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
}

// `tid`: thread index inside the group; `wave_tot` [2][kScoredThreads / 64] and `bb` [2][nb][4] ((xmin, xmax, ymin, ymax)
// of the (x1, y1) columns per list chunk): the group's LDS.  Every barrier is a WORKGROUP barrier: all groups of a
// workgroup walk planes of the same size in lockstep.
__device__ __forceinline__ void cafscored_plane(const ScoredArgs& s, int plane_in, int tid,
                                                int (*wave_tot)[kScoredThreads / 64], float* bb) {

    const float* __restrict__ caf = s.caf; const float* __restrict__ cifhr = s.cifhr;
    const int A = s.A, HW = s.HW, stride = s.stride, F = s.F, hr_rows = s.hr_rows, hr_cols = s.hr_cols, hr_pitch = s.hr_pitch;
    const int64_t* __restrict__ skeleton = s.skeleton;
    const double score_th = s.score_th, cif_floor = s.cif_floor;
    const int no_rescore = s.no_rescore, nb = s.nb, nb_stride = s.nb_stride;
    float* __restrict__ lists = s.lists; int32_t* __restrict__ counts = s.counts; float* __restrict__ chunk_bbox = s.chunk_bbox;
    const bool live = plane_in < s.planes;           // an idle group only keeps the barriers company
    const int plane = live ? plane_in : 0;
    if (chunk_bbox)
        for (int k = tid; k < 2 * nb * 4; k += kScoredThreads) bb[k] = (k & 1) ? -__builtin_inff() : __builtin_inff();
    __syncthreads();
    const int b = plane / A, a = plane - b * A;
    const int lane = tid & 63, w = tid >> 6;
    const float* P = caf + (size_t)plane * 8 * HW;
    const float* hr = cifhr + (size_t)b * s.hr_image_stride;
    const unsigned* touch = s.tile_touch ? s.tile_touch + (size_t)b * F * s.touch_words : nullptr;
    const int32_t* slot = s.hr_slot ? s.hr_slot + (size_t)b * F * s.hr_tpp : nullptr;   // pooled map: the slot table replaces the bitmap test
    float* Lf = lists + ((size_t)plane * 2 + 0) * 7 * HW;
    float* Lb = lists + ((size_t)plane * 2 + 1) * 7 * HW;
    const long long j1 = skeleton[2 * a + 0], j2 = skeleton[2 * a + 1];
    const float stride_f = (float)stride;
    int base_f = 0, base_b = 0, parity = 0;

    for (int c0 = 0; c0 < HW; c0 += kScoredThreads, parity ^= 1) {
        const int o = c0 + tid;
        bool keep_f = false, keep_b = false;
        float c = 0.f, cf = 0.f, cb = 0.f, x1 = 0.f, y1 = 0.f, x2 = 0.f, y2 = 0.f, s1 = 0.f, s2 = 0.f;
        if (o < HW && live) {
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
                    const float fhr = cifhr_value(hr, F, hr_rows, hr_cols, hr_pitch, j2, x2, y2, 0.0f, touch, s.touch_words, s.tiles_x, slot, s.hr_tpp);
                    const float bhr = cifhr_value(hr, F, hr_rows, hr_cols, hr_pitch, j1, x1, y1, 0.0f, touch, s.touch_words, s.tiles_x, slot, s.hr_tpp);
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
            if (nb > kListBboxChunks) {              // long lists: one LDS update per wave and chunk
                widen_boxes(bb, nb, keep_f, off_f, x1, y1);
                widen_boxes(bb + nb * 4, nb, keep_b, off_b, x2, y2);
            } else {
                // short lists, few kept entries per step: LDS float min/max per entry (ds_min_f32 / ds_max_f32); a NaN
                // coordinate never passes the window test and must not poison the box
                auto widen = [](float* q, float x, float y) {
                    if (x == x) { __hip_atomic_fetch_min(q + 0, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                  __hip_atomic_fetch_max(q + 1, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
                    if (y == y) { __hip_atomic_fetch_min(q + 2, y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                  __hip_atomic_fetch_max(q + 3, y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
                };
                if (keep_f && off_f < nb * 64) widen(bb + (off_f >> 6) * 4, x1, y1);
                if (keep_b && off_b < nb * 64) widen(bb + (nb + (off_b >> 6)) * 4, x2, y2);
            }
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
    if (tid == 0 && live) { counts[plane * 2 + 0] = base_f; counts[plane * 2 + 1] = base_b; }
    if (chunk_bbox) {                                 // the chunk boxes (common.hpp), gathered in LDS while the lists were built
        __syncthreads();
        for (int k = tid; k < 2 * nb * 4 && live; k += kScoredThreads) {
            const int dir = k / (nb * 4), rest = k - dir * nb * 4;
            chunk_bbox[((size_t)plane * 2 + dir) * nb_stride * 4 + rest] = bb[k];
        }
    }
}

}  // namespace opa
