// CafScored::fill for one (image, CAF field) plane by one group of kScoredThreads threads: shared by cafscored_kernel
// (cafscored.hip: one group per workgroup) and the fused sort + cafscored launch of the decode (cifseeds.hip: two
// groups per 1024-thread workgroup, beside the seed sort's workgroups).  See cafscored.hip for the description.
#pragma once
#ifndef OPA_SCORED_EAGER
#define OPA_SCORED_EAGER 0
#endif
#ifndef OPA_SCORED_PREFETCH
#define OPA_SCORED_PREFETCH 2      // [r6] 1: the next confidence one step ahead, 250 -> 234 us at 256 images, wholebody 121 -> 110; 2: two steps deep (the
                                   // next cell's six planes too), 233 -> 223 / 110 -> 106.5 / 46.5 -> 43.7 at 32 images (profiles/r6/cafscored_*prefetch_ab.log); 0: round 5's loop
#endif
#include "common.hpp"

namespace opa {

// One workgroup walks a field.  8 waves at 76 VGPRs: three workgroups per CU, so the 608 planes of a bench batch are
// all resident at once; with 1024 threads a CU held one workgroup and the batch took three rounds (50 -> 37 us).
#ifndef OPA_SCORED_THREADS
#define OPA_SCORED_THREADS 512
#endif
constexpr int kScoredThreads = OPA_SCORED_THREADS;

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

#if OPA_SCORED_PREFETCH
    // the confidence of the NEXT step's cell travels while this step is compacted (one load per thread, index clamped: no load
    // under a condition): the first of a step's three dependent round trips is off its critical path
    float c_pre = P[1 * HW + (tid < HW ? tid : HW - 1)];
#endif
#if OPA_SCORED_PREFETCH == 2
    // two steps deep: the confidence two steps ahead, so that the other six planes of the NEXT step's cell can be
    // requested -- where its confidence passes -- while this step waits for its map gathers: one exposed round trip per step
    float c_pre2 = P[1 * HW + (tid + kScoredThreads < HW ? tid + kScoredThreads : HW - 1)];
    float q2 = 0.f, q3 = 0.f, q4 = 0.f, q5 = 0.f, q6 = 0.f, q7 = 0.f;
    if (tid < HW && live && !((double)c_pre < score_th)) {
        q2 = P[2 * HW + tid]; q3 = P[3 * HW + tid]; q4 = P[4 * HW + tid]; q5 = P[5 * HW + tid]; q6 = P[6 * HW + tid]; q7 = P[7 * HW + tid];
    }
#endif
    for (int c0 = 0; c0 < HW; c0 += kScoredThreads, parity ^= 1) {
        const int o = c0 + tid;
        bool keep_f = false, keep_b = false;
        float c = 0.f, cf = 0.f, cb = 0.f, x1 = 0.f, y1 = 0.f, x2 = 0.f, y2 = 0.f, s1 = 0.f, s2 = 0.f;
#if OPA_SCORED_PREFETCH == 2
        const float c_now = c_pre;
        const float p2 = q2, p3 = q3, p4 = q4, p5 = q5, p6 = q6, p7 = q7;
        c_pre = c_pre2;
        { const int on2 = o + 2 * kScoredThreads; c_pre2 = P[1 * HW + (on2 < HW ? on2 : HW - 1)]; }
        {
            const int on = o + kScoredThreads;
            if (on < HW && live && !((double)c_pre < score_th)) {
                q2 = P[2 * HW + on]; q3 = P[3 * HW + on]; q4 = P[4 * HW + on]; q5 = P[5 * HW + on]; q6 = P[6 * HW + on]; q7 = P[7 * HW + on];
            }
        }
#elif OPA_SCORED_PREFETCH
        const float c_now = c_pre;
        { const int on = o + kScoredThreads; c_pre = P[1 * HW + (on < HW ? on : HW - 1)]; }
#endif
        if (o < HW && live) {
            // The source asks for all seven planes of the cell at once; the optimiser SINKS the six coordinate / scale loads
            // into the threshold block below (ISA: confidence load, vmcnt(0), branch, then the six) -- a wave fetches them only
            // where one of its cells passes: 41 MB per 32 images instead of 112 (that is the FETCH_SIZE the review of round 5
            // could not explain), at the price of a second dependent round trip.  Forcing the seven loads to travel together
            // (-DOPA_SCORED_EAGER=1) is slower: 49-50 against 48 us at 32 images, 277-280 against 250-253 at 256, wholebody
            // 144 against 121 (profiles/r6/cafscored_eager_loads_ab.log) -- the compiler's choice stays.
#if OPA_SCORED_PREFETCH
            c = c_now;
#else
            c = P[1 * HW + o];
#endif
#if OPA_SCORED_PREFETCH == 2
            const float r2 = p2, r3 = p3, r4 = p4, r5 = p5, r6 = p6, r7 = p7;
#else
            const float r2 = P[2 * HW + o], r3 = P[3 * HW + o], r4 = P[4 * HW + o], r5 = P[5 * HW + o],
                        r6 = P[6 * HW + o], r7 = P[7 * HW + o];
#endif
#if OPA_SCORED_EAGER
            // (experiment: with the products computed out here the seven loads travel together)
            x1 = r2 * stride_f; y1 = r3 * stride_f; x2 = r4 * stride_f; y2 = r5 * stride_f; s1 = r6 * stride_f; s2 = r7 * stride_f;
            asm volatile("" : "+v"(x1), "+v"(y1), "+v"(x2), "+v"(y2), "+v"(s1), "+v"(s2));
#endif
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


// `tid`: thread index inside the group; `wave_tot` [2][kScoredCells][kScoredThreads / 64] (TWO: [4][...], the second set's behind
// the first's) and `bb` [2][nb][4] ((xmin, xmax, ymin, ymax) of the (x1, y1) columns per list chunk; TWO: the second set's
// [2][nb2][4] behind it): the group's LDS.  Every barrier is a WORKGROUP barrier: all groups of a workgroup walk planes of the
// same size in lockstep.
//
// TWO (round 6): a force-complete decode needs the lists twice -- at caf_th for the seed loop (cifcaf.cpp:153-161) and at
// force_complete_caf_th for _force_complete (:419-420).  Rounds 3-5 ran the pass twice: the 128 MB CAF tensor of a batch read
// twice, the two map values of every cell gathered twice.  Here one read of the field feeds both sets (`s2` = the second set's
// thresholds and outputs; everything that describes the field and the map is taken from `s`).
//
// kScoredCells cells per thread and step, the planes of the next step requested before this step's cells are looked at: both
// are compile-time experiments of round 6 (-DOPA_SCORED_CELLS=n, the prefetch unless -DOPA_SCORED_NO_PREFETCH).  A read-only
// stream of this shape reaches 6.2 TB/s on this chip (tools/gpu/micro/readbw.hip: 144 us for the field of 256 images) where the
// single-set kernel takes 250 us -- 190 of them with a threshold nothing passes, i.e. no gather and no store: the price of a
// barrier per step.  Measured at 256 images (profiles/r6/cafscored_variants.log): 1 / 2 / 4 cells per thread 269 / 267 / 314 us
// with the prefetch, 273 / 271 without; 256-thread groups 271; none beats round 5's loop (252), which therefore stays for
// the single set (cafscored_plane above) -- this routine serves the two-set pass, with one cell per thread.  Cell order
// (r, wave, lane) is raster order, and the lists keep it.
#ifndef OPA_SCORED_CELLS
#define OPA_SCORED_CELLS 1
#endif
constexpr int kScoredCells = OPA_SCORED_CELLS;

template <bool TWO>
__device__ __forceinline__ void cafscored_plane2(const ScoredArgs& s, const ScoredArgs& s2, int plane_in, int tid,
                                                int (*wave_tot)[kScoredCells][kScoredThreads / 64], float* bb) {
    const float* __restrict__ caf = s.caf; const float* __restrict__ cifhr = s.cifhr;
    const int A = s.A, HW = s.HW, stride = s.stride, F = s.F, hr_rows = s.hr_rows, hr_cols = s.hr_cols, hr_pitch = s.hr_pitch;
    const int64_t* __restrict__ skeleton = s.skeleton;
    const double score_th = s.score_th, cif_floor = s.cif_floor;
    const double score_th2 = TWO ? s2.score_th : 0.0, cif_floor2 = TWO ? s2.cif_floor : 0.0;
    const int no_rescore = s.no_rescore, nb = s.nb, nb_stride = s.nb_stride;
    const int nb2 = TWO ? s2.nb : 0;
    float* bb2 = bb + 2 * nb * 4;
    float* __restrict__ lists = s.lists; int32_t* __restrict__ counts = s.counts; float* __restrict__ chunk_bbox = s.chunk_bbox;
    const bool live = plane_in < s.planes;           // an idle group only keeps the barriers company
    const int plane = live ? plane_in : 0;
    if (chunk_bbox)
        for (int k = tid; k < 2 * nb * 4; k += kScoredThreads) bb[k] = (k & 1) ? -__builtin_inff() : __builtin_inff();
    if (TWO && s2.chunk_bbox)
        for (int k = tid; k < 2 * nb2 * 4; k += kScoredThreads) bb2[k] = (k & 1) ? -__builtin_inff() : __builtin_inff();
    __syncthreads();
    const int b = plane / A, a = plane - b * A;
    const int lane = tid & 63, w = tid >> 6;
    const float* P = caf + (size_t)plane * 8 * HW;
    const float* hr = cifhr + (size_t)b * s.hr_image_stride;
    const unsigned* touch = s.tile_touch ? s.tile_touch + (size_t)b * F * s.touch_words : nullptr;
    const int32_t* slot = s.hr_slot ? s.hr_slot + (size_t)b * F * s.hr_tpp : nullptr;   // pooled map: the slot table replaces the bitmap test
    float* Lf = lists + ((size_t)plane * 2 + 0) * 7 * HW;
    float* Lb = lists + ((size_t)plane * 2 + 1) * 7 * HW;
    float* Lf2 = TWO ? s2.lists + ((size_t)plane * 2 + 0) * 7 * HW : nullptr;
    float* Lb2 = TWO ? s2.lists + ((size_t)plane * 2 + 1) * 7 * HW : nullptr;
    const long long j1 = skeleton[2 * a + 0], j2 = skeleton[2 * a + 1];
    const float stride_f = (float)stride;
    // the lower of the two thresholds decides whether a cell is looked at all (caf_scored.cpp:44)
    const double th_low = TWO && score_th2 < score_th ? score_th2 : score_th;
    int base_f = 0, base_b = 0, base_f2 = 0, base_b2 = 0, parity = 0;
    constexpr int kStep = kScoredThreads * kScoredCells;

    // all seven planes of a cell are requested at once (the stage's compulsory bytes), one step ahead
    float pc[kScoredCells], p2[kScoredCells], p3[kScoredCells], p4[kScoredCells], p5[kScoredCells], p6[kScoredCells], p7[kScoredCells];
    auto request = [&](int c0) {
#pragma unroll
        for (int r = 0; r < kScoredCells; r++) {
            const int o = c0 + r * kScoredThreads + tid;
            const int oo = o < HW ? o : 0;
            pc[r] = o < HW && live ? P[1 * HW + oo] : -1.0f;
            p2[r] = P[2 * HW + oo]; p3[r] = P[3 * HW + oo]; p4[r] = P[4 * HW + oo];
            p5[r] = P[5 * HW + oo]; p6[r] = P[6 * HW + oo]; p7[r] = P[7 * HW + oo];
        }
    };
    request(0);
    for (int c0 = 0; c0 < HW; c0 += kStep, parity ^= 1) {
#ifdef OPA_SCORED_NO_PREFETCH
        if (c0) request(c0);
#endif
        bool keep_f[kScoredCells], keep_b[kScoredCells], keep_f2[kScoredCells], keep_b2[kScoredCells];
        float cf[kScoredCells], cb[kScoredCells], cf2[kScoredCells], cb2[kScoredCells];
        float x1[kScoredCells], y1[kScoredCells], x2[kScoredCells], y2[kScoredCells], s1[kScoredCells], s2v[kScoredCells];
        float c[kScoredCells];
        bool look[kScoredCells];
#pragma unroll
        for (int r = 0; r < kScoredCells; r++) {
            const int o = c0 + r * kScoredThreads + tid;
            c[r] = pc[r];
            look[r] = o < HW && live && !((double)c[r] < th_low);        // caf_scored.cpp:44
            x1[r] = p2[r] * stride_f; y1[r] = p3[r] * stride_f;          // :46-54
            x2[r] = p4[r] * stride_f; y2[r] = p5[r] * stride_f;
            s1[r] = p6[r] * stride_f; s2v[r] = p7[r] * stride_f;
        }
#ifndef OPA_SCORED_NO_PREFETCH
        if (c0 + kStep < HW) request(c0 + kStep);
#endif
        // the map values of the step's cells: all slot-table loads first, then all tile loads (cifhr_value is two dependent loads)
        float fhr[kScoredCells], bhr[kScoredCells];
#pragma unroll
        for (int r = 0; r < kScoredCells; r++) {
            fhr[r] = 0.f; bhr[r] = 0.f;
            if (look[r] && !no_rescore) {                                // :66-71
                fhr[r] = cifhr_value(hr, F, hr_rows, hr_cols, hr_pitch, j2, x2[r], y2[r], 0.0f, touch, s.touch_words, s.tiles_x, slot, s.hr_tpp);
                bhr[r] = cifhr_value(hr, F, hr_rows, hr_cols, hr_pitch, j1, x1[r], y1[r], 0.0f, touch, s.touch_words, s.tiles_x, slot, s.hr_tpp);
            }
        }
#pragma unroll
        for (int r = 0; r < kScoredCells; r++) {
            keep_f[r] = keep_b[r] = keep_f2[r] = keep_b2[r] = false;
            cf[r] = cb[r] = cf2[r] = cb2[r] = c[r];
            if (look[r]) {
                if (!no_rescore) {
                    cf[r] = (float)((double)c[r] * (cif_floor + (1.0 - cif_floor) * (double)fhr[r]));
                    cb[r] = (float)((double)c[r] * (cif_floor + (1.0 - cif_floor) * (double)bhr[r]));
                    if (TWO) {
                        cf2[r] = (float)((double)c[r] * (cif_floor2 + (1.0 - cif_floor2) * (double)fhr[r]));
                        cb2[r] = (float)((double)c[r] * (cif_floor2 + (1.0 - cif_floor2) * (double)bhr[r]));
                    }
                }
                const bool in1 = !((double)c[r] < score_th);             // :44 for this set
                keep_f[r] = in1 && (double)cf[r] > score_th;             // :74
                keep_b[r] = in1 && (double)cb[r] > score_th;             // :77
                if (TWO) {
                    const bool in2 = !((double)c[r] < score_th2);
                    keep_f2[r] = in2 && (double)cf2[r] > score_th2;
                    keep_b2[r] = in2 && (double)cb2[r] > score_th2;
                }
            }
        }
        unsigned long long mf[kScoredCells], mb[kScoredCells], mf2[kScoredCells], mb2[kScoredCells];
#pragma unroll
        for (int r = 0; r < kScoredCells; r++) {
            mf[r] = __ballot(keep_f[r]); mb[r] = __ballot(keep_b[r]);
            mf2[r] = TWO ? __ballot(keep_f2[r]) : 0ull; mb2[r] = TWO ? __ballot(keep_b2[r]) : 0ull;
            if (lane == 0) {
                wave_tot[parity][r][w] = __popcll(mf[r]) | (__popcll(mb[r]) << 16);
                if (TWO) wave_tot[2 + parity][r][w] = __popcll(mf2[r]) | (__popcll(mb2[r]) << 16);
            }
        }
        __syncthreads();                              // double-buffered totals: one barrier per step
        const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
        for (int r = 0; r < kScoredCells; r++) {
            int off_f = base_f + __popcll(mf[r] & lt), off_b = base_b + __popcll(mb[r] & lt);
            int off_f2 = base_f2 + __popcll(mf2[r] & lt), off_b2 = base_b2 + __popcll(mb2[r] & lt);
            int tot_f = 0, tot_b = 0, tot_f2 = 0, tot_b2 = 0;
#pragma unroll
            for (int k = 0; k < kScoredThreads / 64; k++) {
                const int t = wave_tot[parity][r][k];
                if (k < w) { off_f += t & 0xffff; off_b += t >> 16; }
                tot_f += t & 0xffff; tot_b += t >> 16;
                if (TWO) {
                    const int u = wave_tot[2 + parity][r][k];
                    if (k < w) { off_f2 += u & 0xffff; off_b2 += u >> 16; }
                    tot_f2 += u & 0xffff; tot_b2 += u >> 16;
                }
            }
            auto boxes = [&](float* q, int n_boxes, bool kf, int of, bool kb, int ob) {
                if (n_boxes > kListBboxChunks) {         // long lists: one LDS update per wave and chunk
                    widen_boxes(q, n_boxes, kf, of, x1[r], y1[r]);
                    widen_boxes(q + n_boxes * 4, n_boxes, kb, ob, x2[r], y2[r]);
                } else {
                    // short lists, few kept entries per step: LDS float min/max per entry (ds_min_f32 / ds_max_f32); a NaN
                    // coordinate never passes the window test and must not poison the box
                    auto widen = [](float* e, float x, float y) {
                        if (x == x) { __hip_atomic_fetch_min(e + 0, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                      __hip_atomic_fetch_max(e + 1, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
                        if (y == y) { __hip_atomic_fetch_min(e + 2, y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                      __hip_atomic_fetch_max(e + 3, y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
                    };
                    if (kf && of < n_boxes * 64) widen(q + (of >> 6) * 4, x1[r], y1[r]);
                    if (kb && ob < n_boxes * 64) widen(q + (n_boxes + (ob >> 6)) * 4, x2[r], y2[r]);
                }
            };
            if (chunk_bbox) boxes(bb, nb, keep_f[r], off_f, keep_b[r], off_b);
            if (TWO && s2.chunk_bbox) boxes(bb2, nb2, keep_f2[r], off_f2, keep_b2[r], off_b2);
            if (keep_f[r]) {
                Lf[0 * HW + off_f] = cf[r]; Lf[1 * HW + off_f] = x1[r]; Lf[2 * HW + off_f] = y1[r];
                Lf[3 * HW + off_f] = x2[r]; Lf[4 * HW + off_f] = y2[r]; Lf[5 * HW + off_f] = s1[r]; Lf[6 * HW + off_f] = s2v[r];
            }
            if (keep_b[r]) {                                                 // mirrored tuple, :55-63
                Lb[0 * HW + off_b] = cb[r]; Lb[1 * HW + off_b] = x2[r]; Lb[2 * HW + off_b] = y2[r];
                Lb[3 * HW + off_b] = x1[r]; Lb[4 * HW + off_b] = y1[r]; Lb[5 * HW + off_b] = s2v[r]; Lb[6 * HW + off_b] = s1[r];
            }
            if (TWO && keep_f2[r]) {
                Lf2[0 * HW + off_f2] = cf2[r]; Lf2[1 * HW + off_f2] = x1[r]; Lf2[2 * HW + off_f2] = y1[r];
                Lf2[3 * HW + off_f2] = x2[r]; Lf2[4 * HW + off_f2] = y2[r]; Lf2[5 * HW + off_f2] = s1[r]; Lf2[6 * HW + off_f2] = s2v[r];
            }
            if (TWO && keep_b2[r]) {
                Lb2[0 * HW + off_b2] = cb2[r]; Lb2[1 * HW + off_b2] = x2[r]; Lb2[2 * HW + off_b2] = y2[r];
                Lb2[3 * HW + off_b2] = x1[r]; Lb2[4 * HW + off_b2] = y1[r]; Lb2[5 * HW + off_b2] = s2v[r]; Lb2[6 * HW + off_b2] = s1[r];
            }
            base_f += tot_f; base_b += tot_b; base_f2 += tot_f2; base_b2 += tot_b2;
        }
    }
    if (tid == 0 && live) {
        counts[plane * 2 + 0] = base_f; counts[plane * 2 + 1] = base_b;
        if (TWO) { s2.counts[plane * 2 + 0] = base_f2; s2.counts[plane * 2 + 1] = base_b2; }
    }
    if (chunk_bbox || (TWO && s2.chunk_bbox)) {       // the chunk boxes (common.hpp), gathered in LDS while the lists were built
        __syncthreads();
        if (chunk_bbox)
            for (int k = tid; k < 2 * nb * 4 && live; k += kScoredThreads) {
                const int dir = k / (nb * 4), rest = k - dir * nb * 4;
                chunk_bbox[((size_t)plane * 2 + dir) * nb_stride * 4 + rest] = bb[k];
            }
        if (TWO && s2.chunk_bbox)
            for (int k = tid; k < 2 * nb2 * 4 && live; k += kScoredThreads) {
                const int dir = k / (nb2 * 4), rest = k - dir * nb2 * 4;
                s2.chunk_bbox[((size_t)plane * 2 + dir) * s2.nb_stride * 4 + rest] = bb2[k];
            }
    }
}

}  // namespace opa
