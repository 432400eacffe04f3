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
#include "cafscored_impl.hpp"

namespace opa {

#ifndef OPA_SCORED_MIN_WAVES
#define OPA_SCORED_MIN_WAVES 2
#endif
__global__ __launch_bounds__(kScoredThreads, OPA_SCORED_MIN_WAVES) void cafscored_kernel(ScoredArgs s) {
#ifdef OPA_SCORED_SINGLE_TEMPLATE                     // experiment: the two-set routine (cells per thread, prefetch) for the single set too
    __shared__ int wave_tot[2][kScoredCells][kScoredThreads / 64];
    extern __shared__ float bb[];
    cafscored_plane2<false>(s, s, blockIdx.x, threadIdx.x, wave_tot, bb);
#else
    __shared__ int wave_tot[2][kScoredThreads / 64];
    extern __shared__ float bb[];
    cafscored_plane(s, blockIdx.x, threadIdx.x, wave_tot, bb);
#endif
}

// both list sets of a force-complete decode from ONE read of the field (cafscored_impl.hpp)
__global__ __launch_bounds__(kScoredThreads, OPA_SCORED_MIN_WAVES) void cafscored2_kernel(ScoredArgs s, ScoredArgs s2) {
    __shared__ int wave_tot[4][kScoredCells][kScoredThreads / 64];
    extern __shared__ float bb[];
    cafscored_plane2<true>(s, s2, blockIdx.x, threadIdx.x, wave_tot, bb);
}

ScoredArgs make_scored_args(const float* caf, int B, int A, int cH, int cW, int cstride,
                            const float* cifhr, int F, int hr_rows, int hr_cols, int hr_pitch,
                            const int64_t* skeleton, double score_th, double cif_floor, int no_rescore,
                            float* lists, int32_t* counts, float* chunk_bbox, int bbox_chunks, int bbox_stride,
                            const unsigned* tile_touch, const HrPool* pool) {
    if (bbox_chunks <= 0) chunk_bbox = nullptr;
    if (bbox_stride < bbox_chunks) bbox_stride = bbox_chunks;
    ScoredArgs s;
    s.caf = caf; s.A = A; s.HW = cH * cW; s.stride = cstride; s.cifhr = cifhr; s.F = F; s.hr_rows = hr_rows; s.hr_cols = hr_cols;
    s.hr_pitch = hr_pitch; s.skeleton = skeleton; s.score_th = score_th; s.cif_floor = cif_floor; s.no_rescore = no_rescore;
    s.lists = lists; s.counts = counts; s.chunk_bbox = chunk_bbox; s.nb = chunk_bbox ? bbox_chunks : 0; s.nb_stride = bbox_stride;
    s.planes = B * A;
    s.tile_touch = tile_touch; s.tiles_x = hr_pitch / kHrTileW;
    s.hr_slot = pool ? pool->slot : nullptr; s.hr_tpp = pool ? pool->tpp : 0;
    s.hr_image_stride = pool ? (size_t)pool->cap * kHrTileH * kHrTileW : (size_t)F * hr_rows * hr_pitch;
    s.touch_words = (s.tiles_x * ((hr_rows + kHrTileH - 1) / kHrTileH) + 31) / 32;
    return s;
}

hipError_t launch_cafscored(const ScoredArgs& s, hipStream_t st) {
    const size_t lds = s.chunk_bbox ? sizeof(float) * 2 * s.nb * 4 : 0;
    cafscored_kernel<<<s.planes, kScoredThreads, lds, st>>>(s);
    prof_mark(st, "cafscored_kernel");
    return hipGetLastError();
}

hipError_t launch_cafscored2(const ScoredArgs& s, const ScoredArgs& s2, hipStream_t st) {
    const size_t lds = sizeof(float) * 2 * 4 * ((s.chunk_bbox ? s.nb : 0) + (s2.chunk_bbox ? s2.nb : 0));
    ScoredArgs a = s, b = s2;
    if (!a.chunk_bbox) a.nb = 0;                      // (the second set's boxes sit behind 2 * nb * 4 floats of the first's)
    if (!b.chunk_bbox) b.nb = 0;
    cafscored2_kernel<<<s.planes, kScoredThreads, lds, st>>>(a, b);
    prof_mark(st, "cafscored_kernel");
    return hipGetLastError();
}

hipError_t launch_cafscored(const float* caf, int B, int A, int cH, int cW, int cstride,
                            const float* cifhr, int F, int hr_rows, int hr_cols, int hr_pitch,
                            const int64_t* skeleton, double score_th, double cif_floor, int no_rescore,
                            float* lists, int32_t* counts, hipStream_t st, float* chunk_bbox, int bbox_chunks,
                            int bbox_stride) {
    return launch_cafscored(make_scored_args(caf, B, A, cH, cW, cstride, cifhr, F, hr_rows, hr_cols, hr_pitch, skeleton, score_th,
                                             cif_floor, no_rescore, lists, counts, chunk_bbox, bbox_chunks, bbox_stride), st);
}

}  // namespace opa
