// Shared declarations of the MI355X (gfx950) CifCaf decode library.
// Internal header: the public boundary is include/openpifpaf_amd.h.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "../../include/openpifpaf_amd.h"

namespace opa {

constexpr int kWave = 64;                 // CDNA4 wavefront width
constexpr int kHrTileW = 64;              // CifHr tile width  (one 256-B row segment)
#ifndef OPA_HR_TILE_H
#define OPA_HR_TILE_H 32
#endif
constexpr int kHrTileH = OPA_HR_TILE_H;   // CifHr tile height
constexpr int kHrLdsPitch = kHrTileW + 16;// LDS row pitch: +16 banks so a 16x4 patch is conflict free
constexpr int kSortLdsKeys = 8192;        // 64 KiB of u64 keys sorted inside LDS

// Launch-time view of one batched decode (device pointers into the workspace).
struct Layout {
    // shapes
    int B, F, K, A, H, W, cH, cW, stride, cstride, max_ann;     // F CIF fields, K >= F joints per annotation
    int hr_rows, hr_cols, hr_pitch;       // high-res map geometry
    int hr_tpp, hr_pool_cap;              // 32x64 tiles per plane; slots of one image's tile pool (opa_shape::cifhr_pool_tiles)
    int hr_spill_cap;                     // slots of the batch's shared spill region behind the pools (automatic sizing only)
    int occ_h, occ_w;                     // occupancy geometry
    int cif_cells;                        // F*H*W  (seed capacity)
    int caf_cells;                        // cH*cW  (list capacity per (field, direction))
    int sort_cap;                         // next pow2 >= cif_cells
    int bbox_chunks;                      // 64-entry chunks of a CAF list that get a bounding box (all of them, up to kListBboxMax)
    // byte offsets into the workspace (all 256-B aligned)
    size_t off_hdr, off_tile_clean, off_cifhr, off_act, off_act_count, off_seed_keys, off_seed_count,
           off_seed_f, off_seed_vxys, off_seed_cell, off_lists, off_list_counts,
           off_lists_fc, off_list_counts_fc, off_list_bbox, off_list_bbox_fc, off_fc_meta, off_occ, off_anns, off_ann_meta, off_status, off_stats, off_trace, off_assoc_queue,
           total_no_fc, total;           // total_no_fc: everything but the regions only a force-complete decode uses (they come last)
    size_t occ_image_words;               // 32-bit words of one image's occupancy bitmap (capacity)
    // scratch of the seed tie pass (cifseeds.hip): `big` lies in the active-cell list (dead once the map is built, and
    // exactly as large), `small` in the occupancy bitmap (cleared by the association kernel afterwards) where it fits
    size_t off_tie_small, tie_small_stride, off_tie_state;
    size_t off_hr_slot, off_hr_overflow;   // the pooled map's slot tables [B][F][tpp]; overflow flags [B], spill counter, work counter, slot counters [B]
    size_t off_hr_work, off_cand_start;    // the tile kernel's work list [B*F*tpp] int2; seed candidates: chunk starts [B*F][chunks] + counts [B*F]
    int cand_chunks;                       // 1024-cell chunks of a CIF plane
};

bool make_layout(const opa_shape& s, Layout* L, const char** why);

// Device-side copy of the tunables.
struct DevParams {
    double cif_threshold, seed_threshold, caf_threshold, cif_floor;
    double keypoint_threshold, keypoint_threshold_rel;
    double nms_suppression, nms_instance_threshold, nms_keypoint_threshold;
    double force_complete_caf_th, occupancy_reduction, occupancy_min_scale_reduced;
    double occupancy_inv_reduction;       // 1 / reduction when the reduction is a power of two (x / r == x * (1/r) exactly), else 0
    int64_t cifhr_neighbors;
    int reverse_match, force_complete, greedy;
    int ablation_cifseeds_nms, ablation_cifseeds_no_rescore, ablation_caf_no_rescore, ablation_cifhr_skip;
};
DevParams to_dev(const opa_params& p);

// Skeleton + adjacency on the device (owned by an opa_cifcaf handle).
struct DevSkeleton {
    int K, A;
    const int64_t* skeleton;   // [A,2] 0-based
    const int32_t* adj_off;    // [K+1]
    const int32_t* adj_other;  // [2A]  other end of the bone
    const int32_t* adj_bone;   // [2A]  CAF field index
    const int32_t* adj_fwd;    // [2A]  1: joint is skeleton[a][0] (walk the forward list)
    const int32_t* adj_first;  // [2A]  first slot of the same joint with the same other end
                               //       (duplicate bones share one frontier slot, cifcaf.cpp:329,361-374)
};

// profiling hook: records an event after an enqueued operation when profiling is on
void prof_mark(hipStream_t st, const char* name);

// ---- kernel launchers (one per .hip file) ---------------------------------
// The CifHr map of the decode path is a POOL of 32x64 tiles per image: only tiles a CIF cell's box reaches get a slot
// (a few hundred of the 3927 of a 641-px COCO image), nothing is cleared, nothing carries over between calls.
// `cifhr` then points at the pools ([B][cap] tiles); with `pool == nullptr` the map is the dense [B][F][rows][pitch] array of
// the stage-level entry points.
struct HrPool {
    int32_t* slot;         // [B][F][tpp] tile -> slot in its image's pool; -1: untouched, -2: the pool was full
    int32_t* overflow;     // [B] set when an image reaches more tiles than its pool holds (the decode then flags the image failed)
    int cap;               // slots per image
    int tpp;               // tiles per plane
    // An image that reaches more tiles than its pool holds takes slots of a region all images of the batch share, right
    // behind the B pools (slot numbers stay relative to the image's own pool: (B - b) * cap + i), handed out by a counter;
    // only when that runs out too is the image flagged.  `spill_cap` 0: no such region.
    int spill_cap, images;
    int32_t* spill_count;  // [1]
    // Work-list form of the tile kernel (cifhr.hip): cif_active_kernel gives every reached tile its slot -- `img_tiles` [B]: slots
    // taken per image -- and appends (plane * tpp + tile, slot) to `work`, `work_count` [1] entries so far.  overflow [B],
    // spill_count, work_count and img_tiles [B] are ONE region of 2B + 2 words, zeroed by a launch before that kernel.
    int2* work;            // [B * F * tpp]
    int32_t* work_count;
    int32_t* img_tiles;
};
// The cells CifSeeds::fill looks at (cif_seeds.cpp:47: !(c < seed_threshold)), written by cif_active_kernel on its one pass
// over the CIF field: per (image, field) plane (cell index as float bits, c, x, y) in raster order, where the candidates of
// every chunk of 1024 cells begin (`start` [planes][chunks]) and how many there are (`count` [planes]).
constexpr int kFillCells = 4;             // cells (or candidates) per thread of the seed fill: 1024 per workgroup
struct SeedCandidates { float4* cand; int32_t* start; int32_t* count; int chunks; bool produced; };
hipError_t launch_cifhr(const float* cif, int B, int F, int H, int W, int stride,
                        double min_scale, double factor, const DevParams& p,
                        float* cifhr, int hr_rows, int hr_pitch,
                        float* act, int32_t* act_count, hipStream_t st, bool det = false,
                        unsigned long long* ws_header = nullptr, unsigned long long layout_hash = 0,
                        unsigned char* tile_state = nullptr, int32_t* zero_per_image = nullptr,
                        const HrPool* pool = nullptr, SeedCandidates* cand = nullptr);
// one image's map as the dense [F][rows][cols] array (get_cifhr of a pooled map)
hipError_t launch_cifhr_gather(const float* pool_image, const int32_t* slot_image, int F, int rows, int cols, int tiles_x, int tpp,
                               float* out, hipStream_t st);

// Workspace header (first 256 bytes): [0] magic, [1] hash of the layout the stored tile bitmap describes,
// [2] 1 = stored bitmap invalid for this call (written by the first kernel of a call).
constexpr unsigned long long kWsMagic = 0x6f70615f63696668ull;   // "opa_cifh"

struct ScoredArgs;
// Scratch of the pass that puts seeds of EQUAL score into libstdc++'s std::sort order (cifseeds.hip); `state` [B]:
// 0 no equal scores, 1 re-sorted, -1 not reproduced (introsort's heapsort branch).  seed_tie_order(): 1 = libstdc++
// (the reference, default), 0 = cell index (opa_set_seed_tie_order / OPA_SEED_TIES=index).
struct TieScratch { unsigned char* big; size_t big_stride; unsigned char* small_; size_t small_stride; int32_t* state;
                    int defer; };   // defer: launch_cifseeds prepares the pass (block table, key copy) and leaves the pass itself to the caller
size_t tie_big_bytes(int cells);
size_t tie_small_bytes(int F, int HW);
int seed_tie_order();
hipError_t launch_cifseeds(const float* cif, int B, int F, int H, int W, int stride,
                           const float* cifhr, int hr_rows, int hr_cols, int hr_pitch, const DevParams& p,
                           unsigned long long* keys, int sort_cap, int32_t* seed_count,
                           int32_t* seed_f, float* seed_vxys, hipStream_t st, bool det = false,
                           int32_t* seed_cell = nullptr, int occ_h = 0, int occ_w = 0, bool count_is_zero = false,
                           const ScoredArgs* scored = nullptr, int n_scored = 0, const TieScratch* ties = nullptr,
                           const HrPool* pool = nullptr, const SeedCandidates* cand = nullptr, bool sort_registers = true);
// (`scored`: up to two CafScored list sets built by the SAME launch as the seed sort -- they only share the finished
// map, and the sort's few workgroups leave the chip to them)

// CafScored::fill of one list set (cafscored_impl.hpp)
struct ScoredArgs {
    const float* caf; int A, HW, stride;
    const float* cifhr; int F, hr_rows, hr_cols, hr_pitch;
    const int64_t* skeleton; double score_th, cif_floor; int no_rescore;
    float* lists; int32_t* counts;
    const unsigned* tile_touch; int touch_words, tiles_x;   // [B][F][touch_words] touched-tile bitmaps of the map (or null)
    const int32_t* hr_slot; int hr_tpp;       // pooled map: [B][F][hr_tpp] slot tables (null: dense map)
    size_t hr_image_stride;                   // floats between the maps of two images (dense: F * rows * pitch; pooled: cap * tile)
    float* chunk_bbox; int nb, nb_stride;     // nb: chunks per list that get a box (the first kListBboxChunks for the caf_th
                                              // set; all of them for the force-complete set); nb_stride: boxes per list in memory
    int planes;                               // B * A
};

ScoredArgs make_scored_args(const float* caf, int B, int A, int cH, int cW, int cstride,
                            const float* cifhr, int F, int hr_rows, int hr_cols, int hr_pitch,
                            const int64_t* skeleton, double score_th, double cif_floor, int no_rescore,
                            float* lists, int32_t* counts, float* chunk_bbox, int bbox_chunks, int bbox_stride,
                            const unsigned* tile_touch = nullptr, const HrPool* pool = nullptr);

hipError_t launch_cafscored(const ScoredArgs& s, hipStream_t st);
hipError_t launch_cafscored2(const ScoredArgs& s, const ScoredArgs& s2, hipStream_t st);   // two list sets, one read of the field
hipError_t launch_cafscored(const float* caf, int B, int A, int cH, int cW, int cstride,
                            const float* cifhr, int F, int hr_rows, int hr_cols, int hr_pitch,
                            const int64_t* skeleton, double score_th, double cif_floor, int no_rescore,
                            float* lists, int32_t* counts, hipStream_t st, float* chunk_bbox = nullptr,
                            int bbox_chunks = 0, int bbox_stride = 0);

// 64-entry chunks of the CAF lists get a bounding box of their (x1, y1) columns (xmin, xmax, ymin, ymax; an empty
// chunk: +inf, -inf, +inf, -inf), written by cafscored next to the list ([list][bbox_chunks][4] floats):
// grow_connection_blend's window test (cifcaf.cpp:54-57) cannot pass for any entry of a chunk whose box misses
// the window, so the association kernels only load the chunks that can matter.  caf_th set: the first
// kListBboxChunks chunks of a list (crowded images have 3-4), kept in LDS by the seed kernel.  Force-complete set (caf_th
// 0.001 keeps most cells of a field: ~100 chunks): every chunk, tested where cafscored wrote them (L2).  Lists of
// more than kListBboxMax chunks (fields of more than 16 320 cells) are scanned without boxes.
constexpr int kListBboxChunks = 16;
constexpr int kListBboxMax = 255;

// zero-fill on the stream with a kernel of this library (the runtime's memset / small-copy nodes are the
// one thing that faulted when the decode was replayed as a captured HIP graph); bytes % 4 == 0
hipError_t launch_zero(void* dst, size_t bytes, hipStream_t st);

// the seed sort's arrays, and the scratch of the pass that reproduces the reference's order of equal scores (cifseeds_tie.hpp)
struct SortArgs {
    unsigned long long* keys; int sort_cap, cap; const int32_t* seed_count;
    const float* cif; int F, NC, HW, stride;
    int32_t* seed_f; float* seed_vxys; int32_t* seed_cell; int occ_h, occ_w;
};
struct TieArgs {
    int cells;                       // F * HW: capacity of the per-image arrays
    unsigned char* big; size_t big_stride;       // per image tie_big_bytes(cells): cells, scores and the two stop lists of an
                                                 // image beyond the LDS arrays
    unsigned char* small_; size_t small_stride;  // per image tie_small_bytes(F, HW): the fill kernel's block table, its prefix,
                                                 // segment lists
    int32_t* tie_state;              // [B] or null: 0 no equal scores, 1 re-sorted in libstdc++'s order, -1 not reproduced
};

struct AssocArgs {
    int B, K, F, A, max_ann, n_initial;      // K joints per annotation, F <= K of them with a CIF field
    int hr_rows, hr_cols;
    int occ_h, occ_w;
    int seed_cap, list_cap;
    const int32_t* seed_f; const float* seed_vxys; const int32_t* seed_count;
    const int32_t* seed_cell;  // occupancy cell of the seed: x | y << 12 | box half-width << 24 (seed_cell_pack)
    const float* lists; const int32_t* list_counts;          // caf_th lists
    const float* lists_fc; const int32_t* list_counts_fc;    // force-complete lists (or null)
    const float* list_bbox;  // [B][A][2][bbox_chunks][4] chunk boxes of `lists` (or null)
    const float* list_bbox_fc;  // ... of `lists_fc` (or null)
    int bbox_chunks;
    const int32_t* hr_overflow; // [B] (or null): 1 = the image's CIF map did not fit its tile pool -- the image is flagged failed (status -2)
    int tie_fused;              // 1: every workgroup first puts its image's seeds of equal score into the reference's order (tie, tie_sort)
    TieArgs tie; SortArgs tie_sort;
    int timing;                 // 1: the coordinator also fills the tick counters of its phases (statistics slots 12, 17-20)
    int collide;                // 1: a growth that assigns a joint inside the same joint's box of an earlier live candidate is stopped (advisory)
    int inherit;                // 1: a candidate inherits the predictions of a growth stopped because of it (advisory; see cifcaf.hip)
    int help;                   // 1: idle growers evaluate connections of the growth that holds the head seed (scan helpers; exact, see cifcaf.hip)
    int spec;                   // 1: the growers walk the skeleton level by level in batched scans first and the search takes connection values from that memo (exact; see cifcaf.hip)
    int dedup;                  // 1: later seeds of an occupancy cell already seen are dropped at the pool refill (exact; see cifcaf.hip)
    const float* caf_raw; int caf_w; float caf_stride;   // the CAF field tensor itself [B][A][8][list_cap] (predict_pose reads single cells of it)
    int coll_shift;             // collision stops: how close to the centre of the earlier candidate's joint box (extent >> shift; 0 = anywhere inside)
    int lookahead;              // 1: (large skeletons) the first free seed outside every box in flight enters the pool ahead of the scan (see cifcaf.hip)
    float predict_min_v;
    int predict; float predict_th;   // 1: a growth first walks the skeleton through single cells of the raw field and publishes the boxes of the joints it expects (advisory; see cifcaf.hip)
    int prededup;               // 1: ... and by the whole workgroup before the coordinator starts (needs dedup; exact; see cifcaf.hip)
    int32_t* fc_meta;           // [B, 4] seed kernel -> force-complete kernel: poses stored, dropped, failed, workgroup counter
    long long watchdog_ticks;   // 10-ns ticks after which every wait inside one launch gives up (status -1)
    int32_t* queue_order; int32_t* queue_head;   // [B] images by seed count, most first + the queue's head (or null: one workgroup per image)
    int max_growers, fc_split;  // opa_debug: at most this many growing waves (0: as many as fit); force-complete workgroups per image (0: automatic)
    unsigned* occ;           // occupancy bitmap [B][occ_image_words]: per image [F][occ_h][(occ_w+31)/32] words, zeroed by the kernel
    size_t occ_image_words;
    int32_t* stats;          // [B, 24] statistics of the association (or null), see include/openpifpaf_amd.h
    int32_t* trace;          // [B, 64, 4] the first commits of each image (or null): commit / hand-out / done tick, seed | grower << 24
    double* anns;            // [B, max_ann, K, 4] doubles (v,x,y,s) scratch
    int64_t* ann_ids;        // [B, max_ann]
    const float* initial; const int64_t* initial_ids;
    float* out; int64_t* out_ids; int32_t* out_count;
    int32_t* status;         // [B] debug/overflow flags
};
// TieArgs / SortArgs of the tie pass the way launch_cifseeds builds them (for a caller that runs the pass itself: TieScratch::defer)
void make_tie_args(TieArgs* a, SortArgs* g, unsigned long long* keys, int sort_cap, const int32_t* seed_count, const float* cif,
                   int F, int NC, int HW, int stride, int32_t* seed_f, float* seed_vxys, int32_t* seed_cell, int occ_h, int occ_w,
                   const TieScratch& t);
hipError_t launch_cifseeds_ties(const TieArgs& a, const SortArgs& g, int B, const DevParams& p, hipStream_t st);
hipError_t launch_assoc(const AssocArgs& a, const DevSkeleton& sk, const DevParams& p, hipStream_t st, const opa_debug& dbg);

struct DetArgs {
    int B, F, max_det, occ_h, occ_w, seed_cap;
    const int32_t* seed_f; const float* seed_vxywh; const int32_t* seed_count;
    unsigned char* occ;
    int64_t* categories; float* scores; float* boxes; int32_t* counts;
};
hipError_t launch_cifdet_collect(const DetArgs& a, const DevParams& p, hipStream_t st);

hipError_t launch_bias_act(void* x, const void* bias, const void* res, long long rows, int channels, int dtype,
                           int relu, hipStream_t st);

hipError_t launch_gemm_bias_act(const void* A, const void* W, const void* bias, const void* res, void* out,
                                int M, int N, int K, int relu, hipStream_t st, const void* a_bias = nullptr);

hipError_t launch_gemm_f32x3_bias_act(const float* A, const unsigned short* W3, const float* bias, const float* res, float* out,
                                      int M, int N, int K, int relu, int terms, hipStream_t st, const float* a_bias);
hipError_t launch_gemm2_f32x3_bias_act(const float* A1, int K1, const float* A2, int K2, int batch, int hi, int wi, int stride,
                                       const unsigned short* W3, const float* bias, float* out, int N, int relu, int terms,
                                       hipStream_t st, const float* a_bias);
hipError_t launch_convrows_f32x3(const float* x, int batch, int hp, int wp, int pix, int ho, int wo, int stride, int ntaps, int tap_floats,
                                 const unsigned short* W3, const float* bias, float* out, int N, int relu, int terms, hipStream_t st);
hipError_t launch_conv3x3_f32x3(const float* x, int batch, int hi, int wi, int C, int stride, const unsigned short* W3,
                                const float* bias, float* out, int N, int relu, int terms, hipStream_t st);
hipError_t launch_gemm_f32_bias_act(const float* A, const float* W, const float* bias, const float* res, float* out,
                                    int M, int N, int K, int relu, hipStream_t st, const float* a_bias = nullptr);
hipError_t launch_winograd_f23(const float* x, const float* U, float* y, const float* bias, int N, int H, int W, int Cin,
                               int Cout, int relu, int variant, int nb_major, hipStream_t st);

hipError_t launch_dwconv(const void* x, long long xs, const void* w, const void* bias, void* out, long long os,
                         int B, int H, int W, int C, int K, int S, int dtype, int relu, hipStream_t st);
hipError_t launch_channel_interleave(const void* a, long long as, const void* b, long long bs, void* out,
                                     long long rows, int half, int dtype, hipStream_t st);

hipError_t launch_head_epilogue(const void* conv, int dtype, int B, int Hc, int Wc, int n_fields, int n_comp, int us,
                                int n_conf, int n_vec, unsigned offset_mask, int n_scales, float* out, hipStream_t st);

hipError_t launch_blend(const float* rows, int n, double x, double y, double s, double filter_sigmas,
                        int only_max, double* out4_dev, hipStream_t st);

}  // namespace opa

// ---- device helpers ---------------------------------------------------------
#ifdef __HIPCC__
namespace opa {

// Workgroup barrier that also orders GLOBAL memory between the waves: the occupancy map and the
// annotation scratch live in HBM and are written by one wave and read by the others.  A plain
// __syncthreads() compiles to "s_waitcnt lgkmcnt(0); s_barrier" -- stores may still be in flight
// when the barrier releases (observed: waves disagreeing on an occupancy test, then running one
// barrier apart).  Release = wait for this wave's stores to be acknowledged by the L2 (workgroup scope: all
// waves of a workgroup sit behind the same L2; an agent-scope release would also write the whole L2 back,
// `buffer_wbl2`, which costs microseconds per call and 170 us when 600 workgroups do it at once), acquire =
// drop stale L1 lines (agent scope: atomics are performed at the L2 and do not update the L1).
__device__ __forceinline__ void sync_global() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0x0F70);              // vmcnt(0): on gfx9 stores count too, until the L2 has them
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

// Occupancy cell of a seed and the half-width of the box its joint occupies (occupancy.cpp:13-43),
// packed for the association kernel's seed pool: x | y << 12 | half << 24.
__device__ __forceinline__ int seed_cell_pack(const DevParams& p, int occ_h, int occ_w, double x, double y, double sigma);

__device__ __forceinline__ int lane_id() { return threadIdx.x & (kWave - 1); }

__device__ __forceinline__ long long clampll(long long v, long long lo, long long hi) {
    return v < lo ? lo : (v > hi ? hi : v);
}

// float -> int64 with x86 cvttss2si semantics for the values that can occur
// (finite, |v| < 2^63): truncation toward zero.
__device__ __forceinline__ long long trunc_ll(float v) { return (long long)v; }
__device__ __forceinline__ long long trunc_ll(double v) { return (long long)v; }

__device__ __forceinline__ int seed_cell_pack(const DevParams& p, int occ_h, int occ_w, double x, double y, double sigma) {
    if (p.occupancy_inv_reduction != 0.0) {
        x *= p.occupancy_inv_reduction; y *= p.occupancy_inv_reduction;
        sigma = fmax(p.occupancy_min_scale_reduced, sigma * p.occupancy_inv_reduction);
    } else if (p.occupancy_reduction != 1.0) {
        x /= p.occupancy_reduction; y /= p.occupancy_reduction;
        sigma = fmax(p.occupancy_min_scale_reduced, sigma / p.occupancy_reduction);
    }
    const int xi = (int)clampll(trunc_ll(x), 0, occ_w - 1), yi = (int)clampll(trunc_ll(y), 0, occ_h - 1);
    const int half = (int)fmin(fmax(sigma, 1.0), 255.0);
    return xi | (yi << 12) | (half << 24);
}

// Reference cifhr_value (cif_seeds.cpp:17-30 == caf_scored.cpp:15-26) on the raw
// revision-1 buffer: 0 = untouched (-> default), else 1 + value.
// `touch` (or null): the image's bitmap of the map's 32x64 tiles that this call's CIF cells reach
// ([F][touch_words] words, written by cif_active_kernel): a pixel of any other tile is 0.0 in the buffer (never
// written, or zeroed by the lazy clear), i.e. "untouched", so its value is known without the gather -- most of a
// map is such tiles, and a random 4-byte gather costs a whole memory line.
// `slot` (or null): the map is a POOL of 32x64 tiles (decode path): `hr_image` is the image's pool, `slot` its table
// [F][tpp] tile -> slot (negative: no cell reached the tile, i.e. untouched) -- one small load instead of the bitmap test.
__device__ __forceinline__ float cifhr_value(const float* hr_image, int F, int rows, int cols, int pitch,
                                             long long f, float x, float y, float default_value,
                                             const unsigned* touch = nullptr, int touch_words = 0, int tiles_x = 0,
                                             const int32_t* slot = nullptr, int tpp = 0) {
    const float max_x = (float)((double)(float)cols - 0.51);
    const float max_y = (float)((double)(float)rows - 0.51);
    if (f >= F || (double)x < -0.49 || (double)y < -0.49 || x > max_x || y > max_y) return default_value;
    const long long yi = (long long)((double)y + 0.5);
    const long long xi = (long long)((double)x + 0.5);
    float raw;
    if (slot) {
        const int t = (int)(yi / kHrTileH) * tiles_x + (int)(xi / kHrTileW);
        const int sl = slot[(size_t)f * tpp + t];
        if (sl < 0) return default_value;
        raw = hr_image[(size_t)sl * (kHrTileH * kHrTileW) + (size_t)(yi % kHrTileH) * kHrTileW + (size_t)(xi % kHrTileW)];
    } else {
        if (touch) {
            const int t = (int)(yi / kHrTileH) * tiles_x + (int)(xi / kHrTileW);
            if (!((touch[(size_t)f * touch_words + (t >> 5)] >> (t & 31)) & 1u)) return default_value;
        }
        raw = hr_image[((size_t)f * rows + yi) * pitch + xi];
    }
    const float value = (float)((double)raw - 1.0);
    if ((double)value < 0.0) return default_value;
    return value;
}

}  // namespace opa
#endif
