// CifCaf greedy keypoint association + keypoint NMS on gfx950.
//
// Replaces reference CifCaf::call_with_initial_annotations, _grow,
// _frontier_add_from, _connection_value, grow_connection_blend, _force_complete,
// _flood_fill (csrc/src/cifcaf.cpp:32-449), Occupancy (occupancy.cpp:13-79) and
// NMSKeypoints::call (nms_keypoints.cpp:17-70).
//
// One workgroup per image (images are the data-parallel unit).  Per image the reference is one serial
// dependency chain -- seed k is skipped iff an earlier pose occupies its cell -- but the GROWTH of a
// pose from a seed reads only the CAF lists and the pose's own joints, never the occupancy map
// (cifcaf.cpp:265-411).  So poses are grown SPECULATIVELY by several wavefronts at once and only the
// accept/reject decision stays sequential.  The workgroup is a small in-order-commit machine:
//
//   wave 0, the coordinator, owns everything sequential: the pool of up to 256 / 512 live, undecided seeds
//       (4 or 8 slots per lane, kPoolSlots; a seed is fetched and tested against the occupancy bitmap ONCE, when
//       it enters the pool), the hand-out of candidates to idle growers, and the commits;
//   waves 1.., the growers, poll a task slot in LDS, grow the pose of the seed they are handed (best-first
//       search with the reference's lazy frontier, no barriers), leave pose + occupancy boxes in their
//       private LDS block and report DONE.  There is no round barrier: a grower that finishes is handed
//       the next candidate at once.
//   predictions keep the growers on different people.  A grower PUBLISHES the occupancy box of every joint the
//       moment it assigns it (the joint is final from then on), and tests the coordinator's pool -- mirrored in
//       LDS -- against the box: pooled seeds that come later in seed order and lie inside are "shadowed" (one bit
//       per pool slot and grower).  A shadowed seed dies if that candidate is accepted, so the coordinator does not
//       hand it out, and if it is being grown already it stops the growth (the seed stays pooled).  The prediction
//       is wrong when the candidate dies itself -- covered by a pose committed before it, typically an overlapping
//       person -- and then costs nothing but time: the shadow bits of a dead candidate are ignored and the seed is
//       handed out after all;
//   commit is strictly in seed order: the HEAD = the smallest-index live pooled seed.  Everything before it is
//       decided, so it is free exactly when the sequential loop would find it free (:211); a candidate that
//       shadows it would be the head itself.  When the head's growth is DONE the pose is accepted: every pooled
//       seed inside one of its joint boxes dies (box containment = what the map would say, occupancy.cpp:13-43),
//       growths of dead seeds are cancelled (they poll a flag), the boxes go to the bitmap for the seeds that enter
//       the pool later.  A head that is not being grown (its growth was stopped on a prediction that did not come
//       true) is handed out then; if every grower holds a later result, the latest one is dropped and redone.  So
//       the accepted seeds are exactly those of the sequential loop, with exactly the poses it grows, whatever
//       the predictions and the interleaving (tests run the same inputs with 1, 3 and all growers).
//
// Inside a growth the wave uses its 64 lanes where the reference has inner loops:
// grow_connection_blend scans a CAF candidate list 64 entries per lane-step (coalesced
// SoA planes from L2, ALL loads of a scan in flight at once, scores kept in registers,
// top-1/top-2 by DPP row-op reductions reproducing the reference's ">=" / ">" position
// tie rules).  The frontier is an exact re-implementation of the binary max-heap behind
// std::priority_queue (sift-up on push, sift-to-leaf + sift-up on pop), because equal
// priorities are the norm (all edges leaving one joint share the bound sqrt(v); every
// flood-filled joint carries 1e-5) and the pop order decides results; for skeletons that fit a
// wave (K <= 64, 2A <= 64) pose, frontier and heap live in VGPR lanes (readlane), else in LDS.
// Force-complete growth and flood fill run one pose per grower; keypoint NMS needs no map at all (box
// containment per field, one field per wave).  The occupancy bitmap is written and read by the coordinator
// only; the pose scratch in HBM is written by the coordinator and read by everyone after sync_global().
//
// Joint confidences are double like the reference's Joint struct; every
// float/double promotion follows the reference operation by operation and the
// library is built with -ffp-contract=off.
#include "cifseeds_tie.hpp"

#include <cstdlib>

namespace opa {

constexpr int kAssocWavesDefault = 12;   // waves per workgroup of the association kernel (OPA_ASSOC_WAVES = 8 | 12 | 16)
constexpr int kBlendChunks = 8;
// Seed-pool slots per coordinator lane.  Every serial step of the coordinator (commit test, head search, hand-out) and every
// published joint of a grower walks ALL slots, so the pool is as small as the window of live seeds needs to be:
//   skeletons whose growth state lives in register lanes (COCO): 4 slots = 256 seeds.  Round 5: with the seed list reduced to the
//     first seed of every cell before the coordinator starts (below), 256 DISTINCT cells reach as far into a crowded image as 512
//     undeduped seeds did; measured 8 / 4 / 2 / 1 slots: 533 / 503 / 518 / 556 us (COCO batch 32, profiles/r5/assoc_pool_slots.log);
//   large skeletons (LDS variant, wholebody): 8.  16 (a person has a thousand seeds: a window of several people) was measured in
//     round 4: the image it was meant for gained 15 %, the batch lost 4 % -- every published joint tests twice the slots, and the
//     pool's LDS costs the eleventh grower.
#ifndef OPA_POOL_SLOTS
#define OPA_POOL_SLOTS 4
#endif
#ifndef OPA_POOL_SLOTS_LDS
#define OPA_POOL_SLOTS_LDS 8
#endif
constexpr int kPoolSlots = OPA_POOL_SLOTS;
// Who hands a candidate to an idle grower (-DOPA_ASSOC_SELFSERVE=1 builds the second variant; build.py builds it as
// lib/libopenpifpaf_amd_selfserve.so and tests/test_gpu_selfserve.py decodes through it).  0, the default: the coordinator (one
// more serial step of the one wave everything waits for, 0.7 us per hand-out, a fifth of its time on a crowded image).  1 (round
// 5): the idle grower itself -- it looks at the mirrored pool the way the coordinator did (smallest eligible seed index, not
// predicted dead, tested by every candidate in flight, not inside the seed box of a candidate that has not published yet) and
// CLAIMS the slot with a compare-and-swap on the slot's owner word; the coordinator learns who grows what from those words.
// Which seeds are grown when is advisory either way: the commit decides, in seed order, from final boxes.  Measured (COCO batch
// 32, profiles/r5/assoc_selfserve.log): the coordinator then waits for the head's growth 60 % of its time (305 of 513 us on the
// crowded image, hand-out 88 -> 0, other 127 -> 95) -- and the launch lasts as long as before (508 against 505 us): what is left
// is the chain of growths itself, not the wave that serialises them.  As a run-time switch the second path costs the default one
// 1.3 % (eight spilled registers in the growers): compiled out by default; statistics slot 5 is not counted in that variant.
#ifndef OPA_ASSOC_SELFSERVE
#define OPA_ASSOC_SELFSERVE 0
#endif
constexpr bool kSelfServe = OPA_ASSOC_SELFSERVE != 0;
constexpr int kPoolSlotsLds = OPA_POOL_SLOTS_LDS;
       // list entries per lane held in registers by the single-pass scan

// LDS words shared between the coordinator and the growers: plain loads/stores made atomic at workgroup
// scope (release = this wave's earlier LDS writes are visible before the flag, acquire = the reverse).
__device__ __forceinline__ int flag_load(const int* p) {
    return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ int flag_peek(const int* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void flag_store(int* p, int v) {
    __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// set bits of a ballot below this lane: v_mbcnt_lo / v_mbcnt_hi on the scalar mask -- no (1 << lane) - 1 pair held in
// vector registers through the scans
__device__ __forceinline__ int prefix_count(unsigned long long m) {
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}

// 7 SoA planes: c,x1,y1,x2,y2,s1,s2; `bbox` (LDS, or null): (xmin, xmax, ymin, ymax) of the (x1, y1) columns of the
// list's first kListBboxChunks chunks of 64 entries (cafscored writes them, common.hpp)
// `gbbox` (global memory, or null): the boxes of ALL chunks (up to `nb`) where cafscored wrote them
struct ListView { const float* base; int cap; int n; const float4* bbox; const float4* gbbox; int nb; };

// Diagnostic builds only (-DOPA_ASSOC_PHASE_TIMING, tools/gpu/assoc_probe.py): shader-clock time of the growers
// by phase of the search, summed over the growers of an image; printed by the kernel for images 0 and 3.
#ifdef OPA_ASSOC_PHASE_TIMING
constexpr int kPhases = 32;
__shared__ int g_ph[kPhases], g_phn[kPhases];
__shared__ long long g_ph_last[16];
__device__ __forceinline__ void ph_stamp(int k) {
    const long long t = clock64();
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&g_ph[k], (int)(t - g_ph_last[w])); atomicAdd(&g_phn[k], 1);
    }
    __builtin_amdgcn_wave_barrier();
    if ((threadIdx.x & 63) == 0) g_ph_last[w] = clock64();
}
#define PH(k) ph_stamp(k)
#else
#define PH(k)
#endif

extern __shared__ __attribute__((aligned(16))) unsigned char opa_dyn_lds[];   // every kernel's dynamic LDS starts here

struct ImageCtx {
    int K, A, F;                         // F = occupancy fields = n_cif
    int wave;
    const int32_t *adj_off, *slot_info, *adj_first;   // LDS: adjacency range of a joint; per directed-bone slot: start | other << 8 |
                                                      // bone << 16 | forward << 24; first slot of the same (start, other) pair
    const float* lists; const int32_t* list_counts; int list_cap;
    int coll_shift;                      // collision test: within (box extent >> shift) / 2 cells of the box centre (0: anywhere in the box)
    int* n_predicted;                    // LDS counter: joint boxes published from predictions (statistics)
    const float* raw_caf; int raw_w; float raw_stride, predict_th;   // the image's CAF field itself [A][8][H*W] (predict_pose), or null
    unsigned* occ; int occ_h, occ_w, occ_wpr;   // occupancy bitmap [F][occ_h][occ_wpr] (one bit per cell)
    const int* cancel;                   // this grower's cancel flag in LDS (polled between frontier pops), or null
    int aborted;                         // set when a growth stopped because of it
    int* pub; int n_pub;                 // the grower's "boxes published" counter in LDS (or null) and its value
    const int *pool_if, *pool_pack;      // LDS mirror of the coordinator's seed pool [8][64] (slot r of lane l at r*64+l)
    const int* pool_ep;                  // ... and the refill epoch at which each slot got its occupant
    const int* epoch; int* ack;          // the coordinator's refill epoch; this grower's "tested everything up to" answer
    int my_epoch;
    unsigned* shadow_mine;               // [64] bit r of word l: pool slot (r, l) lies in a box this grower published
    int my_idx;                          // index of the seed being grown
    const int* head_g; int prio;         // LDS: the grower that holds the HEAD seed (or -1); whether this wave runs at raised priority
    // private LDS (one block per growing wave)
    float* tgt;                          // [3][kBlendChunks][64] target columns of the list being scanned
    struct OccBox* jbox;                 // occupancy boxes of the grown pose [K]
    double* jv; float *jx, *jy, *js;     // current pose [K]
    unsigned long long* heap;            // [A] nodes: float bits of max_score << 32 | directed-bone slot
    double* e_v; float *e_x, *e_y, *e_s;   // LDS variant: the frontier entry of every bone [A] (v == 0: not computed yet)
    unsigned char* in_frontier;          // [A] (LDS variant)
    int heap_n;
    int max_r;                           // chunks of a list the LDS target area holds (8, or 2 when LDS is short)
    // shared LDS
    const float4* bbox;                  // [2A][kListBboxChunks] chunk boxes of the active list set, or null
    const float4* gbbox; int nb;         // global memory: [2A][nb] boxes of every chunk of the active list set, or null
    int* sh_counts;                      // [2A] list lengths of the active list set
    int n_blend;                         // list scans of this wave (statistics)
    int t_blend, t_blend_mem;            // ticks inside the scans, and of those until the loads had returned
    // speculative batched evaluation (spec_phase): per joint `kind | flags | slot << 8`, per bone `state | slot << 8` and the
    // connection value the bone's scans gave (private LDS, null: off)
    int *sp_j, *sp_b;
    double* sp_v; float *sp_x, *sp_y, *sp_s;
    unsigned char *sp_match, *sp_fromc;  // LDS variant of the growth state: joint assigned with its candidate's values [K]; entry of the bone came from the cache [A]
    int sp_pcap;                         // passing entries the scan area holds during a batch
    // scan helpers (help_*): requests of this growth [kHelpRing], a word per bone [A], the slot each joint was assigned through [K],
    // control words (0 epoch, 1 closed, 2-3 mask of assigned joints for the register variant); null: off
    int *hq, *rq, *jsrc, *hctl;
    int hq_n, h_epoch;                   // requests posted so far, this growth's epoch
    int n_hit, n_miss;                   // statistics: connection values taken from the cache / evaluated on demand
};

__device__ __forceinline__ ListView list_view(const ImageCtx& c, int bone, int dir) {
    ListView v;
    v.base = c.lists + ((size_t)bone * 2 + dir) * 7 * c.list_cap;
    v.cap = c.list_cap;
    v.n = c.sh_counts[bone * 2 + dir];
    v.bbox = c.bbox ? c.bbox + (bone * 2 + dir) * kListBboxChunks : nullptr;
    v.gbbox = c.gbbox ? c.gbbox + (size_t)(bone * 2 + dir) * c.nb : nullptr; v.nb = c.nb;
    return v;
}

// -------------------------------------------------------- grow_connection_blend
// cifcaf.cpp:32-103.  Returns false for the all-zero joint.
struct BlendQuery { double x, y, xlo, xhi, ylo, yhi; float sigma2; float fxlo, fxhi, fylo, fyhi; };
struct BlendResult { double v; float x, y, s; int ok; };

__device__ __forceinline__ BlendQuery make_query(double x, double y, double xy_scale, double filter_sigmas) {
    xy_scale = fmax(xy_scale, 0.5);                            // :44
    const float sigma_filter = (float)(filter_sigmas * xy_scale / 2.0);   // :47
    BlendQuery q;
    q.x = x; q.y = y;
    q.sigma2 = (float)(0.25 * xy_scale * xy_scale);            // :48
    q.xlo = x - (double)sigma_filter; q.xhi = x + (double)sigma_filter;
    q.ylo = y - (double)sigma_filter; q.yhi = y + (double)sigma_filter;
    // The window test (:54-57) compares a float entry with these doubles.  For a float v: v >= lo  <=>  v >= the
    // smallest float >= lo, and v <= hi  <=>  v <= the largest float <= hi -- four float compares per entry
    // instead of two conversions and four double compares, same outcome for every input.  The directed roundings are
    // v_cvt_f32_f64 under the MODE register's round-to-+inf / round-to--inf (both the single and the double/half
    // field are set; one asm block, so no other floating-point instruction can be scheduled into it).
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 4), 5\n\t"
                 "s_nop 2\n\t"
                 "v_cvt_f32_f64 %0, %4\n\t"
                 "v_cvt_f32_f64 %1, %5\n\t"
                 "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 4), 10\n\t"
                 "s_nop 2\n\t"
                 "v_cvt_f32_f64 %2, %6\n\t"
                 "v_cvt_f32_f64 %3, %7\n\t"
                 "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 4), 0\n\t"
                 "s_nop 2"
                 : "=&v"(q.fxlo), "=&v"(q.fylo), "=&v"(q.fxhi), "=&v"(q.fyhi)
                 : "v"(q.xlo), "v"(q.ylo), "v"(q.xhi), "v"(q.yhi));
    return q;
}

// (A NaN x1 / y1 fails these compares and is skipped.  The reference's early-out form `if (x1 < lo) continue; ...` lets it
// through to the score -- but the score is then NaN, and a NaN never satisfies `score >= score_1` or `score > score_2`
// (cifcaf.cpp:65-73): the entry is ignored there too.  Same result; tests/test_gpu_parity.py::test_grow_connection_blend_nan.)
__device__ __forceinline__ bool passes_f(const BlendQuery& q, float x1, float y1) {
    return x1 >= q.fxlo && x1 <= q.fxhi && y1 >= q.fylo && y1 <= q.fyhi;
}

__device__ __forceinline__ bool passes(const BlendQuery& q, float x1, float y1) {
    if ((double)x1 < q.xlo) return false;                      // cifcaf.cpp:54-57
    if ((double)x1 > q.xhi) return false;
    if ((double)y1 < q.ylo) return false;
    if ((double)y1 > q.yhi) return false;
    return true;
}

__device__ __forceinline__ float score_of(const BlendQuery& q, float x1, float y1, float c) {
    const double dx = (double)x1 - q.x, dy = (double)y1 - q.y;
    const float d2 = (float)(dx * dx + dy * dy);               // :60
    return (float)(exp(-0.5 * (double)d2 / (double)q.sigma2) * (double)c);   // :63
}

// Wave-wide max of a 64-bit key with DPP row operations instead of LDS-crossbar shuffles
// (6 register-only steps: xor-1/xor-2 inside quads, half-row mirror, row mirror, then the
// row-15 and row-31 broadcasts; the result lands in lane 63).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned long long dpp_max_step(unsigned long long v) {
    const unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
    const unsigned olo = (unsigned)__builtin_amdgcn_update_dpp((int)lo, (int)lo, CTRL, ROW_MASK, 0xF, false);
    const unsigned ohi = (unsigned)__builtin_amdgcn_update_dpp((int)hi, (int)hi, CTRL, ROW_MASK, 0xF, false);
    const unsigned long long o = ((unsigned long long)ohi << 32) | olo;
    return o > v ? o : v;
}
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
    v = dpp_max_step<0xB1, 0xF>(v);      // quad_perm [1,0,3,2]
    v = dpp_max_step<0x4E, 0xF>(v);      // quad_perm [2,3,0,1]
    v = dpp_max_step<0x141, 0xF>(v);     // row_half_mirror
    v = dpp_max_step<0x140, 0xF>(v);     // row_mirror
    v = dpp_max_step<0x142, 0xA>(v);     // row_bcast:15 -> rows 1 and 3
    v = dpp_max_step<0x143, 0xC>(v);     // row_bcast:31 -> rows 2 and 3
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, 63);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), 63);
    return ((unsigned long long)hi << 32) | lo;
}

__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
    auto step = [](unsigned x, unsigned o) { return o > x ? o : x; };
    v = step(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xF, 0xF, false));
    v = step(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xF, 0xF, false));
    v = step(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x141, 0xF, 0xF, false));
    v = step(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x140, 0xF, 0xF, false));
    v = step(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x142, 0xA, 0xF, false));
    v = step(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x143, 0xC, 0xF, false));
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

// first place = max score, LAST list position among equals (">=", :65).  Scores are
// non-negative floats, so (score bits << 32 | position + 1) orders exactly like that;
// a lane without a candidate contributes key 0.
__device__ __forceinline__ void reduce_first(float& s1, int& i1) {
    const unsigned long long key = i1 < 0 ? 0ull
        : ((unsigned long long)__float_as_uint(s1) << 32) | (unsigned)(i1 + 1);
    const unsigned long long m = wave_max_u64(key);
    s1 = __uint_as_float((unsigned)(m >> 32));
    i1 = (int)(unsigned)(m & 0xffffffffull) - 1;
}
// second place: max score among the rest; among equals the last position before
// i1 if any, else the first after i1 (the outcome of the sequential rule :65-73).
// rank(i) = i < i1 ? n + i : n - i (always >= 1), larger wins.
__device__ __forceinline__ void reduce_second(float& s2, int& r2) {
    const unsigned long long key = r2 < 0 ? 0ull
        : ((unsigned long long)__float_as_uint(s2) << 32) | (unsigned)r2;
    const unsigned long long m = wave_max_u64(key);
    s2 = __uint_as_float((unsigned)(m >> 32));
    r2 = m == 0ull ? -1 : (int)(unsigned)(m & 0xffffffffull);
}

__device__ __forceinline__ BlendResult blend_none() { BlendResult r; r.v = 0.0; r.x = r.y = r.s = 0.f; r.ok = 0; return r; }

__device__ __forceinline__ BlendResult blend_finish(float s1, float s2, bool have2, bool only_max,
                                                    float e1x, float e1y, float e1s, float e2x, float e2y, float e2s) {
    BlendResult r; r.ok = 1;
    e1s = fmaxf(0.0f, e1s);                                    // :78-81
    if (only_max) { r.v = (double)s1; r.x = e1x; r.y = e1y; r.s = e1s; return r; }
    if (!have2 || (double)s2 < 0.01 || (double)s2 < 0.5 * (double)s1) {     // :84-85
        r.v = 0.5 * (double)s1; r.x = e1x; r.y = e1y; r.s = e1s; return r;
    }
    e2s = fmaxf(0.0f, e2s);                                    // :88-91
    const double ddx = (double)(e1x - e2x), ddy = (double)(e1y - e2y);
    const float blend_d2 = (float)(ddx * ddx + ddy * ddy);     // :93
    if ((double)blend_d2 > ((double)e1s * (double)e1s) / 4.0) {             // :94-95
        r.v = 0.5 * (double)s1; r.x = e1x; r.y = e1y; r.s = e1s; return r;
    }
    const float ssum = s1 + s2;                                // :97-102
    r.v = 0.5 * (double)ssum;
    r.x = (s1 * e1x + s2 * e2x) / ssum;
    r.y = (s1 * e1y + s2 * e2y) / ssum;
    r.s = (s1 * e1s + s2 * e2s) / ssum;
    return r;
}

// Lists of up to 64*R entries: ONE memory round trip.  Every lane issues all its loads (6 planes x R
// chunks, global address space) back to back and an empty asm pins them there (hipcc otherwise sinks each
// load into the branch that uses it and waits for them one by one).  The window test (:54-57) runs on all
// R chunks, and the few entries that pass -- a confidence blob's worth -- are COMPACTED, in list order,
// into the first lanes through 1 KB of LDS, so that the double-precision exp (:63) is evaluated once per
// scan instead of once per chunk that holds a passing entry.  Compaction keeps list order, so "position"
// in the tie rules is the lane: first place = max score, LAST lane among equals (">=", :65); second place
// = max score among the rest, among equals the last lane before the first if any, else the first after it
// (the outcome of the sequential rule :65-73).  Both are a 32-bit DPP max plus a ballot.  More than 64
// passing entries (rare: a window holding that many cells): the streamed two-pass scan below.
typedef __attribute__((address_space(1))) const float gfloat;

typedef __attribute__((address_space(3))) float lfloat;
typedef __attribute__((address_space(1))) const int gint;
typedef __attribute__((address_space(3))) int lint;

constexpr int kTgtFloats = 3 * kBlendChunks * kWave;     // target columns (x2, y2, s2) of the list being scanned
constexpr int kBlendLdsFloats = kTgtFloats + 4 * kWave;  // + the compacted (x1, y1, c, position) of a scan

// More than 64 entries pass the window test (rare: a window holding that many cells): blend_cached reports it
// (ok < 0) and the caller takes one of the streamed scans below.
__device__ __forceinline__ BlendResult blend_overflow() { BlendResult r; r.v = 0.0; r.x = r.y = r.s = 0.f; r.ok = -1; return r; }
__device__ __forceinline__ BlendResult blend_streamed(const ListView& L, const BlendQuery& q, bool only_max);

// `chunks`: the R chunks of the list to look at, ascending, 8 bits each (0xff: none) -- all of them
// (kDenseChunks) or, for a list with chunk boxes, those whose box meets the window.
constexpr unsigned long long kDenseChunks = 0x0706050403020100ull;
template <int R>
__device__ __forceinline__ BlendResult blend_cached(const ListView& L, const BlendQuery& q, bool only_max, float* tgt, int* t_mem,
                                                    unsigned long long chunks = kDenseChunks) {
    const int lane = lane_id();
    const long long t_issue = t_mem ? wall_clock64() : 0;
    PH(1);
    const gfloat* g = (const gfloat*)L.base;
    // The target columns (x2, y2, s2) are needed for two entries only: they travel HBM/L2 -> LDS
    // directly (global_load_lds, no VGPRs), in flight together with the register loads below; chunk r of
    // column k lands at tgt[(k * R + r) * 64 + lane].
    lfloat* t3 = (lfloat*)tgt;
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int i = (int)((chunks >> (8 * r)) & 0xffull) * kWave + lane;
        const int ii = i < L.n ? i : 0;
        __builtin_amdgcn_global_load_lds(g + 3 * L.cap + ii, t3 + (0 * R + r) * kWave, 4, 0, 0);
        __builtin_amdgcn_global_load_lds(g + 4 * L.cap + ii, t3 + (1 * R + r) * kWave, 4, 0, 0);
        __builtin_amdgcn_global_load_lds(g + 6 * L.cap + ii, t3 + (2 * R + r) * kWave, 4, 0, 0);
    }
    float x1[R], y1[R], cc[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int i = (int)((chunks >> (8 * r)) & 0xffull) * kWave + lane;
        const int ii = i < L.n ? i : 0;
        x1[r] = g[1 * L.cap + ii]; y1[r] = g[2 * L.cap + ii]; cc[r] = g[ii];
    }
#pragma unroll
    for (int r = 0; r < R; r++) asm volatile("" : "+v"(x1[r]), "+v"(y1[r]), "+v"(cc[r]) :: "memory");
    if (t_mem) *t_mem += (int)(wall_clock64() - t_issue);
    PH(2);
    bool have; float sc = 0.0f; int pos = 0;
    if constexpr (R == 1) {
        // one chunk: its lanes ARE in list order -- no compaction, the score is evaluated where the entry was loaded
        const int i = (int)(chunks & 0xffull) * kWave + lane;
        have = i < L.n && passes_f(q, x1[0], y1[0]);
        if (__ballot(have) == 0ull) {                          // :76
            __builtin_amdgcn_s_waitcnt(0x0F70);                // vmcnt(0): the LDS loads must land before the area is reused
            return blend_none();
        }
        PH(3);
        if (have) sc = score_of(q, x1[0], y1[0], cc[0]);
        pos = lane;
    } else {
        float* cx = tgt + 3 * R * kWave; float* cy = cx + kWave; float* cv = cy + kWave; int* ci = (int*)(cv + kWave);   // behind the R chunks' targets
        int cnt = 0;
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int i = (int)((chunks >> (8 * r)) & 0xffull) * kWave + lane;
            const bool pass = i < L.n && passes_f(q, x1[r], y1[r]);
            const unsigned long long m = __ballot(pass);
            if (m == 0ull) continue;
            const int slot = cnt + prefix_count(m);
            // position among the chunks looked at: ascending like the list index, and where the targets landed
            if (pass && slot < kWave) { cx[slot] = x1[r]; cy[slot] = y1[r]; cv[slot] = cc[r]; ci[slot] = r * kWave + lane; }
            cnt += __popcll(m);
        }
        if (cnt == 0) {                                        // :76
            __builtin_amdgcn_s_waitcnt(0x0F70);                // vmcnt(0): the LDS loads must land before the area is reused
            return blend_none();
        }
        if (cnt > kWave) {
            __builtin_amdgcn_s_waitcnt(0x0F70);
            return blend_overflow();
        }
        wave_sync();
        PH(3);
        have = lane < cnt;
        if (have) { sc = score_of(q, cx[lane], cy[lane], cv[lane]); pos = ci[lane]; }
    }
    asm volatile("" : "+v"(sc));
    PH(4);
    const unsigned b1 = have && sc > 0.0f ? __float_as_uint(sc) : 0u;
    const unsigned s1b = wave_max_u32(b1);
    if (s1b == 0u) {                                           // :76
        __builtin_amdgcn_s_waitcnt(0x0F70);
        return blend_none();
    }
    const int l1 = 63 - __builtin_clzll(__ballot(b1 == s1b));
    const int i1 = __builtin_amdgcn_readlane(pos, l1);
    const float s1 = __uint_as_float(s1b);
    float s2 = 0.0f; int i2 = i1; bool have2 = false;
    if (!only_max) {
        const unsigned b2 = lane != l1 ? b1 : 0u;
        const unsigned s2b = wave_max_u32(b2);
        if (s2b != 0u) {
            const unsigned long long m2 = __ballot(b2 == s2b);
            const unsigned long long before = m2 & ((1ull << l1) - 1ull);
            const int l2 = before ? 63 - __builtin_clzll(before) : __builtin_ctzll(m2);
            i2 = __builtin_amdgcn_readlane(pos, l2); s2 = __uint_as_float(s2b); have2 = true;
        }
    }
    PH(5);
    __builtin_amdgcn_s_waitcnt(0x0F70);                        // vmcnt(0): target columns are in LDS
    wave_sync();
    const float e1x = tgt[0 * R * kWave + i1], e1y = tgt[1 * R * kWave + i1], e1s = tgt[2 * R * kWave + i1];
    const float e2x = tgt[0 * R * kWave + i2], e2y = tgt[1 * R * kWave + i2], e2s = tgt[2 * R * kWave + i2];
    return blend_finish(s1, s2, have2, only_max, e1x, e1y, e1s, e2x, e2y, e2s);
}

// Any list length (force-complete lists at caf_th 0.001 and all-active fields hold thousands of
// entries): two passes over the list, each in groups of 4 chunks whose loads are issued together
// and pinned like in blend_cached, so a pass costs one memory round trip per 256 entries instead of
// one per 64.  Scores are recomputed in pass 2.
__device__ __forceinline__ BlendResult blend_streamed(const ListView& L, const BlendQuery& q, bool only_max) {
    constexpr int G = 4;
    const int lane = lane_id();
    const gfloat* g = (const gfloat*)L.base;
    float s1 = 0.0f; int i1 = -1;
    for (int base = 0; base < L.n; base += G * kWave) {
        float x1[G], y1[G], cc[G];
#pragma unroll
        for (int r = 0; r < G; r++) {
            const int i = base + r * kWave + lane;
            const int ii = i < L.n ? i : 0;
            x1[r] = g[1 * L.cap + ii]; y1[r] = g[2 * L.cap + ii]; cc[r] = g[ii];
        }
#pragma unroll
        for (int r = 0; r < G; r++) asm volatile("" : "+v"(x1[r]), "+v"(y1[r]), "+v"(cc[r]) :: "memory");
#pragma unroll
        for (int r = 0; r < G; r++) {
            const int i = base + r * kWave + lane;
            if (i < L.n && passes_f(q, x1[r], y1[r])) {
                const float sc = score_of(q, x1[r], y1[r], cc[r]);
                if (sc >= s1) { s1 = sc; i1 = i; }
            }
        }
    }
    reduce_first(s1, i1);
    if (s1 == 0.0f || i1 < 0) return blend_none();             // :76
    float s2 = 0.0f; int r2 = -1;
    if (!only_max) {
        for (int base = 0; base < L.n; base += G * kWave) {
            float x1[G], y1[G], cc[G];
#pragma unroll
            for (int r = 0; r < G; r++) {
                const int i = base + r * kWave + lane;
                const int ii = i < L.n ? i : 0;
                x1[r] = g[1 * L.cap + ii]; y1[r] = g[2 * L.cap + ii]; cc[r] = g[ii];
            }
#pragma unroll
            for (int r = 0; r < G; r++) asm volatile("" : "+v"(x1[r]), "+v"(y1[r]), "+v"(cc[r]) :: "memory");
#pragma unroll
            for (int r = 0; r < G; r++) {
                const int i = base + r * kWave + lane;
                if (i >= L.n || i == i1 || !passes_f(q, x1[r], y1[r])) continue;
                const float sc = score_of(q, x1[r], y1[r], cc[r]);
                if (!(sc > 0.0f)) continue;
                const int rank = i < i1 ? L.n + i : L.n - i;
                if (sc > s2 || (sc == s2 && rank > r2)) { s2 = sc; r2 = rank; }
            }
        }
        reduce_second(s2, r2);
    }
    const bool have2 = r2 >= 0;
    const int i2 = have2 ? (r2 >= L.n ? r2 - L.n : L.n - r2) : i1;
    const float e1x = g[3 * L.cap + i1], e1y = g[4 * L.cap + i1], e1s = g[6 * L.cap + i1];
    const float e2x = g[3 * L.cap + i2], e2y = g[4 * L.cap + i2], e2s = g[6 * L.cap + i2];
    return blend_finish(s1, s2, have2, only_max, e1x, e1y, e1s, e2x, e2y, e2s);
}

// The one real (non-inlined) device function of the kernel: everything is passed and
// returned by value in registers.
__device__ __forceinline__ BlendResult blend_impl(const float* base, int cap, int n, double x, double y,
                                               double xy_scale, double filter_sigmas, int only_max, float* tgt,
                                               int* t_mem = nullptr, int max_r = kBlendChunks) {
    if (n <= 0) return blend_none();
    ListView L; L.base = base; L.cap = cap; L.n = n; L.bbox = nullptr; L.gbbox = nullptr; L.nb = 0;
    const BlendQuery q = make_query(x, y, xy_scale, filter_sigmas);
    if (n <= kWave) return blend_cached<1>(L, q, only_max != 0, tgt, t_mem);
    BlendResult r = blend_overflow();
    if (n <= 2 * kWave) r = blend_cached<2>(L, q, only_max != 0, tgt, t_mem);
    else if (n <= 4 * kWave && max_r >= 4) r = blend_cached<4>(L, q, only_max != 0, tgt, t_mem);
    else if (n <= kBlendChunks * kWave && max_r >= kBlendChunks) r = blend_cached<kBlendChunks>(L, q, only_max != 0, tgt, t_mem);
    if (r.ok >= 0) return r;
    return blend_streamed(L, q, only_max != 0);
}

// A list with chunk boxes: only the chunks whose box meets the window are loaded (no entry of any other chunk can
// pass the window test, cifcaf.cpp:54-57); list order among the loaded chunks is kept, so the tie rules hold.
template <int R>
__device__ __forceinline__ unsigned long long pack_chunks(unsigned m) {
    unsigned long long chunks = 0ull;
#pragma unroll
    for (int r = 0; r < R; r++) {
        const unsigned cidx = m ? (unsigned)__builtin_ctz(m) : 0xffu;
        m &= m - 1u;
        chunks |= (unsigned long long)cidx << (8 * r);
    }
    return chunks;
}
__device__ __forceinline__ BlendResult blend_boxed(const ListView& L, double x, double y, double xy_scale,
                                                   double filter_sigmas, float* tgt) {
    const BlendQuery q = make_query(x, y, xy_scale, filter_sigmas);
    const int lane = lane_id();
    const float4 bb = L.bbox[lane & (kListBboxChunks - 1)];
    const bool hit = lane * kWave < L.n && lane < kListBboxChunks &&
                     bb.x <= q.fxhi && bb.y >= q.fxlo && bb.z <= q.fyhi && bb.w >= q.fylo;
    const unsigned m = (unsigned)__ballot(hit);
    const int nh = __popc(m);
    if (nh == 0) return blend_none();                          // :76
    if (nh == 1) return blend_cached<1>(L, q, false, tgt, nullptr, pack_chunks<1>(m));
    BlendResult r = blend_overflow();
    if (nh == 2) r = blend_cached<2>(L, q, false, tgt, nullptr, pack_chunks<2>(m));
    else if (nh <= 4) r = blend_cached<4>(L, q, false, tgt, nullptr, pack_chunks<4>(m));
    else if (nh <= kBlendChunks) r = blend_cached<kBlendChunks>(L, q, false, tgt, nullptr, pack_chunks<kBlendChunks>(m));
    if (r.ok >= 0) return r;
    return blend_streamed(L, q, false);
}

// ---- long lists (more than kListBboxChunks chunks: the force-complete set, all-active fields) ----------------
// The boxes of all chunks are tested where cafscored wrote them: lane l looks at chunks l, l + 64, l + 128,
// l + 192 (<= kListBboxMax chunks), one 16-byte load each, all in flight together.  Up to kBlendChunks hit chunks
// take the one-round-trip scan above.  More hit chunks, or more than 64 passing entries (the force-complete window is
// 4 sigma wide: dozens of cells), are COMPACTED over as many round trips as it takes -- (x1, y1, c, list index) of
// every passing entry, in list order, into the wave's LDS area -- so that the double-precision exp runs once per 64
// passing entries instead of twice per chunk, and the top-2 rules are those of the streamed scan on list indices.
struct ChunkMask { unsigned long long m[4]; };
__device__ __forceinline__ int chunk_pop(ChunkMask& k) {       // lowest set chunk index, or -1 (wave-uniform)
#pragma unroll
    for (int g = 0; g < 4; g++)
        if (k.m[g]) { const int ci = g * kWave + __builtin_ctzll(k.m[g]); k.m[g] &= k.m[g] - 1ull; return ci; }
    return -1;
}

constexpr int kCompactCap = kBlendLdsFloats / 4;              // passing entries the LDS area of a wave holds (448)

// blend_streamed over the chunks of `hit` only (ascending, so list positions -- and with them the tie rules --
// are those of the full scan; no entry of another chunk can pass the window test): the last resort, when more than
// kCompactCap entries pass.
__device__ __forceinline__ BlendResult blend_streamed_masked(const ListView& L, const BlendQuery& q, const ChunkMask& hit) {
    constexpr int G = 4;
    const int lane = lane_id();
    const gfloat* g = (const gfloat*)L.base;
    float s1 = 0.0f; int i1 = -1;
    {
        ChunkMask k = hit;
        for (;;) {
            int ci[G];
#pragma unroll
            for (int r = 0; r < G; r++) ci[r] = chunk_pop(k);
            if (ci[0] < 0) break;
            float x1[G], y1[G], cc[G];
#pragma unroll
            for (int r = 0; r < G; r++) {
                const int i = ci[r] * kWave + lane;
                const int ii = ci[r] >= 0 && i < L.n ? i : 0;
                x1[r] = g[1 * L.cap + ii]; y1[r] = g[2 * L.cap + ii]; cc[r] = g[ii];
            }
#pragma unroll
            for (int r = 0; r < G; r++) asm volatile("" : "+v"(x1[r]), "+v"(y1[r]), "+v"(cc[r]) :: "memory");
#pragma unroll
            for (int r = 0; r < G; r++) {
                const int i = ci[r] * kWave + lane;
                if (ci[r] >= 0 && i < L.n && passes_f(q, x1[r], y1[r])) {
                    const float sc = score_of(q, x1[r], y1[r], cc[r]);
                    if (sc >= s1) { s1 = sc; i1 = i; }
                }
            }
        }
    }
    reduce_first(s1, i1);
    if (s1 == 0.0f || i1 < 0) return blend_none();             // :76
    float s2 = 0.0f; int r2 = -1;
    {
        ChunkMask k = hit;
        for (;;) {
            int ci[G];
#pragma unroll
            for (int r = 0; r < G; r++) ci[r] = chunk_pop(k);
            if (ci[0] < 0) break;
            float x1[G], y1[G], cc[G];
#pragma unroll
            for (int r = 0; r < G; r++) {
                const int i = ci[r] * kWave + lane;
                const int ii = ci[r] >= 0 && i < L.n ? i : 0;
                x1[r] = g[1 * L.cap + ii]; y1[r] = g[2 * L.cap + ii]; cc[r] = g[ii];
            }
#pragma unroll
            for (int r = 0; r < G; r++) asm volatile("" : "+v"(x1[r]), "+v"(y1[r]), "+v"(cc[r]) :: "memory");
#pragma unroll
            for (int r = 0; r < G; r++) {
                const int i = ci[r] * kWave + lane;
                if (ci[r] < 0 || i >= L.n || i == i1 || !passes_f(q, x1[r], y1[r])) continue;
                const float sc = score_of(q, x1[r], y1[r], cc[r]);
                if (!(sc > 0.0f)) continue;
                const int rank = i < i1 ? L.n + i : L.n - i;
                if (sc > s2 || (sc == s2 && rank > r2)) { s2 = sc; r2 = rank; }
            }
        }
    }
    reduce_second(s2, r2);
    const bool have2 = r2 >= 0;
    const int i2 = have2 ? (r2 >= L.n ? r2 - L.n : L.n - r2) : i1;
    const float e1x = g[3 * L.cap + i1], e1y = g[4 * L.cap + i1], e1s = g[6 * L.cap + i1];
    const float e2x = g[3 * L.cap + i2], e2y = g[4 * L.cap + i2], e2s = g[6 * L.cap + i2];
    return blend_finish(s1, s2, have2, false, e1x, e1y, e1s, e2x, e2y, e2s);
}

#ifdef OPA_FC_TIMING
// diagnostic builds (tools/gpu/fc_timing.py): scans of the force-complete kernel by the number of chunks their window meets
// (0..31, 32+ in [32]); [33] scans that were compacted, [34] entries compacted -- rows 40.. of image 0's trace
__device__ int* g_fc_hist;
#endif
__device__ __forceinline__ BlendResult blend_compacted(const ListView& L, const BlendQuery& q, const ChunkMask& hit, float* tgt) {
    constexpr int G = 4;
    const int lane = lane_id();
    const gfloat* g = (const gfloat*)L.base;
    float* cx = tgt; float* cy = cx + kCompactCap; float* cv = cy + kCompactCap; int* ci_ = (int*)(cv + kCompactCap);
    int cnt = 0;
    {
        ChunkMask k = hit;
        for (;;) {
            int ci[G];
#pragma unroll
            for (int r = 0; r < G; r++) ci[r] = chunk_pop(k);
            if (ci[0] < 0) break;
            float x1[G], y1[G], cc[G];
#pragma unroll
            for (int r = 0; r < G; r++) {
                const int i = ci[r] * kWave + lane;
                const int ii = ci[r] >= 0 && i < L.n ? i : 0;
                x1[r] = g[1 * L.cap + ii]; y1[r] = g[2 * L.cap + ii]; cc[r] = g[ii];
            }
#pragma unroll
            for (int r = 0; r < G; r++) asm volatile("" : "+v"(x1[r]), "+v"(y1[r]), "+v"(cc[r]) :: "memory");
#pragma unroll
            for (int r = 0; r < G; r++) {
                const int i = ci[r] * kWave + lane;
                const bool pass = ci[r] >= 0 && i < L.n && passes_f(q, x1[r], y1[r]);
                const unsigned long long m = __ballot(pass);
                if (m == 0ull) continue;
                const int slot = cnt + prefix_count(m);
                if (pass && slot < kCompactCap) { cx[slot] = x1[r]; cy[slot] = y1[r]; cv[slot] = cc[r]; ci_[slot] = i; }
                cnt += __popcll(m);
            }
        }
    }
#ifdef OPA_FC_TIMING
    if (lane == 0) { atomicAdd(&g_fc_hist[33], 1); atomicAdd(&g_fc_hist[34], cnt); }
#endif
    if (cnt == 0) return blend_none();                         // :76
    if (cnt > kCompactCap) return blend_streamed_masked(L, q, hit);
    wave_sync();
    float s1 = 0.0f; int i1 = -1;
    for (int e = lane; e < cnt; e += kWave) {                  // ascending list index per lane: ">=" keeps the last among equals
        const float sc = score_of(q, cx[e], cy[e], cv[e]);
        cv[e] = sc;                                            // (this lane's own slot: read again below)
        if (sc >= s1) { s1 = sc; i1 = ci_[e]; }
    }
    reduce_first(s1, i1);
    if (s1 == 0.0f || i1 < 0) return blend_none();             // :76
    float s2 = 0.0f; int r2 = -1;
    for (int e = lane; e < cnt; e += kWave) {
        const int i = ci_[e];
        const float sc = cv[e];
        if (i == i1 || !(sc > 0.0f)) continue;
        const int rank = i < i1 ? L.n + i : L.n - i;
        if (sc > s2 || (sc == s2 && rank > r2)) { s2 = sc; r2 = rank; }
    }
    reduce_second(s2, r2);
    wave_sync();                                               // the area is free for the next scan
    const bool have2 = r2 >= 0;
    const int i2 = have2 ? (r2 >= L.n ? r2 - L.n : L.n - r2) : i1;
    const float e1x = g[3 * L.cap + i1], e1y = g[4 * L.cap + i1], e1s = g[6 * L.cap + i1];
    const float e2x = g[3 * L.cap + i2], e2y = g[4 * L.cap + i2], e2s = g[6 * L.cap + i2];
    return blend_finish(s1, s2, have2, false, e1x, e1y, e1s, e2x, e2y, e2s);
}

__device__ __forceinline__ BlendResult blend_long(const ListView& L, const BlendQuery& q, float* tgt) {
    const int lane = lane_id();
    const int nch = (L.n + kWave - 1) >> 6;                     // <= L.nb <= kListBboxMax
    const float4* gb = L.gbbox;
    float4 bb[4];
#pragma unroll
    for (int g = 0; g < 4; g++) {
        const int ci = g * kWave + lane;
        if (g * kWave < nch) bb[g] = gb[ci < nch ? ci : 0];
    }
    ChunkMask hit;
    int nh = 0;
#pragma unroll
    for (int g = 0; g < 4; g++) {
        hit.m[g] = 0ull;
        if (g * kWave < nch) {
            const int ci = g * kWave + lane;
            hit.m[g] = __ballot(ci < nch && bb[g].x <= q.fxhi && bb[g].y >= q.fxlo && bb[g].z <= q.fyhi && bb[g].w >= q.fylo);
            nh += __popcll(hit.m[g]);
        }
    }
#ifdef OPA_FC_TIMING
    if (lane == 0) atomicAdd(&g_fc_hist[nh < 32 ? nh : 32], 1);
#endif
    if (nh == 0) return blend_none();                          // :76
    if (nh <= kBlendChunks) {
        unsigned long long chunks = ~0ull;                     // 0xff: no chunk
        ChunkMask k = hit;
#pragma unroll
        for (int r = 0; r < kBlendChunks; r++) {
            const int ci = chunk_pop(k);
            if (ci >= 0) chunks = (chunks & ~(0xffull << (8 * r))) | ((unsigned long long)ci << (8 * r));
        }
        BlendResult r;
        if (nh == 1) r = blend_cached<1>(L, q, false, tgt, nullptr, chunks);
        else if (nh == 2) r = blend_cached<2>(L, q, false, tgt, nullptr, chunks);
        else if (nh <= 4) r = blend_cached<4>(L, q, false, tgt, nullptr, chunks);
        else r = blend_cached<kBlendChunks>(L, q, false, tgt, nullptr, chunks);
        if (r.ok >= 0) return r;
    }
    return blend_compacted(L, q, hit, tgt);
}

// LONG: the list set has a box for every chunk in global memory (the force-complete kernel); the seed kernel keeps
// its scan code as small as its lists are (inlined at every scan site, the long-list path cost it 77 spilled VGPRs).
template <bool LONG>
__device__ __forceinline__ BlendResult blend(ImageCtx& c, const ListView& L, double x, double y, double xy_scale,
                                             double filter_sigmas) {
    c.n_blend++;
    if constexpr (LONG) {
        if (L.gbbox && L.n > kWave && L.n <= L.nb * kWave)
            return blend_long(L, make_query(x, y, xy_scale, filter_sigmas), c.tgt);
    }
    if (L.bbox && L.n > kWave && L.n <= kListBboxChunks * kWave) {
#ifdef OPA_ASSOC_PHASE_TIMING
        PH(17);
        const BlendResult r = blend_boxed(L, x, y, xy_scale, filter_sigmas, c.tgt);
        PH(6);
        return r;
#else
        return blend_boxed(L, x, y, xy_scale, filter_sigmas, c.tgt);
#endif
    }
#ifdef OPA_ASSOC_SCAN_TIMING            // four clock reads per scan: diagnostic builds only (tools/gpu/assoc_probe.py)
    const long long t0 = wall_clock64();
    const BlendResult r = blend_impl(L.base, L.cap, L.n, x, y, xy_scale, filter_sigmas, 0, c.tgt, &c.t_blend_mem, c.max_r);
    c.t_blend += (int)(wall_clock64() - t0);
    return r;
#else
#ifdef OPA_ASSOC_PHASE_TIMING
    PH(17);
    const BlendResult r = blend_impl(L.base, L.cap, L.n, x, y, xy_scale, filter_sigmas, 0, c.tgt, nullptr, c.max_r);
    PH(6);
    return r;
#else
    return blend_impl(L.base, L.cap, L.n, x, y, xy_scale, filter_sigmas, 0, c.tgt, nullptr, c.max_r);
#endif
#endif
}

// ------------------------------------------------------------ batched list scans (lane = job)
// grow_connection_blend (cifcaf.cpp:32-103) for up to 64 (list, query) pairs at once.  _connection_value (:349-411) of a
// bone depends on nothing but its start joint, so the bones a new joint adds -- and the bones of every joint that became a
// candidate in the same level of the search -- are scanned TOGETHER: one query setup for all of them (lane = job), the
// chunk boxes of every job tested side by side, the loads of eight (job, chunk) items in flight at once, the passing
// entries of all jobs compacted job by job into the wave's scan area, ONE pass of the double-precision exp over them
// (lane = entry), the top two of every job by the reference's own sequential rule (:65-73, lane = job, each lane walks
// its job's entries in list order), and one round trip for the target columns of the winners.  A batch costs about what
// one scan costs (two memory round trips); the results are those of the single scans bit for bit (same score_of, same
// blend_finish, and the top-2 rule is the reference's loop itself).
// The level walk (spec_phase, below) is compiled in only with -DOPA_ASSOC_WALK (build.build_diagnostic('OPA_ASSOC_WALK', 'walk')):
// measured in round 5 (profiles/r5/rejected_level_walk.log) it is exact and halves the growths started, but a batch of n scans
// costs what n single scans cost, so the growths get slower, not faster; with the switch off the kernels carry none of it.
#ifdef OPA_ASSOC_WALK
constexpr bool kWalk = true;
#else
constexpr bool kWalk = false;
#endif
// ---- scan helpers.  _connection_value (cifcaf.cpp:349-411) of a bone depends on its start joint alone, so it can be computed
// by ANOTHER wave as soon as that joint's values are known -- i.e. as soon as the connection that leads to the joint has
// been evaluated, usually long before the search assigns the joint and pops the bone.  The grower posts "joint b, reached
// through slot u, has these values" (help_post); idle growers look at the requests of the growth that holds the HEAD seed --
// the one the commit waits for -- claim the bones leaving b one at a time (a word per bone, compare-and-swap) and leave
// the connection value in the master's entry array; the master's heap loop, unchanged, takes it from there when it pops
// the bone, provided b was indeed assigned through u (then the values are the ones _connection_value would compute, bit
// for bit: same code, same inputs); anything else it evaluates itself, as before.  The head's growth becomes a chain of
// heap operations with the list scans done beside it.
// Compiled in only with -DOPA_ASSOC_HELPERS: measured in round 5 (profiles/r5/rejected_scan_helpers.log) it is exact (153 GPU
// tests) and changes nothing -- while the head's pose grows, the other growers are busy with growths of their own (the idle
// half of the growers' time lies in the tail of one- and two-joint poses), and the second copy of the scan code costs the
// 12-wave kernel 18 spilled registers (COCO 597 -> 625 us with it on, 558 without it in the build).
#ifdef OPA_ASSOC_HELPERS
constexpr bool kHelp = true;
#else
constexpr bool kHelp = false;
#endif
constexpr int kHelpRing = 16;
constexpr int kRqFree = 0, kRqTaken = 1, kRqOk = 2, kRqRej = 3, kRqMaster = 4;   // word of a bone: state | slot << 3 | source slot << 12 | epoch << 21
constexpr int kSrcStart = 511;           // "source slot" of a joint the pose started with
__device__ __forceinline__ int rq_word(int state, int slot, int src, int ep) { return state | (slot << 3) | (src << 12) | (ep << 21); }
// memo words of the level walk
constexpr int kSpActual = 1, kSpCand = 2, kSpKind = 3, kSpOpen = 4, kSpNext = 8;     // per joint: kind, "its bones are this level's jobs", "became a candidate in this level"
constexpr int kSbOk = 1, kSbRejected = 2, kSbUnknown = 3, kSbPending = 4;            // per bone: connection holds / _connection_value returns the zero joint / not evaluated here / forward scan done, reverse scan outstanding
constexpr int kSpecItems = 128;          // (job, chunk) items of one batch
constexpr int kSpecGroup = 6;            // items whose loads are in flight together
constexpr int kSpecEntryWords = 7;       // a passing entry in the scan area: x1, y1, c -> score, x2, y2, s2, job
struct SpecOut { int ok; double v; float x, y, s; };   // ok: 1 a joint, 0 the all-zero joint (:76), -1 not evaluated (left to the single scans)

__device__ __forceinline__ int rlane_i(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ float rlane_f(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
__device__ __forceinline__ int bperm_i(int v, int src) { return __builtin_amdgcn_ds_bpermute(src << 2, v); }
__device__ __forceinline__ float bperm_f(float v, int src) { return __int_as_float(bperm_i(__float_as_int(v), src)); }
__device__ __forceinline__ double bperm_d(double v, int src) {
    return __hiloint2double(bperm_i(__double2hiint(v), src), bperm_i(__double2loint(v), src));
}
__device__ __forceinline__ int spec_pcap(int tgt_floats) {   // the scan area: items, a segment per job, then seven words per passing entry
    return (tgt_floats - kSpecItems - 2 * kWave) / kSpecEntryWords;
}

__device__ __forceinline__ SpecOut spec_scan_batch(ImageCtx& c, bool active, int li, double x, double y, double s, double filter_sigmas) {
    const int lane = lane_id();
    int* item = reinterpret_cast<int*>(c.tgt);               // job | chunk << 8 | first item of its job << 16
    int* seg = item + kSpecItems;                            // [2][64] where a job's passing entries start / end
    const int pcap = c.sp_pcap;
    float* cx = reinterpret_cast<float*>(seg + 2 * kWave); float* cy = cx + pcap; float* cv = cy + pcap;
    float* tx = cv + pcap; float* ty = tx + pcap; float* ts = ty + pcap;
    int* cj = reinterpret_cast<int*>(ts + pcap);             // the entry's job
    const BlendQuery q = make_query(x, y, s, filter_sigmas);
    const int n = active ? c.sh_counts[li] : 0;
    const int nch = (n + kWave - 1) >> 6;
    int status = !active ? -1 : n <= 0 ? 0 : nch <= kListBboxChunks ? 1 : -1;   // 1: scanned here
    unsigned mask = 0u;                                      // the chunks of the list whose box meets the window (cifcaf.cpp:54-57)
    if (c.bbox) {
        for (int ch0 = 0; __ballot(status == 1 && ch0 < nch) != 0ull; ch0 += 4) {   // four boxes per LDS round trip
            float4 bb[4];
#pragma unroll
            for (int r = 0; r < 4; r++) bb[r] = c.bbox[(status == 1 ? li : 0) * kListBboxChunks + ((ch0 + r) & (kListBboxChunks - 1))];
#pragma unroll
            for (int r = 0; r < 4; r++)
                if (status == 1 && ch0 + r < nch && bb[r].x <= q.fxhi && bb[r].y >= q.fxlo && bb[r].z <= q.fyhi && bb[r].w >= q.fylo)
                    mask |= 1u << (ch0 + r);
        }
    } else if (status == 1) {
        mask = (1u << nch) - 1u;
    }
    PH(22);
    // where the job's items go: exclusive prefix of the chunk counts (<= 16 each: five ballots)
    const int cnt = __popc(mask);
    int base = 0, total = 0;
#pragma unroll
    for (int bit = 0; bit < 5; bit++) {
        const unsigned long long m = __ballot((cnt >> bit) & 1);
        base += prefix_count(m) << bit; total += __popcll(m) << bit;
    }
    int W = total;
    if (total > kSpecItems) {                                // (the jobs that fit are a prefix of the jobs: the offsets only grow)
        const bool fits = base + cnt <= kSpecItems;
        const unsigned long long fm = __ballot(fits && cnt > 0);
        W = fm ? rlane_i(base + cnt, 63 - __builtin_clzll(fm)) : 0;
        if (!fits && cnt > 0) { status = -1; mask = 0u; }
    }
    {
        unsigned m = mask; int k = 0;
        while (__ballot(m != 0u) != 0ull)
            if (m) { const int ch = __builtin_ctz(m); m &= m - 1u; item[base + k] = lane | (ch << 8) | (k == 0 ? 1 << 16 : 0); k++; }
    }
    seg[lane] = 0; seg[kWave + lane] = 0;
    wave_sync();
    PH(23);
    // ---- the items, kSpecGroup at a time: all six columns of a group's entries back to back, then window test + ordered
    //      compaction of what passes (the target columns travel along: no second round trip for the winners)
    const gfloat* g = (const gfloat*)c.lists;
    const unsigned cap = (unsigned)c.list_cap;
    const unsigned loff = (unsigned)li * 7u * cap;
    int np = 0;                                              // passing entries so far (uniform)
    for (int w0 = 0; w0 < W; w0 += kSpecGroup) {
        int d[kSpecGroup];
#pragma unroll
        for (int r = 0; r < kSpecGroup; r++) d[r] = item[w0 + r < W ? w0 + r : W - 1];
#pragma unroll
        for (int r = 0; r < kSpecGroup; r++) asm volatile("" : "+v"(d[r]) :: "memory");
        float x1[kSpecGroup], y1[kSpecGroup], cc[kSpecGroup], x2[kSpecGroup], y2[kSpecGroup], s2[kSpecGroup];
#pragma unroll
        for (int r = 0; r < kSpecGroup; r++) {
            x1[r] = y1[r] = cc[r] = x2[r] = y2[r] = s2[r] = 0.f;
            if (w0 + r < W) {                                // (uniform)
                const int dd = __builtin_amdgcn_readfirstlane(d[r]);
                const int j = dd & 0xff, ch = (dd >> 8) & 0xff;
                const unsigned i = (unsigned)(ch * kWave + lane);
                const unsigned o = (unsigned)rlane_i((int)loff, j) + (i < (unsigned)rlane_i(n, j) ? i : 0u);
                cc[r] = g[o]; x1[r] = g[o + cap]; y1[r] = g[o + 2u * cap];
                x2[r] = g[o + 3u * cap]; y2[r] = g[o + 4u * cap]; s2[r] = g[o + 6u * cap];
            }
        }
#pragma unroll
        for (int r = 0; r < kSpecGroup; r++)
            asm volatile("" : "+v"(x1[r]), "+v"(y1[r]), "+v"(cc[r]), "+v"(x2[r]), "+v"(y2[r]), "+v"(s2[r]) :: "memory");
#pragma unroll
        for (int r = 0; r < kSpecGroup; r++) {
            if (w0 + r >= W) break;
            const int dd = __builtin_amdgcn_readfirstlane(d[r]);
            const int j = dd & 0xff, ch = (dd >> 8) & 0xff;
            const int i = ch * kWave + lane;
            const bool pass = i < rlane_i(n, j) && x1[r] >= rlane_f(q.fxlo, j) && x1[r] <= rlane_f(q.fxhi, j) &&
                              y1[r] >= rlane_f(q.fylo, j) && y1[r] <= rlane_f(q.fyhi, j);
            const unsigned long long m = __ballot(pass);
            const int pc = __popcll(m);
            if (lane == 0) { if (dd >> 16) seg[j] = np; seg[kWave + j] = np + pc; }
            const int slot = np + prefix_count(m);
            if (pass && slot < pcap) {
                cx[slot] = x1[r]; cy[slot] = y1[r]; cv[slot] = cc[r]; tx[slot] = x2[r]; ty[slot] = y2[r]; ts[slot] = s2[r]; cj[slot] = j;
            }
            np += pc;
        }
    }
    wave_sync();
    PH(24);
    // ---- scores of the passing entries (:60-63), lane = entry; the entry's query comes from its job's lane
    const int tot = np < pcap ? np : pcap;
    for (int e0 = 0; e0 < tot; e0 += kWave) {
        const int e = e0 + lane;
        const bool have = e < tot;
        const int j = have ? cj[e] : 0;
        BlendQuery qq;
        qq.x = bperm_d(q.x, j); qq.y = bperm_d(q.y, j); qq.sigma2 = bperm_f(q.sigma2, j);
        if (have) cv[e] = score_of(qq, cx[e], cy[e], cv[e]);
    }
    wave_sync();
    PH(25);
    // ---- the top two of every job: the reference's loop (:65-73) over the job's entries in list order, lane = job
    //      (four entries per LDS round trip)
    const int lo = seg[lane], hi = seg[kWave + lane];
    if (status == 1 && hi > pcap) status = -1;               // more passing entries than the area holds: the single scans
    float s1 = 0.0f, s2 = 0.0f; int e1 = 0, e2 = 0;
    for (int e0 = lo; __ballot(status == 1 && e0 < hi) != 0ull; e0 += 4) {
        float sc[4];
#pragma unroll
        for (int r = 0; r < 4; r++) sc[r] = cv[status == 1 && e0 + r < hi ? e0 + r : 0];
#pragma unroll
        for (int r = 0; r < 4; r++) asm volatile("" : "+v"(sc[r]) :: "memory");
#pragma unroll
        for (int r = 0; r < 4; r++)
            if (status == 1 && e0 + r < hi) {
                if (sc[r] >= s1) { s2 = s1; e2 = e1; s1 = sc[r]; e1 = e0 + r; }
                else if (sc[r] > s2) { s2 = sc[r]; e2 = e0 + r; }
            }
    }
    PH(26);
    SpecOut o; o.ok = status; o.v = 0.0; o.x = o.y = o.s = 0.f;
    if (status == 1) {
        if (s1 == 0.0f) o.ok = 0;                            // :76
        else {
            const float e1x = tx[e1], e1y = ty[e1], e1s = ts[e1];
            const float e2x = tx[e2], e2y = ty[e2], e2s = ts[e2];     // (s2 == 0: e2 = 0, a valid slot nobody looks at, :84)
            const BlendResult r = blend_finish(s1, s2, true, false, e1x, e1y, e1s, e2x, e2y, e2s);
            o.v = r.v; o.x = r.x; o.y = r.y; o.s = r.s;
        }
    }
    wave_sync();                                             // the area is free for the next batch
    PH(27);
    return o;
}

// A joint was assigned (cifcaf.cpp:310): during the seed pipeline the grower publishes the occupancy box the
// joint WILL occupy if the pose is accepted, so that the coordinator can stop handing out -- and growing --
// seeds this pose is going to cover (defined behind the occupancy helpers).
// task slots of the growers (states, see the kernel)
constexpr int kTaskIdle = 0, kTaskAssigned = 1, kTaskDone = 2, kTaskAccepted = 3;
// 72 bytes = 18 banks between the slots: the coordinator's snapshot and publish_joint read one field of ALL slots with lane =
// grower; at 64 bytes twelve lanes fell on two banks (6-way conflicts on every such read, -DOPA_TASK_SLOT_PAD=0 brings it back)
#ifndef OPA_TASK_SLOT_PAD
#define OPA_TASK_SLOT_PAD 2
#endif
struct __attribute__((aligned(8))) TaskSlot { int state, seed, cancel, epoch, npub, pk, f, pad0; double score; int t_emit, t_done, pad1, coll;
#if OPA_TASK_SLOT_PAD
    int pad_bank[OPA_TASK_SLOT_PAD];
#endif
};
template <int WR> __device__ __forceinline__ void publish_joint(ImageCtx& c, const DevParams& p, int k, float x, float y, float s);
template <int WR> __device__ __forceinline__ void pool_catch_up(ImageCtx& c, int e);
template <int WR> __device__ __forceinline__ void spec_phase(ImageCtx& c, const DevParams& p, bool reverse_match_, double filter_sigmas);
// cancel flag and refill epoch of this grower's task slot, one LDS read
template <int WR>
__device__ __forceinline__ bool poll_task(ImageCtx& c) {
    if (!c.cancel) return false;
    const unsigned long long ce = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(c.cancel), __ATOMIC_RELAXED,
                                                    __HIP_MEMORY_SCOPE_WORKGROUP);
    // The commit waits for the growth of the HEAD seed and for nothing else, and twelve waves share the issue slots of
    // one compute unit: the grower that holds the head runs at raised priority, speculative growths take what is left.
    const int want = c.head_g && flag_peek(c.head_g) == c.wave ? 1 : 0;
    if (want != c.prio) {
        if (want) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(0);
        c.prio = want;
    }
    if ((int)ce) return true;
    if (c.epoch && (int)(ce >> 32) != c.my_epoch) pool_catch_up<WR>(c, (int)(ce >> 32));
    return false;
}

// -------------------------------------------------------------- frontier heap
// Exact behaviour of std::priority_queue<FrontierEntry, vector, FrontierCompare>
// (cifcaf.hpp:93, cifcaf.cpp:27-29): comp(a,b) = a.max_score < b.max_score.
// max_score is a non-negative float, so its bit pattern orders like the value.
__device__ __forceinline__ bool heap_less(unsigned long long a, unsigned long long b) {
    return (unsigned)(a >> 32) < (unsigned)(b >> 32);
}

__device__ __forceinline__ void heap_sift_up(ImageCtx& c, int hole, int top, unsigned long long value) {
    int parent = (hole - 1) / 2;
    while (hole > top) {
        const unsigned long long pv = c.heap[parent];
        if (!heap_less(pv, value)) break;
        c.heap[hole] = pv;
        hole = parent;
        parent = (hole - 1) / 2;
    }
    c.heap[hole] = value;
}

__device__ __forceinline__ void heap_push(ImageCtx& c, float score, int entry) {
    c.heap_n++;
    heap_sift_up(c, c.heap_n - 1, 0, ((unsigned long long)__float_as_uint(score) << 32) | (unsigned)entry);
}

__device__ __forceinline__ int heap_pop(ImageCtx& c) {          // returns the top entry id
    const int top = (int)(c.heap[0] & 0xffffffffull);
    const int len = c.heap_n - 1;               // heap length after removing the back
    if (len > 0) {
        const unsigned long long value = c.heap[len];
        int hole = 0, child = 0;
        while (child < (len - 1) / 2) {
            child = 2 * (child + 1);
            unsigned long long cv = c.heap[child];
            const unsigned long long lv = c.heap[child - 1];
            if (heap_less(cv, lv)) { child--; cv = lv; }
            c.heap[hole] = cv;
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

// LDS variant of the growth state (skeletons past 64 joints / 64 directed bones).  The frontier entry of a bone lives
// in the BONE's slot: an entry is popped before it is pushed again (cifcaf.cpp:298-303), in_frontier admits a
// directed bone once (:329), and of the two directions of a bone only one can ever be pushed during a growth -- (a -> b)
// needs a filled and b empty, (b -> a) the opposite, and joints never empty again -- so A entries, heap nodes and
// in_frontier flags suffice (round 2 appended entries to a pool of 4A and searched the adjacency for the slot at
// every evaluation).  A heap node carries the directed slot (start, end, list), the entry is found through its bone.

// cifcaf.cpp:349-411 for the directed bone `info` (its first slot: the bone _connection_value's scan finds, :361-374) leaving a
// joint with the values (sv, sx, sy, ss) -- the growing wave's own joint, or (scan helpers) a joint of another wave's growth
template <bool LONG>
__device__ __forceinline__ bool connection_value_at(ImageCtx& c, const DevParams& p, int info, double sv, double sx, double sy, double ss,
                                                    bool reverse_match_, double filter_sigmas,
                                                    double* nv, float* nx, float* ny, float* ns) {
    const int start = info & 0xff, bone = (info >> 16) & 0xff, fwd = (info >> 24) & 1;
    const ListView caf_f = list_view(c, bone, fwd ? 0 : 1);
    const BlendResult nj = blend<LONG>(c, caf_f, sx, sy, ss, filter_sigmas);
    if (!nj.ok) return false;
    *nx = nj.x; *ny = nj.y; *ns = nj.s;
    *nv = sqrt(nj.v * sv);                                                      // :386
    if (*nv < p.keypoint_threshold || *nv < sv * p.keypoint_threshold_rel) return false;   // :387-390
    if (p.reverse_match && reverse_match_ && start < c.F) {                     // :397
        const ListView caf_b = list_view(c, bone, fwd ? 1 : 0);
        const BlendResult rj = blend<LONG>(c, caf_b, (double)*nx, (double)*ny, (double)*ns, filter_sigmas);
        if (!rj.ok) return false;
        if (fabs(sx - (double)rj.x) + fabs(sy - (double)rj.y) > ss) return false;   // :404
    }
    return true;
}
template <bool LONG>
__device__ __forceinline__ bool connection_value(ImageCtx& c, const DevParams& p, int info,
                                 bool reverse_match_, double filter_sigmas,
                                 double* nv, float* nx, float* ny, float* ns) {
    const int start = info & 0xff;
    return connection_value_at<LONG>(c, p, info, c.jv[start], (double)c.jx[start], (double)c.jy[start], (double)c.js[start],
                                     reverse_match_, filter_sigmas, nv, nx, ny, ns);
}

// ---- scan helpers, the growing wave's side
// "joint b, reached through slot u (kSrcStart: filled from the start), has its values": the bones leaving it can be evaluated
__device__ __forceinline__ void help_post(ImageCtx& c, int b, int u) {
    if (lane_id() == 0) flag_store(&c.hq[c.hq_n & (kHelpRing - 1)], 1 | (b << 1) | (u << 10) | (c.h_epoch << 19));
    c.hq_n++;
}
// a growth begins: new epoch (words of the last growth mean nothing any more), every bone free, the block open for helpers
__device__ __forceinline__ void help_begin(ImageCtx& c) {
    const int lane = lane_id();
    c.h_epoch = (c.h_epoch + 1) & 0x3ff;
    for (int a = lane; a < c.A; a += kWave) c.rq[a] = rq_word(kRqFree, 0, 0, c.h_epoch);
    if (lane < kHelpRing) c.hq[lane] = 0;
    unsigned long long filled0 = 0ull;
    for (int k0 = 0; k0 < c.K; k0 += kWave) {
        const int k = k0 + lane;
        const bool f = k < c.K && c.jv[k < c.K ? k : 0] != 0.0;
        if (k < c.K) c.jsrc[k] = f ? kSrcStart : -1;
        if (k0 == 0) filled0 = __ballot(f);
    }
    if (lane == 0) { c.hctl[0] = c.h_epoch; c.hctl[2] = (int)(unsigned)filled0; c.hctl[3] = (int)(unsigned)(filled0 >> 32); }
    c.hq_n = 0;
    wave_sync();
    if (lane == 0) flag_store(&c.hctl[1], 0);
    for (int k0 = 0; k0 < c.K; k0 += kWave) {
        unsigned long long filled = __ballot(k0 + lane < c.K && c.jsrc[k0 + lane < c.K ? k0 + lane : 0] == kSrcStart);
        while (filled) { const int j = k0 + __builtin_ctzll(filled); filled &= filled - 1; help_post(c, j, kSrcStart); }
    }
}
// the growth is over (finished or stopped): no new claims, and nobody writes into this block once this returns
__device__ __forceinline__ void help_end(ImageCtx& c) {
    const int lane = lane_id();
    if (lane == 0) flag_store(&c.hctl[1], 1);
    for (int spin = 0; spin < (1 << 20); spin++) {
        bool taken = false;
        for (int a = lane; a < c.A; a += kWave) taken |= (flag_load(&c.rq[a]) & 7) == kRqTaken;
        if (__ballot(taken) == 0ull) break;
        __builtin_amdgcn_s_sleep(1);
    }
}
// The search pops bone `bn` (slot `slot`) for the first time; its start joint was assigned through slot `src`.
// 1: a helper left the connection value in the entry array, 2: a helper found that the connection does not hold,
// 0: the wave evaluates it itself (the word is claimed: no helper will write the entry)
__device__ __forceinline__ int help_claim(ImageCtx& c, int bn, int slot, int src) {
    for (int spin = 0; spin < (1 << 20); spin++) {
        const int w = __builtin_amdgcn_readfirstlane(flag_load(&c.rq[bn]));
        const int st = w & 7;
        if (st == kRqTaken) { __builtin_amdgcn_s_sleep(1); continue; }        // a helper is at it (3-4 us at most)
        if ((st == kRqOk || st == kRqRej) && w == rq_word(st, slot, src, c.h_epoch)) return st == kRqOk ? 1 : 2;
        int old = w;
        if (lane_id() == 0) old = atomicCAS(&c.rq[bn], w, rq_word(kRqMaster, slot, src & 0x1ff, c.h_epoch));
        old = __builtin_amdgcn_readfirstlane(old);
        if (old == w) return 0;                                              // (else a helper was faster: look again)
    }
    return 0;
}

// cifcaf.cpp:316-346: the bones leaving `start` whose other end is still empty and that are not in the frontier yet
// are found by all lanes at once (lane = adjacency slot), then pushed in bone order like the reference's loop.  A
// second bone between the same two joints never gets in: it shares the pair's in_frontier entry with the first.
__device__ __forceinline__ void frontier_add_from(ImageCtx& c, int start) {
    const float max_score = (float)sqrt(c.jv[start]);
    const int t0 = c.adj_off[start], t1 = c.adj_off[start + 1];
    const int lane = lane_id();
    for (int base = t0; base < t1; base += kWave) {
        const int t = base + lane;
        bool cand = false;
        int bone = 0;
        if (t < t1) {
            const int info = c.slot_info[t];
            const int other = (info >> 8) & 0xff;
            bone = (info >> 16) & 0xff;
            cand = c.adj_first[t] == t && !(c.jv[other] > 0.0) && !c.in_frontier[bone];
        }
        unsigned long long m = __ballot(cand);
        while (m) {
            const int l = __builtin_ctzll(m);
            m &= m - 1;
            const int bn = __builtin_amdgcn_readlane(bone, l);
            heap_push(c, max_score, base + l);
            c.in_frontier[bn] = 1;                                       // (bit 1: its entry has been computed)
        }
    }
}

__device__ __forceinline__ void frontier_start(ImageCtx& c) {
    const int lane = lane_id();
    for (int t = lane; t < c.A; t += kWave) c.in_frontier[t] = 0;
    c.heap_n = 0;
    wave_sync();
    for (int j0 = 0; j0 < c.K; j0 += kWave) {
        unsigned long long filled = __ballot(j0 + lane < c.K && c.jv[j0 + lane < c.K ? j0 + lane : 0] != 0.0);
        while (filled) {
            const int j = j0 + __builtin_ctzll(filled);
            filled &= filled - 1;
            frontier_add_from(c, j);
        }
    }
}

// cifcaf.cpp:265-313 -- one wavefront, no workgroup barriers
template <bool LONG>
__device__ __forceinline__ void grow(ImageCtx& c, const DevParams& p, bool reverse_match_, double filter_sigmas) {
    if (kWalk && c.sp_j) {                                               // the bones of the pose, level by level, in batches (spec_phase)
        spec_phase<kPoolSlotsLds>(c, p, reverse_match_, filter_sigmas);
        if (c.aborted) return;
        for (int k = lane_id(); k < c.K; k += kWave) c.sp_match[k] = (c.sp_j[k] & kSpKind) == kSpActual ? 1 : 0;
        for (int a = lane_id(); a < c.A; a += kWave) c.sp_fromc[a] = 0;
        wave_sync();
    }
    const bool help = kHelp && c.hq != nullptr;
    if (help) help_begin(c);
    frontier_start(c);
    while (c.heap_n > 0) {
        if (poll_task<kPoolSlotsLds>(c)) { c.aborted = 1; break; }                      // the seed died while its pose grew
        const int slot = heap_pop(c);
        const int info = c.slot_info[slot];
        const int start = info & 0xff, end = (info >> 8) & 0xff, bn = (info >> 16) & 0xff;
        if (c.jv[end] > 0.0) { PH(0); continue; }                        // :284
        double v = 0.0; float x = 0.f, y = 0.f, s = 0.f;
        const bool computed = (c.in_frontier[bn] & 2) != 0;
        if (computed) { v = c.e_v[bn]; x = c.e_x[bn]; y = c.e_y[bn]; s = c.e_s[bn]; }
        PH(0);
        if (!computed) {                                                 // :287: not computed yet
            // a helper may have evaluated the connection already (from the very joint the search assigned: then it is
            // what _connection_value returns); the level walk's memo likewise (walk builds)
            int memo = 0;
            const int got = help ? help_claim(c, bn, slot, c.jsrc[start]) : 0;
            if (got == 2) { PH(7); continue; }                           // :290-296
            if (kWalk && c.sp_j && c.sp_match[start]) {
                const int w = c.sp_b[bn];
                if (((w & 0xff) == kSbOk || (w & 0xff) == kSbRejected) && (w >> 8) == slot) memo = w & 0xff;
            }
            if (got == 1) {
                v = c.e_v[bn]; x = c.e_x[bn]; y = c.e_y[bn]; s = c.e_s[bn];
            } else if (memo) {
                if (kWalk) c.n_hit++;
                if (memo == kSbRejected) { PH(7); continue; }            // :290-296
                v = c.sp_v[bn]; x = c.sp_x[bn]; y = c.sp_y[bn]; s = c.sp_s[bn];
                c.sp_fromc[bn] = 1;
            } else {
                if (kWalk) c.n_miss++;
                if (!connection_value<LONG>(c, p, info, reverse_match_, filter_sigmas, &v, &x, &y, &s)) {
                    PH(7);
                    continue;                                            // :290-296 (block_joints is a no-op)
                }
                if (kWalk && c.sp_j) c.sp_fromc[bn] = 0;
            }
            PH(7);
            if (!p.greedy) {                                             // :298-303
                if (got != 1) { c.e_v[bn] = v; c.e_x[bn] = x; c.e_y[bn] = y; c.e_s[bn] = s; }
                c.in_frontier[bn] = 3;
                if (help) { wave_sync(); help_post(c, end, slot); }      // the joint this connection leads to: its bones can be evaluated
                heap_push(c, (float)v, slot);
                PH(8);
                continue;
            }
        }
        c.jv[end] = v; c.jx[end] = x; c.jy[end] = y; c.js[end] = s;     // :310
        if (help) c.jsrc[end] = slot;
        PH(9);
        // assigned exactly the candidate the level walk gave this joint (same bone, entry from the memo): its box is
        // published already, and the memo of its bones holds
        bool matched = false;
        if (kWalk && c.sp_j && c.sp_fromc[bn]) { const int w = c.sp_j[end]; matched = (w & kSpKind) == kSpCand && (w >> 8) == slot; }
        if (kWalk && c.sp_j) c.sp_match[end] = matched ? 1 : 0;
        if (!matched) publish_joint<kPoolSlotsLds>(c, p, end, x, y, s);
        PH(10);
        frontier_add_from(c, end);
        PH(11);
    }
    if (help) help_end(c);
}

// cifcaf.cpp:429-449
__device__ __forceinline__ void flood_fill(ImageCtx& c) {
    frontier_start(c);
    while (c.heap_n > 0) {
        const int slot = heap_pop(c);
        const int info = c.slot_info[slot];
        const int start = info & 0xff, end = (info >> 8) & 0xff;
        if (c.jv[end] > 0.0) continue;
        c.jv[end] = 0.00001; c.jx[end] = c.jx[start]; c.jy[end] = c.jy[start]; c.js[end] = c.js[start];
        frontier_add_from(c, end);
    }
}

// ------------------------------------------------ register-resident growth state
// For skeletons with K <= 64 joints and 2A <= 64 directed bones (COCO: 17 / 38) the pose, the
// frontier entries (one per directed bone: an entry is popped before it is pushed again, and
// in_frontier admits a bone once) and the binary heap live in VGPR lanes instead of LDS:
//     lane k: joint k      lane t: entry + static data of directed bone t      lane i: heap node i
// Everything the search does with them is wave-uniform, so it is v_readlane / v_writelane and scalar
// control flow -- a few cycles per access where a dependent LDS round trip costs ~100.  The
// algorithm (cifcaf.cpp:265-346, 429-449 and the std::priority_queue heap) is the same as in the
// LDS variant below, statement by statement.
struct RegSkeleton { int slot_info, slot_first, off, off1; };   // built once per kernel
struct RegState {
    int jv_lo, jv_hi; float jx, jy, js;          // lane k: joint k (v is a double)
    int ev_lo, ev_hi; float ex, ey, es;          // lane t: entry of directed bone t (v == 0: not computed yet)
    int list_n;                                  // lane 2*bone+dir: length of that CAF list
    int h_score, h_slot;                         // lane i: heap node i (float bits of max_score, bone slot)
    int heap_n;                                  // uniform
    unsigned long long in_frontier;              // uniform, one bit per directed bone
};

__device__ __forceinline__ int rlane(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ float rlanef(float v, int l) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}
// (this clang has no writelane builtin: one v_cmp against the lane id + a v_cndmask per register)
__device__ __forceinline__ void wlane(int& v, int val, int l) { v = lane_id() == l ? val : v; }
__device__ __forceinline__ void wlanef(float& v, float val, int l) { v = lane_id() == l ? val : v; }
__device__ __forceinline__ double uniform_f64(double v) {
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)),
                            __builtin_amdgcn_readfirstlane(__double2loint(v)));
}
__device__ __forceinline__ float uniform_f32(float v) {
    return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
}
__device__ __forceinline__ double reg_jv(const RegState& R, int j) {
    return __hiloint2double(rlane(R.jv_hi, j), rlane(R.jv_lo, j));
}
__device__ __forceinline__ void reg_set_joint(RegState& R, int j, double v, float x, float y, float s) {
    wlane(R.jv_lo, __double2loint(v), j); wlane(R.jv_hi, __double2hiint(v), j);
    wlanef(R.jx, x, j); wlanef(R.jy, y, j); wlanef(R.js, s, j);
}

__device__ __forceinline__ void reg_heap_sift_up(RegState& R, int hole, unsigned score, int slot) {
    int parent = (hole - 1) / 2;
    while (hole > 0) {
        const unsigned ps = (unsigned)rlane(R.h_score, parent);
        if (!(ps < score)) break;
        const int pslot = rlane(R.h_slot, parent);
        wlane(R.h_score, (int)ps, hole); wlane(R.h_slot, pslot, hole);
        hole = parent;
        parent = (hole - 1) / 2;
    }
    wlane(R.h_score, (int)score, hole); wlane(R.h_slot, slot, hole);
}
__device__ __forceinline__ void reg_heap_push(RegState& R, float score, int slot) {
    R.heap_n++;
    reg_heap_sift_up(R, R.heap_n - 1, (unsigned)__builtin_amdgcn_readfirstlane(__float_as_int(score)), slot);
}
__device__ __forceinline__ int reg_heap_pop(RegState& R) {      // returns the top node's bone slot
    const int top = rlane(R.h_slot, 0);
    const int len = R.heap_n - 1;
    if (len > 0) {
        const unsigned vscore = (unsigned)rlane(R.h_score, len);
        const int vslot = rlane(R.h_slot, len);
        int hole = 0, child = 0;
        while (child < (len - 1) / 2) {
            child = 2 * (child + 1);
            unsigned cs = (unsigned)rlane(R.h_score, child);
            const unsigned ls = (unsigned)rlane(R.h_score, child - 1);
            if (cs < ls) { child--; cs = ls; }
            const int cslot = rlane(R.h_slot, child);
            wlane(R.h_score, (int)cs, hole); wlane(R.h_slot, cslot, hole);
            hole = child;
        }
        if ((len & 1) == 0 && child == (len - 2) / 2) {
            child = 2 * (child + 1);
            const int cs = rlane(R.h_score, child - 1), cslot = rlane(R.h_slot, child - 1);
            wlane(R.h_score, cs, hole); wlane(R.h_slot, cslot, hole);
            hole = child - 1;
        }
        reg_heap_sift_up(R, hole, vscore, vslot);
    }
    R.heap_n = len;
    return top;
}

// cifcaf.cpp:316-346
__device__ __forceinline__ void reg_frontier_add_from(RegState& R, const RegSkeleton& sk, int start) {
    const float max_score = (float)sqrt(reg_jv(R, start));
    // the bones leaving `start` whose other end is still empty, found by all lanes at once (lane t = bone t, lane j =
    // joint j), then pushed in bone order like the reference's loop
    const int lane = lane_id();
    const unsigned long long filled = __ballot(__hiloint2double(R.jv_hi, R.jv_lo) > 0.0);
    const int other = (sk.slot_info >> 8) & 0xff;
    unsigned long long cand = __ballot(lane >= rlane(sk.off, start) && lane < rlane(sk.off1, start) && !((filled >> other) & 1ull));
    while (cand) {
        const int t = __builtin_ctzll(cand);
        cand &= cand - 1;
        const int first = rlane(sk.slot_first, t);
        if ((R.in_frontier >> first) & 1ull) continue;
        wlane(R.ev_lo, 0, first); wlane(R.ev_hi, 0, first);
        reg_heap_push(R, max_score, first);
        R.in_frontier |= 1ull << first;
    }
}

__device__ __forceinline__ void reg_frontier_start(RegState& R, const RegSkeleton& sk, int K) {
    R.heap_n = 0; R.in_frontier = 0ull;
    unsigned long long filled = __ballot(lane_id() < K && __hiloint2double(R.jv_hi, R.jv_lo) != 0.0);
    while (filled) {
        const int j = __builtin_ctzll(filled);
        filled &= filled - 1;
        reg_frontier_add_from(R, sk, j);
    }
}

// cifcaf.cpp:349-411 for the directed bone `slot` leaving joint `start`
template <bool LONG>
__device__ __forceinline__ bool reg_connection_value(ImageCtx& c, const DevParams& p, const RegState& R, int start,
                                                     int info, bool reverse_match_, double filter_sigmas,
                                                     double* nv, float* nx, float* ny, float* ns) {
    const int bone = (info >> 16) & 0xff, fwd = (info >> 24) & 1;
    // (the view of the reverse list is built after the forward scan: nothing of it has to stay live across that scan)
    auto view = [&](int dir) {
        ListView v;
        v.cap = c.list_cap;
        v.base = c.lists + ((size_t)bone * 2 + dir) * 7 * c.list_cap;
        v.n = rlane(R.list_n, bone * 2 + dir);
        v.bbox = c.bbox ? c.bbox + (bone * 2 + dir) * kListBboxChunks : nullptr;
        v.gbbox = c.gbbox ? c.gbbox + (size_t)(bone * 2 + dir) * c.nb : nullptr; v.nb = c.nb;
        return v;
    };
    const ListView caf_f = view(fwd ? 0 : 1);
    const double sv = reg_jv(R, start);
    const double sx = (double)rlanef(R.jx, start), sy = (double)rlanef(R.jy, start), ss = (double)rlanef(R.js, start);
    const BlendResult nj = blend<LONG>(c, caf_f, sx, sy, ss, filter_sigmas);
    if (!nj.ok) return false;
    *nx = uniform_f32(nj.x); *ny = uniform_f32(nj.y); *ns = uniform_f32(nj.s);
    *nv = uniform_f64(sqrt(nj.v * sv));                                         // :386
    if (*nv < p.keypoint_threshold || *nv < sv * p.keypoint_threshold_rel) return false;   // :387-390
    if (p.reverse_match && reverse_match_ && start < c.F) {                     // :397
        const ListView caf_b = view(fwd ? 1 : 0);
        const BlendResult rj = blend<LONG>(c, caf_b, (double)*nx, (double)*ny, (double)*ns, filter_sigmas);
        if (!rj.ok) return false;
        if (fabs(sx - (double)rj.x) + fabs(sy - (double)rj.y) > ss) return false;   // :404
    }
    return true;
}

// pose: LDS -> lanes, and back
__device__ __forceinline__ void reg_load_pose(const ImageCtx& c, RegState& R) {
    const int lane = lane_id();
    double v = 0.0; R.jx = 0.f; R.jy = 0.f; R.js = 0.f;
    if (lane < c.K) { v = c.jv[lane]; R.jx = c.jx[lane]; R.jy = c.jy[lane]; R.js = c.js[lane]; }
    R.jv_lo = __double2loint(v); R.jv_hi = __double2hiint(v);
    R.list_n = lane < 2 * c.A ? c.sh_counts[lane] : 0;
    R.ev_lo = R.ev_hi = 0; R.ex = R.ey = R.es = 0.f; R.h_score = 0; R.h_slot = 0;
}
__device__ __forceinline__ void reg_store_pose(ImageCtx& c, const RegState& R) {
    const int lane = lane_id();
    if (lane < c.K) {
        c.jv[lane] = __hiloint2double(R.jv_hi, R.jv_lo); c.jx[lane] = R.jx; c.jy[lane] = R.jy; c.js[lane] = R.js;
    }
    wave_sync();
}

// cifcaf.cpp:265-313 on the pose in this wave's LDS block
template <bool LONG>
__device__ __forceinline__ void grow_reg(ImageCtx& c, const DevParams& p, const RegSkeleton& sk, bool reverse_match_,
                                         double filter_sigmas, bool then_flood_fill) {
    // joints assigned exactly their candidate's values / directed bones whose entry came from the memo (one bit each)
    unsigned long long sp_match = 0ull, sp_fromc = 0ull;
    if (kWalk && c.sp_j) {                                                   // the bones of the pose, level by level, in batches (spec_phase)
        spec_phase<kPoolSlots>(c, p, reverse_match_, filter_sigmas);
        if (c.aborted) return;
        sp_match = __ballot(lane_id() < c.K && (c.sp_j[lane_id() < c.K ? lane_id() : 0] & kSpKind) == kSpActual);
    }
    const bool help = kHelp && c.hq != nullptr;
    if (help) help_begin(c);
    unsigned long long amask = help ? ((unsigned long long)(unsigned)c.hctl[3] << 32) | (unsigned)c.hctl[2] : 0ull;   // joints assigned so far (for the helpers)
    RegState R;
    reg_load_pose(c, R);
    reg_frontier_start(R, sk, c.K);
    while (R.heap_n > 0) {
        if (poll_task<kPoolSlots>(c)) { c.aborted = 1; if (help) help_end(c); return; }   // the seed died while its pose grew
        const int slot = reg_heap_pop(R);
        const int info = rlane(sk.slot_info, slot);
        const int start = info & 0xff, end = (info >> 8) & 0xff;
        if (reg_jv(R, end) > 0.0) { PH(0); continue; }                       // :284
        double v = __hiloint2double(rlane(R.ev_hi, slot), rlane(R.ev_lo, slot));
        float x, y, s;
        PH(0);
        if (v == 0.0) {                                                      // :287: not computed yet
            // a helper may have evaluated the connection already (from the very joint the search assigned: then it is
            // what _connection_value returns); the level walk's memo likewise (walk builds)
            int memo = 0;
            const int bn = (info >> 16) & 0xff;
            const int got = help ? help_claim(c, bn, slot, __builtin_amdgcn_readfirstlane(c.jsrc[start])) : 0;
            if (got == 2) { PH(7); continue; }                               // :290-296
            if (kWalk && c.sp_j && ((sp_match >> start) & 1ull)) {
                const int w = __builtin_amdgcn_readfirstlane(c.sp_b[bn]);
                if (((w & 0xff) == kSbOk || (w & 0xff) == kSbRejected) && (w >> 8) == slot) memo = w & 0xff;
            }
            if (got == 1) {
                v = uniform_f64(c.e_v[bn]); x = uniform_f32(c.e_x[bn]); y = uniform_f32(c.e_y[bn]); s = uniform_f32(c.e_s[bn]);
            } else if (memo) {
                if (kWalk) c.n_hit++;
                if (memo == kSbRejected) { PH(7); continue; }                // :290-296
                v = uniform_f64(c.sp_v[bn]); x = uniform_f32(c.sp_x[bn]); y = uniform_f32(c.sp_y[bn]); s = uniform_f32(c.sp_s[bn]);
                if (kWalk) sp_fromc |= 1ull << slot;
            } else {
                if (kWalk) c.n_miss++;
                if (!reg_connection_value<LONG>(c, p, R, start, info, reverse_match_, filter_sigmas, &v, &x, &y, &s)) {
                    PH(7);
                    continue;                                                // :290-296 (block_joints is a no-op)
                }
                if (kWalk) sp_fromc &= ~(1ull << slot);
            }
            PH(7);
            if (!p.greedy) {                                                 // :298-303
                wlane(R.ev_lo, __double2loint(v), slot); wlane(R.ev_hi, __double2hiint(v), slot);
                wlanef(R.ex, x, slot); wlanef(R.ey, y, slot); wlanef(R.es, s, slot);
                if (help) {                                                  // the joint this connection leads to: its bones can be evaluated
                    if (got != 1 && lane_id() == 0) { c.e_v[bn] = v; c.e_x[bn] = x; c.e_y[bn] = y; c.e_s[bn] = s; }
                    wave_sync();
                    help_post(c, end, slot);
                }
                reg_heap_push(R, (float)v, slot);
                PH(8);
                continue;
            }
        } else {
            x = rlanef(R.ex, slot); y = rlanef(R.ey, slot); s = rlanef(R.es, slot);
        }
        reg_set_joint(R, end, v, x, y, s);                                   // :310
        if (help) {
            amask |= 1ull << end;
            if (lane_id() == 0) { c.jsrc[end] = slot; c.hctl[2] = (int)(unsigned)amask; c.hctl[3] = (int)(unsigned)(amask >> 32); }
        }
        PH(9);
        // assigned exactly the candidate the level walk gave this joint (same bone, entry from the memo): its box is
        // published already, and the memo of its bones holds
        bool matched = false;
        if (kWalk && c.sp_j && ((sp_fromc >> slot) & 1ull)) {
            const int w = __builtin_amdgcn_readfirstlane(c.sp_j[end]);
            matched = (w & kSpKind) == kSpCand && (w >> 8) == slot;
        }
        if (kWalk && matched) sp_match |= 1ull << end;
        else publish_joint<kPoolSlots>(c, p, end, x, y, s);
        PH(10);
        reg_frontier_add_from(R, sk, end);
        PH(11);
    }
    if (help) help_end(c);
    if (then_flood_fill) {                                                   // cifcaf.cpp:429-449
        reg_frontier_start(R, sk, c.K);
        while (R.heap_n > 0) {
            const int slot = reg_heap_pop(R);
            const int info = rlane(sk.slot_info, slot);
            const int start = info & 0xff, end = (info >> 8) & 0xff;
            if (reg_jv(R, end) > 0.0) continue;
            reg_set_joint(R, end, 0.00001, rlanef(R.jx, start), rlanef(R.jy, start), rlanef(R.js, start));
            reg_frontier_add_from(R, sk, end);
        }
    }
    reg_store_pose(c, R);
}

template <bool REG, bool LONG = false>
__device__ __forceinline__ void grow_pose(ImageCtx& c, const DevParams& p, const RegSkeleton& sk, bool reverse_match_,
                                          double filter_sigmas, bool then_flood_fill) {
    if constexpr (REG) {
        grow_reg<LONG>(c, p, sk, reverse_match_, filter_sigmas, then_flood_fill);
    } else {
        grow<LONG>(c, p, reverse_match_, filter_sigmas);
        if (then_flood_fill && !c.aborted) flood_fill(c);
        wave_sync();
    }
}


// ---------------------------------------------------------------- occupancy
// occupancy.cpp:32-43: the cell a query (x, y) falls into
__device__ __forceinline__ void occ_xy(const ImageCtx& c, const DevParams& p, double x, double y, int* xi, int* yi) {
    if (p.occupancy_inv_reduction != 0.0) { x *= p.occupancy_inv_reduction; y *= p.occupancy_inv_reduction; }
    else if (p.occupancy_reduction != 1.0) { x /= p.occupancy_reduction; y /= p.occupancy_reduction; }
    *xi = (int)clampll(trunc_ll(x), 0, c.occ_w - 1);
    *yi = (int)clampll(trunc_ll(y), 0, c.occ_h - 1);
}

// occupancy.cpp:13-29: the half-open cell box [minx,maxx) x [miny,maxy) a joint occupies.  Used to fill the
// bitmap and to test containment analytically (commit, keypoint NMS), so the two can never disagree.
struct __attribute__((aligned(16))) OccBox { int minx, miny, maxx, maxy; };
__device__ __forceinline__ OccBox occ_box(const ImageCtx& c, const DevParams& p, double x, double y, double sigma) {
    if (p.occupancy_inv_reduction != 0.0) {           // power-of-two reduction (the reference's 2.0): exact either way
        x *= p.occupancy_inv_reduction; y *= p.occupancy_inv_reduction;
        sigma = fmax(p.occupancy_min_scale_reduced, sigma * p.occupancy_inv_reduction);
    } else if (p.occupancy_reduction != 1.0) {
        x /= p.occupancy_reduction; y /= p.occupancy_reduction;
        sigma = fmax(p.occupancy_min_scale_reduced, sigma / p.occupancy_reduction);
    }
    // clamp(trunc_toward_zero(t), lo, hi) for 0 <= lo <= hi < 2^30 without the 64-bit conversion: the double is
    // clamped to [lo - 1, hi + 1] first (a NaN lands on lo - 1; the 64-bit conversion made it 0, which clamps to lo
    // as well), truncated by v_cvt_i32_f64, and clamped again as an integer -- the same value for every input.
    auto tc = [](double t, int lo, int hi) {
        const int i = (int)fmin(fmax(t, (double)lo - 1.0), (double)hi + 1.0);
        return min(max(i, lo), hi);
    };
    OccBox b;
    b.minx = tc(x - sigma, 0, c.occ_w - 1);
    b.miny = tc(y - sigma, 0, c.occ_h - 1);
    b.maxx = tc(x + sigma, b.minx + 1, c.occ_w);
    b.maxy = tc(y + sigma, b.miny + 1, c.occ_h);
    return b;
}
__device__ __forceinline__ bool box_contains(const OccBox& b, int xi, int yi) {
    return xi >= b.minx && xi < b.maxx && yi >= b.miny && yi < b.maxy;
}

constexpr int kSeedStage = 1024;          // seeds (field, cell) the coordinator keeps staged in LDS beyond its scan position (ring)
constexpr int kDedupBits = 10;            // buckets (log2) of the coordinator's first-seed-of-a-cell table
#ifndef OPA_COMMIT_RUN
#define OPA_COMMIT_RUN 4
#endif
#ifndef OPA_REFILL_NUM
#define OPA_REFILL_NUM 2
#endif
constexpr int kRefillNum = OPA_REFILL_NUM, kRefillDen = 4;   // the pool is refilled when fewer than kRefillNum / kRefillDen of its slots are live
constexpr int kCommitRun = OPA_COMMIT_RUN;             // commits per round of the coordinator before it looks at the idle growers again
constexpr int kPoolIdxMask = 0xFFFFFF;    // seed index bits of a slot word (all ones: empty slot); field above
constexpr int kPreDedupMin = 1024;        // images with fewer seeds go through the pool refill as they are (two refill rounds)

template <int WR>
__device__ __forceinline__ void publish_joint(ImageCtx& c, const DevParams& p, int k, float x, float y, float s) {
    if (!c.pub || k >= c.F) return;
    const OccBox b = occ_box(c, p, (double)x, (double)y, (double)s);
    const int lane = lane_id();
    // every pooled seed of this field that comes later in seed order and lies in the box is shadowed: it dies
    // if this pose is accepted.  Advisory only (the commit re-tests every seed against the final boxes).
    int sif[WR], spk[WR];
#pragma unroll
    for (int r = 0; r < WR; r++) { sif[r] = c.pool_if[r * kWave + lane]; spk[r] = c.pool_pack[r * kWave + lane]; }
    // field k, a seed index in (my_idx, empty): two unsigned compares on the packed word; inside the box: one unsigned
    // compare per axis (cell - min < extent)
    const unsigned lo = ((unsigned)k << 24) | (unsigned)c.my_idx, hi = ((unsigned)k << 24) | (unsigned)kPoolIdxMask;
    const unsigned ex = (unsigned)(b.maxx - b.minx), ey = (unsigned)(b.maxy - b.miny);
    unsigned bits = 0u;
#pragma unroll
    for (int r = 0; r < WR; r++) {
        const unsigned w = (unsigned)sif[r];
        const unsigned dx = (unsigned)((spk[r] & 0xfff) - b.minx), dy = (unsigned)(((spk[r] >> 12) & 0xfff) - b.miny);
        if (w > lo && w < hi && dx < ex && dy < ey) bits |= 1u << r;
    }
    if (bits) atomicOr(&c.shadow_mine[lane], bits);
    ++c.n_pub;
    if (lane == 0) { c.jbox[k] = b; *c.pub = c.n_pub; }
    // A joint of mine inside the box an EARLIER live candidate has published for the same joint: the two growths are,
    // most likely, growing the same person, and the earlier one decides first -- tell the coordinator (it stops this
    // growth, keeps the seed pooled as predicted dead, and hands my predictions on to that candidate).  Advisory.
    // (the task slots open the workgroup's dynamic LDS; where the growers' pose blocks lie, how many there are and whether
    // the test is on: the two control words behind the head grower's, written once by the kernel)
    const int cfg = c.head_g[2];
    if ((cfg >> 28) & 1) {
        const TaskSlot* tasks = reinterpret_cast<const TaskSlot*>(opa_dyn_lds);
        const unsigned char* blocks = opa_dyn_lds + c.head_g[1];
        const int n_growers = (cfg >> 20) & 0xff, block_bytes = cfg & 0xfffff;
        int cx, cy;
        occ_xy(c, p, (double)x, (double)y, &cx, &cy);
        unsigned key = 0xFFFFFFFFu;
        if (lane >= 1 && lane <= n_growers && lane != c.wave) {
            const int st = flag_peek(&tasks[lane].state), cn = flag_peek(&tasks[lane].cancel), sd = tasks[lane].seed;
            if ((st == kTaskAssigned || st == kTaskDone) && !cn && sd < c.my_idx) {
                const OccBox ob = reinterpret_cast<const OccBox*>(blocks + (size_t)(lane - 1) * block_bytes)[k];
                // the SAME joint, not a neighbour's: in the middle of the box (two growths of one person put a joint within a cell
                // of each other; the hands of two people standing side by side lie inside each other's boxes -- stopping THOSE growths
                // made every second person of the wholebody batches wait for the one before, round 5)
                const int ex = ob.maxx - ob.minx, ey = ob.maxy - ob.miny;
                const int ddx = 2 * cx - (ob.minx + ob.maxx - 1), ddy = 2 * cy - (ob.miny + ob.maxy - 1);   // twice the offset from the centre
                const int tx = max(2, ex >> c.coll_shift), ty = max(2, ey >> c.coll_shift);
                if (ex > 0 && ddx >= -tx && ddx <= tx && ddy >= -ty && ddy <= ty) key = ((unsigned)sd << 6) | (unsigned)lane;
            }
        }
        if (__ballot(key != 0xFFFFFFFFu) != 0ull) {
            const unsigned first = ~wave_max_u32(~key);      // the earliest of them
            if (lane == 0) flag_store(const_cast<int*>(&tasks[c.wave].coll), (int)(first & 63u));
        }
    }
}

// Seeds that entered the pool after this growth published a box were not there when publish_joint tested the pool:
// the coordinator bumps an epoch at every refill, and each candidate in flight tests the newcomers against ITS
// boxes (published so far, or final) -- eleven waves in parallel instead of the coordinator alone -- and acknowledges.
template <int WR>
__device__ __forceinline__ void pool_catch_up(ImageCtx& c, int e) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");   // (the epoch was read relaxed: the pool mirror written before it)
    const int lane = lane_id();
    unsigned bits = 0u;
    // four slots at a time, straight-line: their twelve pool words are ONE LDS round trip, their four boxes a second one
    // (inside the `if`s of a per-slot loop the compiler waited for every load on its own: 24 round trips per call)
    constexpr int G = WR < 4 ? WR : 4;
#pragma unroll
    for (int r0 = 0; r0 < WR; r0 += G) {
        int ep[G], sif[G], spk[G];
#pragma unroll
        for (int r = 0; r < G; r++) {
            ep[r] = c.pool_ep[(r0 + r) * kWave + lane]; sif[r] = c.pool_if[(r0 + r) * kWave + lane];
            spk[r] = c.pool_pack[(r0 + r) * kWave + lane];
        }
#pragma unroll
        for (int r = 0; r < G; r++) asm volatile("" : "+v"(ep[r]), "+v"(sif[r]), "+v"(spk[r]) :: "memory");
        OccBox bx[G];
#pragma unroll
        for (int r = 0; r < G; r++) {
            const int f = (int)((unsigned)sif[r] >> 24);
            bx[r] = c.jbox[f < c.F ? f : 0];
        }
#pragma unroll
        for (int r = 0; r < G; r++) asm volatile("" : "+v"(bx[r].minx), "+v"(bx[r].miny), "+v"(bx[r].maxx), "+v"(bx[r].maxy) :: "memory");
#pragma unroll
        for (int r = 0; r < G; r++) {
            const int f = (int)((unsigned)sif[r] >> 24), idx = sif[r] & kPoolIdxMask;
            if (ep[r] - c.my_epoch > 0 && idx != kPoolIdxMask && idx > c.my_idx && f < c.F &&
                box_contains(bx[r], spk[r] & 0xfff, (spk[r] >> 12) & 0xfff)) bits |= 1u << (r0 + r);
        }
    }
    if (bits) atomicOr(&c.shadow_mine[lane], bits);
    c.my_epoch = e;
    wave_sync();
    if (lane == 0) flag_store(c.ack, e);
}


// [r5] Where the pose is going to be, before it is grown: a walk over the skeleton from the seed joint that reads, for every
// bone leaving a joint it has reached, ONE cell of the raw CAF field -- the cell the joint lies in -- and takes the far end
// the field regresses there (no list scan, no blend, no reverse match: a level of the walk is one memory round trip for all its
// bones, lane = directed bone), then publishes the occupancy boxes of all joints it reached in one pass over the pool.  The
// exact search that follows assigns these joints one every ~6 us; until it has, the other seeds of the same person are
// handed out and grown as duplicates -- on a crowded image two growths in three, and the people whose first seed comes
// after them in seed order start 50-140 us late.  Predicted boxes are what published boxes are: advisory --
// pooled seeds inside them are not handed out while this candidate lives, growths that run into them are stopped, the commit
// decides from the final boxes (pose_boxes rewrites every box when the growth ends).
// (A real call, once per growth, with everything it needs passed by value: inlined into the grower's loop it cost the scans
// fourteen spilled registers; handed the context by reference it would have put the whole context into scratch memory.)
struct PredictArgs {
    const float* raw; int HW, W; float stride, th;
    int E, K, F, occ_w, occ_h, my_idx;
    OccBox* jbox; const int* pool_if; const int* pool_pack; unsigned* shadow; int* n_predicted;
};
#ifndef OPA_PREDICT_INLINE
#define OPA_PREDICT_INLINE 1
#endif
#if OPA_PREDICT_INLINE
#define OPA_PREDICT_ATTR __forceinline__
#else
#define OPA_PREDICT_ATTR __attribute__((noinline))
#endif
template <int WR>
__device__ OPA_PREDICT_ATTR void predict_pose_call(PredictArgs q, const DevParams* pp, int info, int first, int seed_joint,
                                                   float sx, float sy, float ss) {
    const DevParams& p = *pp;
    const int lane = lane_id();
    const int HW = q.HW, W = q.W, H = HW / W;
    const int start = info & 0xff, other = (info >> 8) & 0xff, bone = (info >> 16) & 0xff, fwd = (info >> 24) & 1;
    const float inv = 1.0f / q.stride;
    float qx = lane == seed_joint ? sx : 0.f, qy = lane == seed_joint ? sy : 0.f, qs = lane == seed_joint ? ss : 0.f;   // lane j: joint j
    unsigned long long known = 1ull << seed_joint;
    for (int level = 0; level < q.K; level++) {
        const bool act = lane < q.E && first == lane && ((known >> start) & 1ull) && !((known >> other) & 1ull);
        if (__ballot(act) == 0ull) break;
        const float x = bperm_f(qx, start), y = bperm_f(qy, start), s0 = bperm_f(qs, start);   // (all lanes: a permute reads nothing from a masked lane)
        int cx = (int)(x * inv + 0.5f), cy = (int)(y * inv + 0.5f);
        cx = min(max(cx, 0), W - 1); cy = min(max(cy, 0), H - 1);
        const float* P = q.raw + ((size_t)bone * 8) * HW + (size_t)cy * W + cx;
        float cc = 0.f, tx = 0.f, ty = 0.f, ts = 0.f, ox = 0.f, oy = 0.f;
        if (act) {
            cc = P[1 * HW]; tx = P[(fwd ? 4 : 2) * HW]; ty = P[(fwd ? 5 : 3) * HW]; ts = P[(fwd ? 7 : 6) * HW];
            ox = P[(fwd ? 2 : 4) * HW]; oy = P[(fwd ? 3 : 5) * HW];    // where the cell says THIS end of the bone lies
        }
        // (a bone of somebody else that passes through the joint's cell starts somewhere else: the search's reverse match, :404,
        // rejects it; here the cell's own regression of the near end has to agree with the joint -- same loads, no second trip)
        const bool ok = act && cc > q.th && tx == tx && ty == ty && ts == ts &&
                        fabsf(ox * q.stride - x) + fabsf(oy * q.stride - y) <= s0;
        unsigned long long m = __ballot(ok);
        if (m == 0ull) break;
        while (m) {                                  // (two bones into one joint: the first slot's answer)
            const int l = __builtin_ctzll(m);
            const int o = rlane(other, l);
            const float nx = rlanef(tx, l) * q.stride, ny = rlanef(ty, l) * q.stride, ns = rlanef(ts, l) * q.stride;
            if (lane == o) { qx = nx; qy = ny; qs = ns; }
            known |= 1ull << o;
            m &= ~__ballot(ok && other == o);
        }
    }
    known &= ~(1ull << seed_joint);                  // (its box is out already)
    if (q.F < 64) known &= (1ull << q.F) - 1ull;     // joints without a CIF field have no seeds to keep away
    if (known == 0ull) return;
    ImageCtx dims; dims.occ_w = q.occ_w; dims.occ_h = q.occ_h;     // (occ_box reads the map's size from the context)
    if ((known >> lane) & 1ull) q.jbox[lane] = occ_box(dims, p, (double)qx, (double)qy, (double)qs);
    wave_sync();
    unsigned bits = 0u;
#pragma unroll
    for (int r = 0; r < WR; r++) {
        const int sif = q.pool_if[r * kWave + lane], spk = q.pool_pack[r * kWave + lane];
        const int f = (int)((unsigned)sif >> 24), idx = sif & kPoolIdxMask;
        if (idx != kPoolIdxMask && idx > q.my_idx && f < q.F && ((known >> f) & 1ull) &&
            box_contains(q.jbox[f], spk & 0xfff, (spk >> 12) & 0xfff)) bits |= 1u << r;
    }
    if (bits) atomicOr(&q.shadow[lane], bits);
    if (lane == 0) __hip_atomic_fetch_add(q.n_predicted, __popcll(known), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
template <int WR>
__device__ __forceinline__ void predict_pose(ImageCtx& c, const DevParams& p, const RegSkeleton& sk, int seed_joint,
                                             float sx, float sy, float ss) {
    if (!c.raw_caf || !c.pub) return;
    PredictArgs q;
    q.raw = c.raw_caf; q.HW = c.list_cap; q.W = c.raw_w; q.stride = c.raw_stride; q.th = c.predict_th;
    q.E = 2 * c.A; q.K = c.K; q.F = c.F; q.occ_w = c.occ_w; q.occ_h = c.occ_h; q.my_idx = c.my_idx;
    q.jbox = c.jbox; q.pool_if = c.pool_if; q.pool_pack = c.pool_pack; q.shadow = c.shadow_mine; q.n_predicted = c.n_predicted;
    predict_pose_call<WR>(q, &p, sk.slot_info, sk.slot_first, seed_joint, sx, sy, ss);
}

// The same walk for skeletons that do not fit the lanes of a wave (LDS-resident growth state: wholebody, dense connections,
// tracking): the reached joints live in the wave's scan area, which is free until the search starts; directed bones 64 at a time.
template <int WR>
__device__ __forceinline__ void predict_pose_lds(ImageCtx& c, const DevParams& p, int seed_joint, float sx, float sy, float ss,
                                                 int tgt_floats) {
    if (!c.raw_caf || !c.pub || 4 * c.K > tgt_floats) return;
    const int lane = lane_id(), K = c.K, E = 2 * c.A, HW = c.list_cap, W = c.raw_w, H = HW / W;
    float* px = c.tgt; float* py = px + K; float* ps = py + K; int* kn = reinterpret_cast<int*>(ps + K);   // kn: 0 unknown, 1 reached, 2 reached in this level
    for (int k = lane; k < K; k += kWave) kn[k] = 0;
    wave_sync();
    if (lane == 0) { kn[seed_joint] = 1; px[seed_joint] = sx; py[seed_joint] = sy; ps[seed_joint] = ss; }
    wave_sync();
    const float inv = 1.0f / c.raw_stride;
    int n_reached = 0;
    for (int level = 0; level < K; level++) {
        bool any = false;
        for (int t0 = 0; t0 < E; t0 += kWave) {
            const int t = t0 + lane;
            bool act = false; int start = 0, other = 0, bone = 0, fwd = 0;
            if (t < E) {
                const int info = c.slot_info[t];
                start = info & 0xff; other = (info >> 8) & 0xff; bone = (info >> 16) & 0xff; fwd = (info >> 24) & 1;
                act = c.adj_first[t] == t && kn[start] == 1 && kn[other] == 0;
            }
            if (__ballot(act) == 0ull) continue;
            float cc = 0.f, tx = 0.f, ty = 0.f, ts = 0.f, ox = 0.f, oy = 0.f, jx0 = 0.f, jy0 = 0.f, js0 = 0.f;
            if (act) {
                jx0 = px[start]; jy0 = py[start]; js0 = ps[start];
                int cx = (int)(jx0 * inv + 0.5f), cy = (int)(jy0 * inv + 0.5f);
                cx = min(max(cx, 0), W - 1); cy = min(max(cy, 0), H - 1);
                const float* P = c.raw_caf + ((size_t)bone * 8) * HW + (size_t)cy * W + cx;
                cc = P[1 * HW]; tx = P[(fwd ? 4 : 2) * HW]; ty = P[(fwd ? 5 : 3) * HW]; ts = P[(fwd ? 7 : 6) * HW];
                ox = P[(fwd ? 2 : 4) * HW]; oy = P[(fwd ? 3 : 5) * HW];    // where the cell says THIS end of the bone lies
            }
            // (the cell's own regression of the near end has to agree with the joint: see predict_pose_call)
            const bool ok = act && cc > c.predict_th && tx == tx && ty == ty && ts == ts &&
                            fabsf(ox * c.raw_stride - jx0) + fabsf(oy * c.raw_stride - jy0) <= js0;
            if (ok && atomicCAS(&kn[other], 0, 2) == 0) {     // (two bones into one joint: whichever gets there first)
                px[other] = tx * c.raw_stride; py[other] = ty * c.raw_stride; ps[other] = ts * c.raw_stride;
            }
            any |= __ballot(ok) != 0ull;
        }
        if (!any) break;
        wave_sync();
        for (int k = lane; k < K; k += kWave) if (kn[k] == 2) { kn[k] = 1; }
        wave_sync();
    }
    for (int k = lane; k < K; k += kWave) {
        const bool have = k < c.F && k != seed_joint && kn[k] == 1;
        if (have) c.jbox[k] = occ_box(c, p, (double)px[k], (double)py[k], (double)ps[k]);
        n_reached += __popcll(__ballot(have));
    }
    if (n_reached == 0) return;
    wave_sync();
    unsigned bits = 0u;
#pragma unroll
    for (int r = 0; r < WR; r++) {
        const int sif = c.pool_if[r * kWave + lane], spk = c.pool_pack[r * kWave + lane];
        const int f = (int)((unsigned)sif >> 24), idx = sif & kPoolIdxMask;
        if (idx != kPoolIdxMask && idx > c.my_idx && f < c.F && f != seed_joint && kn[f] == 1 &&
            box_contains(c.jbox[f], spk & 0xfff, (spk >> 12) & 0xfff)) bits |= 1u << r;
    }
    if (bits) atomicOr(&c.shadow_mine[lane], bits);
    if (lane == 0) __hip_atomic_fetch_add(c.n_predicted, n_reached, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    wave_sync();                                     // (the scan area is the search's from here on)
}

// ------------------------------------------------ speculative batched evaluation of a growth (spec_phase)
// The reference evaluates one bone per pop of its frontier (cifcaf.cpp:287-303), a chain of ~2 list scans per bone -- but
// _connection_value (:349-411) of a bone depends on its START JOINT alone, and a joint, once assigned, never changes.  So
// before the search itself runs, the wave walks the skeleton level by level from the joints the pose starts with: all
// bones leaving the joints of a level are evaluated in ONE batch of forward scans and ONE batch of reverse scans
// (spec_scan_batch), each result is remembered per bone, and the end joint of every connection that holds becomes a
// CANDIDATE -- the start joint of the next level's bones.  The search (grow / grow_reg) then runs the reference's heap
// loop unchanged and takes a bone's connection value from the memo instead of scanning -- but only when the start joint
// was assigned exactly the candidate the memo was computed from (it was reached through the same bone, whose own entry
// came from the memo): then the value IS what _connection_value returns, bit for bit.  Anything else -- a joint reached
// through another bone first, a batch that did not fit -- is evaluated on demand as before.  The sequence of heap
// operations, and with it every tie, is the reference's.
// The candidates are also this growth's PREDICTIONS: their occupancy boxes are published level by level (a pose of 17
// joints is known after ~8 levels, long before the heap loop has assigned it), which is what keeps the other growers off
// the same person.  Advisory as before: the commit re-tests every seed against the final boxes.

// boxes of the joints that became candidates in this round: published like the boxes of assigned joints (publish_joint)
template <int WR>
__device__ __forceinline__ void spec_publish(ImageCtx& c, const DevParams& p) {
    const int lane = lane_id();
    bool any_new = false;
    for (int k0 = 0; k0 < c.F; k0 += kWave) {
        const int k = k0 + lane;
        const int w = k < c.F ? c.sp_j[k] : 0;
        if (w & kSpNext) {
            const int pb = (c.slot_info[w >> 8] >> 16) & 0xff;
            c.jbox[k] = occ_box(c, p, (double)c.sp_x[pb], (double)c.sp_y[pb], (double)c.sp_s[pb]);
            any_new = true;
        }
    }
    if (__ballot(any_new) == 0ull) return;
    wave_sync();
    // every pooled seed that comes later in seed order and lies in one of this growth's boxes is shadowed (re-testing the
    // boxes of earlier rounds sets the same bits again)
    unsigned bits = 0u;
#pragma unroll
    for (int r0 = 0; r0 < WR; r0 += 4) {
        int sif[4], spk[4];
#pragma unroll
        for (int r = 0; r < 4; r++) { sif[r] = c.pool_if[(r0 + r) * kWave + lane]; spk[r] = c.pool_pack[(r0 + r) * kWave + lane]; }
#pragma unroll
        for (int r = 0; r < 4; r++) asm volatile("" : "+v"(sif[r]), "+v"(spk[r]) :: "memory");
        OccBox bx[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int f = (int)((unsigned)sif[r] >> 24);
            bx[r] = c.jbox[f < c.F ? f : 0];
        }
#pragma unroll
        for (int r = 0; r < 4; r++) asm volatile("" : "+v"(bx[r].minx), "+v"(bx[r].miny), "+v"(bx[r].maxx), "+v"(bx[r].maxy) :: "memory");
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int f = (int)((unsigned)sif[r] >> 24), idx = sif[r] & kPoolIdxMask;
            if (idx != kPoolIdxMask && idx > c.my_idx && f < c.F &&
                box_contains(bx[r], spk[r] & 0xfff, (spk[r] >> 12) & 0xfff)) bits |= 1u << (r0 + r);
        }
    }
    if (bits) atomicOr(&c.shadow_mine[lane], bits);
    ++c.n_pub;
    if (lane == 0) *c.pub = c.n_pub;
    // a new candidate inside the box an EARLIER live growth has published for the same joint: most likely the same person
    // (publish_joint's test: lane = grower, the new joints one after the other)
    const int cfg = c.head_g[2];
    if ((cfg >> 28) & 1) {
        const TaskSlot* tasks = reinterpret_cast<const TaskSlot*>(opa_dyn_lds);
        const unsigned char* blocks = opa_dyn_lds + c.head_g[1];
        const int n_growers = (cfg >> 20) & 0xff, block_bytes = cfg & 0xfffff;
        bool earlier = false; int sd = 0;
        if (lane >= 1 && lane <= n_growers && lane != c.wave) {
            const int st = flag_peek(&tasks[lane].state), cn = flag_peek(&tasks[lane].cancel);
            sd = tasks[lane].seed;
            earlier = (st == kTaskAssigned || st == kTaskDone) && !cn && sd < c.my_idx;
        }
        if (__ballot(earlier) != 0ull) {
            unsigned key = 0xFFFFFFFFu;
            const OccBox* theirs = reinterpret_cast<const OccBox*>(blocks + (size_t)(earlier ? lane - 1 : 0) * block_bytes);
            for (int k0 = 0; k0 < c.F; k0 += kWave) {
                const int kk = k0 + lane;
                unsigned long long nm = __ballot(kk < c.F && (c.sp_j[kk < c.F ? kk : 0] & kSpNext));
                while (nm) {
                    const int k = k0 + __builtin_ctzll(nm);
                    nm &= nm - 1;
                    const int pb = (c.slot_info[c.sp_j[k] >> 8] >> 16) & 0xff;
                    int cx, cy;
                    occ_xy(c, p, (double)c.sp_x[pb], (double)c.sp_y[pb], &cx, &cy);
                    const OccBox ob = theirs[k];
                    if (earlier && box_contains(ob, cx, cy)) key = min(key, ((unsigned)sd << 6) | (unsigned)lane);
                }
            }
            if (__ballot(key != 0xFFFFFFFFu) != 0ull) {
                const unsigned first = ~wave_max_u32(~key);      // the earliest of them
                if (lane == 0) flag_store(const_cast<int*>(&tasks[c.wave].coll), (int)(first & 63u));
            }
        }
    }
}

// One round = ONE batch: the forward scans of the bones leaving the joints that became candidates in the round before
// TOGETHER with the reverse scans (cifcaf.cpp:397-409) of that round's connections -- a candidate's values are known
// after its forward scan, the reverse scan only confirms or rejects the connection.  A rejected connection leaves its
// candidate (and what was computed from it) in the memo, where nothing will ever match it: the search never assigns a
// joint through a rejected bone.
template <int WR>
__device__ __forceinline__ void spec_phase(ImageCtx& c, const DevParams& p, bool reverse_match_, double filter_sigmas) {
    const int lane = lane_id();
    const int K = c.K, A = c.A, E = 2 * A;
    int* joblist = reinterpret_cast<int*>(c.tgt);            // (the scan area is free between two batches)
    for (int k = lane; k < K; k += kWave) c.sp_j[k] = c.jv[k] != 0.0 ? (kSpActual | kSpOpen) : 0;
    for (int a = lane; a < A; a += kWave) c.sp_b[a] = 0;
    wave_sync();
    bool pending = false;                                    // connections whose reverse scan is outstanding (uniform)
    for (;;) {
        int nj = 0;
        // ---- reverse jobs: the connections of the round before
        if (pending)
            for (int a0 = 0; a0 < A; a0 += kWave) {
                const int bn = a0 + lane;
                const int w = bn < A ? c.sp_b[bn] : 0;
                const bool cand = (w & 0xff) == kSbPending;
                const unsigned long long m = __ballot(cand);
                const int slot = nj + prefix_count(m);
                if (cand && slot < kWave) joblist[slot] = (w >> 8) | (1 << 16);
                nj += __popcll(m);
                if (nj >= kWave) break;
            }
        const bool rev_left = nj > kWave;                    // (more than a batch holds: the rest in the next round)
        if (nj > kWave) nj = kWave;
        // ---- forward jobs: the first slot of every (start, end) pair whose start joint is open, whose end is not filled
        //      from the start (cifcaf.cpp:327,336) and whose bone has not been evaluated in either direction; of a bone
        //      between two open joints the direction from the lower joint
        bool fwd_left = nj >= kWave;                         // (the batch is full of reverse jobs: the open joints wait)
        for (int t0 = 0; t0 < E && nj < kWave; t0 += kWave) {
            const int t = t0 + lane;
            bool cand = false;
            if (t < E) {
                const int info = c.slot_info[t];
                const int a = info & 0xff, b = (info >> 8) & 0xff, bn = (info >> 16) & 0xff;
                const int wa = c.sp_j[a], wb = c.sp_j[b];
                cand = c.adj_first[t] == t && (wa & kSpOpen) && (wb & kSpKind) != kSpActual && (c.sp_b[bn] & 0xff) == 0 &&
                       !((wb & kSpOpen) && b < a);
            }
            const unsigned long long m = __ballot(cand);
            const int slot = nj + prefix_count(m);
            if (cand && slot < kWave) joblist[slot] = t;
            nj += __popcll(m);
            if (nj > kWave) { fwd_left = true; nj = kWave; }
            else if (nj == kWave && t0 + kWave < E) fwd_left = true;       // (possibly: look again next round)
        }
        if (nj == 0) break;
        wave_sync();
        PH(20);
        const bool job = lane < nj;
        const int jw = job ? joblist[lane] : 0;
        wave_sync();
        const int t = jw & 0xffff;
        const bool is_rev = (jw >> 16) != 0;
        const int info = c.slot_info[t];
        const int a = info & 0xff, b = (info >> 8) & 0xff, bn = (info >> 16) & 0xff, fwd = (info >> 24) & 1;
        // the start joint: filled from the start, or the candidate its bone gave it
        const int wa = c.sp_j[a];
        double sv; float sxf, syf, ssf;
        if ((wa & kSpKind) == kSpActual) { sv = c.jv[a]; sxf = c.jx[a]; syf = c.jy[a]; ssf = c.js[a]; }
        else { const int pb = (c.slot_info[wa >> 8] >> 16) & 0xff; sv = c.sp_v[pb]; sxf = c.sp_x[pb]; syf = c.sp_y[pb]; ssf = c.sp_s[pb]; }
        const double sx = (double)sxf, sy = (double)syf, ss = (double)ssf;
        // forward: the start joint against the bone's list in its direction; reverse: the new joint against the other list
        double qx = sx, qy = sy, qs = ss;
        if (is_rev) { qx = (double)c.sp_x[bn]; qy = (double)c.sp_y[bn]; qs = (double)c.sp_s[bn]; }
        const int li = bn * 2 + ((fwd != 0) != is_rev ? 0 : 1);
        PH(21);
        c.n_blend += nj;
        const SpecOut o = spec_scan_batch(c, job, li, qx, qy, qs, filter_sigmas);
        int res = 0;
        if (job) {
            if (o.ok < 0) res = kSbUnknown;
            else if (o.ok == 0) res = kSbRejected;                                          // :384, :400-403
            else if (is_rev) res = fabs(sx - (double)o.x) + fabs(sy - (double)o.y) > ss ? kSbRejected : kSbOk;   // :404
            else {
                const double nv = sqrt(o.v * sv);                                           // :386
                if (nv < p.keypoint_threshold || nv < sv * p.keypoint_threshold_rel) res = kSbRejected;          // :387-390
                else {
                    res = p.reverse_match && reverse_match_ && a < c.F ? kSbPending : kSbOk;                     // :397
                    c.sp_v[bn] = nv; c.sp_x[bn] = o.x; c.sp_y[bn] = o.y; c.sp_s[bn] = o.s;
                }
            }
            c.sp_b[bn] = res | (t << 8);
        }
        pending = __ballot(job && res == kSbPending) != 0ull || rev_left;
        wave_sync();
        // the level is done (unless jobs were left over): the joints that became candidates in the round before are closed,
        // the end joint of every new connection becomes a candidate -- unless it is one already (first come) -- and open
        if (!fwd_left)
            for (int k = lane; k < K; k += kWave) { const int w = c.sp_j[k]; if (w & kSpOpen) c.sp_j[k] = w & ~kSpOpen; }
        wave_sync();
        if (job && !is_rev && (res == kSbOk || res == kSbPending)) atomicCAS(&c.sp_j[b], 0, kSpCand | kSpNext | kSpOpen | (t << 8));
        wave_sync();
        PH(28);
        if (c.pub) spec_publish<WR>(c, p);
        for (int k = lane; k < K; k += kWave) { const int w = c.sp_j[k]; if (w & kSpNext) c.sp_j[k] = w & ~kSpNext; }
        wave_sync();
        PH(29);
        if (poll_task<WR>(c)) { c.aborted = 1; return; }
        PH(30);
    }
}

// Occupancy::get on the bitmap (one bit per cell, rows of occ_wpr 32-bit words)
__device__ __forceinline__ bool occ_test(const ImageCtx& c, int f, int xi, int yi) {
    const unsigned w = c.occ[((size_t)f * c.occ_h + yi) * c.occ_wpr + (xi >> 5)];
    return (w >> (xi & 31)) & 1u;
}

// Private LDS block of one growing wave: the pose's occupancy boxes and the pose (the coordinator reads both at a
// commit), for the LDS variant the frontier (entries + heap + in_frontier, one per directed-bone slot), and the scan
// area (`tgt_floats`: kBlendLdsFloats, or kSmallTgtFloats when that buys more growers).  16-byte sized.
constexpr int kSmallTgtChunks = 2;
constexpr int kSmallTgtFloats = 3 * kSmallTgtChunks * kWave + 4 * kWave;
constexpr int kSpecTgtFloats = kSpecItems + 2 * kWave + kSpecEntryWords * 160;   // the small scan area when the level walk runs: batches of up to 160 passing entries
__host__ __device__ inline size_t assoc_pose_bytes(int K) {
    return (16 * (size_t)K + sizeof(double) * K + sizeof(float) * 3 * K + 15) / 16 * 16;
}
// the level walk's memo (spec_phase): a word per joint, a word + a connection value per bone; the LDS variant also keeps its two flag sets here
__host__ __device__ inline size_t assoc_spec_bytes(int K, int A, bool reg) {
    return (sizeof(double) * A + sizeof(int) * (K + A) + sizeof(float) * 3 * A + (reg ? 0 : K + A) + 15) / 16 * 16;
}
// the scan helpers' words (requests, a word per bone, the slot each joint was assigned through, control words); the
// register variant also keeps an LDS copy of its computed entries here (what the helpers read and write)
__host__ __device__ inline size_t assoc_help_bytes(int K, int A, bool reg) {
    return (sizeof(int) * (kHelpRing + A + K + 4) + (reg ? (sizeof(double) + 3 * sizeof(float)) * A + 8 : 0) + 15) / 16 * 16;
}
__host__ __device__ inline size_t assoc_private_bytes(int K, int A, bool reg, int tgt_floats, bool spec = false, bool help = false) {
    size_t b = assoc_pose_bytes(K);
    if (!reg) b += sizeof(double) * A + sizeof(unsigned long long) * A + sizeof(float) * 3 * A + (A + 15) / 16 * 16 + (A & 1 ? 4 : 0);
    b = (b + 15) / 16 * 16;
    if (spec) b += assoc_spec_bytes(K, A, reg);
    if (help) b += assoc_help_bytes(K, A, reg);
#ifdef OPA_ASSOC_PRIVATE_PAD     // experiment: other distances between the growers' blocks (LDS bank conflicts)
    b += OPA_ASSOC_PRIVATE_PAD;
#endif
    return b + sizeof(float) * tgt_floats;
}
template <bool REG>
__device__ __forceinline__ void carve_private(ImageCtx& c, unsigned char* sp, int tgt_floats, bool spec = false, bool help = false) {
    const int K = c.K, A = c.A;
    unsigned char* base = sp;
    c.jbox = (OccBox*)sp; sp += sizeof(OccBox) * K;
    c.jv = (double*)sp; sp += sizeof(double) * K;
    c.jx = (float*)sp; sp += sizeof(float) * K;
    c.jy = (float*)sp; sp += sizeof(float) * K;
    c.js = (float*)sp; sp += sizeof(float) * K;
    sp = base + assoc_pose_bytes(K);
    c.e_v = nullptr; c.heap = nullptr; c.e_x = c.e_y = c.e_s = nullptr; c.in_frontier = nullptr;
    if constexpr (!REG) {
        c.e_v = (double*)sp; sp += sizeof(double) * A;
        c.heap = (unsigned long long*)sp; sp += sizeof(unsigned long long) * A;
        c.e_x = (float*)sp; sp += sizeof(float) * A;
        c.e_y = (float*)sp; sp += sizeof(float) * A;
        c.e_s = (float*)sp; sp += sizeof(float) * A;
        c.in_frontier = sp; sp += (A + 15) / 16 * 16 + (A & 1 ? 4 : 0);
        sp = base + (((size_t)(sp - base) + 15) & ~(size_t)15);
    }
    c.sp_j = c.sp_b = nullptr; c.sp_v = nullptr; c.sp_x = c.sp_y = c.sp_s = nullptr; c.sp_match = c.sp_fromc = nullptr;
    c.sp_pcap = 0; c.n_hit = c.n_miss = 0;
    if (kWalk && spec) {
        unsigned char* s0 = sp;
        c.sp_v = (double*)sp; sp += sizeof(double) * A;
        c.sp_j = (int*)sp; sp += sizeof(int) * K;
        c.sp_b = (int*)sp; sp += sizeof(int) * A;
        c.sp_x = (float*)sp; sp += sizeof(float) * A;
        c.sp_y = (float*)sp; sp += sizeof(float) * A;
        c.sp_s = (float*)sp; sp += sizeof(float) * A;
        if constexpr (!REG) { c.sp_match = sp; sp += K; c.sp_fromc = sp; sp += A; }
        sp = s0 + assoc_spec_bytes(K, A, REG);
        c.sp_pcap = spec_pcap(tgt_floats);
    }
    c.hq = c.rq = c.jsrc = c.hctl = nullptr;
    if (kHelp && help) {
        unsigned char* s0 = sp;
        if constexpr (REG) {                                 // (the LDS variant's entry arrays are the ones above)
            c.e_v = (double*)sp; sp += sizeof(double) * A + (A & 1 ? 8 : 0);
            c.e_x = (float*)sp; sp += sizeof(float) * A;
            c.e_y = (float*)sp; sp += sizeof(float) * A;
            c.e_s = (float*)sp; sp += sizeof(float) * A;
        }
        c.hq = (int*)sp; sp += sizeof(int) * kHelpRing;
        c.rq = (int*)sp; sp += sizeof(int) * A;
        c.jsrc = (int*)sp; sp += sizeof(int) * K;
        c.hctl = (int*)sp; sp += sizeof(int) * 4;
        sp = s0 + assoc_help_bytes(K, A, REG);
    }
    c.tgt = (float*)sp;
    c.max_r = tgt_floats >= kBlendLdsFloats ? kBlendChunks : kSmallTgtChunks;
    c.heap_n = 0;
}

// ---- scan helpers, the helping wave's side: ONE connection of the growth in block `mblock` (the one that holds the head
// seed), if any is waiting.  The bones leaving a posted joint are claimed one at a time (compare-and-swap on the bone's
// word), evaluated with this wave's own scan area, and left in the master's entry array.
template <bool REG>
__device__ __forceinline__ void help_once(ImageCtx& c, const DevParams& p, unsigned char* mblock, int tgt_floats, bool spec) {
    ImageCtx m;
    m.K = c.K; m.A = c.A;
    carve_private<REG>(m, mblock, tgt_floats, spec, true);
    const int lane = lane_id();
    if (flag_load(&m.hctl[1])) return;                       // not growing
    const int ep = flag_load(&m.hctl[0]);
    const int free_w = rq_word(kRqFree, 0, 0, ep);
    const int hw = lane < kHelpRing ? flag_peek(&m.hq[lane]) : 0;
    unsigned long long vm = __ballot(hw != 0 && ((hw >> 19) & 0x3ff) == ep);
    while (vm) {
        const int i = __builtin_ctzll(vm); vm &= vm - 1;
        const int w = rlane_i(hw, i);
        const int b = (w >> 1) & 0x1ff, u = (w >> 10) & 0x1ff;
        const int parent = u == kSrcStart ? -1 : (c.slot_info[u] & 0xff);
        const int t1 = c.adj_off[b + 1];
        for (int t0 = c.adj_off[b]; t0 < t1; t0 += kWave) {
            const int t = t0 + lane;
            bool cand = false; int bn = 0;
            if (t < t1) {
                const int info = c.slot_info[t];
                const int other = (info >> 8) & 0xff; bn = (info >> 16) & 0xff;
                bool assigned;
                if constexpr (REG) assigned = (((((unsigned long long)(unsigned)m.hctl[3]) << 32) | (unsigned)m.hctl[2]) >> other) & 1ull;
                else assigned = m.jv[other] > 0.0;
                cand = c.adj_first[t] == t && other != parent && !assigned && flag_peek(&m.rq[bn]) == free_w;
            }
            unsigned long long cm = __ballot(cand);
            while (cm) {
                const int l = __builtin_ctzll(cm); cm &= cm - 1;
                const int ts = t0 + l, bs = rlane_i(bn, l);
                const int mine = rq_word(kRqTaken, ts, u, ep);
                int old = 0;
                if (lane == 0) old = atomicCAS(&m.rq[bs], free_w, mine);
                if (__builtin_amdgcn_readfirstlane(old) != free_w) continue;   // somebody else has it
                // the block closed, or went on to its next growth, in between: hand the bone back untouched
                if (flag_load(&m.hctl[1]) || flag_load(&m.hctl[0]) != ep) {
                    if (lane == 0) atomicCAS(&m.rq[bs], mine, free_w);
                    return;
                }
                double sv; float sx, sy, ss;
                if (u == kSrcStart) { sv = m.jv[b]; sx = m.jx[b]; sy = m.jy[b]; ss = m.js[b]; }
                else { const int bu = (c.slot_info[u] >> 16) & 0xff; sv = m.e_v[bu]; sx = m.e_x[bu]; sy = m.e_y[bu]; ss = m.e_s[bu]; }
                double nv = 0.0; float nx = 0.f, ny = 0.f, ns = 0.f;
                const bool ok = connection_value_at<false>(c, p, c.slot_info[ts], sv, (double)sx, (double)sy, (double)ss, true, 1.0,
                                                           &nv, &nx, &ny, &ns);
                if (ok && lane == 0) { m.e_v[bs] = nv; m.e_x[bs] = nx; m.e_y[bs] = ny; m.e_s[bs] = ns; }
                wave_sync();
                if (lane == 0) flag_store(&m.rq[bs], rq_word(ok ? kRqOk : kRqRej, ts, u, ep));
                return;                                      // one connection at a time: back to the own task slot
            }
        }
        if (lane == 0) atomicCAS(&m.hq[i], w, 0);            // nothing left of this request
    }
}

// LDS scratch of one wave during keypoint NMS (aliases the growth state): a box and a cell per pose
__host__ __device__ inline size_t nms_scratch_bytes(int max_ann) {
    return (sizeof(int) * 4 + sizeof(int) * 2) * (size_t)max_ann;
}
// ... behind the arrays all waves share during the NMS: score, suppression bits, order, rank
__host__ __device__ inline size_t nms_shared_bytes(int max_ann, int K) {
    const int KC = (K + kWave - 1) / kWave;
    return (sizeof(double) * max_ann + sizeof(unsigned long long) * (size_t)max_ann * KC + sizeof(int) * 2 * max_ann + 15) / 16 * 16;
}

struct PoseView { const OccBox* box; const double* v; const float *x, *y, *s; };

__device__ __forceinline__ PoseView pose_of_block(unsigned char* private_base, int block, size_t private_bytes, int K) {
    unsigned char* sp = private_base + (size_t)block * private_bytes;
    PoseView q;
    q.box = (const OccBox*)sp; sp += sizeof(OccBox) * K;
    q.v = (const double*)sp; sp += sizeof(double) * K;
    q.x = (const float*)sp; q.y = q.x + K; q.s = q.y + K;
    return q;
}

// occupancy boxes of the pose this wave just grew (empty box for an unfilled joint or one without a field)
__device__ __forceinline__ void pose_boxes(ImageCtx& c, const DevParams& p) {
    for (int k = lane_id(); k < c.K; k += kWave) {
        OccBox b; b.minx = b.miny = b.maxx = b.maxy = 0;
        if (k < c.F && c.jv[k] != 0.0) b = occ_box(c, p, (double)c.jx[k], (double)c.jy[k], (double)c.js[k]);
        c.jbox[k] = b;
    }
}

// Occupancy::set for every filled joint of a pose (cifcaf.cpp:225-229), by the 64 lanes of ONE wave: four
// boxes per step, 16 rows each, fire-and-forget atomics (nothing waits for them; the reader fences).
__device__ __forceinline__ void occ_mark_pose(const ImageCtx& c, const PoseView& q) {
    const int lane = lane_id(), sub = lane & 15, grp = lane >> 4;
    for (int f0 = 0; f0 < c.F; f0 += 4) {
        const int f = f0 + grp;
        if (f >= c.F) continue;
        const OccBox b = q.box[f];                    // empty for an unfilled joint
        for (int yy = b.miny + sub; yy < b.maxy; yy += 16) {
            unsigned* row = c.occ + ((size_t)f * c.occ_h + yy) * c.occ_wpr;
            for (int w = b.minx >> 5; w <= (b.maxx - 1) >> 5; w++) {
                const int lo = max(b.minx - w * 32, 0), hi = min(b.maxx - w * 32, 32);
                const unsigned mask = (hi >= 32 ? 0xFFFFFFFFu : (1u << hi) - 1u) & ~((1u << lo) - 1u);
                atomicOr(row + w, mask);
            }
        }
    }
}

// nms_keypoints.hpp:25-32 on an LDS pose
__device__ __forceinline__ double pose_score(const double* v, int K) {
    double acc = 0.0;
    for (int k = 0; k < K; k++) { const float i = (float)acc; acc = (double)i + v[k]; }
    return acc / (double)K;
}

// ------------------------------------------------------------------- kernel
// One LDS task slot per wave: the coordinator hands a seed to grower g by filling task[g] and setting
// state = ASSIGNED; the grower answers DONE (pose, boxes and score are in its private block / slot) or, if
// `cancel` was raised while it grew, IDLE.  Only the coordinator moves a slot out of DONE: to IDLE (result
// dropped) or to ACCEPTED -- then the grower itself marks the pose's boxes in the bitmap and stores the pose at
// scratch slot `pad0` (-1: not stored), off the coordinator's critical path, and returns to IDLE.
// (`cancel` and `epoch` share an aligned 8 bytes: a grower polls both with one LDS read between frontier pops)
constexpr int kAssocTrace = 64;           // commits recorded per image in the optional trace ("assoc_trace")

// statistics of one image, int32[kAssocStats] in the workspace ("assoc_stats"): see include/openpifpaf_amd.h
constexpr int kAssocStats = 24;
// The coordinator and the growers wait for each other in LDS polling loops.  A protocol error must not hang
// the device: after this many 10-ns ticks inside one launch every wait gives up, the image reports no poses
// and status -1 (never seen in the tests; one second is ~1000x the slowest image).
constexpr long long kWatchdogTicksDefault = 100000000ll;   // (OPA_ASSOC_WATCHDOG_TICKS overrides: tests provoke the failure)

// ---- keypoint NMS + output (nms_keypoints.cpp:17-70, cifcaf.cpp:246-261), by all threads of a workgroup, on the
// poses stored in the HBM scratch `anns`.  Shared by the seed kernel (default flags) and the force-complete kernel.
struct NmsLds { double* nms_score; unsigned long long* nms_supp; int* nms_order; int* nms_rank; unsigned char* work_base; };
// the NMS arrays alias the work area (the growers' private blocks are free by then): shared arrays first, then one
// scratch block per wave
__device__ __forceinline__ NmsLds nms_carve(unsigned char* work_base, int max_ann, int K) {
    const int KC = (K + kWave - 1) / kWave;
    NmsLds l; unsigned char* sp = work_base;
    l.nms_score = (double*)sp; sp += sizeof(double) * max_ann;
    l.nms_supp = (unsigned long long*)sp; sp += sizeof(unsigned long long) * (size_t)max_ann * KC;
    l.nms_order = (int*)sp; sp += sizeof(int) * max_ann;
    l.nms_rank = (int*)sp;
    l.work_base = work_base + nms_shared_bytes(max_ann, K);
    return l;
}
template <int kThreads>
__device__ __forceinline__ void nms_and_store(const AssocArgs& a, const DevParams& p, const ImageCtx& c, const NmsLds& l, int b,
                                              int n_kept, int n_dropped, int failed, double* anns, const int64_t* ann_ids,
                                              int nms_waves) {
    const int tid = threadIdx.x, lane = lane_id(), wave = c.wave, K = a.K;
    const int KC = (K + kWave - 1) / kWave;
    double* nms_score = l.nms_score; unsigned long long* nms_supp = l.nms_supp;
    int* nms_order = l.nms_order; int* nms_rank = l.nms_rank; unsigned char* work_base = l.work_base;
    // ---- keypoint NMS, nms_keypoints.cpp:17-70
    for (int n = tid; n < n_kept; n += kThreads) {       // UniformScore of every stored pose
        const double* src = anns + (size_t)n * K * 4;
        double acc = 0.0;
        for (int k = 0; k < K; k++) { const float i = (float)acc; acc = (double)i + src[4 * k]; }
        nms_score[n] = acc / (double)K;
    }
    __syncthreads();
    for (int n = tid; n < n_kept; n += kThreads) {       // rank by score desc (ties: creation order)
        const double sn = nms_score[n];
        int rank = 0;
        for (int m = 0; m < n_kept; m++) { const double sm = nms_score[m]; rank += (sm > sn || (sm == sn && m < n)) ? 1 : 0; }
        nms_order[rank] = n;
        for (int kc = 0; kc < KC; kc++) nms_supp[n * KC + kc] = 0ull;
    }
    __syncthreads();
    // Occupancy pass (:27-43) without a map: joints of different fields never interact, so wave w takes
    // fields w, w+nms_waves, ...; per field the poses are visited in score order and pose r's joint is suppressed
    // iff its cell lies in the box of an earlier, still unsuppressed joint (= Occupancy::get after the
    // earlier Occupancy::set calls).  Boxes and cells of the field sit in this wave's LDS scratch.
    {
        unsigned char* nsp = work_base + (size_t)wave * nms_scratch_bytes(a.max_ann);
        OccBox* my_box = (OccBox*)nsp;
        int2* my_cell = (int2*)(nsp + sizeof(OccBox) * a.max_ann);
        // :27-30: only joints with an occupancy field take part; nms_waves = waves whose scratch fits the LDS
        for (int k = wave; k < a.F && wave < nms_waves; k += nms_waves) {
            for (int r = lane; r < n_kept; r += kWave) {
                const double* pose = anns + ((size_t)nms_order[r] * K + k) * 4;
                OccBox bx; bx.minx = bx.miny = bx.maxx = bx.maxy = 0;
                int2 cell; cell.x = -1; cell.y = -1;          // v == 0: neither tested nor set (:36)
                if (pose[0] != 0.0) {
                    occ_xy(c, p, pose[1], pose[2], &cell.x, &cell.y);
                    bx = occ_box(c, p, pose[1], pose[2], pose[3]);
                }
                my_box[r] = bx; my_cell[r] = cell;
            }
            wave_sync();
            for (int r = 1; r < n_kept; r++) {
                const int2 cell = my_cell[r];
                if (cell.x < 0) continue;
                bool cover = false;
                for (int q = lane; q < r; q += kWave) cover |= box_contains(my_box[q], cell.x, cell.y);
                if (__ballot(cover) != 0ull) {                // :37-38 suppressed, sets no box
                    if (lane == 0) {
                        atomicOr(&nms_supp[r * KC + (k >> 6)], 1ull << (k & 63));
                        OccBox e; e.minx = e.miny = e.maxx = e.maxy = 0;
                        my_box[r] = e;
                    }
                    wave_sync();
                }
            }
            wave_sync();
        }
    }
    __syncthreads();
    // suppression, keypoint threshold, instance threshold (:50,58-66); one thread per pose
    for (int r = tid; r < n_kept; r += kThreads) {
        double* pose = anns + (size_t)nms_order[r] * K * 4;
        double acc = 0.0;
        for (int k = 0; k < K; k++) {
            double v = pose[4 * k];
            if ((nms_supp[r * KC + (k >> 6)] >> (k & 63)) & 1ull) v *= p.nms_suppression;
            if (!(v > p.nms_keypoint_threshold)) v = 0.0;
            pose[4 * k] = v;
            const float i = (float)acc; acc = (double)i + v;
        }
        nms_score[r] = acc / (double)K;                  // indexed by sorted position r now
    }
    sync_global();                                       // the rewritten confidences are read by other threads below
    for (int r = tid; r < n_kept; r += kThreads) {       // final order (:69); ties keep the previous order
        const double sr = nms_score[r];
        int rank = -1;
        if (!(sr < p.nms_instance_threshold)) {
            rank = 0;
            for (int m = 0; m < n_kept; m++) {
                const double sm = nms_score[m];
                if (sm < p.nms_instance_threshold) continue;
                rank += (sm > sr || (sm == sr && m < r)) ? 1 : 0;
            }
        }
        nms_rank[r] = rank;
    }
    __syncthreads();
    float* out = a.out + (size_t)b * a.max_ann * K * 4;
    int64_t* out_ids = a.out_ids + (size_t)b * a.max_ann;
    int n_out = 0;
    for (int r = 0; r < n_kept; r++) n_out += nms_rank[r] >= 0 ? 1 : 0;
    for (int idx = tid; idx < n_kept * K; idx += kThreads) {          // cifcaf.cpp:250-258
        const int r = idx / K, k = idx - r * K;
        const int dst = nms_rank[r];
        if (dst < 0) continue;
        const int src_n = nms_order[r];
        const double* pose = anns + (size_t)src_n * K * 4;
        float4 o;
        o.x = (float)pose[4 * k]; o.y = (float)pose[4 * k + 1];
        o.z = (float)pose[4 * k + 2]; o.w = (float)pose[4 * k + 3];
        reinterpret_cast<float4*>(out)[(size_t)dst * K + k] = o;
        if (k == 0) out_ids[dst] = ann_ids[src_n];
    }
    if (tid == 0) {
        // rows [0, n_out) are valid; poses dropped for lack of capacity raise the overflow flag
        a.out_count[b] = n_out | (n_dropped > 0 ? OPA_COUNT_OVERFLOW : 0) | (failed ? OPA_COUNT_FAILED : 0);
        a.status[b] = failed ? -failed : n_dropped;      // (failure codes: see the seed kernel)
    }
}

// One image, by the NW waves of a workgroup (the kernel below calls it once -- one workgroup per image -- or, as a persistent
// workgroup, for one image after the other).  `smem`: the workgroup's dynamic LDS; every word of it that is read is written here first.
template <bool REG, int NW>
__device__ __forceinline__ void assoc_image(const AssocArgs& a, const DevSkeleton& sk, const DevParams& p,
                                            const int n_growers, const int nms_waves, const int tgt_floats, const int b,
                                            unsigned char* smem) {
    constexpr int kThreads = NW * kWave;
    constexpr int WR = REG ? kPoolSlots : kPoolSlotsLds;     // seed-pool slots per coordinator lane
    const bool use_bbox = REG && a.list_bbox != nullptr;
    const int tid = threadIdx.x, lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int K = a.K, A = a.A, E = 2 * A;
    const int S = n_growers;                         // growers = waves 1..S, each with a private LDS block
    const long long t_kernel = wall_clock64();
    const long long kWatchdogTicks = a.watchdog_ticks;

    // ---- seeds of equal score into the order the reference's unstable std::sort leaves them in (cif_seeds.cpp:94), this
    // image's only: images without equal scores leave after one look at their sorted scores, and nobody waits for
    // another image's ties.  The pass borrows the whole LDS block; what it stores is read after the sync_global() below.
    if (a.tie_fused) {
        cifseeds_tie_body<kThreads>(a.tie, a.tie_sort, p, b, smem);
        sync_global();                               // (the seed list it re-ordered is read by every wave below)
    }

    ImageCtx c;
    c.K = K; c.A = A; c.F = a.F; c.wave = wave;
    c.lists = a.lists + (size_t)b * A * 2 * 7 * a.list_cap;
    c.list_counts = a.list_counts + (size_t)b * A * 2;
    c.list_cap = a.list_cap;
    c.raw_caf = a.predict && a.caf_raw ? a.caf_raw + (size_t)b * A * 8 * a.list_cap : nullptr;
    c.raw_w = a.caf_w; c.raw_stride = a.caf_stride; c.predict_th = a.predict_th; c.n_predicted = nullptr; c.coll_shift = a.coll_shift;
    c.occ_h = a.occ_h; c.occ_w = a.occ_w; c.occ_wpr = (a.occ_w + 31) >> 5;
    c.occ = a.occ + (size_t)b * a.occ_image_words;
    c.cancel = nullptr; c.aborted = 0; c.n_blend = 0; c.t_blend = 0; c.t_blend_mem = 0; c.pub = nullptr; c.n_pub = 0;
    c.pool_if = nullptr; c.pool_pack = nullptr; c.shadow_mine = nullptr; c.my_idx = 0;
    c.pool_ep = nullptr; c.epoch = nullptr; c.ack = nullptr; c.my_epoch = 0;
    c.head_g = nullptr; c.prio = 0;
    c.bbox = nullptr;
    c.nb = a.bbox_chunks;
    c.gbbox = a.list_bbox ? reinterpret_cast<const float4*>(a.list_bbox) + (size_t)b * 2 * a.A * a.bbox_chunks : nullptr;

    // ---- LDS carve: shared part, then the work area: one private block per growing wave while poses grow, the
    //      keypoint-NMS arrays afterwards
    unsigned char* sp = smem;
    TaskSlot* task = (TaskSlot*)sp; sp += sizeof(TaskSlot) * NW;            // (16-byte aligned: first)
    unsigned long long* dedup = (unsigned long long*)sp; sp += sizeof(unsigned long long) << kDedupBits;   // refill: first seed of a cell
    c.sh_counts = (int*)sp; sp += sizeof(int) * E;
    int* l_off = (int*)sp; sp += sizeof(int) * (K + 1);
    int* l_info = (int*)sp; sp += sizeof(int) * E;
    int* l_first = (int*)sp; sp += sizeof(int) * E;
    int* sh_ctl = (int*)sp; sp += sizeof(int) * 16;  // 0 exit flag, 1 n_kept, 2 n_dropped, 3 grower ticks, 4 list scans, 5 watchdog,
                                                     // 6-7 scan timing (diagnostic builds), 8 refill epoch, 9 grower of the head seed,
                                                     // 10-11 see below, 12 the head seed's index (self-serve hand-out)
    int* sh_stats = (int*)sp; sp += sizeof(int) * kAssocStats;
    int* pool_if = (int*)sp; sp += sizeof(int) * WR * kWave;        // the coordinator's seed pool, mirrored for the growers
    int* pool_pack = (int*)sp; sp += sizeof(int) * WR * kWave;
    int* pool_ep = (int*)sp; sp += sizeof(int) * WR * kWave;
    int* pool_own = (int*)sp; if (kSelfServe) sp += sizeof(int) * WR * kWave;   // 0, or the grower that claimed the slot's seed (self-serve variant only)
    unsigned* shadow_by = (unsigned*)sp; sp += sizeof(unsigned) * NW * kWave; // [grower][lane]: pool slots in its published boxes
    int* stage_f = (int*)sp; sp += sizeof(int) * kSeedStage;                // the next seeds' field and cell, staged ahead of the pool refill
    int* stage_pk = (int*)sp; sp += sizeof(int) * kSeedStage;
    sp = smem + (((size_t)(sp - smem) + 15) & ~(size_t)15);
    float4* sh_bbox = (float4*)sp;                   // chunk boxes of the caf_th lists (register variant only)
    if (use_bbox) sp += sizeof(float4) * E * kListBboxChunks;
    unsigned char* work_base = sp;                   // growth phase: private blocks; NMS phase: its arrays and scratch
    unsigned char* private_base = sp;
    const bool help_on = kHelp && a.help != 0;
    const size_t private_bytes = assoc_private_bytes(K, A, REG, tgt_floats, a.spec != 0, help_on);
    carve_private<REG>(c, private_base + (size_t)(wave >= 1 && wave <= S ? wave - 1 : 0) * private_bytes, tgt_floats, a.spec != 0, help_on);   // other waves never touch theirs
    c.hq_n = 0; c.h_epoch = 0;
    if (help_on && wave >= 1 && wave <= S && lane == 0) { c.hctl[0] = 0; c.hctl[1] = 1; }     // (no growth yet: closed to helpers)

    // the image's occupancy bitmap starts empty (cifcaf.cpp:173); 16-byte stores, region is 256-B aligned
    {
        const int n16 = (a.F * a.occ_h * c.occ_wpr + 3) >> 2;
        uint4* z = reinterpret_cast<uint4*>(c.occ);
        for (int k = tid; k < n16; k += kThreads) z[k] = make_uint4(0u, 0u, 0u, 0u);
    }
    // list lengths and the skeleton adjacency are consulted at every step of the search: keep them in LDS
    for (int k = tid; k < E; k += kThreads) {
        c.sh_counts[k] = c.list_counts[k];
        int lo = 0, hi = K;                          // the joint whose adjacency range holds slot k
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (sk.adj_off[mid] <= k) lo = mid; else hi = mid; }
        l_info[k] = lo | (sk.adj_other[k] << 8) | (sk.adj_bone[k] << 16) | (sk.adj_fwd[k] << 24);
        l_first[k] = sk.adj_first[k];
    }
    for (int k = tid; k <= K; k += kThreads) l_off[k] = sk.adj_off[k];
    if (use_bbox) {                                  // the first kListBboxChunks boxes of every caf_th list
        const float4* src = reinterpret_cast<const float4*>(a.list_bbox) + (size_t)b * E * a.bbox_chunks;
        const float inf = __builtin_inff();
        for (int k = tid; k < E * kListBboxChunks; k += kThreads) {
            const int ch = k & (kListBboxChunks - 1);
            sh_bbox[k] = ch < a.bbox_chunks ? src[(size_t)(k / kListBboxChunks) * a.bbox_chunks + ch] : make_float4(inf, -inf, inf, -inf);
        }
        c.bbox = sh_bbox;
    }
    if (tid < NW) {
        TaskSlot t; t.state = kTaskIdle; t.cancel = 0; t.epoch = 0; t.seed = -1; t.npub = 0; t.pk = 0; t.f = 0; t.score = 0.0;
        t.t_emit = t.t_done = t.pad0 = t.pad1 = t.coll = 0;
        task[tid] = t;
    }
    if (tid < 16) sh_ctl[tid] = tid == 9 || tid == 12 ? -1 : 0;   // (9: the grower holding the head seed, 12: the head seed)
    if (tid == 10) sh_ctl[10] = (int)(private_base - smem);        // 10, 11: for publish_joint's look at the other growers' boxes
    if (tid == 11) sh_ctl[11] = (int)private_bytes | (S << 20) | ((a.collide ? 1 : 0) << 28);
    if (tid < kAssocStats) sh_stats[tid] = 0;
#ifdef OPA_ASSOC_PHASE_TIMING
    if (tid < kPhases) { g_ph[tid] = 0; g_phn[tid] = 0; }
    if (tid < 16) g_ph_last[tid] = clock64();
#endif
    for (int k = tid; k < WR * kWave; k += kThreads) { pool_if[k] = kPoolIdxMask; pool_pack[k] = 0; pool_ep[k] = 0; if (kSelfServe) pool_own[k] = 0; }
    for (int k = tid; k < NW * kWave; k += kThreads) shadow_by[k] = 0u;
    for (int k = tid; k < (1 << kDedupBits); k += kThreads) dedup[k] = ~0ull;
    const bool dedup_on = a.dedup != 0;
    constexpr bool self = kSelfServe;                // idle growers take their next candidate themselves (see kPoolSlots' neighbour above)
    c.adj_off = l_off; c.slot_info = l_info; c.adj_first = l_first;
    c.n_predicted = &sh_ctl[6];                      // (statistics slot 21: joint boxes published from predictions)

    // ---- seeds in score order, cifcaf.cpp:206-231
    int n_seeds = a.seed_count[b];
    if (n_seeds > a.seed_cap) n_seeds = a.seed_cap;
    const int n_seeds_all = n_seeds;
    const int32_t* seed_f = a.seed_f + (size_t)b * a.seed_cap;
    const float4* seed_vxys = reinterpret_cast<const float4*>(a.seed_vxys) + (size_t)b * a.seed_cap;
    const int32_t* seed_cell = a.seed_cell + (size_t)b * a.seed_cap;

    // ---- [r5] the first seed of every occupancy cell, found by the whole workgroup before the coordinator sees any of them.
    // The refill's dedupe argument (below: of the seeds of ONE cell of a field only the first can ever be free at its turn)
    // does not need the pool: it is a property of the seed list.  The coordinator walked all ~5 000 seeds of a crowded COCO
    // image through its refill (two dependent memory round trips per round of 512, eight or nine refills, 60-70 us of the ONE
    // wave everything waits for) to admit the ~900 that are the first of their cell; here twelve waves do that walk once, in a
    // few microseconds, while the LDS work area is still free: a table of (cell key << 32 | seed index) minima in it, an ordered
    // compaction of the survivors (original index | field << 24, cell) into the image's sort-key array -- the sort and the tie
    // pass are done with it -- and the coordinator scans that list instead.  A bucket shared by two cells keeps the smaller key's
    // seeds exact and admits the other's (the refill's own test against the bitmap, which stays, sees those).  Seed order is
    // preserved, the dropped seeds are dead for good: the accepted seeds and their poses are the sequential loop's.
    const int32_t* scan_f = seed_f;                  // what the coordinator's refill reads at scan position i: the seed's field ...
    const int32_t* scan_pk = seed_cell;              // ... and its cell word
    bool pre = false;
    {
        const int n_it = (n_seeds + kThreads - 1) / kThreads;
        pre = dedup_on && a.prededup && a.tie_sort.keys != nullptr && n_seeds > kPreDedupMin && n_it * NW < 2 * kSeedStage;
        if (pre) {
            int32_t* kept_if = reinterpret_cast<int32_t*>(a.tie_sort.keys + (size_t)b * a.tie_sort.sort_cap);
            int32_t* kept_pk = kept_if + a.tie_sort.sort_cap;
            const size_t area = (size_t)S * private_bytes;
            int tb = 31 - __clz((int)(area >> 3));
            if (tb > 14) tb = 14;
            unsigned long long* tbl = reinterpret_cast<unsigned long long*>(work_base);
            int* cnt = stage_f;                      // (stage_f and stage_pk: 2 * kSeedStage words, empty until the coordinator starts)
            for (int k = tid; k < (1 << tb); k += kThreads) tbl[k] = ~0ull;
            __syncthreads();
            for (int i = tid; i < n_seeds; i += kThreads) {
                const unsigned key = ((unsigned)seed_f[i] << 24) | ((unsigned)seed_cell[i] & 0xFFFFFFu);
                __hip_atomic_fetch_min(&tbl[(key * 2654435761u) >> (32 - tb)], ((unsigned long long)key << 32) | (unsigned)i,
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            __syncthreads();
            auto first_of_cell = [&](int i, int* f, int* pk) -> bool {
                if (i >= n_seeds) return false;
                *f = seed_f[i]; *pk = seed_cell[i];
                const unsigned key = ((unsigned)*f << 24) | ((unsigned)*pk & 0xFFFFFFu);
                const unsigned long long v = tbl[(key * 2654435761u) >> (32 - tb)];
                return !((unsigned)(v >> 32) == key && (unsigned)v != (unsigned)i);
            };
            for (int it = 0; it < n_it; it++) {
                int f, pk;
                const unsigned long long m = __ballot(first_of_cell(it * kThreads + tid, &f, &pk));
                if (lane == 0) cnt[it * NW + wave] = __popcll(m);
            }
            __syncthreads();
            if (wave == 0) {                         // exclusive prefix over the (iteration, wave) counts, in seed order
                const int n_e = n_it * NW, per = (n_e + kWave - 1) / kWave;
                int sum = 0;
                for (int k = 0; k < per; k++) { const int e = lane * per + k; if (e < n_e) sum += cnt[e]; }
                int incl = sum;
                for (int d = 1; d < kWave; d <<= 1) { const int t = __shfl_up(incl, d); if (lane >= d) incl += t; }
                int run = incl - sum;
                for (int k = 0; k < per; k++) { const int e = lane * per + k; if (e < n_e) { const int t = cnt[e]; cnt[e] = run; run += t; } }
                if (lane == kWave - 1) cnt[2 * kSeedStage - 1] = incl;
            }
            __syncthreads();
            for (int it = 0; it < n_it; it++) {
                int f = 0, pk = 0;
                const int i = it * kThreads + tid;
                const bool keep = first_of_cell(i, &f, &pk);
                const unsigned long long m = __ballot(keep);
                if (keep) {
                    const int pos = cnt[it * NW + wave] + prefix_count(m);
                    kept_if[pos] = i | (f << 24); kept_pk[pos] = pk;
                }
            }
            n_seeds = cnt[2 * kSeedStage - 1];
            scan_f = kept_if; scan_pk = kept_pk;
        }
    }
    sync_global();                                   // bitmap zeros are in memory before anyone marks or tests (and the compacted seed list)
    RegSkeleton rs; rs.slot_info = 0; rs.slot_first = 0; rs.off = 0; rs.off1 = 0;
    if constexpr (REG) {                             // lane t: directed bone t; lane j: adjacency range of joint j
        if (lane < E) { rs.slot_info = l_info[lane]; rs.slot_first = l_first[lane]; }
        if (lane < K) { rs.off = l_off[lane]; rs.off1 = l_off[lane + 1]; }
    }

    double* anns = a.anns + (size_t)b * a.max_ann * K * 4;
    int64_t* ann_ids = a.ann_ids + (size_t)b * a.max_ann;
    int n_kept = 0, n_dropped = 0;                   // coordinator's counters until the growth phase ends
    const bool prune = !p.force_complete;     // a pose scoring below the instance threshold before NMS cannot survive it

    // Coordinator: accept the pose in private block `blk`: mark its joints in the bitmap and, unless it
    // cannot survive NMS, store it at slot n_kept.
    auto accept_pose = [&](int blk, double score, long long id) {
        const PoseView q = pose_of_block(private_base, blk, private_bytes, K);
        occ_mark_pose(c, q);
        if (prune && score < p.nms_instance_threshold) return;
        if (n_kept >= a.max_ann) { n_dropped++; return; }
        double* dst = anns + (size_t)n_kept * K * 4;
        for (int k = lane; k < K; k += kWave) {
            dst[4 * k + 0] = q.v[k]; dst[4 * k + 1] = (double)q.x[k];
            dst[4 * k + 2] = (double)q.y[k]; dst[4 * k + 3] = (double)q.s[k];
        }
        if (lane == 0) ann_ids[n_kept] = id;
        n_kept++;
    };

    // ---- initial annotations (tracking API), cifcaf.cpp:177-202: S growths at a time, all of them accepted
    for (int n0 = 0; n0 < a.n_initial; n0 += S) {
        const int n = n0 + wave - 1;
        if (wave >= 1 && wave <= S && n < a.n_initial) {
            const float* src = a.initial + ((size_t)b * a.n_initial + n) * K * 4;
            for (int k = lane; k < K; k += kWave) {
                c.jv[k] = (double)src[4 * k + 0]; c.jx[k] = src[4 * k + 1];
                c.jy[k] = src[4 * k + 2]; c.js[k] = src[4 * k + 3];
            }
            wave_sync();
            grow_pose<REG>(c, p, rs, true, 1.0, false);
            pose_boxes(c, p);
            wave_sync();
            const double sc = pose_score(c.jv, K);
            if (lane == 0) task[wave].score = sc;
        }
        __syncthreads();
        if (wave == 0)
            for (int g = 0; g < S && n0 + g < a.n_initial; g++)
                accept_pose(g, task[g + 1].score, a.initial_ids ? a.initial_ids[(size_t)b * a.n_initial + n0 + g] : -1);
        __syncthreads();
    }

    if (wave == 0) {
        // the always-on statistics live in LDS (lane 0 adds, no return value): two dozen scalar counters carried through this
        // loop spill scalar registers into vector lanes, and the growers' scans pay for that
        auto stat = [&](int k, int v) {
            if (lane == 0) __hip_atomic_fetch_add(&sh_stats[k], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        };
        int n_commits = 0;
        if (pre) stat(23, n_seeds_all - n_seeds);        // later seeds of a cell, dropped before the pool saw them
        // the per-phase tick counters (slots 12, 17-20) cost a clock read and a wait each: only when asked for (OPA_ASSOC_TIMING=1,
        // tools/gpu/r3_probe.py); the event counters are always on
        const bool timing = a.timing != 0;
        auto tick = [&]() -> long long { return timing ? wall_clock64() : 0ll; };
        // ================================================================= coordinator
        // The pool: up to WR * 64 LIVE, undecided seeds, WR slots per lane in any order (mirrored in LDS for the
        // growers).  Invariant: every seed below scan_pos is either in a slot or dead for good (inside a box of
        // an accepted pose), so each seed is fetched and tested against the bitmap exactly once; afterwards it
        // is tested against every newly accepted pose by box containment.
        constexpr int kIdxMask = kPoolIdxMask;
        constexpr unsigned kNone = 0xFFFFFFFFu;
        int s_pack[WR], s_if[WR];                        // cell x | cell y << 12 | box half-width << 24 ;  seed index | field << 24
        unsigned occupied = 0u;                          // bit r: slot r holds a live undecided seed
        unsigned emitted = 0u;                           //        ... handed to a grower (nibble r of gmap says which)
        unsigned ever = 0u;                              //        ... was shadowed at some time (statistics)
        unsigned gmap[(WR + 7) / 8];                     // nibble r: the grower slot r was handed to
#pragma unroll
        for (int k = 0; k < (WR + 7) / 8; k++) gmap[k] = 0u;
        constexpr int HR = WR < 8 ? WR : 8;      // slots per lane one refill round fills (the round's arrays live in registers)
        auto gm_get = [&](int r) -> int { return (int)((gmap[r >> 3] >> (4 * (r & 7))) & 15u); };
        auto gm_set = [&](int r, int g) {
#pragma unroll
            for (int k = 0; k < (WR + 7) / 8; k++)
                if ((r >> 3) == k) gmap[k] = (gmap[k] & ~(15u << (4 * (r & 7)))) | ((unsigned)g << (4 * (r & 7)));
        };
        int scan_pos = 0, n_live = 0;
        // Lookahead (large skeletons, with the compacted seed list): when growers are idle and every pooled seed is somebody's,
        // predicted dead or handed out, the list is searched AHEAD of the scan position for the next seed that is free in the
        // bitmap and outside every box the candidates in flight have published or predicted -- the first seed of the next
        // person, whose thousand seeds otherwise reach the window only when the people before are committed -- and that ONE
        // seed enters the pool early (its list entry is marked: the scan skips it when it gets there).  It is grown like any
        // other candidate and committed at its turn: `bound`, the index of the first seed the scan has not reached, stops a
        // commit of anything beyond it.
        const bool la_on = !REG && pre && a.lookahead != 0 && a.F < 255;
        int la_pos = 0;                                  // lookahead cursor (>= scan_pos)
        bool want_la = false;
        int la_snap = -1, la_restarts = 0;               // lane g: the seed grower g's candidate had while the current pass over the list ran (-1: none)
        unsigned bound = 0u;                             // seed index of list position scan_pos (all ones: the scan is through); set by the refill
        int32_t* list_w = const_cast<int32_t*>(scan_f);  // (the compacted list is this kernel's own: lookahead marks entries in it)
        bool watchdog = false, marks_pending = false;
        int last_hg = -1;                                // what sh_ctl[9] says
        int epoch = 0;                                   // refills so far; `unver`: slots filled by refills not every candidate in flight has tested yet
        unsigned unver = 0u;
        long long wait_ticks = 0;
        unsigned iter = 0;
        const bool is_grower_lane = lane >= 1 && lane <= S;
        __builtin_amdgcn_s_setprio(3);                   // everything sequential runs here: issue ahead of the growers
#pragma unroll
        for (int r = 0; r < WR; r++) { s_pack[r] = 0; s_if[r] = kIdxMask; }

        // Seeds enter the pool through two dependent memory round trips (their field and cell, then the bitmap
        // word of that cell).  The first is taken off the refill: the (field, cell) of the next kSeedStage
        // seeds travel HBM/L2 -> LDS directly, in 64-seed blocks, issued after a refill and landed by the next.
        int pf_end = 0;                                  // seeds below it are staged (multiple of 64)
        auto stage_seeds = [&]() {
            int limit = (scan_pos & ~(kWave - 1)) + kSeedStage;
            const int n_round = (n_seeds + kWave - 1) & ~(kWave - 1);
            if (limit > n_round) limit = n_round;
#pragma unroll
            for (int r = 0; r < 8; r++)
                if (pf_end < limit) {
                    const int idx = pf_end + lane, ii = idx < n_seeds ? idx : 0;
                    __builtin_amdgcn_global_load_lds((gint*)scan_f + ii, (lint*)stage_f + (pf_end & (kSeedStage - 1)), 4, 0, 0);
                    __builtin_amdgcn_global_load_lds((gint*)scan_pk + ii, (lint*)stage_pk + (pf_end & (kSeedStage - 1)), 4, 0, 0);
                    pf_end += kWave;
                }
        };
        stage_seeds(); stage_seeds();
        auto count_live = [&]() {
            n_live = 0;
#pragma unroll
            for (int r = 0; r < WR; r++) n_live += __popcll(__ballot((occupied >> r) & 1u));
        };

        // the head = the smallest-index live seed (everything before it is decided), and the grower that has it
        unsigned hd = kNone;
        auto find_head = [&]() {
            unsigned m = kNone;
#pragma unroll
            for (int r = 0; r < WR; r++)
                if ((occupied >> r) & 1u) m = min(m, (unsigned)(s_if[r] & kIdxMask));
            const unsigned was = hd;
            hd = ~wave_max_u32(~m);
            if (self && hd != was && lane == 0) __hip_atomic_store(&sh_ctl[12], (int)hd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        };
        // self-serve: who grows what is read from the slots' owner words (the growers claim them), once per round
        auto read_owners = [&]() {
            emitted = 0u;
#pragma unroll
            for (int k = 0; k < (WR + 7) / 8; k++) gmap[k] = 0u;
#pragma unroll
            for (int r = 0; r < WR; r++) {
                const int o = flag_peek(&pool_own[r * kWave + lane]);
                if (o != 0 && (occupied >> r) & 1u) { emitted |= 1u << r; gm_set(r, o); }
            }
        };
        auto head_grower = [&]() -> int {
            int mine = -1;
#pragma unroll
            for (int r = 0; r < WR; r++)
                if (((occupied & emitted) >> r) & 1u && (unsigned)(s_if[r] & kIdxMask) == hd) mine = gm_get(r);
            const unsigned long long m = __ballot(mine >= 0);
            return m ? rlane(mine, __builtin_ctzll(m)) : -1;
        };
        find_head();
        // the new occupants of pool slots: mirror them, forget what the growers said about the slots' former occupants, and
        // have every candidate in flight test them against the boxes it has published (pool_catch_up)
        auto publish_fresh = [&](unsigned fresh) {
            epoch++;
#pragma unroll
            for (int r = 0; r < WR; r++)
                if ((fresh >> r) & 1u) {
                    pool_pack[r * kWave + lane] = s_pack[r]; pool_ep[r * kWave + lane] = epoch;
                    __hip_atomic_store(&pool_if[r * kWave + lane], s_if[r], __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);   // (the claimers read this word first)
                }
            if (fresh)
                for (int g = 1; g <= S; g++) atomicAnd(&shadow_by[g * kWave + lane], ~fresh);
            unver |= fresh;
            wave_sync();
            if (is_grower_lane) flag_store(&task[lane].epoch, epoch);
        };

        for (;;) {
            const long long t_iter = wall_clock64();
            if (t_iter - t_kernel > kWatchdogTicks) { watchdog = true; break; }
            iter++;
            bool progress = false;                       // this iteration committed, refilled or handed out something

            // ---- 1. commit the head while its growth is done (:213-230): every commit of a run costs the commit alone,
            //         not a round of the whole loop (a crowded image ends in dozens of poses of one or two joints)
            if (self) read_owners();
            int hg = hd == kNone ? -1 : head_grower();
            for (int run = 0; run < kCommitRun && hg >= 0 && flag_load(&task[hg].state) == kTaskDone &&
                              (!self || task[hg].seed == (int)hd) && (!la_on || hd < bound); run++) {
                const long long t_cm = tick();
                const PoseView q = pose_of_block(private_base, hg - 1, private_bytes, K);
                unsigned dead = 0u;                      // pooled seeds inside one of its joint boxes (:211 for them)
                constexpr int CG = WR < 4 ? WR : 4;
#pragma unroll
                for (int r0 = 0; r0 < WR; r0 += CG) {    // four boxes per LDS round trip
                    OccBox bb[CG];
#pragma unroll
                    for (int r = 0; r < CG; r++) bb[r] = q.box[(occupied >> (r0 + r)) & 1u ? (unsigned)s_if[r0 + r] >> 24 : 0u];
#pragma unroll
                    for (int r = 0; r < CG; r++)
                        if ((occupied >> (r0 + r)) & 1u &&
                            (box_contains(bb[r], s_pack[r0 + r] & 0xfff, (s_pack[r0 + r] >> 12) & 0xfff) ||
                             (unsigned)(s_if[r0 + r] & kIdxMask) == hd))
                            dead |= 1u << (r0 + r);
                }
                if (!self) {
                if (__ballot((dead & emitted) != 0u) != 0ull) {   // growths of seeds that just died: drop finished ones, stop running ones
                    int n_drop = 0, n_stop = 0;
#pragma unroll
                    for (int r = 0; r < WR; r++)
                        if (((dead & emitted) >> r) & 1u && (unsigned)(s_if[r] & kIdxMask) != hd) {
                            const int g = gm_get(r);
                            if (flag_load(&task[g].state) == kTaskDone) { flag_store(&task[g].state, kTaskIdle); n_drop++; }
                            else { flag_store(&task[g].cancel, 1); n_stop++; }
                        }
                    if (__ballot(n_drop + n_stop > 0) != 0ull) {
#pragma unroll
                        for (int k = 0; k < WR; k++) { stat(3, __popcll(__ballot(n_drop > k))); stat(2, __popcll(__ballot(n_stop > k))); }
                    }
                }
#pragma unroll
                for (int r = 0; r < WR; r++)
                    if ((dead >> r) & 1u) {
                        occupied &= ~(1u << r); emitted &= ~(1u << r); s_if[r] |= kIdxMask;
                        pool_if[r * kWave + lane] = s_if[r];
                    }
                } else {
                    // the slot is emptied FIRST, then its owner word is taken: a grower that claims the slot in between finds it
                    // empty when it looks again (try_claim), one that claimed it before is found here and stopped
                    int n_drop = 0, n_stop = 0;
#pragma unroll
                    for (int r = 0; r < WR; r++)
                        if ((dead >> r) & 1u) {
                            const int idx = s_if[r] & kIdxMask;
                            occupied &= ~(1u << r); emitted &= ~(1u << r); s_if[r] |= kIdxMask;
                            __hip_atomic_store(&pool_if[r * kWave + lane], s_if[r], __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                            const int g = __hip_atomic_exchange(&pool_own[r * kWave + lane], 0, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
                            if (g != 0 && (unsigned)idx != hd) {
                                // (a claim whose task is not set up yet -- the seed word is still the last task's -- finds the flag when it starts)
                                if (flag_load(&task[g].state) == kTaskDone && task[g].seed == idx) { flag_store(&task[g].state, kTaskIdle); n_drop++; }
                                else { flag_store(&task[g].cancel, 1); n_stop++; }
                            }
                        }
                    if (__ballot(n_drop + n_stop > 0) != 0ull) {
#pragma unroll
                        for (int k = 0; k < WR; k++) { stat(3, __popcll(__ballot(n_drop > k))); stat(2, __popcll(__ballot(n_stop > k))); }
                    }
                }
                if (a.trace && n_commits < kAssocTrace && lane == 0) {
                    int* tr = a.trace + ((size_t)b * kAssocTrace + n_commits) * 4;
                    tr[0] = (int)(wall_clock64() - t_kernel); tr[1] = task[hg].t_emit; tr[2] = task[hg].t_done;
                    tr[3] = (int)hd | (hg << 24);
                }
                {   // accepted: its grower marks the bitmap and stores the pose (cifcaf.cpp:225-230)
                    const double score = task[hg].score;
                    int slot = -1;
                    if (!(prune && score < p.nms_instance_threshold)) {
                        if (n_kept >= a.max_ann) n_dropped++;
                        else slot = n_kept++;
                    }
                    if (lane == 0) task[hg].pad0 = slot;
                    marks_pending = true;
                }
                n_commits++;
                stat(1, 1);
                wave_sync();                             // every lane has read block hg-1
                if (lane == 0) flag_store(&task[hg].state, kTaskAccepted);
                progress = true;
                find_head();
                hg = hd == kNone ? -1 : head_grower();
                if (timing) stat(17, (int)(wall_clock64() - t_cm));
            }
            if (progress) count_live();

            // ---- 2. the growers (lane g looks at grower g): state, cancel flag, seed, published boxes
            int g_state = -1, g_cancel = 0, g_seed = -1, g_pub = 0, g_ack = 0, g_pk = 0, g_f = -1, g_coll = 0;
            if (is_grower_lane) {
                g_state = flag_load(&task[lane].state);
                g_ack = flag_peek(&task[lane].pad1);
                g_cancel = flag_peek(&task[lane].cancel); g_seed = task[lane].seed; g_pub = flag_peek(&task[lane].npub);
                g_pk = task[lane].pk; g_f = task[lane].f; g_coll = flag_peek(&task[lane].coll);
                if (g_state == kTaskDone && g_cancel) {  // a growth that finished after its seed died: drop the result
                    flag_store(&task[lane].state, kTaskIdle); g_state = kTaskIdle;
                }
            }
            const bool g_live = (g_state == kTaskAssigned || g_state == kTaskDone) && !g_cancel;
            const unsigned long long live_mask = __ballot(g_live);
            if (!self && __ballot(unver != 0u) != 0ull && __ballot(g_live && g_ack != epoch) == 0ull) unver = 0u;   // everyone has tested the newcomers

            // ---- 3. refill free slots with the next seeds that are still free in the bitmap (:211 for the
            //         poses accepted so far); slot (r, lane) takes the seed of its rank among the free slots
            const bool head_early = la_on && hd != kNone && hd >= bound;     // the scan has to reach the head before it can be committed
            if (scan_pos < n_seeds && (kRefillDen * n_live < kRefillNum * WR * kWave || head_early)) {
                const long long t_ph = tick();
                if (marks_pending) {                     // accepted poses are marked by their growers: all of them are done
                    while (__ballot(is_grower_lane && flag_load(&task[lane].state) == kTaskAccepted) != 0ull &&
                           wall_clock64() - t_kernel <= kWatchdogTicks)
                        __builtin_amdgcn_s_sleep(1);
                    marks_pending = false;
                    if (timing) stat(20, (int)(wall_clock64() - t_ph));
                }
                // this wave's marks (atomics, performed at L2) before its own reads, which bypass the L1 (agent-scope
                // loads): vmcnt(0) is all it takes -- and the staged seeds have landed in LDS
                __builtin_amdgcn_s_waitcnt(0x0F70);
                wave_sync();
                unsigned fresh = 0u;                     // slots filled by this refill
                bool first_round = true;
                int r0 = 0, dry = 0;                     // a round fills slots [r0, r0 + HR) of every lane; rounds without a free slot in a row
                while (scan_pos < n_seeds) {
                    if (!first_round) __builtin_amdgcn_s_waitcnt(0x0F70);   // the cells marked by the round before are at the L2
                    first_round = false;
                    int nidx[HR], base = 0;
#pragma unroll
                    for (int r = 0; r < HR; r++) {
                        const bool fr = !((occupied >> (r0 + r)) & 1u);
                        const unsigned long long m = __ballot(fr);
                        nidx[r] = fr ? scan_pos + base + prefix_count(m) : n_seeds;
                        base += __popcll(m);
                    }
                    if (base == 0) {
                        if (++dry >= WR / HR) break;
                        if constexpr (WR > HR) r0 = r0 + HR < WR ? r0 + HR : 0;
                        continue;
                    }
                    dry = 0;
                    // Straight-line code on clamped indices: the eight bitmap words of a lane are ONE memory round
                    // trip (loads inside `if`s are waited for one by one).
                    int ff[HR], pk[HR]; unsigned ow[HR];
                    bool beyond = false;                 // a seed beyond the staged window (second round of a refill)
#pragma unroll
                    for (int r = 0; r < HR; r++) beyond |= nidx[r] >= pf_end && nidx[r] < n_seeds;
                    if (__ballot(beyond) != 0ull) {
#pragma unroll
                        for (int r = 0; r < HR; r++) {
                            const int ii = nidx[r] < n_seeds ? nidx[r] : 0;
                            ff[r] = __hip_atomic_load(&scan_f[ii], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); pk[r] = scan_pk[ii];
                        }
#pragma unroll
                        for (int r = 0; r < HR; r++) asm volatile("" : "+v"(ff[r]), "+v"(pk[r]) :: "memory");
                    } else {
#pragma unroll
                        for (int r = 0; r < HR; r++) { ff[r] = stage_f[nidx[r] & (kSeedStage - 1)]; pk[r] = stage_pk[nidx[r] & (kSeedStage - 1)]; }
                    }
                    // sw: the slot word of the seed (its index in the image's seed list | field << 24) -- what the compacted list holds
                    int sw[HR];
#pragma unroll
                    for (int r = 0; r < HR; r++) {
                        sw[r] = pre ? ff[r] : (nidx[r] | (ff[r] << 24));
                        ff[r] = (int)((unsigned)sw[r] >> 24);
                    }
#pragma unroll
                    for (int r = 0; r < HR; r++) {
                        const bool valid = nidx[r] < n_seeds && ff[r] != 0xff;     // (field 255: an entry the lookahead took)
                        const size_t word = valid ? ((size_t)ff[r] * c.occ_h + ((pk[r] >> 12) & 0xfff)) * c.occ_wpr + ((pk[r] & 0xfff) >> 5) : 0;
                        ow[r] = __hip_atomic_load(&c.occ[word], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
#pragma unroll
                    for (int r = 0; r < HR; r++) asm volatile("" : "+v"(ow[r]) :: "memory");
                    // Of the seeds of ONE occupancy cell of a field only the first can ever be free at its turn: if it is,
                    // its pose is accepted with the seed as joint f and the box of that joint contains the seed's own cell
                    // (occupancy.cpp:13-29: sigma >= 2 cells around it); if it is not, the box that covers its cell covers
                    // the same cell of the later seeds.  So a later seed of a cell already seen is dead for good, whatever
                    // happens to the first -- and a confidence blob's cells all regress to the same point, i.e. mostly the
                    // same cell.  Each admitted seed therefore marks its own cell in the bitmap (the next refills test
                    // against it), and inside one refill round the first seed of a cell is found through a small LDS
                    // table (64-bit min of cell key << 32 | seed index; a bucket taken by another key just admits).
                    bool cand[HR]; unsigned key[HR]; int bkt[HR];
#pragma unroll
                    for (int r = 0; r < HR; r++) {
                        cand[r] = nidx[r] < n_seeds && !((ow[r] >> (pk[r] & 31)) & 1u) && (!la_on || ff[r] != 0xff);   // (field 255: taken early by the lookahead)
                        key[r] = ((unsigned)ff[r] << 24) | ((unsigned)pk[r] & 0xFFFFFFu);
                        bkt[r] = (int)((key[r] * 2654435761u) >> (32 - kDedupBits));
                        if (cand[r] && dedup_on)
                            __hip_atomic_fetch_min(&dedup[bkt[r]], ((unsigned long long)key[r] << 32) | (unsigned)nidx[r],
                                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                    wave_sync();
                    unsigned long long seen[HR];
#pragma unroll
                    for (int r = 0; r < HR; r++) seen[r] = dedup[bkt[r]];
#pragma unroll
                    for (int r = 0; r < HR; r++) asm volatile("" : "+v"(seen[r]) :: "memory");
                    wave_sync();                         // every lane has read its buckets: give them back
#pragma unroll
                    for (int r = 0; r < HR; r++) if (cand[r] && dedup_on) dedup[bkt[r]] = ~0ull;
                    int n_dup = 0;
#pragma unroll
                    for (int r = 0; r < HR; r++) {
                        const bool later = dedup_on && (unsigned)(seen[r] >> 32) == key[r] && (unsigned)seen[r] != (unsigned)nidx[r];
                        if (cand[r] && !later) {
                            const unsigned sb = 1u << (r0 + r);
#pragma unroll
                            for (int q = 0; q < WR; q += HR)     // (static register indices: the slot is r0 + r)
                                if (q == r0) { s_pack[q + r] = pk[r]; s_if[q + r] = sw[r]; }
                            occupied |= sb; emitted &= ~sb; ever &= ~sb; fresh |= sb;
                            if (dedup_on && !la_on) {       // (with the lookahead a seed may enter before earlier ones of its cell: no admission marks)
                                const size_t word = ((size_t)ff[r] * c.occ_h + ((pk[r] >> 12) & 0xfff)) * c.occ_wpr + ((pk[r] & 0xfff) >> 5);
                                atomicOr(&c.occ[word], 1u << (pk[r] & 31));
                            }
                        }
                        n_dup += cand[r] && later ? 1 : 0;
                    }
                    if (__ballot(n_dup > 0) != 0ull) {
#pragma unroll
                        for (int k = 0; k < HR; k++) stat(23, __popcll(__ballot(n_dup > k)));
                    }
                    scan_pos = scan_pos + base < n_seeds ? scan_pos + base : n_seeds;
                    count_live();
                    bool need_more = false;              // the smallest pooled seed still lies beyond the scan (it entered early): go on
                    if (la_on) {
                        find_head();                     // (what this round admitted comes BEFORE an early seed)
                        bound = scan_pos < n_seeds ? (unsigned)__hip_atomic_load(&scan_f[scan_pos], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & (unsigned)kIdxMask : kNone;
                        need_more = hd != kNone && hd >= bound;
                    }
                    if (kRefillDen * n_live >= kRefillNum * WR * kWave && !need_more) break;
                    if constexpr (WR > HR) r0 = r0 + HR < WR ? r0 + HR : 0;
                }
                if (la_on) {
                    find_head();
                    bound = scan_pos < n_seeds ? (unsigned)__hip_atomic_load(&scan_f[scan_pos], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & (unsigned)kIdxMask : kNone;
                    if (la_pos < scan_pos) la_pos = scan_pos;
                    hg = -1;
                }
                stat(6, 1);
                stage_seeds();
                publish_fresh(fresh);
                if (timing) stat(18, (int)(wall_clock64() - t_ph));
                progress = true;
                if (hd == kNone) { find_head(); hg = -1; }   // (newcomers come after everything pooled)
            }

            // ---- 3b. lookahead (see la_on): ONE seed ahead of the scan, the first that is free and outside every box in flight
            if (la_on && want_la) {
                want_la = false;
                if (la_pos < scan_pos) la_pos = scan_pos;
                // A pass skips what lies in a box of a candidate in flight.  When such a candidate is gone -- stopped as a duplicate,
                // committed -- what only IT covered is free: the pass starts again behind the scan (entries taken early are marked).
                if (la_pos >= n_seeds && la_restarts < 256 &&
                    __ballot(la_snap >= 0 && !(g_live && g_seed == la_snap)) != 0ull) {
                    la_pos = scan_pos; la_snap = -1; la_restarts++;
                }
                if (g_live && la_snap < 0) la_snap = g_seed;
                // (every candidate in flight has said where it expects its joints: count 2, or it is done)
                // ... and what entered the pool last has been looked at by everybody (it is handed out first: its candidate's
                // boxes are what the next step must see -- else the step takes the same person's next seed, and the next ...)
                const bool settled = __ballot(g_live && g_state == kTaskAssigned && g_pub < 2) == 0ull && __ballot(unver != 0u) == 0ull;
                if (la_pos < n_seeds && settled) {
                    if (marks_pending) {
                        while (__ballot(is_grower_lane && flag_load(&task[lane].state) == kTaskAccepted) != 0ull &&
                               wall_clock64() - t_kernel <= kWatchdogTicks)
                            __builtin_amdgcn_s_sleep(1);
                        marks_pending = false;
                    }
                    __builtin_amdgcn_s_waitcnt(0x0F70);
                    constexpr int LR = 8;                // 512 list positions per step
                    int q[LR], sifq[LR], pkq[LR]; unsigned owq[LR];
#pragma unroll
                    for (int r = 0; r < LR; r++) {
                        q[r] = la_pos + r * kWave + lane;
                        const int ii = q[r] < n_seeds ? q[r] : 0;
                        sifq[r] = __hip_atomic_load(&scan_f[ii], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); pkq[r] = scan_pk[ii];
                    }
#pragma unroll
                    for (int r = 0; r < LR; r++) asm volatile("" : "+v"(sifq[r]), "+v"(pkq[r]) :: "memory");
                    bool candq[LR];
#pragma unroll
                    for (int r = 0; r < LR; r++) {
                        const int f = (int)((unsigned)sifq[r] >> 24);
                        candq[r] = q[r] < n_seeds && f < c.F;                        // (255: taken early before)
                        const size_t word = candq[r] ? ((size_t)f * c.occ_h + ((pkq[r] >> 12) & 0xfff)) * c.occ_wpr + ((pkq[r] & 0xfff) >> 5) : 0;
                        owq[r] = __hip_atomic_load(&c.occ[word], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
#pragma unroll
                    for (int r = 0; r < LR; r++) asm volatile("" : "+v"(owq[r]) :: "memory");
#pragma unroll
                    for (int r = 0; r < LR; r++) candq[r] = candq[r] && !((owq[r] >> (pkq[r] & 31)) & 1u);
                    for (int g = 1; g <= S; g++)
                        if ((live_mask >> g) & 1ull) {
                            const OccBox* jb = pose_of_block(private_base, g - 1, private_bytes, K).box;
#pragma unroll
                            for (int r = 0; r < LR; r++)
                                if (candq[r] && box_contains(jb[(unsigned)sifq[r] >> 24], pkq[r] & 0xfff, (pkq[r] >> 12) & 0xfff)) candq[r] = false;
                        }
                    unsigned lmin = kNone; int vs = 0, vp = 0;
#pragma unroll
                    for (int r = LR - 1; r >= 0; r--)
                        if (candq[r]) { lmin = (unsigned)q[r]; vs = sifq[r]; vp = pkq[r]; }   // (descending r: the smallest position of the lane stays)
                    const unsigned pstar = ~wave_max_u32(~lmin);
                    if (pstar == kNone) {
                        la_pos = la_pos + LR * kWave < n_seeds ? la_pos + LR * kWave : n_seeds;
                        want_la = true;                  // (nothing in these 512: go on with the next step at once)
                        progress = true;
                    } else {
                        const int owner = __builtin_ctzll(__ballot(lmin == pstar));
                        const int sif_star = rlane(vs, owner), pk_star = rlane(vp, owner);
                        int fr_r = -1, fr_lane = 0;
#pragma unroll
                        for (int r = 0; r < WR; r++) {
                            const unsigned long long m = __ballot(!((occupied >> r) & 1u));
                            if (fr_r < 0 && m) { fr_r = r; fr_lane = __builtin_ctzll(m); }
                        }
                        if (fr_r >= 0) {
                            unsigned fresh = 0u;
#pragma unroll
                            for (int r = 0; r < WR; r++)
                                if (r == fr_r && lane == fr_lane) {
                                    s_if[r] = sif_star; s_pack[r] = pk_star;
                                    occupied |= 1u << r; emitted &= ~(1u << r); ever &= ~(1u << r); fresh |= 1u << r;
                                }
                            // the list entry says so (field 255, the index stays: `bound` reads it): the scan skips it when it gets there
                            if (lane == 0) {
                                const int marked = (sif_star & kIdxMask) | (0xff << 24);
                                __hip_atomic_store(&list_w[pstar], marked, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                if ((int)pstar < pf_end && (int)pstar >= pf_end - kSeedStage) stage_f[pstar & (kSeedStage - 1)] = marked;
                                __hip_atomic_fetch_add(&sh_ctl[7], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // (statistics slot 22)
                            }
                            la_pos = (int)pstar + 1;
                            count_live();
                            publish_fresh(fresh);
                            find_head(); hg = -1;
                            progress = true;
                        }
                    }
                }
            }

            // ---- 4. Which pooled seeds lie in a joint box an EARLIER live candidate has published so far?  (The
            // growers test the mirrored pool against every box they publish.)  Such a seed dies if that candidate
            // is accepted: it is not handed out, and if it is being grown the growth is stopped and the seed waits
            // -- a prediction; should the candidate die instead, the seed is handed out again.
            unsigned shadow = 0u;
            {
                {
                    unsigned w[NW];
#pragma unroll
                    for (int g = 1; g < NW; g++) w[g] = shadow_by[g * kWave + lane];
#pragma unroll
                    for (int g = 1; g < NW; g++) shadow |= (live_mask >> g) & 1ull ? w[g] : 0u;
                }
                shadow &= occupied;
                ever |= shadow;
                const unsigned pc = shadow & emitted;    // growths of shadowed seeds stop; the seeds stay pooled
                if (__ballot(pc != 0u) != 0ull) {
                    int n_pc = 0, d_sel = 0, h_sel = 0;
#pragma unroll
                    for (int r = 0; r < WR; r++)
                        if ((pc >> r) & 1u) {
                            const int g = gm_get(r);
                            if (flag_load(&task[g].state) == kTaskAssigned &&
                                (!self || (task[g].seed == (s_if[r] & kIdxMask) && !flag_peek(&task[g].cancel)))) {
                                flag_store(&task[g].cancel, 1);
                                if (!self) emitted &= ~(1u << r);   // (self-serve: the grower gives the slot back when it has stopped)
                                n_pc++;
                                d_sel = g;               // the stopped growth, and the (first) live candidate whose box holds its seed
                                for (int h = NW - 1; h >= 1; h--)
                                    if ((live_mask >> h) & 1ull && (shadow_by[h * kWave + lane] >> r) & 1u) h_sel = h;
                            }
                        }
#pragma unroll
                    for (int k = 0; k < WR; k++) stat(4, __popcll(__ballot(n_pc > k)));
                    // The stopped growth D was growing the person candidate H is growing (its seed lies in a box of H): the
                    // boxes D has published are, most likely, boxes H will publish.  H inherits D's predictions -- otherwise
                    // they lapse with D, and the seeds they covered are handed out as the next duplicates of the same person.
                    // (Advisory like every prediction: H's commit re-tests every seed against H's final boxes.)
                    if (a.inherit) {
                        unsigned long long m = __ballot(n_pc > 0 && h_sel > 0 && h_sel != d_sel);
                        while (m) {
                            const int l = __builtin_ctzll(m);
                            m &= m - 1;
                            const int d = rlane(d_sel, l), h = rlane(h_sel, l);
                            const unsigned v = shadow_by[d * kWave + lane] & occupied;
                            if (v) atomicOr(&shadow_by[h * kWave + lane], v);
                            shadow |= v;
                        }
                        ever |= shadow;
                    }
                }
            }

            // ---- 4b. growths that ran into a joint box of an earlier live candidate (their grower says so: publish_joint):
            //          stopped like the growths of shadowed seeds; the seed stays pooled, predicted dead with that candidate,
            //          which also inherits what the stopped growth had predicted
            {
                unsigned long long cm = __ballot(g_live && g_state == kTaskAssigned && g_coll > 0 && g_coll != lane);
                while (cm) {
                    const int d = __builtin_ctzll(cm);
                    cm &= cm - 1;
                    const int h = rlane(g_coll, d);
                    if (!((live_mask >> h) & 1ull) || rlane(g_seed, h) >= rlane(g_seed, d)) continue;   // (that candidate is gone)
                    unsigned bit = 0u;
#pragma unroll
                    for (int r = 0; r < WR; r++)
                        if (((occupied & emitted) >> r) & 1u && gm_get(r) == d &&
                            (s_if[r] & kIdxMask) == rlane(g_seed, d)) bit |= 1u << r;
                    if (__ballot(bit != 0u) == 0ull) continue;
                    if (self && rlane(g_cancel, d)) continue;   // (told already)
                    if (lane == 0) flag_store(&task[d].cancel, 1);
                    if (!self) emitted &= ~bit;
                    const unsigned v = (shadow_by[d * kWave + lane] & occupied) | bit;
                    if (v) atomicOr(&shadow_by[h * kWave + lane], v);
                    shadow |= v; ever |= v;
                    stat(4, 1);
                }
            }

            // ---- 5. hand the next candidates, in seed order, to the idle growers (newcomers the candidates in
            //         flight have not tested yet wait for that -- except the head, which nothing can shadow)
            unsigned long long idle = self ? 0ull : __ballot(g_state == kTaskIdle);
            if (idle) {
                const long long t_em = tick();
                // A candidate's own seed box is published by its grower a moment after the hand-out.  Until then (npub == 0)
                // the coordinator stands in for it, on the side of the NEXT candidate: one test of that seed against the
                // seed boxes of the unpublished candidates before it (lane g: grower g) instead of a sweep over the pool.
                bool unp = g_live && g_pub == 0;
                while (idle) {
                    const unsigned elig = occupied & ~emitted & ~shadow;
                    unsigned l_min = kNone; int l_r = 0, l_pk = 0, l_if = 0;
#pragma unroll
                    for (int r = 0; r < WR; r++) {
                        const unsigned idx = (unsigned)(s_if[r] & kIdxMask);
                        if ((elig >> r) & 1u && (!((unver >> r) & 1u) || idx == hd) && idx < l_min) { l_min = idx; l_r = r; l_pk = s_pack[r]; l_if = s_if[r]; }
                    }
                    const unsigned mn = ~wave_max_u32(~l_min);
                    if (mn == kNone) break;
                    const bool own = l_min == mn;
                    const int owner = __builtin_ctzll(__ballot(own));
                    const int pk = rlane(l_pk, owner), fo = (int)((unsigned)rlane(l_if, owner) >> 24);
                    {   // inside the seed box of an earlier candidate that has not published it yet?
                        const int ccx = g_pk & 0xfff, ccy = (g_pk >> 12) & 0xfff, half = (g_pk >> 24) & 0xff;
                        const int dx = (pk & 0xfff) - ccx, dy = ((pk >> 12) & 0xfff) - ccy;
                        const bool hit = unp && g_f == fo && (unsigned)g_seed < mn && dx > -half && dx < half && dy > -half && dy < half;
                        if (__ballot(hit) != 0ull) {
                            if (own) { shadow |= 1u << l_r; ever |= 1u << l_r; }
                            continue;
                        }
                    }
                    const int g = __builtin_ctzll(idle);
                    idle &= idle - 1;
                    bool was_shadowed = false;
                    if (own) {
                        was_shadowed = (ever >> l_r) & 1u; emitted |= 1u << l_r;
                        gm_set(l_r, g);
                    }
                    shadow_by[g * kWave + lane] = 0u;    // nothing published for this task yet
                    if (lane == 0) {
                        task[g].seed = (int)mn; task[g].pk = pk; task[g].f = fo; task[g].npub = 0; task[g].coll = 0;
                        task[g].t_emit = (int)(t_iter - t_kernel);
                        flag_store(&task[g].cancel, 0);
                    }
                    wave_sync();
                    if (lane == 0) flag_store(&task[g].state, kTaskAssigned);
                    if (lane == g) { g_seed = (int)mn; g_pk = pk; g_f = fo; unp = true; }
                    stat(0, 1);
                    if (__ballot(was_shadowed) != 0ull) stat(5, 1);
                    progress = true;
                }
                if (timing) stat(19, (int)(wall_clock64() - t_em));
                if (la_on && idle != 0ull) want_la = true;   // growers left idle, nothing to hand out: look ahead in the next round
            }

            // ---- 6. what the next round waits for
            if (hd == kNone) {
                if (scan_pos >= n_seeds) break;          // no live seed in the pool, none left to scan
                continue;                                // pool ran empty: refill
            }
            if (hg < 0) hg = head_grower();
            if (self && hg < 0) { read_owners(); hg = head_grower(); }   // (claimed since the round's look at the owner words?)
            if (hg != last_hg) { if (lane == 0) __hip_atomic_store(&sh_ctl[9], hg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); last_hg = hg; }
            if (hg < 0) {
                // A head that was never handed out, or whose growth was stopped by a prediction that did not
                // come true.  It is never shadowed (a live candidate shadowing it would be the head), so step 5
                // takes it as soon as a grower is idle.  If every grower holds or grows a LATER seed that
                // none of the commits to come can free, the latest of them is given up.
                const int vstate = is_grower_lane ? flag_load(&task[lane].state) : kTaskAssigned;
                if (__ballot(is_grower_lane && (vstate == kTaskIdle || vstate == kTaskAccepted || flag_peek(&task[lane].cancel))) != 0ull) {
                    __builtin_amdgcn_s_sleep(1);         // a grower is idle or about to be
                    continue;
                }
                // (self-serve: a grower may have claimed the head a moment ago -- its slot says so before the owner word is read here)
                if (self && __ballot(is_grower_lane && vstate == kTaskAssigned && task[lane].seed == (int)hd) != 0ull) continue;
                const unsigned key = is_grower_lane ? ((unsigned)task[lane].seed << 6) | (unsigned)lane : 0u;
                const int victim = (int)(wave_max_u32(key) & 63u);
                const int vseed = task[victim].seed;
                if (lane == 0) flag_store(&task[victim].cancel, 1);
                if (self) {                              // (it stops or its result is dropped by step 2, gives its slot back and takes the head)
                    stat(4, 1);
                    continue;
                }
                while (flag_load(&task[victim].state) == kTaskAssigned && wall_clock64() - t_kernel <= kWatchdogTicks)
                    __builtin_amdgcn_s_sleep(2);
                if (lane == 0) flag_store(&task[victim].state, kTaskIdle);
#pragma unroll
                for (int r = 0; r < WR; r++)
                    if ((occupied >> r) & 1u && (s_if[r] & kIdxMask) == vseed) emitted &= ~(1u << r);
                stat(4, 1);
                continue;
            }
            if (!progress && flag_load(&task[hg].state) != kTaskDone) {
                __builtin_amdgcn_s_sleep(2);
                if (timing) wait_ticks += wall_clock64() - t_iter;   // an iteration that only waited for the head's growth
                stat(16, 1);
            }
        }
        // The last commit may still be in its grower's hands (ACCEPTED: marks + pose store).  A grower reads its
        // state once per polling round and then the exit flag: raised inside that window it would leave with the
        // pose unstored.  So the flag goes up only when no slot is in ACCEPTED any more (the same wait a refill does).
        while (__ballot(is_grower_lane && flag_load(&task[lane].state) == kTaskAccepted) != 0ull &&
               wall_clock64() - t_kernel <= kWatchdogTicks)
            __builtin_amdgcn_s_sleep(1);
        if (wall_clock64() - t_kernel > kWatchdogTicks) watchdog = true;
#ifdef OPA_ASSOC_WATCHDOG_DUMP
        if (watchdog && a.trace) {                       // diagnostic builds: what everybody was doing when the watchdog fired
            int* tr = a.trace + (size_t)b * kAssocTrace * 4;
            if (lane < NW) { int* t4 = tr + (40 + lane) * 4; t4[0] = task[lane].state; t4[1] = task[lane].seed; t4[2] = task[lane].cancel; t4[3] = task[lane].npub | (task[lane].pad1 << 16); }
            if (lane == 0) { int* t4 = tr + 60 * 4; t4[0] = (int)hd; t4[1] = scan_pos; t4[2] = n_live; t4[3] = epoch; }
            unsigned best = kNone; int bo = 0, be = 0, bs = 0;
#pragma unroll
            for (int r = 0; r < WR; r++) {
                const unsigned idx = (unsigned)(s_if[r] & kIdxMask);
                if ((occupied >> r) & 1u && idx < best) { best = idx; bo = pool_own[r * kWave + lane]; be = pool_ep[r * kWave + lane]; bs = 0;
                    for (int g = 1; g < NW; g++) bs |= ((shadow_by[g * kWave + lane] >> r) & 1u) << g; }
            }
            const unsigned mn = ~wave_max_u32(~best);
            if (best == mn && mn != kNone) { int* t4 = tr + 61 * 4; t4[0] = (int)mn; t4[1] = bo; t4[2] = be; t4[3] = bs; }
            if (lane == 0) { int* t4 = tr + 62 * 4; t4[0] = sh_ctl[12]; t4[1] = sh_ctl[9]; t4[2] = (int)iter; t4[3] = n_commits; }
        }
#endif
        if (lane == 0) {
            // failure codes (the image then reports no poses and OPA_COUNT_FAILED, its status word is minus the code):
            // 1 the watchdog fired, 2 the image's CIF map did not fit its tile pool (every lookup into a missing tile was wrong)
            const int fail_code = watchdog ? 1 : (a.hr_overflow && a.hr_overflow[b] != 0) ? 2 : 0;
            sh_ctl[1] = fail_code ? 0 : n_kept; sh_ctl[2] = n_dropped; sh_ctl[5] = fail_code;
            flag_store(&sh_ctl[0], 1);                   // growers leave
        }
        __builtin_amdgcn_s_setprio(0);
        if (lane == 0) {
            sh_stats[7] = n_seeds_all; sh_stats[8] = (int)(wall_clock64() - t_kernel);
            sh_stats[12] = (int)wait_ticks; sh_stats[15] = (int)iter;
        }
    } else if (wave <= S) {
        // ================================================================= grower
        TaskSlot* my = &task[wave];
        c.cancel = &my->cancel;
        long long busy_ticks = 0;
        int my_pos = -1;                                 // self-serve: the pool slot (slot * 64 + lane) this wave claimed last
        // Self-serve hand-out: take the next candidate -- the smallest-index pooled seed that is nobody's, is not predicted dead by
        // a candidate in flight, has been tested by every candidate in flight (or is the head, which nothing can shadow), and does
        // not lie in the seed box of an earlier candidate that has not published that box yet -- by a compare-and-swap on the
        // slot's owner word.  What the coordinator's hand-out loop did (step 5), done by the wave that would otherwise wait for it.
        auto try_claim = [&]() -> bool {
            constexpr unsigned kNone = 0xFFFFFFFFu;
            int st = kTaskIdle, cn = 0, sd = -1, npub = 0, gpk = 0, gf = -1, ack = 0;
            const bool gl = lane >= 1 && lane <= S && lane != wave;
            if (gl) {
                st = flag_load(&task[lane].state); cn = flag_peek(&task[lane].cancel); sd = task[lane].seed;
                npub = flag_peek(&task[lane].npub); gpk = task[lane].pk; gf = task[lane].f; ack = flag_peek(&task[lane].pad1);
            }
            const bool live = gl && (st == kTaskAssigned || st == kTaskDone) && !cn;
            const unsigned long long live_mask = __ballot(live);
            const unsigned min_ack = ~wave_max_u32(~(live ? (unsigned)ack : kNone));   // (all ones: no candidate in flight)
            const unsigned hd_now = (unsigned)flag_peek(&sh_ctl[12]);
            int sif[WR], spk[WR];
            unsigned elig = 0u;
            {
                int sep[WR], own[WR];
#pragma unroll
                for (int r = 0; r < WR; r++) {
                    sif[r] = flag_peek(&pool_if[r * kWave + lane]); spk[r] = pool_pack[r * kWave + lane];
                    sep[r] = pool_ep[r * kWave + lane]; own[r] = flag_peek(&pool_own[r * kWave + lane]);
                }
                unsigned sh = 0u;
                for (int g = 1; g <= S; g++)
                    if ((live_mask >> g) & 1ull) sh |= shadow_by[g * kWave + lane];
#pragma unroll
                for (int r = 0; r < WR; r++) {
                    const unsigned idx = (unsigned)(sif[r] & kPoolIdxMask);
                    const bool tested = min_ack == kNone || sep[r] - (int)min_ack <= 0 || idx == hd_now;
                    if (idx != (unsigned)kPoolIdxMask && own[r] == 0 && !((sh >> r) & 1u) && tested) elig |= 1u << r;
                }
            }
            const bool unp = live && npub == 0;
            for (;;) {
                unsigned l_min = kNone; int l_r = 0, l_pk = 0, l_if = 0;
#pragma unroll
                for (int r = 0; r < WR; r++) {
                    const unsigned idx = (unsigned)(sif[r] & kPoolIdxMask);
                    if ((elig >> r) & 1u && idx < l_min) { l_min = idx; l_r = r; l_pk = spk[r]; l_if = sif[r]; }
                }
                const unsigned mn = ~wave_max_u32(~l_min);
                if (mn == kNone) return false;
                const bool own_lane = l_min == mn;
                const int owner = __builtin_ctzll(__ballot(own_lane));
                const int pk = rlane(l_pk, owner), fo = (int)((unsigned)rlane(l_if, owner) >> 24), slot = rlane(l_r, owner);
                {   // inside the seed box of an earlier candidate that has not published it yet?
                    const int ccx = gpk & 0xfff, ccy = (gpk >> 12) & 0xfff, half = (gpk >> 24) & 0xff;
                    const int dx = (pk & 0xfff) - ccx, dy = ((pk >> 12) & 0xfff) - ccy;
                    const bool hit = unp && gf == fo && (unsigned)sd < mn && dx > -half && dx < half && dy > -half && dy < half;
                    if (__ballot(hit) != 0ull) { if (own_lane) elig &= ~(1u << l_r); continue; }
                }
                int won = 0;
                if (lane == owner) {
                    flag_store(&my->cancel, 0);          // (before the claim: a cancel that follows it is meant for it)
                    int* ow = &pool_own[slot * kWave + lane];
                    int expected = 0;
                    if (__hip_atomic_compare_exchange_strong(ow, &expected, wave, __ATOMIC_ACQ_REL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
                        // the slot may have changed hands between the look at it and the claim: the occupant has to be the same
                        if (flag_load(&pool_if[slot * kWave + lane]) == l_if) won = 1;
                        else { expected = wave; __hip_atomic_compare_exchange_strong(ow, &expected, 0, __ATOMIC_ACQ_REL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
                    }
                }
                if (__ballot(won) == 0ull) return false; // (somebody else's: look again at everything -- the winner's seed box is among the unpublished ones then)
                my_pos = slot * kWave + owner;
                shadow_by[wave * kWave + lane] = 0u;     // nothing published for this task yet
                if (lane == 0) {
                    my->seed = (int)mn; my->pk = pk; my->f = fo; my->npub = 0; my->coll = 0;
                    my->t_emit = (int)(wall_clock64() - t_kernel);
                    __hip_atomic_fetch_add(&sh_stats[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                wave_sync();
                if (lane == 0) flag_store(&my->state, kTaskAssigned);
                return true;
            }
        };
        for (;;) {
            bool leave = false;
            for (;;) {
                const int state = flag_load(&my->state);
                if (state == kTaskAssigned) break;
                if (state == kTaskDone && c.epoch) { const int e = flag_load(c.epoch); if (e != c.my_epoch) pool_catch_up<WR>(c, e); }
                else c.epoch = nullptr;
                if (state == kTaskAccepted) {            // the pose this wave grew was accepted: Occupancy::set + store
                    const PoseView q = pose_of_block(private_base, wave - 1, private_bytes, K);
                    const int slot = __builtin_amdgcn_readfirstlane(my->pad0);
                    occ_mark_pose(c, q);
                    if (slot >= 0) {
                        double* dst = anns + (size_t)slot * K * 4;
                        for (int k = lane; k < K; k += kWave) {
                            dst[4 * k + 0] = q.v[k]; dst[4 * k + 1] = (double)q.x[k];
                            dst[4 * k + 2] = (double)q.y[k]; dst[4 * k + 3] = (double)q.s[k];
                        }
                        if (lane == 0) ann_ids[slot] = -1;
                    }
                    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the marks are at the L2 before the coordinator reads the bitmap
                    wave_sync();
                    if (lane == 0) flag_store(&my->state, kTaskIdle);
                    continue;
                }
                if (flag_load(&sh_ctl[0]) || wall_clock64() - t_kernel > 2 * kWatchdogTicks) { leave = true; break; }
                if (self && state == kTaskIdle) {
                    if (my_pos >= 0) {                   // a growth that was stopped, or a result that was dropped: the seed is somebody's again
                        if (lane == 0) {
                            int expected = wave;
                            __hip_atomic_compare_exchange_strong(&pool_own[my_pos], &expected, 0, __ATOMIC_ACQ_REL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                        my_pos = -1;
                    }
                    if (try_claim()) break;
                }
                if (help_on) {                           // nothing of its own to do: a connection of the growth the commit waits for
                    const int hgw = flag_peek(&sh_ctl[9]);
                    if (hgw >= 1 && hgw <= S && hgw != wave) {
                        help_once<REG>(c, p, private_base + (size_t)(hgw - 1) * private_bytes, tgt_floats, a.spec != 0);
                        continue;
                    }
                }
                __builtin_amdgcn_s_sleep(2);
            }
            if (leave) break;
            const long long t0 = wall_clock64();
            PH(15);                                      // idle / waiting for a task
            PH(16);                                      // (empty interval: the cost of a stamp)
            const int mine = __builtin_amdgcn_readfirstlane(my->seed);
            const int sf = seed_f[mine]; const float4 sd = seed_vxys[mine];
            for (int k = lane; k < K; k += kWave) {
                c.jv[k] = 0.0; c.jx[k] = 0.f; c.jy[k] = 0.f; c.js[k] = 0.f;
                OccBox e; e.minx = e.miny = e.maxx = e.maxy = 0;
                c.jbox[k] = e;                           // nothing published yet
            }
            wave_sync();
            c.jv[sf] = (double)sd.x; c.jx[sf] = sd.y; c.jy[sf] = sd.z; c.js[sf] = sd.w;   // :213-218
            wave_sync();
            c.aborted = 0;
            c.pub = &my->npub; c.n_pub = 0; c.my_idx = mine;
            c.pool_if = pool_if; c.pool_pack = pool_pack; c.shadow_mine = shadow_by + wave * kWave;
            c.pool_ep = pool_ep; c.epoch = &my->epoch; c.ack = &my->pad1; c.head_g = &sh_ctl[9];
            c.my_epoch = flag_load(c.epoch);             // publish_joint tests the whole pool as of now; later refills: pool_catch_up
            if (lane == 0) flag_store(c.ack, c.my_epoch);
            publish_joint<WR>(c, p, sf, sd.y, sd.z, sd.w);   // the seed joint's own box: the rest of its blob
            if (__builtin_expect(sd.x >= a.predict_min_v, 0)) {   // ... and where its other joints will be (advisory; strong seeds only:
                // the one- and two-joint poses of weak seeds predict joints their search rejects)
                if constexpr (REG) predict_pose<WR>(c, p, rs, sf, sd.y, sd.z, sd.w);
                else predict_pose_lds<WR>(c, p, sf, sd.y, sd.z, sd.w, tgt_floats);
            }
            ++c.n_pub;                                   // (count 2: whatever this growth predicts is out -- the lookahead waits for that)
            if (lane == 0) flag_store(&my->npub, c.n_pub);
            PH(12);
            grow_pose<REG>(c, p, rs, true, 1.0, false);
            PH(13);
            c.pub = nullptr;
            if (c.prio) { __builtin_amdgcn_s_setprio(0); c.prio = 0; }
            const int* epoch_ptr = c.epoch;
            c.epoch = nullptr;                           // (pose_boxes rewrites the boxes: no tests in between)
            if (c.aborted) {
                if (lane == 0) flag_store(&my->state, kTaskIdle);
            } else {
                pose_boxes(c, p);
                wave_sync();
                const double sc = pose_score(c.jv, K);
                if (lane == 0) { my->score = sc; my->t_done = (int)(wall_clock64() - t_kernel); flag_store(&my->state, kTaskDone); }
                c.epoch = epoch_ptr;                     // still a candidate in flight: keeps testing newcomers while it waits
            }
            PH(14);
            busy_ticks += wall_clock64() - t0;
        }
        c.cancel = nullptr;
#ifdef OPA_ASSOC_SCAN_TIMING
        if (lane == 0) { atomicAdd(&sh_ctl[3], (int)busy_ticks); atomicAdd(&sh_ctl[4], c.n_blend); atomicAdd(&sh_ctl[6], c.t_blend); atomicAdd(&sh_ctl[7], c.t_blend_mem); }
#else
        if (lane == 0) { atomicAdd(&sh_ctl[3], (int)busy_ticks); atomicAdd(&sh_ctl[4], c.n_blend); atomicAdd(&sh_ctl[6], c.n_hit); atomicAdd(&sh_ctl[7], c.n_miss); }
#endif
    }
    sync_global();                        // stored poses visible to every wave; the private blocks are free
#ifdef OPA_ASSOC_PHASE_TIMING
    if (tid == 0 && (b == 0 || b == 3))
        for (int k = 0; k < kPhases; k++) printf("PHASE img %d k %d cycles %d n %d\n", b, k, g_ph[k], g_phn[k]);
#endif
    n_kept = sh_ctl[1]; n_dropped = sh_ctl[2];

    // ---- force complete (cifcaf.cpp:233-236,414-449) and the keypoint NMS behind it run in cifcaf_fc_kernel: the
    // stored poses are independent there, so they spread over several workgroups per image
    if (p.force_complete) {
        if (tid == 0) {
            int* meta = a.fc_meta + (size_t)b * 4;
            meta[0] = n_kept; meta[1] = n_dropped; meta[2] = sh_ctl[5]; meta[3] = 0;
        }
    } else {
        const NmsLds nl = nms_carve(work_base, a.max_ann, K);
        nms_and_store<kThreads>(a, p, c, nl, b, n_kept, n_dropped, sh_ctl[5], anns, ann_ids, nms_waves);
    }
    if (tid == 0 && a.stats) {
        sh_stats[9] = (int)(wall_clock64() - t_kernel);
        sh_stats[10] = sh_ctl[3]; sh_stats[11] = sh_ctl[4]; sh_stats[21] = sh_ctl[6]; sh_stats[22] = sh_ctl[7];
        sh_stats[13] = S; sh_stats[14] = n_kept;
        for (int k = 0; k < kAssocStats; k++) a.stats[(size_t)b * kAssocStats + k] = sh_stats[k];
    }
}

// One workgroup per image.  Which image: its block index -- or, round 6, for batches of more images than the chip has compute
// units, the next entry of a QUEUE: `queue_order` [B] lists the images by seed count, most seeds first (assoc_order_kernel),
// `queue_head` is the next entry, taken with one atomic when the workgroup starts.  The hardware's dispatcher is the pool of
// persistent workers (a workgroup starts when a compute unit is free); what the queue adds is the ORDER: a launch lasts as long
// as its most crowded image while the mean image takes half of that, so the long ones go first and the tail is the short ones.
// (A loop over images inside the kernel -- one resident workgroup per compute unit pulling from the same queue -- was built first:
// the loop costs the kernel 170 spilled vector registers, with the image's code inlined once or twice alike.)
// Replaces the reference's per-image loop over a fork pool (decoder/decoder.py:33-47,130-131).
template <bool REG, int NW>
__global__ __launch_bounds__(NW * kWave, 1) void cifcaf_assoc_kernel(AssocArgs a, DevSkeleton sk, DevParams p,
                                                                     int n_growers, int nms_waves, int tgt_floats) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int b = blockIdx.x;
    if (a.queue_order) {
        int* entry = reinterpret_cast<int*>(smem);
        if (threadIdx.x == 0) *entry = atomicAdd(a.queue_head, 1);
        __syncthreads();
        b = __builtin_amdgcn_readfirstlane(a.queue_order[__builtin_amdgcn_readfirstlane(*entry)]);   // (uniform: keep it scalar)
        __syncthreads();                              // (read by everybody before the image's code lays the LDS out)
    }
    assoc_image<REG, NW>(a, sk, p, n_growers, nms_waves, tgt_floats, b, smem);
}

// images by seed count, most first (ties by index): rank by counting, one thread per image
__global__ __launch_bounds__(1024) void assoc_order_kernel(const int32_t* __restrict__ seed_count, int B, int32_t* __restrict__ order,
                                                           int32_t* __restrict__ head) {
    __shared__ int tile[1024];
    if (threadIdx.x == 0 && blockIdx.x == 0) *head = 0;
    const int i = blockIdx.x * 1024 + threadIdx.x;
    const int mine = i < B ? seed_count[i] : 0;
    int rank = 0;
    for (int j0 = 0; j0 < B; j0 += 1024) {
        __syncthreads();
        tile[threadIdx.x] = j0 + threadIdx.x < B ? seed_count[j0 + threadIdx.x] : -1;
        __syncthreads();
        const int m = min(1024, B - j0);
        for (int k = 0; k < m; k++) { const int c = tile[k]; rank += (c > mine || (c == mine && j0 + k < i)) ? 1 : 0; }
    }
    if (i < B) order[rank] = i;
}

// ------------------------------------------------------------ force complete
// cifcaf.cpp:233-236,414-449 + the keypoint NMS behind it, as a kernel of its own.  The stored poses of an image
// are independent here (each is grown on with the caf_th 0.001 lists, no reverse match, a 4 sigma window, then flood
// filled), so they are spread over `S` workgroups per image, every wave a grower: a crowded image's 60 poses grow at
// once instead of eleven at a time inside the seed kernel's one workgroup.  The lists of this set hold most cells
// of a field (~100 chunks of 64): scans go through the chunk boxes cafscored left in global memory (blend_long).
// The workgroup of an image that finishes LAST (a counter in the workspace) runs the keypoint NMS and writes the
// output.  Keeping this code out of the seed kernel also keeps that kernel's registers where round 2 left them.
template <bool REG, int NW>
__global__ __launch_bounds__(NW * kWave, 1) void cifcaf_fc_kernel(AssocArgs a, DevSkeleton sk, DevParams p,
                                                                  int n_growers, int nms_waves, int S) {
    constexpr int kThreads = NW * kWave;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int b = blockIdx.x / S, part = blockIdx.x - b * S, tid = threadIdx.x, lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int K = a.K, A = a.A, E = 2 * A;
    int* meta = a.fc_meta + (size_t)b * 4;
    const int n_kept = meta[0], n_dropped = meta[1], failed = meta[2];

    ImageCtx c;
    c.K = K; c.A = A; c.F = a.F; c.wave = wave;
    c.lists = a.lists_fc + (size_t)b * A * 2 * 7 * a.list_cap;
    c.list_counts = a.list_counts_fc + (size_t)b * A * 2;
    c.list_cap = a.list_cap;
    c.raw_caf = nullptr; c.raw_w = 0; c.raw_stride = 1.f; c.predict_th = 0.f; c.n_predicted = nullptr; c.coll_shift = 0;
    c.occ_h = a.occ_h; c.occ_w = a.occ_w; c.occ_wpr = (a.occ_w + 31) >> 5;
    c.occ = nullptr;
    c.cancel = nullptr; c.aborted = 0; c.n_blend = 0; c.t_blend = 0; c.t_blend_mem = 0; c.pub = nullptr; c.n_pub = 0;
    c.pool_if = nullptr; c.pool_pack = nullptr; c.shadow_mine = nullptr; c.my_idx = 0;
    c.pool_ep = nullptr; c.epoch = nullptr; c.ack = nullptr; c.my_epoch = 0;
    c.head_g = nullptr; c.prio = 0;
    c.bbox = nullptr;
    c.nb = a.bbox_chunks;
    c.gbbox = a.list_bbox_fc ? reinterpret_cast<const float4*>(a.list_bbox_fc) + (size_t)b * E * a.bbox_chunks : nullptr;

    unsigned char* sp = smem;
    c.sh_counts = (int*)sp; sp += sizeof(int) * E;
    int* l_off = (int*)sp; sp += sizeof(int) * (K + 1);
    int* l_info = (int*)sp; sp += sizeof(int) * E;
    int* l_first = (int*)sp; sp += sizeof(int) * E;
    int* sh_last = (int*)sp; sp += sizeof(int) * 4;
    sp = smem + (((size_t)(sp - smem) + 15) & ~(size_t)15);
    unsigned char* work_base = sp;                   // the growers' private blocks, then the keypoint-NMS arrays and scratch
    const size_t private_bytes = assoc_private_bytes(K, A, REG, kBlendLdsFloats);
    carve_private<REG>(c, work_base + (size_t)(wave < n_growers ? wave : 0) * private_bytes, kBlendLdsFloats);

    for (int k = tid; k < E; k += kThreads) {
        c.sh_counts[k] = c.list_counts[k];
        int lo = 0, hi = K;                          // the joint whose adjacency range holds slot k
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (sk.adj_off[mid] <= k) lo = mid; else hi = mid; }
        l_info[k] = lo | (sk.adj_other[k] << 8) | (sk.adj_bone[k] << 16) | (sk.adj_fwd[k] << 24);
        l_first[k] = sk.adj_first[k];
    }
    for (int k = tid; k <= K; k += kThreads) l_off[k] = sk.adj_off[k];
    c.adj_off = l_off; c.slot_info = l_info; c.adj_first = l_first;
    __syncthreads();
    RegSkeleton rs; rs.slot_info = 0; rs.slot_first = 0; rs.off = 0; rs.off1 = 0;
    if constexpr (REG) {
        if (lane < E) { rs.slot_info = l_info[lane]; rs.slot_first = l_first[lane]; }
        if (lane < K) { rs.off = l_off[lane]; rs.off1 = l_off[lane + 1]; }
    }

    double* anns = a.anns + (size_t)b * a.max_ann * K * 4;
    const int64_t* ann_ids = a.ann_ids + (size_t)b * a.max_ann;
#ifdef OPA_FC_TIMING
    // diagnostic builds: per image, in 10-ns ticks -- trace row 56: longest growth phase of a workgroup [0], longest single pose [1],
    // the keypoint NMS of the last workgroup [2], poses [3]; row 57 [0]: (ticks << 10 | list scans) of the longest pose
    if (tid == 0) g_fc_hist = a.trace + 40 * 4;
    if (a.trace && tid < 8 && part == 0) a.trace[((size_t)b * kAssocTrace + 56) * 4 + tid] = 0;   // (racy against fast workgroups: diagnostic only)
    __syncthreads();
    const long long t_fc0 = wall_clock64();
    int t_pose_max = 0;
#endif
    if (wave < n_growers && !failed) {
        // pose n belongs to workgroup n % S, and there to wave (n / S) % n_growers
        for (int n = part + S * wave; n < n_kept; n += S * n_growers) {
#ifdef OPA_FC_TIMING
            const long long t_p0 = wall_clock64();
            const int nb0 = c.n_blend;
#endif
            double* src = anns + (size_t)n * K * 4;
            for (int k = lane; k < K; k += kWave) {
                c.jv[k] = src[4 * k + 0]; c.jx[k] = (float)src[4 * k + 1];
                c.jy[k] = (float)src[4 * k + 2]; c.js[k] = (float)src[4 * k + 3];
            }
            wave_sync();
            grow_pose<REG, true>(c, p, rs, false, 4.0, true);    // :419-425, :235
            for (int k = lane; k < K; k += kWave) {
                src[4 * k + 0] = c.jv[k]; src[4 * k + 1] = (double)c.jx[k];
                src[4 * k + 2] = (double)c.jy[k]; src[4 * k + 3] = (double)c.js[k];
            }
            wave_sync();
#ifdef OPA_FC_TIMING
            {
                const int dt = (int)(wall_clock64() - t_p0);
                t_pose_max = max(t_pose_max, dt);
                if (a.trace && lane == 0) atomicMax(a.trace + ((size_t)b * kAssocTrace + 57) * 4, (dt << 10) | min(c.n_blend - nb0, 1023));
            }
#endif
        }
    }
#ifdef OPA_FC_TIMING
    if (a.trace && lane == 0) {
        int* t4 = a.trace + ((size_t)b * kAssocTrace + 56) * 4;
        atomicMax(&t4[0], (int)(wall_clock64() - t_fc0));
        atomicMax(&t4[1], t_pose_max);
        t4[3] = n_kept;
    }
#endif
    // the last workgroup of the image to get here sees every pose (agent-scope release / acquire around the counter)
    __threadfence();
    __syncthreads();
    if (tid == 0) sh_last[0] = (atomicAdd(&meta[3], 1) == S - 1) ? 1 : 0;
    __syncthreads();
    if (!sh_last[0]) return;
    __threadfence();
    const NmsLds nl = nms_carve(work_base, a.max_ann, K);
#ifdef OPA_FC_TIMING
    const long long t_nms0 = wall_clock64();
#endif
    nms_and_store<kThreads>(a, p, c, nl, b, failed ? 0 : n_kept, n_dropped, failed, anns, ann_ids, nms_waves);
#ifdef OPA_FC_TIMING
    if (a.trace && tid == 0) a.trace[((size_t)b * kAssocTrace + 56) * 4 + 2] = (int)(wall_clock64() - t_nms0);
#endif
}

static int assoc_compute_units() {                    // of the current device (asked once per device)
    static int n_cu[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (n_cu[dev] == 0) {
        int v = 0;
        n_cu[dev] = hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0 ? v : 256;
    }
    return n_cu[dev];
}

template <bool REG, int NW>
static hipError_t launch_fc_nw(const AssocArgs& a, const DevSkeleton& sk, const DevParams& p, hipStream_t st) {
    const int K = a.K, A = a.A, E = 2 * A;
    const size_t shared = (sizeof(int) * (3 * E + K + 1 + 4) + 15) / 16 * 16;
    const size_t priv = assoc_private_bytes(K, A, REG, kBlendLdsFloats);
    const size_t budget = 160 * 1024;
    if (shared + priv > budget) return hipErrorInvalidValue;
    int growers = (int)((budget - shared) / priv);
    if (growers > NW) growers = NW;
    int nms_waves = NW;
    const size_t nms_fixed = nms_shared_bytes(a.max_ann, K);
    while (nms_waves > 1 && shared + nms_fixed + (size_t)nms_waves * nms_scratch_bytes(a.max_ann) > budget) nms_waves--;
    const size_t nms = nms_fixed + (size_t)nms_waves * nms_scratch_bytes(a.max_ann);
    if (shared + nms > budget) return hipErrorInvalidValue;
    const size_t grow_bytes = (size_t)growers * priv;
    const size_t lds = shared + (grow_bytes > nms ? grow_bytes : nms);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)cifcaf_fc_kernel<REG, NW>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    // workgroups per image: enough to put every stored pose of a crowded image on a wave of its own, and the chip to work
    int S = (a.max_ann + growers - 1) / growers;
    const int fill = (256 + a.B - 1) / a.B;
    if (S > fill) S = fill;
    if (S < 1) S = 1;
    if (a.fc_split >= 1 && a.fc_split <= 64) S = a.fc_split;   // tests: other splits
    cifcaf_fc_kernel<REG, NW><<<a.B * S, NW * kWave, lds, st>>>(a, sk, p, growers, nms_waves, S);
    return hipGetLastError();
}

template <bool REG, int NW>
static hipError_t launch_assoc_nw(const AssocArgs& a, const DevSkeleton& sk, const DevParams& p, hipStream_t st) {
    const int K = a.K, A = a.A, E = 2 * A;
    size_t shared = sizeof(TaskSlot) * NW + (sizeof(unsigned long long) << kDedupBits) + sizeof(int) * (3 * E + K + 1 + 16 + kAssocStats)
                  + sizeof(int) * ((kSelfServe ? 4 : 3) * (REG ? kPoolSlots : kPoolSlotsLds) + NW) * kWave + sizeof(int) * 2 * kSeedStage;
    shared = (shared + 15) / 16 * 16 + (REG && a.list_bbox ? sizeof(float4) * E * kListBboxChunks : 0);
    // work area behind it: one private block per grower while poses grow, the keypoint-NMS arrays afterwards
#ifdef OPA_ASSOC_PHASE_TIMING
    const size_t budget = 159 * 1024;                // the diagnostic's static LDS
#else
    const size_t budget = 160 * 1024;
#endif
    // the scan area of a grower: the full one (lists of up to 8 chunks in one round trip), or the small one when that
    // buys more growers (large skeletons: the frontier lives in LDS too, and their lists are short)
    const bool spec = a.spec != 0;
    int tgt_floats = kBlendLdsFloats;
    const bool help = kHelp && a.help != 0;
    size_t priv = assoc_private_bytes(K, A, REG, tgt_floats, spec, help);
    int growers = shared + priv <= budget ? (int)((budget - shared) / priv) : 0;
    if (growers < NW - 1) {
        const int small_floats = spec ? kSpecTgtFloats : kSmallTgtFloats;     // (the level walk's batches need more of it than two chunks)
        const size_t priv_small = assoc_private_bytes(K, A, REG, small_floats, spec, help);
        const int growers_small = shared + priv_small <= budget ? (int)((budget - shared) / priv_small) : 0;
        if (growers_small > growers) { growers = growers_small; priv = priv_small; tgt_floats = small_floats; }
    }
    if (growers < 1) return hipErrorInvalidValue;
    if (growers > NW - 1) growers = NW - 1;
    if (a.max_growers >= 1 && a.max_growers < growers) growers = a.max_growers;   // tests: other interleavings of the same result
    int nms_waves = NW;                              // large annotation capacities: fewer waves share the NMS pass
    const size_t nms_fixed = nms_shared_bytes(a.max_ann, K);
    while (nms_waves > 1 && shared + nms_fixed + (size_t)nms_waves * nms_scratch_bytes(a.max_ann) > budget) nms_waves--;
    const size_t nms = nms_fixed + (size_t)nms_waves * nms_scratch_bytes(a.max_ann);
    if (shared + nms > budget) return hipErrorInvalidValue;
    const size_t grow_bytes = (size_t)growers * priv;
    size_t lds = shared + (grow_bytes > nms ? grow_bytes : nms);
    if (a.tie_fused && lds < tie_lds_bytes<NW * kWave>()) lds = tie_lds_bytes<NW * kWave>();   // the tie pass borrows the block first
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)cifcaf_assoc_kernel<REG, NW>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    if (a.queue_order)                                // more images than compute units: longest first
        assoc_order_kernel<<<(a.B + 1023) / 1024, 1024, 0, st>>>(a.seed_count, a.B, a.queue_order, a.queue_head);
    cifcaf_assoc_kernel<REG, NW><<<a.B, NW * kWave, lds, st>>>(a, sk, p, growers, nms_waves, tgt_floats);
    return hipGetLastError();
}

// waves per workgroup of the association kernel: 1 coordinator + up to NW-1 growers.  12 is what ships; the 8- and
// 16-wave instantiations (OPA_ASSOC_WAVES, interleaving experiments of round 2) double the compile time of this file
// and are built only with -DOPA_ASSOC_ALL_WAVES.
static int assoc_waves(int v) {
#ifdef OPA_ASSOC_ALL_WAVES
    return v == 8 || v == 12 || v == 16 ? v : kAssocWavesDefault;
#else
    return v == 8 || v == 12 ? v : kAssocWavesDefault;
#endif
}

hipError_t launch_assoc(const AssocArgs& args, const DevSkeleton& sk, const DevParams& p, hipStream_t st, const opa_debug& dbg) {
    AssocArgs a = args;
    // (every switch comes from the decoder's opa_debug -- opa_cifcaf_set_debug; its defaults took the OPA_* environment variables
    // in when the library was loaded -- no launch reads the environment)
    if (!dbg.assoc_bbox) a.list_bbox = a.list_bbox_fc = nullptr;                          // A/B: scan every chunk
    // (the argument needs a joint's box to hold the joint's own cell: true for a reduced minimum scale >= 1 cell --
    // the reference's is 2 -- not for a box that may shrink to the one cell next to it; and the minimum is only applied
    // when the map is reduced at all, occupancy.cpp:14-18: with reduction == 1 the box is the raw joint scale)
    a.dedup = (p.occupancy_reduction != 1.0 && p.occupancy_min_scale_reduced >= 1.0) ? 1 : 0;
    if (!dbg.assoc_dedup) a.dedup = 0;                                                    // A/B and tests: same result without it
    a.predict = dbg.assoc_predict != 0;           // 0: boxes are published only for assigned joints
    a.predict_th = dbg.assoc_predict_th;          // raw CAF confidence a predicted bone needs
    a.predict_min_v = dbg.assoc_predict_min_v;    // seeds below this confidence grow without the walk
    a.coll_shift = dbg.assoc_collide_shift;       // 0: anywhere inside the earlier candidate's box (round 4)
    a.lookahead = dbg.assoc_lookahead != 0;       // 0: seeds enter the pool in list order only
    a.prededup = dbg.assoc_prededup != 0;         // 0: the coordinator's refill walks every seed
    a.inherit = dbg.assoc_inherit != 0;           // 0: predictions lapse with the growth that made them
    a.collide = dbg.assoc_collide != 0;           // 0: growths stop only when their SEED is covered
    a.timing = dbg.assoc_timing != 0;             // phase tick counters of the coordinator (statistics slots 12, 17-20)
    a.help = kHelp && dbg.assoc_help != 0;        // (helper builds) 0: every growth scans its own lists
    a.spec = kWalk && dbg.assoc_spec != 0;        // (walk builds) 0: every bone scanned on demand, one at a time
    a.watchdog_ticks = dbg.assoc_watchdog_ticks > 0 ? dbg.assoc_watchdog_ticks : kWatchdogTicksDefault;
    a.max_growers = dbg.assoc_growers;
    if (dbg.assoc_persistent < 0 || (dbg.assoc_persistent == 0 && a.B <= assoc_compute_units())) a.queue_order = a.queue_head = nullptr;
    a.fc_split = dbg.fc_split;
    const int waves = assoc_waves(dbg.assoc_waves);
    const int K = a.K, E = 2 * a.A;
    // the seed pool packs cell coordinates into 12 bits, the field into 8 and the seed index into 24
    if (a.occ_w > 4096 || a.occ_h > 4096 || a.F > 256 || a.seed_cap > 0xFFFFFF) return hipErrorInvalidValue;
    if (K > 256 || a.A > 256) return hipErrorInvalidValue;      // a slot's joints and bone are packed into 8 bits each
    const bool reg = K <= kWave && E <= kWave;       // pose, frontier and heap fit the lanes of a wave
    hipError_t e;
    if (!reg) e = waves == 8 ? launch_assoc_nw<false, 8>(a, sk, p, st) : launch_assoc_nw<false, 12>(a, sk, p, st);  // LDS-resident growth state
    else switch (waves) {
        case 8: e = launch_assoc_nw<true, 8>(a, sk, p, st); break;
#ifdef OPA_ASSOC_ALL_WAVES
        case 16: e = launch_assoc_nw<true, 16>(a, sk, p, st); break;
#endif
        default: e = launch_assoc_nw<true, 12>(a, sk, p, st); break;
    }
    prof_mark(st, "cifcaf_assoc_kernel");
    if (e == hipSuccess && p.force_complete) {
        if (!a.fc_meta || !a.lists_fc) return hipErrorInvalidValue;
        e = reg ? launch_fc_nw<true, 12>(a, sk, p, st) : launch_fc_nw<false, 12>(a, sk, p, st);
        prof_mark(st, "cifcaf_fc_kernel");
    }
    return e;
}

// ------------------------------------------- exported grow_connection_blend op
// cifcaf.cpp:105-113 : rows [n,7] (AoS, as the reference's op takes it)
__global__ __launch_bounds__(64) void blend_rows_kernel(const float* rows, int n, double x, double y, double s,
                                                        double filter_sigmas, int only_max, float* soa, double* out4) {
    const int lane = lane_id();
    for (int i = lane; i < n; i += kWave)
        for (int k = 0; k < 7; k++) soa[(size_t)k * n + i] = rows[(size_t)i * 7 + k];
    __threadfence_block();
    __shared__ float tgt[kBlendLdsFloats];
    ListView L; L.base = soa; L.cap = n; L.n = n; L.bbox = nullptr; L.gbbox = nullptr; L.nb = 0;
    const BlendResult r = blend_impl(L.base, L.cap, L.n, x, y, s, filter_sigmas, only_max, tgt);
    if (lane == 0) {
        if (r.ok) { out4[0] = (double)r.x; out4[1] = (double)r.y; out4[2] = (double)r.s; out4[3] = r.v; }
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
