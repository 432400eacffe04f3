// The reference's order of EQUAL seed scores (cif_seeds.cpp:94: an unstable std::sort) as device code shared by the
// stand-alone tie kernel (cifseeds.hip: stage-level entry points) and the association kernel (cifcaf.hip: inside the
// decode every image fixes its own ties before its seeds are read -- images without equal scores pay nothing, and the
// batch no longer waits for a launch that lasts as long as its most tied image).
#pragma once
#include "common.hpp"

namespace opa {

__device__ __forceinline__ unsigned sortable_bits(float v) {
    const unsigned u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float from_sortable(unsigned s) {
    return __uint_as_float((s & 0x80000000u) ? (s & 0x7fffffffu) : ~s);
}


// keys -> sorted seeds (cif_seeds.cpp:100-113): the seed of rank t
__device__ __forceinline__ void store_seed(unsigned long long key, int t, int b, const float* __restrict__ cif, int F, int NC,
                                           int HW, int stride, int cap, int32_t* __restrict__ seed_f,
                                           float* __restrict__ seed_vxys, int32_t* __restrict__ seed_cell, int occ_h,
                                           int occ_w, const DevParams& p) {
    int32_t* sf = seed_f + (size_t)b * cap;
    const int ncol = NC - 1;                    // (v,x,y,s) for CIF; (v,x,y,w,h) for CifDet, cif_seeds.cpp:124-137
    float* sv = seed_vxys + (size_t)b * cap * ncol;
    const float* image = cif + (size_t)b * F * NC * HW;
    const unsigned idx = 0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull);
    const int f = (int)(idx / (unsigned)HW), o = (int)(idx - (unsigned)f * (unsigned)HW);
    const float* P = image + (size_t)f * NC * HW;
    sf[t] = f;
    float4 r;
    r.x = from_sortable((unsigned)(key >> 32));
    r.y = P[2 * HW + o] * (float)stride;
    r.z = P[3 * HW + o] * (float)stride;
    r.w = P[4 * HW + o] * (float)stride;                        // cif_seeds.cpp:61
    if (seed_cell) seed_cell[(size_t)b * cap + t] = seed_cell_pack(p, occ_h, occ_w, (double)r.y, (double)r.z, (double)r.w);
    if (NC == 5) {
        reinterpret_cast<float4*>(sv)[t] = r;
    } else {                                                    // cif_seeds.cpp:85-87
        float* row = sv + (size_t)t * 5;
        row[0] = r.x; row[1] = r.y; row[2] = r.z; row[3] = r.w; row[4] = P[5 * HW + o] * (float)stride;
    }
}


// ------------------------------------------------------------------ the reference's order of EQUAL scores
// CifSeeds::get sorts with std::sort (cif_seeds.cpp:94), which is not stable: where two seeds have the same score,
// their order is whatever libstdc++'s introsort leaves -- and that order decides which of them is grown first.  Float32
// fields of a network rarely tie; fields rounded to bfloat16 do all the time (round 2: 4 % of such images decode
// differently, by up to 0.8 px).  The sort above orders equal scores by cell index.  This pass, for the images that
// have ties, reproduces libstdc++ (bits/stl_algo.h: __introsort_loop, __unguarded_partition_pivot,
// __move_median_to_first, __unguarded_partition, __final_insertion_sort) on the sequence the reference sorts -- the
// seeds in raster order (field, row, column: cif_seeds.cpp:33-66):
//   * raster position of every seed: the fill kernel writes one block of keys per (field, 1024 cells), in cell order, and
//     notes where each block went; a prefix sum over the blocks' lengths in (field, chunk) order places them;
//   * the introsort loop, level by level (every partition of a level takes a depth step, like the recursion), one wave per
//     segment.  A Hoare partition's swaps are fixed by the ORIGINAL segment: scanning from the left it stops at
//     elements with !(x > pivot), from the right at !(pivot > x), and the k-th stop on the left is swapped with the k-th
//     on the right while it lies to the left of it -- one scan that numbers the stops, one pass that swaps the pairs,
//     no sequential two-pointer walk; a segment of up to 64 elements lives in one wave's registers;
//   * only segments that hold a seed whose score occurs twice are followed: a seed with a score of its own ends up at
//     its rank whatever the loop does to it, which is where the first sort put it;
//   * __final_insertion_sort only moves an element past strictly smaller scores, and the loop leaves segments of at most
//     16 elements in their final places relative to each other: it is a stable sort INSIDE every such segment -- one
//     thread per tied seed counts the larger (and the equal, earlier) elements of its segment and stores the seed there.
// Heapsort (the depth limit, 2 log2 n levels; round 6): a segment that HOLDS EQUAL SCORES and reaches the limit is heap-sorted
// like std::__partial_sort does it (tie_heapsort: one lane, serially); segments without equal scores that reach it do not matter
// (whatever sorts them leaves them as the first sort did).  `tie_state` -1 is left for a protocol failure (a partition without a
// stop on one side: cannot happen after the median step).  Images without ties leave after one look at their sorted scores.
constexpr int kTieLdsKeys = 8192;
constexpr int kTieThreads = 1024;          // of the stand-alone kernel; the pass itself is a template on the workgroup size
constexpr unsigned kTiedBit = 0x80000000u;      // in a cell index: the seed's score occurs more than once


__host__ __device__ inline size_t tie_seg_cap(int cells) { return (size_t)cells / 17 + 2; }
__host__ __device__ inline int tie_blocks(int F, int HW) { return F * ((HW + 256 * kFillCells - 1) / (256 * kFillCells)); }
// the fill kernel's copy of the keys lies in the two stop-list arrays (not in use before the partitions start)
__host__ __device__ inline unsigned long long* tie_key_copy(unsigned char* big, int cells) { return (unsigned long long*)(big + 2 * (size_t)cells * sizeof(unsigned)); }

// the arrays of an image live in LDS (ds_read / ds_write through address-space-3 pointers) or in global memory
typedef __attribute__((address_space(3))) unsigned lds_u32;
template <bool LDS> struct TiePtr { typedef unsigned* type; };
template <> struct TiePtr<true> { typedef lds_u32* type; };
// ... read and written through this: in global memory every load bypasses the L1 (agent scope: it sees what other lanes and
// waves have stored, once their stores are acknowledged) -- no cache invalidation between the steps of a partition
template <bool LDS> struct TieArr {
    typename TiePtr<LDS>::type p;
    __device__ __forceinline__ unsigned operator()(int i) const {
        if constexpr (LDS) return p[i];
        else return __hip_atomic_load(p + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __device__ __forceinline__ void set(int i, unsigned v) const { p[i] = v; }
};
template <bool LDS> __device__ __forceinline__ TieArr<LDS> tie_arr(unsigned* p) { TieArr<LDS> a; a.p = (typename TiePtr<LDS>::type)p; return a; }

template <typename A>
__device__ __forceinline__ void tie_swap(const A& BITS, const A& IDX, int i, int j) {
    const unsigned bi = BITS(i), bj = BITS(j), xi = IDX(i), xj = IDX(j);
    BITS.set(i, bj); BITS.set(j, bi); IDX.set(i, xj); IDX.set(j, xi);
}
// what one wave's stores must be before the same wave's other lanes read them back: LDS is in order per wave; global
// memory goes through the L2 (release: stores acknowledged; acquire: stale L1 lines dropped)
template <bool LDS>
__device__ __forceinline__ void tie_wave_fence() {
    if constexpr (LDS) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_s_waitcnt(0xC07F);          // lgkmcnt(0)
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_waitcnt(0x0070);          // vmcnt(0): on gfx9 stores count too, until the L2 has them
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}
template <bool LDS>
__device__ __forceinline__ void tie_group_sync() {   // the same between the waves of the workgroup
    if constexpr (LDS) __syncthreads();
    else {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_waitcnt(0x0070);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
}
__device__ __forceinline__ void tie_group_sync_rt(bool lds) { if (lds) __syncthreads(); else sync_global(); }

// __unguarded_partition_pivot(first, last) on scores BITS (comp(a, b) = a > b) with the cells IDX moving along; returns
// the cut; -2 if no element of the segment has a tied score (the segment is left alone); -1 if the segment has no stop
// on one side (cannot happen after the median step; the caller gives up).  LPOS / RPOS [first, last): scratch of this segment.
template <bool LDS>
__device__ __forceinline__ int tie_partition(unsigned* bits_, unsigned* idx_, unsigned* lpos_, unsigned* rpos_, int first, int last) {
    const TieArr<LDS> BITS = tie_arr<LDS>(bits_), IDX = tie_arr<LDS>(idx_), LPOS = tie_arr<LDS>(lpos_), RPOS = tie_arr<LDS>(rpos_);
    const int lane = threadIdx.x & 63;
    const int len = last - first;
    const unsigned long long below = (1ull << lane) - 1ull;
    if (len <= 64) {
        // ---- the whole segment in one wave's registers: one load, one store
        const int i = first + lane;
        const bool in = lane < len;
        unsigned x = in ? BITS(i) : 0u, id = in ? IDX(i) : 0u;
        if (__ballot((id & kTiedBit) != 0u) == 0ull) return -2;
        {   // __move_median_to_first(first, first + 1, mid, last - 1): lanes 0, 1, len / 2, len - 1
            const int lb = len / 2, lc = len - 1;
            const unsigned va = (unsigned)__builtin_amdgcn_readlane((int)x, 1), vb = (unsigned)__builtin_amdgcn_readlane((int)x, lb),
                           vc = (unsigned)__builtin_amdgcn_readlane((int)x, lc);
            int lt;
            if (va > vb) { if (vb > vc) lt = lb; else if (va > vc) lt = lc; else lt = 1; }
            else if (va > vc) lt = 1;
            else if (vb > vc) lt = lc;
            else lt = lb;
            const unsigned x0 = (unsigned)__builtin_amdgcn_readlane((int)x, 0), xt = (unsigned)__builtin_amdgcn_readlane((int)x, lt);
            const unsigned d0 = (unsigned)__builtin_amdgcn_readlane((int)id, 0), dt = (unsigned)__builtin_amdgcn_readlane((int)id, lt);
            if (lane == 0) { x = xt; id = dt; } else if (lane == lt) { x = x0; id = d0; }
        }
        const unsigned pv = (unsigned)__builtin_amdgcn_readlane((int)x, 0);
        const bool valid = in && lane >= 1;
        const bool is_l = valid && !(x > pv), is_r = valid && !(pv > x);
        const unsigned long long ml = __ballot(is_l), mr = __ballot(is_r);
        const int tot_l = __popcll(ml);
        if (ml == 0ull || mr == 0ull) return -1;
        const int cl = __popcll(ml & below);                                   // left stops before this lane
        const int cr = __popcll(mr & ~(below | (1ull << lane)));               // right stops behind it
        const bool sw_l = is_l && cr >= cl + 1;      // the cl-th left stop meets the cl-th right stop (from the right) to its right
        const int m = __popcll(__ballot(sw_l));
        const bool sw_r = is_r && cr < m;            // the m rightmost right stops are their partners
        if (sw_l) LPOS.set(first + cl, (unsigned)lane);
        if (sw_r) RPOS.set(first + cr, (unsigned)lane);
        tie_wave_fence<LDS>();
        int partner = lane;
        if (sw_l) partner = (int)RPOS(first + cl);
        else if (sw_r) partner = (int)LPOS(first + cr);
        const unsigned nx = (unsigned)__shfl((int)x, partner, 64), nd = (unsigned)__shfl((int)id, partner, 64);
        if (in) { BITS.set(i, nx); IDX.set(i, nd); }
        int cut;
        if (m == 0) cut = __builtin_ctzll(ml);
        else {
            cut = __builtin_ctzll(__ballot(is_r && cr == m - 1));              // R_(m-1)
            if (m < tot_l) { const int l = __builtin_ctzll(__ballot(is_l && cl == m)); if (l < cut) cut = l; }
        }
        tie_wave_fence<LDS>();
        return first + cut;
    }
    {   // a segment without a tied score is left alone: eight chunks of flags per round trip
        bool tied = false;
        for (int i0 = first; i0 < last; i0 += 8 * 64) {
            unsigned d[8];
#pragma unroll
            for (int u = 0; u < 8; u++) { const int i = i0 + u * 64 + lane; d[u] = IDX(i < last ? i : first); }   // (unconditional loads: all in flight)
#pragma unroll
            for (int u = 0; u < 8; u++) asm volatile("" : "+v"(d[u]) :: "memory");
#pragma unroll
            for (int u = 0; u < 8; u++) tied |= i0 + u * 64 + lane < last && (d[u] & kTiedBit) != 0u;
        }
        if (__ballot(tied) == 0ull) return -2;
    }
    {   // __move_median_to_first(first, first + 1, mid, last - 1)
        const int ia = first + 1, ib = first + len / 2, ic = last - 1;
        const unsigned va = BITS(ia), vb = BITS(ib), vc = BITS(ic);
        int t;
        if (va > vb) { if (vb > vc) t = ib; else if (va > vc) t = ic; else t = ia; }
        else if (va > vc) t = ia;
        else if (vb > vc) t = ic;
        else t = ib;
        tie_wave_fence<LDS>();                       // (every lane has read the three before lane 0 swaps)
        if (lane == 0) tie_swap(BITS, IDX, first, t);
        tie_wave_fence<LDS>();
    }
    const unsigned pv = BITS(first);
    // one scan: the left stops in order, and the right stops in order FROM THE LEFT (the k-th from the right is
    // RPOS(first + tot_r - 1 - k) once the total is known); four chunks' loads in flight together
    int nl = 0, nr = 0;
    for (int i0 = first + 1; i0 < last; i0 += 4 * 64) {
        unsigned x[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { const int i = i0 + u * 64 + lane; x[u] = BITS(i < last ? i : first); }
#pragma unroll
        for (int u = 0; u < 4; u++) asm volatile("" : "+v"(x[u]) :: "memory");
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int i = i0 + u * 64 + lane;
            const bool is_l = i < last && !(x[u] > pv), is_r = i < last && !(pv > x[u]);
            const unsigned long long ml = __ballot(is_l), mr = __ballot(is_r);
            if (is_l) LPOS.set(first + nl + __popcll(ml & below), (unsigned)i);
            if (is_r) RPOS.set(first + nr + __popcll(mr & below), (unsigned)i);
            nl += __popcll(ml); nr += __popcll(mr);
        }
    }
    if (nl == 0 || nr == 0) return -1;
    tie_wave_fence<LDS>();
    // the k-th left stop is swapped with the k-th right stop from the right while it lies to its left
    int m = 0;
    const int kmax = nl < nr ? nl : nr;
    for (int k0 = 0; k0 < kmax; k0 += 64) {
        const int k = k0 + lane;
        const int l = k < kmax ? (int)LPOS(first + k) : 0, r = k < kmax ? (int)RPOS(first + nr - 1 - k) : 0;
        const bool sw = k < kmax && l < r;
        if (sw) tie_swap(BITS, IDX, l, r);
        const int c = __popcll(__ballot(sw));
        m += c;
        if (c < 64) break;
    }
    int cut;
    if (m == 0) cut = (int)LPOS(first);
    else {
        cut = (int)RPOS(first + nr - m);                                       // R_(m-1)
        if (m < nl) { const int l = (int)LPOS(first + m); if (l < cut) cut = l; }
    }
    tie_wave_fence<LDS>();
    return cut;
}

// The same partition by ALL waves of the workgroup, for the few long segments at the top of the recursion: every wave
// scans a slice (counts first, then the numbered stops behind the counts of the waves before it), all threads swap pairs.
// `sh`: 2 * 16 + 2 ints of LDS.  Returns like tie_partition (the value is the same in every thread).
// (worth its barriers from ~1 000 elements in LDS, ~4 000 in global memory, where a wave's own partition is a handful of L2
// round trips and sixteen of them run side by side)
template <bool LDS> constexpr int tie_coop_len() { return LDS ? 1024 : 4096; }
template <bool LDS, int NT>
__device__ __forceinline__ int tie_partition_block(unsigned* bits_, unsigned* idx_, unsigned* lpos_, unsigned* rpos_, int first,
                                                   int last, int* sh) {
    const TieArr<LDS> BITS = tie_arr<LDS>(bits_), IDX = tie_arr<LDS>(idx_), LPOS = tie_arr<LDS>(lpos_), RPOS = tie_arr<LDS>(rpos_);
    constexpr int NW = NT / 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int len = last - first;
    const unsigned long long below = (1ull << lane) - 1ull;
    int* s_l = sh; int* s_r = sh + NW; int* s_tied = sh + 2 * NW; int* s_m = sh + 2 * NW + 1;
    if (tid == 0) {   // __move_median_to_first(first, first + 1, mid, last - 1)
        const int ia = first + 1, ib = first + len / 2, ic = last - 1;
        const unsigned va = BITS(ia), vb = BITS(ib), vc = BITS(ic);
        int t;
        if (va > vb) { if (vb > vc) t = ib; else if (va > vc) t = ic; else t = ia; }
        else if (va > vc) t = ia;
        else if (vb > vc) t = ic;
        else t = ib;
        tie_swap(BITS, IDX, first, t);
        *s_tied = (IDX(first) & kTiedBit) ? 1 : 0; *s_m = 0;
    }
    tie_group_sync<LDS>();
    const unsigned pv = BITS(first);
    const int per = ((len - 1 + NW * 64 - 1) / (NW * 64)) * 64;      // a multiple of 64 per wave
    const int a0 = first + 1 + wave * per, a1 = min(last, a0 + per);
    int nl = 0, nr = 0;
    bool tied = false;
    for (int i0 = a0; i0 < a1; i0 += 4 * 64) {       // four chunks' loads in flight together
        unsigned x[4], d[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { const int i = i0 + u * 64 + lane; x[u] = BITS(i < a1 ? i : first); d[u] = IDX(i < a1 ? i : first); }
#pragma unroll
        for (int u = 0; u < 4; u++) asm volatile("" : "+v"(x[u]), "+v"(d[u]) :: "memory");
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int i = i0 + u * 64 + lane;
            nl += __popcll(__ballot(i < a1 && !(x[u] > pv)));
            nr += __popcll(__ballot(i < a1 && !(pv > x[u])));
            tied |= i < a1 && (d[u] & kTiedBit) != 0u;
        }
    }
    if (lane == 0) { s_l[wave] = nl; s_r[wave] = nr; }
    if (__ballot(tied) != 0ull && lane == 0) *s_tied = 1;
    tie_group_sync<LDS>();
    if (!*s_tied) { tie_group_sync<LDS>(); return -2; }
    int off_l = 0, off_r = 0, tot_l = 0, tot_r = 0;
    for (int k = 0; k < NW; k++) { const int cl = s_l[k], cr = s_r[k]; if (k < wave) { off_l += cl; off_r += cr; } tot_l += cl; tot_r += cr; }
    if (tot_l == 0 || tot_r == 0) { tie_group_sync<LDS>(); return -1; }
    for (int i0 = a0; i0 < a1; i0 += 4 * 64) {
        unsigned x[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { const int i = i0 + u * 64 + lane; x[u] = BITS(i < a1 ? i : first); }
#pragma unroll
        for (int u = 0; u < 4; u++) asm volatile("" : "+v"(x[u]) :: "memory");
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int i = i0 + u * 64 + lane;
            const bool is_l = i < a1 && !(x[u] > pv), is_r = i < a1 && !(pv > x[u]);
            const unsigned long long ml = __ballot(is_l), mr = __ballot(is_r);
            if (is_l) LPOS.set(first + off_l + __popcll(ml & below), (unsigned)i);
            if (is_r) RPOS.set(first + off_r + __popcll(mr & below), (unsigned)i);
            off_l += __popcll(ml); off_r += __popcll(mr);
        }
    }
    tie_group_sync<LDS>();
    const int kmax = tot_l < tot_r ? tot_l : tot_r;
    int cnt = 0;
    for (int k = tid; k < kmax; k += NT) {
        const int l = (int)LPOS(first + k), r = (int)RPOS(first + tot_r - 1 - k);
        if (l < r) { tie_swap(BITS, IDX, l, r); cnt++; }
    }
    for (int d = 32; d > 0; d >>= 1) cnt += __shfl_xor(cnt, d, 64);
    if (lane == 0 && cnt) atomicAdd(s_m, cnt);
    tie_group_sync<LDS>();
    const int m = *s_m;
    int cut;
    if (m == 0) cut = (int)LPOS(first);
    else {
        cut = (int)RPOS(first + tot_r - m);                                    // R_(m-1)
        if (m < tot_l) { const int l = (int)LPOS(first + m); if (l < cut) cut = l; }
    }
    tie_group_sync<LDS>();
    return cut;
}

// std::__partial_sort(first, last, last, comp) -- what __introsort_loop does with a segment that is still longer than 16
// elements at its depth limit (bits/stl_algo.h:1944-1948): __make_heap + __sort_heap (bits/stl_heap.h: __adjust_heap, __push_heap,
// __pop_heap), comp(a, b) = a.v > b.v, on the (score, cell) pairs of [first, last).  ONE lane, serially, exactness over speed:
// only a sequence built against libstdc++'s pivot rule gets here (tests/test_tie_depth_limit.py: McIlroy's adversary), and only
// segments that hold equal scores are sorted at all.  The caller marks every position of the segment afterwards: it is sorted,
// so __final_insertion_sort moves nothing in it.  (Arrays in global memory: relaxed agent-scope atomics both ways, which keeps
// the lane's own loads behind its own stores.)
template <bool LDS>
__device__ __noinline__ void tie_heapsort(unsigned* bits_, unsigned* idx_, int first, int last) {
    auto ld = [](unsigned* p, int i) -> unsigned {
        if constexpr (LDS) return p[i];
        else return __hip_atomic_load(p + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    auto st = [](unsigned* p, int i, unsigned v) {
        if constexpr (LDS) p[i] = v;
        else __hip_atomic_store(p + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    unsigned* B = bits_ + first; unsigned* I = idx_ + first;
    const int len0 = last - first;
    // __adjust_heap(first, hole, len, value) with __push_heap at its end; comp(x, y) = x > y on the scores
    auto adjust = [&](int hole, int len, unsigned vb, unsigned vi) {
        const int top = hole;
        int child = hole;
        while (child < (len - 1) / 2) {
            child = 2 * (child + 1);
            if (ld(B, child) > ld(B, child - 1)) child--;
            st(B, hole, ld(B, child)); st(I, hole, ld(I, child));
            hole = child;
        }
        if ((len & 1) == 0 && child == (len - 2) / 2) {
            child = 2 * (child + 1);
            st(B, hole, ld(B, child - 1)); st(I, hole, ld(I, child - 1));
            hole = child - 1;
        }
        int parent = (hole - 1) / 2;                  // __push_heap(first, hole, top, value)
        while (hole > top && ld(B, parent) > vb) {
            st(B, hole, ld(B, parent)); st(I, hole, ld(I, parent));
            hole = parent;
            parent = (hole - 1) / 2;
        }
        st(B, hole, vb); st(I, hole, vi);
    };
    if (len0 >= 2)                                    // __make_heap
        for (int parent = (len0 - 2) / 2; ; parent--) {
            adjust(parent, len0, ld(B, parent), ld(I, parent));
            if (parent == 0) break;
        }
    for (int end = len0; end > 1; ) {                 // __sort_heap: __pop_heap(first, end - 1, end - 1)
        end--;
        const unsigned vb = ld(B, end), vi = ld(I, end);
        st(B, end, ld(B, 0)); st(I, end, ld(I, 0));
        adjust(0, end, vb, vi);
    }
}

// A segment of at most kTieSubtree elements is finished by ONE wave, all the levels below it (no workgroup barrier per
// level: most of an image's partitions are down here).  An image in global memory brings the segment into the wave's
// own LDS area first (scores, cells, two stop lists of kTieSubtree entries) and takes it back afterwards.  `stack`: 3 *
// kTieStack ints of LDS per wave: (first, last, depth) of the segments still to be partitioned.
constexpr int kTieSubtree = 512;
constexpr int kTieSubtreeLds = 0;        // (an image in LDS pays little per level: sharing every level among its waves is faster -- 78 vs 84 us)
constexpr int kTieStack = kTieSubtree / 17 + 4;
template <bool LDS>
__device__ __forceinline__ void tie_subtree(unsigned* BITS, unsigned* IDX, unsigned* LPOS, unsigned* RPOS, int first, int last,
                                            int depth, unsigned* mark, int* s_fail, int* stack, unsigned* area) {
    const int lane = threadIdx.x & 63;
    unsigned *B = BITS, *I = IDX, *L = LPOS, *R = RPOS;
    int off = 0;                                                   // position in the image = position here + off
    if constexpr (!LDS) {
        const TieArr<false> gb = tie_arr<false>(BITS), gi = tie_arr<false>(IDX);
        B = area; I = area + kTieSubtree; L = I + kTieSubtree; R = L + kTieSubtree;
        off = first;
        for (int j = lane; j < last - first; j += 64) { B[j] = gb(first + j); I[j] = gi(first + j); }
        tie_wave_fence<true>();
    }
    int top = 0;
    if (lane == 0) { stack[0] = first - off; stack[1] = last - off; stack[2] = depth; }
    top = 1;
    tie_wave_fence<true>();
    while (top > 0) {
        top--;
        const int f = stack[3 * top], l = stack[3 * top + 1], d = stack[3 * top + 2];
        tie_wave_fence<true>();                                    // (every lane has read the entry before it is overwritten)
        if (d == 0) {
            // the heapsort branch (std::__partial_sort) -- it only matters where it has to order equal scores: a segment without
            // a tied score ends up sorted whatever sorts it, i.e. as the first sort left it
            bool tied = false;
            for (int j = f + lane; j < l; j += 64) tied |= (I[j] & kTiedBit) != 0u;
            if (__ballot(tied) == 0ull) continue;
            if (lane == 0) tie_heapsort<true>(B, I, f, l);
            tie_wave_fence<true>();
            for (int j = f + lane; j < l; j += 64) { const int c = j + off; atomicOr(&mark[c >> 5], 1u << (c & 31)); }   // sorted: every element a segment of its own
            continue;
        }
        const int cut = tie_partition<true>(B, I, L, R, f, l);
        if (cut == -2) continue;
        if (cut < 0) { if (lane == 0) *s_fail = 1; break; }
        if (lane == 0) {
            if (cut < l) { const int c = cut + off; atomicOr(&mark[c >> 5], 1u << (c & 31)); }
            int t = top;
            if (cut - f > 16) { stack[3 * t] = f; stack[3 * t + 1] = cut; stack[3 * t + 2] = d - 1; t++; }
            if (l - cut > 16) { stack[3 * t] = cut; stack[3 * t + 1] = l; stack[3 * t + 2] = d - 1; t++; }
        }
        top += (cut - f > 16 ? 1 : 0) + (l - cut > 16 ? 1 : 0);
        tie_wave_fence<true>();
    }
    if constexpr (!LDS) {
        tie_wave_fence<true>();
        for (int j = lane; j < last - first; j += 64) { BITS[first + j] = B[j]; IDX[first + j] = I[j]; }
        tie_wave_fence<false>();
    }
}

// __introsort_loop(0, n): the segments of one recursion level in `cur`, their children in `nxt`; one wave per segment.
// `mark`: one bit per position, set where a partition cut its segment (and at 0): the segments of at most 16 elements
// the loop leaves to the insertion sort lie between two marks.
template <bool LDS, int NT>
__device__ __forceinline__ void tie_levels(unsigned* BITS, unsigned* IDX, unsigned* LPOS, unsigned* RPOS, int2* cur, int2* nxt,
                                           int n, int* s_next, int* s_fail, unsigned* mark, int* coop, int* stacks,
                                           unsigned* areas) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int depth = 2 * (31 - __clz(n));                               // std::__lg(n) * 2
    int n_cur = 0;
    if (tid == 0) atomicOr(&mark[0], 1u);
    if (n > 16) { if (tid == 0) cur[0] = make_int2(0, n); n_cur = 1; }
    tie_group_sync<LDS>();
    while (n_cur > 0) {
        if (depth == 0) {
            // the heapsort branch (std::__partial_sort) for the segments that are left and hold a tied score (the others end up
            // sorted whatever sorts them): one wave per segment, its first lane sorts
            const TieArr<LDS> IDXA = tie_arr<LDS>(IDX);
            for (int s = wave; s < n_cur; s += NT / 64) {
                const int2 sg = cur[s];
                bool tied = false;
                for (int j = sg.x + lane; j < sg.y; j += 64) tied |= (IDXA(j) & kTiedBit) != 0u;
                if (__ballot(tied) == 0ull) continue;
                if (lane == 0) tie_heapsort<LDS>(BITS, IDX, sg.x, sg.y);
                tie_wave_fence<LDS>();
                for (int j = sg.x + lane; j < sg.y; j += 64) atomicOr(&mark[j >> 5], 1u << (j & 31));   // sorted: every element a segment of its own
            }
            tie_group_sync<LDS>();
            break;
        }
        depth--;
        auto children = [&](const int2 sg, int cut) {               // (one thread)
            if (cut < sg.y) atomicOr(&mark[cut >> 5], 1u << (cut & 31));
            if (cut - sg.x > 16) nxt[atomicAdd(s_next, 1)] = make_int2(sg.x, cut);
            if (sg.y - cut > 16) nxt[atomicAdd(s_next, 1)] = make_int2(cut, sg.y);
        };
        for (int s = 0; s < n_cur; s++) {                          // the long ones: the whole workgroup on each
            const int2 sg = cur[s];
            if (sg.y - sg.x <= tie_coop_len<LDS>()) continue;
            const int cut = tie_partition_block<LDS, NT>(BITS, IDX, LPOS, RPOS, sg.x, sg.y, coop);
            if (cut == -2) continue;
            if (cut < 0) { if (tid == 0) *s_fail = 1; continue; }
            if (tid == 0) children(sg, cut);
        }
        int mine = 0;                                              // the others: one wave per segment
        for (int s = 0; s < n_cur; s++) {
            const int2 sg = cur[s];
            if (sg.y - sg.x > tie_coop_len<LDS>()) continue;
            if ((mine++ % (NT / 64)) != wave) continue;
            if (sg.y - sg.x <= (LDS ? kTieSubtreeLds : kTieSubtree)) {   // short enough: this wave finishes it, all levels
                tie_subtree<LDS>(BITS, IDX, LPOS, RPOS, sg.x, sg.y, depth + 1, mark, s_fail, stacks + wave * 3 * kTieStack,
                                 areas + (size_t)wave * 4 * kTieSubtree);
                continue;
            }
            const int cut = tie_partition<LDS>(BITS, IDX, LPOS, RPOS, sg.x, sg.y);
            if (cut == -2) continue;                               // no tied score in it: nobody asks where its seeds end up
            if (cut < 0) { if (lane == 0) *s_fail = 1; continue; }
            if (lane == 0) children(sg, cut);
        }
        tie_group_sync<LDS>();
        n_cur = *s_next;
        __syncthreads();
        if (tid == 0) *s_next = 0;
        int2* t2 = cur; cur = nxt; nxt = t2;
        tie_group_sync<LDS>();
    }
}

// The tie pass of image `b` by the NT threads of a workgroup.  `tie_lds`: tie_lds_bytes<NT>() of LDS, 16-byte aligned (four arrays
// of kTieLdsKeys words, then the small state).  Called by cifseeds_tie_kernel (stage-level entry points) and, inside the
// decode, by the association kernel before it touches the image's seeds: an image pays for its own ties only.
template <int NT> struct TieSmall {
    int flag, next, fail, pad;
    int wave_tot[NT / 64];
    int2 seg[2][kTieLdsKeys / 17 + 2];                 // the two segment lists of an image that lives in LDS
    unsigned mark[kTieLdsKeys / 32];
    int coop[2 * (NT / 64) + 2];
    int stack[(NT / 64) * 3 * kTieStack];
};
template <int NT> __host__ __device__ constexpr size_t tie_lds_bytes() { return 4 * kTieLdsKeys * sizeof(unsigned) + ((sizeof(TieSmall<NT>) + 15) & ~(size_t)15); }

template <int NT>
__device__ __forceinline__ void cifseeds_tie_body(const TieArgs& a, const SortArgs& g, const DevParams& p, int b, unsigned char* tie_lds) {
    TieSmall<NT>& sm = *reinterpret_cast<TieSmall<NT>*>(tie_lds + 4 * kTieLdsKeys * sizeof(unsigned));
    int& s_flag = sm.flag; int& s_next = sm.next; int& s_fail = sm.fail;
    int* s_wave_tot = sm.wave_tot; int2 (*s_seg)[kTieLdsKeys / 17 + 2] = sm.seg; unsigned* s_mark = sm.mark;
    int* s_coop = sm.coop; int* s_stack = sm.stack;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int n = g.seed_count[b];
    if (n > g.cap) n = g.cap;
    const bool in_lds = n <= kTieLdsKeys;
    const int cells = a.cells;
    unsigned *BITS, *IDX, *LPOS, *RPOS;
    if (in_lds) {
        BITS = (unsigned*)tie_lds; IDX = BITS + kTieLdsKeys; LPOS = IDX + kTieLdsKeys; RPOS = LPOS + kTieLdsKeys;
    } else {
        IDX = (unsigned*)(a.big + (size_t)b * a.big_stride); BITS = IDX + cells; LPOS = BITS + cells; RPOS = LPOS + cells;
    }
    if (tid == 0) { s_flag = 0; s_next = 0; s_fail = 0; }
    __syncthreads();
    // ---- equal neighbours in the sorted scores?  (An image in LDS keeps the sorted scores: SV, in the LPOS array.)
    const int ncol = g.NC - 1;
    const float* sv = g.seed_vxys + (size_t)b * g.cap * ncol;
    unsigned* SV = LPOS;
    {
        bool any = false;
        if (in_lds) {
            for (int t = tid; t < n; t += NT) SV[t] = sortable_bits(sv[(size_t)t * ncol]);
            __syncthreads();
            for (int t = tid; t + 1 < n; t += NT) any |= SV[t] == SV[t + 1];
        } else {
            for (int t = tid; t + 1 < n; t += NT) any |= sv[(size_t)t * ncol] == sv[(size_t)(t + 1) * ncol];
        }
        if (any) s_flag = 1;
    }
    __syncthreads();
    if (a.tie_state && tid == 0) a.tie_state[b] = s_flag;
    if (!s_flag) return;

    const int E = tie_blocks(g.F, g.HW);
    unsigned char* sp = a.small_ + (size_t)b * a.small_stride;
    const int2* TAB = (const int2*)sp;                          // (begin, length) of the key block of (field, chunk)
    int* PRE = (int*)(sp + (size_t)E * sizeof(int2));           // seeds in the blocks before it
    sp += ((size_t)E * (sizeof(int2) + sizeof(int)) + 15) & ~(size_t)15;
    int2* seg_a = (int2*)sp; int2* seg_b = seg_a + tie_seg_cap(cells);
    const unsigned long long* K = tie_key_copy(a.big + (size_t)b * a.big_stride, cells);   // the image's keys, block by block as the fill kernel wrote them

    // ---- raster position of every seed: exclusive prefix of the block lengths in (field, chunk) order ...
    {
        const int per = (E + NT - 1) / NT;
        const int e0 = min(E, tid * per), e1 = min(E, e0 + per);
        int mine = 0;
        for (int e = e0; e < e1; e++) mine += TAB[e].y;
        int incl = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d, 64); if (lane >= d) incl += o; }
        if (lane == 63) s_wave_tot[wave] = incl;
        __syncthreads();
        int run = incl - mine;
        for (int k = 0; k < wave; k++) run += s_wave_tot[k];
        for (int e = e0; e < e1; e++) { PRE[e] = run; run += TAB[e].y; }
    }
    sync_global();
    // An image beyond the LDS arrays: the scores that occur twice, in descending order, into the (otherwise unused) LDS
    // area -- its elements look themselves up there; more than the area holds: every segment is followed.
    unsigned* TV = (unsigned*)tie_lds;
    constexpr int kTvCap = 4 * kTieLdsKeys;
    int n_tv = 0;
    if (!in_lds) {
        const int per = (n + NT - 1) / NT;
        const int r0 = min(n, tid * per), r1 = min(n, r0 + per);
        auto first_of_group = [&](int t) {
            const float v = sv[(size_t)t * ncol];
            return t + 1 < n && sv[(size_t)(t + 1) * ncol] == v && (t == 0 || sv[(size_t)(t - 1) * ncol] != v);
        };
        int mine = 0;
        for (int t = r0; t < r1; t++) mine += first_of_group(t) ? 1 : 0;
        int incl = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d, 64); if (lane >= d) incl += o; }
        __syncthreads();                                           // (s_wave_tot was read above)
        if (lane == 63) s_wave_tot[wave] = incl;
        __syncthreads();
        int run = incl - mine;
        for (int k = 0; k < NT / 64; k++) { if (k < wave) run += s_wave_tot[k]; n_tv += s_wave_tot[k]; }
        if (n_tv <= kTvCap)
            for (int t = r0; t < r1; t++)
                if (first_of_group(t)) TV[run++] = sortable_bits(sv[(size_t)t * ncol]);
        __syncthreads();
    }
    // ... and every key goes to (seeds before its block) + (its offset in the block); its block follows from its cell
    const int chunks = E / g.F;
    for (int t0 = tid; t0 < n; t0 += 4 * NT) {
        unsigned long long key[4]; int pre[4], beg[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { const int t = t0 + u * NT; key[u] = K[t < n ? t : 0]; }
#pragma unroll
        for (int u = 0; u < 4; u++) asm volatile("" : "+v"(key[u]) :: "memory");
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const unsigned idx = 0xFFFFFFFFu - (unsigned)(key[u] & 0xFFFFFFFFull);
            const int f = (int)(idx / (unsigned)g.HW), o = (int)(idx - (unsigned)f * (unsigned)g.HW);
            const int e = f * chunks + o / (256 * kFillCells);
            pre[u] = PRE[e]; beg[u] = TAB[e].x;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) asm volatile("" : "+v"(pre[u]), "+v"(beg[u]) :: "memory");
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int t = t0 + u * NT;
            if (t < n) {
                const unsigned idx = 0xFFFFFFFFu - (unsigned)(key[u] & 0xFFFFFFFFull);
                const unsigned bits = (unsigned)(key[u] >> 32);
                // does the score occur twice?  An image in LDS looks it up in its sorted scores, a larger one in the list
                unsigned tied = kTiedBit;
                if (in_lds) {
                    int lo = 0, hi = n;                            // first index with SV[i] <= bits (descending)
                    while (lo < hi) { const int mid = (lo + hi) >> 1; if (SV[mid] > bits) lo = mid + 1; else hi = mid; }
                    tied = (lo + 1 < n && SV[lo + 1] == bits) ? kTiedBit : 0u;
                } else if (n_tv <= kTvCap) {
                    int lo = 0, hi = n_tv;
                    while (lo < hi) { const int mid = (lo + hi) >> 1; if (TV[mid] > bits) lo = mid + 1; else hi = mid; }
                    tied = (lo < n_tv && TV[lo] == bits) ? kTiedBit : 0u;
                }
                const int pos = pre[u] + (t - beg[u]);
                BITS[pos] = bits; IDX[pos] = idx | tied;
            }
        }
    }
    unsigned* mark = in_lds ? s_mark : (unsigned*)(seg_b + tie_seg_cap(cells));   // (behind the segment lists: n / 32 words)
    tie_group_sync_rt(in_lds);
    for (int w = tid; w < (n + 31) / 32; w += NT) mark[w] = 0u;
    tie_group_sync_rt(in_lds);

    // ---- __introsort_loop, one level of the recursion at a time
    if (in_lds) tie_levels<true, NT>(BITS, IDX, LPOS, RPOS, s_seg[0], s_seg[1], n, &s_next, &s_fail, mark, s_coop, s_stack, nullptr);
    else tie_levels<false, NT>(BITS, IDX, LPOS, RPOS, seg_a, seg_b, n, &s_next, &s_fail, mark, s_coop, s_stack, (unsigned*)tie_lds);
    __syncthreads();
    if (s_fail) {                                                  // the seeds stay as the first sort left them
        if (tid == 0 && a.tie_state) a.tie_state[b] = -1;
        return;
    }

    // ---- __final_insertion_sort: a tied element ends up behind the larger and the equal-and-earlier elements of its
    //      segment (between the mark at or before it and the next one, at most 16 positions)
    auto mk = [&](int w) { return in_lds ? mark[w] : __hip_atomic_load(&mark[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    auto rd = [&](const unsigned* q, int i) { return in_lds ? q[i] : __hip_atomic_load(q + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    for (int k = tid; k < n; k += NT) {
        const unsigned cell = rd(IDX, k);
        if (!(cell & kTiedBit)) continue;
        const unsigned at_or_below = mk(k >> 5) & (0xFFFFFFFFu >> (31 - (k & 31)));
        int ls;
        if (at_or_below) ls = (k & ~31) + 31 - __clz(at_or_below);
        else ls = (k & ~31) - 32 + 31 - __clz(mk((k >> 5) - 1));
        int le = ls + 16 < n ? ls + 16 : n;
        for (int j = k + 1; j < le; j++)
            if ((mk(j >> 5) >> (j & 31)) & 1u) { le = j; break; }
        const unsigned mine = rd(BITS, k);
        int pos = ls;
        for (int j = ls; j < le; j++) {
            const unsigned o = rd(BITS, j);
            pos += (o > mine || (o == mine && j < k)) ? 1 : 0;
        }
        store_seed(((unsigned long long)mine << 32) | (unsigned long long)(0xFFFFFFFFu - (cell & ~kTiedBit)), pos, b, g.cif, g.F,
                   g.NC, g.HW, g.stride, g.cap, g.seed_f, g.seed_vxys, g.seed_cell, g.occ_h, g.occ_w, p);
    }
}

}  // namespace opa
