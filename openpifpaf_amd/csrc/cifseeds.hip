// CifSeeds: seed extraction and ordering on gfx950.
//
// Replaces reference CifSeeds::fill + CifSeeds::get (csrc/src/cif_seeds.cpp:33-66,
// 93-114): threshold every CIF cell, rescore it with the high-resolution map
// (0.9*hr + 0.1*c), threshold again, and return the survivors sorted by score.
//
//  cifseeds_fill_kernel  four CIF cells per thread (coalesced plane reads, one
//      gather into the L2-resident CifHr map), survivors appended to the image's
//      key array with ONE atomic per workgroup (ballot + popcount prefix, wave totals in LDS).
//      key = sortable(score) << 32 | ~cell_index, so that a descending key sort is
//      "score descending, then cell index ascending" -- a total order, which makes
//      the result independent of the append order and of the sort algorithm.
//      (The reference's std::sort leaves the order of equal scores unspecified.)
//  cifseeds_sort_kernel  one 1024-thread workgroup per image: bitonic sort of the
//      u64 keys, entirely in LDS when n <= 8192 (64 KiB), otherwise LDS-blocked
//      with the far strides done through L2.  The epilogue decodes the keys and
//      writes the sorted (f, v, x, y, s) arrays the association kernel streams.
#include "cafscored_impl.hpp"


namespace opa {

__device__ __forceinline__ unsigned sortable_bits(float v) {
    const unsigned u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float from_sortable(unsigned s) {
    return __uint_as_float((s & 0x80000000u) ? (s & 0x7fffffffu) : ~s);
}

constexpr int kFillCells = 4;            // CIF cells per thread: their confidence loads are in flight together

__global__ __launch_bounds__(256) void cifseeds_fill_kernel(
        const float* __restrict__ cif, int F, int NC, int H, int W, int stride,
        const float* __restrict__ cifhr, int hr_rows, int hr_cols, int hr_pitch,
        double threshold, int ablation_nms, int no_rescore,
        unsigned long long* __restrict__ keys, int sort_cap, int cap, int32_t* __restrict__ seed_count,
        int2* __restrict__ wg_tab, size_t tab_stride, unsigned long long* __restrict__ key_copy, size_t copy_stride) {
    const int HW = H * W;
    const int plane = blockIdx.x;              // b*F + f
    const int b = plane / F, f = plane - b * F;
    const int lane = threadIdx.x & 63;
    const float* P = cif + (size_t)plane * NC * HW;
    int o[kFillCells]; float c[kFillCells], xin[kFillCells], yin[kFillCells]; bool on[kFillCells];
#pragma unroll
    for (int r = 0; r < kFillCells; r++) {                           // confidence and position of every cell in one round trip
        o[r] = (blockIdx.y * kFillCells + r) * 256 + threadIdx.x;
        const int oo = o[r] < HW ? o[r] : 0;
        c[r] = o[r] < HW ? P[HW + oo] : -1.0f;
        xin[r] = P[2 * HW + oo]; yin[r] = P[3 * HW + oo];
    }
#pragma unroll
    for (int r = 0; r < kFillCells; r++) {
        on[r] = false;
        if (o[r] < HW && !((double)c[r] < threshold)) {              // cif_seeds.cpp:47
            on[r] = true;
            if (ablation_nms) {                                      // :35-40,49-51: 3x3 max-pool gate
                const int j = o[r] / W, i = o[r] - j * W;
                float m = c[r];
                for (int dj = -1; dj <= 1; dj++) for (int di = -1; di <= 1; di++) {
                    const int jj = j + dj, ii = i + di;
                    if (jj < 0 || jj >= H || ii < 0 || ii >= W) continue;
                    m = fmaxf(m, P[HW + jj * W + ii]);
                }
                if (c[r] < m) on[r] = false;
            }
            if (on[r]) {
                const float x = xin[r] * (float)stride;              // :53-54
                const float y = yin[r] * (float)stride;
                if (!no_rescore) {                                   // :56-58
                    const float hv = cifhr_value(cifhr + (size_t)b * F * hr_rows * hr_pitch,
                                                 F, hr_rows, hr_cols, hr_pitch, f, x, y, -1.0f);
                    c[r] = (float)(0.9 * (double)hv + 0.1 * (double)c[r]);
                }
                if ((double)c[r] < threshold) on[r] = false;         // :59
            }
        }
    }
    // Survivors are appended to the image's key array.  ONE atomic per workgroup: 15 000 per-wave atomics on the 32
    // per-image counters of a batch serialise at the L2 (~470 per address) and were most of this kernel's time.
    // Inside the workgroup's block the keys stand in raster order (r, wave, lane -- the order of the cells), and
    // `wg_tab` remembers where the block of (field, chunk) begins and how long it is: with that (and a copy of the keys
    // the sort does not touch) the tie pass lays the image's seeds out in the order the reference pushes them
    // (cif_seeds.cpp:41-65) without sorting them by cell.
    __shared__ int wave_total[kFillCells][4];
    __shared__ int wg_base;
    unsigned long long mask[kFillCells];
    const int w = threadIdx.x >> 6;
#pragma unroll
    for (int r = 0; r < kFillCells; r++) { mask[r] = __ballot(on[r]); if (lane == 0) wave_total[r][w] = __popcll(mask[r]); }
    __syncthreads();
    int total = 0, before[kFillCells];
#pragma unroll
    for (int r = 0; r < kFillCells; r++) {
        before[r] = total;
#pragma unroll
        for (int k = 0; k < 4; k++) { if (k < w) before[r] += wave_total[r][k]; total += wave_total[r][k]; }
    }
    if (threadIdx.x == 0) {
        wg_base = total ? atomicAdd(&seed_count[b], total) : 0;
        if (wg_tab) wg_tab[(size_t)b * tab_stride + (size_t)f * gridDim.y + blockIdx.y] = make_int2(wg_base, total);
    }
    if (total == 0) return;                          // (uniform over the workgroup)
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kFillCells; r++) {
        if (on[r]) {
            const int slot = wg_base + before[r] + __popcll(mask[r] & ((1ull << lane) - 1ull));
            if (slot < cap) {
                const unsigned idx = (unsigned)(f * HW + o[r]);
                const unsigned long long key = ((unsigned long long)sortable_bits(c[r]) << 32) | (unsigned long long)(0xFFFFFFFFu - idx);
                keys[(size_t)b * sort_cap + slot] = key;
                if (key_copy) key_copy[(size_t)b * copy_stride + slot] = key;   // (the sort reorders `keys`; the tie pass wants the blocks)
            }
        }
    }
}

__device__ __forceinline__ void compare_exchange_desc(unsigned long long& a, unsigned long long& b, bool desc) {
    if ((a < b) == desc) { const unsigned long long t = a; a = b; b = t; }
}

// n <= 8192 keys: a block-wide LSD radix sort (rocPRIM's block primitive, 8 keys per thread, 64 KB LDS) over the
// bits that can differ -- the 32 score bits and as many index bits as the field has cells -- instead of the 91
// compare-exchange passes of a bitonic network.  The key order (score descending, then cell index ascending) is a
// total order, so the result does not depend on the algorithm.
// (Measured on the bench batch: 72.7 us against 74.0 us for the bitonic network below -- the kernel is bound by the
// busiest image's serial passes either way -- so the hand-written network stays the default; -DOPA_SORT_RADIX.)

// Images with more than 8192 seeds (wholebody: 133 fields, ~20 000 seeds): every 8192-key block is sorted by a
// workgroup of its own (kSortBlocksMax per image in the sort kernel's grid), then cifseeds_rankmerge_kernel gives every
// key its final position -- its rank in its own block plus, by binary search, the number of larger keys in each
// other block (keys are unique: the cell index is part of them) -- and writes the decoded seed there.  One workgroup
// per image bitonic-merging the blocks through L2 took 430 us for 16 wholebody images.  Beyond kSortBlocksMax
// blocks (all-active fields) the single-workgroup network below still does it.
constexpr int kSortBlocksMax = 8;
constexpr int kSortSmallBlock = 2048;
__host__ __device__ inline int sort_block_size(int n) { return n <= kSortLdsKeys ? kSortSmallBlock : kSortLdsKeys; }

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

struct SortArgs {
    unsigned long long* keys; int sort_cap, cap; const int32_t* seed_count;
    const float* cif; int F, NC, HW, stride;
    int32_t* seed_f; float* seed_vxys; int32_t* seed_cell; int occ_h, occ_w;
};

#ifdef OPA_SORT_RADIX
#error "the rocPRIM variant predates the block split (round 2 experiment)"
#endif

// the sort of one workgroup (block `block` of the grid's sort part); `sk`: 64 KiB of LDS
__device__ __forceinline__ void cifseeds_sort_body(const SortArgs& g, const DevParams& p, int block, unsigned long long* sk) {
    unsigned long long* keys = g.keys; const int sort_cap = g.sort_cap, cap = g.cap;
    const int32_t* __restrict__ seed_count = g.seed_count; const float* __restrict__ cif = g.cif;
    const int F = g.F, NC = g.NC, HW = g.HW, stride = g.stride, occ_h = g.occ_h, occ_w = g.occ_w;
    int32_t* __restrict__ seed_f = g.seed_f; float* __restrict__ seed_vxys = g.seed_vxys; int32_t* __restrict__ seed_cell = g.seed_cell;
    const int b = block / kSortBlocksMax, part = block - b * kSortBlocksMax, tid = threadIdx.x;
    unsigned long long* K = keys + (size_t)b * sort_cap;
    int n = seed_count[b];
    if (n > cap) n = cap;
    int n_pad = 2;
    while (n_pad < n) n_pad <<= 1;
    // one workgroup per block of BS keys + the rank merge: 2048-key blocks for up to 8192 seeds (66 compare-exchange
    // passes on a quarter of the keys instead of 91 on all of them), 8192-key blocks beyond
    const int BS = sort_block_size(n);
    const bool in_lds = n <= BS;                       // one block: sorted and decoded here
    const bool split = !in_lds && n <= kSortBlocksMax * BS;
    if (part != 0 && !split) return;
    if (split && part * BS >= n) return;
    if (in_lds) n_pad = n_pad < 2 ? 2 : n_pad;

    // strides j_first, j_first/2, ..., 1 of stage k over the 8192 keys in LDS (directions come from the key's
    // GLOBAL index blk + i).  Like the in-LDS sort below: wave w owns keys [512 w, 512 w + 512), strides
    // below 512 stay inside a wave (two per LDS round trip, no workgroup barrier), only 512..4096 synchronise.
    auto lds_strides = [&](int blk, int k, int j_first) {
        constexpr int EB = kSortLdsKeys / 16;
        const int wave = tid >> 6, lane = tid & 63;
        bool synced = true;
        for (int jj = j_first; jj > 0;) {
            if (jj >= EB) {
                if (!synced) __syncthreads();
                for (int t = tid; t < (kSortLdsKeys >> 1); t += 1024) {
                    const int i = ((t & ~(jj - 1)) << 1) | (t & (jj - 1));
                    const int p = i | jj;
                    unsigned long long a = sk[i], c = sk[p];
                    compare_exchange_desc(a, c, ((blk + i) & k) == 0);
                    sk[i] = a; sk[p] = c;
                }
                __syncthreads();
                synced = true;
                jj >>= 1;
                continue;
            }
            if (jj >= 2) {
                const int h = jj >> 1;
                for (int q = lane; q < (EB >> 2); q += 64) {
                    const int i0 = wave * EB + (((q & ~(h - 1)) << 2) | (q & (h - 1)));
                    const bool desc = ((blk + i0) & k) == 0;
                    unsigned long long v0 = sk[i0], v1 = sk[i0 | h], v2 = sk[i0 | jj], v3 = sk[i0 | jj | h];
                    compare_exchange_desc(v0, v2, desc); compare_exchange_desc(v1, v3, desc);
                    compare_exchange_desc(v0, v1, desc); compare_exchange_desc(v2, v3, desc);
                    sk[i0] = v0; sk[i0 | h] = v1; sk[i0 | jj] = v2; sk[i0 | jj | h] = v3;
                }
                jj >>= 2;
            } else {
                for (int q = lane; q < (EB >> 1); q += 64) {
                    const int i = wave * EB + 2 * q;
                    unsigned long long a = sk[i], c = sk[i | 1];
                    compare_exchange_desc(a, c, ((blk + i) & k) == 0);
                    sk[i] = a; sk[i | 1] = c;
                }
                jj >>= 1;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            synced = false;
        }
        __syncthreads();
    };
    // bitonic sort (descending) of sk[0, m), m a power of two, in LDS
    auto sort_lds = [&](const int m) {
        // Wave w owns the contiguous block of E = m / 16 keys: every pass whose stride stays inside a
        // block (j < E) touches only keys this wave wrote, so it needs no workgroup barrier -- for 8192 keys
        // that is 81 of the 91 passes.  Only the far strides (j >= E) synchronise the workgroup.
        const int E = m >= 32 ? m / 16 : m;      // tiny inputs: wave 0 does everything
        const int wave = tid >> 6, lane = tid & 63;
        const bool owner = m >= 32 || wave == 0;
        bool synced = true;                                  // a workgroup barrier separates us from the last local pass
        for (int k = 2; k <= m; k <<= 1) {
            for (int j = k >> 1; j > 0;) {
                if (j >= E) {                                // far stride: any thread, any pair
                    if (!synced) __syncthreads();
                    for (int t = tid; t < (m >> 1); t += 1024) {
                        const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                        const int p = i | j;
                        unsigned long long a = sk[i], c = sk[p];
                        compare_exchange_desc(a, c, (i & k) == 0);
                        sk[i] = a; sk[p] = c;
                    }
                    __syncthreads();
                    synced = true;
                    j >>= 1;
                    continue;
                }
                if (j >= 2) {                                // two strides (j, j/2) per LDS round trip: 4 keys in registers
                    const int h = j >> 1;
                    if (owner) {
                        for (int q = lane; q < (E >> 2); q += 64) {
                            const int i0 = wave * E + (((q & ~(h - 1)) << 2) | (q & (h - 1)));
                            const bool desc = (i0 & k) == 0;
                            unsigned long long v0 = sk[i0], v1 = sk[i0 | h], v2 = sk[i0 | j], v3 = sk[i0 | j | h];
                            compare_exchange_desc(v0, v2, desc); compare_exchange_desc(v1, v3, desc);   // stride j
                            compare_exchange_desc(v0, v1, desc); compare_exchange_desc(v2, v3, desc);   // stride j/2
                            sk[i0] = v0; sk[i0 | h] = v1; sk[i0 | j] = v2; sk[i0 | j | h] = v3;
                        }
                    }
                    j >>= 2;
                } else {                                     // last single stride of the stage
                    if (owner) {
                        for (int q = lane; q < (E >> 1); q += 64) {
                            const int i = wave * E + (((q & ~(j - 1)) << 1) | (q & (j - 1)));
                            const int p = i | j;
                            unsigned long long a = sk[i], c = sk[p];
                            compare_exchange_desc(a, c, (i & k) == 0);
                            sk[i] = a; sk[p] = c;
                        }
                    }
                    j >>= 1;
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                synced = false;
            }
        }
        __syncthreads();
    };
    if (split) {            // this workgroup's block, sorted descending on its own; the rank merge kernel does the rest
        const int blk = part * BS;
        for (int t = tid; t < BS; t += 1024) sk[t] = blk + t < n ? K[blk + t] : 0ull;
        __syncthreads();
        sort_lds(BS);
        for (int t = tid; t < BS; t += 1024) K[blk + t] = sk[t];
        return;
    }

    if (in_lds) {
        for (int t = tid; t < n_pad; t += 1024) sk[t] = t < n ? K[t] : 0ull;
        __syncthreads();
        sort_lds(n_pad);
    } else {
        for (int t = n + tid; t < n_pad; t += 1024) K[t] = 0ull;
        sync_global();                                              // keys travel through HBM between threads here
        // stages k <= 8192 never leave a block: ALL of them in one LDS visit per block
        for (int blk = 0; blk < n_pad; blk += kSortLdsKeys) {
            for (int t = tid; t < kSortLdsKeys; t += 1024) sk[t] = K[blk + t];
            __syncthreads();
            for (int k = 2; k <= kSortLdsKeys; k <<= 1) lds_strides(blk, k, k >> 1);
            for (int t = tid; t < kSortLdsKeys; t += 1024) K[blk + t] = sk[t];
            __syncthreads();
        }
        sync_global();
        // stages k > 8192: far strides through L2, then the strides below 8192 in one LDS visit per block
        for (int k = 2 * kSortLdsKeys; k <= n_pad; k <<= 1) {
            for (int j = k >> 1; j >= kSortLdsKeys; j >>= 1) {
                for (int t = tid; t < (n_pad >> 1); t += 1024) {
                    const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                    const int p = i | j;
                    unsigned long long a = K[i], c = K[p];
                    compare_exchange_desc(a, c, (i & k) == 0);
                    K[i] = a; K[p] = c;
                }
                sync_global();
            }
            for (int blk = 0; blk < n_pad; blk += kSortLdsKeys) {
                for (int t = tid; t < kSortLdsKeys; t += 1024) sk[t] = K[blk + t];
                __syncthreads();
                lds_strides(blk, k, kSortLdsKeys >> 1);
                for (int t = tid; t < kSortLdsKeys; t += 1024) K[blk + t] = sk[t];
                __syncthreads();
            }
            sync_global();
        }
    }

    // epilogue: decode keys -> sorted seeds (cif_seeds.cpp:100-113)
    for (int t = tid; t < n; t += 1024)
        store_seed(in_lds ? sk[t] : K[t], t, b, cif, F, NC, HW, stride, cap, seed_f, seed_vxys, seed_cell, occ_h, occ_w, p);
}

__global__ __launch_bounds__(1024) void cifseeds_sort_kernel(SortArgs g, DevParams p) {
    __shared__ unsigned long long sk[kSortLdsKeys];
    cifseeds_sort_body(g, p, blockIdx.x, sk);
}

// The decode's form: the same launch also builds the CAF lists (CafScored::fill).  Seed sort and list building only
// share the finished CifHr map; the sort keeps a handful of workgroups busy for ~40 us (its passes are serial), the list
// building streams 128 MB through all of the chip in about the same time -- side by side instead of one after the other.
// Blocks [0, n_sort): sort; behind them, two 512-thread groups per workgroup, one (image, CAF field) plane each, list set
// 0 first, then list set 1 (force complete).
__global__ __launch_bounds__(1024) void cifseeds_sort_scored_kernel(SortArgs g, DevParams p, int n_sort, ScoredArgs s0, ScoredArgs s1,
                                                                    int wgs0) {
    __shared__ unsigned long long sk[kSortLdsKeys];
    if ((int)blockIdx.x < n_sort) { cifseeds_sort_body(g, p, blockIdx.x, sk); return; }
    __shared__ int wave_tot[2][2][kScoredThreads / 64];
    extern __shared__ float bb_dyn[];
    const int wg = blockIdx.x - n_sort, group = threadIdx.x >> 9, tid = threadIdx.x & (kScoredThreads - 1);
    if (wg < wgs0) cafscored_plane(s0, 2 * wg + group, tid, wave_tot[group], bb_dyn + group * 2 * s0.nb * 4);
    else cafscored_plane(s1, 2 * (wg - wgs0) + group, tid, wave_tot[group], bb_dyn + group * 2 * s1.nb * 4);
}

// final position of every key of a block-sorted image (see kSortBlocksMax) + the decoded seed
__global__ __launch_bounds__(256) void cifseeds_rankmerge_kernel(
        const unsigned long long* __restrict__ keys, int sort_cap, int cap, const int32_t* __restrict__ seed_count,
        const float* __restrict__ cif, int F, int NC, int HW, int stride,
        int32_t* __restrict__ seed_f, float* __restrict__ seed_vxys,
        int32_t* __restrict__ seed_cell, int occ_h, int occ_w, DevParams p) {
    const int b = blockIdx.y;
    int n = seed_count[b];
    if (n > cap) n = cap;
    const int BS = sort_block_size(n);
    if (n <= BS || n > kSortBlocksMax * BS) return;    // the sort kernel did (or does) it all
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= n) {
        // (positions behind the keys of the last block hold its zero padding: nothing to place) -- but a key's
        // position inside its block can exceed n only in the last block, whose keys sit before the padding
        return;
    }
    const unsigned long long* K = keys + (size_t)b * sort_cap;
    const int blk = t / BS;
    const unsigned long long key = K[t];
    int pos = t - blk * BS;                            // larger keys of its own block
    const int nblk = (n + BS - 1) / BS;
    for (int o = 0; o < nblk; o++) {
        if (o == blk) continue;
        const unsigned long long* Ko = K + (size_t)o * BS;
        int lo = 0, hi = min(BS, n - o * BS);          // first index whose key is smaller (descending block)
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (Ko[mid] > key) lo = mid + 1; else hi = mid;
        }
        pos += lo;
    }
    store_seed(key, pos, b, cif, F, NC, HW, stride, cap, seed_f, seed_vxys, seed_cell, occ_h, occ_w, p);
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
// Heapsort (the depth limit, 2 log2 n levels) is not reproduced: such an image keeps the cell-index order and says so
// in `tie_state` (-1).  Images without ties leave after one look at their sorted scores.
constexpr int kTieLdsKeys = 8192;
constexpr int kTieThreads = 1024;
constexpr unsigned kTiedBit = 0x80000000u;      // in a cell index: the seed's score occurs more than once

struct TieArgs {
    int cells;                       // F * HW: capacity of the per-image arrays
    unsigned char* big; size_t big_stride;       // per image tie_big_bytes(cells): cells, scores and the two stop lists of an
                                                 // image beyond the LDS arrays
    unsigned char* small_; size_t small_stride;  // per image tie_small_bytes(F, HW): the fill kernel's block table, its prefix,
                                                 // segment lists
    int32_t* tie_state;              // [B] or null: 0 no equal scores, 1 re-sorted in libstdc++'s order, -1 not reproduced
};

__host__ __device__ inline size_t tie_seg_cap(int cells) { return (size_t)cells / 17 + 2; }
__host__ __device__ inline int tie_blocks(int F, int HW) { return F * ((HW + 256 * kFillCells - 1) / (256 * kFillCells)); }
size_t tie_big_bytes(int cells) { return 4 * (size_t)cells * sizeof(unsigned); }
// the fill kernel's copy of the keys lies in the two stop-list arrays (not in use before the partitions start)
__host__ __device__ inline unsigned long long* tie_key_copy(unsigned char* big, int cells) { return (unsigned long long*)(big + 2 * (size_t)cells * sizeof(unsigned)); }
size_t tie_small_bytes(int F, int HW) {       // block table (begin, length), its prefix, two segment lists
    size_t b = ((size_t)tie_blocks(F, HW) * (sizeof(int2) + sizeof(int)) + 15) & ~(size_t)15;
    b += 2 * tie_seg_cap(F * HW) * sizeof(int2);
    b += (((size_t)F * HW + 31) / 32) * sizeof(unsigned);  // cut marks of an image beyond the LDS arrays
    return (b + 255) & ~(size_t)255;
}

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
template <bool LDS>
__device__ __forceinline__ int tie_partition_block(unsigned* bits_, unsigned* idx_, unsigned* lpos_, unsigned* rpos_, int first,
                                                   int last, int* sh) {
    const TieArr<LDS> BITS = tie_arr<LDS>(bits_), IDX = tie_arr<LDS>(idx_), LPOS = tie_arr<LDS>(lpos_), RPOS = tie_arr<LDS>(rpos_);
    constexpr int NW = kTieThreads / 64;
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
    for (int k = tid; k < kmax; k += kTieThreads) {
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
        if (d == 0) { if (lane == 0) *s_fail = 1; break; }         // the heapsort branch: not reproduced
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
template <bool LDS>
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
        if (depth == 0) { if (tid == 0) *s_fail = 1; break; }      // the heapsort branch: not reproduced
        depth--;
        auto children = [&](const int2 sg, int cut) {               // (one thread)
            if (cut < sg.y) atomicOr(&mark[cut >> 5], 1u << (cut & 31));
            if (cut - sg.x > 16) nxt[atomicAdd(s_next, 1)] = make_int2(sg.x, cut);
            if (sg.y - cut > 16) nxt[atomicAdd(s_next, 1)] = make_int2(cut, sg.y);
        };
        for (int s = 0; s < n_cur; s++) {                          // the long ones: the whole workgroup on each
            const int2 sg = cur[s];
            if (sg.y - sg.x <= tie_coop_len<LDS>()) continue;
            const int cut = tie_partition_block<LDS>(BITS, IDX, LPOS, RPOS, sg.x, sg.y, coop);
            if (cut == -2) continue;
            if (cut < 0) { if (tid == 0) *s_fail = 1; continue; }
            if (tid == 0) children(sg, cut);
        }
        int mine = 0;                                              // the others: one wave per segment
        for (int s = 0; s < n_cur; s++) {
            const int2 sg = cur[s];
            if (sg.y - sg.x > tie_coop_len<LDS>()) continue;
            if ((mine++ % (kTieThreads / 64)) != wave) continue;
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

__global__ __launch_bounds__(kTieThreads) void cifseeds_tie_kernel(TieArgs a, SortArgs g, DevParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char tie_lds[];    // 128 KiB: four arrays of 8192 words
    __shared__ int s_flag, s_next, s_fail;
    __shared__ int s_wave_tot[kTieThreads / 64];
    __shared__ int2 s_seg[2][kTieLdsKeys / 17 + 2];           // the two segment lists of an image that lives in LDS
    __shared__ unsigned s_mark[kTieLdsKeys / 32];
    __shared__ int s_coop[2 * (kTieThreads / 64) + 2];
    __shared__ int s_stack[(kTieThreads / 64) * 3 * kTieStack];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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
            for (int t = tid; t < n; t += kTieThreads) SV[t] = sortable_bits(sv[(size_t)t * ncol]);
            __syncthreads();
            for (int t = tid; t + 1 < n; t += kTieThreads) any |= SV[t] == SV[t + 1];
        } else {
            for (int t = tid; t + 1 < n; t += kTieThreads) any |= sv[(size_t)t * ncol] == sv[(size_t)(t + 1) * ncol];
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
        const int per = (E + kTieThreads - 1) / kTieThreads;
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
        const int per = (n + kTieThreads - 1) / kTieThreads;
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
        for (int k = 0; k < kTieThreads / 64; k++) { if (k < wave) run += s_wave_tot[k]; n_tv += s_wave_tot[k]; }
        if (n_tv <= kTvCap)
            for (int t = r0; t < r1; t++)
                if (first_of_group(t)) TV[run++] = sortable_bits(sv[(size_t)t * ncol]);
        __syncthreads();
    }
    // ... and every key goes to (seeds before its block) + (its offset in the block); its block follows from its cell
    const int chunks = E / g.F;
    for (int t0 = tid; t0 < n; t0 += 4 * kTieThreads) {
        unsigned long long key[4]; int pre[4], beg[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { const int t = t0 + u * kTieThreads; key[u] = K[t < n ? t : 0]; }
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
            const int t = t0 + u * kTieThreads;
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
    for (int w = tid; w < (n + 31) / 32; w += kTieThreads) mark[w] = 0u;
    tie_group_sync_rt(in_lds);

    // ---- __introsort_loop, one level of the recursion at a time
    if (in_lds) tie_levels<true>(BITS, IDX, LPOS, RPOS, s_seg[0], s_seg[1], n, &s_next, &s_fail, mark, s_coop, s_stack, nullptr);
    else tie_levels<false>(BITS, IDX, LPOS, RPOS, seg_a, seg_b, n, &s_next, &s_fail, mark, s_coop, s_stack, (unsigned*)tie_lds);
    __syncthreads();
    if (s_fail) {                                                  // the seeds stay as the first sort left them
        if (tid == 0 && a.tie_state) a.tie_state[b] = -1;
        return;
    }

    // ---- __final_insertion_sort: a tied element ends up behind the larger and the equal-and-earlier elements of its
    //      segment (between the mark at or before it and the next one, at most 16 positions)
    auto mk = [&](int w) { return in_lds ? mark[w] : __hip_atomic_load(&mark[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    auto rd = [&](const unsigned* q, int i) { return in_lds ? q[i] : __hip_atomic_load(q + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    for (int k = tid; k < n; k += kTieThreads) {
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

hipError_t launch_cifseeds_ties(unsigned long long* keys, int sort_cap, const int32_t* seed_count, const float* cif, int B,
                                int F, int NC, int HW, int stride, int32_t* seed_f, float* seed_vxys, int32_t* seed_cell,
                                int occ_h, int occ_w, const DevParams& p, unsigned char* big, size_t big_stride,
                                unsigned char* small_, size_t small_stride, int32_t* tie_state, hipStream_t st) {
    SortArgs g;
    g.keys = keys; g.sort_cap = sort_cap; g.cap = F * HW; g.seed_count = seed_count; g.cif = cif; g.F = F; g.NC = NC; g.HW = HW;
    g.stride = stride; g.seed_f = seed_f; g.seed_vxys = seed_vxys; g.seed_cell = seed_cell; g.occ_h = occ_h; g.occ_w = occ_w;
    TieArgs a;
    a.cells = F * HW; a.big = big; a.big_stride = big_stride; a.small_ = small_; a.small_stride = small_stride; a.tie_state = tie_state;
    const int lds = 4 * kTieLdsKeys * (int)sizeof(unsigned);
    {   // (per device, not per process: set on every launch like the association kernel's)
        hipError_t e = hipFuncSetAttribute((const void*)cifseeds_tie_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
    }
    cifseeds_tie_kernel<<<B, kTieThreads, lds, st>>>(a, g, p);
    prof_mark(st, "cifseeds_tie_kernel");
    return hipGetLastError();
}

hipError_t launch_cifseeds(const float* cif, int B, int F, int H, int W, int stride,
                           const float* cifhr, int hr_rows, int hr_cols, int hr_pitch, const DevParams& p,
                           unsigned long long* keys, int sort_cap, int32_t* seed_count,
                           int32_t* seed_f, float* seed_vxys, hipStream_t st, bool det,
                           int32_t* seed_cell, int occ_h, int occ_w, bool count_is_zero,
                           const ScoredArgs* scored, int n_scored, const TieScratch* ties) {
    static_assert(kScoredThreads == 512, "the fused launch packs two cafscored groups into a 1024-thread workgroup");
    const int HW = H * W, cap = F * HW, NC = det ? 6 : 5;
    if (!count_is_zero) {                             // (the decode pipeline clears the counters in its first kernel)
        hipError_t e = launch_zero(seed_count, sizeof(int32_t) * B, st);
        if (e != hipSuccess) return e;
        prof_mark(st, "memset_seed_count");
    }
    const bool tie_pass = ties && ties->big && seed_tie_order() == 1;
    dim3 grid(B * F, (HW + 256 * kFillCells - 1) / (256 * kFillCells));
    cifseeds_fill_kernel<<<grid, 256, 0, st>>>(cif, F, NC, H, W, stride, cifhr, hr_rows, hr_cols, hr_pitch,
                                               p.seed_threshold, det ? 0 : p.ablation_cifseeds_nms,
                                               det ? 0 : p.ablation_cifseeds_no_rescore, keys, sort_cap, cap, seed_count,
                                               tie_pass ? (int2*)ties->small_ : nullptr,
                                               tie_pass ? ties->small_stride / sizeof(int2) : 0,
                                               tie_pass ? tie_key_copy(ties->big, cap) : nullptr,
                                               tie_pass ? ties->big_stride / sizeof(unsigned long long) : 0);
    prof_mark(st, "cifseeds_fill_kernel");
    SortArgs g;
    g.keys = keys; g.sort_cap = sort_cap; g.cap = cap; g.seed_count = seed_count; g.cif = cif; g.F = F; g.NC = NC; g.HW = HW;
    g.stride = stride; g.seed_f = seed_f; g.seed_vxys = seed_vxys; g.seed_cell = seed_cell; g.occ_h = occ_h; g.occ_w = occ_w;
    const int n_sort = B * kSortBlocksMax;
    if (scored && n_scored > 0) {
        const ScoredArgs& s0 = scored[0];
        const ScoredArgs& s1 = scored[n_scored > 1 ? 1 : 0];
        const int wgs0 = (s0.planes + 1) / 2, wgs1 = n_scored > 1 ? (s1.planes + 1) / 2 : 0;
        const int nb_max = s0.nb > s1.nb || n_scored < 2 ? s0.nb : s1.nb;
        const size_t lds = sizeof(float) * 2 * 2 * nb_max * 4;
        cifseeds_sort_scored_kernel<<<n_sort + wgs0 + wgs1, 1024, lds, st>>>(g, p, n_sort, s0, s1, wgs0);
    } else {
        cifseeds_sort_kernel<<<n_sort, 1024, 0, st>>>(g, p);
    }
    if (cap > kSortSmallBlock) {                      // images of more than one block of seeds are possible
        const int most = cap < kSortBlocksMax * kSortLdsKeys ? cap : kSortBlocksMax * kSortLdsKeys;
        cifseeds_rankmerge_kernel<<<dim3((most + 255) / 256, B), 256, 0, st>>>(keys, sort_cap, cap, seed_count, cif, F, NC, HW,
                                                                                 stride, seed_f, seed_vxys, seed_cell, occ_h, occ_w, p);
    }
    prof_mark(st, scored && n_scored > 0 ? "sort_cafscored_kernel" : "cifseeds_sort_kernel");
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && tie_pass)                  // cif_seeds.cpp:94: std::sort's order of equal scores
        e = launch_cifseeds_ties(keys, sort_cap, seed_count, cif, B, F, NC, HW, stride, seed_f, seed_vxys, seed_cell, occ_h, occ_w,
                                 p, ties->big, ties->big_stride, ties->small_, ties->small_stride, ties->state, st);
    return e;
}

}  // namespace opa
