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
#include "cifseeds_tie.hpp"



namespace opa {


__global__ __launch_bounds__(256) void cifseeds_fill_kernel(
        const float* __restrict__ cif, int F, int NC, int H, int W, int stride,
        const float* __restrict__ cifhr, int hr_rows, int hr_cols, int hr_pitch,
        double threshold, int ablation_nms, int no_rescore,
        unsigned long long* __restrict__ keys, int sort_cap, int cap, int32_t* __restrict__ seed_count,
        int2* __restrict__ wg_tab, size_t tab_stride, unsigned long long* __restrict__ key_copy, size_t copy_stride,
        const int32_t* __restrict__ hr_slot, int hr_tpp, size_t hr_image_stride) {
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
                    const float hv = cifhr_value(cifhr + (size_t)b * hr_image_stride, F, hr_rows, hr_cols, hr_pitch, f, x, y, -1.0f,
                                                 nullptr, 0, hr_pitch / kHrTileW, hr_slot ? hr_slot + (size_t)b * F * hr_tpp : nullptr, hr_tpp);
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

// The decode path's form: the candidates cif_active_kernel left (common.hpp: SeedCandidates) instead of the field.  Same grid,
// same blocks -- workgroup (plane, chunk) holds the candidates among cells [1024 chunk, 1024 chunk + 1024) of its plane, in
// cell order -- so the block table the tie pass reads (a key's block follows from its cell index) means what it meant; what
// changes is that a batch of 32 COCO images reads ~150 000 candidates of 16 bytes instead of three planes of 71 MB.
__global__ __launch_bounds__(256) void cifseeds_fill_cand_kernel(
        const float* __restrict__ cif, const float4* __restrict__ cand, const int32_t* __restrict__ cand_start,
        const int32_t* __restrict__ cand_count, int F, int NC, int H, int W, int stride,
        const float* __restrict__ cifhr, int hr_rows, int hr_cols, int hr_pitch,
        double threshold, int ablation_nms, int no_rescore,
        unsigned long long* __restrict__ keys, int sort_cap, int cap, int32_t* __restrict__ seed_count,
        int2* __restrict__ wg_tab, size_t tab_stride, unsigned long long* __restrict__ key_copy, size_t copy_stride,
        const int32_t* __restrict__ hr_slot, int hr_tpp, size_t hr_image_stride) {
    const int HW = H * W;
    const int plane = blockIdx.x;              // b*F + f
    const int b = plane / F, f = plane - b * F;
    const int lane = threadIdx.x & 63;
    const int chunks = gridDim.y;
    const int first = cand_start[(size_t)plane * chunks + blockIdx.y];
    const int last = blockIdx.y + 1 < chunks ? cand_start[(size_t)plane * chunks + blockIdx.y + 1] : cand_count[plane];
    if (first >= last) {                       // (uniform) nothing in this chunk -- most chunks of most planes
        if (threadIdx.x == 0 && wg_tab) wg_tab[(size_t)b * tab_stride + (size_t)f * chunks + blockIdx.y] = make_int2(0, 0);
        return;
    }
    const float* P = cif + (size_t)plane * NC * HW;
    const float4* C = cand + (size_t)plane * HW;
    int o[kFillCells]; float c[kFillCells], xin[kFillCells], yin[kFillCells]; bool on[kFillCells];
#pragma unroll
    for (int r = 0; r < kFillCells; r++) {
        const int i = first + r * 256 + threadIdx.x;
        const float4 q = C[i < last ? i : first];
        on[r] = i < last;                                            // (a candidate passed cif_seeds.cpp:47 already)
        o[r] = __float_as_int(q.x); c[r] = q.y; xin[r] = q.z; yin[r] = q.w;
    }
#pragma unroll
    for (int r = 0; r < kFillCells; r++) {
        if (on[r]) {
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
                    const float hv = cifhr_value(cifhr + (size_t)b * hr_image_stride, F, hr_rows, hr_cols, hr_pitch, f, x, y, -1.0f,
                                                 nullptr, 0, hr_pitch / kHrTileW, hr_slot ? hr_slot + (size_t)b * F * hr_tpp : nullptr, hr_tpp);
                    c[r] = (float)(0.9 * (double)hv + 0.1 * (double)c[r]);
                }
                if ((double)c[r] < threshold) on[r] = false;         // :59
            }
        }
    }
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
        if (wg_tab) wg_tab[(size_t)b * tab_stride + (size_t)f * chunks + blockIdx.y] = make_int2(wg_base, total);
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
                if (key_copy) key_copy[(size_t)b * copy_stride + slot] = key;
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

// ---- round 6: the 2048-key blocks in REGISTERS.  Round 3's network keeps the keys in LDS: 66 passes of one compare-exchange per
// thread, every pass an LDS round trip (four 8-byte accesses per lane, bank conflicts measured at 89 % of the LDS cycles) and a
// barrier or wave fence -- 25 us for a block that holds 16 KB.  Here a block is 256 threads x 8 keys: position p = 8 t + r, so
//   strides 1, 2, 4          exchange two registers of one thread            (30 of the 66 passes)
//   strides 8 ... 256        exchange with lane ^ (stride / 8): ds_bpermute  (33 passes; no memory, no bank conflicts)
//   strides 512, 1024        exchange with another wave: through LDS, SoA [r][t] (3 passes, 2 barriers each)
// The order is the same total order (score descending, then cell index ascending), so the result does not depend on the network.
// Blocks of images with more than 8192 seeds (wholebody) stay with the kernel below (8192-key blocks).
constexpr int kSort2kThreads = 256, kSort2kKeys = 8;
static_assert(kSort2kThreads * kSort2kKeys == kSortSmallBlock, "one block = 256 threads x 8 keys");

__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long v, int mask) {
    const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)v, mask, 64), hi = (unsigned)__shfl_xor((int)(unsigned)(v >> 32), mask, 64);
    return ((unsigned long long)hi << 32) | lo;
}

__global__ __launch_bounds__(kSort2kThreads) void cifseeds_sort2k_kernel(SortArgs g, DevParams p) {
    __shared__ unsigned long long xs[kSort2kKeys][kSort2kThreads];   // 16 KB: the cross-wave exchanges
    const int b = blockIdx.x / 4, part = blockIdx.x & 3, t = threadIdx.x;
    int n = g.seed_count[b];
    if (n > g.cap) n = g.cap;
    if (n > 4 * kSortSmallBlock || part * kSortSmallBlock >= n) return;   // (more than 8192 seeds: the 8192-key blocks of the kernel below)
    const bool single = n <= kSortSmallBlock;         // one block: sorted and decoded here; else the rank merge kernel places the keys
    unsigned long long* K = g.keys + (size_t)b * g.sort_cap;
    const int blk = part * kSortSmallBlock;
    unsigned long long key[kSort2kKeys];
#pragma unroll
    for (int r = 0; r < kSort2kKeys; r++) { const int i = blk + t * kSort2kKeys + r; key[r] = i < n ? K[i] : 0ull; }
    auto cx = [](unsigned long long& lo, unsigned long long& hi, bool desc) {     // positions lo < hi: descending = the larger key first
        const bool swap = (lo < hi) == desc;          // (one 64-bit compare per exchange; equal keys are padding: either way)
        const unsigned long long a = lo, c = hi;
        lo = swap ? c : a; hi = swap ? a : c;
    };
#pragma unroll
    for (int k = 2; k <= kSortSmallBlock; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j >= 512) {                           // another wave's keys: through LDS
#pragma unroll
                for (int r = 0; r < kSort2kKeys; r++) xs[r][t] = key[r];
                __syncthreads();
                const int pt = t ^ (j >> 3);
                const bool lower = (t & (j >> 3)) == 0, desc = ((t * kSort2kKeys) & k) == 0;
                const bool take_max = lower == desc;
#pragma unroll
                for (int r = 0; r < kSort2kKeys; r++) {
                    const unsigned long long o = xs[r][pt];
                    key[r] = ((o > key[r]) == take_max) ? o : key[r];
                }
                __syncthreads();
            } else if (j >= kSort2kKeys) {            // another lane's keys
                const int m = j >> 3;
                const bool lower = (t & m) == 0, desc = ((t * kSort2kKeys) & k) == 0;
                const bool take_max = lower == desc;
                unsigned long long o[kSort2kKeys];
#pragma unroll
                for (int r = 0; r < kSort2kKeys; r++) o[r] = shfl_xor_u64(key[r], m);      // sixteen ds_bpermute in flight
#pragma unroll
                for (int r = 0; r < kSort2kKeys; r++) key[r] = ((o[r] > key[r]) == take_max) ? o[r] : key[r];
            } else {                                  // two registers of this thread
#pragma unroll
                for (int r = 0; r < kSort2kKeys; r++)
                    if ((r & j) == 0) cx(key[r], key[r | j], ((t * kSort2kKeys + r) & k) == 0);
            }
        }
    }
    if (single) {                                     // epilogue: decode keys -> sorted seeds (cif_seeds.cpp:100-113)
#pragma unroll
        for (int r = 0; r < kSort2kKeys; r++) {
            const int pos = t * kSort2kKeys + r;
            if (pos < n) store_seed(key[r], pos, b, g.cif, g.F, g.NC, g.HW, g.stride, g.cap, g.seed_f, g.seed_vxys, g.seed_cell, g.occ_h, g.occ_w, p);
        }
    } else {
#pragma unroll
        for (int r = 0; r < kSort2kKeys; r++) K[blk + t * kSort2kKeys + r] = key[r];
    }
}

__global__ __launch_bounds__(1024) void cifseeds_sort_kernel(SortArgs g, DevParams p, int only_large) {
    __shared__ unsigned long long sk[kSortLdsKeys];
    if (only_large) {                                 // (images of up to 8192 seeds were sorted by cifseeds_sort2k_kernel)
        int n = g.seed_count[blockIdx.x / kSortBlocksMax];
        if (n > g.cap) n = g.cap;
        if (n <= 4 * kSortSmallBlock) return;
    }
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

// (Round 6 also built the rank merge with the other blocks in LDS -- one workgroup per block, eight keys per thread searched in
// lockstep, branch-free: 71 us for 256 images against this kernel's 47.5, 15.5 against 14 for 32: one key per thread and thousands of
// threads hide the L2 round trips better than 48 KB of LDS per workgroup allow; removed.)

// ------------------------------------------------------------------ the reference's order of EQUAL scores: cifseeds_tie.hpp
size_t tie_big_bytes(int cells) { return 4 * (size_t)cells * sizeof(unsigned); }

size_t tie_small_bytes(int F, int HW) {       // block table (begin, length), its prefix, two segment lists
    size_t b = ((size_t)tie_blocks(F, HW) * (sizeof(int2) + sizeof(int)) + 15) & ~(size_t)15;
    b += 2 * tie_seg_cap(F * HW) * sizeof(int2);
    b += (((size_t)F * HW + 31) / 32) * sizeof(unsigned);  // cut marks of an image beyond the LDS arrays
    return (b + 255) & ~(size_t)255;
}

__global__ __launch_bounds__(kTieThreads) void cifseeds_tie_kernel(TieArgs a, SortArgs g, DevParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char tie_lds[];    // tie_lds_bytes<kTieThreads>()
    cifseeds_tie_body<kTieThreads>(a, g, p, blockIdx.x, tie_lds);
}


void make_tie_args(TieArgs* a, SortArgs* g, unsigned long long* keys, int sort_cap, const int32_t* seed_count, const float* cif,
                   int F, int NC, int HW, int stride, int32_t* seed_f, float* seed_vxys, int32_t* seed_cell, int occ_h, int occ_w,
                   const TieScratch& t) {
    g->keys = keys; g->sort_cap = sort_cap; g->cap = F * HW; g->seed_count = seed_count; g->cif = cif; g->F = F; g->NC = NC; g->HW = HW;
    g->stride = stride; g->seed_f = seed_f; g->seed_vxys = seed_vxys; g->seed_cell = seed_cell; g->occ_h = occ_h; g->occ_w = occ_w;
    a->cells = F * HW; a->big = t.big; a->big_stride = t.big_stride; a->small_ = t.small_; a->small_stride = t.small_stride;
    a->tie_state = t.state;
}

hipError_t launch_cifseeds_ties(const TieArgs& a, const SortArgs& g, int B, const DevParams& p, hipStream_t st) {
    const int lds = (int)tie_lds_bytes<kTieThreads>();
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
                           const ScoredArgs* scored, int n_scored, const TieScratch* ties, const HrPool* pool, const SeedCandidates* cand,
                           bool sort_registers) {
    static_assert(kScoredThreads == 512, "the fused launch packs two cafscored groups into a 1024-thread workgroup");
    const int HW = H * W, cap = F * HW, NC = det ? 6 : 5;
    if (!count_is_zero) {                             // (the decode pipeline clears the counters in its first kernel)
        hipError_t e = launch_zero(seed_count, sizeof(int32_t) * B, st);
        if (e != hipSuccess) return e;
        prof_mark(st, "memset_seed_count");
    }
    const bool tie_pass = ties && ties->big && seed_tie_order() >= 1;
    dim3 grid(B * F, (HW + 256 * kFillCells - 1) / (256 * kFillCells));
    if (cand && cand->produced && !det && cand->chunks == (int)grid.y)
        cifseeds_fill_cand_kernel<<<grid, 256, 0, st>>>(cif, cand->cand, cand->start, cand->count, F, NC, H, W, stride, cifhr, hr_rows,
                                                        hr_cols, hr_pitch, p.seed_threshold, p.ablation_cifseeds_nms,
                                                        p.ablation_cifseeds_no_rescore, keys, sort_cap, cap, seed_count,
                                                        tie_pass ? (int2*)ties->small_ : nullptr,
                                                        tie_pass ? ties->small_stride / sizeof(int2) : 0,
                                                        tie_pass ? tie_key_copy(ties->big, cap) : nullptr,
                                                        tie_pass ? ties->big_stride / sizeof(unsigned long long) : 0,
                                                        pool ? pool->slot : nullptr, pool ? pool->tpp : 0,
                                                        pool ? (size_t)pool->cap * kHrTileH * kHrTileW : (size_t)F * hr_rows * hr_pitch);
    else
        cifseeds_fill_kernel<<<grid, 256, 0, st>>>(cif, F, NC, H, W, stride, cifhr, hr_rows, hr_cols, hr_pitch,
                                               p.seed_threshold, det ? 0 : p.ablation_cifseeds_nms,
                                               det ? 0 : p.ablation_cifseeds_no_rescore, keys, sort_cap, cap, seed_count,
                                               tie_pass ? (int2*)ties->small_ : nullptr,
                                               tie_pass ? ties->small_stride / sizeof(int2) : 0,
                                               tie_pass ? tie_key_copy(ties->big, cap) : nullptr,
                                               tie_pass ? ties->big_stride / sizeof(unsigned long long) : 0,
                                               pool ? pool->slot : nullptr, pool ? pool->tpp : 0,
                                               pool ? (size_t)pool->cap * kHrTileH * kHrTileW : (size_t)F * hr_rows * hr_pitch);
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
    } else if (sort_registers) {
        cifseeds_sort2k_kernel<<<B * 4, kSort2kThreads, 0, st>>>(g, p);
        if (cap > 4 * kSortSmallBlock) cifseeds_sort_kernel<<<n_sort, 1024, 0, st>>>(g, p, 1);
    } else {
        cifseeds_sort_kernel<<<n_sort, 1024, 0, st>>>(g, p, 0);
    }
    if (cap > kSortSmallBlock) {                      // images of more than one block of seeds are possible
        const int most = cap < kSortBlocksMax * kSortLdsKeys ? cap : kSortBlocksMax * kSortLdsKeys;
        cifseeds_rankmerge_kernel<<<dim3((most + 255) / 256, B), 256, 0, st>>>(keys, sort_cap, cap, seed_count, cif, F, NC, HW,
                                                                                 stride, seed_f, seed_vxys, seed_cell, occ_h, occ_w, p);
    }
    prof_mark(st, scored && n_scored > 0 ? "sort_cafscored_kernel" : "cifseeds_sort_kernel");
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && tie_pass && !ties->defer) {   // cif_seeds.cpp:94: std::sort's order of equal scores
        TieArgs ta; SortArgs tg;
        make_tie_args(&ta, &tg, keys, sort_cap, seed_count, cif, F, NC, HW, stride, seed_f, seed_vxys, seed_cell, occ_h, occ_w, *ties);
        e = launch_cifseeds_ties(ta, tg, B, p, st);
    }
    return e;
}

}  // namespace opa
