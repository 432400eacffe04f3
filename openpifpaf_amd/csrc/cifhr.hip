// CifHr: high-resolution confidence accumulation on gfx950.
//
// Replaces reference CifHr::accumulate / add_gauss / reset
// (csrc/src/cif_hr.cpp:18-121).  The reference scatter-adds one truncated
// Gaussian per active CIF cell into a [F, Hhr, Whr] float map with a serial
// triple loop.  Here:
//
//  pass 1  cif_active_kernel   one workgroup per (image, field) plane: ordered
//          stream compaction (wave ballot + prefix) of the cells that pass the
//          confidence/scale thresholds into (v/16, x, y, sigma) arrays, in the
//          reference's raster order.
//  pass 2  cifhr_tile_kernel   one wavefront per 32x64 high-res tile, tile held
//          in LDS.  The wave walks the plane's active list 64 cells at a time,
//          ballots "box overlaps my tile", and applies the overlapping cells IN
//          LIST ORDER with 16x4-pixel lane patches.  Because every pixel sees its
//          contributions in exactly the reference's order and with the reference's
//          float/double operation sequence (no FMA contraction), the map is
//          bit-identical to a fresh reference instance.  Tiles are then written
//          with coalesced 16-B stores (each tile row = two 128-B lines).
//
// Map content = the reference buffer at revision 1.0: 0.0 where untouched,
// otherwise 1.0 + min(1, accumulated).
#include "common.hpp"

namespace opa {

// cif_hr.cpp:61-64 (truncate = 1.0): the pixel box add_gauss walks for a cell
__device__ __forceinline__ void gauss_box(float cx, float cy, float sigma, int rows, int cols,
                                          int* minx, int* miny, int* maxx, int* maxy) {
    *minx = (int)clampll(trunc_ll(cx - sigma), 0, cols - 1);
    *miny = (int)clampll(trunc_ll(cy - sigma), 0, rows - 1);
    *maxx = (int)clampll(trunc_ll(cx + sigma + 1.0f), *minx + 1, cols);
    *maxy = (int)clampll(trunc_ll(cy + sigma + 1.0f), *miny + 1, rows);
}

// ---------------------------------------------------------------- pass 1
// DET: CifDet fields [F,6,H,W] (w,h instead of scale), CifDetHr::accumulate cif_hr.cpp:124-150
#ifndef OPA_ACTIVE_THREADS
#define OPA_ACTIVE_THREADS 256
#endif
constexpr int kActiveThreads = OPA_ACTIVE_THREADS;
constexpr int kActiveCells = 4;

// WL (the decode path's pooled map): the kernel also
//  * gives every tile its cells reach a slot of the image's pool (one returned atomic per plane on the image's slot counter,
//    the spill region's counter for what the pool cannot hold) and appends (tile, slot) to the batch's WORK LIST of the tile
//    kernel (one returned atomic per plane): cifhr_worktile_kernel then builds one tile per workgroup, whatever plane it
//    belongs to -- no plane keeps the chip waiting for its crowded tiles;
//  * writes the plane's SEED CANDIDATES -- the cells CifSeeds::fill would look at, cif_seeds.cpp:47: (cell, c, x, y) of
//    every cell with !(c < seed_threshold), in raster order, and where the candidates of every 1024-cell chunk begin --
//    so that cifseeds_fill reads a few thousand candidates per image instead of streaming the field a second time.
// The counters (pool.img_tiles, pool.work_count, pool.spill_count, pool.overflow) are zeroed by a launch before this one.
template <bool DET, bool WL>
__global__ __launch_bounds__(kActiveThreads) void cif_active_kernel(
        const float* __restrict__ cif, int HW, int stride, float min_scale_f, double threshold,
        float neighbors_f, double factor, float* __restrict__ act, int32_t* __restrict__ act_count,
        unsigned long long* ws_header, unsigned long long layout_hash,
        unsigned* __restrict__ tile_touch, int touch_words, int rows, int cols, int tiles_x,
        int32_t* __restrict__ zero_per_image, int F, HrPool pool,
        float4* __restrict__ cand, int32_t* __restrict__ cand_start, int32_t* __restrict__ cand_count, int cand_chunks,
        double seed_threshold) {
    __shared__ int wave_tot[2][kActiveCells][kActiveThreads / 64];
    extern __shared__ unsigned sh_touch[];            // WL: the plane's touched-tile bitmap (touch_words words)
    const int plane = blockIdx.x;
    if (zero_per_image && threadIdx.x == 0 && plane % F == 0) zero_per_image[plane / F] = 0;   // the image's seed counter
    if (pool.slot) {                                  // pooled map: no tile of this plane has a slot yet
        for (int k = threadIdx.x; k < pool.tpp; k += kActiveThreads) pool.slot[(size_t)plane * pool.tpp + k] = -1;
        if (!WL) {
            if (threadIdx.x == 0 && plane % F == 0) pool.overflow[plane / F] = 0;
            if (threadIdx.x == 0 && plane == 0 && pool.spill_count) *pool.spill_count = 0;
        }
    }
    unsigned* touch = tile_touch ? tile_touch + (size_t)plane * touch_words : nullptr;   // one bit per tile of this plane
    if (WL) {
        for (int k = threadIdx.x; k < touch_words; k += kActiveThreads) sh_touch[k] = 0u;
        __syncthreads();
    } else if (touch) {
        for (int k = threadIdx.x; k < touch_words; k += kActiveThreads) touch[k] = 0u;
        __syncthreads();
    }
    if (ws_header && blockIdx.x == 0 && threadIdx.x == 0) {       // do the clean-tile flags describe this layout?
        const bool valid = ws_header[0] == kWsMagic && ws_header[1] == layout_hash;
        ws_header[2] = valid ? 0ull : 1ull;
        ws_header[0] = kWsMagic; ws_header[1] = layout_hash;
    }
    const float* P = cif + (size_t)plane * (DET ? 6 : 5) * HW;
    float* out = act + (size_t)plane * 4 * HW;
    float4* cout = WL ? cand + (size_t)plane * HW : nullptr;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const float stride_f = (float)stride;
    int base = 0, cbase = 0;
    int parity = 0;
    // kActiveCells cells per thread and step: their confidence loads are in flight together; cell order
    // (r, wave, lane) is raster order, and the list keeps it.  The planes of the NEXT step are requested before this step's
    // cells are looked at: the memory round trip of a step overlaps the compaction (and its barrier) of the step before.
    float vin[kActiveCells], xin[kActiveCells], yin[kActiveCells], sin_[kActiveCells], hin[kActiveCells];
    auto request = [&](int c0, float* v_, float* x_, float* y_, float* s_, float* h_) {
#pragma unroll
        for (int r = 0; r < kActiveCells; r++) {
            const int o = c0 + r * kActiveThreads + tid;
            const int oo = o < HW ? o : 0;
            v_[r] = P[HW + oo];                       // (unconditional: every consumer tests o < HW itself; a load under a
                                                      // condition makes the compiler's waits behind the join conservative)
            x_[r] = P[2 * HW + oo]; y_[r] = P[3 * HW + oo]; s_[r] = P[4 * HW + oo];
            h_[r] = DET ? P[5 * HW + oo] : 0.0f;
        }
    };
    request(0, vin, xin, yin, sin_, hin);
    for (int c0 = 0, step = 0; c0 < HW; c0 += kActiveThreads * kActiveCells, parity ^= 1, step++) {
        float vn[kActiveCells], xn[kActiveCells], yn[kActiveCells], sn[kActiveCells], hn[kActiveCells];
        const bool more = c0 + kActiveThreads * kActiveCells < HW;
        // (always: the last step requests its own cells again -- with the request under `if (more)` the step's first cell
        // waited for part of the NEXT step's loads, vmcnt(12) with sixteen of them just issued)
        request(more ? c0 + kActiveThreads * kActiveCells : c0, vn, xn, yn, sn, hn);
        if (WL && tid == 0) cand_start[(size_t)plane * cand_chunks + step] = cbase;
        bool on[kActiveCells], con[kActiveCells];
        float v16[kActiveCells], x[kActiveCells], y[kActiveCells], sigma[kActiveCells];
#pragma unroll
        for (int r = 0; r < kActiveCells; r++) {
            const int o = c0 + r * kActiveThreads + tid;
            on[r] = false; v16[r] = 0.f; x[r] = 0.f; y[r] = 0.f; sigma[r] = 0.f;
            const float v = vin[r];
            con[r] = WL && o < HW && !((double)v < seed_threshold);            // cif_seeds.cpp:47
            if (o < HW && !((double)v < threshold)) {             // cif_hr.cpp:39
                const float scale = sin_[r];
                bool big_enough;
                double sigma_d;
                if (DET) {                                        // cif_hr.cpp:135-141
                    const float h = hin[r];
                    big_enough = !(scale < min_scale_f || h < min_scale_f);
                    sigma_d = 0.1 * (double)fminf(scale, h) * (double)stride;
                } else {                                          // cif_hr.cpp:42,46
                    big_enough = !(scale < min_scale_f);
                    sigma_d = 0.5 * (double)scale * (double)stride;
                }
                if (big_enough) {
                    on[r] = true;
                    x[r] = xin[r] * stride_f;                     // cif_hr.cpp:44-45
                    y[r] = yin[r] * stride_f;
                    sigma[r] = fmaxf(1.0f, (float)sigma_d);
                    v16[r] = (float)((double)(v / neighbors_f) * factor);                 // :51
                    if (WL || touch) {                            // tiles this cell's box overlaps
                        int minx, miny, maxx, maxy;
                        gauss_box(x[r], y[r], sigma[r], rows, cols, &minx, &miny, &maxx, &maxy);
                        for (int ty = miny / kHrTileH; ty <= (maxy - 1) / kHrTileH; ty++)
                            for (int tx = minx / kHrTileW; tx <= (maxx - 1) / kHrTileW; tx++) {
                                const int t = ty * tiles_x + tx;
                                if (WL) atomicOr(&sh_touch[t >> 5], 1u << (t & 31));
                                else atomicOr(&touch[t >> 5], 1u << (t & 31));
                            }
                    }
                }
            }
        }
        unsigned long long mask[kActiveCells], cmask[kActiveCells];
#pragma unroll
        for (int r = 0; r < kActiveCells; r++) {
            mask[r] = __ballot(on[r]);
            cmask[r] = WL ? __ballot(con[r]) : 0ull;
            if (lane == 0) wave_tot[parity][r][w] = __popcll(mask[r]) | (__popcll(cmask[r]) << 16);
        }
        __syncthreads();                              // double-buffered totals: one barrier per step
#pragma unroll
        for (int r = 0; r < kActiveCells; r++) {
            const unsigned long long lt = (1ull << lane) - 1ull;
            int off = base + __popcll(mask[r] & lt), coff = cbase + __popcll(cmask[r] & lt), tot = 0, ctot = 0;
#pragma unroll
            for (int k = 0; k < kActiveThreads / 64; k++) {
                const int t = wave_tot[parity][r][k];
                if (k < w) { off += t & 0xffff; coff += t >> 16; }
                tot += t & 0xffff; ctot += t >> 16;
            }
            if (on[r]) {
                out[0 * HW + off] = v16[r]; out[1 * HW + off] = x[r];
                out[2 * HW + off] = y[r];   out[3 * HW + off] = sigma[r];
            }
            if (WL && con[r])
                cout[coff] = make_float4(__int_as_float(c0 + r * kActiveThreads + tid), vin[r], xin[r], yin[r]);
            base += tot; cbase += ctot;
        }
#pragma unroll
        for (int r = 0; r < kActiveCells; r++) { vin[r] = vn[r]; xin[r] = xn[r]; yin[r] = yn[r]; sin_[r] = sn[r]; hin[r] = hn[r]; }
    }
    if (tid == 0) act_count[plane] = base;
    if (!WL) return;
    if (tid == 0) cand_count[plane] = cbase;
    // ---- slots and work items of the tiles this plane's cells reach
    __shared__ int sh_slot0, sh_work0;
    __syncthreads();                                  // (the LDS bitmap is complete)
    if (w == 0) {
        int n = 0;
        for (int k = lane; k < touch_words; k += 64) n += __popc(sh_touch[k]);
        for (int d = 32; d > 0; d >>= 1) n += __shfl_xor(n, d, 64);
        if (lane == 0) {
            sh_slot0 = n ? atomicAdd(&pool.img_tiles[plane / F], n) : 0;
            sh_work0 = n ? atomicAdd(pool.work_count, n) : 0;
        }
    }
    __syncthreads();
    const int b = plane / F;
    for (int k = tid; k < touch_words; k += kActiveThreads) {
        unsigned bits = sh_touch[k];
        if (touch) touch[k] = bits;                   // (the bitmap view of the workspace: tests, the bench's tile count)
        if (!bits) continue;
        int r = 0;
        for (int q = 0; q < k; q++) r += __popc(sh_touch[q]);
        while (bits) {
            const int t = k * 32 + __builtin_ctz(bits);
            bits &= bits - 1;
            int sl = sh_slot0 + r;
            if (sl >= pool.cap) {                     // more tiles than the image's pool holds: a slot of the batch's spill region
                const int i = pool.spill_cap > 0 ? atomicAdd(pool.spill_count, 1) : pool.spill_cap;
                if (i >= pool.spill_cap) { sl = -2; pool.overflow[b] = 1; }   // that ran out too: the image is flagged, not decoded wrongly
                else sl = (pool.images - b) * pool.cap + i;
            }
            pool.slot[(size_t)plane * pool.tpp + t] = sl;
            pool.work[sh_work0 + r] = make_int2(plane * pool.tpp + t, sl);
            r++;
        }
    }
}

// cif_hr.cpp:18-25.  The reference evaluates `1.0 + x / 8.0` and the caller's `-0.5 * d2 / sigma2` in double
// and rounds to float; for one +, / of float operands that double rounding is innocuous (53 >= 2*24+2
// bits), so the correctly rounded float operation gives the same bits at a third of the instructions.
__device__ __forceinline__ float approx_exp(float x) {
    if (x > 2.0f || x < -2.0f) return 0.0f;
    x = 1.0f + x * 0.125f;
    x *= x; x *= x; x *= x;
    return x;
}

// One active cell applied to a wave's band of a tile in LDS (cif_hr.cpp:66-89): `bv` = v / neighbors, (bx, by) its centre,
// bs its sigma, [bx0, bx1) x [by0, by1) its box clipped to the band, (x0, y0) the band's first pixel; lanes are a 16x4 patch.
__device__ __forceinline__ void apply_cell(float* __restrict__ T, float bv, float bx, float by, float bs, int bx0, int bx1,
                                           int by0, int by1, int x0, int y0, int lx, int ly) {
    const float sigma2 = bs * bs;                         // cif_hr.cpp:66-67
    for (int py = by0; py < by1; py += 4) {
        const int yy = py + ly;
        const float dy = (float)yy - by;
        const float dy2 = dy * dy;
        for (int px = bx0; px < bx1; px += 16) {
            const int xx = px + lx;
            if (xx < bx1 && yy < by1) {
                const float dx = (float)xx - bx;
                const float dx2 = dx * dx;
                const float d2 = dx2 + dy2;
                if (!(d2 > sigma2)) {                     // cif_hr.cpp:75
                    float vv;
                    if (dx2 < 0.25f && dy2 < 0.25f) vv = bv;     // :77-79
                    else vv = bv * approx_exp(__fdiv_rn(-0.5f * d2, sigma2));   // :81
                    float* e = T + (yy - y0) * kHrLdsPitch + (xx - x0);
                    float a = fmaxf(*e, 1.0f) + vv;       // :84-86 at revision 1.0
                    *e = fminf(a, 2.0f);
                }
            }
        }
    }
}

// ---------------------------------------------------------------- pass 2
// The four waves of a workgroup build one 32x64 tile in LDS, each its own band of 8 rows (pixels are
// independent of each other; only the order of the cells applied to ONE pixel matters), and write it out.
constexpr int kBandH = kHrTileH / 4;
// `out`: where the tile's pixel (0, 0) goes, `out_pitch` floats per tile row (the dense map: hr_plane + ytile * pitch + x0
// with the map's pitch; a pool slot: the slot's 32x64 block, pitch 64); `clip_rows`: rows below the map are not written.
__device__ __forceinline__ void build_tile(const float* __restrict__ A, int n, int HW, float* __restrict__ T,
                                           float* __restrict__ out, int out_pitch, bool clip_rows, int rows, int cols,
                                           int tx, int ty, int band, bool touched) {
    const int lane = threadIdx.x & 63;
    const int lx = lane & 15, ly = lane >> 4;
    const int x0 = tx * kHrTileW, ytile = ty * kHrTileH, y0 = ytile + band * kBandH;
    if (!touched) {                                   // no cell reaches this tile: it only has to be zero
        for (int r = ly; r < kBandH; r += 4)
            if (!clip_rows || y0 + r < rows)
                *reinterpret_cast<float4*>(out + (size_t)(band * kBandH + r) * out_pitch + lx * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
        return;
    }
    const int x1 = min(x0 + kHrTileW, cols), y1 = min(y0 + kBandH, rows);
    T += band * kBandH * kHrLdsPitch;                 // this wave's rows of the tile
    for (int k = lane; k < kBandH * kHrLdsPitch / 4; k += 64)
        reinterpret_cast<float4*>(T)[k] = make_float4(0.f, 0.f, 0.f, 0.f);

    // the list chunk after the current one is already in flight while the current one is applied
    float nv = 0.f, nx = 0.f, ny = 0.f, ns = 1.f;
    if (lane < n) { nv = A[0 * HW + lane]; nx = A[1 * HW + lane]; ny = A[2 * HW + lane]; ns = A[3 * HW + lane]; }
    for (int c0 = 0; c0 < n; c0 += 64) {
        const int i = c0 + lane;
        const float v16 = nv, cx = nx, cy = ny, sigma = ns;
        if (i + 64 < n) {
            nv = A[0 * HW + i + 64]; nx = A[1 * HW + i + 64]; ny = A[2 * HW + i + 64]; ns = A[3 * HW + i + 64];
        }
        int minx = 0, maxx = 0, miny = 0, maxy = 0;
        bool hit = false;
        if (i < n) {
            gauss_box(cx, cy, sigma, rows, cols, &minx, &miny, &maxx, &maxy);
            hit = minx < x1 && maxx > x0 && miny < y1 && maxy > y0;
        }
        unsigned long long mask = __ballot(hit);
#ifdef OPA_TILE_WALK_ONLY               // diagnostic: the list walk without the accumulation (what tile-binned lists could save)
        mask = 0ull;
#endif
        while (mask) {
            const int l = __builtin_ctzll(mask);
            mask &= mask - 1;
            // l is wave-uniform: v_readlane (a few cycles) instead of a ds_bpermute round trip per value
            auto rl = [l](float v) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); };
            const float bv = rl(v16), bx = rl(cx), by = rl(cy), bs = rl(sigma);
            const int bx0 = max(__builtin_amdgcn_readlane(minx, l), x0), bx1 = min(__builtin_amdgcn_readlane(maxx, l), x1);
            const int by0 = max(__builtin_amdgcn_readlane(miny, l), y0), by1 = min(__builtin_amdgcn_readlane(maxy, l), y1);
            apply_cell(T, bv, bx, by, bs, bx0, bx1, by0, by1, x0, y0, lx, ly);
        }
    }
    // coalesced write-out: 16 lanes x float4 = one 256-B tile row, 4 rows per instruction
    for (int r = ly; r < kBandH; r += 4) {
        const int yy = y0 + r;
        if (!clip_rows || yy < rows) {
            const float4 val = *reinterpret_cast<const float4*>(T + r * kHrLdsPitch + lx * 4);
            *reinterpret_cast<float4*>(out + (size_t)(band * kBandH + r) * out_pitch + lx * 4) = val;
        }
    }
}

// kTileGroups 4-wave workgroups per (image, field) plane share the plane's tiles, one tile at a time per
// workgroup.  Lazy clear, the reference's revision trick
// (cif_hr.cpp:97-121) per tile: pass 1 left a bitmap of the tiles this call's cells reach (`cur`), the
// workspace remembers the bitmap of the previous call (`prev`); only tiles in cur | prev are visited --
// built if in cur, zeroed if only in prev -- and every other tile is already all-zero in HBM.  Most of the
// map is such tiles.  Without workspace state (stage-level entry point, invalid header) every tile is built.
#ifndef OPA_TILE_GROUPS
#define OPA_TILE_GROUPS 8
#endif
constexpr int kTileGroups = OPA_TILE_GROUPS;

__global__ __launch_bounds__(256) void cifhr_tile_kernel(
        const float* __restrict__ act, const int32_t* __restrict__ act_count, int HW,
        float* __restrict__ hr, int rows, int cols, int pitch, int tiles_x, int tiles_y,
        const unsigned long long* __restrict__ ws_header, const unsigned* __restrict__ tile_prev,
        const unsigned* __restrict__ tile_cur, int touch_words, HrPool pool, int F) {
    __shared__ __attribute__((aligned(16))) float T[kHrTileH * kHrLdsPitch];
    __shared__ int spill_slot;
    const int band = threadIdx.x >> 6;
    const int g = blockIdx.x % kTileGroups;          // this workgroup's number within the plane
    const int plane = blockIdx.x / kTileGroups;
    const int tpp = tiles_x * tiles_y;
    const int n = act_count[plane];
    const float* A = act + (size_t)plane * 4 * HW;
    float* hr_plane = hr + (size_t)plane * rows * pitch;
    const bool pooled = pool.slot != nullptr;
    const bool stateful = tile_cur != nullptr;
    const bool valid = stateful && !pooled && ws_header[2] == 0ull;
    int k = 0;                                       // running index of the tiles that need work, dealt round-robin
    const int lane = threadIdx.x & 63;
    // Pooled map: the touched tiles of the image's planes stand one after the other in its pool, in plane order and, inside
    // a plane, in bitmap order -- the k-th touched tile of this plane has slot (tiles of the planes before it) + k.
    int plane_base = 0;
    float* pool_image = nullptr;
    if (pooled) {
        const int b = plane / F, f = plane - b * F;
        int before = 0;                              // (bitmap words of the planes before this one: a handful of loads per lane)
        const unsigned* img = tile_cur + (size_t)b * F * touch_words;
        for (int q = lane; q < f * touch_words; q += 64) before += __popc(img[q]);
        for (int d = 32; d > 0; d >>= 1) before += __shfl_xor(before, d, 64);
        plane_base = before;
        pool_image = hr + (size_t)b * pool.cap * (kHrTileH * kHrTileW);
    }
    for (int w0 = 0; w0 < touch_words; w0 += 64) {   // 64 bitmap words per step: one load per lane, then readlane
        const int wl = w0 + lane;
        unsigned my_need = 0u, my_cur = 0u;
        if (wl < touch_words) {
            const unsigned in_range = tpp - wl * 32 >= 32 ? 0xFFFFFFFFu : (1u << (tpp - wl * 32)) - 1u;
            my_cur = stateful ? tile_cur[(size_t)plane * touch_words + wl] : in_range;
            const unsigned prev = pooled ? 0u : valid ? tile_prev[(size_t)plane * touch_words + wl] : in_range;   // (a pool has no stale tiles)
            my_need = (my_cur | prev) & in_range;
        }
        unsigned long long words = __ballot(my_need != 0u);
        while (words) {
            const int wi = __builtin_ctzll(words);
            words &= words - 1;
            unsigned need = (unsigned)__builtin_amdgcn_readlane((int)my_need, wi);
            const unsigned cur = (unsigned)__builtin_amdgcn_readlane((int)my_cur, wi);
            while (need) {
                const int bit = __builtin_ctz(need);
                need &= need - 1;
                if ((k++ % kTileGroups) != g) continue;
                const int t = (w0 + wi) * 32 + bit;
                const int ty = t / tiles_x, tx = t - ty * tiles_x;
                if (pooled) {
                    int sl = plane_base + k - 1;
                    if (sl >= pool.cap) {             // more tiles than the image's pool holds: a slot of the batch's spill region
                        if (threadIdx.x == 0) spill_slot = pool.spill_cap > 0 ? atomicAdd(pool.spill_count, 1) : pool.spill_cap;
                        __syncthreads();              // (every thread of the workgroup walks the same tiles)
                        const int i = spill_slot;
                        __syncthreads();
                        if (i >= pool.spill_cap) {    // that ran out too: the image is flagged, not decoded wrongly
                            if (threadIdx.x == 0) { pool.slot[(size_t)plane * tpp + t] = -2; pool.overflow[plane / F] = 1; }
                            continue;
                        }
                        sl = (pool.images - plane / F) * pool.cap + i;
                    }
                    if (threadIdx.x == 0) pool.slot[(size_t)plane * tpp + t] = sl;
                    build_tile(A, n, HW, T, pool_image + (size_t)sl * (kHrTileH * kHrTileW), kHrTileW, false, rows, cols, tx, ty, band, true);
                } else {
                    build_tile(A, n, HW, T, hr_plane + (size_t)ty * kHrTileH * pitch + tx * kHrTileW, pitch, true, rows, cols, tx, ty,
                               band, (cur >> bit) & 1u);
                }
            }
        }
    }
}

// ---------------------------------------------------------------- pass 2, the decode path: one tile per WAVE
// The pooled map's tile kernel.  cif_active_kernel<., true> left a work list of (plane * tpp + tile, slot) -- every tile a cell of
// the batch reaches, once.  A persistent grid walks it, one tile per wave at a time (four independent waves per workgroup, no
// barrier anywhere): the wave walks the plane's active list 64 cells per step, ballots "box overlaps my tile" and applies the
// overlapping cells IN LIST ORDER to the whole 32x64 tile in its LDS block -- every pixel sees its contributions in the
// reference's order: bit-identical map, as before -- then writes the tile to its pool slot.
//
// Round 5's kernel was bound by its VALU instruction count, not by memory (counters; 8 workgroups per plane, each of the four
// waves of a workgroup one 8-row band of a tile at a time): a cell's box of ~15x15 pixels spans two or three bands, and every
// band's wave paid the per-cell set-up (eight v_readlane, the clipping) for one or two 16x4-pixel steps.  Here a cell is set up
// ONCE per tile, and
//  * the lanes' patch follows the box: 8x8 pixels for boxes up to 8 wide, 16x4 up to 16, 32x2, 64x1 (a 7x7 box: one step at
//    49 of 64 lanes instead of two at 28; the LDS pitch of 80 floats keeps every shape bank-conflict free: lanes conflict only
//    within a half wave, whose rows start 16 banks apart);
//  * the correctly rounded quotient -0.5 d^2 / sigma^2 (cif_hr.cpp:81; the reference computes it in double and rounds to float,
//    which for one division of float operands IS the correctly rounded float quotient) takes three instructions instead of the
//    compiler's eleven: with r = RN(1 / sigma^2) -- once per cell -- q0 = a r, rem = fma(-sigma^2, q0, a) (exact), q = fma(rem, r, q0)
//    is RN(a / sigma^2) (Markstein's theorem: r correctly rounded, q0 within one ulp -- it is, for every operand -- and the
//    significand of sigma^2 not all ones; such a cell takes the compiler's division; no under- or overflow here: a in
//    [-0.5 sigma^2, -0.125], sigma^2 >= 1).  tests/test_exact_division_model.py checks the identity on 10^7 operand pairs of this
//    kernel's domain, the bit-exact map tests and the randomised sweeps check the kernel;
//  * approx_exp's range test is gone: d^2 <= sigma^2 inside the circle, so its argument lies in [-0.5, 0].
// (Measured and dropped: HALF a tile per wave for small batches -- twice the work items at half the LDS each, eight waves per
// workgroup -- 45.8 us against 42.8 us for 32 images: the launch is bound by its most crowded tiles' chains of cells, which a
// split by rows does not shorten, not by the number of tiles in flight.)
// the rare cell whose sigma^2 has an all-ones significand (Markstein's exception): the compiler's division, the plain loop
__device__ __noinline__ void apply_cell_tile_slow(float* __restrict__ T, float bv, float bx, float by, float sigma2, int bx0, int bx1,
                                                  int by0, int by1, int x0, int y0, int lane) {
    const int lx = lane & 15, ly = lane >> 4;
    for (int py = by0; py < by1; py += 4) {
        const int yy = py + ly;
        const float dy = (float)yy - by;
        const float dy2 = dy * dy;
        for (int px = bx0; px < bx1; px += 16) {
            const int xx = px + lx;
            if (xx < bx1 && yy < by1) {
                const float dx = (float)xx - bx;
                const float dx2 = dx * dx;
                const float d2 = dx2 + dy2;
                if (!(d2 > sigma2)) {
                    float vv;
                    if (dx2 < 0.25f && dy2 < 0.25f) vv = bv;
                    else vv = bv * approx_exp(__fdiv_rn(-0.5f * d2, sigma2));
                    float* e = T + (yy - y0) * kHrLdsPitch + (xx - x0);
                    const float acc = fmaxf(*e, 1.0f) + vv;
                    *e = fminf(acc, 2.0f);
                }
            }
        }
    }
}

// One cell applied to a wave's whole tile.  The patch is as wide as the clipped box (a power of two: 8, 16, 32 or 64 columns --
// the box never leaves the tile's 64), so a lane keeps its column and the loop runs over rows alone: branch-free, the tile value
// of the NEXT step read (from a clamped address: rows of one cell's steps never overlap) before this step's arithmetic.
__device__ __forceinline__ void apply_cell_tile(float* __restrict__ T, float bv, float bx, float by, float sigma2, float rcp,
                                                int bx0, int bx1, int by0, int by1, int sh, int x0, int y0, int lane) {
    const int ph = 64 >> sh;
    const int lx = lane & ((1 << sh) - 1), ly = lane >> sh;
    const int xx = bx0 + lx;
    const bool xin = xx < bx1;
    const float dx = (float)xx - bx;
    const float dx2 = dx * dx;
    const bool cx = dx2 < 0.25f;
    float* col = T + (min(xx, x0 + kHrTileW - 1) - x0);
    int yy = by0 + ly;
    float cur = col[min(yy - y0, kHrTileH - 1) * kHrLdsPitch];
    for (int py = by0; py < by1; py += ph) {
        const float nxt = col[min(yy + ph - y0, kHrTileH - 1) * kHrLdsPitch];
        const float dy = (float)yy - by;
        const float dy2 = dy * dy;
        const float d2 = dx2 + dy2;
        const bool ok = xin && yy < by1 && !(d2 > sigma2);        // cif_hr.cpp:75
        const float a = -0.5f * d2;                               // :81  (-0.5 * d2 / sigma2, correctly rounded)
        const float q0 = a * rcp;
        const float q = __fmaf_rn(__fmaf_rn(-sigma2, q0, a), rcp, q0);
        float e8 = 1.0f + q * 0.125f;                             // approx_exp, cif_hr.cpp:18-25 (|q| <= 0.5: no range test)
        e8 *= e8; e8 *= e8; e8 *= e8;
        const float vv = (cx && dy2 < 0.25f) ? bv : bv * e8;      // :77-81
        const float acc = fminf(fmaxf(cur, 1.0f) + vv, 2.0f);     // :84-86 at revision 1.0
        if (ok) col[(yy - y0) * kHrLdsPitch] = acc;
        cur = nxt; yy += ph;
    }
}

__global__ __launch_bounds__(256) void cifhr_worktile_kernel(
        const float* __restrict__ act, const int32_t* __restrict__ act_count, int HW,
        float* __restrict__ hr, int rows, int cols, int tiles_x, HrPool pool, int F) {
    __shared__ __attribute__((aligned(16))) float Tall[4][kHrTileH * kHrLdsPitch];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* T = Tall[wave];
    const int n_items = *pool.work_count;
    const int step = gridDim.x * 4;
    int item = blockIdx.x * 4 + wave;
    int2 wk = item < n_items ? pool.work[item] : make_int2(0, -1);
    for (; item < n_items; item += step) {
        const int2 cur = wk;
        if (item + step < n_items) wk = pool.work[item + step];       // the next tile's entry travels while this one is built
        if (cur.y < 0) continue;                      // (no slot left for this tile: its image is flagged)
        const int plane = cur.x / pool.tpp, t = cur.x - plane * pool.tpp;
        const int ty = t / tiles_x, tx = t - ty * tiles_x;
        const int b = plane / F;
        const int n = act_count[plane];
        const float* A = act + (size_t)plane * 4 * HW;
        float* out = hr + ((size_t)b * pool.cap + cur.y) * (kHrTileH * kHrTileW);
        const int x0 = tx * kHrTileW, y0 = ty * kHrTileH;
        const int x1 = min(x0 + kHrTileW, cols), y1 = min(y0 + kHrTileH, rows);
        // the first list chunk is in flight while the tile is zeroed
        float nv = 0.f, nx = 0.f, ny = 0.f, ns = 1.f;
        if (lane < n) { nv = A[0 * HW + lane]; nx = A[1 * HW + lane]; ny = A[2 * HW + lane]; ns = A[3 * HW + lane]; }
        for (int k = lane; k < kHrTileH * kHrLdsPitch / 4; k += 64)
            reinterpret_cast<float4*>(T)[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int c0 = 0; c0 < n; c0 += 64) {
            const int i = c0 + lane;
            const float v16 = nv, cx = nx, cy = ny, sigma = ns;
            if (i + 64 < n) {
                nv = A[0 * HW + i + 64]; nx = A[1 * HW + i + 64]; ny = A[2 * HW + i + 64]; ns = A[3 * HW + i + 64];
            }
            // everything a cell needs is computed for 64 cells at once, lane = cell: its box clipped to the tile, the patch
            // shape, sigma^2 and its correctly rounded reciprocal; applying a cell then starts with ten v_readlane and nothing else
            int bx0 = 0, bx1 = 0, by0 = 0, by1 = 0, shf = 4;
            float sigma2 = 1.f, rcp = 1.f;
            bool hit = false;
            if (i < n) {
                int minx, miny, maxx, maxy;
                gauss_box(cx, cy, sigma, rows, cols, &minx, &miny, &maxx, &maxy);
                bx0 = max(minx, x0); bx1 = min(maxx, x1); by0 = max(miny, y0); by1 = min(maxy, y1);
                hit = bx0 < bx1 && by0 < by1;
                const int bw = bx1 - bx0;
                shf = bw <= 8 ? 3 : bw <= 16 ? 4 : bw <= 32 ? 5 : 6;
                sigma2 = sigma * sigma;                              // cif_hr.cpp:66-67
                rcp = __fdiv_rn(1.0f, sigma2);                       // RN(1 / sigma^2)
                if ((__float_as_uint(sigma2) & 0x7FFFFFu) == 0x7FFFFFu) shf |= 8;   // Markstein's exception: the slow routine
            }
            unsigned long long mask = __ballot(hit);
            while (mask) {
                const int l = __builtin_ctzll(mask);
                mask &= mask - 1;
                auto rl = [l](float v) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); };
                auto ri = [l](int v) { return __builtin_amdgcn_readlane(v, l); };
                const int s_sh = ri(shf);
                if (s_sh & 8) apply_cell_tile_slow(T, rl(v16), rl(cx), rl(cy), rl(sigma2), ri(bx0), ri(bx1), ri(by0), ri(by1), x0, y0, lane);
                else apply_cell_tile(T, rl(v16), rl(cx), rl(cy), rl(sigma2), rl(rcp), ri(bx0), ri(bx1), ri(by0), ri(by1), s_sh, x0, y0, lane);
            }
        }
        // coalesced write-out: 16 lanes x float4 = one 256-B tile row, 4 rows per instruction (rows below the map too: the slot is the tile's)
        const int lx = lane & 15, ly = lane >> 4;
        for (int r = ly; r < kHrTileH; r += 4) {
            const float4 val = *reinterpret_cast<const float4*>(T + r * kHrLdsPitch + lx * 4);
            *reinterpret_cast<float4*>(out + (size_t)r * kHrTileW + lx * 4) = val;
        }
    }
}

// this call's touched-tile bitmap becomes the next call's "previous" (a kernel rather than a D2D
// hipMemcpyAsync: small copy nodes inside a captured HIP graph fault on replay with ROCm 7.2)
__global__ __launch_bounds__(256) void tile_state_roll_kernel(unsigned* __restrict__ prev, const unsigned* __restrict__ cur, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) prev[i] = cur[i];
}

hipError_t launch_cifhr(const float* cif, int B, int F, int H, int W, int stride,
                        double min_scale, double factor, const DevParams& p,
                        float* cifhr, int hr_rows, int hr_pitch,
                        float* act, int32_t* act_count, hipStream_t st, bool det,
                        unsigned long long* ws_header, unsigned long long layout_hash, unsigned char* tile_state,
                        int32_t* zero_per_image, const HrPool* pool_in, SeedCandidates* cand) {
    const int planes = B * F, HW = H * W;
    const int hr_cols = (W - 1) * stride + 1;
    const int tiles_x = hr_pitch / kHrTileW;
    const int tiles_y = (hr_rows + kHrTileH - 1) / kHrTileH;
    // two per-plane tile bitmaps in the workspace region `tile_state`: previous call, this call
    const int touch_words = (tiles_x * tiles_y + 31) / 32;
    HrPool pool; pool.slot = nullptr; pool.overflow = nullptr; pool.cap = 0; pool.tpp = tiles_x * tiles_y;
    pool.spill_cap = 0; pool.images = B; pool.spill_count = nullptr; pool.work = nullptr; pool.work_count = nullptr; pool.img_tiles = nullptr;
    if (pool_in) pool = *pool_in;
    // the decode path: tiles through a work list, seed candidates on the way (see cif_active_kernel)
    const bool worklist = pool.slot && pool.work && cand && cand->cand && !det && (size_t)touch_words * sizeof(unsigned) <= 48 * 1024;
    unsigned* tile_prev = ws_header ? reinterpret_cast<unsigned*>(tile_state) : nullptr;
    unsigned* tile_touch = ws_header ? tile_prev + (size_t)planes * touch_words : nullptr;
    if (worklist) {                                   // overflow flags [B], spill counter, work counter, slot counters [B]: one region
        hipError_t e = launch_zero(pool.overflow, sizeof(int32_t) * (2 * (size_t)B + 2), st);
        if (e != hipSuccess) return e;
    }
    if (p.ablation_cifhr_skip && !det) {              // cif_hr.cpp:29
        hipError_t e = hipMemsetAsync(act_count, 0, sizeof(int32_t) * planes, st);
        if (e != hipSuccess) return e;
        if (zero_per_image) { e = hipMemsetAsync(zero_per_image, 0, sizeof(int32_t) * B, st); if (e != hipSuccess) return e; }
        if (pool.slot) {                              // pooled map, no cell: no tile has a slot
            e = hipMemsetAsync(pool.slot, 0xFF, sizeof(int32_t) * (size_t)planes * pool.tpp, st);
            if (e != hipSuccess) return e;
            e = hipMemsetAsync(pool.overflow, 0, sizeof(int32_t) * (B + (pool.spill_count ? 1 : 0)), st);   // (+ the spill counter behind the flags)
            if (e != hipSuccess) return e;
        }
        prof_mark(st, "memset_act_count");
        if (ws_header) {                              // no kernel validates the flags on this path: invalidate them
            e = hipMemsetAsync(ws_header, 0xFF, 32, st);
            if (e != hipSuccess) return e;
            e = hipMemsetAsync(tile_touch, 0, sizeof(unsigned) * touch_words * planes, st);
            if (e != hipSuccess) return e;
        }
    }
    if (cand) cand->produced = false;                 // (no candidate lists: the seed fill streams the field itself)
    const float min_scale_f = (float)(min_scale / (double)stride);       // cif_hr.cpp:32
    if (p.ablation_cifhr_skip && !det) {
    } else if (worklist) {
        static_assert(kActiveThreads * kActiveCells == 256 * kFillCells, "a step of cif_active is a block of the seed fill");
        cif_active_kernel<false, true><<<planes, kActiveThreads, sizeof(unsigned) * touch_words, st>>>(
            cif, HW, stride, min_scale_f, p.cif_threshold, (float)p.cifhr_neighbors, factor, act, act_count, ws_header, layout_hash,
            tile_touch, touch_words, hr_rows, hr_cols, tiles_x, zero_per_image, F, pool,
            cand->cand, cand->start, cand->count, cand->chunks, p.seed_threshold);
        prof_mark(st, "cif_active_kernel");
        // a persistent grid over the work list: a compute unit holds four of these workgroups (40 KB of LDS each), sixteen tiles
        long long grid = ((long long)planes * pool.tpp + 3) / 4;
        if (grid > 256 * 4) grid = 256 * 4;
        cifhr_worktile_kernel<<<(int)grid, 256, 0, st>>>(act, act_count, HW, cifhr, hr_rows, hr_cols, tiles_x, pool, F);
        prof_mark(st, "cifhr_tile_kernel");
        cand->produced = true;
        return hipGetLastError();
    } else if (det)
        cif_active_kernel<true, false><<<planes, kActiveThreads, 0, st>>>(cif, HW, stride, min_scale_f, p.cif_threshold,
                                                         (float)p.cifhr_neighbors, factor, act, act_count, ws_header, layout_hash,
                                                         tile_touch, touch_words, hr_rows, hr_cols, tiles_x, zero_per_image, F, pool,
                                                         nullptr, nullptr, nullptr, 0, 0.0);
    else
        cif_active_kernel<false, false><<<planes, kActiveThreads, 0, st>>>(cif, HW, stride, min_scale_f, p.cif_threshold,
                                                          (float)p.cifhr_neighbors, factor, act, act_count, ws_header, layout_hash,
                                                          tile_touch, touch_words, hr_rows, hr_cols, tiles_x, zero_per_image, F, pool,
                                                          nullptr, nullptr, nullptr, 0, 0.0);
    if (!(p.ablation_cifhr_skip && !det)) prof_mark(st, "cif_active_kernel");
    cifhr_tile_kernel<<<planes * kTileGroups, 256, 0, st>>>(act, act_count, HW, cifhr, hr_rows, hr_cols, hr_pitch,
                                                             tiles_x, tiles_y, ws_header, tile_prev, tile_touch, touch_words, pool, F);
    if (ws_header && !pool.slot) {                    // this call's bitmap is the next call's "previous" (a pooled map keeps no state)
        const int n = touch_words * planes;
        tile_state_roll_kernel<<<(n + 255) / 256, 256, 0, st>>>(tile_prev, tile_touch, n);
    }
    prof_mark(st, "cifhr_tile_kernel");
    return hipGetLastError();
}

// get_cifhr of a pooled map: one image's [F][rows][cols] array (0.0 where no tile was built, like the dense buffer)
__global__ __launch_bounds__(256) void cifhr_gather_kernel(const float* __restrict__ pool_image, const int32_t* __restrict__ slot_image,
                                                           int F, int rows, int cols, int tiles_x, int tpp, float* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t n = (size_t)F * rows * cols;
    if (i >= n) return;
    const int x = (int)(i % cols), y = (int)((i / cols) % rows), f = (int)(i / ((size_t)cols * rows));
    const int sl = slot_image[(size_t)f * tpp + (y / kHrTileH) * tiles_x + x / kHrTileW];
    out[i] = sl < 0 ? 0.0f : pool_image[(size_t)sl * (kHrTileH * kHrTileW) + (y % kHrTileH) * kHrTileW + (x % kHrTileW)];
}

hipError_t launch_cifhr_gather(const float* pool_image, const int32_t* slot_image, int F, int rows, int cols, int tiles_x, int tpp,
                               float* out, hipStream_t st) {
    const size_t n = (size_t)F * rows * cols;
    cifhr_gather_kernel<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(pool_image, slot_image, F, rows, cols, tiles_x, tpp, out);
    return hipGetLastError();
}

}  // namespace opa
