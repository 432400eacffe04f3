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

template <bool DET>
__global__ __launch_bounds__(kActiveThreads) void cif_active_kernel(
        const float* __restrict__ cif, int HW, int stride, float min_scale_f, double threshold,
        float neighbors_f, double factor, float* __restrict__ act, int32_t* __restrict__ act_count,
        unsigned long long* ws_header, unsigned long long layout_hash,
        unsigned* __restrict__ tile_touch, int touch_words, int rows, int cols, int tiles_x,
        int32_t* __restrict__ zero_per_image, int F, HrPool pool) {
    __shared__ int wave_tot[2][kActiveCells][kActiveThreads / 64];
    const int plane = blockIdx.x;
    if (zero_per_image && threadIdx.x == 0 && plane % F == 0) zero_per_image[plane / F] = 0;   // the image's seed counter
    if (pool.slot) {                                  // pooled map: no tile of this plane has a slot yet
        for (int k = threadIdx.x; k < pool.tpp; k += kActiveThreads) pool.slot[(size_t)plane * pool.tpp + k] = -1;
        if (threadIdx.x == 0 && plane % F == 0) pool.overflow[plane / F] = 0;
        if (threadIdx.x == 0 && plane == 0 && pool.spill_count) *pool.spill_count = 0;
    }
    unsigned* touch = tile_touch ? tile_touch + (size_t)plane * touch_words : nullptr;   // one bit per tile of this plane
    if (touch) {
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
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const float stride_f = (float)stride;
    int base = 0;
    int parity = 0;
    // kActiveCells cells per thread and step: their confidence loads are in flight together; cell order
    // (r, wave, lane) is raster order, and the list keeps it
    for (int c0 = 0; c0 < HW; c0 += kActiveThreads * kActiveCells, parity ^= 1) {
        // all four planes of every cell are requested at once (they are this kernel's compulsory bytes anyway):
        // one memory round trip per step instead of two dependent ones
        float vin[kActiveCells], xin[kActiveCells], yin[kActiveCells], sin_[kActiveCells], hin[kActiveCells];
#pragma unroll
        for (int r = 0; r < kActiveCells; r++) {
            const int o = c0 + r * kActiveThreads + tid;
            const int oo = o < HW ? o : 0;
            vin[r] = o < HW ? P[HW + oo] : -1.0f;
            xin[r] = P[2 * HW + oo]; yin[r] = P[3 * HW + oo]; sin_[r] = P[4 * HW + oo];
            hin[r] = DET ? P[5 * HW + oo] : 0.0f;
        }
        bool on[kActiveCells];
        float v16[kActiveCells], x[kActiveCells], y[kActiveCells], sigma[kActiveCells];
#pragma unroll
        for (int r = 0; r < kActiveCells; r++) {
            const int o = c0 + r * kActiveThreads + tid;
            on[r] = false; v16[r] = 0.f; x[r] = 0.f; y[r] = 0.f; sigma[r] = 0.f;
            const float v = vin[r];
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
                    if (touch) {                                  // tiles this cell's box overlaps
                        int minx, miny, maxx, maxy;
                        gauss_box(x[r], y[r], sigma[r], rows, cols, &minx, &miny, &maxx, &maxy);
                        for (int ty = miny / kHrTileH; ty <= (maxy - 1) / kHrTileH; ty++)
                            for (int tx = minx / kHrTileW; tx <= (maxx - 1) / kHrTileW; tx++) {
                                const int t = ty * tiles_x + tx;
                                atomicOr(&touch[t >> 5], 1u << (t & 31));
                            }
                    }
                }
            }
        }
        unsigned long long mask[kActiveCells];
#pragma unroll
        for (int r = 0; r < kActiveCells; r++) {
            mask[r] = __ballot(on[r]);
            if (lane == 0) wave_tot[parity][r][w] = __popcll(mask[r]);
        }
        __syncthreads();                              // double-buffered totals: one barrier per step
#pragma unroll
        for (int r = 0; r < kActiveCells; r++) {
            int off = base + __popcll(mask[r] & ((1ull << lane) - 1ull)), tot = 0;
#pragma unroll
            for (int k = 0; k < kActiveThreads / 64; k++) { const int t = wave_tot[parity][r][k]; if (k < w) off += t; tot += t; }
            if (on[r]) {
                out[0 * HW + off] = v16[r]; out[1 * HW + off] = x[r];
                out[2 * HW + off] = y[r];   out[3 * HW + off] = sigma[r];
            }
            base += tot;
        }
    }
    if (tid == 0) act_count[plane] = base;
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
                        int32_t* zero_per_image, const HrPool* pool_in) {
    const int planes = B * F, HW = H * W;
    const int hr_cols = (W - 1) * stride + 1;
    const int tiles_x = hr_pitch / kHrTileW;
    const int tiles_y = (hr_rows + kHrTileH - 1) / kHrTileH;
    // two per-plane tile bitmaps in the workspace region `tile_state`: previous call, this call
    const int touch_words = (tiles_x * tiles_y + 31) / 32;
    HrPool pool; pool.slot = nullptr; pool.overflow = nullptr; pool.cap = 0; pool.tpp = tiles_x * tiles_y;
    pool.spill_cap = 0; pool.images = B; pool.spill_count = nullptr;
    if (pool_in) pool = *pool_in;
    unsigned* tile_prev = ws_header ? reinterpret_cast<unsigned*>(tile_state) : nullptr;
    unsigned* tile_touch = ws_header ? tile_prev + (size_t)planes * touch_words : nullptr;
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
    } else {
        const float min_scale_f = (float)(min_scale / (double)stride);       // cif_hr.cpp:32
        if (det)
            cif_active_kernel<true><<<planes, kActiveThreads, 0, st>>>(cif, HW, stride, min_scale_f, p.cif_threshold,
                                                             (float)p.cifhr_neighbors, factor, act, act_count, ws_header, layout_hash,
                                                             tile_touch, touch_words, hr_rows, hr_cols, tiles_x, zero_per_image, F, pool);
        else
            cif_active_kernel<false><<<planes, kActiveThreads, 0, st>>>(cif, HW, stride, min_scale_f, p.cif_threshold,
                                                              (float)p.cifhr_neighbors, factor, act, act_count, ws_header, layout_hash,
                                                              tile_touch, touch_words, hr_rows, hr_cols, tiles_x, zero_per_image, F, pool);
        prof_mark(st, "cif_active_kernel");
    }
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
