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

// ---------------------------------------------------------------- pass 1
// DET: CifDet fields [F,6,H,W] (w,h instead of scale), CifDetHr::accumulate cif_hr.cpp:124-150
template <bool DET>
__global__ __launch_bounds__(256) void cif_active_kernel(
        const float* __restrict__ cif, int HW, int stride, float min_scale_f, double threshold,
        float neighbors_f, double factor, float* __restrict__ act, int32_t* __restrict__ act_count) {
    __shared__ int wave_tot[4];
    const int plane = blockIdx.x;
    const float* P = cif + (size_t)plane * (DET ? 6 : 5) * HW;
    float* out = act + (size_t)plane * 4 * HW;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const float stride_f = (float)stride;
    int base = 0;
    for (int c0 = 0; c0 < HW; c0 += 256) {
        const int o = c0 + tid;
        bool on = false;
        float v16 = 0.f, x = 0.f, y = 0.f, sigma = 0.f;
        if (o < HW) {
            const float v = P[HW + o];
            if (!((double)v < threshold)) {                       // cif_hr.cpp:39
                const float scale = P[4 * HW + o];
                bool big_enough;
                double sigma_d;
                if (DET) {                                        // cif_hr.cpp:135-141
                    const float h = P[5 * HW + o];
                    big_enough = !(scale < min_scale_f || h < min_scale_f);
                    sigma_d = 0.1 * (double)fminf(scale, h) * (double)stride;
                } else {                                          // cif_hr.cpp:42,46
                    big_enough = !(scale < min_scale_f);
                    sigma_d = 0.5 * (double)scale * (double)stride;
                }
                if (big_enough) {
                    on = true;
                    x = P[2 * HW + o] * stride_f;                 // cif_hr.cpp:44-45
                    y = P[3 * HW + o] * stride_f;
                    sigma = fmaxf(1.0f, (float)sigma_d);
                    v16 = (float)((double)(v / neighbors_f) * factor);                    // :51
                }
            }
        }
        const unsigned long long mask = __ballot(on);
        const int pre = __popcll(mask & ((1ull << lane) - 1ull));
        if (lane == 0) wave_tot[w] = __popcll(mask);
        __syncthreads();
        int off = base + pre, tot = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) { const int t = wave_tot[k]; if (k < w) off += t; tot += t; }
        if (on) {
            out[0 * HW + off] = v16; out[1 * HW + off] = x;
            out[2 * HW + off] = y;   out[3 * HW + off] = sigma;
        }
        base += tot;
        __syncthreads();
    }
    if (tid == 0) act_count[plane] = base;
}

// cif_hr.cpp:18-25
__device__ __forceinline__ float approx_exp(float x) {
    if ((double)x > 2.0 || (double)x < -2.0) return 0.0f;
    x = (float)(1.0 + (double)x / 8.0);
    x *= x; x *= x; x *= x;
    return x;
}

// ---------------------------------------------------------------- pass 2
__global__ __launch_bounds__(256) void cifhr_tile_kernel(
        const float* __restrict__ act, const int32_t* __restrict__ act_count, int HW,
        float* __restrict__ hr, int rows, int cols, int pitch,
        int tiles_x, int tiles_y, long long total_tiles) {
    __shared__ __attribute__((aligned(16))) float lds[4 * kHrTileH * kHrLdsPitch];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long long tile = (long long)blockIdx.x * 4 + w;
    if (tile >= total_tiles) return;                 // no workgroup barriers below
    const int tpp = tiles_x * tiles_y;
    const int plane = (int)(tile / tpp);
    const int rem = (int)(tile - (long long)plane * tpp);
    const int ty = rem / tiles_x, tx = rem - ty * tiles_x;
    const int x0 = tx * kHrTileW, y0 = ty * kHrTileH;
    const int x1 = min(x0 + kHrTileW, cols), y1 = min(y0 + kHrTileH, rows);
    float* T = lds + w * (kHrTileH * kHrLdsPitch);

    for (int k = lane; k < kHrTileH * kHrLdsPitch / 4; k += 64)
        reinterpret_cast<float4*>(T)[k] = make_float4(0.f, 0.f, 0.f, 0.f);

    const int n = act_count[plane];
    const float* A = act + (size_t)plane * 4 * HW;
    const int lx = lane & 15, ly = lane >> 4;

    for (int c0 = 0; c0 < n; c0 += 64) {
        const int i = c0 + lane;
        float v16 = 0.f, cx = 0.f, cy = 0.f, sigma = 1.f;
        int minx = 0, maxx = 0, miny = 0, maxy = 0;
        bool hit = false;
        if (i < n) {
            v16 = A[0 * HW + i]; cx = A[1 * HW + i]; cy = A[2 * HW + i]; sigma = A[3 * HW + i];
            // cif_hr.cpp:61-64 (truncate = 1.0)
            minx = (int)clampll(trunc_ll(cx - sigma), 0, cols - 1);
            miny = (int)clampll(trunc_ll(cy - sigma), 0, rows - 1);
            maxx = (int)clampll(trunc_ll(cx + sigma + 1.0f), minx + 1, cols);
            maxy = (int)clampll(trunc_ll(cy + sigma + 1.0f), miny + 1, rows);
            hit = minx < x1 && maxx > x0 && miny < y1 && maxy > y0;
        }
        unsigned long long mask = __ballot(hit);
        while (mask) {
            const int l = __builtin_ctzll(mask);
            mask &= mask - 1;
            const float bv = __shfl(v16, l), bx = __shfl(cx, l), by = __shfl(cy, l), bs = __shfl(sigma, l);
            const int bx0 = max(__shfl(minx, l), x0), bx1 = min(__shfl(maxx, l), x1);
            const int by0 = max(__shfl(miny, l), y0), by1 = min(__shfl(maxy, l), y1);
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
                            if ((double)dx2 < 0.25 && (double)dy2 < 0.25) vv = bv;   // :77-79
                            else vv = bv * approx_exp((float)(-0.5 * (double)d2 / (double)sigma2));
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
    for (int r = ly; r < kHrTileH; r += 4) {
        const int yy = y0 + r;
        if (yy < rows) {
            const float4 val = *reinterpret_cast<const float4*>(T + r * kHrLdsPitch + lx * 4);
            *reinterpret_cast<float4*>(hr + ((size_t)plane * rows + yy) * pitch + x0 + lx * 4) = val;
        }
    }
}

hipError_t launch_cifhr(const float* cif, int B, int F, int H, int W, int stride,
                        double min_scale, double factor, const DevParams& p,
                        float* cifhr, int hr_rows, int hr_pitch,
                        float* act, int32_t* act_count, hipStream_t st, bool det) {
    const int planes = B * F, HW = H * W;
    const int hr_cols = (W - 1) * stride + 1;
    if (p.ablation_cifhr_skip && !det) {              // cif_hr.cpp:29
        hipError_t e = hipMemsetAsync(act_count, 0, sizeof(int32_t) * planes, st);
        if (e != hipSuccess) return e;
        prof_mark(st, "memset_act_count");
    } else {
        const float min_scale_f = (float)(min_scale / (double)stride);       // cif_hr.cpp:32
        if (det)
            cif_active_kernel<true><<<planes, 256, 0, st>>>(cif, HW, stride, min_scale_f, p.cif_threshold,
                                                             (float)p.cifhr_neighbors, factor, act, act_count);
        else
            cif_active_kernel<false><<<planes, 256, 0, st>>>(cif, HW, stride, min_scale_f, p.cif_threshold,
                                                              (float)p.cifhr_neighbors, factor, act, act_count);
        prof_mark(st, "cif_active_kernel");
    }
    const int tiles_x = hr_pitch / kHrTileW;
    const int tiles_y = (hr_rows + kHrTileH - 1) / kHrTileH;
    const long long total = (long long)planes * tiles_x * tiles_y;
    const unsigned grid = (unsigned)((total + 3) / 4);
    cifhr_tile_kernel<<<grid, 256, 0, st>>>(act, act_count, HW, cifhr, hr_rows, hr_cols, hr_pitch,
                                             tiles_x, tiles_y, total);
    prof_mark(st, "cifhr_tile_kernel");
    return hipGetLastError();
}

}  // namespace opa
