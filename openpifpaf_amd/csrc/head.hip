// Fused head epilogue on gfx950 (producer side of the decode path).
//
// Replaces, in ONE pass, what the reference's CompositeField4 does after its 1x1 convolution
// (network/heads.py:330-378): PixelShuffle(upsample) -> crop of the last rows / columns -> view as
// [B, fields, components, H, W] -> float32 -> sigmoid on the confidences, cell-index offsets on the regression
// vectors, softplus on the scales.  The convolution output is channels-last ([B, Hc, Wc, fields * components *
// upsample^2], the layout the GEMM / MIOpen kernels write), so the pixel shuffle is a transpose: a workgroup loads
// one source row of 64 consecutive channels (coalesced 128-B / 256-B segments) into LDS and writes the 2 x 16 output
// rows of the 16 planes those channels feed (coalesced along x).  Reads 0.1 GB + writes 0.2 GB per batch of 32
// instead of five PyTorch passes over the 0.2 GB field tensors.
#include "common.hpp"

#include <hip/hip_fp16.h>

namespace opa {

constexpr int kHeadPlanes = 16;          // output planes (field, component) per workgroup
constexpr int kHeadThreads = 256;

__device__ __forceinline__ float head_load(const void* p, size_t i, int dtype) {
    if (dtype == 0) return reinterpret_cast<const float*>(p)[i];
    const unsigned short h = reinterpret_cast<const unsigned short*>(p)[i];
    if (dtype == 2) return __uint_as_float((unsigned)h << 16);                        // bfloat16
    _Float16 f; __builtin_memcpy(&f, &h, 2);
    return (float)f;                                                                   // float16
}

template <int US>
__global__ __launch_bounds__(kHeadThreads) void head_epilogue_kernel(
        const void* __restrict__ conv, int dtype, int Hc, int Wc, int n_planes, int n_comp,
        int n_conf, int n_vec, unsigned offset_mask, int n_scales, int Ho, int Wo, int low_cut,
        float* __restrict__ out) {
    constexpr int SUB = US * US;                         // source channels per output plane
    extern __shared__ float tile[];                      // [kHeadPlanes * SUB][Wc + 1]
    const int pitch = Wc + 1;
    const int bh = blockIdx.x, b = bh / Hc, h = bh - b * Hc;
    const int p0 = blockIdx.y * kHeadPlanes;
    const int np = min(kHeadPlanes, n_planes - p0);
    const int nch = np * SUB, Ctot = n_planes * SUB;
    const size_t src_row = ((size_t)b * Hc + h) * Wc * Ctot + (size_t)p0 * SUB;
    for (int k = threadIdx.x; k < Wc * nch; k += kHeadThreads) {
        const int w = k / nch, ch = k - w * nch;
        tile[ch * pitch + w] = head_load(conv, src_row + (size_t)w * Ctot + ch, dtype);
    }
    __syncthreads();
    const int row_len = Wo;
    for (int k = threadIdx.x; k < np * US * row_len; k += kHeadThreads) {
        const int x = k % row_len, pi = k / row_len, i = pi % US, p = pi / US;
        const int ys = h * US + i, y = ys - low_cut;     // row after the pixel shuffle, after the crop
        if (y < 0 || y >= Ho) continue;
        const int xs = x + low_cut, j = xs % US, w = xs / US;
        float v = tile[(p * SUB + i * US + j) * pitch + w];
        const int plane = p0 + p, c = plane % n_comp;    // component within its field
        if (c >= 1 && c < 1 + n_conf) {
            v = 1.0f / (1.0f + expf(-v));                // sigmoid (heads.py:364)
        } else if (c >= 1 + n_conf && c < 1 + n_conf + 2 * n_vec) {
            const int vi = (c - 1 - n_conf) >> 1, is_y = (c - 1 - n_conf) & 1;
            if ((offset_mask >> vi) & 1u) v += is_y ? (float)y : (float)x;              // heads.py:366-370
        } else if (c >= 1 + n_conf + 2 * n_vec && c < 1 + n_conf + 2 * n_vec + n_scales) {
            v = v > 20.0f ? v : log1pf(expf(v));         // softplus, beta 1, threshold 20 (heads.py:374)
        }
        out[(((size_t)b * n_planes + plane) * Ho + y) * Wo + x] = v;
    }
}

hipError_t launch_head_epilogue(const void* conv, int dtype, int B, int Hc, int Wc, int n_fields, int n_comp, int us,
                                int n_conf, int n_vec, unsigned offset_mask, int n_scales, float* out, hipStream_t st) {
    const int n_planes = n_fields * n_comp;
    const int low_cut = (us - 1) / 2, high_cut = us - 1 - low_cut;      // heads.py:336-343
    const int Ho = Hc * us - low_cut - high_cut, Wo = Wc * us - low_cut - high_cut;
    dim3 grid(B * Hc, (n_planes + kHeadPlanes - 1) / kHeadPlanes);
    const size_t lds = sizeof(float) * kHeadPlanes * us * us * (Wc + 1);
    if (lds > 64 * 1024) return hipErrorInvalidValue;
    if (us == 1)
        head_epilogue_kernel<1><<<grid, kHeadThreads, lds, st>>>(conv, dtype, Hc, Wc, n_planes, n_comp, n_conf, n_vec,
                                                                 offset_mask, n_scales, Ho, Wo, low_cut, out);
    else if (us == 2)
        head_epilogue_kernel<2><<<grid, kHeadThreads, lds, st>>>(conv, dtype, Hc, Wc, n_planes, n_comp, n_conf, n_vec,
                                                                 offset_mask, n_scales, Ho, Wo, low_cut, out);
    else
        return hipErrorInvalidValue;
    prof_mark(st, "head_epilogue_kernel");
    return hipGetLastError();
}

}  // namespace opa
