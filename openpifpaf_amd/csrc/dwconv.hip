// Depthwise k x k convolution and channel interleave on gfx950 (producer side: the ShuffleNetV2K backbones of
// BASELINE configs 3-5, reference network/basenetworks.py:186-268).
//
// MIOpen runs a depthwise convolution as a grouped MFMA convolution with one channel per group: 18-27 ms per
// layer at 641 px and batch 32 (236 of the 317 ms of a shufflenetv2k16 forward in float32, 349 of 400 ms in
// bfloat16).  The operation is a stencil -- 25 multiply-adds per output for two bytes moved: HBM-bound by a wide
// margin -- so it gets a plain stencil kernel: channels-last, one thread per channel and strip of four output
// pixels (a wave reads 64 consecutive channels: coalesced), the k rows of the window slide through registers,
// float32 accumulation, the folded batch-norm bias (and optionally ReLU) applied before the single store.
//
// channel_interleave_kernel fuses the unit's torch.cat + channel_shuffle(groups = 2): out[.., 2i] = a[.., i],
// out[.., 2i+1] = b[.., i].
#include "common.hpp"

namespace opa {

constexpr int kDwStrip = 4;              // output pixels per thread along x

template <typename T, int V> struct DwVec;                       // V consecutive channels of one pixel
template <int V> struct DwVec<float, V> {
    float v[V];
    __device__ __forceinline__ void load(const float* p) {
        if constexpr (V == 4) { const float4 t = *reinterpret_cast<const float4*>(p); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
        else if constexpr (V == 2) { const float2 t = *reinterpret_cast<const float2*>(p); v[0] = t.x; v[1] = t.y; }
        else v[0] = *p;
    }
    __device__ __forceinline__ void store(float* p) const {
        if constexpr (V == 4) *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
        else if constexpr (V == 2) *reinterpret_cast<float2*>(p) = make_float2(v[0], v[1]);
        else *p = v[0];
    }
};
template <int V> struct DwVec<unsigned short, V> {               // bfloat16
    float v[V];
    __device__ __forceinline__ void load(const unsigned short* p) {
        if constexpr (V == 4) {
            const uint2 t = *reinterpret_cast<const uint2*>(p);
            v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xFFFF0000u);
            v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xFFFF0000u);
        } else if constexpr (V == 2) {
            const unsigned t = *reinterpret_cast<const unsigned*>(p);
            v[0] = __uint_as_float(t << 16); v[1] = __uint_as_float(t & 0xFFFF0000u);
        } else v[0] = __uint_as_float((unsigned)*p << 16);
    }
    static __device__ __forceinline__ unsigned rne(float f) {
        const unsigned u = __float_as_uint(f);
        return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
    }
    __device__ __forceinline__ void store(unsigned short* p) const {
        if constexpr (V == 4) *reinterpret_cast<uint2*>(p) = make_uint2(rne(v[0]) | (rne(v[1]) << 16), rne(v[2]) | (rne(v[3]) << 16));
        else if constexpr (V == 2) *reinterpret_cast<unsigned*>(p) = rne(v[0]) | (rne(v[1]) << 16);
        else *p = (unsigned short)rne(v[0]);
    }
};

// x: [B, H, W, *] with pixel stride xs (a channel slice of a wider tensor is fine), w: [K*K, C] (tap-major),
// out: [B, Ho, Wo, *] with pixel stride os.  A work item = V channels x a strip of kDwStrip output pixels of one
// output row; the items of a row are dealt to the threads channel-vector first (coalesced within a pixel).
template <typename T, int K, int S, int V>
__global__ __launch_bounds__(256) void dwconv_kernel(const T* __restrict__ x, long long xs, const T* __restrict__ w,
                                                     const T* __restrict__ bias, T* __restrict__ out, long long os,
                                                     int H, int W, int C, int Ho, int Wo, int relu) {
    constexpr int P = K / 2;
    constexpr int NIN = (kDwStrip - 1) * S + K;          // input columns a strip of outputs needs
    const int cvecs = C / V, strips = (Wo + kDwStrip - 1) / kDwStrip;
    const int item = blockIdx.x * 256 + threadIdx.x;
    if (item >= cvecs * strips) return;
    const int strip = item / cvecs, c = (item - strip * cvecs) * V;
    const int x0 = strip * kDwStrip;
    const int by = blockIdx.y, b = by / Ho, y = by - b * Ho;
    float acc[kDwStrip][V];
    DwVec<T, V> bv;
#pragma unroll
    for (int q = 0; q < V; q++) bv.v[q] = 0.0f;
    if (bias) bv.load(bias + c);
#pragma unroll
    for (int i = 0; i < kDwStrip; i++)
#pragma unroll
        for (int q = 0; q < V; q++) acc[i][q] = 0.0f;
#pragma unroll
    for (int ky = 0; ky < K; ky++) {
        const int yy = y * S - P + ky;
        if (yy < 0 || yy >= H) continue;
        const T* row = x + ((size_t)b * H + yy) * W * xs + c;
        DwVec<T, V> in[NIN];
#pragma unroll
        for (int j = 0; j < NIN; j++) {
            const int xx = x0 * S - P + j;
#pragma unroll
            for (int q = 0; q < V; q++) in[j].v[q] = 0.0f;
            if (xx >= 0 && xx < W) in[j].load(row + (size_t)xx * xs);
        }
#pragma unroll
        for (int kx = 0; kx < K; kx++) {
            DwVec<T, V> wv;
            wv.load(w + (size_t)(ky * K + kx) * C + c);
#pragma unroll
            for (int i = 0; i < kDwStrip; i++)
#pragma unroll
                for (int q = 0; q < V; q++) acc[i][q] = fmaf(in[i * S + kx].v[q], wv.v[q], acc[i][q]);
        }
    }
#pragma unroll
    for (int i = 0; i < kDwStrip; i++) {
        const int xo = x0 + i;
        if (xo >= Wo) break;
        DwVec<T, V> o;
#pragma unroll
        for (int q = 0; q < V; q++) { o.v[q] = acc[i][q] + bv.v[q]; if (relu) o.v[q] = fmaxf(o.v[q], 0.0f); }
        o.store(out + (((size_t)b * Ho + y) * Wo + xo) * os + c);
    }
}

// V channels of a and of b per thread (V * sizeof(T)-byte loads, one 2V-element store)
template <typename T, int V>
__global__ __launch_bounds__(256) void channel_interleave_kernel(const T* __restrict__ a, long long as,
                                                                 const T* __restrict__ b, long long bs,
                                                                 T* __restrict__ out, long long rows, int half) {
    const int hv = half / V;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * hv) return;
    const long long r = i / hv;
    const int c = (int)(i - r * hv) * V;
    T va[V], vb[V], vo[2 * V];
    __builtin_memcpy(va, a + r * as + c, sizeof(va));
    __builtin_memcpy(vb, b + r * bs + c, sizeof(vb));
#pragma unroll
    for (int q = 0; q < V; q++) { vo[2 * q] = va[q]; vo[2 * q + 1] = vb[q]; }
    __builtin_memcpy(out + r * 2 * half + 2 * c, vo, sizeof(vo));
}

template <typename T, int V>
static hipError_t launch_dw_v(const void* x, long long xs, const void* w, const void* bias, void* out, long long os,
                              int B, int H, int W, int C, int K, int S, int relu, hipStream_t st) {
    const int P = K / 2;
    const int Ho = (H + 2 * P - K) / S + 1, Wo = (W + 2 * P - K) / S + 1;
    const long long items = (long long)(C / V) * ((Wo + kDwStrip - 1) / kDwStrip);
    dim3 grid((unsigned)((items + 255) / 256), B * Ho);
#define OPA_DW(KK, SS) dwconv_kernel<T, KK, SS, V><<<grid, 256, 0, st>>>((const T*)x, xs, (const T*)w, (const T*)bias, \
                                                                         (T*)out, os, H, W, C, Ho, Wo, relu)
    if (K == 5 && S == 1) OPA_DW(5, 1);
    else if (K == 5 && S == 2) OPA_DW(5, 2);
    else if (K == 3 && S == 1) OPA_DW(3, 1);
    else if (K == 3 && S == 2) OPA_DW(3, 2);
    else return hipErrorInvalidValue;
#undef OPA_DW
    prof_mark(st, "dwconv_kernel");
    return hipGetLastError();
}

// the widest channel vector the shapes and addresses allow
template <typename T>
static hipError_t launch_dw_t(const void* x, long long xs, const void* w, const void* bias, void* out, long long os,
                              int B, int H, int W, int C, int K, int S, int relu, hipStream_t st) {
    auto ok = [&](int v) {
        const size_t a = v * sizeof(T);
        return C % v == 0 && xs % v == 0 && os % v == 0 && (uintptr_t)x % a == 0 && (uintptr_t)out % a == 0 &&
               (uintptr_t)w % a == 0 && (!bias || (uintptr_t)bias % a == 0);
    };
    if (ok(4)) return launch_dw_v<T, 4>(x, xs, w, bias, out, os, B, H, W, C, K, S, relu, st);
    if (ok(2)) return launch_dw_v<T, 2>(x, xs, w, bias, out, os, B, H, W, C, K, S, relu, st);
    return launch_dw_v<T, 1>(x, xs, w, bias, out, os, B, H, W, C, K, S, relu, st);
}

hipError_t launch_dwconv(const void* x, long long xs, const void* w, const void* bias, void* out, long long os,
                         int B, int H, int W, int C, int K, int S, int dtype, int relu, hipStream_t st) {
    if (dtype == 0) return launch_dw_t<float>(x, xs, w, bias, out, os, B, H, W, C, K, S, relu, st);
    if (dtype == 2) return launch_dw_t<unsigned short>(x, xs, w, bias, out, os, B, H, W, C, K, S, relu, st);
    return hipErrorInvalidValue;
}

template <typename T>
static hipError_t launch_il_t(const void* a, long long as, const void* b, long long bs, void* out, long long rows, int half,
                              hipStream_t st) {
    auto ok = [&](int v) {
        const size_t al = v * sizeof(T);
        return half % v == 0 && as % v == 0 && bs % v == 0 && (uintptr_t)a % al == 0 && (uintptr_t)b % al == 0 &&
               (uintptr_t)out % (2 * al) == 0;
    };
    const int v = ok(4) ? 4 : ok(2) ? 2 : 1;
    const unsigned blocks = (unsigned)((rows * (half / v) + 255) / 256);
    if (v == 4) channel_interleave_kernel<T, 4><<<blocks, 256, 0, st>>>((const T*)a, as, (const T*)b, bs, (T*)out, rows, half);
    else if (v == 2) channel_interleave_kernel<T, 2><<<blocks, 256, 0, st>>>((const T*)a, as, (const T*)b, bs, (T*)out, rows, half);
    else channel_interleave_kernel<T, 1><<<blocks, 256, 0, st>>>((const T*)a, as, (const T*)b, bs, (T*)out, rows, half);
    prof_mark(st, "channel_interleave_kernel");
    return hipGetLastError();
}

hipError_t launch_channel_interleave(const void* a, long long as, const void* b, long long bs, void* out,
                                     long long rows, int half, int dtype, hipStream_t st) {
    if (dtype == 0) return launch_il_t<float>(a, as, b, bs, out, rows, half, st);
    if (dtype == 1 || dtype == 2) return launch_il_t<unsigned short>(a, as, b, bs, out, rows, half, st);
    return hipErrorInvalidValue;
}

}  // namespace opa
