// Fused convolution epilogue for the field-producing network (inference):
//     x[r, c] = act(x[r, c] + bias[c] (+ residual[r, c]))      in place, NHWC rows.
// PyTorch-ROCm runs a folded conv+BN as conv -> bias add -> ReLU (-> residual add -> ReLU):
// up to four full read+write passes over activations that are 1.7 GB per tensor at
// 641 px / batch 32 -- more time than the convolutions themselves.  This kernel does it in
// one pass: 16-B vector loads/stores (8 bf16/f16 or 4 f32 per lane), grid-stride, fp32 math.
#include "common.hpp"
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>

namespace opa {

struct alignas(16) Vec16 { unsigned int w[4]; };

template <int DT> struct Elem;
template <> struct Elem<0> {   // f32
    static constexpr int kPerVec = 4;
    static __device__ __forceinline__ void unpack(const Vec16& v, float* f) {
        for (int i = 0; i < 4; i++) f[i] = __uint_as_float(v.w[i]);
    }
    static __device__ __forceinline__ void pack(const float* f, Vec16& v) {
        for (int i = 0; i < 4; i++) v.w[i] = __float_as_uint(f[i]);
    }
};
template <> struct Elem<1> {   // f16
    static constexpr int kPerVec = 8;
    static __device__ __forceinline__ void unpack(const Vec16& v, float* f) {
        for (int i = 0; i < 4; i++) {
            const __half2 h = *reinterpret_cast<const __half2*>(&v.w[i]);
            f[2 * i] = __low2float(h); f[2 * i + 1] = __high2float(h);
        }
    }
    static __device__ __forceinline__ void pack(const float* f, Vec16& v) {
        for (int i = 0; i < 4; i++) {
            const __half2 h = __floats2half2_rn(f[2 * i], f[2 * i + 1]);
            v.w[i] = *reinterpret_cast<const unsigned int*>(&h);
        }
    }
};
template <> struct Elem<2> {   // bf16
    static constexpr int kPerVec = 8;
    static __device__ __forceinline__ void unpack(const Vec16& v, float* f) {
        for (int i = 0; i < 4; i++) {
            f[2 * i] = __uint_as_float(v.w[i] << 16);
            f[2 * i + 1] = __uint_as_float(v.w[i] & 0xffff0000u);
        }
    }
    static __device__ __forceinline__ unsigned rne(float x) {      // float -> bf16 bits, round to nearest even
        unsigned u = __float_as_uint(x);
        if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;   // NaN
        u += 0x7fffu + ((u >> 16) & 1u);
        return u >> 16;
    }
    static __device__ __forceinline__ void pack(const float* f, Vec16& v) {
        for (int i = 0; i < 4; i++) v.w[i] = rne(f[2 * i]) | (rne(f[2 * i + 1]) << 16);
    }
};

// Four vectors per thread and step: their loads (x and residual: up to 128 B per thread) are all in flight before the
// first add, and the bias column advances by a 32-bit add instead of a 64-bit modulo per vector.  Measured:
// 4.4-4.9 TB/s with a residual, up to 6.7 TB/s without (tools/gpu/epilogue_probe.py); in the float32 ResNet-50 step
// the layer-1 pass moves 3 x 3.4 GB (32 x 256 x 321 x 321 floats per tensor) in 2.27 ms = 4.5 TB/s.
constexpr int kEpiUnroll = 4;

template <int DT, bool RES, bool RELU>
__global__ __launch_bounds__(256) void bias_act_kernel(Vec16* __restrict__ x, const Vec16* __restrict__ bias,
                                                       const Vec16* __restrict__ res, long long n_vec, int vec_per_row) {
    constexpr int N = Elem<DT>::kPerVec;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const int col_step = (int)(stride % vec_per_row);
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    int col = (int)(i % vec_per_row);
    for (; i < n_vec; i += kEpiUnroll * stride) {
        Vec16 xv[kEpiUnroll], rv[kEpiUnroll];
        int cols[kEpiUnroll];
#pragma unroll
        for (int u = 0; u < kEpiUnroll; u++) {
            const long long j = i + u * stride;
            cols[u] = col;
            col += col_step; if (col >= vec_per_row) col -= vec_per_row;
            if (j < n_vec) { xv[u] = x[j]; if (RES) rv[u] = res[j]; }
        }
#pragma unroll
        for (int u = 0; u < kEpiUnroll; u++) {
            const long long j = i + u * stride;
            if (j >= n_vec) break;
            const Vec16 bv = bias[cols[u]];
            float a[N], b[N];
            Elem<DT>::unpack(xv[u], a);
            Elem<DT>::unpack(bv, b);
            if (RES) {
                float r[N];
                Elem<DT>::unpack(rv[u], r);
#pragma unroll
                for (int k = 0; k < N; k++) a[k] = a[k] + b[k] + r[k];
            } else {
#pragma unroll
                for (int k = 0; k < N; k++) a[k] = a[k] + b[k];
            }
            if (RELU) {
#pragma unroll
                for (int k = 0; k < N; k++) a[k] = fmaxf(a[k], 0.0f);
            }
            Vec16 o;
            Elem<DT>::pack(a, o);
            x[j] = o;
        }
    }
}

template <int DT>
static hipError_t launch_dt(void* x, const void* bias, const void* res, long long rows, int channels, int relu,
                            hipStream_t st) {
    constexpr int N = Elem<DT>::kPerVec;
    const int vec_per_row = channels / N;
    const long long n_vec = rows * vec_per_row;
    long long blocks = (n_vec + 255) / 256;
    blocks = (blocks + kEpiUnroll - 1) / kEpiUnroll;
    if (blocks > 256 * 16) blocks = 256 * 16;          // 256 CUs x 16 blocks, grid-stride the rest
    if (blocks < 1) blocks = 1;
    Vec16* xv = (Vec16*)x; const Vec16* bv = (const Vec16*)bias; const Vec16* rv = (const Vec16*)res;
    if (res) {
        if (relu) bias_act_kernel<DT, true, true><<<(unsigned)blocks, 256, 0, st>>>(xv, bv, rv, n_vec, vec_per_row);
        else bias_act_kernel<DT, true, false><<<(unsigned)blocks, 256, 0, st>>>(xv, bv, rv, n_vec, vec_per_row);
    } else {
        if (relu) bias_act_kernel<DT, false, true><<<(unsigned)blocks, 256, 0, st>>>(xv, bv, rv, n_vec, vec_per_row);
        else bias_act_kernel<DT, false, false><<<(unsigned)blocks, 256, 0, st>>>(xv, bv, rv, n_vec, vec_per_row);
    }
    return hipGetLastError();
}

hipError_t launch_bias_act(void* x, const void* bias, const void* res, long long rows, int channels, int dtype,
                           int relu, hipStream_t st) {
    switch (dtype) {
        case 0: return launch_dt<0>(x, bias, res, rows, channels, relu, st);
        case 1: return launch_dt<1>(x, bias, res, rows, channels, relu, st);
        case 2: return launch_dt<2>(x, bias, res, rows, channels, relu, st);
        default: return hipErrorInvalidValue;
    }
}

// ---- zero fill (see common.hpp) ------------------------------------------------------------------
__global__ __launch_bounds__(256) void zero_kernel(unsigned* __restrict__ dst, size_t n_words) {
    const size_t n16 = n_words / 4;
    uint4* d16 = reinterpret_cast<uint4*>(dst);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256)
        d16[i] = make_uint4(0u, 0u, 0u, 0u);
    if (blockIdx.x == 0 && threadIdx.x < (n_words & 3)) dst[n16 * 4 + threadIdx.x] = 0u;
}

hipError_t launch_zero(void* dst, size_t bytes, hipStream_t st) {
    if (bytes == 0) return hipSuccess;
    if ((bytes & 3) || ((uintptr_t)dst & 15)) return hipMemsetAsync(dst, 0, bytes, st);
    const size_t n_words = bytes / 4;
    size_t blocks = (n_words / 4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks == 0) blocks = 1;
    zero_kernel<<<(unsigned)blocks, 256, 0, st>>>((unsigned*)dst, n_words);
    return hipGetLastError();
}

}  // namespace opa
