// 1x1 convolution of the field-producing network as an MFMA GEMM with the whole epilogue fused:
//     out[M,N] = act( A[M,K] * W[N,K]^T + bias[N] (+ residual[M,N]) )        (bf16 in/out, f32 math)
// A is an NHWC activation viewed as [M = B*H*W, K = C_in] (row-major, K contiguous), W the conv
// weight [C_out, C_in] -- both operands are "K-major", exactly the fragment order of
// v_mfma_f32_32x32x16_bf16 (lane l holds 8 consecutive k of row l&31, k-offset 8*(l>>5)).
//
// Why: at 641 px / batch 32 the 1x1 convs of ResNet blocks 2-3 are bandwidth bound (CK's kernels
// already run them at ~5 TB/s) and the separate bias/residual/ReLU pass over the 1.7 GB outputs
// costs more than the convolution.  Fusing it removes one read and one write of the output.
//
// Tile: 128 x BN (BN = 128 or 64) per 256-thread workgroup, BK = 64; 4 waves as 2(M) x 2(N), each
// wave a 64 x BN/2 sub-tile of 32x32 MFMA blocks; operands staged global -> registers -> LDS
// (row pitch 72 bf16 = 144 B keeps ds_read_b128 fragment reads at <= 2-way bank conflicts);
// several workgroups per CU overlap each other's loads and MFMAs.  Epilogue: accumulators
// (+bias) go through a wave-private f32 LDS patch so that the residual load and the output store are
// row-contiguous 16-B vectors.
#include "common.hpp"

namespace opa {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

constexpr int kGemmBM = 128, kGemmBK = 64, kGemmPitch = kGemmBK + 8;   // LDS row pitch in bf16

__device__ __forceinline__ float bf16_lo(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16_hi(unsigned w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ unsigned bf16_rne(float x) {
    unsigned u = __float_as_uint(x);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
}

// PRO: the A operand is the RAW output of the preceding convolution; its bias + ReLU epilogue
// (a_bias[k], per input channel) is applied while the tile is staged into LDS -- same f32 add, max and
// round-to-nearest-even as the separate bias_act pass, which is thereby saved (one read and one write
// of the 3x3 convolution's output per bottleneck).
template <int BN, bool RES, bool RELU, bool PRO>
__global__ __launch_bounds__(256, 3) void gemm_bias_act_kernel(
        const unsigned short* __restrict__ A, const unsigned short* __restrict__ W,
        const unsigned short* __restrict__ bias, const unsigned short* __restrict__ res,
        unsigned short* __restrict__ out, int M, int N, int K, const unsigned short* __restrict__ a_bias) {
    constexpr int WN = BN / 2;                 // wave tile width
    constexpr int NT = WN / 32;                // 32-wide MFMA blocks per wave in N (2 or 1)
    constexpr int LDS_A = kGemmBM * kGemmPitch, LDS_B = BN * kGemmPitch;
    constexpr int STAGE_BYTES = (LDS_A + LDS_B) * 2;
    constexpr int EPI_BYTES = 4 * 32 * WN * 4; // per wave a 32 x WN f32 patch
    __shared__ __attribute__((aligned(16))) unsigned char smem[STAGE_BYTES > EPI_BYTES ? STAGE_BYTES : EPI_BYTES];
    unsigned short* sA = reinterpret_cast<unsigned short*>(smem);
    unsigned short* sB = sA + LDS_A;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int n_tiles = N / BN;
    // XCD-aware tile order: workgroup b runs on XCD b % 8 (each XCD has its own L2).  Give every XCD
    // a contiguous range of logical tile ids so that the N-tiles sharing one A row-block are
    // neighbours in time on ONE L2 instead of being fetched by eight (bijective for any grid size).
    const unsigned nwg = gridDim.x, xcd = blockIdx.x & 7u, in_xcd = blockIdx.x >> 3;
    const unsigned q = nwg >> 3, r8 = nwg & 7u;
    const unsigned logical = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + in_xcd;
    const int m0 = (int)(logical / n_tiles) * kGemmBM;
    const int n0 = (int)(logical % n_tiles) * BN;

    f32x16_t acc[2][NT];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < NT; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.0f;

    // The residual tile does not depend on the product: fetch this lane's share (the vectors it
    // will combine in the epilogue) before anything else so that its HBM latency hides behind the
    // operand loads and the MFMAs.
    constexpr int VEC_PER_ROW = WN / 8;        // 16-B output vectors per patch row
    constexpr int VPL = 32 * VEC_PER_ROW / 64; // epilogue vectors per lane and 32-row block
    u32x4_t rpre[2][VPL];
    auto fetch_residual = [&](int i) {
#pragma unroll
        for (int t = 0; t < VPL; t++) {
            const int v = t * 64 + lane;
            const int row = v / VEC_PER_ROW, c8 = (v % VEC_PER_ROW) * 8;
            const int m = m0 + wm * 64 + i * 32 + row;
            rpre[i][t] = (u32x4_t){0u, 0u, 0u, 0u};
            if (m < M) rpre[i][t] = *reinterpret_cast<const u32x4_t*>(res + (size_t)m * N + n0 + wn * WN + c8);
        }
    };
    if (RES) fetch_residual(0);                // the second half is fetched when the operand registers are free

    // staging map: a 64-wide bf16 row is 8 x 16 B; 256 threads cover 32 rows per pass
    const int s_row = tid >> 3, s_col = (tid & 7) * 8;
    u32x4_t ra[kGemmBM / 32], rb[BN / 32];
    auto fetch = [&](int k0) {                 // global -> registers for K-step k0 (with the operand prologue)
#pragma unroll
        for (int p = 0; p < kGemmBM / 32; p++) {
            const int m = m0 + p * 32 + s_row;
            ra[p] = (u32x4_t){0u, 0u, 0u, 0u};
            if (m < M) ra[p] = *reinterpret_cast<const u32x4_t*>(A + (size_t)m * K + k0 + s_col);
        }
#pragma unroll
        for (int p = 0; p < BN / 32; p++)
            rb[p] = *reinterpret_cast<const u32x4_t*>(W + (size_t)(n0 + p * 32 + s_row) * K + k0 + s_col);
        if (PRO) {
            const u32x4_t ab = *reinterpret_cast<const u32x4_t*>(a_bias + k0 + s_col);
#pragma unroll
            for (int p = 0; p < kGemmBM / 32; p++)
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const float lo = fmaxf(bf16_lo(ra[p][q]) + bf16_lo(ab[q]), 0.0f);
                    const float hi = fmaxf(bf16_hi(ra[p][q]) + bf16_hi(ab[q]), 0.0f);
                    ra[p][q] = bf16_rne(lo) | (bf16_rne(hi) << 16);
                }
        }
    };
    fetch(0);
    for (int k0 = 0; k0 < K; k0 += kGemmBK) {
        __syncthreads();                       // previous step's fragment reads are done
#pragma unroll
        for (int p = 0; p < kGemmBM / 32; p++)
            *reinterpret_cast<u32x4_t*>(sA + (p * 32 + s_row) * kGemmPitch + s_col) = ra[p];
#pragma unroll
        for (int p = 0; p < BN / 32; p++)
            *reinterpret_cast<u32x4_t*>(sB + (p * 32 + s_row) * kGemmPitch + s_col) = rb[p];
        __syncthreads();
        if (k0 + kGemmBK < K) fetch(k0 + kGemmBK);   // the next K-step's operands travel while this one multiplies
#pragma unroll
        for (int kk = 0; kk < kGemmBK; kk += 16) {
            bf16x8_t fa[2], fb[NT];
            const int kof = kk + (lane >> 5) * 8;
#pragma unroll
            for (int i = 0; i < 2; i++)
                fa[i] = *reinterpret_cast<const bf16x8_t*>(sA + (wm * 64 + i * 32 + (lane & 31)) * kGemmPitch + kof);
#pragma unroll
            for (int j = 0; j < NT; j++)
                fb[j] = *reinterpret_cast<const bf16x8_t*>(sB + (wn * WN + j * 32 + (lane & 31)) * kGemmPitch + kof);
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < NT; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();                           // staging LDS is free: reuse it for the epilogue
    if (RES) fetch_residual(1);

    // epilogue, one 32-row block of the wave tile at a time through a wave-private f32 patch
    float* patch = reinterpret_cast<float*>(smem) + wave * (32 * WN);
#pragma unroll
    for (int i = 0; i < 2; i++) {
#pragma unroll
        for (int j = 0; j < NT; j++) {
            const int col = j * 32 + (lane & 31);
            const float b = bf16_lo((unsigned)bias[n0 + wn * WN + col]);
#pragma unroll
            for (int r = 0; r < 16; r++) {     // C layout: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
                const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                patch[row * WN + col] = acc[i][j][r] + b;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int t = 0; t < VPL; t++) {
            const int v = t * 64 + lane;
            const int row = v / VEC_PER_ROW, c8 = (v % VEC_PER_ROW) * 8;
            const int m = m0 + wm * 64 + i * 32 + row;
            if (m < M) {
                const size_t g = (size_t)m * N + n0 + wn * WN + c8;
                const float4 lo = *reinterpret_cast<const float4*>(patch + row * WN + c8);
                const float4 hi = *reinterpret_cast<const float4*>(patch + row * WN + c8 + 4);
                float f[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                if (RES) {
                    const u32x4_t rv = rpre[i][t];
#pragma unroll
                    for (int q = 0; q < 4; q++) { f[2 * q] += bf16_lo(rv[q]); f[2 * q + 1] += bf16_hi(rv[q]); }
                }
                if (RELU) {
#pragma unroll
                    for (int q = 0; q < 8; q++) f[q] = fmaxf(f[q], 0.0f);
                }
                u32x4_t ov;
#pragma unroll
                for (int q = 0; q < 4; q++) ov[q] = bf16_rne(f[2 * q]) | (bf16_rne(f[2 * q + 1]) << 16);
                *reinterpret_cast<u32x4_t*>(out + g) = ov;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

template <int BN, bool PRO>
static hipError_t launch_bn(const void* A, const void* W, const void* bias, const void* res, void* out,
                            int M, int N, int K, int relu, const void* a_bias, hipStream_t st) {
    const long long blocks = (long long)((M + kGemmBM - 1) / kGemmBM) * (N / BN);
    const unsigned short *a = (const unsigned short*)A, *w = (const unsigned short*)W,
                         *b = (const unsigned short*)bias, *r = (const unsigned short*)res,
                         *ab = (const unsigned short*)a_bias;
    unsigned short* o = (unsigned short*)out;
    if (res) {
        if (relu) gemm_bias_act_kernel<BN, true, true, PRO><<<(unsigned)blocks, 256, 0, st>>>(a, w, b, r, o, M, N, K, ab);
        else gemm_bias_act_kernel<BN, true, false, PRO><<<(unsigned)blocks, 256, 0, st>>>(a, w, b, r, o, M, N, K, ab);
    } else {
        if (relu) gemm_bias_act_kernel<BN, false, true, PRO><<<(unsigned)blocks, 256, 0, st>>>(a, w, b, r, o, M, N, K, ab);
        else gemm_bias_act_kernel<BN, false, false, PRO><<<(unsigned)blocks, 256, 0, st>>>(a, w, b, r, o, M, N, K, ab);
    }
    return hipGetLastError();
}

hipError_t launch_gemm_bias_act(const void* A, const void* W, const void* bias, const void* res, void* out,
                                int M, int N, int K, int relu, hipStream_t st, const void* a_bias) {
    if (a_bias) {
        if (N % 128 == 0) return launch_bn<128, true>(A, W, bias, res, out, M, N, K, relu, a_bias, st);
        return launch_bn<64, true>(A, W, bias, res, out, M, N, K, relu, a_bias, st);
    }
    if (N % 128 == 0) return launch_bn<128, false>(A, W, bias, res, out, M, N, K, relu, nullptr, st);
    return launch_bn<64, false>(A, W, bias, res, out, M, N, K, relu, nullptr, st);
}

}  // namespace opa
