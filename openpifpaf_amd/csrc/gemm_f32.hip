// The float32 twin of gemm_epilogue.hip: 1x1 convolution as an MFMA GEMM with the whole epilogue fused,
//     out[M,N] = act( A[M,K] * W[N,K]^T + bias[N] (+ residual[M,N]) )        (f32 in/out, f32 math)
// for the network at the reference's precision.  At 641 px / batch 32 the expanding 1x1 convolution of a
// ResNet bottleneck writes a 3.4 GB (layer 1) tensor that the separate bias + residual + ReLU pass then reads,
// adds the 3.4 GB residual to and writes again: 2.1 ms beside a 1.2 ms convolution.  Fused, the product never
// leaves the registers before the epilogue: one read of the residual, one write of the output.
//
// v_mfma_f32_32x32x2f32: lane l supplies A[row l&31][k = l>>5] and B[k = l>>5][col l&31].  Operands are staged
// K-major in LDS like in the bf16 kernel; a lane reads FOUR consecutive k of its row with one ds_read_b128 and
// feeds them to four MFMAs, so MFMA m of a group multiplies the k pair (8j + m, 8j + 4 + m) -- every k once, the
// order of the sum over k being as arbitrary as in any GEMM.
// Tile 128 x BN (BN = 128 | 64) per 256-thread workgroup, BK = 32 floats (the bf16 kernel's 128-B rows and its
// 144-B LDS pitch); 4 waves as 2(M) x 2(N); XCD-aware tile order; epilogue through a wave-private LDS patch.
// Measured (tools/gpu/gemm_f32_probe.py, batch 32 at 641 px): layer-1 expand 64->256 + residual 1.70 ms against 3.3 ms
// for MIOpen's convolution + the epilogue pass; 105-119 TFLOP/s on the compute-bound shapes (dense f32 MFMA peak 157).
#include "common.hpp"

namespace opa {

typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;      // (HIP's float4 struct keeps register arrays on the stack)

constexpr int kF32BM = 128, kF32BK = 32, kF32Pitch = kF32BK + 4;      // LDS row pitch in floats
// One LDS stage and three workgroups per CU (three waves per SIMD take turns on the MFMA pipe while the others load,
// store and wait at barriers) beat two stages with two workgroups on every bench shape: 105-119 against 95-106
// TFLOP/s, 4.5 against 3.9 TB/s on the bandwidth-bound layer-1 shape.  (-DOPA_F32_STAGES=2 -DOPA_F32_WGS=2: the other.)
#ifndef OPA_F32_STAGES
#define OPA_F32_STAGES 1
#endif
#ifndef OPA_F32_WGS
#define OPA_F32_WGS 3
#endif
#ifndef OPA_GEMM_DIAG              // timing experiments (wrong results): 1 no global loads in the K loop, 2 no LDS stores either,
#define OPA_GEMM_DIAG 0            // 3 = 2 without the epilogue's residual load and output store
#endif
constexpr int kF32Stages = OPA_F32_STAGES;       // LDS stages (2: the next K-step is stored while this one multiplies)

template <int BN, bool RES, bool RELU, bool PRO>
__global__ __launch_bounds__(256, BN == 128 ? OPA_F32_WGS : 3) void gemm_f32_bias_act_kernel(
        const float* __restrict__ A, const float* __restrict__ W, const float* __restrict__ bias,
        const float* __restrict__ res, float* __restrict__ out, int M, int N, int K, const float* __restrict__ a_bias) {
    constexpr int WN = BN / 2;                 // wave tile width
    constexpr int NT = WN / 32;                // 32-wide MFMA blocks per wave in N (2 or 1)
    constexpr int LDS_A = kF32BM * kF32Pitch, LDS_B = BN * kF32Pitch;
    constexpr int STAGE_BYTES = kF32Stages * (LDS_A + LDS_B) * 4;
    constexpr int EPI_BYTES = 4 * 32 * WN * 4; // per wave a 32 x WN f32 patch
    __shared__ __attribute__((aligned(16))) unsigned char smem[STAGE_BYTES > EPI_BYTES ? STAGE_BYTES : EPI_BYTES];
    float* stage0 = reinterpret_cast<float*>(smem);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int n_tiles = N / BN;
    // XCD-aware tile order (see gemm_epilogue.hip): the N-tiles sharing one A row-block run on ONE L2
    const unsigned nwg = gridDim.x, xcd = blockIdx.x & 7u, in_xcd = blockIdx.x >> 3;
    const unsigned q = nwg >> 3, r8 = nwg & 7u;
    const unsigned logical = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + in_xcd;
    const int m0 = (int)(logical / n_tiles) * kF32BM;
    const int n0 = (int)(logical % n_tiles) * BN;

    f32x16_t acc[2][NT];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < NT; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.0f;

    // staging map: a 32-float row is 8 x 16 B; 256 threads cover 32 rows per pass
    const int s_row = tid >> 3, s_col = (tid & 7) * 4;
    f32x4_t ra[kF32BM / 32], rb[BN / 32];
    // Row pointers once, outside the K loop.  Rows past M read row M - 1 (valid memory; the epilogue never stores them): no
    // zero fill, no branch around a load -- with both in the loop the compiler put `s_waitcnt vmcnt(0)` BETWEEN the A and the
    // B loads of a K-step (destination registers doubled as address temporaries), i.e. the A operand's memory latency sat in
    // front of every K-step's multiplications instead of behind the previous step's (round 5: found in the ISA).
    const float* pa[kF32BM / 32]; const float* pb[BN / 32];
#pragma unroll
    for (int p = 0; p < kF32BM / 32; p++) {
        int m = m0 + p * 32 + s_row;
        if (m > M - 1) m = M - 1;
        pa[p] = A + (size_t)m * K + s_col;
    }
#pragma unroll
    for (int p = 0; p < BN / 32; p++) pb[p] = W + (size_t)(n0 + p * 32 + s_row) * K + s_col;
    auto fetch = [&](int k0) {                 // global -> registers for K-step k0 (with the operand prologue): all loads back to back
#pragma unroll
        for (int p = 0; p < kF32BM / 32; p++) ra[p] = *reinterpret_cast<const f32x4_t*>(pa[p] + k0);
#pragma unroll
        for (int p = 0; p < BN / 32; p++) rb[p] = *reinterpret_cast<const f32x4_t*>(pb[p] + k0);
        if (PRO) {                             // the preceding convolution's bias + ReLU, applied to the raw operand
            const f32x4_t ab = *reinterpret_cast<const f32x4_t*>(a_bias + k0 + s_col);
#pragma unroll
            for (int p = 0; p < kF32BM / 32; p++)
#pragma unroll
                for (int e = 0; e < 4; e++) ra[p][e] = fmaxf(ra[p][e] + ab[e], 0.0f);
        }
    };
    auto store = [&](int buf) {                // registers -> LDS stage `buf`
        float* sA = stage0 + buf * (LDS_A + LDS_B);
        float* sB = sA + LDS_A;
#pragma unroll
        for (int p = 0; p < kF32BM / 32; p++)
            *reinterpret_cast<f32x4_t*>(sA + (p * 32 + s_row) * kF32Pitch + s_col) = ra[p];
#pragma unroll
        for (int p = 0; p < BN / 32; p++)
            *reinterpret_cast<f32x4_t*>(sB + (p * 32 + s_row) * kF32Pitch + s_col) = rb[p];
    };
    fetch(0);
    store(0);
    __syncthreads();
    int buf = 0;
    for (int k0 = 0; k0 < K; k0 += kF32BK, buf ^= (kF32Stages - 1)) {
        const bool more = k0 + kF32BK < K;
        if (more && OPA_GEMM_DIAG == 0) fetch(k0 + kF32BK);          // the next K-step's operands travel while this one multiplies
        const float* sA = stage0 + buf * (LDS_A + LDS_B);
        const float* sB = sA + LDS_A;
#pragma unroll
        for (int kk = 0; kk < kF32BK; kk += 8) {
            f32x4_t fa[2], fb[NT];
            const int kof = kk + (lane >> 5) * 4;
#pragma unroll
            for (int i = 0; i < 2; i++)
                fa[i] = *reinterpret_cast<const f32x4_t*>(sA + (wm * 64 + i * 32 + (lane & 31)) * kF32Pitch + kof);
#pragma unroll
            for (int j = 0; j < NT; j++)
                fb[j] = *reinterpret_cast<const f32x4_t*>(sB + (wn * WN + j * 32 + (lane & 31)) * kF32Pitch + kof);
            // consecutive MFMAs go to different accumulators (a dependent MFMA waits for all 16 passes of its predecessor)
#pragma unroll
            for (int m = 0; m < 4; m++)
#pragma unroll
                for (int i = 0; i < 2; i++)
#pragma unroll
                    for (int j = 0; j < NT; j++)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][m], fb[j][m], acc[i][j], 0, 0, 0);
        }
        if (kF32Stages == 1) __syncthreads(); // one stage: every wave is done reading it
        if (more && OPA_GEMM_DIAG < 2) store(buf ^ (kF32Stages - 1)); // (two stages: the other one, whose readers passed the previous step's barrier)
        __syncthreads();
    }
    // (the loop's last barrier: staging LDS is free, reuse it for the epilogue)

    // epilogue, one 32-row block of the wave tile at a time through a wave-private f32 patch: the residual load and
    // the output store are row-contiguous 16-B vectors
    constexpr int VEC_PER_ROW = WN / 4;
    constexpr int VPL = 32 * VEC_PER_ROW / 64;
    float* patch = reinterpret_cast<float*>(smem) + wave * (32 * WN);
#pragma unroll
    for (int i = 0; i < 2; i++) {
        constexpr int VPRE = VPL > 4 ? 4 : VPL; // residual vectors fetched ahead (16 registers); the rest in the loop
        f32x4_t rv[VPRE];
        if (RES) {                             // they travel while the patch is written
#pragma unroll
            for (int t = 0; t < VPRE; t++) {
                const int v = t * 64 + lane;
                const int row = v / VEC_PER_ROW, c4 = (v % VEC_PER_ROW) * 4;
                const int m = m0 + wm * 64 + i * 32 + row;
                rv[t] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
                if (m < M) rv[t] = *reinterpret_cast<const f32x4_t*>(res + (size_t)m * N + n0 + wn * WN + c4);
            }
        }
#pragma unroll
        for (int j = 0; j < NT; j++) {
            const int col = j * 32 + (lane & 31);
            const float b = bias[n0 + wn * WN + col];
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
            const int row = v / VEC_PER_ROW, c4 = (v % VEC_PER_ROW) * 4;
            const int m = m0 + wm * 64 + i * 32 + row;
            if (m < M) {
                f32x4_t f = *reinterpret_cast<const f32x4_t*>(patch + row * WN + c4);
                if (RES) {
                    if (t < VPRE) f += rv[t < VPRE ? t : 0];
                    else f += *reinterpret_cast<const f32x4_t*>(res + (size_t)m * N + n0 + wn * WN + c4);
                }
                if (RELU) {
#pragma unroll
                    for (int e = 0; e < 4; e++) f[e] = fmaxf(f[e], 0.0f);
                }
                if (OPA_GEMM_DIAG < 3 || f[0] == 12345.678f) *reinterpret_cast<f32x4_t*>(out + (size_t)m * N + n0 + wn * WN + c4) = f;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        asm volatile("" ::: "memory");         // (keeps the second block's residual loads -- 32 registers -- behind this block)
    }
}

template <int BN, bool PRO>
static hipError_t launch_f32_bn(const float* a, const float* w, const float* b, const float* r, float* o,
                                int M, int N, int K, int relu, const float* ab, hipStream_t st) {
    const long long blocks = (long long)((M + kF32BM - 1) / kF32BM) * (N / BN);
    if (r) {
        if (relu) gemm_f32_bias_act_kernel<BN, true, true, PRO><<<(unsigned)blocks, 256, 0, st>>>(a, w, b, r, o, M, N, K, ab);
        else gemm_f32_bias_act_kernel<BN, true, false, PRO><<<(unsigned)blocks, 256, 0, st>>>(a, w, b, r, o, M, N, K, ab);
    } else {
        if (relu) gemm_f32_bias_act_kernel<BN, false, true, PRO><<<(unsigned)blocks, 256, 0, st>>>(a, w, b, r, o, M, N, K, ab);
        else gemm_f32_bias_act_kernel<BN, false, false, PRO><<<(unsigned)blocks, 256, 0, st>>>(a, w, b, r, o, M, N, K, ab);
    }
    return hipGetLastError();
}

hipError_t launch_gemm_f32_bias_act(const float* A, const float* W, const float* bias, const float* res, float* out,
                                    int M, int N, int K, int relu, hipStream_t st, const float* a_bias) {
    if (a_bias) {
        if (N % 128 == 0) return launch_f32_bn<128, true>(A, W, bias, res, out, M, N, K, relu, a_bias, st);
        return launch_f32_bn<64, true>(A, W, bias, res, out, M, N, K, relu, a_bias, st);
    }
    if (N % 128 == 0) return launch_f32_bn<128, false>(A, W, bias, res, out, M, N, K, relu, nullptr, st);
    return launch_f32_bn<64, false>(A, W, bias, res, out, M, N, K, relu, nullptr, st);
}

}  // namespace opa
