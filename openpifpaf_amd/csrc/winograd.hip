// 3x3 stride-1 convolution of the float32 network as Winograd F(2x2, 3x3), ONE kernel: input transform -> sixteen
// MFMA GEMMs -> output transform, nothing of the transformed domain ever leaves the compute unit.
//     y[n, h, w, :] = sum_{r,s} x[n, h + r - 1, w + s - 1, :] * g[:, :, r, s]^T        (NHWC float32, zero padding 1)
// The ResNet-50 trunk at 641 px / batch 32 spends 39 of its 75 ms in thirteen such convolutions (250 GFLOP each as a direct
// convolution, CK's implicit GEMM at 112 TFLOP/s = 0.71 of the dense float32 MFMA peak: there is no TF32 on gfx950, the
// float32 MFMA runs at the vector rate).  F(2x2, 3x3) needs 16 multiplications per 2x2 outputs instead of 36: 2.25x fewer
// MFMA flops, in float32 arithmetic throughout (the transforms are additions; G g G^T is computed once on the host in
// float64).  Separate transform kernels would write and re-read the 4x larger transformed tensors (15 GB per layer-1
// convolution) -- so one workgroup does all of it for 64 tiles x BN output channels.  Four kernels share that plan: the
// four-wave one described here (variants 0 / 1, the first version), the eight-wave one in two shifts that the Python side uses
// (variant 2, further down: what the four-wave kernel leaves idle and why) and its persistent form (variant 3: measured slower).
//   * 256 threads, four waves, wave w owns the four transform positions (xi = w, nu = 0..3) as 64 x BN accumulators each:
//     v_mfma_f32_32x32x2f32, 4 x 2 x NB accumulator blocks of 16 registers (NB = 2: 256 accumulator registers, one wave per
//     SIMD -- the float32 MFMA issues once per 64 cycles, one wave's other instructions fit into its shadow);
//   * per K-chunk of KC input channels: thread (tile, channel group) loads its tile's 4x4 pixels (KC/4 channels each) from
//     global memory one chunk AHEAD, transforms them in registers (B^T d B: 32 additions per channel) and stores the sixteen
//     positions into the other LDS buffer as V[position][k][tile] (pitch 66 / 68: conflict-free for the thread map of the
//     stores AND the MFMA operand reads); one barrier per chunk;
//   * the filter operand U[position][k][cout] is used by exactly one wave (the owner of the position), so it does not go
//     through LDS at all: the host lays it out so that a lane's values of a chunk are float4-contiguous, and every register
//     group is reloaded for the next chunk right after its last MFMA of this one;
//   * output transform: A^T m A splits into the nu-sum (inside the owning wave, registers) and the xi-sum across the four
//     waves through LDS (the staging buffers, reused), then bias / ReLU if asked and 256-B contiguous stores.
#include "common.hpp"

namespace opa {

typedef __attribute__((ext_vector_type(16))) float wf32x16_t;
typedef __attribute__((ext_vector_type(4))) float wf32x4_t;

template <int KC, int NB>
struct WinoCfg {
    static constexpr int VW = KC / 4;                  // channels per thread and pixel (float2 / float4 loads)
    static constexpr int P = VW == 4 ? 66 : 68;        // floats between two k rows of the LDS operand (64 tiles + pad)
    static constexpr int ROWS = 16 * KC;
    static constexpr int VBUF = ROWS * P;              // floats per stage
    static constexpr int BN = 32 * NB;                 // output channels per workgroup
    static constexpr int KQ = KC / 8;                  // float4 groups of k pairs per chunk
    static constexpr int STAGE_BYTES = 2 * VBUF * 4;
    static constexpr int EPI_BYTES = 4 * 2 * 64 * BN * 4;
    static constexpr int LDS_BYTES = STAGE_BYTES > EPI_BYTES ? STAGE_BYTES : EPI_BYTES;
};

template <int KC, int NB, int WGS>
__global__ __launch_bounds__(256, WGS) void winograd_f23_kernel(
        const float* __restrict__ x, const float* __restrict__ U, float* __restrict__ y, const float* __restrict__ bias,
        int H, int W, int Cin, int Cout, int TH, int TW, int T, int relu, int nb_major) {
    using C = WinoCfg<KC, NB>;
    constexpr int VW = C::VW, P = C::P, BN = C::BN, KQ = C::KQ;
    extern __shared__ __attribute__((aligned(16))) unsigned char wino_smem[];
    float* const lds = reinterpret_cast<float*>(wino_smem);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_nb = Cout / BN;
    // XCD-aware order (gemm_f32.hip): consecutive logical workgroups run on ONE XCD and share its L2
    const unsigned nwg = gridDim.x, xcd = blockIdx.x & 7u, in_xcd = blockIdx.x >> 3;
    const unsigned q8 = nwg >> 3, r8 = nwg & 7u;
    const unsigned logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + in_xcd;
    const unsigned n_tb = nwg / (unsigned)n_nb;
    const int tb = nb_major ? (int)(logical % n_tb) : (int)(logical / (unsigned)n_nb);
    const int nb = nb_major ? (int)(logical / n_tb) : (int)(logical % (unsigned)n_nb);
    const int nchunks = Cin / KC;

    // ---- the staging thread's tile: pixel offsets (clamped into the image) and validity, once
    const int tile_l = tid >> 2, cg = tid & 3;
    int t = tb * 64 + tile_l;
    if (t > T - 1) t = T - 1;
    const int tx = t % TW, ty = (t / TW) % TH, n = t / (TW * TH);
    unsigned rowoff[4], coloff[4];
    bool rv[4], cv[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int r = 2 * ty - 1 + i, c = 2 * tx - 1 + i;
        rv[i] = r >= 0 && r < H;
        cv[i] = c >= 0 && c < W;
        const int rc = r < 0 ? 0 : (r > H - 1 ? H - 1 : r), cc = c < 0 ? 0 : (c > W - 1 ? W - 1 : c);
        rowoff[i] = (unsigned)((n * H + rc) * W) * (unsigned)Cin;
        coloff[i] = (unsigned)cc * (unsigned)Cin + (unsigned)(cg * VW);
    }
    float d[16][VW];
    auto fetch_x = [&](int chunk) {
        const unsigned c0 = (unsigned)(chunk * KC);
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float* p = x + (size_t)(rowoff[i] + coloff[j] + c0);
                if (VW == 4) {
                    const wf32x4_t v = *reinterpret_cast<const wf32x4_t*>(p);
#pragma unroll
                    for (int e = 0; e < VW; e++) d[i * 4 + j][e] = v[e];
                } else {
                    const float2 v = *reinterpret_cast<const float2*>(p);
                    d[i * 4 + j][0] = v.x; d[i * 4 + j][VW - 1] = v.y;
                }
            }
    };
    // B^T d B in registers, then the sixteen positions of this thread's VW channels to stage `buf`
    auto transform_store = [&](int buf) {
        float* const vb = lds + buf * C::VBUF + (cg * VW) * P + tile_l;
#pragma unroll
        for (int e = 0; e < VW; e++) {
            float z[16];
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) z[i * 4 + j] = (rv[i] && cv[j]) ? d[i * 4 + j][e] : 0.0f;
            float s[16];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                s[0 * 4 + j] = z[0 * 4 + j] - z[2 * 4 + j];
                s[1 * 4 + j] = z[1 * 4 + j] + z[2 * 4 + j];
                s[2 * 4 + j] = z[2 * 4 + j] - z[1 * 4 + j];
                s[3 * 4 + j] = z[1 * 4 + j] - z[3 * 4 + j];
            }
#pragma unroll
            for (int i = 0; i < 4; i++) {
                vb[((i * 4 + 0) * KC + e) * P] = s[i * 4 + 0] - s[i * 4 + 2];
                vb[((i * 4 + 1) * KC + e) * P] = s[i * 4 + 1] + s[i * 4 + 2];
                vb[((i * 4 + 2) * KC + e) * P] = s[i * 4 + 2] - s[i * 4 + 1];
                vb[((i * 4 + 3) * KC + e) * P] = s[i * 4 + 1] - s[i * 4 + 3];
            }
        }
    };

    // ---- the filter operand of this wave: U laid out [nb][chunk][position][j][kq][lane] float4 (winograd.py)
    wf32x4_t ub[4][KQ][NB];
    const wf32x4_t* const ubase = reinterpret_cast<const wf32x4_t*>(U) + (size_t)nb * nchunks * (16 * NB * KQ * 64) + lane;
    auto u_ptr = [&](int chunk, int pl, int kq, int j) {
        return ubase + ((size_t)((chunk * 16 + wave * 4 + pl) * NB + j) * KQ + kq) * 64;
    };

    wf32x16_t acc[4][2][NB];
#pragma unroll
    for (int pl = 0; pl < 4; pl++)
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < NB; j++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[pl][i][j][r] = 0.0f;

    fetch_x(0);
#pragma unroll
    for (int pl = 0; pl < 4; pl++)
#pragma unroll
        for (int kq = 0; kq < KQ; kq++)
#pragma unroll
            for (int j = 0; j < NB; j++) ub[pl][kq][j] = *u_ptr(0, pl, kq, j);
    transform_store(0);
    __syncthreads();

    const int a_lane = (lane >> 5) * P + (lane & 31);
    // the sixteen GEMM steps of one chunk for this wave's four positions; every filter register group is reloaded for chunk
    // `nxt` right behind its last use (nxt < 0: the last chunk, nothing to reload)
    auto mfma_phase = [&](int chunk, int nxt) {
        const float* const vb = lds + (chunk & 1) * C::VBUF + a_lane;
#pragma unroll
        for (int pl = 0; pl < 4; pl++) {
            const float* const vp = vb + ((wave * 4 + pl) * KC) * P;
#pragma unroll
            for (int kq = 0; kq < KQ; kq++) {
#pragma unroll
                for (int m = 0; m < 4; m++) {
                    const int kp = kq * 4 + m;
                    const float a0 = vp[(2 * kp) * P], a1 = vp[(2 * kp) * P + 32];
#pragma unroll
                    for (int j = 0; j < NB; j++) {
                        acc[pl][0][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, ub[pl][kq][j][m], acc[pl][0][j], 0, 0, 0);
                        acc[pl][1][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, ub[pl][kq][j][m], acc[pl][1][j], 0, 0, 0);
                    }
                }
                if (nxt >= 0) {
#pragma unroll
                    for (int j = 0; j < NB; j++) ub[pl][kq][j] = *u_ptr(nxt, pl, kq, j);
                }
            }
        }
    };
    // (the last chunk is peeled off: with `if (more)` around the transform the optimiser sinks the loads of fetch_x into that
    // block, i.e. behind the MFMA phase, and the memory latency of every chunk is exposed)
    for (int chunk = 0; chunk + 1 < nchunks; chunk++) {
        fetch_x(chunk + 1);
        __builtin_amdgcn_sched_barrier(0);
        mfma_phase(chunk, chunk + 1);
        __builtin_amdgcn_sched_barrier(0);
        transform_store((chunk + 1) & 1);
        __syncthreads();
    }
    mfma_phase(nchunks - 1, -1);
    __syncthreads();

    // ---- output transform.  nu-sum in registers: P0 = m0 + m1 + m2, P1 = m1 - m2 - m3 (A^T = [1 1 1 0; 0 1 -1 -1])
    float* const sw = lds + wave * (2 * 64 * BN);
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < NB; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int tile = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int col = j * 32 + (lane & 31);
                const float m0 = acc[0][i][j][r], m1 = acc[1][i][j][r], m2 = acc[2][i][j][r], m3 = acc[3][i][j][r];
                sw[tile * BN + col] = (m0 + m1) + m2;
                sw[64 * BN + tile * BN + col] = (m1 - m2) - m3;
            }
    __syncthreads();
    // xi-sum across the waves: Y[0][.] = P(0) + P(1) + P(2), Y[1][.] = P(1) - P(2) - P(3); item = (tile, 4 channels)
    constexpr int V4 = BN / 4;
#pragma unroll
    for (int it = 0; it < 64 * V4 / 256; it++) {
        const int item = it * 256 + tid;
        const int tile = item / V4, c4 = (item % V4) * 4;
        const int tt = tb * 64 + tile;
        if (tt >= T) continue;
        const int ox = tt % TW, oy = (tt / TW) % TH, on = tt / (TW * TH);
        wf32x4_t pq[4][2];
#pragma unroll
        for (int w = 0; w < 4; w++)
#pragma unroll
            for (int jj = 0; jj < 2; jj++)
                pq[w][jj] = *reinterpret_cast<const wf32x4_t*>(lds + (w * 2 + jj) * (64 * BN) + tile * BN + c4);
        wf32x4_t bv = {0.f, 0.f, 0.f, 0.f};
        if (bias) bv = *reinterpret_cast<const wf32x4_t*>(bias + nb * BN + c4);
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int oh = 2 * oy + i;
            if (oh >= H) continue;
#pragma unroll
            for (int jj = 0; jj < 2; jj++) {
                const int ow = 2 * ox + jj;
                if (ow >= W) continue;
                wf32x4_t v = i == 0 ? (pq[0][jj] + pq[1][jj]) + pq[2][jj] : (pq[1][jj] - pq[2][jj]) - pq[3][jj];
                v += bv;
                if (relu) {
#pragma unroll
                    for (int e = 0; e < 4; e++) v[e] = fmaxf(v[e], 0.0f);
                }
                *reinterpret_cast<wf32x4_t*>(y + ((size_t)(on * H + oh) * W + ow) * Cout + nb * BN + c4) = v;
            }
        }
    }
}

// ---- variant 2: the same 64 tiles x 64 channels with EIGHT waves (512 threads), two per SIMD, in two shifts.
// Variant 0's single wave per SIMD does its transform and its LDS round trips in front of the idle MFMA pipe (the compiler
// schedules them behind the multiplications whatever the source says: 20 % + 20 % of a chunk).  Here wave w owns TWO positions
// (xi = w / 2, nu = 2 * (w % 2) + {0, 1}: 128 accumulator registers), waves w and w + 4 share a SIMD, and the two halves of
// the workgroup take the phases of a chunk in opposite order:
//     waves 0-3:  fetch x(c + 1) | multiply chunk c                | transform x(c + 1) -> other stage | barrier
//     waves 4-7:  transform x(c + 1) -> other stage | fetch x(c + 2) | multiply chunk c                 | barrier
// so that one wave of every SIMD has MFMAs to issue while the other one adds, stores and waits for LDS.  Same filter layout
// as variant 0.  Thread (tile = tid / 8, channel pair = tid % 8) stages float2's; LDS pitch 66 is conflict-free for that map.
// Output transform in two halves of 32 tiles (8 waves x 2 x 32 x 64 floats = 128 KB per half).
constexpr int kW8KC = 16, kW8P = 66, kW8VBUF = 16 * kW8KC * kW8P;
constexpr int kW8LdsBytes = 2 * kW8VBUF * 4;            // 135 168 B (>= the 131 072 B of an output half)

template <int DIAG>
__global__ __launch_bounds__(512, 1) void winograd_f23_w8_kernel(
        const float* __restrict__ x, const float* __restrict__ U, float* __restrict__ y, const float* __restrict__ bias,
        int H, int W, int Cin, int Cout, int TH, int TW, int T, int relu, int nb_major) {
    constexpr int KC = kW8KC, P = kW8P, NB = 2, BN = 64, KQ = 2, S = 2 * KQ;
    extern __shared__ __attribute__((aligned(16))) unsigned char wino_smem[];
    float* const lds = reinterpret_cast<float*>(wino_smem);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_nb = Cout / BN;
    const unsigned nwg = gridDim.x, xcd = blockIdx.x & 7u, in_xcd = blockIdx.x >> 3;
    const unsigned q8 = nwg >> 3, r8 = nwg & 7u;
    const unsigned logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + in_xcd;
    const unsigned n_tb = nwg / (unsigned)n_nb;
    const int tb = nb_major ? (int)(logical % n_tb) : (int)(logical / (unsigned)n_nb);
    const int nb = nb_major ? (int)(logical / n_tb) : (int)(logical % (unsigned)n_nb);
    const int nchunks = Cin / KC;

    const int tile_l = tid >> 3, cg = tid & 7;
    int t = tb * 64 + tile_l;
    if (t > T - 1) t = T - 1;
    const int tx = t % TW, ty = (t / TW) % TH, n = t / (TW * TH);
    // byte offsets of the tile's sixteen pixels (clamped into the image), once: a fetch is then sixteen loads of the form
    // uniform base (x + chunk * KC, scalar registers) + 32-bit lane offset -- no address arithmetic in the loop
    unsigned off[16];
    unsigned valid = 0u;                               // bit i * 4 + j: pixel (i, j) lies inside the image
    {
        unsigned rowoff[4], coloff[4];
        bool rv[4], cv[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int r = 2 * ty - 1 + i, c = 2 * tx - 1 + i;
            rv[i] = r >= 0 && r < H;
            cv[i] = c >= 0 && c < W;
            const int rc = r < 0 ? 0 : (r > H - 1 ? H - 1 : r), cc = c < 0 ? 0 : (c > W - 1 ? W - 1 : c);
            rowoff[i] = (unsigned)((n * H + rc) * W) * (unsigned)Cin;
            coloff[i] = (unsigned)cc * (unsigned)Cin + (unsigned)(cg * 2);
        }
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                off[i * 4 + j] = (rowoff[i] + coloff[j]) * 4u;
                if (rv[i] && cv[j]) valid |= 1u << (i * 4 + j);
            }
    }
    const bool wave_inside = __ballot(valid != 0xffffu) == 0ull;   // no lane of this wave has a pixel in the padding
    float d[16][2];
    auto fetch_x = [&](int chunk) {
        const char* const base = reinterpret_cast<const char*>(x + chunk * KC);
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const float2 v = *reinterpret_cast<const float2*>(base + off[k]);
            d[k][0] = v.x; d[k][1] = v.y;
        }
    };
    auto transform_store = [&](int buf) {
        float* const vb = lds + buf * kW8VBUF + (cg * 2) * P + tile_l;
        if (!wave_inside) {                            // zero padding: only the waves that hold a border tile pay for it
#pragma unroll
            for (int k = 0; k < 16; k++)
                if (!((valid >> k) & 1u)) { d[k][0] = 0.0f; d[k][1] = 0.0f; }
        }
#pragma unroll
        for (int e = 0; e < 2; e++) {
            float s[16];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                s[0 * 4 + j] = d[0 * 4 + j][e] - d[2 * 4 + j][e];
                s[1 * 4 + j] = d[1 * 4 + j][e] + d[2 * 4 + j][e];
                s[2 * 4 + j] = d[2 * 4 + j][e] - d[1 * 4 + j][e];
                s[3 * 4 + j] = d[1 * 4 + j][e] - d[3 * 4 + j][e];
            }
#pragma unroll
            for (int i = 0; i < 4; i++) {
                vb[((i * 4 + 0) * KC + e) * P] = s[i * 4 + 0] - s[i * 4 + 2];
                vb[((i * 4 + 1) * KC + e) * P] = s[i * 4 + 1] + s[i * 4 + 2];
                vb[((i * 4 + 2) * KC + e) * P] = s[i * 4 + 2] - s[i * 4 + 1];
                vb[((i * 4 + 3) * KC + e) * P] = s[i * 4 + 1] - s[i * 4 + 3];
            }
        }
    };

    // filter operand: step st = (position 2 * wave + st / KQ, kq = st % KQ); all four steps of a chunk in registers, each
    // reloaded for the next chunk behind its last MFMA
    wf32x4_t ub[S][NB];
    const wf32x4_t* const ubase = reinterpret_cast<const wf32x4_t*>(U) + (size_t)nb * nchunks * (16 * NB * KQ * 64) + lane;
    auto u_ptr = [&](int chunk, int st, int j) {
        return ubase + ((size_t)((chunk * 16 + wave * 2 + st / KQ) * NB + j) * KQ + st % KQ) * 64;
    };
    wf32x16_t acc[2][2][NB];
#pragma unroll
    for (int pl = 0; pl < 2; pl++)
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < NB; j++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[pl][i][j][r] = 0.0f;

    const int a_lane = (lane >> 5) * P + (lane & 31) + (wave * 2 * KC) * P;
    auto mfma_phase = [&](int chunk, int nxt) {
        const float* const vb = lds + (chunk & 1) * kW8VBUF + a_lane;
#pragma unroll
        for (int st = 0; st < S; st++) {
            const int pl = st / KQ;
            const float* const vp = vb + (pl * KC + (st % KQ) * 8) * P;
#pragma unroll
            for (int m = 0; m < 4; m++) {
                const float a0 = DIAG == 5 ? ub[st][0][m] : vp[(2 * m) * P], a1 = DIAG == 5 ? ub[st][1][m] : vp[(2 * m) * P + 32];
#pragma unroll
                for (int j = 0; j < NB; j++) {
                    acc[pl][0][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, ub[st][j][m], acc[pl][0][j], 0, 0, 0);
                    acc[pl][1][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, ub[st][j][m], acc[pl][1][j], 0, 0, 0);
                }
            }
            if (DIAG != 1 && DIAG != 4 && DIAG != 5) {
#pragma unroll
                for (int j = 0; j < NB; j++) ub[st][j] = *u_ptr(nxt, st, j);
            }
        }
    };

    // Prologue in the loops' own order of loads -- pixels first, filter behind them: the compiler's s_waitcnt at a loop head
    // covers the entry edge as well, and with the filter loads issued first the second shift's `wait for the pixels` became
    // vmcnt(0), i.e. every period began by waiting for the filter reloads issued at the end of the period before.
    const int last = nchunks - 1;
    fetch_x(0);
    transform_store(0);
    __builtin_amdgcn_sched_barrier(0);
    if (wave >= 4) fetch_x(nchunks > 1 ? 1 : 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int st = 0; st < S; st++)
#pragma unroll
        for (int j = 0; j < NB; j++) ub[st][j] = *u_ptr(0, st, j);
    __builtin_amdgcn_sched_barrier(0);
    if (wave < 4) {                                    // first shift: multiply, then transform
        __syncthreads();
        for (int chunk = 0; chunk < nchunks; chunk++) {
            const int nxt = chunk < last ? chunk + 1 : last;
            if (DIAG != 2 && DIAG != 4 && DIAG != 5) fetch_x(nxt);
            __builtin_amdgcn_sched_barrier(0);
            mfma_phase(chunk, nxt);
            __builtin_amdgcn_sched_barrier(0);
            if (DIAG != 3 && DIAG != 4 && DIAG != 5) transform_store((chunk + 1) & 1);          // (the last chunk's goes to the stage nobody reads any more)
            __syncthreads();
        }
    } else {                                           // second shift: transform, then multiply
        __syncthreads();
        for (int chunk = 0; chunk < nchunks; chunk++) {
            const int nxt = chunk < last ? chunk + 1 : last;
            if (DIAG != 3 && DIAG != 4 && DIAG != 5) transform_store((chunk + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
            if (DIAG != 2 && DIAG != 4 && DIAG != 5) fetch_x(chunk + 2 < nchunks ? chunk + 2 : last);
            __builtin_amdgcn_sched_barrier(0);
            mfma_phase(chunk, nxt);
            __syncthreads();
        }
    }

    if (DIAG == 6) {                                   // timing experiment: no output transform (one store keeps the MFMAs alive)
        float sum = 0.0f;
#pragma unroll
        for (int pl = 0; pl < 2; pl++)
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < NB; j++)
#pragma unroll
                    for (int r = 0; r < 16; r++) sum += acc[pl][i][j][r];
        if (sum == 12345.678f) y[0] = sum;
        return;
    }
    // ---- output transform, 32 tiles (accumulator row block i) at a time.  This wave's part of the nu-sum:
    //      nu in {0, 1}: P0 = m0 + m1, P1 = m1;   nu in {2, 3}: P0 = m2, P1 = -(m2 + m3)
    const int h = wave & 1;
#pragma unroll
    for (int i = 0; i < 2; i++) {
        float* const sw = lds + wave * (2 * 32 * BN);
#pragma unroll
        for (int j = 0; j < NB; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int tile = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int col = j * 32 + (lane & 31);
                const float ma = acc[0][i][j][r], mb = acc[1][i][j][r];
                sw[tile * BN + col] = h == 0 ? ma + mb : ma;
                sw[32 * BN + tile * BN + col] = h == 0 ? mb : -(ma + mb);
            }
        __syncthreads();
        // item = (tile of this half, 4 channels): 32 x 16 = 512 = one per thread.  xi-sum over the wave pairs:
        //      Y[0][.] = P(0) + P(1) + P(2), Y[1][.] = P(1) - P(2) - P(3), P(xi) = the two waves 2 xi, 2 xi + 1
        {
            const int tile = tid >> 4, c4 = (tid & 15) * 4;
            const int tt = tb * 64 + i * 32 + tile;
            if (tt < T) {
                const int ox = tt % TW, oy = (tt / TW) % TH, on = tt / (TW * TH);
                wf32x4_t pq[4][2];
#pragma unroll
                for (int xi = 0; xi < 4; xi++)
#pragma unroll
                    for (int jj = 0; jj < 2; jj++)
                        pq[xi][jj] = *reinterpret_cast<const wf32x4_t*>(lds + ((2 * xi) * 2 + jj) * (32 * BN) + tile * BN + c4) +
                                     *reinterpret_cast<const wf32x4_t*>(lds + ((2 * xi + 1) * 2 + jj) * (32 * BN) + tile * BN + c4);
                wf32x4_t bv = {0.f, 0.f, 0.f, 0.f};
                if (bias) bv = *reinterpret_cast<const wf32x4_t*>(bias + nb * BN + c4);
#pragma unroll
                for (int a = 0; a < 2; a++) {
                    const int oh = 2 * oy + a;
                    if (oh >= H) continue;
#pragma unroll
                    for (int jj = 0; jj < 2; jj++) {
                        const int ow = 2 * ox + jj;
                        if (ow >= W) continue;
                        wf32x4_t v = a == 0 ? (pq[0][jj] + pq[1][jj]) + pq[2][jj] : (pq[1][jj] - pq[2][jj]) - pq[3][jj];
                        v += bv;
                        if (relu) {
#pragma unroll
                            for (int e = 0; e < 4; e++) v[e] = fmaxf(v[e], 0.0f);
                        }
                        if (DIAG != 7 || v[0] == 12345.678f)       // (7: timing experiment, the output transform without its stores)
                            {
                            wf32x4_t* const dst = reinterpret_cast<wf32x4_t*>(y + ((size_t)(on * H + oh) * W + ow) * Cout + nb * BN + c4);
                            if (DIAG == 8) __builtin_nontemporal_store(v, dst); else *dst = v;   // (8: timing experiment, nontemporal stores)
                        }
                    }
                }
            }
        }
        __syncthreads();
    }
}

// ---- variant 3: variant 2 as PERSISTENT workgroups.  With one workgroup per compute unit nothing overlaps a workgroup's
// prologue (the first pixels come from HBM: 2-3 us) -- a third of a workgroup's time at 64 input channels.  Here a workgroup
// walks tile blocks tb, tb + G, ... of ONE channel block, and during a block's LAST chunk both shifts fetch the first chunk
// of the NEXT block; its B^T d B is computed in registers before the output transform starts (so that the transform's
// stores are not in front of it in the vmcnt queue) and stored to stage 0 behind it.  Every iteration of the chunk loops
// issues the same loads in the same order (a chunk with nothing to fetch re-fetches its own pixels): the compiler's
// s_waitcnt counts are static, a load under a condition makes every wait behind the join conservative.
__global__ __launch_bounds__(512, 1) void winograd_f23_w8p_kernel(
        const float* __restrict__ x, const float* __restrict__ U, float* __restrict__ y, const float* __restrict__ bias,
        int H, int W, int Cin, int Cout, int TH, int TW, int T, int relu, int n_tb, int G) {
    constexpr int KC = kW8KC, P = kW8P, NB = 2, BN = 64, KQ = 2, S = 2 * KQ;
    extern __shared__ __attribute__((aligned(16))) unsigned char wino_smem[];
    float* const lds = reinterpret_cast<float*>(wino_smem);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_nb = Cout / BN;
    const unsigned nwg = gridDim.x, xcd = blockIdx.x & 7u, in_xcd = blockIdx.x >> 3;
    const unsigned q8 = nwg >> 3, r8 = nwg & 7u;
    const unsigned logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + in_xcd;
    const int nb = (int)(logical % (unsigned)n_nb);
    int tb = (int)(logical / (unsigned)n_nb);          // < G <= n_tb
    const int nchunks = Cin / KC, last = nchunks - 1;
    const bool shift_b = wave >= 4;

    const int tile_l = tid >> 3, cg = tid & 7;
    unsigned off[16];
    unsigned valid = 0u;
    auto set_block = [&](int blk) {                    // this thread's tile of block `blk`: pixel offsets and validity
        int t = blk * 64 + tile_l;
        if (t > T - 1) t = T - 1;
        const int tx = t % TW, ty = (t / TW) % TH, n = t / (TW * TH);
        unsigned rowoff[4], coloff[4];
        bool rv[4], cv[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int r = 2 * ty - 1 + i, c = 2 * tx - 1 + i;
            rv[i] = r >= 0 && r < H;
            cv[i] = c >= 0 && c < W;
            const int rc = r < 0 ? 0 : (r > H - 1 ? H - 1 : r), cc = c < 0 ? 0 : (c > W - 1 ? W - 1 : c);
            rowoff[i] = (unsigned)((n * H + rc) * W) * (unsigned)Cin;
            coloff[i] = (unsigned)cc * (unsigned)Cin + (unsigned)(cg * 2);
        }
        valid = 0u;
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                off[i * 4 + j] = (rowoff[i] + coloff[j]) * 4u;
                if (rv[i] && cv[j]) valid |= 1u << (i * 4 + j);
            }
    };
    float d[16][2];
    auto fetch_x = [&](int chunk) {
        const char* const base = reinterpret_cast<const char*>(x + chunk * KC);
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const float2 v = *reinterpret_cast<const float2*>(base + off[k]);
            d[k][0] = v.x; d[k][1] = v.y;
        }
    };
    auto transform_regs = [&]() {                      // d <- B^T d B, in place
        // (zero padding as 32 unconditional selects: a branch around them -- only the waves with a border tile need them --
        // splits the loop body, and the compiler's wait for the pixels at the loop head degrades to vmcnt(0), i.e. to a wait
        // for the filter reloads issued at the end of the period before)
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const bool ok = (valid >> k) & 1u;
            d[k][0] = ok ? d[k][0] : 0.0f; d[k][1] = ok ? d[k][1] : 0.0f;
        }
#pragma unroll
        for (int e = 0; e < 2; e++) {
            float s[16];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                s[0 * 4 + j] = d[0 * 4 + j][e] - d[2 * 4 + j][e];
                s[1 * 4 + j] = d[1 * 4 + j][e] + d[2 * 4 + j][e];
                s[2 * 4 + j] = d[2 * 4 + j][e] - d[1 * 4 + j][e];
                s[3 * 4 + j] = d[1 * 4 + j][e] - d[3 * 4 + j][e];
            }
#pragma unroll
            for (int i = 0; i < 4; i++) {
                d[i * 4 + 0][e] = s[i * 4 + 0] - s[i * 4 + 2];
                d[i * 4 + 1][e] = s[i * 4 + 1] + s[i * 4 + 2];
                d[i * 4 + 2][e] = s[i * 4 + 2] - s[i * 4 + 1];
                d[i * 4 + 3][e] = s[i * 4 + 1] - s[i * 4 + 3];
            }
        }
    };
    auto store_v = [&](int buf) {                      // the sixteen positions of this thread's two channels to stage `buf`
        float* const vb = lds + buf * kW8VBUF + (cg * 2) * P + tile_l;
#pragma unroll
        for (int pos = 0; pos < 16; pos++)
#pragma unroll
            for (int e = 0; e < 2; e++) vb[(pos * KC + e) * P] = d[pos][e];
    };

    wf32x4_t ub[S][NB];
    const wf32x4_t* const ubase = reinterpret_cast<const wf32x4_t*>(U) + (size_t)nb * nchunks * (16 * NB * KQ * 64) + lane;
    auto u_ptr = [&](int chunk, int st, int j) {
        return ubase + ((size_t)((chunk * 16 + wave * 2 + st / KQ) * NB + j) * KQ + st % KQ) * 64;
    };
    wf32x16_t acc[2][2][NB];
    auto zero_acc = [&]() {
#pragma unroll
        for (int pl = 0; pl < 2; pl++)
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < NB; j++)
#pragma unroll
                    for (int r = 0; r < 16; r++) acc[pl][i][j][r] = 0.0f;
    };
    zero_acc();

    const int a_lane = (lane >> 5) * P + (lane & 31) + (wave * 2 * KC) * P;
    auto mfma_phase = [&](int chunk, int nxt) {
        const float* const vb = lds + (chunk & 1) * kW8VBUF + a_lane;
#pragma unroll
        for (int st = 0; st < S; st++) {
            const int pl = st / KQ;
            const float* const vp = vb + (pl * KC + (st % KQ) * 8) * P;
#pragma unroll
            for (int m = 0; m < 4; m++) {
                const float a0 = vp[(2 * m) * P], a1 = vp[(2 * m) * P + 32];
#pragma unroll
                for (int j = 0; j < NB; j++) {
                    acc[pl][0][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, ub[st][j][m], acc[pl][0][j], 0, 0, 0);
                    acc[pl][1][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, ub[st][j][m], acc[pl][1][j], 0, 0, 0);
                }
            }
#pragma unroll
            for (int j = 0; j < NB; j++) ub[st][j] = *u_ptr(nxt, st, j);
        }
    };
    // output transform of block `blk` (as in variant 2: two halves of 32 tiles through all of LDS)
    const int h = wave & 1;
    auto epilogue = [&](int blk) {
#pragma unroll
        for (int i = 0; i < 2; i++) {
            float* const sw = lds + wave * (2 * 32 * BN);
#pragma unroll
            for (int j = 0; j < NB; j++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int tile = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    const int col = j * 32 + (lane & 31);
                    const float ma = acc[0][i][j][r], mb = acc[1][i][j][r];
                    sw[tile * BN + col] = h == 0 ? ma + mb : ma;
                    sw[32 * BN + tile * BN + col] = h == 0 ? mb : -(ma + mb);
                }
            __syncthreads();
            {
                const int tile = tid >> 4, c4 = (tid & 15) * 4;
                const int tt = blk * 64 + i * 32 + tile;
                if (tt < T) {
                    const int ox = tt % TW, oy = (tt / TW) % TH, on = tt / (TW * TH);
                    wf32x4_t pq[4][2];
#pragma unroll
                    for (int xi = 0; xi < 4; xi++)
#pragma unroll
                        for (int jj = 0; jj < 2; jj++)
                            pq[xi][jj] = *reinterpret_cast<const wf32x4_t*>(lds + ((2 * xi) * 2 + jj) * (32 * BN) + tile * BN + c4) +
                                         *reinterpret_cast<const wf32x4_t*>(lds + ((2 * xi + 1) * 2 + jj) * (32 * BN) + tile * BN + c4);
                    wf32x4_t bv = {0.f, 0.f, 0.f, 0.f};
                    if (bias) bv = *reinterpret_cast<const wf32x4_t*>(bias + nb * BN + c4);
#pragma unroll
                    for (int a = 0; a < 2; a++) {
                        const int oh = 2 * oy + a;
                        if (oh >= H) continue;
#pragma unroll
                        for (int jj = 0; jj < 2; jj++) {
                            const int ow = 2 * ox + jj;
                            if (ow >= W) continue;
                            wf32x4_t v = a == 0 ? (pq[0][jj] + pq[1][jj]) + pq[2][jj] : (pq[1][jj] - pq[2][jj]) - pq[3][jj];
                            v += bv;
                            if (relu) {
#pragma unroll
                                for (int e = 0; e < 4; e++) v[e] = fmaxf(v[e], 0.0f);
                            }
                            *reinterpret_cast<wf32x4_t*>(y + ((size_t)(on * H + oh) * W + ow) * Cout + nb * BN + c4) = v;
                        }
                    }
                }
            }
            __syncthreads();
        }
    };

    set_block(tb);
    fetch_x(0);
    int prev = -1;
    while (true) {
        __builtin_amdgcn_sched_barrier(0);
        transform_regs();                              // (waits for the pixels only: they were fetched before the filter reloads)
        __builtin_amdgcn_sched_barrier(0);
        if (prev >= 0) { epilogue(prev); zero_acc(); }
        if (tb >= n_tb) break;
        store_v(0);
        __builtin_amdgcn_sched_barrier(0);
        // The output transform's stores must be out of the vmcnt queue before the chunk loops are entered: with loads AND
        // stores pending the compiler's waits cannot count (the two kinds return out of order) and every wait of the second
        // shift's loop becomes vmcnt(0).  They drained while the accumulators were zeroed and stage 0 was written.
        __builtin_amdgcn_s_waitcnt(0x0F70);            // vmcnt(0)
        __builtin_amdgcn_sched_barrier(0);
        // loads in the chunk loops' order: second shift pixels first, filter behind them
        if (shift_b) fetch_x(nchunks > 1 ? 1 : 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int st = 0; st < S; st++)
#pragma unroll
            for (int j = 0; j < NB; j++) ub[st][j] = *u_ptr(0, st, j);
        __builtin_amdgcn_sched_barrier(0);
        const int nxt_tb = tb + G < n_tb ? tb + G : tb; // (no next block: its own pixels again, never used)
        __syncthreads();
        if (!shift_b) {                                // first shift: fetch, multiply, transform
            for (int chunk = 0; chunk < nchunks; chunk++) {
                const bool is_last = chunk == last;
                if (is_last) set_block(nxt_tb);
                fetch_x(is_last ? 0 : chunk + 1);
                __builtin_amdgcn_sched_barrier(0);
                mfma_phase(chunk, is_last ? 0 : chunk + 1);
                __builtin_amdgcn_sched_barrier(0);
                if (!is_last) { transform_regs(); store_v((chunk + 1) & 1); }
                __syncthreads();
            }
        } else {                                       // second shift: transform, fetch, multiply
            for (int chunk = 0; chunk < nchunks; chunk++) {
                const bool is_last = chunk == last;
                if (!is_last) { transform_regs(); store_v((chunk + 1) & 1); }
                __builtin_amdgcn_sched_barrier(0);
                if (is_last) set_block(nxt_tb);
                fetch_x(is_last ? 0 : (chunk + 2 < nchunks ? chunk + 2 : last));
                __builtin_amdgcn_sched_barrier(0);
                mfma_phase(chunk, is_last ? 0 : chunk + 1);
                __syncthreads();
            }
        }
        prev = tb;
        tb += G;
    }
}

static hipError_t launch_wino_w8p(const float* x, const float* U, float* y, const float* bias, int N, int H, int W, int Cin,
                                  int Cout, int relu, hipStream_t st) {
    const int TH = (H + 1) / 2, TW = (W + 1) / 2, T = N * TH * TW;
    const int n_tb = (T + 63) / 64, n_nb = Cout / 64;
    static int cus = 0;
    if (!cus) {
        int dev = 0; hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return hipGetLastError();
        cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    int G = cus / n_nb;                                // workgroups per channel block: one workgroup per compute unit in all
    if (G < 1) G = 1;
    if (G > n_tb) G = n_tb;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&winograd_f23_w8p_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, kW8LdsBytes);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    winograd_f23_w8p_kernel<<<(unsigned)(G * n_nb), 512, kW8LdsBytes, st>>>(x, U, y, bias, H, W, Cin, Cout, TH, TW, T, relu, n_tb, G);
    return hipGetLastError();
}

template <int DIAG>
static hipError_t launch_wino_w8(const float* x, const float* U, float* y, const float* bias, int N, int H, int W, int Cin,
                                 int Cout, int relu, int nb_major, hipStream_t st) {
    const int TH = (H + 1) / 2, TW = (W + 1) / 2, T = N * TH * TW;
    const long long blocks = (long long)((T + 63) / 64) * (Cout / 64);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&winograd_f23_w8_kernel<DIAG>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, kW8LdsBytes);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    winograd_f23_w8_kernel<DIAG><<<(unsigned)blocks, 512, kW8LdsBytes, st>>>(x, U, y, bias, H, W, Cin, Cout, TH, TW, T, relu, nb_major);
    return hipGetLastError();
}

template <int KC, int NB, int WGS>
static hipError_t launch_wino(const float* x, const float* U, float* y, const float* bias, int N, int H, int W, int Cin,
                              int Cout, int relu, int nb_major, hipStream_t st) {
    using C = WinoCfg<KC, NB>;
    const int TH = (H + 1) / 2, TW = (W + 1) / 2, T = N * TH * TW;
    const long long blocks = (long long)((T + 63) / 64) * (Cout / C::BN);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&winograd_f23_kernel<KC, NB, WGS>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    winograd_f23_kernel<KC, NB, WGS><<<(unsigned)blocks, 256, C::LDS_BYTES, st>>>(x, U, y, bias, H, W, Cin, Cout, TH, TW, T,
                                                                                  relu, nb_major);
    return hipGetLastError();
}

// variant 0: 64 tiles x 64 channels, K-chunks of 16, one workgroup per compute unit (256 accumulator registers per lane);
// variant 1: 64 tiles x 32 channels, K-chunks of 8, two workgroups per compute unit;
// variant 2: variant 0's tile and filter layout, eight waves in two shifts (openpifpaf_amd.winograd.DEFAULT_VARIANT);
// variant 3: variant 2 as persistent workgroups.  The filter layout depends on the variant (0, 2, 3 share one).
hipError_t launch_winograd_f23(const float* x, const float* U, float* y, const float* bias, int N, int H, int W, int Cin,
                               int Cout, int relu, int variant, int nb_major, hipStream_t st) {
    if (variant == 0) return launch_wino<16, 2, 1>(x, U, y, bias, N, H, W, Cin, Cout, relu, nb_major, st);
    if (variant == 2) return launch_wino_w8<0>(x, U, y, bias, N, H, W, Cin, Cout, relu, nb_major, st);
    if (variant == 3) return launch_wino_w8p(x, U, y, bias, N, H, W, Cin, Cout, relu, st);
#ifdef OPA_WINO_DIAG          // timing experiments (wrong results): without the filter reloads / pixel fetches / transforms
    if (variant == 11) return launch_wino_w8<1>(x, U, y, bias, N, H, W, Cin, Cout, relu, nb_major, st);
    if (variant == 12) return launch_wino_w8<2>(x, U, y, bias, N, H, W, Cin, Cout, relu, nb_major, st);
    if (variant == 13) return launch_wino_w8<3>(x, U, y, bias, N, H, W, Cin, Cout, relu, nb_major, st);
    if (variant == 14) return launch_wino_w8<4>(x, U, y, bias, N, H, W, Cin, Cout, relu, nb_major, st);
    if (variant == 15) return launch_wino_w8<5>(x, U, y, bias, N, H, W, Cin, Cout, relu, nb_major, st);
    if (variant == 16) return launch_wino_w8<6>(x, U, y, bias, N, H, W, Cin, Cout, relu, nb_major, st);
    if (variant == 17) return launch_wino_w8<7>(x, U, y, bias, N, H, W, Cin, Cout, relu, nb_major, st);
    if (variant == 18) return launch_wino_w8<8>(x, U, y, bias, N, H, W, Cin, Cout, relu, nb_major, st);
#endif
    return launch_wino<8, 1, 2>(x, U, y, bias, N, H, W, Cin, Cout, relu, nb_major, st);
}

}  // namespace opa
