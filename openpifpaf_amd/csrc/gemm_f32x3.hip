// The float32 1x1 convolution on the bf16 MFMA pipe, without giving up a bit of either operand:
//     out[M,N] = act( A[M,K] * W[N,K]^T + bias[N] (+ residual[M,N]) )        (f32 in, f32 out, f32 accumulators)
// gfx950 has no TF32 and its float32 MFMA (v_mfma_f32_32x32x2f32) runs at 1/16 of the bf16 rate (157 against 2 500 TFLOP/s
// dense), so the compute-bound 1x1 convolutions of ResNet layers 3-4 sit at 115-118 TFLOP/s however the kernel around them
// is built (gemm_f32.hip).  A float32 number is, exactly, the sum of three bfloat16 numbers: its 24-bit
// significand cut into 8 + 8 + 8 bits (a1 = a with the low 16 bits cleared, r = a - a1, a2 = r with the low 16 bits cleared,
// a3 = r - a2: every step exact, every piece representable -- bf16 has float32's exponent range).  So
//     a * b = sum over i, j of a_i * b_j                                       (nine products, each EXACT in float32:
// 8 x 8 significand bits), and v_mfma_f32_32x32x16_bf16 forms and accumulates them in float32 like the float32 MFMA
// accumulates its own exact products.  TERMS = 9: all of them -- the GEMM's only rounding is the accumulation's, as in
// any float32 GEMM.  TERMS = 6: without a2*b3, a3*b2, a3*b3 (below 2^-23 of the product each).  The leading products
// a1*b1 and the corrections go to accumulators of their own (the corrections are 2^-8 and less of the sum: added
// among themselves first they are not rounded away against it), joined in the epilogue.
// Nine bf16 MFMAs of 8 passes against eight float32 MFMAs of 16 passes per 32x32x16 block: 0.56 of the MFMA time
// (six: 0.375).  The split costs ~5.5 VALU instructions per operand element, once per element and tile: the weight is
// split on the host, once ([3][N][K] bf16, openpifpaf_amd.fused.split_weight), the activation while its tile is staged.
// Error against a float64 product: tests/test_gpu_gemm_x3.py, tools/gpu/gemm_x3_probe.py (both variants next to
// gemm_f32.hip's float32 MFMA and torch's float32 convolution): six terms 2.4-2.8x BELOW the float32 MFMA's.
// Measured (profiles/r6/gemm_x3_probe.log, gemm_x3_pmc.log): 0.62 ms against 0.93 on the layer-3 reduce shape; the kernel is
// POWER-limited -- the clock falls to 1.8 GHz and the bf16 MFMA sustains 1.0e9 busy cycles per second per SIMD with six and with
// nine terms alike, so MFMA time adds to the launch however well the rest overlaps.
// The same kernel takes a SECOND activation along K (SRC = 1: a block's last 1x1 convolution + its downsampling convolution as
// one product) or NINE / EIGHT shifted views of one activation (SRC = 2: strided 3x3 convolutions and the 7x7 stem as implicit
// GEMMs; padding through out-of-range buffer offsets).
//
// Tile 128 x BN (BN = 128 | 64) per 256-thread workgroup, BK = 32; 4 waves as 2(M) x 2(N), a wave 64 x BN/2 of 32x32
// blocks; operands K-major in LDS, three bf16 planes each, rows of 64 B whose 16-byte chunks are XOR-swizzled by (row / 4) % 4
// (ds_read_b128 fragments of 16 consecutive rows fall on 16 different bank groups AND the 8-byte plane stores of the split do
// not collide: with rows padded to 80 B a third of the LDS cycles were store conflicts; profiles/r6/gemm_x3_swizzle_ab.log:
// 2-7 % on the reduce shapes); XCD-aware tile order and epilogue as in gemm_f32.hip.
#include "common.hpp"

namespace opa {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;

#ifndef OPA_X3_BK                  // tuning switches (tools/gpu/gemm_x3_probe.py builds the variants): K-step,
#define OPA_X3_BK 32               // workgroups per compute unit the registers are budgeted for,
#endif
#ifndef OPA_X3_WGS
#define OPA_X3_WGS 2
#endif
#ifndef OPA_X3_DIAG                // timing experiments (WRONG results): 1 no split arithmetic, 2 no LDS stores in the K loop,
#define OPA_X3_DIAG 0              // 3 only the leading MFMA of every block, 4 fragments read once per K-step (kk = 0 only)
#endif
#ifndef OPA_X3_ONE_ACC             // 1: the corrections go to the leading products' accumulators (64 registers less)
#define OPA_X3_ONE_ACC 0
#endif
#ifndef OPA_X3_SWIZZLE             // 1: LDS rows without padding, the 16-byte chunks of a row XOR-swizzled by (row / 4) % 4: fragment
#define OPA_X3_SWIZZLE 1           //    reads of 16 consecutive rows AND the 8-byte plane stores of the split fall on distinct banks
#endif                             //    (0: rows 80 bytes apart -- the stores collide two-way, a third of the LDS cycles, profiles/r6/gemm_x3_pmc.log)
constexpr int kX3BM = 128, kX3BK = OPA_X3_BK, kX3Pitch = OPA_X3_SWIZZLE ? kX3BK : kX3BK + 8;      // LDS row pitch in bf16
// bf16 offset of element k (a multiple of 4) of row `row` in a plane
__device__ __forceinline__ int x3_lds(int row, int k) {
    if (OPA_X3_SWIZZLE && kX3BK == 32) return row * kX3Pitch + ((((k >> 3) ^ (row >> 2)) & 3) << 3) + (k & 7);
    return row * kX3Pitch + k;
}

// four float32 -> their three bf16 pieces, packed pairwise (element e in the low half of word e / 2 ... K-major order)
__device__ __forceinline__ void split4(const f32x4_t a, u32x2_t& p1, u32x2_t& p2, u32x2_t& p3) {
    unsigned u[4], v[4], w[4];
#pragma unroll
    for (int e = 0; e < 4; e++) {
        u[e] = __float_as_uint(a[e]);
        const float r1 = a[e] - __uint_as_float(u[e] & 0xffff0000u);       // exact: the low 16 significand bits
        v[e] = __float_as_uint(r1);
        const float r2 = r1 - __uint_as_float(v[e] & 0xffff0000u);         // exact: at most 8 significant bits are left
        w[e] = __float_as_uint(r2);
    }
    // high halves of two words side by side: bytes {hi.3, hi.2, lo.3, lo.2}
    p1[0] = __builtin_amdgcn_perm(u[1], u[0], 0x07060302u); p1[1] = __builtin_amdgcn_perm(u[3], u[2], 0x07060302u);
    p2[0] = __builtin_amdgcn_perm(v[1], v[0], 0x07060302u); p2[1] = __builtin_amdgcn_perm(v[3], v[2], 0x07060302u);
    p3[0] = __builtin_amdgcn_perm(w[1], w[0], 0x07060302u); p3[1] = __builtin_amdgcn_perm(w[3], w[2], 0x07060302u);
}

// A SECOND activation behind the first along K (TWO): out = act([A | A2'] * W^T + bias) with A2'[m] = the pixel of A2 that
// output pixel m reads through a 1x1 convolution of stride `stride` -- the last 1x1 convolution of a ResNet block and the
// block's downsampling convolution as ONE product (the weights concatenated along K on the host), so that the identity tensor
// is neither written nor read back.  Columns [0, K1) come from A (row length K1), [K1, K) from A2 (row length K - K1).
// NINE activations along K (SRC = 2): a 3x3 convolution with padding 1 and any stride as an implicit GEMM -- column block
// t = 3 ky + kx of K holds the C channels of input pixel (stride * oy - 1 + ky, stride * ox - 1 + kx); a pixel in the padding is
// requested beyond the end of the buffer, which a buffer load answers with zeros (one bit per row and tap decides).
struct X3Second {
    const float* A2;           // [B, hi, wi, K - K1] channels-last, or null
    int K1;                    // columns of the first activation (a multiple of the K-step)
    int ho_wo, wo, hi_wi, wi, stride;
    int hi, C, batch;          // (SRC = 2) input rows, floats per tap (a multiple of the K-step), images
    int pix, taps_x, ntaps, padded;   // (SRC = 2) floats per input pixel; taps per window row; taps; 1: the input is padded in
                               // memory (window origin = (stride oy, stride ox), every tap valid), 0: padding 1 by the tap mask
};

template <int BN, bool RES, bool RELU, bool PRO, int TERMS, int SRC>
__global__ __launch_bounds__(256, OPA_X3_WGS) void gemm_f32x3_bias_act_kernel(
        const float* __restrict__ A, const unsigned short* __restrict__ W3, const float* __restrict__ bias,
        const float* __restrict__ res, float* __restrict__ out, int M, int N, int K, const float* __restrict__ a_bias,
        const X3Second sec) {
    constexpr int WN = BN / 2;                 // wave tile width
    constexpr int NT = WN / 32;                // 32-wide MFMA blocks per wave in N (2 or 1)
    constexpr int LDS_A = kX3BM * kX3Pitch, LDS_B = BN * kX3Pitch;        // bf16 elements of ONE plane
#ifndef OPA_X3_PAD_LDS            // experiment: this many bytes of LDS more (fewer workgroups per compute unit)
#define OPA_X3_PAD_LDS 0
#endif
    constexpr int STAGE_BYTES = 3 * (LDS_A + LDS_B) * 2 + OPA_X3_PAD_LDS;
    constexpr int EPI_BYTES = 4 * 32 * WN * 4; // per wave a 32 x WN f32 patch
    __shared__ __attribute__((aligned(16))) unsigned char smem[STAGE_BYTES > EPI_BYTES ? STAGE_BYTES : EPI_BYTES];
    unsigned short* sA = reinterpret_cast<unsigned short*>(smem);          // [3][128][pitch]
    unsigned short* sB = sA + 3 * LDS_A;                                    // [3][BN][pitch]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int n_tiles = N / BN;
    // XCD-aware tile order (see gemm_epilogue.hip): the N-tiles sharing one A row-block run on ONE L2
    const unsigned nwg = gridDim.x, xcd = blockIdx.x & 7u, in_xcd = blockIdx.x >> 3;
    const unsigned q = nwg >> 3, r8 = nwg & 7u;
    const unsigned logical = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + in_xcd;
    const int m0 = (int)(logical / n_tiles) * kX3BM;
    const int n0 = (int)(logical % n_tiles) * BN;

    constexpr int NL = OPA_X3_ONE_ACC ? 1 : 2;
    f32x16_t acc[2][NT], low_[NL][NT];         // the leading products a1*b1; everything else
    f32x16_t (&low)[2][NT] = OPA_X3_ONE_ACC ? acc : reinterpret_cast<f32x16_t (&)[2][NT]>(low_);
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < NT; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) { acc[i][j][r] = 0.0f; if (!OPA_X3_ONE_ACC) low[i][j][r] = 0.0f; }

    // staging maps.  A: a 32-float row is 8 x 16 B, 256 threads cover 32 rows per pass.  W3: a 32-bf16 row of one plane is
    // 4 x 16 B; vector v of the 3 * BN * 4 of a K-step = (plane, row, quarter)
    constexpr int VA = kX3BK / 4;              // 16-B vectors of an A row (8 | 4)
    constexpr int RPP = 256 / VA;              // A rows per pass (32 | 64)
    constexpr int NPA = kX3BM / RPP;           // passes (4 | 2)
    constexpr int WQ = kX3BK / 8;              // 16-B vectors of a W row of one plane (4 | 2)
    const int s_row = tid / VA, s_col = (tid % VA) * 4;
    constexpr int WTOT = 3 * BN * WQ;          // W vectors of a K-step
    constexpr int WV = (WTOT + 255) / 256;     // ... per thread
    f32x4_t ra0[NPA], ra1[NPA];                // the activation travels TWO K-steps ahead (HBM), the weight one (L2)
    u32x4_t rb[WV];
    // uniform 64-bit bases (scalar registers) + one 32-bit offset per load: the loads keep their address registers to
    // themselves (with 64-bit per-thread pointers the compiler, short of registers, computed every address INTO the load's
    // destination, which made each K-step wait for all loads in flight before it could issue its own)
    // (buffer loads: a descriptor in scalar registers, ONE 32-bit byte offset per load, the K-step as the scalar offset)
    constexpr bool TWO = SRC == 1, TAPS = SRC == 2;
    const int rows_here = M - m0 < kX3BM ? M - m0 : kX3BM;
    const int KA = TWO ? sec.K1 : TAPS ? sec.C : K;          // row length of the first activation
    // (TAPS: the whole tensor, from wi + 1 pixels BEFORE its start -- the window of output pixel (oy, ox) begins at input pixel
    //  (stride oy - 1, stride ox - 1); what lies before the tensor is only ever asked for by rows whose tap bit is clear)
    const int shift = TAPS && !sec.padded ? (sec.wi + 1) * sec.pix : 0;
    const float* a1_base = TAPS ? A - (size_t)shift : A + (size_t)m0 * KA;
    const int a1_bytes = TAPS ? (int)(((size_t)sec.batch * sec.hi_wi * sec.pix + shift) * 4) : (int)((size_t)rows_here * KA * 4);
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a1_base), 0, a1_bytes, 0x00020000);
    // the second activation: its rows are the input pixels the tile's output pixels read (monotonic in m: offsets from the first)
    const int K2 = K - KA;
    auto in_row = [&](int m) -> long long {
        const int b = m / sec.ho_wo, r = m - b * sec.ho_wo, oy = r / sec.wo, ox = r - oy * sec.wo;
        return (long long)b * sec.hi_wi + (long long)oy * sec.stride * sec.wi + (long long)ox * sec.stride;
    };
    long long row0_2 = 0;
    const float* a2_base = nullptr;
    int a2_bytes = 0;
    unsigned pa2[NPA];
    if constexpr (TWO) {
        row0_2 = in_row(__builtin_amdgcn_readfirstlane(m0));
        const int m_end = m0 + rows_here - 1;
        a2_base = sec.A2 + (size_t)row0_2 * K2;
        a2_bytes = (int)((size_t)(in_row(__builtin_amdgcn_readfirstlane(m_end)) - row0_2 + 1) * K2 * 4);
    }
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned short*>(W3 + (size_t)n0 * K), 0, (int)(((size_t)3 * N - n0) * K * 2), 0x00020000);
    unsigned pa[NPA], pb[WV];
    unsigned vmask[NPA];                       // (TAPS) bit t: tap t of this row lies inside the image
    int sb_off[WV];
#pragma unroll
    for (int p = 0; p < NPA; p++) {            // rows past M read row M - 1 (valid memory; the epilogue never stores them)
        int m = m0 + p * RPP + s_row;
        if (m > M - 1) m = M - 1;
        pa[p] = ((unsigned)(m - m0) * (unsigned)KA + (unsigned)s_col) * 4u;
        vmask[p] = 0u;
        if constexpr (TAPS) {
            const int b = m / sec.ho_wo, r = m - b * sec.ho_wo, oy = r / sec.wo, ox = r - oy * sec.wo;
            pa[p] = (unsigned)((long long)b * sec.hi_wi + (long long)oy * sec.stride * sec.wi + (long long)ox * sec.stride) * (unsigned)sec.pix * 4u
                    + (unsigned)s_col * 4u;
            if (sec.padded) vmask[p] = 0xffffffffu;
            else
                for (int t = 0; t < sec.ntaps; t++) {
                    const int iy = oy * sec.stride - 1 + t / sec.taps_x, ix = ox * sec.stride - 1 + t % sec.taps_x;
                    if (iy >= 0 && iy < sec.hi && ix >= 0 && ix < sec.wi) vmask[p] |= 1u << t;
                }
        }
        if constexpr (TWO) pa2[p] = ((unsigned)(in_row(m) - row0_2) * (unsigned)K2 + (unsigned)s_col) * 4u;
    }
#pragma unroll
    for (int t = 0; t < WV; t++) {
        const int v = t * 256 + tid < WTOT ? t * 256 + tid : WTOT - 1;     // (a thread without a vector of its own repeats the last one)
        const int plane = v / (BN * WQ), rem = v - plane * (BN * WQ), row = rem / WQ, c = (rem % WQ) * 8;
        pb[t] = (((unsigned)plane * (unsigned)N + (unsigned)row) * (unsigned)K + (unsigned)c) * 2u;
        sb_off[t] = plane * LDS_B + x3_lds(row, c);
    }
    auto fetch_a = [&](f32x4_t (&ra)[NPA], int k0) {      // global -> registers for K-step k0 (with the operand prologue)
        if constexpr (TAPS) {                  // tap t = k0 / C: a uniform shift of every row's window origin
            const int t = k0 / KA, kc = k0 - t * KA;
            const int soff = (((t / sec.taps_x) * sec.wi + t % sec.taps_x) * sec.pix + kc) * 4;
#pragma unroll
            for (int p = 0; p < NPA; p++)
                ra[p] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, (vmask[p] >> t) & 1u ? pa[p] : 0xFFFFFFF0u, soff, 0));
        } else if constexpr (TWO) {            // (uniform selects, no branch around the loads)
            const bool second = k0 >= KA;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(second ? a2_base : a1_base), 0, second ? a2_bytes : a1_bytes, 0x00020000);
            const int ks = (second ? k0 - KA : k0) * 4;
#pragma unroll
            for (int p = 0; p < NPA; p++)
                ra[p] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rs, second ? pa2[p] : pa[p], ks, 0));
        } else {
#pragma unroll
            for (int p = 0; p < NPA; p++) ra[p] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, pa[p], k0 * 4, 0));
        }
        if (PRO) {                             // the preceding convolution's bias + ReLU, applied to the raw operand
            const f32x4_t ab = *reinterpret_cast<const f32x4_t*>(a_bias + k0 + s_col);
#pragma unroll
            for (int p = 0; p < NPA; p++)
#pragma unroll
                for (int e = 0; e < 4; e++) ra[p][e] = fmaxf(ra[p][e] + ab[e], 0.0f);
        }
    };
    auto fetch_b = [&](int k0) {
#pragma unroll
        for (int t = 0; t < WV; t++) rb[t] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, pb[t], k0 * 2, 0));
    };
    auto store = [&](const f32x4_t (&ra)[NPA]) {          // registers -> LDS: the activation is split here, once per element and tile
#pragma unroll
        for (int p = 0; p < NPA; p++) {
            u32x2_t p1, p2, p3;
            if (OPA_X3_DIAG == 1) { p1[0] = __float_as_uint(ra[p][0]); p1[1] = __float_as_uint(ra[p][2]); p2 = p1; p3 = p1; }
            else split4(ra[p], p1, p2, p3);
            unsigned short* d = sA + x3_lds(p * RPP + s_row, s_col);
            *reinterpret_cast<u32x2_t*>(d) = p1;
            *reinterpret_cast<u32x2_t*>(d + LDS_A) = p2;
            *reinterpret_cast<u32x2_t*>(d + 2 * LDS_A) = p3;
        }
#pragma unroll
        for (int t = 0; t < WV; t++)
            if (WTOT % 256 == 0 || t * 256 + tid < WTOT) *reinterpret_cast<u32x4_t*>(sB + sb_off[t]) = rb[t];
    };
    // one K-step: LDS holds tile k0, `cur` the activation of tile k0 + BK (requested a step ago), `nxt` is free.  The weight's
    // loads are issued BEFORE the activation's: vmcnt counts in order, so the wait for the weight of tile k0 + BK (behind this
    // step's MFMAs) leaves the activation of tile k0 + 2 BK in flight
    auto step = [&](int k0, f32x4_t (&cur)[NPA], f32x4_t (&nxt)[NPA]) {
        // (no branch around a load: past the end the last tile is requested again -- behind a conditional load the compiler's
        // wait counts assume the worst path, i.e. they wait for everything in flight, which is what two steps ahead is there to avoid)
        const bool more = k0 + kX3BK < K;
        const int k_last = K - kX3BK;
        fetch_b(k0 + kX3BK < k_last ? k0 + kX3BK : k_last);
        fetch_a(nxt, k0 + 2 * kX3BK < k_last ? k0 + 2 * kX3BK : k_last);
        __builtin_amdgcn_sched_barrier(0);     // (the scheduler otherwise sinks the loads behind the MFMAs and lets fragments share their registers)
#pragma unroll
        for (int kk = 0; kk < kX3BK; kk += 16) {
            bf16x8_t fa[3][2], fb[3][NT];
            const int kof = (OPA_X3_DIAG == 4 ? 0 : kk) + (lane >> 5) * 8;
#pragma unroll
            for (int pl = 0; pl < 3; pl++) {
#pragma unroll
                for (int i = 0; i < 2; i++)
                    fa[pl][i] = *reinterpret_cast<const bf16x8_t*>(sA + pl * LDS_A + x3_lds(wm * 64 + i * 32 + (lane & 31), kof));
#pragma unroll
                for (int j = 0; j < NT; j++)
                    fb[pl][j] = *reinterpret_cast<const bf16x8_t*>(sB + pl * LDS_B + x3_lds(wn * WN + j * 32 + (lane & 31), kof));
            }
            // consecutive MFMAs go to different accumulators (a dependent MFMA waits for its predecessor's passes)
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < NT; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][i], fb[0][j], acc[i][j], 0, 0, 0);
            // the corrections, smallest first
#pragma unroll
            for (int s = 4; s >= 1; s--) {
                if ((TERMS == 6 && s > 2) || OPA_X3_DIAG == 3) continue;
#pragma unroll
                for (int pa_ = 0; pa_ < 3; pa_++) {
                    const int pb_ = s - pa_;
                    if (pb_ < 0 || pb_ > 2) continue;
#pragma unroll
                    for (int i = 0; i < 2; i++)
#pragma unroll
                        for (int j = 0; j < NT; j++)
                            low[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[pa_][i], fb[pb_][j], low[i][j], 0, 0, 0);
                }
            }
        }
        __syncthreads();                       // every wave is done reading the stage
        if (more && (OPA_X3_DIAG != 2 || cur[0][0] == 12345.678f)) store(cur);
        __syncthreads();
    };
    fetch_b(0);
    fetch_a(ra0, 0);
    store(ra0);
    __syncthreads();
    fetch_a(ra1, kX3BK);                       // (K is a multiple of 2 BK: the launcher checks)
    for (int k0 = 0; k0 < K; k0 += 2 * kX3BK) {
        step(k0, ra1, ra0);
        step(k0 + kX3BK, ra0, ra1);
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
                patch[row * WN + col] = (OPA_X3_ONE_ACC ? acc[i][j][r] : acc[i][j][r] + low[i][j][r]) + b;
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
                *reinterpret_cast<f32x4_t*>(out + (size_t)m * N + n0 + wn * WN + c4) = f;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        asm volatile("" ::: "memory");         // (keeps the second block's residual loads behind this block)
    }
}

template <int BN, bool PRO, int TERMS>
static hipError_t launch_x3_bn(const float* a, const unsigned short* w, const float* b, const float* r, float* o,
                               int M, int N, int K, int relu, const float* ab, hipStream_t st, const X3Second& sec) {
    const long long blocks = (long long)((M + kX3BM - 1) / kX3BM) * (N / BN);
    if (sec.C > 0) {                           // 3x3 implicit GEMM: no operand prologue, no residual
        if constexpr (!PRO) {
            if (relu) gemm_f32x3_bias_act_kernel<BN, false, true, false, TERMS, 2><<<(unsigned)blocks, 256, 0, st>>>(a, w, b, nullptr, o, M, N, K, nullptr, sec);
            else gemm_f32x3_bias_act_kernel<BN, false, false, false, TERMS, 2><<<(unsigned)blocks, 256, 0, st>>>(a, w, b, nullptr, o, M, N, K, nullptr, sec);
        }
    } else if (sec.A2) {                       // (the pair has no residual: the second activation IS the identity branch)
        if (relu) gemm_f32x3_bias_act_kernel<BN, false, true, PRO, TERMS, 1><<<(unsigned)blocks, 256, 0, st>>>(a, w, b, nullptr, o, M, N, K, ab, sec);
        else gemm_f32x3_bias_act_kernel<BN, false, false, PRO, TERMS, 1><<<(unsigned)blocks, 256, 0, st>>>(a, w, b, nullptr, o, M, N, K, ab, sec);
    } else if (r) {
        if (relu) gemm_f32x3_bias_act_kernel<BN, true, true, PRO, TERMS, 0><<<(unsigned)blocks, 256, 0, st>>>(a, w, b, r, o, M, N, K, ab, sec);
        else gemm_f32x3_bias_act_kernel<BN, true, false, PRO, TERMS, 0><<<(unsigned)blocks, 256, 0, st>>>(a, w, b, r, o, M, N, K, ab, sec);
    } else {
        if (relu) gemm_f32x3_bias_act_kernel<BN, false, true, PRO, TERMS, 0><<<(unsigned)blocks, 256, 0, st>>>(a, w, b, r, o, M, N, K, ab, sec);
        else gemm_f32x3_bias_act_kernel<BN, false, false, PRO, TERMS, 0><<<(unsigned)blocks, 256, 0, st>>>(a, w, b, r, o, M, N, K, ab, sec);
    }
    return hipGetLastError();
}

template <int TERMS>
static hipError_t launch_x3_terms(const float* A, const unsigned short* W3, const float* bias, const float* res, float* out,
                                  int M, int N, int K, int relu, hipStream_t st, const float* a_bias, const X3Second& sec) {
    if (a_bias) {
        if (N % 128 == 0) return launch_x3_bn<128, true, TERMS>(A, W3, bias, res, out, M, N, K, relu, a_bias, st, sec);
        return launch_x3_bn<64, true, TERMS>(A, W3, bias, res, out, M, N, K, relu, a_bias, st, sec);
    }
    if (N % 128 == 0) return launch_x3_bn<128, false, TERMS>(A, W3, bias, res, out, M, N, K, relu, nullptr, st, sec);
    return launch_x3_bn<64, false, TERMS>(A, W3, bias, res, out, M, N, K, relu, nullptr, st, sec);
}

hipError_t launch_gemm_f32x3_bias_act(const float* A, const unsigned short* W3, const float* bias, const float* res, float* out,
                                      int M, int N, int K, int relu, int terms, hipStream_t st, const float* a_bias) {
    X3Second none; none.A2 = nullptr; none.K1 = K; none.ho_wo = none.wo = none.hi_wi = none.wi = none.stride = 1; none.hi = 1; none.C = 0; none.batch = 1;
    none.pix = 1; none.taps_x = 1; none.ntaps = 1; none.padded = 0;
    if (terms == 6) return launch_x3_terms<6>(A, W3, bias, res, out, M, N, K, relu, st, a_bias, none);
    return launch_x3_terms<9>(A, W3, bias, res, out, M, N, K, relu, st, a_bias, none);
}

// out[B, ho, wo, N] = act([A1 | pixels of A2 at stride s] * W3cat^T + bias): A1 [B*ho*wo, K1], A2 [B, hi, wi, K2], W3cat [3][N][K1 + K2]
hipError_t launch_gemm2_f32x3_bias_act(const float* A1, int K1, const float* A2, int K2, int batch, int hi, int wi, int stride,
                                       const unsigned short* W3, const float* bias, float* out, int N, int relu, int terms,
                                       hipStream_t st, const float* a_bias) {
    const int ho = (hi - 1) / stride + 1, wo = (wi - 1) / stride + 1;
    X3Second sec; sec.A2 = A2; sec.K1 = K1; sec.ho_wo = ho * wo; sec.wo = wo; sec.hi_wi = hi * wi; sec.wi = wi; sec.stride = stride;
    sec.hi = hi; sec.C = 0; sec.batch = batch; sec.pix = 1; sec.taps_x = 1; sec.ntaps = 1; sec.padded = 0;
    const int M = batch * ho * wo, K = K1 + K2;
    if (terms == 6) return launch_x3_terms<6>(A1, W3, bias, nullptr, out, M, N, K, relu, st, a_bias, sec);
    return launch_x3_terms<9>(A1, W3, bias, nullptr, out, M, N, K, relu, st, a_bias, sec);
}

// 3x3 convolution, padding 1, stride s: out[B, ho, wo, N] = act(im2col(x) * W3^T + bias), x [B, hi, wi, C] channels-last,
// W3 = split_weight of the weight as [N, (ky, kx, c)] ([3][N][9 C]); C % 32 == 0, 9 C % 64 == 0, tensor < 2 GB
hipError_t launch_conv3x3_f32x3(const float* x, int batch, int hi, int wi, int C, int stride, const unsigned short* W3,
                                const float* bias, float* out, int N, int relu, int terms, hipStream_t st) {
    const int ho = (hi - 1) / stride + 1, wo = (wi - 1) / stride + 1;
    X3Second sec; sec.A2 = nullptr; sec.K1 = C; sec.ho_wo = ho * wo; sec.wo = wo; sec.hi_wi = hi * wi; sec.wi = wi; sec.stride = stride;
    sec.hi = hi; sec.C = C; sec.batch = batch; sec.pix = C; sec.taps_x = 3; sec.ntaps = 9; sec.padded = 0;
    const int M = batch * ho * wo, K = 9 * C;
    if (terms == 6) return launch_x3_terms<6>(x, W3, bias, nullptr, out, M, N, K, relu, st, nullptr, sec);
    return launch_x3_terms<9>(x, W3, bias, nullptr, out, M, N, K, relu, st, nullptr, sec);
}

// A convolution whose window ROWS are the taps: x [B, hp, wp, pix] is padded in memory (pix floats per pixel), output pixel
// (oy, ox) reads, for tap t, the `tap_floats` contiguous floats that begin at input pixel (stride oy + t, stride ox) -- the 7x7
// stride-2 stem of a ResNet on a 4-channel copy of the image: 8 taps of 8 pixels x 4 channels (the eighth row and column and the
// fourth channel meet zero weights).  W3 [3][N][ntaps * tap_floats].  out [B, ho, wo, N].
hipError_t launch_convrows_f32x3(const float* x, int batch, int hp, int wp, int pix, int ho, int wo, int stride, int ntaps, int tap_floats,
                                 const unsigned short* W3, const float* bias, float* out, int N, int relu, int terms, hipStream_t st) {
    X3Second sec; sec.A2 = nullptr; sec.K1 = tap_floats; sec.ho_wo = ho * wo; sec.wo = wo; sec.hi_wi = hp * wp; sec.wi = wp; sec.stride = stride;
    sec.hi = hp; sec.C = tap_floats; sec.batch = batch; sec.pix = pix; sec.taps_x = 1; sec.ntaps = ntaps; sec.padded = 1;
    const int M = batch * ho * wo, K = ntaps * tap_floats;
    if (terms == 6) return launch_x3_terms<6>(x, W3, bias, nullptr, out, M, N, K, relu, st, nullptr, sec);
    return launch_x3_terms<9>(x, W3, bias, nullptr, out, M, N, K, relu, st, nullptr, sec);
}

}  // namespace opa
