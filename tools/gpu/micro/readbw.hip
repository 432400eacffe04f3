// Micro-benchmark behind DESIGN's stage-kernel rooflines (round 6): what a read-only stream reaches on this chip with the
// access shapes the stage kernels use -- 4-byte loads per lane, several of them in flight per thread, 16-byte loads, and
// plane-strided reads of a [planes][8][6561] tensor that skip one plane in eight (the CAF field's unused component).
// hipcc --offload-arch=gfx950 -O3 -o readbw readbw.hip && ./readbw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int U>
__global__ __launch_bounds__(256) void read_dword(const float* __restrict__ p, size_t n, float* out) {
    float acc = 0.f;
    const size_t stride = (size_t)gridDim.x * 256 * U;
    for (size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x; i < n; i += stride) {
        float v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = i + u * 256 < n ? p[i + u * 256] : 0.f;
#pragma unroll
        for (int u = 0; u < U; u++) acc += v[u];
    }
    if (acc == 123.456f) out[0] = acc;
}
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
template <int U>
__global__ __launch_bounds__(256) void read_x4(const float* __restrict__ p, size_t n, float* out, int misalign) {
    float acc = 0.f;
    const size_t stride = (size_t)gridDim.x * 256 * U * 4;
    for (size_t i = ((size_t)blockIdx.x * 256 * U + threadIdx.x) * 4 + misalign; i + 4 * 256 * U < n; i += stride) {
        f4u v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = *reinterpret_cast<const f4u*>(p + i + u * 1024);
#pragma unroll
        for (int u = 0; u < U; u++) acc += v[u].x + v[u].y + v[u].z + v[u].w;
    }
    if (acc == 123.456f) out[0] = acc;
}
// one 512-thread workgroup per plane group of 8 x 6561 floats, 7 of 8 planes, `CELLS` cells per thread and step (cafscored's shape)
template <int CELLS>
__global__ __launch_bounds__(512) void read_planes(const float* __restrict__ p, int HW, float* out) {
    const float* P = p + (size_t)blockIdx.x * 8 * HW;
    float acc = 0.f;
    for (int c0 = 0; c0 < HW; c0 += 512 * CELLS) {
        float v[CELLS][7];
#pragma unroll
        for (int r = 0; r < CELLS; r++) {
            const int o = c0 + r * 512 + threadIdx.x;
#pragma unroll
            for (int k = 0; k < 7; k++) v[r][k] = o < HW ? P[(k + 1) * HW + o] : 0.f;
        }
#pragma unroll
        for (int r = 0; r < CELLS; r++)
#pragma unroll
            for (int k = 0; k < 7; k++) acc += v[r][k];
    }
    if (acc == 123.456f) out[0] = acc;
}

int main(int argc, char** argv) {
    const size_t images = argc > 1 ? (size_t)atoi(argv[1]) : 256;      // 32: the field of one bench batch (112 MB: inside the Infinity Cache)
    const size_t planes = images * 19, HW = 6561;
    const size_t n = planes * 8 * HW;                 // the CAF tensor of `images` COCO images (256: 1.02 GB)
    printf("buffer %.1f MB (%zu images)\n", n * 4 / 1e6, images);
    float *p, *out;
    CK(hipMalloc(&p, n * sizeof(float))); CK(hipMalloc(&out, 256));
    CK(hipMemset(p, 0, n * sizeof(float)));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time = [&](const char* name, auto launch, double bytes) {
        launch(); hipDeviceSynchronize();
        float best = 1e9f;
        for (int rep = 0; rep < 5; rep++) {
            hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        printf("%-44s %8.1f us  %6.2f TB/s\n", name, best * 1e3, bytes / (best * 1e-3) / 1e12);
    };
    const double all = (double)n * 4, seven = all * 7 / 8;
    for (int g : {2048, 8192, 32768}) {
        char nm[96];
        snprintf(nm, 96, "dword x1, grid %d", g); time(nm, [&] { read_dword<1><<<g, 256>>>(p, n, out); }, all);
        snprintf(nm, 96, "dword x4, grid %d", g); time(nm, [&] { read_dword<4><<<g, 256>>>(p, n, out); }, all);
        snprintf(nm, 96, "dword x16, grid %d", g); time(nm, [&] { read_dword<16><<<g, 256>>>(p, n, out); }, all);
        snprintf(nm, 96, "dwordx4 x1 aligned, grid %d", g); time(nm, [&] { read_x4<1><<<g, 256>>>(p, n, out, 0); }, all);
        snprintf(nm, 96, "dwordx4 x4 aligned, grid %d", g); time(nm, [&] { read_x4<4><<<g, 256>>>(p, n, out, 0); }, all);
        snprintf(nm, 96, "dwordx4 x4 misaligned by 4 B, grid %d", g); time(nm, [&] { read_x4<4><<<g, 256>>>(p, n, out, 1); }, all);
    }
    time("planes 7/8, 1 cell per thread and step", [&] { read_planes<1><<<planes, 512>>>(p, (int)HW, out); }, seven);
    time("planes 7/8, 2 cells", [&] { read_planes<2><<<planes, 512>>>(p, (int)HW, out); }, seven);
    time("planes 7/8, 4 cells", [&] { read_planes<4><<<planes, 512>>>(p, (int)HW, out); }, seven);
    time("planes 7/8, 13 cells (whole plane)", [&] { read_planes<13><<<planes, 512>>>(p, (int)HW, out); }, seven);
    return 0;
}
