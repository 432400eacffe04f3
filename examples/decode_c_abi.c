/* Minimal C host for the drop-in boundary (include/openpifpaf_amd.h): decodes one batch of CIF/CAF fields that
 * already sit in device memory.  Plain C99 + the HIP runtime for the two allocations; no torch.
 *
 *   hipcc -x c examples/decode_c_abi.c -Iinclude -Lopenpifpaf_amd/lib -lopenpifpaf_amd -Wl,-rpath,openpifpaf_amd/lib -o decode_c_abi
 *   (or any C compiler with -I/opt/rocm/include -L/opt/rocm/lib -lamdhip64 -D__HIP_PLATFORM_AMD__)
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <hip/hip_runtime_api.h>

#include "openpifpaf_amd.h"

/* COCO person skeleton, 0-based (reference plugins/coco/constants.py:16-20) */
static const int64_t SKELETON[19][2] = {
    {15, 13}, {13, 11}, {16, 14}, {14, 12}, {11, 12}, {5, 11}, {6, 12}, {5, 6}, {5, 7}, {6, 8},
    {7, 9}, {8, 10}, {1, 2}, {0, 1}, {0, 2}, {1, 3}, {2, 4}, {3, 5}, {4, 6}};

#define CHECK_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { \
    fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_)); return 1; } } while (0)
#define CHECK_OPA(call) do { int rc_ = (call); if (rc_ != OPA_OK) { \
    fprintf(stderr, "%s: error %d: %s\n", #call, rc_, opa_last_error()); return 1; } } while (0)

int main(void) {
    opa_shape shape;
    opa_cifcaf* decoder = NULL;
    float *cif = NULL, *caf = NULL, *out = NULL;
    int64_t* ids = NULL;
    int32_t* counts = NULL;
    void* workspace = NULL;
    size_t cif_bytes, caf_bytes, ws_bytes;
    int32_t host_counts[2];

    printf("%s, %d device(s)\n", opa_version(), opa_device_count());
    memset(&shape, 0, sizeof(shape));
    shape.batch = 2; shape.n_cif = 17; shape.n_caf = 19;
    shape.cif_h = shape.caf_h = 41; shape.cif_w = shape.caf_w = 41;
    shape.cif_stride = shape.caf_stride = 8; shape.max_annotations = 64;

    CHECK_OPA(opa_cifcaf_create(&decoder, 17, &SKELETON[0][0], 19));
    cif_bytes = sizeof(float) * 2 * 17 * 5 * 41 * 41;
    caf_bytes = sizeof(float) * 2 * 19 * 8 * 41 * 41;
    ws_bytes = opa_cifcaf_workspace_bytes(&shape);
    if (ws_bytes == 0) { fprintf(stderr, "%s\n", opa_last_error()); return 1; }
    CHECK_HIP(hipMalloc((void**)&cif, cif_bytes));
    CHECK_HIP(hipMalloc((void**)&caf, caf_bytes));
    CHECK_HIP(hipMalloc(&workspace, ws_bytes));
    CHECK_HIP(hipMalloc((void**)&out, sizeof(float) * 2 * 64 * 17 * 4));
    CHECK_HIP(hipMalloc((void**)&ids, sizeof(int64_t) * 2 * 64));
    CHECK_HIP(hipMalloc((void**)&counts, sizeof(int32_t) * 2));
    CHECK_HIP(hipMemset(cif, 0, cif_bytes));           /* a network would have written its heads here */
    CHECK_HIP(hipMemset(caf, 0, caf_bytes));

    CHECK_OPA(opa_cifcaf_decode(decoder, &shape, NULL, cif, caf, NULL, NULL, 0, workspace, ws_bytes,
                                out, ids, counts, NULL /* default stream */));
    CHECK_HIP(hipMemcpy(host_counts, counts, sizeof(host_counts), hipMemcpyDeviceToHost));
    printf("annotations per image: %d %d\n", host_counts[0], host_counts[1]);

    /* ---- several batches in flight: one decoder handle, workspace, output block and HIP
     * stream PER LANE.  A decode allocates nothing and synchronises nothing, so batch i+1 is queued on its lane while
     * batch i is still decoding on the other one; the annotations travel to pinned host memory on the lane's stream
     * and an event says when they are there.  (The fields of a lane must stay untouched until its event has fired.) */
    {
        enum { LANES = 2, BATCHES = 6 };
        opa_cifcaf* dec[LANES]; void* ws[LANES]; float* o[LANES]; int64_t* id[LANES]; int32_t* cn[LANES];
        hipStream_t st[LANES]; hipEvent_t done[LANES]; int32_t* host_cn[LANES]; float* host_o[LANES];
        const size_t out_bytes = sizeof(float) * 2 * 64 * 17 * 4;
        int l, i, total = 0;
        for (l = 0; l < LANES; l++) {
            CHECK_OPA(opa_cifcaf_create(&dec[l], 17, &SKELETON[0][0], 19));
            CHECK_HIP(hipMalloc(&ws[l], ws_bytes));
            CHECK_HIP(hipMemset(ws[l], 0, 256));                 /* workspace contract: the header starts out invalid */
            CHECK_HIP(hipMalloc((void**)&o[l], out_bytes));
            CHECK_HIP(hipMalloc((void**)&id[l], sizeof(int64_t) * 2 * 64));
            CHECK_HIP(hipMalloc((void**)&cn[l], sizeof(int32_t) * 2));
            CHECK_HIP(hipHostMalloc((void**)&host_cn[l], sizeof(int32_t) * 2, hipHostMallocDefault));
            CHECK_HIP(hipHostMalloc((void**)&host_o[l], out_bytes, hipHostMallocDefault));
            CHECK_HIP(hipStreamCreateWithFlags(&st[l], hipStreamNonBlocking));
            CHECK_HIP(hipEventCreateWithFlags(&done[l], hipEventDisableTiming));
        }
        CHECK_HIP(hipDeviceSynchronize());
        for (i = 0; i < BATCHES + LANES; i++) {
            l = i % LANES;
            if (i >= LANES) {                                    /* collect what this lane decoded last time */
                CHECK_HIP(hipEventSynchronize(done[l]));
                if ((host_cn[l][0] | host_cn[l][1]) & OPA_COUNT_FAILED) { fprintf(stderr, "decode failed\n"); return 1; }
                total += OPA_COUNT_ROWS(host_cn[l][0]) + OPA_COUNT_ROWS(host_cn[l][1]);
            }
            if (i < BATCHES) {                                   /* ... and queue its next batch */
                CHECK_OPA(opa_cifcaf_decode(dec[l], &shape, NULL, cif, caf, NULL, NULL, 0, ws[l], ws_bytes,
                                            o[l], id[l], cn[l], st[l]));
                CHECK_HIP(hipMemcpyAsync(host_o[l], o[l], out_bytes, hipMemcpyDeviceToHost, st[l]));
                CHECK_HIP(hipMemcpyAsync(host_cn[l], cn[l], sizeof(int32_t) * 2, hipMemcpyDeviceToHost, st[l]));
                CHECK_HIP(hipEventRecord(done[l], st[l]));
            }
        }
        printf("%d batches over %d lanes: %d annotations\n", BATCHES, LANES, total);
        for (l = 0; l < LANES; l++) {
            opa_cifcaf_destroy(dec[l]);
            hipFree(ws[l]); hipFree(o[l]); hipFree(id[l]); hipFree(cn[l]); hipHostFree(host_cn[l]); hipHostFree(host_o[l]);
            hipStreamDestroy(st[l]); hipEventDestroy(done[l]);
        }
    }

    opa_cifcaf_destroy(decoder);
    hipFree(cif); hipFree(caf); hipFree(workspace); hipFree(out); hipFree(ids); hipFree(counts);
    return 0;
}
