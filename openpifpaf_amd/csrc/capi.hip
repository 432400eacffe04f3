// C ABI of the library (include/openpifpaf_amd.h): argument checking, workspace
// layout, decoder handle, and the kernel pipeline of one batched decode.
#include "common.hpp"

#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

namespace opa {

static thread_local std::string g_error;
static std::mutex g_params_mutex;
static int g_quiet = 0;
static int g_seed_tie_order = -1;          // -1: not set (OPA_SEED_TIES at load time, else libstdc++'s order)

// The environment is read ONCE, when the library is loaded (static initialisation): no decode looks a variable up.
static int env_int(const char* name, int dflt) { const char* e = std::getenv(name); return e && *e ? std::atoi(e) : dflt; }
static float env_float(const char* name, float dflt) { const char* e = std::getenv(name); return e && *e ? (float)std::atof(e) : dflt; }
static opa_debug debug_from_environment() {
    opa_debug d;
    d.stage_worklist = env_int("OPA_STAGE_WORKLIST", 1) != 0;
    d.fuse_scored = env_int("OPA_FUSE_SCORED", 0) != 0;
    d.scored_one_pass = env_int("OPA_SCORED_ONE_PASS", 1) != 0;
    d.assoc_waves = env_int("OPA_ASSOC_WAVES", 0);
    d.assoc_growers = env_int("OPA_ASSOC_GROWERS", 0);
    d.assoc_bbox = env_int("OPA_ASSOC_BBOX", 1) != 0;
    d.assoc_dedup = env_int("OPA_ASSOC_DEDUP", 1) != 0;
    d.assoc_prededup = env_int("OPA_ASSOC_PREDEDUP", 1) != 0;
    d.assoc_predict = env_int("OPA_ASSOC_PREDICT", 1) != 0;
    d.assoc_predict_min_v = env_float("OPA_ASSOC_PREDICT_MINV", 0.5f);
    d.assoc_predict_th = env_float("OPA_ASSOC_PREDICT_TH", 0.3f);
    d.assoc_collide = env_int("OPA_ASSOC_COLLIDE", 1) != 0;
    d.assoc_collide_shift = env_int("OPA_ASSOC_COLLIDE_SHIFT", 1);
    d.assoc_inherit = env_int("OPA_ASSOC_INHERIT", 1) != 0;
    d.assoc_lookahead = env_int("OPA_ASSOC_LOOKAHEAD", 1) != 0;
    d.assoc_help = env_int("OPA_ASSOC_HELP", 1) != 0;
    d.assoc_spec = env_int("OPA_ASSOC_SPEC", 1) != 0;
    d.assoc_timing = env_int("OPA_ASSOC_TIMING", 0) != 0;
    d.assoc_persistent = env_int("OPA_ASSOC_PERSISTENT", 0);
    d.side_stream = env_int("OPA_SIDE_STREAM", 0);
    d.fc_split = env_int("OPA_FC_SPLIT", 0);
    d.assoc_watchdog_ticks = 100000000ll;
    if (const char* e = std::getenv("OPA_ASSOC_WATCHDOG_TICKS")) { const long long v = std::atoll(e); if (v > 0) d.assoc_watchdog_ticks = v; }
    return d;
}
static const opa_debug g_debug_default = debug_from_environment();
static int tie_order_from_environment() {
    const char* e = std::getenv("OPA_SEED_TIES");
    if (e && (std::strcmp(e, "index") == 0 || std::strcmp(e, "0") == 0)) return 0;
    if (e && (std::strcmp(e, "libstdcxx-fused") == 0 || std::strcmp(e, "2") == 0)) return 2;
    return env_int("OPA_FUSE_TIES", 0) != 0 ? 2 : 1;
}
static const int g_tie_order_default = tie_order_from_environment();

int seed_tie_order() { return g_seed_tie_order >= 0 ? g_seed_tie_order : g_tie_order_default; }

static opa_params default_params() {
    opa_params p;
    p.cif_threshold = 0.3; p.cifhr_neighbors = 16; p.seed_threshold = 0.2; p.caf_threshold = 0.3;
    p.cif_floor = 0.1; p.keypoint_threshold = 0.15; p.keypoint_threshold_rel = 0.5;
    p.nms_suppression = 0.00001; p.nms_instance_threshold = 0.15; p.nms_keypoint_threshold = 0.15;
    p.force_complete_caf_th = 0.001; p.occupancy_reduction = 2.0; p.occupancy_min_scale = 4.0;
    p.greedy = 0; p.reverse_match = 1; p.force_complete = 0; p.block_joints = 0;
    p.ablation_cifseeds_nms = 0; p.ablation_cifseeds_no_rescore = 0;
    p.ablation_caf_no_rescore = 0; p.ablation_cifhr_skip = 0;
    return p;
}
static opa_params g_params = default_params();

struct Profiler {
    bool on = false;
    hipStream_t st = nullptr;
    std::vector<hipEvent_t> events;
    std::vector<const char*> names;
};
static thread_local Profiler g_prof;

void prof_mark(hipStream_t st, const char* name) {
    if (!g_prof.on || st != g_prof.st) return;
    hipEvent_t ev;
    if (hipEventCreate(&ev) != hipSuccess) return;
    if (hipEventRecord(ev, st) != hipSuccess) { (void)hipEventDestroy(ev); return; }
    g_prof.events.push_back(ev);
    g_prof.names.push_back(name);
}

static int fail(int code, const std::string& msg) { g_error = msg; return code; }
static int fail_hip(hipError_t e, const char* where) {
    g_error = std::string(where) + ": " + hipGetErrorString(e);
    return OPA_ERR_HIP;
}

DevParams to_dev(const opa_params& p) {
    DevParams d;
    d.cif_threshold = p.cif_threshold; d.seed_threshold = p.seed_threshold;
    d.caf_threshold = p.caf_threshold; d.cif_floor = p.cif_floor;
    d.keypoint_threshold = p.keypoint_threshold; d.keypoint_threshold_rel = p.keypoint_threshold_rel;
    d.nms_suppression = p.nms_suppression; d.nms_instance_threshold = p.nms_instance_threshold;
    d.nms_keypoint_threshold = p.nms_keypoint_threshold; d.force_complete_caf_th = p.force_complete_caf_th;
    d.occupancy_reduction = p.occupancy_reduction;
    d.occupancy_min_scale_reduced = p.occupancy_min_scale / p.occupancy_reduction;   // occupancy.hpp:29
    {   // dividing by a power of two is multiplying by its (exact) reciprocal: three double divisions less per joint box
        int e = 0;
        const double m = std::frexp(p.occupancy_reduction, &e);
        d.occupancy_inv_reduction = (m == 0.5 && p.occupancy_reduction != 1.0) ? 1.0 / p.occupancy_reduction : 0.0;
    }
    d.cifhr_neighbors = p.cifhr_neighbors;
    d.reverse_match = p.reverse_match; d.force_complete = p.force_complete; d.greedy = p.greedy;
    d.ablation_cifseeds_nms = p.ablation_cifseeds_nms;
    d.ablation_cifseeds_no_rescore = p.ablation_cifseeds_no_rescore;
    d.ablation_caf_no_rescore = p.ablation_caf_no_rescore;
    d.ablation_cifhr_skip = p.ablation_cifhr_skip;
    return d;
}

static bool check_params(const opa_params& p, const char** why) {
    if (!(p.occupancy_reduction > 0.0)) { *why = "occupancy_reduction must be > 0"; return false; }
    if (p.cifhr_neighbors <= 0) { *why = "cifhr_neighbors must be > 0"; return false; }
    return true;
}

static size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

bool make_layout(const opa_shape& s, Layout* L, const char** why) {
    if (s.batch <= 0 || s.n_cif <= 0 || s.n_caf <= 0 || s.cif_h <= 0 || s.cif_w <= 0 || s.caf_h <= 0 ||
        s.caf_w <= 0 || s.cif_stride <= 0 || s.caf_stride <= 0 || s.max_annotations <= 0) {
        *why = "opa_shape: every field must be positive"; return false;
    }
    if (s.n_cif > 32767 || s.n_caf > 16383) { *why = "opa_shape: too many fields"; return false; }
    if ((long long)s.n_cif * s.cif_h * s.cif_w > (1ll << 30)) { *why = "opa_shape: CIF field too large"; return false; }
    L->B = s.batch; L->F = s.n_cif; L->A = s.n_caf; L->H = s.cif_h; L->W = s.cif_w;
    L->K = s.n_keypoints > 0 ? s.n_keypoints : s.n_cif;
    if (L->K < L->F || L->K > 32767) { *why = "opa_shape: n_keypoints must be 0 or in [n_cif, 32767]"; return false; }
    L->cH = s.caf_h; L->cW = s.caf_w; L->stride = s.cif_stride; L->cstride = s.caf_stride;
    L->max_ann = s.max_annotations;
    L->hr_rows = (s.cif_h - 1) * s.cif_stride + 1;                       // cif_hr.cpp:110-112
    L->hr_cols = (s.cif_w - 1) * s.cif_stride + 1;
    L->hr_pitch = (L->hr_cols + kHrTileW - 1) / kHrTileW * kHrTileW;
    // capacity for occupancy_reduction >= 1 (the reference hard-codes 2.0, cifcaf.hpp:103);
    // the per-call geometry (occupancy.cpp:47-48) is computed in opa_cifcaf_decode
    L->occ_h = L->hr_rows + 1;
    L->occ_w = L->hr_cols + 1;
    L->cif_cells = s.n_cif * s.cif_h * s.cif_w;
    L->caf_cells = s.caf_h * s.caf_w;
    int sc = 2; while (sc < L->cif_cells) sc <<= 1;
    if (sc < kSortLdsKeys) sc = kSortLdsKeys;
    L->sort_cap = sc;
    L->bbox_chunks = (L->caf_cells + kWave - 1) / kWave;
    if (L->bbox_chunks > kListBboxMax) L->bbox_chunks = kListBboxMax;
    const size_t B = s.batch;
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off = align_up(off + bytes); return o; };
    L->off_hdr = take(256);                                           // see kWsMagic (common.hpp)
    {   // two tile bitmaps per (image, field) plane: the previous call's and this call's touched tiles
        const size_t tpp = (size_t)(L->hr_pitch / kHrTileW) * ((L->hr_rows + kHrTileH - 1) / kHrTileH);
        L->off_tile_clean = take(2 * B * L->F * ((tpp + 31) / 32) * sizeof(unsigned));
    }
    {   // the map: a pool of 32x64 tiles per image (cifhr.hip) + the slot table of every plane
        const size_t tpp = (size_t)(L->hr_pitch / kHrTileW) * ((L->hr_rows + kHrTileH - 1) / kHrTileH);
        const size_t all = tpp * L->F;
        // automatic: the whole map where that is small (<= 32 MB per image: nothing can run out), else an eighth of it and at
        // least 1024 tiles, plus a spill region for the whole batch that holds what ONE image's pool cannot (a single
        // structureless image in a batch decodes; several of them are flagged and decoded again by the host paths)
        const size_t tile_bytes = (size_t)(kHrTileH * kHrTileW) * sizeof(float);
        size_t cap = s.cifhr_pool_tiles < 0 ? all : s.cifhr_pool_tiles > 0 ? (size_t)s.cifhr_pool_tiles
                   : all * tile_bytes <= (size_t)32000000 ? all : std::max<size_t>(1024, all / 8);   // (32 MB: a 641-px COCO map, 32.2 MB, is pooled)
        if (cap > all) cap = all;
        const size_t spill = s.cifhr_pool_tiles == 0 ? all - cap : 0;
        L->hr_tpp = (int)tpp; L->hr_pool_cap = (int)cap; L->hr_spill_cap = (int)spill;
        L->off_cifhr = take((B * cap + spill) * tile_bytes);
        L->off_hr_slot = take(B * all * sizeof(int32_t));
        L->off_hr_overflow = take((2 * B + 2) * sizeof(int32_t));     // overflow flags [B], spill counter, work counter, slot counters [B]
        L->off_hr_work = take(B * all * sizeof(int2));                // the tile kernel's work list
        L->cand_chunks = (L->H * L->W + 256 * kFillCells - 1) / (256 * kFillCells);
        L->off_cand_start = take(B * L->F * (size_t)(L->cand_chunks + 1) * sizeof(int32_t));   // seed candidates: chunk starts, then counts
    }
    L->off_act = take(B * L->F * 4 * (size_t)(L->H * L->W) * sizeof(float));
    L->off_act_count = take(B * L->F * sizeof(int32_t));
    L->off_seed_keys = take(B * (size_t)L->sort_cap * sizeof(unsigned long long));
    L->off_seed_count = take(B * sizeof(int32_t));
    L->off_seed_f = take(B * (size_t)L->cif_cells * sizeof(int32_t));
    L->off_seed_vxys = take(B * (size_t)L->cif_cells * 4 * sizeof(float));
    L->off_seed_cell = take(B * (size_t)L->cif_cells * sizeof(int32_t));
    const size_t list_bytes = B * L->A * 2 * 7 * (size_t)L->caf_cells * sizeof(float);
    L->off_lists = take(list_bytes);
    L->off_list_counts = take(B * L->A * 2 * sizeof(int32_t));
    L->off_list_bbox = take(B * L->A * 2 * (size_t)L->bbox_chunks * 4 * sizeof(float));
    // occupancy bitmap, one bit per cell (occupancy.cpp:46-68 keeps an int16 map); 256-B multiple per image
    L->occ_image_words = align_up((size_t)L->F * L->occ_h * ((L->occ_w + 31) / 32) * sizeof(unsigned)) / sizeof(unsigned);
    L->off_occ = take(B * L->occ_image_words * sizeof(unsigned));
    L->off_anns = take(B * (size_t)L->max_ann * L->K * 4 * sizeof(double));
    L->off_ann_meta = take(B * (size_t)L->max_ann * sizeof(int64_t));
    L->off_status = take(B * sizeof(int32_t));
    L->off_stats = take(B * 24 * sizeof(int32_t));
    L->off_trace = take(B * 64 * 4 * sizeof(int32_t));
    L->off_assoc_queue = take((B + 1) * sizeof(int32_t));             // the association kernel's image queue (batches beyond one workgroup per compute unit)
    L->off_tie_state = take(B * sizeof(int32_t));
    if (L->occ_image_words * sizeof(unsigned) >= tie_small_bytes(L->F, L->H * L->W)) {
        L->off_tie_small = L->off_occ; L->tie_small_stride = L->occ_image_words * sizeof(unsigned);
    } else {
        L->tie_small_stride = tie_small_bytes(L->F, L->H * L->W);
        L->off_tie_small = take(B * L->tie_small_stride);
    }
    L->total_no_fc = off;
    // what only a force-complete decode touches sits behind everything else: a workspace for decodes without it can stop here
    L->off_lists_fc = take(list_bytes);
    L->off_list_counts_fc = take(B * L->A * 2 * sizeof(int32_t));
    L->off_list_bbox_fc = take(B * L->A * 2 * (size_t)L->bbox_chunks * 4 * sizeof(float));
    L->off_fc_meta = take(B * 4 * sizeof(int32_t));
    L->total = off;
    return true;
}

}  // namespace opa

using namespace opa;

struct opa_cifcaf {
    int32_t K, A;
    std::vector<int64_t> skeleton;     // host copy [A,2]
    void* dev_block;                   // one allocation holding everything below
    DevSkeleton dev;
    int device;
    int tie_inside;                    // opa_cifcaf_set_tie_placement: -1 process-wide choice, 0 own launch, 1 inside the association kernel
    opa_debug debug;                   // opa_cifcaf_set_debug
    // Side streams (round 6): CafScored::fill only needs the finished map, the seed chain (fill, sort, rank merge, tie pass) only
    // needs the finished map -- two branches that meet at the association kernel.  The list building runs on a stream of the
    // library's own beside the seed chain: one per caller stream, created at the first decode on that stream, joined back before
    // the association kernel is queued (event fork / join: capturable into a HIP graph like everything else).
    struct Side { hipStream_t main, side; hipEvent_t fork, join; };
    std::vector<Side> sides;
    std::mutex sides_mutex;
};

static bool side_for(opa_cifcaf* dec, hipStream_t main, opa_cifcaf::Side* out) {
    std::lock_guard<std::mutex> lock(dec->sides_mutex);
    for (const auto& s : dec->sides)
        if (s.main == main) { *out = s; return true; }
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(main, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) {
        (void)hipGetLastError();
        return false;                                 // (no stream is created in the middle of a capture: this decode runs on one stream)
    }
    opa_cifcaf::Side s; s.main = main;
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    // (mode 1 -- the lists on the side stream -- wants the LOWEST priority there, mode 2 -- the tie pass -- the highest: its fat
    // workgroups only find room beside the list building when the dispatcher prefers them)
    if (hipStreamCreateWithPriority(&s.side, hipStreamNonBlocking, dec->debug.side_stream == 2 ? hi : lo) != hipSuccess) { (void)hipGetLastError(); return false; }
    if (hipEventCreateWithFlags(&s.fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&s.join, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); return false; }
    dec->sides.push_back(s);
    *out = s;
    return true;
}

extern "C" {

const char* opa_version(void) { return "openpifpaf_amd 0.1 (gfx950)"; }
int opa_abi_version(void) { return OPA_ABI_VERSION; }
size_t opa_shape_bytes(void) { return sizeof(opa_shape); }
size_t opa_params_bytes(void) { return sizeof(opa_params); }
size_t opa_debug_bytes(void) { return sizeof(opa_debug); }
void opa_default_debug(opa_debug* out) { if (out) *out = g_debug_default; }
const char* opa_last_error(void) { return g_error.c_str(); }

int opa_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

void opa_set_quiet(int quiet) { g_quiet = quiet; }

void opa_set_seed_tie_order(int order) { g_seed_tie_order = order == 2 ? 2 : order ? 1 : 0; }
int opa_get_seed_tie_order(void) { return opa::seed_tie_order(); }

void opa_default_params(opa_params* out) { if (out) *out = default_params(); }
void opa_get_params(opa_params* out) {
    std::lock_guard<std::mutex> lock(g_params_mutex);
    if (out) *out = g_params;
}
int opa_set_params(const opa_params* in) {
    if (!in) return fail(OPA_ERR_INVALID_ARGUMENT, "opa_set_params: null");
    const char* why = nullptr;
    if (!check_params(*in, &why)) return fail(OPA_ERR_INVALID_ARGUMENT, why);
    std::lock_guard<std::mutex> lock(g_params_mutex);
    g_params = *in;
    return OPA_OK;
}

int opa_cifcaf_create(opa_cifcaf** out, int32_t n_keypoints, const int64_t* skeleton_host, int32_t n_bones) {
    if (!out || !skeleton_host || n_keypoints <= 0 || n_bones <= 0 || n_keypoints > 32767 || n_bones > 16383)
        return fail(OPA_ERR_INVALID_ARGUMENT, "opa_cifcaf_create: bad arguments");
    for (int a = 0; a < 2 * n_bones; a++)
        if (skeleton_host[a] < 0 || skeleton_host[a] >= n_keypoints)
            return fail(OPA_ERR_INVALID_ARGUMENT, "opa_cifcaf_create: skeleton index out of range (must be 0-based)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(OPA_ERR_NO_DEVICE, "opa_cifcaf_create: no HIP device visible (this library has no CPU path)");
    const int K = n_keypoints, A = n_bones, E = 2 * A;
    // adjacency in bone order, exactly the visiting order of cifcaf.cpp:323-345
    std::vector<int32_t> off(K + 1, 0), other(E), bone(E), fwd(E), first(E);
    std::vector<std::vector<int32_t>> slots(K);
    std::vector<int32_t> s_other, s_bone, s_fwd;
    for (int j = 0; j < K; j++) {
        off[j] = (int32_t)s_other.size();
        for (int a = 0; a < A; a++) {
            const int64_t p0 = skeleton_host[2 * a], p1 = skeleton_host[2 * a + 1];
            if (p0 == j) { s_other.push_back((int32_t)p1); s_bone.push_back(a); s_fwd.push_back(1); }
            else if (p1 == j) { s_other.push_back((int32_t)p0); s_bone.push_back(a); s_fwd.push_back(0); }
        }
    }
    off[K] = (int32_t)s_other.size();
    s_other.resize(E, 0); s_bone.resize(E, 0); s_fwd.resize(E, 0);
    for (int j = 0; j < K; j++)
        for (int t = off[j]; t < off[j + 1]; t++) {
            int f = t;
            for (int u = off[j]; u < t; u++) if (s_other[u] == s_other[t]) { f = u; break; }
            first[t] = f;
        }
    for (int t = off[K]; t < E; t++) first[t] = t;

    const size_t skel_bytes = align_up(sizeof(int64_t) * 2 * A, 256);
    const size_t off_bytes = align_up(sizeof(int32_t) * (K + 1), 256);
    const size_t e_bytes = align_up(sizeof(int32_t) * E, 256);
    const size_t total = skel_bytes + off_bytes + 4 * e_bytes;
    void* block = nullptr;
    hipError_t e = hipMalloc(&block, total);
    if (e != hipSuccess) return fail_hip(e, "opa_cifcaf_create: hipMalloc");
    unsigned char* base = (unsigned char*)block;
    std::vector<unsigned char> host(total, 0);
    std::memcpy(host.data(), skeleton_host, sizeof(int64_t) * 2 * A);
    std::memcpy(host.data() + skel_bytes, off.data(), sizeof(int32_t) * (K + 1));
    std::memcpy(host.data() + skel_bytes + off_bytes + 0 * e_bytes, s_other.data(), sizeof(int32_t) * E);
    std::memcpy(host.data() + skel_bytes + off_bytes + 1 * e_bytes, s_bone.data(), sizeof(int32_t) * E);
    std::memcpy(host.data() + skel_bytes + off_bytes + 2 * e_bytes, s_fwd.data(), sizeof(int32_t) * E);
    std::memcpy(host.data() + skel_bytes + off_bytes + 3 * e_bytes, first.data(), sizeof(int32_t) * E);
    e = hipMemcpy(block, host.data(), total, hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(block); return fail_hip(e, "opa_cifcaf_create: hipMemcpy"); }

    opa_cifcaf* d = new opa_cifcaf();
    d->K = K; d->A = A;
    d->skeleton.assign(skeleton_host, skeleton_host + 2 * A);
    d->dev_block = block;
    d->dev.K = K; d->dev.A = A;
    d->dev.skeleton = (const int64_t*)base;
    d->dev.adj_off = (const int32_t*)(base + skel_bytes);
    d->dev.adj_other = (const int32_t*)(base + skel_bytes + off_bytes + 0 * e_bytes);
    d->dev.adj_bone = (const int32_t*)(base + skel_bytes + off_bytes + 1 * e_bytes);
    d->dev.adj_fwd = (const int32_t*)(base + skel_bytes + off_bytes + 2 * e_bytes);
    d->dev.adj_first = (const int32_t*)(base + skel_bytes + off_bytes + 3 * e_bytes);
    (void)hipGetDevice(&d->device);
    d->tie_inside = -1;
    d->debug = g_debug_default;
    *out = d;
    return OPA_OK;
}

void opa_cifcaf_destroy(opa_cifcaf* dec) {
    if (!dec) return;
    if (dec->dev_block) (void)hipFree(dec->dev_block);
    for (auto& s : dec->sides) { (void)hipStreamDestroy(s.side); (void)hipEventDestroy(s.fork); (void)hipEventDestroy(s.join); }
    delete dec;
}

int opa_cifcaf_set_tie_placement(opa_cifcaf* dec, int32_t inside_association) {
    if (!dec) return fail(OPA_ERR_INVALID_ARGUMENT, "opa_cifcaf_set_tie_placement: null handle");
    dec->tie_inside = inside_association < 0 ? -1 : inside_association ? 1 : 0;
    return OPA_OK;
}

int opa_cifcaf_set_debug(opa_cifcaf* dec, const opa_debug* in) {
    if (!dec || !in) return fail(OPA_ERR_INVALID_ARGUMENT, "opa_cifcaf_set_debug: null argument");
    if (in->assoc_watchdog_ticks <= 0 || in->assoc_growers < 0 || in->fc_split < 0 || in->fc_split > 64)
        return fail(OPA_ERR_INVALID_ARGUMENT, "opa_cifcaf_set_debug: value out of range");
    dec->debug = *in;
    return OPA_OK;
}
int opa_cifcaf_get_debug(const opa_cifcaf* dec, opa_debug* out) {
    if (!dec || !out) return fail(OPA_ERR_INVALID_ARGUMENT, "opa_cifcaf_get_debug: null argument");
    *out = dec->debug;
    return OPA_OK;
}

int opa_cifcaf_get_state(const opa_cifcaf* dec, int32_t* n_keypoints, int64_t* skeleton_host, int32_t* n_bones) {
    if (!dec) return fail(OPA_ERR_INVALID_ARGUMENT, "opa_cifcaf_get_state: null handle");
    if (n_keypoints) *n_keypoints = dec->K;
    if (n_bones) *n_bones = dec->A;
    if (skeleton_host) std::memcpy(skeleton_host, dec->skeleton.data(), sizeof(int64_t) * 2 * dec->A);
    return OPA_OK;
}

size_t opa_cifcaf_workspace_bytes(const opa_shape* shape) {
    Layout L; const char* why = nullptr;
    if (!shape || !make_layout(*shape, &L, &why)) { g_error = why ? why : "null shape"; return 0; }
    return L.total;
}

size_t opa_cifcaf_workspace_bytes_for(const opa_shape* shape, const opa_params* params) {
    Layout L; const char* why = nullptr;
    if (!shape || !make_layout(*shape, &L, &why)) { g_error = why ? why : "null shape"; return 0; }
    opa_params hp;
    if (params) hp = *params; else opa_get_params(&hp);
    return hp.force_complete ? L.total : L.total_no_fc;
}

int opa_cifcaf_cifhr_view(const opa_shape* shape, size_t* offset_floats, int32_t* rows, int32_t* cols,
                          int32_t* pitch, double* revision) {
    Layout L; const char* why = nullptr;
    if (!shape || !make_layout(*shape, &L, &why)) return fail(OPA_ERR_INVALID_ARGUMENT, why ? why : "null shape");
    if (offset_floats) *offset_floats = SIZE_MAX;            // (not a dense array in the workspace: opa_cifcaf_get_cifhr)
    if (rows) *rows = L.hr_rows;
    if (cols) *cols = L.hr_cols;
    if (pitch) *pitch = L.hr_cols;
    if (revision) *revision = 1.0;
    return OPA_OK;
}

int opa_cifcaf_get_cifhr(const opa_shape* shape, const void* workspace_dev, int32_t image, float* out_dev, void* stream) {
    Layout L; const char* why = nullptr;
    if (!shape || !make_layout(*shape, &L, &why)) return fail(OPA_ERR_INVALID_ARGUMENT, why ? why : "null shape");
    if (!workspace_dev || !out_dev || image < 0 || image >= L.B) return fail(OPA_ERR_INVALID_ARGUMENT, "opa_cifcaf_get_cifhr: bad argument");
    const unsigned char* ws = (const unsigned char*)workspace_dev;
    const float* pool_image = (const float*)(ws + L.off_cifhr) + (size_t)image * L.hr_pool_cap * (kHrTileH * kHrTileW);
    const int32_t* slot_image = (const int32_t*)(ws + L.off_hr_slot) + (size_t)image * L.F * L.hr_tpp;
    hipError_t e = launch_cifhr_gather(pool_image, slot_image, L.F, L.hr_rows, L.hr_cols, L.hr_pitch / kHrTileW, L.hr_tpp, out_dev,
                                       (hipStream_t)stream);
    if (e != hipSuccess) return fail_hip(e, "cifhr gather");
    return OPA_OK;
}

int opa_cifcaf_workspace_view(const opa_shape* shape, const char* what, size_t* offset_bytes, size_t* size_bytes) {
    Layout L; const char* why = nullptr;
    if (!shape || !what || !make_layout(*shape, &L, &why)) return fail(OPA_ERR_INVALID_ARGUMENT, why ? why : "null argument");
    struct Entry { const char* name; size_t off, end; };
    const Entry table[] = {
        {"tile_bitmaps", L.off_tile_clean, L.off_cifhr}, {"cifhr", L.off_cifhr, L.off_hr_slot}, {"cifhr_slots", L.off_hr_slot, L.off_hr_overflow},
        {"cifhr_overflow", L.off_hr_overflow, L.off_hr_overflow + (size_t)L.B * sizeof(int32_t)}, {"cifhr_work", L.off_hr_work, L.off_cand_start}, {"seed_count", L.off_seed_count, L.off_seed_f},
        {"seed_f", L.off_seed_f, L.off_seed_vxys}, {"seed_vxys", L.off_seed_vxys, L.off_seed_cell}, {"seed_cell", L.off_seed_cell, L.off_lists},
        {"lists", L.off_lists, L.off_list_counts}, {"list_counts", L.off_list_counts, L.off_list_bbox},
        {"list_bbox", L.off_list_bbox, L.off_occ},
        {"occupancy", L.off_occ, L.off_anns}, {"annotation_scratch", L.off_anns, L.off_ann_meta},
        {"status", L.off_status, L.off_stats}, {"assoc_stats", L.off_stats, L.off_trace}, {"assoc_trace", L.off_trace, L.off_assoc_queue}, {"assoc_queue", L.off_assoc_queue, L.off_tie_state}, {"seed_ties", L.off_tie_state, L.off_tie_state + (size_t)L.B * sizeof(int32_t)},
        {"lists_fc", L.off_lists_fc, L.off_list_counts_fc}, {"list_counts_fc", L.off_list_counts_fc, L.off_list_bbox_fc},
        {"list_bbox_fc", L.off_list_bbox_fc, L.off_fc_meta},
    };
    for (const Entry& e : table)
        if (std::strcmp(e.name, what) == 0) {
            if (offset_bytes) *offset_bytes = e.off;
            if (size_bytes) *size_bytes = e.end - e.off;
            return OPA_OK;
        }
    return fail(OPA_ERR_INVALID_ARGUMENT, std::string("opa_cifcaf_workspace_view: unknown buffer ") + what);
}

int opa_cifcaf_decode(const opa_cifcaf* dec_in, const opa_shape* shape, const opa_params* params,
                      const float* cif_dev, const float* caf_dev,
                      const float* initial_dev, const int64_t* initial_ids_dev, int32_t n_initial,
                      void* workspace_dev, size_t workspace_bytes,
                      float* out_dev, int64_t* out_ids_dev, int32_t* out_count_dev, void* stream) {
    opa_cifcaf* dec = const_cast<opa_cifcaf*>(dec_in);   // (the handle's side streams are created on first use, under its mutex)
    if (!dec || !shape || !cif_dev || !caf_dev || !workspace_dev || !out_dev || !out_ids_dev || !out_count_dev)
        return fail(OPA_ERR_INVALID_ARGUMENT, "opa_cifcaf_decode: null argument");
    if (n_initial < 0 || (n_initial > 0 && !initial_dev))
        return fail(OPA_ERR_INVALID_ARGUMENT, "opa_cifcaf_decode: initial annotations without data");
    opa_params hp;
    if (params) hp = *params; else opa_get_params(&hp);
    const char* why = nullptr;
    if (!check_params(hp, &why)) return fail(OPA_ERR_INVALID_ARGUMENT, why);
    Layout L;
    if (!make_layout(*shape, &L, &why)) return fail(OPA_ERR_INVALID_ARGUMENT, why);
    if (L.K != dec->K || shape->n_caf != dec->A)
        return fail(OPA_ERR_INVALID_ARGUMENT, "opa_cifcaf_decode: shape does not match the decoder's keypoints/skeleton");
    if (hp.occupancy_reduction < 1.0)
        return fail(OPA_ERR_UNSUPPORTED, "opa_cifcaf_decode: occupancy_reduction < 1 is not supported");
    L.occ_h = (int)((double)L.hr_rows / hp.occupancy_reduction) + 1;       // occupancy.cpp:47-48
    L.occ_w = (int)((double)L.hr_cols / hp.occupancy_reduction) + 1;
    if (workspace_bytes < (hp.force_complete ? L.total : L.total_no_fc))
        return fail(OPA_ERR_WORKSPACE, hp.force_complete && workspace_bytes >= L.total_no_fc
                    ? "opa_cifcaf_decode: workspace too small for a force-complete decode (sized with opa_cifcaf_workspace_bytes_for "
                      "without the flag?)" : "opa_cifcaf_decode: workspace too small");
    if (((uintptr_t)workspace_dev & 255) != 0 || ((uintptr_t)out_dev & 15) != 0)
        return fail(OPA_ERR_INVALID_ARGUMENT, "opa_cifcaf_decode: workspace must be 256-B and out 16-B aligned");

    hipStream_t st = (hipStream_t)stream;
    unsigned char* ws = (unsigned char*)workspace_dev;
    const DevParams p = to_dev(hp);
    float* cifhr = (float*)(ws + L.off_cifhr);
    hipError_t e;

    // the clean-tile flags are valid for exactly this carving of the workspace
    unsigned long long layout_hash = 1469598103934665603ull;
    for (long long v : {(long long)L.B, (long long)L.F, (long long)L.H, (long long)L.W, (long long)L.stride,
                        (long long)L.A, (long long)L.cH, (long long)L.cW, (long long)L.max_ann, (long long)L.K, (long long)L.total_no_fc})
        layout_hash = (layout_hash ^ (unsigned long long)v) * 1099511628211ull;
    HrPool pool;                                      // the map is a pool of tiles (cifhr.hip)
    pool.slot = (int32_t*)(ws + L.off_hr_slot); pool.overflow = (int32_t*)(ws + L.off_hr_overflow); pool.cap = L.hr_pool_cap; pool.tpp = L.hr_tpp;
    pool.spill_cap = L.hr_spill_cap; pool.images = L.B; pool.spill_count = pool.overflow + L.B;
    pool.work_count = pool.overflow + L.B + 1; pool.img_tiles = pool.overflow + L.B + 2; pool.work = (int2*)(ws + L.off_hr_work);
    // the seed candidates lie where the sorted seeds will: the fill kernel is done with them before the sort writes those
    SeedCandidates cand;
    cand.cand = (float4*)(ws + L.off_seed_vxys); cand.start = (int32_t*)(ws + L.off_cand_start);
    cand.count = cand.start + (size_t)L.B * L.F * L.cand_chunks; cand.chunks = L.cand_chunks; cand.produced = false;
    e = launch_cifhr(cif_dev, L.B, L.F, L.H, L.W, L.stride, 0.0, 1.0, p, cifhr, L.hr_rows, L.hr_pitch,
                     (float*)(ws + L.off_act), (int32_t*)(ws + L.off_act_count), st, false,
                     (unsigned long long*)(ws + L.off_hdr), layout_hash, ws + L.off_tile_clean,
                     (int32_t*)(ws + L.off_seed_count), &pool, dec->debug.stage_worklist ? &cand : nullptr);   // cifcaf.cpp:140-141
    if (e != hipSuccess) return fail_hip(e, "cifhr");
    // CafScored::fill (:153-161) of the caf_th list set and, for force complete, of the second one (:419-420): launches of
    // their own.  (OPA_FUSE_SCORED=1 lets them ride in the seed sort's launch, two 512-thread groups per workgroup beside
    // the sort's workgroups -- measured in round 3: 90 us against 42 + 41 us one after the other, 323 against 170 us with
    // the force-complete set: under the sort kernel's 64 KiB of static LDS and 1024-thread workgroups the list building
    // gets two workgroups per compute unit instead of its six, and loses more than the overlap gives.)
    ScoredArgs scored[2];
    int n_scored = 0;
    scored[n_scored++] = make_scored_args(caf_dev, L.B, L.A, L.cH, L.cW, L.cstride, cifhr, L.F, L.hr_rows, L.hr_cols, L.hr_pitch,
                                          dec->dev.skeleton, p.caf_threshold, p.cif_floor, p.ablation_caf_no_rescore,
                                          (float*)(ws + L.off_lists), (int32_t*)(ws + L.off_list_counts),
                                          (float*)(ws + L.off_list_bbox),
                                          L.bbox_chunks < kListBboxChunks ? L.bbox_chunks : kListBboxChunks, L.bbox_chunks,
                                          nullptr, &pool);   // (pooled map: the slot table says which tiles exist)
    if (p.force_complete)
        scored[n_scored++] = make_scored_args(caf_dev, L.B, L.A, L.cH, L.cW, L.cstride, cifhr, L.F, L.hr_rows, L.hr_cols,
                                              L.hr_pitch, dec->dev.skeleton, p.force_complete_caf_th, 0.1,
                                              p.ablation_caf_no_rescore, (float*)(ws + L.off_lists_fc),
                                              (int32_t*)(ws + L.off_list_counts_fc), (float*)(ws + L.off_list_bbox_fc),
                                              L.bbox_chunks, L.bbox_chunks, nullptr, &pool);
    const bool fuse = dec->debug.fuse_scored != 0;
    TieScratch ties;
    ties.big = ws + L.off_act; ties.big_stride = (size_t)L.F * 4 * (L.H * L.W) * sizeof(float);
    ties.small_ = ws + L.off_tie_small; ties.small_stride = L.tie_small_stride;
    ties.state = (int32_t*)(ws + L.off_tie_state);
    // Where the tie pass runs: in the association kernel (every image its own ties, before its seeds are read) or as a launch of its
    // own.  Round 4 measured ONE decode 2 % shorter with the pass inside -- the images with the most seeds are both the likeliest
    // to hold equal scores and the slowest to associate -- and twelve lanes 11 % faster; round 6 measured it again on the new stage
    // kernels (wall per decode: 32 COCO images 0.730 -> 0.713 ms, 256 images 1.563 -> 1.502 ms, 16 wholebody images 3.42 -> 3.36 ms).
    // (round 6: inside the association kernel unless the decoder asks for the launch of its own -- measured shorter for one decode
    // of 32 or 256 COCO images and of 16 wholebody images alike, profiles/r6/tie_placement.log)
    // The branches behind the finished map -- the CAF lists, the seed chain (fill, sort, rank merge), the tie pass -- meet at the
    // association kernel.  opa_debug::side_stream puts one of them on the handle's side stream (not while this thread profiles the
    // stream with events -- their times are per kernel in a row -- and not when the side stream cannot be had, e.g. the first
    // decode of a handle inside a graph capture): 1 = the lists beside the whole seed chain (measured: no overlap, the list
    // building fills the chip and the sort's fat workgroups find no room beside it); 2 = the TIE PASS, a launch of its own, beside
    // the list building -- one workgroup per image that needs most of a compute unit's LDS but few of its wave slots.
    opa_cifcaf::Side side;
    const bool forked = !fuse && dec->debug.side_stream != 0 && !(g_prof.on && g_prof.st == st) && side_for(dec, st, &side);
    const bool tie_on_side = forked && dec->debug.side_stream == 2 && seed_tie_order() >= 1 && dec->tie_inside != 1;
    const bool lists_on_side = forked && dec->debug.side_stream == 1;
    const bool fuse_ties = seed_tie_order() >= 1 && !tie_on_side && (dec->tie_inside >= 0 ? dec->tie_inside == 1 : true);
    ties.defer = fuse_ties || tie_on_side ? 1 : 0;
    auto lists = [&](hipStream_t ls) -> int {
        if (n_scored == 2 && dec->debug.scored_one_pass) {     // both list sets from one read of the field (round 6)
            const hipError_t le = launch_cafscored2(scored[0], scored[1], ls);
            if (le != hipSuccess) return fail_hip(le, "cafscored(both list sets)");
        } else
            for (int k = 0; k < n_scored; k++) {
                const hipError_t le = launch_cafscored(scored[k], ls);
                if (le != hipSuccess) return fail_hip(le, k ? "cafscored(force complete)" : "cafscored");
            }
        return OPA_OK;
    };
    if (lists_on_side) {
        e = hipEventRecord(side.fork, st);
        if (e == hipSuccess) e = hipStreamWaitEvent(side.side, side.fork, 0);
        if (e != hipSuccess) return fail_hip(e, "fork to the side stream");
        const int rc = lists(side.side);
        if (rc != OPA_OK) return rc;
        e = hipEventRecord(side.join, side.side);
        if (e != hipSuccess) return fail_hip(e, "side stream");
    }
    e = launch_cifseeds(cif_dev, L.B, L.F, L.H, L.W, L.stride, cifhr, L.hr_rows, L.hr_cols, L.hr_pitch, p,
                        (unsigned long long*)(ws + L.off_seed_keys), L.sort_cap,
                        (int32_t*)(ws + L.off_seed_count), (int32_t*)(ws + L.off_seed_f),
                        (float*)(ws + L.off_seed_vxys), st, false, (int32_t*)(ws + L.off_seed_cell),
                        L.occ_h, L.occ_w, true, fuse ? scored : nullptr, fuse ? n_scored : 0, &ties, &pool, &cand,
                        dec->debug.stage_worklist != 0);                                              // :144-146
    if (e != hipSuccess) return fail_hip(e, "cifseeds");
    if (lists_on_side) {
        e = hipStreamWaitEvent(st, side.join, 0);
        if (e != hipSuccess) return fail_hip(e, "join of the side stream");
    } else if (!fuse) {
        if (tie_on_side) {                            // the seeds are sorted: the tie pass beside the list building
            TieArgs ta; SortArgs tg;
            make_tie_args(&ta, &tg, (unsigned long long*)(ws + L.off_seed_keys), L.sort_cap, (int32_t*)(ws + L.off_seed_count), cif_dev,
                          L.F, 5, L.H * L.W, L.stride, (int32_t*)(ws + L.off_seed_f), (float*)(ws + L.off_seed_vxys),
                          (int32_t*)(ws + L.off_seed_cell), L.occ_h, L.occ_w, ties);
            e = hipEventRecord(side.fork, st);
            if (e == hipSuccess) e = hipStreamWaitEvent(side.side, side.fork, 0);
            if (e == hipSuccess) e = launch_cifseeds_ties(ta, tg, L.B, p, side.side);
            if (e == hipSuccess) e = hipEventRecord(side.join, side.side);
            if (e != hipSuccess) return fail_hip(e, "tie pass on the side stream");
        }
        const int rc = lists(st);
        if (rc != OPA_OK) return rc;
        if (tie_on_side) {
            e = hipStreamWaitEvent(st, side.join, 0);
            if (e != hipSuccess) return fail_hip(e, "join of the side stream");
        }
    }
    // (the occupancy map of :173 is a bitmap the association kernel clears itself)
    AssocArgs a;
    a.B = L.B; a.K = L.K; a.F = L.F; a.A = L.A; a.max_ann = L.max_ann; a.n_initial = n_initial;
    a.hr_rows = L.hr_rows; a.hr_cols = L.hr_cols; a.occ_h = L.occ_h; a.occ_w = L.occ_w;
    a.seed_cap = L.cif_cells; a.list_cap = L.caf_cells;
    a.seed_f = (const int32_t*)(ws + L.off_seed_f); a.seed_vxys = (const float*)(ws + L.off_seed_vxys);
    a.seed_count = (const int32_t*)(ws + L.off_seed_count);
    a.seed_cell = (const int32_t*)(ws + L.off_seed_cell);
    a.lists = (const float*)(ws + L.off_lists); a.list_counts = (const int32_t*)(ws + L.off_list_counts);
    a.lists_fc = (const float*)(ws + L.off_lists_fc); a.list_counts_fc = (const int32_t*)(ws + L.off_list_counts_fc);
    a.caf_raw = caf_dev; a.caf_w = L.cW; a.caf_stride = (float)L.cstride;
    a.list_bbox = (const float*)(ws + L.off_list_bbox);
    a.list_bbox_fc = (const float*)(ws + L.off_list_bbox_fc);
    a.bbox_chunks = L.bbox_chunks;
    a.fc_meta = (int32_t*)(ws + L.off_fc_meta);
    a.occ = (unsigned*)(ws + L.off_occ); a.occ_image_words = L.occ_image_words;
    a.stats = (int32_t*)(ws + L.off_stats);
    a.trace = (int32_t*)(ws + L.off_trace);
    a.anns = (double*)(ws + L.off_anns); a.ann_ids = (int64_t*)(ws + L.off_ann_meta);
    a.initial = initial_dev; a.initial_ids = initial_ids_dev;
    a.out = out_dev; a.out_ids = out_ids_dev; a.out_count = out_count_dev;
    a.status = (int32_t*)(ws + L.off_status);
    a.hr_overflow = pool.overflow;
    a.queue_order = (int32_t*)(ws + L.off_assoc_queue); a.queue_head = a.queue_order + L.B;
    a.tie_fused = fuse_ties ? 1 : 0;
    make_tie_args(&a.tie, &a.tie_sort, (unsigned long long*)(ws + L.off_seed_keys), L.sort_cap, (int32_t*)(ws + L.off_seed_count), cif_dev,
                  L.F, 5, L.H * L.W, L.stride, (int32_t*)(ws + L.off_seed_f), (float*)(ws + L.off_seed_vxys),
                  (int32_t*)(ws + L.off_seed_cell), L.occ_h, L.occ_w, ties);
    e = launch_assoc(a, dec->dev, p, st, dec->debug);                                         // :176-261
    if (e != hipSuccess) return fail_hip(e, "association");
    return OPA_OK;
}

// ---- stage-level entry points ----------------------------------------------
int32_t opa_cifhr_pitch(int32_t cif_w, int32_t stride) {
    const int cols = (cif_w - 1) * stride + 1;
    return (cols + kHrTileW - 1) / kHrTileW * kHrTileW;
}

size_t opa_cifhr_scratch_bytes(int32_t batch, int32_t n_cif, int32_t cif_h, int32_t cif_w) {
    return align_up((size_t)batch * n_cif * 4 * cif_h * cif_w * sizeof(float)) +
           align_up((size_t)batch * n_cif * sizeof(int32_t));
}

int opa_cifhr_accumulate(const float* cif_dev, int32_t batch, int32_t n_cif, int32_t cif_h, int32_t cif_w,
                         int32_t stride, double min_scale, double factor, const opa_params* params,
                         float* cifhr_dev, void* scratch_dev, size_t scratch_bytes, void* stream) {
    if (!cif_dev || !cifhr_dev || !scratch_dev || batch <= 0 || n_cif <= 0 || cif_h <= 0 || cif_w <= 0 || stride <= 0)
        return fail(OPA_ERR_INVALID_ARGUMENT, "opa_cifhr_accumulate: bad arguments");
    if (scratch_bytes < opa_cifhr_scratch_bytes(batch, n_cif, cif_h, cif_w))
        return fail(OPA_ERR_WORKSPACE, "opa_cifhr_accumulate: scratch too small");
    opa_params hp; if (params) hp = *params; else opa_get_params(&hp);
    const DevParams p = to_dev(hp);
    unsigned char* sc = (unsigned char*)scratch_dev;
    const size_t act_bytes = align_up((size_t)batch * n_cif * 4 * cif_h * cif_w * sizeof(float));
    hipError_t e = launch_cifhr(cif_dev, batch, n_cif, cif_h, cif_w, stride, min_scale, factor, p, cifhr_dev,
                                (cif_h - 1) * stride + 1, opa_cifhr_pitch(cif_w, stride),
                                (float*)sc, (int32_t*)(sc + act_bytes), (hipStream_t)stream);
    if (e != hipSuccess) return fail_hip(e, "cifhr");
    return OPA_OK;
}

static int sort_cap_for(int cells) { int sc = 2; while (sc < cells) sc <<= 1; return sc < kSortLdsKeys ? kSortLdsKeys : sc; }

// keys, then the scratch of the tie pass (position table + stop lists, bitmap + segment lists, one state word per image)
static size_t seeds_keys_bytes(int batch, int cells) { return align_up((size_t)batch * sort_cap_for(cells) * sizeof(unsigned long long)); }
size_t opa_cifseeds_scratch_bytes(int32_t batch, int32_t n_cif, int32_t cif_h, int32_t cif_w) {
    const int cells = n_cif * cif_h * cif_w;
    return seeds_keys_bytes(batch, cells) + align_up((size_t)batch * tie_big_bytes(cells)) +
           align_up((size_t)batch * tie_small_bytes(n_cif, cif_h * cif_w)) + align_up((size_t)batch * sizeof(int32_t));
}
static TieScratch stage_ties(void* scratch_dev, int batch, int F, int HW) {
    const int cells = F * HW;
    unsigned char* sp = (unsigned char*)scratch_dev + seeds_keys_bytes(batch, cells);
    TieScratch t;
    t.big = sp; t.big_stride = tie_big_bytes(cells); sp += align_up((size_t)batch * t.big_stride);
    t.small_ = sp; t.small_stride = tie_small_bytes(F, HW); sp += align_up((size_t)batch * t.small_stride);
    t.state = (int32_t*)sp;
    t.defer = 0;
    return t;
}

int opa_cifseeds_fill(const float* cif_dev, int32_t batch, int32_t n_cif, int32_t cif_h, int32_t cif_w,
                      int32_t stride, const float* cifhr_dev, const opa_params* params,
                      int32_t* seed_f_dev, float* seed_vxys_dev, int32_t* seed_count_dev,
                      void* scratch_dev, size_t scratch_bytes, void* stream) {
    if (!cif_dev || !cifhr_dev || !seed_f_dev || !seed_vxys_dev || !seed_count_dev || !scratch_dev ||
        batch <= 0 || n_cif <= 0 || cif_h <= 0 || cif_w <= 0 || stride <= 0)
        return fail(OPA_ERR_INVALID_ARGUMENT, "opa_cifseeds_fill: bad arguments");
    if (scratch_bytes < opa_cifseeds_scratch_bytes(batch, n_cif, cif_h, cif_w))
        return fail(OPA_ERR_WORKSPACE, "opa_cifseeds_fill: scratch too small");
    opa_params hp; if (params) hp = *params; else opa_get_params(&hp);
    const DevParams p = to_dev(hp);
    const TieScratch ties = stage_ties(scratch_dev, batch, n_cif, cif_h * cif_w);
    hipError_t e = launch_cifseeds(cif_dev, batch, n_cif, cif_h, cif_w, stride, cifhr_dev,
                                   (cif_h - 1) * stride + 1, (cif_w - 1) * stride + 1, opa_cifhr_pitch(cif_w, stride),
                                   p, (unsigned long long*)scratch_dev, sort_cap_for(n_cif * cif_h * cif_w),
                                   seed_count_dev, seed_f_dev, seed_vxys_dev, (hipStream_t)stream, false, nullptr, 0, 0, false,
                                   nullptr, 0, &ties);
    if (e != hipSuccess) return fail_hip(e, "cifseeds");
    return OPA_OK;
}

int opa_cifdetseeds_fill(const float* field_dev, int32_t batch, int32_t n_fields, int32_t field_h, int32_t field_w,
                         int32_t stride, const float* cifhr_dev, const opa_params* params,
                         int32_t* seed_f_dev, float* seed_vxywh_dev, int32_t* seed_count_dev,
                         void* scratch_dev, size_t scratch_bytes, void* stream) {
    if (!field_dev || !cifhr_dev || !seed_f_dev || !seed_vxywh_dev || !seed_count_dev || !scratch_dev ||
        batch <= 0 || n_fields <= 0 || field_h <= 0 || field_w <= 0 || stride <= 0)
        return fail(OPA_ERR_INVALID_ARGUMENT, "opa_cifdetseeds_fill: bad arguments");
    if (scratch_bytes < opa_cifseeds_scratch_bytes(batch, n_fields, field_h, field_w))
        return fail(OPA_ERR_WORKSPACE, "opa_cifdetseeds_fill: scratch too small");
    opa_params hp; if (params) hp = *params; else opa_get_params(&hp);
    const DevParams p = to_dev(hp);
    const TieScratch ties = stage_ties(scratch_dev, batch, n_fields, field_h * field_w);
    hipError_t e = launch_cifseeds(field_dev, batch, n_fields, field_h, field_w, stride, cifhr_dev,
                                   (field_h - 1) * stride + 1, (field_w - 1) * stride + 1,
                                   opa_cifhr_pitch(field_w, stride), p, (unsigned long long*)scratch_dev,
                                   sort_cap_for(n_fields * field_h * field_w), seed_count_dev, seed_f_dev,
                                   seed_vxywh_dev, (hipStream_t)stream, true, nullptr, 0, 0, false, nullptr, 0, &ties);
    if (e != hipSuccess) return fail_hip(e, "cifdetseeds");
    return OPA_OK;
}

int opa_cafscored_fill(const float* caf_dev, int32_t batch, int32_t n_caf, int32_t caf_h, int32_t caf_w,
                       int32_t stride, const float* cifhr_dev, int32_t n_cif, int32_t cif_h, int32_t cif_w,
                       int32_t cif_stride, const int64_t* skeleton_dev, double score_th, double cif_floor,
                       const opa_params* params, float* lists_dev, int32_t* counts_dev, void* stream) {
    if (!caf_dev || !cifhr_dev || !skeleton_dev || !lists_dev || !counts_dev || batch <= 0 || n_caf <= 0 ||
        caf_h <= 0 || caf_w <= 0 || stride <= 0 || n_cif <= 0 || cif_h <= 0 || cif_w <= 0 || cif_stride <= 0)
        return fail(OPA_ERR_INVALID_ARGUMENT, "opa_cafscored_fill: bad arguments");
    opa_params hp; if (params) hp = *params; else opa_get_params(&hp);
    if (score_th < 0.0) score_th = hp.caf_threshold;                       // caf_scored.hpp:52
    hipError_t e = launch_cafscored(caf_dev, batch, n_caf, caf_h, caf_w, stride, cifhr_dev, n_cif,
                                    (cif_h - 1) * cif_stride + 1, (cif_w - 1) * cif_stride + 1,
                                    opa_cifhr_pitch(cif_w, cif_stride), skeleton_dev, score_th, cif_floor,
                                    hp.ablation_caf_no_rescore, lists_dev, counts_dev, (hipStream_t)stream);
    if (e != hipSuccess) return fail_hip(e, "cafscored");
    return OPA_OK;
}

int opa_grow_connection_blend(const float* rows_dev, int32_t n, double x, double y, double s,
                              double filter_sigmas, int32_t only_max, double* out_host, void* stream) {
    if (!out_host || n < 0 || (n > 0 && !rows_dev))
        return fail(OPA_ERR_INVALID_ARGUMENT, "opa_grow_connection_blend: bad arguments");
    // a test/debug op: it is synchronous and owns a tiny temporary (the only allocating entry point)
    void* tmp = nullptr;
    hipError_t e = hipMalloc(&tmp, 4 * sizeof(double) + (size_t)(n > 0 ? n : 1) * 7 * sizeof(float));
    if (e != hipSuccess) return fail_hip(e, "opa_grow_connection_blend: hipMalloc");
    e = launch_blend(rows_dev, n, x, y, s, filter_sigmas, only_max, (double*)tmp, (hipStream_t)stream);
    if (e == hipSuccess) e = hipMemcpyAsync(out_host, tmp, 4 * sizeof(double), hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    (void)hipFree(tmp);
    if (e != hipSuccess) return fail_hip(e, "opa_grow_connection_blend");
    return OPA_OK;
}

struct DetLayout {
    int hr_rows, hr_cols, hr_pitch, occ_h, occ_w, cells, sort_cap;
    size_t off_cifhr, off_act, off_act_count, off_keys, off_seed_count, off_seed_f, off_seed_v, off_occ, total;
    size_t off_tie_small, tie_small_stride, off_tie_state;    // (see Layout)
};

static bool make_det_layout(const opa_det_shape& s, DetLayout* L, const char** why) {
    if (s.batch <= 0 || s.n_fields <= 0 || s.field_h <= 0 || s.field_w <= 0 || s.stride <= 0 || s.max_detections <= 0) {
        *why = "opa_det_shape: every field must be positive"; return false;
    }
    if ((long long)s.n_fields * s.field_h * s.field_w > (1ll << 30)) { *why = "opa_det_shape: field too large"; return false; }
    L->hr_rows = (s.field_h - 1) * s.stride + 1;
    L->hr_cols = (s.field_w - 1) * s.stride + 1;
    L->hr_pitch = (L->hr_cols + kHrTileW - 1) / kHrTileW * kHrTileW;
    L->occ_h = L->hr_rows + 1; L->occ_w = L->hr_cols + 1;
    L->cells = s.n_fields * s.field_h * s.field_w;
    int sc = 2; while (sc < L->cells) sc <<= 1;
    L->sort_cap = sc < kSortLdsKeys ? kSortLdsKeys : sc;
    const size_t B = s.batch;
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off = align_up(off + bytes); return o; };
    L->off_cifhr = take(B * s.n_fields * L->hr_rows * (size_t)L->hr_pitch * sizeof(float));
    L->off_act = take(B * s.n_fields * 4 * (size_t)(s.field_h * s.field_w) * sizeof(float));
    L->off_act_count = take(B * s.n_fields * sizeof(int32_t));
    L->off_keys = take(B * (size_t)L->sort_cap * sizeof(unsigned long long));
    L->off_seed_count = take(B * sizeof(int32_t));
    L->off_seed_f = take(B * (size_t)L->cells * sizeof(int32_t));
    L->off_seed_v = take(B * (size_t)L->cells * 5 * sizeof(float));
    L->off_occ = take(B * s.n_fields * (size_t)L->occ_h * L->occ_w);
    L->off_tie_state = take(B * sizeof(int32_t));
    if (s.n_fields * (size_t)L->occ_h * L->occ_w >= tie_small_bytes(s.n_fields, s.field_h * s.field_w) && (s.n_fields * (size_t)L->occ_h * L->occ_w) % 16 == 0) {
        L->off_tie_small = L->off_occ; L->tie_small_stride = s.n_fields * (size_t)L->occ_h * L->occ_w;
    } else {
        L->tie_small_stride = tie_small_bytes(s.n_fields, s.field_h * s.field_w);
        L->off_tie_small = take(B * L->tie_small_stride);
    }
    L->total = off;
    return true;
}

size_t opa_cifdet_workspace_bytes(const opa_det_shape* shape) {
    DetLayout L; const char* why = nullptr;
    if (!shape || !make_det_layout(*shape, &L, &why)) { g_error = why ? why : "null shape"; return 0; }
    return L.total;
}

int opa_cifdet_decode(const opa_det_shape* shape, const opa_params* params, const float* field_dev,
                      void* workspace_dev, size_t workspace_bytes,
                      int64_t* categories_dev, float* scores_dev, float* boxes_dev, int32_t* counts_dev,
                      void* stream) {
    if (!shape || !field_dev || !workspace_dev || !categories_dev || !scores_dev || !boxes_dev || !counts_dev)
        return fail(OPA_ERR_INVALID_ARGUMENT, "opa_cifdet_decode: null argument");
    opa_params hp;
    if (params) hp = *params; else opa_get_params(&hp);
    const char* why = nullptr;
    if (!check_params(hp, &why)) return fail(OPA_ERR_INVALID_ARGUMENT, why);
    if (hp.occupancy_reduction < 1.0) return fail(OPA_ERR_UNSUPPORTED, "opa_cifdet_decode: occupancy_reduction < 1");
    DetLayout L;
    if (!make_det_layout(*shape, &L, &why)) return fail(OPA_ERR_INVALID_ARGUMENT, why);
    if (workspace_bytes < L.total) return fail(OPA_ERR_WORKSPACE, "opa_cifdet_decode: workspace too small");
    if (((uintptr_t)workspace_dev & 255) != 0) return fail(OPA_ERR_INVALID_ARGUMENT, "opa_cifdet_decode: workspace must be 256-B aligned");
    hipStream_t st = (hipStream_t)stream;
    unsigned char* ws = (unsigned char*)workspace_dev;
    const DevParams p = to_dev(hp);
    const int B = shape->batch, F = shape->n_fields, H = shape->field_h, W = shape->field_w;
    float* cifhr = (float*)(ws + L.off_cifhr);
    hipError_t e = launch_cifhr(field_dev, B, F, H, W, shape->stride, 0.0, 1.0, p, cifhr, L.hr_rows, L.hr_pitch,
                                (float*)(ws + L.off_act), (int32_t*)(ws + L.off_act_count), st, true);   // cifdet.cpp:30-31
    if (e != hipSuccess) return fail_hip(e, "cifdethr");
    TieScratch ties;
    ties.big = ws + L.off_act; ties.big_stride = (size_t)F * 4 * (H * W) * sizeof(float);
    ties.small_ = ws + L.off_tie_small; ties.small_stride = L.tie_small_stride;
    ties.state = (int32_t*)(ws + L.off_tie_state);
    ties.defer = 0;
    e = launch_cifseeds(field_dev, B, F, H, W, shape->stride, cifhr, L.hr_rows, L.hr_cols, L.hr_pitch, p,
                        (unsigned long long*)(ws + L.off_keys), L.sort_cap, (int32_t*)(ws + L.off_seed_count),
                        (int32_t*)(ws + L.off_seed_f), (float*)(ws + L.off_seed_v), st, true, nullptr, 0, 0, false, nullptr, 0,
                        &ties);                                                                          // :34-36
    if (e != hipSuccess) return fail_hip(e, "cifdetseeds");
    const int occ_h = (int)((double)L.hr_rows / hp.occupancy_reduction) + 1;
    const int occ_w = (int)((double)L.hr_cols / hp.occupancy_reduction) + 1;
    e = launch_zero(ws + L.off_occ, ((size_t)B * F * occ_h * occ_w + 3) & ~(size_t)3, st);                // :44
    if (e != hipSuccess) return fail_hip(e, "occupancy memset");
    prof_mark(st, "memset_occupancy");
    DetArgs a;
    a.B = B; a.F = F; a.max_det = shape->max_detections; a.occ_h = occ_h; a.occ_w = occ_w; a.seed_cap = L.cells;
    a.seed_f = (const int32_t*)(ws + L.off_seed_f); a.seed_vxywh = (const float*)(ws + L.off_seed_v);
    a.seed_count = (const int32_t*)(ws + L.off_seed_count); a.occ = ws + L.off_occ;
    a.categories = categories_dev; a.scores = scores_dev; a.boxes = boxes_dev; a.counts = counts_dev;
    e = launch_cifdet_collect(a, p, st);                                                                // :50-67
    if (e != hipSuccess) return fail_hip(e, "cifdet collect");
    return OPA_OK;
}

int opa_bias_act(void* x_dev, const void* bias_dev, const void* residual_dev, int64_t rows, int32_t channels,
                 int32_t dtype, int32_t relu, void* stream) {
    if (!x_dev || !bias_dev || rows < 0 || channels <= 0 || dtype < 0 || dtype > 2)
        return fail(OPA_ERR_INVALID_ARGUMENT, "opa_bias_act: bad arguments");
    const int per_vec = dtype == 0 ? 4 : 8;
    if (channels % per_vec != 0 || ((uintptr_t)x_dev & 15) || ((uintptr_t)bias_dev & 15) || ((uintptr_t)residual_dev & 15))
        return fail(OPA_ERR_INVALID_ARGUMENT, "opa_bias_act: channels must fill 16-byte vectors and pointers be 16-B aligned");
    if (rows == 0) return OPA_OK;
    hipError_t e = launch_bias_act(x_dev, bias_dev, residual_dev, rows, channels, dtype, relu, (hipStream_t)stream);
    if (e != hipSuccess) return fail_hip(e, "bias_act");
    prof_mark((hipStream_t)stream, "bias_act_kernel");
    return OPA_OK;
}

int opa_gemm_bias_act_bf16(const void* a_dev, const void* w_dev, const void* bias_dev, const void* residual_dev,
                           void* out_dev, int64_t m, int32_t n, int32_t k, int32_t relu, void* stream) {
    if (!a_dev || !w_dev || !bias_dev || !out_dev || m < 0 || n <= 0 || k <= 0 || m > 0x7fffffffll)
        return fail(OPA_ERR_INVALID_ARGUMENT, "opa_gemm_bias_act_bf16: bad arguments");
    if (k % 64 != 0 || n % 64 != 0)
        return fail(OPA_ERR_INVALID_ARGUMENT, "opa_gemm_bias_act_bf16: K and N must be multiples of 64");
    if (((uintptr_t)a_dev | (uintptr_t)w_dev | (uintptr_t)out_dev | (uintptr_t)residual_dev) & 15)
        return fail(OPA_ERR_INVALID_ARGUMENT, "opa_gemm_bias_act_bf16: pointers must be 16-B aligned");
    if (m == 0) return OPA_OK;
    hipError_t e = launch_gemm_bias_act(a_dev, w_dev, bias_dev, residual_dev, out_dev, (int)m, n, k, relu,
                                        (hipStream_t)stream);
    if (e != hipSuccess) return fail_hip(e, "gemm_bias_act");
    prof_mark((hipStream_t)stream, "gemm_bias_act_kernel");
    return OPA_OK;
}

int opa_gemm_pro_bias_act_bf16(const void* a_dev, const void* a_bias_dev, const void* w_dev, const void* bias_dev,
                               const void* residual_dev, void* out_dev, int64_t m, int32_t n, int32_t k,
                               int32_t relu, void* stream) {
    if (!a_dev || !a_bias_dev || !w_dev || !bias_dev || !out_dev || m < 0 || n <= 0 || k <= 0 || m > 0x7fffffffll)
        return fail(OPA_ERR_INVALID_ARGUMENT, "opa_gemm_pro_bias_act_bf16: bad arguments");
    if (k % 64 != 0 || n % 64 != 0)
        return fail(OPA_ERR_INVALID_ARGUMENT, "opa_gemm_pro_bias_act_bf16: K and N must be multiples of 64");
    if (((uintptr_t)a_dev | (uintptr_t)a_bias_dev | (uintptr_t)w_dev | (uintptr_t)out_dev | (uintptr_t)residual_dev) & 15)
        return fail(OPA_ERR_INVALID_ARGUMENT, "opa_gemm_pro_bias_act_bf16: pointers must be 16-B aligned");
    if (m == 0) return OPA_OK;
    hipError_t e = launch_gemm_bias_act(a_dev, w_dev, bias_dev, residual_dev, out_dev, (int)m, n, k, relu,
                                        (hipStream_t)stream, a_bias_dev);
    if (e != hipSuccess) return fail_hip(e, "gemm_pro_bias_act");
    prof_mark((hipStream_t)stream, "gemm_pro_bias_act_kernel");
    return OPA_OK;
}

int opa_gemm_bias_act_f32(const float* a_dev, const float* a_bias_dev, const float* w_dev, const float* bias_dev,
                          const float* residual_dev, float* out_dev, int64_t m, int32_t n, int32_t k,
                          int32_t relu, void* stream) {
    if (!a_dev || !w_dev || !bias_dev || !out_dev || m < 0 || n <= 0 || k <= 0 || m > 0x7fffffffll)
        return fail(OPA_ERR_INVALID_ARGUMENT, "opa_gemm_bias_act_f32: bad arguments");
    if (k % 32 != 0 || n % 64 != 0)
        return fail(OPA_ERR_INVALID_ARGUMENT, "opa_gemm_bias_act_f32: K must be a multiple of 32 and N of 64");
    if (((uintptr_t)a_dev | (uintptr_t)a_bias_dev | (uintptr_t)w_dev | (uintptr_t)out_dev | (uintptr_t)residual_dev |
         (uintptr_t)bias_dev) & 15)
        return fail(OPA_ERR_INVALID_ARGUMENT, "opa_gemm_bias_act_f32: pointers must be 16-B aligned");
    if (m == 0) return OPA_OK;
    hipError_t e = launch_gemm_f32_bias_act(a_dev, w_dev, bias_dev, residual_dev, out_dev, (int)m, n, k, relu,
                                            (hipStream_t)stream, a_bias_dev);
    if (e != hipSuccess) return fail_hip(e, "gemm_f32_bias_act");
    prof_mark((hipStream_t)stream, a_bias_dev ? "gemm_f32_pro_bias_act_kernel" : "gemm_f32_bias_act_kernel");
    return OPA_OK;
}

int opa_gemm_bias_act_f32x3(const float* a_dev, const float* a_bias_dev, const void* w3_dev, const float* bias_dev,
                            const float* residual_dev, float* out_dev, int64_t m, int32_t n, int32_t k,
                            int32_t relu, int32_t terms, void* stream) {
    if (!a_dev || !w3_dev || !bias_dev || !out_dev || m < 0 || n <= 0 || k <= 0 || m > 0x7fffffffll || (terms != 6 && terms != 9))
        return fail(OPA_ERR_INVALID_ARGUMENT, "opa_gemm_bias_act_f32x3: bad arguments");
    if (k % 64 != 0 || n % 64 != 0)
        return fail(OPA_ERR_INVALID_ARGUMENT, "opa_gemm_bias_act_f32x3: K and N must be multiples of 64");
    if (((uintptr_t)a_dev | (uintptr_t)a_bias_dev | (uintptr_t)w3_dev | (uintptr_t)out_dev | (uintptr_t)residual_dev |
         (uintptr_t)bias_dev) & 15)
        return fail(OPA_ERR_INVALID_ARGUMENT, "opa_gemm_bias_act_f32x3: pointers must be 16-B aligned");
    if (m == 0) return OPA_OK;
    hipError_t e = launch_gemm_f32x3_bias_act(a_dev, (const unsigned short*)w3_dev, bias_dev, residual_dev, out_dev, (int)m, n, k,
                                              relu, terms, (hipStream_t)stream, a_bias_dev);
    if (e != hipSuccess) return fail_hip(e, "gemm_f32x3_bias_act");
    prof_mark((hipStream_t)stream, "gemm_f32x3_bias_act_kernel");
    return OPA_OK;
}

int opa_gemm2_bias_act_f32x3(const float* a1_dev, int32_t k1, const float* a2_dev, int32_t k2, int32_t batch, int32_t h_in,
                             int32_t w_in, int32_t stride, const float* a_bias_dev, const void* w3cat_dev, const float* bias_dev,
                             float* out_dev, int32_t n, int32_t relu, int32_t terms, void* stream) {
    if (!a1_dev || !a2_dev || !w3cat_dev || !bias_dev || !out_dev || batch <= 0 || h_in <= 0 || w_in <= 0 || stride < 1 || n <= 0 ||
        k1 <= 0 || k2 <= 0 || (terms != 6 && terms != 9))
        return fail(OPA_ERR_INVALID_ARGUMENT, "opa_gemm2_bias_act_f32x3: bad arguments");
    if (k1 % 32 != 0 || (k1 + k2) % 64 != 0 || k2 % 4 != 0 || n % 64 != 0)
        return fail(OPA_ERR_INVALID_ARGUMENT, "opa_gemm2_bias_act_f32x3: k1 % 32, (k1 + k2) % 64, k2 % 4, N % 64 must be 0");
    if (((uintptr_t)a1_dev | (uintptr_t)a2_dev | (uintptr_t)a_bias_dev | (uintptr_t)w3cat_dev | (uintptr_t)out_dev | (uintptr_t)bias_dev) & 15)
        return fail(OPA_ERR_INVALID_ARGUMENT, "opa_gemm2_bias_act_f32x3: pointers must be 16-B aligned");
    const long long ho = (h_in - 1) / stride + 1, wo = (w_in - 1) / stride + 1;
    if ((long long)batch * ho * wo > 0x7fffffffll || (long long)batch * h_in * w_in > 0x7fffffffll)
        return fail(OPA_ERR_INVALID_ARGUMENT, "opa_gemm2_bias_act_f32x3: too many pixels");
    hipError_t e = launch_gemm2_f32x3_bias_act(a1_dev, k1, a2_dev, k2, batch, h_in, w_in, stride, (const unsigned short*)w3cat_dev,
                                               bias_dev, out_dev, n, relu, terms, (hipStream_t)stream, a_bias_dev);
    if (e != hipSuccess) return fail_hip(e, "gemm2_f32x3_bias_act");
    prof_mark((hipStream_t)stream, "gemm2_f32x3_bias_act_kernel");
    return OPA_OK;
}

int opa_conv_rows_f32x3(const float* x_dev, const void* w3_dev, const float* bias_dev, float* out_dev, int32_t batch, int32_t hp,
                        int32_t wp, int32_t pix, int32_t ho, int32_t wo, int32_t stride, int32_t ntaps, int32_t tap_floats,
                        int32_t c_out, int32_t relu, int32_t terms, void* stream) {
    if (!x_dev || !w3_dev || !bias_dev || !out_dev || batch <= 0 || hp <= 0 || wp <= 0 || pix <= 0 || ho <= 0 || wo <= 0 || stride < 1 ||
        ntaps < 1 || ntaps > 32 || tap_floats <= 0 || c_out <= 0 || (terms != 6 && terms != 9))
        return fail(OPA_ERR_INVALID_ARGUMENT, "opa_conv_rows_f32x3: bad arguments");
    if (tap_floats % 32 != 0 || (ntaps * tap_floats) % 64 != 0 || c_out % 64 != 0 || pix % 4 != 0)
        return fail(OPA_ERR_INVALID_ARGUMENT, "opa_conv_rows_f32x3: tap_floats % 32, ntaps * tap_floats % 64, c_out % 64, pix % 4 must be 0");
    if (((uintptr_t)x_dev | (uintptr_t)w3_dev | (uintptr_t)out_dev | (uintptr_t)bias_dev) & 15)
        return fail(OPA_ERR_INVALID_ARGUMENT, "opa_conv_rows_f32x3: pointers must be 16-B aligned");
    // every tap of the last output pixel inside the tensor
    if ((long long)(ho - 1) * stride + ntaps > hp || ((long long)(wo - 1) * stride) * pix + tap_floats > (long long)wp * pix ||
        (long long)batch * hp * wp * pix * 4 > 0x7fffffffll || (long long)batch * ho * wo > 0x7fffffffll)
        return fail(OPA_ERR_INVALID_ARGUMENT, "opa_conv_rows_f32x3: the taps leave the (padded) input, or it is 2 GB or more");
    hipError_t e = launch_convrows_f32x3(x_dev, batch, hp, wp, pix, ho, wo, stride, ntaps, tap_floats, (const unsigned short*)w3_dev,
                                         bias_dev, out_dev, c_out, relu, terms, (hipStream_t)stream);
    if (e != hipSuccess) return fail_hip(e, "conv_rows_f32x3");
    prof_mark((hipStream_t)stream, "conv_rows_f32x3_kernel");
    return OPA_OK;
}

int opa_conv3x3_f32x3(const float* x_dev, const void* w3_dev, const float* bias_dev, float* out_dev, int32_t batch, int32_t h_in,
                      int32_t w_in, int32_t c_in, int32_t c_out, int32_t stride, int32_t relu, int32_t terms, void* stream) {
    if (!x_dev || !w3_dev || !bias_dev || !out_dev || batch <= 0 || h_in <= 0 || w_in <= 0 || stride < 1 || c_in <= 0 || c_out <= 0 ||
        (terms != 6 && terms != 9))
        return fail(OPA_ERR_INVALID_ARGUMENT, "opa_conv3x3_f32x3: bad arguments");
    if (c_in % 64 != 0 || c_out % 64 != 0)
        return fail(OPA_ERR_INVALID_ARGUMENT, "opa_conv3x3_f32x3: c_in and c_out must be multiples of 64");
    if (((uintptr_t)x_dev | (uintptr_t)w3_dev | (uintptr_t)out_dev | (uintptr_t)bias_dev) & 15)
        return fail(OPA_ERR_INVALID_ARGUMENT, "opa_conv3x3_f32x3: pointers must be 16-B aligned");
    if (((long long)batch * h_in * w_in + w_in + 1) * c_in * 4 > 0x7fffffffll)
        return fail(OPA_ERR_INVALID_ARGUMENT, "opa_conv3x3_f32x3: the activation must be smaller than 2 GB");
    hipError_t e = launch_conv3x3_f32x3(x_dev, batch, h_in, w_in, c_in, stride, (const unsigned short*)w3_dev, bias_dev, out_dev, c_out,
                                        relu, terms, (hipStream_t)stream);
    if (e != hipSuccess) return fail_hip(e, "conv3x3_f32x3");
    prof_mark((hipStream_t)stream, "conv3x3_f32x3_kernel");
    return OPA_OK;
}

int opa_conv3x3_winograd_f32(const float* x_dev, const float* u_dev, const float* bias_dev, float* out_dev, int32_t batch,
                             int32_t h, int32_t w, int32_t c_in, int32_t c_out, int32_t relu, int32_t variant,
                             int32_t order, void* stream) {
    if (!x_dev || !u_dev || !out_dev || batch <= 0 || h <= 0 || w <= 0 || c_in <= 0 || c_out <= 0 || variant < 0 || (variant > 3 && (variant < 11 || variant > 18)))
        return fail(OPA_ERR_INVALID_ARGUMENT, "opa_conv3x3_winograd_f32: bad arguments");
    if (variant != 1 ? (c_in % 16 != 0 || c_out % 64 != 0) : (c_in % 8 != 0 || c_out % 32 != 0))
        return fail(OPA_ERR_INVALID_ARGUMENT, "opa_conv3x3_winograd_f32: channel counts do not fit the variant's tiles");
    if ((double)batch * h * w * c_in >= 4294967296.0 || (double)batch * ((h + 1) / 2) * ((w + 1) / 2) >= 2147483647.0)
        return fail(OPA_ERR_INVALID_ARGUMENT, "opa_conv3x3_winograd_f32: the activation needs 32-bit element offsets");
    if (((uintptr_t)x_dev | (uintptr_t)u_dev | (uintptr_t)out_dev | (uintptr_t)bias_dev) & 15)
        return fail(OPA_ERR_INVALID_ARGUMENT, "opa_conv3x3_winograd_f32: pointers must be 16-B aligned");
    hipError_t e = launch_winograd_f23(x_dev, u_dev, out_dev, bias_dev, batch, h, w, c_in, c_out, relu, variant, order,
                                       (hipStream_t)stream);
    if (e != hipSuccess) return fail_hip(e, "winograd_f23");
    prof_mark((hipStream_t)stream, "winograd_f23_kernel");
    return OPA_OK;
}

int opa_dwconv_bias_act(const void* x_dev, int64_t x_pixel_stride, const void* w_dev, const void* bias_dev,
                        void* out_dev, int64_t out_pixel_stride, int32_t batch, int32_t h, int32_t w,
                        int32_t channels, int32_t k, int32_t stride, int32_t dtype, int32_t relu, void* stream) {
    if (!x_dev || !w_dev || !out_dev || batch <= 0 || h <= 0 || w <= 0 || channels <= 0 || x_pixel_stride < channels ||
        out_pixel_stride < channels || (k != 3 && k != 5) || (stride != 1 && stride != 2) || (dtype != 0 && dtype != 2))
        return fail(OPA_ERR_INVALID_ARGUMENT, "opa_dwconv_bias_act: bad arguments");
    hipError_t e = launch_dwconv(x_dev, x_pixel_stride, w_dev, bias_dev, out_dev, out_pixel_stride, batch, h, w, channels,
                                 k, stride, dtype, relu, (hipStream_t)stream);
    if (e != hipSuccess) return fail_hip(e, "depthwise convolution");
    return OPA_OK;
}

int opa_channel_interleave(const void* a_dev, int64_t a_pixel_stride, const void* b_dev, int64_t b_pixel_stride,
                           void* out_dev, int64_t rows, int32_t half, int32_t dtype, void* stream) {
    if (!a_dev || !b_dev || !out_dev || rows <= 0 || half <= 0 || a_pixel_stride < half || b_pixel_stride < half ||
        dtype < 0 || dtype > 2)
        return fail(OPA_ERR_INVALID_ARGUMENT, "opa_channel_interleave: bad arguments");
    hipError_t e = launch_channel_interleave(a_dev, a_pixel_stride, b_dev, b_pixel_stride, out_dev, rows, half, dtype,
                                             (hipStream_t)stream);
    if (e != hipSuccess) return fail_hip(e, "channel interleave");
    return OPA_OK;
}

int opa_head_epilogue(const void* conv_dev, int32_t dtype, int32_t batch, int32_t hc, int32_t wc,
                      int32_t n_fields, int32_t n_components, int32_t upsample, int32_t n_confidences,
                      int32_t n_vectors, uint32_t vector_offset_mask, int32_t n_scales, float* out_dev, void* stream) {
    if (!conv_dev || !out_dev || batch <= 0 || hc <= 0 || wc <= 0 || n_fields <= 0 || n_components <= 0 ||
        dtype < 0 || dtype > 2 || (upsample != 1 && upsample != 2) || n_confidences < 0 || n_vectors < 0 || n_scales < 0 ||
        1 + n_confidences + 2 * n_vectors + n_scales > n_components)
        return fail(OPA_ERR_INVALID_ARGUMENT, "opa_head_epilogue: bad arguments");
    hipError_t e = launch_head_epilogue(conv_dev, dtype, batch, hc, wc, n_fields, n_components, upsample, n_confidences,
                                        n_vectors, vector_offset_mask, n_scales, out_dev, (hipStream_t)stream);
    if (e != hipSuccess) return fail_hip(e, "head epilogue");
    return OPA_OK;
}

int opa_profile_begin(void* stream) {
    for (hipEvent_t ev : g_prof.events) (void)hipEventDestroy(ev);
    g_prof.events.clear(); g_prof.names.clear();
    g_prof.st = (hipStream_t)stream;
    g_prof.on = true;
    prof_mark(g_prof.st, "begin");
    if (g_prof.events.empty()) { g_prof.on = false; return fail(OPA_ERR_HIP, "opa_profile_begin: cannot record an event"); }
    return OPA_OK;
}

int opa_profile_end(int32_t capacity, const char** names_out, float* ms_out, int32_t* n_out) {
    if (!g_prof.on) return fail(OPA_ERR_INVALID_ARGUMENT, "opa_profile_end: no profile in progress");
    g_prof.on = false;
    hipError_t e = hipStreamSynchronize(g_prof.st);
    if (e != hipSuccess) return fail_hip(e, "opa_profile_end: sync");
    int32_t n = 0;
    for (size_t i = 1; i < g_prof.events.size(); i++) {
        float ms = 0.f;
        e = hipEventElapsedTime(&ms, g_prof.events[i - 1], g_prof.events[i]);
        if (e != hipSuccess) return fail_hip(e, "opa_profile_end: elapsed");
        if (n < capacity) { if (names_out) names_out[n] = g_prof.names[i]; if (ms_out) ms_out[n] = ms; }
        n++;
    }
    if (n_out) *n_out = n;
    for (hipEvent_t ev : g_prof.events) (void)hipEventDestroy(ev);
    g_prof.events.clear(); g_prof.names.clear();
    return OPA_OK;
}

}  // extern "C"
