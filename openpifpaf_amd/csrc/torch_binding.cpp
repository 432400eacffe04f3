// TorchScript custom-class binding over the C ABI (include/openpifpaf_amd.h).
//
// The reference registers its native decoder as TorchScript classes/ops
// (csrc/src/module.cpp:19-118) so that it can be used from Python
// (decoder/cifcaf.py:119), embedded in an exported TorchScript model
// (export_torchscript.py:15-43) and loaded from C++ (cpp/cli_video.cpp:48-64).  This file is the
// same surface for the HIP path, under the namespaces
//     torch.ops.openpifpaf_amd.set_quiet
//     torch.classes.openpifpaf_amd_decoder.{CifCaf,CifDet} (+ call_batch)
//     torch.ops.openpifpaf_amd_decoder.grow_connection_blend
//     torch.classes.openpifpaf_amd_decoder_utils.{CifHr,CifSeeds,CafScored} (stage objects + static tunables),
//                                               CifDetSeeds, NMSKeypoints (static tunables),
//                                               Occupancy (host-side box map for host callers such as the
//                                               trackers, decoder/tracking_pose.py:81 -- the decode itself
//                                               tests occupancy inside the association kernel)
// Apart from that small host map it contains no compute: every method marshals tensors into the C ABI of
// libopenpifpaf_amd.so.
// Host C++ only (g++); built by openpifpaf_amd/build.py into lib/libopenpifpaf_amd_torch.so.
#include <torch/script.h>
#include <torch/custom_class.h>
#include <c10/hip/HIPStream.h>

#include <algorithm>
#include <cmath>
#include <tuple>
#include <vector>

#include "../../include/openpifpaf_amd.h"

namespace {

void check(int code, const char* what) {
    TORCH_CHECK(code == OPA_OK, what, " failed: ", opa_last_error());
}

void* current_stream(const torch::Tensor& t) {
    return (void*)c10::hip::getCurrentHIPStream(t.device().index()).stream();
}

torch::Tensor to_device_f32(const torch::Tensor& t) {
    torch::Tensor x = t;
    if (!x.is_cuda()) x = x.to(torch::Device(torch::kCUDA, c10::hip::current_device()));
    if (x.scalar_type() != torch::kFloat32) x = x.to(torch::kFloat32);
    return x.contiguous();
}

#define OPA_STATIC_GETSET(FIELD, T)                                                    \
    .def_static("set_" #FIELD, [](T v) { opa_params p; opa_get_params(&p); p.FIELD = v; \
                                          check(opa_set_params(&p), "opa_set_params"); }) \
    .def_static("get_" #FIELD, []() { opa_params p; opa_get_params(&p); return (T)p.FIELD; })

#define OPA_STATIC_GETSET_AS(NAME, FIELD, T)                                           \
    .def_static("set_" #NAME, [](T v) { opa_params p; opa_get_params(&p); p.FIELD = v;  \
                                         check(opa_set_params(&p), "opa_set_params"); }) \
    .def_static("get_" #NAME, []() { opa_params p; opa_get_params(&p); return (T)p.FIELD; })

struct CifCaf : torch::CustomClassHolder {
    int64_t n_keypoints;
    torch::Tensor skeleton;          // [A,2] int64, 0-based, CPU
    int64_t max_annotations = 128;
    opa_cifcaf* handle = nullptr;
    torch::Tensor workspace;         // caller-owned device workspace of the last call
    opa_shape last_shape{};
    int64_t cifhr_pool_tiles = 0;                    // opa_shape::cifhr_pool_tiles of the next decode
    bool has_last = false;

    CifCaf(int64_t n_keypoints_, const torch::Tensor& skeleton_) : n_keypoints(n_keypoints_) {
        TORCH_CHECK(skeleton_.dtype() == torch::kInt64, "skeleton must be of type LongTensor");   // cifcaf.hpp:106
        skeleton = skeleton_.detach().cpu().contiguous().view({-1, 2});
        check(opa_cifcaf_create(&handle, (int32_t)n_keypoints, skeleton.data_ptr<int64_t>(), (int32_t)skeleton.size(0)),
              "opa_cifcaf_create");
    }
    ~CifCaf() override { opa_cifcaf_destroy(handle); }

    void set_max_annotations(int64_t n) { max_annotations = n; }
    // opa_shape::cifhr_pool_tiles of the next decodes: 0 automatic, -1 every tile (can never run out), n tiles per image
    void set_cifhr_pool_tiles(int64_t n) { cifhr_pool_tiles = n < 0 ? -1 : n; }
    int64_t get_cifhr_pool_tiles() { return cifhr_pool_tiles; }
    void use_full_pool() { cifhr_pool_tiles = -1; }
    // opa_debug of this decoder's handle by field name (A/B and test switches: exact kernel variants, the watchdog; none changes
    // a result): set_debug("assoc_growers", 3); an unknown name raises
    void set_debug(const std::string& name, double value) {
        opa_debug d;
        check(opa_cifcaf_get_debug(handle, &d), "opa_cifcaf_get_debug");
#define OPA_DBG_FIELD(F, T) if (name == #F) { d.F = (T)value; check(opa_cifcaf_set_debug(handle, &d), "opa_cifcaf_set_debug"); return; }
        OPA_DBG_FIELD(stage_worklist, int32_t) OPA_DBG_FIELD(fuse_scored, int32_t) OPA_DBG_FIELD(scored_one_pass, int32_t)
        OPA_DBG_FIELD(assoc_waves, int32_t) OPA_DBG_FIELD(assoc_growers, int32_t) OPA_DBG_FIELD(assoc_bbox, int32_t)
        OPA_DBG_FIELD(assoc_dedup, int32_t) OPA_DBG_FIELD(assoc_prededup, int32_t) OPA_DBG_FIELD(assoc_predict, int32_t)
        OPA_DBG_FIELD(assoc_predict_min_v, float) OPA_DBG_FIELD(assoc_predict_th, float) OPA_DBG_FIELD(assoc_collide, int32_t)
        OPA_DBG_FIELD(assoc_collide_shift, int32_t) OPA_DBG_FIELD(assoc_inherit, int32_t) OPA_DBG_FIELD(assoc_lookahead, int32_t)
        OPA_DBG_FIELD(assoc_help, int32_t) OPA_DBG_FIELD(assoc_spec, int32_t) OPA_DBG_FIELD(assoc_timing, int32_t)
        OPA_DBG_FIELD(assoc_persistent, int32_t) OPA_DBG_FIELD(fc_split, int32_t) OPA_DBG_FIELD(side_stream, int32_t) OPA_DBG_FIELD(assoc_watchdog_ticks, int64_t)
#undef OPA_DBG_FIELD
        TORCH_CHECK(false, "opa_debug has no field ", name);
    }
    // did an image of the last decode run out of its tile pool (status -2)?  The only failure a larger pool cures.
    bool pool_overflowed() {
        if (!has_last) return false;
        size_t off = 0, bytes = 0;
        check(opa_cifcaf_workspace_view(&last_shape, "cifhr_overflow", &off, &bytes), "opa_cifcaf_workspace_view");
        const torch::Tensor flags = workspace.narrow(0, (int64_t)off, (int64_t)last_shape.batch * 4).view(torch::kInt32);
        return flags.ne(0).any().item<bool>();
    }

    // batched extension: cif [B,F,5,H,W], caf [B,A,8,H,W] -> (ann [B,max,K,4], ids [B,max], counts [B])
    std::tuple<torch::Tensor, torch::Tensor, torch::Tensor> call_batch_impl(
            const torch::Tensor& cif_in, int64_t cif_stride, const torch::Tensor& caf_in, int64_t caf_stride,
            const torch::optional<torch::Tensor>& initial, const torch::optional<torch::Tensor>& initial_ids) {
        torch::Tensor cif = to_device_f32(cif_in), caf = to_device_f32(caf_in);
        TORCH_CHECK(cif.dim() == 5 && caf.dim() == 5 && cif.size(2) == 5 && caf.size(2) == 8 && cif.size(0) == caf.size(0),
                    "expected cif [B,F,5,H,W] and caf [B,A,8,H,W]");
        opa_shape s;
        s.batch = (int32_t)cif.size(0); s.n_cif = (int32_t)cif.size(1); s.n_caf = (int32_t)caf.size(1);
        s.cif_h = (int32_t)cif.size(3); s.cif_w = (int32_t)cif.size(4);
        s.caf_h = (int32_t)caf.size(3); s.caf_w = (int32_t)caf.size(4);
        s.cif_stride = (int32_t)cif_stride; s.caf_stride = (int32_t)caf_stride;
        s.max_annotations = (int32_t)max_annotations;
        s.n_keypoints = (int32_t)n_keypoints;        // > n_cif in the tracking setup
        s.cifhr_pool_tiles = (int32_t)cifhr_pool_tiles;   // 0: automatic (+ the batch's spill region); -1 after an image did not fit, or on request
        // (the decode below runs with the process-global tunables: without force_complete the second list set is left out)
        const size_t need = opa_cifcaf_workspace_bytes_for(&s, nullptr);
        TORCH_CHECK(need > 0, "opa_cifcaf_workspace_bytes_for: ", opa_last_error());
        if (!workspace.defined() || (size_t)workspace.numel() < need || workspace.device() != cif.device()) {
            workspace = torch::empty({(int64_t)need}, torch::dtype(torch::kUInt8).device(cif.device()));
            workspace.narrow(0, 0, 256).zero_();     // recycled allocator memory: the lazy-clear header starts invalid
        }
        auto opts = torch::TensorOptions().device(cif.device());
        torch::Tensor out = torch::empty({s.batch, max_annotations, n_keypoints, 4}, opts.dtype(torch::kFloat32));
        torch::Tensor ids = torch::empty({s.batch, max_annotations}, opts.dtype(torch::kInt64));
        torch::Tensor counts = torch::empty({s.batch}, opts.dtype(torch::kInt32));
        torch::Tensor init_t, ids_t;
        int32_t n_initial = 0;
        if (initial.has_value() && initial->numel() > 0) {
            TORCH_CHECK(initial_ids.has_value(), "require initial_ids when initial_annotations are given");   // cifcaf.cpp:178
            init_t = to_device_f32(*initial).view({s.batch, -1, n_keypoints, 4});
            n_initial = (int32_t)init_t.size(1);
            ids_t = initial_ids->to(cif.device(), torch::kInt64).contiguous().view({s.batch, n_initial});
        }
        check(opa_cifcaf_decode(handle, &s, nullptr, cif.data_ptr<float>(), caf.data_ptr<float>(),
                                n_initial ? init_t.data_ptr<float>() : nullptr,
                                n_initial ? ids_t.data_ptr<int64_t>() : nullptr, n_initial,
                                workspace.data_ptr(), (size_t)workspace.numel(),
                                out.data_ptr<float>(), ids.data_ptr<int64_t>(), counts.data_ptr<int32_t>(),
                                current_stream(cif)),
              "opa_cifcaf_decode");
        last_shape = s; has_last = true;
        return std::make_tuple(out, ids, counts);
    }

    std::tuple<torch::Tensor, torch::Tensor, torch::Tensor> call_batch(
            const torch::Tensor& cif, int64_t cif_stride, const torch::Tensor& caf, int64_t caf_stride) {
        auto result = call_batch_impl(cif, cif_stride, caf, caf_stride, torch::nullopt, torch::nullopt);
        if (cifhr_pool_tiles == 0) {
            // Scripted and batched callers have no other place to recover: an image whose CIF map ran out of the automatic
            // pool AND the batch's spill region (several structureless, all-active fields in one batch) comes back flagged
            // with no rows.  One look at the counts (they are what a caller reads first anyway), and if need be the batch is
            // decoded again with a pool that holds every tile -- from then on this decoder keeps that pool.
            // (a failure of another kind -- the watchdog, status -1 -- is not cured by a larger pool: the flagged counts go back as
            // they are and the pool setting stays; this look at the counts synchronises with the stream, so a caller that wants
            // call_batch asynchronous chooses the pool up front: set_cifhr_pool_tiles(n) or use_full_pool())
            const torch::Tensor c = std::get<2>(result).cpu();
            if (c.bitwise_and(OPA_COUNT_FAILED).any().item<bool>() && pool_overflowed()) {
                cifhr_pool_tiles = -1;
                result = call_batch_impl(cif, cif_stride, caf, caf_stride, torch::nullopt, torch::nullopt);
            }
        }
        return result;
    }

    // module.cpp:36 -- single image, results on the device the fields came from
    std::tuple<torch::Tensor, torch::Tensor> call_with_initial_annotations(
            const torch::Tensor& cif, int64_t cif_stride, const torch::Tensor& caf, int64_t caf_stride,
            torch::optional<torch::Tensor> initial, torch::optional<torch::Tensor> initial_ids) {
        torch::optional<torch::Tensor> ia, ii;
        if (initial.has_value()) ia = initial->unsqueeze(0);
        if (initial_ids.has_value()) ii = initial_ids->unsqueeze(0);
        auto [out, ids, counts] = call_batch_impl(cif.unsqueeze(0), cif_stride, caf.unsqueeze(0), caf_stride, ia, ii);
        int64_t c = counts.cpu().item<int32_t>();
        if ((c & OPA_COUNT_FAILED) && cifhr_pool_tiles == 0 && pool_overflowed()) {
            //  the image's CIF cells reach more map tiles than the automatic pool holds (structureless
            // all-active fields do): once more with a pool that holds every tile
            cifhr_pool_tiles = -1;
            std::tie(out, ids, counts) = call_batch_impl(cif.unsqueeze(0), cif_stride, caf.unsqueeze(0), caf_stride, ia, ii);
            c = counts.cpu().item<int32_t>();
        }
        TORCH_CHECK(!(c & OPA_COUNT_FAILED), "the decode of the image failed (the association kernel's watchdog, status -1, "
                    "or a CIF map beyond its tile pool, status -2): the result is invalid");
        TORCH_CHECK(!(c & OPA_COUNT_OVERFLOW), "annotation capacity overflow: poses were dropped; call "
                    "set_max_annotations with a larger value");
        const int64_t n = OPA_COUNT_ROWS(c);
        torch::Tensor o = out[0].narrow(0, 0, n).clone(), i = ids[0].narrow(0, 0, n).clone();
        if (!cif.is_cuda()) { o = o.cpu(); i = i.cpu(); }
        return std::make_tuple(o, i);
    }

    // module.cpp:35
    std::tuple<torch::Tensor, torch::Tensor> call(const torch::Tensor& cif, int64_t cif_stride,
                                                  const torch::Tensor& caf, int64_t caf_stride) {
        return call_with_initial_annotations(cif, cif_stride, caf, caf_stride, torch::nullopt, torch::nullopt);
    }

    // module.cpp:37-39 -- view of the internal buffer (image 0 of the last call), revision
    std::tuple<torch::Tensor, double> get_cifhr() {
        if (!has_last) return std::make_tuple(torch::zeros({1, 1, 1}), 0.0);
        size_t off = 0; int32_t rows = 0, cols = 0, pitch = 0; double rev = 0.0;
        check(opa_cifcaf_cifhr_view(&last_shape, &off, &rows, &cols, &pitch, &rev), "opa_cifcaf_cifhr_view");
        const int64_t F = last_shape.n_cif;
        // the decode keeps the map as a pool of tiles: gathered into the dense array the reference returns
        torch::Tensor dense = torch::empty({F, rows, cols}, torch::dtype(torch::kFloat32).device(workspace.device()));
        check(opa_cifcaf_get_cifhr(&last_shape, workspace.data_ptr(), 0, dense.data_ptr<float>(), current_stream(workspace)),
              "opa_cifcaf_get_cifhr");
        return std::make_tuple(dense, rev);
    }
};

// module.cpp:57-62
struct CifDet : torch::CustomClassHolder {
    static int64_t max_detections_before_nms;        // cifdet.cpp:16
    torch::Tensor workspace;

    std::tuple<torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor> call_batch(const torch::Tensor& field_in,
                                                                                       int64_t stride) {
        torch::Tensor field = to_device_f32(field_in);
        TORCH_CHECK(field.dim() == 5 && field.size(2) == 6, "expected a CifDet field [B,F,6,H,W]");
        opa_det_shape s;
        s.batch = (int32_t)field.size(0); s.n_fields = (int32_t)field.size(1);
        s.field_h = (int32_t)field.size(3); s.field_w = (int32_t)field.size(4);
        s.stride = (int32_t)stride; s.max_detections = (int32_t)max_detections_before_nms;
        const size_t need = opa_cifdet_workspace_bytes(&s);
        TORCH_CHECK(need > 0, "opa_cifdet_workspace_bytes: ", opa_last_error());
        if (!workspace.defined() || (size_t)workspace.numel() < need || workspace.device() != field.device())
            workspace = torch::empty({(int64_t)need}, torch::dtype(torch::kUInt8).device(field.device()));
        auto opts = torch::TensorOptions().device(field.device());
        const int64_t M = s.max_detections;
        torch::Tensor cat = torch::empty({s.batch, M}, opts.dtype(torch::kInt64));
        torch::Tensor sc = torch::empty({s.batch, M}, opts.dtype(torch::kFloat32));
        torch::Tensor bx = torch::empty({s.batch, M, 4}, opts.dtype(torch::kFloat32));
        torch::Tensor cnt = torch::empty({s.batch}, opts.dtype(torch::kInt32));
        check(opa_cifdet_decode(&s, nullptr, field.data_ptr<float>(), workspace.data_ptr(), (size_t)workspace.numel(),
                                cat.data_ptr<int64_t>(), sc.data_ptr<float>(), bx.data_ptr<float>(),
                                cnt.data_ptr<int32_t>(), current_stream(field)),
              "opa_cifdet_decode");
        return std::make_tuple(cat, sc, bx, cnt);
    }

    std::tuple<torch::Tensor, torch::Tensor, torch::Tensor> call(const torch::Tensor& field, int64_t stride) {
        auto [cat, sc, bx, cnt] = call_batch(field.unsqueeze(0), stride);
        const int64_t n = cnt.cpu().item<int32_t>();
        torch::Tensor c = cat[0].narrow(0, 0, n).clone(), v = sc[0].narrow(0, 0, n).clone(),
                      b = bx[0].narrow(0, 0, n).clone();
        if (!field.is_cuda()) { c = c.cpu(); v = v.cpu(); b = b.cpu(); }
        return std::make_tuple(c, v, b);
    }
};
int64_t CifDet::max_detections_before_nms = 120;

// The pitched layout the stage entry points share: [F, rows, pitch] with pitch = opa_cifhr_pitch.
// `hr` may be the view get_accumulated() returned (used in place) or any [F, rows, cols] tensor (copied).
torch::Tensor pitched_cifhr(const torch::Tensor& hr_in) {
    TORCH_CHECK(hr_in.dim() == 3, "cifhr must be [F, rows, cols]");
    torch::Tensor hr = hr_in;
    if (!hr.is_cuda()) hr = hr.to(torch::Device(torch::kCUDA, c10::hip::current_device()));
    if (hr.scalar_type() != torch::kFloat32) hr = hr.to(torch::kFloat32);
    const int64_t F = hr.size(0), rows = hr.size(1), cols = hr.size(2);
    const int64_t pitch = opa_cifhr_pitch((int32_t)cols, 1);
    if (hr.stride(2) == 1 && hr.stride(1) == pitch && hr.stride(0) == rows * pitch) return hr;
    torch::Tensor buf = torch::zeros({F, rows, pitch}, hr.options());
    buf.narrow(2, 0, cols).copy_(hr);
    return buf.narrow(2, 0, cols);
}

// module.cpp:75-84
struct CifHr : torch::CustomClassHolder {
    torch::Tensor accumulated;       // [F, rows, pitch]
    torch::Tensor scratch;
    int64_t cols = 0;
    double revision = 0.0;

    void reset(std::vector<int64_t> shape, int64_t stride) {      // sizes are taken from the field in accumulate()
        accumulated = torch::Tensor(); cols = 0; revision = 0.0;
    }
    void accumulate(const torch::Tensor& cif_in, int64_t stride, double min_scale, double factor) {
        torch::Tensor cif = to_device_f32(cif_in);
        TORCH_CHECK(cif.dim() == 4 && cif.size(1) == 5, "expected a CIF field [F,5,H,W]");
        const int32_t F = (int32_t)cif.size(0), H = (int32_t)cif.size(2), W = (int32_t)cif.size(3);
        const int64_t rows = (int64_t)(H - 1) * stride + 1, pitch = opa_cifhr_pitch(W, (int32_t)stride);
        cols = (int64_t)(W - 1) * stride + 1;
        accumulated = torch::empty({F, rows, pitch}, cif.options());
        const size_t nbytes = opa_cifhr_scratch_bytes(1, F, H, W);
        scratch = torch::empty({(int64_t)nbytes}, torch::dtype(torch::kUInt8).device(cif.device()));
        check(opa_cifhr_accumulate(cif.data_ptr<float>(), 1, F, H, W, (int32_t)stride, min_scale, factor, nullptr,
                                   accumulated.data_ptr<float>(), scratch.data_ptr(), nbytes, current_stream(cif)),
              "opa_cifhr_accumulate");
        revision = 1.0;
    }
    std::tuple<torch::Tensor, double> get_accumulated() {
        TORCH_CHECK(accumulated.defined(), "CifHr.accumulate() has not been called");
        return std::make_tuple(accumulated.narrow(2, 0, cols), revision);
    }
};

// module.cpp:86-94
struct CifSeeds : torch::CustomClassHolder {
    torch::Tensor hr;                // view [F, rows, cols] on a pitched buffer
    torch::Tensor f, vxys, count, scratch;

    CifSeeds(const torch::Tensor& cifhr, double revision) : hr(pitched_cifhr(cifhr)) {
        TORCH_CHECK(revision == 1.0, "the HIP path stores the map at revision 1.0 (a fresh reference instance)");
    }
    void fill(const torch::Tensor& cif_in, int64_t stride) {
        torch::Tensor cif = to_device_f32(cif_in);
        TORCH_CHECK(cif.dim() == 4 && cif.size(1) == 5, "expected a CIF field [F,5,H,W]");
        const int32_t F = (int32_t)cif.size(0), H = (int32_t)cif.size(2), W = (int32_t)cif.size(3);
        TORCH_CHECK(hr.size(0) == F && hr.size(1) == (int64_t)(H - 1) * stride + 1 && hr.size(2) == (int64_t)(W - 1) * stride + 1,
                    "cifhr does not match the field shape and stride");
        const int64_t cap = (int64_t)F * H * W;
        auto opts = torch::TensorOptions().device(cif.device());
        f = torch::empty({cap}, opts.dtype(torch::kInt32));
        vxys = torch::empty({cap, 4}, opts.dtype(torch::kFloat32));
        count = torch::empty({1}, opts.dtype(torch::kInt32));
        const size_t nbytes = opa_cifseeds_scratch_bytes(1, F, H, W);
        scratch = torch::empty({(int64_t)nbytes}, opts.dtype(torch::kUInt8));
        check(opa_cifseeds_fill(cif.data_ptr<float>(), 1, F, H, W, (int32_t)stride, hr.data_ptr<float>(), nullptr,
                                f.data_ptr<int32_t>(), vxys.data_ptr<float>(), count.data_ptr<int32_t>(),
                                scratch.data_ptr(), nbytes, current_stream(cif)),
              "opa_cifseeds_fill");
    }
    std::tuple<torch::Tensor, torch::Tensor> get() {              // cif_seeds.cpp:93-114
        TORCH_CHECK(count.defined(), "CifSeeds.fill() has not been called");
        const int64_t n = count.cpu().item<int32_t>();
        return std::make_tuple(f.narrow(0, 0, n).to(torch::kInt64), vxys.narrow(0, 0, n).clone());
    }
};

// module.cpp:96-102
struct CifDetSeeds : torch::CustomClassHolder {
    torch::Tensor hr;
    torch::Tensor f, vxywh, count, scratch;

    CifDetSeeds(const torch::Tensor& cifhr, double revision) : hr(pitched_cifhr(cifhr)) {
        TORCH_CHECK(revision == 1.0, "the HIP path stores the map at revision 1.0 (a fresh reference instance)");
    }
    void fill(const torch::Tensor& field_in, int64_t stride) {
        torch::Tensor field = to_device_f32(field_in);
        TORCH_CHECK(field.dim() == 4 && field.size(1) == 6, "expected a CifDet field [F,6,H,W]");
        const int32_t F = (int32_t)field.size(0), H = (int32_t)field.size(2), W = (int32_t)field.size(3);
        TORCH_CHECK(hr.size(0) == F && hr.size(1) == (int64_t)(H - 1) * stride + 1 && hr.size(2) == (int64_t)(W - 1) * stride + 1,
                    "cifhr does not match the field shape and stride");
        const int64_t cap = (int64_t)F * H * W;
        auto opts = torch::TensorOptions().device(field.device());
        f = torch::empty({cap}, opts.dtype(torch::kInt32));
        vxywh = torch::empty({cap, 5}, opts.dtype(torch::kFloat32));
        count = torch::empty({1}, opts.dtype(torch::kInt32));
        const size_t nbytes = opa_cifseeds_scratch_bytes(1, F, H, W);
        scratch = torch::empty({(int64_t)nbytes}, opts.dtype(torch::kUInt8));
        check(opa_cifdetseeds_fill(field.data_ptr<float>(), 1, F, H, W, (int32_t)stride, hr.data_ptr<float>(), nullptr,
                                   f.data_ptr<int32_t>(), vxywh.data_ptr<float>(), count.data_ptr<int32_t>(),
                                   scratch.data_ptr(), nbytes, current_stream(field)),
              "opa_cifdetseeds_fill");
    }
    std::tuple<torch::Tensor, torch::Tensor> get() {              // cif_seeds.cpp:117-139
        TORCH_CHECK(count.defined(), "CifDetSeeds.fill() has not been called");
        const int64_t n = count.cpu().item<int32_t>();
        return std::make_tuple(f.narrow(0, 0, n).to(torch::kInt64), vxywh.narrow(0, 0, n).clone());
    }
};

// module.cpp:67-73; semantics of csrc/src/occupancy.cpp:13-79.  A host-side map for host-side callers: cells carry
// the epoch of their last set(), clear() starts a new epoch.
struct Occupancy : torch::CustomClassHolder {
    double reduction, min_scale_reduced;
    int64_t n = 1, rows = 1, cols = 1;
    uint32_t epoch = 1;
    std::vector<uint32_t> stamp = std::vector<uint32_t>(1, 0u);

    Occupancy(double reduction_, double min_scale) : reduction(reduction_), min_scale_reduced(min_scale / reduction_) {
        TORCH_CHECK(reduction_ > 0.0, "Occupancy: reduction must be > 0");
    }
    static int64_t clampi(int64_t v, int64_t lo, int64_t hi) { return v < lo ? lo : (v > hi ? hi : v); }
    void set(int64_t f, double x, double y, double sigma) {        // occupancy.cpp:13-29
        TORCH_CHECK(f >= 0 && f < n, "Occupancy.set: field index out of range");
        if (reduction != 1.0) {
            x /= reduction; y /= reduction;
            sigma = std::fmax(min_scale_reduced, sigma / reduction);
        }
        const int64_t minx = clampi((int64_t)(x - sigma), 0, cols - 1), miny = clampi((int64_t)(y - sigma), 0, rows - 1);
        const int64_t maxx = clampi((int64_t)(x + sigma), minx + 1, cols), maxy = clampi((int64_t)(y + sigma), miny + 1, rows);
        for (int64_t j = miny; j < maxy; j++) {
            uint32_t* row = stamp.data() + (f * rows + j) * cols;
            for (int64_t i = minx; i < maxx; i++) row[i] = epoch;
        }
    }
    bool get(int64_t f, double x, double y) {                      // occupancy.cpp:32-43
        if (f >= n) return true;
        TORCH_CHECK(f >= 0, "Occupancy.get: negative field index");
        if (reduction != 1.0) { x /= reduction; y /= reduction; }
        const int64_t xi = clampi((int64_t)x, 0, cols - 1), yi = clampi((int64_t)y, 0, rows - 1);
        return stamp[(f * rows + yi) * cols + xi] == epoch;
    }
    void reset(std::vector<int64_t> shape) {                       // occupancy.cpp:46-68
        TORCH_CHECK(shape.size() == 3 && shape[0] > 0 && shape[1] >= 0 && shape[2] >= 0, "Occupancy.reset: shape must be [n, rows, cols]");
        n = shape[0];
        rows = (int64_t)((double)shape[1] / reduction) + 1;
        cols = (int64_t)((double)shape[2] / reduction) + 1;
        if ((int64_t)stamp.size() < n * rows * cols) stamp.assign((size_t)(n * rows * cols), 0u);
        clear();
    }
    void clear() {                                                 // occupancy.cpp:71-77
        if (++epoch == 0u) { std::fill(stamp.begin(), stamp.end(), 0u); epoch = 1; }
    }
};

// module.cpp:104-111
struct CafScored : torch::CustomClassHolder {
    torch::Tensor hr, lists, counts;
    double score_th, cif_floor;

    CafScored(const torch::Tensor& cifhr, double revision, double score_th_, double cif_floor_)
        : hr(pitched_cifhr(cifhr)), score_th(score_th_), cif_floor(cif_floor_) {
        TORCH_CHECK(revision == 1.0, "the HIP path stores the map at revision 1.0 (a fresh reference instance)");
    }
    void fill(const torch::Tensor& caf_in, int64_t stride, const torch::Tensor& skeleton) {
        torch::Tensor caf = to_device_f32(caf_in);
        TORCH_CHECK(caf.dim() == 4 && caf.size(1) == 8, "expected a CAF field [A,8,H,W]");
        TORCH_CHECK(skeleton.dtype() == torch::kInt64, "skeleton must be of type LongTensor");
        const int32_t A = (int32_t)caf.size(0), H = (int32_t)caf.size(2), W = (int32_t)caf.size(3);
        torch::Tensor skel = skeleton.to(caf.device()).contiguous();
        auto opts = torch::TensorOptions().device(caf.device());
        lists = torch::empty({A, 2, 7, (int64_t)H * W}, opts.dtype(torch::kFloat32));
        counts = torch::empty({A, 2}, opts.dtype(torch::kInt32));
        // the map's own size is all the stage needs: describe it as a stride-1 field of that size
        check(opa_cafscored_fill(caf.data_ptr<float>(), 1, A, H, W, (int32_t)stride, hr.data_ptr<float>(),
                                 (int32_t)hr.size(0), (int32_t)hr.size(1), (int32_t)hr.size(2), 1,
                                 skel.data_ptr<int64_t>(), score_th, cif_floor, nullptr,
                                 lists.data_ptr<float>(), counts.data_ptr<int32_t>(), current_stream(caf)),
              "opa_cafscored_fill");
    }
    std::tuple<std::vector<torch::Tensor>, std::vector<torch::Tensor>> get() {   // caf_scored.cpp:86-104
        TORCH_CHECK(counts.defined(), "CafScored.fill() has not been called");
        torch::Tensor cnt = counts.cpu();
        auto acc = cnt.accessor<int32_t, 2>();
        std::vector<torch::Tensor> fwd, bwd;
        for (int64_t a = 0; a < lists.size(0); a++) {
            fwd.push_back(lists[a][0].narrow(1, 0, acc[a][0]).t().contiguous());
            bwd.push_back(lists[a][1].narrow(1, 0, acc[a][1]).t().contiguous());
        }
        return std::make_tuple(fwd, bwd);
    }
};

// static-only holder (module.cpp:113-117)
struct NMSKeypointsStatics : torch::CustomClassHolder {};

std::vector<double> grow_connection_blend(const torch::Tensor& caf, double x, double y, double s,
                                          double filter_sigmas, bool only_max) {   // module.cpp:55
    torch::Tensor rows = to_device_f32(caf).view({-1, 7});
    double out[4] = {0, 0, 0, 0};
    check(opa_grow_connection_blend(rows.numel() ? rows.data_ptr<float>() : nullptr, (int32_t)rows.size(0), x, y, s,
                                    filter_sigmas, only_max ? 1 : 0, out, current_stream(rows)),
          "opa_grow_connection_blend");
    return {out[0], out[1], out[2], out[3]};
}

}  // namespace

TORCH_LIBRARY(openpifpaf_amd, m) {
    m.def("set_quiet", [](bool quiet) { opa_set_quiet(quiet ? 1 : 0); });                // module.cpp:19-21
    m.def("set_seed_tie_order", [](bool libstdcxx) { opa_set_seed_tie_order(libstdcxx ? 1 : 0); });   // (no counterpart: cif_seeds.cpp:94 is what it is)
}

TORCH_LIBRARY(openpifpaf_amd_decoder, m) {
    m.class_<CifCaf>("CifCaf")
        OPA_STATIC_GETSET(block_joints, bool)                                           // module.cpp:26-32
        OPA_STATIC_GETSET(greedy, bool)
        OPA_STATIC_GETSET(keypoint_threshold, double)
        OPA_STATIC_GETSET(keypoint_threshold_rel, double)
        OPA_STATIC_GETSET(reverse_match, bool)
        OPA_STATIC_GETSET(force_complete, bool)
        OPA_STATIC_GETSET(force_complete_caf_th, double)
        .def(torch::init<int64_t, const torch::Tensor&>())                               // :34
        .def("call", &CifCaf::call)                                                      // :35
        .def("call_with_initial_annotations", &CifCaf::call_with_initial_annotations)    // :36
        .def("call_batch", &CifCaf::call_batch)
        .def("set_max_annotations", &CifCaf::set_max_annotations)
        .def("set_cifhr_pool_tiles", &CifCaf::set_cifhr_pool_tiles)
        .def("get_cifhr_pool_tiles", &CifCaf::get_cifhr_pool_tiles)
        .def("use_full_pool", &CifCaf::use_full_pool)
        .def("set_debug", &CifCaf::set_debug)
        .def("get_cifhr", &CifCaf::get_cifhr)                                            // :37-39
        // :41-53: the state is the reference's (n_keypoints, skeleton) pair.  This build's two capacities travel INSIDE the skeleton
        // tensor, as one extra row (-1 - max_annotations, cifhr_pool_tiles) -- no joint index is negative -- so that a saved module
        // decodes like the live one while modules saved by a build (or a reference) that knows only the pair still load.
        .def_pickle(
            [](const c10::intrusive_ptr<CifCaf>& self) -> std::tuple<int64_t, torch::Tensor> {
                torch::Tensor extra = torch::empty({1, 2}, torch::kInt64);
                extra[0][0] = -1 - self->max_annotations; extra[0][1] = self->cifhr_pool_tiles;
                return std::make_tuple(self->n_keypoints, torch::cat({self->skeleton, extra}, 0));
            },
            [](std::tuple<int64_t, torch::Tensor> state) -> c10::intrusive_ptr<CifCaf> {
                torch::Tensor sk = std::get<1>(state).detach().cpu().contiguous().view({-1, 2});
                int64_t max_ann = -1, pool = 0;
                const int64_t n = sk.size(0);
                if (n > 0 && sk[n - 1][0].item<int64_t>() < 0) {
                    max_ann = -1 - sk[n - 1][0].item<int64_t>(); pool = sk[n - 1][1].item<int64_t>();
                    sk = sk.narrow(0, 0, n - 1).contiguous();
                }
                auto obj = c10::make_intrusive<CifCaf>(std::get<0>(state), sk);
                if (max_ann > 0) obj->max_annotations = max_ann;
                obj->cifhr_pool_tiles = pool;
                return obj;
            });
    m.def("grow_connection_blend", grow_connection_blend);                               // :55
    m.class_<CifDet>("CifDet")                                                           // :57-62
        .def_static("set_max_detections_before_nms", [](int64_t v) { CifDet::max_detections_before_nms = v; })
        .def_static("get_max_detections_before_nms", []() { return CifDet::max_detections_before_nms; })
        .def(torch::init<>())
        .def("call", &CifDet::call)
        .def("call_batch", &CifDet::call_batch);
}

TORCH_LIBRARY(openpifpaf_amd_decoder_utils, m) {
    m.class_<CifHr>("CifHr")                                                             // :75-84
        OPA_STATIC_GETSET_AS(neighbors, cifhr_neighbors, int64_t)
        OPA_STATIC_GETSET_AS(threshold, cif_threshold, double)
        OPA_STATIC_GETSET_AS(ablation_skip, ablation_cifhr_skip, bool)
        .def(torch::init<>())
        .def("accumulate", &CifHr::accumulate)
        .def("get_accumulated", &CifHr::get_accumulated)
        .def("reset", &CifHr::reset);
    m.class_<CifSeeds>("CifSeeds")                                                       // :86-94
        OPA_STATIC_GETSET_AS(threshold, seed_threshold, double)
        OPA_STATIC_GETSET_AS(ablation_nms, ablation_cifseeds_nms, bool)
        OPA_STATIC_GETSET_AS(ablation_no_rescore, ablation_cifseeds_no_rescore, bool)
        .def(torch::init<const torch::Tensor&, double>())
        .def("fill", &CifSeeds::fill)
        .def("get", &CifSeeds::get);
    m.class_<CifDetSeeds>("CifDetSeeds")                                                 // :96-102
        OPA_STATIC_GETSET_AS(threshold, seed_threshold, double)
        .def(torch::init<const torch::Tensor&, double>())
        .def("fill", &CifDetSeeds::fill)
        .def("get", &CifDetSeeds::get);
    m.class_<Occupancy>("Occupancy")                                                     // :67-73
        .def(torch::init<double, double>())
        .def("get", &Occupancy::get)
        .def("set", &Occupancy::set)
        .def("reset", &Occupancy::reset)
        .def("clear", &Occupancy::clear);
    m.class_<CafScored>("CafScored")                                                     // :104-111
        OPA_STATIC_GETSET_AS(default_score_th, caf_threshold, double)
        OPA_STATIC_GETSET_AS(ablation_no_rescore, ablation_caf_no_rescore, bool)
        .def(torch::init<const torch::Tensor&, double, double, double>())
        .def("fill", &CafScored::fill)
        .def("get", &CafScored::get);
    m.class_<NMSKeypointsStatics>("NMSKeypoints")                                        // :113-117
        OPA_STATIC_GETSET_AS(instance_threshold, nms_instance_threshold, double)
        OPA_STATIC_GETSET_AS(keypoint_threshold, nms_keypoint_threshold, double)
        OPA_STATIC_GETSET_AS(suppression, nms_suppression, double);
}
