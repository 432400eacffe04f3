// TorchScript custom-class binding over the C ABI (include/openpifpaf_amd.h).
//
// The reference registers its native decoder as TorchScript classes/ops
// (csrc/src/module.cpp:19-118) so that it can be used from Python
// (decoder/cifcaf.py:119), embedded in an exported TorchScript model
// (export_torchscript.py:15-43) and loaded from C++ (cpp/cli_video.cpp:48-64).  This file is the
// same surface for the HIP path, under the namespaces
//     torch.ops.openpifpaf_amd.set_quiet
//     torch.classes.openpifpaf_amd_decoder.CifCaf          (+ call_batch)
//     torch.ops.openpifpaf_amd_decoder.grow_connection_blend
//     torch.classes.openpifpaf_amd_decoder_utils.{CifHr,CifSeeds,CafScored,NMSKeypoints}  (static tunables)
// It contains no compute: every method marshals tensors into the C ABI of libopenpifpaf_amd.so.
// Host C++ only (g++); built by openpifpaf_amd/build.py into lib/libopenpifpaf_amd_torch.so.
#include <torch/script.h>
#include <torch/custom_class.h>
#include <c10/hip/HIPStream.h>

#include <tuple>

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
    bool has_last = false;

    CifCaf(int64_t n_keypoints_, const torch::Tensor& skeleton_) : n_keypoints(n_keypoints_) {
        TORCH_CHECK(skeleton_.dtype() == torch::kInt64, "skeleton must be of type LongTensor");   // cifcaf.hpp:106
        skeleton = skeleton_.detach().cpu().contiguous().view({-1, 2});
        check(opa_cifcaf_create(&handle, (int32_t)n_keypoints, skeleton.data_ptr<int64_t>(), (int32_t)skeleton.size(0)),
              "opa_cifcaf_create");
    }
    ~CifCaf() override { opa_cifcaf_destroy(handle); }

    void set_max_annotations(int64_t n) { max_annotations = n; }

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
        const size_t need = opa_cifcaf_workspace_bytes(&s);
        TORCH_CHECK(need > 0, "opa_cifcaf_workspace_bytes: ", opa_last_error());
        if (!workspace.defined() || (size_t)workspace.numel() < need || workspace.device() != cif.device())
            workspace = torch::empty({(int64_t)need}, torch::dtype(torch::kUInt8).device(cif.device()));
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
        return call_batch_impl(cif, cif_stride, caf, caf_stride, torch::nullopt, torch::nullopt);
    }

    // module.cpp:36 -- single image, results on the device the fields came from
    std::tuple<torch::Tensor, torch::Tensor> call_with_initial_annotations(
            const torch::Tensor& cif, int64_t cif_stride, const torch::Tensor& caf, int64_t caf_stride,
            torch::optional<torch::Tensor> initial, torch::optional<torch::Tensor> initial_ids) {
        torch::optional<torch::Tensor> ia, ii;
        if (initial.has_value()) ia = initial->unsqueeze(0);
        if (initial_ids.has_value()) ii = initial_ids->unsqueeze(0);
        auto [out, ids, counts] = call_batch_impl(cif.unsqueeze(0), cif_stride, caf.unsqueeze(0), caf_stride, ia, ii);
        const int64_t n = counts.cpu().item<int32_t>();
        TORCH_CHECK(n <= max_annotations, "annotation capacity overflow: ", n - max_annotations,
                    " dropped; call set_max_annotations with a larger value");
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
        torch::Tensor flat = workspace.narrow(0, (int64_t)off * 4, F * rows * pitch * 4).view(torch::kFloat32);
        return std::make_tuple(flat.view({F, rows, pitch}).narrow(2, 0, cols), rev);
    }
};

// static-only holders for the utility classes' tunables (module.cpp:75-117)
struct CifHrStatics : torch::CustomClassHolder {};
struct CifSeedsStatics : torch::CustomClassHolder {};
struct CafScoredStatics : torch::CustomClassHolder {};
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
        .def("get_cifhr", &CifCaf::get_cifhr)                                            // :37-39
        .def_pickle(                                                                      // :41-53
            [](const c10::intrusive_ptr<CifCaf>& self) -> std::tuple<int64_t, torch::Tensor> {
                return std::make_tuple(self->n_keypoints, self->skeleton);
            },
            [](std::tuple<int64_t, torch::Tensor> state) -> c10::intrusive_ptr<CifCaf> {
                return c10::make_intrusive<CifCaf>(std::get<0>(state), std::get<1>(state));
            });
    m.def("grow_connection_blend", grow_connection_blend);                               // :55
}

TORCH_LIBRARY(openpifpaf_amd_decoder_utils, m) {
    m.class_<CifHrStatics>("CifHr")                                                      // :75-79
        OPA_STATIC_GETSET_AS(neighbors, cifhr_neighbors, int64_t)
        OPA_STATIC_GETSET_AS(threshold, cif_threshold, double)
        OPA_STATIC_GETSET_AS(ablation_skip, ablation_cifhr_skip, bool);
    m.class_<CifSeedsStatics>("CifSeeds")                                                // :86-90
        OPA_STATIC_GETSET_AS(threshold, seed_threshold, double)
        OPA_STATIC_GETSET_AS(ablation_nms, ablation_cifseeds_nms, bool)
        OPA_STATIC_GETSET_AS(ablation_no_rescore, ablation_cifseeds_no_rescore, bool);
    m.class_<CafScoredStatics>("CafScored")                                              // :104-107
        OPA_STATIC_GETSET_AS(default_score_th, caf_threshold, double)
        OPA_STATIC_GETSET_AS(ablation_no_rescore, ablation_caf_no_rescore, bool);
    m.class_<NMSKeypointsStatics>("NMSKeypoints")                                        // :113-117
        OPA_STATIC_GETSET_AS(instance_threshold, nms_instance_threshold, double)
        OPA_STATIC_GETSET_AS(keypoint_threshold, nms_keypoint_threshold, double)
        OPA_STATIC_GETSET_AS(suppression, nms_suppression, double);
}
