/* openpifpaf_amd -- C ABI of the MI355X-native CifCaf decode path.
 *
 * This header is the drop-in boundary.  Every entry point replaces one piece
 * of the reference's native decoder interface, the TorchScript registrations
 * in /root/reference/src/openpifpaf/csrc/src/module.cpp:19-118 (cited per
 * function below as "ref: module.cpp:<line>").  Signatures are plain C: raw
 * pointers, sizes and a HIP stream passed as void*.  No torch types.
 *
 * Conventions
 *  - "dev" pointers are HIP device pointers (MI355X HBM); "host" pointers are
 *    ordinary host memory.  All tensors are dense, row-major, float32 unless
 *    stated otherwise, and batched: the leading dimension is the image.
 *  - Field layouts are the reference's CompositeField4 layouts
 *    (ref: network/heads.py:290,360-378):
 *      CIF [B, F, 5, H, W]  comps: 0 unused, 1 conf, 2 x, 3 y, 4 scale
 *      CAF [B, A, 8, H, W]  comps: 0 unused, 1 conf, 2 x1, 3 y1, 4 x2, 5 y2, 6 s1, 7 s2
 *  - Work is enqueued on `stream` and NOT synchronised; results are valid once
 *    the stream reaches that point.  No entry point allocates device memory:
 *    the caller owns the workspace (size from opa_cifcaf_workspace_bytes).
 *  - Return value: OPA_OK or an error code; opa_last_error() gives the text.
 *  - Semantics are those of a FRESH reference decoder instance per image
 *    (CifHr revision 1.0): see DESIGN.md "revision".
 */
#ifndef OPENPIFPAF_AMD_H_
#define OPENPIFPAF_AMD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OPA_OK 0
#define OPA_ERR_INVALID_ARGUMENT 1   /* bad shape / null pointer / bad handle      */
#define OPA_ERR_HIP 2                /* a HIP runtime call or kernel launch failed  */
#define OPA_ERR_UNSUPPORTED 3        /* option the HIP path does not implement      */
#define OPA_ERR_WORKSPACE 4          /* workspace too small                         */
#define OPA_ERR_NO_DEVICE 5          /* no gfx950 device visible                    */

/* out_count of opa_cifcaf_decode: number of valid rows, plus OPA_COUNT_OVERFLOW when poses were dropped for lack
 * of capacity (max_annotations too small), plus OPA_COUNT_FAILED when the image could not be decoded: the association
 * kernel's watchdog fired (a protocol error, never seen outside the test that provokes it; status word -1), or the image's
 * CIF map reaches more tiles than its pool holds (opa_shape::cifhr_pool_tiles; status word -2: decode again with a
 * larger pool).  The image then has 0 rows and its result must not be used.  Both flags travel with the count, so a
 * caller that reads the counts sees them. */
#define OPA_COUNT_OVERFLOW 0x40000000
#define OPA_COUNT_FAILED 0x20000000
#define OPA_COUNT_ROWS(c) ((c) & 0x0FFFFFFF)

/* The reference's process-global tunables (its C++ static members), one field
 * per STATIC_GETSET line of module.cpp:26-32,76-116 plus the constructor
 * constants of cifcaf.cpp:153 / cifcaf.hpp:103.  Defaults in comments. */
typedef struct opa_params {
    double cif_threshold;            /* CifHr::threshold               0.3     cif_hr.cpp:14        */
    int64_t cifhr_neighbors;         /* CifHr::neighbors               16      cif_hr.cpp:13        */
    double seed_threshold;           /* CifSeeds::threshold            0.2     cif_seeds.cpp:11     */
    double caf_threshold;            /* CafScored::default_score_th    0.3     caf_scored.cpp:11    */
    double cif_floor;                /* CafScored cif_floor            0.1     cifcaf.cpp:153       */
    double keypoint_threshold;       /* CifCaf::keypoint_threshold     0.15    cifcaf.cpp:20        */
    double keypoint_threshold_rel;   /* CifCaf::keypoint_threshold_rel 0.5     cifcaf.cpp:21        */
    double nms_suppression;          /* NMSKeypoints::suppression      1e-5    nms_keypoints.cpp:12 */
    double nms_instance_threshold;   /* NMSKeypoints::instance_threshold 0.15  nms_keypoints.cpp:13 */
    double nms_keypoint_threshold;   /* NMSKeypoints::keypoint_threshold 0.15  nms_keypoints.cpp:14 */
    double force_complete_caf_th;    /* CifCaf::force_complete_caf_th  0.001   cifcaf.cpp:24        */
    double occupancy_reduction;      /* Occupancy(reduction, .)        2.0     cifcaf.hpp:103       */
    double occupancy_min_scale;      /* Occupancy(., min_scale)        4.0     cifcaf.hpp:103       */
    int32_t greedy;                  /* CifCaf::greedy                 0       cifcaf.cpp:19        */
    int32_t reverse_match;           /* CifCaf::reverse_match          1       cifcaf.cpp:22        */
    int32_t force_complete;          /* CifCaf::force_complete         0       cifcaf.cpp:23        */
    int32_t block_joints;            /* CifCaf::block_joints           0       cifcaf.cpp:18 (no-op in the reference) */
    int32_t ablation_cifseeds_nms;        /* CifSeeds::ablation_nms          0  cif_seeds.cpp:13  */
    int32_t ablation_cifseeds_no_rescore; /* CifSeeds::ablation_no_rescore   0  cif_seeds.cpp:14  */
    int32_t ablation_caf_no_rescore;      /* CafScored::ablation_no_rescore  0  caf_scored.cpp:12 */
    int32_t ablation_cifhr_skip;          /* CifHr::ablation_skip            0  cif_hr.cpp:15     */
} opa_params;

/* Shapes of one batched decode call. */
typedef struct opa_shape {
    int32_t batch;            /* B images                                           */
    int32_t n_cif;            /* F = number of CIF fields (= number of keypoints K unless n_keypoints says otherwise) */
    int32_t n_caf;            /* A = number of CAF fields = number of bones         */
    int32_t cif_h, cif_w;     /* CIF field height/width                             */
    int32_t caf_h, caf_w;     /* CAF field height/width                             */
    int32_t cif_stride;       /* pixels per CIF cell (meta.stride)                  */
    int32_t caf_stride;       /* pixels per CAF cell                                */
    int32_t max_annotations;  /* capacity of the per-image annotation output        */
    int32_t n_keypoints;      /* K joints per annotation; 0 = n_cif.  K > n_cif is the reference's tracking
                               * setup (decoder/tracking_pose.py:47-80): joints F..K-1 belong to earlier
                               * frames, have no CIF field -- no seeds, no rescoring, no occupancy, no reverse
                               * match from them (cifcaf.cpp:173,225-229,397; caf_scored.cpp:15-18) -- and are
                               * reached only through initial annotations and bones                    */
    int32_t cifhr_pool_tiles; /* capacity, per image, of the high-resolution CIF map, which the decode keeps as a pool
                               * of 32x64-pixel tiles: only tiles the box of an active CIF cell reaches take a slot
                               * (a 20-person 641-px COCO image: ~450 of its 3927).  0 = automatic: every tile where
                               * the whole map is at most 32 MB per image (nothing can run out), else an eighth of the
                               * map and at least 1024 tiles, plus ONE spill region shared by the batch that holds what
                               * a single image's pool cannot (the rest of a whole map); -1 = every tile (the dense
                               * map's size: can never run out); > 0 = exactly that many, no spill region.  An image
                               * that reaches more tiles than pool + spill region hold is not decoded wrongly: its
                               * count carries OPA_COUNT_FAILED and its status word is -2 (decode it again with -1).  */
} opa_shape;

/* ---- library ------------------------------------------------------------ */
/* The structs above are passed by pointer and have grown over time (opa_shape::cifhr_pool_tiles is the latest field): a
 * caller built against another header would make the library read past its struct.  Check once at start-up that
 * opa_abi_version() == OPA_ABI_VERSION and opa_shape_bytes() == sizeof(opa_shape), opa_params_bytes() == sizeof(opa_params). */
#define OPA_ABI_VERSION 6
int opa_abi_version(void);
size_t opa_shape_bytes(void);
size_t opa_params_bytes(void);
size_t opa_debug_bytes(void);
const char* opa_version(void);
const char* opa_last_error(void);            /* thread-local, never NULL            */
int opa_device_count(void);                  /* number of visible gfx950 devices    */

/* ref: module.cpp:19-21  torch.ops.openpifpaf.set_quiet(bool) */
void opa_set_quiet(int quiet);

/* Order of seeds with EQUAL scores (ref: cif_seeds.cpp:94,118: an unstable std::sort, so the reference's order is
 * whatever libstdc++'s introsort leaves -- it decides which of two equally scored seeds is grown first).
 *   1 (default)  libstdc++'s order, reproduced on the device for the images that have such seeds (float32 fields of a
 *                network practically never do; fields rounded to bfloat16 do) -- introsort's partitions and, since round 6,
 *                its heapsort branch (std::__partial_sort for a segment at the depth limit).  The workspace view "seed_ties"
 *                (int32 [B]) says per image: 0 no equal scores, 1 re-ordered (-1: a protocol failure of the pass, never seen:
 *                the image keeps the order below).
 *   0            cell index ascending (field, row, column) -- one launch less.
 *   2            like 1 (rounds 4-5: "with the pass inside the association kernel" -- that is a decoder's own choice now,
 *                opa_cifcaf_set_tie_placement, and the default).
 * Process-global like the reference's statics; the environment variable OPA_SEED_TIES=index / libstdcxx-fused selects 0 / 2
 * when this function was never called (read once, when the library is loaded). */
void opa_set_seed_tie_order(int order);
int opa_get_seed_tie_order(void);

/* The reference keeps its tunables in process-global statics that are set
 * before decoders are built (ref: decoder/factory.py:52-82, decoder/cifcaf.py:175-211).
 * opa_get_params/opa_set_params are that global; every call also accepts an
 * explicit opa_params* (NULL = use the global). */
void opa_default_params(opa_params* out);
void opa_get_params(opa_params* out);
int opa_set_params(const opa_params* in);

/* ---- CifCaf decoder object ---------------------------------------------- */
/* ref: module.cpp:25,34  torch.classes.openpifpaf_decoder.CifCaf(n_keypoints, skeleton)
 * skeleton_host: int64 [n_bones, 2], 0-based joint indices, CAF field order
 * (ref: decoder/cifcaf.py:119-122).  The handle owns a small device copy of
 * the skeleton and its adjacency -- and, per caller stream it has decoded on, one side stream and two events (opa_debug::
 * side_stream), created at the first decode on that stream; it holds no per-call state, so one handle may be used from
 * several streams. */
typedef struct opa_cifcaf opa_cifcaf;
int opa_cifcaf_create(opa_cifcaf** out, int32_t n_keypoints,
                      const int64_t* skeleton_host, int32_t n_bones);
void opa_cifcaf_destroy(opa_cifcaf* dec);
/* ref: module.cpp:41-53 (pickle state = (n_keypoints, skeleton)) */
int opa_cifcaf_get_state(const opa_cifcaf* dec, int32_t* n_keypoints,
                         int64_t* skeleton_host /* [n_bones,2] or NULL */, int32_t* n_bones);

/* Where the pass that puts seeds of EQUAL score into the reference's order runs for THIS decoder's decodes (seed tie
 * order 1, see opa_set_seed_tie_order): 1 = inside the association kernel, every image its own ties; 0 = a launch of its own
 * (its time then shows up under its own name); -1 (default) = automatic, which since round 6 means inside the kernel -- measured
 * shorter for one decode at a time as well as for several in flight.  The results are the same bit for bit. */
int opa_cifcaf_set_tie_placement(opa_cifcaf* dec, int32_t inside_association);

/* A/B and test switches of ONE decoder handle (no reference counterpart; the reference's tunables are opa_params).  None of
 * them changes a result: they select between exact variants of the kernels, shorten the watchdog, or switch measurements on.
 * opa_default_debug() = the library's defaults with the environment variables named below applied -- the environment is read
 * ONCE, when the library is loaded, never during a decode (rounds 2-5 looked 20 variables up at every launch, and tests
 * flipped them mid-process next to running decode lanes).  opa_cifcaf_set_debug() copies the struct into the handle; a decode
 * reads only that copy. */
typedef struct opa_debug {
    int32_t stage_worklist;        /* 1: tiles of the CIF map through a work list, seeds from candidate lists (round 6); 0: round 5's
                                    *    per-plane tile kernel and a seed fill that streams the field      OPA_STAGE_WORKLIST      */
    int32_t fuse_scored;           /* 0; 1: CafScored::fill rides in the seed sort's launch (round 3, slower) OPA_FUSE_SCORED       */
    int32_t scored_one_pass;       /* 1: a force-complete decode builds both CAF list sets from ONE read of the field (round 6);
                                    *    0: two passes                                                       OPA_SCORED_ONE_PASS   */
    int32_t assoc_waves;           /* 0 = 12 waves per association workgroup; 8 (and 16 in -DOPA_ASSOC_ALL_WAVES builds) OPA_ASSOC_WAVES */
    int32_t assoc_growers;         /* 0 = as many growing waves as fit; n: at most n (other interleavings) OPA_ASSOC_GROWERS      */
    int32_t assoc_bbox;            /* 1: list scans skip chunks whose box misses the window                  OPA_ASSOC_BBOX         */
    int32_t assoc_dedup;           /* 1: later seeds of an occupancy cell already seen are dropped           OPA_ASSOC_DEDUP        */
    int32_t assoc_prededup;        /* 1: ... by the whole workgroup before the coordinator starts            OPA_ASSOC_PREDEDUP     */
    int32_t assoc_predict;         /* 1: joint boxes predicted from single cells of the raw CAF field        OPA_ASSOC_PREDICT      */
    float assoc_predict_min_v;     /* 0.5: seeds below this confidence grow without that walk                OPA_ASSOC_PREDICT_MINV */
    float assoc_predict_th;        /* 0.3: raw CAF confidence a predicted bone needs                         OPA_ASSOC_PREDICT_TH   */
    int32_t assoc_collide;         /* 1: a growth that runs into an earlier candidate's joint box is stopped OPA_ASSOC_COLLIDE      */
    int32_t assoc_collide_shift;   /* 1: ... only near the centre of that box (0: anywhere inside, round 4)  OPA_ASSOC_COLLIDE_SHIFT */
    int32_t assoc_inherit;         /* 1: a candidate inherits the predictions of growths stopped for it      OPA_ASSOC_INHERIT      */
    int32_t assoc_lookahead;       /* 1: large skeletons: the next person's first seed enters the pool early OPA_ASSOC_LOOKAHEAD    */
    int32_t assoc_help;            /* compiled-in variants only (-DOPA_ASSOC_HELPERS)                        OPA_ASSOC_HELP         */
    int32_t assoc_spec;            /* compiled-in variants only (-DOPA_ASSOC_WALK)                           OPA_ASSOC_SPEC         */
    int32_t assoc_timing;          /* 0; 1: the coordinator fills its per-phase tick counters                OPA_ASSOC_TIMING       */
    int32_t assoc_persistent;      /* 0 = automatic: in a batch of more images than the chip has compute units the association
                                    *    workgroups take their images from a queue, most seeds first (round 6); 1: always;
                                    *    -1: never (workgroup b = image b)                                   OPA_ASSOC_PERSISTENT   */
    int32_t fc_split;              /* 0 = automatic; n: force-complete workgroups per image                  OPA_FC_SPLIT           */
    int32_t side_stream;           /* one branch of the decode on a stream of the handle's own, joined before the association kernel
                                    *    (round 6): 0 none; 1 the CAF lists beside the seed chain (measured: the two do not overlap, the
                                    *    list building fills the chip); 2 the tie pass (as a launch of its own) beside the list building
                                    *                                                                        OPA_SIDE_STREAM        */
    int64_t assoc_watchdog_ticks;  /* 1e8 (one second): 10-ns ticks after which a wait inside the association kernel gives up
                                    *    and the image is flagged OPA_COUNT_FAILED                           OPA_ASSOC_WATCHDOG_TICKS */
} opa_debug;
void opa_default_debug(opa_debug* out);
int opa_cifcaf_set_debug(opa_cifcaf* dec, const opa_debug* in);
int opa_cifcaf_get_debug(const opa_cifcaf* dec, opa_debug* out);

/* Bytes of device workspace opa_cifcaf_decode needs for `shape`
 * (0 and an error text if the shape is invalid).
 *
 * The workspace may hold anything when it is first used.  Between calls it carries one piece of
 * state: a 256-byte header plus a bitmap of the 32x64 tiles of the high-resolution map the previous
 * call wrote to, so that tiles no CIF cell touches and that are already zero are not written again
 * (the per-tile form of the reference's lazy clear, cif_hr.cpp:97-121).  The bitmap is trusted only
 * while the header matches the layout of the call.  So: do not write into a workspace between the calls that use
 * it; after lending the memory to anything else, overwrite its first 256 bytes (hipMemset 0) --
 * that alone makes the next call treat every tile as dirty. */
size_t opa_cifcaf_workspace_bytes(const opa_shape* shape);

/* The same for one set of tunables (NULL = the process-global ones): without params->force_complete the second
 * CAF list set, its chunk boxes and counters -- about 30 % of the workspace -- are left out (they sit at the end of
 * the layout; every other offset is unchanged).  opa_cifcaf_decode with force_complete set refuses such a
 * workspace (OPA_ERR_WORKSPACE).  opa_cifcaf_workspace_bytes(shape) is the size that serves every setting. */
size_t opa_cifcaf_workspace_bytes_for(const opa_shape* shape, const opa_params* params);

/* ref: module.cpp:35-36  CifCaf.call / CifCaf.call_with_initial_annotations,
 * i.e. cifcaf.cpp:116-262, batched over B images and reading the field
 * tensors where the network wrote them (device memory; the reference
 * hard-requires CPU tensors, cifcaf.cpp:137-138).
 *
 *  cif_dev  [B,F,5,H,W], caf_dev [B,A,8,H,W]: read by the kernels the call queues on `stream` -- the CAF tensor by the LAST of
 *                   them too (the association kernel reads single cells of it, round 5): both have to stay as they are until
 *                   the work queued by this call has run (the call itself returns at once)
 *  initial_dev      optional [B, n_initial, K, 4] (v,x,y,s) or NULL
 *  initial_ids_dev  optional int64 [B, n_initial] or NULL
 *  out_dev          [B, max_annotations, K, 4] (v,x,y,s)   (ref: cifcaf.cpp:246-258), K = the decoder's n_keypoints
 *  out_ids_dev      int64 [B, max_annotations]             (ref: cifcaf.cpp:259)
 *  out_count_dev    int32 [B]  OPA_COUNT_ROWS(c) = number of valid rows of each image (<= max_annotations, in the
 *                   reference's output order, score descending); rows behind them are not written.  The
 *                   OPA_COUNT_OVERFLOW bit is set when the annotation capacity was too small: the poses the
 *                   seed loop produced after the first max_annotations (in seed order, cifcaf.cpp:206-231) were
 *                   dropped before keypoint NMS, and the workspace's "status" buffer holds how many.
 *                   OPA_COUNT_FAILED: see above (opa_debug::assoc_watchdog_ticks = the watchdog in 10-ns
 *                   ticks, default 1e8 = one second; tests shorten it to provoke the failure).
 */
int opa_cifcaf_decode(const opa_cifcaf* dec, const opa_shape* shape, const opa_params* params,
                      const float* cif_dev, const float* caf_dev,
                      const float* initial_dev, const int64_t* initial_ids_dev, int32_t n_initial,
                      void* workspace_dev, size_t workspace_bytes,
                      float* out_dev, int64_t* out_ids_dev, int32_t* out_count_dev,
                      void* stream);

/* ref: module.cpp:37-39  CifCaf.get_cifhr() -> (Tensor[F,Hhr,Whr], revision).
 * The decode keeps the high-resolution map as a pool of tiles (opa_shape::cifhr_pool_tiles); this call writes the map of
 * image `image` of the LAST opa_cifcaf_decode into that workspace as the dense float32 [n_cif, rows, cols] array the
 * reference returns -- out_dev, device memory of n_cif * rows * cols floats (rows / cols: opa_cifcaf_cifhr_view) --
 * asynchronously on `stream`.  Cell content is the reference buffer's content at revision 1.0: 0.0 = never touched,
 * otherwise 1.0 + accumulated confidence. */
int opa_cifcaf_get_cifhr(const opa_shape* shape, const void* workspace_dev, int32_t image, float* out_dev, void* stream);

/* Geometry of that array: rows = (cif_h - 1) * stride + 1, cols likewise; pitch = cols; revision = 1.0.
 * offset_floats: SIZE_MAX -- the map is NOT a dense array inside the workspace any more (rounds 1-3 returned its offset;
 * a caller that still indexes the workspace with it fails loudly instead of reading tile pools): use opa_cifcaf_get_cifhr. */
int opa_cifcaf_cifhr_view(const opa_shape* shape, size_t* offset_floats,
                          int32_t* rows, int32_t* cols, int32_t* pitch, double* revision);

/* Locates an intermediate buffer of the last opa_cifcaf_decode inside the workspace
 * (debugging / tests; the reference exposes its intermediates through the utility
 * classes of module.cpp:66-117).  what: "tile_bitmaps" (u32 [2][B*F][words]: tiles of the map written by
 * the previous / by this call), "cifhr", "seed_count", "seed_f", "seed_vxys", "seed_cell", "seed_ties",
 * "lists", "list_counts", "lists_fc", "list_counts_fc", "list_bbox" (f32 [B,A,2,C,4], C = min(ceil(caf_h*caf_w/64),
 * 255): xmin, xmax, ymin, ymax of the (x1, y1) columns of the first min(C, 16) chunks of 64 entries of every "lists"
 * list -- the rest of a list's C boxes is not written; an empty chunk: +inf, -inf, +inf, -inf), "list_bbox_fc" (the
 * boxes of ALL C chunks of every "lists_fc" list; written only by a force-complete decode), "occupancy",
 * "annotation_scratch", "status" (int32 [B]: poses dropped for lack of capacity; -1: the kernel's watchdog
 * fired; -2: the image's CIF map did not fit its tile pool), "cifhr_slots" (int32 [B, n_cif, tiles per plane]: slot of a map
 * tile in its image's pool, negative: no cell reaches the tile), "cifhr_overflow" (int32 [B]), "assoc_stats" (int32 [B,24] per image: 0 growths started, 1 poses accepted, 2 growths stopped because
 * their seed died, 3 finished growths dropped for the same reason, 4 growths stopped or given up on a prediction
 * (the seed stays pooled), 5 seeds handed out after having been predicted dead, 6 pool refills, 7 seeds,
 * 8 ticks until the growth phase ended, 9 ticks of the kernel, 10 sum of the growers' busy ticks, 11 list
 * scans, 12 coordinator ticks spent in iterations that only waited for the head's growth, 13 growers,
 * 14 poses stored, 15 coordinator iterations, 16 of them waiting, 17/18/19 ticks in commits / refills / hand-outs,
 * 20 ticks of the refills spent waiting for the growers' occupancy marks, 21 joint boxes published from PREDICTIONS (a growth's walk over
 * single cells of the raw CAF field before its search; round 5) and 22 seeds that entered the pool ahead of the scan (large skeletons' lookahead;
 * round 5) -- in diagnostic builds both also carry scan timing --, 23 seeds
 * dropped as later seeds of an occupancy cell already seen -- by the workgroup's pass over the seed list before the pool sees it
 * (round 5) and, for what that pass admits, at the refills; ticks are 10 ns; the tick
 * counters 12 and 17-20 are filled only with opa_debug::assoc_timing: each costs clock reads in the coordinator's loop), "assoc_trace" (int32 [B,64,4]: for the first
 * 64 accepted poses of an image the tick of the commit, of the hand-out and of the end of the growth, and
 * seed index | grower << 24), "assoc_queue" (int32 [B + 1]: batches of more images than compute units -- the images by seed count,
 * most first, then the queue's head; round 6), "cifhr_work" (int2 [..]: the tile kernel's work list, round 6).
 * The three "*_fc" regions lie at the END of the layout: their offsets are only inside a workspace of
 * opa_cifcaf_workspace_bytes() (or ..._bytes_for() with force_complete set); a workspace sized without the flag
 * ends before them -- do not dereference them there. */
int opa_cifcaf_workspace_view(const opa_shape* shape, const char* what,
                              size_t* offset_bytes, size_t* size_bytes);

/* ---- stage-level entry points (openpifpaf_decoder_utils) ---------------- */
/* ref: module.cpp:75-84  CifHr.reset + CifHr.accumulate (cif_hr.cpp:28-121).
 *  cifhr_dev [B, F, rows, pitch] with rows=(H-1)*stride+1, pitch from opa_cifhr_pitch();
 *  scratch_dev: opa_cifhr_scratch_bytes() bytes. */
int32_t opa_cifhr_pitch(int32_t cif_w, int32_t stride);
size_t opa_cifhr_scratch_bytes(int32_t batch, int32_t n_cif, int32_t cif_h, int32_t cif_w);
int opa_cifhr_accumulate(const float* cif_dev, int32_t batch, int32_t n_cif, int32_t cif_h, int32_t cif_w,
                         int32_t stride, double min_scale, double factor, const opa_params* params,
                         float* cifhr_dev, void* scratch_dev, size_t scratch_bytes, void* stream);

/* ref: module.cpp:86-94  CifSeeds(cifhr, revision).fill(cif, stride) + get()
 * (cif_seeds.cpp:33-66,93-114).  Output sorted by v descending; equal scores in
 * the order the reference's std::sort leaves them in (opa_set_seed_tie_order).
 *  seed_f_dev int32 [B, cap], seed_vxys_dev [B, cap, 4] (v,x,y,s), seed_count_dev int32 [B],
 *  cap = F*H*W;  scratch_dev: opa_cifseeds_scratch_bytes() bytes. */
size_t opa_cifseeds_scratch_bytes(int32_t batch, int32_t n_cif, int32_t cif_h, int32_t cif_w);
int opa_cifseeds_fill(const float* cif_dev, int32_t batch, int32_t n_cif, int32_t cif_h, int32_t cif_w,
                      int32_t stride, const float* cifhr_dev, const opa_params* params,
                      int32_t* seed_f_dev, float* seed_vxys_dev, int32_t* seed_count_dev,
                      void* scratch_dev, size_t scratch_bytes, void* stream);

/* ref: module.cpp:96-102  CifDetSeeds(cifhr, revision).fill(cifdet_field, stride) + get()
 * (cif_seeds.cpp:69-90,117-139).  field_dev [B, F, 6, H, W]; same ordering rule as opa_cifseeds_fill.
 *  seed_f_dev int32 [B, cap], seed_vxywh_dev [B, cap, 5] (v,x,y,w,h), seed_count_dev int32 [B],
 *  cap = F*H*W;  scratch_dev: opa_cifseeds_scratch_bytes() bytes. */
int opa_cifdetseeds_fill(const float* field_dev, int32_t batch, int32_t n_fields, int32_t field_h, int32_t field_w,
                         int32_t stride, const float* cifhr_dev, const opa_params* params,
                         int32_t* seed_f_dev, float* seed_vxywh_dev, int32_t* seed_count_dev,
                         void* scratch_dev, size_t scratch_bytes, void* stream);

/* ref: module.cpp:104-111  CafScored(cifhr, revision, score_th, cif_floor).fill(caf, stride, skeleton) + get()
 * (caf_scored.cpp:29-104).  Lists keep the reference's raster (j,i) order.
 *  skeleton_dev int64 [A,2] 0-based (device);  score_th < 0 -> params->caf_threshold
 *  lists_dev [B, A, 2(dir: 0 forward, 1 backward), 7, cap] planes (c,x1,y1,x2,y2,s1,s2), cap = caf_h*caf_w
 *  counts_dev int32 [B, A, 2] */
int opa_cafscored_fill(const float* caf_dev, int32_t batch, int32_t n_caf, int32_t caf_h, int32_t caf_w,
                       int32_t stride, const float* cifhr_dev, int32_t n_cif, int32_t cif_h, int32_t cif_w,
                       int32_t cif_stride, const int64_t* skeleton_dev, double score_th, double cif_floor,
                       const opa_params* params, float* lists_dev, int32_t* counts_dev, void* stream);

/* ref: module.cpp:55  torch.ops.openpifpaf_decoder.grow_connection_blend(caf[n,7], x, y, s, filter_sigmas, only_max)
 * (cifcaf.cpp:32-113).  rows_dev: device [n,7] row list; out_host[4] = (x, y, s, v).
 * Synchronous (copies the result back). */
int opa_grow_connection_blend(const float* rows_dev, int32_t n, double x, double y, double s,
                              double filter_sigmas, int32_t only_max, double* out_host, void* stream);

/* ---- CifDet decoder -------------------------------------------------------- */
/* ref: module.cpp:57-62  torch.classes.openpifpaf_decoder.CifDet().call(cifdet_field, stride)
 * (cifdet.cpp:24-80, CifDetHr cif_hr.cpp:124-150, CifDetSeeds cif_seeds.cpp:69-90,117-139), batched.
 *  field_dev       [B, F, 6, H, W]  comps: 0 unused, 1 conf, 2 x, 3 y, 4 w, 5 h
 *  categories_dev  int64 [B, max_detections]  (1-based field index, cifdet.cpp:61)
 *  scores_dev      [B, max_detections]
 *  boxes_dev       [B, max_detections, 4]  (x0, y0, x1, y1)
 *  counts_dev      int32 [B]
 * max_detections is the reference's static CifDet::max_detections_before_nms (120, cifdet.cpp:16).
 * The IoU NMS that follows is host-side Python in the reference too (decoder/cifdet.py:62-72). */
typedef struct opa_det_shape {
    int32_t batch, n_fields, field_h, field_w, stride, max_detections;
} opa_det_shape;
size_t opa_cifdet_workspace_bytes(const opa_det_shape* shape);
int opa_cifdet_decode(const opa_det_shape* shape, const opa_params* params, const float* field_dev,
                      void* workspace_dev, size_t workspace_bytes,
                      int64_t* categories_dev, float* scores_dev, float* boxes_dev, int32_t* counts_dev,
                      void* stream);

/* ---- producer-side helper ------------------------------------------------ */
/* Fused convolution epilogue of the field-producing network (no reference counterpart:
 * the reference runs conv -> batch-norm -> ReLU (-> add -> ReLU) as separate PyTorch ops,
 * network/basenetworks.py).  In place on an NHWC activation viewed as [rows, channels]:
 *     x = act(x + bias[channel] (+ residual))
 * dtype: 0 = float32, 1 = float16, 2 = bfloat16; channels must be a multiple of 16 bytes
 * worth of elements; x/residual 16-B aligned; residual may be NULL; relu 0/1. */
int opa_bias_act(void* x_dev, const void* bias_dev, const void* residual_dev,
                 int64_t rows, int32_t channels, int32_t dtype, int32_t relu, void* stream);

/* 1x1 convolution as an MFMA GEMM with the epilogue fused (bfloat16 in/out, f32 accumulate):
 *     out[M,N] = act(A[M,K] * W[N,K]^T + bias[N] (+ residual[M,N]))
 * A = NHWC activation viewed as [B*H*W, C_in], W = conv weight [C_out, C_in].  K % 64 == 0,
 * N % 64 == 0, all pointers 16-B aligned; residual may be NULL. */
int opa_gemm_bias_act_bf16(const void* a_dev, const void* w_dev, const void* bias_dev, const void* residual_dev,
                           void* out_dev, int64_t m, int32_t n, int32_t k, int32_t relu, void* stream);

/* The same GEMM with a fused PROLOGUE on its A operand: A is the raw output of the preceding (3x3)
 * convolution and  A'[m,k] = relu(A[m,k] + a_bias[k])  -- that convolution's bias + ReLU -- is applied while
 * the tile is staged, so its separate epilogue pass (opa_bias_act) disappears:
 *     out[M,N] = act(relu(A + a_bias) * W^T + bias (+ residual)) */
int opa_gemm_pro_bias_act_bf16(const void* a_dev, const void* a_bias_dev, const void* w_dev, const void* bias_dev,
                               const void* residual_dev, void* out_dev, int64_t m, int32_t n, int32_t k,
                               int32_t relu, void* stream);

/* Both of the above in float32 (v_mfma_f32_32x32x2f32, f32 operands and accumulation): the network at the
 * reference's precision.  a_bias_dev may be NULL (no prologue).  K % 32 == 0, N % 64 == 0, pointers 16-B aligned. */
int opa_gemm_bias_act_f32(const float* a_dev, const float* a_bias_dev, const float* w_dev, const float* bias_dev,
                          const float* residual_dev, float* out_dev, int64_t m, int32_t n, int32_t k,
                          int32_t relu, void* stream);

/* The float32 GEMM above on the bfloat16 MFMA pipe with NOTHING of either operand dropped (csrc/gemm_f32x3.hip): a float32 is
 * exactly the sum of three bfloat16 (its 24-bit significand cut into 8 + 8 + 8 bits), so a * b is the sum of nine bf16 x bf16
 * products, each exact in the float32 accumulator.  w3_dev = the weight split on the host ([3][N][K] bfloat16:
 * openpifpaf_amd.fused.split_weight), the activation is split while its tile is staged.  terms = 9: all nine products (the
 * only rounding left is the accumulation's, as in any float32 GEMM); 6: without the three smallest (< 2^-23 of a product each).
 * float32 in, float32 out; K % 64 == 0, N % 64 == 0, pointers 16-B aligned; a_bias_dev / residual_dev may be NULL. */
int opa_gemm_bias_act_f32x3(const float* a_dev, const float* a_bias_dev, const void* w3_dev, const float* bias_dev,
                            const float* residual_dev, float* out_dev, int64_t m, int32_t n, int32_t k,
                            int32_t relu, int32_t terms, void* stream);

/* Two 1x1 convolutions that are added -- the last convolution of a ResNet block and the block's downsampling convolution
 * (reference network/basenetworks.py:71-150: torchvision's Bottleneck, out = relu(bn3(conv3(h)) + downsample(x))) -- as ONE product
 * of the same kernel:  out[B, ho, wo, N] = act([a1 | a2 at stride] * w3cat^T + bias),  a1 [B*ho*wo, k1] (the block's inner
 * activation), a2 [B, h_in, w_in, k2] channels-last (the block's input; output pixel (y, x) reads input pixel (stride*y,
 * stride*x), ho = (h_in - 1) / stride + 1), w3cat = split_weight of the two weights concatenated along K ([3][N][k1 + k2]),
 * a_bias_dev [k1 + k2] or NULL (relu(. + a_bias) on the operand: zeros behind k1 leave the non-negative a2 as it is).  The
 * identity tensor is neither written nor read back.  k1 % 32 == 0, (k1 + k2) % 64 == 0, k2 % 4 == 0, N % 64 == 0. */
int opa_gemm2_bias_act_f32x3(const float* a1_dev, int32_t k1, const float* a2_dev, int32_t k2, int32_t batch, int32_t h_in,
                             int32_t w_in, int32_t stride, const float* a_bias_dev, const void* w3cat_dev, const float* bias_dev,
                             float* out_dev, int32_t n, int32_t relu, int32_t terms, void* stream);

/* 3x3 convolution, padding 1, ANY stride, of an NHWC float32 activation as an implicit GEMM of the split-operand kernel: the
 * strided convolution at the head of ResNet layers 2-4 (reference network/basenetworks.py:71-150: torch.nn.Conv2d -> MIOpen).
 * Column block t = 3 ky + kx of K = 9 c_in holds the channels of input pixel (stride*y - 1 + ky, stride*x - 1 + kx); pixels in
 * the padding are zeros.  x_dev [B, h_in, w_in, c_in] (< 2 GB), w3_dev = split_weight of the weight as [c_out, (ky, kx, c_in)]
 * ([3][c_out][9 c_in] bfloat16: openpifpaf_amd.fused.split_weight_3x3), out_dev [B, ho, wo, c_out], ho = (h_in - 1) / stride + 1.
 * c_in % 64 == 0, c_out % 64 == 0; terms 6 or 9 as above. */
int opa_conv3x3_f32x3(const float* x_dev, const void* w3_dev, const float* bias_dev, float* out_dev, int32_t batch, int32_t h_in,
                      int32_t w_in, int32_t c_in, int32_t c_out, int32_t stride, int32_t relu, int32_t terms, void* stream);

/* A convolution whose window ROWS are the taps of the same implicit GEMM, on an input that is padded IN MEMORY: the 7x7 stride-2
 * stem of a ResNet (reference network/basenetworks.py:71-150) on a 4-channel, zero-padded copy of the image.  x_dev [B, hp, wp, pix]
 * (pix floats per pixel); output pixel (y, x) reads for tap t the tap_floats contiguous floats that begin at input pixel
 * (stride*y + t, stride*x); w3_dev = split_weight of the weight laid out to match ([3][c_out][ntaps * tap_floats];
 * openpifpaf_amd.fused.stem7x7_bias_act_x3 pads the image by 3 + 4 pixels and the weight to 8 rows x 8 columns x 4 channels);
 * out_dev [B, ho, wo, c_out].  tap_floats % 32 == 0, ntaps * tap_floats % 64 == 0, ntaps <= 32, c_out % 64 == 0, x < 2 GB; the caller
 * guarantees that every tap of every output pixel lies inside x. */
int opa_conv_rows_f32x3(const float* x_dev, const void* w3_dev, const float* bias_dev, float* out_dev, int32_t batch, int32_t hp,
                        int32_t wp, int32_t pix, int32_t ho, int32_t wo, int32_t stride, int32_t ntaps, int32_t tap_floats,
                        int32_t c_out, int32_t relu, int32_t terms, void* stream);

/* 3x3 convolution, stride 1, padding 1, of an NHWC float32 activation as Winograd F(2x2, 3x3) in ONE kernel (input
 * transform -> sixteen float32 MFMA GEMMs -> output transform; csrc/winograd.hip): the bottleneck convolutions of the
 * ResNet trunk (reference network/basenetworks.py:71-150 runs them through torch.nn.Conv2d), 2.25x fewer multiplications
 * than the direct form, float32 arithmetic throughout.
 *  x_dev [B, h, w, c_in], u_dev = the filter transformed and laid out by openpifpaf_amd.winograd.transform_filter for
 *  `variant` (0, 2 and 3: 64 output channels per workgroup, c_in % 16 == 0, c_out % 64 == 0 -- 2 runs eight waves in two
 *  shifts on the same operand (the default of the Python side), 3 the same as persistent workgroups; 1: 32 channels, c_in % 8 == 0, c_out % 32 == 0), out_dev [B, h, w, c_out]; bias_dev [c_out] or NULL; relu 0/1; order 0 = workgroups of one tile block
 *  on one XCD, 1 = workgroups of one channel block on one XCD.  B*h*w*c_in < 2^32; pointers 16-B aligned. */
int opa_conv3x3_winograd_f32(const float* x_dev, const float* u_dev, const float* bias_dev, float* out_dev, int32_t batch,
                             int32_t h, int32_t w, int32_t c_in, int32_t c_out, int32_t relu, int32_t variant,
                             int32_t order, void* stream);

/* Depthwise k x k convolution (k = 3 or 5, stride 1 or 2, padding k/2) of a channels-last activation, with the
 * folded batch-norm bias and optionally ReLU fused (the ShuffleNetV2K unit of the reference,
 * network/basenetworks.py:186-268; MIOpen runs it as a grouped MFMA convolution, ~100x slower).
 *  x_dev [B, h, w, *] with x_pixel_stride elements between pixels (a channel slice of a wider tensor is fine),
 *  w_dev [k*k, channels] (tap-major), bias_dev [channels] or NULL, out_dev [B, ho, wo, *] with out_pixel_stride;
 *  dtype 0 = float32, 2 = bfloat16 (float32 accumulation). */
int opa_dwconv_bias_act(const void* x_dev, int64_t x_pixel_stride, const void* w_dev, const void* bias_dev,
                        void* out_dev, int64_t out_pixel_stride, int32_t batch, int32_t h, int32_t w,
                        int32_t channels, int32_t k, int32_t stride, int32_t dtype, int32_t relu, void* stream);

/* torch.cat((a, b), 1) followed by channel_shuffle(groups = 2) of the same unit, in one pass over channels-last rows:
 * out[r, 2i] = a[r, i], out[r, 2i+1] = b[r, i]; a / b with their own pixel strides, out dense [rows, 2*half]. */
int opa_channel_interleave(const void* a_dev, int64_t a_pixel_stride, const void* b_dev, int64_t b_pixel_stride,
                           void* out_dev, int64_t rows, int32_t half, int32_t dtype, void* stream);

/* The head of the field-producing network after its 1x1 convolution, in one pass (ref: network/heads.py:330-378
 * CompositeField4.forward): PixelShuffle(upsample) -> crop -> [B, n_fields, n_components, H, W] float32 -> sigmoid on
 * the n_confidences components after component 0, cell-index offsets on the vector components whose bit is set in
 * vector_offset_mask (x index on the first, y index on the second of each pair), softplus on the n_scales
 * components behind them.
 *  conv_dev  the convolution's output, channels-last [B, hc, wc, n_fields*n_components*upsample^2],
 *            dtype 0 = float32, 1 = float16, 2 = bfloat16
 *  out_dev   float32 [B, n_fields, n_components, H, W], H = hc*upsample - (upsample-1) (likewise W); upsample 1 or 2 */
int opa_head_epilogue(const void* conv_dev, int32_t dtype, int32_t batch, int32_t hc, int32_t wc,
                      int32_t n_fields, int32_t n_components, int32_t upsample, int32_t n_confidences,
                      int32_t n_vectors, uint32_t vector_offset_mask, int32_t n_scales, float* out_dev, void* stream);

/* ---- measurement -------------------------------------------------------- */
/* Per-kernel timing with HIP events on the launch stream (no reference
 * counterpart; bench.py's roofline leg uses it).  Between opa_profile_begin and
 * opa_profile_end every kernel/memset the library enqueues from THIS host thread
 * is followed by an event record; opa_profile_end synchronises the stream and
 * returns, per enqueued operation in order, its name (static string) and the
 * elapsed milliseconds since the previous event. */
int opa_profile_begin(void* stream);
int opa_profile_end(int32_t capacity, const char** names_out, float* ms_out, int32_t* n_out);

#ifdef __cplusplus
}
#endif
#endif  /* OPENPIFPAF_AMD_H_ */
