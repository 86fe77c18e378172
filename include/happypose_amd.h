/*
 * happypose_amd -- C ABI of the MI355X-native render-and-compare refinement path.
 *
 * This header is the drop-in boundary.  The reference is pure
 * Python with no FFI on this path, so each entry point replaces a Python-level
 * interface of the reference; the citation next to it is the reference function it
 * stands in for (paths relative to the reference root; TB/ = happypose/toolbox/,
 * MP/ = happypose/pose_estimators/megapose/, CP/ = happypose/pose_estimators/cosypose/cosypose/).
 * INTEGRATION.md shows the ctypes stub a reference maintainer would add.
 *
 * Conventions
 *  - plain pointers and sizes only; every `d_*` pointer is DEVICE memory (HBM), every
 *    `h_*` pointer is host memory; float = IEEE fp32; poses are row-major 4x4, intrinsics
 *    row-major 3x3.
 *  - all work is enqueued on `stream` (a hipStream_t passed as void*; NULL = default
 *    stream) and returns without synchronising; no hidden allocations after create().
 *  - return value 0 = success, negative = error (hp_last_error() gives the text).  Shape
 *    or argument errors are reported, never silently repaired -- mirroring the reference's
 *    asserts (e.g. TB/renderer/panda3d_batch_renderer.py:166-169).
 *  - non-finite poses/intrinsics render as all-zero images, not an error
 *    (TB/renderer/panda3d_batch_renderer.py:81-111).
 */
#ifndef HAPPYPOSE_AMD_H
#define HAPPYPOSE_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* The library is built with -fvisibility=hidden: the declarations of this header -- and nothing else -- are its dynamic
 * symbols (tests/test_abi.py: nm -D shows the hp_* entry points only). */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

#define HP_OK 0
#define HP_ERR_ARG -1
#define HP_ERR_HIP -2
#define HP_ERR_STATE -3

int hp_version(void);
const char* hp_last_error(void);
/* number of visible HIP devices / name of the current one (diagnostics) */
int hp_device_count(void);
int hp_device_name(char* buf, int len);

/* ------------------------------------------------------------------------------------
 * Mesh store: device-resident geometry + textures of an object set, and the padded
 * mesh-point database.  Replaces Panda3dBatchRenderer.__init__(asset_dataset, ...)
 * (TB/renderer/panda3d_batch_renderer.py:129-142, worker start-up :288-330, model cache
 * TB/renderer/panda3d_scene_renderer.py:206-219) and MeshDataBase.batched().to(device)
 * (TB/lib3d/rigid_mesh_database.py:84-130).
 *
 * h_obj: [n_obj][8] int64 = vert_off, n_verts, face_off, n_faces, tex_off(-1 none),
 *        tex_w, tex_h, 0.   faces hold vertex ids LOCAL to their object.
 * h_points: [n_obj][n_pad][3] metres (may be NULL when only rendering is needed).
 * ---------------------------------------------------------------------------------- */
typedef struct hp_mesh_store hp_mesh_store;

hp_mesh_store* hp_mesh_store_create(const float* h_verts, const float* h_normals,
                                    const float* h_uvs, const uint8_t* h_colors,
                                    int64_t n_verts_total, const int32_t* h_faces,
                                    int64_t n_faces_total, const uint8_t* h_tex,
                                    int64_t tex_bytes, const int64_t* h_obj, int n_obj,
                                    const float* h_points, int n_pad);
void hp_mesh_store_destroy(hp_mesh_store* store);
/* device pointer of the [n_obj][n_pad][3] point table (NULL if not uploaded) */
const float* hp_mesh_store_points(const hp_mesh_store* store);
/* The rasteriser keeps per-(view, band) triangle lists and per-(view, vertex) records in scratch memory owned by the
 * store; it grows on demand (never under stream capture: hp_rasterize then fails with HP_ERR_ARG).  Reserve it for the
 * largest call -- n_views views of h x w, flags: HP_RASTER_MSAA4 if multisampled renders will be asked for -- so that no
 * later call reallocates it (the reference's worker pool pre-loads its scene the same way,
 * TB/renderer/panda3d_batch_renderer.py:129-142).  hp_mesh_store_scratch_generation counts the reallocations: a holder
 * of captured hipGraphs that launch hp_rasterize compares it before every replay. */
int hp_mesh_store_reserve_raster(hp_mesh_store* store, int n_views, int h, int w, int flags);
int64_t hp_mesh_store_scratch_generation(const hp_mesh_store* store);

/* ------------------------------------------------------------------------------------
 * Rasteriser.  Replaces Panda3dBatchRenderer.render(labels, TCO, K, light_datas,
 * resolution, render_normals, render_depth, render_binary_mask) -> BatchRenderOutput
 * (TB/renderer/panda3d_batch_renderer.py:194-286; worker :62-125; scene renderer
 * TB/renderer/panda3d_scene_renderer.py:320-390; camera model TB/renderer/types.py:92-137;
 * depth decode / normal code TB/renderer/utils.py:46-79).
 *
 * One mesh per view, pinhole camera K, pose TCO, clip range [0.1, 10] m, two-sided,
 * black background.  Outputs (each may be NULL = not rendered):
 *   rgb   3 ch f32 in [0,1]   albedo * (ambient + Lambert point lights), 8-bit quantised
 *   nrm   3 ch f32 in [0,1]   eye-space normal colour code
 *   depth 1 ch f32 metres, 0 = background (or normalised, see below)
 *   mask  u8 [n][h][w]        depth > 0   (requires depth, as the reference asserts)
 * Addressing of the float outputs (element strides), view = 0..n-1:
 *   off(view, c, row, col) = (view / views_per_item) * s_item + (view % views_per_item) * s_view
 *                            + c * s_chan + row * s_row + col * s_col
 * so both the reference's NCHW BatchRenderOutput tensors and channel slices of the
 * NHWC network input are expressible.  depth uses (ds_item, ds_view, ds_row, ds_col).
 * d_depth_norm_z (optional, [n / views_per_item]): fuses normalize_images
 * (MP/models/pose_rigid.py:455-544) into the epilogue; depth_norm_mode selects
 * 0 none, 1 tCR_scale (d/z), 2 tCR_scale_clamp_center (clamp(d/z,0,2)-1),
 * 3 tCR_center_clamp (clamp(d-z,-2,2)).
 * ---------------------------------------------------------------------------------- */
#define HP_RASTER_QUANT8 8
/* d_rgb / d_nrm / d_depth point at fp16 tensors (strides in fp16 elements): views rendered straight into
 * the input of an fp16 network plan (hp_net_forward_f16in).  The mask stays uint8. */
#define HP_RASTER_OUT_F16 16
/* 4x multisampled colour / normal buffers, the framebuffer state of the reference's renderer
 * (TB/renderer/panda3d_scene_renderer.py:70-71 "framebuffer-multisample 1 / multisamples 4"): coverage and depth per
 * sample (standard 4x pattern), one shading per pixel and triangle at the pixel centre, 8-bit resolve = mean of the four
 * samples.  Depth and mask stay sampled at the pixel centre.  Ignored by depth-only renders. */
#define HP_RASTER_MSAA4 32
/* Texture filtering of the reference's renderer (TB/renderer/panda3d_scene_renderer.py:68-69 "texture-minfilter mipmap",
 * "texture-anisotropic-degree 16"): trilinear over the mip chain the store keeps behind level 0 (obj row entry 7 = number
 * of levels) + up to 16 probes along the major axis of the pixel footprint (EXT_texture_filter_anisotropic's sketch).
 * Default (flag clear): bilinear on level 0. */
#define HP_RASTER_TEX_ANISO 64

typedef struct {
  int64_t s_item, s_view, s_chan, s_row, s_col;
} hp_strides;

/* The three conventions of the reference's renderer that OpenGL / Panda3D leave to the implementation and that cannot be
 * pinned without Panda3D: where the four multisample positions sit, how the anisotropic filter
 * derives its probe count and level of detail from the pixel footprint, and which eye-space axis (and sign) each
 * channel of the normal-code render shows (TB/renderer/panda3d_scene_renderer.py:68-71,221-230, TB/renderer/utils.py:63-79).
 * They are a RECORD, not compile-time constants, so that an owner of a Panda3D installation can fit them:
 * tools/calibrate_renderer.py scores candidate records against Panda3D renders of the reference's own test scene
 * (tests/test_batch_renderer_panda3d.py:43-69); oracle/csrc/oracle.c mirrors the record (hp_oracle_set_raster_conventions).
 * The record belongs to a MESH STORE (hp_mesh_store_set_raster_conventions; NULL restores the defaults below) and is read at
 * launch time: two stores -- two renderers -- in one process may differ; launches already enqueued keep the values they were
 * launched with; captured hipGraphs must be re-captured after a change.  Sample positions are used on a 1/256-pixel grid
 * (rounded to nearest; hardware keeps them on such a grid -- D3D: 1/16 --, and so does the oracle). */
typedef struct {
  float msaa_x[4], msaa_y[4];  /* sample positions inside the pixel, each in (0, 1).  Default: the standard 4x pattern
                                  (0.375, 0.125) (0.875, 0.375) (0.125, 0.625) (0.625, 0.875) */
  int aniso_max;               /* cap of the probe count, 1..16 ("texture-anisotropic-degree").  Default 16 */
  int aniso_round;             /* probes from the footprint ratio r = Pmax / Pmin: 0 ceil(r) (default, the extension's sketch),
                                  1 nearest integer (ties to even), 2 floor(r) */
  int lod_from;                /* level of detail = log2 of: 0 Pmax / N (default), 1 Pmin, 2 Pmax (no anisotropic compensation) */
  float lod_bias;              /* added to the level of detail before clamping.  Default 0 */
  float aniso_ratio_bias;      /* added to r before it is rounded to the probe count.  Default 0 */
  int normal_axis[3];          /* channel c of the normal-code render shows component normal_axis[c] of the unit normal in the
                                  CAMERA frame of this library (OpenCV: x right, y down, z forward) ... */
  float normal_sign[3];        /* ... times normal_sign[c] (+1 / -1).  Default axes {0, 1, 2}, signs {+1, -1, -1}: GL eye space
                                  (x right, y up, z backward) as R, G, B */
} hp_raster_conventions;
int hp_mesh_store_set_raster_conventions(hp_mesh_store* store, const hp_raster_conventions* conventions /* NULL = defaults */);
int hp_mesh_store_get_raster_conventions(const hp_mesh_store* store, hp_raster_conventions* out);
/* Back-face culling in the set-up pass of this store's renders (read at launch time; default on, HP_RASTER_NO_CULL=1 creates
 * stores with it off).  The reference renders two-sided (TB/renderer/panda3d_scene_renderer.py:102).  A triangle whose
 * inward side is turned to the camera is dropped only when it cannot be seen: it belongs to a CONNECTED COMPONENT of the
 * (position-welded) mesh that is a closed, consistently oriented surface with a non-zero signed volume -- decided per
 * component at hp_mesh_store_create, so nested shells, parts with flipped winding and open sheets each get their own
 * answer --, the camera is outside the object's bounding sphere and the sphere lies beyond the near plane.  The facing test
 * is the exact sign of the triangle's area on the 1/256-px vertex grid.  Pixels may differ from the two-sided render only
 * where a sample lies exactly on a silhouette edge.  ASSUMPTION: a closed component does not intersect itself.  The edge test
 * and the net signed volume cannot see a self-intersecting component with a lobe of the opposite winding: its visible faces
 * would be culled (holes in rgb / depth / mask where the reference, two-sided, has none).  For object sets that may hold such
 * meshes switch culling off on the store -- hp_mesh_store_set_backface_culling(store, 0) right after hp_mesh_store_create, or
 * HP_RASTER_NO_CULL=1 for every store of the process; renders are then two-sided everywhere, ~25 % slower.
 * Returns the previous setting (-1: null store). */
int hp_mesh_store_set_backface_culling(hp_mesh_store* store, int on);
/* The current setting (1 / 0; -1: null store): what a lane's store copies from the store it was cloned from. */
int hp_mesh_store_get_backface_culling(const hp_mesh_store* store);

int hp_rasterize(const hp_mesh_store* store, int n, int views_per_item,
                 const int32_t* d_obj_ids /* [n / views_per_item] */,
                 const float* d_TCO /* [n][16] */, const float* d_K /* [n][9] */,
                 const float* d_ambient /* [n][3] or NULL = (1,1,1) */, int n_lights,
                 const float* d_light_pos /* [n][n_lights][3] object frame */,
                 const float* d_light_col /* [n][n_lights][3] */, int h, int w, int flags,
                 float* d_rgb, float* d_nrm, const hp_strides* color_strides,
                 float* d_depth, const hp_strides* depth_strides, uint8_t* d_mask,
                 const float* d_depth_norm_z, int depth_norm_mode, void* stream);

/* ------------------------------------------------------------------------------------
 * Network input of one refiner / scoring iteration in ONE pass: render the views of every hypothesis AND crop the
 * observation, each pixel record of the NHWC input written once.  Replaces, per iteration,
 *   crop_inputs -> crop_images (torchvision roi_align)       MP/models/pose_rigid.py:199-277, TB/lib3d/cropping.py:155-197
 *   render_images_multiview -> Panda3dBatchRenderer.render    MP/models/pose_rigid.py:376-453
 *   normalize_images + torch.cat((images_crop, renders))      MP/models/pose_rigid.py:455-544,624-629
 * (CosyPose: CP/models/pose.py:58-93,129-157).  Same arithmetic as hp_crop_roi_align followed by hp_rasterize into channel
 * slices; what changes is the memory traffic: no kernel touches a 32-B sector another kernel (or another workgroup) also
 * writes -- rocprofv3 counted 3x the algorithmic bytes for the two-launch form (profiles/r03a_raster_hbm_traffic.json).
 *
 * d_x: [n_items][h][w][record_elems] fp32, or fp16 with HP_RASTER_OUT_F16.  View v of an item writes
 *   its render channels -- rgb, then normals (HP_RENDER_NORMALS), then depth (HP_RENDER_DEPTH, normalised per
 *   depth_norm_mode as in hp_rasterize) -- at elements [view_c0[v], ...) of every pixel record, and
 *   crop_n[v] channels of the observed crop, source channels [crop_src0[v], crop_src0[v] + crop_n[v]) of the frame
 *   d_images[d_im_ids[item]] resampled from d_boxes[item] with torchvision's roi_align (sampling_ratio^2 samples,
 *   aligned = False; source channel 3 = depth: validity rule of TB/lib3d/cropping.py:184-195 and the same
 *   normalisation), at elements [crop_c0[v], ...).
 * The reference's channel order is {crop_n = {C_img, 0, ...}, crop_c0 = {0}, view_c0[v] = C_img + v * C_r}; any other
 * assignment (e.g. view v = one 32-B sector holding its 7 channels + crop channel v) needs the network's first layer
 * permuted accordingly.  Elements of a record nobody is assigned keep their value (the pads of a zeroed input stay 0).
 * ---------------------------------------------------------------------------------- */
#define HP_RENDER_NORMALS 0x1000
#define HP_RENDER_DEPTH 0x2000

typedef struct {
  int view_c0[8];   /* first record element of view v's render channels */
  int crop_c0[8];   /* first record element of the crop channels view v produces */
  int crop_src0[8]; /* first source channel of those */
  int crop_n[8];    /* how many (0 = this view produces no crop channel) */
} hp_input_layout;

int hp_render_inputs(const hp_mesh_store* store, int n_items, int views_per_item, const int32_t* d_obj_ids /* [n_items] */,
                     const float* d_TCV_O /* [n_items][V][16] */, const float* d_KV /* [n_items][V][9] */,
                     const float* d_ambient /* [n_items * V][3] or NULL */, int n_lights,
                     const float* d_light_pos /* [n_items * V][n_lights][3] */, const float* d_light_col, int h, int w,
                     int flags /* HP_RASTER_* | HP_RENDER_* */, const float* d_images /* [Bi][Ct][H][W] */, int Bi, int Ct,
                     int H, int W, const float* d_boxes /* [n_items][4] xyxy */, const int32_t* d_im_ids /* [n_items] */,
                     int sampling_ratio, const float* d_depth_norm_z /* [n_items] */, int depth_norm_mode, void* d_x,
                     int record_elems, const hp_input_layout* layout, void* stream);

/* ------------------------------------------------------------------------------------
 * Iteration prologue ("pose prep"), one launch for all hypotheses and views:
 *   [normalize_T]  TB/lib3d/transform_ops.py:107-120   (MP/models/pose_rigid.py:571)
 *   tCR            MP/models/pose_rigid.py:574-576     (tOR = 0 -> tCR = tCO)
 *   TCV_O          make_TCO_multiview, TB/lib3d/multiview.py:166-251 (:28-92)
 *   per view: project_points_robust -> boxes_from_uv -> deepim_boxes -> get_K_crop_resize
 *                  MP/models/pose_rigid.py:199-337 (crop_inputs, compute_crops_multiview),
 *                  CP/models/pose.py:58-93; TB/lib3d/camera_geometry.py:40-122;
 *                  TB/lib3d/cropping.py:27-75,113-152
 * View 0 uses n_points_main sub-sampled mesh points (2000), the extra views
 * n_points_extra (200); point ids are the deterministic RandomState(0) lists
 * (TB/lib3d/mesh_ops.py:74-84) computed once on the host.
 * multiview_type: 0 = single view (TCV_O = TCO), 1 = "TCO+front_1view",
 * 3 = "TCO+front_3views", 5 = "TCO+front_5views"; n_views must be 1 / 2 / 4 / 6.
 * Outputs: d_TCO_out [b][16] (normalised input pose), d_tCR [b][3], d_TCV_O [b][V][16],
 * d_boxes_rend [b][4], d_boxes_crop [b][4], d_K_crop [b][V][9] (view 0 = K_crop).
 * Index convention of the whole library: intrinsics and frames are tables [n_images] indexed by d_im_ids, never
 * per-hypothesis copies.  Ids live on the device, so they cannot be asserted here the way the reference's indexing
 * raises; a hypothesis whose image or object id lies outside its table reads row 0 and gets NaN outputs (NaN poses
 * render as zero images, an out-of-range image id crops to zeros) -- out-of-bounds memory is never touched.
 * ---------------------------------------------------------------------------------- */
int hp_pose_prep(const hp_mesh_store* store, int b, int n_views, int multiview_type,
                 int normalize, const float* d_TCO_in, const float* d_K /* [n_images][9] */, int n_images,
                 const int32_t* d_im_ids /* [b], values in [0, n_images) */, const int32_t* d_obj_ids /* [b] */,
                 const int32_t* d_point_ids_main, int n_points_main,
                 const int32_t* d_point_ids_extra, int n_points_extra, int im_h, int im_w,
                 int crop_h, int crop_w, float lamb, float* d_TCO_out, float* d_tCR,
                 float* d_TCV_O, float* d_boxes_rend, float* d_boxes_crop, float* d_K_crop,
                 void* stream);
/* The same with `remove_TCO_rendering` (TB/lib3d/multiview.py:189-236, MP/models/pose_rigid.py:609-611): the TCO view
 * itself is not rendered.  n_views counts the RENDERED views (multiview_type 3 -> 3, 5 -> 5; >= 2); d_TCV_O / d_K_crop hold
 * the look-at views only, each with the intrinsics of its own 200-point crop (compute_crops_multiview), and the K of the
 * observed crop (crop_inputs; what hp_pose_update needs) goes to d_K_crop_main [b][9].  remove_tco_rendering = 0 behaves
 * like hp_pose_prep (d_K_crop_main optional: a copy of view 0's K). */
int hp_pose_prep_views(const hp_mesh_store* store, int b, int n_views, int multiview_type, int remove_tco_rendering,
                       int normalize, const float* d_TCO_in, const float* d_K, int n_images,
                       const int32_t* d_im_ids, const int32_t* d_obj_ids,
                       const int32_t* d_point_ids_main, int n_points_main,
                       const int32_t* d_point_ids_extra, int n_points_extra, int im_h, int im_w,
                       int crop_h, int crop_w, float lamb, float* d_TCO_out, float* d_tCR,
                       float* d_TCV_O, float* d_boxes_rend, float* d_boxes_crop, float* d_K_crop,
                       float* d_K_crop_main, void* stream);

/* ------------------------------------------------------------------------------------
 * Crop.  Replaces crop_images / torchvision.ops.roi_align(images, [k,x1,y1,x2,y2],
 * (240,320), sampling_ratio=4, aligned=False)  (TB/lib3d/cropping.py:155-197,
 * CP/lib3d/cropping.py:129-134) incl. the RGB-D rule (depth zeroed where the roi-aligned
 * validity mask < 0.99) and, optionally, the depth normalisation of normalize_images.
 * d_images: [Bi][C][H][W] f32 (the reference's ObservationTensor layout); the first
 * n_channels (3 = rgb, 4 = rgbd) planes are cropped (a model without input_depth drops the
 * depth plane of an RGB-D observation, MP/models/pose_rigid.py:557-559).  Output
 * addressing as in hp_rasterize with views_per_item = 1 (s_view unused).
 * depth_norm_mode may carry HP_CROP_FULL_RECORD8: the destination is an NHWC tensor whose pixel records are
 * multiples of 32 B (8 floats / 16 halves; 32-B aligned) and the crop owns the first 32 B of each: its 3 / 4 channels AND zeros for the
 * rest of that 32-B sector are written as one full-sector store (the rasteriser writes its channels afterwards): a
 * partial-sector store costs the memory system a read-modify-write.
 * ---------------------------------------------------------------------------------- */
#define HP_CROP_FULL_RECORD8 0x100
int hp_crop_roi_align(const float* d_images, int Bi, int C, int n_channels, int H, int W,
                      const float* d_boxes /* [n][4] */, const int32_t* d_im_ids /* [n] */,
                      int n, int out_h, int out_w, int sampling_ratio, float* d_out,
                      const hp_strides* out_strides, const float* d_depth_norm_z,
                      int depth_norm_mode, void* stream);
/* The same with an fp16 destination (strides in fp16 elements): the crop of an fp16 network plan
 * (hp_net_set_precision) written straight into the tensor hp_net_forward_f16in reads. */
int hp_crop_roi_align_f16(const float* d_images, int Bi, int C, int n_channels, int H, int W,
                          const float* d_boxes, const int32_t* d_im_ids, int n, int out_h, int out_w,
                          int sampling_ratio, void* d_out_f16, const hp_strides* out_strides,
                          const float* d_depth_norm_z, int depth_norm_mode, void* stream);

/* ------------------------------------------------------------------------------------
 * Pose update.  Replaces PosePredictor.update_pose (MP/models/pose_rigid.py:339-350 ->
 * TB/lib3d/rotations.py:22-36 + TB/lib3d/cosypose_ops.py:34-62) and CosyPose's
 * apply_imagespace_predictions (CP/lib3d/cosypose_ops.py:18-42; d_tCR = NULL).
 * d_K_crop is [b][k_stride floats] (k_stride = 9 * n_views when taken from hp_pose_prep).
 * ---------------------------------------------------------------------------------- */
int hp_pose_update(int b, const float* d_TCO, const float* d_K_crop, int k_stride,
                   const float* d_pose9 /* [b][9] */, const float* d_tCR /* [b][3] or NULL */,
                   float* d_TCO_out, void* stream);

/* Coarse initialisation.  Replaces TCO_init_from_boxes_autodepth_with_R
 * (TB/lib3d/cosypose_ops.py:184-238; d_R != NULL), TCO_init_from_boxes_zup_autodepth
 * (:241-283; d_R = NULL).  Extents are taken over the full padded point set of the object
 * (MegaPose, MP/inference/pose_estimator.py:393-410) or, when d_point_ids is given, over
 * that deterministic sub-sample (CosyPose, CP/integrated/pose_estimator.py:128-130).
 * Hypothesis i uses box d_boxes[d_box_ids ? d_box_ids[i] : i], intrinsics
 * d_K[d_im_ids[i]], object d_obj_ids[i], rotation d_R[d_rot_ids ? d_rot_ids[i] : i]; an id outside its table
 * (sizes n_boxes / n_images / n_rots / objects of the store) gives a NaN pose, see hp_pose_prep. */
int hp_tco_init_autodepth(const hp_mesh_store* store, int n, const float* d_boxes /* [n_boxes][4] */, int n_boxes,
                          const int32_t* d_box_ids, const float* d_K /* [n_images][9] */, int n_images,
                          const int32_t* d_im_ids, const int32_t* d_obj_ids,
                          const float* d_R /* [n_rots][9] */, int n_rots, const int32_t* d_rot_ids,
                          const int32_t* d_point_ids, int n_points, float* d_TCO_out, void* stream);

/* ------------------------------------------------------------------------------------
 * Network (backbone + heads).  Replaces PosePredictor.net_forward
 * (MP/models/pose_rigid.py:352-374, CP/models/pose.py:108-114) for the backbones of
 * MP/training/pose_models_cfg.py:106-122 / CP/training/pose_models_cfg.py:39-42:
 *   HP_ARCH_VANILLA_RESNET34  MP/models/torchvision_resnet.py:191-344 (num_classes=512)
 *   HP_ARCH_WIDE_RESNET34/18  MP/models/wide_resnet.py:68-154 == CP/models/wide_resnet.py
 *   HP_ARCH_EFFICIENTNET_B3   CP/models/efficientnet.py:153-277 (extract_features; BN eps 1e-3,
 *                             static "same" padding for image_size 300), features [batch][1536]
 * Parameters are handed over by their reference state_dict names ("backbone.conv1.weight",
 * "backbone.layer1.0.bn1.running_var", "pose_fc.weight", "views_logits_head.bias", ...;
 * legacy names of TB/utils/models_compat.py are translated by the host side), fp32 host
 * arrays in PyTorch layout ([Cout][Cin][kh][kw]).  hp_net_finalize folds eval-mode
 * BatchNorm (eps 1e-5) and repacks to the kernels' layout.
 * Input: d_x NHWC [batch][h][w][c_pad], c_pad = hp_net_input_channels_padded(), pad
 * channels zero.  Outputs (each may be NULL): d_pose [batch][pose_dim],
 * d_logits [batch][n_logits], d_features [batch][512] (1536 for EfficientNet-b3).
 * ---------------------------------------------------------------------------------- */
#define HP_ARCH_VANILLA_RESNET34 0
#define HP_ARCH_WIDE_RESNET34 1
#define HP_ARCH_WIDE_RESNET18 2
#define HP_ARCH_EFFICIENTNET_B3 3 /* CP/models/efficientnet.py (CosyPose's released checkpoints): features [b,1536] */
/* ResNet-50 + FPN, the backbone of the Mask-RCNN detector (MP/models/mask_rcnn.py:22-42 -> torchvision
 * resnet_fpn_backbone("resnet50")); parameters by the names DetectorMaskRCNN registers ("backbone.body.layer1.0.conv1.weight",
 * "backbone.fpn.inner_blocks.0.0.weight", ...).  No heads: hp_net_forward(net, x, batch <= max_batch, NULL, NULL, NULL) fills
 * the five pyramid levels, read back with hp_net_feature_map. */
#define HP_ARCH_RESNET50_FPN 4

/* A feed-forward graph of convolutions the CALLER describes (the detector's RoI heads: fully connected layers are 7x7 /
 * 1x1 convolutions on [n][7][7][256], the mask head 3x3 convolutions on [n][14][14][256]).  hp_net_create(HP_ARCH_CUSTOM,
 * c_in, h, w), then hp_net_add_conv per layer in execution order (arena slots 0..31; in_slot -1 = the network input;
 * weight [cout][cin][k][k] and optional bias [cout] by parameter name; act 0 none / 1 ReLU; res_slot >= 0 adds that slot
 * before the activation), hp_net_add_output for every slot to read back, hp_net_set_param, hp_net_finalize, hp_net_forward
 * (heads NULL), hp_net_copy_feature_map.  Output channel counts are rounded up to 4 (zero rows). */
#define HP_ARCH_CUSTOM 5

typedef struct hp_net hp_net;

hp_net* hp_net_create(int arch, int n_inputs, int h, int w);
void hp_net_destroy(hp_net* net);
int hp_net_add_conv(hp_net* net, const char* weight_name, const char* bias_name, int cin, int cout, int k, int stride, int pad,
                    int act, int H, int W, int in_slot, int out_slot, int res_slot);
int hp_net_add_output(hp_net* net, int slot, int H, int W, int C);
int hp_net_input_channels_padded(const hp_net* net);
int hp_net_set_param(hp_net* net, const char* name, const float* h_data, int64_t numel);
/* Arithmetic of the conv stack, to be chosen before hp_net_finalize:
 *   HP_PRECISION_F32 (default): fp32 MFMA, the reference's precision;
 *   HP_PRECISION_F16: weights / activations rounded to fp16 once per layer, fp32 accumulation
 *     (v_mfma_f32_32x32x16_f16) -- configuration C5 of SURVEY.md 8; the reference has no fp16
 *     path, the tolerance is stated in DESIGN.md.  hp_net_forward still takes / returns fp32. */
#define HP_PRECISION_F32 0
#define HP_PRECISION_F16 1
int hp_net_set_precision(hp_net* net, int precision);
int hp_net_precision(const hp_net* net);
int hp_net_finalize(hp_net* net, int max_batch);
/* Widths of the three outputs of a finalized network: pose_dim / n_logits are 0 when the checkpoint has no such head (what
 * PosePredictor.net_forward returns, MP/models/pose_rigid.py:352-374); the compiled operator library sizes its outputs with it. */
int hp_net_output_dims(const hp_net* net, int* pose_dim, int* n_logits, int* n_features);
/* The planned input map of a network: hp_net_forward strides its input by h * w * c_pad floats per sample (c_pad = channels
 * rounded up to 4; the fp16 plan's record width is hp_net_input_channels_f16), so a caller -- the compiled operator library's
 * net_forward and its Meta kernel -- checks a tensor's height and width against these before handing over the pointer
 * (the reference's backbone takes any H x W, MP/models/pose_rigid.py:352-374; a plan is built for one).  device = the HIP
 * device that was current in hp_net_create (weights and arena live there). */
int hp_net_input_dims(const hp_net* net, int* h, int* w, int* c_pad, int* device);
int hp_net_forward(hp_net* net, const float* d_x, int batch, float* d_pose, float* d_logits,
                   float* d_features, void* stream);
/* Feature-pyramid networks (HP_ARCH_RESNET50_FPN): number of output maps ('0', '1', '2', '3', 'pool' of torchvision's
 * BackboneWithFPN) and the map itself -- NHWC [batch][h][w][c] fp32 in the network's arena, valid until the next forward. */
int hp_net_n_feature_maps(const hp_net* net);
int hp_net_feature_map(const hp_net* net, int index, const float** d_ptr, int* h, int* w, int* c);
/* ... or copied (device to device, asynchronously on `stream`) into caller-owned memory [batch][h][w][c].  For
 * HP_ARCH_RESNET50_FPN maps 0..4 are the pyramid levels, 5..9 the RPN objectness logits of those levels ([h][w][4], 3 anchors
 * + one padding channel) and 10..14 the RPN box deltas ([h][w][12]) (torchvision models/detection/rpn.py: RPNHead, keys
 * "rpn.head.conv.0.0.*", "rpn.head.cls_logits.*", "rpn.head.bbox_pred.*"). */
int hp_net_copy_feature_map(const hp_net* net, int index, int batch, float* d_dst, void* stream);
/* GeneralizedRCNNTransform.normalize of the detector (torchvision models/detection/transform.py: (image - mean) / std per
 * channel; MaskRCNN defaults mean (0.485, 0.456, 0.406), std (0.229, 0.224, 0.225)) fused with the layout change the conv
 * stack wants: d_images NCHW [n][3][h][w] in [0,1] (ObservationTensor.images[:, :3]) -> d_x NHWC [n][h][w][4], pad channel 0. */
int hp_detector_preprocess(const float* d_images, int n, int h, int w, const float* h_mean3, const float* h_std3,
                           float* d_x_nhwc4, void* stream);
/* The same with GeneralizedRCNNTransform.resize + batch_images in front (torchvision models/detection/transform.py:
 * _resize_image_and_masks = F.interpolate(bilinear, align_corners=False, recompute_scale_factor=True); batch_images pads
 * to a multiple of 32 with zeros): images [n,3,h_in,w_in] -> resized to [h_out,w_out], normalised, written into the
 * top-left of the zeroed canvas [n,h_pad,w_pad,4].  The caller computes the sizes (happypose_amd/detector.py). */
int hp_detector_preprocess_resize(const float* d_images, int n, int h_in, int w_in, int h_out, int w_out, int h_pad, int w_pad,
                                  const float* h_mean3, const float* h_std3, float* d_x_nhwc4, void* stream);
/* fp16 plan only: the input already in fp16, NHWC [batch][h][w][hp_net_input_channels_f16()] with the
 * channels past n_inputs zero (hp_crop_roi_align_f16 / hp_rasterize with HP_RASTER_OUT_F16 write it):
 * saves the fp32 -> fp16 conversion pass of hp_net_forward (5 % of a coarse-scoring step). */
int hp_net_input_channels_f16(const hp_net* net);
int hp_net_forward_f16in(hp_net* net, const void* d_x16, int batch, float* d_pose, float* d_logits,
                         float* d_features, void* stream);
/* total multiply-accumulate FLOPs (2*MAC) of one sample through conv + linear layers */
double hp_net_flops_per_sample(const hp_net* net);
/* Profiling of the dominant kernel: with hp_net_set_profiling(net, 1) every conv launch is
 * bracketed by a pair of HIP events recorded on the launch stream (no synchronisation).
 * hp_net_profile_collect waits for the recorded pairs and returns the summed kernel time
 * (ms), the number of launches, their ALGORITHMIC FLOPs (2 x M x Cout x kh x kw x Cin of the
 * direct convolution) and the FLOPs the matrix cores actually executed (less for the Winograd
 * layers, more where tiles / K are padded; in fp32-MFMA equivalents: an fp16 MFMA FLOP of the
 * split-fp16 layers counts 1/16, its share of matrix-pipe time) since the previous collect. */
int hp_net_set_profiling(hp_net* net, int enabled);
/* Parity tests / layer-level users: the kernel family hp_conv2d_nhwc / hp_conv2d_nhwc_f16 -- the SINGLE-LAYER entry points, which
 * have no network to carry a choice -- pick from.  Networks never read it (no process-wide state on a network's launch path:
 * hp_net_set_conv_algo below).  AUTO = for 3x3 layers the split-fp16 kernels (fp32 operands as two fp16 halves, three fp16 MFMAs
 * per product, fp32 accumulation: fp32-level accuracy while |activations| < 65504), else Winograd F(2x2,3x3), else the
 * patch-staged direct kernel, else the generic implicit GEMM; WINOGRAD = exact-fp32 arithmetic only; DIRECT = no Winograd either;
 * IGEMM = generic kernel only. */
#define HP_CONV_ALGO_AUTO 0
#define HP_CONV_ALGO_DIRECT 1
#define HP_CONV_ALGO_IGEMM 2
#define HP_CONV_ALGO_WINOGRAD_1WAVE 3 /* WINOGRAD, but the one-wave-per-SIMD schedule of the Winograd kernel */
#define HP_CONV_ALGO_WINOGRAD 4 /* exact-fp32 kernels only: Winograd, else patch-staged, else generic */
#define HP_CONV_ALGO_SPLIT 5 /* split-fp16 kernels (3 fp16 MFMAs per fp32 product) wherever they apply */
int hp_conv_select_algo(int algo);
/* The choice of ONE network (what the predictors and bench.py's exact-fp32 pass use; networks on different host threads /
 * streams do not interfere); algo = -1 returns the network to AUTO. */
int hp_net_set_conv_algo(hp_net* net, int algo);
/* The conv kernels cut the tiles of a partially filled last round along K so that one launch fills the GPU.  When
 * independent launches share the GPU (the two half-batch lanes of a predictor on two streams) the other stream fills
 * those CUs and the slicing only costs its reduction: 0 switches it off for this network, 1 back on (the default). */
int hp_net_set_tail_split(hp_net* net, int enabled);
/* Dynamic activation scale of the split-fp16 kernels (default on).  x = x_hi + x_lo in fp16 has an ABSOLUTE floor of 2^-25:
 * activations below ~0.1 lose relative bits (x_lo falls into the fp16 subnormals).  With the scale on, every launch tracks
 * max|y| of what it stores (one atomicMax per wave into a per-layer word) and a split-fp16 consumer multiplies what it
 * splits by the power of two that puts its input's bound at 2^13 -- exact, undone in the epilogue together with the
 * weights' scale; a layer whose activations sit at 1e-4 then keeps the 22 significant bits of the scheme, and the fp16
 * range cannot be left by a finite input.  0 restores the unscaled arithmetic (A/B, tests). */
int hp_net_set_act_scale(hp_net* net, int enabled);
/* Numerical guard of the default (split-fp16) kernels.  They carry fp32 activations through the fp16 matrix path as
 * hi/lo halves, which needs |activation| < 65504; beyond that a half becomes inf and the layer's output inf / NaN where
 * the reference's fp32 arithmetic stays finite.  Every split-fp16 launch reports a non-finite output to a host-visible
 * word of its network.  hp_net_status synchronises `stream`, returns the flags and clears HP_STATUS_NONFINITE:
 *   HP_STATUS_NONFINITE  a forward since the last call produced a non-finite value: ITS OUTPUTS ARE INVALID;
 *   HP_STATUS_EXACT_ONLY the network has switched to the exact-fp32 kernels (Winograd / direct; sticky): re-running
 *                        the same inputs now gives the reference's arithmetic.
 * hp_net_forward also reads the word (without synchronising) on entry, so once a completed forward has tripped the
 * guard every later forward runs on the exact kernels by itself. */
#define HP_STATUS_NONFINITE 1
#define HP_STATUS_EXACT_ONLY 2
int hp_net_status(hp_net* net, void* stream, int* flags);
/* Sets (1) or clears (0) the sticky exact-fp32 state the guard enters by itself.  Multi-rank callers use it to keep the
 * ranks uniform: when ANY rank's guard fired, every rank forces its networks exact before the stage is repeated, so that
 * the merged rows come from one arithmetic and no rank is left alone on the slower kernels (pose_estimator.py::_guarded;
 * no counterpart in the reference, whose ranks all run ATen's fp32 convolutions).  0 returns to the default kernels. */
int hp_net_force_exact(hp_net* net, int enabled);
/* diagnostics: workgroups per CU the runtime grants conv tile variant 0 (128x128) / 1 (128x64) */
int hp_conv_occupancy(int variant);
/* hipGraph safety: launches so far, in this process, of kernels whose code object uses scratch (private segment > 0:
 * register spills of a rarely used tile variant; counted on the host at launch time, so also while a stream captures,
 * never during a replay).  A captured graph that contains such a launch replays wrongly on this runtime; the host layer
 * (happypose_amd/pose_predictor.py::_run_refine, no counterpart in the reference) compares the count around a predictor's
 * eager call and keeps that predictor on eager launches when it moved.  Stays 0 on the benchmarked configurations. */
long long hp_scratch_launches(void);
int hp_net_profile_collect(hp_net* net, double* conv_ms, int64_t* n_launches, double* conv_flops,
                           double* mfma_flops);
/* Several networks on several streams (the two half-batch lanes of a predictor run concurrently, so their summed
 * kernel time exceeds the wall time): hp_profile_mark_reference records a process-wide reference event on `stream`;
 * hp_net_profile_intervals returns the number of timed stretches pending for `net` and writes the start / end of the
 * first `cap` of them in ms after the reference (call it BEFORE hp_net_profile_collect, which releases them).  The
 * caller takes the union over the networks: the time during which any conv kernel was running. */
int hp_profile_mark_reference(void* stream);
int hp_net_profile_intervals(hp_net* net, double* t0_ms, double* t1_ms, int cap);

/* Diagnostics: the fp16 matrix-pipe rate this GPU SUSTAINS (back-to-back v_mfma_f32_32x32x16_f16 on every SIMD, operands
 * in registers), chip-wide TFLOP/s and the shader clock it settles at.  gfx950 clocks to its power budget: zero operands
 * reach the dense peak (~2.5 PFLOP/s at ~2.4 GHz), random fp16 operands -- what activations and weights are -- about two
 * thirds of it at ~1.6 GHz.  bench.py reports the conv roofline against the nominal peak AND against this measurement. */
int hp_probe_mfma_rate(int random_data, double* tflops, double* shader_mhz, void* stream);

/* ------------------------------------------------------------------------------------
 * Depth refinement (run_depth_refiner=True): point-to-plane ICP between the depth rendered at
 * the predicted pose and the measured depth.  Replaces icp_refinement / ICPRefiner.refine_poses
 * (MP/inference/icp_refiner.py:135-303, masks of MP/inference/refiner_utils.py:27-53); the
 * registration itself is OpenCV's ppf_match_3d_ICP there and a projective point-to-plane ICP here
 * (parity unpinned, see csrc/icp.hip).  d_depth_rendered [n][H][W] (hp_rasterize at d_TCO),
 * d_depth_measured [B][H][W] metres, d_masks [B][H][W] u8 or NULL (= the "threshold" mask with
 * depth_delta_thresh), im_ids on device and host, d_K [n][9].  Outputs: d_TCO_out [n][16]
 * (= input pose where the registration is rejected), d_retval [n] (0 / -1), d_residual [n]
 * (RMS point-to-plane distance of the inliers, metres); the last two may be NULL.
 * ---------------------------------------------------------------------------------- */
int hp_icp_refine(int n, int B, int H, int W, const float* d_depth_rendered, const float* d_depth_measured,
                  const uint8_t* d_masks, const int32_t* d_im_ids, const int32_t* h_im_ids, const float* d_K,
                  const float* d_TCO, int n_iterations, int n_min_points, float tolerance,
                  float depth_delta_thresh, float* d_TCO_out, int32_t* d_retval, float* d_residual,
                  void* stream);

/* ------------------------------------------------------------------------------------
 * Detector stages that are not convolutions (the reference's Detector.get_detections, MP/inference/detector.py:34-156, runs
 * torchvision's MaskRCNN.forward; file references below are torchvision 0.14.1, pinned by the reference's pyproject.toml).
 * The dense networks are hp_net objects: HP_ARCH_RESNET50_FPN (backbone + FPN + RPN head) and two HP_ARCH_CUSTOM graphs
 * (box head, mask head).
 *  hp_rpn_decode: for the n anchors a pyramid level's top-k selected (d_anchor_idx = (y * map_w + x) * 3 + a, objectness
 *    logits gathered alongside): AnchorGenerator.grid_anchors (anchor_utils.py), BoxCoder.decode with weights (1,1,1,1) and
 *    the log(1000/16) clamp (_utils.py), clip_boxes_to_image, remove_small_boxes(min_size) as a validity flag, sigmoid
 *    (rpn.py: RegionProposalNetwork.filter_proposals).  d_deltas_map = that level's [h][map_w][12] map of ONE image.
 *  hp_nms: boxes sorted by decreasing score; h_keep[i] = 1 when box i survives greedy suppression (IoU > threshold) inside its
 *    group (ops/boxes.py: batched_nms).  Pairwise masks on the device, greedy pass on the host: SYNCHRONISES `stream`.
 *  hp_roi_align_levels: MultiScaleRoIAlign (ops/poolers.py): level k = floor(4 + log2(sqrt(area) / 224) + 1e-6) clamped to
 *    [k_min, k_min + n_levels), then roi_align(aligned=False) with spatial_scale h_scales[level] on that NHWC map
 *    [n_img][h][w][C]; d_rois [K][5] = image index, x1, y1, x2, y2 (image coordinates); d_out [K][out][out][C].
 *  hp_box_postprocess: softmax over n_classes + BoxCoder.decode with weights (10,10,5,5) per class + clip
 *    (roi_heads.py: postprocess_detections); d_scores [n][n_classes], d_boxes [n][n_classes][4].
 *  hp_paste_masks: maskrcnn_inference + paste_masks_in_image (roi_heads.py): sigmoid of channel d_labels[k] of the mask
 *    logits [n][14][14][2][2][ld] (the mask head's 2x2 deconvolution kept as four 1x1 convolutions), padding 1, box expanded
 *    and truncated to integers, bilinear resize (align_corners=False) to the box, pasted into d_out [n][H][W] (0 elsewhere).
 * ---------------------------------------------------------------------------------- */
int hp_rpn_decode(const float* d_objectness, const int32_t* d_anchor_idx, int n, const float* d_deltas_map, int map_w,
                  int n_anchors, const float* h_base_anchors /* [3][4] */, int stride_h, int stride_w, float im_h, float im_w,
                  float min_size, float* d_boxes, float* d_scores, uint8_t* d_valid, void* stream);
int hp_nms(const float* d_boxes, const int32_t* d_group, int n, float iou_threshold, uint8_t* h_keep, void* stream);
int hp_roi_align_levels(const float* const* h_feat_ptrs, const int* h_heights, const int* h_widths, const float* h_scales,
                        int n_levels, int k_min, int C, const float* d_rois, int K, int out_size, int sampling_ratio, float* d_out,
                        int32_t* d_levels /* [K] or NULL */, void* stream);
int hp_box_postprocess(const float* d_class_logits, int ld_logits, const float* d_box_regression, int ld_regression,
                       const float* d_proposals, int n, int n_classes, float im_h, float im_w, float* d_scores, float* d_boxes,
                       void* stream);
int hp_paste_masks(const float* d_mask_logits, int ld, const int32_t* d_labels, const float* d_boxes, int n, int H, int W,
                   float* d_out, void* stream);

/* Single layer entry (used by the parity tests of the conv kernels themselves):
 * y[n][ho][wo][cout] = act( conv(x_act, w) + bias + residual ),   act = relu: 0 none, 1 ReLU, 2 swish
 * x_act = x                                              (pre_scale == NULL)
 *       = relu(x * pre_scale[c] + pre_shift[c])           (both given: pre-activation BatchNorm, zero
 *                                                          padding AFTER it)
 *       = x * pre_scale[img][c]                           (pre_shift == NULL: squeeze-excitation gate
 *                                                          [n][cin] of an MBConv projection)
 * x NHWC [n][h][w][cin] (cin % 4 == 0), w packed [cout][kh][kw][cin] (cout % 4 == 0), stride 1|2,
 * symmetric padding.  The kernel family follows hp_conv_select_algo (Winograd / patch-staged direct
 * for 3x3 stride-1 layers with cin % 32 == 0 and cout % 64 == 0, the generic implicit GEMM else). */
int hp_conv2d_nhwc(const float* d_x, int n, int h, int w, int cin, const float* d_w, int cout,
                   int kh, int kw, int stride, int pad, const float* d_bias,
                   const float* d_residual, const float* d_pre_scale, const float* d_pre_shift,
                   int relu, float* d_y, void* stream);
/* the same for the fp16 kernel: x, w, residual, pre_scale / pre_shift and y are fp16 device
 * arrays (bias stays fp32); cin % 8 == 0, kh*kw*cin % 64 == 0, cout % 64 == 0 */
int hp_conv2d_nhwc_f16(const void* d_x, int n, int h, int w, int cin, const void* d_w, int cout, int kh,
                       int kw, int stride, int pad, const float* d_bias, const void* d_residual,
                       const void* d_pre_scale, const void* d_pre_shift, int relu, void* d_y,
                       void* stream);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* HAPPYPOSE_AMD_H */
