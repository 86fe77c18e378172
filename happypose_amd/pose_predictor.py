"""Render-and-compare pose predictors (the refiner / coarse models).

Host-side mirror of the reference's two ``PosePredictor`` classes -- same
constructor arguments, ``forward`` / ``forward_coarse`` signatures, output
dictionaries and error behaviour:

* MegaPose  ``MP/models/pose_rigid.py:96-788``   -> :class:`PosePredictor`
* CosyPose  ``CP/models/pose.py:33-199``         -> :class:`CosyPosePosePredictor`

Every iteration is five asynchronous HIP launches on the current stream, with
nothing leaving the device:

  hp_pose_prep     normalize_T, tCR, multi-view cameras, projected boxes, crop box, K_crop
  hp_crop_roi_align  observed crop  -> channels [0, C_img) of the NHWC network input
  hp_rasterize     V rendered views -> channels [C_img, C_img + V*C_r)
  hp_net_forward   BN-folded MFMA conv stack + heads
  hp_pose_update   9-D update -> TCO_output

The reference's per-iteration host work (pickling to renderer processes, numpy
``make_TCO_multiview``, ``clone``/``cat``) has no counterpart here.
"""

from __future__ import annotations

import os
import time
from collections import defaultdict
from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional, Sequence

import torch

from . import ops
from .renderer import BatchRenderer


@dataclass
class PosePredictorOutput:
    """``MP/models/pose_rigid.py:60-80``.  ``renders`` / ``images_crop`` (NCHW) are
    materialised only when the predictor runs with ``keep_pixels=True``."""

    TCO_output: torch.Tensor
    TCO_input: torch.Tensor
    renders: Optional[torch.Tensor]
    images_crop: Optional[torch.Tensor]
    TCV_O_input: torch.Tensor
    KV_crop: torch.Tensor
    tCR: torch.Tensor
    labels: List[str]
    K: torch.Tensor
    K_crop: torch.Tensor
    network_outputs: Dict[str, torch.Tensor]
    boxes_rend: torch.Tensor
    boxes_crop: torch.Tensor
    renderings_logits: Optional[torch.Tensor] = None
    timing_dict: Dict[str, float] = field(default_factory=dict)

    # the CosyPose flavour is consumed with item access in parts of the reference
    def __getitem__(self, k):
        return getattr(self, k)


class _Timer:
    """``SimpleTimer`` / ``CudaTimer`` (``MP/training/utils.py:218-277``)."""

    def __init__(self, device, cuda: bool):
        self.device, self.cuda = device, cuda
        self._t = 0.0

    def start(self):
        if self.cuda:
            self._e0 = torch.cuda.Event(enable_timing=True)
            self._e1 = torch.cuda.Event(enable_timing=True)
            self._e0.record(torch.cuda.current_stream(self.device))
        else:
            self._t0 = time.time()

    def stop(self):
        if self.cuda:
            self._e1.record(torch.cuda.current_stream(self.device))
            self._e1.synchronize()
            self._t = self._e0.elapsed_time(self._e1) / 1000.0
        else:
            self._t = time.time() - self._t0

    end = stop

    def elapsed(self) -> float:
        return self._t


class _RenderAndCompare:
    """Shared device pipeline of both predictors."""

    def _setup(self, backbone: ops.Net, renderer: BatchRenderer, mesh_db, render_size):
        assert isinstance(renderer, BatchRenderer), "renderer must be a happypose_amd BatchRenderer"
        assert isinstance(backbone, ops.Net), "backbone must be a happypose_amd.ops.Net"
        assert isinstance(backbone.n_features if hasattr(backbone, "n_features") else 512, int)
        self.backbone = backbone
        self.renderer = renderer
        self.store = renderer.store
        self.mesh_db = mesh_db if mesh_db is not None else self.store.mesh_db
        self.render_size = tuple(render_size)
        self.device = self.store.device
        self.keep_pixels = False
        self.debug = False
        # True: the observed crop is produced by the render launch itself (hp_render_inputs with the crop arguments)
        self.fuse_crop = os.environ.get("HP_FUSE_CROP", "") == "1"
        self._x: Optional[torch.Tensor] = None
        # hipGraph replay of forward() (happypose_amd.graphs): off unless asked for -- the first two calls of a
        # signature run eagerly / capture, and callers that time single launches want the eager path
        self.use_graphs = False
        self._graphs = None
        self._graph_epoch = None

    def _reserve(self, n_views_per_hypothesis: int):
        """Pre-size the store's rasteriser scratch for this predictor's largest call (``max_batch`` hypotheses x views):
        it then never reallocates in normal use, which keeps captured graphs valid and ``hipFree`` out of the loop."""
        mb = getattr(self.backbone, "max_batch", None)
        if mb:
            self.store.reserve_raster(int(mb) * n_views_per_hypothesis, self.render_size, msaa=self.renderer.msaa)

    def eval(self):
        return self

    def to(self, device):
        assert torch.device(device).type == self.device.type, "models live on the device of their MeshStore"
        return self

    def _input_buffer(self, b: int) -> torch.Tensor:
        if self._x is None or self._x.shape[0] < b:
            self._x = self.backbone.new_input(b)  # zeroed once: pad channels are never written
        return self._x[:b]

    def _ids(self, images, K, labels, im_ids):
        """The ONE index convention of the path: ``images [Bi,C,H,W]`` and ``K [Bi,3,3]`` are per-FRAME tables and
        hypothesis ``i`` uses row ``im_ids[i]`` of both.  ``im_ids=None`` is the reference's calling convention
        (images / K already gathered per hypothesis): ``Bi == bsz`` and ``im_ids = arange(bsz)``.  Ids handed over
        on the host are range-checked here (``IndexError``, like the reference's indexing); ids already on the
        device are guarded inside the kernels (``ops._check_ids``)."""
        bsz = len(labels)
        assert K.shape[0] == images.shape[0], "K must hold one intrinsics matrix per frame of `images`"
        if im_ids is None:
            assert images.shape[0] == bsz, "without im_ids, images and K must hold one row per hypothesis"
            im_ids = torch.arange(bsz, dtype=torch.int32, device=self.device)
        else:
            ops._check_ids(im_ids, images.shape[0], "im_ids -> images / K")
            im_ids = torch.as_tensor(im_ids).to(device=self.device, dtype=torch.int32)
            assert im_ids.shape == (bsz,)
        return im_ids, self.store.ids_of(labels)

    def _run_refine(self, consts, inputs, fn):
        """``fn(*inputs)`` eagerly, or through the graph cache when ``use_graphs`` is set and nothing in the call
        resists capture (kept pixels are large and change what is launched; profiling records events)."""
        if not self.use_graphs or self.keep_pixels or self.debug or self._profiling():
            return fn(*inputs)
        if getattr(self, "_no_graphs", False):
            return fn(*inputs)
        scratch0 = ops.scratch_launches()
        from .graphs import GraphCache

        # what invalidates captured launches: the process-wide launch-plan epoch and this store's scratch generation (a
        # larger eager call, or another predictor sharing the store, reallocated the rasteriser scratch the graphs point to)
        epoch = (ops.graph_epoch(), self.store.scratch_generation())
        if self._graphs is None or self._graph_epoch != epoch:
            self._graphs, self._graph_epoch = GraphCache(self.device), epoch
        # everything else the launch plan depends on is part of the signature: tail K-slicing differs between the
        # single-lane call (on) and a lane of TwoLanePredictor (off) on the same shapes
        consts = tuple(consts) + (("tail_split", bool(self.backbone.tail_split)), ("msaa", self.renderer.msaa), ("aniso", self.renderer.aniso))
        out = self._graphs.run(consts, inputs, fn, keepalive=self._graph_keepalive)
        if ops.scratch_launches() != scratch0:
            # this predictor launches a tile variant that spills registers to scratch (EfficientNet's narrow layers, odd
            # resolutions); a graph holding such a launch replays wrongly.  The count moves on the host at launch time, i.e.
            # during the eager first call of a signature: nothing of it has been captured yet -- stay eager from here on
            self._no_graphs, self._graphs = True, None
        if self.store.scratch_generation() != epoch[1]:
            # the eager first call of a signature larger than the reservation grew the scratch: every graph captured
            # before holds freed pointers -- start over (this call's result is valid: it ran eagerly)
            self._graphs = None
        return out

    def _profiling(self) -> bool:
        return bool(self.backbone.profiling)

    @property
    def max_batch(self) -> int:
        """Largest hypothesis batch one ``forward`` call takes (the backbone's activation arena)."""
        return int(self.backbone.max_batch)

    def _graph_keepalive(self):
        return [self._x]

    def numerics_status(self) -> int:
        """Guard flags of the backbone (``ops.Net.status``; synchronises the current stream): non-zero bit 0 means
        the outputs of a forward since the last call are invalid (an activation left the fp16 range of the default
        split-fp16 conv kernels); the backbone has then switched to the exact-fp32 kernels and the call can simply
        be repeated.  The estimators do that at their existing synchronisation points."""
        return self.backbone.status()

    def _one_pass(self, images, K, im_ids, obj_ids, TCO_in, *, n_img_channels, multiview_type, normalize,
                  render_normals, render_depth, depth_mode, want_pose, want_logits, remove_TCO_rendering=False,
                  scene_lights=False):
        b = TCO_in.shape[0]
        prep = ops.pose_prep(self.store, TCO_in, K, im_ids, obj_ids, tuple(images.shape[-2:]),
                             self.render_size, multiview_type=multiview_type, normalize=normalize,
                             remove_TCO_rendering=remove_TCO_rendering)
        x = self._input_buffer(b)
        z = prep["tCR"][:, 2].contiguous() if depth_mode else None
        t0 = time.time()
        lights = {}
        if scene_lights:  # the reference's make_scene_lights() per view: ambient 0.1 + six point lights around the object
            amb, pos, col = self.renderer.scene_light_tables()
            V = prep["TCV_O"].shape[1]
            ov = obj_ids.long().repeat_interleave(V)
            lights = dict(ambient=amb[None].expand(b * V, 3).contiguous(), light_pos=pos[ov].contiguous(),
                          light_col=col[None].expand(b * V, -1, 3).contiguous())
        V = prep["TCV_O"].shape[1]
        fuse = bool(self.fuse_crop)
        if fuse:
            # crop of the observation + the rendered views in ONE launch, every pixel record of x written once (opt-in:
            # ``fuse_crop = True`` / HP_FUSE_CROP=1).  Built to remove the partial-sector traffic of two writers; measured
            # in the pipeline it is a wash for MegaPose (2261 vs 2270 poses/s) and slower for one-view models (C2 4563
            # vs 4753, C5 37.9 k vs 38.9 k views/s): every workgroup then carries the crop's taps on top of a VALU-bound
            # shading pass, so the stand-alone crop launch in front stays the default
            ops.render_inputs(self.store, x, obj_ids, prep["TCV_O"], prep["K_crop"], render_normals, render_depth,
                              images=images, boxes=prep["boxes_crop"], im_ids=im_ids, n_img_channels=n_img_channels,
                              depth_norm_z=z, depth_norm_mode=depth_mode, msaa=self.renderer.msaa, aniso=self.renderer.aniso, **lights)
        else:
            ops.crop_roi_align(images, prep["boxes_crop"], im_ids, self.render_size, out=x, depth_norm_z=z,
                               depth_norm_mode=depth_mode if n_img_channels == 4 else 0, n_channels=n_img_channels, owns_record=True)
            ops.render_inputs(self.store, x, obj_ids, prep["TCV_O"], prep["K_crop"], render_normals, render_depth,
                              chan0=n_img_channels, depth_norm_z=z, depth_norm_mode=depth_mode, msaa=self.renderer.msaa,
                              aniso=self.renderer.aniso, **lights)
        render_time = time.time() - t0
        pose, logits, _ = self.backbone.forward(x, want_pose=want_pose, want_logits=want_logits)
        return prep, x, pose, logits, render_time

    def _pixels(self, x, n_img_channels, n_render_channels):
        if not self.keep_pixels:
            return None, None
        nchw = x.permute(0, 3, 1, 2).float()  # fp16 when the backbone is planned in fp16
        return (nchw[:, :n_img_channels].contiguous(),
                nchw[:, n_img_channels:n_img_channels + n_render_channels].contiguous())


class PosePredictor(_RenderAndCompare):
    """MegaPose predictor, ``MP/models/pose_rigid.py:96-788``."""

    def __init__(self, backbone: ops.Net, renderer: BatchRenderer, mesh_db=None, render_size=(240, 320),
                 multiview_type: str = "front_3views", views_inplane_rotations: bool = False,
                 remove_TCO_rendering: bool = False, predict_pose_update: bool = True,
                 predict_rendered_views_logits: bool = False, render_normals: bool = True,
                 n_rendered_views: int = 1, input_depth: bool = False, render_depth: bool = False,
                 depth_normalization_type: Optional[str] = None):
        self._setup(backbone, renderer, mesh_db, render_size)
        # views_inplane_rotations only reaches make_TCO_multiview from the training loss
        # (MP/training/megapose_forward_loss.py:122); forward_refiner / forward_coarse never pass it
        # (MP/models/pose_rigid.py:578-584), so at inference the flag is stored and has no effect -- same here.
        self.views_inplane_rotations = views_inplane_rotations
        # remove_TCO_rendering: forward_refiner renders the look-at views only (MP/models/pose_rigid.py:578-611; with one
        # rendered view make_TCO_multiview short-cuts to the TCO view and only the K of the render changes -- not built);
        # forward_coarse renders the hypothesis itself either way (:483-532)
        if remove_TCO_rendering and predict_pose_update and n_rendered_views < 2:
            raise NotImplementedError("refiner with remove_TCO_rendering and a single rendered view")
        # legacy names (MP/training/pose_models_cfg.py:48-53)
        multiview_type = {"front_3views": "TCO+front_3views", "front_5views": "TCO+front_5views",
                          "front_1view": "TCO+front_1view"}.get(multiview_type, multiview_type)
        self.n_rendered_views = n_rendered_views
        self.multiview_type = multiview_type if n_rendered_views > 1 else "TCO"
        self._skip_tco = bool(remove_TCO_rendering and predict_pose_update and n_rendered_views > 1)
        if self.multiview_type not in ops.MULTIVIEW or ops.MULTIVIEW[self.multiview_type][1] != n_rendered_views + int(self._skip_tco):
            raise ValueError(multiview_type)
        self.input_depth = input_depth
        self.render_normals = render_normals
        self.render_depth = render_depth
        self.depth_normalization_type = depth_normalization_type
        if (input_depth or render_depth) and depth_normalization_type not in ops.DEPTH_NORM:
            raise ValueError(f"Unknown depth_normalization_type = {depth_normalization_type}")
        self._depth_mode = ops.DEPTH_NORM.get(depth_normalization_type, 0) if (input_depth or render_depth) else 0
        self.predict_pose_update = predict_pose_update
        self.predict_rendered_views_logits = predict_rendered_views_logits
        self.remove_TCO_rendering = remove_TCO_rendering
        self._n_img = 4 if input_depth else 3
        self._reserve(n_rendered_views)
        self._n_single_render_channels = 3 + (3 if render_normals else 0) + (1 if render_depth else 0)
        n_inputs = self._n_img + self._n_single_render_channels * n_rendered_views
        assert backbone.n_inputs == n_inputs, (
            f"backbone expects {backbone.n_inputs} input channels, configuration needs {n_inputs}")
        if predict_pose_update:
            assert backbone.pose_dim == 9
        if predict_rendered_views_logits:
            assert backbone.n_logits == n_rendered_views
        self.timing_dict: Dict[str, float] = defaultdict(float)

    # -- refiner ---------------------------------------------------------------------------
    @torch.no_grad()
    def forward(self, images: torch.Tensor, K: torch.Tensor, labels: Sequence[str], TCO: torch.Tensor,
                n_iterations: int = 1, random_ambient_light: bool = False,
                im_ids: Optional[torch.Tensor] = None) -> Dict[str, PosePredictorOutput]:
        assert not random_ambient_light, "random_ambient_light is a training-time augmentation"
        bsz = len(labels)
        assert TCO.shape == (bsz, 4, 4)
        assert K.dim() == 3 and K.shape[1:] == (3, 3)
        assert images.dim() == 4 and images.shape[1] >= self._n_img, "images must be [B,C,H,W] with C>=3 (4 if input_depth)"
        per_hyp = im_ids is None
        im_ids, obj_ids = self._ids(images, K, labels, im_ids)
        TCO_input = TCO.to(self.device, torch.float32)
        recs = self._run_refine((n_iterations,), [images, K, im_ids, obj_ids, TCO_input],
                                lambda *t: self._refine_device(*t, n_iterations=n_iterations))
        return self._build_outputs(recs, list(labels), K, per_hyp, im_ids)

    def _refine_device(self, images, K, im_ids, obj_ids, TCO_input, *, n_iterations: int):
        """The device program of ``forward``: only launches on tensors (capturable as a hipGraph).  One record per
        iteration."""
        return list(self._refine_iter(images, K, im_ids, obj_ids, TCO_input, n_iterations=n_iterations))

    def _refine_iter(self, images, K, im_ids, obj_ids, TCO_input, *, n_iterations: int):
        """``_refine_device`` as a generator: yields an iteration's record once its launches are enqueued, so that
        ``TwoLanePredictor`` can enqueue its lanes' chains alternately (a lane whose whole chain is enqueued only after
        the other's starts milliseconds late whenever the caller synchronises between calls, as the estimators do)."""
        for _ in range(n_iterations):
            prep, x, pose, logits, render_time = self._one_pass(
                images, K, im_ids, obj_ids, TCO_input, n_img_channels=self._n_img,
                multiview_type=self.multiview_type, normalize=True, render_normals=self.render_normals,
                render_depth=self.render_depth, depth_mode=self._depth_mode,
                want_pose=self.predict_pose_update, want_logits=self.predict_rendered_views_logits,
                remove_TCO_rendering=self._skip_tco, scene_lights=not self.render_normals)
            TCO_norm = prep["TCO"]
            if self.predict_pose_update:
                TCO_output = ops.pose_update(TCO_norm, prep["K_crop_main"].contiguous() if self._skip_tco else prep["K_crop"], pose,
                                             prep["tCR"])
            else:
                TCO_output = TCO_norm.clone()
            images_crop, renders = self._pixels(x, self._n_img, self._n_single_render_channels * self.n_rendered_views)
            yield dict(TCO_input=TCO_norm, TCO_output=TCO_output, TCV_O=prep["TCV_O"], tCR=prep["tCR"],
                       KV_crop=prep["K_crop"], K_crop=prep["K_crop_main"] if self._skip_tco else None,
                       boxes_rend=prep["boxes_rend"], boxes_crop=prep["boxes_crop"],
                       pose=pose, logits=logits, images_crop=images_crop, renders=renders, render_time=render_time)
            TCO_input = TCO_output

    def _build_outputs(self, recs, labels, K, per_hyp, im_ids) -> Dict[str, PosePredictorOutput]:
        outputs: Dict[str, PosePredictorOutput] = {}
        Kb = K if per_hyp else K[im_ids.long()]
        for n, r in enumerate(recs):
            net_out = {}
            if r["pose"] is not None:
                net_out["pose"] = r["pose"]
            if r["logits"] is not None:
                net_out["renderings_logits"] = r["logits"]
            outputs[f"iteration={n + 1}"] = PosePredictorOutput(
                renders=r["renders"], images_crop=r["images_crop"], TCO_input=r["TCO_input"], TCO_output=r["TCO_output"],
                TCV_O_input=r["TCV_O"], tCR=r["tCR"], labels=labels, K=Kb,
                K_crop=r["K_crop"] if r.get("K_crop") is not None else r["KV_crop"][:, 0],
                KV_crop=r["KV_crop"], network_outputs=net_out, boxes_rend=r["boxes_rend"], boxes_crop=r["boxes_crop"],
                renderings_logits=r["logits"] if r["logits"] is not None else torch.empty(
                    len(labels), self.n_rendered_views, dtype=torch.float32, device=self.device),
                timing_dict={"render": r["render_time"]})
        return outputs

    __call__ = forward

    # -- coarse / scoring ------------------------------------------------------------------
    @torch.no_grad()
    def forward_coarse(self, images: torch.Tensor, K: torch.Tensor, labels: Sequence[str],
                       TCO_input: torch.Tensor, cuda_timer: bool = False, return_debug_data: bool = False,
                       im_ids: Optional[torch.Tensor] = None) -> Dict[str, Any]:
        """``MP/models/pose_rigid.py:708-788``: logits/scores of the rendered view."""
        assert self.predict_rendered_views_logits, "Method only valid if coarse classification model"
        bsz = len(labels)
        assert TCO_input.shape == (bsz, 4, 4)
        im_ids, obj_ids = self._ids(images, K, labels, im_ids)
        timer = _Timer(self.device, cuda_timer)
        timer.start()
        keep, self.keep_pixels = self.keep_pixels, self.keep_pixels or return_debug_data
        prep, x, _, logits, render_time = self._one_pass(
            images, K, im_ids, obj_ids, TCO_input.to(self.device, torch.float32), n_img_channels=self._n_img,
            multiview_type="TCO", normalize=True, render_normals=self.render_normals,
            render_depth=self.render_depth, depth_mode=self._depth_mode, want_pose=False, want_logits=True,
            scene_lights=not self.render_normals)
        timer.stop()
        out = {"logits": logits, "scores": torch.sigmoid(logits), "time": timer.elapsed(),
               "render_time": render_time, "model_time": timer.elapsed()}
        if return_debug_data:
            out["images_crop"], out["renders"] = self._pixels(x, self._n_img, self._n_single_render_channels)
        self.keep_pixels = keep
        return out


class CosyPosePosePredictor(_RenderAndCompare):
    """CosyPose predictor, ``CP/models/pose.py:33-199``: one rendered RGB view under white
    ambient light, 6 input channels, no ``normalize_T``, reference point = object origin
    (``apply_imagespace_predictions``)."""

    def __init__(self, backbone: ops.Net, renderer: BatchRenderer, mesh_db=None, render_size=(240, 320),
                 pose_dim: int = 9):
        self._setup(backbone, renderer, mesh_db, render_size)
        if pose_dim != 9:
            raise ValueError(f"pose_dim={pose_dim} not supported")
        self.pose_dim = pose_dim
        assert backbone.n_inputs == 6 and backbone.pose_dim == 9
        self._reserve(1)

    @torch.no_grad()
    def forward(self, images: torch.Tensor, K: torch.Tensor, labels: Sequence[str], TCO: torch.Tensor,
                n_iterations: int = 1, im_ids: Optional[torch.Tensor] = None) -> Dict[str, PosePredictorOutput]:
        bsz = len(labels)
        assert images.dim() == 4 and images.shape[1] >= 3
        assert K.dim() == 3 and K.shape[1:] == (3, 3)
        assert TCO.shape == (bsz, 4, 4)
        per_hyp = im_ids is None
        im_ids, obj_ids = self._ids(images, K, labels, im_ids)
        TCO_input = TCO.to(self.device, torch.float32)
        recs = self._run_refine((n_iterations,), [images, K, im_ids, obj_ids, TCO_input],
                                lambda *t: self._refine_device(*t, n_iterations=n_iterations))
        return self._build_outputs(recs, list(labels), K, per_hyp, im_ids)

    def _refine_device(self, images, K, im_ids, obj_ids, TCO_input, *, n_iterations: int):
        """The device program of ``forward`` (see ``PosePredictor._refine_device``)."""
        return list(self._refine_iter(images, K, im_ids, obj_ids, TCO_input, n_iterations=n_iterations))

    def _refine_iter(self, images, K, im_ids, obj_ids, TCO_input, *, n_iterations: int):
        for _ in range(n_iterations):
            prep, x, pose, _, render_time = self._one_pass(
                images, K, im_ids, obj_ids, TCO_input, n_img_channels=3, multiview_type="TCO", normalize=False,
                render_normals=False, render_depth=False, depth_mode=0, want_pose=True, want_logits=False)
            TCO_output = ops.pose_update(TCO_input, prep["K_crop"], pose, None)
            images_crop, renders = self._pixels(x, 3, 3)
            yield dict(TCO_input=TCO_input, TCO_output=TCO_output, TCV_O=prep["TCV_O"], tCR=prep["tCR"],
                       KV_crop=prep["K_crop"], boxes_rend=prep["boxes_rend"], boxes_crop=prep["boxes_crop"],
                       pose=pose, images_crop=images_crop, renders=renders, render_time=render_time)
            TCO_input = TCO_output

    def _build_outputs(self, recs, labels, K, per_hyp, im_ids) -> Dict[str, PosePredictorOutput]:
        outputs: Dict[str, PosePredictorOutput] = {}
        Kb = K if per_hyp else K[im_ids.long()]
        for n, r in enumerate(recs):
            outputs[f"iteration={n + 1}"] = PosePredictorOutput(
                renders=r["renders"], images_crop=r["images_crop"], TCO_input=r["TCO_input"], TCO_output=r["TCO_output"],
                TCV_O_input=r["TCV_O"], tCR=r["tCR"], labels=labels, K=Kb, K_crop=r["KV_crop"][:, 0],
                KV_crop=r["KV_crop"], network_outputs={"pose": r["pose"]}, boxes_rend=r["boxes_rend"],
                boxes_crop=r["boxes_crop"], timing_dict={"render": r["render_time"]})
        return outputs

    __call__ = forward


# ---------------------------------------------------------------------------------------------
# Two half-batch lanes
# ---------------------------------------------------------------------------------------------
class _LaneBackbones:
    """What ``bench.py`` needs from ``model.backbone`` when the model has two lanes: profiling of both networks."""

    def __init__(self, nets):
        self.nets = list(nets)

    def __getattr__(self, name):
        return getattr(self.nets[0], name)

    def set_profiling(self, on: bool):
        for n in self.nets:
            n.set_profiling(on)

    def set_conv_algo(self, name=None):
        for n in self.nets:
            n.set_conv_algo(name)

    def set_tail_split(self, on: bool):
        for n in self.nets:
            n.set_tail_split(on)

    def force_exact(self, on: bool = True):
        for n in self.nets:
            n.force_exact(on)

    def status(self, streams=None) -> int:
        flags = 0
        for i, n in enumerate(self.nets):
            flags |= n.status(None if streams is None else streams[i])
        return flags

    def profile_intervals(self):
        return [iv for n in self.nets for iv in n.profile_intervals()]

    def profile_collect(self):
        tot = [0.0, 0, 0.0, 0.0]
        for n in self.nets:
            for i, v in enumerate(n.profile_collect()):
                tot[i] += v
        return tuple(tot)


class _LaneOutputs:
    """The ``PosePredictorOutput`` of one iteration over both lanes: a field is concatenated when it is first read
    (a caller wants one or two of the dozen fields of one or two iterations; concatenating all of them for every
    iteration cost 2.5 % of a step in small launches)."""

    def __init__(self, parts: List[PosePredictorOutput]):
        object.__setattr__(self, "_parts", list(parts))
        object.__setattr__(self, "_cache", {})

    def __getattr__(self, name):
        cache = object.__getattribute__(self, "_cache")
        if name in cache:
            return cache[name]
        parts = object.__getattribute__(self, "_parts")
        if name not in PosePredictorOutput.__dataclass_fields__:
            raise AttributeError(name)
        vals = [getattr(p, name) for p in parts]
        v0 = vals[0]
        if isinstance(v0, torch.Tensor):
            out = torch.cat(vals, 0)
        elif isinstance(v0, list):
            out = [x for v in vals for x in v]
        elif name == "network_outputs":
            out = {k: torch.cat([v[k] for v in vals], 0) for k in v0}
        elif name == "timing_dict":
            out = {k: max(v.get(k, 0.0) for v in vals) for k in v0}
        else:
            out = v0  # None (pixels not kept)
        cache[name] = out
        return out

    def __getitem__(self, k):
        return getattr(self, k)


def _cat_outputs(parts: List[PosePredictorOutput]) -> "_LaneOutputs":
    return _LaneOutputs(parts)


class TwoLanePredictor:
    """A refiner whose ``forward`` runs the two halves of the hypothesis batch as two independent chains
    (prep -> crop -> rasterise -> network -> update, all iterations) on two HIP streams.

    Hypotheses are independent given (frame, meshes, weights), and a conv launch leaves CUs idle in its last,
    partially filled round of tiles (8x10 layers: 160 tiles on 256 CUs); a second, independent chain fills them
    (C2: 26.0 -> 23.1 ms per step).  Each lane owns everything that is written: its network (activation arena,
    K-slice workspaces are per stream), its input buffer and its mesh store (rasteriser scratch).  Results are the
    per-lane results concatenated -- identical to the single-lane ones up to the summation order inside K-sliced
    tiles.  Everything but ``forward`` (coarse scoring, attributes) is lane 0's."""

    MIN_BATCH = 32
    CHUNKS_TAIL_SPLIT = True  # forward_chunks: tail K-slicing of the lanes' networks while whole chunks run side by side

    def __init__(self, lanes):
        assert len(lanes) >= 2
        self.lanes = list(lanes)
        self.device = lanes[0].device
        self.streams = self._lane_streams(self.device, len(lanes))
        self.backbone = _LaneBackbones([l.backbone for l in lanes])
        self.use_graphs = False  # hipGraph replay of forward() (happypose_amd.graphs)
        self._graphs = None
        self._graph_epoch = None

    _STREAMS: dict = {}  # device -> the lane streams every instance on that device uses

    @classmethod
    def _lane_streams(cls, device, n: int):
        """One set of lane streams per device for the whole process.  HIP maps streams onto a few hardware queues in creation
        order; a predictor whose two fresh streams happened to share a queue ran its lanes one after the other (measured:
        every other predictor built in a process stepped in 32.3 instead of 23.5 ms).  So the streams are made once, each
        new one checked to run CONCURRENTLY with the ones already chosen (two one-thread spin kernels: together they take
        the time of one, or of two), and every predictor of the process uses them -- predictors are called one at a time."""
        key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
        pool = cls._STREAMS.setdefault(key, [])
        tries = 0
        while len(pool) < n:
            cand = torch.cuda.Stream(device=device)
            tries += 1
            if tries > 12 or all(cls._concurrent(device, cand, s) for s in pool):
                pool.append(cand)
        return pool[:n]

    @staticmethod
    def _concurrent(device, a, b, cycles: int = 400_000) -> bool:
        """Do kernels on streams ``a`` and ``b`` overlap?  (True when it cannot be measured.)"""
        import time

        sleep = getattr(torch.cuda, "_sleep", None)
        if sleep is None or torch.cuda.is_current_stream_capturing():
            return True

        def spin(streams):
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            for st in streams:
                with torch.cuda.stream(st):
                    sleep(cycles)
            for st in streams:
                st.synchronize()
            return time.perf_counter() - t0

        try:
            with torch.cuda.device(device):
                spin((a, b))  # first launches: lazy initialisation
                one = min(spin((a,)) for _ in range(3))
                two = min(spin((a, b)) for _ in range(3))
        except RuntimeError:
            return True
        return two < 1.6 * one

    def __getattr__(self, name):
        if name in ("lanes", "streams", "backbone", "device", "use_graphs", "_graphs", "_graph_epoch", "max_batch"):  # not set yet: no recursion through lanes[0]
            raise AttributeError(name)
        return getattr(self.lanes[0], name)

    def eval(self):
        return self

    @property
    def max_batch(self) -> int:
        return sum(int(l.backbone.max_batch) for l in self.lanes)

    def to(self, device):
        self.lanes[0].to(device)
        return self

    def _cuts(self, bsz: int):
        """Contiguous shares of a batch, one per lane, ``distributed.shard_range`` semantics: the first ``bsz % n`` lanes
        take one row more, so no lane exceeds ``ceil(bsz / n)`` -- what its backbone arena and rasteriser scratch were
        reserved for (``models._two_lanes``)."""
        n = len(self.lanes)
        q, r = divmod(bsz, n)
        edges = [i * q + min(i, r) for i in range(n)] + [bsz]
        return [slice(edges[i], edges[i + 1]) for i in range(n)]

    def numerics_status(self) -> int:
        """Guard flags over both lanes' backbones (see ``_RenderAndCompare.numerics_status``)."""
        torch.cuda.current_stream(self.device).synchronize()  # forward() already joined the lane streams into it
        return self.backbone.status()

    @torch.no_grad()
    def forward(self, images, K, labels, TCO, n_iterations: int = 1, *, im_ids=None, **kw):
        """``kw``: what the lanes' ``forward`` takes beyond this (MegaPose: ``random_ambient_light``)."""
        bsz = len(labels)
        if bsz < self.MIN_BATCH:
            self.lanes[0].use_graphs = self.use_graphs
            return self.lanes[0].forward(images, K, labels, TCO, n_iterations=n_iterations, im_ids=im_ids, **kw)
        assert not kw.get("random_ambient_light", False), "random_ambient_light is a training-time augmentation"
        labels = list(labels)
        cuts = self._cuts(bsz)
        lane0 = self.lanes[0]
        assert TCO.shape == (bsz, 4, 4) and K.dim() == 3 and K.shape[1:] == (3, 3) and images.dim() == 4
        per_hyp = im_ids is None  # the reference's calling convention: images / K already gathered per hypothesis
        im_ids, obj_ids = lane0._ids(images, K, labels, im_ids)
        TCO_input = TCO.to(self.device, torch.float32)
        # one chain per lane on its own stream; with use_graphs each lane replays ITS OWN captured graph there (a single
        # graph holding both chains is executed as one serial node order by this runtime: the lanes would not overlap)
        cur = torch.cuda.current_stream(self.device)
        parts = []
        # the other lane fills the CUs a partially filled round of tiles leaves idle: K-slicing those tiles would only
        # add its reduction (C2: 24.5 ms with, 22.5 ms without)
        self.backbone.set_tail_split(False)  # per network: other predictors are not affected
        try:
            eager = not self.use_graphs or any(l.keep_pixels or l.debug or l._profiling() or getattr(l, "_no_graphs", False)
                                               for l in self.lanes)
            plan = []
            for lane, stream, sl in zip(self.lanes, self.streams, cuts):
                lane.use_graphs = self.use_graphs
                stream.wait_stream(cur)
                with torch.cuda.stream(stream):
                    n_l = sl.stop - sl.start
                    ids_l = torch.arange(n_l, dtype=torch.int32, device=self.device) if per_hyp else im_ids[sl]
                    img_l, K_l = (images[sl], K[sl]) if per_hyp else (images, K)
                    if eager:  # enqueue the chains ALTERNATELY, one iteration at a time (see _refine_iter)
                        plan.append((lane, stream, sl, img_l, K_l, ids_l,
                                     lane._refine_iter(img_l, K_l, ids_l, obj_ids[sl], TCO_input[sl], n_iterations=n_iterations), []))
                        continue
                    recs = lane._run_refine((n_iterations,), [img_l, K_l, ids_l, obj_ids[sl], TCO_input[sl]],
                                            lambda *t, lane=lane: lane._refine_device(*t, n_iterations=n_iterations))
                    parts.append(lane._build_outputs(recs, labels[sl], K_l, per_hyp, None if per_hyp else ids_l))
            for _ in range(n_iterations if eager else 0):
                for lane, stream, sl, img_l, K_l, ids_l, gen, recs in plan:
                    with torch.cuda.stream(stream):
                        recs.append(next(gen))
            for lane, stream, sl, img_l, K_l, ids_l, gen, recs in plan:
                with torch.cuda.stream(stream):
                    parts.append(lane._build_outputs(recs, labels[sl], K_l, per_hyp, None if per_hyp else ids_l))
        finally:
            self.backbone.set_tail_split(True)
        for stream in self.streams:
            cur.wait_stream(stream)
        return {k: _cat_outputs([p[k] for p in parts]) for k in parts[0]}

    __call__ = forward

    @torch.no_grad()
    def forward_chunks(self, images, K, chunks, n_iterations: int = 1, **kw):
        """Independent SMALL batches (each below ``MIN_BATCH``: the ``bsz_objects`` chunks of an estimator stage, reference
        default 8, ``MP/inference/pose_estimator.py:74``) as concurrent chains: chunk ``c`` runs whole -- all its iterations --
        on lane ``c % n_lanes`` on that lane's stream.  ``forward`` sends such a batch to lane 0, and an estimator that calls
        it chunk after chunk leaves the other lanes (and, at batch 8, half of the CUs) idle: the refiner stage of the E2E
        frame, 5 chunks of 8 hypotheses, took 35 ms of 133.  Same launches per chunk as lane 0 would make (every lane holds the
        same plan and weights): the results do not depend on which lane ran a chunk.
        ``chunks``: ``[(labels, TCO, im_ids), ...]``; returns the chunks' outputs in order."""
        n = len(self.lanes)
        cur = torch.cuda.current_stream(self.device)
        used = self.streams[:min(n, len(chunks))]
        for stream in used:
            stream.wait_stream(cur)
        outs = []
        # Tail K-slicing stays ON here, unlike forward(): a chunk of 8 hypotheses is 40 - 160 tiles per layer, the lanes together
        # do not fill 256 CUs, and the slices are what spreads a layer over the idle ones (E2E refiner stage, three lanes, chunks
        # of 16: CHUNKS_TAIL_SPLIT on / off measured in CHANGELOG round 6); it also keeps a chunk's launches -- and so its bits --
        # those of lane 0 running it alone.
        self.backbone.set_tail_split(self.CHUNKS_TAIL_SPLIT)
        try:
            for c, (labels, TCO, im_ids) in enumerate(chunks):
                assert len(labels) < self.MIN_BATCH, "forward_chunks is for batches that forward() would not split"
                lane = self.lanes[c % n]
                lane.use_graphs = self.use_graphs
                with torch.cuda.stream(self.streams[c % n]):
                    outs.append(lane.forward(images, K, labels, TCO, n_iterations=n_iterations, im_ids=im_ids, **kw))
        finally:
            self.backbone.set_tail_split(True)
        for stream in used:
            cur.wait_stream(stream)
        return outs

    @torch.no_grad()
    def forward_coarse(self, images, K, labels, TCO_input, cuda_timer: bool = False, return_debug_data: bool = False,
                       im_ids=None):
        """Coarse scoring with the two halves of the views on the two lanes (``MP/models/pose_rigid.py:708-788``)."""
        bsz = len(labels)
        if bsz < 2 * self.MIN_BATCH or cuda_timer or return_debug_data:
            return self.lanes[0].forward_coarse(images, K, labels, TCO_input, cuda_timer, return_debug_data, im_ids)
        labels = list(labels)
        cuts = self._cuts(bsz)
        cur = torch.cuda.current_stream(self.device)
        parts = []
        self.backbone.set_tail_split(False)
        try:
            for lane, stream, sl in zip(self.lanes, self.streams, cuts):
                per_hyp = im_ids is None
                stream.wait_stream(cur)
                with torch.cuda.stream(stream):
                    parts.append(lane.forward_coarse(images[sl] if per_hyp else images, K[sl] if per_hyp else K, labels[sl],
                                                     TCO_input[sl], im_ids=None if per_hyp else torch.as_tensor(im_ids)[sl]))
        finally:
            self.backbone.set_tail_split(True)
        for stream in self.streams:
            cur.wait_stream(stream)
        out = {k: torch.cat([p[k] for p in parts], 0) for k in ("logits", "scores")}
        for k in ("time", "render_time", "model_time"):
            out[k] = max(p[k] for p in parts)
        return out
