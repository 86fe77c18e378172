#!/usr/bin/env python3
"""Headline benchmark: refined poses / second of the render-and-compare refiner.

Workload (BASELINE.json ``configs[1]``, "C2"): CosyPose refiner on one 640x480 frame,
8 detections x 16 hypotheses = 128 hypotheses per GPU, 5 refiner iterations,
WideResNet-34 backbone on 6 x 240 x 320 inputs, fp32.  A *step* is one pass of the hot path
over one such batch: 5 x (pose prep, roi_align crop, rasterise, conv stack, pose update).
Synthetic seeded inputs: there are no datasets/checkpoints offline.

Multi-GPU (``torch.distributed.run``, one rank per GPU, RCCL): weak scaling -- every rank
refines its own 128-hypothesis shard (N = 8 is BASELINE's "1k-hypothesis batch sharded
across 8 GPUs") and the step ends with ONE all-gather of the refined poses.

Prints one JSON line (rank 0) with the throughput, the roofline of the dominant kernel
(fp32 MFMA implicit-GEMM conv, timed live with HIP events on its launch stream) and the
CPU baseline (the oracle port of the reference algorithm on the host cores, bounded sample).
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_HBM_BPS = 8.0e12        # MI355X_MICROARCH.md: HBM3E ~8 TB/s
PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_F16_MFMA_TFLOPS = 2516.6  # dense fp16 MFMA: 16x the fp32 rate (same table)
N_DET, N_HYP, N_ITERS = 8, 16, 5


WORKLOADS = ("C2", "C3", "C5", "E2E")
EFFICIENTNET_LANES = 4  # C2 with the EfficientNet-b3 backbone: 153 small launches per forward; 3414 / 3455 / 3470 poses/s at 2 / 3 / 4 lanes
COARSE_HEAD_SCALE = 1.0  # the coarse / scoring head of the C5 and E2E worlds: logits that discriminate between grid poses (std ~ O(1) over an object's 576 poses, as golden G10's `coarse` case); at the pose head's 0.002 their whole range was 4e-4 and every tolerance blind
DEFAULT_LANES = {"C2": 2, "C3": 3, "C5": 2, "E2E": 3}  # --lanes (E2E: the refiner's bsz_objects chunks run one per lane; f16 coarse 7.99 / 8.53 / 8.28 frames/s at 2 / 3 / 4)


def build_world(device, arch="resnet34", seed=0, workload="C2", precision="f32", n_lanes=1, update_scale=0.002, renderer_kw=None):
    """Synthetic world of SURVEY.md 8(d) for a BASELINE.json config:
    C2 CosyPose refiner (8 det x 16 hyp, WideResNet-34 on 6 channels, 1 RGB view);
    C3 MegaPose RGB-D refiner (64 hypotheses, 4 views x (RGB + normals + depth), ResNet-34 on 32 ch);
    C5 MegaPose coarse scoring (8 objects x 576 SO(3)-grid poses, 1 view RGB + normals, ResNet-34 on 9 ch)."""
    from happypose_amd.models import create_model_pose, create_pose_model_cosypose, pose_model_param_shapes
    from happypose_amd.renderer import BatchRenderer
    from happypose_amd.synthetic import make_object_dataset, make_scene, predictor_weights

    ds = make_object_dataset(8, seed=1, tex_size=1024)
    renderer = BatchRenderer(ds, device=device, **(renderer_kw or {}))
    if workload == "C2":
        scene = make_scene(n_detections=N_DET, n_hypotheses=N_HYP, n_objects=8, seed=2 + seed)
        weights = predictor_weights(pose_model_param_shapes(arch, 6), seed=0, update_scale=update_scale)
        model = create_pose_model_cosypose(dict(backbone_str=arch), renderer, state_dict=weights,
                                           max_batch=N_DET * N_HYP, precision=precision, n_lanes=n_lanes)
    elif workload == "C3":
        scene = make_scene(n_detections=8, n_hypotheses=8, n_objects=8, seed=2 + seed, with_depth=True)
        weights = predictor_weights(pose_model_param_shapes("vanilla_resnet34", 32), seed=0, update_scale=update_scale)
        cfg = dict(backbone_str="vanilla_resnet34", n_rendered_views=4, multiview_type="front_3views",
                   render_normals=True, render_depth=True, input_depth=True, predict_pose_update=True,
                   depth_augmentation=False, depth_normalization_type="tCR_scale_clamp_center")
        model = create_model_pose(cfg, renderer, state_dict=weights, max_batch=64, precision=precision, n_lanes=n_lanes)
    else:
        from happypose_amd.pose_estimator import load_SO3_grid

        scene = make_scene(n_detections=8, n_hypotheses=1, n_objects=8, seed=2 + seed)
        grid = load_SO3_grid(576).numpy()
        T = np.repeat(scene["TCO_det"], 576, axis=0)
        T[:, :3, :3] = np.tile(grid, (8, 1, 1))
        scene["TCO_hyp"] = T.astype(np.float32)
        scene["hyp_obj_ids"] = np.repeat(scene["det_obj_ids"], 576).astype(np.int32)
        weights = predictor_weights(pose_model_param_shapes("vanilla_resnet34", 9, pose_dim=0, n_views_logits=1), seed=0, update_scale=COARSE_HEAD_SCALE)
        cfg = dict(backbone_str="vanilla_resnet34", n_rendered_views=1, multiview_type="TCO", render_normals=True,
                   predict_rendered_views_logits=True, predict_pose_update=False, depth_augmentation=False)
        model = create_model_pose(cfg, renderer, state_dict=weights, max_batch=576 * n_lanes, precision=precision, n_lanes=n_lanes)
    return ds, renderer, scene, weights, model


def e2e_run(device, rank, world, lanes, coarse_precision, steps, warmup, run_detector=False):
    """End-to-end frames/s of SURVEY.md 8(d): the whole ``PoseEstimator.run_inference_pipeline`` of the
    ``megapose-1.0-RGB-multi-hypothesis`` configuration (TB/utils/load_model.py:26-34) on one 640x480 frame
    with 8 detections per GPU: coarse scoring of 8 x 576 SO(3)-grid poses, top-5 hypotheses per detection,
    5 refiner iterations over the 40 hypotheses (4 views x RGB + normals, 27 channels), re-scoring, top-1 --
    host orchestration (pandas bookkeeping, chunking) included: what ``MP/evaluation/prediction_runner.py:228-236,
    265-291`` times.  ``run_detector``: the frame additionally goes through the Mask-RCNN detector (random weights: its
    detections are timed and discarded, the pose stages run on the projected-object boxes so that their work is the same
    in every run).  Returns the measurements of rank 0's line (``None`` on the other ranks)."""
    from happypose_amd.models import create_model_pose, pose_model_param_shapes
    from happypose_amd.pose_estimator import ObservationTensor, PoseEstimator, make_detections_from_object_data
    from happypose_amd.renderer import BatchRenderer
    from happypose_amd.synthetic import make_object_dataset, make_scene, predictor_weights

    ds = make_object_dataset(8, seed=1, tex_size=1024)
    renderer = BatchRenderer(ds, device=device)
    store = renderer.store
    scene = make_scene(n_detections=N_DET, n_hypotheses=1, n_objects=8, seed=2 + rank)
    ccfg = dict(backbone_str="vanilla_resnet34", n_rendered_views=1, multiview_type="TCO", render_normals=True,
                predict_rendered_views_logits=True, predict_pose_update=False, depth_augmentation=False)
    rcfg = dict(backbone_str="vanilla_resnet34", n_rendered_views=4, multiview_type="front_3views", render_normals=True,
                depth_augmentation=False)
    wc = predictor_weights(pose_model_param_shapes("vanilla_resnet34", 9, pose_dim=0, n_views_logits=1), seed=0, update_scale=COARSE_HEAD_SCALE)
    wr = predictor_weights(pose_model_param_shapes("vanilla_resnet34", 27), seed=1)
    coarse = create_model_pose(ccfg, renderer, state_dict=wc, max_batch=576, precision=coarse_precision, n_lanes=lanes)
    refiner = create_model_pose(rcfg, renderer, state_dict=wr, max_batch=64, precision="f32", n_lanes=lanes)
    detector = None
    if run_detector:
        from happypose_amd.detector import Detector, synthetic_maskrcnn

        detector = Detector(synthetic_maskrcnn(device, n_classes=len(store.labels) + 1, seed=3),
                            {f"{l}": i + 1 for i, l in enumerate(store.labels)})
    # chunk sizes of the reference's InferenceConfig (TB/inference/types.py:97-98), which its prediction runner passes to
    # run_inference_pipeline (MP/evaluation/prediction_runner.py:125-126): 16 refiner hypotheses, 576 coarse views per chunk
    est = PoseEstimator(refiner_model=refiner, coarse_model=coarse, detector_model=detector, bsz_objects=int(os.environ.get("HP_E2E_BSZ_OBJECTS", "16")),
                        bsz_images=576, SO3_grid_size=576)

    # detections = bounding boxes of the projected objects (what a detector would hand over)
    pts = store.mesh_db.points[scene["det_obj_ids"]].astype(np.float64)
    T = scene["TCO_det"].astype(np.float64)
    pc = np.einsum("nij,npj->npi", T[:, :3, :3], pts) + T[:, None, :3, 3]
    uv = np.einsum("ij,npj->npi", scene["K"][0].astype(np.float64), pc)
    uv = uv[..., :2] / uv[..., 2:]
    boxes = np.concatenate([uv.min(1), uv.max(1)], -1).astype(np.float32)
    det = make_detections_from_object_data([store.labels[i] for i in scene["det_obj_ids"]], boxes).to(device)
    obs = ObservationTensor(torch.as_tensor(scene["images"][:, :3].copy(), device=device),
                            torch.as_tensor(scene["K"], device=device))
    det_s = [0.0]

    def step():
        if detector is not None:  # timed, its (random-weight) detections discarded
            t_d = time.perf_counter()
            detector.get_detections(observation=obs, detection_th=0.0, output_masks=True)
            torch.cuda.synchronize(device)
            det_s[0] += time.perf_counter() - t_d
        final, extra = est.run_inference_pipeline(obs, detections=det, n_refiner_iterations=N_ITERS, n_pose_hypotheses=5)
        return final, extra

    def fence():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize(device)

    for _ in range(warmup):
        step()
    from happypose_amd import ops as _ops

    for m in (coarse, refiner):
        m.backbone.set_profiling(True)
    det_s[0] = 0.0
    stage_s = {"coarse": 0.0, "refiner": 0.0, "scoring": 0.0}
    fence()
    _ops.profile_mark_reference(device)
    t0 = time.perf_counter()
    for _ in range(steps):
        final, extra = step()
        stage_s["coarse"] += extra["coarse"]["data"]["time"]
        stage_s["refiner"] += extra["refiner"]["data"]["time"]
        stage_s["scoring"] += extra["scoring"]["data"]["time"]
    fence()
    elapsed = time.perf_counter() - t0
    # the time during which ANY conv kernel of either backbone ran (lanes and backbones overlap: their sum is not a share)
    conv_union_ms = union_ms([iv for m in (coarse, refiner) for iv in m.backbone.profile_intervals()])
    prof = [m.backbone.profile_collect() for m in (coarse, refiner)]
    for m in (coarse, refiner):
        m.backbone.set_profiling(False)
    assert len(final) == N_DET and torch.isfinite(final.poses).all()
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank != 0:
        return None
    stage_ms = {k: 1e3 * v / steps for k, v in stage_s.items()}
    if detector is not None:
        stage_ms["detection"] = 1e3 * det_s[0] / steps
    stage_ms["host_and_bookkeeping"] = 1e3 * elapsed / steps - sum(stage_ms.values())
    cp = extra["coarse"]["preds"]
    product = dict(coarse_logit=cp.infos.coarse_logit.values.astype(np.float64), final_hyp=final.infos.hypothesis_id.tolist(),
                   final_labels=final.infos.label.tolist(), final_poses=final.poses.cpu().numpy(),
                   filtered_hyp=extra["coarse_filter"]["preds"].infos.hypothesis_id.tolist())
    return {"elapsed": elapsed, "steps": steps, "prof": prof, "conv_union_ms": conv_union_ms, "stage_ms_per_frame": stage_ms,
            "timing_str": extra["timing_str"], "product": product,
            "job_inputs": dict(images=scene["images"][:, :3], K=scene["K"], labels=[store.labels[i] for i in scene["det_obj_ids"]], boxes=boxes,
                               wc=wc, wr=wr, packed=store.packed, points=store.mesh_db.points, store_labels=list(store.labels),
                               render_kw=dict(msaa=renderer.msaa, aniso=renderer.aniso))}


def e2e_oracle(job, cores=None):
    """The SAME end-to-end job (8 detections x 576 coarse poses, top-5, 5 refiner iterations x 40 hypotheses x 4 views, re-scoring,
    top-1) through the CPU oracle's estimator (oracle/estimator.py: the checker, fp32 throughout), once, on every host core --
    about a minute.  Returns its result dict and the seconds it took."""
    from oracle import native as oracle_native
    from oracle.estimator import OracleEstimator
    from oracle.pipeline import OraclePredictor

    cores = cores or effective_cpu_count()
    oracle_native.set_threads(cores)
    torch.set_num_threads(cores)
    rk = job["render_kw"]
    oc = OraclePredictor(job["wc"], job["packed"], job["points"], arch="vanilla_resnet34", render_normals=True, **rk)
    orf = OraclePredictor(job["wr"], job["packed"], job["points"], arch="vanilla_resnet34", n_views=4, multiview_type="TCO+front_3views",
                          render_normals=True, **rk)
    t0 = time.time()
    ref = OracleEstimator(orf, oc, job["store_labels"], SO3_grid_size=576, bsz_objects=8, bsz_images=576).run_inference_pipeline(
        job["images"], job["K"], job["labels"], job["boxes"], n_refiner_iterations=N_ITERS, n_pose_hypotheses=5,
        instance_id=np.arange(len(job["labels"])))
    return ref, time.time() - t0


# coarse logits of an end-to-end run against the oracle estimator's, in units of the oracle's SPREAD over a detection's 576 grid
# poses (std; the coarse head of the bench worlds has COARSE_HEAD_SCALE = 1.0): the measured error of the healthy path with
# head-room, as tests/test_gpu_pipeline.py::test_c5_coarse_scoring_vs_oracle (tools/probes/c5_parity_probe.py; a network with
# one conv layer off by 1 % is outside them)
COARSE_LOGIT_REL = {"f32": 0.04, "f16": 0.10}


def e2e_parity(product, ref, coarse_precision):
    """Final-pose parity of an end-to-end run against the oracle estimator's run of the same job: coarse logits of all 4608 poses
    RELATIVE to the oracle's own spread per detection, the top-5 selection, and -- for the detections where both sides picked the
    same hypothesis in the end -- the final poses."""
    cl = ref["coarse_df"]["coarse_logit"].values.astype(np.float64)
    det = ref["coarse_df"]["instance_id"].values if "instance_id" in ref["coarse_df"] else np.arange(len(cl)) // 576
    diff = np.abs(product["coarse_logit"] - cl)
    rel, spreads = [], []
    for d in np.unique(det):
        m = det == d
        spreads.append(float(cl[m].std()))
        rel.append(float(diff[m].max() / max(cl[m].std(), 1e-30)))
    rel_tol = COARSE_LOGIT_REL[coarse_precision]
    same_top = sorted(product["filtered_hyp"]) == sorted(ref["filtered_df"]["hypothesis_id"].tolist())
    rl = dict(zip(ref["final_df"]["label"].tolist(), zip(ref["final_df"]["hypothesis_id"].tolist(), range(len(ref["final_df"])))))
    match, dts, drs = 0, [], []
    for k, (lab, hyp) in enumerate(zip(product["final_labels"], product["final_hyp"])):
        if lab in rl and rl[lab][0] == hyp:
            match += 1
            pp = pose_parity(product["final_poses"][k:k + 1], ref["final_TCO"][rl[lab][1]:rl[lab][1] + 1])
            dts.append(pp["max_dt_m"]); drs.append(pp["max_dR_rad"])
    return {"coarse_logit_max_abs_diff": float(diff.max()), "coarse_logit_spread_per_detection": [round(x, 5) for x in spreads],
            "coarse_logit_max_diff_over_spread": max(rel), "coarse_logit_rel_tol": rel_tol,
            "top5_sets_equal": bool(same_top), "detections": len(product["final_hyp"]), "final_hypothesis_agrees": match,
            "max_dt_m": max(dts) if dts else None, "max_dR_rad": max(drs) if drs else None, "tol": {"dt_m": T_TOL, "dR_rad": R_TOL},
            "ok": bool(max(rel) <= rel_tol and (not dts or (max(dts) <= T_TOL and max(drs) <= R_TOL))),
            "note": "coarse logits are compared in units of the oracle's own standard deviation over a detection's 576 grid poses; a detection "
                    "whose two best hypotheses score within the error of each other may legitimately end on the other one: poses are "
                    "compared where the final hypothesis ids agree"}


def bench_e2e(args, device, rank, world):
    """``--workload E2E``: the line of :func:`e2e_run`."""
    coarse_precision = args.precision or "f32"
    r = e2e_run(device, rank, world, args.lanes or DEFAULT_LANES["E2E"], coarse_precision, args.steps, args.warmup, run_detector=args.run_detector)
    if r is None:
        return
    elapsed, prof, conv_union_ms = r["elapsed"], r["prof"], r["conv_union_ms"]
    # dominant backbone = the one with more conv time (the coarse net unless it runs in fp16); roofline of the issued
    # instruction, v_mfma_f32_32x32x16_f16: executed fp16-MFMA FLOPs (fp32-MFMA-time equivalents x 16) / conv time
    names = ("coarse", "refiner")
    dom = 0 if prof[0][0] >= prof[1][0] else 1
    conv_ms = sum(p[0] for p in prof)
    per_net = {names[i]: {"achieved": 16.0 * prof[i][3] / (prof[i][0] * 1e-3) / 1e12, "peak": PEAK_F16_MFMA_TFLOPS,
                          "frac": 16.0 * prof[i][3] / (prof[i][0] * 1e-3) / 1e12 / PEAK_F16_MFMA_TFLOPS,
                          "algorithmic_tflops": prof[i][2] / (prof[i][0] * 1e-3) / 1e12, "conv_ms_per_frame": prof[i][0] / args.steps}
               for i in range(2)}
    achieved, peak = per_net[names[dom]]["achieved"], PEAK_F16_MFMA_TFLOPS
    line = {
        "metric": "end-to-end frames/sec (640x480, 8 detections, 576-pose coarse grid, 5 hyp/det, 5 refiner iters)",
        "value": world * args.steps / elapsed, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32" if coarse_precision == "f32" else "f32 refiner + f16 coarse net",
        "data": "synthetic",
        "config": {"workload": "E2E: PoseEstimator.run_inference_pipeline, megapose-1.0-RGB-multi-hypothesis shape: one 640x480 "
                               "frame per GPU, 8 detections, coarse 8 x 576 views (vanilla_resnet34 on 9 ch), top-5 hypotheses, "
                               "5 refiner iterations x 40 hypotheses x 4 views (vanilla_resnet34 on 27 ch), re-scoring, top-1"
                               + (", Mask-RCNN detector on the frame first (run_detector)" if args.run_detector else ""),
                   "detections_per_gpu": N_DET, "parallelism": f"frame-replica x{world}"},
        "roofline": {"bound": "mfma", "kernel": f"all conv launches of the {names[dom]} backbone (the one with more conv time)", "per_backbone": per_net,
                     "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak, "traffic": None,
                     "launches": int(sum(p[1] for p in prof)), "conv_time_share": conv_union_ms * 1e-3 / elapsed,
                     "conv_time_sum_over_wall": conv_ms * 1e-3 / elapsed},
        "stage_ms_per_frame": r["stage_ms_per_frame"],
        "stage_seconds_last_frame": r["timing_str"],
    }
    print(json.dumps(line), flush=True)


def union_ms(intervals) -> float:
    """Length of the union of ``[(t0, t1), ...]``: the time during which ANY of the timed stretches ran (the lanes /
    backbones overlap, so their sum can exceed the wall time)."""
    union, end = 0.0, -1.0
    for a0, a1 in sorted(intervals):
        if a1 > end:
            union += a1 - max(a0, end)
            end = a1
    return union


def recorded_traffic(kind="conv"):
    """HBM bytes per launch from the PMC passes of THIS command committed under profiles/ (counters cannot be read
    inside the run; collected with rocprofv3 ``--pmc FETCH_SIZE`` / ``WRITE_SIZE`` in separate passes, tools/pmc_traffic.py):
    the newest ``profiles/*_{kind}_hbm_traffic.json``.  Returns ``(bytes_per_launch | None, source file | None)``."""
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"*_{kind}_hbm_traffic.json")))
    if not files:
        return None, None
    try:
        with open(files[-1]) as fh:
            d = json.load(fh)
        return float(d["hbm_bytes_per_launch"]), os.path.relpath(files[-1], ROOT)
    except (OSError, ValueError, KeyError):
        return None, None


def quick_workload(device, workload, precision, n_lanes, steps=3, warmup=2, renderer_kw=None, arch="resnet34", graphs=False):
    """A short run of another BASELINE.json config in the same process (C3: MegaPose RGB-D refiner, fp32; C5: coarse
    scoring in fp16; C2 again in another render state): ``{value, unit, ms_per_step, frac, algorithmic_tflops, steps}`` --
    the driver-visible twin of ``bench.py --workload C3|C5`` / ``--render-state single-sample``."""
    from happypose_amd import ops as _ops

    ds, renderer, scene, weights, model = build_world(device, arch, seed=0, workload=workload, precision=precision, n_lanes=n_lanes,
                                                      renderer_kw=renderer_kw)
    store = renderer.store
    B = len(scene["TCO_hyp"])
    images, K = torch.as_tensor(scene["images"], device=device), torch.as_tensor(scene["K"], device=device)
    TCO0 = torch.as_tensor(scene["TCO_hyp"], device=device)
    labels = [store.labels[i] for i in scene["hyp_obj_ids"]]
    im_ids = torch.zeros(B, dtype=torch.int32, device=device)

    def step():
        if workload == "C5":
            ck = 576 * n_lanes
            return torch.cat([model.forward_coarse(images, K, labels[i:i + ck], TCO0[i:i + ck], im_ids=im_ids[i:i + ck])["logits"]
                              for i in range(0, B, ck)])
        return model.forward(images, K, labels, TCO0, n_iterations=N_ITERS, im_ids=im_ids)[f"iteration={N_ITERS}"].TCO_output

    import gc

    for _ in range(warmup):
        step()
    # the previous workload's world (device buffers behind reference cycles) must not be collected -- hipFree synchronises
    # the device -- inside the timed steps: 4007 instead of 5620 poses/s was measured when it happened
    torch.cuda.synchronize(device)
    gc.collect()
    gc.disable()
    replayed = False
    try:
        step()
        if graphs and workload != "C5":
            # each lane replays its step as a captured hipGraph (happypose_amd/graphs.py; create_*_model(graphs=True)): the timed pass is
            # the replay, the roofline's per-launch timing comes from the same steps run once more eagerly (events cannot be recorded
            # inside a replayed graph) -- as `bench.py --graphs on` does for the headline
            try:
                model.use_graphs = True
                step(); step()
                torch.cuda.synchronize(device)
                t0 = time.perf_counter()
                for _ in range(steps):
                    out = step()
                torch.cuda.synchronize(device)
                elapsed, replayed = time.perf_counter() - t0, True
            except Exception:
                replayed = False
            model.use_graphs = False
        model.backbone.set_profiling(True)  # (also drops captured graphs)
        if replayed:
            step()
            model.backbone.profile_collect(); model.backbone.profile_intervals()
        torch.cuda.synchronize(device)
        _ops.profile_mark_reference(device)
        t0 = time.perf_counter()
        for _ in range(steps):
            out = step()
        torch.cuda.synchronize(device)
        eager_elapsed = time.perf_counter() - t0
        if not replayed:
            elapsed = eager_elapsed
    finally:
        gc.enable()
    conv_ms = union_ms(model.backbone.profile_intervals())
    _, n_launch, conv_flops, mfma_flops = model.backbone.profile_collect()
    model.backbone.set_profiling(False)
    assert torch.isfinite(out).all() and model.numerics_status() == 0
    sec = conv_ms * 1e-3
    executed_f16 = (16.0 if precision == "f32" else 1.0) * mfma_flops / sec / 1e12 if sec > 0 else 0.0
    res = {"value": B * steps / elapsed, "unit": "refined poses/s" if workload != "C5" else "views/s", "dtype": precision,
           "ms_per_step": 1e3 * elapsed / steps, "steps": steps, "frac": executed_f16 / PEAK_F16_MFMA_TFLOPS,
           "algorithmic_tflops": conv_flops / sec / 1e12 if sec > 0 else 0.0, "conv_time_share": sec / eager_elapsed,
           "hypotheses_per_step": B, "lanes": n_lanes, "scratch_launches": int(_ops.scratch_launches())}
    if replayed:
        res["graphs"] = "each lane replays its 5-iteration step as a captured hipGraph; frac / conv_time_share from the same steps run eagerly"
        res["eager_value"] = B * steps / eager_elapsed
    return res


def estimator_entry(model, images, K, labels, TCO0, device):
    """``step()`` through the entry point: ``CosyPoseEstimator.forward_refiner`` (bsz_objects = the whole table) on the C2 job;
    returns the final poses ``[B, 4, 4]`` like the direct call's ``iteration=5`` output."""
    import pandas as pd

    from happypose_amd.pose_estimator import CosyPoseEstimator, ObservationTensor
    from happypose_amd.tensor_collection import PandasTensorCollection

    B = len(labels)
    est = CosyPoseEstimator(refiner_model=model, coarse_model=model, bsz_objects=B)
    obs = ObservationTensor(images, K)
    infos = pd.DataFrame({"label": list(labels), "batch_im_id": np.zeros(B, dtype=np.int64), "instance_id": np.arange(B) // N_HYP,
                          "hypothesis_id": np.arange(B) % N_HYP})
    data = PandasTensorCollection(infos=infos, poses=TCO0)

    def step():
        preds, _ = est.forward_refiner(obs, data, n_iterations=N_ITERS)
        return preds[f"iteration={N_ITERS}"].poses

    return step


def side_timing(fn, device, steps, poses_main, B):
    """The OTHER way into the same job (the bare predictor when the headline goes through the estimator, and vice versa): ms per
    step over ``steps`` steps, and the largest difference of its final poses to the headline's (0: the same launches)."""
    for _ in range(2):
        out = fn()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(steps):
        out = fn()
    torch.cuda.synchronize(device)
    ms = 1e3 * (time.perf_counter() - t0) / steps
    return {"ms_per_step": ms, "steps": steps, "value": B / (ms * 1e-3), "unit": "refined poses/s",
            "max_abs_pose_diff_vs_headline": float((out[:B] - poses_main[:B]).abs().max())}


def effective_cpu_count() -> int:
    """Cores this process may actually use: affinity mask capped by the cgroup CPU quota
    (the GPU boxes expose 256 logical CPUs but grant a 16-CPU quota; oversubscribing the
    quota makes oneDNN ~300x slower, which would flatter the GPU)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(np.ceil(int(quota) / int(period)))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline(ds_store, scene, weights, arch, budget_s=15.0, cores=None, render_kw=None):
    """The oracle port of the reference algorithm (torch-CPU conv stack, C rasteriser and
    roi_align, reference batching bsz_objects=8) timed on the host cores on a bounded
    sample of the same workload.  cores=None: every core the box grants; cores=1: the setting the
    reference itself forces at import (OMP_NUM_THREADS=MKL_NUM_THREADS=1, MP/__init__.py:35-36)."""
    cores = cores or effective_cpu_count()
    from oracle import native as oracle_native
    from oracle.pipeline import OraclePredictor

    oracle_native.set_threads(cores)
    torch.set_num_threads(cores)
    ora = OraclePredictor(weights, ds_store.packed, ds_store.mesh_db.points, arch=arch, cosypose=True, **(render_kw or {}))

    last = {}

    def run(n):
        t0 = time.time()
        its = ora.forward(scene["images"][:, :3], scene["K"], np.zeros(n, np.int32), scene["hyp_obj_ids"][:n],
                          scene["TCO_hyp"][:n], N_ITERS, bsz_objects=8)
        last["poses"] = its[-1]["TCO_output"]
        return time.time() - t0

    run(8)  # warm-up (thread pools, oneDNN primitive cache)
    t8 = run(8)
    n = int(min(N_DET * N_HYP, max(8, 8 * round(budget_s / max(t8, 1e-3)))))
    t = run(n) if n != 8 else t8
    return {"value": n / t, "unit": "refined poses/s", "cores": cores, "kind": "port",
            "sample": f"{n} of {N_DET * N_HYP} hypotheses x {N_ITERS} iterations in chunks of 8 "
                      f"(oracle/pipeline.py: torch-CPU unfused conv stack + C rasteriser/roi_align), {t:.1f} s"}, last["poses"]


T_TOL, R_TOL = 2e-5, 1e-4  # the stated tolerance of the fp32 path (tests/test_gpu_pipeline.py; SURVEY.md 8d proposed 1e-4 / 1e-3)


def pose_parity(gpu_poses: np.ndarray, cpu_poses: np.ndarray) -> dict:
    """Same-run CPU <-> GPU check (last row): the CPU baseline's final poses of its sample against the
    poses the timed HIP path produced for the same hypotheses."""
    n = len(cpu_poses)
    A, B = np.asarray(gpu_poses[:n], np.float64), np.asarray(cpu_poses, np.float64)
    dt = np.linalg.norm(A[:, :3, 3] - B[:, :3, 3], axis=1)
    # geodesic angle through the chord: ||R_A - R_B||_F = 2 sqrt(2) sin(theta / 2).  (arccos of the trace has a floor of
    # ~sqrt(2 * 1e-7) = 5e-4 rad on fp32 matrices -- it reported 8e-4 rad for poses that agree to 1e-6)
    chord = np.linalg.norm(A[:, :3, :3] - B[:, :3, :3], axis=(1, 2))
    ang = 2.0 * np.arcsin(np.clip(chord / (2.0 * np.sqrt(2.0)), 0.0, 1.0))
    ok = bool(np.isfinite(A).all() and dt.max() <= T_TOL and ang.max() <= R_TOL)
    return {"hypotheses_compared": n, "iterations": N_ITERS, "max_dt_m": float(dt.max()), "max_dR_rad": float(ang.max()),
            "tol": {"dt_m": T_TOL, "dR_rad": R_TOL}, "ok": ok}


def stage_rates(store, scene, images, K, TCO0, im_ids, device, reps=20):
    """The input stage (render + crop) on the C2 inputs in the product layout -- ``hp_render_inputs``: 128 rendered RGB views
    and the 128 observed crops into the 6-float pixel records of the network input, one launch, every record written once
    -- in BOTH render states: the reference's (4x MSAA + mip-mapped anisotropic-16 texturing, the product default) and the
    single-sample / bilinear one; plus the two stand-alone kernels (rasteriser into its channel slice, crop into its
    slice) in the single-sample state, as round 2 reported them.  Timed with events on the launch stream.  Bytes are the
    ALGORITHMIC bytes of SURVEY.md 8(d): per view V*32 + F*12 + h*w*4*C_out + 4 B of texture per covered pixel
    (rasteriser) + the crop's output bytes."""
    from happypose_amd import ops

    B = TCO0.shape[0]
    obj = torch.as_tensor(scene["hyp_obj_ids"][:B], device=device)
    prep = ops.pose_prep(store, TCO0, K, im_ids, obj, (480, 640))
    Kc = prep["K_crop"][:, 0].contiguous()
    x = torch.zeros((B, 240, 320, 6), device=device)
    depth = ops.rasterize(store, obj, TCO0, Kc, (240, 320), render_depth=True)[2]
    covered = float((depth > 0).sum().item())
    rows = store.packed.obj[np.asarray(scene["hyp_obj_ids"][:B])]  # (voff, nv, foff, nf, ...)
    nv, nf = rows[:, 1].astype(np.float64), rows[:, 3].astype(np.float64)
    raster_bytes = float((nv * 32 + nf * 12).sum()) + B * 76800 * 3 * 4 + covered * 4
    crop_bytes = float(B * 76800 * 3 * 4)

    def timeit(fn):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize(device)
        return e0.elapsed_time(e1) / reps * 1e-3

    def fused(msaa, aniso):
        return lambda: ops.render_inputs(store, x, obj, TCO0[:, None], Kc[:, None], False, False, images=images, boxes=prep["boxes_crop"],
                                         im_ids=im_ids, n_img_channels=3, msaa=msaa, aniso=aniso)

    def entry(t, nbytes, **kw):
        return dict(us=t * 1e6, algorithmic_MB=nbytes / 1e6, **{"GB/s": nbytes / t / 1e9}, frac_hbm_peak=nbytes / t / PEAK_HBM_BPS, **kw)

    t_ref, t_fast = timeit(fused(True, True)), timeit(fused(False, False))
    t_r = timeit(lambda: ops.rasterize_into(store, x, 3, obj, TCO0[:, None], Kc[:, None], False, False))
    t_rr = timeit(lambda: ops.rasterize_into(store, x, 3, obj, TCO0[:, None], Kc[:, None], False, False, msaa=True, aniso=True))
    t_c = timeit(lambda: ops.crop_roi_align(images, prep["boxes_crop"], im_ids, out=x, n_channels=3))
    return {
        "rasterize_reference_state": entry(t_rr, raster_bytes, views=B, coverage=covered / (B * 76800),
                                           render_state="reference: 4x MSAA + mipmap / anisotropic-16 (product default); what the "
                                                        "predictors launch after the stand-alone crop"),
        "render_inputs": entry(t_ref, raster_bytes + crop_bytes, views=B, crops=B,
                               render_state="reference state, crop fused into the render launch (opt-in: fuse_crop)"),
        "render_inputs_single_sample": entry(t_fast, raster_bytes + crop_bytes, views=B, crops=B,
                                             render_state="single sample, bilinear level 0 (BatchRenderer(msaa=False, aniso=False))"),
        "rasterize": entry(t_r, raster_bytes, views=B, render_state="single sample, bilinear; stand-alone launch into its channel slice"),
        "crop_roi_align": entry(t_c, crop_bytes, crops=B),
    }


def spawn_ranks(n: int) -> int:
    """``python bench.py --gpus N`` without a launcher: start N ranks (one per GPU) as child processes BEFORE this
    process touches the GPU, wire them up like ``torch.distributed.run`` does (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_*), pass their output through (rank 0 prints the JSON line) and return the worst exit code."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--arch", default="resnet34", choices=["resnet34", "resnet18", "efficientnet-b3"],
                    help="CosyPose backbone for C2: resnet34 = the WideResNet-34 BASELINE.json quotes; efficientnet-b3 = the released checkpoints' architecture")
    ap.add_argument("--workload", default="C2", choices=list(WORKLOADS),
                    help="BASELINE.json config (default C2 = the one the headline metric is quoted on)")
    ap.add_argument("--no-exact-fp32", action="store_true", help="skip the secondary run on the exact-fp32 kernels")
    ap.add_argument("--lanes", type=int, default=None, choices=[1, 2, 3, 4],
                    help="shares of the batch as independent chains on their own streams (TwoLanePredictor).  Default: 2, and 3 for "
                         "C3 (64 hypotheses of 4 views: measured 2747 / 2828 / 2790 poses/s at 2 / 3 / 4 lanes; C2: 5273 / 5330 / "
                         "4970 with the conv-busy share of the step going from 0.87 to 0.94 to 0.96, left at 2)")
    ap.add_argument("--precision", default=None, choices=["f32", "f16"],
                    help="conv arithmetic (default: f32, the reference's; f16 for C5 as BASELINE.json names it)")
    ap.add_argument("--graphs", default="off", choices=["on", "off"],
                    help="replay the refiner step as a captured hipGraph (happypose_amd.graphs).  Measured: within 2 %% of the "
                         "eager path on C2 and C3 -- the steps are bound by the GPU, not by the host's launch rate")
    ap.add_argument("--render-state", default="reference", choices=["reference", "single-sample"],
                    help="reference = the reference renderer's state (4x MSAA, mipmap + anisotropic-16; product default); single-sample = "
                         "one sample per pixel, bilinear level 0 (what rounds 1-2 measured)")
    ap.add_argument("--run-detector", action="store_true", help="E2E: the frame also goes through the Mask-RCNN detector (random weights)")
    ap.add_argument("--entry", default="estimator", choices=["estimator", "predictor"],
                    help="C2 on one GPU: what a step calls -- CosyPoseEstimator.forward_refiner (the drop-in entry point, SURVEY.md 0.8: the "
                         "default, pandas bookkeeping and the numerical guard's status query included) or the bare predictor's forward; the "
                         "other one is timed beside it (key `estimator`)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-workloads", action="store_true", help="skip the 3-step C3 / C5 runs appended to the C2 line")
    ap.add_argument("--no-cpu-1thread", action="store_true")
    ap.add_argument("--no-e2e-parity", action="store_true", help="skip the oracle estimator's run of the end-to-end job (about a minute of CPU)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:  # no launcher: this process only spawns the ranks
        # (no device query here: it would initialise HIP in this parent, which only spawns; a rank without a GPU fails
        # with its own message)
        sys.exit(spawn_ranks(args.gpus))

    from happypose_amd import distributed as D

    # HP_BENCH_DIST_BACKEND=gloo + HP_BENCH_ONE_DEVICE=1: every rank on cuda:0 with gloo collectives -- how the N > 1 path
    # (sharded batch, refine_sharded with the real predictor, the all-gather, max-over-ranks timing) is exercised on a
    # one-GPU box (tests/test_gpu_pipeline.py); RCCL refuses two ranks on one device.  Not a measurement.
    backend = os.environ.get("HP_BENCH_DIST_BACKEND", "nccl")
    one_device = os.environ.get("HP_BENCH_ONE_DEVICE") == "1"
    rank, local_rank, world = D.init_distributed(backend if args.gpus > 1 else None)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py measures the HIP path: a GPU is required (no CPU fallback)"
    device = torch.device("cuda:0" if one_device else f"cuda:{local_rank}")
    torch.cuda.set_device(device)

    if args.workload == "E2E":
        bench_e2e(args, device, rank, world)
        if world > 1:
            torch.distributed.barrier()
            torch.distributed.destroy_process_group()
        return
    precision = args.precision or ("f16" if args.workload == "C5" else "f32")
    peak = PEAK_F16_MFMA_TFLOPS if precision == "f16" else PEAK_F32_MFMA_TFLOPS
    n_lanes = args.lanes or (EFFICIENTNET_LANES if args.arch == "efficientnet-b3" and args.workload == "C2" else DEFAULT_LANES.get(args.workload, 2))
    rkw = None if args.render_state == "reference" else dict(msaa=False, aniso=False)
    ds, renderer, scene, weights, model = build_world(device, args.arch, seed=rank, workload=args.workload,
                                                      precision=precision, n_lanes=n_lanes, renderer_kw=rkw)
    store = renderer.store
    B = len(scene["TCO_hyp"])
    sharded = world > 1 and args.workload == "C2"
    if sharded:
        # C4: ONE hypothesis batch of world x 128 rows on one frame (8 x world detections x 16 hypotheses), the same on
        # every rank; distributed.refine_sharded cuts it into contiguous shards of 128 and merges the refined poses
        # with one all-gather -- weak scaling: 128 rows per GPU whatever the number of GPUs
        from happypose_amd.synthetic import make_scene
        scene = make_scene(n_detections=N_DET * world, n_hypotheses=N_HYP, n_objects=8, seed=2)
        assert len(scene["TCO_hyp"]) == world * B
    images = torch.as_tensor(scene["images"], device=device)  # inputs resident in HBM
    K = torch.as_tensor(scene["K"], device=device)
    TCO0 = torch.as_tensor(scene["TCO_hyp"], device=device)
    labels = [store.labels[i] for i in scene["hyp_obj_ids"]]
    im_ids = torch.zeros(len(labels), dtype=torch.int32, device=device)

    # C2 on one GPU: the step goes through the ENTRY POINT, CosyPoseEstimator.forward_refiner (CP/integrated/pose_estimator.py:249-356)
    # on the 128-row hypothesis table, bsz_objects = 128 -- pandas bookkeeping, the per-iteration result tables and the guard's status
    # query included (north_star: "run_inference_pipeline() stays the drop-in entry point"; SURVEY.md 0.8 names forward_refiner as
    # what is timed).  --entry predictor times the bare model.forward instead; either way the other one is timed beside it.
    via_estimator = args.workload == "C2" and world == 1 and args.entry == "estimator"
    entry_step = None
    if args.workload == "C2" and world == 1:
        entry_step = estimator_entry(model, images, K, labels, TCO0, device)

    def step():
        if via_estimator:
            return entry_step()
        if args.workload == "C5":  # coarse scoring, one object (576 grid poses) per chunk and lane as in 8(e)
            ck = 576 * n_lanes
            scores = [model.forward_coarse(images, K, labels[i:i + ck], TCO0[i:i + ck], im_ids=im_ids[i:i + ck])["logits"]
                      for i in range(0, B, ck)]
            poses, logits = TCO0, torch.cat(scores).reshape(-1)
        elif sharded:
            return D.refine_sharded(model, images, K, labels, TCO0, N_ITERS, im_ids=im_ids)[0]
        else:
            out = model.forward(images, K, labels, TCO0, n_iterations=N_ITERS, im_ids=im_ids)
            poses, logits = out[f"iteration={N_ITERS}"].TCO_output, None
        if world > 1:
            poses, _ = D.gather_poses(poses, logits, rank * B, world * B)
        return poses

    def fence():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize(device)

    graphs = args.workload != "C5" and args.graphs == "on"
    if graphs:  # part of the set-up, like building the network plan: eager call, then capture
        model.use_graphs = True
        step(); step()
    if via_estimator:
        # set-up, like building the plan: the first process on a fresh box pages the pandas / ctypes code of the entry point in while it
        # runs (round 6: 24.4 instead of 22.7 ms per step over the first 23 calls of a box's first process) -- a few untimed calls first
        for _ in range(6):
            step()
    for _ in range(args.warmup):
        step()
    from happypose_amd import ops as _ops
    if not graphs:  # HIP events around the conv stretches of the timed steps themselves
        model.backbone.set_profiling(True)
    fence()
    _ops.profile_mark_reference(device)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        poses = step()
    fence()
    elapsed = time.perf_counter() - t0
    instrumented_ms = None
    if graphs:
        # events cannot be recorded inside a replayed graph: the per-launch timing of the roofline comes from the same K
        # steps run once more eagerly (the launches are identical; only the host's part differs), reported beside it
        model.backbone.set_profiling(True)  # also drops the captured graphs (ops.graph_epoch)
        step()
        model.backbone.profile_collect(); model.backbone.profile_intervals()
        fence()
        _ops.profile_mark_reference(device)
        t_i = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        instrumented_ms = 1e3 * (time.perf_counter() - t_i) / args.steps
    def conv_profile():
        """(union ms, sum ms, launches, algorithmic FLOPs, matrix-pipe FLOPs in fp32-MFMA-time equivalents) of the conv
        launches timed since profiling was switched on.  Union = the time during which ANY conv kernel was running: with
        two lanes the timed stretches of the two networks overlap (the summed kernel time exceeds the wall time)."""
        return (union_ms(model.backbone.profile_intervals()),) + tuple(model.backbone.profile_collect())

    conv_ms, conv_sum_ms, n_launch, conv_flops, mfma_flops = conv_profile()
    model.backbone.set_profiling(False)
    assert torch.isfinite(poses).all()
    if hasattr(model, "numerics_status"):
        assert model.numerics_status() == 0, "the split-fp16 guard fired inside the timed region"

    # all-gather of the refined poses alone (N > 1): [N_local, 18] fp32 rows over RCCL
    all_gather_us = None
    if world > 1:
        g_poses = poses[rank * B:(rank + 1) * B].contiguous() if poses.shape[0] == world * B else poses
        for _ in range(3):
            D.gather_poses(g_poses, None, rank * B, world * B)
        fence()
        t_ag = time.perf_counter()
        for _ in range(20):
            D.gather_poses(g_poses, None, rank * B, world * B)
        fence()
        all_gather_us = (time.perf_counter() - t_ag) / 20 * 1e6

    # the same job the OTHER way in (see `via_estimator` above), half as many steps
    estimator = None
    if entry_step is not None:
        def direct_step():
            return model.forward(images, K, labels, TCO0, n_iterations=N_ITERS, im_ids=im_ids)[f"iteration={N_ITERS}"].TCO_output
        estimator = side_timing(direct_step if via_estimator else entry_step, device, max(3, args.steps // 2), poses, B)

    ranks_block = None
    if world > 1:  # what the collective actually ran on: gathered from every rank
        import ctypes

        bus = ctypes.create_string_buffer(64)
        try:
            hip = ctypes.CDLL("libamdhip64.so")
            hip.hipDeviceGetPCIBusId(bus, 64, ctypes.c_int(device.index or 0))
        except OSError:
            pass
        mine = {"rank": rank, "local_rank": local_rank, "device": str(device), "name": torch.cuda.get_device_name(device),
                "pci_bus_id": bus.value.decode() or None, "hostname": os.uname().nodename, "pid": os.getpid()}
        gathered = [None] * world
        torch.distributed.all_gather_object(gathered, mine)
        ranks_block = {"backend": torch.distributed.get_backend(), "world_size": torch.distributed.get_world_size(),
                       "ranks": gathered, "distinct_devices": len({(g["hostname"], g["pci_bus_id"]) for g in gathered})}

    # the same job restricted to the exact-fp32 kernels (fp32 MFMA: Winograd / direct), a quarter of the steps, so the
    # line also carries the number of the build whose every multiply is an fp32 FMA (reported beside `value`)
    exact = None
    if precision == "f32" and not args.no_exact_fp32:
        model.backbone.set_conv_algo("winograd")  # this network only
        k_exact = max(2, args.steps // 4)
        step()
        model.backbone.set_profiling(True)
        fence()
        _ops.profile_mark_reference(device)
        t1 = time.perf_counter()
        for _ in range(k_exact):
            poses_exact = step()
        fence()
        exact = (time.perf_counter() - t1, k_exact, float((poses_exact - poses).abs().max())) + conv_profile()
        model.backbone.set_profiling(False)
        model.backbone.set_conv_algo(None)

    if world > 1:
        t = torch.tensor([elapsed, exact[0] if exact else 0.0], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t[0].item())
        if exact:
            exact = (float(t[1].item()),) + exact[1:]

    parity_failed = False
    if rank == 0:
        total = world * B * args.steps
        sec = conv_ms * 1e-3
        algorithmic = conv_flops / sec / 1e12 if sec > 0 else 0.0
        # matrix-pipe work in units of the instruction that dominates it.  net.cpp counts executed MFMA FLOPs (padded
        # tiles and the 3 MFMAs per product of the split scheme included) in fp32-MFMA-time equivalents: an fp16 MFMA
        # FLOP occupies the pipe 1/16 as long as an fp32 one.  fp16-rate units = x16.
        # fp16-rate units = x16 (the fp16 plan's count is in fp16 FLOPs already).
        executed_f16 = (16.0 if precision == "f32" else 1.0) * mfma_flops / sec / 1e12 if sec > 0 else 0.0
        traffic, traffic_src = recorded_traffic("conv") if (args.workload == "C2" and args.arch == "resnet34" and precision == "f32") else (None, None)
        desc = {
            "C2": f"C2: CosyPose refiner, one 640x480 frame per GPU, {N_DET} detections x {N_HYP} hypotheses = {B} "
                  f"hypotheses/GPU, {N_ITERS} iterations, {args.arch}{' (WideResNet)' if 'resnet' in args.arch else ''} on 6x240x320",
            "C3": f"C3: MegaPose RGB-D refiner, one 640x480 RGB-D frame per GPU, {B} hypotheses/GPU, {N_ITERS} iterations, "
                  "4 views x (RGB + normals + depth), vanilla_resnet34 on 32x240x320",
            "C5": f"C5: MegaPose coarse scoring, 8 objects x 576 SO(3)-grid "
                  f"poses = {B} views/GPU, RGB + normals, vanilla_resnet34 on 9x240x320",
        }[args.workload] + ", 8 objects of 8.2k vertices / 16.1k faces, 1024^2 textures"
        line = {
            "metric": "refined poses/sec (640x480, 16 hyp/det, 5 refiner iters)" if args.workload != "C5"
                      else "coarse-scoring views/sec (640x480, 576 SO(3)-grid poses / object)",
            "value": total / elapsed, "unit": "refined poses/s" if args.workload != "C5" else "views/s",
            "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": precision, "data": "synthetic",
            "config": {"workload": desc + (", renders in the reference's state (4x MSAA, mipmap + anisotropic-16)" if args.render_state == "reference"
                                           else ", single-sample bilinear renders") +
                       (f", {n_lanes} lanes (shares of the batch as independent chains on their own streams)" if n_lanes >= 2 else "") +
                       (", the 5-iteration step replayed as one captured hipGraph" if graphs else ""),
                       "hypotheses_per_gpu": B, "iterations": N_ITERS if args.workload != "C5" else 1,
                       "parallelism": f"hypothesis-shard x{world}" + (f" (one {world * B}-hypothesis batch, distributed.refine_sharded)" if sharded else "")},
            # The roofline of the instruction that is issued: v_mfma_f32_32x32x16_f16 (dense fp16 MFMA peak 2516.6 TFLOP/s).
            # achieved = fp16 MFMA FLOPs the matrix cores EXECUTE per second of conv time (for the fp32 path: three fp16
            # MFMAs per fp32 product, padded tiles included) -> frac = busy fraction of the matrix pipe, always <= 1.
            # algorithmic_tflops = direct-convolution 2*MAC FLOPs of SURVEY.md 8(d) per second, reported beside it.
            "roofline": {"bound": "mfma",
                         "kernel": ("conv3x3_split_f32 / conv3x3s2_split_f32 / conv_igemm_split_f32 / conv_stem5x5s2_pool_split: fp32 "
                                    "operands as fp16 hi/lo halves, three v_mfma_f32_32x32x16_f16 per product, fp32 accumulate"
                                    if precision == "f32" else
                                    "conv_igemm_f16 / conv3x3_patch_f16: v_mfma_f32_32x32x16_f16 on fp16 operands, fp32 accumulate")
                                   + "; all conv launches of a forward",
                         "achieved": executed_f16, "peak": PEAK_F16_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": executed_f16 / PEAK_F16_MFMA_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_tflops": algorithmic,
                         "algorithmic_ceiling_tflops": PEAK_F16_MFMA_TFLOPS / (3.0 if precision == "f32" else 1.0),
                         "launches": n_launch, "avg_launch_us": 1e3 * conv_sum_ms / max(n_launch, 1),
                         "lanes": n_lanes, "conv_busy_ms_per_step": conv_ms / args.steps,
                         "note": ("achieved = executed fp16-MFMA FLOPs / conv-busy time; conv-busy = the union over both lanes of the "
                                  "timed conv stretches (HIP events on the launch streams; the two lanes run concurrently, so a "
                                  "launch shares the machine with the other lane's and avg_launch_us is its duration as it ran); "
                                  "traffic: HBM bytes per conv launch from the rocprofv3 PMC passes of this command committed under "
                                  "profiles/ (traffic_source; FETCH_SIZE x 2 + WRITE_SIZE) -- counters cannot be read inside the run"),
                         "conv_time_share": conv_ms * 1e-3 / (elapsed if instrumented_ms is None else instrumented_ms * 1e-3 * args.steps)},
        }
        if instrumented_ms is not None:
            line["roofline"]["timed_in"] = ("a second, eager pass of the same %d steps (%.2f ms/step): HIP events cannot be recorded inside "
                                            "a replayed graph; `value` is the graph-replayed pass" % (args.steps, instrumented_ms))
            line["eager_ms_per_step"] = instrumented_ms
        # what the matrix pipe SUSTAINS on this box (hp_probe_mfma_rate, measured now): gfx950 clocks to its power
        # budget, so back-to-back fp16 MFMAs on random operands settle at ~1.6 GHz and ~2/3 of the nominal dense peak --
        # the ceiling any dense fp16-MFMA kernel on real data is bound by
        try:
            sus_tf, sus_mhz = _ops.probe_mfma_rate(device, random_data=True)
            nom_tf, nom_mhz = _ops.probe_mfma_rate(device, random_data=False)
            line["roofline"]["sustained"] = {
                "random_operands": {"tflops": sus_tf, "shader_mhz": sus_mhz}, "zero_operands": {"tflops": nom_tf, "shader_mhz": nom_mhz},
                "frac_of_sustained": executed_f16 / sus_tf if sus_tf > 0 else None,
                "note": "hp_probe_mfma_rate in this run: v_mfma_f32_32x32x16_f16 back to back on every SIMD, operands in registers; "
                        "the power budget, not the issue rate, sets the clock (zeros reach the nominal peak)"}
        except Exception as e:  # diagnostics only
            line["roofline"]["sustained"] = {"error": str(e)}
        if all_gather_us is not None:
            line["all_gather_us"] = all_gather_us
        if ranks_block is not None:
            line["ranks"] = ranks_block
        if estimator is not None:
            # `value` is timed through `entry`; `estimator` holds both ways in and the entry point's cost over the bare predictor
            est_ms, pred_ms = (line["ms_per_step"], estimator["ms_per_step"]) if via_estimator else (estimator["ms_per_step"], line["ms_per_step"])
            line["entry"] = ("CosyPoseEstimator.forward_refiner(observation, data_TCO_input[128], n_iterations=5), bsz_objects=128" if via_estimator
                             else "CosyPosePosePredictor.forward (two lanes)")
            line["estimator_ms_per_step"] = est_ms
            line["estimator"] = {"entry_ms_per_step": est_ms, "predictor_ms_per_step": pred_ms, "entry_value": B / (est_ms * 1e-3),
                                 "predictor_value": B / (pred_ms * 1e-3), "unit": "refined poses/s", "overhead_vs_predictor": est_ms / pred_ms - 1.0,
                                 "headline_is": "entry" if via_estimator else "predictor", "side_steps": estimator["steps"],
                                 "max_abs_pose_diff_between_them": estimator["max_abs_pose_diff_vs_headline"],
                                 "entry": "CosyPoseEstimator.forward_refiner(observation, data_TCO_input[128], n_iterations=5), bsz_objects=128 "
                                          "(pandas bookkeeping, per-iteration result tables, the numerical guard's status query)"}
        line["arithmetic"] = ("3xf16-split products (22 significant bits), fp32 accumulate" if precision == "f32" else "fp16 products, fp32 accumulate")
        if precision == "f32":
            line["dtype_note"] = ("fp32 tensors and fp32 accumulation everywhere; the convolutions multiply fp16 hi/lo halves of the fp32 "
                                  "operands (three fp16 MFMAs per product, 22 significant bits: per-layer error vs fp64 as the exact-fp32 "
                                  "kernels, 2e-5 of max|ref|), guarded against activations beyond the fp16 range (hp_net_status); "
                                  "exact_fp32_kernels = the same job on fp32 MFMA only")
        if exact:
            e_sec = exact[3] * 1e-3
            e_exec = exact[7] / e_sec / 1e12 if e_sec > 0 else 0.0
            line["exact_fp32_kernels"] = {
                "value": world * B * exact[1] / exact[0], "unit": line["unit"], "steps": exact[1], "ms_per_step": 1e3 * exact[0] / exact[1],
                "max_abs_pose_diff_vs_default": exact[2],
                "roofline": {"bound": "mfma", "kernel": "conv3x3_wino8_f32 / conv3x3_patch_f32 / conv_igemm_f32: v_mfma_f32_16x16x4_f32 / "
                                                        "v_mfma_f32_32x32x2_f32",
                             "achieved": e_exec, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": e_exec / PEAK_F32_MFMA_TFLOPS,
                             "algorithmic_tflops": exact[6] / e_sec / 1e12 if e_sec > 0 else 0.0,
                             "note": "executed fp32 MFMA FLOPs (Winograd layers execute 2.25x fewer than the direct convolution)"},
                "note": "same job with hp_net_set_conv_algo(WINOGRAD): every multiply an fp32 FMA on the fp32 matrix path"}
        # launches of kernels that use scratch (spilled tile variants; they would also keep hipGraph replay off): 0 expected
        line["scratch_launches"] = int(_ops.scratch_launches())
        if args.workload == "C2":
            line["stages"] = stage_rates(store, scene, images[:1], K[:1], TCO0[:B], im_ids[:B], device)
        if args.workload == "C2" and world == 1 and not args.no_extra_workloads:
            # the other single-GPU configurations of BASELINE.json, 3 steps each, so that the driver's record carries them
            extras = [("c3", "C3", "f32", None), ("c5", "C5", "f16", None)]
            if args.render_state == "reference" and args.arch == "resnet34" and precision == "f32":
                # the job rounds 1-2 measured (one sample per pixel, bilinear level-0 texturing), for continuity
                extras.append(("c2_single_sample_renders", "C2", "f32", dict(msaa=False, aniso=False)))
            for key, wl, prec, rk in extras:
                try:
                    line[key] = quick_workload(device, wl, prec, args.lanes or DEFAULT_LANES.get(wl, 2), steps=5 if wl == "C2" else 3, renderer_kw=rk)
                except Exception as e:  # never lose the headline line to an extra
                    line[key] = {"error": f"{type(e).__name__}: {e}"}
            if args.arch == "resnet34" and precision == "f32":
                # C2 on the backbone the released CosyPose checkpoints use (CP/models/efficientnet.py), 3 steps
                try:
                    # 153 launches of 5 - 130 us per forward and lane: the one workload whose step is bound by launch hand-over enough for
                    # graph replay to matter (round 5: 3083 vs 2913 poses/s) -- replayed, with the eager number beside it
                    line["c2_efficientnet_b3"] = quick_workload(device, "C2", "f32", args.lanes or EFFICIENTNET_LANES, steps=3, arch="efficientnet-b3", graphs=True)
                except Exception as e:
                    line["c2_efficientnet_b3"] = {"error": f"{type(e).__name__}: {e}"}
                # the whole PoseEstimator.run_inference_pipeline on one frame (bench.py --workload E2E), detector included
                try:
                    import gc

                    gc.collect()
                    e2e_products = {}
                    for key, cprec in (("e2e", "f32"), ("e2e_f16_coarse", "f16")):
                        # the second one: the coarse / scoring model on the fp16 plan (BASELINE.json config 5 names fp16 for that
                        # stage), the refiner in fp32 as always
                        r = e2e_run(device, 0, 1, args.lanes or DEFAULT_LANES["E2E"], cprec, steps=3, warmup=2, run_detector=True)
                        line[key] = {"value": r["steps"] / r["elapsed"], "unit": "frames/s", "ms_per_frame": 1e3 * r["elapsed"] / r["steps"],
                                     "lanes": args.lanes or DEFAULT_LANES["E2E"],
                                     "steps": r["steps"], "stage_ms_per_frame": r["stage_ms_per_frame"],
                                     "conv_time_share": r["conv_union_ms"] * 1e-3 / r["elapsed"], "coarse_precision": cprec,
                                     "job": "PoseEstimator.run_inference_pipeline: Mask-RCNN detector (random weights, timed, its detections "
                                            "discarded) + 8 detections x 576-pose coarse grid, top-5, 5 refiner iterations x 40 hypotheses x 4 "
                                            "views, re-scoring, top-1; chunks of bsz_objects = 16 / bsz_images = 576 (the reference's InferenceConfig); pandas bookkeeping included"}
                        e2e_products[key] = (r["product"], r["job_inputs"])
                        gc.collect()
                    if not args.no_cpu_baseline and not args.no_e2e_parity and world == 1:
                        # final-pose parity of BOTH end-to-end runs against ONE run of the oracle estimator on the same job
                        ref, secs = e2e_oracle(e2e_products["e2e"][1])
                        for key, cprec in (("e2e", "f32"), ("e2e_f16_coarse", "f16")):
                            line[key]["parity"] = dict(e2e_parity(e2e_products[key][0], ref, cprec), oracle_seconds=round(secs, 1))
                except Exception as e:
                    line.setdefault("e2e", {"error": f"{type(e).__name__}: {e}"})
                    line.setdefault("e2e_f16_coarse", {"error": f"{type(e).__name__}: {e}"})
        if not args.no_cpu_baseline and args.workload == "C2" and world == 1:  # the CPU baseline is a 1-GPU-run item
            rk = dict(msaa=renderer.msaa, aniso=renderer.aniso)  # the same render state on both sides
            base, cpu_poses = cpu_baseline(store, scene, weights, args.arch, args.cpu_seconds, render_kw=rk)
            line["cpu_baseline"] = base
            line["speedup_vs_cpu"] = line["value"] / base["value"]
            # same-run CPU <-> GPU parity: the oracle's final poses of its sample vs the timed HIP path's
            line["parity"] = pose_parity(poses[:B].cpu().numpy(), cpu_poses)
            parity_failed = not line["parity"]["ok"]
            if not args.no_cpu_1thread:
                line["cpu_baseline_1thread"] = cpu_baseline(store, scene, weights, args.arch, args.cpu_seconds / 2, cores=1, render_kw=rk)[0]
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    if parity_failed:
        sys.exit("bench.py: the HIP path's poses differ from the CPU oracle's beyond the stated tolerance (see \"parity\")")


if __name__ == "__main__":
    main()
