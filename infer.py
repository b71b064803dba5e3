#!/usr/bin/env python
"""EPOS inference on MI355X -- drop-in for ``scripts/infer.py`` of thodan/epos.

    python infer.py --model=<model_name> [flags]                       (infer.py:6-8)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 \
        --master-addr 127.0.0.1 infer.py --model=<model_name> ...

Same contract as the reference script:
  * env TF_DATA_PATH / TF_MODELS_PATH / BOP_PATH (config.py:9-16);
  * <TF_MODELS_PATH>/<model>/params.yml overrides flag defaults (common.py:157-177,
    list-valued crop sizes parsed from "w,h" strings);
  * weights from <model>/train/ -- the TensorFlow checkpoint itself
    (model.ckpt-N.index/.data-*, latest or --checkpoint_name, read without
    TensorFlow by epos_amd/tf_checkpoint.py) or an ``.npz`` keyed by the TF
    variable names; fragments from <model>/fragments.pkl
    (datagen.py:254-268) or <model>/fragments.npz;
  * poses written to <model>/infer/estimated-poses[_<infer_name>].csv in BOP'19
    format (infer.py:753-760); optional corr_*/NNNNNN_corr_OO.txt (infer.py:294-345).

Input frames: ``--infer_tfrecord_names a,b`` reads <TF_DATA_PATH>/<name>.tfrecord
like the reference (infer.py:581-583) but without TensorFlow (epos_amd/tfrecord.py:
TFRecord framing + tf.Example parsing + PIL decoding; frames that would need the
reference's resize raise). Alternatively ``--frames <dir>`` holding ``frames.json``
(list of {scene_id, im_id, path, K[9], targets {obj_id: count}}) + images (.npy
HxWx3 or anything PIL reads), or ``--synthetic N`` seeded synthetic frames.

Reference quirks kept on purpose: only the *localization* task has targets (the
reference dereferences gt_poses=None in detection mode, infer.py:400); here
--task_type=detection fits every object of the model store with unlimited
instances. ``infer_crop_size`` is (width, height) as consumed by the reference
(datagen.py:448-449), despite its help string.
"""
import argparse
import glob
import json
import os
import pickle
import sys
import time

# One hardware queue per pipeline stream: with the runtime's default of 4 queues the null
# stream and the four pipeline streams (--pipeline_depth 4) make five, two pipelines share a
# queue and serialise against each other (bench.py sets the same; without it this entry point
# ran at 0.81 of bench.py's rate, profiles/r06/infer_end_to_end.txt). Must be in the
# environment before the HIP runtime initialises, i.e. before torch is imported.
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

import numpy as np     # noqa: E402
import torch           # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

from epos_amd import bop_io, dist as edist, fitting, model, pipeline   # noqa: E402
from epos_amd import synthetic, weights                                # noqa: E402

PARAMS_FILENAME = 'params.yml'   # common.py


# "all found" (num_instances = -1, detection) needs a bound for the static buffers
# (--detection_instance_cap); every (frame, object) that reaches it is reported
# (EposPipeline.cap_hits, printed per frame and summed up at the end of the run). This is a
# DEVIATION from the reference, where -1 is unbounded (scripts/infer.py:463-468:
# min(-1, max_instances_to_fit) == -1, so that flag never limits detection either -- nor
# does it here).
DETECTION_INSTANCE_CAP = 16


def str2bool(v):
  return str(v).lower() in ('1', 'true', 'yes', 'y')


def build_parser():
  ap = argparse.ArgumentParser(description=__doc__,
                               formatter_class=argparse.RawTextHelpFormatter)
  a = ap.add_argument
  # scripts/infer.py:37-120
  a('--master', default='',
    help='accepted and ignored (scripts/infer.py:38-40: BNS name of a TensorFlow master)')
  a('--model', required=True)
  a('--cpu_only', type=str2bool, default=False)
  a('--task_type', default=pipeline.LOCALIZATION)
  a('--infer_tfrecord_names', default=None)
  a('--infer_max_height_before_crop', type=int, default=480)
  a('--infer_crop_size', default='640,480')
  a('--checkpoint_name', default=None)
  a('--project_to_surface', type=str2bool, default=False)
  a('--save_estimates', type=str2bool, default=True)
  a('--save_corresp', type=str2bool, default=False)
  a('--infer_name', default=None)
  a('--fitting_method', default='progressive_x')
  a('--inlier_thresh', type=float, default=4.0)
  a('--neighbour_max_dist', type=float, default=20.0)
  a('--min_hypothesis_quality', type=float, default=0.5)
  a('--required_progx_confidence', type=float, default=0.5)
  a('--required_ransac_confidence', type=float, default=1.0)
  a('--min_triangle_area', type=float, default=0.0)
  a('--use_prosac', type=str2bool, default=False)
  a('--max_model_number_for_pearl', type=int, default=5)
  a('--spatial_coherence_weight', type=float, default=0.1)
  a('--scaling_from_millimeters', type=float, default=0.1)
  a('--max_tanimoto_similarity', type=float, default=0.9)
  a('--max_correspondences', type=int, default=None)
  a('--max_instances_to_fit', type=int, default=None)
  a('--detection_instance_cap', type=int, default=DETECTION_INSTANCE_CAP)   # not in the reference
  a('--max_fitting_iterations', type=int, default=400)
  a('--vis', type=str2bool, default=False)
  a('--vis_gt_poses', type=str2bool, default=True)           # infer.py:126-146
  a('--vis_pred_poses', type=str2bool, default=True)
  a('--vis_gt_obj_labels', type=str2bool, default=True)
  a('--vis_pred_obj_labels', type=str2bool, default=True)
  a('--vis_pred_obj_confs', type=str2bool, default=False)
  a('--vis_gt_frag_fields', type=str2bool, default=False)
  a('--vis_pred_frag_fields', type=str2bool, default=False)
  # epos_lib/common.py:60-154 (the model flags the hot path reads)
  a('--dataset', default=None)
  a('--num_frags', type=int, default=64)
  a('--min_visib_fract', type=float, default=0.1)
  a('--corr_min_obj_conf', type=float, default=0.1)
  a('--corr_min_frag_rel_conf', type=float, default=0.5)
  a('--corr_project_to_model', type=str2bool, default=False,
    help='accepted and ignored, as in the reference: common.py:78-80 defines the flag and '
         'nothing reads it (the projection switch that acts is --project_to_surface)')
  a('--model_variant', default='xception_65')
  a('--atrous_rates', default='12,24,36')
  a('--encoder_output_stride', type=int, default=8)
  a('--decoder_output_stride', default='4')
  a('--upsample_logits', type=str2bool, default=False)
  a('--frag_cls_agnostic', type=str2bool, default=False)
  a('--frag_loc_agnostic', type=str2bool, default=False)
  a('--multi_grid', default=None,
    help='e.g. 1,2,4 for the resnet_v1_*_beta checkpoints (common.py:111-115)')
  # common.py:96-154: known to the reference, supported here at their defaults only
  # (check_supported_flags raises otherwise -- a params.yml must not be half-applied)
  a('--logits_kernel_size', type=int, default=1)
  a('--image_pyramid', default=None)
  a('--add_image_level_feature', type=str2bool, default=True)
  a('--image_pooling_stride', default='1,1')
  a('--aspp_with_batch_norm', type=str2bool, default=True)
  a('--aspp_with_separable_conv', type=str2bool, default=True)
  a('--depth_multiplier', type=float, default=1.0)
  a('--divisible_by', type=int, default=None)
  a('--decoder_use_separable_conv', type=str2bool, default=True)
  a('--merge_method', default='max')
  a('--prediction_with_upsampled_logits', type=str2bool, default=True)
  a('--use_bounded_activation', type=str2bool, default=False)
  # this build
  a('--frames', default=None, help='directory with frames.json + images')
  a('--synthetic', type=int, default=0, help='number of synthetic frames')
  a('--num_objs', type=int, default=None,
    help='object channels (default: from the checkpoint)')
  a('--batch', type=int, default=1, help='images per GPU per step')
  a('--seed', type=int, default=0)
  a('--pipeline_depth', type=int, default=0,
    help='batches in flight (independent plans on their own HIP streams, each with its own '
         'per-stage timers). 0 (default) = bench.py\'s rule: 4 at one image per batch, 2 from '
         'four images per batch on, 3 in between; 1 = strictly one batch at a time (the '
         'reference\'s loop). --vis, --save_corresp and the operator path (--use_prosac, '
         '--max_correspondences, --project_to_surface) read the plan\'s buffers after every '
         'step and always run at depth 1. The poses do not depend on the depth.')
  a('--launch_queue', type=int, default=int(os.environ.get('EPOS_LAUNCH_QUEUE', '2')),
    help='batches enqueued per plan before the oldest is collected: with 2 a plan\'s next '
         'batch is already in its stream when the current one finishes (the stream does not wait '
         'for the host between two batches). 1 whenever --vis / --save_corresp / the operator '
         'path read the plan\'s buffers after a step. The poses do not depend on it.')
  a('--sparse_heads', default='auto',
    help='evaluate the fragment heads only for the target objects of each image. '
         'corresp.establish_many_to_many never reads the other objects\' channels in '
         'localization mode (corresp.py:39-43), so the poses are bit-identical and ~7 %% of the '
         'step is saved. auto (default) = on for --task_type=localization unless --vis, '
         '--save_corresp or the operator path need the dense prediction; true / false force it')
  a('--decode_threads', type=int, default=0,
    help='decoder processes working ahead of the GPU (0 = min(8, cores - 2); '
         'EPOS_DECODE_PROCS=0 makes them in-process threads)')
  a('--prefetch', type=int, default=6, help='batches decoded ahead of the GPU')
  return ap


def update_flags(args, params_path):
  """common.py:157-177: YAML values override flag DEFAULTS."""
  if not os.path.exists(params_path):
    return
  if os.path.basename(params_path).split('.')[1] not in ['yml', 'yaml']:
    raise ValueError('Only YAML format is currently supported.')
  import yaml
  with open(params_path, 'r') as f:
    params = yaml.safe_load(f) or {}
  for name, val in params.items():
    if hasattr(args, name):
      setattr(args, name, val)


def _as_list(v, cast):
  if v is None:
    return None
  if isinstance(v, (list, tuple)):
    return [cast(x) for x in v]
  return [cast(x) for x in str(v).strip('[]()').split(',') if str(x).strip()]


# Flags of common.py:60-154 the network plan implements at ONE value only. A model
# trained with another value has a different graph (other layers, other head layout),
# so running it through this plan would silently produce garbage: raise instead.
_FIXED_FLAGS = [
    ('upsample_logits', False, 'model.py:661-672: logits stay at the decoder stride'),
    ('frag_cls_agnostic', False, 'common.py:198-202: per-object fragment heads only'),
    ('frag_loc_agnostic', False, 'common.py:61-66: per-object fragment heads only'),
    ('logits_kernel_size', 1, 'model.py:428-431'),
    ('add_image_level_feature', True, 'model.py:217-226'),
    ('aspp_with_batch_norm', True, 'model.py:187-199'),
    ('aspp_with_separable_conv', True, 'model.py:243-256'),
    ('decoder_use_separable_conv', True, 'model.py:369-392'),
    ('use_bounded_activation', False, 'model.py:202,317: ReLU, not ReLU6'),
    ('depth_multiplier', 1.0, 'MobileNet only'),
    ('divisible_by', None, 'MobileNet only'),
]


def check_supported_flags(args):
  """Raises NotImplementedError for a known common.py flag set (on the command line
  or by params.yml) to a value this build's network plan does not implement."""
  bad = []
  for name, want, why in _FIXED_FLAGS:
    got = getattr(args, name)
    if isinstance(want, bool):
      got = str2bool(got) if not isinstance(got, bool) else got
    if got != want and not (want is None and got in (None, 'None', '')):
      bad.append('%s=%r (supported: %r; %s)' % (name, getattr(args, name), want, why))
  pyr = _as_list(args.image_pyramid, float)
  if pyr not in (None, [], [1.0]):
    bad.append('image_pyramid=%r (single scale only, model.py:545-546,597)' % (
        args.image_pyramid,))
  if _as_list(args.image_pooling_stride, int) not in ([1, 1],):
    bad.append('image_pooling_stride=%r (supported: 1,1)' % (args.image_pooling_stride,))
  if args.model_variant not in ('xception_65', 'resnet_v1_101_beta'):
    bad.append('model_variant=%r (xception_65, resnet_v1_101_beta)' % args.model_variant)
  if int(args.encoder_output_stride) != 8:
    bad.append('encoder_output_stride=%r (supported: 8)' % args.encoder_output_stride)
  if _as_list(args.decoder_output_stride, int) != [4]:
    bad.append('decoder_output_stride=%r (supported: 4)' % (args.decoder_output_stride,))
  if bad:
    raise NotImplementedError(
        'flags outside what this build implements (common.py:60-154): ' + '; '.join(bad))


def load_fragments(model_dir, num_frags):
  """fragments.pkl (datagen.py:254-268) or fragments.npz."""
  pkl = os.path.join(model_dir, 'fragments.pkl')
  npz = os.path.join(model_dir, 'fragments.npz')
  if os.path.exists(pkl):
    with open(pkl, 'rb') as f:
      fr = pickle.load(f)
    centers, sizes = fr['frag_centers'], fr['frag_sizes']
  elif os.path.exists(npz):
    z = np.load(npz)
    centers = {int(o): z['frag_centers'][i] for i, o in enumerate(z['obj_ids'])}
    sizes = {int(o): z['frag_sizes'][i] for i, o in enumerate(z['obj_ids'])}
  else:
    return None
  for o in centers:                                   # datagen.py:264-268
    if centers[o].shape[0] != num_frags or sizes[o].shape[0] != num_frags:
      raise ValueError('The loaded fragmentation is not valid.')
  store = synthetic.ModelStore(0, num_frags)
  store.dp_model = {'obj_ids': sorted(int(o) for o in centers)}
  store.frag_centers = {int(o): np.asarray(v, np.float64) for o, v in centers.items()}
  store.frag_sizes = {int(o): np.asarray(v, np.float64) for o, v in sizes.items()}
  return store


def fragment_from_bop_models(model_dir, args, dev):
  """datagen.py:238-296: no fragments.pkl yet -> load the object models of the
  dataset (<BOP_PATH>/<dataset>/models[_<type>]/obj_XXXXXX.ply; 'reconst' for T-LESS,
  'dense' for ITODD, 'eval' for TUD-L, the original ones otherwise), fragment them by
  furthest-point sampling on the GPU, save fragments.pkl next to params.yml."""
  from epos_amd import fragment, ply
  dataset = args.dataset
  bop = os.environ.get('BOP_PATH')
  if not dataset or not bop or dataset not in ply.BOP_OBJ_IDS:
    return None
  mtype = {'tless': 'reconst', 'itodd': 'dense', 'tudl': 'eval'}.get(dataset)
  if not os.path.exists(ply.model_path(bop, dataset, ply.BOP_OBJ_IDS[dataset][0], mtype)):
    return None
  models = ply.load_models(bop, dataset, mtype)
  centers, sizes = fragment.fragment_models(
      {o: m['pts'] for o, m in models.items()}, args.num_frags, device=dev)
  fragment.save_fragments(os.path.join(model_dir, 'fragments.pkl'), centers, sizes)
  return load_fragments(model_dir, args.num_frags)


def find_checkpoint(checkpoint_dir, name):
  if name is not None:
    path = os.path.join(checkpoint_dir, name)
    if not path.endswith('.npz'):
      path += '.npz'
    return path
  cands = sorted(glob.glob(os.path.join(checkpoint_dir, '*.npz')),
                 key=os.path.getmtime)
  return cands[-1] if cands else None


def load_frames(args, num_objs, rank, world, store_obj_ids=None):
  """Returns this rank's list of epos_amd.frames.Frame (ids, K, targets known; pixels decoded
  on demand by the prefetcher's threads), plus the frame height and width."""
  from epos_amd import frames as eframes
  w, h = [int(x) for x in str(args.infer_crop_size).split(',')][:2] \
      if not isinstance(args.infer_crop_size, (list, tuple)) \
      else args.infer_crop_size[:2]
  if args.infer_tfrecord_names:
    # <TF_DATA_PATH>/<name>.tfrecord for each name (infer.py:581-583,
    # datagen.py:707-723), read without TensorFlow (epos_amd/tfrecord.py).
    names = args.infer_tfrecord_names
    if not isinstance(names, (list, tuple)):
      names = [n for n in str(names).split(',') if n]
    data_path = os.environ.get('TF_DATA_PATH', '.')
    paths = []
    for name in names:
      path = os.path.join(data_path, name + '.tfrecord')
      if not os.path.exists(path):
        raise ValueError('No input files: {}'.format(path))   # datagen.py:720-721
      paths.append(path)
    # min_visib_fract=None: the reference builds its inference Dataset without a
    # visibility filter (scripts/infer.py:614), every annotated instance is a target
    all_frames = eframes.scan_tfrecords(
        paths, (w, h), args.infer_max_height_before_crop,
        store_obj_ids if store_obj_ids else None, crop_seed=args.seed)
    b, e = edist.shard_range(len(all_frames), rank, world)
    frames = all_frames[b:e]
  elif args.frames:
    meta = json.load(open(os.path.join(args.frames, 'frames.json')))
    b, e = edist.shard_range(len(meta), rank, world)
    frames = eframes.frames_from_dir(args.frames, meta[b:e], h, w)
  elif args.synthetic:
    b, e = edist.shard_range(args.synthetic, rank, world)
    frames = eframes.synthetic_frames(range(b, e), h, w, num_objs, 5)
  else:
    raise ValueError(
        'No input files: give --infer_tfrecord_names, --frames <dir> or '
        '--synthetic N.')
  return frames, h, w


def save_correspondences(infer_dir, infer_name, frame, im_ind, corr, pred_time):
  """infer.py:294-345 text dump (sorted by confidence)."""
  scene_id, im_id, K = frame.scene_id, frame.im_id, frame.K
  suffix = '' if infer_name is None else '_' + infer_name
  for obj_id, c in corr.items():
    txt = '# Corr format: u v x y z px_id frag_id conf conf_obj conf_frag\n'
    txt += 'synthetic\n{} {} {} {}\n'.format(scene_id, im_id, obj_id, pred_time)
    for i in range(3):
      txt += '{} {} {}\n'.format(K[i, 0], K[i, 1], K[i, 2])
    txt += '0\n'
    order = np.argsort(c['conf'])[::-1]
    txt += '{}\n'.format(len(order))
    for i in order:
      txt += '{} {} {} {} {} {} {} {} {} {}\n'.format(
          c['coord_2d'][i, 0], c['coord_2d'][i, 1], c['coord_3d'][i, 0],
          c['coord_3d'][i, 1], c['coord_3d'][i, 2], c['px_id'][i],
          c['frag_id'][i], c['conf'][i], c['conf_obj'][i], c['conf_frag'][i])
    path = os.path.join(infer_dir, 'corr' + suffix,
                        '{:06d}_corr_{:02d}.txt'.format(im_ind, obj_id))
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, 'w') as f:
      f.write(txt)


def process_by_operators(pipe, store, imgs, chunk, targets, args, fit):
  """process_image (infer.py:348-554) call by call: predict -> corresp ->
  [PROSAC sort / top-K by confidence, infer.py:425-440] -> find6DPoses per object."""
  from epos_amd import corresp as ecorresp
  t0 = time.time()
  pred = pipe.net.forward(imgs, use_graph=pipe.use_graph)
  torch.cuda.synchronize()
  t1 = time.time()
  poses = []
  t_corr = t_fit = 0.0
  for b, f in enumerate(chunk):
    tc = time.time()
    corr = ecorresp.establish_many_to_many(
        pred['pred_obj_conf'][b], pred['pred_frag_conf'][b],
        pred['pred_frag_loc'][b], list(targets[b]), store, pipe.output_scale,
        args.corr_min_obj_conf, args.corr_min_frag_rel_conf,
        bool(args.project_to_surface),
        args.task_type == pipeline.LOCALIZATION, device=str(pipe.dev))
    t_corr += time.time() - tc
    tf_ = time.time()
    for obj_id, c in corr.items():
      n = c['coord_2d'].shape[0]
      if n < 6:                                           # infer.py:420-422
        continue
      if args.use_prosac:                                 # infer.py:425-428
        order = np.argsort(c['conf'])[::-1]
        c = {k: v[order] for k, v in c.items()}
      if args.max_correspondences is not None and n > args.max_correspondences:
        keep = (np.arange(n) if args.use_prosac else
                np.argsort(c['conf'])[::-1])[:args.max_correspondences]
        c = {k: v[keep] for k, v in c.items()}           # infer.py:431-440
      num_inst = (targets[b].get(obj_id, 1)
                  if args.task_type == pipeline.LOCALIZATION else -1)
      if args.max_instances_to_fit is not None:
        num_inst = min(num_inst, args.max_instances_to_fit)
      if args.fitting_method == 'opencv_ransac':
        # infer.py:505-528: cv2.solvePnPRansac(EPNP) -- ONE instance per object, score 0.0.
        # OpenCV is not in this stack; epos_amd.fitting.solvePnPRansac runs the same
        # algorithm (5-point EPnP sets drawn by cv::RNG, float32 inlier rule, the 0.99
        # confidence bound, EPnP over the inliers) in HIP -- csrc/epnp_ransac.hip.
        ok, r_est, t_est, _ = fitting.solvePnPRansac(
            objectPoints=c['coord_3d'], imagePoints=c['coord_2d'], cameraMatrix=f.K,
            distCoeffs=None, iterationsCount=fit.max_iters,
            reprojectionError=fit.threshold, confidence=0.99,
            flags=fitting.SOLVEPNP_EPNP)
        if ok:
          poses.append({'scene_id': f.scene_id, 'im_id': f.im_id, 'obj_id': obj_id,
                        'R': fitting.Rodrigues(r_est), 't': t_est, 'score': 0.0})
        continue
      est, _, quals = fitting.find6DPoses(
          c['coord_2d'], c['coord_3d'], f.K, threshold=fit.threshold,
          neighborhood_ball_radius=fit.neighborhood_ball_radius,
          spatial_coherence_weight=fit.spatial_coherence_weight,
          scaling_from_millimeters=fit.scaling_from_millimeters,
          max_tanimoto_similarity=fit.max_tanimoto_similarity,
          max_iters=fit.max_iters, conf=fit.conf,
          proposal_engine_conf=fit.proposal_engine_conf,
          min_coverage=fit.min_coverage,
          min_triangle_area=fit.min_triangle_area, min_point_number=6,
          max_model_number=num_inst,
          max_model_number_for_optimization=fit.max_model_number_for_optimization,
          use_prosac=args.use_prosac,
          seed=args.seed * 1000003 + f.im_id * 1009 + obj_id)
      if est is not None:                                 # infer.py:490-503
        for i in range(est.shape[0] // 3):
          poses.append({'scene_id': f.scene_id, 'im_id': f.im_id, 'obj_id': obj_id,
                        'R': est[3 * i:3 * i + 3, :3],
                        't': est[3 * i:3 * i + 3, 3].reshape(3, 1),
                        'score': float(quals[i])})
    t_fit += time.time() - tf_
  rt = {'prediction': t1 - t0, 'establish_corr': t_corr, 'fitting': t_fit}
  rt['total'] = sum(rt.values())
  for p in poses:
    p['time'] = rt['total']
  return poses, rt


def main(argv=None):
  args = build_parser().parse_args(argv)
  rank, world, local_rank = edist.init_from_env()
  models_path = os.environ.get('TF_MODELS_PATH', '.')          # config.py:9-16
  model_dir = os.path.join(models_path, args.model)
  update_flags(args, os.path.join(model_dir, PARAMS_FILENAME))  # infer.py:561-564
  check_supported_flags(args)
  if args.cpu_only:
    raise SystemExit('--cpu_only: this build has no CPU path (MI355X only).')
  if args.fitting_method not in ('progressive_x', 'opencv_ransac'):
    raise ValueError('Unknown pose fitting method ({}).'.format(
        args.fitting_method))                                   # infer.py:530-532
  if args.fitting_method == 'opencv_ransac':
    # not bit-for-bit cv2 (which cannot be installed or checked here): say so at run time
    print('NOTE --fitting_method=opencv_ransac: the algorithm cv2.solvePnPRansac(EPNP) '
          'publishes (cv::RNG 5-point sets, EPnP, float32 inlier rule, the shrinking 0.99 '
          'bound with a once-rounded w^5, the float32 round trip of the normalised image '
          'points, EPnP over the inliers) runs in HIP, but its poses are NOT guaranteed to '
          'equal cv2\'s bit for bit: the factorisations are this build\'s own and the '
          'R -> rvec -> R round trip through cv::Rodrigues in front of the inlier test is '
          'not made (DESIGN.md (f2)).')
  if args.vis and args.vis_gt_frag_fields:
    raise NotImplementedError(
        '--vis_gt_frag_fields needs the ground-truth fragment fields of the training '
        'pipeline (datagen.py:478-544), which the inference reader does not build.')
  if args.vis and args.vis_gt_obj_labels:
    # the GT label map comes from the instance masks of the training reader
    # (datagen.py:478-544); the inference reader holds none, so that tile is left out --
    # said once here instead of silently (scripts/infer.py:150-291 draws it)
    print('note: --vis_gt_obj_labels has no ground-truth label map at inference; the tile '
          'is omitted (use --vis_gt_obj_labels=False to silence this)', file=sys.stderr)
  checkpoint_dir = os.path.join(model_dir, 'train')             # infer.py:570
  infer_dir = os.path.join(model_dir, 'infer')
  os.makedirs(infer_dir, exist_ok=True)
  # EPOS_FORCE_DEVICE=0 maps every rank onto one GPU (multi-rank flow on a one-GPU
  # test box, together with EPOS_DIST_BACKEND=gloo), as in bench.py
  dev_index = int(os.environ.get('EPOS_FORCE_DEVICE', local_rank))
  dev = 'cuda:%d' % dev_index
  torch.cuda.set_device(dev_index)

  ckpt_path = find_checkpoint(checkpoint_dir, args.checkpoint_name)
  tf_prefix = None
  if args.checkpoint_name is not None and os.path.exists(
      os.path.join(checkpoint_dir, args.checkpoint_name + '.index')):
    tf_prefix = os.path.join(checkpoint_dir, args.checkpoint_name)
  elif not (ckpt_path and os.path.exists(ckpt_path)):
    from epos_amd import tf_checkpoint
    tf_prefix = tf_checkpoint.latest_checkpoint(checkpoint_dir)  # infer.py:670-674
  if tf_prefix is not None:
    # A TensorFlow checkpoint (model.ckpt-N.index/.data-*), read without TF.
    from epos_amd import tf_checkpoint
    ckpt = tf_checkpoint.to_epos_checkpoint(
        tf_checkpoint.load_checkpoint(tf_prefix))
    num_objs = ckpt['logits/pred_obj_conf/biases'].shape[0] - 1
  elif ckpt_path and os.path.exists(ckpt_path):
    ckpt = weights.load_npz(ckpt_path)
    num_objs = ckpt['logits/pred_obj_conf/biases'].shape[0] - 1
  elif args.synthetic:
    num_objs = args.num_objs or 21
    ckpt = weights.random_init(num_objs=num_objs, num_frags=args.num_frags,
                               seed=0, randomize_bn=True)
  else:
    raise ValueError('No checkpoint (.npz) found in {}'.format(checkpoint_dir))
  store = load_fragments(model_dir, args.num_frags)
  if store is None and not args.synthetic:
    store = fragment_from_bop_models(model_dir, args, dev)
  if store is None:
    if not args.synthetic:
      raise ValueError('fragments.pkl / fragments.npz not found in ' + model_dir +
                       ' and no BOP models under $BOP_PATH/<dataset>/models*')
    store = synthetic.ModelStore(num_objs, args.num_frags, seed=0)

  frames, h, w = load_frames(args, num_objs, rank, world,
                            store.dp_model['obj_ids'])
  atrous = _as_list(args.atrous_rates, int)
  mo = model.ModelOptions(
      model.get_outputs_to_num_channels(num_objs, args.num_frags),
      crop_size=(w, h), atrous_rates=atrous,
      encoder_output_stride=args.encoder_output_stride,
      decoder_output_stride=_as_list(args.decoder_output_stride, int),
      model_variant=args.model_variant,
      multi_grid=_as_list(args.multi_grid, int))
  fit = fitting.fit_params(
      threshold=args.inlier_thresh,
      neighborhood_ball_radius=args.neighbour_max_dist,
      spatial_coherence_weight=args.spatial_coherence_weight,
      scaling_from_millimeters=args.scaling_from_millimeters,
      max_tanimoto_similarity=args.max_tanimoto_similarity,
      max_iters=args.max_fitting_iterations,
      conf=args.required_progx_confidence,
      proposal_engine_conf=args.required_ransac_confidence,
      min_coverage=args.min_hypothesis_quality,
      min_triangle_area=args.min_triangle_area, min_point_number=6,
      max_model_number_for_optimization=args.max_model_number_for_pearl,
      use_prosac=args.use_prosac)
  # max_correspondences / use_prosac (both off by default, infer.py:95-97,115-117)
  # re-order the correspondences by confidence on the host (infer.py:425-440), and
  # project_to_surface (off by default) needs the object meshes, so
  # those runs go operator by operator (HIP network -> HIP correspondences -> host
  # sort -> HIP fitting per object) instead of through the fused device pipeline.
  operator_path = (args.max_correspondences is not None or args.use_prosac or
                   args.project_to_surface)
  if args.project_to_surface:
    # infer.py:622 prepare_for_projection: the 'eval' models of the dataset
    # (datagen.py:250-252,299-306), closest-point queries on the GPU
    from epos_amd import ply
    bop = os.environ.get('BOP_PATH')
    if not (args.dataset and bop):
      raise ValueError('--project_to_surface needs --dataset and $BOP_PATH (object models)')
    store.models = ply.load_models(bop, args.dataset, 'eval',
                                   obj_ids=store.dp_model['obj_ids'])
  B = args.batch
  # Instances per object (infer.py:456-468 of the reference): localization fits as many as
  # the frame's annotations hold -- the plan is sized for the largest count among the frames
  # read, nothing is clamped; --max_instances_to_fit lowers the counts as in the reference
  # (scripts/infer.py:467-468). Detection ("all found", -1): the reference's
  # min(-1, max_instances_to_fit) stays -1, i.e. the flag does not act there; this build
  # bounds "all found" by --detection_instance_cap (static buffers) and reports every
  # (frame, object) that reaches the cap.
  if args.task_type == pipeline.LOCALIZATION:
    max_inst = max([1] + [int(c) for f in frames for c in f.targets.values()])
    if args.max_instances_to_fit is not None:
      max_inst = max(1, min(max_inst, args.max_instances_to_fit))
  else:
    max_inst = max(1, int(args.detection_instance_cap))
  needs_dense = bool(operator_path or args.save_corresp or args.vis)
  depth = args.pipeline_depth if args.pipeline_depth > 0 else (
      4 if B == 1 else 2 if B >= 4 else 3)         # bench.py's rule
  lq = max(1, args.launch_queue)
  if needs_dense:
    depth = 1                      # those paths read the plan's buffers after the step
    lq = 1
  if depth == 1:
    lq = 1                         # strictly one batch at a time means launch -> collect
  sh = str(args.sparse_heads).lower()
  if sh == 'auto':
    sparse_heads = args.task_type == pipeline.LOCALIZATION and not needs_dense
  else:
    sparse_heads = str2bool(sh)
    if sparse_heads and (needs_dense or args.task_type != pipeline.LOCALIZATION):
      raise ValueError('--sparse_heads=true needs --task_type=localization without --vis / '
                       '--save_corresp / the operator path (they read every object\'s heads)')
  # the decoder processes start (import numpy / PIL) while the plans are built; nothing is
  # decoded before the loop below asks for it
  from epos_amd import frames as eframes
  feed = eframes.Prefetcher(frames, B, h, w, workers=args.decode_threads or None,
                            ahead=max(1, args.prefetch), inflight=depth * lq)
  pipes = [pipeline.EposPipeline(
      ckpt, B, h, w, num_objs, args.num_frags, store, fit_params=fit,
      corr_min_obj_conf=args.corr_min_obj_conf,
      corr_min_frag_rel_conf=args.corr_min_frag_rel_conf,
      max_instances=max_inst, model_options=mo, device=dev, instance=j,
      sparse_heads=sparse_heads, fitting_method=args.fitting_method, queue=lq)
           for j in range(depth)]
  pipe = pipes[0]
  if rank == 0:
    print('plan: {} image(s) per step, {} step(s) in flight{}, {} heads, {} GEMM layers on the '
          'fp16-pair kernel ({} on the bf16 x 6 fallback)'.format(
              B, depth, ' x {} enqueued per plan'.format(lq) if lq > 1 else '',
              'sparse' if sparse_heads else 'dense', len(pipe.net.h2_layers),
              len(pipe.net.h2_refused)))

  # Set-up, not inference: every plan captures its hipGraph and loads its kernels on first
  # use, so each one runs once on a blank frame here (results discarded) -- the reference
  # likewise treats its first sess.run as warm-up (scripts/infer.py:741-749 replaces the first
  # image's time by the mean of the others).
  if frames and not operator_path:
    blank = torch.zeros((B, h, w, 3), dtype=torch.uint8).pin_memory()
    f0 = frames[0]
    tg0 = [dict(f0.targets)] * B
    if args.max_instances_to_fit is not None:
      tg0 = [{o: min(c, args.max_instances_to_fit) for o, c in t.items()} for t in tg0]
    import collections
    for q in pipes:
      q.process_batch(blank, np.stack([f0.K] * B), tg0, task_type=args.task_type, seed=args.seed)
      q.cap_hits = collections.deque(maxlen=q.CAP_HITS_KEPT)
      q.last_cap_hits, q.cap_hit_count = [], 0
      q._warned_cap = False
    torch.cuda.synchronize()

  poses_all = []
  time_start = time.time()           # first decode -> CSV written

  def finish(i0, chunk, poses, rt):
    n_real = len(frames[i0:i0 + B])
    real_ids = set((f.scene_id, f.im_id) for f in frames[i0:i0 + n_real])
    seen = set()
    for p in poses:
      p.setdefault('time', rt.get('total', 0.0))
      key = (p['scene_id'], p['im_id'], p['obj_id'], float(p['score']))
      if (p['scene_id'], p['im_id']) in real_ids and key not in seen:
        seen.add(key)
        poses_all.append(p)
    if args.save_corresp:
      from epos_amd import corresp as ecorresp
      pred = pipe.net.outputs()
      for b, f in enumerate(chunk[:n_real]):
        c = ecorresp.establish_many_to_many(
            pred['pred_obj_conf'][b], pred['pred_frag_conf'][b],
            pred['pred_frag_loc'][b], list(f.targets), store, 0.25,
            args.corr_min_obj_conf, args.corr_min_frag_rel_conf, False,
            args.task_type == pipeline.LOCALIZATION, device=dev)
        save_correspondences(infer_dir, args.infer_name, f, i0 + b, c,
                             rt.get('total', 0.0))
    if args.vis:                                # infer.py:540-552, <model>/vis (:577)
      from epos_amd import vis as evis
      pred = {k: v.cpu().numpy() for k, v in pipe.net.outputs().items()}
      flags = {k: getattr(args, k) for k in vars(args) if k.startswith('vis_')}
      for b, f in enumerate(chunk[:n_real]):
        est = [p for p in poses if (p['scene_id'], p['im_id']) == (f.scene_id, f.im_id)]
        evis.visualize(f.image_f32(), f.K, {k: v[b] for k, v in pred.items()}, est, i0 + b,
                       store, os.path.join(model_dir, 'vis'),
                       gt_poses=f.gt_poses, flags=flags)
    if rank == 0:                               # infer.py:730-734
      print('Image: {}, prediction: {:.3f}, establish_corr: {:.3f}, fitting: '
            '{:.3f}, total time: {:.3f}'.format(
                i0, rt.get('prediction', 0), rt.get('establish_corr', 0),
                rt.get('fitting', 0), rt.get('total', 0)))
    feed.release(i0)                            # its staging buffer may be decoded into again

  inflight = []                                 # (pipeline, i0, chunk), oldest first
  # HIP events around the stages (the per-image times the reference prints, infer.py:730-734);
  # EPOS_INFER_TIMING=0 runs without them (diagnostics: tools/infer_diag.sh)
  stage_timing = os.environ.get('EPOS_INFER_TIMING', '1') != '0'
  # where the host's time goes, per loop phase (printed with the throughput line): waiting for
  # decoded frames, waiting for the oldest step in flight, enqueueing a step, bookkeeping
  host_s = {'wait_frames': 0.0, 'wait_gpu': 0.0, 'launch': 0.0, 'finish': 0.0}
  clock = time.perf_counter
  t_mark = clock()
  for step, (i0, chunk, imgs) in enumerate(feed):
    # imgs: pinned host memory, uint8 as decoded (float32 only for frames that are not
    # byte-valued); the upload is enqueued on the step's own stream and the cast to float32
    # (datagen.py:435-436) runs on the device
    t_now = clock(); host_s['wait_frames'] += t_now - t_mark; t_mark = t_now
    Ks = np.stack([f.K for f in chunk])
    tg = [f.targets for f in chunk]
    if args.max_instances_to_fit is not None:  # infer.py:467-468
      tg = [{o: min(c, args.max_instances_to_fit) for o, c in t.items()}
            for t in tg]
    if operator_path:
      poses, rt = process_by_operators(pipe, store, imgs, chunk, tg, args, fit)
      finish(i0, chunk, poses, rt)
      t_mark = clock()
      continue
    if len(inflight) == depth * lq:
      q, j0, ch = inflight.pop(0)
      res = q.collect()
      t_now = clock(); host_s['wait_gpu'] += t_now - t_mark; t_mark = t_now
      finish(j0, ch, *res)
      t_now = clock(); host_s['finish'] += t_now - t_mark; t_mark = t_now
    p = pipes[step % depth]
    p.launch(imgs, Ks, tg, task_type=args.task_type,
             image_ids=[f.im_id for f in chunk], scene_ids=[f.scene_id for f in chunk],
             seed=args.seed, timing=stage_timing)
    inflight.append((p, i0, chunk))
    t_now = clock(); host_s['launch'] += t_now - t_mark; t_mark = t_now
  while inflight:
    q, j0, ch = inflight.pop(0)
    res = q.collect()
    t_now = clock(); host_s['wait_gpu'] += t_now - t_mark; t_mark = t_now
    finish(j0, ch, *res)
    t_now = clock(); host_s['finish'] += t_now - t_mark; t_mark = t_now
  loop_s = time.time() - time_start
  hits = sorted(set(h for q in pipes for h in q.cap_hits))
  n_hits = sum(q.cap_hit_count for q in pipes)
  if hits:
    overflowed = any(q.cap_hit_count > q.CAP_HITS_KEPT for q in pipes)
    print('Instance cap (--detection_instance_cap={}) reached {} time(s), for {} distinct '
          '(scene, image, object) triples{}; more instances may exist there:'.format(
              max_inst, n_hits, len(hits),
              ' among the last {} hits kept per pipeline'.format(pipe.CAP_HITS_KEPT)
              if overflowed else ''))
    for sc_, im_, ob_, n_ in hits:
      print('  scene {} image {} object {}: {} instances'.format(sc_, im_, ob_, n_))
  # First-image time := mean time of the others (infer.py:741-749).
  if len(poses_all) > 1 and frames:
    first = (frames[0].scene_id, frames[0].im_id)
    rest = [p['time'] for p in poses_all if (p['scene_id'], p['im_id']) != first]
    if rest:
      for p in poses_all:
        if (p['scene_id'], p['im_id']) == first:
          p['time'] = float(np.mean(rest))
  # max_records=None: the ranks first agree on the largest local pose count (their
  # shards differ by a frame whenever N % world != 0, and a rank may hold none)
  merged = edist.gather_poses(poses_all, max_records=None) if world > 1 else poses_all
  if rank == 0 and args.save_estimates:
    suffix = '' if args.infer_name is None else '_' + args.infer_name
    path = os.path.join(infer_dir, 'estimated-poses{}.csv'.format(suffix))
    bop_io.save_bop_results(path, merged, version='bop19')
    total_s = time.time() - time_start
    print('Saved {} pose estimates to: {}  ({:.2f} s)'.format(len(merged), path, total_s))
    # from the first decode to the CSV on disk (this rank's frames; with N ranks each rank
    # runs its own shard concurrently, so the job's rate is ~N x this)
    print('Throughput: {} images in {:.3f} s = {:.1f} images/s (inference loop {:.3f} s = '
          '{:.1f} images/s; first decode -> CSV written; plan construction and weight packing '
          'before it are not included)'.format(
              len(frames), total_s, len(frames) / max(total_s, 1e-9), loop_s,
              len(frames) / max(loop_s, 1e-9)))
    if not operator_path:
      n_steps = max(1, (len(frames) + B - 1) // B)
      print('Host time per step [ms]: ' + ', '.join(
          '{} {:.3f}'.format(k, v / n_steps * 1e3) for k, v in host_s.items()))
  if world > 1:
    torch.distributed.destroy_process_group()


if __name__ == '__main__':
  main()
