#!/usr/bin/env python3
"""Predict every frame of a dataset (same command line as the reference's predict_dataset.py:58-64).

    python predict_dataset.py <dataset> [-angs SLU]
    python -m torch.distributed.run --nproc-per-node N ... predict_dataset.py <dataset>    # frames shard over N GPUs

Frames are independent, so rank r predicts a contiguous block of frames on its own GPU and
the only communication is one all-gather of the (n, 6) float64 results (RCCL over xGMI).
Output: predictions_<dataset>.npy, as the reference (predict_dataset.py:47-49).
"""
import argparse
import os

# dmabuf IPC: what RCCL needs on this driver (bench.py says the same); before anything imports torch, whoever started the ranks
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

import numpy as np

from robotpose import Dataset, Grapher, Predictor
from rope_s3d_amd.parallel import dist_env, gather_rows, shard_range


def run(args):
    rank, world, local_rank = dist_env()
    # ROPE_FORCE_DEVICE / ROPE_DIST_BACKEND: rehearse the N>1 flow on a one-GPU box (all ranks on device 0, gloo)
    gpu = int(os.environ.get('ROPE_FORCE_DEVICE', local_rank))
    device = None
    # ROPE_DIST_ALWAYS: a world of one goes through the process group and the gather too (RCCL executed on a one-GPU box)
    use_dist = world > 1 or bool(os.environ.get('ROPE_DIST_ALWAYS'))
    if use_dist:
        import torch
        import torch.distributed as dist
        backend = os.environ.get('ROPE_DIST_BACKEND', 'nccl')
        if backend == 'nccl':
            torch.cuda.set_device(gpu)
            device = torch.device('cuda', gpu)
            dist.init_process_group('nccl', device_id=device)
        else:
            dist.init_process_group(backend)

    import time
    t_begin = time.perf_counter()
    from rope_s3d_amd.data.dataset import open_dataset
    ds = open_dataset(args.dataset, gpu)                 # a stored set, or 'synthetic:<frames>[:<seed>[:<preset>]]' rendered as it is read
    kwargs = {}
    if getattr(args, 'segmenter', None) == 'maskrcnn':
        # BASELINE configs[2]: the segmentation stage on PyTorch-ROCm in front of the engine (predict.py:94-98,416).
        # No trained weights exist offline: without -weights the network is random and only the plumbing and the
        # timing are meaningful.
        import torch
        from rope_s3d_amd.maskrcnn import MaskRCNNSegmenter
        sd = None
        if getattr(args, 'weights', None):                    # a trained Keras file of the reference (models/<set>/*.h5), or a state_dict
            from rope_s3d_amd.maskrcnn import load_matterport_weights
            sd = load_matterport_weights(args.weights) if args.weights.endswith('.h5') else torch.load(args.weights, map_location='cpu')
        from rope_s3d_amd.maskrcnn import BatchAheadSegmenter
        kwargs['segmenter'] = BatchAheadSegmenter(MaskRCNNSegmenter(7, device=f'cuda:{gpu}', state_dict=sd,
                                                                    min_confidence=0.7 if sd is not None else 0.0), batch=8)
    elif getattr(args, 'segmenter', None) == 'color':
        # colour-coded frames through the SEGMENTATION path (_segmentLoad: instance merge, dilate 8 / erode 7 body mask) with
        # exact masks in place of the network's: the whole of configs[2] minus the network's own errors
        from rope_s3d_amd.segmentation import ColorSegmenter
        from rope_s3d_amd.urdf import URDFReader
        kwargs['segmenter'] = ColorSegmenter(['BG'] + list(URDFReader().mesh_names[:6]), ds.attrs.get('color_dict'))
    elif ds.attrs.get('synthetic'):
        kwargs['color_dict'] = ds.attrs['color_dict']         # link masks are read from the colour render
    if getattr(args, 'lookup_divisions', None):                # default: the reference's size rule (simulation/lookup.py)
        kwargs['lookup_divisions'] = int(args.lookup_divisions)
    am = Predictor(ds_factor=args.ds_factor, camera_pose=ds.camera_pose[0], preview=False, base_intrin=ds.intrinsics,
                   do_angles=args.angs, model_ds=args.dataset, device=gpu, **kwargs)
    # Frames are independent (predict_dataset.py:43-44 is a plain loop; fresh state per frame, predict.py:144-148): ONE Predictor
    # takes a chunk's frames through the stage list in lockstep (Predictor.run_many -> rope_predict_batch: every step one device
    # batch over all frames' rows), the next frames prepared on worker threads meanwhile.  -predictors k > 1 is the older way of
    # filling the GPU — k Predictors (own context and stream each) fed by k threads (prediction/pool.py) — kept for comparison.
    n_pred = max(1, int(getattr(args, 'predictors', 1) or 1))
    pool = None
    if n_pred > 1 and am.synthetic:
        from rope_s3d_amd.prediction.pool import PredictorPool
        pool = PredictorPool(n_pred - 1, ds_factor=args.ds_factor, camera_pose=ds.camera_pose[0], preview=False, base_intrin=ds.intrinsics,
                             do_angles=args.angs, model_ds=args.dataset, device=gpu, **kwargs)
        pool.predictors.insert(0, am)
        pool._tune()

    lo, hi = shard_range(ds.length, rank, world)
    out = np.zeros((hi - lo, 6))
    t_ready = time.perf_counter()
    # The reference reads ~200 frames at a time (predict_dataset.py:27-41).  Here a chunk is several lockstep batches, so that inside
    # run_many the preparation and upload of one batch hide behind the stages of the one before; bounded by the bytes of raw
    # frames held (two chunks: the one being predicted and the one being read).
    batch = getattr(args, 'batch', None) or am.default_batch()
    frame_bytes = max(1, int(np.prod(ds.og_img.shape[1:])) * ds.og_img.dtype.itemsize + int(np.prod(ds.depthmaps.shape[1:])) * ds.depthmaps.dtype.itemsize)
    chunk = max(200, min(4 * batch, max(batch, (4 << 30) // frame_bytes)))

    def read(start):
        end = min(start + chunk, hi)
        # slices, not copies (predict_dataset.py:39-41 copies): an HDF5 slice is already a fresh array, and a memory-mapped
        # one is read by the worker that needs it — with ds_factor 8 the down-sampling touches a third of its pages
        return end, ds.og_img[start:end], ds.depthmaps[start:end], np.copy(ds.camera_pose[start:end])

    from concurrent.futures import ThreadPoolExecutor
    reader = ThreadPoolExecutor(max_workers=1)                  # the next chunk comes off the disk while this one is predicted
    nxt = reader.submit(read, lo) if lo < hi else None
    for start in range(lo, hi, chunk):
        end, og_imgs, dms, cam_poses = nxt.result()
        nxt = reader.submit(read, end) if end < hi else None
        seg = None if am.synthetic else getattr(am, 'seg', None)
        if hasattr(seg, 'announce'):                          # the chunk's frames through the network in batches of 8
            seg.announce([am._downsample(og_imgs[i], am.ds_factor) for i in range(end - start)])
        # frame by frame as predict_dataset.py:43-44, with the next frame's host preparation (down-sampling, masks,
        # segmentation) running beside the current frame's device work
        if pool is not None:
            out[start - lo:end - lo] = pool.run_many(og_imgs, dms, cam_poses)
        else:
            out[start - lo:end - lo] = am.run_many(og_imgs, dms, cam_poses, batch=getattr(args, 'batch', None))
    reader.shutdown()
    if rank == 0 and os.environ.get('ROPE_TIMING'):
        t_end = time.perf_counter()
        print(f"rank 0: set-up {t_ready - t_begin:.2f} s (data set, segmenter, Predictor, lookup table), {hi - lo} frames in {t_end - t_ready:.2f} s = "
              f"{(hi - lo) / max(t_end - t_ready, 1e-9):.1f} frames/s")
    full = gather_rows(out, ds.length, device=device, single_rank_too=use_dist)
    if rank == 0:
        if use_dist:
            print(f"{ds.length} frames of {world} rank(s) gathered over {os.environ.get('ROPE_DIST_BACKEND', 'nccl')}")
        np.save(f"predictions_{os.path.basename(os.path.normpath(args.dataset)).replace(':', '_')}.npy", full)
        Grapher(args.angs, full, np.copy(ds.angles)).plot()
    if use_dist:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    return full


if __name__ == "__main__":
    parser = argparse.ArgumentParser()
    parser.add_argument('dataset', type=str, help="The dataset to predict on.")
    parser.add_argument('-angs', type=str, default='SLU', help="The joints to predict.")
    parser.add_argument('-ds_factor', type=int, default=8, help="Downsampling factor (the reference hard-codes 8).")
    parser.add_argument('-segmenter', type=str, default=None, choices=[None, 'maskrcnn', 'color'],
                        help="'maskrcnn': segment every frame with the Mask R-CNN stage instead of reading a synthetic set's colours; "
                             "'color': a colour-coded set through the segmentation path with exact masks.")
    parser.add_argument('-lookup_divisions', type=int, default=None, help="Lookup grid divisions per joint (default: the reference's size rule).")
    parser.add_argument('-predictors', type=int, default=1,
                        help="Predictors (engine contexts + threads) per GPU when the link masks come from the colour render; default 1: "
                             "one Predictor walking -batch frames in lockstep.")
    parser.add_argument('-batch', type=int, default=None,
                        help="Frames that walk the stage list in lockstep (one device batch per step over all of them); default: by frame size, 16..1024; 1: frame after frame.")
    parser.add_argument('-weights', type=str, default=None, help="weights for -segmenter maskrcnn: the reference's trained Keras .h5 or a torch state_dict (random weights otherwise).")
    run(parser.parse_args())
