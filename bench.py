#!/usr/bin/env python3
"""Benchmark of the render-and-compare hot path on MI355X.

Metric (BASELINE.json): rendered+scored candidate poses / second at 640x480.
Workload at every N: BASELINE.json configs[1] — motoman mh5l (limited URDF), one
synthetic 640x480 RGB-D frame per rank, 4096 candidate poses per frame (16^3 S/L/U
lookup grid, robotpose/simulation/lookup.py:56-66 order), depth-only loss (last term of
Predictor._error, predict.py:503-507).  One "step" = FK + raster of 6 links + loss
reduction + argmin for the 4096 candidates of one frame, everything resident in HBM.
Frames shard across ranks (one process per GPU, no data-path collective); the only
exchange is one RCCL all-gather of the ranks' best joint vectors, inside the timed region.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N ...      # without WORLD_SIZE in the environment: starts the N ranks itself (same launcher)

The number of ranks that took part is checked against --gpus on every rank and reported as "ranks_seen"; a mismatch is an error.
"""
import argparse
import json
import os
import sys
import time

# dmabuf IPC: what RCCL (and any sharing of device memory between the ranks' processes) needs on this driver; set before torch
# is imported, for ranks started by the driver's launcher as much as for the ones self_launch starts
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic bytes per candidate, mh5l 6 links, 640x480, f32 depth, depth-only loss
#   12·V + 12·T (indexed mesh read, V=59 167, T=118 466) + 2·W·H·4 (one depth write + one read)
B_CAND = 12 * 59167 + 12 * 118466 + 2 * 640 * 480 * 4          # = 4 589 196
HBM_PEAK_GBS = 8000.0                                           # MI355X_MICROARCH.md: 8.0 TB/s spec


def slu_grid(limits, d):
    divs = np.array([d, d, d, 1, 1, 1])
    num = int(np.prod(divs))
    ang = np.zeros((num, 6))
    for idx in range(3):
        rng = np.linspace(limits[idx, 0], limits[idx, 1], divs[idx])
        repeat = int(np.prod(divs[:idx]))
        ang[:, idx] = np.tile(np.repeat(rng, repeat), num // (repeat * divs[idx]))
    return ang


PROFILE_ROUND = 'r03'          # profiles/<round>_pmc.json, <round>_valu_issue.json, <round>_kernel_clock.json


def committed_profile(name, build_id):
    """profiles/<name> if it belongs to the build being timed: tools/summarize_prof.py stamps the counters with the hash of the
    sources they were taken on (rope_build_id of the library profiled).  -> (dict or None, stale?)"""
    try:
        with open(os.path.join(ROOT, 'profiles', name)) as f:
            d = json.load(f)
    except (OSError, ValueError):
        return None, False
    if d.get('build_id') != build_id:
        return None, True
    return d, False


def valu_issue_peak():
    """SIMD cycles one wave64 VALU instruction costs at the raster kernel's occupancy (6 waves per SIMD), from the committed
    micro-benchmark (tools/valu_issue_bench.hip -> profiles/<round>_valu_issue.json): wall time of a launch of independent
    instructions x the clock held inside that launch (s_memtime / s_memrealtime stamps).  -> {full-rate op, kernel-like mix,
    half-rate op} cycles and the micro-benchmark's own clock; {} if absent.  (Independent of the library's build.)"""
    try:
        with open(os.path.join(ROOT, 'profiles', PROFILE_ROUND + '_valu_issue.json')) as f:
            kinds = {k['instruction']: k for k in json.load(f)['kinds']}
        mix = [v for k, v in kinds.items() if k.startswith('raster-like mix')][0]
        return {'full_rate': kinds['v_add_u32']['cycles_per_wave_instruction_per_simd']['6'],
                'half_rate': kinds['v_mul_i32_i24']['cycles_per_wave_instruction_per_simd']['6'],
                'mix': mix['cycles_per_wave_instruction_per_simd']['6'], 'clock_ghz_microbench': mix['clock_ghz']['6']}
    except (OSError, KeyError, ValueError, IndexError):
        return {}


def self_launch(args):
    """--gpus N > 1 without a launcher's environment: become the launcher.  Runs before anything touches the GPU (the
    children are fresh processes of torch.distributed.run, one rank per GPU), relays their output and exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')       # dmabuf IPC: what RCCL needs on this driver
    return subprocess.call(cmd, env=env)


def cpu_baseline(robot, PV, W, H, znear, zfar, cand, tq, n_sample, gpu_err=None, loss=0, flags=None):
    """The CPU oracle (C restatement of the reference path) on this host's cores, bounded sample."""
    from oracle import oracle as orc
    # the GPU box shows every host core but a 1-GPU job's share is a cgroup quota of 16 CPUs: more threads than that are
    # throttled by the scheduler, not faster
    from rope_s3d_amd.utils import cpu_budget
    threads = min(cpu_budget(), 16)
    o = orc.Oracle(robot.verts, robot.faces, robot.vtx_off, robot.tri_off, robot.joint_fixed, robot.joint_axes,
                   PV, W, H, znear, zfar)
    sample = np.ascontiguousarray(cand[:n_sample])
    o.eval(sample[:threads], loss, 6, tq, link_flags=flags, threads=threads)          # warm-up
    t0 = time.perf_counter()
    cpu_err = o.eval(sample, loss, 6, tq, link_flags=flags, threads=threads)
    dt = time.perf_counter() - t0
    one = sample[:max(threads * 4, 64)]
    t1 = time.perf_counter()
    o.eval(one, loss, 6, tq, link_flags=flags, threads=1)
    dt1 = time.perf_counter() - t1
    parity = None
    if gpu_err is not None:       # the same rows as the GPU scored them in the timed passes: the checker's other job
        g = np.asarray(gpu_err[:len(sample)], np.float64)
        parity = {"rows": int(len(sample)), "identical_bits": bool(np.array_equal(g.view(np.uint64), cpu_err.view(np.uint64))),
                  "max_abs_diff": float(np.nanmax(np.abs(g - cpu_err))) if len(sample) else 0.0}
    return {"value": len(sample) / dt, "unit": "poses/s", "cores": threads, "kind": "port",
            "single_thread_value": len(one) / dt1, "gpu_errors_vs_port": parity,
            "sample": f"first {len(sample)} of the {len(cand)} grid candidates, {threads} threads over candidates, "
                      f"{dt:.2f} s wall; single thread: first {len(one)} candidates, {dt1:.2f} s"}


def end_to_end(device, n_frames=4096):
    """The whole prediction path at the metric's resolution on one engine context, after the timed region: n_frames synthetic
    640x480 RGB-D frames (already in host memory, as a camera or a dataset reader hands them over) through Predictor.run_many: lockstep
    batches of the default size (582 frames at this resolution), the next batch prepared on worker threads and uploaded on the engine's second stream while one is on the GPU —
    host preparation, upload, the Lookup stage (9^3 grid, the reference's size rule) and every stage of the 'SLU' list, the frames
    walking the stage list in lockstep batches (rope_predict_batch).  Poses = candidate poses rendered AND scored, lookup rows
    included.  Not `value`: an extra figure beside it."""
    from rope_s3d_amd import SyntheticPredictor
    from rope_s3d_amd.constants import DEFAULT_CAMERA_POSE
    sp = SyntheticPredictor(DEFAULT_CAMERA_POSE, '640_480_color', 1, 'SLU', noise=False, seed=1, device=device)
    p = sp.predictor
    lim = sp.urdf_reader.joint_limits
    frames = []
    for f in range(n_frames):
        sp.renderer.setJointAngles(np.random.default_rng(7919 + f).uniform(lim[:, 0], lim[:, 1]) * np.array([1, 1, 1, 0, 0, 0]))
        frames.append(sp.renderer.render())
    colors, depths = [c for c, _ in frames], [d for _, d in frames]
    b = p.default_batch()
    p.run_many(colors[:2 * b], depths[:2 * b])                  # warm-up: two full lockstep batches (both sets of page-locked planes get allocated here)
    p.evaluations = 0
    t0 = time.perf_counter()
    got = p.run_many(colors, depths)
    dt = time.perf_counter() - t0
    truth = np.array([np.random.default_rng(7919 + f).uniform(lim[:, 0], lim[:, 1]) * np.array([1, 1, 1, 0, 0, 0]) for f in range(n_frames)])
    return {"frames_per_s": n_frames / dt, "poses_per_s": p.evaluations / dt, "frames": n_frames,
            "evaluations_per_frame": p.evaluations / n_frames, "lookup_grid": int(len(p.lookup_angles)), "lockstep_batch": b,
            "median_abs_joint_error_rad": float(np.median(np.abs(got - truth)[:, :3])),
            "workload": "640x480 / 1, 'SLU' stage list, one Predictor (one engine context), frames in lockstep batches; "
                        "frames in host memory when the clock starts, angles back in host memory when it stops"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--grid', type=int, default=None, help='S/L/U divisions per joint (default 16 -> 4096 candidates; cfg5: 32)')
    ap.add_argument('--workload', default='cfg1', choices=['cfg1', 'cfg5'],
                    help="cfg1 = BASELINE configs[1] (default, the metric's configuration); cfg5 = configs[4] geometry on one GPU: "
                         "mh50, 1280x720, 32^3 candidates")
    ap.add_argument('--cpu-sample', type=int, default=4096)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-unshared', action='store_true',
                    help="skip the extra passes behind unshared_value (profiling runs: they launch the same kernels on six links per "
                         "candidate and would mix into the per-kernel averages)")
    ap.add_argument('--loss', default='depth', choices=['depth', 'full'],
                    help="depth = the metric's loss (last term of Predictor._error); full = the whole _error with the link masks "
                         "(SURVEY §8d: +2*W*H bytes per candidate for the id image)")
    ap.add_argument('--split-candidates', action='store_true',
                    help="strong scaling (SURVEY §8e, optional): ONE frame, its candidates split over the ranks, all-gather of "
                         "(best error, best index) and a global argmin; default is one frame per rank (weak scaling)")
    ap.add_argument('--no-end-to-end', action='store_true',
                    help="skip the end-to-end figure (N = 1 only, after the timed region): 256 synthetic 640x480 frames through "
                         "Predictor.run_many — preparation, upload, lookup and every stage in lockstep batches")
    ap.add_argument('--backend', default='nccl', help="'nccl' (RCCL over xGMI); 'gloo' only to rehearse N>1 on a one-GPU box")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        raise SystemExit(self_launch(args))

    import torch
    import torch.distributed as dist

    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s) (WORLD_SIZE); refusing to "
                         "print a line whose n_gpus would be wrong")
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU fallback")
    # ROPE_FORCE_DEVICE: rehearsal of the N>1 control flow on a box with one GPU (all ranks share device 0)
    device = int(os.environ.get('ROPE_FORCE_DEVICE', local_rank))
    torch.cuda.set_device(device)
    coll_dev = torch.device('cuda', device) if args.backend == 'nccl' else torch.device('cpu')
    # ROPE_DIST_ALWAYS: also a world of one goes through the process group and its collectives (tests/test_gpu_runtime.py: RCCL
    # itself gets executed on a one-GPU box)
    use_dist = world > 1 or bool(os.environ.get('ROPE_DIST_ALWAYS'))
    if use_dist:
        if args.backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', device))
        else:
            dist.init_process_group(args.backend)

    ranks_seen = dist.get_world_size() if use_dist else 1
    if ranks_seen != args.gpus:
        raise SystemExit(f"bench.py: process group holds {ranks_seen} ranks, --gpus says {args.gpus}")

    from rope_s3d_amd import engine as eng
    from rope_s3d_amd.constants import DEFAULT_CAMERA_POSE, ZFAR, ZNEAR
    from rope_s3d_amd.projection import Intrinsics, camera_matrix
    from rope_s3d_amd.robot import RobotModel

    from rope_s3d_amd.urdf import URDFReader
    if args.workload == 'cfg5':
        robot = RobotModel.from_urdf(URDFReader(os.path.join(ROOT, 'urdfs/motoman_mh50_support/urdf/mh50.urdf')))
        intr, cam_pose = Intrinsics('1280_720_color'), [0, -4.0, 1.5, 0, 0, 0]      # camera pulled back for the larger arm
        b_cand = 12 * 27498 + 12 * 55421 + 2 * 1280 * 720 * 2                     # = 4 681 428 (fp16 depth in the §8d model)
        label = "configs[4] geometry on one GPU: mh50 URDF, 1280x720, %d candidates/frame (%d^3 SLU grid), depth-only loss, 6 links"
        args.grid = args.grid or 32
    else:
        robot = RobotModel.from_urdf()
        intr, cam_pose = Intrinsics('640_480_color'), DEFAULT_CAMERA_POSE
        b_cand = B_CAND
        label = "configs[1]: mh5l_limited URDF, 640x480, %d candidates/frame (%d^3 SLU grid), depth-only loss, 6 links, one frame per rank"
        args.grid = args.grid or 16
    W, H = intr.width, intr.height
    PV = camera_matrix(cam_pose, intr, ZNEAR, ZFAR)

    e = eng.Engine(device)
    e.set_robot(robot)
    e.set_camera(PV, W, H, ZNEAR, ZFAR)

    # synthetic frame of this rank: pose uniform in the S/L/U limits, target = engine render
    rng = np.random.default_rng(7919 + (0 if args.split_candidates else rank))
    q_true = rng.uniform(robot.joint_limits[:, 0], robot.joint_limits[:, 1]) * np.array([1, 1, 1, 0, 0, 0])
    depth, ids = e.render(q_true, 6)
    loss = eng.LOSS_FULL if args.loss == 'full' else eng.LOSS_DEPTH
    bits, flags = None, np.zeros(8, np.uint8)
    if args.loss == 'full':                       # link masks of the synthetic frame, as _loadSynthetic reads them
        bits = np.zeros(ids.shape, np.uint64)
        for l in range(6):
            m = (ids == l) | ((ids == 255) if l == 0 else False)      # base_link's colour 0 equals the background's
            bits |= m.astype(np.uint64) << np.uint64(l)
            flags[l] = 1 | (2 if np.count_nonzero(m & (depth != 0)) > .05 * np.count_nonzero(m) else 0)
        b_cand += 2 * W * H
        label = label.replace('depth-only loss', 'full _error loss (link masks + depth)')
    tq = eng.pack_target(depth.astype(np.float64), bits)
    e.set_target(tq, None, flags)

    cand = slu_grid(robot.joint_limits, args.grid)
    C_total = len(cand)
    first = 0
    if args.split_candidates:
        # contiguous blocks: the grid has joint 0 fastest, so a block keeps whole (q0, q1) groups together for the layers
        per = -(-C_total // world)
        first = rank * per
        cand = cand[first:first + per]
    C = len(cand)
    e.upload_candidates(cand)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        e.eval_resident(6, loss)
    e.sync()

    best = torch.zeros(8 if args.split_candidates else 6, dtype=torch.float64, device=coll_dev)
    if use_dist:                                 # part of the warm-up: the collective's first call sets up its channels
        dist.all_gather([torch.zeros_like(best) for _ in range(world)], best)
    barrier()
    t0 = time.perf_counter()
    # the K timed steps run inside rope_profile_eval, which brackets every kernel with HIP events
    # on the engine's own stream (torch.cuda.Event would only see torch's current stream)
    kern = e.profile_eval(6, loss, None, reps=args.steps)
    _, _, bi, be = e.download(want_err=False)
    if args.split_candidates:
        best.copy_(torch.from_numpy(np.concatenate([cand[bi], [be, first + bi]])))
    else:
        best.copy_(torch.from_numpy(cand[bi]))
    if use_dist:
        gathered = [torch.zeros_like(best) for _ in range(world)]
        dist.all_gather(gathered, best)         # the single collective: final joint angles (+ score and index) over xGMI
        if args.split_candidates:               # global argmin: smallest error, then smallest index; NaN never wins
            g = torch.stack(gathered).cpu().numpy()
            order = np.lexsort((g[:, 7], np.where(np.isnan(g[:, 6]), np.inf, g[:, 6])))
            be, bi = float(g[order[0], 6]), int(g[order[0], 7])
    barrier()
    dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # outside the timed region: the same candidates with nothing shared between them (six links drawn per candidate) —
    # the rate a caller sees when no two candidates agree in their first two joint angles
    kern_u = None
    if not args.no_unshared:
        e.set_strategy(e.NO_LAYERS)
        e.eval_resident(6, loss)
        kern_u = e.profile_eval(6, loss, None, reps=max(2, min(args.steps, 5)))
        e.set_strategy(0)

    if rank == 0:
        poses = (C_total if args.split_candidates else world * C) * args.steps
        raster_s = kern['raster'] * 1e-3
        achieved = b_cand * C / raster_s / 1e9
        default_wl = args.loss == 'depth' and args.workload == 'cfg1' and args.grid == 16
        # Counters come from committed rocprofv3 passes, the times from this run: only counters taken on THIS build may be combined
        # with them (a kernel edit changes the instruction count).  Stale files give nulls and "pmc_stale": true.
        lib_id = eng.build_id()
        traffic_d, stale_t = committed_profile('traffic.json', lib_id) if default_wl else (None, False)
        pmc_d, stale_p = committed_profile(PROFILE_ROUND + '_pmc.json', lib_id) if default_wl else (None, False)
        clock_d, stale_c = committed_profile(PROFILE_ROUND + '_kernel_clock.json', lib_id) if default_wl else (None, False)
        traffic = traffic_d['hbm_bytes_per_launch'] if traffic_d else None
        pmc = {k: v['avg_per_launch'] for k, v in pmc_d['counters'].items()} if pmc_d else {}
        peak = valu_issue_peak()
        n_simd = 256 * 4
        valu = None
        if pmc.get('SQ_INSTS_VALU') and peak:
            rate = pmc['SQ_INSTS_VALU'] / (kern['score'] * 1e-3)            # wave64 VALU instructions per second, whole chip
            # the clock this kernel runs at: stamped inside it (profiling build, tools/kernel_clock.py) and, as a cross-check,
            # GRBM_GUI_ACTIVE / 8 XCDs / the profiled launch's duration (MI355X_MICROARCH.md, "DVFS give-back")
            clk = clock_d['layers']['clock_ghz_median'] if clock_d else None
            clk_grbm = pmc_d.get('clock_ghz_grbm')
            use_clk = clk or clk_grbm
            cyc = (n_simd * use_clk * 1e9 / rate) if use_clk else None      # SIMD cycles per VALU instruction issued, as the kernel runs
            valu = {"bound": "valu_issue", "achieved": rate / 1e9, "unit": "G wave-instructions/s",
                    "clock_ghz_kernel": clk, "clock_ghz_kernel_grbm": clk_grbm, "clock_ghz_microbench": peak['clock_ghz_microbench'],
                    "simd_cycles_per_valu_instruction": cyc,
                    "microbench_cycles_per_instruction": {k: peak[k] for k in ('full_rate', 'mix', 'half_rate')},
                    "peak": (n_simd * use_clk / peak['mix']) if use_clk else None,
                    "frac": (peak['mix'] / cyc) if cyc else None,
                    "frac_of_full_rate_peak": (peak['full_rate'] / cyc) if cyc else None,
                    "valu_instructions_per_launch": pmc['SQ_INSTS_VALU'], "active_lanes_of_64": pmc.get('active_lanes'),
                    "wave_wait_fraction": (pmc['SQ_WAIT_ANY'] / pmc['SQ_WAVE_CYCLES']) if pmc.get('SQ_WAVE_CYCLES') else None,
                    "source": f"SQ_INSTS_VALU of the scoring launch from profiles/{PROFILE_ROUND}_pmc.json (rocprofv3 --pmc, same command, same build: "
                              f"build_id {lib_id}) / its live launch time; peak = 1024 SIMDs x the kernel's own clock / the SIMD cycles a wave64 instruction of a "
                              f"kernel-like mix costs at 6 waves per SIMD (tools/valu_issue_bench.hip, profiles/{PROFILE_ROUND}_valu_issue.json: wall time x in-loop clock)"}
        out = {
            "metric": "rendered+scored candidate poses/sec @%dx%d" % (W, H),
            "value": poses / dt, "unit": "poses/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong" if args.split_candidates else "weak",
            "vs_baseline": None, "dtype": "i32/i64 fixed-point edge functions, f32 depth, u64 Q32 sums",
            "data": "synthetic",
            "config": {"workload": label % (C_total, args.grid),
                       "candidates_per_step": C, "frames_per_rank": 1,
                       "parallelism": f"candidates of one frame /{world}" if args.split_candidates else f"frames x{world}",
                       "argmin_error": be, "argmin_index": bi},
            "ranks_seen": ranks_seen, "collective": (args.backend if use_dist else None),
            "unshared_value": C / (kern_u['total'] * 1e-3) if kern_u else None,
            "unshared_note": "poses/s per GPU with rope_set_strategy(NO_LAYERS): links 0-2 drawn for every candidate instead of once "
                             "per distinct (S, L); same results bit for bit; measured after the timed region",
            # SURVEY §8d's contract: algorithmic bytes per candidate x candidates per launch / live launch time against the HBM
            # peak.  The kernel keeps a candidate's depth image in LDS, so the bytes that really cross the HBM pins (`traffic`,
            # PMC) are a small fraction of the algorithmic ones and HBM is not what binds it: see measured_hbm_* and valu_issue.
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "pmc_stale": bool(stale_t or stale_p or stale_c),
                         "build_id": lib_id,
                         "measured_hbm_GBs": (traffic / (kern['score'] * 1e-3) / 1e9) if traffic else None,
                         "measured_hbm_frac": (traffic / (kern['score'] * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
                         "binding_resource": "vector instruction issue (valu_issue below), not HBM",
                         "valu_issue": valu,
                         "kernel": "raster_queue_kernel<%s,SCORE> (per-candidate links + loss, from the queue of (candidate, tile) pairs, heaviest first) + the shared-layer launches raster_score_kernel / raster_queue_kernel<%s,LAYER> (links 0-2 once per distinct (S,L))" % ((args.loss.upper(),) * 2),
                         "kernel_ms": kern['raster'], "score_launch_ms": kern['score'], "layer_launch_ms": kern['layer'],
                         "bytes_per_candidate": b_cand, "candidates_per_launch": C,
                         "other_kernels_ms": {"fk_mvp+bounds": kern['fk'], "finalize+argmin": kern['finalize'],
                                              "pass_total": kern['total']}},
        }
        if not args.no_end_to_end and world == 1 and args.workload == 'cfg1':
            out["end_to_end"] = end_to_end(device)
        if not args.no_cpu_baseline and world == 1:           # reported at N=1 only
            gpu_err = e.download(want_err=True)[0]                 # errors of the last timed pass, outside the timed region
            out["cpu_baseline"] = cpu_baseline(robot, PV, W, H, ZNEAR, ZFAR, cand, tq, min(args.cpu_sample, C), gpu_err, loss, flags)
        print(json.dumps(out))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
