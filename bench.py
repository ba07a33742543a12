#!/usr/bin/env python3
"""bench.py -- frames/sec @1080p with 4k PyrLK tracks (BASELINE.json metric), N GPUs of one node.

A step = one frame on each of S independent camera streams of this GPU (colour convert ->
5-level Gaussian pyramid -> PyrLK 21x21 on <=4096 points -> Harris track/merge), inputs resident
in HBM (a ring of pre-generated RGBX frames).  Streams are sharded one-or-more per GPU; there is
no data-path collective -- the process group only carries the timing
barrier and the max-over-ranks reduction.

  python bench.py [--gpus N --steps K --warmup W --streams S]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# SURVEY.md section 8d, configs[1]: algorithmic (compulsory, unique) bytes per frame
WIDTH, HEIGHT, LEVELS, CAPACITY, CELL, WIN, ITERS = 1920, 1080, 5, 4096, 18, 21, 5
BYTES_PYRAMID = 2754000 + 688560
BYTES_DETECT = 2073600 + 32768
BYTES_LK = 5524320 + 65536          # both pyramids read once + 16 B per point: the PyrLK launch
BYTES_CVT = 10368000
BYTES_FRAME = BYTES_PYRAMID + BYTES_DETECT + BYTES_LK + BYTES_CVT
HBM_PEAK_GBS = 8000.0               # MI355X_MICROARCH.md: 8 TB/s spec
WORKLOAD = "1080p synthetic RGBX frames, 5-level pyramid, 4k Harris features (cell 18), 21x21 PyrLK window, 5 iters"


def shard_streams(total_streams, world_size, rank):
    """camera stream s runs on rank s % world_size (independent units, no exchange step)"""
    return [s for s in range(total_streams) if s % world_size == rank]


def reduce_max(value, world_size):
    if world_size == 1:
        return value
    import torch
    import torch.distributed as dist
    t = torch.tensor([value], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier(world_size):
    if world_size > 1:
        import torch.distributed as dist
        dist.barrier()


def cpu_baseline(frames_rgbx, params, budget_s=12.0, max_frames=48):
    """The oracle (a port: the reference implementation is closed-source) timed on this host's
    cores, OpenMP over points/rows, on a bounded sample of the same workload."""
    from oracle.pipeline import OracleTracker
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or len(os.sched_getaffinity(0))
    os.environ["OMP_NUM_THREADS"] = str(threads)
    trk = OracleTracker(params, omp=True)
    trk.init(frames_rgbx[0])
    n, t0 = 0, time.perf_counter()
    while n < max_frames:
        trk.track(frames_rgbx[(n + 1) % len(frames_rgbx)])
        n += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": f"{n} frames of the same 1080p workload through oracle/ (OpenMP, {threads} threads)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--streams", type=int, default=32, help="camera streams per GPU")
    ap.add_argument("--ring", type=int, default=8, help="pre-generated frames resident in HBM")
    ap.add_argument("--api", choices=["ovx", "thin"], default="ovx",
                    help="ovx: OpenVX graph API (default, the north-star path); thin: nvxcu primitives")
    ap.add_argument("--host-threads", type=int, default=1,
                    help="ovx: 1 = one host thread per camera stream calling vxProcessGraph (default); 0 = one thread scheduling all graphs")
    ap.add_argument("--driver", choices=["native", "python"], default="native",
                    help="ovx with host threads: native = the per-frame C API calls are made by C++ threads "
                         "(visualslam_amd/csrc/stream_driver.cpp), as the reference's C++ application makes them; "
                         "python = by Python threads through ctypes (they take turns on the interpreter lock)")
    ap.add_argument("--homography", action="store_true",
                    help="BASELINE configs[4]: append nvxFindHomographyNode (batched RANSAC on the PyrLK matches, the video "
                         "stabilizer's arguments) to every stream's graph; a separate workload, never the headline line")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import numpy as np
    import torch

    from visualslam_amd import nvxcu, synth
    from visualslam_amd.thin_tracker import Params

    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU fallback)"
    # one GPU per rank; on a box with fewer GPUs than ranks (a rehearsal) ranks share GPUs round-robin
    local_device = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_device)
    os.environ["VISUALSLAM_AMD_DEVICE"] = str(local_device)     # vxCreateContext binds this rank's GPU explicitly
    device = f"cuda:{local_device}"
    nvxcu.lib()

    params = Params(pyr_levels=LEVELS, lk_num_iters=ITERS, lk_win_size=WIN, array_capacity=CAPACITY,
                    detector_cell_size=CELL, use_harris_detector=True, harris_k=0.04, harris_thresh=100.0)

    # weak scaling: S streams on every GPU; global stream ids sharded round-robin over ranks
    my_streams = shard_streams(args.streams * world, world, rank)
    assert len(my_streams) == args.streams

    base = synth.base_frame(WIDTH, HEIGHT, 1234)
    host_frames = [synth.to_rgbx(synth.gray_frame(base, t, 1234)) for t in range(args.ring)]
    phases = [(s * 3) % args.ring for s in my_streams]     # streams look at different frames

    if args.homography and args.api != "ovx":
        raise SystemExit("--homography needs --api ovx (nvxFindHomographyNode is a graph node)")
    if args.api == "ovx":
        # the north-star path: one OpenVX graph per camera stream (ColorConvert -> GaussianPyramid ->
        # OpticalFlowPyrLK -> HarrisTrack), per frame vxSetParameterByIndex x2 + vxAgeDelay x2 +
        # vxScheduleGraph, as demos/feature_tracker/feature_tracker.cpp:216-224 drives it
        import ctypes as C
        from visualslam_amd import ovx
        from visualslam_amd.vx_tracker import VxFeatureTracker
        L = ovx.lib()
        ctx = L.vxCreateContext()
        ovx.check(L.vxGetStatus(ctx), "vxCreateContext")
        ring = []
        for f in host_frames:
            img = L.vxCreateImage(ctx, WIDTH, HEIGHT, ovx.VX_DF_IMAGE_RGBX)
            ovx.write_image(img, f)
            ring.append(img)
        trackers = [VxFeatureTracker(ctx, params, WIDTH, HEIGHT, ring[ph], None, homography=args.homography) for ph in phases]
        if args.homography:
            ovx.check(L.vxDirective(trackers[0].hg, ovx.VX_DIRECTIVE_ENABLE_PERFORMANCE))
        # HIP events around the PyrLK node of stream 0 only (two event records per frame on that graph's stream)
        ovx.check(L.vxDirective(trackers[0].lk, ovx.VX_DIRECTIVE_ENABLE_PERFORMANCE))

        def step(k):
            for trk, ph in zip(trackers, phases):
                trk.track(ring[(ph + 1 + k) % args.ring], None, schedule_only=True)
            for trk in trackers:
                trk.wait()

        def run_stream(i, k0, n):
            # one host thread per camera stream, synchronous vxProcessGraph per frame exactly as the
            # reference application calls it; streams overlap on the GPU because each graph owns a HIP stream
            trk, ph = trackers[i], phases[i]
            for k in range(k0, k0 + n):
                trk.track(ring[(ph + 1 + k) % args.ring], None)

        native = None
        if args.host_threads and args.driver == "native":
            drv = C.CDLL(os.path.join(os.path.dirname(nvxcu.LIB_PATH), "libvsa_stream_driver.so"))
            n = len(trackers)
            arr = lambda xs: (C.c_void_p * len(xs))(*xs)    # noqa: E731
            drv.vsa_drive_streams.restype = C.c_int
            drv.vsa_drive_streams.argtypes = [C.c_int] + [C.c_void_p] * 6 + [C.c_int, C.POINTER(C.c_int), C.c_int, C.c_int]
            handles = (arr([t.g for t in trackers]), arr([t.cvt for t in trackers]), arr([t.det for t in trackers]),
                       arr([t.pyr_delay for t in trackers]), arr([t.pts_delay for t in trackers]), arr(ring))

            def native(k0, n_frames):
                ovx.check(drv.vsa_drive_streams(n, *handles, len(ring), (C.c_int * n)(*phases), k0, n_frames), "vsa_drive_streams")

        def lk_launch_ms():
            perf = ovx.query("vxQueryNode", trackers[0].lk, ovx.VX_NODE_PERFORMANCE, ovx.vx_perf_t)
            return perf.avg * 1e-6, int(perf.num)

        def counts():
            cur = [trk.curr() for trk in trackers]
            return [int((c[:, 0] >= 0).sum()) for c in cur], [len(c) for c in cur]
        api_desc = ("OpenVX graph C API (VX/vx.h): one graph + HIP stream per camera stream, " +
                    (("one native host thread per stream calling vxProcessGraph" if native else "one Python thread per stream calling vxProcessGraph")
                     if args.host_threads else "one host thread, vxScheduleGraph/vxWaitGraph"))
    else:
        from visualslam_amd.devmem import DeviceImage
        from visualslam_amd.thin_tracker import ThinFeatureTracker
        ring = [DeviceImage(WIDTH, HEIGHT, nvxcu.NVXCU_DF_IMAGE_RGBX, device).upload(f) for f in host_frames]
        trackers = [ThinFeatureTracker(params, WIDTH, HEIGHT, device, torch.cuda.Stream(device)) for _ in my_streams]
        for trk, ph in zip(trackers, phases):
            trk.init(ring[ph].desc)

        def step(k):
            for trk, ph in zip(trackers, phases):
                trk.track(ring[(ph + 1 + k) % args.ring].desc)

        def lk_launch_ms():
            pairs = trackers[0].lk_events or []
            return (float(np.mean([a.elapsed_time(b) for a, b in pairs])) if pairs else float("nan")), len(pairs)

        def counts():
            cur = [trk.curr_features() for trk in trackers]
            return [int((c[:, 0] >= 0).sum()) for c in cur], [len(c) for c in cur]
        api_desc = "thin C ABI (NVX/nvxcu.h), eager launches, one HIP stream per camera stream"
    torch.cuda.synchronize()

    def run(k0, n):
        """n steps (= n frames on every camera stream of this GPU) starting at frame index k0"""
        if args.api == "ovx" and native:
            native(k0, n)
        elif args.api == "ovx" and args.host_threads:
            import threading
            ths = [threading.Thread(target=run_stream, args=(i, k0, n)) for i in range(len(trackers))]
            for t in ths:
                t.start()
            for t in ths:
                t.join()
        else:
            for k in range(k0, k0 + n):
                step(k)

    run(0, args.warmup)
    torch.cuda.synchronize()
    barrier(world)
    if args.api == "thin":
        trackers[0].lk_events = []
    lk_before = lk_launch_ms() if args.api == "ovx" else (0.0, 0)
    t0 = time.perf_counter()
    run(args.warmup, args.steps)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    barrier(world)
    elapsed = reduce_max(elapsed, world)

    # PyrLK launch duration from HIP events recorded on its stream during the timed region
    lk_ms, lk_n = lk_launch_ms()
    if args.api == "ovx" and lk_n > lk_before[1]:
        lk_ms = (lk_ms * lk_n - lk_before[0] * lk_before[1]) / (lk_n - lk_before[1])
    n_tracks, n_points = counts()

    if rank == 0:
        frames = args.steps * args.streams * world
        fps = frames / elapsed
        lk_gbs = BYTES_LK / (max(lk_ms, 1e-9) * 1e-3) / 1e9
        traffic = None
        tf = os.path.join(ROOT, "profiles", "lk_hbm_traffic.json")
        if os.path.exists(tf):
            with open(tf) as f:
                traffic = json.load(f).get("bytes_per_launch")
        out = {
            "metric": "frames/sec @1080p with 4k PyrLK tracks",
            "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": WORKLOAD + (" + nvxFindHomographyNode (RANSAC 3.0 px, 56 hypotheses, <=10 refits) on the matches"
                                               if args.homography else ""), "streams_per_gpu": args.streams, "frames_per_step": args.streams * world,
                       "frame_ring": args.ring, "api": api_desc},
            "tracks_per_sec": fps * float(np.mean(n_tracks)),
            "points_per_frame": float(np.mean(n_points)), "tracked_per_frame": float(np.mean(n_tracks)),
            "frame_algorithmic_GBps": BYTES_FRAME * fps / 1e9,
            "frame_algorithmic_frac_of_hbm_peak": BYTES_FRAME * fps / 1e9 / (HBM_PEAK_GBS * world),
            "roofline": {"kernel": "k_pyrlk<7,21,true> (PyrLK, 4096 points x 5 levels, 21x21)", "bound": "hbm",
                         "achieved": lk_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": lk_gbs / HBM_PEAK_GBS,
                         "traffic": traffic, "launch_ms": lk_ms, "algorithmic_bytes_per_launch": BYTES_LK},
        }
        vf = os.path.join(ROOT, "profiles", "frame_valu_instructions.json")
        if os.path.exists(vf) and not args.homography:
            # what actually bounds the path: vector-instruction issue.  Wave-instructions per frame
            # from the SQ counters, against the f32 FMA issue rate measured on this GPU -- an extra, not the roofline object
            with open(vf) as f:
                v = json.load(f)
            rate = v["wave_instructions_per_frame"] * fps / 1e9 / world
            out["valu_issue"] = {"wave_instructions_per_frame": v["wave_instructions_per_frame"], "achieved": rate,
                                 "peak": v["peak_wave_instructions_per_ns"], "unit": "G wave-instructions/s per GPU",
                                 "frac": rate / v["peak_wave_instructions_per_ns"]}
        if args.homography:
            perf = ovx.query("vxQueryNode", trackers[0].hg, ovx.VX_NODE_PERFORMANCE, ovx.vx_perf_t)
            H, mask = trackers[0].homography()
            out["homography"] = {"node_ms": perf.avg * 1e-6, "inliers_stream0": int((mask != 0).sum()), "points_stream0": int(len(mask)),
                                 "matrix_stream0": [float(x) for x in H.T.reshape(-1)]}
        if not args.no_cpu_baseline and world == 1 and not args.homography:
            out["cpu_baseline"] = cpu_baseline(host_frames, params)
        print(json.dumps(out), flush=True)
    barrier(world)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
