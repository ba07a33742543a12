#!/usr/bin/env python3
"""bench.py -- frames/sec @1080p with 4k PyrLK tracks (BASELINE.json metric), N GPUs of one node.

A step = one frame on each of S independent camera streams of this GPU (colour convert -> Gaussian pyramid ->
PyrLK -> Harris/FAST track and merge), inputs resident in HBM (a ring of pre-generated RGBX frames).  Streams are
sharded one-or-more per GPU; there is no data-path collective -- the process group only
carries the timing barrier and the max-over-ranks reduction.

  python bench.py [--gpus N --steps K --warmup W --streams S --config c1|c2|c3 --homography]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

--gpus N without a launcher (WORLD_SIZE unset) starts the N ranks itself, as fresh child processes, before this
process has touched a GPU.  The default line is BASELINE configs[1] (--config c2); c1 and c3 are configs[0] and [2],
--homography is configs[4]: other workloads, never the headline line.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0               # MI355X_MICROARCH.md: 8 TB/s spec

# the workloads and their ALGORITHMIC (compulsory, unique) bytes per frame
CONFIGS = {
    "c1": dict(baseline_config=0, width=640, height=480, levels=6, capacity=2000, cell=18, win=6, iters=5,
               harris=True, harris_thresh=2000.0, ring=64, streams=32,
               frames="640x480 RGBX frames synthesised from warps of the reference's lena.jpg / baboon.jpg "
                      "(tests/golden/c1_sources.npz; the bundled video named by configs[0] is not in the repository)",
               workload="640x480 frames, 6-level pyramid, <=2000 Harris features (cell 18, k 0.04, threshold 2000), 6x6 PyrLK window, "
                        "5 iters (data/feature_tracker_demo_config.ini)",
               bytes=dict(pyramid=409200 + 102300, detect=307200 + 16000, lk=819000 + 32000, cvt=1536000),
               ),
    "c2": dict(baseline_config=1, width=1920, height=1080, levels=5, capacity=4096, cell=18, win=21, iters=5,
               harris=True, harris_thresh=100.0, ring=64, streams=32,
               frames="1080p synthetic RGBX frames (seed 1234: low-pass noise + blobs, known sub-pixel drift, per-frame noise)",
               workload="1080p synthetic RGBX frames, 5-level pyramid, 4k Harris features (cell 18), 21x21 PyrLK window, 5 iters",
               bytes=dict(pyramid=2754000 + 688560, detect=2073600 + 32768, lk=5524320 + 65536, cvt=10368000),
               ),
    "c3": dict(baseline_config=2, width=3840, height=2160, levels=7, capacity=10000, cell=18, win=21, iters=5,
               harris=False, harris_thresh=100.0, ring=16, streams=16,
               frames="4K synthetic RGBX frames (seed 1234: low-pass noise + blobs, known sub-pixel drift, per-frame noise)",
               workload="4K synthetic RGBX frames, 7-level pyramid, FAST-9 (threshold 25) track with 10k features (cell 18), 21x21 PyrLK window, 5 iters",
               bytes=dict(pyramid=11056560 + 2764200, detect=8294400 + 80000, lk=22117200 + 160000, cvt=41472000),
               ),
}


def lk_kernel_names(win, det=0):
    """(PyrLK alone, PyrLK + detector cell pass): the instantiations pyrlk.hip's launch_lk / launch_bucket pick for this window on
    4-byte-aligned pyramids -- the same bucket and lane-map arithmetic, so the bench line names the kernel that ran.
    det: 0 = Harris cell pass, 9 .. 12 = FAST-N (the last template argument of the fused kernel)"""
    segs = max(1, min(64 // win, win))
    seglen = (win + segs - 1) // segs
    segs = (win + seglen - 1) // seglen
    nseg, wmax = next(b for b in ((1, 8), (2, 10), (4, 16), (7, 21), (16, 32)) if win <= b[1])
    full = "true" if seglen == nseg and segs * seglen == win else "false"
    cap = "_104" if nseg == 7 else ""
    return f"k_pyrlk{cap}<{nseg},{wmax},true,{full}>", f"k_pyrlk_cells{cap}<{nseg},{wmax},{full},{det}>"


def metric_name(config, homography):
    """BASELINE.json's metric is quoted on configs[1]; the other configurations are labelled as what they are"""
    if homography:
        return "frames/sec @1080p with 4k PyrLK tracks + RANSAC homography (BASELINE configs[4], not the headline)"
    return {"c1": "frames/sec @640x480 with 2k PyrLK tracks (BASELINE configs[0], not the headline)",
            "c2": "frames/sec @1080p with 4k PyrLK tracks",
            "c3": "frames/sec @4K with 10k PyrLK tracks, FAST track (BASELINE configs[2], not the headline)"}[config]


def shard_streams(total_streams, world_size, rank):
    """camera stream s runs on rank s % world_size (independent units, no exchange step)"""
    return [s for s in range(total_streams) if s % world_size == rank]


def rank_device(local_rank, device_count):
    """one GPU per rank; with fewer GPUs than ranks (a rehearsal on a 1-GPU box) ranks share GPUs round-robin"""
    return local_rank % max(device_count, 1)


def reduce_max(value, world_size):
    if world_size == 1:
        return value
    import torch
    import torch.distributed as dist
    t = torch.tensor([value], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_floats(value, world_size):
    if world_size == 1:
        return [value]
    import torch
    import torch.distributed as dist
    out = [torch.zeros(1, dtype=torch.float64) for _ in range(world_size)]
    dist.all_gather(out, torch.tensor([value], dtype=torch.float64))
    return [float(t.item()) for t in out]


def barrier(world_size):
    if world_size > 1:
        import torch.distributed as dist
        dist.barrier()


def make_frames(cfg, n):
    """the host copies of the frame ring (also what the CPU baseline runs on)"""
    import numpy as np

    from visualslam_amd import synth
    if cfg["baseline_config"] == 0:
        src = np.load(os.path.join(ROOT, "tests", "golden", "c1_sources.npz"))
        return synth.c1_sequence(src, n, cfg["width"], cfg["height"])
    base = synth.base_frame(cfg["width"], cfg["height"], 1234)
    return [synth.to_rgbx(synth.gray_frame(base, t, 1234)) for t in range(n)]


def cpu_baseline(frames_rgbx, params, budget_s=10.0):
    """The oracle (a port: the reference implementation is closed-source) timed on this host's cores on a bounded
    sample of the same workload: the OpenMP build on all host threads, median of 200 frames
    after 20 warm-up frames, and the scalar build on one thread, median of as many frames as fit in the budget
    (after 2 warm-ups).  `value` is the OpenMP figure; a reported baseline, never credit."""
    import numpy as np

    from oracle.pipeline import OracleTracker
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or len(os.sched_getaffinity(0))
    os.environ["OMP_NUM_THREADS"] = str(threads)

    def run(omp, warm, frames, budget):
        trk = OracleTracker(params, omp=omp)
        trk.init(frames_rgbx[0])
        times, t_end = [], time.perf_counter() + budget
        for k in range(warm + frames):
            t0 = time.perf_counter()
            trk.track(frames_rgbx[(k + 1) % len(frames_rgbx)])
            if k >= warm:
                times.append(time.perf_counter() - t0)
            if time.perf_counter() > t_end and len(times) >= 5:
                break
        return 1.0 / float(np.median(times)), len(times)

    omp_fps, omp_n = run(True, 20, 200, 3.0 * budget_s)
    one_fps, one_n = run(False, 2, 200, budget_s)
    return {"value": omp_fps, "unit": "frames/s", "cores": threads, "kind": "port",
            "single_thread": {"value": one_fps, "cores": 1, "frames": one_n},
            "sample": f"oracle/ on the same workload and frames: OpenMP build, {threads} threads, median of {omp_n} frames after 20 warm-ups; "
                      f"scalar build, 1 thread, median of {one_n} frames after 2 warm-ups (time-bounded)"}


def ratio(a, b):
    return (a / b) if (a is not None and b) else None


def load_profile_json(name):
    path = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(path):
        return None
    with open(path) as f:
        return json.load(f)


def profiled_launch_ms(kernel):
    """average duration of `kernel` in the committed rocprofv3 --kernel-trace --stats summary of a single-stream run of this bench
    (profiles/r03_single_stream_kernel_stats.csv): the kernel's own begin-to-end time, without the few microseconds of dispatch
    latency that the HIP events around a launch include"""
    import csv
    path = os.path.join(ROOT, "profiles", "r03_single_stream_kernel_stats.csv")
    if not os.path.exists(path):
        return None
    want = kernel.replace(" ", "")
    for r in csv.DictReader(open(path)):
        if want in r["Name"].replace(" ", ""):
            return float(r["AverageNs"]) * 1e-6
    return None


def host_cores():
    """The CPU share this job may use: the scheduler affinity mask, cut down by a cgroup CPU quota when there is one.  (On the GPU
    boxes os.cpu_count() shows the whole machine; host threads beyond the share only take turns on its cores.)"""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return n


def default_streams(cfg, world, cores):
    """Camera streams per GPU: a host core for every stream's thread, at most the configuration's own count (32; 16 for 4K), at least a
    hardware queue each (4).  Every stream has a host thread that polls for its frame's end (hipStreamSynchronize spins), and 8 streams
    already saturate a GPU (profiles/r03_stream_sweep.txt: 4 -> 39.1 k, 8 -> 39.7 k, 16 -> 39.4 k, 32 -> 37.6 k frames/s), so more threads
    than cores buy nothing and cost a short run its steadiness: on the 16-core GPU box one driver-sized block in three reads 33-35 k
    with 32 streams where 16 streams read 38.4-38.7 k every time (profiles/r03_streams_short_blocks.txt).  With several ranks on one
    host the ranks share its cores."""
    return min(cfg["streams"], max(4, cores // max(world, 1)))


def launch_ranks(args, argv, poll_s=0.05):
    """--gpus N with no launcher: start the N ranks as fresh child processes (this parent never touches a GPU, and
    nothing that has initialised a GPU is ever re-executed).  Rank 0's stdout carries the JSON line.  The first rank that
    ends with an error ends the job: its siblings (which would otherwise wait in the process group for it until the
    group's timeout) are terminated, and the launcher returns that rank's exit code."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    children = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), LOCAL_WORLD_SIZE=str(args.gpus))
        children.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
    deadline = time.monotonic() + args.launch_timeout
    failed = 0
    try:
        while True:
            codes = [c.poll() for c in children]
            bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
            if bad:
                r, failed = bad[0]
                print(f"bench.py: rank {r} ended with exit code {failed}; ending the other ranks", file=sys.stderr, flush=True)
                break
            if all(c == 0 for c in codes):
                return 0
            if time.monotonic() > deadline:
                failed = 124
                print(f"bench.py: ranks still running after {args.launch_timeout:.0f} s; ending them", file=sys.stderr, flush=True)
                break
            time.sleep(poll_s)
    finally:
        live = [c for c in children if c.poll() is None]
        for c in live:
            c.terminate()
        t_kill = time.monotonic() + 5.0
        for c in live:
            try:
                c.wait(timeout=max(0.0, t_kill - time.monotonic()))
            except subprocess.TimeoutExpired:
                c.kill()
                c.wait()
    return abs(failed) or 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c2",
                    help="c2 = BASELINE configs[1], the headline (default); c1 = configs[0] (640x480, demo ini); c3 = configs[2] (4K, FAST, 10k features)")
    ap.add_argument("--streams", type=int, default=0,
                    help="camera streams per GPU (default: a host core per stream, min(32 (16 for c3), max(4, host cores // N)))")
    ap.add_argument("--ring", type=int, default=0, help="pre-generated frames resident in HBM (default: 64, SURVEY.md section 8d; 16 for c3)")
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
    ap.add_argument("--repeats", type=int, default=-1,
                    help="after the timed block, repeat it this many times for repeat_frames_per_s (default: 15, fewer when a block is long)")
    ap.add_argument("--preheat-s", type=float, default=0.5,
                    help="untimed run of the same workload before the warm-up steps, this many seconds: brings the GPU from its idle clock "
                         "to its operating clocks (5 warm-up steps are 1.3 ms; the ramp takes ~50 ms). 0 = none")
    ap.add_argument("--launch-timeout", type=float, default=900.0, help="--gpus N without a launcher: end the ranks after this many seconds")
    ap.add_argument("--pg-timeout", type=float, default=120.0, help="process-group timeout in seconds (a rank that never arrives is an error)")
    ap.add_argument("--dry-run", action="store_true",
                    help="print this rank's device and camera streams as one JSON line and exit (no GPU is touched)")
    ap.add_argument("--dry-run-devices", type=int, default=0, help="--dry-run: pretend the node has this many GPUs (default: --gpus)")
    args = ap.parse_args()
    cfg = CONFIGS[args.config]
    args.ring = args.ring or cfg["ring"]

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args, sys.argv[1:]))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    cores = host_cores()
    args.streams = args.streams or default_streams(cfg, world, cores)
    if os.environ.get("VSA_BENCH_TEST_FAIL_RANK") == str(rank):     # tests/test_bench_sharding.py: a rank that dies at start-up
        sys.exit(3)

    # weak scaling: S streams on every GPU; global stream ids sharded round-robin over ranks
    my_streams = shard_streams(args.streams * world, world, rank)
    assert len(my_streams) == args.streams
    phases = [(s * 3) % args.ring for s in my_streams]     # streams look at different frames

    if args.dry_run:
        dev = rank_device(local_rank, args.dry_run_devices or args.gpus)
        print(json.dumps({"rank": rank, "world": world, "device": dev, "streams": my_streams, "phases": phases,
                          "host_cores": cores, "host_threads_total": len(my_streams) * world}), flush=True)
        return

    import numpy as np
    import torch

    from visualslam_amd import nvxcu
    from visualslam_amd.thin_tracker import Params

    json_fd = None
    if world > 1:
        # stdout carries the ONE JSON line: what libraries print there from native code (Gloo's connection banner) goes to stderr
        sys.stdout.flush()
        json_fd = os.dup(1)
        os.dup2(2, 1)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import datetime
        dist.init_process_group(backend="gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=args.pg_timeout))
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU fallback)"
    local_device = rank_device(local_rank, torch.cuda.device_count())
    torch.cuda.set_device(local_device)
    os.environ["VISUALSLAM_AMD_DEVICE"] = str(local_device)     # vxCreateContext binds this rank's GPU explicitly
    # the camera streams' graphs get HIP streams of greatest priority: a pool of hardware queues of their own, on which HIP deals
    # them out evenly; nothing else runs on this GPU during the timed region
    os.environ.setdefault("VISUALSLAM_AMD_STREAM_PRIORITY", "-1")
    device = f"cuda:{local_device}"
    nvxcu.lib()

    W, H = cfg["width"], cfg["height"]
    params = Params(pyr_levels=cfg["levels"], lk_num_iters=cfg["iters"], lk_win_size=cfg["win"], array_capacity=cfg["capacity"],
                    detector_cell_size=cfg["cell"], use_harris_detector=cfg["harris"], harris_k=0.04, harris_thresh=cfg["harris_thresh"],
                    fast_type=9, fast_thresh=25)
    host_frames = make_frames(cfg, args.ring)

    if args.homography and args.api != "ovx":
        raise SystemExit("--homography needs --api ovx (nvxFindHomographyNode is a graph node)")
    driver = None
    if args.api == "ovx":
        # the north-star path: one OpenVX graph per camera stream (ColorConvert -> GaussianPyramid ->
        # OpticalFlowPyrLK -> HarrisTrack), per frame vxSetParameterByIndex x2 + vxAgeDelay x2 +
        # vxProcessGraph, as demos/feature_tracker/feature_tracker.cpp:216-224 drives it
        import ctypes as C
        from visualslam_amd import ovx
        from visualslam_amd.vx_tracker import VxFeatureTracker
        L = ovx.lib()
        ctx = L.vxCreateContext()
        ovx.check(L.vxGetStatus(ctx), "vxCreateContext")
        ring = []
        for f in host_frames:
            img = L.vxCreateImage(ctx, W, H, ovx.VX_DF_IMAGE_RGBX)
            ovx.write_image(img, f)
            ring.append(img)
        trackers = [VxFeatureTracker(ctx, params, W, H, ring[ph], None, homography=args.homography) for ph in phases]

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

        if args.host_threads and args.driver == "native":
            # persistent native host threads, released together for each run: a timed run holds the per-frame API calls only
            drv = C.CDLL(os.path.join(os.path.dirname(nvxcu.LIB_PATH), "libvsa_stream_driver.so"))
            arr = lambda xs: (C.c_void_p * len(xs))(*xs)    # noqa: E731
            drv.vsa_driver_create.restype = C.c_void_p
            drv.vsa_driver_create.argtypes = [C.c_int] + [C.c_void_p] * 6 + [C.c_int, C.POINTER(C.c_int)]
            drv.vsa_driver_run.restype = C.c_int
            drv.vsa_driver_run.argtypes = [C.c_void_p, C.c_int, C.c_int]
            drv.vsa_driver_destroy.argtypes = [C.c_void_p]

            def make_driver(trks, phs):
                n = len(trks)
                h = drv.vsa_driver_create(n, arr([t.g for t in trks]), arr([t.cvt for t in trks]), arr([t.det for t in trks]),
                                          arr([t.pyr_delay for t in trks]), arr([t.pts_delay for t in trks]), arr(ring), len(ring),
                                          (C.c_int * n)(*phs))
                assert h, "vsa_driver_create"
                return h
            driver = make_driver(trackers, phases)

        def node_perf(node):
            perf = ovx.query("vxQueryNode", node, ovx.VX_NODE_PERFORMANCE, ovx.vx_perf_t)
            return int(perf.sum), int(perf.num)

        def counts():
            cur = [trk.curr() for trk in trackers]
            return [int((c[:, 0] >= 0).sum()) for c in cur], [len(c) for c in cur]
        api_desc = ("OpenVX graph C API (VX/vx.h): one graph + HIP stream per camera stream, " +
                    (("one persistent native host thread per stream calling vxProcessGraph" if driver else "one Python thread per stream calling vxProcessGraph")
                     if args.host_threads else "one host thread, vxScheduleGraph/vxWaitGraph"))
    else:
        from visualslam_amd.devmem import DeviceImage
        from visualslam_amd.thin_tracker import ThinFeatureTracker
        ring = [DeviceImage(W, H, nvxcu.NVXCU_DF_IMAGE_RGBX, device).upload(f) for f in host_frames]
        trackers = [ThinFeatureTracker(params, W, H, device, torch.cuda.Stream(device)) for _ in my_streams]
        for trk, ph in zip(trackers, phases):
            trk.init(ring[ph].desc)

        def step(k):
            for trk, ph in zip(trackers, phases):
                trk.track(ring[(ph + 1 + k) % args.ring].desc)

        def counts():
            cur = [trk.curr_features() for trk in trackers]
            return [int((c[:, 0] >= 0).sum()) for c in cur], [len(c) for c in cur]
        api_desc = "thin C ABI (NVX/nvxcu.h), eager launches, one HIP stream per camera stream"
    torch.cuda.synchronize()

    def run(k0, n):
        """n steps (= n frames on every camera stream of this GPU) starting at frame index k0"""
        if driver:
            ovx.check(drv.vsa_driver_run(driver, k0, n), "vsa_driver_run")
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

    # A GPU that has been idle sits at its lowest clock (rocm-smi: sclk 534 MHz) and takes tens of milliseconds of load to reach its
    # operating point; W = 5 warm-up steps are over after 1.3 ms (profiles/r03_kernel_experiments.txt).  The same workload is therefore run
    # untimed for --preheat-s seconds first -- set-up, like building the graphs -- and the W warm-up steps and K timed steps follow at once.
    k_at = 0
    preheat_steps = 0
    t_pre = time.perf_counter()
    while args.preheat_s > 0 and time.perf_counter() - t_pre < args.preheat_s:
        run(k_at, 25)
        torch.cuda.synchronize()
        k_at += 25
        preheat_steps += 25
    run(k_at, args.warmup)
    k_at += args.warmup
    torch.cuda.synchronize()
    if args.api == "ovx":
        # HIP events around the PyrLK node of stream 0 only (two event records per frame on that graph's stream)
        ovx.check(L.vxDirective(trackers[0].lk, ovx.VX_DIRECTIVE_ENABLE_PERFORMANCE))
        if args.homography:
            ovx.check(L.vxDirective(trackers[0].hg, ovx.VX_DIRECTIVE_ENABLE_PERFORMANCE))
    else:
        trackers[0].lk_events = []
    barrier(world)
    t0 = time.perf_counter()
    run(k_at, args.steps)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    barrier(world)
    my_elapsed = elapsed
    elapsed = reduce_max(elapsed, world)
    rank_fps = gather_floats(args.steps * args.streams / my_elapsed, world)

    # PyrLK launch duration (HIP events on its stream): (1) over the timed region, where kernels of ~4 streams share the
    # GPU; (2) alone -- stream 0 driven by itself for 50 frames after the timed region, nothing else on the GPU
    lk_alone_ms = lk_loaded_ms = float("nan")
    hg_ms = None
    n_tracks, n_points = counts()
    if args.api == "ovx":
        s1, n1 = node_perf(trackers[0].lk)
        lk_loaded_ms = s1 / max(n1, 1) * 1e-6
        if args.homography:
            hs, hn = node_perf(trackers[0].hg)
            hg_ms = hs / max(hn, 1) * 1e-6

    # The spread of the figure: the same block (same steps, same barriers and synchronisation) again, back to back.  `value` stays the
    # first block's; a block of 20 steps is ~18 ms of GPU time, and how the streams' kernels interleave from a standing start differs
    # from block to block, so one block alone cannot show a regression of a few percent.
    n_rep = args.repeats if args.repeats >= 0 else max(3, min(15, int(8.0 / max(elapsed, 1e-3))))
    rep_fps = []
    k_next = k_at + args.steps
    # the read-back of the counts above left the GPU idle for milliseconds, and its clock with it (the next three 20-step blocks read
    # 32.8 / 35.6 / 37.6 k where the later ones read 38.6-39.0 k): the same untimed pre-heat as before the first block
    t_pre = time.perf_counter()
    while n_rep and args.preheat_s > 0 and time.perf_counter() - t_pre < args.preheat_s:
        run(k_next, 25)
        torch.cuda.synchronize()
        k_next += 25
    for _ in range(n_rep):
        barrier(world)
        r0 = time.perf_counter()
        run(k_next, args.steps)
        torch.cuda.synchronize()
        r_el = time.perf_counter() - r0
        barrier(world)
        rep_fps.append(args.steps * args.streams * world / reduce_max(r_el, world))
        k_next += args.steps

    if args.api == "ovx":
        solo = make_driver(trackers[:1], phases[:1]) if driver else None
        if solo:
            ovx.check(drv.vsa_driver_run(solo, k_next, 10), "solo warm-up")
            s2, n2 = node_perf(trackers[0].lk)
            ovx.check(drv.vsa_driver_run(solo, k_next + 10, 50), "solo run")
            drv.vsa_driver_destroy(solo)
        else:
            run_stream(0, k_next, 10)
            s2, n2 = node_perf(trackers[0].lk)
            run_stream(0, k_next + 10, 50)
        s3, n3 = node_perf(trackers[0].lk)
        lk_alone_ms = (s3 - s2) / max(n3 - n2, 1) * 1e-6
    else:
        pairs = trackers[0].lk_events or []
        if pairs:
            lk_loaded_ms = float(np.mean([a.elapsed_time(b) for a, b in pairs]))
        trackers[0].lk_events = []
        for k in range(60):
            trackers[0].track(ring[(phases[0] + 1 + k_next + k) % args.ring].desc)
        torch.cuda.synchronize()
        lk_alone_ms = float(np.mean([a.elapsed_time(b) for a, b in trackers[0].lk_events[10:]]))

    if rank == 0:
        B = cfg["bytes"]
        bytes_frame = B["pyramid"] + B["detect"] + B["lk"] + B["cvt"]
        frames = args.steps * args.streams * world
        fps = frames / elapsed
        # The OpenVX graph runs the HarrisTrack / FastTrack node's cell pass inside the PyrLK launch (vxVerifyGraph's node fusion): the dominant
        # kernel is then PyrLK + cell pass, and its algorithmic bytes are PyrLK's plus the detector's read of level 0
        lk_alone_name, lk_cells_name = lk_kernel_names(cfg["win"], 0 if cfg["harris"] else 9)
        fused = args.api == "ovx" and "VISUALSLAM_AMD_NO_FUSION" not in os.environ
        lk_bytes = B["lk"] + (W * H if fused else 0)
        lk_name = lk_cells_name if fused else lk_alone_name
        lk_what = ("PyrLK + HarrisTrack cell pass" if cfg["harris"] else "PyrLK + FastTrack cell pass") if fused else "PyrLK"
        lk_gbs = lk_bytes / (max(lk_alone_ms, 1e-9) * 1e-3) / 1e9
        counters = load_profile_json("frame_counters.json") if args.config == "c2" and not args.homography and fused and cfg["harris"] else None
        lk_counters = (counters or {}).get("by_kernel", {}).get("k_pyrlk", {})
        out = {
            "metric": metric_name(args.config, args.homography),
            "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": cfg["workload"] + (" + nvxFindHomographyNode (RANSAC 3.0 px, 56 hypotheses, <=10 refits) on the matches"
                                                      if args.homography else ""),
                       "baseline_config": 4 if args.homography else cfg["baseline_config"], "frames": cfg["frames"],
                       "streams_per_gpu": args.streams, "frames_per_step": args.streams * world,
                       "frame_ring": args.ring, "api": api_desc,
                       "graph_stream_priority": os.environ["VISUALSLAM_AMD_STREAM_PRIORITY"] if args.api == "ovx" else None,
                       "graph_fusions": ovx.graph_fusions(trackers[0].g) if args.api == "ovx" else None},
            "per_rank_frames_per_s": rank_fps,
            "repeat_frames_per_s": ({"median": float(np.median(rep_fps)), "min": min(rep_fps), "max": max(rep_fps), "blocks": len(rep_fps),
                                     "all": [round(v) for v in rep_fps],
                                     "note": "the timed block run again back to back (same steps, barriers and synchronisation) after the same untimed "
                                             "pre-heat; value / ms_per_step are the first block's"} if rep_fps else None),
            "preheat": {"seconds": args.preheat_s, "steps": preheat_steps,
                        "note": "untimed set-up run of the same workload so that the GPU is at its operating clocks; the W warm-up steps follow"},
            "host_cores": cores, "host_threads_total": args.streams * world if (args.api == "ovx" and args.host_threads) else world,
            "tracks_per_sec": fps * float(np.mean(n_tracks)),
            "points_per_frame": float(np.mean(n_points)), "tracked_per_frame": float(np.mean(n_tracks)),
            "frame_algorithmic_GBps": bytes_frame * fps / 1e9,
            "frame_algorithmic_frac_of_hbm_peak": bytes_frame * fps / 1e9 / (HBM_PEAK_GBS * world),
            # the dominant kernel against the HBM roof.  launch_ms: the kernel ALONE (stream 0 driven by itself after the
            # timed region, HIP events, 50 launches); launch_ms_in_timed_region: the same events while the other streams'
            # kernels share the GPU (mean concurrency ~2.3, profiles/r03_bench_concurrency.txt) -- longer, not slower work
            "roofline": {"kernel": f"{lk_name} ({lk_what}, {cfg['capacity']} points x {cfg['levels']} levels, {cfg['win']}x{cfg['win']})",
                         "bound": "hbm", "achieved": lk_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": lk_gbs / HBM_PEAK_GBS,
                         "traffic": lk_counters.get("hbm_bytes"), "launch_ms": lk_alone_ms,
                         # SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE and SQ_INSTS_SALU / SQ_INSTS_VALU of this kernel (profiles/frame_counters.json)
                         "lds_bank_conflict_frac": ratio(lk_counters.get("lds_bank_conflict_cycles"), lk_counters.get("lds_active_cycles")),
                         "salu_per_valu": ratio(lk_counters.get("salu_wave_instructions"), lk_counters.get("valu_wave_instructions")),
                         "launch_ms_in_timed_region": lk_loaded_ms, "algorithmic_bytes_per_launch": lk_bytes,
                         "launch_ms_rocprof_single_stream": profiled_launch_ms(lk_name) if args.config == "c2" and not args.homography else None},
        }
        if counters and lk_counters.get("valu_wave_instructions"):
            # what actually bounds this kernel and the path: vector-instruction issue.  Wave-instructions
            # from the SQ counters (profiles/frame_counters.json, stamped with the commit they were measured at) against the
            # f32 FMA issue rate measured on this GPU
            peak = counters["peak_wave_instructions_per_ns"]
            lk_rate = lk_counters["valu_wave_instructions"] / (lk_alone_ms * 1e6)
            out["roofline_valu"] = {"kernel": lk_name, "bound": "valu issue", "wave_instructions_per_launch": lk_counters["valu_wave_instructions"],
                                    "achieved": lk_rate, "peak": peak, "unit": "G wave-instructions/s", "frac": lk_rate / peak,
                                    "launch_ms": lk_alone_ms}
            per_frame = counters["valu_wave_instructions_per_frame"]
            rate = per_frame * fps / 1e9 / world
            out["valu_issue"] = {"wave_instructions_per_frame": per_frame, "achieved": rate, "peak": peak,
                                 "unit": "G wave-instructions/s per GPU", "frac": rate / peak}
            out["profiles_commit"] = counters.get("commit")
        if args.homography:
            H_, mask = trackers[0].homography()
            out["homography"] = {"node_ms_in_timed_region": hg_ms, "inliers_stream0": int((mask != 0).sum()), "points_stream0": int(len(mask)),
                                 "matrix_stream0": [float(x) for x in H_.T.reshape(-1)]}
        if not args.no_cpu_baseline and world == 1 and not args.homography:
            out["cpu_baseline"] = cpu_baseline(host_frames, params)
        if json_fd is not None:
            sys.stdout.flush()
            os.dup2(json_fd, 1)
        print(json.dumps(out), flush=True)
        if json_fd is not None:
            os.dup2(2, 1)
    if driver:
        drv.vsa_driver_destroy(driver)
    barrier(world)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
