#!/usr/bin/env python3
"""Column-DP throughput bench (BASELINE.json metric: images/s on 1024x2048x128-disparity frames).

One "step" = one pass of the hot path (JoinColumns + per-column preparation + DP + back-trace)
over a batch of synthetic frames that is already resident in HBM; outputs (Section arrays) stay
on the device.  With N > 1 (launched by torch.distributed.run, one rank per GPU) every rank
processes its own shard of the batch -- the path has no data-path collective -- and the only
communication is the final gather of the stixel outputs to rank 0 over RCCL, inside the timed
region.  Rank 0 prints ONE JSON line.

    python bench.py [--gpus N --steps K --warmup W] [--batch B] [--preset drn_d_22_unary]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s peak
VALU_PEAK_LANEOPS = 78.6e12    # 157.3 TFLOP/s fp32 vector = 78.6 T lane-FMA/s


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=64, help="frames per GPU per step")
    ap.add_argument("--preset", default="drn_d_22_unary")
    ap.add_argument("--rows", type=int, default=1024)
    ap.add_argument("--cols", type=int, default=2048)
    ap.add_argument("--max-dis", type=int, default=128)
    ap.add_argument("--distinct", type=int, default=4, help="distinct synthetic frames per rank")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise RCCL and run the gather pipeline even with one rank (plumbing test)")
    ap.add_argument("--pcie", action="store_true",
                    help="also measure value_incl_h2d_d2h (inputs from pinned host memory each step)")
    ap.add_argument("--no-single", action="store_true",
                    help="skip the extra single-frame (batch 1) measurement")
    ap.add_argument("--no-d2h", action="store_true",
                    help="skip the extra value_incl_d2h measurement")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=6.0)
    ap.add_argument("--no-gather", action="store_true")
    return ap.parse_args()


def usable_cores():
    """Host threads this process may really run on: the affinity mask, capped by the cgroup CPU
    quota when the container has one (os.cpu_count() reports the whole machine)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as fh:
                txt = fh.read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(round(int(txt[0]) / int(txt[1])))))
            else:
                q = int(txt[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fh:
                        n = min(n, max(1, int(round(q / int(fh.read())))))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def cpu_baseline(cfg, frame, target_seconds):
    """The oracle (kind "port": the reference has no CPU path) on a bounded sample of the same
    workload: whole frames, all host cores (OpenMP over stixel columns), repeated until about
    `target_seconds` of wall time have been spent."""
    from oracle import oracle
    params, lut, odr = oracle.host_initialize(cfg)
    gf, ng, ig, vhor = oracle.host_ground(cfg, frame.vhor_image, frame.camera_tilt,
                                          frame.camera_height, frame.alpha_ground)
    joined = oracle.join_columns(cfg, frame.disparity)
    cores = usable_cores()
    run = lambda: oracle.compute(params, lut, odr, joined, frame.segmentation, gf, ng, ig, vhor,
                                 cfg.pairwise, nthreads=cores, want_tables=False)
    run()                                            # warm-up (thread pool, page faults)
    n, t0 = 0, time.perf_counter()
    while True:
        run()
        n += 1
        dt = time.perf_counter() - t0
        if dt >= target_seconds or n >= 64:
            break
    # single-thread figure on a few columns of the same frame, scaled to the frame
    ncol1 = min(4, cfg.realcols)
    t1 = time.perf_counter()
    oracle.compute(params, lut, odr, joined, frame.segmentation, gf, ng, ig, vhor, cfg.pairwise,
                   nthreads=1, want_tables=False, col_range=(0, ncol1))
    dt1 = time.perf_counter() - t1
    return dict(value=n / dt, unit="images/s", cores=cores, kind="port",
                sample=f"{n} x one {cfg.rows}x{cfg.cols}x{cfg.max_dis} frame "
                       f"({cfg.realcols} stixel columns) in {dt:.2f} s wall, OpenMP over columns "
                       f"on {cores} threads = {dt * cores:.0f} core-seconds",
                single_thread_value=ncol1 / dt1 / cfg.realcols,
                single_thread_sample=f"{ncol1} of {cfg.realcols} columns of that frame on one "
                                     f"thread in {dt1:.2f} s, scaled to the frame")


def committed_traffic(cfg, B, H, W, D):
    """HBM bytes per launch of the dominant DP kernel from the committed PMC passes
    (profiles/r01_traffic.json; counters cannot be read from inside the timed run), or None when
    that profile was taken on another mode / shape / batch."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_traffic.json")
    try:
        with open(path) as fh:
            t = json.load(fh).get("pairwise" if cfg.pairwise else "unary")
    except (OSError, ValueError):
        return None
    if not t or (t["batch"], t["rows"], t["cols"], t["max_dis"]) != (B, H, W, D):
        return None
    return (2.0 * t["fetch_size_kb"] + t["write_size_kb"]) * 1024.0


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    import torch
    import torch.distributed as dist
    from instance_stixels_amd import make_config, synthetic, host
    from instance_stixels_amd.core import Core, InstanceBuffers
    from instance_stixels_amd.parallel import PipelinedGather

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_dist  # --force-dist: RCCL plumbing with a single rank
    if use_dist:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            dist.init_process_group("nccl", device_id=dev, rank=0, world_size=1)
        else:
            dist.init_process_group("nccl", device_id=dev)

    cfg = make_config(args.preset, args.rows, args.cols, args.max_dis)
    B = args.batch
    H, W, C, D = int(cfg.rows), int(cfg.cols), cfg.realcols, int(cfg.max_dis)

    # ---- host side: the C++ Stixels class computes the parameter block, LUTs, ground model
    st = host.Stixels()
    st.SetConfig(cfg)
    st.PrecomputeHost()
    params = st.GetParameters()
    lut, odr = st.GetLUTs()
    frames = [synthetic.make_frame(cfg, seed=17 + 101 * rank + i) for i in range(args.distinct)]
    gfs, ngs, igs, vhs = [], [], [], []
    for f in frames:
        st.SetRoadParameters(f.vhor_image, f.camera_tilt, f.camera_height, f.alpha_ground)
        gf, ng, ig, vh = st.GetGroundModel()
        gfs.append(gf); ngs.append(ng); igs.append(ig); vhs.append(vh)
    pick = [i % args.distinct for i in range(B)]
    gf = np.stack([gfs[i] for i in pick]); ng = np.stack([ngs[i] for i in pick])
    ig = np.stack([igs[i] for i in pick]); vh = np.array([vhs[i] for i in pick], np.int32)

    # ---- inputs resident in HBM before the timed region
    d_frames = torch.from_numpy(np.stack([f.disparity for f in frames])).to(dev)
    s_frames = torch.from_numpy(np.stack([f.segmentation for f in frames])).to(dev)
    idx = torch.tensor(pick, device=dev)
    d_big = d_frames[idx].contiguous()                    # [B][H][W] f32
    d_seg = s_frames[idx].contiguous()                    # [B][C][21][P2S] i32
    d_joined = torch.empty((B, C, H), dtype=torch.float32, device=dev)
    S = params.max_sections
    d_sections = torch.empty((B, C, S, 8), dtype=torch.int32, device=dev)
    del d_frames, s_frames

    core = Core(params, lut, odr, max_batch=B, device=local_rank)
    core.set_kernel_timing(True)
    stream = torch.cuda.current_stream(dev).cuda_stream
    # N > 1: the stixel outputs of every step are gathered on rank 0 (RCCL over xGMI); the gather
    # of step k overlaps the compute of step k+1 (double-buffered outputs)
    pipe = PipelinedGather(d_sections, depth=2, dst=0) if (use_dist and not args.no_gather) else None

    def step():
        out = pipe.next_buffer() if pipe is not None else d_sections
        core.join_columns_ptr(d_big.data_ptr(), W, cfg.median_join, d_joined.data_ptr(), B, stream)
        core.compute_ptr(d_joined.data_ptr(), d_seg.data_ptr(), gf, ng, ig, vh, cfg.pairwise, B,
                         out.data_ptr(), None, None, None, stream)
        if pipe is not None:
            pipe.submit()

    def barrier():
        if pipe is not None:
            pipe.flush()
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    # per-kernel durations of the LAST timed step, measured with HIP events on the launch stream
    kt = core.kernel_times_ms()

    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # BASELINE.json configs[1] (ONE 1024x2048 frame per call) next to the batched headline value:
    # the same entry points with n_images = 1, i.e. the latency a per-frame caller sees
    single = None
    if world == 1 and not args.no_single:
        def step1():
            core.join_columns_ptr(d_big.data_ptr(), W, cfg.median_join, d_joined.data_ptr(), 1, stream)
            core.compute_ptr(d_joined.data_ptr(), d_seg.data_ptr(), gf[:1], ng[:1], ig[:1], vh[:1],
                             cfg.pairwise, 1, d_sections.data_ptr(), None, None, None, stream)
        for _ in range(10):
            step1()
        torch.cuda.synchronize(dev)
        n1 = 100
        t1 = time.perf_counter()
        for _ in range(n1):
            step1()
        torch.cuda.synchronize(dev)
        single = (time.perf_counter() - t1) / n1

    # second figure, N = 1 only and outside the judged `value`: the same step
    # followed by the D2H copy of the Section output into pinned host memory (what
    # Stixels::Compute does, Stixels.cu:629-633)
    d2h_value = None
    if world == 1 and not args.no_d2h:
        h_sections = torch.empty(d_sections.shape, dtype=d_sections.dtype, pin_memory=True)
        for _ in range(2):
            step(); h_sections.copy_(d_sections, non_blocking=True)
        torch.cuda.synchronize(dev)
        k = max(2, min(args.steps, 5))
        t1 = time.perf_counter()
        for _ in range(k):
            step(); h_sections.copy_(d_sections, non_blocking=True)
        torch.cuda.synchronize(dev)
        d2h_value = B * k / (time.perf_counter() - t1)

    # PCIe-inclusive figure for DESIGN.md (--pcie): inputs come from pinned host memory every step
    # and the sections go back, all on the compute stream, nothing overlapped
    pcie_value = None
    if world == 1 and args.pcie:
        h_big = torch.empty(d_big.shape, dtype=d_big.dtype, pin_memory=True).copy_(d_big)
        h_seg = torch.empty(d_seg.shape, dtype=d_seg.dtype, pin_memory=True).copy_(d_seg)
        h_out = torch.empty(d_sections.shape, dtype=d_sections.dtype, pin_memory=True)
        def step_pcie():
            d_big.copy_(h_big, non_blocking=True); d_seg.copy_(h_seg, non_blocking=True)
            step(); h_out.copy_(d_sections, non_blocking=True)
        step_pcie(); torch.cuda.synchronize(dev)
        k = 3
        t1 = time.perf_counter()
        for _ in range(k):
            step_pcie()
        torch.cuda.synchronize(dev)
        pcie_value = B * k / (time.perf_counter() - t1)

    if rank == 0:
        images = B * world * args.steps
        value = images / dt
        alg_bytes_img = synthetic.algorithmic_bytes_per_image(cfg)
        pairs_img = synthetic.pair_evaluations_per_image(cfg)
        dp_s = kt["dp_ms"] * 1e-3
        achieved = alg_bytes_img * B / dp_s / 1e9
        traffic = committed_traffic(cfg, B, H, W, D)
        out = {
            "metric": "images/s on 1024x2048x128-disp column DP",
            "value": value, "unit": "images/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32+i32", "data": "synthetic",
            "config": {"workload": f"C2/C3: {B} frames/GPU of {H}x{W}, {D} disparity bins, "
                                   f"19 classes + 2 offset channels, preset {args.preset} "
                                   f"({'pairwise' if cfg.pairwise else 'unary'}), JoinColumns + "
                                   "prepare + DP + back-trace, device-resident in/out",
                       "batch_per_gpu": B, "rows": H, "cols": W, "max_dis": D,
                       "preset": args.preset,
                       "parallelism": f"batch shards x{world}, RCCL gather of sections to rank 0 "
                                      "(overlapped with the next step)"
                                      if world > 1 else "single GPU"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "k_pw_phase1 + k_pw_phase2, all tiles of the step"
                                   if cfg.pairwise else "k_dp_unary",
                         "kernel_ms": kt["dp_ms"],
                         "algorithmic_bytes_per_image": alg_bytes_img,
                         "note": "the column DP is bound by VALU issue, not by HBM (SURVEY.md "
                                 "H1, DESIGN.md section 5): see the valu fields; traffic = "
                                 "(2*FETCH_SIZE + WRITE_SIZE) per launch from the committed "
                                 "rocprofv3 PMC passes (profiles/r01_traffic.json)"},
            "valu": {"pair_evals_per_s": pairs_img * B / dp_s,
                     "pair_evals_per_image": pairs_img,
                     "lane_ops_peak_per_s": VALU_PEAK_LANEOPS},
            "kernel_ms": kt,
        }
        if d2h_value is not None:
            out["value_incl_d2h"] = d2h_value
        if pcie_value is not None:
            out["value_incl_h2d_d2h"] = pcie_value
        if single is not None:
            out["single_frame"] = {"workload": "BASELINE configs[1]: one frame per call (batch 1), "
                                               "device-resident in/out",
                                   "images_per_s": 1.0 / single, "ms_per_frame": single * 1e3}
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(cfg, frames[0], args.cpu_seconds)
    else:
        out = None

    core.close()

    def flush_c_stdio():
        sys.stdout.flush()
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass

    # RCCL writes a version banner into the C-level stdout buffer, which would otherwise be flushed
    # at process exit, i.e. AFTER the result: every rank flushes, all ranks meet, the process
    # group goes away, and only then rank 0 prints -- the JSON line is the last line of the job
    if use_dist:
        flush_c_stdio()
        dist.barrier()
        dist.destroy_process_group()
    flush_c_stdio()
    if out is not None:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
