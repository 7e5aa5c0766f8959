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
    ap.add_argument("--verify", action="store_true",
                    help="after the timed region: check frames of the timed output against the oracle")
    ap.add_argument("--no-variants", action="store_true",
                    help="skip the extra figures (pruning off, invalid-disparity kernels, generic "
                         "column encoding, the C++ host class)")
    ap.add_argument("--min-seconds", type=float, default=2.0,
                    help="repeat the timed K-step block until this much time has been measured; "
                         "the MEDIAN block is reported")
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
    (profiles/r02_traffic.json; counters cannot be read from inside the timed run), or None when
    that profile was taken on another mode / shape / batch."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r02_traffic.json")
    try:
        with open(path) as fh:
            t = json.load(fh).get("pairwise" if cfg.pairwise else "unary")
    except (OSError, ValueError):
        return None
    if not t or (t["batch"], t["rows"], t["cols"], t["max_dis"]) != (B, H, W, D):
        return None
    return (2.0 * t["fetch_size_kb"] + t["write_size_kb"]) * 1024.0



def verify_frames(cfg, frames, pick, d_sections, B, C, S):
    """Oracle check of the first, the middle (where a two-stream split would cut the batch) and
    the last frame of the batch that was just timed.  Returns a dict for the JSON line; raises
    SystemExit when a frame differs."""
    import torch
    from oracle import oracle
    from instance_stixels_amd.config import SECTION_DTYPE
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import helpers
    params, lut, odr = oracle.host_initialize(cfg)
    checked = []
    for i in sorted({0, B // 2 - 1 if B > 1 else 0, B // 2, B - 1}):
        if i < 0 or i >= B:
            continue
        f = frames[pick[i]]
        gf, ng, ig, vh = oracle.host_ground(cfg, f.vhor_image, f.camera_tilt, f.camera_height,
                                            f.alpha_ground)
        joined = oracle.join_columns(cfg, f.disparity)
        ref = oracle.compute(params, lut, odr, joined, f.segmentation, gf, ng, ig, vh, cfg.pairwise,
                             want_tables=False)
        got = d_sections[i].cpu().numpy().view(SECTION_DTYPE).reshape(C, S)
        if not helpers.sections_equal(ref["sections"], got):
            raise SystemExit(f"bench.py --verify: frame {i} of the timed batch differs from the oracle")
        checked.append(i)
    return {"frames_checked": checked, "against": "oracle (bit-exact Section arrays)", "ok": True}


def measure_variants(args, cfg, frames, pick, dev, local_rank):
    """Throughput of the same step on the other kernel variants the library contains: pruning
    switched off (the worst case of the branch-and-bound: every (vB, vT) pair is evaluated), an
    invalid-disparity value with 5 % holes (HAS_INVALID kernels), all columns in the generic
    int32/int64 encoding, and a single frame through the C++ `Stixels::Compute` host class."""
    import torch
    from instance_stixels_amd import make_config, host
    from instance_stixels_amd.core import Core
    B = args.batch
    H, W, D = int(cfg.rows), int(cfg.cols), int(cfg.max_dis)
    stream = torch.cuda.current_stream(dev).cuda_stream
    out = {}

    def run(tag, vcfg, disp, seg, env=None, steps=3):
        st = host.Stixels()
        st.SetConfig(vcfg)
        st.PrecomputeHost()
        params = st.GetParameters()
        lut, odr = st.GetLUTs()
        gfs, ngs, igs, vhs = [], [], [], []
        for f in frames:
            st.SetRoadParameters(f.vhor_image, f.camera_tilt, f.camera_height, f.alpha_ground)
            gf, ng, ig, vh = st.GetGroundModel()
            gfs.append(gf); ngs.append(ng); igs.append(ig); vhs.append(vh)
        st.close()
        gf = np.stack([gfs[i] for i in pick]); ng = np.stack([ngs[i] for i in pick])
        ig = np.stack([igs[i] for i in pick]); vh = np.array([vhs[i] for i in pick], np.int32)
        old = {k: os.environ.get(k) for k in (env or {})}
        os.environ.update(env or {})
        try:
            core = Core(params, lut, odr, max_batch=B, device=local_rank)
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        C, S = params.cols, params.max_sections
        d_joined = torch.empty((B, C, H), dtype=torch.float32, device=dev)
        d_sections = torch.empty((B, C, S, 8), dtype=torch.int32, device=dev)

        def step():
            core.join_columns_ptr(disp.data_ptr(), W, vcfg.median_join, d_joined.data_ptr(), B, stream)
            core.compute_ptr(d_joined.data_ptr(), seg.data_ptr(), gf, ng, ig, vh, vcfg.pairwise, B,
                             d_sections.data_ptr(), None, None, None, stream)
        step()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize(dev)
        out[tag] = {"images_per_s": B * steps / (time.perf_counter() - t0), "steps": steps}
        core.close()
        del d_joined, d_sections

    idx = torch.tensor(pick, device=dev)
    disp = torch.from_numpy(np.stack([f.disparity for f in frames])).to(dev)[idx].contiguous()
    seg = torch.from_numpy(np.stack([f.segmentation for f in frames])).to(dev)[idx].contiguous()
    run("pruning_off", cfg, disp, seg, env={"IS_NO_PRUNE": "1"})
    out["pruning_off"]["what"] = ("IS_NO_PRUNE=1: the exact branch-and-bound never fires, every "
                                  "(vB, vT) pair is evaluated (data-independent worst case)")
    # invalid-disparity value 0 with 5 % holes: the HAS_INVALID kernel variants
    icfg = make_config(args.preset, H, W, D, invalid_disparity=0.0)
    g = torch.Generator(device=dev); g.manual_seed(5)
    holes = torch.rand(disp.shape, device=dev, generator=g) < 0.05
    run("invalid_disparity_0", icfg, torch.where(holes, torch.zeros_like(disp), disp), seg)
    out["invalid_disparity_0"]["what"] = "invalid_disparity = 0, 5 % of the pixels invalid (HAS_INVALID kernels)"
    del holes
    # every column in the generic encoding: one negative class value per column
    gseg = seg.clone()
    gseg[:, :, 0, 0] = -1
    run("generic_encoding", cfg, disp, gseg, steps=2)
    out["generic_encoding"]["what"] = ("one negative class value per column: int32 / int64 records, "
                                       "IEEE division, no pruning (the hostile-input path)")
    del gseg, disp, seg
    # BASELINE configs[1] through the C++ host class: Stixels::Compute() incl. the device-side
    # clustering, the D2H copy of the sections and one synchronisation per frame
    st = host.Stixels()
    st.SetConfig(cfg)
    st.SetDevice(local_rank)
    st.Initialize()
    f = frames[0]
    st.SetDisparityImage(f.disparity)
    st.SetSegmentation(f.segmentation)
    st.SetRoadParameters(f.vhor_image, f.camera_tilt, f.camera_height, f.alpha_ground)
    t_plain = st.time_compute(cfg.pairwise, 200, False)
    t_inst = st.time_compute(cfg.pairwise, 200, True)
    st.close()
    out["stixels_compute_host_class"] = {
        "images_per_s": 1.0 / t_plain, "ms_per_frame": t_plain * 1e3,
        "images_per_s_with_GetInstanceStixels": 1.0 / t_inst,
        "what": "one frame per Stixels::Compute() call, timed inside the C++ library "
                "(ish_time_compute): ground model on the host, JoinColumns + DP + back-trace + "
                "instance candidates + clustering on the device, sections copied to the host"}
    return out


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.gpus != world:
        # one process per GPU: N > 1 must come from `python -m torch.distributed.run
        # --nproc-per-node N ... bench.py --gpus N` (nothing here re-executes a process that may
        # have touched the GPU); checked before torch is imported
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE is {world}: launch with "
                         f"python -m torch.distributed.run --nnodes=1 --nproc-per-node {args.gpus} "
                         f"--master-addr 127.0.0.1 --master-port <P> bench.py --gpus {args.gpus} ...")

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

    def timed_block():
        """EXACTLY args.steps steps between barrier + synchronize on both sides; max over ranks."""
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        d = time.perf_counter() - t0
        if use_dist:
            t = torch.tensor([d], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            d = float(t.item())
        return d

    # the K-step block is repeated until --min-seconds have been measured (every rank sees the
    # same max-reduced times, so all ranks stop together) and the MEDIAN block is reported: a
    # 0.3 s measurement is over before a power / utilisation sampler sees the GPU busy
    blocks = [timed_block()]
    while sum(blocks) < args.min_seconds and len(blocks) < 200:
        blocks.append(timed_block())
    dt = float(np.median(blocks))
    # per-kernel durations of the LAST timed step, measured with HIP events on the launch stream
    kt = core.kernel_times_ms()

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

    # ---- --verify: frames of the TIMED output (the batch geometry the value is measured on)
    verify = None
    if args.verify and rank == 0:
        verify = verify_frames(cfg, frames, pick, d_sections if pipe is None else pipe.last_local(),
                               B, C, S)
        if pipe is not None:  # what the RCCL gather delivered to rank 0 is what the ranks computed
            got = pipe.last_gathered()
            same = bool(torch.equal(got[0], pipe.last_local()))
            verify["rccl_gather"] = {"tensors": len(got), "bytes_per_rank": got[0].numel() * 4,
                                     "rank0_copy_equals_local": same}
            if not same:
                raise SystemExit("bench.py --verify: gathered copy of rank 0 differs from its output")

    # ---- extra figures (N = 1): other kernel variants of the same step, never part of `value`
    variants = None
    if world == 1 and not args.no_variants:
        core.close()
        variants = measure_variants(args, cfg, frames, pick, dev, local_rank)
        core = Core(params, lut, odr, max_batch=1, device=local_rank)  # closed again below

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
                         "note": "the column DP is bound by VALU issue and per-step latency, "
                                 "not by HBM (SURVEY.md H1, DESIGN.md sections 6-7): see the valu "
                                 "fields; traffic = (2*FETCH_SIZE + WRITE_SIZE) of the DP kernels "
                                 "per step from the committed rocprofv3 PMC passes "
                                 "(profiles/r02_traffic.json)"},
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
        out["timed_blocks"] = {"count": len(blocks), "steps_per_block": args.steps,
                               "seconds": [round(x, 5) for x in blocks], "reported": "median"}
        if traffic is not None:
            out["roofline"]["measured_hbm_gbps"] = traffic / dp_s / 1e9
            out["roofline"]["measured_hbm_frac"] = traffic / dp_s / 1e9 / HBM_PEAK_GBS
        if verify is not None:
            out["verify"] = verify
        if variants is not None:
            out["variants"] = variants
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
