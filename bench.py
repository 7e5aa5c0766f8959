#!/usr/bin/env python3
"""Column-DP throughput bench (BASELINE.json metric: images/s on 1024x2048x128-disparity frames).

One "step" = one pass of the hot path (JoinColumns + per-column preparation + DP + back-trace)
over a batch of synthetic frames that is already resident in HBM; outputs (Section arrays) stay
on the device.  With N > 1 (launched by torch.distributed.run, one rank per GPU) every rank
processes its own shard of the batch -- the path has no data-path collective -- and the only
communication is the final gather of the stixel outputs to rank 0 over RCCL, inside the timed
region.  Rank 0 prints ONE JSON line.

    python bench.py [--gpus N --steps K --warmup W] [--batch B] [--preset drn_d_22_unary]

stdout carries ONE compact JSON line (< 4 KB: compact_line()); the complete object (per-kernel times,
timed blocks, every oracle check, the --full sweeps) goes to bench_full.json next to this file (and
under gpurun_out/ when that directory exists).

The default run (N = 1, about half a minute) also reports, outside `value`:
  verify      frames of the TIMED output against the CPU oracle (bit-exact Section arrays)
  prune       what the exact branch-and-bound evaluated (device counters, separate untimed pass)
  value_pruning_off / value_floor_families (the slowest known input family, cityscapes_like; with
              --full every family) / lut_fused_repaired over the timed steps / cpu_baseline
--full adds `variants` and `value_spread`: three more input families, the OTHER model (pairwise when
the preset is unary) at the same batch with its own roofline / pruning-off / verify, BASELINE
configs[4] (1024x4096x256) in both modes, configs[0] (512x1024x64, disparity only) on the CPU and
the GPU, invalid-disparity kernels, generic column encoding, the C++ host class.

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE starts `python -m torch.distributed.run
--nproc-per-node N bench.py --gpus N ...` as a CHILD process (before anything here touches the GPU)
and relays rank 0's line as its own last line.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

VERIFIED = []                  # (what, ok) of every oracle check of this run -> verify_all_ok
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s peak
VALU_PEAK_LANEOPS = 78.6e12    # 157.3 TFLOP/s fp32 vector = 78.6 T lane-FMA/s
OTHER_PRESET = {"drn_d_22_unary": "drn_d_38_pairwise", "drn_d_38_unary": "drn_d_38_pairwise",
                "drn_d_22_pairwise": "drn_d_22_unary", "drn_d_38_pairwise": "drn_d_22_unary"}


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
    ap.add_argument("--invalid-disparity", type=float, default=None,
                    help="invalid_disparity of the configuration (e.g. 0: 5 %% of the pixels invalid, the HAS_INVALID kernels)")
    ap.add_argument("--family", default="scene", help="synthetic input family (synthetic.FAMILIES)")
    ap.add_argument("--distinct", type=int, default=0,
                    help="distinct synthetic frames per rank (default 0 = --batch: every frame of the batch is its own scene)")
    ap.add_argument("--spread-batches", type=int, default=8,
                    help="--full: disjoint batches of distinct frames behind value_spread (0 = skip)")
    ap.add_argument("--full", action="store_true",
                    help="also run the sweeps behind `variants` / `value_spread` (minutes; into bench_full.json)")
    ap.add_argument("--out", default=None,
                    help="where the complete result object goes (default: bench_full.json next to bench.py)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise RCCL and run the gather pipeline even with one rank (plumbing test)")
    ap.add_argument("--gather", choices=("fixed", "compact"), default="compact",
                    help="N > 1: gather fixed-stride Section tensors, or per-column counts + packed sections")
    ap.add_argument("--pcie", action="store_true",
                    help="also measure value_incl_h2d_d2h (inputs from pinned host memory each step)")
    ap.add_argument("--no-single", action="store_true",
                    help="skip the extra single-frame (batch 1) measurement")
    ap.add_argument("--no-d2h", action="store_true",
                    help="skip the extra value_incl_d2h measurement")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=6.0)
    ap.add_argument("--no-gather", action="store_true")
    ap.add_argument("--no-verify", action="store_true",
                    help="skip the oracle check of the timed output (about one CPU-second per frame)")
    ap.add_argument("--verify", action="store_true", help="(default since round 3; kept for old command lines)")
    ap.add_argument("--side-stream", action="store_true",
                    help="run the steps on a torch side stream instead of the (legacy NULL) default stream")
    ap.add_argument("--no-prune-stats", action="store_true",
                    help="skip the untimed device-counter pass (profiling runs: its atomics distort per-kernel averages)")
    ap.add_argument("--no-variants", action="store_true",
                    help="(kept for old command lines: the sweeps only run with --full) also skips the "
                         "pruning-off / floor-family figures of the default run")
    ap.add_argument("--min-seconds", type=float, default=1.0,
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


def cpu_baseline(cfg, frame, target_seconds, single_thread=True):
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
    out = dict(value=n / dt, unit="images/s", cores=cores, kind="port",
               sample=f"{n} x one {cfg.rows}x{cfg.cols}x{cfg.max_dis} frame "
                      f"({cfg.realcols} stixel columns) in {dt:.2f} s wall, OpenMP over columns "
                      f"on {cores} threads = {dt * cores:.0f} core-seconds")
    if single_thread:   # single-thread figure on a few columns of the same frame, scaled to the frame
        ncol1 = min(4, cfg.realcols)
        t1 = time.perf_counter()
        oracle.compute(params, lut, odr, joined, frame.segmentation, gf, ng, ig, vhor, cfg.pairwise,
                       nthreads=1, want_tables=False, col_range=(0, ncol1))
        dt1 = time.perf_counter() - t1
        out["single_thread_value"] = ncol1 / dt1 / cfg.realcols
        out["single_thread_sample"] = (f"{ncol1} of {cfg.realcols} columns of that frame on one "
                                       f"thread in {dt1:.2f} s, scaled to the frame")
    return out


def committed_traffic(cfg, B, H, W, D):
    """HBM bytes per step of the DP kernels from the committed PMC passes (profiles/rNN_traffic.json,
    newest round first; counters cannot be read from inside the timed run), or None when no profile
    was taken on this mode / shape / batch.  Returns (bytes, file name)."""
    t, name = committed_profile(cfg, B, H, W, D)
    if t is None:
        return None, None
    return (2.0 * t["fetch_size_kb"] + t["write_size_kb"]) * 1024.0, name


def committed_profile(cfg, B, H, W, D):
    """The newest profiles/rNN_traffic.json entry of this mode / shape / batch (or None)."""
    prof = os.path.join(ROOT, "profiles")
    for name in ("r06_traffic.json", "r05_traffic.json", "r04_traffic.json", "r03_traffic.json", "r02_traffic.json"):
        try:
            with open(os.path.join(prof, name)) as fh:
                t = json.load(fh).get("pairwise" if cfg.pairwise else "unary")
        except (OSError, ValueError):
            continue
        if not t or (t["batch"], t["rows"], t["cols"], t["max_dis"]) != (B, H, W, D):
            continue
        return t, name
    return None, None


def committed_valu(cfg, B, H, W, D):
    """VALU issue fraction of the dominant DP kernel from the committed PMC passes:
    the path's true roofline -- it is bound by vector issue, not by HBM."""
    t, name = committed_profile(cfg, B, H, W, D)
    v = (t or {}).get("valu")
    if not v:
        return None
    return {"issue_frac": v.get("issue_frac"), "issue_frac_pmc_upper": v["issue_frac_pmc_upper"],
            "kernel": v["kernel"], "sq_insts_valu": v["sq_insts_valu"],
            "sq_active_inst_valu_quadcycles": v["sq_active_inst_valu_quadcycles"],
            "kernel_cycles": v["kernel_cycles_grbm_gui_active_over_8"],
            "isa_cycles_per_valu_inst": v["isa_cycles_per_valu_inst"], "formula": v["formula"],
            "source": f"profiles/{name}"}


def gen_frames(cfg, seeds, family="scene", zero_segmentation=False):
    """synthetic.make_frame for a list of seeds on a thread pool (numpy releases the GIL in the large
    fills): 64 distinct 1024x2048 frames cost 1-2 s instead of 6.  `family` may be a list (one per seed)."""
    from concurrent.futures import ThreadPoolExecutor
    from instance_stixels_amd import synthetic
    fams = [family] * len(seeds) if isinstance(family, str) else list(family)
    one = lambda a: synthetic.make_frame(cfg, seed=a[0], family=a[1], zero_segmentation=zero_segmentation)
    if len(seeds) <= 2:
        return [one(a) for a in zip(seeds, fams)]
    with ThreadPoolExecutor(max(1, min(usable_cores(), 16, len(seeds)))) as ex:
        return list(ex.map(one, zip(seeds, fams)))


class Workload:
    """One configuration's batch, resident in HBM: host tables through the C++ Stixels class,
    `distinct` synthetic frames repeated to `batch` (or the given `frames`), device inputs and output
    buffers."""

    side_stream = False   # --side-stream: steps on a torch side stream instead of the default stream

    def __init__(self, preset, H, W, D, batch, distinct, dev, local_rank, seed0=17, family="scene",
                 frames=None, **overrides):
        import torch
        from instance_stixels_amd import make_config, synthetic, host
        self.torch = torch
        self.dev, self.local_rank, self.B = dev, local_rank, batch
        self.cfg = cfg = make_config(preset, H, W, D, **overrides)
        distinct = len(frames) if frames is not None else max(1, min(distinct or batch, batch))
        self.distinct, self.family = distinct, family
        self.H, self.W, self.D, self.C = int(cfg.rows), int(cfg.cols), int(cfg.max_dis), cfg.realcols
        st = host.Stixels()
        st.SetConfig(cfg)
        st.PrecomputeHost()
        self.params = st.GetParameters()
        self.lut, self.odr = st.GetLUTs()
        zero_seg = preset.startswith("disparity_only")
        self.frames = frames if frames is not None else gen_frames(
            cfg, [seed0 + i for i in range(distinct)], family, zero_seg)
        g = []
        for f in self.frames:
            st.SetRoadParameters(f.vhor_image, f.camera_tilt, f.camera_height, f.alpha_ground)
            g.append(st.GetGroundModel())
        st.close()
        self.pick = pick = [i % distinct for i in range(batch)]
        self.gf = np.stack([g[i][0] for i in pick]); self.ng = np.stack([g[i][1] for i in pick])
        self.ig = np.stack([g[i][2] for i in pick]); self.vh = np.array([g[i][3] for i in pick], np.int32)
        idx = torch.tensor(pick, device=dev)
        self.d_big = torch.from_numpy(np.stack([f.disparity for f in self.frames])).to(dev)[idx].contiguous()
        self.d_seg = torch.from_numpy(np.stack([f.segmentation for f in self.frames])).to(dev)[idx].contiguous()
        self.S = self.params.max_sections
        self.d_joined = torch.empty((batch, self.C, self.H), dtype=torch.float32, device=dev)
        self.d_sections = torch.empty((batch, self.C, self.S, 8), dtype=torch.int32, device=dev)
        self._side = torch.cuda.Stream(dev) if Workload.side_stream else None
        self.stream = (self._side or torch.cuda.current_stream(dev)).cuda_stream

    def make_core(self, env=None, max_batch=None):
        """A context; `env`: IS_* knobs, which the library reads once, when a context is created."""
        from instance_stixels_amd.core import Core
        old = {k: os.environ.get(k) for k in (env or {})}
        os.environ.update(env or {})
        try:
            return Core(self.params, self.lut, self.odr, max_batch=max_batch or self.B,
                        device=self.local_rank)
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v

    def step(self, core, out=None, n=None, instances=None):
        n = n or self.B
        out = self.d_sections if out is None else out
        core.join_columns_ptr(self.d_big.data_ptr(), self.W, self.cfg.median_join,
                              self.d_joined.data_ptr(), n, self.stream)
        core.compute_ptr(self.d_joined.data_ptr(), self.d_seg.data_ptr(), self.gf[:n], self.ng[:n],
                         self.ig[:n], self.vh[:n], self.cfg.pairwise, n, out.data_ptr(), instances,
                         None, None, self.stream)

    def time_steps(self, core, steps, **kw):
        for _ in range(2):   # (two warm-up steps: the first one of a fresh context also loads code objects)
            self.step(core, **kw)
        self.torch.cuda.synchronize(self.dev)
        t0 = time.perf_counter()
        for _ in range(steps):
            self.step(core, **kw)
        self.torch.cuda.synchronize(self.dev)
        return (time.perf_counter() - t0) / steps

    def instance_buffers(self):
        """Per-image candidate / label arrays for the batched instance path."""
        from instance_stixels_amd.core import InstanceBuffers
        from instance_stixels_amd.config import INSTANCE_CLASSES
        torch, n, slots = self.torch, self.B, self.C * self.S
        com = torch.zeros((n, INSTANCE_CLASSES, slots, 2), dtype=torch.float32, device=self.dev)
        idx = torch.zeros((n, INSTANCE_CLASSES, slots, 2), dtype=torch.int32, device=self.dev)
        cand = torch.zeros((n, INSTANCE_CLASSES, slots), dtype=torch.uint8, device=self.dev)
        per = torch.zeros((n, INSTANCE_CLASSES), dtype=torch.int32, device=self.dev)
        lab = torch.zeros((n, INSTANCE_CLASSES, slots), dtype=torch.int32, device=self.dev)
        pk = torch.zeros((n, 1 + 3 * INSTANCE_CLASSES * slots), dtype=torch.int32, device=self.dev)
        keep = (com, idx, cand, per, lab, pk)
        return keep, [InstanceBuffers(com[i].data_ptr(), idx[i].data_ptr(), cand[i].data_ptr(),
                                      per[i].data_ptr(), lab[i].data_ptr(), pk[i].data_ptr())
                      for i in range(n)]

    def prune_stats(self, core):
        """Device counters of one untimed step: what the branch-and-bound evaluated.  All columns
        of the synthetic families are FAST columns (class values >= 0), whose walks are counted."""
        core.set_eval_counters(True)
        self.step(core)
        c = core.eval_counters()
        core.set_eval_counters(False)
        H, ncols = self.H, self.B * self.C
        tiles = [min(64, H - lo) for lo in range(0, H, 64)]
        diag = sum(n * (n - 1) // 2 for n in tiles)        # vB inside the tile: always evaluated
        nominal = ncols * H * (H + 1) // 2
        if self.cfg.pairwise:
            full, gs = c["p1_full"], c["p1_gs"]
        else:
            full, gs = c["unary_full"], c["unary_gs"]
        ev = ncols * diag + 64 * (full + gs)
        return {"evaluated_frac": ev / nominal, "full_eval_frac": (ncols * diag + 64 * full) / nominal,
                "ground_sky_only_frac": 64 * gs / nominal, "pairs_nominal": nominal,
                "pairs_evaluated": ev,
                "how": "device counters of one untimed step (is_set_eval_counters): 64-pair wave-steps "
                       "below the diagonal blocks + the always-evaluated diagonal blocks, over the "
                       "nominal C*H*(H+1)/2 pairs; full = all three candidates, ground_sky_only = the "
                       "cheap steps after the object bound has closed"}

    def verify(self, d_sections, images=None, fatal=False):
        """Oracle check of frames of a finished step (first, either side of the middle, last).  A
        mismatch is reported as ok = False (and collected into the line's verify_all_ok); with `fatal`
        -- the headline value -- the bench stops instead: a number whose output differs is no number."""
        from oracle import oracle
        from instance_stixels_amd.config import SECTION_DTYPE
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import helpers
        cfg, B = self.cfg, self.B
        params, lut, odr = oracle.host_initialize(cfg)
        checked, bad = [], []
        for i in (sorted({0, B // 2 - 1 if B > 1 else 0, B // 2, B - 1}) if images is None else images):
            if i < 0 or i >= B:
                continue
            f = self.frames[self.pick[i]]
            gf, ng, ig, vh = oracle.host_ground(cfg, f.vhor_image, f.camera_tilt, f.camera_height,
                                                f.alpha_ground)
            joined = oracle.join_columns(cfg, f.disparity)
            ref = oracle.compute(params, lut, odr, joined, f.segmentation, gf, ng, ig, vh, cfg.pairwise,
                                 want_tables=False)
            got = d_sections[i].cpu().numpy().view(SECTION_DTYPE).reshape(self.C, self.S)
            if not helpers.sections_equal(ref["sections"], got):
                if fatal:
                    raise SystemExit(f"bench.py verify: frame {i} of the timed batch "
                                     f"({cfg.rows}x{cfg.cols}, pairwise={cfg.pairwise}) differs from the oracle")
                bad.append(i)
            checked.append(i)
        out = {"frames_checked": checked, "against": "oracle (bit-exact Section arrays)", "ok": not bad}
        if bad:
            out["frames_differing"] = bad
        VERIFIED.append((f"{cfg.rows}x{cfg.cols}x{cfg.max_dis} {'pairwise' if cfg.pairwise else 'unary'} "
                         f"batch {B} family {self.family} invalid {cfg.invalid_disparity:g}", not bad))
        return out

    def roofline(self, dp_ms, kernel, traffic=None, traffic_file=None):
        from instance_stixels_amd import synthetic
        alg = synthetic.algorithmic_bytes_per_image(self.cfg)
        achieved = alg * self.B / (dp_ms * 1e-3) / 1e9
        r = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
             "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "kernel": kernel, "kernel_ms": dp_ms,
             "algorithmic_bytes_per_image": alg}
        if traffic is not None:
            r["measured_hbm_gbps"] = traffic / (dp_ms * 1e-3) / 1e9
            r["measured_hbm_frac"] = r["measured_hbm_gbps"] / HBM_PEAK_GBS
            r["traffic_source"] = f"profiles/{traffic_file}"
        return r

    def free(self):
        del self.d_big, self.d_seg, self.d_joined, self.d_sections


def dp_kernel_name(cfg):
    return ("k_pw_phase1 + k_pw_phase2, all 64-row tiles of the step" if cfg.pairwise
            else "k_dp_unary_fast (FAST columns; k_dp_unary takes generic columns: none here); in calls of >= 2048 "
                 "columns the launch also holds the object-LUT units of the prepare step (their 8.6 GB of writes per 64 "
                 "frames are part of `traffic`, their time part of kernel_ms)")


def measure_mode(wl, steps=3, with_pruning_off=True, with_verify=True, with_prune=True, with_single=False):
    """images/s, DP time, roofline, pruning statistics, pruning-off figure and oracle check of one
    workload -- the fields a non-default mode / shape reports under `variants`."""
    core = wl.make_core()
    core.set_kernel_timing(True)
    dt = wl.time_steps(core, steps)
    kt = core.kernel_times_ms()
    traffic, tfile = committed_traffic(wl.cfg, wl.B, wl.H, wl.W, wl.D)
    out = {"preset_pairwise": bool(wl.cfg.pairwise), "batch": wl.B, "images_per_s": wl.B / dt,
           "ms_per_step": dt * 1e3, "dp_ms": kt["dp_ms"], "kernel_ms": kt, "steps": steps,
           "roofline": wl.roofline(kt["dp_ms"], dp_kernel_name(wl.cfg), traffic, tfile),
           "valu": committed_valu(wl.cfg, wl.B, wl.H, wl.W, wl.D)}
    if with_verify:
        out["verify"] = wl.verify(wl.d_sections, images=[0, wl.B - 1] if wl.B > 1 else [0])
    if with_prune:
        out["prune"] = wl.prune_stats(core)
    if with_single and wl.B > 1:   # BASELINE configs[1]: ONE frame per call, the latency a per-frame caller sees
        for _ in range(5):
            wl.step(core, n=1)
        wl.torch.cuda.synchronize(wl.dev)
        t1 = time.perf_counter()
        for _ in range(50):
            wl.step(core, n=1)
        wl.torch.cuda.synchronize(wl.dev)
        single = (time.perf_counter() - t1) / 50
        out["single_frame"] = {"workload": "BASELINE configs[1]: one frame per call (batch 1), device-resident "
                                           "in/out", "images_per_s": 1.0 / single, "ms_per_frame": single * 1e3}
    core.close()
    if with_pruning_off:
        core = wl.make_core(env={"IS_NO_PRUNE": "1"})
        core.set_kernel_timing(True)
        dt0 = wl.time_steps(core, max(2, steps - 1))
        out["pruning_off"] = {"images_per_s": wl.B / dt0, "dp_ms": core.kernel_times_ms()["dp_ms"]}
        core.close()
    return out


def measure_in_flight(wl, counts, steps=12):
    """The headline step with several contexts in flight (tools/experiments/two_ctx.py): context k on its own stream
    takes every k-th batch; all results are compared with the single-context output."""
    torch = wl.torch
    res = {"what": "N contexts on N streams, batches alternate between them; images/s over all of them "
                   "(the headline `value` is ONE context, one batch at a time)"}
    for n in counts:
        cores = [wl.make_core() for _ in range(n)]
        streams = [torch.cuda.Stream(wl.dev) for _ in range(n)]
        joined = [torch.empty_like(wl.d_joined) for _ in range(n)]
        # (a step writes a column's sections up to its terminator: zeroed buffers compare as wholes)
        outs = [torch.zeros_like(wl.d_sections) for _ in range(n)]
        ref = torch.zeros_like(wl.d_sections)
        wl.step(cores[0], out=ref)
        torch.cuda.synchronize(wl.dev)
        if "verify" not in res:   # the output every context's result is compared with, against the oracle
            res["verify"] = wl.verify(ref, images=[wl.B - 1])

        def step(i):
            k = i % n
            sp = streams[k].cuda_stream
            cores[k].join_columns_ptr(wl.d_big.data_ptr(), wl.W, wl.cfg.median_join, joined[k].data_ptr(), wl.B, sp)
            cores[k].compute_ptr(joined[k].data_ptr(), wl.d_seg.data_ptr(), wl.gf, wl.ng, wl.ig, wl.vh,
                                 wl.cfg.pairwise, wl.B, outs[k].data_ptr(), None, None, None, sp)

        for i in range(2 * n):
            step(i)
        torch.cuda.synchronize(wl.dev)
        t0 = time.perf_counter()
        for i in range(steps):
            step(i)
        torch.cuda.synchronize(wl.dev)
        dt = (time.perf_counter() - t0) / steps
        res[str(n)] = {"images_per_s": wl.B / dt, "ms_per_step": dt * 1e3, "steps": steps,
                       "outputs_equal_single_context": bool(all(torch.equal(ref, o) for o in outs))}
        if not wl.cfg.pairwise:   # did a fused LUT + DP launch of a context ever distrust its hand-over beside the others'?
            res[str(n)]["lut_fused_repaired_last_call"] = [c.lut_fused_repaired() for c in cores]
        for c in cores:
            c.close()
        del cores, streams, joined, outs, ref
        torch.cuda.empty_cache()
    return res


EXTRA_FAMILIES = ("iid_noise", "low_confidence", "flat_disparity", "homogeneous", "many_thin_objects",
                  "noisy_disparity", "cityscapes_like")


def measure_families(preset, H, W, D, B, dev, local_rank, distinct=2):
    """images/s and what the branch-and-bound evaluated on the other input families of
    synthetic.make_frame (the headline family is "scene"), same preset, shape and batch; one frame of
    every timed output is compared with the oracle."""
    import torch
    fam = {}
    for name in EXTRA_FAMILIES + ("mixed",):
        if name == "mixed":   # every frame of the batch from another family, distinct seeds: do slow frames gate a batch?
            from instance_stixels_amd import make_config, synthetic
            cfg = make_config(preset, H, W, D)
            frames = gen_frames(cfg, [700 + i for i in range(B)],
                                [synthetic.FAMILIES[i % len(synthetic.FAMILIES)] for i in range(B)])
            wf = Workload(preset, H, W, D, B, B, dev, local_rank, family=name, frames=frames)
        else:
            wf = Workload(preset, H, W, D, B, distinct, dev, local_rank, family=name)
        core = wf.make_core()
        dtf = wf.time_steps(core, 3)
        v = wf.verify(wf.d_sections, images=[B - 1])
        ps = wf.prune_stats(core)
        fam[name] = {"images_per_s": B / dtf, "evaluated_frac": ps["evaluated_frac"],
                     "full_eval_frac": ps["full_eval_frac"], "distinct_frames": wf.distinct, "verify": v}
        core.close(); wf.free(); del wf
        torch.cuda.empty_cache()
    fam["what"] = ("synthetic.make_frame(family=...): iid_noise = every class logit N(0,1), no scene in "
                   "the segmentation; low_confidence = true-class logit +1..2 instead of +4..5; "
                   "flat_disparity = the scene's segmentation over constant + U(0,1) disparity; "
                   "homogeneous = road below the horizon, sky above, no object, confident CNN (logit +8..9); "
                   "many_thin_objects = sixty slabs 8..24 px wide; noisy_disparity = the scene with N(0,3) "
                   "disparity noise; cityscapes_like = spatially correlated CNN errors, a building band, 0.25 px "
                   "disparity noise (see synthetic.make_frame); mixed = frame i of the batch from family i mod 8, every frame its own "
                   "seed; the others: 2 distinct frames repeated to the batch (value_spread has whole "
                   "batches of distinct frames for the scene and floor families)")
    return fam


def measure_spread(presets, H, W, D, B, dev, local_rank, families, n_batches):
    """Sampling error of a data-dependent number: `n_batches` DISJOINT batches of B distinct frames per
    family (frame generation does not depend on the preset, so every batch serves both models): images/s
    and evaluated_frac of each batch -> min / median / max."""
    import torch
    from instance_stixels_amd import make_config
    out = {}
    for fam in families:
        rows = {p: [] for p in presets}
        for b in range(n_batches):
            frames = gen_frames(make_config(presets[0], H, W, D), [50000 + 1000 * b + i for i in range(B)], fam)
            for preset in presets:
                w = Workload(preset, H, W, D, B, B, dev, local_rank, family=fam, frames=frames)
                core = w.make_core()
                dt = w.time_steps(core, 3)
                ps = w.prune_stats(core)
                rows[preset].append((B / dt, ps["evaluated_frac"]))
                core.close(); w.free(); del w
                torch.cuda.empty_cache()
            del frames
        for preset in presets:
            v = np.array([r[0] for r in rows[preset]]); e = np.array([r[1] for r in rows[preset]])
            out.setdefault(preset, {})[fam] = {
                "images_per_s": {"min": float(v.min()), "median": float(np.median(v)), "max": float(v.max())},
                "evaluated_frac": {"min": float(e.min()), "median": float(np.median(e)), "max": float(e.max())},
                "batches": n_batches, "frames_per_batch": B, "images_per_s_all": [round(float(x), 1) for x in v]}
    out["what"] = (f"{n_batches} disjoint batches of {B} distinct frames (seeds 50000 + 1000 b + i) per family, "
                   "3 timed steps each, wall clock around the steps; the pruning counters from an untimed pass")
    return out


def floor_over_families(scene_value, fam):
    """The smallest images/s over every measured input family (the headline family included)."""
    vals = {"scene": scene_value}
    vals.update({k: v["images_per_s"] for k, v in fam.items() if isinstance(v, dict) and k != "mixed"})
    worst = min(vals, key=vals.get)
    return {"images_per_s": vals[worst], "family": worst, "families_measured": sorted(vals)}


def measure_variants(args, wl, dev, local_rank):
    """Figures of the same step on the other kernel variants, input families, the other model and
    the other BASELINE shapes.  Never part of `value`."""
    import torch
    from instance_stixels_amd import host
    B, H, W, D = wl.B, wl.H, wl.W, wl.D
    out = {}

    # ---- pruning off on the headline workload
    core = wl.make_core(env={"IS_NO_PRUNE": "1"})
    core.set_kernel_timing(True)
    dt = wl.time_steps(core, 3)
    out["pruning_off"] = {
        "images_per_s": B / dt, "dp_ms": core.kernel_times_ms()["dp_ms"], "steps": 3,
        "evaluated_frac": wl.prune_stats(core)["evaluated_frac"],
        "what": "IS_NO_PRUNE=1: the exact branch-and-bound never fires, every (vB, vT) pair is "
                "evaluated (data-independent worst case)"}
    core.close()

    # ---- the batched instance path inside the step (candidates + clustering of all frames)
    core = wl.make_core()
    keep, ibs = wl.instance_buffers()
    dt_i = wl.time_steps(core, 3, instances=ibs)
    dt_p = wl.time_steps(core, 3)
    out["with_instances"] = {
        "images_per_s": B / dt_i, "images_per_s_without": B / dt_p, "ratio": dt_p / dt_i,
        "candidates_per_image": float(keep[3].sum().item()) / B,
        "what": "the same step with the instance candidates (StixelsKernels.cu:926-942) and their "
                "size-filtered DBSCAN (Stixels.cu:613) for every frame: two more launches per batch"}
    del keep, ibs
    core.close()

    # ---- pairwise: ONE column group instead of the default three (IS_PW_GROUPS=1: what a caller that pipelines an
    # RCCL gather beside the next step's compute uses -- there the groups cost 5 % --, and the setting under which the
    # per-kernel durations of a profile are not those of overlapping launches)
    if wl.cfg.pairwise:
        core = wl.make_core(env={"IS_PW_GROUPS": "1"})
        dt_g = wl.time_steps(core, 3)
        out["column_groups_1"] = {"images_per_s": B / dt_g, "steps": 3,
                                  "what": "IS_PW_GROUPS=1: the phase-1 / phase-2 chain of all columns on one stream "
                                          "(default: three column groups on three streams of the context, the "
                                          "latency-bound phase 2 of one group beside the launches of the others)"}
        core.close()

    # ---- unary: the object-LUT units back in the prepare launch (IS_LUT_FUSED=0; by default they run as workgroups
    # of the DP launch in calls of >= 2048 columns), and the repair path of the fused form forced (IS_LUT_FUSED=2)
    if not wl.cfg.pairwise:
        for knob, key, what in (("0", "lut_fused_0", "IS_LUT_FUSED=0: the object-LUT units in the prepare launch (k_prepare_fused), "
                                 "as in every round before 5; default: inside the k_dp_unary_fast launch"),
                                ("2", "lut_fused_repair_forced", "IS_LUT_FUSED=2: the fused launch publishes a wrong XCC id, every DP "
                                 "workgroup distrusts the hand-over and the repair launches (ordinary LUT kernel + ordinary DP "
                                 "launch) redo the call: the price of the safety net when it fires")):
            core = wl.make_core(env={"IS_LUT_FUSED": knob})
            dt_f = wl.time_steps(core, 5)
            out[key] = {"images_per_s": B / dt_f, "steps": 5, "verify": wl.verify(wl.d_sections, images=[0, B - 1]),
                        "what": what}
            core.close()

    # ---- two / three batches in flight: one context and one stream each, batches alternate (what a
    # double-buffered caller does; the kernels of one batch fill the launch gaps and tails of another)
    out["in_flight"] = measure_in_flight(wl, (2, 3))

    # ---- invalid-disparity value 0 with 5 % holes: the HAS_INVALID kernel variants (the mode the
    # reference's own CLI hard-codes, apps/run_cityscapes.cu:188), both models
    wl.free()
    torch.cuda.empty_cache()
    other = OTHER_PRESET.get(args.preset)
    inv0 = {}
    for preset in (args.preset, other):
        if not preset:
            continue
        wi = Workload(preset, H, W, D, B, min(wl.distinct, 8), dev, local_rank, invalid_disparity=0.0)
        core = wi.make_core()
        core.set_kernel_timing(True)
        dti = wi.time_steps(core, 3)
        inv0[preset] = {"images_per_s": B / dti, "steps": 3, "kernel_ms": core.kernel_times_ms(),
                        "verify": wi.verify(wi.d_sections, images=[0, B - 1]),
                        "what": "invalid_disparity = 0, 5 % of the pixels invalid (HAS_INVALID kernels)"}
        core.close(); wi.free(); del wi
        torch.cuda.empty_cache()
    out["invalid_disparity_0"] = inv0[args.preset]

    # ---- other input families (SURVEY.md 8d generator + six harder / different ones)
    out["families"] = measure_families(args.preset, H, W, D, B, dev, local_rank)

    # ---- every column in the generic encoding: one negative class value per column
    wg = Workload(args.preset, H, W, D, B, 4, dev, local_rank)
    wg.d_seg[:, :, 0, 0] = -1
    for f in wg.frames:                  # (the host copies the oracle check reads)
        f.segmentation[:, 0, 0] = -1
    core = wg.make_core()
    out["generic_encoding"] = {"images_per_s": B / wg.time_steps(core, 2), "steps": 2,
                               "verify": wg.verify(wg.d_sections, images=[B - 1]),
                               "what": "one negative class value per column: int32 / int64 records, "
                                       "IEEE division, no pruning (the hostile-input path)"}
    core.close(); wg.free(); del wg
    torch.cuda.empty_cache()

    # ---- the OTHER model at the same shape and batch (BASELINE configs[3]'s per-GPU share when
    # the headline is the unary preset): own roofline, pruning statistics, pruning off, verify
    if other:
        wo = Workload(other, H, W, D, B, args.distinct, dev, local_rank)
        key = ("pairwise" if wo.cfg.pairwise else "unary") + f"_batch{B}"
        out[key] = measure_mode(wo, steps=3, with_single=True)
        out[key]["preset"] = other
        out[key]["distinct_frames"] = wo.distinct
        out[key]["invalid_disparity_0"] = inv0[other]
        wo.free(); del wo
        torch.cuda.empty_cache()
        out[key]["families"] = measure_families(other, H, W, D, B, dev, local_rank)
        out[key]["value_floor_families"] = floor_over_families(out[key]["images_per_s"], out[key]["families"])

    # ---- BASELINE configs[4]: 1024x4096 frames, 256 disparity bins, both models
    c5 = {}
    for preset in (args.preset, other):
        if not preset:
            continue
        w5 = Workload(preset, 1024, 4096, 256, 32, 4, dev, local_rank, seed0=91)
        m = measure_mode(w5, steps=2)
        m["preset"] = preset
        c5["pairwise" if w5.cfg.pairwise else "unary"] = m
        w5.free(); del w5
        torch.cuda.empty_cache()
    c5["what"] = ("BASELINE configs[4]: 32 frames of 1024x4096, 256 disparity bins per call (512 stixel "
                  "columns per frame, 2.2 GB of lutT per frame; the windowed D = 256 kernels), first and last "
                  "frame of the timed output against the oracle; parity at this instantiation: "
                  "tests/test_parity_gpu.py test_config5_batch_of_8_windowed_kernels_as_timed")
    out["c5_1024x4096x256"] = c5

    # ---- the reference's OWN operating point: the 784x1792 crop, 128 disparities, invalid_disparity = 0
    # (tests/run_test.sh:84, apps/run_cityscapes.cu:129-134 and :188, apps/stixels_node.cu:162-176): batch 64
    # through the C ABI and one frame per call through Stixels::Compute, both models, verified
    rs = {}
    for preset in (args.preset, other):
        if not preset:
            continue
        wr = Workload(preset, 784, 1792, 128, B, min(B, 16), dev, local_rank, seed0=211, invalid_disparity=0.0)
        m = measure_mode(wr, steps=3, with_pruning_off=False)
        m["preset"] = preset
        st = host.Stixels()
        st.SetConfig(wr.cfg)
        st.SetDevice(local_rank)
        st.Initialize()
        f = wr.frames[0]
        st.SetDisparityImage(f.disparity)
        st.SetSegmentation(f.segmentation)
        st.SetRoadParameters(f.vhor_image, f.camera_tilt, f.camera_height, f.alpha_ground)
        data = st.Compute(wr.cfg.pairwise)
        t1 = st.time_compute(wr.cfg.pairwise, 100 if wr.cfg.pairwise else 200, True)
        st.close()
        one = Workload(preset, 784, 1792, 128, 1, 1, dev, local_rank, frames=[f], invalid_disparity=0.0)
        m["stixels_compute_one_frame_per_call"] = {
            "images_per_s": 1.0 / t1, "ms_per_frame": t1 * 1e3,
            "stixels_in_frame": int(sum((c["type"] != -1).argmin() for c in data.sections)),
            "verify": one.verify(torch.from_numpy(data.sections.view(np.int32).reshape(1, one.C, one.S, 8))),
            "what": "Stixels::Compute + GetInstanceStixels per frame (host ground model, device clustering, "
                    "D2H of the sections), timed inside the C++ library"}
        one.free(); del one
        wr.free(); del wr
        torch.cuda.empty_cache()
        # the same shape and mode on the cityscapes_like family: sky and occlusion bands invalid as REGIONS,
        # ~3400 / 2500 stixels per frame (the reference's pins on real data: 2278 / 1421)
        wc = Workload(preset, 784, 1792, 128, B, min(B, 8), dev, local_rank, seed0=311, family="cityscapes_like",
                      invalid_disparity=0.0)
        core = wc.make_core()
        dtc = wc.time_steps(core, 3)
        m["cityscapes_like"] = {"images_per_s": B / dtc, "verify": wc.verify(wc.d_sections, images=[B - 1]),
                                "evaluated_frac": wc.prune_stats(core)["evaluated_frac"]}
        core.close(); wc.free(); del wc
        torch.cuda.empty_cache()
        rs["pairwise" if m["preset_pairwise"] else "unary"] = m
    rs["what"] = ("the only shape / mode the reference itself launches: 784x1792 crop (224 stixel columns, 12.25 "
                  "tiles), 128 disparities, invalid_disparity = 0 with 5 % holes; 16 distinct frames per batch of "
                  "64.  Context, not a baseline for this metric: BASELINE.md quotes ~19 fps for the reference's "
                  "WHOLE pipeline (CNN included) at this shape on a Titan V")
    out["ref_shape_784x1792"] = rs

    # ---- BASELINE configs[0]: one 512x1024 frame, 64 bins, disparity only -- CPU path + GPU
    c1 = {}
    for preset in ("disparity_only_unary", "disparity_only_pairwise"):
        w1 = Workload(preset, 512, 1024, 64, 1, 1, dev, local_rank, seed0=12)
        core = w1.make_core()
        dt1 = w1.time_steps(core, 50)
        v = w1.verify(w1.d_sections, images=[0])
        core.close()
        c1[preset] = {"gpu_images_per_s": 1.0 / dt1, "gpu_ms_per_frame": dt1 * 1e3, "verify_ok": v["ok"],
                      "cpu": cpu_baseline(w1.cfg, w1.frames[0], 1.5, single_thread=True)}
        w1.free(); del w1
    c1["what"] = ("BASELINE configs[0]: single 512x1024 frame, 64 disparity bins, segmentation weight 0; "
                  "cpu = the oracle (kind port) on the GPU box's host cores, gpu = one frame per call")
    out["c1_cpu_512x1024x64"] = c1

    # ---- BASELINE configs[1] through the C++ host class: Stixels::Compute() incl. the device-side
    # clustering, the D2H copy of the sections and one synchronisation per frame
    wh = Workload(args.preset, H, W, D, 8, args.distinct, dev, local_rank)
    st = host.Stixels()
    st.SetConfig(wh.cfg)
    st.SetDevice(local_rank)
    st.Initialize(max_batch=8)
    f = wh.frames[0]
    st.SetDisparityImage(f.disparity)
    st.SetSegmentation(f.segmentation)
    st.SetRoadParameters(f.vhor_image, f.camera_tilt, f.camera_height, f.alpha_ground)
    t_plain = st.time_compute(wh.cfg.pairwise, 200, False)
    t_inst = st.time_compute(wh.cfg.pairwise, 200, True)
    road = [(wh.frames[i].vhor_image, wh.frames[i].camera_tilt, wh.frames[i].camera_height,
             wh.frames[i].alpha_ground) for i in wh.pick]
    t_b = st.time_compute_batch(wh.cfg.pairwise, wh.d_big.data_ptr(), wh.d_seg.data_ptr(), road, 5, False)
    t_bi = st.time_compute_batch(wh.cfg.pairwise, wh.d_big.data_ptr(), wh.d_seg.data_ptr(), road, 5, True)
    st.close()
    wh.free(); del wh
    out["stixels_compute_host_class"] = {
        "images_per_s": 1.0 / t_plain, "ms_per_frame": t_plain * 1e3,
        "images_per_s_with_GetInstanceStixels": 1.0 / t_inst,
        "compute_batch8_images_per_s": 8 / t_b,
        "compute_batch8_with_instance_mappings_images_per_s": 8 / t_bi,
        "what": "one frame per Stixels::Compute() call, timed inside the C++ library "
                "(ish_time_compute): ground model on the host, JoinColumns + DP + back-trace + "
                "instance candidates + clustering on the device, sections copied to the host; "
                "compute_batch8: Stixels::ComputeBatch of 8 frames incl. the D2H copy of all "
                "sections (and of the per-frame instance mappings)"}
    return out


COMPACT_LIMIT = 4096   # bytes of the stdout line (the driver keeps a bounded tail of stdout)


def _r(x, nd=3):
    return round(x, nd) if isinstance(x, float) else x


def compact_line(out):
    """The ONE line that goes to stdout: the contract keys + the figures the headline has to be read
    with, always shorter than COMPACT_LIMIT bytes.  Everything else is in bench_full.json."""
    roof = out.get("roofline") or {}
    cpu = out.get("cpu_baseline")
    cfg = out.get("config") or {}
    c = {k: out.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
                                 "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    c["value"], c["ms_per_step"] = _r(c["value"], 1), _r(c["ms_per_step"], 4)
    c["config"] = {k: cfg.get(k) for k in ("workload", "batch_per_gpu", "rows", "cols", "max_dis", "preset",
                                           "family", "parallelism") if k in cfg}
    c["roofline"] = {k: _r(roof.get(k), 5) for k in (
        "bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms", "kernel_ms_how",
        "algorithmic_bytes_per_image", "traffic_source", "from_committed_profile", "kernel_ms_profile",
        "frac_profile", "measured_hbm_frac", "note") if k in roof}
    if cpu:
        c["cpu_baseline"] = {k: _r(cpu.get(k), 4) for k in ("value", "unit", "cores", "kind", "sample")}
    else:
        c["cpu_baseline"] = None
    c["verify_all_ok"] = out.get("verify_all_ok")
    c["verified_count"] = len(out.get("verified") or [])
    c["value_pruning_off"] = _r(out.get("value_pruning_off"), 1)
    fl = out.get("value_floor_families")
    c["value_floor_families"] = ({"images_per_s": _r(fl["images_per_s"], 1), "family": fl["family"],
                                  "families_measured": len(fl["families_measured"])} if fl else None)
    c["lut_fused_repaired"] = out.get("lut_fused_repaired")
    for k in ("value_incl_d2h", "value_incl_instances", "value_incl_h2d_d2h"):
        if out.get(k) is not None:
            c[k] = _r(out[k], 1)
    if out.get("value_incl_d2h_compact"):
        c["value_incl_d2h_compact"] = _r(out["value_incl_d2h_compact"]["images_per_s"], 1)
    if out.get("single_frame"):
        c["single_frame_ms"] = _r(out["single_frame"]["ms_per_frame"], 4)
    if out.get("prune"):
        c["evaluated_frac"] = _r(out["prune"]["evaluated_frac"], 4)
    if out.get("valu") and out["valu"].get("issue_frac") is not None:
        c["valu_issue_frac"] = {"value": _r(out["valu"]["issue_frac"], 3), "from_committed_profile": True}
    if out.get("kernel_ms"):
        c["kernel_ms"] = {k: _r(v, 4) for k, v in out["kernel_ms"].items()}
    if out.get("gather"):
        g = out["gather"]
        c["gather"] = {k: _r(g[k], 4) for k in ("kind", "exposed_ms_per_step", "ms_per_step_without_gather",
                                                "ratio_vs_fixed") if k in g}
        if out.get("verify") and "rccl_gather" in out["verify"]:
            c["gather"]["rank0_copy_equals_local"] = out["verify"]["rccl_gather"].get("rank0_copy_equals_local")
    if out.get("other_model"):
        c["other_model"] = out["other_model"]
    c["full"] = out.get("full_json")
    line = json.dumps(c, separators=(",", ":"))
    # never longer than the limit: drop the optional fields, longest first, then the free-text ones
    for k in ("other_model", "gather", "kernel_ms", "valu_issue_frac"):
        if len(line) >= COMPACT_LIMIT and k in c:
            del c[k]
            line = json.dumps(c, separators=(",", ":"))
    def clip(x, n):   # free-text fields are what can grow: clip every string, harder until the line fits
        if isinstance(x, dict):
            return {k: clip(v, n) for k, v in x.items()}
        return x[:n] if isinstance(x, str) and len(x) > n else x
    for n in (240, 120, 60, 24):
        if len(line) < COMPACT_LIMIT:
            break
        line = json.dumps(clip(c, n), separators=(",", ":"))
    assert len(line) < COMPACT_LIMIT, len(line)
    return line


def emit(out, path=None):
    """Complete object -> bench_full.json (+ gpurun_out/), compact line -> stdout (the last line)."""
    path = path or os.path.join(ROOT, "bench_full.json")
    written = []
    targets = [path]
    if os.path.isdir(os.path.join(ROOT, "gpurun_out")):
        targets.append(os.path.join(ROOT, "gpurun_out", os.path.basename(path)))
    for t in targets:
        try:
            with open(t, "w") as fh:
                json.dump(out, fh)
            written.append(os.path.relpath(t, ROOT))
        except OSError:
            pass
    out["full_json"] = written[0] if written else None
    line = compact_line(out)
    sys.stdout.write(line + "\n")
    sys.stdout.flush()
    return line


def launch_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start torch.distributed.run as a CHILD process
    (this process has not imported torch nor touched HIP; nothing is exec'ed), pass its output through
    and make rank 0's JSON line the last line of OUR stdout.  Returns the child's exit code.
    IS_BENCH_CHILD_CMD (a JSON list) replaces the child command: tests/test_bench_line.py."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    if os.environ.get("IS_BENCH_CHILD_CMD"):
        cmd = json.loads(os.environ["IS_BENCH_CHILD_CMD"])
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.pop("IS_BENCH_CHILD_CMD", None)
    child = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env, cwd=ROOT)
    result, last = None, None
    for ln in child.stdout:
        ln = ln.rstrip("\n")
        print(ln, flush=True)
        last = ln
        if ln.startswith("{") and '"metric"' in ln:
            try:
                json.loads(ln)
                result = ln
            except ValueError:
                pass
    rc = child.wait()
    if result is not None and last != result:
        print(result, flush=True)
    return rc


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE is {world}")

    import torch
    import torch.distributed as dist
    from instance_stixels_amd import synthetic
    from instance_stixels_amd.parallel import PipelinedGather, PipelinedCompactGather

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    Workload.side_stream = args.side_stream
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

    # ---- host side (C++ Stixels class) + inputs resident in HBM before the timed region
    extra = {} if args.invalid_disparity is None else {"invalid_disparity": args.invalid_disparity}
    wl = Workload(args.preset, args.rows, args.cols, args.max_dis, args.batch, args.distinct, dev,
                  local_rank, seed0=17 + 101 * rank, family=args.family, **extra)
    cfg, B, H, W, C, D, S = wl.cfg, wl.B, wl.H, wl.W, wl.C, wl.D, wl.S
    # (beside the pipelined RCCL gather the pairwise DP runs as ONE column group: measured, is_device.h)
    core = wl.make_core(env={"IS_PW_GROUPS": "1"} if (use_dist and not args.no_gather and cfg.pairwise) else None)
    core.set_kernel_timing(True)
    # N > 1: the stixel outputs of every step are gathered on rank 0 (RCCL over xGMI); the gather
    # of step k overlaps the compute of step k+1 (double-buffered outputs)
    pipe = None
    if use_dist and not args.no_gather:
        if args.gather == "compact":
            pipe = PipelinedCompactGather(wl.d_sections, core, depth=4, dst=0)
        else:
            pipe = PipelinedGather(wl.d_sections, depth=2, dst=0)

    def step():
        out = pipe.next_buffer() if pipe is not None else wl.d_sections
        wl.step(core, out=out)
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
    repairs0 = core.lut_fused_repairs()   # (sticky count of repaired calls of this context)

    local_blocks = []   # this rank's own clock of every block (before the max over ranks)
    kt_blocks = []      # HIP-event kernel times of the last step of every timed block

    def timed_block(fn=None):
        """EXACTLY args.steps steps between barrier + synchronize on both sides; max over ranks."""
        fn = fn or step
        t0 = time.perf_counter()
        for _ in range(args.steps):
            fn()
        barrier()
        d = time.perf_counter() - t0
        local_blocks.append(d)
        if use_dist:
            t = torch.tensor([d], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            d = float(t.item())
        return d

    # the K-step block is repeated until --min-seconds have been measured (every rank sees the
    # same max-reduced times, so all ranks stop together) and the MEDIAN block is reported: a
    # 0.1 s measurement is over before a power / utilisation sampler sees the GPU busy
    blocks = []
    while not blocks or (sum(blocks) < args.min_seconds and len(blocks) < 200):
        blocks.append(timed_block())
        kt_blocks.append(core.kernel_times_ms())   # (outside the block's clock: the events of its last step)
    dt = float(np.median(blocks))
    # per-kernel durations, HIP events on the launch stream: the mean over the last steps of all timed blocks
    kt = {k: float(np.mean([b[k] for b in kt_blocks])) for k in kt_blocks[0]}
    repaired = core.lut_fused_repairs() - repairs0
    timed_out = wl.d_sections if pipe is None else pipe.last_local()
    gather_stats = pipe.stats() if pipe is not None else None

    # N > 1 (or --force-dist): what every rank measured by itself, and the same K steps WITHOUT the
    # gather (outputs stay on the rank) -- the difference is what the gather costs the step
    # ("exposed": not hidden behind the compute); both outside `value`
    per_rank = None
    if use_dist:
        mine = {"rank": rank, "ms_per_step_own_clock": float(np.median(local_blocks)) / args.steps * 1e3,
                "dp_ms": kt["dp_ms"], "prepare_ms": kt["prepare_ms"], "lut_fused_repaired": repaired}
        if pipe is not None:
            n_before = len(local_blocks)
            d_plain = timed_block(lambda: wl.step(core))
            mine["ms_per_step_without_gather_own_clock"] = local_blocks[n_before] / args.steps * 1e3
            gather_stats["ms_per_step_without_gather"] = d_plain / args.steps * 1e3
            gather_stats["exposed_ms_per_step"] = (dt - d_plain) / args.steps * 1e3
            local_blocks.pop()
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        per_rank = gathered
        repaired = sum(g["lut_fused_repaired"] for g in gathered)

    # ---- verify (default): frames of the TIMED output (the batch geometry the value is measured on)
    verify = None
    if not args.no_verify and rank == 0:
        verify = wl.verify(timed_out, fatal=True)
        if pipe is not None:  # what the RCCL gather delivered to rank 0 is what the ranks computed
            verify["rccl_gather"] = pipe.check_last(S)
            if not verify["rccl_gather"]["rank0_copy_equals_local"]:
                raise SystemExit("bench.py verify: gathered copy of rank 0 differs from its output")

    # ---- what the branch-and-bound evaluated (device counters, separate untimed step)
    prune = wl.prune_stats(core) if (rank == 0 and not args.no_prune_stats) else None

    # BASELINE.json configs[1] (ONE 1024x2048 frame per call) next to the batched headline value:
    # the same entry points with n_images = 1, i.e. the latency a per-frame caller sees
    single = None
    if world == 1 and not args.no_single:
        for _ in range(10):
            wl.step(core, n=1)
        torch.cuda.synchronize(dev)
        n1 = 100
        t1 = time.perf_counter()
        for _ in range(n1):
            wl.step(core, n=1)
        torch.cuda.synchronize(dev)
        single = (time.perf_counter() - t1) / n1

    # second figure, N = 1 only and outside the judged `value`: the same step
    # followed by the D2H copy of the Section output into pinned host memory (what
    # Stixels::Compute does, Stixels.cu:629-633)
    d2h_value = d2h_compact = None
    if world == 1 and not args.no_d2h:
        h_sections = torch.empty(wl.d_sections.shape, dtype=wl.d_sections.dtype, pin_memory=True)
        for _ in range(2):
            wl.step(core); h_sections.copy_(wl.d_sections, non_blocking=True)
        torch.cuda.synchronize(dev)
        k = max(2, min(args.steps, 5))
        t1 = time.perf_counter()
        for _ in range(k):
            wl.step(core); h_sections.copy_(wl.d_sections, non_blocking=True)
        torch.cuda.synchronize(dev)
        d2h_value = B * k / (time.perf_counter() - t1)
        del h_sections
        # ... and the same with the output COMPACTED first (is_pack_sections: per-column offsets + the used sections
        # only, what Stixels::ComputeBatch copies): two pinned copies per step, the second sized by the first
        ncol = B * C
        counts = torch.empty(ncol, dtype=torch.int32, device=dev)
        offsets = torch.empty(ncol + 1, dtype=torch.int32, device=dev)
        packed = torch.empty((ncol * (S - 1), 8), dtype=torch.int32, device=dev)
        h_off = torch.empty(ncol + 1, dtype=torch.int32, pin_memory=True)
        h_packed = torch.empty((ncol * (S - 1), 8), dtype=torch.int32, pin_memory=True)
        from instance_stixels_amd import core as core_mod

        def step_compact():
            wl.step(core)
            core_mod.pack_sections_ptr(wl.d_sections.data_ptr(), ncol, S, counts.data_ptr(), offsets.data_ptr(),
                                       packed.data_ptr(), wl.stream)
            h_off.copy_(offsets, non_blocking=True)
            torch.cuda.synchronize(dev)
            total = int(h_off[-1])
            h_packed[:total].copy_(packed[:total], non_blocking=True)
            return total
        for _ in range(2):
            step_compact()
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        for _ in range(k):
            total = step_compact()
        torch.cuda.synchronize(dev)
        d2h_compact = {"images_per_s": B * k / (time.perf_counter() - t1), "bytes_per_step": 4 * (ncol + 1) + 32 * total,
                       "fixed_stride_bytes_per_step": int(wl.d_sections.numel()) * 4}
        del counts, offsets, packed, h_off, h_packed

    # PCIe-inclusive figure for DESIGN.md (--pcie): inputs come from pinned host memory every step
    # and the sections go back, all on the compute stream, nothing overlapped
    pcie_value = None
    if world == 1 and args.pcie:
        h_big = torch.empty(wl.d_big.shape, dtype=wl.d_big.dtype, pin_memory=True).copy_(wl.d_big)
        h_seg = torch.empty(wl.d_seg.shape, dtype=wl.d_seg.dtype, pin_memory=True).copy_(wl.d_seg)
        h_out = torch.empty(wl.d_sections.shape, dtype=wl.d_sections.dtype, pin_memory=True)
        def step_pcie():
            wl.d_big.copy_(h_big, non_blocking=True); wl.d_seg.copy_(h_seg, non_blocking=True)
            wl.step(core); h_out.copy_(wl.d_sections, non_blocking=True)
        step_pcie(); torch.cuda.synchronize(dev)
        k = 3
        t1 = time.perf_counter()
        for _ in range(k):
            step_pcie()
        torch.cuda.synchronize(dev)
        pcie_value = B * k / (time.perf_counter() - t1)
        del h_big, h_seg, h_out

    cpu = None
    if not args.no_cpu_baseline and world == 1 and rank == 0:
        cpu = cpu_baseline(cfg, wl.frames[0], args.cpu_seconds)

    core.close()
    # ---- the two floors the headline is read with (N = 1, a few steps each, outside `value`): pruning
    # off (data-independent) and the slowest input family any full sweep has found (cityscapes_like)
    pruning_off = floor = None
    quick = world == 1 and rank == 0 and not args.no_variants and not args.full
    if quick:
        c0 = wl.make_core(env={"IS_NO_PRUNE": "1"})
        pruning_off = B / wl.time_steps(c0, 3)
        c0.close()
        fam = "cityscapes_like"
        if args.family != fam:
            wf = Workload(args.preset, H, W, D, B, min(B, 8), dev, local_rank, family=fam, seed0=311, **extra)
            cf = wf.make_core()
            vf = B / wf.time_steps(cf, 3)
            wf.verify(wf.d_sections, images=[B - 1])
            cf.close(); wf.free(); del wf
            head = B * args.steps / dt
            floor = {"images_per_s": min(vf, head), "family": fam if vf < head else args.family,
                     "families_measured": sorted({fam, args.family})}

    # ---- --full (N = 1): other kernel variants / families / models / shapes
    variants = None
    spread = None
    if world == 1 and args.full:
        variants = measure_variants(args, wl, dev, local_rank)   # (frees wl's device buffers)
        if args.spread_batches > 0:
            other = OTHER_PRESET.get(args.preset)
            floors = {floor_over_families(float("inf"), variants["families"])["family"]}
            okey = [k for k in variants if k.endswith(f"_batch{B}")]
            if okey:
                floors.add(floor_over_families(float("inf"), variants[okey[0]]["families"])["family"])
            floors.discard("scene")
            spread = measure_spread([p for p in (args.preset, other) if p], H, W, D, B, dev, local_rank,
                                    ["scene"] + sorted(floors), args.spread_batches)

    if rank == 0:
        images = B * world * args.steps
        value = images / dt
        pairs_img = synthetic.pair_evaluations_per_image(cfg)
        dp_s = kt["dp_ms"] * 1e-3
        traffic, tfile = committed_traffic(cfg, B, H, W, D)
        roof = wl.roofline(kt["dp_ms"], dp_kernel_name(cfg), traffic, tfile)
        roof["kernel_ms_how"] = (f"HIP events on the launch stream around the DP launches, mean over the last step of "
                                 f"each of the {len(kt_blocks)} timed blocks")
        prof, pname = committed_profile(cfg, B, H, W, D)
        if prof and prof.get("rocprof_kernel_ms"):   # the rocprofv3 --kernel-trace average of the same launches
            roof["kernel_ms_profile"] = prof["rocprof_kernel_ms"]
            roof["frac_profile"] = (roof["algorithmic_bytes_per_image"] * B / (prof["rocprof_kernel_ms"] * 1e-3)
                                    / 1e9 / HBM_PEAK_GBS)
        roof["from_committed_profile"] = traffic is not None   # (of `traffic`: PMC counters cannot be read inside the timed run)
        roof["note"] = ("VALU-issue / latency bound, not HBM bound (DESIGN.md); traffic = 2*FETCH_SIZE + WRITE_SIZE "
                        "of the DP kernels per step, NOT measured in this run: copied from the committed "
                        "rocprofv3 --pmc passes (traffic_source)" if traffic is not None else
                        "VALU-issue / latency bound, not HBM bound (DESIGN.md); no committed PMC pass for this shape")
        out = {
            "metric": "images/s on 1024x2048x128-disp column DP",
            "value": value, "unit": "images/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32+i32", "data": "synthetic",
            "config": {"workload": f"C2/C3: {B} frames/GPU of {H}x{W}, {D} disparity bins, "
                                   f"19 classes + 2 offset channels, preset {args.preset} "
                                   f"({'pairwise' if cfg.pairwise else 'unary'}), input family "
                                   f"{args.family}, JoinColumns + prepare + DP + back-trace, "
                                   "device-resident in/out",
                       "batch_per_gpu": B, "rows": H, "cols": W, "max_dis": D,
                       "preset": args.preset, "family": args.family,
                       "parallelism": (f"batch shards x{world}, RCCL gather ({args.gather}) of the "
                                       "sections to rank 0, overlapped with the next step")
                                      if world > 1 else "single GPU"},
            "roofline": roof,
            "valu": {**(committed_valu(cfg, B, H, W, D) or {"issue_frac": None}),
                     "from_committed_profile": True,
                     "pair_evals_per_s": (pairs_img * B * prune["evaluated_frac"] / dp_s) if prune else None,
                     "pair_evals_per_s_nominal": pairs_img * B / dp_s,
                     "pair_evals_per_image_nominal": pairs_img,
                     "lane_ops_peak_per_s": VALU_PEAK_LANEOPS,
                     "note": "pair_evals_per_s counts the pairs the branch-and-bound actually "
                             "evaluated (prune.evaluated_frac), _nominal all C*H*(H+1)/2"},
            "prune": prune,
            "kernel_ms": kt,
            "lut_fused_repaired": repaired,
        }
        if d2h_value is not None:
            out["value_incl_d2h"] = d2h_value
            out["value_incl_d2h_compact"] = d2h_compact
        if pcie_value is not None:
            out["value_incl_h2d_d2h"] = pcie_value
        if single is not None:
            out["single_frame"] = {"workload": "BASELINE configs[1]: one frame per call (batch 1), "
                                               "device-resident in/out",
                                   "images_per_s": 1.0 / single, "ms_per_frame": single * 1e3}
        out["timed_blocks"] = {"count": len(blocks), "steps_per_block": args.steps,
                               "seconds": [round(x, 5) for x in blocks], "reported": "median",
                               "kernel_ms_per_block": kt_blocks}
        if pipe is not None:
            out["gather"] = gather_stats
        if per_rank is not None:
            out["per_rank"] = per_rank
        if verify is not None:
            out["verify"] = verify
        if spread is not None:
            out["value_spread"] = spread
        if pruning_off is not None:
            out["value_pruning_off"] = pruning_off
        if floor is not None:
            out["value_floor_families"] = floor
        if variants is not None:
            out["variants"] = variants
            out["value_incl_instances"] = variants["with_instances"]["images_per_s"]
            # how far the headline can fall on inputs unlike the generator's: the slowest of the seven
            # input families, and the data-independent floor (pruning off), side by side with `value`
            out["value_floor_families"] = floor_over_families(value, variants["families"])
            out["value_pruning_off"] = variants["pruning_off"]["images_per_s"]
            okey = [k for k in variants if k.endswith(f"_batch{B}")]
            if okey:   # the OTHER model at the same batch, in the compact line too
                o = variants[okey[0]]
                out["other_model"] = {"preset": o["preset"], "images_per_s": _r(o["images_per_s"], 1),
                                      "dp_ms": _r(o["dp_ms"], 3),
                                      "pruning_off": _r(o.get("pruning_off", {}).get("images_per_s"), 1),
                                      "single_frame_ms": _r(o.get("single_frame", {}).get("ms_per_frame"), 4)}
        if cpu is not None:
            out["cpu_baseline"] = cpu
        out["distinct_frames"] = wl.distinct
        out["verify_all_ok"] = bool(VERIFIED) and all(ok for _, ok in VERIFIED)
        out["verified"] = [{"what": w, "ok": ok} for w, ok in VERIFIED]
    else:
        out = None

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
        emit(out, args.out)


if __name__ == "__main__":
    main()
