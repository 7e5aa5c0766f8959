"""ctypes binding of the HIP column-DP core (include/instance_stixels_core.h).

This is the product path: it loads instance_stixels_amd/lib/libis_core.so and fails loudly if the
library is missing -- there is no CPU fallback.  PyTorch is used only as the owner of device
memory / streams (plumbing); every compute call goes through the C ABI.
"""
import ctypes
import os

import numpy as np

from .config import StixelParams, SECTION_DTYPE, INSTANCE_CLASSES

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("IS_CORE_LIB", os.path.join(_HERE, "lib", "libis_core.so"))
_LIB = None
EVAL_COUNTERS = 200   # IS_EVAL_COUNTERS of include/instance_stixels_core.h (checked by tests/test_host_and_abi.py)

EXPORTS = [
    "is_ctx_create", "is_ctx_destroy", "is_join_columns", "is_compute", "is_device_malloc",
    "is_device_free", "is_memcpy_h2d", "is_memcpy_d2h", "is_memcpy2d_d2h", "is_memset", "is_stream_synchronize",
    "is_device_synchronize", "is_last_error", "is_version", "is_set_kernel_timing",
    "is_get_kernel_times_ms", "is_scratch_bytes", "is_flip_and_pad", "is_road_vdisparity",
    "is_cluster_instances", "is_host_malloc", "is_host_free", "is_get_device", "is_set_device",
    "is_ctx_device", "is_set_eval_counters", "is_get_eval_counters",
    "is_pack_sections", "is_unpack_sections", "is_stream_create", "is_stream_destroy",
    "is_debug_read_object_lut", "is_debug_read_block_summaries", "is_debug_lut_fused_state",
    "is_lut_fused_repairs",
    "is_comm_unique_id", "is_comm_init_rank", "is_comm_destroy", "is_comm_rank", "is_gather_i32",
    "is_gather_sections",
]


class InstanceBuffers(ctypes.Structure):
    _fields_ = [("d_centerofmass", ctypes.c_void_p), ("d_indices", ctypes.c_void_p),
                ("d_core_candidates", ctypes.c_void_p), ("d_instances_per_class", ctypes.c_void_p),
                ("d_labels", ctypes.c_void_p), ("d_packed", ctypes.c_void_p)]


class CoreError(RuntimeError):
    pass


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise CoreError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback.")
        # PyTorch-ROCm wheels bundle their own libamdhip64.so (SONAME libamdhip64.so.7).  Two HIP
        # runtimes in one process do not work ("No HIP GPUs are available"), so when torch is
        # importable it must be loaded FIRST; the dynamic linker then resolves this library's
        # libamdhip64.so.7 dependency to the already loaded copy.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = ctypes.CDLL(LIB_PATH)
        vp, ci, cf = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
        L.is_ctx_create.argtypes = [ctypes.POINTER(StixelParams), vp, vp, ci, ci,
                                    ctypes.POINTER(vp)]
        L.is_ctx_destroy.argtypes = [vp]
        L.is_join_columns.argtypes = [vp, vp, ci, ci, vp, ci, vp]
        L.is_compute.argtypes = [vp, vp, vp, vp, vp, vp, vp, ci, ci, vp, vp, vp, vp, vp]
        L.is_device_malloc.argtypes = [ctypes.POINTER(vp), ctypes.c_size_t]
        L.is_device_free.argtypes = [vp]
        L.is_memcpy_h2d.argtypes = [vp, vp, ctypes.c_size_t, vp]
        L.is_memcpy_d2h.argtypes = [vp, vp, ctypes.c_size_t, vp]
        L.is_memcpy2d_d2h.argtypes = [vp, ctypes.c_size_t, vp, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t, vp]
        L.is_memset.argtypes = [vp, ci, ctypes.c_size_t, vp]
        L.is_stream_synchronize.argtypes = [vp]
        L.is_last_error.restype = ctypes.c_char_p
        L.is_version.restype = ctypes.c_char_p
        L.is_set_kernel_timing.argtypes = [vp, ci]
        L.is_get_kernel_times_ms.argtypes = [vp, ctypes.POINTER(cf), ctypes.POINTER(cf),
                                             ctypes.POINTER(cf)]
        L.is_flip_and_pad.argtypes = [vp, vp, ci, ci, ci, ci, ci, vp]
        L.is_road_vdisparity.argtypes = [vp, ci, ci, ci, cf, vp, vp, vp, vp]
        L.is_cluster_instances.argtypes = [vp, ctypes.POINTER(InstanceBuffers), vp]
        L.is_host_malloc.argtypes = [ctypes.POINTER(vp), ctypes.c_size_t]
        L.is_host_free.argtypes = [vp]
        L.is_get_device.argtypes = [ctypes.POINTER(ci)]
        L.is_set_device.argtypes = [ci]
        L.is_ctx_device.argtypes = [vp]
        L.is_set_eval_counters.argtypes = [vp, ci]
        L.is_get_eval_counters.argtypes = [vp, vp, ci]
        L.is_pack_sections.argtypes = [vp, ci, ci, vp, vp, vp, vp]
        L.is_unpack_sections.argtypes = [vp, vp, vp, ci, ci, vp, vp]
        L.is_scratch_bytes.argtypes = [vp]
        L.is_scratch_bytes.restype = ctypes.c_size_t
        L.is_comm_unique_id.argtypes = [vp, ctypes.c_size_t]
        L.is_comm_init_rank.argtypes = [ctypes.POINTER(vp), ci, vp, ci]
        L.is_comm_destroy.argtypes = [vp]
        L.is_comm_rank.argtypes = [vp, ctypes.POINTER(ci), ctypes.POINTER(ci)]
        L.is_gather_i32.argtypes = [vp, ci, vp, vp, vp, vp]
        L.is_gather_sections.argtypes = [vp, ci, vp, vp, vp, vp, vp, vp, ctypes.c_size_t, vp, vp]
        L.is_debug_read_object_lut.argtypes = [vp, ci, vp]
        L.is_debug_lut_fused_state.argtypes = [vp, ctypes.POINTER(ci)]
        L.is_lut_fused_repairs.argtypes = [vp, ctypes.POINTER(ci)]
        try:   # (an experiment library built from an older tree may lack the newest test hook)
            L.is_debug_read_block_summaries.argtypes = [vp, ci, vp, ci, ctypes.POINTER(ci)]
        except AttributeError:
            pass
        _LIB = L
    return _LIB


def _check(rc, what):
    if rc != 0:
        raise CoreError(f"{what} failed (rc={rc}): {lib().is_last_error().decode()}")


def _hp(a):
    return a.ctypes.data_as(ctypes.c_void_p)


class Core:
    """One `is_ctx`: the device half of Stixels::Initialize .. Finish for a fixed configuration."""

    def __init__(self, params: StixelParams, obj_cost_lut, obj_disparity_range, max_batch=1,
                 device=0):
        self.params = StixelParams.from_buffer_copy(params)
        self.max_batch = int(max_batch)
        self.device = int(device)
        lut = np.ascontiguousarray(obj_cost_lut, np.float32)
        odr = np.ascontiguousarray(obj_disparity_range, np.float32)
        D = self.params.max_dis
        assert lut.size == D * D and odr.size == D
        self._ctx = ctypes.c_void_p()
        _check(lib().is_ctx_create(ctypes.byref(self.params), _hp(lut), _hp(odr), self.max_batch,
                                   self.device, ctypes.byref(self._ctx)), "is_ctx_create")

    def close(self):
        if self._ctx:
            lib().is_ctx_destroy(self._ctx)
            self._ctx = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def scratch_bytes(self):
        return int(lib().is_scratch_bytes(self._ctx))

    def set_kernel_timing(self, enabled=True):
        _check(lib().is_set_kernel_timing(self._ctx, int(enabled)), "is_set_kernel_timing")

    def set_eval_counters(self, enabled=True):
        """Evaluation counters of the branch-and-bound (never inside a timed region)."""
        _check(lib().is_set_eval_counters(self._ctx, int(enabled)), "is_set_eval_counters")

    def eval_counters(self):
        out = np.zeros(EVAL_COUNTERS, np.uint64)
        rc = lib().is_get_eval_counters(self._ctx, _hp(out), EVAL_COUNTERS)
        if rc != 0:   # (an experiment library built from an older tree only knows the first eight)
            _check(lib().is_get_eval_counters(self._ctx, _hp(out), 8), "is_get_eval_counters")
        return dict(unary_full=int(out[0]), unary_gs=int(out[1]), p1_full=int(out[2]), p1_gs=int(out[3]),
                    p1_window_miss=int(out[4]), unary_window_miss=int(out[5]),
                    lutf_spins=int(out[6]), lutf_unit_cycles=int(out[7]),
                    # per phase-1 launch (tile): [full, window misses, ground / sky-only]
                    p1_per_tile=[[int(out[8 + 3 * t + j]) for j in range(3)] for t in range(64)])

    def lut_fused_repaired(self):
        """1 when the last unary call ran its repair launches (test hook, see instance_stixels_core.h)."""
        out = ctypes.c_int(-1)
        _check(lib().is_debug_lut_fused_state(self._ctx, ctypes.byref(out)), "is_debug_lut_fused_state")
        return int(out.value)

    def lut_fused_repairs(self):
        """Calls of this context whose fused LUT hand-over was distrusted and repaired (sticky count)."""
        out = ctypes.c_int(-1)
        _check(lib().is_lut_fused_repairs(self._ctx, ctypes.byref(out)), "is_lut_fused_repairs")
        return int(out.value)

    def read_object_lut(self, column):
        """lutT[v][fn] of one stixel column as the last compute call left it (test hook, A4)."""
        out = np.zeros((self.params.rows + 1, self.params.max_dis), np.float32)
        _check(lib().is_debug_read_object_lut(self._ctx, int(column), _hp(out)), "is_debug_read_object_lut")
        return out

    def read_block_summaries(self, column):
        """[n_blocks][24] bound-block summaries of one column after a pairwise call (test hook)."""
        out = np.zeros(4096 * 24, np.float32)
        n = ctypes.c_int(0)
        _check(lib().is_debug_read_block_summaries(self._ctx, int(column), _hp(out), out.size, ctypes.byref(n)),
               "is_debug_read_block_summaries")
        return out[: n.value * 24].reshape(n.value, 24).copy()

    def kernel_times_ms(self):
        a, b, c = ctypes.c_float(), ctypes.c_float(), ctypes.c_float()
        _check(lib().is_get_kernel_times_ms(self._ctx, ctypes.byref(a), ctypes.byref(b),
                                            ctypes.byref(c)), "is_get_kernel_times_ms")
        return dict(prepare_ms=a.value, dp_ms=b.value, backtrace_ms=c.value)

    # ---- raw-pointer API (device pointers as ints) -------------------------------------
    def join_columns_ptr(self, d_big, full_cols, median_join, d_joined, n_images, stream=0):
        _check(lib().is_join_columns(self._ctx, d_big, int(full_cols), int(bool(median_join)),
                                     d_joined, int(n_images), stream), "is_join_columns")

    def compute_ptr(self, d_joined, d_seg, ground_function, normalization_ground,
                    inv_sigma2_ground, vhor, pairwise, n_images, d_sections, instances=None,
                    d_cost_table=None, d_index_table=None, stream=0):
        H = self.params.rows
        gf = np.ascontiguousarray(ground_function, np.float32).reshape(n_images, H)
        ng = np.ascontiguousarray(normalization_ground, np.float32).reshape(n_images, H)
        ig = np.ascontiguousarray(inv_sigma2_ground, np.float32).reshape(n_images, H)
        vh = np.ascontiguousarray(vhor, np.int32).reshape(n_images)
        inst = None
        if instances is not None:
            arr = (InstanceBuffers * n_images)(*instances)
            inst = ctypes.cast(arr, ctypes.c_void_p)
        _check(lib().is_compute(self._ctx, d_joined, d_seg, _hp(gf), _hp(ng), _hp(ig), _hp(vh),
                                int(bool(pairwise)), int(n_images), d_sections, inst,
                                d_cost_table, d_index_table, stream), "is_compute")

    # ---- torch-tensor convenience API ----------------------------------------------------
    def run(self, disparity_big=None, joined=None, segmentation=None, ground_function=None,
            normalization_ground=None, inv_sigma2_ground=None, vhor=None, pairwise=False,
            median_join=False, want_tables=False, want_instances=True):
        """Runs a batch given numpy inputs; returns numpy outputs (one sync at the end).

        disparity_big [n][H][W] or joined [n][C][H]; segmentation [n][C][CH][P2S]."""
        import torch
        p = self.params
        C, H, S = p.cols, p.rows, p.max_sections
        dev = torch.device("cuda", self.device)
        seg = torch.from_numpy(np.ascontiguousarray(segmentation, np.int32)).to(dev)
        n = seg.shape[0]
        stream = torch.cuda.current_stream(dev).cuda_stream
        if joined is None:
            big = torch.from_numpy(np.ascontiguousarray(disparity_big, np.float32)).to(dev)
            assert big.shape[0] == n and big.shape[1] == H
            d_joined = torch.empty((n, C, H), dtype=torch.float32, device=dev)
            self.join_columns_ptr(big.data_ptr(), big.shape[2], median_join, d_joined.data_ptr(), n,
                                  stream)
        else:
            d_joined = torch.from_numpy(np.ascontiguousarray(joined, np.float32)).to(dev)
        sections = torch.empty((n, C, S, 8), dtype=torch.int32, device=dev)
        cost = torch.empty((n, C, H, 3), dtype=torch.float32, device=dev) if want_tables else None
        index = torch.empty((n, C, H, 3), dtype=torch.int32, device=dev) if want_tables else None
        inst_t, inst_s = None, None
        if want_instances:
            com = torch.zeros((n, INSTANCE_CLASSES, C * S, 2), dtype=torch.float32, device=dev)
            idx = torch.zeros((n, INSTANCE_CLASSES, C * S, 2), dtype=torch.int32, device=dev)
            core = torch.zeros((n, INSTANCE_CLASSES, C * S), dtype=torch.uint8, device=dev)
            per = torch.zeros((n, INSTANCE_CLASSES), dtype=torch.int32, device=dev)
            lab = torch.full((n, INSTANCE_CLASSES, C * S), -9, dtype=torch.int32, device=dev)
            inst_t = (com, idx, core, per, lab)
            inst_s = [InstanceBuffers(com[i].data_ptr(), idx[i].data_ptr(), core[i].data_ptr(),
                                      per[i].data_ptr(), lab[i].data_ptr(), None)
                      for i in range(n)]
        self.compute_ptr(d_joined.data_ptr(), seg.data_ptr(), ground_function,
                         normalization_ground, inv_sigma2_ground, vhor, pairwise, n,
                         sections.data_ptr(), inst_s,
                         cost.data_ptr() if cost is not None else None,
                         index.data_ptr() if index is not None else None, stream)
        torch.cuda.synchronize(dev)
        out = dict(joined=d_joined.cpu().numpy(),
                   sections=sections.cpu().numpy().view(SECTION_DTYPE).reshape(n, C, S))
        if want_tables:
            out["cost_table"] = cost.cpu().numpy()
            out["index_table"] = index.cpu().numpy()
        if want_instances:
            out["inst_centerofmass"] = inst_t[0].cpu().numpy()
            out["inst_indices"] = inst_t[1].cpu().numpy()
            out["inst_core"] = inst_t[2].cpu().numpy()
            out["inst_per_class"] = inst_t[3].cpu().numpy()
            out["inst_labels"] = inst_t[4].cpu().numpy()
        return out

    def cluster_instances(self, centerofmass, core_candidates, per_class):
        """Size-filtered DBSCAN (is_cluster_instances) of one image's candidate arrays given as
        numpy: centerofmass [8][n_slots][2] f32, core_candidates [8][n_slots] u8, per_class [8].
        Returns (labels [8][n_slots] int32, packed triples [total][3])."""
        import torch
        p = self.params
        slots = p.cols * p.max_sections
        dev = torch.device("cuda", self.device)
        com = torch.from_numpy(np.ascontiguousarray(centerofmass, np.float32)).to(dev)
        cand = torch.from_numpy(np.ascontiguousarray(core_candidates, np.uint8)).to(dev)
        per = torch.from_numpy(np.ascontiguousarray(per_class, np.int32)).to(dev)
        assert com.shape == (INSTANCE_CLASSES, slots, 2) and cand.shape == (INSTANCE_CLASSES, slots)
        idx = torch.zeros((INSTANCE_CLASSES, slots, 2), dtype=torch.int32, device=dev)
        idx[:, :, 0] = torch.arange(INSTANCE_CLASSES, device=dev, dtype=torch.int32)[:, None]
        idx[:, :, 1] = torch.arange(slots, device=dev, dtype=torch.int32)[None, :]
        lab = torch.full((INSTANCE_CLASSES, slots), -9, dtype=torch.int32, device=dev)
        packed = torch.full((1 + 3 * INSTANCE_CLASSES * slots,), -9, dtype=torch.int32, device=dev)
        ib = InstanceBuffers(com.data_ptr(), idx.data_ptr(), cand.data_ptr(), per.data_ptr(),
                             lab.data_ptr(), packed.data_ptr())
        _check(lib().is_cluster_instances(self._ctx, ctypes.byref(ib),
                                          torch.cuda.current_stream(dev).cuda_stream),
               "is_cluster_instances")
        torch.cuda.synchronize(dev)
        pk = packed.cpu().numpy()
        return lab.cpu().numpy(), pk[1:1 + 3 * int(pk[0])].reshape(-1, 3)


def comm_unique_id():
    """128 bytes of an ncclUniqueId (is_comm_unique_id): created by one rank, handed to all."""
    buf = ctypes.create_string_buffer(128)
    _check(lib().is_comm_unique_id(buf, 128), "is_comm_unique_id")
    return buf.raw


def comm_init_rank(nranks, uid, rank):
    """An RCCL communicator on the current device (is_comm_init_rank) -> ncclComm_t as int."""
    comm = ctypes.c_void_p()
    _check(lib().is_comm_init_rank(ctypes.byref(comm), int(nranks), ctypes.c_char_p(uid), int(rank)),
           "is_comm_init_rank")
    return comm.value


def comm_destroy(comm):
    _check(lib().is_comm_destroy(ctypes.c_void_p(comm)), "is_comm_destroy")


def gather_sections_ptr(comm, dst, columns, d_counts, d_offsets, d_packed, d_all_counts, d_all_packed,
                        cap_sections, stream=0):
    """is_gather_sections on raw device pointers; returns the per-rank section totals (dst: all of them)."""
    cols = np.ascontiguousarray(columns, np.int32)
    totals = np.zeros(cols.size, np.int64)
    _check(lib().is_gather_sections(ctypes.c_void_p(comm), int(dst), _hp(cols), d_counts, d_offsets, d_packed,
                                    d_all_counts, d_all_packed, int(cap_sections), _hp(totals), stream),
           "is_gather_sections")
    return totals


def pack_sections_ptr(d_sections, n_columns, max_sections, d_counts, d_offsets, d_packed, stream=0):
    """is_pack_sections on raw device pointers (ints)."""
    _check(lib().is_pack_sections(d_sections, int(n_columns), int(max_sections), d_counts, d_offsets,
                                  d_packed, stream), "is_pack_sections")


def unpack_sections_ptr(d_counts, d_offsets, d_packed, n_columns, max_sections, d_sections, stream=0):
    _check(lib().is_unpack_sections(d_counts, d_offsets, d_packed, int(n_columns), int(max_sections),
                                    d_sections, stream), "is_unpack_sections")


def flip_and_pad(cnn_out, rows_power2_segmentation, device=0):
    """CNN output [n][CH][Hs][Ws] float32 (numpy) -> DP input [n][Ws][CH][P2S] int32 (numpy)."""
    import torch
    dev = torch.device("cuda", device)
    x = torch.from_numpy(np.ascontiguousarray(cnn_out, np.float32)).to(dev)
    n, CH, Hs, Ws = x.shape
    out = torch.empty((n, Ws, CH, int(rows_power2_segmentation)), dtype=torch.int32, device=dev)
    _check(lib().is_flip_and_pad(x.data_ptr(), out.data_ptr(), n, CH, Hs, Ws,
                                 int(rows_power2_segmentation),
                                 torch.cuda.current_stream(dev).cuda_stream), "is_flip_and_pad")
    torch.cuda.synchronize(dev)
    return out.cpu().numpy()
