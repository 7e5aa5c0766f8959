"""Python view of the C++ `Stixels` host class (instance_stixels_amd/host/Stixels.cpp).

Method names, argument meaning and call order are those of the reference's class
(/root/reference/InstanceStixels/include/InstanceStixels/Stixels.hpp:40-96) so that tests read
like its callers (apps/run_cityscapes.cu:328-449).  Everything executes in the C++ library; this
module only marshals numpy arrays.
"""
import ctypes
import dataclasses
import os

import numpy as np

from . import core as _core
from .config import StixelConfig, StixelParams, SECTION_DTYPE

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libInstanceStixels.so")
_LIB = None

EXPORTS = [
    "ish_last_error", "ish_create", "ish_destroy", "ish_set_config", "ish_initialize",
    "ish_precompute_host",
    "ish_finish", "ish_is_initialized", "ish_real_cols", "ish_max_sections",
    "ish_get_parameters", "ish_get_luts", "ish_core_context", "ish_set_disparity_image",
    "ish_set_segmentation", "ish_set_road_parameters", "ish_get_ground_model", "ish_compute",
    "ish_get_instance_stixels", "ish_get_3d_vertices", "ish_save_stixels", "ish_time_compute",
    "ish_set_device", "ish_compute_batch", "ish_time_compute_batch", "ish_compute_batch_gather",
    "ire_create", "ire_destroy", "ire_initialize", "ire_finish", "ire_compute", "ire_get_binary",
    "ire_hough_lines", "ire_set_device", "ire_active_device", "ire_compute_device",
    "ish_get_input_disparity_on_device",
]


class _IshConfig(ctypes.Structure):
    _f, _i = ctypes.c_float, ctypes.c_int
    _fields_ = [
        ("rows", _f), ("cols", _f), ("max_dis", _i), ("invalid_disparity", _f), ("eps", _f),
        ("min_pts", _i), ("size_filter", _i), ("n_semantic_classes", _i),
        ("n_offset_channels", _i), ("prior_weight", _f), ("segmentation_weight", _f),
        ("instance_weight", _f), ("disparity_weight", _f), ("pairwise", _i), ("column_step", _i),
        ("focal", _f), ("baseline", _f), ("camera_center_x", _f), ("camera_center_y", _f),
        ("sigma_disparity_object", _f), ("sigma_disparity_ground", _f), ("sigma_sky", _f),
        ("pout", _f), ("pout_sky", _f), ("pord", _f), ("pgrav", _f), ("pblg", _f),
        ("pground_given_nexist", _f), ("pobject_given_nexist", _f), ("psky_given_nexist", _f),
        ("pnexist_dis", _f), ("pground", _f), ("pobject", _f), ("psky", _f), ("width_margin", _i),
        ("sigma_camera_tilt", _f), ("sigma_camera_height", _f), ("median_join", _i),
        ("epsilon", _f), ("range_objects_z", _f), ("road_vdisparity_threshold", _f),
    ]


def lib():
    global _LIB
    if _LIB is None:
        _core.lib()  # loads torch's HIP runtime first (if any) and libis_core.so
        if not os.path.exists(LIB_PATH):
            raise _core.CoreError(f"{LIB_PATH} is missing: run __graft_entry__.build()")
        L = ctypes.CDLL(LIB_PATH)
        vp, ci, cf = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
        L.ish_last_error.restype = ctypes.c_char_p
        L.ish_create.restype = vp
        L.ish_destroy.argtypes = [vp]
        L.ish_set_config.argtypes = [vp, ctypes.POINTER(_IshConfig)]
        L.ish_initialize.argtypes = [vp, ci]
        L.ish_finish.argtypes = [vp]
        L.ish_precompute_host.argtypes = [vp]
        L.ish_is_initialized.argtypes = [vp]
        L.ish_real_cols.argtypes = [vp]
        L.ish_max_sections.argtypes = [vp]
        L.ish_get_parameters.argtypes = [vp, ctypes.POINTER(StixelParams)]
        L.ish_get_luts.argtypes = [vp, vp, vp]
        L.ish_core_context.argtypes = [vp]
        L.ish_core_context.restype = vp
        L.ish_set_disparity_image.argtypes = [vp, vp, ctypes.c_size_t]
        L.ish_set_segmentation.argtypes = [vp, vp, ctypes.c_size_t]
        L.ish_set_road_parameters.argtypes = [vp, ci, cf, cf, cf]
        L.ish_get_ground_model.argtypes = [vp, vp, vp, vp, ctypes.POINTER(ci)]
        L.ish_compute.argtypes = [vp, ci, vp, vp, ctypes.POINTER(cf), ctypes.POINTER(cf)]
        L.ish_get_instance_stixels.argtypes = [vp, vp, ci]
        L.ish_time_compute.argtypes = [vp, ci, ci, ci, ctypes.POINTER(ctypes.c_double)]
        L.ish_set_device.argtypes = [vp, ci]
        L.ish_compute_batch.argtypes = [vp, ci, ci, vp, vp, vp, vp, vp, vp, ci, vp, vp]
        L.ish_compute_batch_gather.argtypes = [vp, ci, ci, vp, vp, vp, vp, ci, vp, vp, ci, vp, vp, vp, vp]
        L.ish_time_compute_batch.argtypes = [vp, ci, ci, vp, vp, vp, ci, ci,
                                             ctypes.POINTER(ctypes.c_double)]
        L.ish_get_3d_vertices.argtypes = [vp, vp, cf, ci, vp, ci]
        L.ish_save_stixels.argtypes = [vp, vp, vp, ci, cf, ci, ctypes.c_char_p]
        L.ire_create.restype = vp
        L.ire_destroy.argtypes = [vp]
        L.ire_initialize.argtypes = [vp, cf, cf, cf, ci, ci, ci, cf]
        L.ire_finish.argtypes = [vp]
        L.ire_compute.argtypes = [vp, vp, ctypes.c_size_t, vp]
        L.ire_get_binary.argtypes = [vp, vp, ctypes.c_size_t]
        L.ire_hough_lines.argtypes = [vp, ci, ci, cf, cf, ci, vp, ci]
        L.ire_set_device.argtypes = [vp, ci]
        L.ire_active_device.argtypes = [vp]
        L.ire_compute_device.argtypes = [vp, vp, vp]
        L.ish_get_input_disparity_on_device.argtypes = [vp]
        L.ish_get_input_disparity_on_device.restype = vp
        _LIB = L
    return _LIB


@dataclasses.dataclass
class StixelsData:               # types.h:196-205
    sections: np.ndarray         # [realcols][max_sections] SECTION_DTYPE
    rows: int
    cols: int
    realcols: int
    max_sections: int
    max_dis: int
    column_step: int
    semantic_classes: int
    alpha_ground: float
    vhor: int


class Stixels:
    def __init__(self):
        self._h = ctypes.c_void_p(lib().ish_create())
        self._cfg = None

    def _check(self, rc, what):
        if rc == -1:
            raise ValueError(lib().ish_last_error().decode())   # std::invalid_argument
        if rc < 0:
            raise RuntimeError(f"{what}: {lib().ish_last_error().decode()}")
        return rc

    def close(self):
        if self._h:
            lib().ish_finish(self._h)
            lib().ish_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- reference API ---------------------------------------------------------------
    def SetConfig(self, cfg: StixelConfig):
        c = _IshConfig()
        for name, _ in _IshConfig._fields_:
            v = getattr(cfg, name)
            setattr(c, name, int(v) if isinstance(v, (bool, np.bool_)) else v)
        self._check(lib().ish_set_config(self._h, ctypes.byref(c)), "SetConfig")
        self._cfg = cfg

    def Initialize(self, max_batch=1):
        self._check(lib().ish_initialize(self._h, int(max_batch)), "Initialize")

    def SetDevice(self, device):
        """GPU of the next Initialize() (default: the caller's current HIP device)."""
        self._check(lib().ish_set_device(self._h, int(device)), "SetDevice")

    def GetInputDisparityImageOnDevice(self):
        """Device address of the internal full-resolution disparity buffer (Stixels.cu:357-359)."""
        return int(lib().ish_get_input_disparity_on_device(self._h) or 0)

    def time_compute(self, pairwise, n_iter=100, with_instances=False):
        """Seconds per frame of n_iter Stixels::Compute() calls timed inside the C++ library."""
        t = ctypes.c_double()
        self._check(lib().ish_time_compute(self._h, int(bool(pairwise)), int(n_iter),
                                           int(bool(with_instances)), ctypes.byref(t)),
                    "time_compute")
        return t.value

    def PrecomputeHost(self):
        """Host half of Initialize (tables + StixelParameters); needs no GPU."""
        self._check(lib().ish_precompute_host(self._h), "PrecomputeHost")

    def Finish(self):
        self._check(lib().ish_finish(self._h), "Finish")

    def IsInitialized(self):
        return bool(lib().ish_is_initialized(self._h))

    def GetRealCols(self):
        return lib().ish_real_cols(self._h)

    def GetMaxSections(self):
        return lib().ish_max_sections(self._h)

    def SetDisparityImage(self, disp):
        a = np.ascontiguousarray(disp, np.float32)
        self._check(lib().ish_set_disparity_image(self._h, a.ctypes.data, a.size),
                    "SetDisparityImage")

    def SetSegmentation(self, seg):
        a = np.ascontiguousarray(seg, np.int32)
        self._check(lib().ish_set_segmentation(self._h, a.ctypes.data, a.size), "SetSegmentation")

    def SetRoadParameters(self, vhor, camera_tilt, camera_height, alpha_ground):
        self._check(lib().ish_set_road_parameters(self._h, int(vhor), camera_tilt, camera_height,
                                                  alpha_ground), "SetRoadParameters")

    def Compute(self, pairwise) -> StixelsData:
        C, S = self.GetRealCols(), self.GetMaxSections()
        sec = np.zeros((C, S), SECTION_DTYPE)
        hdr = np.zeros(9, np.int32)
        alpha, ret = ctypes.c_float(), ctypes.c_float()
        self._check(lib().ish_compute(self._h, int(bool(pairwise)), sec.ctypes.data,
                                      hdr.ctypes.data, ctypes.byref(alpha), ctypes.byref(ret)),
                    "Compute")
        return StixelsData(sec, *[int(x) for x in hdr[:7]], float(alpha.value), int(hdr[7]))

    def ComputeBatch(self, pairwise, d_disparity_big, d_segmentation, road, with_instances=True,
                     stream=0):
        """Stixels::ComputeBatch on device-resident inputs (device pointers as ints):
        d_disparity_big [n][rows][cols] f32, d_segmentation [n][realcols][channels][P2S] i32,
        road: n tuples (vhor_image, camera_tilt, camera_height, alpha_ground).
        Returns (list of StixelsData, list of instance mappings or None)."""
        n = len(road)
        C, S = self.GetRealCols(), self.GetMaxSections()
        rp = np.ascontiguousarray(road, np.float32).reshape(n, 4)
        sec = np.zeros((n, C, S), SECTION_DTYPE)
        vh = np.zeros(n, np.int32)
        cap = C * S
        tri = np.zeros((n, cap, 3), np.int32) if with_instances else None
        cnt = np.zeros(n, np.int32)
        self._check(lib().ish_compute_batch(self._h, int(bool(pairwise)), n, d_disparity_big,
                                            d_segmentation, rp.ctypes.data, sec.ctypes.data,
                                            vh.ctypes.data, tri.ctypes.data if with_instances else None,
                                            cap, cnt.ctypes.data, stream), "ComputeBatch")
        cfg = self._cfg
        data = [StixelsData(sec[i], int(cfg.rows), int(cfg.cols), C, S, int(cfg.max_dis),
                            int(cfg.column_step), int(cfg.n_semantic_classes), float(rp[i, 3]),
                            int(vh[i])) for i in range(n)]
        maps = None
        if with_instances:
            maps = [{(int(u), int(v)): int(l) for u, v, l in tri[i, :cnt[i]]} for i in range(n)]
        return data, maps

    def ComputeBatchGather(self, pairwise, d_disparity_big, d_segmentation, road, comm, dst,
                           images_per_rank, road_all=None, stream=0):
        """Stixels::ComputeBatchGather: this rank's shard, then the RCCL gather of every rank's Sections on
        rank `dst` of `comm` (an ncclComm_t as int, core.comm_init_rank).  road: this rank's tuples
        (vhor_image, camera_tilt, camera_height, alpha_ground); road_all: those of ALL frames in rank
        order (dst only).  Returns the list of StixelsData of all frames on dst, [] elsewhere."""
        n = len(road)
        C, S = self.GetRealCols(), self.GetMaxSections()
        rp = np.ascontiguousarray(road, np.float32).reshape(n, 4)
        ipr = np.ascontiguousarray(images_per_rank, np.int32)
        n_all = int(ipr.sum())
        ra = None if road_all is None else np.ascontiguousarray(road_all, np.float32).reshape(n_all, 4)
        sec = np.zeros((n_all, C, S), SECTION_DTYPE)
        vh = np.zeros(n_all, np.int32)
        n_out = ctypes.c_int(0)
        self._check(lib().ish_compute_batch_gather(
            self._h, int(bool(pairwise)), n, d_disparity_big, d_segmentation, rp.ctypes.data,
            ctypes.c_void_p(int(comm)), int(dst), ipr.ctypes.data, None if ra is None else ra.ctypes.data, n_all,
            sec.ctypes.data, vh.ctypes.data, ctypes.byref(n_out), stream), "ComputeBatchGather")
        cfg = self._cfg
        return [StixelsData(sec[i], int(cfg.rows), int(cfg.cols), C, S, int(cfg.max_dis), int(cfg.column_step),
                            int(cfg.n_semantic_classes), float(ra[i, 3]), int(vh[i])) for i in range(n_out.value)]

    def time_compute_batch(self, pairwise, d_disparity_big, d_segmentation, road, n_iter=5,
                           with_instances=False):
        """Seconds per ComputeBatch() call, timed inside the C++ library."""
        rp = np.ascontiguousarray(road, np.float32).reshape(len(road), 4)
        t = ctypes.c_double()
        self._check(lib().ish_time_compute_batch(self._h, int(bool(pairwise)), len(road),
                                                 d_disparity_big, d_segmentation, rp.ctypes.data,
                                                 int(n_iter), int(bool(with_instances)),
                                                 ctypes.byref(t)), "time_compute_batch")
        return t.value

    def GetInstanceStixels(self):
        cap = self.GetRealCols() * self.GetMaxSections()
        t = np.zeros((cap, 3), np.int32)
        n = self._check(lib().ish_get_instance_stixels(self._h, t.ctypes.data, cap),
                        "GetInstanceStixels")
        return {(int(u), int(v)): int(l) for u, v, l in t[:n]}

    def Get3DVertices(self, data: StixelsData):
        cap = data.sections.size * 12
        out = np.zeros(cap, np.float32)
        sec = np.ascontiguousarray(data.sections)
        n = self._check(lib().ish_get_3d_vertices(self._h, sec.ctypes.data, data.alpha_ground,
                                                  data.vhor, out.ctypes.data, cap),
                        "Get3DVertices")
        return out[:n].copy()

    def SaveStixels(self, data: StixelsData, instance_stixels, alpha_ground, vhor, fname):
        t = np.array([[u, v, l] for (u, v), l in instance_stixels.items()],
                     np.int32).reshape(-1, 3)
        sec = np.ascontiguousarray(data.sections)
        self._check(lib().ish_save_stixels(self._h, sec.ctypes.data, t.ctypes.data, len(t),
                                           alpha_ground, int(vhor), fname.encode()),
                    "SaveStixels")

    # ---- introspection ---------------------------------------------------------------
    def GetParameters(self) -> StixelParams:
        p = StixelParams()
        lib().ish_get_parameters(self._h, ctypes.byref(p))
        return p

    def GetLUTs(self):
        D = self.GetParameters().max_dis
        lut = np.zeros((D, D), np.float32)
        odr = np.zeros(D, np.float32)
        lib().ish_get_luts(self._h, lut.ctypes.data, odr.ctypes.data)
        return lut, odr

    def GetGroundModel(self):
        H = self.GetParameters().rows
        gf, ng, ig = (np.zeros(H, np.float32) for _ in range(3))
        vh = ctypes.c_int()
        self._check(lib().ish_get_ground_model(self._h, gf.ctypes.data, ng.ctypes.data,
                                               ig.ctypes.data, ctypes.byref(vh)),
                    "GetGroundModel")
        return gf, ng, ig, vh.value


def hough_lines(image, rho=1.0, theta=float(np.pi / 180), threshold=25, cap=4096):
    """Standard Hough transform of the host library (RoadEstimation::HoughLines)."""
    img = np.ascontiguousarray(image, np.uint8)
    out = np.zeros((cap, 2), np.float32)
    n = lib().ire_hough_lines(img.ctypes.data, img.shape[0], img.shape[1], rho, theta, threshold,
                              out.ctypes.data, cap)
    return out[:min(n, cap)].copy()


class RoadEstimation:
    """Python view of the C++ RoadEstimation class (RoadEstimation.h of the reference)."""

    def __init__(self):
        self._h = ctypes.c_void_p(lib().ire_create())
        self._shape = None

    def Initialize(self, camera_center_y, baseline, focal, rows, cols, max_dis,
                   road_vdisparity_threshold=0.2):
        rc = lib().ire_initialize(self._h, camera_center_y, baseline, focal, rows, cols, max_dis,
                                  road_vdisparity_threshold)
        if rc < 0:
            raise RuntimeError(lib().ish_last_error().decode())
        self._shape = (rows, max_dis)

    def Compute(self, disparity):
        a = np.ascontiguousarray(disparity, np.float32)
        out = np.zeros(4, np.float32)
        rc = lib().ire_compute(self._h, a.ctypes.data, a.size, out.ctypes.data)
        if rc < 0:
            raise RuntimeError(lib().ish_last_error().decode())
        self.pitch, self.camera_height, self.slope = float(out[0]), float(out[1]), float(out[2])
        self.horizon_point = int(out[3])
        return bool(rc)

    def SetDevice(self, device):
        """Device of the next Initialize() (RoadEstimation::SetDevice, an addition like Stixels::SetDevice)."""
        lib().ire_set_device(self._h, int(device))

    def GetActiveDevice(self):
        return int(lib().ire_active_device(self._h))

    def ComputeOnDevice(self, d_ptr):
        """RoadEstimation::Compute(pixel_t* d_im): the image is on the object's device already."""
        out = np.zeros(4, np.float32)
        rc = lib().ire_compute_device(self._h, ctypes.c_void_p(int(d_ptr)), out.ctypes.data)
        if rc < 0:
            raise RuntimeError(lib().ish_last_error().decode())
        self.pitch, self.camera_height, self.slope = float(out[0]), float(out[1]), float(out[2])
        self.horizon_point = int(out[3])
        return bool(rc)

    def GetBinaryVDisparity(self):
        out = np.zeros(self._shape, np.uint8)
        lib().ire_get_binary(self._h, out.ctypes.data, out.size)
        return out

    def Finish(self):
        lib().ire_finish(self._h)

    def close(self):
        if self._h:
            lib().ire_finish(self._h)
            lib().ire_destroy(self._h)
            self._h = ctypes.c_void_p()
