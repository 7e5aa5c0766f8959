"""Host-side configuration types of the column-DP path.

`StixelConfig` mirrors `struct StixelConfig` of the reference
(/root/reference/InstanceStixels/include/InstanceStixels/types.h:30-141): same field names, same
defaults.  `StixelParams` / `SECTION_DTYPE` mirror `StixelParameters` (types.h:145-184) and
`Section` (types.h:186-194) byte for byte; they are the ctypes/numpy view of the C structs in
include/instance_stixels_core.h.
"""
import ctypes
import dataclasses

import numpy as np

GROUND, OBJECT, SKY = 0, 1, 2
DOWNSAMPLE_FACTOR = 8          # configuration.h:31
MAX_STIXELS_PER_COLUMN = 200   # configuration.h:32
INSTANCE_CLASSES = 8           # Stixels.cu:47
FIRST_INSTANCE_CLASS = 11      # StixelsKernels.cu:926


@dataclasses.dataclass
class StixelConfig:
    # --- no defaults in the reference: must be set (types.h:31-69)
    rows: int = -1
    cols: int = -1
    max_dis: int = -1
    invalid_disparity: float = -1.0
    eps: float = -1
    min_pts: int = -1
    size_filter: int = -1
    n_semantic_classes: int = -1
    n_offset_channels: int = -1
    prior_weight: float = -1
    segmentation_weight: float = -1
    instance_weight: float = -1
    disparity_weight: float = -1
    pairwise: bool = False
    column_step: int = -1
    focal: float = -1
    baseline: float = -1
    camera_center_x: float = -1
    camera_center_y: float = -1
    # --- defaults (types.h:96-140)
    sigma_disparity_object: float = 1.0
    sigma_disparity_ground: float = 2.0
    sigma_sky: float = 0.1
    pout: float = 0.15
    pout_sky: float = 0.4
    pord: float = 0.2
    pgrav: float = 0.1
    pblg: float = 0.04
    pground_given_nexist: float = 0.28
    pobject_given_nexist: float = 0.44
    psky_given_nexist: float = 0.28
    pnexist_dis: float = 0.25
    pground: float = 1.0 / 3.0
    pobject: float = 1.0 / 3.0
    psky: float = 1.0 / 3.0
    width_margin: int = 0
    sigma_camera_tilt: float = 0.05
    sigma_camera_height: float = 0.05
    median_join: bool = False
    epsilon: float = 3.0
    range_objects_z: float = 10.20
    road_vdisparity_threshold: float = 0.2

    @property
    def realcols(self) -> int:          # Stixels.cu:44
        return (int(self.cols) - self.width_margin) // self.column_step


class StixelParams(ctypes.Structure):
    """`is_stixel_params` == `StixelParameters` (types.h:145-184)."""
    _fields_ = [
        ("vhor", ctypes.c_int), ("rows", ctypes.c_int), ("rows_power2", ctypes.c_int),
        ("rows_power2_segmentation", ctypes.c_int), ("cols", ctypes.c_int),
        ("max_dis", ctypes.c_int), ("rows_log", ctypes.c_float),
        ("pnexists_given_sky_log", ctypes.c_float), ("normalization_sky", ctypes.c_float),
        ("inv_sigma2_sky", ctypes.c_float), ("puniform_sky", ctypes.c_float),
        ("nopnexists_given_sky_log", ctypes.c_float),
        ("pnexists_given_ground_log", ctypes.c_float), ("puniform", ctypes.c_float),
        ("nopnexists_given_ground_log", ctypes.c_float),
        ("pnexists_given_object_log", ctypes.c_float),
        ("nopnexists_given_object_log", ctypes.c_float), ("baseline", ctypes.c_float),
        ("focal", ctypes.c_float), ("range_objects_z", ctypes.c_float),
        ("pord", ctypes.c_float), ("epsilon", ctypes.c_float), ("pgrav", ctypes.c_float),
        ("pblg", ctypes.c_float), ("max_dis_log", ctypes.c_float),
        ("max_sections", ctypes.c_int), ("width_margin", ctypes.c_int),
        ("segmentation_classes", ctypes.c_int), ("segmentation_channels", ctypes.c_int),
        ("prior_weight", ctypes.c_float), ("disparity_weight", ctypes.c_float),
        ("segmentation_weight", ctypes.c_float), ("instance_weight", ctypes.c_float),
        ("column_step", ctypes.c_int), ("clustering_eps", ctypes.c_float),
        ("clustering_min_pts", ctypes.c_int), ("clustering_size_filter", ctypes.c_int),
        ("invalid_disparity", ctypes.c_float),
    ]

    def as_dict(self):
        return {name: getattr(self, name) for name, _ in self._fields_}


SECTION_DTYPE = np.dtype([
    ("type", np.int32), ("vB", np.int32), ("vT", np.int32), ("disparity", np.float32),
    ("semantic_class", np.int32), ("cost", np.float32), ("instance_meanx", np.float32),
    ("instance_meany", np.float32),
])
assert SECTION_DTYPE.itemsize == 32 and ctypes.sizeof(StixelParams) == 152


# ---------------------------------------------------------------------------------------------
# Named parameter sets.  The YAML dumps in /root/reference/cfg/ are dynamic_reconfigure presets
# that no reference code loads; the values below are those numbers.
# ---------------------------------------------------------------------------------------------
_CITYSCAPES_CAMERA = dict(focal=2262.52, baseline=0.209313, camera_center_x=1096.98,
                          camera_center_y=513.137)  # types.h:80-82 (comment block)

PRESETS = {
    # cfg/drn_d_22_unary_cfg.yaml
    "drn_d_22_unary": dict(disparity_weight=0.006993, segmentation_weight=11.241965,
                           instance_weight=0.001731, prior_weight=1e4, pairwise=False,
                           eps=23.89408, min_pts=4, size_filter=42, invalid_disparity=-1.0,
                           pground=0.33, pobject=0.33, psky=0.33),
    # cfg/drn_d_38_unary_cfg.yaml
    "drn_d_38_unary": dict(disparity_weight=0.000637, segmentation_weight=14.949844,
                           instance_weight=0.013686, prior_weight=1e4, pairwise=False,
                           eps=18.542413, min_pts=4, size_filter=35, invalid_disparity=-1.0,
                           pground=0.33, pobject=0.33, psky=0.33),
    # cfg/drn_d_22_pairwise_cfg.yaml
    "drn_d_22_pairwise": dict(disparity_weight=0.000314, segmentation_weight=2.553681,
                              instance_weight=0.000918, prior_weight=1.0, pairwise=True,
                              eps=15.417949, min_pts=3, size_filter=1, invalid_disparity=-1.0,
                              pground=0.33, pobject=0.33, psky=0.33),
    # cfg/drn_d_38_pairwise_cfg.yaml
    "drn_d_38_pairwise": dict(disparity_weight=0.0001, segmentation_weight=4.7095,
                              instance_weight=0.003131, prior_weight=1.0, pairwise=True,
                              eps=18.822322, min_pts=3, size_filter=25, invalid_disparity=-1.0,
                              pground=0.33, pobject=0.33, psky=0.33),
    # BASELINE.json configs[0]: disparity-only (segmentation_weight = 0 => instance weight 0,
    # Stixels.cu:416-422)
    "disparity_only_unary": dict(disparity_weight=1.0, segmentation_weight=0.0,
                                 instance_weight=0.0, prior_weight=1e4, pairwise=False,
                                 eps=20.0, min_pts=4, size_filter=30, invalid_disparity=-1.0),
    "disparity_only_pairwise": dict(disparity_weight=1.0, segmentation_weight=0.0,
                                    instance_weight=0.0, prior_weight=1.0, pairwise=True,
                                    eps=20.0, min_pts=4, size_filter=30, invalid_disparity=-1.0),
}


def make_config(preset: str, rows: int, cols: int, max_dis: int, **overrides) -> StixelConfig:
    """StixelConfig for a named preset at a given frame shape (column_step 8, 19+2 channels)."""
    kw = dict(rows=rows, cols=cols, max_dis=max_dis, column_step=8, n_semantic_classes=19,
              n_offset_channels=2, **_CITYSCAPES_CAMERA)
    kw.update(PRESETS[preset])
    kw.update(overrides)
    return StixelConfig(**kw)
