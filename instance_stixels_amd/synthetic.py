"""Seeded synthetic inputs for the column-DP path (BASELINE.md §3 / SURVEY.md §8d).

There is no dataset and no CNN in this image, so every test and the bench use this generator:

* disparity, full resolution H x W float32, image row 0 = top: ground ramp below the horizon
  (`alpha*(row-vhor_img) + U(0,1)`), a few vertical object slabs of constant disparity
  (`+ U(0,1)`), sky `U(0,0.5)`; clamped to [0, D-1.01] (input domain);
* segmentation tensor int32 `[C][classes+2][P2S]` in the layout the reference's CNN wrapper
  emits (/root/reference/tools/CNN_training/models/wrappers.py:35-61): per stixel column,
  per channel, 1/8-resolution rows flipped so that index 0 is the image bottom, zero padded to
  P2S = 2^ceil(log2(H/8+1)); channels < classes hold int(8 * -log_softmax), the two offset
  channels hold int(8 * offset) (channel `classes` = y, `classes+1` = x);
* road parameters vhor_img = 0.45 H, alpha = 0.8 D / (H - vhor_img), tilt 0.05, height 1.2.
"""
import dataclasses
import math

import numpy as np

from .config import StixelConfig, DOWNSAMPLE_FACTOR


@dataclasses.dataclass
class Frame:
    disparity: np.ndarray      # [H][W] float32
    segmentation: np.ndarray   # [C][CH][P2S] int32
    vhor_image: int
    camera_tilt: float
    camera_height: float
    alpha_ground: float


def rows_power2(rows: int) -> int:               # Stixels.cu:131
    return int(2 ** math.ceil(math.log2(rows + 1)))


def rows_power2_segmentation(rows: int) -> int:  # Stixels.cu:132-133
    return int(2 ** math.ceil(math.log2(rows // 8 + 1)))


CITY_NOISE = 0.6   # amplitude of the correlated logit noise of the "cityscapes_like" family
FAMILIES = ("scene", "iid_noise", "low_confidence", "flat_disparity", "homogeneous", "many_thin_objects",
            "noisy_disparity", "cityscapes_like")


def make_frame(cfg: StixelConfig, seed: int = 0, n_slabs: int = 6, hole_fraction: float = 0.05,
               zero_segmentation: bool = False, offset_scale: float = 8.0,
               family: str = "scene") -> Frame:
    """family (bench.py `variants.families`; the default "scene" is the data of every earlier
    round, bit for bit): "iid_noise" -- every class logit N(0, 1), the segmentation carries no
    scene at all; "low_confidence" -- the true class's logit is only 1 .. 2 above the N(0, 1)
    rest (a hesitant CNN: class sums separate slowly, the branch-and-bound prunes late);
    "flat_disparity" -- the scene's segmentation over a disparity image without structure
    (constant + U(0, 1) everywhere); "homogeneous" -- road below the horizon, sky above, no object
    at all and a confident CNN (true-class logit +8 .. 9: the class values are 0 almost
    everywhere), the long uniform stretches in which every split is a near-optimal candidate;
    "many_thin_objects" -- about sixty object slabs 8 .. 24 px wide (a crowd / pole scene: hardly a
    column without an object, many short segments); "noisy_disparity" -- the scene with
    N(0, 3) disparity noise on every pixel (a poor stereo matcher: the data terms carry little).
    "cityscapes_like" -- statistics closer to what the reference reports on real data (there is no CNN
    output and no stereo pair in this image): a confident CNN whose errors are SPATIALLY CORRELATED
    (true-class logit +7 .. 8 over low-pass filtered noise of amplitude 0.6, so that the arg-max holds
    over runs of rows instead of flipping per 8x8 cell), a building / vegetation band behind the
    objects above the horizon, disparity noise of 0.25 px, and -- with an invalid-disparity value -- the
    sky and an occlusion band left of every object invalid as a REGION next to 3 % pixel holes.
    On the reference's own shape (784x1792, invalid_disparity = 0) it yields ~3430 stixels per frame in the
    unary and ~2510 in the pairwise model (median 12 per column), where the reference's regression pins are
    2278 and 1421 stixels per Cityscapes image (tests/run_test.sh:124,93); the iid-noise "scene" family
    yields ~9950 / 8270.  The count is set by the models' own splitting of long uniform regions, not by the
    CNN noise (amplitudes 0.3 .. 0.6 give the same count).
    The value range of the class channels is that of the reference's CNN wrapper,
    8 * -log_softmax (wrappers.py:50-60).

    offset_scale: the offset channels hold int(offset_scale * offset in full-resolution
    pixels).  The kernel adds the channel value to the pixel position as it is
    (StixelsKernels.cu:401-405), so 1.0 makes the predicted centres of an object coincide (what a
    trained CNN delivers; used by the clustering tests); the default 8.0 is the bench / parity
    data of round 1 (centres scattered 8x wider: hardly any instance clusters), kept so that
    throughput numbers stay comparable across rounds."""
    rng = np.random.Generator(np.random.PCG64(seed))
    H, W, D = int(cfg.rows), int(cfg.cols), int(cfg.max_dis)
    C = cfg.realcols
    K = cfg.n_semantic_classes
    CH = K + cfg.n_offset_channels
    P2S = rows_power2_segmentation(H)
    Hs = H // DOWNSAMPLE_FACTOR

    vhor_img = int(0.45 * H)
    alpha = 0.8 * D / (H - vhor_img)

    rows = np.arange(H, dtype=np.float32)[:, None]
    disp = np.where(rows > vhor_img, alpha * (rows - vhor_img), 0.0).astype(np.float32)
    disp = np.broadcast_to(disp, (H, W)).copy()
    below = (rows > vhor_img)
    disp += np.where(below, rng.random((H, W), dtype=np.float32),
                     0.5 * rng.random((H, W), dtype=np.float32))

    # label image at full resolution: 0 road, 1 sidewalk, 10 sky, objects 2..18 (not 10)
    label = np.where(below, 0, 10).astype(np.int32)
    label = np.broadcast_to(label, (H, W)).copy()
    label[:, : W // 6][np.broadcast_to(below, (H, W))[:, : W // 6]] = 1   # sidewalk strip
    centre_x = np.zeros((H, W), np.float32)
    centre_y = np.zeros((H, W), np.float32)
    has_obj = np.zeros((H, W), bool)
    obj_classes = [2, 5, 8, 11, 13, 12, 17, 18, 14, 3]
    if family not in FAMILIES:
        raise ValueError(f"unknown input family {family!r}")
    city = family == "cityscapes_like"
    if city:   # a band of buildings / vegetation above the horizon, behind everything else
        band_top = max(0, vhor_img - H // 4)
        x = 0
        while x < W:
            w = int(rng.integers(max(1, W // 16), max(2, W // 5)))
            cls_b = 2 if rng.random() < 0.6 else 8
            top_b = int(rng.integers(max(0, band_top - H // 10), band_top + H // 10 + 1))
            d_b = float(rng.uniform(2.0, 6.0))
            label[top_b:vhor_img + 1, x:x + w] = cls_b
            disp[top_b:vhor_img + 1, x:x + w] = d_b + 0.5 * rng.random((vhor_img + 1 - top_b, min(w, W - x)),
                                                                      dtype=np.float32)
            x += w
        n_slabs = 10
    if family == "homogeneous":
        n_slabs = 0
    elif family == "many_thin_objects":
        n_slabs = 60
    for s in range(n_slabs):
        if family == "many_thin_objects":
            w = min(int(rng.integers(8, 25)), W)
        else:
            w = int(rng.integers(max(8, W // 40), max(9, W // 8)))
        x0 = int(rng.integers(0, max(1, W - w)))
        foot = int(rng.integers(vhor_img + max(2, H // 16), H))          # image row of the foot
        top = int(rng.integers(max(0, vhor_img - H // 3), max(1, foot - H // 16)))
        d_obj = float(np.clip(alpha * (foot - vhor_img), 1.5, D - 2.5))
        disp[top:foot, x0:x0 + w] = d_obj + rng.random((foot - top, w), dtype=np.float32)
        cls = obj_classes[s % len(obj_classes)]
        label[top:foot, x0:x0 + w] = cls
        has_obj[top:foot, x0:x0 + w] = cls >= 11
        centre_x[top:foot, x0:x0 + w] = x0 + 0.5 * w
        centre_y[top:foot, x0:x0 + w] = 0.5 * (top + foot)
    if family == "flat_disparity":   # keep only the U(0, 1) part of every pixel
        disp = np.float32(D // 3) + (disp - np.floor(disp))
    if family == "noisy_disparity":
        disp = disp + rng.normal(0.0, 3.0, (H, W)).astype(np.float32)
    if city:
        disp = disp - (disp - np.floor(disp)) + 0.5 + rng.normal(0.0, 0.25, (H, W)).astype(np.float32)
    disp = np.clip(disp, 0.0, D - 1.01).astype(np.float32)
    if cfg.invalid_disparity >= 0 and hole_fraction > 0:
        holes = rng.random((H, W)) < (0.03 if city else hole_fraction)
        if city:   # no match in the sky, and an occlusion band left of every object
            holes |= label == 10
            obj = (label >= 11) | (label == 5)
            edge = obj & ~np.roll(obj, 1, axis=1)
            for k in range(1, 13):
                holes |= np.roll(edge, -k + 1 - 12, axis=1) & ~obj
        disp[holes] = np.float32(cfg.invalid_disparity)

    seg = np.zeros((C, CH, P2S), np.int32)
    if not zero_segmentation:
        # sample the label / centre images at the 1/8 grid (centre of each 8x8 cell)
        ys = np.arange(Hs) * DOWNSAMPLE_FACTOR + DOWNSAMPLE_FACTOR // 2
        xs = cfg.width_margin + np.arange(C) * cfg.column_step + cfg.column_step // 2
        ys = np.minimum(ys, H - 1)
        xs = np.minimum(xs, W - 1)
        lab = label[np.ix_(ys, xs)]                                     # [Hs][C]
        logits = rng.normal(0.0, 1.0, (Hs, C, K)).astype(np.float32)
        if city:   # spatially correlated errors: low-pass filtered noise (a separable box filter, twice)
            from scipy.ndimage import uniform_filter1d
            for ax, n in ((0, 9), (1, 5), (0, 9), (1, 5)):
                logits = uniform_filter1d(logits, size=n, axis=ax, mode="nearest")
            logits = (logits / max(float(logits.std()), 1e-6) * CITY_NOISE).astype(np.float32)
        if family != "iid_noise":
            true_logit = {"low_confidence": 1.0, "homogeneous": 8.0, "cityscapes_like": 7.0}.get(family, 4.0)
            np.put_along_axis(logits, lab[..., None],
                              np.float32(true_logit) + rng.random((Hs, C, 1), dtype=np.float32), axis=2)
        logits -= logits.max(axis=2, keepdims=True)
        nlogp = -(logits - np.log(np.exp(logits).sum(axis=2, keepdims=True)))
        sem = (8.0 * nlogp).astype(np.int32)                            # [Hs][C][K]
        oy = np.where(has_obj[np.ix_(ys, xs)], centre_y[np.ix_(ys, xs)] - ys[:, None], 0.0)
        ox = np.where(has_obj[np.ix_(ys, xs)], centre_x[np.ix_(ys, xs)] - xs[None, :], 0.0)
        oy = oy + rng.normal(0.0, 2.0, oy.shape)
        ox = ox + rng.normal(0.0, 2.0, ox.shape)
        # the reference's convention: my = row - offy (rows counted from the bottom), so a
        # positive stored y offset points DOWN in the image (StixelsKernels.cu:400-405)
        off_y = (offset_scale * -oy).astype(np.int32)
        off_x = (offset_scale * ox).astype(np.int32)
        # flip rows: index 0 = bottom of the image
        seg[:, :K, :Hs] = sem[::-1].transpose(1, 2, 0)
        seg[:, K, :Hs] = off_y[::-1].T
        seg[:, K + 1, :Hs] = off_x[::-1].T
    return Frame(disparity=disp, segmentation=np.ascontiguousarray(seg), vhor_image=vhor_img,
                 camera_tilt=0.05, camera_height=1.2, alpha_ground=float(alpha))


def algorithmic_bytes_per_image(cfg: StixelConfig, joined_input: bool = False) -> int:
    """4*H*W (or 4*H*C) + 4*C*CH*P2S + 32*C*200."""
    H, W, C = int(cfg.rows), int(cfg.cols), cfg.realcols
    CH = cfg.n_semantic_classes + cfg.n_offset_channels
    disp = 4 * H * (C if joined_input else W)
    return disp + 4 * C * CH * rows_power2_segmentation(H) + 32 * C * 200


def pair_evaluations_per_image(cfg: StixelConfig) -> int:
    H, C = int(cfg.rows), cfg.realcols
    return C * H * (H + 1) // 2
