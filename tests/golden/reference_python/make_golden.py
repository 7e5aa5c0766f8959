"""Generates tests/golden/reference_python/f1_f2_reference_python.npz -- the pin of SURVEY rows f1 / f2.

BUILD-CONTAINER ONLY: reads /root/reference at generation time; only the resulting vectors (data)
are committed and travel to the GPU box.

What is executed from the reference: the three functions `read_stixel_file`,
`get_instance_means` and `assign_instances` of
/root/reference/tools/visualization/clustering_visualization.py (lines 73-116, 821-844, 894-960),
byte for byte as they stand there.  The module cannot be imported as a whole: its top level builds
colour tables through `cv2.cvtColor` and `cityscapesscripts` (both absent in this image, lines
32-52), i.e. it touches the absent packages on import.  So the three function definitions are cut
out of the parsed source (ast) and executed in a namespace that holds ONLY numpy, copy and
sklearn's DBSCAN -- any use of cv2 / h5py / cityscapesscripts / matplotlib inside them would be a
NameError, which proves they do not depend on the absent packages.  No stand-in module is written.

Pipeline per case (small frames, CPU only):
  oracle DP -> Section[]                      (sections are bit-equal to the HIP path, parity tests)
  product  Stixels::SaveStixels  -> text A (no instance labels)        [f2, Stixels.cu:889-926]
  reference read_stixel_file(text A)          -> parsed stixels        [the format's only reader]
  reference assign_instances(parsed, cfg)     -> instance labels       [twin of Stixels.cu:639-681]
  product  Stixels::SaveStixels with the twin-oracle labels -> text B
  reference read_stixel_file(text B)          -> parsed instance labels (label + class*1000)

    python tests/golden/reference_python/make_golden.py
"""
import ast
import copy
import io
import os
import sys
import tempfile
import contextlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

REF = "/root/reference/tools/visualization/clustering_visualization.py"
WANTED = ("read_stixel_file", "get_instance_means", "assign_instances")

CASES = [  # preset, rows, cols, max_dis, seed, n_slabs, overrides
    ("drn_d_22_unary", 256, 1024, 64, 5, 14, dict(size_filter=12, eps=23.89408, min_pts=4)),
    ("drn_d_38_pairwise", 256, 1024, 64, 9, 18, dict(size_filter=8, eps=18.822322, min_pts=3)),
    ("drn_d_22_unary", 128, 512, 32, 2, 8, dict(size_filter=6, eps=30.0, min_pts=2)),
]


def reference_functions():
    from sklearn.cluster import DBSCAN
    src = open(REF).read()
    tree = ast.parse(src)
    picked = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in WANTED]
    assert sorted(n.name for n in picked) == sorted(WANTED)
    mod = ast.Module(body=picked, type_ignores=[])
    ns = {"np": np, "copy": copy, "DBSCAN": DBSCAN}
    exec(compile(mod, REF, "exec"), ns)
    return ns


def sections_of(case, ref):
    """[C][S] SECTION_DTYPE -> list of (col, idx) in file order."""
    import helpers
    secs = ref["sections"]
    return [(c, i) for c in range(secs.shape[0]) for i in range(helpers.n_sections(secs[c]))]


def main():
    import helpers
    from oracle import oracle
    from instance_stixels_amd import host, synthetic
    fns = reference_functions()
    out = {}
    for k, (preset, rows, cols, D, seed, n_slabs, ov) in enumerate(CASES):
        case = helpers.build_case(preset, rows, cols, D, seed=seed, **ov)
        cfg = case["cfg"]
        frame = synthetic.make_frame(cfg, seed=seed, n_slabs=n_slabs, offset_scale=1.0)
        case["frames"] = [frame]
        case["disparity"] = frame.disparity[None]
        case["segmentation"] = frame.segmentation[None]
        ref = helpers.run_oracle(case)
        secs = ref["sections"]
        C, S = secs.shape
        order = sections_of(case, ref)

        st = host.Stixels()
        st.SetConfig(cfg)
        st.PrecomputeHost()
        data = host.StixelsData(secs, rows, cols, C, S, D, 8, 19, frame.alpha_ground,
                                int(case["vhor"][0]))
        with tempfile.TemporaryDirectory() as tmp:
            fa = os.path.join(tmp, "a.stixels")
            st.SaveStixels(data, {}, frame.alpha_ground, int(case["vhor"][0]), fa)
            text_a = open(fa, "rb").read()
            with contextlib.redirect_stdout(io.StringIO()):
                parsed, ground = fns["read_stixel_file"](fa)
                labelled = fns["assign_instances"](
                    parsed, dict(eps=float(cfg.eps), min_size=int(cfg.min_pts),
                                 size_filter=int(cfg.size_filter), use_instance_disparity=""))
            # ---- what the reference reader saw, in file order
            flat = [s for col in parsed for s in col]
            assert len(flat) == len(order) and len(parsed) == C
            rp = np.array([[s["type"], s["vB"], s["vT"], s["class"]] for s in flat], np.int32)
            rf = np.array([[s["disparity"], s["cost"], s["instance_mean_x"], s["instance_mean_y"]]
                           for s in flat], np.float64)
            ref_labels = np.array([s.get("instance_label", -2)
                                   for col in labelled for s in col], np.int32)  # -2: class < 11
            # ---- candidate arrays per instance class as the twin derives them (class, size)
            for cls in range(11, 19):
                m = rp[:, 3] == cls
                if not m.any():
                    continue
                Xc = rf[m][:, 2:4]
                # margin: no pair may sit on the eps boundary (float32 vs float64 evaluation)
                d2 = ((Xc[:, None, :] - Xc[None, :, :]) ** 2).sum(-1)
                margin = np.abs(d2 - float(cfg.eps) ** 2).min() / float(cfg.eps) ** 2
                assert margin > 1e-4, f"case {k} class {cls}: pair on the eps boundary ({margin})"
            # ---- text B: labels of the oracle twin through the product writer
            mapping = {}
            for cls in range(8):
                n = int(ref["inst_per_class"][cls])
                if n == 0:
                    continue
                lab = oracle.cluster_instances(ref["inst_centerofmass"][cls][:n],
                                               ref["inst_core"][cls][:n], cfg.eps, cfg.min_pts)
                for (u, v), l in zip(ref["inst_indices"][cls][:n].tolist(), lab.tolist()):
                    mapping[(u, v)] = l
            fb = os.path.join(tmp, "b.stixels")
            st.SaveStixels(data, mapping, frame.alpha_ground, int(case["vhor"][0]), fb)
            text_b = open(fb, "rb").read()
            with contextlib.redirect_stdout(io.StringIO()):
                parsed_b, ground_b = fns["read_stixel_file"](fb)
            ref_labels_b = np.array([s.get("instance_label", -2) for col in parsed_b for s in col],
                                    np.int32)
        st.close()
        n_inst = int((rp[:, 3] >= 11).sum())
        n_lab = int((ref_labels >= 0).sum())
        print(f"case {k}: {preset} {rows}x{cols}x{D}: {len(flat)} stixels, {n_inst} of an instance "
              f"class, {n_lab} labelled by the reference's assign_instances, "
              f"{len(np.unique(ref_labels[ref_labels >= 0]))} instances")
        out[f"c{k}_sections"] = secs.view(np.int32).reshape(C, S, 8)
        out[f"c{k}_meta"] = np.array([rows, cols, D, seed, n_slabs, int(case["vhor"][0]),
                                      int(cfg.size_filter), int(cfg.min_pts)], np.int32)
        out[f"c{k}_fmeta"] = np.array([cfg.eps, frame.alpha_ground], np.float64)
        out[f"c{k}_preset"] = np.frombuffer(preset.encode(), np.uint8)
        out[f"c{k}_text_a"] = np.frombuffer(text_a, np.uint8)
        out[f"c{k}_text_b"] = np.frombuffer(text_b, np.uint8)
        out[f"c{k}_ref_ints"] = rp
        out[f"c{k}_ref_floats"] = rf
        out[f"c{k}_ref_ground"] = np.array([ground[0], ground[1]], np.float64)
        out[f"c{k}_ref_labels"] = ref_labels
        out[f"c{k}_ref_labels_b"] = ref_labels_b
        out[f"c{k}_mapping"] = np.array([[u, v, l] for (u, v), l in sorted(mapping.items())],
                                        np.int32).reshape(-1, 3)
    out["n_cases"] = np.array(len(CASES), np.int32)
    np.savez_compressed(os.path.join(HERE, "f1_f2_reference_python.npz"), **out)
    print("written", os.path.join(HERE, "f1_f2_reference_python.npz"))


if __name__ == "__main__":
    main()
