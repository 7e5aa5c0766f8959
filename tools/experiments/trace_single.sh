#!/bin/bash
# GPU box: kernel trace of the single-frame host-class path (Stixels::Compute + GetInstanceStixels).
set -u
OUT=gpurun_out/trace_single
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cat > $OUT/run.py <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from instance_stixels_amd import make_config, synthetic, host
preset = sys.argv[1] if len(sys.argv) > 1 else "drn_d_22_unary"
cfg = make_config(preset, 1024, 2048, 128)
f = synthetic.make_frame(cfg, seed=17)
st = host.Stixels(); st.SetConfig(cfg); st.Initialize()
st.SetDisparityImage(f.disparity); st.SetSegmentation(f.segmentation)
st.SetRoadParameters(f.vhor_image, f.camera_tilt, f.camera_height, f.alpha_ground)
print("compute", st.time_compute(cfg.pairwise, 50, False) * 1e3, "ms; with GetInstanceStixels", st.time_compute(cfg.pairwise, 50, True) * 1e3)
st.close()
PY
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT -- python3 $OUT/run.py ${1:-drn_d_22_unary} > $OUT/log.txt 2>&1
tail -2 $OUT/log.txt
python3 - <<PY
import csv, glob
rows = []
for f in glob.glob("$OUT/**/*kernel_trace.csv", recursive=True):
    rows += [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40]) for r in csv.DictReader(open(f))]
for f in glob.glob("$OUT/**/*memory_copy_trace.csv", recursive=True):
    rows += [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "")[:30]) for r in csv.DictReader(open(f))]
rows.sort()
# the last frame: from the last k_join_columns start
idx = max(i for i, r in enumerate(rows) if "k_join" in r[2])
prev = max(i for i, r in enumerate(rows[:idx]) if "k_join" in r[2])
t0 = rows[prev][0]
for s, e, n in rows[prev:idx]:
    print(f"{(s - t0) / 1e3:8.1f} us  +{(e - s) / 1e3:7.1f} us  {n}")
print("frame period", (rows[idx][0] - t0) / 1e3, "us")
PY
