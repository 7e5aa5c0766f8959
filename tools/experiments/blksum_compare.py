"""Block summaries (lemmas L7 / L8) of two builds of libis_core.so, bit for bit (run on the GPU box):
    python tools/blksum_compare.py instance_stixels_amd/lib/libis_core_ref.so
dumps the summaries of a few columns of every input family with the in-tree library and with the given
one (a child process each: the library is chosen at import, IS_CORE_LIB) and compares the raw words."""
import os
import subprocess
import sys

sys.path.insert(0, ".")
COLS = (0, 3, 17, 100, 185, 235, 255, 256 + 31, 256 * 3 + 77)


def dump(path):
    import numpy as np
    import torch
    import bench
    from instance_stixels_amd.synthetic import FAMILIES
    dev = torch.device("cuda", 0)
    out = []
    for fam in FAMILIES:
        wl = bench.Workload("drn_d_38_pairwise", 1024, 2048, 128, 4, 2, dev, 0, family=fam)
        core = wl.make_core()
        wl.step(core)
        torch.cuda.synchronize()
        out.append(np.stack([core.read_block_summaries(c) for c in COLS]))
        core.close()
    np.save(path, np.stack(out))


if __name__ == "__main__":
    if sys.argv[1] == "dump":
        dump(sys.argv[2])
        sys.exit(0)
    import numpy as np
    os.makedirs("gpurun_out", exist_ok=True)
    paths = []
    for tag, lib in (("tree", None), ("other", sys.argv[1])):
        env = dict(os.environ)
        if lib:
            env["IS_CORE_LIB"] = os.path.abspath(lib)
        p = f"gpurun_out/blksum_{tag}.npy"
        subprocess.run([sys.executable, __file__, "dump", p], check=True, env=env)
        paths.append(p)
    a, b = (np.load(p).view(np.uint32) for p in paths)
    print("shape", a.shape, "identical words:", int((a == b).sum()), "of", a.size)
    if not (a == b).all():
        idx = np.argwhere(a != b)
        print("first differences (family, column, block, float):", idx[:10].tolist())
        sys.exit(1)
