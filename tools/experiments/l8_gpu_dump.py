import sys
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, torch
import bench
dev = torch.device("cuda", 0)
wl = bench.Workload("drn_d_38_pairwise", 1024, 2048, 128, 16, 1, dev, 0)
core = wl.make_core()
wl.step(core); torch.cuda.synchronize()
bs = core.read_block_summaries(3)
np.save("gpurun_out/blksum_col3.npy", bs)
print(bs.shape); print(bs[24:28])
core.close()
