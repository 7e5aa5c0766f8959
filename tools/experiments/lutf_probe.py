"""IS_LUT_FUSED=1 (GPU box): polls of waiting DP workgroups and the average life of a fused LUT unit."""
import os, sys
os.environ.setdefault("IS_LUT_FUSED", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
dev = torch.device("cuda", 0)
wl = bench.Workload("drn_d_22_unary", 1024, 2048, 128, 64, 8, dev, 0)
core = wl.make_core()
wl.step(core); torch.cuda.synchronize()
core.set_eval_counters(True)
wl.step(core)
c = core.eval_counters()
units = 64 * 256 * 2
print("spins", c["lutf_spins"], "unit cycles avg", c["lutf_unit_cycles"] / units, "=", c["lutf_unit_cycles"] / units / 2.1e3, "us at 2.1 GHz")
