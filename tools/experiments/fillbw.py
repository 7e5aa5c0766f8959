import torch, time
x = torch.empty(int(8.6e9)//4, dtype=torch.float32, device="cuda")
for _ in range(2): x.fill_(1.0)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): x.fill_(2.0)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
print("fill 8.6 GB: %.3f ms = %.2f TB/s" % (ms, x.numel() * 4 / ms / 1e9))
y = torch.empty_like(x)
for _ in range(2): y.copy_(x)
torch.cuda.synchronize()
e0.record()
for _ in range(5): y.copy_(x)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
print("copy 8.6 GB: %.3f ms = %.2f TB/s (read+write)" % (ms, 2 * x.numel() * 4 / ms / 1e9))
