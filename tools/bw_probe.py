"""GPU box: reference streaming bandwidth (torch copy) to calibrate the encoder's HBM fractions."""
import torch, time
x = torch.empty(256 * 1024 * 1024, dtype=torch.float32, device="cuda")  # 1 GiB
y = torch.empty_like(x)
for n in (2**20, 2**24, 2**26, 2**28):
  a, b = x[:n], y[:n]
  for _ in range(3): b.copy_(a)
  torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(10): b.copy_(a)
  e1.record(); torch.cuda.synchronize()
  ms = e0.elapsed_time(e1) / 10
  print("copy %5d MiB: %.1f us  -> %.2f TB/s (read+write)" % (n * 4 >> 20, ms * 1e3, 2 * n * 4 / ms / 1e9))
