"""GPU box (dev): test_split_kernel_operand_ranges under other seeds of its inputs — how far does the max-statistic of an
ill-conditioned case move from seed to seed, per kernel build?   python tools/dev/range_seeds.py 10.0 1.0 [seeds...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from tests import test_gpu_parity as T
dev = torch.device("cuda", 0)
w, z = float(sys.argv[1]), float(sys.argv[2])
for seed in [int(a) for a in sys.argv[3:]] or [11, 12, 13, 14]:
  try:
    print(T._operand_range_case(dev, z, w, seed))
    print("  seed %d: within the bound" % seed)
  except AssertionError as e:
    print("  seed %d: OUTSIDE the bound" % seed)
