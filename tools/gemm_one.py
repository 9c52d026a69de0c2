"""The level-0 correlation GEMM of a 1080p pair (32400 x 32400 x 256, plain f32 out), timed (dev tool, GPU only)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from conv_microbench import bench
bench("gemm 32400x32400x256", 1, 1, 32400, 256, 32400, 1, 1, reps=5, s16=True, gemm=True)
bench("gemm 32400x8040x256", 1, 1, 32400, 256, 8040, 1, 1, reps=5, s16=True, gemm=True)
