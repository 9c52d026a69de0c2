import sys, os
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "video-flow-ml_amd"))
os.environ["MB_MFMA"] = "1"; os.environ["VFML_DIRECT_MIN"] = "64"
import conv_microbench as mb
for cout in (128, 256, 512):
    mb.bench(f"readout 32400 x 32448 -> {cout}", 1, 1, 32400, 32448, cout, 1, 1, reps=5, s16=True, gemm=True)
