"""AlignPoints_KNN batched on the device tier: B supports x n points in one launch, HIP-event time (bench.py's extra_align shape)."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pcreg_amd._lib import check, lib
B, n = (int(a) for a in (sys.argv[1:3] if len(sys.argv) > 2 else (4096, 3000)))
dev = torch.device("cuda", 0)
rng = np.random.default_rng(3)
A = np.linalg.qr(rng.normal(size=(3, 3)))[0]
sup = (rng.normal(size=(B * n, 3)) * np.array([3.0, 1.5, 0.4])) @ A + rng.uniform(-50, 50, 3)
pts = torch.from_numpy(np.ascontiguousarray(sup.T)).to(dev)
off = torch.arange(0, (B + 1) * n, n, dtype=torch.int32, device=dev)
al = torch.empty_like(pts); co = torch.empty(9 * B, dtype=torch.float64, device=dev); c = torch.empty(3 * B, dtype=torch.float64, device=dev)
st = torch.empty(B, dtype=torch.int32, device=dev)
L = lib(); p = lambda t: C.c_void_p(t.data_ptr())
shape = int(os.environ.get("ALIGN_SHAPE", "0"))             # this script's own knob -> pcreg_debug_set("align_shape", .)
check(L.pcreg_debug_set(b"align_shape", shape))
def run():
    check(L.pcreg_dev_align_points_knn_batched(p(pts), B * n, B * n, p(off), B, n, 0, 0, p(al), p(co), p(c), p(st), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
for _ in range(2): run()
torch.cuda.synchronize(); ts = []
for _ in range(int(os.environ.get("REPS", "5"))):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); run(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
print(f"shape {shape}: B={B} n={n} min {min(ts):.4f} ms  {48.0 * B * n / min(ts) / 1e6:.0f} GB/s ({48.0 * B * n / min(ts) / 1e6 / 8000:.3f} of 8 TB/s)", flush=True)
