"""The RANSAC-bound extras of bench.py alone (cfg 1, cfg 1 batched, the sweep): one JSON object per line.  For A/B runs of
ransac.hip on the GPU box without the full bench."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda", 0)
which = sys.argv[1:] or ["ransac_cfg1", "ransac_cfg1_batched", "sweep"]
fns = {"ransac_cfg1": bench.extra_ransac_cfg1, "ransac_cfg1_batched": bench.extra_ransac_cfg1_batched, "sweep": bench.extra_sweep}
for name in which:
    r = fns[name](dev, False)
    r.pop("roofline", None)
    print(json.dumps({name: r}), flush=True)
