"""cfg 2's descriptor variant once (bench.py's extra_get_matches) for rocprofv3."""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
print(json.dumps({k: v for k, v in bench.extra_get_matches(torch.device("cuda", 0), False).items() if k != "roofline"}))
