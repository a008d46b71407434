import sys, json
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
print(json.dumps(bench.extra_host_tier(torch.device("cuda", 0), False), indent=1))
