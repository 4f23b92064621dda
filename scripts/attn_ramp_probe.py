import json, sys, torch
sys.path.insert(0, ".")
import bench_legs
print(json.dumps(bench_legs.attention_ramp_leg(torch.device("cuda:0")), indent=1))
