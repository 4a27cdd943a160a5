import json, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, torch
import randt_slam_amd as R
ctx = R.Context(0, torch.cuda.current_stream().cuda_stream)
print(json.dumps(bench.slam_loop(ctx, 300), indent=1))
