"""bench.py's config5_polar_filter section on its own (16 scans per launch, ONE scan per launch, 64 scans per launch)."""
import json, sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
import randt_slam_amd as R
ctx = R.Context(0, torch.cuda.current_stream().cuda_stream)
bench.FILTER_ROWS = {}
print(json.dumps(bench.polar_filter(ctx, 16), indent=1))
