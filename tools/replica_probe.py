"""bench.py's config3_replicas section on its own: R replicas of the streaming-odometry loop in lock-step on one GPU."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import torch  # noqa: E402

import randt_slam_amd as R  # noqa: E402

bench.load_counters()
ctx = R.Context(0, torch.cuda.current_stream().cuda_stream)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 150
counts = tuple(int(v) for v in sys.argv[2].split(",")) if len(sys.argv) > 2 else (1, 16, 64, 256)
print(json.dumps(bench.config3_replicas(ctx, steps, counts), indent=1))
