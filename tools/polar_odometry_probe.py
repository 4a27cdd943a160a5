"""bench.py's config5_polar_odometry section on its own (device-resident Python loop + the C++ leg from host buffers)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import torch  # noqa: E402

import randt_slam_amd as R  # noqa: E402

ctx = R.Context(0, torch.cuda.current_stream().cuda_stream)
print(json.dumps(bench.polar_odometry(ctx, int(sys.argv[1]) if len(sys.argv) > 1 else 60), indent=1))
