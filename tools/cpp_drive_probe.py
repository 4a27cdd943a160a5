"""bench.py's cpp_local_fuser_drive section on its own (the C++ drop-in drive from host buffers: ms per scan + allocator /
synchronisation counters per scan), beside the Python resident loop on the same drive."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    import torch

    import randt_slam_amd as R

    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    ctx = R.Context(0, torch.cuda.current_stream().cuda_stream)
    py = bench.streaming_odometry(ctx, n, False)
    out = bench.cpp_local_fuser_drive(ctx, n, py["ms_per_scan"])
    out["python_resident_loop"] = py
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
