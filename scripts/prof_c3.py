"""rocprofv3 target: training steps of config C3 (scripts/bench_configs.py), the DEFAULT schedule only
(no A/B leg inside the profiled process: every launch in the stats belongs to gemm mode 0).
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c3 -- python3 scripts/prof_c3.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_configs  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
print(bench_configs.c3(steps=steps, warm=5, profile_steps=0, ab=False))
