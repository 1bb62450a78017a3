"""rocprofv3 target: training steps of config C3 (scripts/bench_configs.py).
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c3 -- python3 scripts/prof_c3.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_configs  # noqa: E402

print(bench_configs.c3(steps=30, warm=5, profile_steps=0))
