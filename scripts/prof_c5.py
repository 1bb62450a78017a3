"""rocprofv3 target: TTA / plain inference batches of config C5 (scripts/bench_configs.py).
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c5 -- python3 scripts/prof_c5.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_configs  # noqa: E402

print(bench_configs.c5(speed_tta=False, n=10, warm=3))
