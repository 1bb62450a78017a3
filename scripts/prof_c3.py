"""rocprofv3 target: 20 training steps of config C3 (see scripts/bench_configs.py)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_configs  # noqa: E402

bench_configs.c3()
