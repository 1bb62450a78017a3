"""Host-side pieces of bench.py that need no GPU: the preflight watchdog of N > 1 runs."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(code, **env):
    e = dict(os.environ, **env)
    return subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)


def test_watchdog_reports_and_kills_a_rank_that_hangs_in_its_first_collective():
    code = ("import sys, time; sys.argv = ['bench.py']; import bench\n"
            "dog = bench.PreflightWatchdog(sys.stdout, 8)\n"
            "dog.say('start')\n"
            "time.sleep(60)\n"
            "print('not reached')\n")
    res = _run(code, KWS_BENCH_PREFLIGHT_TIMEOUT="1")
    assert res.returncode == -9                         # SIGKILL from the fresh child, never an exec of the hung process
    lines = [l for l in res.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    assert out["value"] is None and out["n_gpus"] == 8 and out["stage"] == "preflight all-reduce" and "watchdog" in out["error"]


def test_watchdog_covers_the_rendezvous_too_and_is_silent_when_all_is_well():
    hang_in_init = ("import sys, time; sys.argv = ['bench.py']; import bench\n"
                    "dog = bench.PreflightWatchdog(sys.stdout, 2)\n"
                    "time.sleep(60)\n")
    res = _run(hang_in_init, KWS_BENCH_INIT_TIMEOUT="1")
    assert res.returncode == -9 and json.loads(res.stdout.decode().strip())["stage"] == "init_process_group"
    fine = ("import sys, time; sys.argv = ['bench.py']; import bench\n"
            "dog = bench.PreflightWatchdog(sys.stdout, 2)\n"
            "dog.say('start'); time.sleep(0.2); dog.done()\n"
            "time.sleep(1.5)\n")                        # well past the limit: the watchdog has gone
    res = _run(fine, KWS_BENCH_PREFLIGHT_TIMEOUT="1")
    assert res.returncode == 0 and not res.stdout.strip(), (res.stdout, res.stderr)
    dies_by_itself = ("import sys; sys.argv = ['bench.py']; import bench\n"
                      "dog = bench.PreflightWatchdog(sys.stdout, 2)\n"
                      "raise SystemExit(5)\n")          # the rank ends on its own (it printed its own reason): pipe closes, no line
    res = _run(dies_by_itself, KWS_BENCH_PREFLIGHT_TIMEOUT="1")
    assert res.returncode == 5 and not res.stdout.strip()
