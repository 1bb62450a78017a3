"""bench.py control flow with more than one rank.  A 1-GPU box cannot run RCCL across two ranks, so the test hook
KWS_BENCH_ONE_DEVICE puts both ranks on cuda:0 over gloo: every collective of the data-parallel path (gradient
all-reduce inside each step, barriers, the MAX of the timings) and the rank-0-only reporting are exercised exactly as
under torch.distributed.run --nproc-per-node N on an N-GPU node.  (A rank-0-only profiling loop once left the other
ranks out of the per-step all-reduce and hung the job: this test is the guard.)"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_ranks_print_one_json_line():
    env = dict(os.environ, KWS_BENCH_ONE_DEVICE="1", KWS_BENCH_TRACE="240", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29541", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6",
           "--warmup", "2", "--bank", "8192", "--batch", "256"]
    res = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=400)
    assert res.returncode == 0, res.stderr.decode()[-2000:]
    lines = [l for l in res.stdout.decode().splitlines() if l.strip()]
    json_lines = [l for l in lines if l.lstrip().startswith("{")]
    other = [l for l in lines if not l.lstrip().startswith("{")]
    assert len(json_lines) == 1, lines                 # rank 0 prints ONE JSON line, rank 1 nothing
    assert all("bench.py" not in l for l in other), other    # anything else on stdout is the launcher's own banner
    out = json.loads(json_lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 6 and out["warmup"] == 2 and out["scaling"] == "weak"
    assert out["config"]["global_batch"] == 512 and out["config"]["parallelism"] == "dp2"
    assert out["value"] > 0 and out["roofline"] is not None and out["rccl_ranks"] == 2
    pf = out["preflight"]                               # first contact: ones + one gradient-sized all-reduce, under the watchdog
    assert pf["rccl_ranks"] == 2 and pf["grad_allreduce_ok"] is True and pf["grad_allreduce_floats"] == 1191436 and pf["seconds"] < 60
    assert out["cpu_baseline"]["value"] > 0 and out["cpu_baseline"]["kind"] == "port"      # N>1 lines carry it too
    assert out["train_loss_first_last"][0] == out["train_loss_first_last"][0]      # finite
    # the N > 1 line diagnoses itself: the gradient exchange alone, the split-overlap form of the same step, every rank's time
    ar = out["allreduce_only"]
    # the flat gradient buffer: 1,191,433 trainable floats, each tensor padded to 16 bytes
    assert 1191433 <= ar["floats"] <= 1191433 + 64 and ar["bytes"] == 4 * ar["floats"] and ar["us_per_allreduce"] > 0
    assert 0 < ar["pct_of_ms_per_step"] and ar["algorithmic_bus_GBps"] > 0
    sp = out["ab_allreduce_split"]
    assert sp["split_block"] == 6 and sp["ms_per_step"] > 0 and sp["value"] > 0 and 0.3 < sp["vs_one_buffer"] < 3.0
    assert len(out["per_rank_ms"]) == 2 and all(v > 0 for v in out["per_rank_ms"])
    assert max(out["per_rank_ms"]) <= out["ms_per_step"] * 1.001          # the line's time is the MAX over ranks
    assert out["scaling_vs_n1"] is None                                   # no --n1-value given
    assert out["ab_gemm_f16x2"] is None and out["configs"] is None        # single-GPU legs stay out of N > 1 lines
    print("non-JSON stdout lines of the launcher:", other)


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it (how the driver may call it): bench.py starts the two
    ranks itself as child processes BEFORE touching the GPU, relays rank 0's line and reports n_gpus 2 - never a
    silent 1-GPU run.  (--no-cpu-baseline: the other test covers that leg.)"""
    env = dict(os.environ, KWS_BENCH_ONE_DEVICE="1", KWS_BENCH_TRACE="240", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--bank", "8192",
           "--batch", "256", "--no-cpu-baseline", "--allreduce-split", "3", "--n1-value", "100000"]
    res = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=400)
    assert res.returncode == 0, res.stderr.decode()[-2000:]
    lines = [l for l in res.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["config"]["global_batch"] == 512
    assert out["preflight"]["grad_allreduce_ok"] is True
    assert out["ab_allreduce_split"]["split_block"] == 3 and out["ab_allreduce_split"]["value"] > 0     # the argument, not the environment
    assert out["allreduce_only"]["us_per_allreduce"] > 0 and len(out["per_rank_ms"]) == 2
    assert abs(out["scaling_vs_n1"] - out["value"] / (2 * 100000.0)) < 1e-9


def test_bench_four_ranks_on_one_device_rehearsal():
    """VERDICT r5 item 5, within this pool's rules: at most 6 processes may use one card, and this pytest process is one of them,
    so the widest world that may touch the GPU here is FOUR ranks (the host-side world-8 paths - row sharding, the test-set
    partition, gathers of eight, eight sampler seeds - run over gloo on the CPU in tests/test_data_parallel_cpu.py).  bench.py starts
    its own ranks; both all-reduce forms run (the headline's one buffer, `ab_allreduce_split`); one JSON line."""
    env = dict(os.environ, KWS_BENCH_ONE_DEVICE="1", KWS_BENCH_TRACE="240", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "4", "--warmup", "2", "--bank", "512",
           "--batch", "64", "--no-cpu-baseline", "--n1-value", "50000"]
    res = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=500)
    assert res.returncode == 0, res.stderr.decode()[-2000:]
    lines = [l for l in res.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    assert out["n_gpus"] == 4 and out["rccl_ranks"] == 4 and out["preflight"]["rccl_ranks"] == 4
    assert out["config"]["global_batch"] == 256 and out["config"]["parallelism"] == "dp4" and out["scaling"] == "weak"
    assert len(out["per_rank_ms"]) == 4 and all(v > 0 for v in out["per_rank_ms"])
    assert max(out["per_rank_ms"]) <= out["ms_per_step"] * 1.001
    assert out["sampler_seeds"] == [1234, 1235, 1236, 1237]                      # four ranks, four different clip streams
    assert abs(out["scaling_vs_n1"] - out["value"] / (4 * 50000.0)) < 1e-9
    assert out["allreduce_only"]["us_per_allreduce"] > 0 and out["ab_allreduce_split"]["value"] > 0
    assert out["train_loss_first_last"][1] == out["train_loss_first_last"][1]    # finite after the all-reduced steps


def test_launcher_counts_devices_without_the_hip_runtime():
    """The parent of `bench.py --gpus N` must not touch the GPU before it starts its ranks: the count comes from the KFD
    topology in sysfs and agrees with what the runtime reports on this box."""
    code = ("import sys; sys.argv = ['bench.py']; import bench, torch\n"
            "n = bench.visible_gpu_count()\n"
            "assert not torch.cuda.is_initialized()\n"
            "print(n, torch.cuda.device_count())\n")
    res = subprocess.run([sys.executable, "-c", code], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert res.returncode == 0, res.stderr.decode()[-2000:]
    n_sysfs, n_rt = res.stdout.decode().split()[-2:]
    assert n_sysfs == "None" or int(n_sysfs) == int(n_rt) >= 1, (n_sysfs, n_rt)


def test_bench_refuses_more_gpus_than_visible():
    """--gpus 64 on this box must fail, not degrade to the devices that exist."""
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "KWS_BENCH_ONE_DEVICE"):
        env.pop(k, None)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64", "--steps", "1", "--warmup", "0"],
                         cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert res.returncode != 0 and not res.stdout.strip()
    assert b"GPU(s) visible" in res.stderr


def test_tta_inference_sharded_over_two_ranks_matches_one_rank(tmp_path):
    """BASELINE config C5's multi-GPU form (scripts/tta_infer.py -> tta.predict_test_set): the test set is split into
    contiguous ranges by rank, no collective on the data path, results exchanged once at the end.  Two ranks (on the one
    GPU over gloo) must return exactly what one rank returns."""
    import numpy as np
    outs = {}
    for world in (1, 2):
        out = str(tmp_path / ("probs_w%d.npy" % world))
        env = dict(os.environ, KWS_TTA_OUT=out, HSA_ENABLE_IPC_MODE_LEGACY="0")
        for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
            env.pop(k, None)
        script = os.path.join(ROOT, "scripts", "tta_infer.py")
        if world == 1:
            cmd = [sys.executable, script, "5003"]
        else:
            env.update(KWS_ONE_DEVICE="1")
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                   "127.0.0.1", "--master-port", "29549", script, "5003"]
        res = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        assert res.returncode == 0, res.stderr.decode()[-2000:]
        line = [l for l in res.stdout.decode().splitlines() if l.lstrip().startswith("{")][-1]
        info = json.loads(line)
        assert info["n_gpus"] == world and sum(info["argmax_hist"]) == 5003
        outs[world] = np.load(out)
    assert outs[1].shape == (5003, 12)
    assert np.array_equal(outs[1], outs[2])                 # inference is per clip: the split cannot change a bit


@pytest.mark.parametrize("mode", ["hang", "raise", "init"])
def test_bench_preflight_failure_prints_one_error_line_and_fails(mode):
    """the first collectives of an N > 1 run go under a watchdog, before the clip bank is built.  A
    collective that never returns ('hang': the fresh watchdog child reports and kills rank 0) or that raises ('raise': rank 0
    reports itself) must leave exactly ONE JSON line carrying "error" on stdout and a non-zero exit status."""
    env = dict(os.environ, KWS_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0", KWS_BENCH_PREFLIGHT_FAIL=mode,
               KWS_BENCH_PREFLIGHT_TIMEOUT="5")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", {"hang": "29543", "raise": "29545", "init": "29547"}[mode], os.path.join(ROOT, "bench.py"), "--gpus", "2",
           "--steps", "2", "--warmup", "1", "--bank", "8192", "--batch", "256", "--no-cpu-baseline"]
    res = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=400)
    assert res.returncode != 0
    json_lines = [l for l in res.stdout.decode().splitlines() if l.lstrip().startswith("{")]
    assert len(json_lines) == 1, res.stdout.decode()[-2000:] + res.stderr.decode()[-2000:]
    out = json.loads(json_lines[0])
    assert out["value"] is None and out["n_gpus"] == 2
    assert out["stage"] == ("init_process_group" if mode == "init" else "preflight all-reduce")
    assert ("watchdog" in out["error"]) == (mode == "hang")


def test_bench_single_gpu_line_carries_the_round4_keys():
    """The N = 1 line: configs[1]'s own A/B (generator 'raw' vs 'mfcc_and_raw'), and the measured error of the STFT+mel stage
    against the float64 oracle on the run's own clips (VERDICT r3 item 4 ii / iii)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "KWS_BENCH_ONE_DEVICE"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "6", "--warmup", "3", "--bank", "8192", "--batch", "256",
           "--no-val-acc", "--no-configs", "--profile-steps", "1"]
    res = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert res.returncode == 0, res.stderr.decode()[-2000:]
    lines = [l for l in res.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["preflight"] is None and out["value"] > 0
    ab = out["ab_features"]
    assert ab["raw"]["value"] > 0 and ab["mfcc_and_raw"]["value"] > 0 and len(ab["raw"]["rounds_ms"]) == 2
    assert abs(ab["stft_mel_cost_us_per_step"] - 1e3 * (ab["mfcc_and_raw"]["ms_per_step"] - ab["raw"]["ms_per_step"])) < 1e-6
    bp = out["ab_bwd_pair"]                              # one launch for a layer's two backward GEMMs vs two (mode 1)
    assert bp["paired"]["value"] > 0 and bp["separate"]["value"] > 0 and len(bp["paired"]["rounds_ms"]) == 2
    assert out["ab_bwd_pair_error"] is None and "ab_wgrad_beside_dwbwd" not in out      # round 5's third schedule left the library
    assert out["roofline"]["family"] in ("gemm_bwd_pair", "gemm_nn") and out["roofline"]["frac"] > 0
    err = out["stft_mel_error"]
    assert err["clips"] == 16
    assert err["log_mel"]["max_abs_err_vs_f64"] < 1e-3 and err["mfcc"]["max_abs_err_vs_f64"] < 2e-3     # the test-suite bars
    assert err["spectrogram"]["max_abs_err_vs_f64"] < 2e-5 * max(1.0, err["spectrogram"]["max_abs_value"])
