"""AddressSanitizer + UBSan builds of the HOST-side native code - SURVEY 5's build note: csrc/sampler.cpp (the per-clip draw loop of
input_data.py:457-514) and, since round 6, the host-side PLANNERS of the C ABI (the layer tables and workspace layouts of net.hip /
net_logmfcc.hip, nn_plan / tn_plan of gemm.hip, the depthwise geometry of dwconv.hip, error.cpp).  GPU sanitizers are not available
on this pool, so the device code is covered by the guard-band tests in tests/test_guards_gpu.py instead."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="no host C++ compiler")
def test_sampler_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "sampler_san")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
           "-fno-omit-frame-pointer", os.path.join(ROOT, "speech_recognition_amd", "csrc", "sampler.cpp"),
           os.path.join(ROOT, "tests", "native", "sampler_sanitize_main.cpp"), "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    if r.returncode != 0 and "asan" in (r.stderr or "").lower() and "cannot find" in r.stderr:
        pytest.skip("libasan not installed")
    assert r.returncode == 0, r.stderr[-2000:]
    run = subprocess.run([exe], capture_output=True, text=True, timeout=120,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1"))
    assert run.returncode == 0, (run.stdout + run.stderr)[-3000:]
    assert "draws ok" in run.stdout


HIPCC = "/opt/rocm/bin/hipcc"
SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-Wno-option-ignored"]
PLANNER_TUS = ["gemm.hip", "net.hip", "net_logmfcc.hip", "dwconv.hip", "error.cpp"]


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
@pytest.mark.timeout(600)
def test_host_side_planners_under_asan_ubsan(tmp_path):
    """The translation units that hold host-side pointer / offset arithmetic are compiled with the sanitizers on their HOST half
    (hipcc ignores -fsanitize for the gfx950 half, which never runs here), linked with the library's other objects and driven,
    without a GPU, by tests/native/planner_sanitize_main.cpp over every net kind, awkward batches (1, 3, 70, 384, 1024, 2048) and
    every pointwise shape: tensors inside their buffers, workspace views inside the reported size and pairwise disjoint in
    training mode, the GEMM planners' row / slab counts, tn_plan's memo from four threads."""
    csrc = os.path.join(ROOT, "speech_recognition_amd", "csrc")
    subprocess.check_call(["make", "-C", csrc, "-j", "8"], stdout=subprocess.DEVNULL)     # the other TUs' regular objects
    procs, objs = [], []
    for tu in PLANNER_TUS:
        obj = str(tmp_path / (os.path.splitext(tu)[0] + ".o"))
        objs.append(obj)
        cmd = [HIPCC, "--offload-arch=gfx950", "-O1", "-g", "-std=c++17", "-fPIC", "-Wno-unused-function"] + SAN + \
              ["-I", os.path.join(ROOT, "include")] + (["-x", "hip"] if tu.endswith(".cpp") else []) + ["-c", os.path.join(csrc, tu), "-o", obj]
        procs.append((tu, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)))
    for tu, p in procs:
        out, err = p.communicate(timeout=500)
        if p.returncode != 0 and "asan" in err.lower() and "cannot find" in err:
            pytest.skip("sanitizer runtime not installed")
        assert p.returncode == 0, (tu, err[-2000:])
    main_o = str(tmp_path / "main.o")
    r = subprocess.run(["/opt/rocm/lib/llvm/bin/clang++", "-O1", "-g", "-std=c++17"] + SAN[:3] + ["-I", os.path.join(ROOT, "include"), "-c",
                        os.path.join(ROOT, "tests", "native", "planner_sanitize_main.cpp"), "-o", main_o], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    sanitized = set(os.path.splitext(t)[0] + ".o" for t in PLANNER_TUS)
    others = [os.path.join(csrc, "build", f) for f in sorted(os.listdir(os.path.join(csrc, "build"))) if f.endswith(".o") and f not in sanitized]
    exe = str(tmp_path / "planner_san")
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-fsanitize=address,undefined", "-Wno-option-ignored", main_o] + objs + others + ["-o", exe],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    run = subprocess.run([exe], capture_output=True, text=True, timeout=300,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1"))
    assert run.returncode == 0, (run.stdout + run.stderr)[-3000:]
    assert "planners ok" in run.stdout
