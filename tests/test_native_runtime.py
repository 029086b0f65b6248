"""The torch-free host runtime (csrc/amdsmi_sampler.cpp): the sampler thread and its lock-free
ring through ctypes, and the sanitizer builds of `make sanitize`."""

import os
import shutil
import subprocess
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_native_sampler_thread_with_a_synthetic_source():
    from kvmini.telemetry.native_rt import NativeSampler, load

    if load() is None:
        pytest.skip("libkvmini_rt.so not built")
    s = NativeSampler(interval_s=0.002, capacity=256, synthetic=2)
    assert s.working()
    s.start()
    with pytest.raises(RuntimeError):
        NativeSampler(interval_s=0.01, synthetic=1).start()  # one sampler thread per process
    time.sleep(0.25)
    last = s.latest()
    assert last is not None and len(last["per_gpu"]) in (1, 2)
    samples = s.stop()
    assert 40 <= len(samples) <= 256
    ts = [x["ts_s"] for x in samples]
    assert ts == sorted(ts)
    assert all(g["watts"] >= 100.0 for x in samples for g in x["per_gpu"]) and all(len(x["per_gpu"]) == 2 for x in samples[1:-1])
    assert s.to_power_json()["source"].startswith("synthetic")
    # the ring holds the NEWEST records once the thread has lapped it
    s2 = NativeSampler(interval_s=0.001, capacity=64, synthetic=1).start()
    time.sleep(0.2)
    got = s2.stop()
    assert len(got) <= 64 and got[-1]["ts_s"] > 64


@pytest.mark.parametrize("flags,env", [("-fsanitize=thread", {"TSAN_OPTIONS": "halt_on_error=1"}), ("-fsanitize=address,undefined -fno-sanitize-recover=all", {"ASAN_OPTIONS": "detect_leaks=1"})])
def test_ring_buffer_stress_under_sanitizers(tmp_path, flags, env):
    """What `make sanitize` runs, shortened: writer thread laps a 64-slot ring while two readers
    copy it out; no torn record, no report from ThreadSanitizer / AddressSanitizer / UBSan."""
    cxx = os.environ.get("CXX", "g++")
    if shutil.which(cxx) is None:
        pytest.skip("no C++ compiler")
    exe = tmp_path / "ring_stress"
    cmd = [cxx, "-std=c++17", "-O1", "-g", *flags.split(), os.path.join(ROOT, "csrc/amdsmi_sampler.cpp"), os.path.join(ROOT, "csrc/tests/ring_stress.cpp"), "-lpthread", "-ldl", "-o", str(exe)]
    build = subprocess.run(cmd, capture_output=True, text=True)
    if build.returncode != 0 and ("cannot find" in build.stderr or "unrecognized" in build.stderr):
        pytest.skip(f"sanitizer runtime not installed: {build.stderr.splitlines()[-1]}")
    assert build.returncode == 0, build.stderr
    run = subprocess.run([str(exe), "0.9"], capture_output=True, text=True, env={**os.environ, **env}, timeout=120)
    assert run.returncode == 0, run.stdout + run.stderr
    assert "0 bad" in run.stdout and "WARNING: ThreadSanitizer" not in run.stderr and "ERROR: AddressSanitizer" not in run.stderr and "runtime error" not in run.stderr


def test_makefile_has_the_sanitize_target():
    mk = open(os.path.join(ROOT, "Makefile")).read()
    assert "sanitize:" in mk and "-fsanitize=thread" in mk and "-fsanitize=address,undefined" in mk and "ring_stress" in mk
