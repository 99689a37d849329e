"""Runs the C++ restatement of the reference's ScalaTest specs (tests/cpp/specs.cpp) against the C++
mirror of the Scala API (include/epidemic.hpp) on the GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "cpp", "specs")


@pytest.mark.gpu
def test_cpp_specs_pass_on_gpu():
    assert os.path.exists(BIN), "tests/cpp/specs is built by __graft_entry__.build()"
    res = subprocess.run([BIN], capture_output=True, text=True, timeout=300)
    print(res.stdout)
    assert res.returncode == 0, res.stdout + res.stderr
    for name in ("ValidationSpec", "MobilitySpec", "CalibrationSpec", "EvalDeterminismSpec", "HeterogeneitySpec"):
        assert "PASS " + name in res.stdout


def test_cpp_specs_binary_is_built_and_refuses_without_gpu():
    import epidemicsimulator_amd as E
    assert os.path.exists(BIN), "tests/cpp/specs is built by __graft_entry__.build()"
    if E.lib().epi_device_count() > 0:
        pytest.skip("a HIP device is present")
    res = subprocess.run([BIN], capture_output=True, text=True, timeout=60)
    assert res.returncode == 2 and "no HIP device" in res.stdout
