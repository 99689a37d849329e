"""Seeded slices of the randomised differential soaks under tools/ (random sizes, graphs, policies and
hyper-parameters; engine against the oracle, everything compared with ==).  The tools run for as long as one
likes from the command line; here each gets a fixed seed and a few seconds."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tool(name):
    spec = importlib.util.spec_from_file_location("epi_tool_" + name, os.path.join(ROOT, "tools", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("name,seconds,seed,least", [("soak", 14.0, 11, 5), ("soak_world", 14.0, 12, 20),
                                                     ("soak_train", 12.0, 13, 20), ("soak_learn", 12.0, 14, 50),
                                                     ("soak_sweep_learn", 12.0, 15, 10)])
def test_soak_slice(name, seconds, seed, least):
    import epidemicsimulator_amd as E
    assert E.lib().epi_device_count() >= 1, "no HIP device: GPU tests must run on the MI355X box"
    cases = _tool(name).run(seconds, seed, verbose=False)
    assert cases >= least, "%s ran only %d cases in %.0f s" % (name, cases, seconds)
