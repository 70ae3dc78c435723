"""Shared fixture: run the facade tests against the HIP backend (-m gpu) and against the TEST-ONLY 64-lane
host emulation of the same core (CPU suite)."""
import pytest

BACKENDS = [pytest.param("emu", id="emu"), pytest.param("hip", id="hip", marks=pytest.mark.gpu)]


@pytest.fixture(params=BACKENDS)
def facade(request):
    import azul_deep_reinforcement_learning_amd.facade_backend as fb
    saved = fb._FACTORY
    if request.param == "emu":
        from tests.hostcheck import hostcheck as hc
        fb._FACTORY = hc.call_backend_class()      # the product's host logic on the emulated device core
    else:
        fb._FACTORY = fb.HipBackend
    import azul_deep_reinforcement_learning_amd as pkg
    yield pkg
    fb._FACTORY = saved
