"""The C-ABI library loads and exports every symbol include/jsdr_hip.h declares (no compute, no GPU)."""
import os

import java_sdr_amd as J


def test_library_is_built_in_tree():
    assert os.path.exists(J.library_path()), "run `python java-sdr_amd/build.py` (or __graft_entry__.build())"
    assert os.path.dirname(J.library_path()).endswith("java-sdr_amd")


def test_every_declared_symbol_is_exported():
    lib = J.lib()
    assert len(J.EXPORTED_SYMBOLS) >= 40
    missing = [s for s in J.EXPORTED_SYMBOLS if not hasattr(lib, s)]
    assert not missing, missing


def test_version_and_error_string():
    lib = J.lib()
    assert lib.jsdr_version() == 1
    assert isinstance(lib.jsdr_last_error(), (bytes, type(None)))


def test_compute_fails_loudly_without_a_device():
    """No CPU fallback: on a box without a GPU every compute entry point must raise, never return data."""
    import numpy as np
    import pytest
    if J.have_gpu():
        pytest.skip("GPU present")
    with pytest.raises(J.JsdrError):
        J.Fft(2048, 96000)
    with pytest.raises(J.JsdrError):
        J.fec_decode(np.zeros(5200, np.uint8))
    with pytest.raises(J.JsdrError):
        J.Bpsk()
    with pytest.raises(J.JsdrError):
        J.Group(1, 4, 4096)
    with pytest.raises(J.JsdrError):
        J.Group(2, 4, 4096, devices=[0, 0], gather_copy=True)


def test_group_arguments_are_checked_before_any_device_work():
    import pytest
    with pytest.raises(J.JsdrError, match="split evenly"):
        J.Group(3, 8, 4096, gather_copy=True)
    with pytest.raises(J.JsdrError):
        J.Group(0, 8, 4096)
