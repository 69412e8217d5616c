"""CPU checks of device-side index logic through a g++ build of the shared inline headers."""
import ctypes
import pathlib
import subprocess

import numpy as np
import pytest

HERE = pathlib.Path(__file__).resolve().parent


@pytest.fixture(scope='module')
def emu(tmp_path_factory):
    so = tmp_path_factory.mktemp('emu') / 'libhost_emu.so'
    subprocess.check_call(['g++', '-O2', '-shared', '-fPIC', '-std=c++17', str(HERE / 'host' / 'host_emu.cpp'), '-o', str(so)])
    return ctypes.CDLL(str(so))


def test_fft_core_matches_numpy_rfft(emu):
    rng = np.random.default_rng(0)
    x = rng.standard_normal(2048).astype(np.float32)
    n = np.arange(2048)
    win = (0.5 - 0.5 * np.cos(2 * np.pi * n / 2048)).astype(np.float32)
    mag = np.zeros(1025, dtype=np.float32)
    fp = ctypes.POINTER(ctypes.c_float)
    emu.emu_rfft_mag(x.ctypes.data_as(fp), win.ctypes.data_as(fp), mag.ctypes.data_as(fp))
    ref = np.abs(np.fft.rfft(x.astype(np.float64) * win.astype(np.float64)))
    assert np.max(np.abs(mag - ref)) < 2e-4 * np.max(ref)
    np.testing.assert_allclose(mag, ref, rtol=0, atol=1e-5 * np.max(ref))
