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
    subprocess.check_call(['g++', '-O2', '-ffp-contract=off', '-shared', '-fPIC', '-std=c++17', str(HERE / 'host' / 'host_emu.cpp'), '-o', str(so)])
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


@pytest.mark.parametrize('frame_length,hop,n', [(3528, 882, 44100 * 3 + 17), (2048, 512, 20000), (100, 30, 1000), (7, 3, 50),
                                                (129, 64, 700), (3528, 882, 1000)])
def test_rms_core_is_bitwise_numpy(emu, frame_length, hop, n):
    """The summation order the device RMS kernel uses (rms_core.h) reproduces numpy's float32 pairwise reduction,
    i.e. get_rms (utils/slicer2.py:5-38), bit for bit - silence decisions compare and argmin these values."""
    from some_amd.utils.slicer2 import get_rms
    rng = np.random.default_rng(frame_length + n)
    y = (rng.standard_normal(n) * 0.05).astype(np.float32)
    y[n // 3: n // 2] *= np.float32(1e-3)
    ref = get_rms(y, frame_length=frame_length, hop_length=hop)[0]
    out = np.zeros(1 + n // hop, dtype=np.float32)
    fp = ctypes.POINTER(ctypes.c_float)
    emu.emu_slicer_rms(y.ctypes.data_as(fp), ctypes.c_int64(n), frame_length, hop, out.ctypes.data_as(fp))
    assert out.shape == ref.shape
    assert np.array_equal(out.view(np.uint32), ref.view(np.uint32))
