"""The product's line-FFT core (datum_amd/csrc/ocean_fft_core.h: plans, butterflies, index maps, LDS padding /
swizzle) run on the CPU by emulating the threads of a line one after another, phase by phase
(tests/cpu/fft_core_emul.cpp), against numpy.  Tolerance: fp32 Stockham, max error < 1e-6 of the output scale."""

import ctypes
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def emul():
    lib = ctypes.CDLL(os.path.join(ROOT, "tests", "cpu", "libfft_core_emul.so"))
    return lib


def _check(got, x, N):
    z = x[:, 0].astype(np.float64) + 1j * x[:, 1]
    want = np.fft.ifft(z) * N
    err = np.abs((got[:, 0] + 1j * got[:, 1]) - want).max() / np.abs(want).max()
    assert err < 1e-6, err


@pytest.mark.parametrize("N", [64, 128, 256, 512, 1024, 2048, 4096])
def test_line_transform_all_plans(emul, N):
    rng = np.random.default_rng(N)
    for trial in range(3):
        x = rng.standard_normal((N, 2)).astype(np.float32)
        out = np.empty_like(x)
        assert emul.emul_line_ifft(N, x.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p)) == 0
        _check(out, x, N)


def test_line_transform_impulses(emul):
    # a unit impulse at k gives exp(+2 pi i k n / N): catches index-map mistakes that random data can hide
    N = 1024
    for k in (0, 1, 7, 8, 63, 64, 129, 511, 512, 1023):
        x = np.zeros((N, 2), np.float32)
        x[k, 0] = 1
        out = np.empty_like(x)
        assert emul.emul_line_ifft(N, x.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p)) == 0
        n = np.arange(N)
        want = np.exp(2j * np.pi * ((k * n) % N) / N)
        assert np.abs((out[:, 0] + 1j * out[:, 1]) - want).max() < 2e-6, k


@pytest.mark.parametrize("N", [256, 512, 1024, 2048, 4096])
def test_line_transform_sixteen_points_per_thread(emul, N):
    # the plans with radix 16 (4096 = 16^3 is what the column pass of the largest grid runs: ColCfg::E)
    rng = np.random.default_rng(16 * N)
    x = rng.standard_normal((N, 2)).astype(np.float32)
    out = np.empty_like(x)
    assert emul.emul_line_ifft16(N, x.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p)) == 0
    _check(out, x, N)
    for k in (1, 15, 16, 255, N // 2, N - 1):
        x = np.zeros((N, 2), np.float32)
        x[k, 0] = 1
        assert emul.emul_line_ifft16(N, x.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p)) == 0
        n = np.arange(N)
        want = np.exp(2j * np.pi * ((k * n) % N) / N)
        assert np.abs((out[:, 0] + 1j * out[:, 1]) - want).max() < 2e-6, k


@pytest.mark.parametrize("E,W", [(8, 2), (8, 4), (8, 8), (16, 2), (16, 4)])
@pytest.mark.parametrize("N", [256, 512, 1024, 2048, 4096])
def test_line_transform_interleaved_columns(emul, N, E, W):
    # the column pass keeps its W columns element by element in one LDS array (position * W + column) and the exchange
    # layouts depend on W (ocean_fft_core.h, "LDS layout of the exchanges"): every (radix, W) the kernels instantiate
    rng = np.random.default_rng(N * E + W)
    x = rng.standard_normal((N, 2)).astype(np.float32)
    out = np.empty_like(x)
    assert emul.emul_line_ifft_w(N, E, W, x.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p)) == 0
    _check(out, x, N)
    for k in (1, E - 1, E, (E * E + 1) % N, N // 2 + 3, N - 1):
        x = np.zeros((N, 2), np.float32)
        x[k, 0] = 1
        assert emul.emul_line_ifft_w(N, E, W, x.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p)) == 0
        n = np.arange(N)
        want = np.exp(2j * np.pi * ((k * n) % N) / N)
        assert np.abs((out[:, 0] + 1j * out[:, 1]) - want).max() < 2e-6, k
