"""CPU: the gfx950 fit kernel's own arithmetic (csrc/fit_core.hpp, csrc/smpl_host.hpp), run lane-by-lane on
the host (tests/emul/fit_emul.cpp), against the autograd + torch.optim.Adam oracle (oracle/fit.py) and the
golden trajectories.  This is a check of the kernel source, not a product code path."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest
import torch

from soccerplayershapepose_amd import _abi
from oracle import smpl as OS, fit as OF

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope='module')
def emul():
    out = os.path.join(HERE, 'emul', '_build')
    os.makedirs(out, exist_ok=True)
    so = os.path.join(out, 'libfit_emul.so')
    subprocess.check_call(['g++', '-O2', '-std=c++17', '-shared', '-fPIC', '-o', so,
                           os.path.join(HERE, 'emul', 'fit_emul.cpp')])
    lib = C.CDLL(so)
    fp = C.POINTER(C.c_float)
    lib.fit_emulate.restype = C.c_int
    lib.fit_emulate.argtypes = [C.POINTER(_abi.SmplArrays), C.c_int, C.c_double, C.c_double, C.c_double, C.c_double,
                                C.c_uint, C.c_float, C.c_float, C.c_double, C.c_int, fp, fp, fp, fp, C.c_int, fp, fp, fp]
    return lib


def run_emul(lib, model, R, betas, cam, j2d, *, iters, lr, mask, select=0):
    arrs, keep = _abi.smpl_arrays(model)
    B = R.shape[0]
    R = np.ascontiguousarray(R.reshape(B, 216), np.float32).copy()
    betas = np.ascontiguousarray(betas, np.float32).copy()
    cam = np.ascontiguousarray(cam, np.float32).copy()
    j = np.ascontiguousarray(j2d[:, :, :2], np.float32)
    loss = np.zeros((B, iters), np.float32)
    l2e = np.zeros((B, iters), np.float32)
    grad = np.zeros((B, 229), np.float32)
    p = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    rc = lib.fit_emulate(C.byref(arrs), iters, lr, 0.9, 0.999, 1e-8, mask, 512.0, 256.0, 1.0, select,
                         p(R), p(betas), p(cam), p(j), B, p(loss), p(l2e), p(grad))
    assert rc == 0
    return R.reshape(B, 24, 3, 3), betas, cam, loss, l2e, grad


def test_first_gradient_matches_autograd(emul, golden, smpl_model_np):
    g = golden('fit')
    _, _, _, loss, l2e, grad = run_emul(emul, smpl_model_np, g['R'], g['betas'], g['cam'], g['joints2D'],
                                        iters=1, lr=1e-3, mask=_abi.FIT_MASK_POSE_SHAPE)
    for i in range(3):
        assert abs(loss[i, 0] - g[f'B{i}_loss'][0]) < 2e-6 * max(1, abs(g[f'B{i}_loss'][0]))
        assert abs(l2e[i, 0] - g[f'B{i}_l2e'][0]) < 1e-3
        np.testing.assert_allclose(grad[i, :3], g[f'B{i}_grad_cam'], rtol=2e-4, atol=2e-7)
        np.testing.assert_allclose(grad[i, 3:12].reshape(3, 3), g[f'B{i}_grad_G'].reshape(3, 3), rtol=2e-4, atol=2e-7)
        gb = grad[i, 12:219].reshape(23, 9)
        opt = np.concatenate([gb[:6], gb[8:21]]).reshape(19, 3, 3)
        np.testing.assert_allclose(opt, g[f'B{i}_grad_body'], rtol=5e-4, atol=3e-7)
        np.testing.assert_allclose(grad[i, 219:], g[f'B{i}_grad_betas'], rtol=5e-4, atol=3e-7)
        assert np.all(grad[i, 3 + 6:3 + 9] == 0)      # z-row of G: exactly zero (Appendix D)


def test_variant_a_trajectory(emul, golden, smpl_model_np):
    g = golden('fit')
    keep = list(g['keep_iters'])
    for it_idx, it in enumerate(keep):
        R, betas, cam, loss, l2e, _ = run_emul(emul, smpl_model_np, g['R'], g['betas'], g['cam'], g['joints2D'],
                                               iters=it + 1, lr=0.01, mask=_abi.FIT_MASK_CAMERA)
        for i in range(3):
            np.testing.assert_allclose(cam[i], g[f'A{i}_cam'][it_idx], atol=5e-5)
            np.testing.assert_allclose(R[i, 0], g[f'A{i}_G'][it_idx], atol=5e-5)
            np.testing.assert_array_equal(R[i, 1:], g['R'][i, 1:])          # body pose untouched
            np.testing.assert_array_equal(betas[i], g['betas'][i])
            np.testing.assert_allclose(loss[i], g[f'A{i}_loss'][:it + 1], rtol=5e-3, atol=2e-7)


def test_variant_b_trajectory_and_frozen_joints(emul, golden, smpl_model_np):
    g = golden('fit')
    keep = list(g['keep_iters'])
    for it_idx, it in enumerate(keep):
        R, betas, cam, loss, l2e, _ = run_emul(emul, smpl_model_np, g['R'], g['betas'], g['cam'], g['joints2D'],
                                               iters=it + 1, lr=1e-3, mask=_abi.FIT_MASK_POSE_SHAPE)
        for i in range(3):
            np.testing.assert_allclose(cam[i], g[f'B{i}_cam'][it_idx], atol=5e-5)
            np.testing.assert_allclose(R[i, 0], g[f'B{i}_G'][it_idx].reshape(3, 3), atol=5e-5)
            np.testing.assert_allclose(R[i, 1:], g[f'B{i}_body'][it_idx], atol=5e-5)
            np.testing.assert_allclose(betas[i], g[f'B{i}_betas'][it_idx], atol=5e-5)
            for idx in (6, 7, 21, 22):                                       # ankles / hands stay frozen
                np.testing.assert_array_equal(R[i, 1 + idx], g['R'][i, 1 + idx])
            np.testing.assert_allclose(loss[i], g[f'B{i}_loss'][:it + 1], rtol=5e-3, atol=2e-7)
            np.testing.assert_allclose(l2e[i], g[f'B{i}_l2e'][:it + 1], rtol=2e-3)


def test_best_iterate_selection_against_oracle(emul, golden, smpl_model_np):
    g = golden('fit')
    smpl = OS.OracleSMPL(smpl_model_np)
    # large lr makes the loss non-monotonic so "best" differs from "last"
    i = 2
    R, b, cam = (torch.from_numpy(g[k][i:i + 1]) for k in ('R', 'betas', 'cam'))
    res = OF.fit_pose_shape(smpl, R[:, 1:], R[:, :1], b, g['joints2D'][i], cam, lr=0.05, iters=30)
    Re, be, ce, loss, l2e, _ = run_emul(emul, smpl_model_np, g['R'][i:i + 1], g['betas'][i:i + 1], g['cam'][i:i + 1],
                                        g['joints2D'][i:i + 1], iters=30, lr=0.05, mask=_abi.FIT_MASK_POSE_SHAPE,
                                        select=_abi.FIT_SELECT_BEST_L2E)
    assert int(np.argmin(l2e[0])) < 29, 'test wants a non-monotonic trajectory'
    assert abs(float(l2e[0].min()) - res['best_l2e']) < 0.05
    np.testing.assert_allclose(ce[0], res['best'][3].numpy()[0], atol=5e-3)
    np.testing.assert_allclose(Re[0, 1:], res['best'][0].numpy()[0], atol=5e-3)
