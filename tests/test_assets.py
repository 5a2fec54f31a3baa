"""CPU: the loaders of the REAL assets a reference user points the package at first (`PR/config.py:3-8`): the SMPL model pickle
`<SMPL_MODEL_DIR>/SMPL_NEUTRAL.pkl` as smplx reads it (`PR/models/smpl_official.py:15-25` -> smplx.SMPL [EXT]: scipy-sparse
`J_regressor`, `posedirs` [6890,3,207] reshaped to [207,20670], `kintree_table` with 2^32-1 as the root's parent, `weights`, `f`),
the three extra joint regressors (.npy) and the mean-parameter file of the IEF (`PR/models/ief_module.py:33-46`).  The licensed
files do not ship with the reference, so the test WRITES files of exactly those formats from `synthetic.smpl_model` and loads them
back through `load_smpl_assets` / `SMPL(path)` / `SingleInputRegressor()`: every array must come back as the synthetic one."""
import os
import pickle

import numpy as np
import pytest

from soccerplayershapepose_amd import _abi, config, synthetic as S
from soccerplayershapepose_amd.models.smpl_official import SMPL, load_smpl_assets


def _write_reference_layout(root, m):
    """The file tree `PR/config.py:3-8` names, relative to `root` (the reference runs from its Python/Soccer directory)."""
    import scipy.sparse as sp
    os.makedirs(os.path.join(root, config.SMPL_MODEL_DIR), exist_ok=True)
    kin = np.stack([m['parents'].astype(np.int64), np.arange(24)]).astype(np.int64)
    kin[0, 0] = 2 ** 32 - 1                                                     # the official pickle stores uint32: -1 wraps
    raw = {'v_template': m['v_template'].astype(np.float64),
           'shapedirs': m['shapedirs'].astype(np.float64),                      # [6890,3,10]
           'posedirs': m['posedirs'].T.reshape(6890, 3, 207).astype(np.float64),  # smplx: reshape(-1, 207).T -> [207, 20670]
           'J_regressor': sp.csc_matrix(m['J_regressor'].astype(np.float64)),
           'weights': m['lbs_weights'].astype(np.float64),
           'f': m['faces'].astype(np.uint32),
           'kintree_table': kin.astype(np.uint32),
           'bs_type': 'lrotmin', 'bs_style': 'lbs'}
    with open(os.path.join(root, config.SMPL_MODEL_DIR, 'SMPL_NEUTRAL.pkl'), 'wb') as f:
        pickle.dump(raw, f, protocol=2)
    np.save(os.path.join(root, config.J_REGRESSOR_EXTRA_PATH), m['J_regressor_extra'])
    np.save(os.path.join(root, config.COCOPLUS_REGRESSOR_PATH), m['J_regressor_cocoplus'])
    np.save(os.path.join(root, config.H36M_REGRESSOR_PATH), m['J_regressor_h36m'])
    mp = S.mean_params()
    np.savez(os.path.join(root, config.SMPL_MEAN_PARAMS_PATH), pose=mp['pose'] + 0.01, shape=mp['shape'] + 0.02)


def test_smpl_pickle_and_extra_regressors_load_into_the_arrays_the_kernels_take(tmp_path, monkeypatch, smpl_model_np):
    m = smpl_model_np
    _write_reference_layout(str(tmp_path), m)
    monkeypatch.chdir(tmp_path)
    for arg in (config.SMPL_MODEL_DIR, os.path.join(config.SMPL_MODEL_DIR, 'SMPL_NEUTRAL.pkl')):     # directory or file, as smplx accepts
        d = load_smpl_assets(arg)
        for k in ('v_template', 'shapedirs', 'posedirs', 'J_regressor', 'lbs_weights', 'J_regressor_extra', 'J_regressor_cocoplus',
                  'J_regressor_h36m'):
            assert np.asarray(d[k]).shape == np.asarray(m[k]).shape, k
            np.testing.assert_array_equal(np.asarray(d[k], np.float32), np.asarray(m[k], np.float32), err_msg=k)
        np.testing.assert_array_equal(np.asarray(d['parents']), m['parents'])
        assert d['parents'][0] == -1 and np.asarray(d['parents']).dtype == np.int32
        np.testing.assert_array_equal(d['faces'], m['faces'])
        np.testing.assert_array_equal(d['vertex_joint_ids'], m['vertex_joint_ids'])
        _abi.smpl_arrays(d)                                                      # passes the host-side validation of spsp_smpl_load's input
    smpl = SMPL(config.SMPL_MODEL_DIR, batch_size=1)                             # `SMPL(config.SMPL_MODEL_DIR, batch_size=1)`, PR/player_recon.py
    assert smpl.faces.shape == (13776, 3) and smpl._handle is None               # (no GPU touched until .to(device))


def test_mean_params_file_is_picked_up_like_the_reference_ief_module(tmp_path, monkeypatch, smpl_model_np):
    from soccerplayershapepose_amd.models.regressor import SingleInputRegressor
    _write_reference_layout(str(tmp_path), smpl_model_np)
    monkeypatch.chdir(tmp_path)
    reg = SingleInputRegressor(18, 18, 3)                                        # no mean_params argument: the file at config.SMPL_MEAN_PARAMS_PATH
    mp = S.mean_params()
    want = np.zeros(157, np.float32)
    want[0] = 0.9
    want[3:147], want[147:] = mp['pose'] + 0.01, mp['shape'] + 0.02
    np.testing.assert_array_equal(reg.initial_params_estimate, want)


def test_npz_with_the_same_array_names_is_accepted(tmp_path, smpl_model_np):
    path = str(tmp_path / 'model.npz')
    np.savez(path, **smpl_model_np)
    d = load_smpl_assets(path)
    for k, v in smpl_model_np.items():
        np.testing.assert_array_equal(d[k], v)
    with pytest.raises(FileNotFoundError):
        load_smpl_assets(str(tmp_path / 'nowhere'))
