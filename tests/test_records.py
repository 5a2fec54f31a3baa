"""CPU: the on-disk record layout of the fit drivers -- `np.savez(body_pose [1,23,3,3], global_orient
[1,1,3,3], betas [1,10], translation [1,3])` (`PR/player_recon.py:1293-1294`), the multi-view split into `data.npz` +
one file per view (`:1976-1979`), and the camera the warm start recovers from the saved translation (`:1689`)."""
import os

import numpy as np
import pytest
import torch

from soccerplayershapepose_amd import records, synthetic as S
from oracle import geometry as G


def _player(seed):
    R, b, cam = S.pose_params(1, seed=seed)
    trans = G.wp_to_translation(torch.from_numpy(cam)).numpy()
    return R[:, 1:], R[:, :1], b, trans, cam


def test_single_view_record_layout_and_roundtrip(tmp_path):
    body, go, betas, trans, cam = _player(3)
    path = str(tmp_path / 'player_7')
    records.save_record(path, torch.from_numpy(body), go, betas, trans)          # tensors and arrays both accepted
    assert os.path.exists(path + '.npz')                                            # numpy appends the suffix, as in the reference
    with np.load(path + '.npz') as f:
        assert sorted(f.files) == ['betas', 'body_pose', 'global_orient', 'translation']
        assert f['body_pose'].shape == (1, 23, 3, 3) and f['global_orient'].shape == (1, 1, 3, 3)
        assert f['betas'].shape == (1, 10) and f['translation'].shape == (1, 3)
        assert all(f[k].dtype == np.float32 for k in f.files)
    rec = records.load_record(path + '.npz')
    np.testing.assert_array_equal(rec['body_pose'], body)
    np.testing.assert_array_equal(rec['translation'], trans)
    # the reference's init path (:1673-1689): arrays -> cam_wp via convert_camera_translation_to_weak_perspective
    np.testing.assert_allclose(records.translation_to_cam_wp(rec['translation']), G.translation_to_wp(torch.from_numpy(trans)).numpy(), rtol=1e-6)
    np.testing.assert_allclose(records.translation_to_cam_wp(rec['translation']), cam, rtol=2e-6)
    with pytest.raises(ValueError):
        records.save_record(path, body[0], go, betas, trans)                        # missing batch dimension
    np.savez(str(tmp_path / 'broken'), body_pose=body, betas=betas)
    with pytest.raises(KeyError):
        records.load_record(str(tmp_path / 'broken.npz'))


def test_multi_view_record_layout_and_roundtrip(tmp_path):
    body, _, betas, _, _ = _player(5)
    views = {}
    for v, name in enumerate(['cam_a.png', 'cam_b.png', 'cam_c']):
        _, go, _, trans, _ = _player(10 + v)
        views[name] = (trans, go)
    d = str(tmp_path / 'player')
    records.save_multiview_record(d, body, betas, views)
    assert sorted(os.listdir(d)) == ['cam_a.npz', 'cam_b.npz', 'cam_c.npz', 'data.npz']
    with np.load(os.path.join(d, 'data.npz')) as f:
        assert sorted(f.files) == ['betas', 'body_pose']                             # :1976
    with np.load(os.path.join(d, 'cam_b.npz')) as f:
        assert sorted(f.files) == ['global_orient', 'translation']                   # :1979
    b2, be2, v2 = records.load_multiview_record(d)
    np.testing.assert_array_equal(b2, body)
    np.testing.assert_array_equal(be2, betas)
    assert list(v2) == ['cam_a', 'cam_b', 'cam_c']
    np.testing.assert_array_equal(v2['cam_c'][1], views['cam_c'][1])
    with pytest.raises(ValueError):
        records.save_multiview_record(d, body, betas, {'data.png': views['cam_c']})
