"""GPU parity tests: the HIP path (through the Python mirror -> ctypes -> C ABI -> kernels) against the CPU
oracle on identical seeded inputs, and against the committed golden vectors.  Tolerances are stated per test:
fp32 kernels (SMPL, fit, geometry) are held to ~1e-5; the bf16 MFMA backbone is checked tightly against the
oracle's bf16-emulation mode and loosely against the fp32 reference outputs."""
import os

import numpy as np
import pytest
import torch

from soccerplayershapepose_amd import _abi, synthetic as S

pytestmark = pytest.mark.gpu
# SPSP_TEST_FULL=1: the oracle-heavy variants round 4 thinned for suite time (every player under the two-metric rules, two players
# per multi-view case, the free-running oracle chain over all 30 frames of the clip) -- run once per round, profiles/r05_full_suite.txt
FULL_ORACLE = bool(os.environ.get('SPSP_TEST_FULL'))


def _oracle_fit_vertices(oracle_smpl, R, betas, tgt, cam, iters=50):
    """Vertices of the CPU oracle's pose+shape fit for a batch of players (oracle/fit.py:fit_pose_shape_batch -- the per-player
    reference loop in one autograd graph, checked against that loop in tests/test_oracle_golden.py)."""
    from oracle import fit as OF
    r = OF.fit_pose_shape_batch(oracle_smpl, R[:, 1:], R[:, :1], betas, np.asarray(tgt), cam, iters=iters)
    body, Gm, be, ca, _ = r['last']
    with torch.no_grad():
        return oracle_smpl(body_pose=body, global_orient=Gm, betas=be).vertices



@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'GPU tests need a ROCm device'
    return torch.device('cuda', 0)


@pytest.fixture(scope='module')
def smpl_gpu(smpl_model_np, dev):
    from soccerplayershapepose_amd.models.smpl_official import SMPL
    return SMPL(smpl_model_np, batch_size=1, max_batch=64).to(dev)


@pytest.fixture(scope='module')
def oracle_smpl(smpl_model_np):
    from oracle import smpl as OS
    return OS.OracleSMPL(smpl_model_np)


@pytest.fixture(scope='module')
def regressor_gpu(regressor_sd_np, dev):
    from soccerplayershapepose_amd.models.regressor import SingleInputRegressor
    reg = SingleInputRegressor(18, 18, 3, mean_params=S.mean_params(), max_batch=32, precision='bf16')
    reg.load_state_dict(regressor_sd_np)
    return reg.to(dev).eval()


@pytest.fixture(scope='module')
def regressor_gpu_f16(regressor_sd_np, dev):
    from soccerplayershapepose_amd.models.regressor import SingleInputRegressor
    reg = SingleInputRegressor(18, 18, 3, mean_params=S.mean_params(), max_batch=32, precision='f16')
    reg.load_state_dict(regressor_sd_np)
    return reg.to(dev).eval()


def test_native_library_is_the_code_that_runs():
    lib = _abi.load()
    assert lib.spsp_version().startswith(b'spsp')
    with open('/proc/self/maps') as f:
        assert 'libspsp.so' in f.read()


# ---------------------------------------------------------------- geometry (a5, a7, a12)
def test_geometry_against_reference_golden(golden, dev):
    from soccerplayershapepose_amd.utils.rigid_transform_utils import rot6d_to_rotmat
    from soccerplayershapepose_amd.utils.cam_utils import (orthographic_project_torch, orthographic_project_to_pixels,
        convert_weak_perspective_to_camera_translation_torch, convert_camera_translation_to_weak_perspective_torch)
    from soccerplayershapepose_amd.utils.joints2d_utils import undo_keypoint_normalisation
    g = golden('geometry')
    R = rot6d_to_rotmat(torch.from_numpy(g['rot6d_in']).to(dev))
    assert R.shape == (120, 3, 3)
    np.testing.assert_allclose(R.cpu().numpy(), g['rot6d_out'], atol=2e-6)
    pts, cam = torch.from_numpy(g['proj_pts']).to(dev), torch.from_numpy(g['proj_cam']).to(dev)
    proj = orthographic_project_torch(pts, cam)
    np.testing.assert_allclose(proj.cpu().numpy(), g['proj_out'], atol=1e-6)
    np.testing.assert_allclose(undo_keypoint_normalisation(proj, 512).cpu().numpy(), g['proj_px512'], atol=2e-4)
    np.testing.assert_allclose(orthographic_project_to_pixels(pts, cam, 512).cpu().numpy(), g['proj_px512'], atol=2e-4)
    tr = convert_weak_perspective_to_camera_translation_torch(cam, 5000., 512)
    np.testing.assert_allclose(tr.cpu().numpy(), g['wp2t_out'], rtol=2e-6)
    np.testing.assert_allclose(convert_camera_translation_to_weak_perspective_torch(tr, 5000., 512).cpu().numpy(),
                               g['t2wp_out'], rtol=2e-6)


def test_rot6d_degenerate_inputs_follow_f_normalize(dev):
    from soccerplayershapepose_amd.utils.rigid_transform_utils import rot6d_to_rotmat
    from oracle import geometry as G
    x = torch.zeros(3, 6)
    x[1] = torch.tensor([1., 1., 0., 0., 0., 0.])       # a2 parallel to a1
    x[2] = torch.tensor([1e-20, 0., 0., 1., 0., 0.])
    got = rot6d_to_rotmat(x.to(dev)).cpu()
    ref = G.rot6d_to_rotmat(x)
    assert torch.isfinite(got).all()
    np.testing.assert_allclose(got.numpy(), ref.numpy(), atol=1e-6)


def test_record_pack_roundtrip_is_bit_exact(dev):
    from soccerplayershapepose_amd.player_recon import pack_record, unpack_record
    R, b, cam = (torch.from_numpy(a).to(dev) for a in S.pose_params(37, seed=3))
    rec = pack_record(cam, R, b)
    assert rec.shape == (37, 229)
    c2, R2, b2 = unpack_record(rec)
    assert torch.equal(c2, cam) and torch.equal(R2, R) and torch.equal(b2, b)
    assert torch.equal(rec[:, :3], cam) and torch.equal(rec[:, 3:219], R.reshape(37, 216)) and torch.equal(rec[:, 219:], b)


# ---------------------------------------------------------------- SMPL forward (a6)
def test_smpl_forward_matches_oracle_and_fixture(golden, smpl_gpu, oracle_smpl, smpl_model_np, dev):
    from oracle import smpl as OS
    g = golden('smpl')
    R, b = torch.from_numpy(g['R']).to(dev), torch.from_numpy(g['betas']).to(dev)
    out = smpl_gpu(body_pose=R[:, 1:], global_orient=R[:, :1], betas=b, pose2rot=False)
    assert out.vertices.shape == (4, 6890, 3) and out.joints.shape == (4, 90, 3)
    v64, j64 = OS.smpl_forward_np(smpl_model_np, g['R'], g['betas'])
    assert np.abs(out.vertices.cpu().numpy() - v64).max() < 1e-5        # tolerance: 1e-5 m (fp32 vs float64 twin)
    assert np.abs(out.joints.cpu().numpy() - j64).max() < 1e-5
    np.testing.assert_allclose(out.vertices.cpu().numpy()[:, ::13], g['verts_sub'], atol=1e-5)
    np.testing.assert_allclose(out.joints.cpu().numpy(), g['joints'], atol=1e-5)
    assert out.full_pose.shape == (4, 24, 3, 3) and smpl_gpu.faces.shape == (13776, 3)


def test_smpl_ragged_batches_tpose_and_unconstrained_matrices(smpl_gpu, smpl_model_np, dev):
    from oracle import smpl as OS
    for B in (1, 7, 8, 9, 33):           # ragged vs the 8-player skin workgroup
        R, b, _ = S.pose_params(B, seed=11 + B)
        R = R + 0.05 * S.hash_normal(R.size, f'unc{B}').reshape(R.shape).astype(np.float32)   # NOT orthonormal (fit iterates)
        out = smpl_gpu(body_pose=torch.from_numpy(R[:, 1:]).to(dev), global_orient=torch.from_numpy(R[:, :1]).to(dev),
                       betas=torch.from_numpy(b).to(dev))
        v64, j64 = OS.smpl_forward_np(smpl_model_np, R, b)
        assert np.abs(out.vertices.cpu().numpy() - v64).max() < 2e-5
        assert np.abs(out.joints.cpu().numpy() - j64).max() < 2e-5
    t = smpl_gpu(betas=torch.from_numpy(b).to(dev))                       # predict_3D.py:148 path
    vt, _ = OS.smpl_forward_np(smpl_model_np, np.tile(np.eye(3, dtype=np.float32), (33, 24, 1, 1)), b)
    assert np.abs(t.vertices.cpu().numpy() - vt).max() < 1e-5


def test_smpl_linearity_in_betas_at_bench_size(smpl_gpu, dev):
    """Size-independent property at B=32: with identity pose the mesh is affine in betas."""
    B = 32
    b1 = torch.from_numpy(S.pose_params(B, seed=5)[1]).to(dev)
    b2 = torch.from_numpy(S.pose_params(B, seed=6)[1]).to(dev)
    v = lambda b: smpl_gpu(betas=b, return_joints=False).vertices
    lhs = v(0.25 * b1 + 0.75 * b2)
    rhs = 0.25 * v(b1) + 0.75 * v(b2)
    assert (lhs - rhs).abs().max().item() < 5e-6


def test_smpl_rejects_oversized_and_handles_empty_batch(smpl_gpu, dev):
    with pytest.raises(ValueError):
        smpl_gpu(betas=torch.zeros(65, 10, device=dev))
    out = smpl_gpu(betas=torch.zeros(0, 10, device=dev))
    assert out.vertices.shape == (0, 6890, 3)


# ---------------------------------------------------------------- fit (a9, a10, a11)
def _fit_inputs(g, dev):
    R = torch.from_numpy(g['R']).to(dev).contiguous()
    return R, torch.from_numpy(g['betas']).to(dev), torch.from_numpy(g['cam']).to(dev), g['joints2D']


def test_fit_variant_a_trajectory_and_inplace_views(golden, smpl_gpu, dev):
    from soccerplayershapepose_amd.player_recon import fit_players, optimize_camera
    g = golden('fit')
    keep = list(g['keep_iters'])
    for it_idx, it in enumerate(keep):
        R, b, cam, j = _fit_inputs(g, dev)
        loss, l2e = fit_players(smpl_gpu, R, b, cam, j, variant=0, iters=it + 1, return_history=True)
        for i in range(3):
            np.testing.assert_allclose(cam[i].cpu().numpy(), g[f'A{i}_cam'][it_idx], atol=5e-5)
            np.testing.assert_allclose(R[i, 0].cpu().numpy(), g[f'A{i}_G'][it_idx], atol=5e-5)
            np.testing.assert_allclose(loss[i].cpu().numpy(), g[f'A{i}_loss'][:it + 1], rtol=5e-3, atol=2e-7)
        assert torch.equal(R[:, 1:].cpu(), torch.from_numpy(g['R'][:, 1:]))
    # reference calling convention: per-player VIEWS into batch tensors are updated in place (:3766-3767)
    R, b, cam, j = _fit_inputs(g, dev)
    body, orient = R[:, 1:].clone(), R[:, :1].clone()
    for i in range(3):
        optimize_camera(body[i:i + 1], orient[i:i + 1], b[i:i + 1], j[i], None, cam[i:i + 1], smpl=smpl_gpu)
    for i in range(3):
        np.testing.assert_allclose(cam[i].cpu().numpy(), g[f'A{i}_cam'][-1], atol=5e-5)
        np.testing.assert_allclose(orient[i, 0].cpu().numpy(), g[f'A{i}_G'][-1], atol=5e-5)


def test_fit_variant_b_trajectory_frozen_joints_and_record(golden, smpl_gpu, dev):
    from soccerplayershapepose_amd.player_recon import fit_players, single_view_optimization
    g = golden('fit')
    keep = list(g['keep_iters'])
    for it_idx, it in enumerate(keep):
        R, b, cam, j = _fit_inputs(g, dev)
        loss, l2e = fit_players(smpl_gpu, R, b, cam, j, variant=1, iters=it + 1, return_history=True)
        for i in range(3):
            np.testing.assert_allclose(cam[i].cpu().numpy(), g[f'B{i}_cam'][it_idx], atol=5e-5)
            np.testing.assert_allclose(R[i, 1:].cpu().numpy(), g[f'B{i}_body'][it_idx], atol=5e-5)
            np.testing.assert_allclose(b[i].cpu().numpy(), g[f'B{i}_betas'][it_idx], atol=5e-5)
            np.testing.assert_allclose(loss[i].cpu().numpy(), g[f'B{i}_loss'][:it + 1], rtol=5e-3, atol=2e-7)
            np.testing.assert_allclose(l2e[i].cpu().numpy(), g[f'B{i}_l2e'][:it + 1], rtol=2e-3)
            for idx in (6, 7, 21, 22):
                np.testing.assert_array_equal(R[i, 1 + idx].cpu().numpy(), g['R'][i, 1 + idx])
    R, b, cam, j = _fit_inputs(g, dev)
    rec = single_view_optimization(smpl_gpu, R[:, 1:], R[:, :1], b, cam, j, interation=50)
    for i in range(3):
        np.testing.assert_allclose(rec.translation[i].cpu().numpy(), g[f'B{i}_translation'], rtol=1e-4)
    assert torch.equal(R.cpu(), torch.from_numpy(g['R']))                 # inputs untouched


def test_fit_against_live_oracle_on_fresh_players_and_best_iterate(smpl_gpu, oracle_smpl, smpl_model_np, dev):
    from soccerplayershapepose_amd.player_recon import fit_players
    from oracle import fit as OF, smpl as OS
    B = 4
    R, b, cam = S.pose_params(B, seed=21)
    R2, b2, _ = S.pose_params(B, seed=22, body_deg=10, global_deg=12)
    Rp = (R2.astype(np.float64) @ R.astype(np.float64)).astype(np.float32)
    _, jp = OS.smpl_forward_np(smpl_model_np, Rp, b + b2 / 3)
    tgt = S.fit_targets_from_joints(jp, cam * np.array([0.93, 1, 1], np.float32), seed=9)
    Rt, bt, ct = (torch.from_numpy(a) for a in (R, b, cam))
    Rg, bg, cg = Rt.to(dev).contiguous(), bt.to(dev), ct.to(dev)
    fit_players(smpl_gpu, Rg, bg, cg, tgt, variant=1, iters=50)
    for i in range(B):
        res = OF.fit_pose_shape(oracle_smpl, Rt[i:i + 1, 1:], Rt[i:i + 1, :1], bt[i:i + 1], tgt[i], ct[i:i + 1], iters=50)
        body, G, be, ca, _ = res['last']
        assert np.abs(Rg[i, 1:].cpu().numpy() - body.numpy()[0]).max() < 1e-4   # tolerance: 1e-4 on raw matrix entries
        assert np.abs(Rg[i, 0].cpu().numpy() - G.numpy()[0, 0]).max() < 1e-4
        assert np.abs(bg[i].cpu().numpy() - be.numpy()[0]).max() < 1e-4
        assert np.abs(cg[i].cpu().numpy() - ca.numpy()[0]).max() < 1e-4
    # best-iterate rule (large lr -> non-monotonic l2e)
    i = 0
    res = OF.fit_pose_shape(oracle_smpl, Rt[i:i + 1, 1:], Rt[i:i + 1, :1], bt[i:i + 1], tgt[i], ct[i:i + 1], lr=0.05, iters=30)
    Rg, bg, cg = Rt[i:i + 1].to(dev).contiguous(), bt[i:i + 1].to(dev), ct[i:i + 1].to(dev)
    loss, l2e = fit_players(smpl_gpu, Rg, bg, cg, tgt[i:i + 1], variant=1, iters=30, lr=0.05,
                            select_mode=_abi.FIT_SELECT_BEST_L2E, return_history=True)
    assert abs(l2e.min().item() - res['best_l2e']) < 0.05
    np.testing.assert_allclose(cg.cpu().numpy()[0], res['best'][3].numpy()[0], atol=5e-3)


def test_fit_variant_b_at_the_reference_default_of_100_iterations(smpl_gpu, oracle_smpl, smpl_model_np, dev):
    """`single_view_optimization` runs `interation=player_recon_single_view_iteration` = 100 Adam steps unless told otherwise
    (`PR/player_recon.py:1051,1200`, `S/global_var.py:75`); the bench and most parity tests stop at 50 (north star).  Here the
    mirror's DEFAULT length against the autograd oracle on five fresh players: raw parameters within 2e-4 after 100 steps (1e-4 is
    the bound at 50; Adam's sign-sensitive steps let the two fp32 trajectories drift apart with length), vertices within 2e-4 m."""
    from soccerplayershapepose_amd.player_recon import single_view_optimization
    from oracle import fit as OF, smpl as OS
    B = 5
    R, b, cam = S.pose_params(B, seed=121)
    R2, b2, _ = S.pose_params(B, seed=122, body_deg=10, global_deg=12)
    Rp = (R2.astype(np.float64) @ R.astype(np.float64)).astype(np.float32)
    _, jp = OS.smpl_forward_np(smpl_model_np, Rp, b + b2 / 3)
    tgt = S.fit_targets_from_joints(jp, cam * np.array([0.93, 1, 1], np.float32), seed=19)
    Rt, bt, ct = (torch.from_numpy(a) for a in (R, b, cam))
    rec = single_view_optimization(smpl_gpu, Rt[:, 1:].to(dev), Rt[:, :1].to(dev), bt.to(dev), ct.to(dev), tgt)     # no `interation`: 100
    o = OF.fit_pose_shape_batch(oracle_smpl, Rt[:, 1:], Rt[:, :1], bt, tgt, ct, iters=100)['last']
    got = (rec.body_pose, rec.global_orient, rec.betas, rec.cam_wp)
    d = [(g.cpu() - w).abs().max().item() for g, w in zip(got, o[:4])]
    assert max(d) < 2e-4, d
    np.testing.assert_allclose(rec.translation.cpu().numpy(), o[4].numpy(), rtol=2e-4, atol=1e-5)
    o50 = OF.fit_pose_shape_batch(oracle_smpl, Rt[:, 1:], Rt[:, :1], bt, tgt, ct, iters=50)['last']
    assert (o50[0] - o[0]).abs().max().item() > 1e-3          # (the second 50 steps do move the pose: the default length matters)
    with torch.no_grad():
        v = oracle_smpl(body_pose=o[0], global_orient=o[1], betas=o[2]).vertices
    vg = smpl_gpu(body_pose=rec.body_pose, global_orient=rec.global_orient, betas=rec.betas).vertices
    assert (vg.cpu() - v).abs().max().item() < 2e-4


def test_fit_is_deterministic_and_batch_independent(smpl_gpu, dev):
    """Run-twice bit compare + a player's result must not depend on its batch neighbours (B=32 bench size)."""
    from soccerplayershapepose_amd.player_recon import fit_players
    B = 32
    R, b, cam = S.pose_params(B, seed=31)
    out = smpl_gpu(body_pose=torch.from_numpy(R[:, 1:]).to(dev), global_orient=torch.from_numpy(R[:, :1]).to(dev),
                   betas=torch.from_numpy(b).to(dev))
    tgt = S.fit_targets_from_joints(out.joints.cpu().numpy(), cam * np.array([1.08, 1, 1], np.float32), seed=2)
    runs = []
    for _ in range(2):
        Rg, bg, cg = (torch.from_numpy(a).to(dev).contiguous() for a in (R, b, cam))
        fit_players(smpl_gpu, Rg, bg, cg, tgt, variant=1, iters=50)
        runs.append((Rg.clone(), bg.clone(), cg.clone()))
    assert all(torch.equal(x, y) for x, y in zip(*runs))
    Rg, bg, cg = (torch.from_numpy(a[5:6]).to(dev).contiguous() for a in (R, b, cam))
    fit_players(smpl_gpu, Rg, bg, cg, tgt[5:6], variant=1, iters=50)
    assert torch.equal(Rg[0], runs[0][0][5]) and torch.equal(cg[0], runs[0][2][5])


def test_patch_kernels_with_producer_waves_are_bit_identical(dev):
    """csrc/conv_patch.hip, PROD: four DMA-only wavefronts per workgroup issue the LDS-DMA pieces in place of the accumulating
    waves (layer3 / layer4, 128-pixel tiles; `SPSP_DEBUG_PLAN_PATCH_PRODUCER`).  The arithmetic is untouched, so every output of
    the regressor and the layer3 / layer4 feature maps must equal the plain kernels' bit for bit -- both precisions, both layer4
    block widths, B = 32 (full grid) and B = 6 (ragged last tiles)."""
    from soccerplayershapepose_amd import runtime as rt
    from soccerplayershapepose_amd.models.regressor import SingleInputRegressor
    sd = S.regressor_state_dict(seed=0)
    for prec in ('f16', 'bf16'):
        for B in (32, 6):
            reg = SingleInputRegressor(18, 18, 3, mean_params=S.mean_params(), max_batch=B, precision=prec).load_state_dict(sd).to(dev)
            r = rt.get(dev)
            x, _ = S.proxy_crops(B, seed=40 + B)
            native = reg.pack(torch.from_numpy(x).to(dev))
            for nch in (64, 32):
                got = []
                for prod in (0, 1):
                    r.check(r.lib.spsp_regressor_debug(reg.handle, _abi.DEBUG_PLAN_LAYER4_CHANNELS, nch), 'debug')
                    for L in (2, 3):
                        r.check(r.lib.spsp_regressor_debug(reg.handle, _abi.DEBUG_PLAN_PATCH_PRODUCER, L * 10 + prod), 'debug')
                    out = [t.clone() for t in reg(native)]
                    out += [reg.read_stage(st, B).clone() for st in (4, 5)]
                    torch.cuda.synchronize()
                    reg.check_status()
                    got.append(out)
                for a, b in zip(*got):
                    assert torch.isfinite(a).all() and torch.equal(a, b)


def test_layer2_stride2_pair_with_register_stationary_weights_is_bit_identical(dev):
    """csrc/conv_s2rs.hip: layer2.0's 3x3 / 2 convolution + 1x1 / 2 downsample with the block's weights resident in registers
    and two 128-pixel tiles per workgroup (`SPSP_DEBUG_PLAN_S2RS`; 0 = conv_s2.hip, the streaming kernel of layer3.0 / layer4.0).
    Same products accumulated in the same order, so the regressor outputs and layer2's feature map must equal the streaming
    kernel's bit for bit: B = 32 and 22 (two tiles per workgroup, full and partial grid), 16 and 8 (one tile per workgroup), 3
    (a grid that is no multiple of 16: the unpaired tile map), both precisions; and with four / eight tiles per workgroup (plan
    "throughput", `SPSP_DEBUG_PLAN_S2RS_CUS`: the tile after next is requested into the buffer a finished tile leaves)."""
    from soccerplayershapepose_amd import runtime as rt
    from soccerplayershapepose_amd.models.regressor import SingleInputRegressor
    sd = S.regressor_state_dict(seed=0)
    for prec in ('f16', 'bf16'):
        for B in (32, 22, 16, 8, 3):
            reg = SingleInputRegressor(18, 18, 3, mean_params=S.mean_params(), max_batch=B, precision=prec).load_state_dict(sd).to(dev)
            r = rt.get(dev)
            x, _ = S.proxy_crops(B, seed=60 + B)
            native = reg.pack(torch.from_numpy(x).to(dev))
            got = []
            r.check(r.lib.spsp_regressor_debug(reg.handle, _abi.DEBUG_PLAN_S2Q, 0), 'debug')      # (round 5's conv_s2q.hip off: below)
            for rs, cus in ((0, 1), (1, 1), (1, 2), (1, 4)):      # (cus: 4 and 8 tiles per workgroup where they divide -- plan "throughput")
                r.check(r.lib.spsp_regressor_debug(reg.handle, _abi.DEBUG_PLAN_S2RS, rs), 'debug')
                r.check(r.lib.spsp_regressor_debug(reg.handle, _abi.DEBUG_PLAN_S2RS_CUS, cus), 'debug')
                out = [t.clone() for t in reg(native)]
                out.append(reg.read_stage(3, B).clone())
                torch.cuda.synchronize()
                reg.check_status()
                got.append(out)
            for other in got[1:]:
                for a, b in zip(got[0], other):
                    assert torch.isfinite(a).all() and torch.equal(a, b)


def test_layer2_stride2_pair_with_all_channels_in_one_workgroup_matches_the_channel_block_kernel(dev):
    """csrc/conv_s2q.hip (round 5, the default): layer2.0's 3x3 / 2 convolution + 1x1 / 2 downsample with all 128 output channels in
    one workgroup -- the parity planes staged once, a wave = a 32-channel quarter over the whole K, weights in registers, tiles of
    4 x 16 pixels -- against conv_s2rs.hip (`SPSP_DEBUG_PLAN_S2Q` = 0: 64-channel blocks, K halves added through LDS).  The same
    products added in another order (taps 0..8 in a row instead of plane by plane, no split of K): layer2's OUTPUT map -- three more
    convolutions behind the pair -- within one unit in the last place of the element type at its largest value (f16: 2^-10 = 1e-3 of
    it, measured exactly that: 0.00195 at 2.86; bf16: 2^-7), the regressor outputs within 1e-4 / 5e-4 (measured 3.0e-5); against the rounding-emulating
    oracle in the stage tests, which run it.  Two, four and eight tiles per workgroup (the grids of
    both plans) are the same bits."""
    from soccerplayershapepose_amd import runtime as rt
    from soccerplayershapepose_amd.models.regressor import SingleInputRegressor
    sd = S.regressor_state_dict(seed=0)
    for prec, batches in (('f16', (32, 22, 3, 1)), ('bf16', (32, 5))):
        for B in batches:
            reg = SingleInputRegressor(18, 18, 3, mean_params=S.mean_params(), max_batch=B, precision=prec).load_state_dict(sd).to(dev)
            r = rt.get(dev)
            x, _ = S.proxy_crops(B, seed=90 + B)
            native = reg.pack(torch.from_numpy(x).to(dev))
            got = {}
            for s2q, cus in ((0, 1), (1, 1), (1, 2), (1, 4)):
                r.check(r.lib.spsp_regressor_debug(reg.handle, _abi.DEBUG_PLAN_S2Q, s2q), 'debug')
                r.check(r.lib.spsp_regressor_debug(reg.handle, _abi.DEBUG_PLAN_S2RS_CUS, cus), 'debug')
                out = [t.clone() for t in reg(native)] + [reg.read_stage(3, B).clone()]
                torch.cuda.synchronize()
                reg.check_status()
                got[(s2q, cus)] = out
            for cus in (2, 4):
                assert all(torch.equal(a, b) for a, b in zip(got[(1, 1)], got[(1, cus)])), (prec, B, cus)
            ref, new = got[(0, 1)], got[(1, 1)]
            assert all(torch.isfinite(t).all() for t in new)
            scale = ref[3].abs().max().item()
            assert (new[3] - ref[3]).abs().max().item() <= (1e-3 if prec == 'f16' else 8e-3) * scale, (prec, B)
            assert max((a - b).abs().max().item() for a, b in zip(new[:3], ref[:3])) <= (1e-4 if prec == 'f16' else 5e-4), (prec, B)


def test_ief_forms_agree(dev):
    """The fused IEF (csrc/regressor.hip).  `ief_poll_kernel` (round 5) hands its hidden vectors over without an arrival counter: every
    hidden value is >= +0, is stored NEGATED (sign bit set) into a per-phase buffer that the forward's stem kernel has zeroed, and a
    consumer loads until none of its words is zero.  With 16 lanes per player pair (`SPSP_DEBUG_IEF_POLL` = 1) its sums are
    `ief_fused_kernel`'s (counter hand-offs, `SPSP_DEBUG_IEF_POLL` = 0, blocks of 16 or 32 players: `SPSP_DEBUG_IEF_BLOCK`) in the same
    order -- the stored sign is undone by the FMA's source modifier --: cam, pose6d and betas bit for bit; the default, 32 lanes per pair
    (= 2: a product is 256 instead of 512 FMAs per thread, the 32-lane sum by a v_permlane16_swap on top of the DPP row steps), adds the
    same products in another order: 1e-6 on outputs of order 1.  B = 32, 22 (a ragged last block: idle lanes recompute the last player),
    3, 1 and 72 (five blocks), three forwards in a row on the handle (the buffers re-zeroed in between).  Against the oracle in the stage
    tests, which run the default."""
    from soccerplayershapepose_amd import runtime as rt
    from soccerplayershapepose_amd.models.regressor import SingleInputRegressor
    sd = S.regressor_state_dict(seed=0)
    r = rt.get(dev)
    for B in (32, 22, 3, 1, 72):
        reg = SingleInputRegressor(18, 18, 3, mean_params=S.mean_params(), max_batch=B, precision='f16').load_state_dict(sd).to(dev)
        natives = []
        for k in range(2):
            x, _ = S.proxy_crops(B, seed=150 + B + 1000 * k)
            natives.append(reg.pack(torch.from_numpy(x).to(dev)))
        got = {}
        for poll, blk in ((0, 32), (0, 16), (1, 16), (2, 16)):
            r.check(r.lib.spsp_regressor_debug(reg.handle, _abi.DEBUG_IEF_POLL, poll), 'debug')
            r.check(r.lib.spsp_regressor_debug(reg.handle, _abi.DEBUG_IEF_BLOCK, blk), 'debug')
            outs = []
            for nat in natives + natives[:1]:                   # three forwards in a row on the handle
                outs += [t.clone() for t in reg(nat)]
            torch.cuda.synchronize()
            reg.check_status()
            got[(poll, blk)] = outs
        assert all(torch.isfinite(t).all() for t in got[(2, 16)])
        assert not torch.equal(got[(2, 16)][0], got[(2, 16)][3])            # (the two inputs differ)
        for key in ((0, 16), (1, 16)):
            assert all(torch.equal(a, b) for a, b in zip(got[(0, 32)], got[key])), (B, key)
        assert max(float((a - b).abs().max()) for a, b in zip(got[(0, 32)], got[(2, 16)])) <= 1e-6, B


def test_layer2_with_register_stationary_weights_matches_the_streaming_kernel(dev):
    """csrc/conv_rs.hip (round 5, plan "throughput"): layer2's three 3x3 / 1 convolutions with a 64-channel block's weights stationary
    in registers and persistent workgroups over 8 x 16-pixel tiles (`SPSP_DEBUG_PLAN_RS` = workgroups; 0 = conv_p4.hip, which streams
    the weight slabs through an LDS ring per tile).  The same products added in another order (K-step = 4 tap + 2 chunk + half instead
    of chunk-major): layer2's map within one unit in the last place of the element type at its largest value, the regressor outputs
    within 1e-4 / 5e-4; B = 32, 22 (176 tiles over 32 ranges: ragged), 3 and 1; 64, 32, 128 and 256 workgroups give the same bits.
    conv_p4.hip itself is held to the rounding-emulating oracle in the stage tests."""
    from soccerplayershapepose_amd import runtime as rt
    from soccerplayershapepose_amd.models.regressor import SingleInputRegressor
    sd = S.regressor_state_dict(seed=0)
    for prec, batches in (('f16', (32, 22, 3, 1)), ('bf16', (32, 5))):
        for B in batches:
            reg = SingleInputRegressor(18, 18, 3, mean_params=S.mean_params(), max_batch=B, precision=prec).load_state_dict(sd).to(dev)
            r = rt.get(dev)
            x, _ = S.proxy_crops(B, seed=130 + B)
            native = reg.pack(torch.from_numpy(x).to(dev))
            got = {}
            for wgs in (0, 64, 32, 128, 256):
                r.check(r.lib.spsp_regressor_debug(reg.handle, _abi.DEBUG_PLAN_RS, wgs), 'debug')
                out = [t.clone() for t in reg(native)] + [reg.read_stage(3, B).clone()]
                torch.cuda.synchronize()
                reg.check_status()
                got[wgs] = out
            for wgs in (32, 128, 256):
                assert all(torch.equal(a, b) for a, b in zip(got[64], got[wgs])), (prec, B, wgs)
            ref, new = got[0], got[64]
            assert all(torch.isfinite(t).all() for t in new)
            scale = ref[3].abs().max().item()
            assert (new[3] - ref[3]).abs().max().item() <= (1e-3 if prec == 'f16' else 8e-3) * scale, (prec, B, (new[3] - ref[3]).abs().max().item(), scale)
            assert max((a - b).abs().max().item() for a, b in zip(new[:3], ref[:3])) <= (1e-4 if prec == 'f16' else 5e-4), (prec, B)
    # the plans select it: "throughput" runs conv_rs (64 workgroups), "latency" conv_p4
    reg = SingleInputRegressor(18, 18, 3, mean_params=S.mean_params(), max_batch=32, precision='f16').load_state_dict(sd).to(dev)
    x, _ = S.proxy_crops(32, seed=131)
    native = reg.pack(torch.from_numpy(x).to(dev))
    thr = [t.clone() for t in reg.set_plan('throughput')(native)]
    lat = [t.clone() for t in reg.set_plan('latency')(native)]
    torch.cuda.synchronize()
    assert all(float((a - b).abs().max()) <= 1e-4 for a, b in zip(lat, thr))


def test_layer2_layer3_without_the_k_split_match_the_k_split_kernel(dev):
    """csrc/conv_p4.hip: the 3x3 / 1 convolutions of layer2 (both plans) and layer3 (plan "throughput") on 256-pixel tiles with
    four accumulating waves over the whole K and four producer waves (`SPSP_DEBUG_PLAN_P4`, bit L = residual stage L; 0 =
    conv_patch.hip everywhere: K split over wave pairs, halves added through LDS).  The same products, added in another order
    (and starting from the bias instead of adding it last): layer2's and layer3's feature maps within 1.5e-3 (f16) / 1e-2 (bf16)
    of the map's largest value, the regressor outputs within 1e-4 / 1e-3; B = 32 and 22 (XCD-aware and plain tile order), 8 and 3
    (grids below one workgroup per CU).  The default plan (stages against the rounding-emulating oracle, tests above) runs this
    kernel; this test keeps conv_patch.hip's layer2 / layer3 forms covered and pins the two against each other."""
    from soccerplayershapepose_amd import runtime as rt
    from soccerplayershapepose_amd.models.regressor import SingleInputRegressor
    sd = S.regressor_state_dict(seed=0)
    for prec, tol_map, tol_out, batches in (('f16', 1.5e-3, 1e-4, (32, 22, 3)), ('bf16', 1e-2, 1e-3, (32, 8))):
        for B in batches:
            reg = SingleInputRegressor(18, 18, 3, mean_params=S.mean_params(), max_batch=B, precision=prec).load_state_dict(sd).to(dev)
            r = rt.get(dev)
            x, _ = S.proxy_crops(B, seed=90 + B)
            native = reg.pack(torch.from_numpy(x).to(dev))
            got = {}
            for plan in (0, 2, 6):
                r.check(r.lib.spsp_regressor_debug(reg.handle, _abi.DEBUG_PLAN_P4, plan), 'debug')
                out = [t.clone() for t in reg(native)]
                maps = [reg.read_stage(s, B).clone() for s in (3, 4)]
                torch.cuda.synchronize()
                reg.check_status()
                got[plan] = (out, maps)
            for plan in (2, 6):
                for a, b in zip(got[plan][0], got[0][0]):
                    assert torch.isfinite(a).all() and float((a - b).abs().max()) <= tol_out
                for a, b in zip(got[plan][1], got[0][1]):
                    assert float((a - b).abs().max()) <= tol_map * float(b.abs().max())
            # plan 2 leaves layer3 on conv_patch.hip: given the same layer2 map, plan 6's layer2 map is plan 2's bit for bit
            assert torch.equal(got[2][1][0], got[6][1][0])


def test_stem_on_32x32x16_mfma_matches_the_16x16x32_form(dev):
    """csrc/conv_stem.hip, conv1_pool32r_kernel (round 5, the default): conv1 + BN + ReLU + max-pool on v_mfma_f32_32x32x16 -- one MFMA per
    32 cycles for a loop that is bound by its issue slots, even / odd conv columns in even / odd 16-lane rows, the horizontal pool by
    v_permlane16_swap + a DPP row shift, and both pools RIDING in the K loop of the following step(s): pixel tiles as two pairs, pooled
    rows two deep by step parity, the row above carried in registers, one barrier per step -- (i) bit for bit against the same MFMAs
    with the pooling behind the loop (`SPSP_DEBUG_PLAN_STEM32` = 2: conv1_pool32_kernel), at every band height from the 4-row bands of B = 1 and 3
    (two and three steps per workgroup: every guard of the riding schedule) to 64 rows; (ii) against the 16x16x32 kernel of rounds 1-4
    (`SPSP_DEBUG_PLAN_STEM32` = 0).  The same products
    summed in another order (56 MFMAs of depth 16 per output instead of 28 of depth 32): the pooled map within one unit in the last
    place of the element type at its largest value, the regressor outputs within 1e-4 / 5e-4; B = 32, 22 (16-row bands), 3 and 1
    (finer bands), the band heights of plan "throughput", both element types.  Against the rounding-emulating oracle in the stage
    tests, which run it."""
    from soccerplayershapepose_amd import runtime as rt
    from soccerplayershapepose_amd.models.regressor import SingleInputRegressor
    sd = S.regressor_state_dict(seed=0)
    for prec, batches in (('f16', (32, 22, 3, 1)), ('bf16', (32, 5))):
        for B in batches:
            reg = SingleInputRegressor(18, 18, 3, mean_params=S.mean_params(), max_batch=B, precision=prec).load_state_dict(sd).to(dev)
            r = rt.get(dev)
            x, _ = S.proxy_crops(B, seed=110 + B)
            native = reg.pack(torch.from_numpy(x).to(dev))
            got = {}
            for m32, band in ((0, 0), (1, 0), (1, 32), (1, 64), (2, 0), (2, 64)):
                r.check(r.lib.spsp_regressor_debug(reg.handle, _abi.DEBUG_PLAN_STEM32, m32), 'debug')
                r.check(r.lib.spsp_regressor_debug(reg.handle, _abi.DEBUG_PLAN_STEM_BAND, band), 'debug')
                out = [t.clone() for t in reg(native)] + [reg.read_stage(1, B).clone()]
                torch.cuda.synchronize()
                reg.check_status()
                got[(m32, band)] = out
            for key in ((1, 32), (1, 64), (2, 0), (2, 64)):   # which workgroup computes a row, and when it is pooled, changes; nothing about the row
                assert all(torch.equal(a, b) for a, b in zip(got[(1, 0)], got[key])), (prec, B, key, (got[(1, 0)][3] - got[key][3]).abs().max().item())
            ref, new = got[(0, 0)], got[(1, 0)]
            assert all(torch.isfinite(t).all() for t in new)
            scale = ref[3].abs().max().item()
            assert (new[3] - ref[3]).abs().max().item() <= (1e-3 if prec == 'f16' else 8e-3) * scale, (prec, B, (new[3] - ref[3]).abs().max().item(), scale)
            assert max((a - b).abs().max().item() for a, b in zip(new[:3], ref[:3])) <= (1e-4 if prec == 'f16' else 5e-4), (prec, B)


def test_layer1_basic_block_in_one_launch_is_bit_identical_to_two_launches(dev):
    """csrc/conv_fb.hip (round 5): a BasicBlock of layer1 as ONE launch -- conv1 + ReLU into LDS on the 18 x 18 pixels conv2 needs
    (zeros outside the image), conv2 + residual + ReLU out of LDS, both convolutions' weights stationary in registers -- against the
    two conv_wres.hip launches per block it replaces (`SPSP_DEBUG_PLAN_FB` = 0).  Same products in the same order with the same
    rounding points (the intermediate map is rounded to the element type either way): layer1's map and everything behind it equal
    bit for bit, at B = 32 (two tiles per workgroup), 22 (352 tiles: ragged ranges), 3 and 1, both element types, chip-filling and
    narrow grids (eight tiles per resident weight set)."""
    from soccerplayershapepose_amd import runtime as rt
    from soccerplayershapepose_amd.models.regressor import SingleInputRegressor
    sd = S.regressor_state_dict(seed=0)
    for prec, batches in (('f16', (32, 22, 3, 1)), ('bf16', (32, 5))):
        for B in batches:
            reg = SingleInputRegressor(18, 18, 3, mean_params=S.mean_params(), max_batch=B, precision=prec).load_state_dict(sd).to(dev)
            r = rt.get(dev)
            x, _ = S.proxy_crops(B, seed=70 + B)
            native = reg.pack(torch.from_numpy(x).to(dev))
            got = {}
            for fb, div in ((0, 1), (1, 1), (1, 4), (1, 16)):
                r.check(r.lib.spsp_regressor_debug(reg.handle, _abi.DEBUG_PLAN_FB, fb), 'debug')
                r.check(r.lib.spsp_regressor_debug(reg.handle, _abi.DEBUG_PLAN_WRES_CUS, div), 'debug')
                out = [t.clone() for t in reg(native)] + [reg.read_stage(s_, B).clone() for s_ in (2, 5)]
                torch.cuda.synchronize()
                reg.check_status()
                got[(fb, div)] = out
            for key, other in got.items():
                for a, b in zip(got[(0, 1)], other):
                    assert torch.isfinite(a).all() and torch.equal(a, b), (prec, B, key, (a - b).abs().max().item())


def test_narrow_persistent_grids_of_plan_throughput_are_bit_identical(dev):
    """Plan "throughput" launches the persistent kernels narrow -- layer1 (conv_wres.hip) on a quarter of the CUs, the stem
    (conv_stem.hip) in bands of up to 64 conv rows, layer2 / layer3 (conv_p4.hip) on 64 workgroups that walk 4 / 2 tiles each
    with the DMA stream running on across the tile boundary -- and plan "latency" chip-filling (`SPSP_DEBUG_PLAN_WRES_CUS`,
    `SPSP_DEBUG_PLAN_STEM_BAND`, `SPSP_DEBUG_PLAN_P4_CUS`).  Which workgroup computes a tile changes, nothing about the tile: the pooled stem map, layer1's
    map, layer2's / layer3's maps and the regressor outputs are equal bit for bit at B = 32 (64-row bands), 22 (32-row bands; 176
    conv_p4 tiles on 64 / 32 workgroups: uneven tile counts) and 3, both element types;
    and the two named plans differ only by layer2's (conv_rs.hip against conv_p4.hip, round 5) and layer3's (conv_p4.hip against
    conv_patch.hip) summation order: the map in front of layer2 is equal bit for bit."""
    from soccerplayershapepose_amd import runtime as rt
    from soccerplayershapepose_amd.models.regressor import SingleInputRegressor
    sd = S.regressor_state_dict(seed=0)
    for prec, batches in (('f16', (32, 22, 3)), ('bf16', (32,))):
        for B in batches:
            reg = SingleInputRegressor(18, 18, 3, mean_params=S.mean_params(), max_batch=B, precision=prec).load_state_dict(sd).to(dev)
            r = rt.get(dev)
            x, _ = S.proxy_crops(B, seed=40 + B)
            native = reg.pack(torch.from_numpy(x).to(dev))
            got = []
            r.check(r.lib.spsp_regressor_debug(reg.handle, _abi.DEBUG_PLAN_P4, 6), 'debug')
            r.check(r.lib.spsp_regressor_debug(reg.handle, _abi.DEBUG_PLAN_RS, 0), 'debug')
            for div, band, p4div in ((1, 0, 1), (4, 64, 4), (2, 32, 2), (16, 16, 8)):
                r.check(r.lib.spsp_regressor_debug(reg.handle, _abi.DEBUG_PLAN_WRES_CUS, div), 'debug')
                r.check(r.lib.spsp_regressor_debug(reg.handle, _abi.DEBUG_PLAN_STEM_BAND, band), 'debug')
                r.check(r.lib.spsp_regressor_debug(reg.handle, _abi.DEBUG_PLAN_P4_CUS, p4div), 'debug')
                out = [t.clone() for t in reg(native)] + [reg.read_stage(s_, B).clone() for s_ in (1, 2, 3, 4)]
                torch.cuda.synchronize()
                reg.check_status()
                got.append(out)
            for other in got[1:]:
                for a, b in zip(got[0], other):
                    assert torch.isfinite(a).all() and torch.equal(a, b)
            if B == 32:
                lat = [t.clone() for t in reg.set_plan('latency')(native)] + [reg.read_stage(2, B).clone()]
                thr = [t.clone() for t in reg.set_plan('throughput')(native)] + [reg.read_stage(2, B).clone()]
                torch.cuda.synchronize()
                assert torch.equal(lat[-1], thr[-1])                      # up to layer1 the plans run the same arithmetic
                assert all(float((a - b).abs().max()) <= (1e-4 if prec == 'f16' else 1e-3) for a, b in zip(lat[:-1], thr[:-1]))


def test_fit_tree_walks_are_bit_identical_to_the_level_sweeps(smpl_gpu, dev):
    """The fit's kinematic chain and reverse sweep run as register-resident walks over quads of lanes (csrc/fit_core.hpp:
    fit_chain_walk, fit_reverse_walk_a / _b -- DPP row broadcasts in place of eight LDS hand-offs per sweep); the level sweeps
    they replaced remain for models with another tree and are selected by `spsp_smpl_debug(SPSP_DEBUG_FIT_SCHEDULE, 0)`; 1 = the walks
    in the same ten-barrier schedule.  Same
    operations in the same order: every iterate of both variants, all 229 parameters of 7 players over 50 iterations, bit for bit."""
    from soccerplayershapepose_amd.player_recon import fit_players
    B = 7
    R, b, cam = S.pose_params(B, seed=77, body_deg=25, global_deg=40)
    out = smpl_gpu(body_pose=torch.from_numpy(R[:, 1:]).to(dev), global_orient=torch.from_numpy(R[:, :1]).to(dev), betas=torch.from_numpy(b).to(dev))
    tgt = S.fit_targets_from_joints(out.joints.cpu().numpy(), cam * np.array([1.1, 1, 1], np.float32), seed=5)
    runs = []
    try:
        for sched in (1, 0):
            smpl_gpu.debug(_abi.DEBUG_FIT_SCHEDULE, sched)
            for variant in (0, 1):
                Rg, bg, cg = (torch.from_numpy(a).to(dev).contiguous() for a in (R, b, cam))
                hist = fit_players(smpl_gpu, Rg, bg, cg, tgt, variant=variant, iters=50, return_trajectory=True)
                torch.cuda.synchronize()
                runs.append([Rg.clone(), bg.clone(), cg.clone()] + [h.clone() for h in hist])
    finally:
        smpl_gpu.debug(_abi.DEBUG_FIT_SCHEDULE, 2)
    for walk, level in ((runs[0], runs[2]), (runs[1], runs[3])):
        assert len(walk) == len(level) == 6
        for x, y in zip(walk, level):
            assert torch.equal(x, y)
    assert not torch.equal(runs[1][0], torch.from_numpy(R).to(dev))         # (the fit did move the pose)


def test_fit_schedule_v2_follows_the_ieee_ten_barrier_schedule(smpl_gpu, dev):
    """csrc/fit.hip fit_kernel2 (the default: eight barriers per iteration, Adam steps in the lanes that compute the gradients, waves
    1 and 3 working beside wave 0's tree walks; since round 5 with the hardware's 1-ulp square root / reciprocal in the Adam step
    and the l2e metric and table-driven bias corrections -- csrc/fit_core.hpp) against the schedule of rounds 1-3, which keeps
    every IEEE division and square root (`spsp_smpl_debug(SPSP_DEBUG_FIT_SCHEDULE, 1)`).  Bit-identical through round 4 (and
    still, on the host: tests/emul); on the GPU now held to the oracle's own bar: every iterate of both fit variants and both select
    modes within 5e-5 on the raw parameters, loss / l2e histories within 1e-4 relative, 9 players over 50 iterations."""
    from soccerplayershapepose_amd.player_recon import fit_players
    B = 9
    R, b, cam = S.pose_params(B, seed=91, body_deg=25, global_deg=40)
    out = smpl_gpu(body_pose=torch.from_numpy(R[:, 1:]).to(dev), global_orient=torch.from_numpy(R[:, :1]).to(dev), betas=torch.from_numpy(b).to(dev))
    tgt = S.fit_targets_from_joints(out.joints.cpu().numpy(), cam * np.array([1.1, 1, 1], np.float32), seed=6)
    runs = {}
    try:
        for sched in ('2', '1'):
            smpl_gpu.debug(_abi.DEBUG_FIT_SCHEDULE, int(sched))
            for variant in (0, 1):
                for select in (_abi.FIT_SELECT_LAST, _abi.FIT_SELECT_BEST_L2E):
                    Rg, bg, cg = (torch.from_numpy(a).to(dev).contiguous() for a in (R, b, cam))
                    hist = fit_players(smpl_gpu, Rg, bg, cg, tgt, variant=variant, iters=50, return_trajectory=True, select_mode=select)
                    torch.cuda.synchronize()
                    runs[(sched, variant, select)] = [Rg.clone(), bg.clone(), cg.clone()] + [h.clone() for h in hist]
    finally:
        smpl_gpu.debug(_abi.DEBUG_FIT_SCHEDULE, 2)
    worst = 0.0
    for variant in (0, 1):
        for select in (_abi.FIT_SELECT_LAST, _abi.FIT_SELECT_BEST_L2E):
            new, old = runs[('2', variant, select)], runs[('1', variant, select)]
            assert len(new) == len(old) == 6
            for i, (x, y) in enumerate(zip(new, old)):
                assert torch.isfinite(x).all()
                if i in (3, 4):          # loss, l2e histories
                    assert ((x - y).abs() <= 1e-4 * y.abs() + 1e-9).all()
                else:                    # R, betas, cam, and the trajectory of all 229 parameters
                    worst = max(worst, (x - y).abs().max().item())
    print(f'fit v2 (1-ulp sqrt / rcp) vs the IEEE schedule: worst parameter deviation over all iterates {worst:.2e}')
    assert worst < 5e-5
    assert not torch.equal(runs[('2', 1, _abi.FIT_SELECT_LAST)][0], torch.from_numpy(R).to(dev))


# ---------------------------------------------------------------- silhouette term and the reference's exact best-iterate rule (8f-4, a11)
def test_silhouette_render_and_metric_against_restated_neural_renderer(smpl_gpu, smpl_model_np, dev):
    """csrc/silhouette.hip vs oracle/silhouette.py (neural_renderer's projection + silhouette rasterisation restated):
    the 512 x 512 image of three posed meshes -- one pushed half out of the frame, one far away and small -- sample for
    sample, and the tracker's silhouette metric against a detected silhouette."""
    from soccerplayershapepose_amd.renderer import SilhouetteRenderer
    from oracle import silhouette as OSil, geometry as G
    B = 3
    R, betas, cam = S.pose_params(B, seed=23, body_deg=30, global_deg=50)
    cam = cam.copy()
    cam[1, 1] += 0.9                       # half out of the frame
    cam[2, 0] *= 0.35                      # far away
    Rt, bt = torch.from_numpy(R).to(dev).contiguous(), torch.from_numpy(betas).to(dev).contiguous()
    verts = smpl_gpu(full_pose=Rt, betas=bt, return_joints=False).vertices
    trans = G.wp_to_translation(torch.from_numpy(cam)).numpy()
    rend = SilhouetteRenderer(smpl_gpu.faces, max_batch=8).to(dev)
    target = np.zeros((512, 512), np.uint8)
    target[60:420, 150:380] = 1
    out = rend(verts, trans, target=target)
    assert out.silhouettes.shape == (B, 512, 512) and out.metric.shape == (B,)
    vh = verts.cpu().numpy()
    for b in range(B):
        ref = OSil.render_silhouette(vh[b], smpl_model_np['faces'], trans[b])
        got = out.silhouettes[b].cpu().numpy()
        assert ref.sum() > 200, 'mesh must be visible'
        bad = int((got != ref).sum())
        assert bad == 0, f'mesh {b}: {bad} of 262144 pixels differ from the restated rasteriser'     # same fp32 operations, none fused: bit for bit
        m_ref = OSil.silhouette_metric(ref, target)
        assert abs(out.metric[b].item() - m_ref) <= 2e-5 * m_ref              # fp32 block sums against a float64 sum
    # a per-mesh target gives the same numbers as the shared one; rendering without a target gives the same image
    t3 = torch.from_numpy(np.repeat(target[None], B, axis=0)).to(dev)
    out3 = rend(verts, trans, target=t3, return_image=False)
    assert torch.equal(out3.metric, out.metric) and torch.equal(rend(verts, trans).silhouettes, out.silhouettes)
    with pytest.raises(ValueError):
        rend(verts[:, :100], trans)


def test_reference_best_iterate_rule_with_the_rendered_silhouette_metric(smpl_gpu, oracle_smpl, smpl_model_np, dev):
    """a11 as the reference applies it (`PR/player_recon.py:1259-1271`): an iterate is saved iff joints2D_l2es AND the
    silhouette metric of its render are both <= their values at the previous save.  GPU: recorded trajectory -> SMPL
    forward + render + metric per iterate -> spsp_fit_select; oracle: autograd trajectory -> restated renderer -> the
    same rule.  A large learning rate makes both metrics non-monotonic, so the rule has something to decide."""
    from soccerplayershapepose_amd.player_recon import single_view_optimization, FIT_SELECT_REFERENCE
    from soccerplayershapepose_amd.renderer import SilhouetteRenderer
    from oracle import fit as OF, silhouette as OSil, smpl as OS, geometry as G
    B, iters, lr = 2, 10, 0.03
    R, betas, cam = S.pose_params(B, seed=61, body_deg=20, global_deg=30)
    vt, jt = OS.smpl_forward_np(smpl_model_np, R, betas)
    tgt = S.fit_targets_from_joints(jt, cam, seed=62)
    trans_t = G.wp_to_translation(torch.from_numpy(cam)).numpy()
    sil_t = np.stack([(OSil.render_silhouette(vt[b].astype(np.float32), smpl_model_np['faces'], trans_t[b]) > 0.5) for b in range(B)]).astype(np.uint8)
    Rn, bn, _ = S.pose_params(B, seed=63, body_deg=10, global_deg=12)
    R0 = (Rn.astype(np.float64) @ R.astype(np.float64)).astype(np.float32)
    b0, cam0 = (betas + bn / 2).astype(np.float32), (cam * np.array([1.08, 1, 1])).astype(np.float32)
    rend = SilhouetteRenderer(smpl_gpu.faces, max_batch=8).to(dev)
    out = single_view_optimization(smpl_gpu, R0[:, 1:], R0[:, :1], b0, cam0, tgt, interation=iters, lr=lr,
                                   select_mode=FIT_SELECT_REFERENCE, silhouette=sil_t, renderer=rend)
    for b in range(B):
        rec = {}
        OF.fit_pose_shape(oracle_smpl, torch.from_numpy(R0[b:b + 1, 1:]), torch.from_numpy(R0[b:b + 1, :1]), torch.from_numpy(b0[b:b + 1]),
                          tgt[b], torch.from_numpy(cam0[b:b + 1]), lr=lr, iters=iters, record=rec)
        # parameters each iteration EVALUATED: the initial ones, then the result of the previous step
        pre = [(R0[b:b + 1, 1:], R0[b:b + 1, :1], b0[b:b + 1], cam0[b:b + 1])] + \
              [(rec['body'][i], rec['G'][i], rec['betas'][i], rec['cam'][i]) for i in range(iters - 1)]
        sil_o = []
        for body, Gm, be, ca in pre:
            full = np.concatenate([np.asarray(Gm).reshape(1, 1, 3, 3), np.asarray(body).reshape(1, 23, 3, 3)], axis=1)
            v, _ = OS.smpl_forward_np(smpl_model_np, full.astype(np.float32), np.asarray(be, np.float32))
            t = G.wp_to_translation(torch.from_numpy(np.asarray(ca, np.float32))).numpy()[0]
            sil_o.append(OSil.silhouette_metric(OSil.render_silhouette(v[0].astype(np.float32), smpl_model_np['faces'], t), sil_t[b]))
        np.testing.assert_allclose(out.l2e_history[b].cpu().numpy(), rec['l2e'], rtol=2e-3)
        np.testing.assert_allclose(out.silhouette_metric_history[b].cpu().numpy(), sil_o, rtol=5e-3)
        best = OSil.select_best(rec['l2e'], sil_o)
        assert int(out.best_iter[b]) == best, (int(out.best_iter[b]), best, rec['l2e'], sil_o)
        body, Gm, be, ca = pre[best]
        np.testing.assert_allclose(out.body_pose[b].cpu().numpy(), np.asarray(body)[0], atol=2e-4)
        np.testing.assert_allclose(out.cam_wp[b].cpu().numpy(), np.asarray(ca)[0], atol=2e-4)
        np.testing.assert_allclose(out.betas[b].cpu().numpy(), np.asarray(be)[0], atol=2e-4)
    with pytest.raises(ValueError):
        single_view_optimization(smpl_gpu, R0[:, 1:], R0[:, :1], b0, cam0, tgt, select_mode=FIT_SELECT_REFERENCE)


def test_smpl_vertices_vjp_against_autograd(smpl_gpu, oracle_smpl, dev):
    """csrc/smpl_vjp.hip -- dL/dvertices -> (dL/dR, dL/dbetas), what `loss.backward()` does to the mesh when the silhouette is
    in the loss -- against torch.autograd through the oracle SMPL on random non-orthonormal matrices; B = 9 (ragged against the
    skin kernel's 8-player workgroups).  Tolerance: 2e-5 of the largest gradient entry (fp32 sums over 6890 vertices)."""
    B = 9
    R, betas, _ = S.pose_params(B, seed=71, body_deg=35, global_deg=60)
    R = R + 0.05 * S.hash_normal(R.size, 'vjpR').reshape(R.shape).astype(np.float32)
    G = S.hash_normal(B * 6890 * 3, 'vjpG').reshape(B, 6890, 3).astype(np.float32)
    Rt, bt = torch.from_numpy(R).requires_grad_(True), torch.from_numpy(betas).requires_grad_(True)
    v = oracle_smpl(body_pose=Rt[:, 1:], global_orient=Rt[:, :1], betas=bt).vertices
    (v * torch.from_numpy(G)).sum().backward()
    gR, gb = smpl_gpu.vertices_vjp(torch.from_numpy(R).to(dev), torch.from_numpy(betas).to(dev), torch.from_numpy(G).to(dev))
    sR, sb = Rt.grad.abs().max().item(), bt.grad.abs().max().item()
    assert (gR.cpu() - Rt.grad).abs().max().item() < 2e-5 * sR, ((gR.cpu() - Rt.grad).abs().max().item(), sR)
    assert (gb.cpu() - bt.grad).abs().max().item() < 2e-5 * sb, ((gb.cpu() - bt.grad).abs().max().item(), sb)


def _shifted_silhouette_case(smpl_model_np, B, seed):
    """B posed meshes and, as the detected silhouette of each, the render of the same mesh slightly moved and re-posed."""
    from oracle import smpl as OS, silhouette as OSil, geometry as G
    R, betas, cam = S.pose_params(B, seed=seed, body_deg=25, global_deg=40)
    v, j = OS.smpl_forward_np(smpl_model_np, R, betas)
    Rn, bn, _ = S.pose_params(B, seed=seed + 1, body_deg=6, global_deg=8)
    R2 = (Rn.astype(np.float64) @ R.astype(np.float64)).astype(np.float32)
    v2, j2 = OS.smpl_forward_np(smpl_model_np, R2, (betas + bn / 2).astype(np.float32))
    cam2 = (cam * np.array([1.06, 1, 1]) + np.array([0, 0.02, -0.015])).astype(np.float32)
    trans, trans2 = G.wp_to_translation(torch.from_numpy(cam)).numpy(), G.wp_to_translation(torch.from_numpy(cam2)).numpy()
    target = np.stack([(OSil.render_silhouette(v2[b].astype(np.float32), smpl_model_np['faces'], trans2[b]) > 0.5) for b in range(B)]).astype(np.uint8)
    return R, betas, cam, v.astype(np.float32), trans.astype(np.float32), target, j2, cam2


def test_silhouette_loss_and_gradient_against_restated_neural_renderer_backward(smpl_gpu, smpl_model_np, dev):
    """csrc/silhouette_grad.hip vs oracle/silhouette_grad.py (neural_renderer's rasteriser WITH its approximate backward pass,
    restated; parity unpinned): the image of the ownership raster equals the coverage raster's, the loss sum, and
    dL/dvertices / dL/dtranslation of L = -S w against torch.autograd through the restated renderer.  The gradient is a sum
    of +-diff / dist terms of up to 1e6 / 1e-4 each: tolerance 1e-4 of its largest entry."""
    from soccerplayershapepose_amd.renderer import SilhouetteRenderer
    from oracle import silhouette_grad as OG
    B = 2
    R, betas, cam, v, trans, target, _, _ = _shifted_silhouette_case(smpl_model_np, B, seed=81)
    rend = SilhouetteRenderer(smpl_gpu.faces, max_batch=4).to(dev)
    w = np.array([1e6, 3.0], np.float32)
    out = rend.loss_grad(torch.from_numpy(v).to(dev), trans, target, weight=w, return_image=True)
    plain = rend(torch.from_numpy(v).to(dev), trans, target=target)
    assert torch.equal(out.silhouettes, plain.silhouettes)
    np.testing.assert_allclose(-out.sil_sum.cpu().numpy(), plain.metric.cpu().numpy(), rtol=1e-5)
    for b in range(B):
        vt = torch.from_numpy(v[b]).requires_grad_(True)
        tt = torch.from_numpy(trans[b]).requires_grad_(True)
        img = OG.render_silhouette_torch(vt, smpl_model_np['faces'], tt)
        tg = torch.from_numpy(target[b].astype(np.float32))
        S_ = torch.sum(img * torch.log(tg + 1e-6) + (1 - img) * torch.log(1 - tg + 1e-6))
        (-S_ * float(w[b])).backward()
        assert np.array_equal(img.detach().numpy(), out.silhouettes[b].cpu().numpy())
        assert abs(out.sil_sum[b].item() - S_.item()) <= 2e-5 * abs(S_.item())
        gv, gt = out.grad_vertices[b].cpu(), out.grad_translation[b].cpu()
        sc = vt.grad.abs().max().item()
        assert sc > 0
        dv = (gv - vt.grad).abs().amax(dim=1)
        rel = (gv - vt.grad).norm().item() / vt.grad.norm().item()
        print(f'mesh {b}: dL/dvertices relative L2 deviation {rel:.2e}, max {dv.max().item():.3e} of {sc:.3e}, vertices beyond 1e-4 of the scale: {int((dv > 1e-4 * sc).sum())}')
        bad = dv.max().item()
        assert bad < 1e-4 * sc, f'mesh {b}: dL/dvertices deviates by {bad} (largest entry {sc})'
        assert (gt - tt.grad).abs().max().item() < 1e-4 * tt.grad.abs().max().item(), (gt, tt.grad)
    with pytest.raises(ValueError):
        rend.loss_grad(torch.from_numpy(v).to(dev), trans, target[:, :100], weight=w)


def test_fit_with_the_silhouette_in_the_loss_against_the_autograd_oracle(smpl_gpu, oracle_smpl, smpl_model_np, dev):
    """`single_view_optimization_test` (`PR/player_recon.py:1429-1557`): joints2D + silhouette loss, the criterion's two
    log-variances in the optimiser, the silhouette term reaching the camera from the second epoch on.  GPU: one chain of
    launches per epoch (csrc/fit_sil.hip); oracle: torch.autograd through the oracle SMPL and the restated neural_renderer,
    torch.optim.Adam.  (i) the GRADIENT of the whole loss at the first two epochs -- identical parameters, resp. parameters equal
    to 5e-10: every piece of the backward chain (renderer, projection, SMPL, joints2D term, log-variances, camera through the
    translation) against autograd, within 5e-3 of the largest entry of each parameter group.  (Not tighter: the two sides' SMPL
    vertices differ in the last bit, and ONE term of the renderer's backward pass whose pixel distance falls below its eps =
    1e-4 is worth 3.45e6 w / 1e-4 = 3e10 on a gradient of 6e12 -- measured 1e-3.  Fed identical vertices the renderer's gradient
    agrees to 1e-7 and the SMPL reverse mode to 2e-5: the two tests above.)  (ii) the parameters after steps 1 and 2 within 5e-5 (an Adam step
    moves every parameter by lr = 1e-3 whatever its gradient: no sign may flip); (iii) the histories of loss, joints2D_l2es and
    silhouette metric over ten epochs, 2e-3 relative; (iv) after ten steps the parameters within 5e-3 and the log-variances within
    3e-4: the renderer's gradient is discontinuous in the vertices (pixel ownership, 1 / dist terms), so the 5e-10 left after the
    first step grows -- measured 2.4e-3 after nine; with identical inputs the gradient itself agrees to 1e-7 (the test above)."""
    from soccerplayershapepose_amd.player_recon import single_view_optimization_test
    from soccerplayershapepose_amd.renderer import SilhouetteRenderer
    from oracle import silhouette_grad as OG
    iters = 10
    R, betas, cam, v, trans, target, j2, cam2 = _shifted_silhouette_case(smpl_model_np, 1, seed=91)
    tgt_j = S.fit_targets_from_joints(j2, cam2, seed=92)
    rend = SilhouetteRenderer(smpl_gpu.faces, max_batch=4).to(dev)
    out = single_view_optimization_test(smpl_gpu, rend, R[:, 1:], R[:, :1], betas, cam, tgt_j, target, interation=iters, return_history=True)
    rec = {}
    ref = OG.fit_pose_shape_silhouette(oracle_smpl, smpl_model_np['faces'], torch.from_numpy(R[:, 1:]), torch.from_numpy(R[:, :1]),
                                       torch.from_numpy(betas), tgt_j[0], target[0], torch.from_numpy(cam), iters=iters, record=rec)
    ph = out.param_history[0].cpu().numpy()                      # [iters, 231]: the parameters each epoch EVALUATED
    np.testing.assert_allclose(out.l2e_history[0].cpu().numpy(), rec['l2e'], rtol=2e-3)
    np.testing.assert_allclose(out.silhouette_metric_history[0].cpu().numpy(), rec['sil_metric'], rtol=2e-3)
    np.testing.assert_allclose(out.loss_history[0].cpu().numpy(), rec['loss'], rtol=2e-3)
    gh = out.grad_history[0].cpu().numpy()
    for ep in (1, 2):
        want = np.concatenate([rec[f'grad_cam_{ep}'].reshape(-1), rec[f'grad_G_{ep}'].reshape(-1)])
        gtol = 5e-3          # of the largest entry of the group (see the docstring)
        np.testing.assert_allclose(gh[ep - 1][:12], want, rtol=0, atol=gtol * np.abs(want).max())
        body19 = np.concatenate([gh[ep - 1][12:219].reshape(23, 9)[i] for i in list(range(6)) + list(range(8, 21))])
        np.testing.assert_allclose(body19, rec[f'grad_body_{ep}'].reshape(-1), rtol=0, atol=gtol * np.abs(rec[f'grad_body_{ep}']).max())
        np.testing.assert_allclose(gh[ep - 1][219:229], rec[f'grad_betas_{ep}'].reshape(-1), rtol=0, atol=gtol * np.abs(rec[f'grad_betas_{ep}']).max())
        np.testing.assert_allclose(gh[ep - 1][229:231], rec[f'grad_lv_{ep}'], rtol=1e-4)
        if ep == 1:
            np.testing.assert_allclose(gh[0][:3], rec['grad_cam_1'].reshape(-1), rtol=1e-4)     # joints2D term alone: exact
    for ep, tol in ((1, 5e-5), (2, 5e-5), (9, 5e-3)):            # ph[ep] = the result of step ep (0-based: after ep steps)
        want = np.concatenate([rec['cam'][ep - 1].reshape(-1), rec['G'][ep - 1].reshape(-1), rec['body'][ep - 1].reshape(-1),
                               rec['betas'][ep - 1].reshape(-1), rec['lv'][ep - 1].reshape(-1)])
        dev_ = np.abs(ph[ep] - want)
        print(f'silhouette fit, after step {ep}: max parameter deviation {dev_.max():.2e} (parameter {int(dev_.argmax())}), {int((dev_ > tol).sum())} beyond {tol}')
        assert dev_.max() < tol, f'after step {ep}: {dev_.max()} at parameter {int(dev_.argmax())}'
    body, Gm, be, ca, lvj, lvs = ref['last']
    assert (out.body_pose[0].cpu() - body[0]).abs().max().item() < 5e-3 and (out.cam_wp[0].cpu() - ca[0]).abs().max().item() < 5e-3
    assert (out.betas[0].cpu() - be[0]).abs().max().item() < 5e-3 and (out.global_orient[0].cpu() - Gm[0]).abs().max().item() < 5e-3
    np.testing.assert_allclose(out.log_vars[0].cpu().numpy(), [lvj, lvs], atol=3e-4)
    assert rec['sil_metric'][-1] < rec['sil_metric'][0]           # and the silhouette term does pull the mesh into the target


def test_broad_view_optimization_against_the_autograd_oracle(smpl_gpu, oracle_smpl, smpl_model_np, dev):
    """`broad_view_optimization` (`PR/player_recon.py:2128-2229`): camera + global orientation, lr 1e-2, 100
    epochs, train pass then validation pass, the best validation epoch saved.  (i) the reference's sizes with the joints2D
    metric alone; (ii) 12 epochs under the reference's two-metric rule, the silhouette metric of every validation iterate
    rendered on both sides.  Tolerances: validation l2e histories 2e-3 relative, the saved parameters 2e-4 / 5e-5, the saved
    epoch equal."""
    from soccerplayershapepose_amd.player_recon import broad_view_optimization, FIT_SELECT_REFERENCE
    from soccerplayershapepose_amd.renderer import SilhouetteRenderer
    from oracle import fit as OF, smpl as OS, silhouette as OSil, geometry as G
    B = 2
    R, betas, cam = S.pose_params(B, seed=101, body_deg=20, global_deg=30)
    _, jt = OS.smpl_forward_np(smpl_model_np, R, betas)
    tgt = S.fit_targets_from_joints(jt, cam, seed=102)
    Rn, _, _ = S.pose_params(B, seed=103, body_deg=0, global_deg=12)
    G0 = (Rn[:, :1].astype(np.float64) @ R[:, :1].astype(np.float64)).astype(np.float32)     # what the regressor gives on the broad view
    cam0 = (cam * np.array([1.1, 1, 1]) + np.array([0, 0.03, -0.02])).astype(np.float32)
    out = broad_view_optimization(smpl_gpu, R[:, 1:], G0, betas, cam0, tgt)
    assert out.val_l2e_history.shape == (B, 100)
    for b in range(B):
        ref = OF.fit_broad_view(oracle_smpl, torch.from_numpy(R[b:b + 1, 1:]), torch.from_numpy(G0[b:b + 1]), torch.from_numpy(betas[b:b + 1]),
                                tgt[b], torch.from_numpy(cam0[b:b + 1]))
        np.testing.assert_allclose(out.val_l2e_history[b].cpu().numpy(), ref['val_l2e'], rtol=2e-3)
        assert int(out.best_epoch[b]) == ref['best_epoch']
        np.testing.assert_allclose(out.global_orient[b].cpu().numpy(), ref['best'][0][0].numpy(), atol=2e-4)
        np.testing.assert_allclose(out.cam_wp[b].cpu().numpy(), ref['best'][1][0].numpy(), atol=2e-4)
    assert torch.equal(out.body_pose.cpu(), torch.from_numpy(R[:, 1:])) and torch.equal(out.betas.cpu(), torch.from_numpy(betas))
    # (ii) the two-metric rule, a large learning rate so that it has something to decide
    vt, _ = OS.smpl_forward_np(smpl_model_np, R, betas)
    trans_t = G.wp_to_translation(torch.from_numpy(cam)).numpy()
    sil_t = np.stack([(OSil.render_silhouette(vt[b].astype(np.float32), smpl_model_np['faces'], trans_t[b]) > 0.5) for b in range(B)]).astype(np.uint8)
    rend = SilhouetteRenderer(smpl_gpu.faces, max_batch=4).to(dev)
    out = broad_view_optimization(smpl_gpu, R[:, 1:], G0, betas, cam0, tgt, interation=12, lr=0.05, select_mode=FIT_SELECT_REFERENCE,
                                  silhouette=sil_t, renderer=rend)
    for b in (range(B) if FULL_ORACLE else (0,)):        # (the oracle renders 24 silhouettes per player on the CPU: one player under the two-metric rule unless SPSP_TEST_FULL, both above)
        metric = lambda verts, c, b=b: OSil.silhouette_metric(OSil.render_silhouette(verts.astype(np.float32), smpl_model_np['faces'],
                                                               G.wp_to_translation(torch.from_numpy(c[None])).numpy()[0]), sil_t[b])
        ref = OF.fit_broad_view(oracle_smpl, torch.from_numpy(R[b:b + 1, 1:]), torch.from_numpy(G0[b:b + 1]), torch.from_numpy(betas[b:b + 1]),
                                tgt[b], torch.from_numpy(cam0[b:b + 1]), lr=0.05, iters=12, silhouette_metric=metric)
        np.testing.assert_allclose(out.val_l2e_history[b].cpu().numpy(), ref['val_l2e'], rtol=2e-3)
        np.testing.assert_allclose(out.val_silhouette_metric_history[b].cpu().numpy(), ref['val_sil'], rtol=5e-3)
        assert int(out.best_epoch[b]) == ref['best_epoch'], (int(out.best_epoch[b]), ref['best_epoch'], ref['val_l2e'], ref['val_sil'])
        np.testing.assert_allclose(out.cam_wp[b].cpu().numpy(), ref['best'][1][0].numpy(), atol=5e-4)
    with pytest.raises(ValueError):
        broad_view_optimization(smpl_gpu, R[:, 1:], G0, betas, cam0, tgt, select_mode=FIT_SELECT_REFERENCE)


# ---------------------------------------------------------------- multi-view fit (8f-3) and the on-disk records (a14)
@pytest.mark.parametrize('optimise_body_pose', [False, True])
def test_multi_view_fit_against_oracle(smpl_gpu, oracle_smpl, smpl_model_np, dev, optimise_body_pose, tmp_path):
    """`multi_view_optimization` on the GPU (one workgroup per player, all rounds / blocks / epochs in one launch)
    against the autograd restatement of `PR/player_recon.py:1691-1979`, two players with 3 and ... views handled as
    separate calls (V is per call), the whole path starting from and ending in the reference's npz records."""
    from soccerplayershapepose_amd import records
    from soccerplayershapepose_amd.player_recon import multi_view_optimization
    from oracle import fit_multiview as OMV
    from test_fit_multiview_emulation import multiview_case
    iters, rounds = 6, 2
    # (V = 3, 2, 4 with two players each passed in round 3; two players at V = 3, one at V = 2 and 4 keep the CPU oracle -- 48 V
    # autograd passes per player -- inside a minute on a slow host)
    for V, seed, nplayers in ((3, 31, 2 if (optimise_body_pose or FULL_ORACLE) else 1), (2, 7, 2 if FULL_ORACLE else 1), (4, 19, 2 if FULL_ORACLE else 1)):
        cases = [multiview_case(smpl_model_np, V, seed=seed + 100 * b) for b in range(nplayers)]
        # the single-view results live in npz records (a14): write them, load them, start the multi-view fit from them
        recs = []
        for b, views in enumerate(cases):
            row = []
            for v, (body, go, betas, trans, _) in enumerate(views):
                path = str(tmp_path / f'p{b}_V{V}_view{v}.npz')
                records.save_record(path, body, go, betas, trans)
                row.append(records.load_record(path))
            recs.append(row)
        stack = lambda k: np.stack([np.stack([r[k][0] for r in row]) for row in recs])
        j2d = np.stack([np.stack([v[4] for v in views]) for views in cases])
        out = multi_view_optimization(smpl_gpu, stack('body_pose'), stack('global_orient'), stack('betas'), stack('translation'), j2d,
                                      interation=iters, rounds=rounds, optimise_body_pose=optimise_body_pose, return_history=True)
        assert out.body_pose.shape == (nplayers, 23, 3, 3) and out.translation.shape == (nplayers, V, 3) and out.global_orient.shape == (nplayers, V, 1, 3, 3)
        for b, views in enumerate(cases):
            t = lambda k: torch.from_numpy(np.stack([v[k] for v in views]))
            ref = OMV.fit_multi_view(oracle_smpl, t(0), t(1), t(2), t(3), [v[4] for v in views], iters=iters, rounds=rounds,
                                     stale_pose_cat=not optimise_body_pose)
            np.testing.assert_allclose(out.val_l2e_history[b].cpu().numpy().reshape(-1), ref['val_history'], rtol=2e-4, atol=2e-3)
            np.testing.assert_allclose(out.cam_wp[b].cpu().numpy(), ref['cam_wp'][:, 0], atol=5e-5)
            np.testing.assert_allclose(out.global_orient[b, :, 0].cpu().numpy(), ref['global_orient'][:, 0, 0], atol=5e-5)
            np.testing.assert_allclose(out.betas[b].cpu().numpy(), ref['betas'][0], atol=5e-5)
            np.testing.assert_allclose(out.body_pose[b].cpu().numpy(), ref['body_pose'][0], atol=5e-5)
            np.testing.assert_allclose(out.translation[b].cpu().numpy(), ref['translation'][:, 0], rtol=2e-5, atol=5e-5)
            # ... and back to disk in the multi-view layout (:1976-1979)
            d = str(tmp_path / f'out_p{b}_V{V}_{int(optimise_body_pose)}')
            records.save_multiview_record(d, out.body_pose[b:b + 1], out.betas[b:b + 1],
                                          {f'view{v}.png': (out.translation[b, v:v + 1], out.global_orient[b, v][None]) for v in range(V)})
            bp, be, vs = records.load_multiview_record(d)
            assert np.array_equal(bp[0], out.body_pose[b].cpu().numpy()) and len(vs) == V
    with pytest.raises(ValueError):
        multi_view_optimization(smpl_gpu, np.zeros((1, 5, 23, 3, 3), np.float32), np.zeros((1, 5, 1, 3, 3), np.float32),
                                np.zeros((1, 5, 10), np.float32), np.ones((1, 5, 3), np.float32), np.zeros((1, 5, 17, 2), np.float32))


def test_multi_view_fit_at_the_reference_operating_point_50_epochs_3_rounds(smpl_gpu, oracle_smpl, smpl_model_np, dev):
    """The multi-view fit where the reference runs it: 50 epochs per block, 3 rounds (`S/global_var.py:79`,
    `PR/player_recon.py:1720`) = 300 epochs, 1 200 forward passes with V = 2 (2 400 with V = 4) -- one player against the autograd
    restatement.  Tolerances: run on the host against the same oracle (tests/emul), the kernel's phases end 3e-5 apart after
    the 300 epochs in the reference's mode (betas only) and 5e-4 with the body pose optimised (193 Adam-normalised parameters,
    900 steps: the sign sensitivity of DESIGN section 2 accumulates), validation history 2e-4 / 2e-3 relative (on the GPU 2 %
    at two of the 300 epochs -- 0.02 px; parameters 1.5e-3 / 4e-3: fp32 contraction differences grow along 900 sign-sensitive
    steps).  Asserted: 1e-4 / 1e-2 on parameters, 1e-3 / 3e-2 relative on the history, the record's l2e within 2 %."""
    from soccerplayershapepose_amd.player_recon import multi_view_optimization
    from oracle import fit_multiview as OMV
    from test_fit_multiview_emulation import multiview_case
    # (the reference's own mode -- betas only -- at its operating point; the body-pose mode ran here too in round 3 (V = 3: 4e-3 on the
    # parameters after 900 steps) and stays covered at 6 x 2 by test_multi_view_fit_against_oracle[True]: the CPU oracle of one
    # 50 x 3 case takes a minute)
    for V, seed, obp in ((2, 19, False),):      # (V = 4 passed in round 3: 2 400 oracle passes, 70-250 s depending on the host; V = 2: half)
        views = multiview_case(smpl_model_np, V, seed=seed)
        st = lambda k: np.stack([v[k][0] for v in views])[None]
        out = multi_view_optimization(smpl_gpu, st(0), st(1), st(2), st(3), np.stack([v[4] for v in views])[None],
                                      interation=50, rounds=3, optimise_body_pose=obp, return_history=True)
        t = lambda k: torch.from_numpy(np.stack([v[k] for v in views]))
        ref = OMV.fit_multi_view(oracle_smpl, t(0), t(1), t(2), t(3), [v[4] for v in views], iters=50, rounds=3, stale_pose_cat=not obp)
        hist = out.val_l2e_history[0].cpu().numpy().reshape(-1)
        assert hist.shape == (300,)
        np.testing.assert_allclose(hist, ref['val_history'], rtol=3e-2 if obp else 1e-3, atol=5e-3)
        devs = dict(cam=np.abs(out.cam_wp[0].cpu().numpy() - ref['cam_wp'][:, 0]).max(),
                    G=np.abs(out.global_orient[0, :, 0].cpu().numpy() - ref['global_orient'][:, 0, 0]).max(),
                    betas=np.abs(out.betas[0].cpu().numpy() - ref['betas'][0]).max(),
                    body=np.abs(out.body_pose[0].cpu().numpy() - ref['body_pose'][0]).max())
        print(f'multi-view 50 x 3, V = {V}, optimise_body_pose = {obp}: {devs}')
        assert max(devs.values()) < (1e-2 if obp else 1e-4), devs
        assert abs(float(hist.min()) - float(ref['best_val'])) < 2e-2 * float(ref['best_val'])      # the record itself: within 2 %


def test_multi_view_fit_under_the_reference_record_rule_and_shuffled_view_orders(smpl_gpu, oracle_smpl, smpl_model_np, dev):
    """`multi_view_optimization` as the reference decides its record (`PR/player_recon.py:1820, 1937`: mean joints2D_l2es AND the
    mean rendered-silhouette metric over the views) and visits the views (`random.shuffle` before every pass, `:1748, :1876`):
    per-block launches + renders on the GPU against the autograd oracle given the same drawn orders and the restated renderer.
    A large learning rate makes both metrics non-monotonic, so the rule and the block restarts have something to decide."""
    from soccerplayershapepose_amd.player_recon import multi_view_optimization, multi_view_orders, FIT_SELECT_REFERENCE
    from soccerplayershapepose_amd.renderer import SilhouetteRenderer
    from oracle import fit_multiview as OMV, silhouette as OSil, geometry as G, smpl as OS
    from test_fit_multiview_emulation import multiview_case
    V, iters, rounds, lr = 3, 4, 2, 0.02
    views = multiview_case(smpl_model_np, V, seed=57)
    orders = multi_view_orders(V, iters, rounds, seed=5)
    assert orders.shape == (rounds * 2 * iters, V) and len({tuple(o.tolist()) for o in orders}) > 1
    # detected silhouettes: every view's single-view result rendered, shifted a little
    sil = []
    for body, go, betas, trans, _ in views:
        v, _ = OS.smpl_forward_np(smpl_model_np, np.concatenate([go, body], axis=1).astype(np.float32), betas.astype(np.float32))
        sil.append(OSil.render_silhouette(v[0].astype(np.float32), smpl_model_np['faces'], (trans[0] + np.array([0.02, -0.02, 0.3])).astype(np.float32)) > 0.5)
    sil = np.stack(sil).astype(np.uint8)
    st = lambda k: np.stack([v[k][0] for v in views])[None]
    rend = SilhouetteRenderer(smpl_gpu.faces, max_batch=4).to(dev)
    out = multi_view_optimization(smpl_gpu, st(0), st(1), st(2), st(3), np.stack([v[4] for v in views])[None], interation=iters, rounds=rounds,
                                  lr=lr, optimise_body_pose=True, view_order=orders, select_mode=FIT_SELECT_REFERENCE, silhouettes=sil[None],
                                  renderer=rend)
    metric = lambda view, verts, c: OSil.silhouette_metric(OSil.render_silhouette(verts.astype(np.float32), smpl_model_np['faces'],
                                                            G.wp_to_translation(torch.from_numpy(c[None])).numpy()[0]), sil[view])
    t = lambda k: torch.from_numpy(np.stack([v[k] for v in views]))
    ref = OMV.fit_multi_view(oracle_smpl, t(0), t(1), t(2), t(3), [v[4] for v in views], lr=lr, iters=iters, rounds=rounds, stale_pose_cat=False,
                             view_orders=orders.numpy(), silhouette_metric=metric)
    np.testing.assert_allclose(out.val_l2e_history[0].cpu().numpy().reshape(-1), ref['val_history'], rtol=2e-3, atol=2e-2)
    np.testing.assert_allclose(out.val_silhouette_metric_history[0].cpu().numpy().reshape(-1), ref['sil_history'], rtol=5e-3)
    assert abs(float(out.record_l2e[0]) - ref['best_val']) < 2e-3 * ref['best_val'] + 2e-2
    assert abs(float(out.record_silhouette_metric[0]) - ref['best_sil']) < 5e-3 * ref['best_sil']
    np.testing.assert_allclose(out.cam_wp[0].cpu().numpy(), ref['cam_wp'][:, 0], atol=2e-3)
    np.testing.assert_allclose(out.betas[0].cpu().numpy(), ref['betas'][0], atol=2e-3)
    np.testing.assert_allclose(out.body_pose[0].cpu().numpy(), ref['body_pose'][0], atol=2e-3)
    np.testing.assert_allclose(out.global_orient[0, :, 0].cpu().numpy(), ref['global_orient'][:, 0, 0], atol=2e-3)
    # the ordered single-launch form (joints2D_l2es record) against the oracle with the same orders
    out2 = multi_view_optimization(smpl_gpu, st(0), st(1), st(2), st(3), np.stack([v[4] for v in views])[None], interation=iters, rounds=rounds,
                                   lr=lr, optimise_body_pose=True, view_order=orders, return_history=True)
    ref2 = OMV.fit_multi_view(oracle_smpl, t(0), t(1), t(2), t(3), [v[4] for v in views], lr=lr, iters=iters, rounds=rounds, stale_pose_cat=False,
                              view_orders=orders.numpy())
    np.testing.assert_allclose(out2.val_l2e_history[0].cpu().numpy().reshape(-1), ref2['val_history'], rtol=2e-3, atol=2e-2)
    np.testing.assert_allclose(out2.cam_wp[0].cpu().numpy(), ref2['cam_wp'][:, 0], atol=2e-3)
    with pytest.raises(ValueError):
        multi_view_optimization(smpl_gpu, st(0), st(1), st(2), st(3), np.stack([v[4] for v in views])[None], interation=iters, rounds=rounds,
                                view_order=orders[:3])


def test_single_view_fit_warm_started_from_its_own_record_continues_the_trajectory(smpl_gpu, smpl_model_np, dev, tmp_path):
    """a14 as a warm start (`PR/player_recon.py:1673-1689`): fit 20 iterations, save the reference-layout record, load it,
    and fit 1 more iteration from it == what a fresh optimiser does from the same parameters held in memory (the record
    stores translation, so the camera goes through translation -> cam_wp: equal to fp32 rounding of that round trip)."""
    from soccerplayershapepose_amd import records
    from soccerplayershapepose_amd.player_recon import single_view_optimization
    from oracle import smpl as OS
    B = 3
    R, betas, cam = S.pose_params(B, seed=41, body_deg=20, global_deg=30)
    _, j = OS.smpl_forward_np(smpl_model_np, R, betas)
    tgt = S.fit_targets_from_joints(j, cam * np.array([1.1, 1, 1], np.float32), seed=42)
    Rn, bn, _ = S.pose_params(B, seed=43, body_deg=8, global_deg=10)
    R0 = (Rn.astype(np.float64) @ R.astype(np.float64)).astype(np.float32)
    fit = single_view_optimization(smpl_gpu, R0[:, 1:], R0[:, :1], betas + bn / 3, cam, tgt, interation=20)
    paths = records.save_records(str(tmp_path / 'single'), [f'{i}.png' for i in range(B)], fit)
    assert [os.path.basename(p) for p in paths] == ['0.png.npz', '1.png.npz', '2.png.npz']
    full, be, cam_ws = records.warm_start(paths, dev)
    assert full.shape == (B, 24, 3, 3) and be.shape == (B, 10) and cam_ws.shape == (B, 3)
    assert torch.equal(full[:, 1:], fit.body_pose) and torch.equal(full[:, :1], fit.global_orient) and torch.equal(be, fit.betas)
    np.testing.assert_allclose(cam_ws.cpu().numpy(), fit.cam_wp.cpu().numpy(), rtol=3e-6)
    a = single_view_optimization(smpl_gpu, full[:, 1:], full[:, :1], be, cam_ws, tgt, interation=1)
    b = single_view_optimization(smpl_gpu, fit.body_pose, fit.global_orient, fit.betas, fit.cam_wp, tgt, interation=1)
    np.testing.assert_allclose(a.body_pose.cpu().numpy(), b.body_pose.cpu().numpy(), atol=2e-6)
    np.testing.assert_allclose(a.cam_wp.cpu().numpy(), b.cam_wp.cpu().numpy(), atol=2e-5)


# ---------------------------------------------------------------- regressor (a2-a4)
_STAGE_ORACLE = {}


@pytest.mark.parametrize('precision', ['bf16', 'f16'])
# 3: an odd batch: layer4 (tiles = pairs of 8 x 8 images) runs on B + 1 crops, the extra one never read; 6: layer4 in whole 128-pixel tiles (two 8 x 8 images each) that are no whole
# 256-pixel ones, as for the 22-player frame, and the stem's band height by cost; 8: every patch kernel incl. the 8x8 one; 32: the
# BENCHMARK plan (all 32 crops checked); '32-contended': the same while a second regressor keeps the chip busy from another stream,
# so the persistent / dynamically scheduled kernels are checked against the oracle under CU contention, not only against themselves
# '32-throughput': the launch plan PlayerPipeline runs (narrow persistent grids; layer2 on conv_rs.hip, layer3 on conv_p4.hip)
@pytest.mark.parametrize('B', [3, 6, 8, 32, '32-contended', '32-throughput'])
def test_backbone_stages_against_bf16_emulating_oracle(regressor_gpu, regressor_gpu_f16, regressor_sd_np, dev, B, precision):
    """Every stage of the backbone against the oracle rounding weights and stored activations to the SAME element
    type at the same points (bf16: precision 0; fp16: precision 1)."""
    from oracle import regressor as OR
    contended = B == '32-contended'
    throughput = B == '32-throughput'
    B = 32 if contended or throughput else B
    x, _ = S.proxy_crops(B, seed=4)
    regressor_gpu = regressor_gpu_f16 if precision == 'f16' else regressor_gpu
    ulp = 1.0 if precision == 'bf16' else 0.125          # fp16 has 3 more mantissa bits
    xt = torch.from_numpy(x).to(dev)
    if contended:
        from soccerplayershapepose_amd.models.regressor import SingleInputRegressor
        other = SingleInputRegressor(18, 18, 3, mean_params=S.mean_params(), max_batch=32, precision=precision)
        other.load_state_dict(regressor_sd_np).to(dev)
        native_o = other.pack(xt)
        side = torch.cuda.Stream()
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            for _ in range(6):
                other(native_o)
    if throughput:
        regressor_gpu.set_plan('throughput')
    try:
        cam, pose, shape, feats = regressor_gpu(xt, return_feats=True)
        torch.cuda.synchronize()
    finally:
        regressor_gpu.set_plan('latency')                 # the fixture is shared: back to a handle's default
    if contended:
        with torch.cuda.stream(side):
            for _ in range(2):
                other(native_o)
        torch.cuda.synchronize()
    if (B, precision) not in _STAGE_ORACLE:                # the three B = 32 cases share one CPU pass of the oracle
        _STAGE_ORACLE[(B, precision)] = OR.encoder_forward(regressor_sd_np, x, emulate_bf16=precision, return_stages=True)
    f_emu, stages = _STAGE_ORACLE[(B, precision)]
    # stage 0 (the 128 x 128 x 64 map after conv1 + ReLU) does not exist on the GPU: the stem kernel pools in registers
    names = ['conv1', 'maxpool', 'layer1', 'layer2', 'layer3', 'layer4']
    with pytest.raises(ValueError):
        regressor_gpu.read_stage(0, B)
    for si, n in list(enumerate(names))[1:]:
        got = regressor_gpu.read_stage(si, B).cpu()
        ref = stages[n]
        assert got.shape == ref.shape
        err = (got - ref).abs().max().item()
        scale = ref.abs().max().item()
        # same bf16 rounding points, different fp32 summation order: a few 1-ulp(bf16) flips at most
        assert err <= 0.02 * ulp * scale, f'stage {n}: max err {err} vs scale {scale}'
        assert (got - ref).abs().mean().item() <= 5e-4 * ulp * scale, f'stage {n}: mean err'   # (bf16-vs-fp32 gap is ~2.4e-3)
    np.testing.assert_allclose(feats.cpu().numpy(), f_emu.numpy(), rtol=0, atol=3e-3 * ulp * f_emu.abs().max().item())
    c_e, p_e, s_e = OR.ief_forward(regressor_sd_np, feats.cpu(), S.mean_params())
    np.testing.assert_allclose(cam.cpu().numpy(), c_e.numpy(), atol=2e-5)       # IEF itself is fp32
    np.testing.assert_allclose(pose.cpu().numpy(), p_e.numpy(), atol=2e-5)
    np.testing.assert_allclose(shape.cpu().numpy(), s_e.numpy(), atol=2e-5)


def test_regressor_against_reference_golden_outputs(regressor_gpu, golden, dev):
    """bf16 MFMA backbone vs the reference module's fp32 outputs (tests/golden/regressor.npz).
    Stated tolerance: 5e-3 on (cam, pose6d, shape) -- bf16 activations/weights, fp32 accumulate."""
    g = golden('regressor')
    x, _ = S.proxy_crops(int(g['crops_batch']), seed=int(g['crops_seed']))
    cam, pose, shape, feats = regressor_gpu(torch.from_numpy(x).to(dev), return_feats=True)
    assert np.abs(feats.cpu().numpy() - g['feats']).max() < 0.02 * np.abs(g['feats']).max()
    assert np.abs(cam.cpu().numpy() - g['cam']).max() < 5e-3
    assert np.abs(pose.cpu().numpy() - g['pose']).max() < 5e-3
    assert np.abs(shape.cpu().numpy() - g['shape']).max() < 5e-3
    for si, n in ((1, 'maxpool'), (2, 'layer1'), (3, 'layer2'), (4, 'layer3'), (5, 'layer4')):
        got = regressor_gpu.read_stage(si, 2).cpu().numpy()
        ref = g[n + '_slice']
        assert np.abs(got[0, :2] - ref).max() < 0.05 * np.abs(ref).max() + 1e-3


def test_regressor_f16_mode_against_reference_golden_outputs(regressor_gpu_f16, golden, dev):
    """precision 1 (fp16 storage, f16 MFMA) vs the reference module's fp32 outputs: 10x tighter than the bf16 mode
    -- 5e-4 on (cam, pose6d, shape), 0.25 % of the feature scale -- and a dtype check on the native input."""
    g = golden('regressor')
    x, _ = S.proxy_crops(int(g['crops_batch']), seed=int(g['crops_seed']))
    xt = torch.from_numpy(x).to(dev)
    cam, pose, shape, feats = regressor_gpu_f16(xt, return_feats=True)
    assert np.abs(feats.cpu().numpy() - g['feats']).max() < 2.5e-3 * np.abs(g['feats']).max()
    assert np.abs(cam.cpu().numpy() - g['cam']).max() < 5e-4
    assert np.abs(pose.cpu().numpy() - g['pose']).max() < 5e-4
    assert np.abs(shape.cpu().numpy() - g['shape']).max() < 5e-4
    native = regressor_gpu_f16.pack(xt)
    assert native.dtype == torch.float16
    c2, p2, s2 = regressor_gpu_f16(native)
    assert torch.equal(cam, c2) and torch.equal(pose, p2) and torch.equal(shape, s2)
    with pytest.raises(ValueError):
        regressor_gpu_f16(native.to(torch.bfloat16))


def test_regressor_native_input_batch_invariance_and_limits(regressor_gpu, dev):
    x, _ = S.proxy_crops(32, seed=8)
    xt = torch.from_numpy(x).to(dev)
    cam, pose, shape = regressor_gpu(xt)
    native = regressor_gpu.pack(xt)
    assert native.shape == (32, 262, 264, 18) and native.dtype == torch.bfloat16
    assert native[:, :3].abs().max().item() == 0 and native[:, :, -5:].abs().max().item() == 0 and native[:, -3:].abs().max().item() == 0 and native[:, :, :3].abs().max().item() == 0
    c2, p2, s2 = regressor_gpu(native)
    assert torch.equal(cam, c2) and torch.equal(pose, p2) and torch.equal(shape, s2)
    # a crop's result does not depend on its batch neighbours or batch size (BN is eval-mode, resnet is per-sample)
    c1, p1, s1 = regressor_gpu(xt[7:8])
    assert torch.allclose(c1[0], cam[7], atol=2e-5) and torch.allclose(p1[0], pose[7], atol=2e-5)   # tile variants differ with B: summation order only
    with pytest.raises(ValueError):
        regressor_gpu(torch.zeros(33, 18, 256, 256, device=dev))
    with pytest.raises(ValueError):
        regressor_gpu(torch.zeros(2, 17, 256, 256, device=dev))


def test_regressor_batch_72_equals_its_32_crop_blocks(regressor_sd_np, dev):
    """More crops per call than the bench uses: 72 = two full blocks of 32 players + a ragged one for the fused IEF kernel
    (one grid of 64 workgroups and one barrier counter per block), 288 layer1 tiles for 256 persistent workgroups, tile
    counts that are no multiple of the XCD grid.  Every crop's result must be the one it gets in a batch of 32: the kernels
    of both batch sizes add a pixel's products in the same order."""
    from soccerplayershapepose_amd.models.regressor import SingleInputRegressor
    B = 72
    big = SingleInputRegressor(18, 18, 3, mean_params=S.mean_params(), max_batch=B, precision='f16').load_state_dict(regressor_sd_np).to(dev)
    ref = SingleInputRegressor(18, 18, 3, mean_params=S.mean_params(), max_batch=32, precision='f16').load_state_dict(regressor_sd_np).to(dev)
    x, _ = S.proxy_crops(B, seed=21)
    xt = torch.from_numpy(x).to(dev)
    cam, pose, shape = big(xt)
    assert torch.isfinite(cam).all() and torch.isfinite(pose).all() and torch.isfinite(shape).all()
    for lo in (0, 32, 64):
        hi = min(lo + 32, B)
        c, p, s_ = ref(xt[lo:hi])
        # ragged tail (8 crops): the 32-crop regressor then runs its ragged-batch kernels, summation order differs
        tol = 0.0 if hi - lo == 32 else 2e-5
        assert (cam[lo:hi] - c).abs().max().item() <= tol, (lo, (cam[lo:hi] - c).abs().max().item())
        assert (pose[lo:hi] - p).abs().max().item() <= tol, (lo, (pose[lo:hi] - p).abs().max().item())
        assert (shape[lo:hi] - s_).abs().max().item() <= tol, (lo, (shape[lo:hi] - s_).abs().max().item())


# ---------------------------------------------------------------- end to end (a13)
@pytest.mark.parametrize('precision', ['bf16', 'f16'])
def test_end_to_end_against_oracle_pipeline(regressor_gpu, regressor_gpu_f16, smpl_gpu, oracle_smpl, regressor_sd_np, smpl_model_np, dev,
                                            precision):
    """crop -> regressor -> rot6d -> 50-iter pose+shape fit -> SMPL: vertices vs the CPU oracle run on the SAME
    regressor outputs within 1e-3 m (fit + SMPL kernels), and vs the ALL-fp32 oracle (the reference CPU path end to
    end): regressor -> SMPL within 2.5e-4 m with the fp16 backbone (precision 1) and 4e-3 m with bf16 (precision 0, the
    benchmark's configuration); after the fit the median player is within 1e-3 m (fp16), maxima bounded (see below)."""
    regressor_gpu = regressor_gpu_f16 if precision == 'f16' else regressor_gpu
    from soccerplayershapepose_amd.player_recon import reconstruct_players
    from soccerplayershapepose_amd.utils.rigid_transform_utils import rot6d_to_rotmat
    from oracle import regressor as OR, geometry as G, fit as OF, smpl as OS
    B = 5
    x, jin = S.proxy_crops(B, seed=12)
    xt = torch.from_numpy(x).to(dev)
    cam0, pose0, shape0 = regressor_gpu(xt)
    R0 = rot6d_to_rotmat(pose0).view(B, 24, 3, 3)
    # targets: the regressor-initialised model, perturbed
    R2, b2, _ = S.pose_params(B, seed=13, body_deg=8, global_deg=10)
    Rp = (R2.astype(np.float64) @ R0.cpu().numpy().astype(np.float64)).astype(np.float32)
    _, jp = OS.smpl_forward_np(smpl_model_np, Rp, shape0.cpu().numpy() + b2 / 3)
    tgt = S.fit_targets_from_joints(jp, cam0.cpu().numpy() * np.array([1.1, 1, 1], np.float32), seed=14)
    res = reconstruct_players(regressor_gpu, smpl_gpu, xt, tgt, fit='pose_shape', iters=50, return_vertices2d=True)
    assert res.vertices.shape == (B, 6890, 3) and res.vertices2d.shape == (B, 6890, 2)
    # (i) oracle fit + SMPL from the GPU regressor outputs
    v = _oracle_fit_vertices(oracle_smpl, R0.cpu(), shape0.cpu(), tgt, cam0.cpu())
    worst = (res.vertices.cpu() - v).abs().max().item()
    print(f'fit + SMPL kernels vs oracle fit + SMPL from identical regressor outputs ({precision}): {worst:.3e} m')
    assert worst < 1e-3, f'fit+SMPL kernels vs oracle from identical regressor outputs: {worst}'
    # (ii) everything fp32 on the CPU (the reference path end to end)
    cam_o, pose_o, shape_o = OR.regressor_forward(regressor_sd_np, x, S.mean_params())
    R_o = G.rot6d_to_rotmat(pose_o).view(B, 24, 3, 3)
    with torch.no_grad():
        pre_o = oracle_smpl(body_pose=R_o[:, 1:], global_orient=R_o[:, :1], betas=shape_o).vertices
    pre = smpl_gpu(full_pose=R0, betas=shape0).vertices.cpu()
    dev_pre = (pre - pre_o).abs().max().item()
    v = _oracle_fit_vertices(oracle_smpl, R_o, shape_o, tgt, cam_o)
    per_player = (res.vertices.cpu() - v).abs().amax(dim=(1, 2)).tolist()
    print(f'{precision} backbone vs all-fp32 oracle, max vertex deviation: regressor -> SMPL {dev_pre:.3e} m; '
          f'after the 50-iteration fit per player {[f"{d:.2e}" for d in per_player]} m')
    # Backbone -> SMPL (no optimiser in between): the north-star 1e-3 m holds with fp16 storage (measured 9e-5);
    # bf16 storage is the benchmark configuration, ~1.5e-3.
    assert dev_pre < (2.5e-4 if precision == 'f16' else 4e-3)
    # After the fit the figure also contains the reference algorithm's own sensitivity: Adam's update is
    # lr * m/(sqrt(v)+eps) ~ +-1e-3 per iteration however small the gradient, so for weakly observed joints a 1e-4
    # change of the regressor outputs can flip update signs and end 1e-3 apart (the all-fp32 oracle itself, fed its own
    # outputs + N(0, 1e-4): up to 1.1e-3).  Median within 1e-3 for fp16, bounded maxima for both.
    assert sorted(per_player)[len(per_player) // 2] < (1e-3 if precision == 'f16' else 5e-3)
    assert max(per_player) < (3e-3 if precision == 'f16' else 1e-2)


def test_north_star_all_32_players_of_the_bench_batch_within_1e_3_m_of_the_fp32_reference_path(regressor_gpu_f16, smpl_gpu, oracle_smpl,
                                                                                                regressor_sd_np, dev):
    """BASELINE north star, asserted: the benchmark's own batch (B = 32, fp16 backbone, bench.py's input batch 0) through
    regressor -> rot6d -> 50-iteration pose+shape fit -> SMPL, ALL 32 players against the all-fp32 CPU oracle run end to end
    on the same crops and targets: max |vertex difference| < 1e-3 m before AND after the fit (measured 9e-5 / 1.0-1.7e-4)."""
    from soccerplayershapepose_amd.player_recon import reconstruct_players
    from soccerplayershapepose_amd.utils.rigid_transform_utils import rot6d_to_rotmat
    from oracle import regressor as OR, geometry as G, fit as OF
    B = 32
    x, _ = S.proxy_crops(B, seed=1234)
    native = regressor_gpu_f16.pack(torch.from_numpy(x).to(dev))
    cam0, pose0, shape0 = regressor_gpu_f16(native)
    R0 = rot6d_to_rotmat(pose0).view(B, 24, 3, 3)
    pre = smpl_gpu(full_pose=R0, betas=shape0).vertices.cpu()
    R2, b2, _ = S.pose_params(B, seed=77, body_deg=8, global_deg=10)
    Rp = torch.matmul(torch.from_numpy(R2).to(dev), R0)
    jp = smpl_gpu(body_pose=Rp[:, 1:], global_orient=Rp[:, :1], betas=shape0 + torch.from_numpy(b2).to(dev) / 3, return_verts=False).joints
    tgt = S.fit_targets_from_joints(jp.cpu().numpy(), cam0.cpu().numpy() * np.array([1.1, 1, 1], np.float32), seed=4321)
    res = reconstruct_players(regressor_gpu_f16, smpl_gpu, native, tgt, fit='pose_shape', iters=50)
    cam_o, pose_o, shape_o = OR.regressor_forward(regressor_sd_np, x, S.mean_params())
    R_o = G.rot6d_to_rotmat(pose_o).view(B, 24, 3, 3)
    with torch.no_grad():
        pre_o = oracle_smpl(body_pose=R_o[:, 1:], global_orient=R_o[:, :1], betas=shape_o).vertices
    pre_dev = (pre - pre_o).abs().amax(dim=(1, 2))
    v = _oracle_fit_vertices(oracle_smpl, R_o, shape_o, tgt, cam_o)
    post_dev = (res.vertices.cpu() - v).abs().amax(dim=(1, 2)).tolist()
    print(f'north star, 32 players: pre-fit max {pre_dev.max().item():.3e} m, post-fit median {sorted(post_dev)[16]:.3e} max {max(post_dev):.3e} m')
    assert pre_dev.max().item() < 1e-3
    assert max(post_dev) < 1e-3, f'players beyond 1e-3 m after the fit: {[(i, d) for i, d in enumerate(post_dev) if d >= 1e-3]}'


def test_batch_256_every_block_of_32_is_bit_identical_and_a_sample_matches_the_oracle(regressor_sd_np, smpl_model_np, oracle_smpl, dev):
    """BASELINE configs[2]'s workload on one GPU: 256 crops in ONE call (max_batch = 256: 8 IEF blocks, 2 048 layer1 tiles on 256
    persistent workgroups, 256 fit workgroups, SMPL at B = 256).  (i) every block of 32 crops comes out bit for bit as a
    B = 32 call computes it -- regressor outputs and final vertices; (ii) backbone stages of a sample of crops against the
    rounding-emulating oracle; (iii) fit + SMPL of sample players against the oracle from identical regressor outputs."""
    from soccerplayershapepose_amd.models.regressor import SingleInputRegressor
    from soccerplayershapepose_amd.models.smpl_official import SMPL
    from soccerplayershapepose_amd.player_recon import reconstruct_players
    from soccerplayershapepose_amd.utils.rigid_transform_utils import rot6d_to_rotmat
    from oracle import regressor as OR, fit as OF
    B = 256
    big = SingleInputRegressor(18, 18, 3, mean_params=S.mean_params(), max_batch=B, precision='f16').load_state_dict(regressor_sd_np).to(dev)
    ref = SingleInputRegressor(18, 18, 3, mean_params=S.mean_params(), max_batch=32, precision='f16').load_state_dict(regressor_sd_np).to(dev)
    smpl_big, smpl_ref = SMPL(smpl_model_np, max_batch=B).to(dev), SMPL(smpl_model_np, max_batch=32).to(dev)
    with pytest.raises(ValueError):
        SingleInputRegressor(18, 18, 3, mean_params=S.mean_params(), max_batch=257).load_state_dict(regressor_sd_np).to(dev)
    natives = []
    xs = {}
    for blk in range(8):                        # 32 crops at a time on the host (1.2 GB of fp32 otherwise)
        x, _ = S.proxy_crops(32, seed=300 + blk)
        natives.append(big.pack(torch.from_numpy(x).to(dev)))
        xs[blk] = x
    native = torch.cat(natives)
    del natives
    cam, pose, shape = big(native)
    R0 = rot6d_to_rotmat(pose).view(B, 24, 3, 3)
    sample = [0, 37, 128, 255]
    stages = {si: big.read_stage(si, B)[sample].cpu() for si in (1, 2, 3, 4, 5)}
    tgt = (torch.rand(B, 17, 2, generator=torch.Generator().manual_seed(5)) * 200 + 150).to(dev)
    res = reconstruct_players(big, smpl_big, native, tgt, fit='pose_shape', iters=50)
    assert torch.isfinite(res.vertices).all()
    for blk in range(8):
        lo, hi = 32 * blk, 32 * blk + 32
        c, p, s_ = ref(native[lo:hi])
        assert torch.equal(cam[lo:hi], c) and torch.equal(pose[lo:hi], p) and torch.equal(shape[lo:hi], s_), f'block {blk}: regressor outputs'
        if blk in (0, 5, 7):
            r32 = reconstruct_players(ref, smpl_ref, native[lo:hi], tgt[lo:hi], fit='pose_shape', iters=50)
            assert torch.equal(res.vertices[lo:hi], r32.vertices) and torch.equal(res.cam[lo:hi], r32.cam), f'block {blk}: fit / SMPL'
    # (ii) stages of the sample crops against the fp16-emulating oracle
    xsamp = np.stack([xs[i // 32][i % 32] for i in sample])
    _, ost = OR.encoder_forward(regressor_sd_np, xsamp, emulate_bf16='f16', return_stages=True)
    for si, n in ((1, 'maxpool'), (2, 'layer1'), (3, 'layer2'), (4, 'layer3'), (5, 'layer4')):
        err, scale = (stages[si] - ost[n]).abs().max().item(), ost[n].abs().max().item()
        assert err <= 0.02 * 0.125 * scale, f'stage {n} at B = 256: max err {err} vs scale {scale}'
    # (iii) fit + SMPL kernels at B = 256 against the oracle, from the GPU's own regressor outputs
    idx = torch.tensor(list(sample))
    v = _oracle_fit_vertices(oracle_smpl, R0.cpu()[idx], shape.cpu()[idx], tgt.cpu().numpy()[idx.numpy()], cam.cpu()[idx])
    dev_s = (res.vertices.cpu()[idx] - v).abs().amax(dim=(1, 2))
    assert dev_s.max().item() < 1e-3, f'players of 256 beyond 1e-3 m: {[(int(i), float(d)) for i, d in zip(idx, dev_s) if d >= 1e-3]}'


def test_ief_hand_off_time_out_is_reported_as_an_error_not_as_silent_nans(regressor_sd_np, dev):
    """a starved grid-wide hand-off of the fused IEF used to leave NaNs and SPSP_OK.  The test hook launches
    the next forward's IEF one workgroup short (its hand-offs can never complete; short spin bound): the outputs are NaN,
    `check_status()` / the next forward raise, and the handle works again afterwards."""
    from soccerplayershapepose_amd.models.regressor import SingleInputRegressor
    reg = SingleInputRegressor(18, 18, 3, mean_params=S.mean_params(), max_batch=8, precision='f16').load_state_dict(regressor_sd_np).to(dev)
    x, _ = S.proxy_crops(4, seed=2)
    native = reg.pack(torch.from_numpy(x).to(dev))
    good = reg(native)
    torch.cuda.synchronize()
    reg.check_status()
    r = reg._rt
    r.check(r.lib.spsp_regressor_debug(reg.handle, _abi.DEBUG_IEF_STALL, 1), 'spsp_regressor_debug')
    cam, pose, shape = reg(native)
    torch.cuda.synchronize()
    assert torch.isnan(cam).all() and torch.isnan(shape).all()
    with pytest.raises(RuntimeError, match='timed out'):
        reg(native)                                   # the failure of the earlier forward surfaces on the next call ...
    again = reg(native)                               # ... once; the handle itself is fine
    torch.cuda.synchronize()
    reg.check_status()
    assert all(torch.equal(a, b) for a, b in zip(good, again))



def test_ief_called_twice_on_the_same_features_gives_the_same_answer(regressor_sd_np, dev):
    """`spsp_regressor_ief` counts its grid-wide hand-offs on monotonic counters that a backbone call zeroes; run a
    second time without a backbone in between, every hand-off of the launch used to pass at once (stale hidden vectors, silently
    wrong cam / pose / betas, SPSP_OK).  The handle now tracks whether its counters are fresh and zeroes them itself when they are
    not: backbone -> ief -> ief -> ief gives three identical answers, equal to `spsp_regressor_forward`'s, at B = 5 and B = 40
    (two blocks of 32: two counter lines)."""
    from soccerplayershapepose_amd import runtime as rt
    from soccerplayershapepose_amd.models.regressor import SingleInputRegressor
    reg = SingleInputRegressor(18, 18, 3, mean_params=S.mean_params(), max_batch=40, precision='f16').load_state_dict(regressor_sd_np).to(dev)
    r = rt.get(dev)
    for B in (5, 40):
        x, _ = S.proxy_crops(B, seed=60 + B)
        native = reg.pack(torch.from_numpy(x).to(dev))
        want = [t.clone() for t in reg(native)]
        torch.cuda.synchronize()
        r.check(r.lib.spsp_regressor_backbone(r.ctx, reg.handle, rt.ptr(native), _abi.PROXY_NATIVE, B, 0, r.stream()), 'spsp_regressor_backbone')
        for _ in range(3):
            cam, pose, betas = (torch.full((B, n), float('nan'), device=dev) for n in (3, 144, 10))
            r.check(r.lib.spsp_regressor_ief(r.ctx, reg.handle, B, rt.ptr(cam), rt.ptr(pose), rt.ptr(betas), None, r.stream()), 'spsp_regressor_ief')
            torch.cuda.synchronize()
            reg.check_status()
            assert torch.equal(cam, want[0]) and torch.equal(pose, want[1]) and torch.equal(betas, want[2])


def test_predict_3d_mirror_against_the_oracle_path(regressor_gpu_f16, smpl_gpu, oracle_smpl, regressor_sd_np, dev):
    """`predict_3D` (`PR/predict/predict_3D.py:128-149`: regressor -> rot6d -> SMPL mesh -> vertices projected to the 512-px
    image -> the reposed T-pose mesh `smpl(betas=pred_shape)`) by VALUE against the all-fp32 oracle on the same proxies: the
    hand-made silhouette + joints through the reference's own proxy builder, and three synthetic crops.  Tolerances: the fp16
    backbone's pre-fit bounds -- cam / pose6d / betas 5e-4, vertices and reposed vertices 2.5e-4 m, projected vertices
    0.25 px of 512."""
    from soccerplayershapepose_amd.predict.predict_3D import predict_3D, create_proxy_representation
    from oracle import regressor as OR, geometry as G
    sil = np.zeros((512, 512)); sil[100:420, 180:330] = 1
    j = np.stack([np.linspace(190, 320, 17), np.linspace(110, 410, 17), np.ones(17)], axis=1)
    proxy = create_proxy_representation(sil, j, in_wh=512, out_wh=256)
    assert proxy.shape == (18, 256, 256)
    x = np.concatenate([proxy[None].astype(np.float32), S.proxy_crops(3, seed=83)[0]])
    B = x.shape[0]
    out = predict_3D(torch.from_numpy(x).float().to(dev), regressor_gpu_f16, smpl_gpu)
    assert out.pred_vertices.shape == (B, 6890, 3) and out.pred_reposed_vertices.shape == (B, 6890, 3)
    assert out.pred_vertices2d.shape == (B, 6890, 2) and torch.isfinite(out.pred_vertices).all()
    with torch.no_grad():
        cam, pose, shape = OR.regressor_forward(regressor_sd_np, x, S.mean_params())
        R = G.rot6d_to_rotmat(pose).view(B, 24, 3, 3)
        o = oracle_smpl(body_pose=R[:, 1:], global_orient=R[:, :1], betas=shape)
        v2d = G.undo_keypoint_normalisation(G.orthographic_project(o.vertices, cam), 512)
        eye = torch.eye(3).expand(B, 24, 3, 3)
        rep = oracle_smpl(body_pose=eye[:, 1:], global_orient=eye[:, :1], betas=shape).vertices
    dev_of = lambda a, b: (a.cpu() - b).abs().max().item()
    assert dev_of(out.pred_cam_wp, cam) < 5e-4 and dev_of(out.pred_shape, shape) < 5e-4
    assert dev_of(out.pred_pose_rotmats, R) < 1e-3
    assert dev_of(out.pred_vertices, o.vertices) < 2.5e-4, dev_of(out.pred_vertices, o.vertices)
    assert dev_of(out.pred_joints[:, :24], o.joints[:, :24]) < 2.5e-4
    assert dev_of(out.pred_vertices2d, v2d) < 0.25, dev_of(out.pred_vertices2d, v2d)
    assert dev_of(out.pred_reposed_vertices, rep) < 2.5e-4, dev_of(out.pred_reposed_vertices, rep)


# ---------------------------------------------------------------- throughput mode (multi-stream slots, HIP graphs)
def test_player_pipeline_streams_and_graph_replay_are_bit_identical_to_the_serial_path(regressor_sd_np, smpl_model_np, dev):
    """PlayerPipeline runs the same kernels on per-slot streams (eager or replayed from a captured HIP graph):
    every batch must come out bit-identical to a plain `reconstruct_players` call, whichever slot served it, and a
    change of the input buffer's CONTENT must be picked up by a replay (the graph bakes pointers, not data)."""
    from soccerplayershapepose_amd.models.regressor import SingleInputRegressor
    from soccerplayershapepose_amd.models.smpl_official import SMPL
    from soccerplayershapepose_amd.player_recon import PlayerPipeline, reconstruct_players
    B = 8
    mk_r = lambda: SingleInputRegressor(18, 18, 3, mean_params=S.mean_params(), max_batch=B).load_state_dict(regressor_sd_np)
    mk_s = lambda: SMPL(smpl_model_np, max_batch=B)
    x, _ = S.proxy_crops(2 * B, seed=5)
    xa, xb = torch.from_numpy(x[:B]).to(dev), torch.from_numpy(x[B:]).to(dev)
    tgt = (torch.rand(B, 17, 2, generator=torch.Generator().manual_seed(3)) * 300 + 100).to(dev)
    ref_reg, ref_smpl = mk_r().to(dev).set_plan('throughput'), mk_s().to(dev)      # (the plan PlayerPipeline selects for its slots)
    native = ref_reg.pack(xa)
    ref_a = reconstruct_players(ref_reg, ref_smpl, native, tgt, fit='pose_shape', iters=10)
    va, ca = ref_a.vertices.clone(), ref_a.cam.clone()
    ref_b = reconstruct_players(ref_reg, ref_smpl, ref_reg.pack(xb), tgt, fit='pose_shape', iters=10)
    vb = ref_b.vertices.clone()
    assert (va - vb).abs().max().item() > 1e-4          # the two input sets really differ
    for use_graphs in (False, True):
        pipe = PlayerPipeline(mk_r, mk_s, dev, depth=3, use_graphs=use_graphs)
        buf = native.clone()
        outs = [pipe.submit(buf, tgt, fit='pose_shape', iters=10) for _ in range(3)]
        pipe.synchronize()
        da = [(o.vertices - va).abs().max().item() + (o.cam - ca).abs().max().item() for o in outs]
        assert max(da) == 0.0, f'use_graphs={use_graphs}: slots differ from the serial path on batch a: {da}'
        buf.copy_(ref_reg.pack(xb))                      # same buffer, new crops
        torch.cuda.synchronize()
        outs = [pipe.submit(buf, tgt, fit='pose_shape', iters=10) for _ in range(4)]
        pipe.synchronize()
        db = [(o.vertices - vb).abs().max().item() for o in outs]
        assert max(db) == 0.0, f'use_graphs={use_graphs}: slots differ from the serial path on batch b: {db}'
    with pytest.raises(ValueError):
        PlayerPipeline(mk_r, mk_s, dev, depth=1, use_graphs=True).submit(native, tgt.cpu().numpy())


def test_grouped_tail_is_bit_identical_to_the_per_batch_path(regressor_sd_np, smpl_model_np, dev):
    """`spsp_pipeline_group` / `reconstruct_players_group` / `PlayerPipeline.submit_group`: the backbone batch by batch into rows
    of the handle's feature buffer, then IEF -> fit (R built from pose6d inside the kernel) -> SMPL forward ONCE over the
    players of two or three consecutive batches.  Every player is independent in every one of those stages, so each batch of
    a group must equal a plain `reconstruct_players` call bit for bit -- eager, replayed from a HIP graph, for groups of 2
    and 3, at a batch that fills whole IEF blocks (32) and one that does not (5) -- and `spsp_regressor_backbone` +
    `spsp_regressor_ief` must equal `spsp_regressor_forward`."""
    from soccerplayershapepose_amd.models.regressor import SingleInputRegressor
    from soccerplayershapepose_amd.models.smpl_official import SMPL
    from soccerplayershapepose_amd.player_recon import PlayerPipeline, reconstruct_players, reconstruct_players_group
    fields = ('cam', 'pose6d', 'rotmats', 'betas', 'vertices', 'joints', 'translation', 'record')
    for B, G in ((32, 2), (5, 3)):
        mk_r = lambda: SingleInputRegressor(18, 18, 3, mean_params=S.mean_params(), max_batch=G * B, precision='f16').load_state_dict(regressor_sd_np)
        mk_s = lambda: SMPL(smpl_model_np, max_batch=G * B)
        reg, smpl = mk_r().to(dev).set_plan('throughput'), mk_s().to(dev)
        x, _ = S.proxy_crops(G * B, seed=11 + B)
        natives = [reg.pack(torch.from_numpy(x[g * B:(g + 1) * B]).to(dev)) for g in range(G)]
        tgts = [(torch.rand(B, 17, 2, generator=torch.Generator().manual_seed(30 + g)) * 300 + 100).to(dev) for g in range(G)]
        refs = []
        for g in range(G):
            o = reconstruct_players(reg, smpl, natives[g], tgts[g], fit='pose_shape', iters=12, return_record=True)
            torch.cuda.synchronize()
            refs.append({f: getattr(o, f).clone() for f in fields})
        assert (refs[0]['vertices'] - refs[1]['vertices']).abs().max().item() > 1e-4
        got = reconstruct_players_group(reg, smpl, natives, tgts, fit='pose_shape', iters=12, return_record=True)
        torch.cuda.synchronize()
        for g in range(G):
            for f in fields:
                assert torch.equal(getattr(got[g], f), refs[g][f]), f'B={B} group of {G}: batch {g} differs in {f}'
        for use_graphs in (False, True):
            pipe = PlayerPipeline(mk_r, mk_s, dev, depth=2, use_graphs=use_graphs)
            outs = [pipe.submit_group(natives, tgts, fit='pose_shape', iters=12, return_record=True) for _ in range(5)]
            pipe.synchronize()
            for grp in outs:
                for g in range(G):
                    for f in fields:
                        assert torch.equal(getattr(grp[g], f), refs[g][f]), f'use_graphs={use_graphs} B={B}: batch {g} differs in {f}'
        with pytest.raises(ValueError):
            reconstruct_players_group(reg, smpl, natives + natives, tgts + tgts, fit='pose_shape', iters=2)      # more players than max_batch


def test_player_pipeline_is_correct_next_to_kernels_on_the_callers_stream(dev, monkeypatch):
    """The hazard of round 3: with ROCm 7.2 a HIP-graph replay issued behind `hipStreamWaitEvent` on a REAL
    event of another stream -- one recorded behind a kernel, i.e. as soon as the application runs anything of its own on the
    stream it submits from -- ran 13 % faster and returned vertices centimetres off.  `PlayerPipeline.submit` therefore waits
    for a busy caller stream (and for registered readers) on the HOST before a replay.  Here: four slots, graphs, a 64-float
    fill on the caller's stream in the hazardous spot (`SPSP_PIPE_TEST_NUDGE`), 200 batches over four rotating inputs; the last
    result of every slot must equal the serial eager path bit for bit (with `SPSP_PIPE_EVENT_WAIT=1` -- the device-side waits of
    rounds 1-2 -- this test fails: vertices 0.07-0.10 m off, run g3bp)."""
    from soccerplayershapepose_amd.models.regressor import SingleInputRegressor
    from soccerplayershapepose_amd.models.smpl_official import SMPL
    from soccerplayershapepose_amd.player_recon import PlayerPipeline, reconstruct_players
    monkeypatch.setenv('SPSP_PIPE_TEST_NUDGE', '1')
    monkeypatch.setenv('SPSP_PIPE_NUDGE_PERIOD', os.environ.get('TEST_NUDGE_PERIOD', '0'))
    B = 32
    sd, model = S.regressor_state_dict(seed=0), S.smpl_model(seed=0)
    mk_r = lambda: SingleInputRegressor(18, 18, 3, mean_params=S.mean_params(), max_batch=B, precision='f16').load_state_dict(sd).to(dev)
    mk_s = lambda: SMPL(model, max_batch=B)
    ref_reg, ref_smpl = mk_r().set_plan('throughput'), mk_s().to(dev)
    inputs, refs = [], []
    for k in range(4):
        x, _ = S.proxy_crops(B, seed=900 + k)
        nat = ref_reg.pack(torch.from_numpy(x).to(dev))
        tgt = (torch.rand(B, 17, 2, generator=torch.Generator().manual_seed(30 + k)) * 300 + 100).to(dev)
        ref = reconstruct_players(ref_reg, ref_smpl, nat, tgt, fit='pose_shape', iters=50)
        torch.cuda.synchronize()
        inputs.append((nat, tgt))
        refs.append((ref.vertices.clone(), ref.cam.clone(), ref.rotmats.clone()))
    pipe = PlayerPipeline(mk_r, mk_s, dev, depth=4, use_graphs=not os.environ.get('TEST_PIPE_EAGER'))
    # bench.py's sequence, the one that went wrong: four capturing submits, five warm-up submits, a synchronisation, then the
    # loop -- the hook's first fill lands on its third submit (2 x depth submits after the last capture), while every slot has
    # at most one replay outstanding; with SPSP_PIPE_EVENT_WAIT=1 (the event waits of rounds 1-2) this test fails on ROCm 7.2
    for i in range(9):
        pipe.submit(*inputs[i % 4], fit='pose_shape', iters=50)
        if i == 3:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    last = {}
    for i in range(9, 9 + 200):
        last[i % 4] = pipe.submit(*inputs[i % 4], fit='pose_shape', iters=50)
    torch.cuda.synchronize()
    for k, res in last.items():
        for got, want in zip((res.vertices, res.cam, res.rotmats), refs[k]):
            assert torch.equal(got, want), f'slot {k}: max deviation {(got - want).abs().max().item():.3e}'


def test_player_pipeline_keeps_freshly_allocated_inputs_alive_across_streams(regressor_sd_np, smpl_model_np, dev):
    """Every step hands the pipeline a FRESHLY allocated input with different content and drops it right after
    submit(); the caller's stream then allocates and scribbles over same-sized buffers at once.  Without
    record_stream() on the inputs the caching allocator gives the just-dropped block to that scribble while the slot's
    stream still reads it (conv1 of the previous batch): every result must equal the serial path bit for bit.  Results
    are consumed on the caller's stream through res.wait()."""
    from soccerplayershapepose_amd.models.regressor import SingleInputRegressor
    from soccerplayershapepose_amd.models.smpl_official import SMPL
    from soccerplayershapepose_amd.player_recon import PlayerPipeline, reconstruct_players
    B = 8
    mk_r = lambda: SingleInputRegressor(18, 18, 3, mean_params=S.mean_params(), max_batch=B).load_state_dict(regressor_sd_np)
    mk_s = lambda: SMPL(smpl_model_np, max_batch=B)
    x, _ = S.proxy_crops(3 * B, seed=21)
    ref_reg, ref_smpl = mk_r().to(dev).set_plan('throughput'), mk_s().to(dev)      # (the plan PlayerPipeline selects for its slots)
    natives = [ref_reg.pack(torch.from_numpy(x[i * B:(i + 1) * B]).to(dev)) for i in range(3)]
    tgts = [(torch.rand(B, 17, 2, generator=torch.Generator().manual_seed(30 + i)) * 300 + 100).to(dev) for i in range(3)]
    refs = [reconstruct_players(ref_reg, ref_smpl, natives[i], tgts[i], fit='pose_shape', iters=10).vertices.clone() for i in range(3)]
    torch.cuda.synchronize()
    pipe = PlayerPipeline(mk_r, mk_s, dev, depth=4)
    sums, worst = [], 0.0
    for step in range(12):
        i = step % 3
        inp, tg = natives[i].clone(), tgts[i].clone()            # fresh allocations on the caller's stream
        res = pipe.submit(inp, tg, fit='pose_shape', iters=10)
        del inp, tg
        junk = [torch.full((B, 262, 264, 18), 7.0, dtype=natives[0].dtype, device=dev) for _ in range(2)]   # would reuse the dropped blocks
        junk2 = torch.full((B, 17, 2), -1e6, device=dev)
        res.wait()
        sums.append((res.vertices - refs[i]).abs().max())         # consumed on the caller's stream
        del res, junk, junk2
    torch.cuda.synchronize()
    worst = max(float(v) for v in sums)
    assert worst == 0.0, f'pipelined results differ from the serial path by {worst} (input buffer reused while in flight?)'


def test_smpl_forward_is_bit_stable_next_to_convolutions_on_other_streams(regressor_sd_np, smpl_model_np, dev):
    """Regression test for a cross-stream corruption found with tools/attic/race_hunt2.py: with packed-fp32 VALU ops
    (`v_pk_fma_f32 ... op_sel`) in the skin kernel, ~35 % of SMPL forwards came out with wrong y-coordinates in
    lanes 48-63 of even players whenever MFMA / LDS-DMA convolution workgroups of another stream shared the CU.
    The library is built without packed-fp32 ops (csrc/Makefile); every forward must equal the serial result."""
    from soccerplayershapepose_amd.models.regressor import SingleInputRegressor
    from soccerplayershapepose_amd.models.smpl_official import SMPL
    B = 32
    regs = [SingleInputRegressor(18, 18, 3, mean_params=S.mean_params(), max_batch=B).load_state_dict(regressor_sd_np).to(dev)
            for _ in range(2)]
    smpl = SMPL(smpl_model_np, max_batch=B).to(dev)
    x, _ = S.proxy_crops(B, seed=5)
    native = regs[0].pack(torch.from_numpy(x).to(dev))
    R, betas, _ = S.pose_params(B, seed=3, body_deg=20, global_deg=30)
    R, betas = torch.from_numpy(R).to(dev).contiguous(), torch.from_numpy(betas).to(dev).contiguous()
    ref = smpl(full_pose=R, betas=betas)
    ref_v, ref_j = ref.vertices.clone(), ref.joints.clone()
    feats_ref = regs[0](native, return_feats=True)[3].clone()
    torch.cuda.synchronize()
    sa, others = torch.cuda.Stream(), [torch.cuda.Stream() for _ in regs]
    bad = total = 0
    for _ in range(6):
        outs, feats = [], []
        for _rep in range(3):
            for reg, st in zip(regs, others):
                with torch.cuda.stream(st):
                    feats.append(reg(native, return_feats=True)[3])
            with torch.cuda.stream(sa):
                outs += [smpl(full_pose=R, betas=betas) for _ in range(4)]
        torch.cuda.synchronize()
        total += len(outs)
        bad += sum(0 if (torch.equal(o.vertices, ref_v) and torch.equal(o.joints, ref_j)) else 1 for o in outs)
        assert all(torch.equal(f, feats_ref) for f in feats)
    assert bad == 0, f'{bad} of {total} SMPL forwards differ from the serial result while convolutions run on other streams'


# ---------------------------------------------------------------- box -> crop -> pad -> bilinear 512 on the GPU (8f-1, front half)
def test_gpu_player_crops_are_bit_identical_to_the_restated_crop_image_and_cv2_resize(dev):
    """spsp_crop_players vs the numpy restatement of `crop_image` (S/global_utils.py:257-290) + OpenCV's 8-bit
    INTER_LINEAR (oracle/geometry.py): every byte of every crop -- boxes inside the 1080p frame, boxes whose border
    leaves it on each side, odd and even padding on both axes, the 512 identity, the exact 2 x 2 shrink (INTER_AREA
    path), strong up-scaling of a tiny box, and a box wider than tall."""
    from soccerplayershapepose_amd.global_utils import crop_players, player_crop_border, player_crop_border_broad
    from oracle import geometry as G
    rng = np.random.default_rng(11)
    frame = rng.integers(0, 256, (1080, 1920, 3), dtype=np.uint8)
    boxes = np.array([[900, 400, 980, 640], [3, 5, 90, 300], [1850, 800, 1919, 1079], [700, 0, 761, 177], [100, 200, 612, 712],
                      [10, 20, 1034, 1044], [1000, 500, 1003, 503], [300, 900, 900, 1000], [640, 300, 741, 523], [5, 1000, 400, 1075]],
                     dtype=np.float32) + 0.6          # detector boxes are floats; astype(int32) truncates (player_detection.py:30)
    for border in (player_crop_border, player_crop_border_broad, 0):
        got = crop_players(torch.from_numpy(frame).to(dev), boxes, border=border)
        ref = G.crop_players(frame, boxes.astype(np.int32), border)
        assert got.shape == (len(boxes), 512, 512, 3) and got.dtype == torch.uint8
        diff = (got.cpu().numpy().astype(np.int16) - ref.astype(np.int16))
        assert np.abs(diff).max() == 0, f'border {border}: {int((diff != 0).sum())} bytes differ (max {np.abs(diff).max()})'
    # the 512 identity crop is the frame region itself; an empty box list is a no-op
    np.testing.assert_array_equal(crop_players(frame, boxes[4:5], border=0)[0].cpu().numpy(), frame[200:712, 100:612])
    assert crop_players(frame, np.zeros((0, 4), np.float32)).shape == (0, 512, 512, 3)
    with pytest.raises(ValueError):
        crop_players(frame.astype(np.float32), boxes)


# ---------------------------------------------------------------- proxy builder on the GPU (8f-1)
@pytest.mark.parametrize('precision', ['bf16', 'f16'])
def test_gpu_proxy_builder_is_bit_identical_to_the_reference_routine(regressor_gpu, regressor_gpu_f16, golden, dev, precision):
    """spsp_proxy_build vs pack(create_proxy_representation) (the numpy mirror, itself pinned to the reference's
    heat-map routine by tests/golden/geometry.npz): every element of the native tensor, border included -- random
    joints, sub-pixel joints (int16 truncation), and the reference's edge cases: joints at -8 / -7 (not drawn /
    one column), 0, 254 / 255 (last row / column never written), 262 / 263 (one column / not drawn), far outside."""
    from soccerplayershapepose_amd.predict.predict_3D import create_proxy_representation, create_proxy_representation_batch
    reg = regressor_gpu_f16 if precision == 'f16' else regressor_gpu
    rng = np.random.default_rng(5)
    B = 6
    sil = (rng.random((B, 512, 512)) < 0.4).astype(np.uint8)
    sil[0, ::2, ::2] = 1; sil[0, 1::2, :] = 0               # only the sampled pixels set / only the skipped ones set
    j = np.zeros((B, 17, 3), np.float32)
    j[:, :, :2] = rng.uniform(-40, 560, size=(B, 17, 2)).astype(np.float32)
    j[:, :, 2] = 1
    edge = np.array([-16, -15, -14, -1, 0, 1, 15, 16, 17, 508, 509, 510, 511, 524, 525, 526, 527], np.float32)
    j[1, :, 0] = edge; j[1, :, 1] = 200.7
    j[2, :, 1] = edge; j[2, :, 0] = 301.3
    j[3, :, 0] = edge; j[3, :, 1] = edge[::-1]
    j[4, :, :2] = np.floor(j[4, :, :2]) + 0.999                # truncation, not rounding
    ref = np.stack([create_proxy_representation(sil[b].astype(np.float64), j[b].astype(np.float64), 512, 256) for b in range(B)])
    want = reg.pack(torch.from_numpy(ref.astype(np.float32)).to(dev))
    got = create_proxy_representation_batch(torch.from_numpy(sil).to(dev), torch.from_numpy(j).to(dev), precision=precision)
    assert got.dtype == want.dtype and got.shape == want.shape
    assert torch.equal(got.view(torch.int16), want.view(torch.int16))
    got_f = create_proxy_representation_batch(sil.astype(np.float32), j[:, :, :2], precision=precision)     # host inputs, fp32 silhouette, [B,17,2]
    assert torch.equal(got_f.view(torch.int16), want.view(torch.int16))
    assert want[:, 3:259, 3:259, 1:].float().amax().item() > 0.9      # stamps really drawn
    # and it feeds the regressor directly
    a, b = reg(got[:2]), reg(torch.from_numpy(ref[:2].astype(np.float32)).to(dev))
    assert all(torch.equal(u, v) for u, v in zip(a, b))
    with pytest.raises(ValueError):
        create_proxy_representation_batch(sil[:, :256], j, precision=precision)


# ---------------------------------------------------------------- BASELINE configs 4 and 5 as parity cases
def test_ragged_match_frame_and_temporal_warm_start_clip(regressor_gpu, smpl_gpu, oracle_smpl, smpl_model_np, dev):
    """Config 4: one match frame = 22 players in one call (ragged w.r.t. every tile size).  Config 5: the 30-frame clip in
    which frame t's fit starts from frame t-1's result (`single_view_optimization` given the previous record as the warm
    start) while the targets move smoothly (2 deg / frame).  Two of the players are followed through the whole
    clip by the CPU oracle in two ways: TEACHER-FORCED, ALL 30 frames (the oracle fit of a frame starts from the state the GPU fit of
    that frame started from -- the regressor's outputs for frame 0, the GPU's previous result after that: 2e-4 on raw parameters, so
    per-frame errors cannot hide behind each other; the 60 (frame, player) fits are independent, so they run after the GPU clip as
    ONE `fit_pose_shape_batch` call -- seconds, where 30 separate oracle fits of 2.4 s of framework overhead each made round 4 thin
    this check to four frames) and FREE-RUNNING (the oracle chains its own results from the same regressor outputs: 2e-4 per frame
    over the first three frames, and the vertices after ten chained fits within the north-star 1e-3 m.  With SPSP_TEST_FULL=1 the
    oracle chains all 30 and the bound after the last is 1e-2 m: on this trajectory ONE fit (frame 20) parts the two chains by
    2.6 mm from starting states 2e-6 m apart -- the best-iterate rule choosing between two iterates of nearly equal joints2D_l2es,
    a discrete event of the algorithm that the ten-barrier kernel with IEEE division and square root shows as well (3.1 mm,
    tools/clip_chain.py, profiles/r05_clip_chain.txt; on the f16 regressor's trajectory both stay below 1.1e-4 m over 30 frames);
    what holds a frame to the reference is the teacher-forced check above)."""
    from soccerplayershapepose_amd.player_recon import reconstruct_players, single_view_optimization
    from soccerplayershapepose_amd.utils.rigid_transform_utils import rot6d_to_rotmat
    from oracle import fit as OF, smpl as OS
    N, T, follow = 22, 30, (0, 21)
    x, _ = S.proxy_crops(N, seed=31)
    xt = torch.from_numpy(x).to(dev)
    cam0, pose0, shape0 = regressor_gpu(xt)
    R0 = rot6d_to_rotmat(pose0).view(N, 24, 3, 3)
    # a smooth random walk of the "true" pose (2 deg / frame) drives the detected joints of every frame
    Rt_true = R0.cpu().numpy().astype(np.float64)
    targets = []
    for t in range(T):
        step, b2, _ = S.pose_params(N, seed=40 + t, body_deg=2, global_deg=2)
        Rt_true = step.astype(np.float64) @ Rt_true
        _, jp = OS.smpl_forward_np(smpl_model_np, Rt_true.astype(np.float32), shape0.cpu().numpy())
        targets.append(S.fit_targets_from_joints(jp, cam0.cpu().numpy() * np.array([1.05, 1, 1], np.float32), seed=50 + t))
    # frame 0: the whole path from the crops; frames 1..: warm start from the previous record
    res = reconstruct_players(regressor_gpu, smpl_gpu, xt, targets[0], fit='pose_shape', iters=50)
    assert res.vertices.shape == (N, 6890, 3) and torch.isfinite(res.vertices).all()
    body, orient, betas, cam = res.rotmats[:, 1:], res.rotmats[:, :1], res.betas, res.cam
    fl = list(follow)
    free = (R0[fl, 1:].cpu(), R0[fl, :1].cpu(), shape0[fl].cpu(), cam0[fl].cpu())      # the oracle's own chain (both followed players)
    started, ended, tgs = [free], [], []        # per frame: what the GPU fit started from / ended at (followed players), its targets
    worst_free = vdev = 0.0
    FREE = T if FULL_ORACLE else 10             # frames the oracle chains on its own (each a serial 2.4 s oracle fit)
    for t in range(T):
        if t > 0:
            rec = single_view_optimization(smpl_gpu, body, orient, betas, cam, targets[t], interation=50)
            body, orient, betas, cam = rec.body_pose, rec.global_orient, rec.betas, rec.cam_wp
        gpu_now = (body[fl].cpu(), orient[fl].cpu(), betas[fl].cpu(), cam[fl].cpu())
        tg = np.stack([targets[t][i] for i in fl])
        ended.append(gpu_now)
        tgs.append(tg)
        if t + 1 < T:
            started.append(gpu_now)
        if t < FREE:
            f = OF.fit_pose_shape_batch(oracle_smpl, free[0], free[1], free[2], tg, free[3], iters=50)['last']
            d = max((gpu_now[k] - f[k]).abs().max().item() for k in range(4))
            worst_free = max(worst_free, d)
            if t < 3:
                assert d < 2e-4, f'frame {t}: {d} from the free-running oracle chain'
            free = tuple(f[k].detach() for k in range(4))
        if t == FREE - 1:       # the vertices after FREE chained fits, GPU chain against oracle chain
            out = smpl_gpu(body_pose=body, global_orient=orient, betas=betas)
            with torch.no_grad():
                v = oracle_smpl(body_pose=free[0], global_orient=free[1], betas=free[2]).vertices
            vdev = (out.vertices[fl].cpu() - v).abs().max().item()
    out = smpl_gpu(body_pose=body, global_orient=orient, betas=betas)
    assert torch.isfinite(out.vertices).all()
    # teacher-forced, every frame: 2 T independent fits in one autograd graph
    o = OF.fit_pose_shape_batch(oracle_smpl, *(torch.cat([st[k] for st in started]) for k in range(3)), np.concatenate(tgs),
                                torch.cat([st[3] for st in started]), iters=50)['last']
    per_frame = torch.stack([(torch.cat([e[k] for e in ended]) - o[k]).abs().reshape(T, -1).amax(dim=1) for k in range(4)]).amax(dim=0).numpy()
    worst_forced = float(per_frame.max())
    print(f'30-frame clip: worst per-frame parameter deviation teacher-forced (all {T} frames) {worst_forced:.2e}, free-running ({FREE} frames) '
          f'{worst_free:.2e}; vertices after {FREE} chained fits vs the free-running oracle chain {vdev:.2e} m')
    assert worst_forced < 2e-4, f'frame {int(np.argmax(per_frame))}: {worst_forced} from the oracle fit started at the same state'
    assert vdev < (1e-2 if FULL_ORACLE else 1e-3)


# ---------------------------------------------------------------- PoseRelationModule (8f-2)
def test_pose_relation_module_against_reference_golden_and_oracle(regressor_gpu, smpl_gpu, golden, dev):
    """csrc/relate.hip vs the reference's PoseRelationModule outputs (tests/golden/relate.npz, 22-player frame and a
    2-player one) and vs the oracle on a fresh 37-player frame; tolerance 5e-6 on matrix entries (fp32; sin/cos/log/exp
    of different libms).  Then the operator in its place in the pipeline (between rot6d and the fit)."""
    from soccerplayershapepose_amd.models.regressor_relate import PoseRelationModule
    from soccerplayershapepose_amd.player_recon import reconstruct_players
    from oracle import relate as ORL
    g = golden('relate')
    sd = S.relation_state_dict(seed=int(g['weights_seed']))
    rel = PoseRelationModule(max_players=64).load_state_dict(sd).to(dev).eval()
    for tag in ('frame22', 'pair'):
        R, boxes = torch.from_numpy(g[f'{tag}_R']).to(dev), torch.from_numpy(g[f'{tag}_boxes']).to(dev)
        out = rel([R, boxes])
        assert out.shape == R.shape
        np.testing.assert_allclose(out.cpu().numpy(), g[f'{tag}_out'], atol=5e-6, rtol=0)
        assert torch.equal(R.cpu(), torch.from_numpy(g[f'{tag}_R']))        # input untouched
    R, _, _ = S.pose_params(37, seed=71)
    boxes = S.frame_boxes(37, seed=72)
    want = ORL.pose_relation_forward(sd, R, boxes).numpy()
    np.testing.assert_allclose(rel([R, boxes]).cpu().numpy(), want, atol=5e-6, rtol=0)       # host inputs are accepted
    with pytest.raises(ValueError):
        rel([R[:5], boxes[:4]])
    with pytest.raises(ValueError):
        PoseRelationModule(max_players=8).load_state_dict(sd).to(dev)([R, boxes])
    with pytest.raises(KeyError):
        PoseRelationModule().load_state_dict({k: v for k, v in sd.items() if 'relation.3.WQ' not in k})
    # in the pipeline: regressor -> rot6d -> relation -> (no fit) -> SMPL equals the stages run by hand
    from soccerplayershapepose_amd.utils.rigid_transform_utils import rot6d_to_rotmat
    x, _ = S.proxy_crops(6, seed=9)
    xt, bx = torch.from_numpy(x).to(dev), torch.from_numpy(S.frame_boxes(6, seed=3)).to(dev)
    res = reconstruct_players(regressor_gpu, smpl_gpu, xt, None, fit=None, pose_relation=rel, boxes=bx)
    cam, pose, shape = regressor_gpu(xt)
    R2 = rel([rot6d_to_rotmat(pose).view(6, 24, 3, 3), bx])
    assert torch.equal(res.rotmats, R2) and torch.equal(res.vertices, smpl_gpu(full_pose=R2, betas=shape).vertices)
    with pytest.raises(ValueError):
        reconstruct_players(regressor_gpu, smpl_gpu, xt, None, fit=None, pose_relation=rel)


# ---------------------------------------------------------------- exchange step through the C ABI (8e)
def test_c_abi_rccl_allgather_of_records_single_rank(dev):
    """spsp_comm_* / spsp_allgather_records: the torch-free form of the one collective on the path.  A one-GPU box
    can only form a 1-rank communicator; that still goes through ncclCommInitRank + ncclAllGather on the stream.
    Run in a child process so RCCL is initialised by this library alone (no torch.distributed in the process)."""
    import subprocess, sys, textwrap
    code = textwrap.dedent("""
        import ctypes as C, torch
        from soccerplayershapepose_amd import _abi, runtime as rt
        dev = torch.device('cuda', 0)
        r = rt.get(dev)
        uid = (C.c_char * 128)()
        r.check(r.lib.spsp_comm_unique_id(uid), 'uid')
        comm = C.c_void_p()
        r.check(r.lib.spsp_comm_init_rank(r.ctx, uid, 1, 0, C.byref(comm)), 'init')
        local = torch.arange(22 * 229, dtype=torch.float32, device=dev).reshape(22, 229) / 7
        out = torch.full((22, 229), -1.0, device=dev)
        st = torch.cuda.Stream()
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            r.check(r.lib.spsp_allgather_records(r.ctx, comm, rt.ptr(local), 22, rt.ptr(out), r.stream()), 'allgather')
        st.synchronize()
        assert torch.equal(out, local)
        assert r.lib.spsp_comm_init_rank(r.ctx, uid, 2, 5, C.byref(C.c_void_p())) == _abi.EINVAL
        r.lib.spsp_comm_destroy(comm)
        print('rccl-ok')
    """)
    p = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and 'rccl-ok' in p.stdout, p.stdout[-2000:] + p.stderr[-2000:]


def test_record_exchange_lanes_and_nonblocking_form_single_rank(dev):
    """`sharding.RecordExchange` as bench.py drives it (process group of one rank, in a child process): one communicator
    per stream (`lanes`), the blocking and the event-returning form, ragged counts, and the exchange-stream form."""
    import subprocess, sys, textwrap
    code = textwrap.dedent("""
        import os, torch, torch.distributed as dist
        os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29631', RANK='0', WORLD_SIZE='1')
        from soccerplayershapepose_amd.sharding import RecordExchange
        dev = torch.device('cuda', 0); torch.cuda.set_device(dev)
        dist.init_process_group('nccl', rank=0, world_size=1)
        ex = RecordExchange(dev, mode='abi', lanes=3)
        assert ex.mode == 'abi' and '3 communicator' in ex.describe(), ex.describe()
        streams = [torch.cuda.Stream() for _ in range(3)]
        recs = [torch.randn(22, 229, device=dev) + i for i in range(3)]
        torch.cuda.synchronize()
        outs = []
        for rep in range(4):
            for i, st in enumerate(streams):
                with torch.cuda.stream(st):
                    outs.append((i, ex.all_gather(recs[i], 22, lane=i)))
        g, ev = ex.all_gather(recs[1], 22, wait=False)          # exchange-stream form, event instead of a wait
        torch.cuda.current_stream().wait_event(ev)
        g2 = ex.all_gather(recs[2], 22)                            # ... and its blocking form
        torch.cuda.synchronize()
        assert all(torch.equal(o, recs[i]) for i, o in outs) and torch.equal(g, recs[1]) and torch.equal(g2, recs[2])
        ex.close()
        dist.destroy_process_group()
        print('exchange-ok')
    """)
    p = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=300, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert p.returncode == 0 and 'exchange-ok' in p.stdout, p.stdout[-2000:] + p.stderr[-2000:]


def test_handles_release_their_device_memory(regressor_sd_np, smpl_model_np, dev):
    """Loading and dropping handles repeatedly must give all device memory back (DevicePool frees on destroy)."""
    import gc
    from soccerplayershapepose_amd.models.regressor import SingleInputRegressor
    from soccerplayershapepose_amd.models.smpl_official import SMPL
    from soccerplayershapepose_amd.models.regressor_relate import PoseRelationModule

    def cycle():
        reg = SingleInputRegressor(18, 18, 3, mean_params=S.mean_params(), max_batch=16).load_state_dict(regressor_sd_np).to(dev)
        smpl = SMPL(smpl_model_np, max_batch=16).to(dev)
        rel = PoseRelationModule(max_players=16).load_state_dict(S.relation_state_dict(seed=0)).to(dev)
        x, _ = S.proxy_crops(2, seed=3)
        cam, pose, shape = reg(torch.from_numpy(x).to(dev))
        torch.cuda.synchronize()
        del reg, smpl, rel, cam, pose, shape
        gc.collect()

    cycle()
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info(dev)
    for _ in range(6):
        cycle()
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info(dev)
    assert free0 - free1 < 8 << 20, f'{(free0 - free1) / 2**20:.1f} MiB not returned after 6 load/free cycles'


def test_smpl_global_orientation_is_a_rigid_motion_about_the_root_joint(smpl_gpu, smpl_model_np, dev):
    """Oracle-independent property of linear-blend skinning (smplx lbs): the global orientation enters only
    the root transform, so replacing it by G R_0 moves every vertex and joint by the rigid map x -> G' (x - J_0) + J_0
    with G' = G R_0 R_0^-1 = G applied about the root joint J_0 (which itself stays put).  Checked at full batch size
    with random poses and shapes; tolerance 2e-6 m (fp32)."""
    B = 32
    R, betas, _ = S.pose_params(B, seed=91, body_deg=35, global_deg=40)
    G2, _, _ = S.pose_params(B, seed=92, body_deg=0, global_deg=180)
    Rt, bt = torch.from_numpy(R).to(dev), torch.from_numpy(betas).to(dev)
    Gt = torch.from_numpy(G2[:, 0]).to(dev)                                   # [B,3,3] extra rotations
    a = smpl_gpu(body_pose=Rt[:, 1:], global_orient=Rt[:, :1], betas=bt)
    b = smpl_gpu(body_pose=Rt[:, 1:], global_orient=torch.matmul(Gt, Rt[:, 0]).unsqueeze(1), betas=bt)
    J0 = a.joints[:, :1]                                                      # root joint = J_regressor . v_shaped, pose-independent
    assert (b.joints[:, 0] - a.joints[:, 0]).abs().max().item() < 2e-6
    want_v = torch.matmul(a.vertices - J0, Gt.transpose(1, 2)) + J0
    want_j = torch.matmul(a.joints - J0, Gt.transpose(1, 2)) + J0
    assert (b.vertices - want_v).abs().max().item() < 2e-6
    assert (b.joints - want_j).abs().max().item() < 2e-6                      # all 90: chain, picked vertices, regressed
