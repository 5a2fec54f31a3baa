"""CPU ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT.

A PyTorch-CPU fp32 (plus a float64 numpy twin for SMPL) restatement of the reference hot path
`proxy crop -> ResNet-18+IEF -> rot6d -> 50-iter reprojection fit -> SMPL forward -> vertices`
(SURVEY.md section 8a).  Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` may import this package; the product (`soccerplayershapepose_amd`) never does and
fails loudly when its HIP library is missing.

Pinning status:

* regressor (a2-a4), rot6d (a5), weak-perspective projection / key-point un-normalisation /
  camera conversion (a7, a12), joints2D loss (a8), heat-map / proxy builder (a1) and the
  joints2D_l2es metric (a11) are PINNED: `oracle/gen_golden.py` imports the reference's own
  modules from /root/reference in the build container, runs them on hash-seeded inputs and
  commits inputs + reference outputs as `tests/golden/*.npz`; `tests/test_oracle_golden.py`
  checks this restatement against those vectors.
* SMPL forward (a6) lives in the third-party `smplx` package (un-pinned in
  `PR/requirements.txt:6`, not vendored, not installed, no network): PARITY UNPINNED.  It is
  restated from smplx's published `lbs` algorithm (`smplx/lbs.py`: blend_shapes,
  vertices2joints, batch_rigid_transform, lbs; `smplx/vertex_joint_selector.py`) and anchored
  on the reference's call sites `PR/models/smpl_official.py:27-41`.  A float64 numpy twin
  cross-checks the fp32 torch version.
* fit loops (a9, a10) cannot be imported (their files import cv2 / neural_renderer / detectron2
  and run file-system work at import): restated line-for-line from
  `PR/PyTorch3DTest.py:3074-3122` and `PR/player_recon.py:1175-1282` on top of real
  `torch.autograd` + `torch.optim.Adam` -- the same engine the reference uses.  They inherit
  the SMPL "unpinned" status.
"""
