"""Deterministic synthetic assets for the hot path (numpy only, no torch, no oracle).

The reference ships neither the licensed SMPL assets (`PR/config.py:3-8`), nor the STRAPS
checkpoint (`PR/player_recon.py:139-140`), nor any data.  Everything the benchmark and the
parity tests run on is generated here from an integer hash so that the SAME bytes come out
in this container and on the GPU box (no dependence on libm or on a RNG implementation):

* `regressor_state_dict`   -- the 132-key state dict of `SingleInputRegressor(18, 18, 3)`
                              (`PR/models/regressor.py:13-41`, schema in SURVEY Appendix C)
* `mean_params`            -- stand-in for `neutral_smpl_mean_params_6dpose.npz`
                              (`PR/models/ief_module.py:33-46`)
* `smpl_model`             -- an SMPL-shaped body model (6890 verts, 24 joints, 207 pose
                              blend-shapes, 10 betas, 9+19+17 extra joint regressors)
* `proxy_crops`            -- [B,18,256,256] silhouette + 17 joint heat-maps built with the
                              reference heat-map routine (`PR/utils/label_conversions.py:58-87`)
* `fit_targets`            -- detected-2D-joint stand-ins for the reprojection fit
"""
import numpy as np

NUM_VERTS = 6890
NUM_JOINTS = 24
NUM_BETAS = 10
NUM_POSE_BASIS = 207
NUM_FACES = 13776

# SMPL kinematic tree (smplx / SMPL kintree_table) -- SURVEY Appendix A.
PARENTS = np.array([-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14,
                    16, 17, 18, 19, 20, 21], dtype=np.int32)

# smplx VertexJointSelector ids (vertex_ids['smplh']): 5 face, 6 feet, 10 finger tips.
VERTEX_JOINT_IDS = np.array([
    332, 6260, 2800, 4071, 583,                 # nose, reye, leye, rear, lear
    3216, 3226, 3387, 6617, 6624, 6787,         # LBigToe LSmallToe LHeel RBigToe RSmallToe RHeel
    2746, 2319, 2445, 2556, 2673,               # l thumb index middle ring pinky
    6191, 5782, 5905, 6016, 6133,               # r thumb index middle ring pinky
], dtype=np.int32)

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    """Vectorised splitmix64 finaliser on uint64 arrays (wrap-around arithmetic)."""
    with np.errstate(over='ignore'):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        z = z ^ (z >> np.uint64(31))
    return z


def _salt(key):
    """Stable 64-bit salt of a string (FNV-1a), independent of PYTHONHASHSEED."""
    h = 0xCBF29CE484222325
    for ch in key.encode('utf-8'):
        h = ((h ^ ch) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def hash_uniform(n, key, seed=0):
    """n float64 values in [0,1) (exact multiples of 2^-24), a pure function of (key, seed, index)."""
    base = np.uint64((_salt(key) ^ (seed * 0x9E3779B97F4A7C15)) & 0xFFFFFFFFFFFFFFFF)
    with np.errstate(over='ignore'):
        idx = (np.arange(n, dtype=np.uint64) * np.uint64(0xD1342543DE82EF95) + base) & _M64
    z = _splitmix64(idx)
    return (z >> np.uint64(40)).astype(np.float64) * (1.0 / 16777216.0)


def hash_normal(n, key, seed=0):
    """Approximately N(0,1): (sum of 4 uniforms - 2) * sqrt(3); exact arithmetic, no libm."""
    u = hash_uniform(4 * n, key, seed).reshape(4, n)
    return (u.sum(axis=0) - 2.0) * 1.7320508075688772


# ----------------------------------------------------------------------------------------
# Regressor weights
# ----------------------------------------------------------------------------------------
def regressor_layer_table():
    """[(name, cout, cin, k, stride)] for the 20 convs of resnet18(in_channels=18)
    (`PR/models/resnet.py:145-156,177-199`)."""
    t = [('image_encoder.conv1', 64, 18, 7, 2)]
    cin = 64
    for li, c in enumerate([64, 128, 256, 512], start=1):
        for b in range(2):
            s = 2 if (li > 1 and b == 0) else 1
            t.append((f'image_encoder.layer{li}.{b}.conv1', c, cin, 3, s))
            t.append((f'image_encoder.layer{li}.{b}.conv2', c, c, 3, 1))
            if li > 1 and b == 0:
                t.append((f'image_encoder.layer{li}.{b}.downsample.0', c, cin, 1, 2))
            cin = c
    return t


def regressor_state_dict(seed=0):
    """132-key float32 state dict (numpy) for SingleInputRegressor(18,18,3).

    Scales are chosen so activations stay O(1) through the 8 residual blocks and the three
    IEF iterations move the parameters by O(0.1) -- BN statistics are non-trivial so that the
    BN folding done at load time (`SURVEY §5 checkpoint row`) is actually exercised."""
    sd = {}

    def bn(prefix, c, gamma_scale):
        sd[prefix + '.weight'] = (gamma_scale * (0.75 + 0.5 * hash_uniform(c, prefix + '.w', seed))).astype(np.float32)
        sd[prefix + '.bias'] = (0.1 * (hash_uniform(c, prefix + '.b', seed) - 0.5)).astype(np.float32)
        sd[prefix + '.running_mean'] = (0.2 * (hash_uniform(c, prefix + '.m', seed) - 0.5)).astype(np.float32)
        sd[prefix + '.running_var'] = (0.75 + 0.5 * hash_uniform(c, prefix + '.v', seed)).astype(np.float32)
        sd[prefix + '.num_batches_tracked'] = np.array(0, dtype=np.int64)

    for name, cout, cin, k, _ in regressor_layer_table():
        fan_in = cin * k * k
        w = hash_normal(cout * fan_in, name + '.weight', seed) * np.sqrt(2.0 / fan_in)
        sd[name + '.weight'] = w.reshape(cout, cin, k, k).astype(np.float32)
        if name.endswith('downsample.0'):
            bn(name[:-1] + '1', cout, 0.7)
        elif name == 'image_encoder.conv1':
            bn('image_encoder.bn1', cout, 1.0)
        else:
            # conv1 -> bn1 (gamma 1), conv2 -> bn2 (gamma 0.5: damp the residual branch)
            bn(name.replace('conv', 'bn'), cout, 1.0 if name.endswith('conv1') else 0.5)

    def fc(name, nout, nin, wscale, bscale):
        w = hash_normal(nout * nin, name + '.weight', seed) * wscale / np.sqrt(nin)
        sd[name + '.weight'] = w.reshape(nout, nin).astype(np.float32)
        sd[name + '.bias'] = (bscale * (hash_uniform(nout, name + '.bias', seed) - 0.5)).astype(np.float32)

    fc('ief_module.fc1', 512, 669, 1.0, 0.1)
    # pooled features are O(0.2) and vary little between crops: give their 512 columns a larger
    # gain than the 157 parameter-feedback columns so the outputs really depend on the image
    sd['ief_module.fc1.weight'][:, :512] *= np.float32(4.0)
    fc('ief_module.fc2', 512, 512, 1.4, 0.1)
    fc('ief_module.fc3', 157, 512, 0.05, 0.02)
    # the same nn.Linear objects are registered twice (`PR/models/ief_module.py:16-28`)
    for i, n in ((0, 1), (2, 2), (4, 3)):
        sd[f'ief_module.ief_layers.{i}.weight'] = sd[f'ief_module.fc{n}.weight']
        sd[f'ief_module.ief_layers.{i}.bias'] = sd[f'ief_module.fc{n}.bias']
    return sd


def mean_params():
    """{'pose': [144], 'shape': [10]} -- 6-D identity per joint, zero shape."""
    return {'pose': np.tile(np.array([1, 0, 0, 1, 0, 0], np.float32), 24),
            'shape': np.zeros(10, np.float32)}


# ----------------------------------------------------------------------------------------
# SMPL-shaped body model
# ----------------------------------------------------------------------------------------
_REST_JOINTS = np.array([
    [0.00, -0.24, 0.03], [0.07, -0.33, 0.02], [-0.07, -0.33, 0.02], [0.00, -0.12, 0.00],
    [0.10, -0.71, 0.02], [-0.10, -0.71, 0.02], [0.00, 0.02, 0.01], [0.09, -1.11, -0.02],
    [-0.09, -1.11, -0.02], [0.00, 0.07, 0.03], [0.11, -1.17, 0.10], [-0.11, -1.17, 0.10],
    [0.00, 0.28, -0.01], [0.08, 0.19, 0.00], [-0.08, 0.19, 0.00], [0.00, 0.37, 0.04],
    [0.17, 0.22, -0.01], [-0.17, 0.22, -0.01], [0.43, 0.21, -0.03], [-0.43, 0.21, -0.03],
    [0.68, 0.22, -0.03], [-0.68, 0.22, -0.03], [0.77, 0.21, -0.03], [-0.77, 0.21, -0.03],
], dtype=np.float64)

_BONE_RADIUS = np.array([0.12, 0.07, 0.07, 0.12, 0.05, 0.05, 0.13, 0.04, 0.04, 0.13, 0.03, 0.03,
                         0.05, 0.06, 0.06, 0.09, 0.05, 0.05, 0.04, 0.04, 0.03, 0.03, 0.03, 0.03])


def smpl_model(seed=0):
    """Synthetic SMPL-shaped model with the array names/shapes smplx builds from SMPL_NEUTRAL.pkl.

    Returns a dict of numpy arrays:
      v_template [6890,3] f32, shapedirs [6890,3,10] f32, posedirs [207,20670] f32 (smplx layout:
      reshape(6890*3,207).T), J_regressor [24,6890] f32 (dense), lbs_weights [6890,24] f32,
      parents [24] i32, faces [13776,3] i32, J_regressor_extra [9,6890], J_regressor_cocoplus
      [19,6890], J_regressor_h36m [17,6890] (`PR/models/smpl_official.py:17-25`),
      vertex_joint_ids [21] i32."""
    V = NUM_VERTS
    # --- vertices on capsules: vertex v belongs to bone b(v) between joint b and its parent
    u = hash_uniform(V, 'smpl.bone', seed)
    seg_len = np.array([0.15 if p < 0 else np.linalg.norm(_REST_JOINTS[k] - _REST_JOINTS[p])
                        for k, p in enumerate(PARENTS)])
    cdf = np.cumsum(seg_len + 0.08) / np.sum(seg_len + 0.08)
    bone = np.minimum(np.searchsorted(cdf, u), NUM_JOINTS - 1)
    par = np.where(PARENTS[bone] < 0, bone, PARENTS[bone])
    t = hash_uniform(V, 'smpl.t', seed)[:, None]
    axis_pt = _REST_JOINTS[bone] * (0.15 + 0.85 * t) + _REST_JOINTS[par] * (0.85 - 0.85 * t)
    d = hash_normal(3 * V, 'smpl.dir', seed).reshape(V, 3)
    d /= np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-6)
    v_template = axis_pt + d * _BONE_RADIUS[bone][:, None]
    # pin the 21 selector vertices to plausible landmark positions
    head, lank, rank, lw, rw = (_REST_JOINTS[i] for i in (15, 7, 8, 22, 23))
    landmarks = np.array([
        head + [0.0, 0.03, 0.11], head + [-0.035, 0.06, 0.09], head + [0.035, 0.06, 0.09],
        head + [-0.075, 0.04, 0.01], head + [0.075, 0.04, 0.01],
        lank + [0.03, -0.06, 0.17], lank + [0.07, -0.06, 0.13], lank + [0.0, -0.06, -0.06],
        rank + [-0.03, -0.06, 0.17], rank + [-0.07, -0.06, 0.13], rank + [0.0, -0.06, -0.06],
        lw + [0.03, 0.0, 0.04], lw + [0.09, 0.01, 0.02], lw + [0.10, 0.0, 0.0], lw + [0.09, -0.01, -0.02], lw + [0.07, -0.02, -0.03],
        rw + [-0.03, 0.0, 0.04], rw + [-0.09, 0.01, 0.02], rw + [-0.10, 0.0, 0.0], rw + [-0.09, -0.01, -0.02], rw + [-0.07, -0.02, -0.03],
    ])
    v_template[VERTEX_JOINT_IDS] = landmarks

    shapedirs = 0.01 * hash_normal(V * 3 * NUM_BETAS, 'smpl.shapedirs', seed).reshape(V, 3, NUM_BETAS)
    posedirs = 0.002 * hash_normal(NUM_POSE_BASIS * V * 3, 'smpl.posedirs', seed).reshape(NUM_POSE_BASIS, V * 3)

    # --- distances vertex -> joint
    d2 = ((v_template[:, None, :] - _REST_JOINTS[None, :, :]) ** 2).sum(-1)      # [V,24]
    # LBS weights: soft-max over the 4 nearest joints, exactly 0 elsewhere, rows sum to 1
    order = np.argsort(d2, axis=1)[:, :4]
    near = np.take_along_axis(d2, order, axis=1)
    # exp() is libm; keep reproducibility by a rational kernel instead of a true soft-max
    wgt = 1.0 / (1.0 + 400.0 * (near - near[:, :1])) ** 2
    wgt /= wgt.sum(axis=1, keepdims=True)
    lbs_weights = np.zeros((V, NUM_JOINTS))
    np.put_along_axis(lbs_weights, order, wgt, axis=1)

    # J_regressor: normalised weights over the 64 nearest vertices of each joint
    J_regressor = np.zeros((NUM_JOINTS, V))
    nn_idx = np.argsort(d2, axis=0)[:64, :]                                       # [64,24]
    for k in range(NUM_JOINTS):
        w = 0.5 + hash_uniform(64, f'smpl.jreg{k}', seed)
        J_regressor[k, nn_idx[:, k]] = w / w.sum()

    def sparse_rows(nrows, key, nnz=32):
        R = np.zeros((nrows, V))
        for r in range(nrows):
            # anchor each regressed joint near a skeleton joint so it is a plausible body point
            anchor = int(hash_uniform(1, f'{key}.a{r}', seed)[0] * NUM_JOINTS)
            idx = nn_idx[:nnz, anchor]
            w = 0.25 + hash_uniform(nnz, f'{key}.w{r}', seed)
            R[r, idx] = w / w.sum()
        return R.astype(np.float32)

    # faces: small triangles between template neighbours (vertex i with its 1st+2nd and with its 2nd+3rd nearest
    # neighbour), so that the mesh rasterises like a surface: 2 x 6890 = 13 780 candidates, the first 13 776 kept
    from scipy.spatial import cKDTree
    _, nn = cKDTree(v_template).query(v_template, k=4)
    faces = np.concatenate([np.stack([nn[:, 0], nn[:, 1], nn[:, 2]], axis=1),
                            np.stack([nn[:, 0], nn[:, 2], nn[:, 3]], axis=1)])[:NUM_FACES].astype(np.int32)
    return {
        'v_template': v_template.astype(np.float32),
        'shapedirs': shapedirs.astype(np.float32),
        'posedirs': posedirs.astype(np.float32),
        'J_regressor': J_regressor.astype(np.float32),
        'lbs_weights': lbs_weights.astype(np.float32),
        'parents': PARENTS.copy(),
        'faces': faces,
        'J_regressor_extra': sparse_rows(9, 'smpl.extra'),
        'J_regressor_cocoplus': sparse_rows(19, 'smpl.cocoplus'),
        'J_regressor_h36m': sparse_rows(17, 'smpl.h36m'),
        'vertex_joint_ids': VERTEX_JOINT_IDS.copy(),
    }


# ----------------------------------------------------------------------------------------
# Inputs
# ----------------------------------------------------------------------------------------
def gaussian_heatmaps(joints2D, img_wh, std=4):
    """Restates `convert_2Djoints_to_gaussian_heatmaps` (`PR/utils/label_conversions.py:58-87`)
    including its quirks: 16-sample grid over [-8,8] (no exact 0), exclusive end index
    min(img_wh-1, c+size).  joints2D: integer [J,2] (x,y).  Returns [img_wh,img_wh,J] f32."""
    num_joints = joints2D.shape[0]
    size = 2 * std
    heatmaps = np.zeros((img_wh, img_wh, num_joints), dtype=np.float32)
    lin = np.linspace(-size, size, 2 * size)
    x, y = np.meshgrid(lin, lin)
    gaussian = np.exp(-((np.sqrt(x * x + y * y)) ** 2 / (2.0 * std ** 2)))
    for i in range(num_joints):
        cx, cy = int(joints2D[i, 0]), int(joints2D[i, 1])
        if min(cx, cy) > -size and max(cx, cy) < img_wh - 1 + size:
            sx, ex = max(0, cx - size), min(img_wh - 1, cx + size)
            sy, ey = max(0, cy - size), min(img_wh - 1, cy + size)
            gsx, gex = max(0, size - cx), min(2 * size, 2 * size - (size + cx - (img_wh - 1)))
            gsy, gey = max(0, size - cy), min(2 * size, 2 * size - (size + cy - (img_wh - 1)))
            heatmaps[sy:ey, sx:ex, i] = gaussian[gsy:gey, gsx:gex]
    return heatmaps


def relation_state_dict(seed=0):
    """State dict (numpy, 216 arrays) of `PoseRelationModule` (`PR/models/regressor_relate.py`): 27 relation units
    x {WG [1,8], WK/WQ/WV [8,216]} + biases.  Scales keep the geometry gate WG around 1 (so relu/log/clamp all see
    both branches over a frame) and the attention logits O(1)."""
    sd = {}
    for r in range(27):
        p = f'relation_module.relation.{r}'
        sd[p + '.WG.weight'] = (0.6 * hash_normal(8, p + '.WG.w', seed)).reshape(1, 8).astype(np.float32)
        sd[p + '.WG.bias'] = (0.3 + 0.4 * hash_normal(1, p + '.WG.b', seed)).astype(np.float32)
        for n, sc in (('WK', 1.0), ('WQ', 1.0), ('WV', 0.15)):
            sd[f'{p}.{n}.weight'] = (sc * hash_normal(8 * 216, f'{p}.{n}.w', seed) / np.sqrt(216.0)).reshape(8, 216).astype(np.float32)
            sd[f'{p}.{n}.bias'] = (0.05 * hash_normal(8, f'{p}.{n}.b', seed)).astype(np.float32)
    return sd


def frame_boxes(n, seed=0, frame_wh=(1920, 1080)):
    """n player boxes (x_min, y_min, x_max, y_max) in a 1080p frame; two players share a centre column and two
    have identical boxes, so the reference's clamp(|delta|, 1e-3) branch and log(1) = 0 are exercised."""
    u = hash_uniform(4 * n, 'boxes', seed).reshape(n, 4)
    w = 40 + 80 * u[:, 2]
    h = 90 + 160 * u[:, 3]
    cx = 100 + (frame_wh[0] - 200) * u[:, 0]
    cy = 150 + (frame_wh[1] - 300) * u[:, 1]
    if n >= 4:
        cx[1] = cx[0]
        cx[3], cy[3], w[3], h[3] = cx[2], cy[2], w[2], h[2]
    return np.stack([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], axis=1).astype(np.float32)


def proxy_crops(batch, seed=0, wh=256):
    """[B,18,wh,wh] float32 NCHW proxy representations: channel 0 a binary capsule silhouette,
    channels 1..17 joint heat-maps at integer locations inside the silhouette bbox.
    Also returns the joint locations in 2*wh pixel space, [B,17,3] float64 (x, y, conf)."""
    out = np.zeros((batch, 18, wh, wh), dtype=np.float32)
    joints_in = np.zeros((batch, 17, 3), dtype=np.float64)
    yy, xx = np.mgrid[0:wh, 0:wh]
    for b in range(batch):
        u = hash_uniform(8 + 34, f'crop{b}', seed)
        cx = wh / 2 + (u[0] - 0.5) * 32
        cy = wh / 2 + (u[1] - 0.5) * 32
        ax = wh * (0.08 + 0.20 * u[2])
        ay = wh * (0.22 + 0.22 * u[3])
        out[b, 0] = (((xx - cx) / ax) ** 2 + ((yy - cy) / ay) ** 2 <= 1.0).astype(np.float32)
        jx = np.floor(cx - ax + 2 * ax * u[8:25]).astype(np.int64)
        jy = np.floor(cy - ay + 2 * ay * u[25:42]).astype(np.int64)
        j = np.stack([jx, jy], axis=1)
        out[b, 1:] = np.transpose(gaussian_heatmaps(j, wh), (2, 0, 1))
        joints_in[b, :, :2] = j * 2.0
        joints_in[b, :, 2] = 1.0
    return out, joints_in


def random_rotmats(n, max_deg, key, seed=0):
    """n proper rotation matrices with angle <= max_deg about hash-random axes (float64).
    Rodrigues with sin/cos from numpy -- used only to build INPUTS, never expected outputs."""
    ax = hash_normal(3 * n, key + '.axis', seed).reshape(n, 3)
    ax /= np.maximum(np.linalg.norm(ax, axis=1, keepdims=True), 1e-9)
    ang = np.deg2rad(max_deg) * (2.0 * hash_uniform(n, key + '.ang', seed) - 1.0)
    K = np.zeros((n, 3, 3))
    K[:, 0, 1], K[:, 0, 2] = -ax[:, 2], ax[:, 1]
    K[:, 1, 0], K[:, 1, 2] = ax[:, 2], -ax[:, 0]
    K[:, 2, 0], K[:, 2, 1] = -ax[:, 1], ax[:, 0]
    s, c = np.sin(ang)[:, None, None], np.cos(ang)[:, None, None]
    return np.eye(3)[None] + s * K + (1 - c) * (K @ K)


def pose_params(batch, seed=0, body_deg=25.0, global_deg=20.0):
    """Seeded plausible (R[B,24,3,3], betas[B,10], cam[B,3]) float32 -- stand-ins for the
    regressor's output when the fit / SMPL kernels are tested on their own."""
    R = random_rotmats(batch * 24, body_deg, 'pose.R', seed).reshape(batch, 24, 3, 3)
    R[:, 0] = random_rotmats(batch, global_deg, 'pose.G', seed)
    betas = 1.5 * (hash_uniform(batch * 10, 'pose.betas', seed).reshape(batch, 10) - 0.5)
    u = hash_uniform(batch * 3, 'pose.cam', seed).reshape(batch, 3)
    cam = np.stack([0.8 + 0.3 * u[:, 0], 0.2 * (u[:, 1] - 0.5), 0.2 * (u[:, 2] - 0.5) + 0.2], axis=1)
    return R.astype(np.float32), betas.astype(np.float32), cam.astype(np.float32)


SMPL_TO_KPRCNN_MAP = [24, 26, 25, 28, 27, 16, 17, 18, 19, 20, 21, 1, 2, 4, 5, 7, 8]   # PR/config.py:38


def fit_targets_from_joints(joints90, cam, seed=0, noise_px=1.0, img_wh=512):
    """Detected-2D-joint stand-ins [B,17,3] float64 (x, y, conf=1) in `img_wh` pixel space:
    weak-perspective projection (`PR/utils/cam_utils.py:5-26`, `PR/utils/joints2d_utils.py:5-10`)
    of the 17 COCO joints of a (perturbed) model plus hash-normal pixel noise."""
    joints90 = np.asarray(joints90, dtype=np.float64)
    cam = np.asarray(cam, dtype=np.float64)
    B = joints90.shape[0]
    X = joints90[:, SMPL_TO_KPRCNN_MAP, :2]
    uv = cam[:, None, 0:1] * (X + cam[:, None, 1:3])
    px = (uv + 1.0) * (img_wh / 2.0)
    px = px + noise_px * hash_normal(B * 34, 'fit.noise', seed).reshape(B, 17, 2)
    return np.concatenate([px, np.ones((B, 17, 1))], axis=2)
