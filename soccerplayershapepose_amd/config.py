"""Constants of the hot path -- mirrors `PR/config.py` and the parts of `S/global_var.py` the path uses."""
# ------------------------ Paths (same relative layout as the reference, PR/config.py:3-8) --------
SMPL_MODEL_DIR = 'PlayerReconstruction/additional/smpl'
SMPL_MEAN_PARAMS_PATH = 'PlayerReconstruction/additional/neutral_smpl_mean_params_6dpose.npz'
J_REGRESSOR_EXTRA_PATH = 'PlayerReconstruction/additional/J_regressor_extra.npy'
COCOPLUS_REGRESSOR_PATH = 'PlayerReconstruction/additional/cocoplus_regressor.npy'
H36M_REGRESSOR_PATH = 'PlayerReconstruction/additional/J_regressor_h36m.npy'

# ------------------------ Constants ------------------------
FOCAL_LENGTH = 5000.            # PR/config.py:15
REGRESSOR_IMG_WH = 256          # PR/config.py:16
proxy_rep_input_wh = 512        # S/global_var.py:32

# ------------------------ Joint label conventions (PR/config.py:29-38) ------------------------
ALL_JOINTS_TO_COCO_MAP = [24, 26, 25, 28, 27, 16, 17, 18, 19, 20, 21, 1, 2, 4, 5, 7, 8]
ALL_JOINTS_TO_H36M_MAP = list(range(73, 90))
H36M_TO_J17 = [6, 5, 4, 1, 2, 3, 16, 15, 14, 11, 12, 13, 8, 10, 0, 7, 9]
H36M_TO_J14 = H36M_TO_J17[:14]
SMPL_TO_KPRCNN_MAP = [24, 26, 25, 28, 27, 16, 17, 18, 19, 20, 21, 1, 2, 4, 5, 7, 8]

# ------------------------ Fit defaults (S/global_var.py:71,75,110; PR/PyTorch3DTest.py:3077-3080) ----
player_recon_train_regressor_learning_rate = 1e-3
player_recon_single_view_iteration = 100
player_recon_multi_view_iteration = 50       # S/global_var.py:79
texture_single_opt_interation = 50
optimize_camera_learning_rate = 0.01
optimize_camera_iterations = 50
player_recon_broad_view_iteration = 100          # S/global_var.py:83
player_recon_broad_view_learning_rate = 1e-2     # S/global_var.py:84
