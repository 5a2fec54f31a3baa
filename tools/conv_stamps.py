"""Where the conv kernels spend their cycles: runs ONE regressor forward at batch 32 through the diagnostic twin of the
library (csrc/libspsp_stamps.so, `make -C soccerplayershapepose_amd/csrc libspsp_stamps.so`: s_memtime stamps in the
kernels, -DSPSP_CONV_STAMPS); with SPSP_CONV_DEBUG set every stamped launch prints the cycle shares of one workgroup.
The stamped build's run time is not a benchmark figure -- read the SHARES."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ['SPSP_LIB_PATH'] = os.path.join(ROOT, 'soccerplayershapepose_amd', 'csrc', 'libspsp_stamps.so')
os.environ['SPSP_CONV_DEBUG'] = '1'

import torch  # noqa: E402

from soccerplayershapepose_amd import synthetic as S  # noqa: E402
from soccerplayershapepose_amd.models.regressor import SingleInputRegressor  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device('cuda', 0)
reg = SingleInputRegressor(18, 18, 3, mean_params=S.mean_params(), max_batch=B).load_state_dict(S.regressor_state_dict(seed=0)).to(dev)
x, _ = S.proxy_crops(B, seed=1)
native = reg.pack(torch.from_numpy(x).to(dev))
torch.cuda.synchronize()
for rep in range(2):
    print(f'--- forward {rep} (batch {B})', file=sys.stderr)
    reg(native)
    torch.cuda.synchronize()
