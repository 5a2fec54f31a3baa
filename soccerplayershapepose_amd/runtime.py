"""Per-device runtime: the loaded C-ABI library, its context handle and pointer/stream plumbing.

PyTorch-ROCm is used only as the owner of device memory and streams; every kernel is reached through
`_abi` (ctypes -> csrc/libspsp.so).  No op in this package has a CPU or eager-PyTorch fallback."""
import ctypes as C

import torch

from . import _abi

_runtimes = {}


class Runtime:
    def __init__(self, device_index):
        self.lib = _abi.load()
        if not torch.cuda.is_available():
            raise RuntimeError('soccerplayershapepose_amd needs a ROCm GPU (torch.cuda.is_available() is False); '
                               'there is no CPU fallback')
        self.device = torch.device('cuda', device_index)
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            torch.cuda.current_stream()          # make sure torch initialised the device first
            _abi.check(self.lib.spsp_init(device_index, C.byref(h)), 'spsp_init')
        self.ctx = h

    def stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def check(self, rc, what):
        _abi.check(rc, what)


def get(device=None):
    """Runtime of `device` (torch.device / int / None = current CUDA device)."""
    if device is None:
        idx = torch.cuda.current_device() if torch.cuda.is_available() else 0
    elif isinstance(device, int):
        idx = device
    else:
        device = torch.device(device)
        if device.type != 'cuda':
            raise RuntimeError(f'soccerplayershapepose_amd runs on ROCm GPUs only, got device {device}')
        idx = device.index if device.index is not None else torch.cuda.current_device()
    rt = _runtimes.get(idx)
    if rt is None:
        rt = _runtimes[idx] = Runtime(idx)
    return rt


def dev_f32(t, device=None, name='tensor'):
    """Contiguous fp32 tensor on the GPU (moves / casts host inputs; no copy when already right)."""
    if not isinstance(t, torch.Tensor):
        t = torch.as_tensor(t)
    if device is None:
        device = t.device if t.is_cuda else get().device
    return t.to(device=device, dtype=torch.float32).contiguous()


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
