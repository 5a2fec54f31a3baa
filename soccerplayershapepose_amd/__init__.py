"""MI355X-native hot path of SoccerPlayerShapePose (see README.md): Python mirror of the reference's call surface over
`csrc/libspsp.so`."""
import os as _os
import sys as _sys
import warnings as _warnings

# HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues (runtime default: 4).  The four slot streams of a PlayerPipeline plus the
# caller's own stream then share four queues and two slots serialise behind each other: 87 k instead of 129 k players/s
# (profiles/r04_hwq.txt).  The runtime reads the variable ONCE, when it starts: setting it here works only if this package is
# imported before the first GPU call of the process.  `hw_queues()` says what the runtime can have seen; importing the package into
# a process whose GPU runtime is already up, without the variable, is reported at once (and again by PlayerPipeline).


def _runtime_already_up():
    t = _sys.modules.get('torch')
    try:
        return bool(t is not None and t.cuda.is_initialized())
    except Exception:
        return False


def _parse_queues(v):
    try:
        return max(1, int(str(v).strip()))
    except (TypeError, ValueError):
        return None


_HIP_UP_AT_IMPORT = _runtime_already_up()
_QUEUES_AT_IMPORT = _os.environ.get('GPU_MAX_HW_QUEUES')          # what the runtime saw if it was already up
if not _HIP_UP_AT_IMPORT:
    _os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
elif _parse_queues(_QUEUES_AT_IMPORT) is None or _parse_queues(_QUEUES_AT_IMPORT) < 5:
    _warnings.warn('soccerplayershapepose_amd: the GPU runtime of this process was initialised before the package was imported, with '
                   f'GPU_MAX_HW_QUEUES={_QUEUES_AT_IMPORT or "unset (runtime default 4)"}: the slot streams of a PlayerPipeline will share '
                   'hardware queues and serialise (measured 87 k instead of 129 k players/s).  Export GPU_MAX_HW_QUEUES=8 before the '
                   'first GPU call, or import this package first.', RuntimeWarning, stacklevel=2)


def hw_queues():
    """Hardware queues the HIP runtime of this process maps its streams onto, as far as the environment can tell: the value of
    GPU_MAX_HW_QUEUES the runtime saw when it started (4 = its default when unset or unparsable)."""
    v = _QUEUES_AT_IMPORT if _HIP_UP_AT_IMPORT else _os.environ.get('GPU_MAX_HW_QUEUES')
    return _parse_queues(v) or 4
