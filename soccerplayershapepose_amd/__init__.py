"""MI355X-native hot path of SoccerPlayerShapePose (see README.md): Python mirror of the reference's call surface over
`csrc/libspsp.so`."""
import os as _os

# HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues (runtime default: 4).  The four slot streams of a PlayerPipeline plus the
# caller's own stream then share four queues and two slots serialise behind each other: 87 k instead of 129 k players/s
# (profiles/r04_hwq.txt).  Read when the HIP runtime starts: effective if this package is imported before the first GPU call, harmless
# otherwise (PlayerPipeline warns when it finds fewer queues than it has slots + 1).
_os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
