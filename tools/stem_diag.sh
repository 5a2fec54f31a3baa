#!/bin/bash
# diagnostic twin only: the stem's K loop with its LDS-read stream or its MFMA stream removed (results invalid, timing only)
for d in 0 1 2; do SPSP_STEM_DIAG=$d python tools/conv_stamps.py 32 2>&1 | grep "stem\[" | tail -1; done
