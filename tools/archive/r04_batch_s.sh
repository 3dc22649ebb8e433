#!/bin/bash
# Round 4, batch s: fp32 instance of the prefilter on a 12-byte-per-texel padded chain (row pair = one 16-byte + one 8-byte load) against
# the 16-byte-per-texel chain read as four 12-byte loads (tools/ab/libpbr_base.so = the build before), interleaved on one box
for r in 1 2 3; do for t in base rgb12; do
  PBR_HIP_LIB=$PWD/tools/ab/libpbr_$t.so python3 tools/cfg3_ms.py $t || exit 1
done; done
python3 -m pytest tests -x -q -m gpu -k "prefilter" 2>&1 | tail -3
