#!/bin/bash
# round 6, batch i: restored round-5 grid kernel (runtime rows) vs batched list staging: tests on the product lib, interleaved size tables, timelines
set -o pipefail
out=gpurun_out; mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -x -q -rs > $out/r06_i_pytest.log 2>&1; rc=$?
tail -n 5 $out/r06_i_pytest.log
[ $rc -eq 0 ] || exit $rc
for r in 1 2 3; do for v in base batch; do
  PBR_HIP_LIB=tools/ab/libpbr_$v.so timeout -k 10 300 python tools/shade_tile_ms.py ${v}_$r >> $out/r06_i_sizes.jsonl 2>> $out/r06_i_sizes.err || exit 1
done; done
cat $out/r06_i_sizes.jsonl
for v in timing timingb; do
PBR_HIP_LIB=tools/ab/libpbr_$v.so timeout -k 10 300 python tools/shade_timeline.py $v 1440x960 1920x1080 3840x2160 >> $out/r06_i_timeline.jsonl 2>> $out/r06_i_timeline.err || { tail -5 $out/r06_i_timeline.err; exit 1; }
done
cat $out/r06_i_timeline.jsonl
for rows in 3 4; do for v in base batch; do
  PBR_HIP_LIB=tools/ab/libpbr_$v.so PBR_SHADE_ROWS_BIG=$rows timeout -k 10 300 python tools/shade_tile_ms.py ${v}_rows$rows 1440x960 1920x1080 1928x2168@7680x4320 >> $out/r06_i_rows.jsonl 2>> $out/r06_i_rows.err || exit 1
done; done
grep -v fit_ $out/r06_i_rows.jsonl
