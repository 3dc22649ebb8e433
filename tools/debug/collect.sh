tag=${1:-r02_l}
bash tools/collect_profiles.sh $tag > gpurun_out/collect_$tag.log 2>&1; rc=$?
tail -n 5 gpurun_out/collect_$tag.log
[ $rc -eq 0 ] || exit $rc
bash tools/pmc_shade_issue.sh ${tag}_issue > gpurun_out/${tag}_issue.log 2>&1 || { tail -n 20 gpurun_out/${tag}_issue.log; exit 1; }
tail -n 30 gpurun_out/${tag}_issue.log
