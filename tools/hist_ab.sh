cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_harness.py -m gpu -x -q -k "hist or make_input or xcd" 2>&1 | tail -1
for tp in 0 1; do for sh in "166667 50" "1000000 64"; do PCL_BIN_TWOPASS=$tp python tools/hist_stage_bench.py $sh 30 2>&1 | tail -1; done; done
for tp in 0; do for sh in "166667 50" "1000000 64"; do export PCL_BIN_TWOPASS=$tp; rocprofv3 --kernel-trace --output-format csv -d gpurun_out/hp_${tp}_${sh%% *} -- python3 tools/hist_stage_bench.py $sh 20 > /dev/null 2>&1; done; done
python3 tools/kstats.py gpurun_out/hp_0_166667 gpurun_out/hp_0_1000000 | grep -v "accum\|bbox\|fill\|score\|final\|splat\|rocprim\|codes"
