R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04b; mkdir -p $O; cd $R
if [ -f profiles/tools/r04/test_args.txt ]; then ARGS=$(cat profiles/tools/r04/test_args.txt); else ARGS=""; fi
eval "timeout 900 python -m pytest tests -m gpu -q --tb=short $ARGS" > $O/gputests.txt 2>&1
grep -n "^FAILED\|^ERROR\|passed\|failed" $O/gputests.txt | tail -30
