R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04i; mkdir -p $O; cd $R
PF_LIB_PATH=$R/build/variants/libpf_norepair.so timeout 900 python -m pytest tests -m gpu -q --tb=line -k "not isa_lint" > $O/norepair_gputests.txt 2>&1
grep -n "^FAILED\|passed\|failed" $O/norepair_gputests.txt | tail -40
