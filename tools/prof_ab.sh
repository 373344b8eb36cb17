#!/bin/bash
# In-step A/B of two library builds inside ONE gpurun call (same box):  tools/prof_ab.sh TAG_A LIB_A TAG_B LIB_B [pattern]
# ("" for the shipped library); prints the rows matching `pattern` of both kernel tables and both median step spans.
a=$1; la=$2; b=$3; lb=$4; pat=${5:-ln_}
[ -n "$la" ] && la="SSV_HIP_LIB=$GRAFT_REPO_ROOT/$la"
[ -n "$lb" ] && lb="SSV_HIP_LIB=$GRAFT_REPO_ROOT/$lb"
tools/prof_env.sh $a $la && tools/prof_env.sh $b $lb || exit 1
for t in $a $b; do echo "== $t"; grep -E "$pat" gpurun_out/${SSV_PROF_DIR:-r5}/prof_$t.txt | grep -E "TB/s|TF/s" | cut -c1-150; done
