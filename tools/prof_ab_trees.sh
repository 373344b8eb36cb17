#!/bin/bash
# In-step A/B of two whole TREES inside ONE gpurun call (same box): the committed tree exported under abl/base (git archive HEAD + its library)
# against the working tree -- for changes that span the Python side and the library.   tools/prof_ab_trees.sh [pattern]
pat=${1:-gemm_}
R0=$GRAFT_REPO_ROOT
(cd $R0/abl/base && GRAFT_REPO_ROOT=$R0/abl/base SSV_PROF_DIR=ab tools/prof_env.sh base) || exit 1
mkdir -p $R0/gpurun_out/ab && cp $R0/abl/base/gpurun_out/ab/prof_base.* $R0/gpurun_out/ab/
(cd $R0 && SSV_PROF_DIR=ab tools/prof_env.sh new) || exit 1
for t in base new; do echo "== $t"; grep -E "$pat" $R0/gpurun_out/ab/prof_$t.txt | cut -c1-150; done
