set -e
mkdir -p gpurun_out/ge2e
python tools/ge2e_train_time.py 5 > gpurun_out/ge2e/sweep.txt 2>&1
for c in 7 6 4; do
  SSV_NNB_FORCE="1:1536:880=2,$c;1:768:880=2,$c" timeout -k 10 120 python tools/ge2e_train_time.py 5 >> gpurun_out/ge2e/sweep.txt 2>&1
done
SSV_NNB_FORCE="1:1536:880=1,7;1:768:880=1,7" timeout -k 10 120 python tools/ge2e_train_time.py 5 >> gpurun_out/ge2e/sweep.txt 2>&1
SSV_NNB_FORCE="1:1536:880=1,4;1:768:880=1,4" timeout -k 10 120 python tools/ge2e_train_time.py 5 >> gpurun_out/ge2e/sweep.txt 2>&1
cat gpurun_out/ge2e/sweep.txt
cd /tmp && export TMPDIR=/tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pg -- python3 $GRAFT_REPO_ROOT/tools/ge2e_train_time.py 3 > /dev/null 2>&1
f=$(ls /tmp/pg/*/*kernel_stats.csv | head -1); cp $f $GRAFT_REPO_ROOT/gpurun_out/ge2e/train_kernel_stats.csv
