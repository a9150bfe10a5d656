cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for n in 1 8; do
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/trace_$n -- python3 tools/scratch/trace.py $n > gpurun_out/trace_$n.log 2>&1
grep "sites per call" gpurun_out/trace_$n.log
done
