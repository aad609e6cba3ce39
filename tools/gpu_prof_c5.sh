ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out/prof_c5
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_c5 -o bench -- python3 $ROOT/bench.py --workload c5 --steps 3 --warmup 1 --no-cpu-baseline > $ROOT/gpurun_out/prof_c5/bench.json 2> $ROOT/gpurun_out/prof_c5/bench.err
cd $ROOT/gpurun_out/prof_c5
for f in $(find . -name '*_results.db'); do python3 $ROOT/tools/prof_stats.py $f | head -8; python3 $ROOT/tools/prof_steps.py $f; done
find . -name "*.db" -size +30M -delete
