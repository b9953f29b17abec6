cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/r3_hier_prof
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3_hier_prof -- python3 -u bench.py --steps 5 --warmup 2 --no-cpu-baseline --arch hier --batch 16 > gpurun_out/r3_hier_prof.log 2>&1
ls gpurun_out/r3_hier_prof/*/
