cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/loss_stats
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/loss_stats -- python3 $GRAFT_REPO_ROOT/tools/prof_loss.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(ls -t gpurun_out/loss_stats/*/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && python tools/kstats.py "$f" 30
