# usage (GPU box): bash tools/r05_direct_prof.sh -- does rocprofv3 see the library's own queue?  kernel trace + HBM traffic of the direct run
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r05d; rm -rf $OUT; mkdir -p $OUT; cd $R
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 bench.py --launch direct --no-cpu-baseline --no-host-path > $OUT/kt.log 2>&1
tail -3 $OUT/kt.log | cut -c1-300
cp $OUT/kt/*/*kernel_stats.csv $OUT/direct_kernel_stats_c3.csv 2>/dev/null; head -5 $OUT/direct_kernel_stats_c3.csv | cut -c1-200
rm -rf $OUT/kt
timeout 900 bash tools/prof_traffic.sh direct_traffic_c3 --launch direct --no-host-path > $OUT/traffic_c3.log 2>&1; tail -5 $OUT/traffic_c3.log | cut -c1-300
cp gpurun_out/prof/direct_traffic_c3/traffic.json $OUT/direct_traffic_c3.json; cat $OUT/direct_traffic_c3.json | cut -c1-600
rm -rf gpurun_out/prof
