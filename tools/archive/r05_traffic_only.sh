cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r05p; mkdir -p $OUT; cd $R
bash tools/prof_traffic.sh r05_traffic_c3 --no-host-path > $OUT/traffic_c3.log 2>&1; cp gpurun_out/prof/r05_traffic_c3/traffic.json $OUT/r05_traffic_c3.json
bash tools/prof_traffic.sh r05_traffic_16384x50 --envs-per-gpu 16384 --no-host-path > $OUT/traffic_16384.log 2>&1; cp gpurun_out/prof/r05_traffic_16384x50/traffic.json $OUT/r05_traffic_16384x50.json
bash tools/prof_traffic.sh r05_traffic_c5 --config c5 --no-host-path > $OUT/traffic_c5.log 2>&1; cp gpurun_out/prof/r05_traffic_c5/traffic.json $OUT/r05_traffic_c5.json
rm -rf gpurun_out/prof
cp $OUT/r05_traffic_*.json profiles/
bash tools/r05_benchlines.sh
