# usage (GPU box): bash tools/r02_profiles.sh  -- everything profiles/r02_* is made from, into gpurun_out/r02/
#   * rocprofv3 --kernel-trace --stats of the DEFAULT bench command (c3) and of c5 / 16384x50
#   * HBM traffic (FETCH_SIZE / WRITE_SIZE in separate --pmc passes) for c3, 16384x50 and c5
#   * SQ counter passes (instructions, wave cycles, wait cycles) for c3 and 16384x50
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02
rm -rf $OUT; mkdir -p $OUT
cd $R
# the JSON line of the default command, un-profiled
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
for c in c2 c4 c5; do python3 bench.py --config $c --no-cpu-baseline > $OUT/bench_$c.json 2>> $OUT/bench_default.err; done
python3 bench.py --envs-per-gpu 16384 --no-cpu-baseline --no-host-path > $OUT/bench_16384x50.json 2>> $OUT/bench_default.err
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-path > $OUT/bench_default_20steps.json 2>> $OUT/bench_default.err
# kernel-trace stats of the same (default) command
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_c3 -- python3 bench.py --no-cpu-baseline --no-host-path > $OUT/kt_c3.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_16384 -- python3 bench.py --envs-per-gpu 16384 --no-cpu-baseline --no-host-path > $OUT/kt_16384.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_c5 -- python3 bench.py --config c5 --no-cpu-baseline --no-host-path > $OUT/kt_c5.log 2>&1
for d in kt_c3 kt_16384 kt_c5; do cp $OUT/$d/*/*kernel_stats.csv $OUT/${d}_kernel_stats.csv; done
# traffic
bash tools/prof_traffic.sh r02_traffic_c3 --no-host-path > $OUT/traffic_c3.log 2>&1; cp gpurun_out/prof/r02_traffic_c3/traffic.json $OUT/r02_traffic_c3.json
bash tools/prof_traffic.sh r02_traffic_16384x50 --envs-per-gpu 16384 --no-host-path > $OUT/traffic_16384.log 2>&1; cp gpurun_out/prof/r02_traffic_16384x50/traffic.json $OUT/r02_traffic_16384x50.json
bash tools/prof_traffic.sh r02_traffic_c5 --config c5 --no-host-path > $OUT/traffic_c5.log 2>&1; cp gpurun_out/prof/r02_traffic_c5/traffic.json $OUT/r02_traffic_c5.json
# SQ counters
bash tools/prof_step.sh r02_sq_c3 --no-host-path > $OUT/r02_step_kernel_summary_c3.txt 2>&1
bash tools/prof_step.sh r02_sq_16384 --envs-per-gpu 16384 --no-host-path > $OUT/r02_step_kernel_summary_16384x50.txt 2>&1
rm -rf $OUT/kt_c3 $OUT/kt_16384 $OUT/kt_c5
ls -la $OUT
