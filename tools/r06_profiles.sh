# usage (GPU box): bash tools/r06_profiles.sh  -- everything profiles/r06_* is made from, into gpurun_out/r06p/
#   * the JSON lines of the default command (c3), of the driver's short run (--steps 20 --warmup 5) and of the other shapes
#     (each line carries the closed-loop path -- HIP's launches -- and the other episode phase beside the headline)
#   * rocprofv3 --kernel-trace --stats of the DEFAULT bench command (c3; and once through the hipGraph replay) and of c2 / c4 / c5 / 16384x50
#   * HBM traffic (FETCH_SIZE / WRITE_SIZE in separate --pmc passes) for c3, 16384x50 and c5
#   * SQ counter passes (instructions, wave cycles, wait cycles) for c3 and 16384x50
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r06p
rm -rf $OUT; mkdir -p $OUT
cd $R
B="timeout 300 python3 bench.py"
$B > $OUT/r06_bench_default.json 2> $OUT/bench.err
$B --steps 20 --warmup 5 > $OUT/r06_bench_default_20steps.json 2>> $OUT/bench.err
for c in c2 c4 c5; do $B --config $c --no-cpu-baseline > $OUT/r06_bench_$c.json 2>> $OUT/bench.err; done
$B --envs-per-gpu 16384 --no-cpu-baseline --no-host-path > $OUT/r06_bench_16384x50.json 2>> $OUT/bench.err
# configs[3] as a 4-GPU node would run it: two of its 2048-env shards per GPU, one launch
$B --config c4 --envs-per-gpu 4096 --no-cpu-baseline --no-host-path > $OUT/r06_bench_c4_two_shards.json 2>> $OUT/bench.err
# the closed-loop path as the headline (HIP's launches), the staggered episode phase as the headline, one queue for the large batch
$B --launch graph --no-cpu-baseline --no-host-path > $OUT/r06_bench_default_graph.json 2>> $OUT/bench.err
$B --launch graph --steps 20 --warmup 5 --no-cpu-baseline --no-host-path > $OUT/r06_bench_default_20steps_graph.json 2>> $OUT/bench.err
$B --phase staggered --no-cpu-baseline --no-host-path > $OUT/r06_bench_default_staggered.json 2>> $OUT/bench.err
$B --launch direct1 --envs-per-gpu 16384 --no-cpu-baseline --no-host-path > $OUT/r06_bench_16384x50_onequeue.json 2>> $OUT/bench.err
kt() { timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_$1 -- python3 bench.py --no-cpu-baseline --no-host-path --lean "${@:2}" > $OUT/kt_$1.log 2>&1; cp $OUT/kt_$1/*/*kernel_stats.csv $OUT/r06_step_kernel_stats_$1.csv; rm -rf $OUT/kt_$1; }
kt c3
kt c3_graph --launch graph
kt c2 --config c2
kt c4 --config c4
kt c5 --config c5
kt 16384x50 --envs-per-gpu 16384
bash tools/prof_traffic.sh r06_traffic_c3 --no-host-path > $OUT/traffic_c3.log 2>&1; cp gpurun_out/prof/r06_traffic_c3/traffic.json $OUT/r06_traffic_c3.json
bash tools/prof_traffic.sh r06_traffic_16384x50 --envs-per-gpu 16384 --no-host-path > $OUT/traffic_16384.log 2>&1; cp gpurun_out/prof/r06_traffic_16384x50/traffic.json $OUT/r06_traffic_16384x50.json
bash tools/prof_traffic.sh r06_traffic_c5 --config c5 --no-host-path > $OUT/traffic_c5.log 2>&1; cp gpurun_out/prof/r06_traffic_c5/traffic.json $OUT/r06_traffic_c5.json
bash tools/prof_step.sh r06_sq_c3 --no-host-path > $OUT/r06_step_kernel_summary_c3.txt 2>&1
bash tools/prof_step.sh r06_sq_16384 --envs-per-gpu 16384 --no-host-path > $OUT/r06_step_kernel_summary_16384x50.txt 2>&1
rm -rf gpurun_out/prof
ls -la $OUT; head -c 600 $OUT/r06_bench_default.json; echo; head -c 400 $OUT/r06_bench_default_20steps.json
