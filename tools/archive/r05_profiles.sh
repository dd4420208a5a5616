# usage (GPU box): bash tools/r05_profiles.sh  -- everything profiles/r05_* is made from, into gpurun_out/r05p/
#   * the JSON lines of the default command (c3), of the driver's short run (--steps 20 --warmup 5) and of the other shapes
#   * rocprofv3 --kernel-trace --stats of the DEFAULT bench command (c3; and once through the hipGraph replay) and of c2 / c4 / c5 / 16384x50
#   * HBM traffic (FETCH_SIZE / WRITE_SIZE in separate --pmc passes) for c3, 16384x50 and c5
#   * SQ counter passes (instructions, wave cycles, wait cycles) for c3 and 16384x50
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05p
rm -rf $OUT; mkdir -p $OUT
cd $R
python3 bench.py > $OUT/r05_bench_default.json 2> $OUT/bench.err
python3 bench.py --steps 20 --warmup 5 > $OUT/r05_bench_default_20steps.json 2>> $OUT/bench.err
for c in c2 c4 c5; do python3 bench.py --config $c --no-cpu-baseline > $OUT/r05_bench_$c.json 2>> $OUT/bench.err; done
python3 bench.py --envs-per-gpu 16384 --no-cpu-baseline --no-host-path > $OUT/r05_bench_16384x50.json 2>> $OUT/bench.err
# the same shapes through the replayed hipGraph (the launch path of rounds 2-4; --launch graph) beside the default (the library's own queue)
python3 bench.py --launch graph --no-cpu-baseline --no-host-path > $OUT/r05_bench_default_graph.json 2>> $OUT/bench.err
python3 bench.py --launch graph --steps 20 --warmup 5 --no-cpu-baseline --no-host-path > $OUT/r05_bench_default_20steps_graph.json 2>> $OUT/bench.err
python3 bench.py --launch graph --config c4 --no-cpu-baseline --no-host-path > $OUT/r05_bench_c4_graph.json 2>> $OUT/bench.err
python3 bench.py --launch graph --config c5 --no-cpu-baseline --no-host-path > $OUT/r05_bench_c5_graph.json 2>> $OUT/bench.err
python3 bench.py --launch graph --envs-per-gpu 16384 --no-cpu-baseline --no-host-path > $OUT/r05_bench_16384x50_graph.json 2>> $OUT/bench.err
python3 bench.py --launch direct1 --envs-per-gpu 16384 --no-cpu-baseline --no-host-path > $OUT/r05_bench_16384x50_onequeue.json 2>> $OUT/bench.err
kt() { rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_$1 -- python3 bench.py --no-cpu-baseline --no-host-path "${@:2}" > $OUT/kt_$1.log 2>&1; cp $OUT/kt_$1/*/*kernel_stats.csv $OUT/r05_step_kernel_stats_$1.csv; rm -rf $OUT/kt_$1; }
kt c3
kt c3_graph --launch graph
kt c2 --config c2
kt c4 --config c4
kt c5 --config c5
kt 16384x50 --envs-per-gpu 16384
bash tools/prof_traffic.sh r05_traffic_c3 --no-host-path > $OUT/traffic_c3.log 2>&1; cp gpurun_out/prof/r05_traffic_c3/traffic.json $OUT/r05_traffic_c3.json
bash tools/prof_traffic.sh r05_traffic_16384x50 --envs-per-gpu 16384 --no-host-path > $OUT/traffic_16384.log 2>&1; cp gpurun_out/prof/r05_traffic_16384x50/traffic.json $OUT/r05_traffic_16384x50.json
bash tools/prof_traffic.sh r05_traffic_c5 --config c5 --no-host-path > $OUT/traffic_c5.log 2>&1; cp gpurun_out/prof/r05_traffic_c5/traffic.json $OUT/r05_traffic_c5.json
bash tools/prof_step.sh r05_sq_c3 --no-host-path > $OUT/r05_step_kernel_summary_c3.txt 2>&1
bash tools/prof_step.sh r05_sq_16384 --envs-per-gpu 16384 --no-host-path > $OUT/r05_step_kernel_summary_16384x50.txt 2>&1
rm -rf gpurun_out/prof
ls -la $OUT; head -c 600 $OUT/r05_bench_default.json; echo; head -c 400 $OUT/r05_bench_default_20steps.json
# the large shapes once more with a 64-step action tape (round 4's setting: 210 MB / 420 MB of actions streaming through the Infinity Cache)
python3 bench.py --envs-per-gpu 16384 --tape-len 64 --no-cpu-baseline --no-host-path > $OUT/r05_bench_16384x50_tape64.json 2>> $OUT/bench.err
python3 bench.py --config c5 --tape-len 64 --no-cpu-baseline --no-host-path > $OUT/r05_bench_c5_tape64.json 2>> $OUT/bench.err
