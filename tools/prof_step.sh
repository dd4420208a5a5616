# usage: bash tools/prof_step.sh <tag> [bench args...]   (run on the GPU box through gpurun)
# Writes rocprofv3 kernel-trace stats and two PMC passes under gpurun_out/prof/<tag>/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; shift
OUT=$R/gpurun_out/prof/$TAG
mkdir -p $OUT
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 bench.py --steps 500 --warmup 50 --no-cpu-baseline --lean "$@" > $OUT/bench_kt.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc1 -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --lean "$@" > $OUT/bench_pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc2 -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --lean "$@" > $OUT/bench_pmc2.log 2>&1
python3 tools/prof_summary.py $OUT
