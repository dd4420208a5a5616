# usage: bash tools/ablate_mix.sh "<flags1>" ... : per-wave instruction counts of the step kernel for each ablation build
cd $GRAFT_REPO_ROOT
for f in "$@"; do
  echo "== flags: [$f]"; FLEET_EXTRA_HIPCC_FLAGS="$f" python3 -c "from fleetrl_amd import build; build.build(force=True)"
  bash tools/prof_mix.sh mixabl 2>&1 | grep -E "SQ_INSTS_VALU  |SQ_INSTS_SALU|BRANCH|FMA_F64|ADD_F64|MUL_F64"
done
FLEET_EXTRA_HIPCC_FLAGS="" python3 -c "from fleetrl_amd import build; build.build(force=True)"
