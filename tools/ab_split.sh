# A/B of the split kernel's switches: FETCH_SIZE (256 planes) per setting
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r05
for cfg in "0 0" "1 0" "0 1" "1 1"; do
  set -- $cfg
  export SPC_SPLIT_MIRROR=$1 SPC_SPLIT_SYNC=$2
  echo "== mirror $1 sync $2"
  bash $R/tools/pmc_split_fetch.sh m$1s$2 2>&1 | tail -2
done
