# Dev tool (GPU box): A/B of two builds of libdzo_hip.so on the isolated single-pass kernel
# (tools/bin/old/libdzo_hip.so = the previous build, selected through DZO_LIB_PATH).
set -e
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
for r in 1 2; do
  echo "old:"; DZO_LIB_PATH=$PWD/tools/bin/old/libdzo_hip.so AB_MASKS=0 AB_ROUNDS=6 python3 tools/sp_ablate.py 2>/dev/null | tail -1
  echo "new:"; AB_MASKS=0 AB_ROUNDS=6 python3 tools/sp_ablate.py 2>/dev/null | tail -1
done
echo "fp32 old:"; AB_DTYPE=float32 DZO_LIB_PATH=$PWD/tools/bin/old/libdzo_hip.so AB_MASKS=0 AB_ROUNDS=6 python3 tools/sp_ablate.py 2>/dev/null | tail -1
echo "fp32 new:"; AB_DTYPE=float32 AB_MASKS=0 AB_ROUNDS=6 python3 tools/sp_ablate.py 2>/dev/null | tail -1
