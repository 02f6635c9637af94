# Dev tool: instruction mix of one kernel of the in-tree library (or DZO_LIB_PATH).  usage: tools/isa_count.sh <mangled-name-regex>
LIB=${DZO_LIB_PATH:-dzoptimization.jl_amd/libdzo_hip.so}
T=$(mktemp -d); cp $LIB $T/lib.so; (cd $T && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading lib.so >/dev/null 2>&1)
for F in $T/*gfx950*; do /opt/rocm/lib/llvm/bin/llvm-objdump -d $F; done | awk -v pat="$1" '
  /^[0-9a-f]+ <.*>:/ { on = ($0 ~ pat) }
  on && /^[ \t]+[a-z]/ { n++; op=$1; c[op]++ }
  END { printf "total %d\n", n; for (k in c) if (k ~ /s_waitcnt|ds_|global_|s_barrier|v_readlane|v_writelane|scratch_|buffer_/) printf "%s %d\n", k, c[k] }' | sort
rm -rf $T
