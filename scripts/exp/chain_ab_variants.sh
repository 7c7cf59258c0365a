set -e
mkdir -p gpurun_out
for v in "" def4 def12 def15 ""; do
  if [ -z "$v" ]; then lib=whisprrec_amd/libwhisprrec_hip.so; else lib=build_ab/lib_$v.so; fi
  echo "== ${v:-base}"
  WR_LIB=$lib timeout -k 10 120 python scripts/exp/chain_ab.py 1000000 1000000 64 65536 32 2>/dev/null
done
