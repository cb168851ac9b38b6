set -x
mkdir -p gpurun_out
: > gpurun_out/r03r_sweep_nt_gathers_C3.txt
for T in off 0 4 16 64 128; do
  if [ "$T" = off ]; then unset HB_NT_FROM; else export HB_NT_FROM=$T; fi
  echo "HB_NT_FROM=$T" >> gpurun_out/r03r_sweep_nt_gathers_C3.txt
  python tools/sweep.py C3 "0:0:" >> gpurun_out/r03r_sweep_nt_gathers_C3.txt 2>&1
done
cut -c1-175 gpurun_out/r03r_sweep_nt_gathers_C3.txt
