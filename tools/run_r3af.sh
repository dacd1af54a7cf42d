#!/bin/bash
# K-order rotation between the workgroups that share an operand tile (gemm_dma128): time + HBM-side fetch per variant
cd "$(dirname "$0")/.."; mkdir -p gpurun_out
R=$PWD; O=$R/gpurun_out
for lib in default rot1 rot2 rot4; do
  if [ $lib = default ]; then unset IMD_LIB_PATH; else export IMD_LIB_PATH=$R/imagdressing_amd/libimd_$lib.so; fi
  echo "== $lib"
  python tools/l3_conv_bench.py --cfgs 17 --splits 1 --shapes LIN:8192:640:2560,LIN:8192:5120:640,LIN:32768:1280:1280,LIN:2048:1280:5120 --iters 100 2>/dev/null | cut -c1-120
  (cd /tmp && export TMPDIR=/tmp && timeout 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_rot_$lib -o g -- python $R/tools/gemm_bench.py --cfgs 17 --iters 2 --filter "L1 lin 2560->640" > /dev/null 2>&1)
  python - <<P
import csv,glob
v=[float(r['Counter_Value']) for f in glob.glob('$O/pmc_rot_$lib/**/g_counter_collection.csv', recursive=True) for r in csv.DictReader(open(f)) if 'gemm_dma128' in r['Kernel_Name'] and r['Counter_Name']=='FETCH_SIZE']
print('FETCH_SIZE KiB (x2 for bytes)', sum(v)/max(1,len(v)), len(v))
P
done 2>&1 | tee $O/r3af_k_rotation.txt
rm -rf $O/pmc_rot_*
