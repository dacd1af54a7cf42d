#!/bin/bash
# HBM-side traffic of EVERY kernel of a short bench run (10 DDIM steps): FETCH_SIZE and WRITE_SIZE in separate passes.
OUT="${1:-gpurun_out/pmc_traffic}"
R=$PWD
mkdir -p $R/$OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/$OUT/pass$i -o t -- python $R/bench.py --steps 1 --warmup 0 --ddim-steps 10 --no-cpu-baseline > $R/$OUT/pass$i.out 2>&1
done
cd $R
python tools/pmc_traffic_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
find $OUT -name "*.csv" -size +3000k -delete
