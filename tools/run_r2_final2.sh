#!/bin/bash
# round 2, last evidence run on the final binary: GPU suite, smoke, kernel trace of the bench command, default bench line
TAG="${1:-r2zz}"
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -4 > gpurun_out/${TAG}_pytest.txt; cat gpurun_out/${TAG}_pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
bash tools/run_profile.sh ${TAG} 2>&1 | tail -30 | cut -c1-200
timeout 900 python bench.py --steps 4 --warmup 1 > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err; cat gpurun_out/${TAG}_bench_default.json
