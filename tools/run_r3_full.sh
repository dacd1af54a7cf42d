#!/bin/bash
# round 3: full GPU suite, smoke, kernel trace of the bench command, default bench line
TAG="${1:-r3x}"
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -6 > gpurun_out/${TAG}_pytest_gpu.txt; cat gpurun_out/${TAG}_pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
timeout 900 python bench.py --steps 4 --warmup 1 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; cat gpurun_out/${TAG}_bench.json; tail -3 gpurun_out/${TAG}_bench.err
