#!/bin/bash
# ONE parameterised lease script for the GPU box (replaces the per-experiment tools/run_r2*.sh / run_r3*.sh of earlier rounds).
# Run from the repo root on the GPU box, e.g.   gpurun --timeout 900 -- 'bash tools/gpu.sh tests; bash tools/gpu.sh bench r4a'
#
#   tests  [pytest args]           GPU parity suite (tail of the output -> gpurun_out/<tag>_pytest.txt; TAG env, default "t")
#   smoke                          __graft_entry__.smoke()
#   bench  TAG [bench args]        full default bench line -> gpurun_out/TAG_bench.json
#   quick  TAG [bench args]        bench without the diagnostics legs (headline timing only), 3 steps
#   trace  TAG [bench args]        rocprofv3 --kernel-trace --stats of the quick bench -> gpurun_out/TAG_kernel_trace_summary.md
#   ab-env TAG VAR V1 V2 [args]    same-box A/B of one environment variable, interleaved twice (quick bench)
#   ab-lib TAG L1 L2 ...           same-box A/B of library builds imagdressing_amd/libimagdressing_hip_<L>.so ("cur" = shipped)
#   ab-flags TAG V1 V2 [args]      same-box A/B of two values of tuning knob 2 (bench.py --gemm-flags), interleaved twice
#   ab-table TAG T1 T2             same-box A/B of two tuning tables over the tuned workloads (512x512, 512x640, configs[2], configs[4]), interleaved twice
#   retune TAG gemm_tune-args...   re-time one class of layers of the five tuned workloads (tools/gemm_tune.py ARGS: e.g. --only-linear --cands 30,32
#                                  --min-k 64, or --only-conv3x3) and merge the winners (3 % hysteresis) into imagdressing_amd/gemm_tuning.json;
#                                  the table before / after and the log -> gpurun_out/TAG_*
#   pmc    OUT PATTERN cmd...      PMC counters of the kernels matching PATTERN in cmd (separate passes, --kernel-trace only)
#   prof   TAG cmd...              rocprofv3 --kernel-trace --stats of any command -> gpurun_out/TAG_kernel_trace_summary.md (HEAD lines printed, default 30)
#   run    TAG cmd...              any command, stdout+stderr -> gpurun_out/TAG.txt (tail printed)
R="$(cd "$(dirname "$0")/.." && pwd)"
cd "$R"; mkdir -p gpurun_out
QUICK="--no-secondary --no-geometry-secondary --no-parity --no-latency --no-flops --no-live-traffic --no-cpu-baseline --no-power --no-configs"
line() { python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); r=d.get('roofline') or {}; print('$1', d['ms_per_step'], d['value'], r.get('achieved'))"; }
task="$1"; shift
case "$task" in
  tests)
    (timeout 1500 python -m pytest tests -q -m gpu "$@" 2>&1 | tail -8) | tee gpurun_out/${TAG:-t}_pytest.txt ;;
  smoke)
    timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 ;;
  bench)
    tag="$1"; shift
    timeout 900 python bench.py "$@" > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; tail -c 6000 gpurun_out/${tag}_bench.json ;;
  quick)
    tag="$1"; shift
    timeout 400 python bench.py --steps 3 --warmup 1 $QUICK "$@" 2> gpurun_out/${tag}_quick.err | tee gpurun_out/${tag}_quick.json | line "$tag $*" ;;
  trace)
    tag="$1"; shift
    mkdir -p gpurun_out/$tag
    (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$tag -o trace -- python $R/bench.py --steps 2 --warmup 1 $QUICK "$@" \
        > $R/gpurun_out/${tag}_bench_under_rocprof.json 2> $R/gpurun_out/${tag}_rocprof.err)
    DB=$(find gpurun_out/$tag -name "*.db" | head -1)
    python tools/rocprof_summary.py $DB gpurun_out/${tag}_kernel_trace_summary.md
    head -34 gpurun_out/${tag}_kernel_trace_summary.md
    python tools/rocprof_sequence.py $DB "" "" 0 > gpurun_out/${tag}_forward_launches.txt 2>&1 || true
    find gpurun_out/$tag -name "*.db" -size +20000k -delete ;;
  ab-env)
    tag="$1"; var="$2"; v1="$3"; v2="$4"; shift 4
    for v in $v1 $v2 $v1 $v2; do
      env $var=$v timeout 400 python bench.py --steps 3 --warmup 1 $QUICK "$@" 2>/dev/null | line "$var=$v"
    done | tee gpurun_out/${tag}_ab.txt ;;
  ab-lib)
    tag="$1"; shift
    for rep in 1 2; do for lib in "$@"; do
      if [ $lib = cur ]; then unset IMD_LIB_PATH; else export IMD_LIB_PATH=$R/imagdressing_amd/libimagdressing_hip_$lib.so; fi
      timeout 400 python bench.py --steps 3 --warmup 1 $QUICK 2>/dev/null | line "lib=$lib"
    done; done | tee gpurun_out/${tag}_ab.txt ;;
  ab-flags)
    tag="$1"; v1="$2"; v2="$3"; shift 3
    for rep in 1 2; do for v in $v1 $v2; do
      timeout 400 python bench.py --steps 3 --warmup 1 $QUICK --gemm-flags $v "$@" 2>/dev/null | line "gemm_flags=$v"
    done; done | tee gpurun_out/${tag}_flags_ab.txt ;;
  ab-table)
    tag="$1"; A="$2"; B="$3"
    ms() { python -c "import sys,json; print(json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])['ms_per_step'])"; }
    for rep in 1 2; do for T in $A $B; do
      export IMD_GEMM_TUNING=$T
      a=$(timeout 300 python bench.py --steps 3 --warmup 1 $QUICK 2>/dev/null | ms)
      b=$(timeout 300 python bench.py --steps 3 --warmup 1 $QUICK --width 512 --height 640 2>/dev/null | ms)
      c=$(timeout 300 python tools/configs.py --config 3 --steps 10 2>/dev/null | ms)
      d=$(timeout 300 python tools/configs.py --config 5 --steps 10 2>/dev/null | ms)
      echo "table=$T  512x512_ms_per_bench_step=$a  512x640=$b  configs2_ms_per_ddim_step=$c  configs4_ms_per_ddim_step=$d"
    done; done | tee gpurun_out/${tag}_table_ab.txt ;;
  insitu)
    # in-situ A/B of tile configs: every further argument is one run's list of forced table entries, e.g. "2048,1280,2560,1,1,0=32:2 8192,640,1920,1,1,0=32:1"
    # (tools/insitu_conv.py --force: HIP events around each imd_conv_gemm launch inside the running sampling loop, per shape key)
    tag="$1"; shift; i=0
    { timeout 300 python tools/insitu_conv.py --top 400 2>/dev/null | sed "s/^/base /"
      for f in "$@"; do i=$((i+1)); timeout 300 python tools/insitu_conv.py --top 400 --force $f 2>/dev/null | sed "s/^/run$i /"; done
    } > gpurun_out/${tag}_insitu.txt
    grep -c . gpurun_out/${tag}_insitu.txt ;;
  retune)
    tag="$1"; shift
    T=imagdressing_amd/gemm_tuning.json
    cp $T gpurun_out/${tag}_table_before.json
    i=0
    for wl in "--config 1" "--config 1 --width 512 --height 640" "--config 1 --batch 1" "--config 3" "--config 5"; do
      i=$((i+1))
      timeout 600 python tools/gemm_tune.py "$@" $wl --out gpurun_out/gemm_tuning.json > gpurun_out/${tag}_retune_$i.log 2>&1 && cp gpurun_out/gemm_tuning.json $T
      echo "workload $i ($wl): $(grep -c changed gpurun_out/${tag}_retune_$i.log) changed"
    done
    python - "$tag" <<'PY'
import json, sys
a = json.load(open(f"gpurun_out/{sys.argv[1]}_table_before.json"))["shapes"]; b = json.load(open("imagdressing_amd/gemm_tuning.json"))["shapes"]
ch = {k: (a.get(k), b[k]) for k in b if a.get(k) != b[k]}
print(len(ch), "entries changed")
for k, (x, y) in sorted(ch.items()): print(k, x, "->", y)
PY
    cp $T gpurun_out/${tag}_table_after.json
    cat gpurun_out/${tag}_retune_*.log | grep '^{' > gpurun_out/${tag}_retune_log.jsonl ;;
  prof)
    tag="$1"; shift
    mkdir -p gpurun_out/$tag
    (cd /tmp && export TMPDIR=/tmp && cd $R && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$tag -o trace -- "$@" > $R/gpurun_out/${tag}_cmd.out 2> $R/gpurun_out/${tag}_rocprof.err)
    DB=$(find gpurun_out/$tag -name "*.db" | head -1)
    python tools/rocprof_summary.py $DB gpurun_out/${tag}_kernel_trace_summary.md
    head -${HEAD:-30} gpurun_out/${tag}_kernel_trace_summary.md
    find gpurun_out/$tag -name "*.db" -size +20000k -delete ;;
  pmc)
    OUT="$1"; PAT="$2"; shift 2
    mkdir -p $R/$OUT; i=0
    for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA" \
               "GRBM_GUI_ACTIVE GRBM_COUNT" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS" \
               "TCC_HIT_sum TCC_MISS_sum"; do
      i=$((i+1))
      (cd /tmp && export TMPDIR=/tmp && cd $R && timeout 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/$OUT/pass$i -o a -- "$@" > $R/$OUT/pass$i.out 2>&1)
    done
    python tools/pmc_summary.py $OUT "$PAT" > $OUT/summary.txt 2>&1; cat $OUT/summary.txt
    find $OUT -name "*.csv" -size +2000k -delete ;;
  run)
    tag="$1"; shift
    ("$@" 2>&1 | tail -60) | tee gpurun_out/${tag}.txt ;;
  *) echo "tools/gpu.sh: unknown task '$task'" >&2; exit 2 ;;
esac
