#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 passes over bench.py.
#   scripts/profile_gpu.sh stats   -> kernel trace + stats of the default-shaped bench command
#   scripts/profile_gpu.sh pmc     -> FETCH_SIZE and WRITE_SIZE passes (separate runs, no other tracing)
# Output under gpurun_out/; summaries worth judging are copied into profiles/ by hand.
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p "$R/gpurun_out"
cd /tmp && export TMPDIR=/tmp
mode=${1:-stats}
if [ "$mode" = stats ]; then
  rm -rf "$R/gpurun_out/prof_stats"
  timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_stats" -- \
    python3 "$R/bench.py" --steps 24 --warmup 2 --no-cpu-baseline > "$R/gpurun_out/prof_stats.json" 2> "$R/gpurun_out/prof_stats.err"
  echo "stats rc=$?"
elif [ "$mode" = pmc ]; then
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf "$R/gpurun_out/pmc_$c"
    timeout -k 10 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$R/gpurun_out/pmc_$c" -- \
      python3 "$R/bench.py" --steps 6 --warmup 1 --no-cpu-baseline --target-layers 8 > "$R/gpurun_out/pmc_$c.json" 2> "$R/gpurun_out/pmc_$c.err"
    rc=$?
    echo "pmc $c rc=$rc"
    [ $rc -ne 0 ] && { tail -5 "$R/gpurun_out/pmc_$c.err"; exit $rc; }
  done
fi
if [ "$mode" = pmcb ]; then   # the ragged batch (4 requests per GPU)
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf "$R/gpurun_out/pmc_${c}_b4"
    timeout -k 10 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$R/gpurun_out/pmc_${c}_b4" -- \
      python3 "$R/bench.py" --steps 6 --warmup 1 --no-cpu-baseline --target-layers 8 --requests-per-gpu 4 > "$R/gpurun_out/pmc_${c}_b4.json" 2> "$R/gpurun_out/pmc_${c}_b4.err"
    rc=$?
    echo "pmcb $c rc=$rc"
    [ $rc -ne 0 ] && { tail -5 "$R/gpurun_out/pmc_${c}_b4.err"; exit $rc; }
  done
fi
ls -R "$R/gpurun_out" | head -40
if [ "$mode" = mfma ]; then
  rm -rf "$R/gpurun_out/pmc_mfma"
  timeout -k 10 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$R/gpurun_out/pmc_mfma" -- \
    python3 "$R/bench.py" --steps 6 --warmup 1 --no-cpu-baseline --target-layers 8 > "$R/gpurun_out/pmc_mfma.json" 2> "$R/gpurun_out/pmc_mfma.err"
  echo "pmc mfma rc=$?"; tail -3 "$R/gpurun_out/pmc_mfma.err" | grep -v "^E2026\|^W2026"
fi
