#!/bin/bash
# One parametrised runner for the GPU sessions of a round (replaces the per-session gpu_rNN_x.sh scripts of rounds
# 3-5, which are in the git history up to commit bc3c084).  Runs on the GPU box under gpurun:
#
#   gpurun --timeout 3000 -- 'bash scripts/gpu_session.sh TAG STEP [STEP ...]'
#
# writes everything under gpurun_out/TAG/ (merged back by gpurun) and prints the tail of every step.  A STEP is
# "kind:arguments" (quote it); environment assignments may lead the arguments (FOO=1 BAR=2 ...):
#   pytest:ARGS      python -m pytest ARGS -q -m gpu            -> pytest_N.log
#   bench:ARGS       python bench.py ARGS                       -> bench_N.json (+ .err)
#   py:SCRIPT ARGS   python SCRIPT ARGS                         -> <script>_N.jsonl (+ .err)
#   exe:BINARY ARGS  BINARY ARGS (a stand-alone micro-benchmark) -> <binary>_N.txt
#   smoke            python -c "import __graft_entry__ as g; g.smoke()" -> smoke_N.log
#   sh:COMMAND       bash -c COMMAND (profiles: scripts/profile_bench.sh, pmc_*.sh ...) -> sh_N.log
# STEP_TIMEOUT (seconds, default 1500) bounds every step.
TAG=$1; shift
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
N=0
for STEP in "$@"; do
  N=$((N + 1))
  KIND=${STEP%%:*}
  ARGS=""
  if [[ "$STEP" == *:* ]]; then ARGS=${STEP#*:}; fi
  ENVS=()
  read -r -a WORDS <<< "$ARGS"
  while [[ ${#WORDS[@]} -gt 0 && "${WORDS[0]}" =~ ^[A-Za-z_][A-Za-z0-9_]*= ]]; do ENVS+=("${WORDS[0]}"); WORDS=("${WORDS[@]:1}"); done
  T=${STEP_TIMEOUT:-1500}
  echo "== step $N: $STEP"
  case "$KIND" in
    pytest) env "${ENVS[@]}" timeout "$T" python -m pytest "${WORDS[@]}" -q -m gpu > "$OUT/pytest_$N.log" 2>&1; tail -4 "$OUT/pytest_$N.log" ;;
    bench)  env "${ENVS[@]}" timeout "$T" python bench.py "${WORDS[@]}" > "$OUT/bench_$N.json" 2> "$OUT/bench_$N.err"; cut -c1-600 "$OUT/bench_$N.json" ;;
    py)     B=$(basename "${WORDS[0]}" .py); env "${ENVS[@]}" timeout "$T" python "${WORDS[@]}" > "$OUT/${B}_$N.jsonl" 2> "$OUT/${B}_$N.err"; cut -c1-900 "$OUT/${B}_$N.jsonl" | tail -40 ;;
    exe)    B=$(basename "${WORDS[0]}"); env "${ENVS[@]}" timeout "$T" "${WORDS[@]}" > "$OUT/${B}_$N.txt" 2>&1; tail -40 "$OUT/${B}_$N.txt" ;;
    smoke)  timeout "$T" python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke_$N.log" 2>&1; tail -6 "$OUT/smoke_$N.log" ;;
    sh)     env "${ENVS[@]}" timeout "$T" bash -c "${WORDS[*]}" > "$OUT/sh_$N.log" 2>&1; tail -20 "$OUT/sh_$N.log" ;;
    *)      echo "unknown step kind: $KIND" ;;
  esac
done
