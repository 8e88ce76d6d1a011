# A/B on ONE box of the full-size inter-pass twiddle table of the coset LDE (ntt2.hip.h tw_full; VX_NTT2_NO_TW_TABLE=1 switches it off), alternating.
# (The round-4 experiment also had a prescale table, column folding and the table in plain transforms behind a bit mask, and a library built
# with -DVX_NTT2_NO_SHIFT64 for the old shift form: results in profiles/r04_ntt_experiment.md, code removed.)
export VX_JIT_CACHE_DIR=$PWD/.jit_cache
run() {
  python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-host-witness-leg --no-dag-leg --no-chip-leg --no-dag-stark-leg 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', round(d['ms_per_step'], 2), 'lde', d['stage_ms_per_step']['lde'], 'intt', d['stage_ms_per_step']['intt'], 'roofline', d['roofline']['frac'], 'quot_intt', d['stage_ms_per_step']['quotient_intt'], 'fri_lde', d['stage_ms_per_step']['fri_lde'])"
}
for rep in 1 2 3; do
  VX_NTT2_NO_TW_TABLE=1 run "tw_table=off"
  run "tw_table=on"
done
