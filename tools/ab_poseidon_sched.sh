# A/B on ONE box of the two Poseidon instruction-count cuts of round 4: (1) full round 3's dense layer opens the first block of partial rounds, (2) the sponge computes only the live rows of a permutation's last layer;
# libvxprover_prev.so = neither, libvxprover_prev2.so = (1) only (both built from the commits before), alternating
export VX_JIT_CACHE_DIR=$PWD/.jit_cache
run() {
  python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-host-witness-leg --no-dag-leg --no-chip-leg --no-dag-stark-leg 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', round(d['ms_per_step'], 2), 'hash_leaves', d['stage_ms_per_step']['hash_leaves'], 'merkle', d['stage_ms_per_step']['merkle_levels'], 'fri_hash', d['stage_ms_per_step']['fri_hash_leaves'], 'lde', d['stage_ms_per_step']['lde'])"
}
for rep in 1 2 3; do
  VXPROVER_LIB=$PWD/vectorx_amd/libvxprover_prev.so run "round3_schedule"
  VXPROVER_LIB=$PWD/vectorx_amd/libvxprover_prev2.so run "merged_schedule"
  run "merged_schedule+live_rows"
done
