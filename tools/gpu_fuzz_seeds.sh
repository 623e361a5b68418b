#!/bin/bash
# tools/gpu_fuzz_seeds.sh <outdir> <seed> [<seed> ...] — the differential fuzzers (engine vs oracle over corrupted witnesses,
# every HAMT route, self-planned shards) on OTHER corpora: IPCFP_FUZZ_SEED moves every fixed seed of those tests
# (tests/conftest.py fuzz_seed).  A failure here is a finding, not a regression of the suite: read <outdir>/seed_<k>.txt.
# FUZZ_WIDE=1: also the parity tests over generated tipsets whose seeds move with it (events, event table, transport forms, range
# shards, bundles, the ABI boundary) — minutes per seed instead of seconds; their seed-specific count assertions may trip.
out=$1; shift
mkdir -p "$out"
wide=""
[ -n "$FUZZ_WIDE" ] && wide="tests/test_gpu_events.py tests/test_gpu_event_table.py tests/test_gpu_transport.py tests/test_gpu_range_shards.py tests/test_gpu_sharding.py tests/test_gpu_bundle.py tests/test_gpu_boundary.py tests/test_gpu_shard_pull.py"
for k in "$@"; do
  IPCFP_FUZZ_SEED=$k timeout ${FUZZ_TIMEOUT:-600} python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_shard_pull_fuzz.py "tests/test_gpu_limits.py::test_fuzz_of_a_witness_with_long_cids" "tests/test_gpu_limits.py::test_fuzz_of_a_wide_tipset" \
      "tests/test_gpu_hamt_routes.py::test_mutated_witnesses_every_route" "tests/test_gpu_hamt_routes.py::test_state_tree_gets_every_route_and_wrong_types" "tests/test_gpu_hamt_routes.py::test_storage_values_in_every_spelling_both_routes" $wide \
      -q -m gpu --durations=3 > "$out/seed_$k.txt" 2>&1
  echo "seed $k: $(grep -E ' passed| failed| error' "$out/seed_$k.txt" | tail -1)"
done
