#!/bin/bash
# tools/gpu_fuzz_seeds.sh <outdir> <seed> [<seed> ...] — the differential fuzzers (engine vs oracle over corrupted witnesses,
# every HAMT route, self-planned shards) on OTHER corpora: IPCFP_FUZZ_SEED moves every fixed seed of those tests
# (tests/conftest.py fuzz_seed).  A failure here is a finding, not a regression of the suite: read <outdir>/seed_<k>.txt.
out=$1; shift
mkdir -p "$out"
for k in "$@"; do
  IPCFP_FUZZ_SEED=$k timeout 600 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_shard_pull_fuzz.py \
      "tests/test_gpu_hamt_routes.py::test_mutated_witnesses_every_route" "tests/test_gpu_hamt_routes.py::test_state_tree_gets_every_route_and_wrong_types" \
      -q -m gpu --durations=3 > "$out/seed_$k.txt" 2>&1
  echo "seed $k: $(grep -E ' passed| failed| error' "$out/seed_$k.txt" | tail -1)"
done
