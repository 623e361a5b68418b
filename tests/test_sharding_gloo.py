"""CPU, world_size 2, gloo: the N > 1 path of the engine is "shard by index, verify locally, one
all-gather".  No GPU here, so each rank's local verdicts come from the CPU oracle (the checker);
what is under test is the shard plan and the gather: the gathered result must equal the unsharded
one, for uneven shard sizes too."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_receipts, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    import claims
    import oracle_lib
    from ipc_filecoin_proofs_amd import shard
    from tools.synth import Tipset

    dist.init_process_group("gloo", rank=rank, world_size=world)
    tip = Tipset(n_receipts=n_receipts, n_planted=5, variety=1, seed=4242)
    orc = oracle_lib.load()
    st = orc.store(tip.data, tip.off, tip.lens, tip.cids)
    n = len(tip.claim_exec)
    lo, hi = shard.shard_bounds(n, world)[rank]
    ec = claims.EventClaims(tip, indices=np.arange(lo, hi))
    ec_all = claims.EventClaims(tip)
    if hi > lo:
        ec.arr[0].exec_index += 1  # first claim of every shard is wrong: the gather must keep positions
        ec_all.arr[lo].exec_index += 1
    other = shard.shard_bounds(n, world)[1 - rank]
    if other[1] > other[0]:
        ec_all.arr[other[0]].exec_index += 1
    local = st.verify_event_proofs(ec, mode=1, threads=1)
    merged = shard.gather_bytes(local, n, dist)
    want = st.verify_event_proofs(ec_all, mode=1, threads=1)
    ok = np.array_equal(merged, want)
    # bitmap form: block-index shards of the CID check
    nb = tip.n_blocks
    blo, bhi = shard.shard_bounds(nb, world)[rank]
    okb, _ = orc.blake2b256_verify(tip.data, tip.off[blo:bhi], tip.lens[blo:bhi],
                                   np.ascontiguousarray(tip.cids[blo:bhi, 6:38]))
    merged_b = shard.gather_bytes(okb, nb, dist)
    ok = ok and merged_b.sum() == nb and len(merged_b) == nb
    np.save(os.path.join(out_dir, f"r{rank}.npy"), np.array([ok, len(merged), n, int((merged != 1).sum())]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_receipts", [301, 64])
def test_two_rank_shard_and_gather(tmp_path, n_receipts):
    import torch.multiprocessing as mp

    port = _free_port()
    mp.spawn(_worker, args=(2, port, n_receipts, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        ok, m, n, bad = np.load(tmp_path / f"r{r}.npy")
        assert ok and m == n and bad >= 2


def test_shard_bounds():
    from ipc_filecoin_proofs_amd import shard

    assert shard.shard_bounds(10, 3) == [(0, 4), (4, 7), (7, 10)]
    assert shard.shard_bounds(2, 4) == [(0, 1), (1, 2), (2, 2), (2, 2)]
    assert shard.shard_bounds(0, 2) == [(0, 0), (0, 0)]
    b = shard.shard_bounds(1_000_003, 8)
    assert b[0][0] == 0 and b[-1][1] == 1_000_003 and all(x[1] == y[0] for x, y in zip(b, b[1:]))
    assert shard.pack_bits(np.array([1, 0, 0, 0, 0, 0, 0, 0, 1])).tolist() == [1, 1]


def _padded_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist

    from ipc_filecoin_proofs_amd.shard import PaddedGather

    dist.init_process_group("gloo", rank=rank, world_size=world)
    # ranks own independently generated shards: message lengths differ (bench.py's per-step collective)
    local_len = 1000 + 37 * rank
    g = PaddedGather(local_len, dist)
    ok = g.lens == [1000 + 37 * r for r in range(world)] and g.width == 1000 + 37 * (world - 1)
    for step in range(3):
        g.payload[:local_len] = torch.full((local_len,), (rank * 16 + step) & 0xFF, dtype=torch.uint8)
        g.run()
        for r in range(world):
            m = g.message(r)
            ok = ok and len(m) == 1000 + 37 * r and bool((m == ((r * 16 + step) & 0xFF)).all())
    np.save(os.path.join(out_dir, f"p{rank}.npy"), np.array([ok]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_padded_gather(tmp_path):
    """bench.py --gpus N closes every step with this collective; uneven message lengths must work."""
    import torch.multiprocessing as mp

    port = _free_port()
    mp.spawn(_padded_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        assert np.load(tmp_path / f"p{r}.npy")[0]
