"""GPU, two processes: the ENGINE's communicator with n_ranks = 2 — `ipcfp_comm_create` (ncclCommInitRank resolved from
librccl.so by libipcfp.so), `ipcfp_allgather_device` and `ipcfp_allgather_segments` on the engine's stream, and a
query-range shard of config 4 gathered across the two ranks.  Needs two visible GPUs (RCCL refuses two ranks on one
device); skipped on the single-GPU boxes, where tests/test_gpu_range_shards.py drives the same calls with one rank."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist

    import ipc_filecoin_proofs_amd as ipcfp
    from tools.synth import Tipset

    torch.cuda.set_device(rank)
    dist.init_process_group("gloo", rank=rank, world_size=world)  # the HOST channel only (the 128-byte id)
    eng = ipcfp.Engine(rank)
    uid = [ipcfp.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(uid, src=0)
    comm = ipcfp.Comm(eng, uid[0], world, rank)
    dev = f"cuda:{rank}"
    # 1) plain all-gather of distinct bytes
    send = torch.full((4096,), 17 + rank, dtype=torch.uint8, device=dev)
    recv = torch.zeros(world * 4096, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    comm.allgather_device(send.data_ptr(), recv.data_ptr(), 4096)
    eng.sync()
    g = recv.cpu().numpy().reshape(world, 4096)
    ok = all((g[r] == 17 + r).all() for r in range(world))
    # 2) config 4 cut by query range, statuses gathered
    T = Tipset(n_receipts=8, n_planted=0, n_actors=20_000, keep_full_state=0, n_actor_queries=1010, seed=31)

    def idaddr(i):
        b = bytearray([0])
        while True:
            c = i & 0x7F
            i >>= 7
            if i:
                b.append(c | 0x80)
            else:
                b.append(c)
                return bytes(b)

    keys = [idaddr(int(i)) for i in T.query_ids]
    n = len(keys)
    with eng.witness(T.data, T.off, T.lens, T.cids) as w:
        want, _ = w.hamt_get(T.actors_root, 5, "actor_state", keys)
        lo, hi = ipcfp.shard_range(n, world, rank)
        m = hi - lo
        width = ((n + world - 1) // world + 15) & ~15
        kl = np.array([len(k) for k in keys[lo:hi]], dtype=np.uint32)
        ko = np.zeros(m, dtype=np.uint32)
        ko[1:] = np.cumsum(kl[:-1])
        kb = np.frombuffer(b"".join(keys[lo:hi]) + bytes(32), dtype=np.uint8).copy()
        d_kb, d_ko, d_kl = torch.from_numpy(kb).to(dev), torch.from_numpy(ko.view(np.int32)).to(dev), torch.from_numpy(kl.view(np.int32)).to(dev)
        d_st = torch.zeros(width, dtype=torch.uint8, device=dev)
        d_recv = torch.zeros(world * width, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        w.hamt_get_device(T.actors_root, 5, "actor_state", d_kb.data_ptr(), d_ko.data_ptr(), d_kl.data_ptr(), m, d_st.data_ptr())
        comm.allgather_device(d_st.data_ptr(), d_recv.data_ptr(), width)
        eng.sync()
        gg = d_recv.cpu().numpy().reshape(world, width)
        merged = np.concatenate([gg[r, : ipcfp.shard_range(n, world, r)[1] - ipcfp.shard_range(n, world, r)[0]] for r in range(world)])
        ok = ok and np.array_equal(merged, want) and (want == 1).sum() > n // 2
    np.save(os.path.join(out_dir, f"r{rank}.npy"), np.array([int(ok)]))
    comm.close()
    eng.close()
    dist.barrier()
    dist.destroy_process_group()


def test_engine_communicator_with_two_ranks(tmp_path):
    import torch

    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU visible: RCCL cannot put two ranks on one device (the 1-rank path is tested elsewhere)")
    import torch.multiprocessing as mp

    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        assert np.load(tmp_path / f"r{r}.npy")[0] == 1
