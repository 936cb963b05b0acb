"""The N > 1 path on CPU: world_size 2 over gloo (127.0.0.1).  Each rank runs its batch shard through
a forward (here: the CPU oracle on a tiny config -- tests may use it as the compute stand-in), the
shards are gathered, and the result must equal the unsharded forward; the timing reduction is a MAX."""

import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

import helpers as H
from wav2vec2 import dist as D

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_bounds_cover_and_balance():
    for total, world in [(256, 8), (32, 1), (10, 4), (3, 8), (128, 8)]:
        spans = [D.shard_bounds(total, world, r) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == total
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [hi - lo for lo, hi in spans]
        assert max(sizes) - min(sizes) <= 1
    assert D.shard_bounds(256, 8, 3) == (96, 128)          # BASELINE config 3: 8 x 32
    with pytest.raises(ValueError):
        D.shard_bounds(8, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total_rows, out_dir):
    for p in (ROOT, os.path.join(ROOT, "gsoc-wav2vec2_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import helpers as HH
    from oracle import w2v2_oracle as O
    from wav2vec2 import dist as DD
    from wav2vec2 import variables as V
    torch.set_num_threads(1)
    w_, r_ = DD.init(backend="gloo")
    assert (w_, r_) == (world, rank)
    cfg = HH.case_config("tiny_base")
    weights = HH.case_weights("tiny_base")
    x = V.hash_normal("dist/wave", total_rows * 3000, 9).reshape(total_rows, 3000)
    lo, hi = DD.shard_bounds(total_rows, world, rank)
    DD.barrier()
    local = torch.from_numpy(O.ctc_forward(cfg, weights, x[lo:hi]))      # this rank's shard only
    DD.barrier()
    full = DD.gather_rows(local, total_rows)
    slowest = DD.max_over_ranks(1.0 + rank)                              # pretend rank r took 1+r seconds
    if rank == 0:
        np.save(os.path.join(out_dir, "gathered.npy"), full.numpy())
        np.save(os.path.join(out_dir, "slowest.npy"), np.array([slowest]))
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("total_rows", [4, 5])
def test_two_rank_sharded_forward_matches_unsharded(tmp_path, total_rows):
    from oracle import w2v2_oracle as O
    from wav2vec2 import variables as V
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), total_rows, str(tmp_path)), nprocs=world, join=True)
    cfg, weights = H.case_config("tiny_base"), H.case_weights("tiny_base")
    x = V.hash_normal("dist/wave", total_rows * 3000, 9).reshape(total_rows, 3000)
    ref = O.ctc_forward(cfg, weights, x)
    got = np.load(tmp_path / "gathered.npy")
    assert got.shape == ref.shape
    # batch rows are independent: sharding must not change a single bit of any row's result
    assert np.array_equal(got, ref)
    assert np.load(tmp_path / "slowest.npy")[0] == 2.0
