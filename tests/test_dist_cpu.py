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


# ---------------------------------------------------------------------------------------------------------------
# The training collective (reference src/main.py:156,198-200 + losses.py:45): every replica divides its loss by
# the GLOBAL batch, gradients are SUMmed across replicas.  Two gloo ranks take a 2 + 2 split of a 4-row batch through the
# training oracle with division_factor = 4, lay their gradients out in the flat buffer the HIP library uses, and reduce it
# with the SAME bucket / trainable-range / all-reduce code the Trainer runs over RCCL (wav2vec2/dist.py) -- the result must
# equal the unsharded gradients.
# ---------------------------------------------------------------------------------------------------------------
def _flat_from_grads(layout, total, grads, frozen_fill):
    buf = np.full(total, frozen_fill, dtype=np.float64)
    for name, (off, n) in layout.items():
        if name in grads:
            g = grads[name]
            buf[off:off + ((n + 3) & ~3)] = 0.0
            if g is not None:
                buf[off:off + n] = np.asarray(g, np.float64).reshape(-1)
    return buf


def _grad_worker(rank, world, port, out_dir, payload):
    for p in (ROOT, os.path.join(ROOT, "gsoc-wav2vec2_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import helpers as HH
    from oracle import w2v2_torch_train as TT
    from wav2vec2 import dist as DD
    from wav2vec2 import variables as V
    torch.set_num_threads(1)
    DD.init(backend="gloo")
    cfg, weights = HH.case_config("tiny_base"), HH.case_weights("tiny_base")
    total_rows = 4
    x = V.hash_normal("dist/train_wave", total_rows * 4000, 9).reshape(total_rows, 4000)
    labels = np.array([[3, 4, 9, 0], [5, 5, 0, 0], [7, 1, 2, 2], [30, 0, 0, 0]], np.int32)
    lo, hi = DD.shard_bounds(total_rows, world, rank)
    loss, _, _, grads = TT.loss_and_grads(cfg, weights, x[lo:hi], labels[lo:hi], division_factor=total_rows)
    specs = V.variable_specs(cfg)
    layout, total = DD.flat_layout(specs)
    buf = torch.from_numpy(_flat_from_grads(layout, total, grads, frozen_fill=7.0))      # 7.0 marks frozen slots
    if payload == "bf16":
        buf = buf.float()
    trainable = set(grads)                                              # everything but the conv stack
    sent = 0
    finishers = []
    for bucket in DD.gradient_buckets(layout, total, cfg.num_layers):
        for off, n in DD.trainable_ranges(layout, bucket, trainable):
            finishers.append(DD.all_reduce_range(buf, off, n, torch.bfloat16 if payload == "bf16" else None, async_op=True))
            sent += n
    for f in finishers:
        f()
    loss_sum = DD.max_over_ranks(0.0)                                    # (exercise the helper under this group too)
    t = torch.tensor([loss], dtype=torch.float64)
    torch.distributed.all_reduce(t)
    if rank == 0:
        np.save(os.path.join(out_dir, "reduced.npy"), buf.double().numpy())
        np.save(os.path.join(out_dir, "meta.npy"), np.array([sent, float(t.item()), loss_sum]))
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("payload", ["fp32", "bf16"])
def test_two_rank_gradient_all_reduce_equals_unsharded(tmp_path, payload):
    from oracle import w2v2_torch_train as TT
    from wav2vec2 import variables as V
    world = 2
    mp.spawn(_grad_worker, args=(world, _free_port(), str(tmp_path), payload), nprocs=world, join=True)
    cfg, weights = H.case_config("tiny_base"), H.case_weights("tiny_base")
    x = V.hash_normal("dist/train_wave", 4 * 4000, 9).reshape(4, 4000)
    labels = np.array([[3, 4, 9, 0], [5, 5, 0, 0], [7, 1, 2, 2], [30, 0, 0, 0]], np.int32)
    loss, _, _, grads = TT.loss_and_grads(cfg, weights, x, labels, division_factor=4)
    specs = V.variable_specs(cfg)
    layout, total = D.flat_layout(specs)
    want = _flat_from_grads(layout, total, grads, frozen_fill=7.0)
    got = np.load(tmp_path / "reduced.npy")
    sent, loss_sum, _ = np.load(tmp_path / "meta.npy")
    # the summed per-rank losses (each / global batch) are the global loss
    assert abs(loss_sum - loss) < 1e-9 * abs(loss)
    frozen = np.ones(total, bool)
    for name, (off, n) in layout.items():
        if name in grads:
            frozen[off:off + ((n + 3) & ~3)] = False
    # frozen conv-stack slots never travelled: still the marker on rank 0 (a SUM would have made them 14)
    assert np.all(got[frozen] == 7.0)
    trainable_elems = sum(n for name, (off, n) in layout.items() if name in grads)
    assert trainable_elems <= sent < trainable_elems + 4 * len(grads)          # payload = trainable slots (+ alignment pads)
    scale = np.abs(want[~frozen]).max()
    err = np.abs(got[~frozen] - want[~frozen]).max()
    if payload == "fp32":
        assert err < 1e-12 * max(1.0, scale), err
    else:
        assert err < 2e-2 * scale, (err, scale)                                 # two bf16 roundings of the payload


def _payload_worker(rank, world, port, out_dir, n):
    for p in (ROOT, os.path.join(ROOT, "gsoc-wav2vec2_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from wav2vec2 import dist as DD
    from wav2vec2 import variables as V
    torch.set_num_threads(1)
    DD.init(backend="gloo")
    g = torch.from_numpy(V.hash_normal(f"dist/payload{rank}", n, 3).astype(np.float32) * 1e-3)      # a rank's gradient shard contribution
    for dtype, tag in ((None, "fp32"), (torch.bfloat16, "bf16")):
        buf = g.clone()
        DD.all_reduce_range(buf, 0, n, dtype, async_op=False)()
        if rank == 0:
            np.save(os.path.join(out_dir, f"sum_{tag}.npy"), buf.double().numpy())
    torch.distributed.destroy_process_group()


def test_eight_rank_bf16_payload_error_bound(tmp_path):
    """Trainer(allreduce_dtype="bf16") at the world size of BASELINE configs[2] / [4]: the collective SUMs in bf16, so every partial sum of
    the reduction is rounded again.  Eight gloo ranks, one million-element gradient each: the fp32 payload equals the fp64 sum to fp32
    rounding; the bf16 payload stays within 2e-2 of max |sum| (worst element) and 8e-3 rms -- the figure dist.all_reduce_range documents
    (the two-rank test above pins 2e-2 at two roundings; this is the eight-rank case the review asked for)."""
    from wav2vec2 import variables as V
    world, n = 8, 1 << 20
    mp.spawn(_payload_worker, args=(world, _free_port(), str(tmp_path), n), nprocs=world, join=True)
    want = sum(V.hash_normal(f"dist/payload{r}", n, 3).astype(np.float32).astype(np.float64) * np.float64(np.float32(1e-3)) for r in range(world))
    f32, b16 = np.load(tmp_path / "sum_fp32.npy"), np.load(tmp_path / "sum_bf16.npy")
    scale = np.abs(want).max()
    assert np.abs(f32 - want).max() < 1e-6 * scale
    err = np.abs(b16 - want)
    print(f"8-rank bf16 payload: max err {err.max() / scale:.2e} of max |sum|, rms {np.sqrt((err ** 2).mean()) / np.sqrt((want ** 2).mean()):.2e}")
    assert err.max() < 2e-2 * scale
    assert np.sqrt((err ** 2).mean()) < 8e-3 * np.sqrt((want ** 2).mean())


def test_bucket_layout_tiles_the_buffer():
    from wav2vec2 import variables as V
    for case in ("tiny_base", "base_sample_padded", "robust_masked"):
        cfg = H.case_config(case)
        layout, total = D.flat_layout(V.variable_specs(cfg))
        buckets = D.gradient_buckets(layout, total, cfg.num_layers)
        assert len(buckets) == cfg.num_layers + 2
        spans = sorted(buckets)
        assert spans[0][0] == 0 and spans[-1][0] + spans[-1][1] == total
        assert all(a[0] + a[1] == b[0] for a, b in zip(spans, spans[1:]))
        assert buckets[0][0] == layout["lm_head/kernel"][0]
        trainable = {n for n in layout if not n.startswith("feature_extractor/")}
        sent = sum(n for bk in buckets for _, n in D.trainable_ranges(layout, bk, trainable))
        if case == "base_sample_padded":
            # the reference's stage-2 payload: 90,195,104 trainable elements + masked_spec_embed's 768
            assert sent == 90195104 + 768
            assert total - sent == 4200448                                    # the frozen conv stack stays home
