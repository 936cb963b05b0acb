"""The N = 2 data-parallel training step on DEVICE buffers (reference src/main.py:148-156,192,198-200 and
src/wav2vec2/losses.py:45): two processes share the one GPU of the test box and drive the real `Trainer.step` -- training
forward, CTC, backward with its per-bucket events, the communication stream, the trainable send ranges, Adam -- on a
1 + 1 split of a 2-row batch with division_factor = 2.  RCCL refuses two ranks on one device, so the process group is
`gloo` (device ranges are staged through host memory on the communication stream, wav2vec2/dist.py); everything in front
of and behind the transport is the code that runs over RCCL on 8 GPUs.

Asserted: the all-reduced gradients of both ranks equal the unsharded 2-row step's (and a whole-buffer all-reduce of the
local gradients, bit for bit in fp32), frozen slots never travel, and the post-Adam weights equal the unsharded step's.
"""

import os
import socket
import sys

import numpy as np
import pytest

import helpers as H
from wav2vec2 import variables as V

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = 4000
LABELS = np.array([[3, 4, 9, 0], [5, 5, 0, 0]], np.int32)
CHECK = ["lm_head/kernel", "lm_head/bias", "encoder/layers/1/feed_forward/output_dense/kernel",
         "encoder/layers/0/attention/q_proj/kernel", "encoder/layers/0/attention/q_proj/bias", "encoder/layer_norm/gamma",
         "encoder/pos_conv_embed/conv/weight_v", "feature_projection/projection/kernel", "feature_projection/layer_norm/beta"]


def _wave():
    return V.hash_normal("dist_gpu/wave", 2 * L, 4).reshape(2, L)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _make_trainer(rows, payload, overlap):
    import wav2vec2
    cfg = H.case_config("tiny_base")
    m = wav2vec2.Wav2Vec2ForCTC(cfg, input_shape=(rows, L))
    m.set_weights(H.case_weights("tiny_base"))
    m.freeze_feature_extractor()                                    # stage 2 of the reference (main.py:234-237)
    loss = wav2vec2.CTCLoss(cfg, (rows, L), division_factor=2)      # the GLOBAL batch on every replica (losses.py:45)
    tr = wav2vec2.Trainer(m, loss, learning_rate=1e-3, dropout=0.0, apply_spec_augment=False, seed=5,
                          allreduce_dtype=payload, overlap_all_reduce=overlap)
    return m, tr


def _rank_main(rank, world, port, out_dir, payload, overlap):
    for p in (ROOT, os.path.join(ROOT, "gsoc-wav2vec2_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    from wav2vec2 import dist as D
    torch.cuda.set_device(0)                                        # both ranks on the one GPU
    D.init(backend="gloo")
    assert D.describe()["world_size"] == 2 and D.describe()["backend"] == "gloo"
    m, tr = _make_trainer(1, payload, overlap)
    assert tr.seed == 5 * world + rank                              # per-replica randomness (MirroredStrategy replicas)
    x, labels = _wave()[rank:rank + 1], LABELS[rank:rank + 1]
    # -- the pieces of Trainer.step, with a look at the buffer in between
    logits = tr.forward(x)
    nll, grad = tr.loss.per_sample(labels, logits, with_grad=True)
    tr.backward(grad)
    torch.cuda.synchronize()
    local = tr.grad_buffer().clone()
    host = local.cpu()
    dist.all_reduce(host)                                           # whole-buffer SUM of the local gradients
    tr.backward(grad)                                               # enqueue the backward again ...
    tr.all_reduce_gradients()                                       # ... and the per-bucket collectives right behind it
    torch.cuda.synchronize()
    reduced = tr.grad_buffer().cpu()
    ranges = tr.reduce_ranges()
    sent = torch.zeros(reduced.numel(), dtype=torch.bool)
    for runs in ranges:
        for off, n in runs:
            sent[off:off + n] = True
    # frozen slots stayed home: untouched local values (zero) there, and the payload is the trainable set
    assert torch.equal(reduced[~sent], local.cpu()[~sent])
    trainable = sum(int(np.prod(v.shape)) for v in m.trainable_variables)
    assert trainable <= int(sent.sum()) < trainable + 4 * len(m.trainable_variables)
    if payload == "fp32":
        assert torch.equal(reduced[sent], host[sent]), "bucketed all-reduce != whole-buffer all-reduce"
    else:
        scale = float(host[sent].abs().max())
        assert float((reduced[sent] - host[sent]).abs().max()) < 2e-2 * scale
    grads = {n: tr.gradient(n) for n in CHECK}
    tr.apply_gradients()
    torch.cuda.synchronize()
    weights = {n: m.get_weights()[n] for n in CHECK}
    # -- and the step as one call on a fresh replica: same loss contribution, same update
    m2, tr2 = _make_trainer(1, payload, overlap)
    loss2 = float(tr2.step(x, labels))
    w2 = m2.get_weights()
    for n in CHECK:
        assert np.array_equal(w2[n], weights[n]), n
    t = torch.tensor([loss2], dtype=torch.float64)
    dist.all_reduce(t)                                              # SUM of the per-replica losses = the global loss
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), loss=np.array([float(t)]),
             **{"g:" + n: g for n, g in grads.items()}, **{"w:" + n: w for n, w in weights.items()})
    dist.destroy_process_group()


@pytest.mark.parametrize("payload,overlap", [("fp32", True), ("bf16", True), ("fp32", False)])
def test_two_ranks_on_one_gpu_drive_the_real_trainer(tmp_path, payload, overlap):
    import torch
    import torch.multiprocessing as mp
    assert torch.cuda.is_available()
    mp.spawn(_rank_main, args=(2, _free_port(), str(tmp_path), payload, overlap), nprocs=2, join=True)
    # the unsharded reference: ONE process, both rows, the same division factor
    m, tr = _make_trainer(2, "fp32", False)
    loss = float(tr.step(_wave(), LABELS))
    want_w = m.get_weights()
    want_g = {n: tr.gradient(n) for n in CHECK}
    got = [dict(np.load(tmp_path / f"rank{r}.npz")) for r in range(2)]
    # (the first Adam step moves an element by ~lr sign(g): an element whose gradient is at the noise level may flip)
    gtol, wtol = (2e-5, 5e-5) if payload == "fp32" else (2e-2, 2.1e-3)
    for r in range(2):
        assert abs(got[r]["loss"][0] - loss) < 1e-5 * abs(loss)
        for n in CHECK:
            g, w = got[r]["g:" + n], got[r]["w:" + n]
            scale = max(1e-6, float(np.abs(want_g[n]).max()))
            assert np.abs(g - want_g[n]).max() < gtol * scale, (n, float(np.abs(g - want_g[n]).max()), scale)
            assert np.abs(w - want_w[n]).max() < wtol, (n, float(np.abs(w - want_w[n]).max()))
    # both replicas hold the same weights after the step (what keeps data-parallel replicas in sync)
    for n in CHECK:
        assert np.array_equal(got[0]["w:" + n], got[1]["w:" + n]), n


def _bench_objects(r):
    """bench.py's contract: stdout carries ONE compact JSON line (< 4 KB: the driver keeps 8 KB of stdout and parses the last line),
    the complete object is on stderr behind the tag BENCH_FULL (and in gpurun_out/bench_full.json)."""
    import json
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    assert r.stdout.rstrip().splitlines()[-1] == lines[0]
    assert len(lines[0]) < 4096, len(lines[0])
    compact = json.loads(lines[0])
    fulls = [ln for ln in r.stderr.splitlines() if ln.startswith("BENCH_FULL ")]
    assert len(fulls) == 1
    return compact, json.loads(fulls[0][len("BENCH_FULL "):])


def test_bench_two_ranks_report_every_baseline_config_with_the_collective():
    """The one command the driver runs, at N = 2: `bench.py --gpus 2` self-launches two ranks under torch.distributed.run and its
    single JSON line must carry, beside the headline, BASELINE configs[2] / [3] / [4] -- the two training legs with the bucketed
    gradient all-reduce INSIDE their timed steps and the measured `allreduce` object (payload, buckets, step time without the
    collective, exposed time, standalone time, bus bandwidth).  The test box has one GPU and RCCL refuses two ranks on one device, so
    the group is gloo here (`--backend gloo`; the default and the only quotable backend is nccl = RCCL); the side legs run at an
    eighth of their batch to keep the test short.  Everything else is the code path of the driver's 8-GPU run."""
    import json
    import subprocess
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None), env.pop("RANK", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "2", "--warmup", "1",
           "--side-shrink", "8", "--no-cpu-baseline", "--no-alt"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    compact, js = _bench_objects(r)
    assert compact["n_gpus"] == 2 and compact["comm"]["backend"] == "gloo" and compact["roofline"]["frac"] > 0
    for key in ("configs2_train_bf16", "configs4_large_train_bf16"):          # the compact line carries the measured collective
        assert compact[key]["ms_per_step"] > 0 and compact[key]["frac"] > 0 and "exposed_ms" in compact[key]["allreduce"], compact[key]
    assert compact["configs3_large_fwd_f32"]["value"] > 0
    assert js["n_gpus"] == 2 and js["comm"]["backend"] == "gloo" and js["comm"]["world_size"] == 2
    assert js["config"]["global_batch"] == 64 and js["roofline"]["frac"] > 0
    for key, train in (("configs2_train_bf16", True), ("configs3_large_fwd_f32", False), ("configs4_large_train_bf16", True)):
        leg = js[key]
        assert "error" not in leg, leg
        assert leg["n_gpus"] == 2 and leg["ms_per_step"] > 0 and leg["value"] > 0
        assert leg["roofline"]["frac"] > 0 and leg["roofline"]["kernel_launches_per_step"] >= leg["roofline"]["launches_per_step"]
        shares = sum(v["share"] for v in leg["families"].values())
        assert 0.02 < shares < 1.3, shares                  # shares are of the TIMED step (here dominated by gloo's host staging): they need not add to 1
        if train:
            ar = leg["allreduce"]
            assert ar["world_size"] == 2 and ar["buckets"] >= 13 and ar["payload_bytes"] > 0
            assert ar["ms_per_step_without_collective"] > 0 and ar["standalone_ms"] > 0 and ar["busbw_GBps"] > 0
            assert np.isfinite(leg["final_loss"])
    assert js["configs2_train_bf16"]["allreduce"]["payload_bytes"] == 4 * (90195104 + 768)      # 360.8 MB fp32 for base


def test_bench_eight_ranks_over_gloo_is_the_drivers_command():
    """The driver's own 8-GPU command line (`bench.py --gpus 8`) with eight processes on the one GPU over gloo:
    the world size the scaling run uses, the base model's 14 gradient buckets all-reduced among eight ranks inside the timed steps,
    the loss divisor 8 x the per-rank batch -- and the ONE compact stdout line must carry the measured collective of the training
    leg (`allreduce.exposed_ms`, `standalone_ms`, `busbw_GBps`).  Only configs[2] runs as a side leg here (`--side-legs`): gloo stages
    every all-reduce through host memory, and the large model's 1.25-GB payload among eight processes takes minutes per step (the
    two-rank test above runs all three legs).  No number of this run is quotable (one GPU, host-staged collectives) -- it exists so
    that the driver's first real `--gpus 8` run cannot fail on plumbing or on parsing."""
    import subprocess
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None), env.pop("RANK", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--backend", "gloo", "--steps", "1", "--warmup", "1",
           "--side-shrink", "16", "--side-legs", "configs2_train_bf16", "--no-cpu-baseline", "--no-alt", "--no-profile"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    compact, js = _bench_objects(r)
    assert compact["n_gpus"] == 8 and compact["config"]["global_batch"] == 256 and compact["scaling"] == "weak" and compact["value"] > 0
    ar = compact["configs2_train_bf16"]["allreduce"]
    assert "exposed_ms" in ar and ar["standalone_ms"] > 0 and ar["busbw_GBps"] > 0, compact["configs2_train_bf16"]
    assert "configs3_large_fwd_f32" not in compact and "configs4_large_train_bf16" not in compact
    assert js["n_gpus"] == 8 and js["comm"]["world_size"] == 8 and js["config"]["global_batch"] == 256 and js["scaling"] == "weak"
    leg = js["configs2_train_bf16"]
    assert "error" not in leg, leg
    ar = leg["allreduce"]
    assert leg["n_gpus"] == 8 and leg["global_batch"] == 8 * 2 and ar["world_size"] == 8 and ar["buckets"] == 14 and ar["collectives_per_step"] >= 14
    assert ar["standalone_ms"] > 0 and np.isfinite(leg["final_loss"]) and ar["engine"].startswith("torch.distributed")
    assert ar["payload_bytes"] == 4 * (90195104 + 768)


# ---------------------------------------------------------------- the library's own RCCL collective (csrc/comm.hip) ----------
@pytest.mark.parametrize("collective", ["native", "native-rs"])
def test_native_collective_world_of_one(collective):
    """the C ABI's own data-parallel collective (w2v2_comm_init / w2v2_allreduce_bucket / w2v2_allreduce_finish,
    RCCL bound at run time).  One GPU here, so a world of ONE rank: what can be pinned is everything but the wire --
      * the communicator comes up (w2v2_comm_unique_id -> ncclCommInitRank) with the library's own communication stream;
      * the runs it sends per bucket are EXACTLY the host-side ranges of the torch engine (Trainer.reduce_ranges: frozen conv
        stack excluded, adjacent trainable slots merged), in both training stages of the reference (main.py:210,234-237);
      * a forced all-reduce over every bucket waits for the bucket events of the backward enqueued right before it, runs through
        RCCL (both algorithms) and leaves the gradients bit-identical (SUM over one replica); the reported payload is the
        reference's 4 bytes x trainable elements; the optimizer step behind it equals the torch-engine step bit for bit."""
    import torch
    from wav2vec2 import _native as N
    from wav2vec2 import dist as D
    x = _wave()
    m_ref, tr_ref = _make_trainer(2, "fp32", True)
    loss_ref = float(tr_ref.step(x, LABELS))
    m, tr = _make_trainer(2, "fp32", True)
    tr.collective = collective
    assert D.native_comm_info(m)[1] == 0                            # no communicator yet
    assert D.native_comm_init(m) == (0, 1)
    r, w, ver = D.native_comm_info(m)
    assert (r, w) == (0, 1) and ver >= 20000                        # RCCL reports NCCL-style version codes (2.26.6 -> 22606)
    logits = tr.forward(x)
    nll, grad = tr.loss.per_sample(LABELS, logits, with_grad=True)
    tr.backward(grad)
    torch.cuda.synchronize()
    local = tr.grad_buffer().clone()
    assert tr.native_reduce_ranges() == tr.reduce_ranges()
    n_tr = sum(n for runs in tr.reduce_ranges() for _, n in runs)
    tr.backward(grad)                                               # enqueue the backward again ...
    tr.all_reduce_gradients(force=True)                             # ... and the library's per-bucket collectives right behind it
    torch.cuda.synchronize()
    assert torch.equal(tr.grad_buffer(), local) and tr._native_bytes_last == 4 * n_tr
    tr.apply_gradients()
    torch.cuda.synchronize()
    assert abs(float(nll.sum()) / 2 - loss_ref) <= 1e-6 * abs(loss_ref)      # (division_factor = 2: losses.py:45)
    for n in CHECK:
        assert np.array_equal(m.get_weights()[n], m_ref.get_weights()[n]), n
    # stage 1 of the reference: only lm_head trains -> one run, in bucket 0
    m.layers[0].trainable = False
    tr._ranges_cache = {}
    logits = tr.forward(x)
    nll, grad = tr.loss.per_sample(LABELS, logits, with_grad=True)
    tr.backward(grad)
    assert tr.native_reduce_ranges() == tr.reduce_ranges()
    assert [len(r) for r in tr.native_reduce_ranges()] == [1] + [0] * (len(tr.reduce_ranges()) - 1)
    tr.all_reduce_gradients(force=True)
    torch.cuda.synchronize()
    # a destroyed communicator refuses loudly
    N.check(m._lib.w2v2_comm_destroy(m._handle), "w2v2_comm_destroy")
    assert m._lib.w2v2_allreduce_bucket(m._handle, 0, 0) != 0 and b"no communicator" in m._lib.w2v2_last_error()
