"""Data-parallel helpers: one process per GPU, torch.distributed (backend "nccl" = RCCL on ROCm,
"gloo" in the CPU tests).

The forward path shards along batch only -- utterances are independent (GroupNorm / LayerNorm
statistics are per sample, attention is within an utterance, CTC is per sample), which is exactly the
reference's scheme (src/main.py:156,192).  No collective touches the data path of a forward; the only
cross-rank traffic is the timing reduction and an optional gather of the outputs.
"""

import os


def env_world():
    """(world_size, rank, local_rank) as torch.distributed.run exports them."""
    return (int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")),
            int(os.environ.get("LOCAL_RANK", "0")))


def init(backend="nccl", device=None):
    """Initialise the default process group from the launcher's env (MASTER_ADDR defaults to 127.0.0.1)."""
    import torch.distributed as dist
    world, rank, _ = env_world()
    launched = "RANK" in os.environ and "MASTER_PORT" in os.environ      # under torch.distributed.run, even with 1 rank
    if dist.is_initialized() or (world == 1 and not launched):
        return world, rank
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    kw = {}
    if backend == "nccl" and device is not None:
        kw["device_id"] = device
    dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return world, rank


def describe():
    """What the process group itself reports: backend, world size, rank count per node, the RCCL version torch links
    (backend "nccl" IS RCCL on ROCm).  A single un-launched process has no group: backend None, world size 1."""
    import torch
    import torch.distributed as dist
    out = {"backend": None, "world_size": 1, "rccl_version": None, "launcher": "torch.distributed.run" if "RANK" in os.environ else "none"}
    try:
        out["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception:  # noqa: BLE001 -- a build without RCCL reports None
        pass
    if dist.is_available() and dist.is_initialized():
        out["backend"] = dist.get_backend()
        out["world_size"] = dist.get_world_size()
    return out


COMM_ID_BYTES = 128          # include/w2v2.h: W2V2_COMM_ID_BYTES = sizeof(ncclUniqueId)


def native_comm_init(model, rank=None, world=None, unique_id=None):
    """Give `model` the library's own RCCL communicator (include/w2v2.h: w2v2_comm_init; csrc/comm.hip) -- the engine behind
    `Trainer(..., collective="native")`.  Rank 0 draws the rendezvous id (w2v2_comm_unique_id) and the other ranks receive its 128
    bytes: over the default torch.distributed group when one exists (any backend -- it carries bytes, not gradients), or from
    `unique_id` when the host has its own channel (the torch-free stub of INTEGRATION.md section 3 passes it through a file).
    Returns (rank, world)."""
    import ctypes as C

    from . import _native as N
    if rank is None or world is None:
        world, rank, _ = env_world()
        try:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                world, rank = dist.get_world_size(), dist.get_rank()
        except ImportError:
            pass
    lib = model._lib
    buf = (C.c_uint8 * COMM_ID_BYTES)()
    if unique_id is not None:
        if len(unique_id) != COMM_ID_BYTES:
            raise ValueError(f"unique_id must be {COMM_ID_BYTES} bytes")
        C.memmove(buf, bytes(unique_id), COMM_ID_BYTES)
    else:
        if rank == 0:
            N.check(lib.w2v2_comm_unique_id(buf, COMM_ID_BYTES), "w2v2_comm_unique_id")
        if world > 1:
            import torch.distributed as dist
            if not (dist.is_available() and dist.is_initialized()):
                raise RuntimeError("native_comm_init: world > 1 needs a channel for the rendezvous id: pass unique_id, or initialise torch.distributed")
            box = [bytes(buf) if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            C.memmove(buf, box[0], COMM_ID_BYTES)
    N.check(lib.w2v2_comm_init(model._handle, buf, COMM_ID_BYTES, int(rank), int(world)), "w2v2_comm_init")
    return rank, world


def native_comm_info(model):
    """(rank, world, RCCL version code) of the model's native communicator; world 0 = none."""
    import ctypes as C

    from . import _native as N
    r, w, v = C.c_int32(), C.c_int32(), C.c_int32()
    N.check(model._lib.w2v2_comm_info(model._handle, C.byref(r), C.byref(w), C.byref(v)), "w2v2_comm_info")
    return r.value, w.value, v.value


def shard_bounds(total_rows, world, rank):
    """Contiguous [lo, hi) slice of a global batch for `rank`; the first total % world ranks take one
    extra row (the reference keeps a fixed 32 rows per replica, src/main.py:41,156)."""
    if not 0 <= rank < world:
        raise ValueError(f"rank {rank} outside world of {world}")
    q, r = divmod(int(total_rows), int(world))
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def barrier(sync_device=None):
    import torch.distributed as dist
    if sync_device is not None:
        sync_device()
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
    if sync_device is not None:
        sync_device()


def max_over_ranks(value, device="cpu"):
    """MAX all-reduce of a python float (the slowest rank defines the step time)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    if dist.get_backend() == "gloo":
        device = "cpu"                          # (the two-processes-on-one-GPU tests: keep the scalar off the device)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_rows(local, total_rows, device=None):
    """All-gather row shards (possibly ragged) back into the full (total_rows, ...) tensor, in rank order."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return local
    world = dist.get_world_size()
    sizes = [shard_bounds(total_rows, world, r) for r in range(world)]
    max_rows = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((max_rows,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    return torch.cat([p[: hi - lo] for p, (lo, hi) in zip(parts, sizes)], dim=0)


# ------------------------------------------------------------------------------------------------------------
# The gradient all-reduce of the training step (reference: tf.distribute.MirroredStrategy under model.fit,
# src/main.py:156,192; loss / GLOBAL batch then SUM, main.py:198-200, losses.py:45).  Pure index arithmetic plus
# torch.distributed calls on whatever tensor carries the flat gradient buffer -- a view of the HIP library's device
# buffer in the Trainer (backend nccl = RCCL), a CPU tensor in the world-size-2 gloo test.
# ------------------------------------------------------------------------------------------------------------
def flat_layout(specs):
    """{local_name: (offset, numel)} and the total length of the flat gradient / Adam buffers: inventory order, every
    variable in a 16-byte aligned slot -- the layout csrc/w2v2_train.hip::get_state builds (`w2v2_grad_slot` reports it)."""
    out, off = {}, 0
    for name, (shape, _) in specs.items():
        n = 1
        for d in shape:
            n *= int(d)
        out[name] = (off, n)
        off += (n + 3) & ~3
    return out, off


def gradient_buckets(layout, total, num_layers):
    """[(offset, numel)] in the order the backward completes them: lm_head, encoder layers N-1 .. 0, then everything in
    front of layer 0 (include/w2v2.h: w2v2_train_bucket).  They tile [0, total)."""
    def first_with(prefix):
        hits = [o for n, (o, _) in layout.items() if n.startswith(prefix)]
        return min(hits) if hits else total

    def layer_begin(i):
        return first_with(f"encoder/layers/{i}/") if i < num_layers else first_with("lm_head/")

    out = [(first_with("lm_head/"), total - first_with("lm_head/"))]
    for i in range(num_layers - 1, -1, -1):
        out.append((layer_begin(i), layer_begin(i + 1) - layer_begin(i)))
    out.append((0, layer_begin(0)))
    return out


def trainable_ranges(layout, bucket, trainable):
    """The contiguous runs of TRAINABLE variables inside `bucket` = (offset, numel): frozen slots (the 4.2 M conv-stack
    elements of stage 2, everything but lm_head in stage 1) stay zero on every rank and are not sent.  `trainable` is a
    set of local names.  Adjacent trainable slots merge (alignment padding between them rides along)."""
    lo, n = bucket
    hi = lo + n
    runs = []
    for name, (off, numel) in layout.items():
        if off < lo or off >= hi or name not in trainable:
            continue
        end = min(hi, off + ((numel + 3) & ~3))
        if runs and runs[-1][1] == off:
            runs[-1][1] = end
        else:
            runs.append([off, end])
    return [(a, b - a) for a, b in runs]


def all_reduce_range(buf, off, n, payload_dtype=None, async_op=False):
    """SUM all-reduce of buf[off : off + n] in place.  `payload_dtype` (e.g. torch.bfloat16) sends a down-cast copy and
    writes the up-cast sum back -- half the bytes over xGMI for the bf16 fine-tune configurations (180.4 MB
    instead of 360.8 MB for base).  Returns a callable that completes the operation (waits, and for a compressed payload
    copies the result back).

    Accuracy of the compressed payload: the collective SUMs in the payload dtype, so with a bf16 payload every partial sum
    of the ring is rounded to bf16 again -- relative error per element ~ 2^-9 for the down-cast plus up to ~ log2(world)
    further roundings, i.e. ~1e-2 at 8 ranks (the two-rank bound is pinned in tests/test_dist_cpu.py).  That is the price
    of halving the bytes; the default payload is fp32 (exact to fp32 summation order).

    Transport: device buffers go straight to the backend (RCCL).  Under the `gloo` backend (the two-process-one-GPU test;
    RCCL refuses two ranks on one device) a device range is staged through host memory on the CURRENT stream -- the copy
    out is ordered behind whatever that stream waits for (the bucket event), the copy back is ordered before whatever
    waits on it, so the stream / event protocol of Trainer.all_reduce_gradients is exercised unchanged."""
    import torch
    import torch.distributed as dist
    view = buf[off:off + n]
    compress = payload_dtype is not None and payload_dtype != view.dtype
    if view.is_cuda and dist.get_backend() == "gloo":
        host = (view.to(payload_dtype) if compress else view).cpu()      # synchronous D2H on the current stream
        work = dist.all_reduce(host, op=dist.ReduceOp.SUM, async_op=async_op)
        stream = torch.cuda.current_stream()

        def finish_host():
            if async_op:
                work.wait()
            with torch.cuda.stream(stream):
                view.copy_(host.to(view.device, non_blocking=False))
        return finish_host
    if not compress:
        work = dist.all_reduce(view, op=dist.ReduceOp.SUM, async_op=async_op)
        return (work.wait if async_op else (lambda: None))
    small = view.to(payload_dtype)
    work = dist.all_reduce(small, op=dist.ReduceOp.SUM, async_op=async_op)

    def finish():
        if async_op:
            work.wait()
        view.copy_(small)
    return finish
