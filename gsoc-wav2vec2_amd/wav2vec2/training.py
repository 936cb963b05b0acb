"""The training step -- the build's counterpart of what Keras' ``model.fit`` does per batch in the
reference's ``src/main.py:136-259``: training-mode forward (dropout, spec-augment,
stochastic depth), CTC loss divided by the GLOBAL batch, backward of the trainable variables, a SUM
all-reduce of the gradients across data-parallel ranks (RCCL), Adam with Keras defaults.

    model.freeze_feature_extractor()                       # stage 2 of the reference (main.py:234-237)
    trainer = Trainer(model, CTCLoss(config, input_shape, division_factor=global_batch), learning_rate=1e-4)
    loss = trainer.step(batch, labels)

Everything numeric runs in the HIP library (w2v2_train_forward / w2v2_ctc_loss / w2v2_train_backward /
w2v2_adam_step); this class only sequences the calls, draws the host-side randomness the reference also
draws on the host, and issues the collective.
"""

import ctypes as C
import warnings

import numpy as np

from . import _native as N
from . import dist as D


def _world_rank():
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_world_size(), dist.get_rank()
    except ImportError:  # pragma: no cover
        pass
    return 1, 0


class _DeviceBuffer:
    """Zero-copy view of library-owned device memory for torch (``__cuda_array_interface__``)."""

    def __init__(self, ptr, numel):
        self.__cuda_array_interface__ = {"shape": (int(numel),), "typestr": "<f4", "data": (int(ptr), False), "version": 2}


def stage2_learning_rate(epoch, lr1=1e-4, lr2=5e-5, transition_epochs=10):
    """The reference's stage-2 schedule (training_utils.py:24-31 with main.py:56-58 defaults): `lr1` while
    `epoch <= transition_epochs` (Keras passes 0-based epochs), `lr2` afterwards.  Assign the result to
    `Trainer.learning_rate` at the start of each epoch -- what `LearningRateScheduler` does to the optimizer."""
    return lr1 if epoch <= transition_epochs else lr2


class Trainer:
    """One optimizer over one model.  Like `tf.keras.optimizers.Adam(...)` + `model.compile` in the reference, creating a
    Trainer starts a FRESH optimizer: iteration 0 and zero moments (src/main.py:213,240 build a new Adam for each of the
    two stages) -- pass `reset_optimizer=False` to adopt the moments already in the model (resume via `load_state_dict`).

    Randomness is per data-parallel rank, as under MirroredStrategy where every replica draws its own dropout and
    spec-augment masks: the effective seed is `seed * world_size + rank` (kept in `state_dict`, so resume stays exact).

    `allreduce_dtype`: "fp32" (default) or "bf16" -- the gradient payload of the data-parallel all-reduce."""

    def __init__(self, model, loss, learning_rate=1e-4, beta_1=0.9, beta_2=0.999, epsilon=1e-7, seed=0,
                 dropout=None, apply_spec_augment=None, overlap_all_reduce=True, reset_optimizer=True, allreduce_dtype="fp32",
                 collective="torch"):
        if not getattr(model, "_with_lm_head", False):
            raise ValueError("Trainer needs a Wav2Vec2ForCTC model")
        if allreduce_dtype not in ("fp32", "bf16"):
            raise ValueError("allreduce_dtype must be 'fp32' or 'bf16'")
        if collective not in ("torch", "native", "native-rs"):
            raise ValueError("collective must be 'torch', 'native' or 'native-rs'")
        if collective != "torch" and allreduce_dtype != "fp32":
            raise ValueError("the native collective reduces the fp32 gradient buffer in place: allreduce_dtype must be 'fp32'")
        self.collective = collective      # engine of the gradient SUM: torch.distributed, or the library's own RCCL communicator
                                          # (include/w2v2.h w2v2_allreduce_bucket; "native-rs" = reduce-scatter + all-gather)
        self.model, self.loss = model, loss
        self.learning_rate, self.beta_1, self.beta_2, self.epsilon = learning_rate, beta_1, beta_2, epsilon
        cfg = model.config
        self.dropout = cfg.dropout if dropout is None else dropout
        self.apply_spec_augment = cfg.apply_spec_augment if apply_spec_augment is None else apply_spec_augment
        world, rank = _world_rank()
        self._world_rank_seen = (world, rank)             # what the seed below was derived from (checked again in `step`)
        self.base_seed = int(seed)
        self.seed = int(seed) * world + rank              # rank-aware: replicas draw independent masks
        self.iterations = 0                               # Keras optimizer.iterations
        self._rng = np.random.RandomState(self.seed & 0xFFFFFFFF)    # host RNG: spec-augment spans, stochastic depth
        self._ranges_cache = {}                           # trainable set -> per-bucket send ranges (reduce_ranges)
        self._layout_checked = False
        self.last = {}
        self.overlap_all_reduce = bool(overlap_all_reduce)   # per-bucket all-reduces under the backward (all_reduce_gradients)
        self.allreduce_dtype = allreduce_dtype
        self._comm_stream = None
        if reset_optimizer:
            N.check(model._lib.w2v2_adam_reset(model._handle, N.current_stream()), "w2v2_adam_reset")

    # -- pieces (also used by the parity tests) ----------------------------------------------------
    def forward(self, batch, attention_mask=None, spec_mask=None, sd_keep=None, step_seed=None):
        """Training-mode forward = `model(batch, attention_mask, training=True)` with this trainer's dropout rate, host RNG
        and step-derived seed (explicit `spec_mask` / `sd_keep` / `step_seed` override the draws)."""
        seed = self.seed * 1000003 + self.iterations if step_seed is None else int(step_seed)
        logits, self.last = self.model._train_forward(batch, attention_mask, dropout=self.dropout,
                                                      apply_spec_augment=self.apply_spec_augment, spec_mask=spec_mask,
                                                      sd_keep=sd_keep, step_seed=seed, rng=self._rng)
        return logits

    def backward(self, grad_logits):
        m = self.model
        grad_logits = grad_logits.contiguous()
        N.check(m._lib.w2v2_train_backward(m._handle, N.ptr(grad_logits), N.current_stream()), "w2v2_train_backward")

    def gradient(self, local_name):
        m = self.model
        shape = m._specs[local_name][0]
        out = np.empty(shape, dtype=np.float32)
        N.check(m._lib.w2v2_get_grad(m._handle, local_name.encode(), N.ptr(out), out.size, N.current_stream()), "w2v2_get_grad")
        return out

    def grad_buffer(self):
        """The flat fp32 gradient buffer (all variables, inventory order) as a torch CUDA tensor view."""
        import torch
        m = self.model
        ptr, n = C.c_void_p(), C.c_int64()
        N.check(m._lib.w2v2_grad_buffer(m._handle, C.byref(ptr), C.byref(n)), "w2v2_grad_buffer")
        return torch.as_tensor(_DeviceBuffer(ptr.value, n.value), device=torch.device("cuda", torch.cuda.current_device()))

    def gradient_buckets(self):
        """[(offset, numel)] slices of the flat gradient buffer in the order the backward completes them
        (lm_head, layers N-1 .. 0, the front of the model); they tile the buffer."""
        m = self.model
        out = []
        for k in range(m._lib.w2v2_train_num_buckets(m._handle)):
            off, n = C.c_int64(), C.c_int64()
            N.check(m._lib.w2v2_train_bucket(m._handle, k, C.byref(off), C.byref(n)), "w2v2_train_bucket")
            out.append((off.value, n.value))
        return out

    def gradient_slot(self, local_name):
        """(offset, numel) of a variable's slot in the flat gradient buffer."""
        off, n = C.c_int64(), C.c_int64()
        N.check(self.model._lib.w2v2_grad_slot(self.model._handle, local_name.encode(), C.byref(off), C.byref(n)), "w2v2_grad_slot")
        return off.value, n.value

    def reduce_ranges(self):
        """Per gradient bucket (completion order) the [(offset, numel)] runs of TRAINABLE variables that are sent: frozen slots
        (the 4.2 M conv-stack elements in stage 2; everything but lm_head in stage 1) are zero on every rank and stay home.
        For wav2vec2-base in stage 2 that is the reference's 90,195,104-element payload (+ masked_spec_embed), 360.8 MB fp32."""
        m = self.model
        trainable = frozenset(v.local_name for v in m.trainable_variables)
        hit = self._ranges_cache.get(trainable)
        if hit is not None:
            return hit
        layout, total = D.flat_layout(m._specs)
        if not self._layout_checked:
            # the host-side layout arithmetic must BE the native flat buffer's layout: a drift would leave trainable slots
            # unreduced or send frozen ones (checked once, against every slot the library reports)
            for name, (off, n) in layout.items():
                if self.gradient_slot(name) != (off, n):
                    raise RuntimeError(f"gradient slot of `{name}`: host layout {(off, n)} != native {self.gradient_slot(name)}")
            if total != self.grad_buffer().numel():
                raise RuntimeError(f"flat gradient buffer: host layout {total} elements != native {self.grad_buffer().numel()}")
            self._layout_checked = True
        ranges = [D.trainable_ranges(layout, bk, trainable) for bk in self.gradient_buckets()]
        self._ranges_cache = {trainable: ranges}          # (one entry: the trainable set changes between stages, not steps)
        return ranges

    def all_reduce_gradients(self, force=False):
        """SUM over data-parallel ranks: the loss is pre-divided by the global batch (losses.py:45,
        main.py:198-200), so the sum is the global mean.

        Called right after `backward` -- which only ENQUEUES the backward kernels -- this issues the RCCL all-reduces of one
        gradient bucket at a time on a communication stream that waits for that bucket alone (an event the backward records
        when the bucket's slice is final), so the collectives of the upper layers run under the backward of the lower ones;
        the calling stream then waits for all of them.  A layer of wav2vec2-base is a 28 MB bucket (large: 50 MB): big
        enough for the ring to run at link speed over xGMI, small enough that only the last one is exposed.  Only trainable
        slots travel (`reduce_ranges`); `allreduce_dtype="bf16"` halves the payload.
        `overlap_all_reduce=False` (constructor) issues the same ranges after the whole backward on the calling stream."""
        if self.collective != "torch":
            return self._all_reduce_native(force)
        import torch
        import torch.distributed as dist
        active = dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or force)
        if not active:
            return
        if _world_rank() != self._world_rank_seen:
            warnings.warn(f"torch.distributed reports (world, rank) = {_world_rank()} but this Trainer derived its per-replica seed "
                          f"from {self._world_rank_seen}: build the Trainer after init_process_group, or replicas share dropout / "
                          "spec-augment masks", RuntimeWarning, stacklevel=2)
            self._world_rank_seen = _world_rank()
        buf = self.grad_buffer()
        payload = torch.bfloat16 if self.allreduce_dtype == "bf16" else None
        ranges = self.reduce_ranges()
        if not self.overlap_all_reduce:
            for runs in ranges:
                for off, n in runs:
                    D.all_reduce_range(buf, off, n, payload)()
            return
        m = self.model
        if self._comm_stream is None:
            self._comm_stream = torch.cuda.Stream()
        cs = self._comm_stream
        finishers = []
        for k, runs in enumerate(ranges):
            if not runs:
                continue
            N.check(m._lib.w2v2_train_bucket_wait(m._handle, k, C.c_void_p(cs.cuda_stream)), "w2v2_train_bucket_wait")
            with torch.cuda.stream(cs):
                for off, n in runs:
                    finishers.append(D.all_reduce_range(buf, off, n, payload, async_op=True))
        with torch.cuda.stream(cs):
            for f in finishers:
                f()               # (a compressed payload is copied back on the communication stream)
        torch.cuda.current_stream().wait_stream(cs)      # the optimizer step next waits for the collectives

    def _all_reduce_native(self, force=False):
        """The same protocol issued by the library (csrc/comm.hip): per bucket, its communication stream waits for that bucket's
        event and SUMs the bucket's trainable runs in place over RCCL; the calling stream then waits for the communication stream.
        The communicator is created on first use (wav2vec2/dist.py::native_comm_init)."""
        m = self.model
        _, world, _ = D.native_comm_info(m)
        if world == 0:
            _, world = D.native_comm_init(m)
        if world <= 1 and not force:
            return
        algo = 1 if self.collective == "native-rs" else 0
        nb = m._lib.w2v2_train_num_buckets(m._handle)
        for k in range(nb):
            N.check(m._lib.w2v2_allreduce_bucket(m._handle, k, algo), "w2v2_allreduce_bucket")
        sent = C.c_int64()
        N.check(m._lib.w2v2_allreduce_finish(m._handle, N.current_stream(), C.byref(sent)), "w2v2_allreduce_finish")
        self._native_bytes_last = sent.value

    def native_reduce_ranges(self):
        """Per bucket the (offset, numel) runs the NATIVE collective sends (w2v2_allreduce_run): must equal `reduce_ranges()`."""
        m = self.model
        out = []
        for k in range(m._lib.w2v2_train_num_buckets(m._handle)):
            cnt = C.c_int32()
            N.check(m._lib.w2v2_allreduce_num_runs(m._handle, k, C.byref(cnt)), "w2v2_allreduce_num_runs")
            runs = []
            for i in range(cnt.value):
                off, n = C.c_int64(), C.c_int64()
                N.check(m._lib.w2v2_allreduce_run(m._handle, k, i, C.byref(off), C.byref(n)), "w2v2_allreduce_run")
                runs.append((off.value, n.value))
            out.append(runs)
        return out

    def apply_gradients(self):
        m = self.model
        self.iterations += 1
        N.check(m._lib.w2v2_adam_step(m._handle, float(self.learning_rate), float(self.beta_1), float(self.beta_2),
                                      float(self.epsilon), self.iterations, N.current_stream()), "w2v2_adam_step")
        m._dirty = False          # w2v2_adam_step re-derives the packed / normalised tensors itself

    # -- checkpoint / resume ----------------------------------------------------------------------------
    def _adam_views(self):
        import torch
        m = self.model
        pm, pv, n = C.c_void_p(), C.c_void_p(), C.c_int64()
        N.check(m._lib.w2v2_adam_buffers(m._handle, C.byref(pm), C.byref(pv), C.byref(n)), "w2v2_adam_buffers")
        dev = torch.device("cuda", torch.cuda.current_device())
        return (torch.as_tensor(_DeviceBuffer(pm.value, n.value), device=dev),
                torch.as_tensor(_DeviceBuffer(pv.value, n.value), device=dev))

    def state_dict(self):
        """Everything a resumed run needs besides the model's variables (`model.save_weights` / `save_pretrained`): the
        optimizer step count, Adam's moments, the hyper-parameters and the host RNG that draws spec-augment spans and
        stochastic-depth decisions.  The reference's per-epoch ModelCheckpoint (training_utils.py:38-45) is the analogue."""
        am, av = self._adam_views()
        world, rank = _world_rank()
        return dict(iterations=int(self.iterations), learning_rate=float(self.learning_rate), beta_1=float(self.beta_1),
                    beta_2=float(self.beta_2), epsilon=float(self.epsilon), seed=int(self.seed), base_seed=int(self.base_seed),
                    world_size=world, rank=rank, rng=self._rng.get_state(), adam_m=am.cpu().numpy(), adam_v=av.cpu().numpy())

    def load_state_dict(self, state, batch_shape=None):
        """Inverse of `state_dict`.  A state without moments (e.g. one written before any step) zeroes them, so stale
        moments of an earlier optimizer never leak into the resumed run.  (`batch_shape` is accepted for compatibility:
        the moment buffers no longer depend on a batch shape.)"""
        import torch
        views = self._adam_views()
        if "adam_m" in state:
            if views[0].numel() != state["adam_m"].size:
                raise ValueError("optimizer state belongs to a different model (moment buffer size mismatch)")
            for dst, key in zip(views, ("adam_m", "adam_v")):
                dst.copy_(torch.from_numpy(np.ascontiguousarray(state[key], np.float32)).to(dst.device))
        else:
            N.check(self.model._lib.w2v2_adam_reset(self.model._handle, N.current_stream()), "w2v2_adam_reset")
        self.iterations = int(state["iterations"])
        for k in ("learning_rate", "beta_1", "beta_2", "epsilon"):
            setattr(self, k, float(state[k]))
        # Randomness stays per REPLICA: the seed is re-derived for the rank that loads (a rank-0 checkpoint loaded on every
        # rank must not give all replicas rank 0's dropout / spec-augment / stochastic-depth masks).  The saved host RNG
        # state is adopted only by the replica that wrote it (exact resume); any other replica re-seeds from its own seed
        # and the step count.
        world, rank = _world_rank()
        self._world_rank_seen = (world, rank)
        saved_world, saved_rank = int(state.get("world_size", 1)), int(state.get("rank", 0))
        self.base_seed = int(state.get("base_seed", state["seed"]))
        if "base_seed" in state:
            self.seed = self.base_seed * world + rank
        else:                                             # a state from before base_seed existed: single-replica semantics
            self.seed = int(state["seed"])
        if (saved_world, saved_rank) == (world, rank) and self.seed == int(state["seed"]):
            self._rng.set_state(state["rng"])
        else:
            self._rng = np.random.RandomState((self.seed * 1000003 + self.iterations) & 0xFFFFFFFF)

    # -- the step ------------------------------------------------------------------------------------
    def step(self, batch, labels, attention_mask=None, all_reduce=True):
        """One fine-tune step.  `all_reduce=False` skips the data-parallel gradient collective (each replica then applies its
        own gradients): a MEASUREMENT switch -- bench.py times the step with and without it to report the exposed
        communication time -- never the training semantics of the reference (main.py:156,192)."""
        logits = self.forward(batch, attention_mask)
        _, grad, total = self.loss.per_sample(labels, logits, with_grad=True, with_total=True)    # grad already / division_factor;
                                                                             # total = sum_b nll_b / division_factor, summed on the device
        self.backward(grad)
        if all_reduce:
            self.all_reduce_gradients()
        self.apply_gradients()
        return total

    def activation_storage(self):
        """(set of per-layer activations the last training forward kept ONLY as bf16 -- "qkv", "ctx", "ffn", "u", and for prenorm models "ln" --, bytes of
        shape-dependent training workspace allocated).  In precision "bf16" with shadows all four are bf16-only and their fp32 buffers do
        not exist; anything else after an optimizer step means the step fell back to the fp32 activations (include/w2v2.h: w2v2_train_storage)."""
        mask, nbytes = C.c_int32(), C.c_int64()
        N.check(self.model._lib.w2v2_train_storage(self.model._handle, C.byref(mask), C.byref(nbytes)), "w2v2_train_storage")
        names = {n for bit, n in ((1, "qkv"), (2, "ctx"), (4, "ffn"), (8, "u"), (16, "ln")) if mask.value & bit}
        return names, nbytes.value

    def all_reduce_payload(self):
        """(bytes sent per step and replica, number of non-empty buckets, number of collectives) of `all_reduce_gradients`."""
        ranges = self.reduce_ranges()
        elems = sum(n for runs in ranges for _, n in runs)
        return (elems * (2 if self.allreduce_dtype == "bf16" else 4), sum(1 for runs in ranges if runs),
                sum(len(runs) for runs in ranges))
