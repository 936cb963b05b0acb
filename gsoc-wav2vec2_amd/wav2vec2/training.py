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

import numpy as np

from . import _native as N
from .spec_augment import compute_mask_indices


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
    def __init__(self, model, loss, learning_rate=1e-4, beta_1=0.9, beta_2=0.999, epsilon=1e-7, seed=0,
                 dropout=None, apply_spec_augment=None, overlap_all_reduce=True):
        if not getattr(model, "_with_lm_head", False):
            raise ValueError("Trainer needs a Wav2Vec2ForCTC model")
        self.model, self.loss = model, loss
        self.learning_rate, self.beta_1, self.beta_2, self.epsilon = learning_rate, beta_1, beta_2, epsilon
        cfg = model.config
        self.dropout = cfg.dropout if dropout is None else dropout
        self.apply_spec_augment = cfg.apply_spec_augment if apply_spec_augment is None else apply_spec_augment
        self.seed = int(seed)
        self.iterations = 0                               # Keras optimizer.iterations
        self._rng = np.random.RandomState(seed)           # host RNG: spec-augment spans, stochastic depth
        self.last = {}
        self.overlap_all_reduce = bool(overlap_all_reduce)   # per-bucket all-reduces under the backward (all_reduce_gradients)
        self._comm_stream = None

    # -- pieces (also used by the parity tests) ----------------------------------------------------
    def forward(self, batch, attention_mask=None, spec_mask=None, sd_keep=None, step_seed=None):
        import torch
        m = self.model
        batch, attention_mask = m._prepare(batch, attention_mask)
        B, L = batch.shape
        T = m.num_frames(L)
        m._finalize()
        cfg = m.config
        if spec_mask is None and self.apply_spec_augment:
            spec_mask = compute_mask_indices((B, T), cfg.mask_time_prob, cfg.mask_time_length, min_masks=2, rng=self._rng)
        if sd_keep is None and cfg.survival_prob < 1.0:
            # one Bernoulli scalar per StochasticDepth call (tensorflow_addons.py:381)
            sd_keep = (self._rng.uniform(size=cfg.num_layers) < cfg.survival_prob).astype(np.float32)
        sm = None if spec_mask is None else np.ascontiguousarray(spec_mask, dtype=np.uint8).reshape(-1)
        sd = None if sd_keep is None else np.ascontiguousarray(sd_keep, dtype=np.float32)
        seed = self.seed * 1000003 + self.iterations if step_seed is None else int(step_seed)
        logits = torch.empty((B, T, cfg.vocab_size), device=batch.device, dtype=torch.float32)
        N.check(m._lib.w2v2_train_forward(m._handle, N.ptr(batch), B, L, N.ptr(attention_mask), N.ptr(sm), N.ptr(sd),
                                          float(self.dropout), C.c_uint64(seed & 0xFFFFFFFFFFFFFFFF), N.ptr(logits),
                                          N.current_stream()), "w2v2_train_forward")
        self.last = dict(spec_mask=spec_mask, sd_keep=sd_keep, seed=seed & 0xFFFFFFFFFFFFFFFF)
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

    def all_reduce_gradients(self, force=False):
        """SUM over data-parallel ranks: the loss is pre-divided by the global batch (losses.py:45,
        main.py:198-200), so the sum is the global mean.

        Called right after `backward` -- which only ENQUEUES the backward kernels -- this issues one RCCL all-reduce per
        gradient bucket on a communication stream that waits for that bucket alone (an event the backward records when
        the bucket's slice is final), so the collectives of the upper layers run under the backward of the lower ones;
        the calling stream then waits for all of them.  A layer of wav2vec2-base is a 28 MB bucket (large: 50 MB): big
        enough for the ring to run at link speed over xGMI, small enough that only the last one is exposed.
        `overlap_all_reduce=False` (constructor) falls back to a single all-reduce of the whole buffer."""
        import torch
        import torch.distributed as dist
        active = dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or force)
        if not active:
            return
        buf = self.grad_buffer()
        if not self.overlap_all_reduce:
            dist.all_reduce(buf, op=dist.ReduceOp.SUM)
            return
        m = self.model
        if self._comm_stream is None:
            self._comm_stream = torch.cuda.Stream()
        cs = self._comm_stream
        works = []
        for k, (off, n) in enumerate(self.gradient_buckets()):
            if n == 0:
                continue
            N.check(m._lib.w2v2_train_bucket_wait(m._handle, k, C.c_void_p(cs.cuda_stream)), "w2v2_train_bucket_wait")
            with torch.cuda.stream(cs):
                works.append(dist.all_reduce(buf[off:off + n], op=dist.ReduceOp.SUM, async_op=True))
        for w in works:
            w.wait()              # the current stream (optimizer step next) waits for the collectives

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
        state = dict(iterations=int(self.iterations), learning_rate=float(self.learning_rate), beta_1=float(self.beta_1),
                     beta_2=float(self.beta_2), epsilon=float(self.epsilon), seed=int(self.seed), rng=self._rng.get_state())
        if self.iterations > 0:
            am, av = self._adam_views()
            state["adam_m"], state["adam_v"] = am.cpu().numpy(), av.cpu().numpy()
        return state

    def load_state_dict(self, state, batch_shape=None):
        """Inverse of `state_dict`.  The moment buffers live in the model's training state, which is sized at the first
        training forward: pass `batch_shape=(B, L)` (any shape; it only triggers the allocation) when loading into a
        trainer that has not run a step yet."""
        import torch
        views = None
        if "adam_m" in state:
            try:
                views = self._adam_views()
            except Exception:
                if batch_shape is None:
                    raise
                B, L = batch_shape                       # allocate the training state (before the host RNG is restored:
                self.forward(torch.zeros((B, L), device="cuda"), step_seed=0)      # this forward may draw spec-augment spans)
                views = self._adam_views()
            if views[0].numel() != state["adam_m"].size:
                raise ValueError("optimizer state belongs to a different model (moment buffer size mismatch)")
        self.iterations = int(state["iterations"])
        for k in ("learning_rate", "beta_1", "beta_2", "epsilon"):
            setattr(self, k, float(state[k]))
        self.seed = int(state["seed"])
        self._rng.set_state(state["rng"])
        if views is not None:
            for dst, key in zip(views, ("adam_m", "adam_v")):
                dst.copy_(torch.from_numpy(np.ascontiguousarray(state[key], np.float32)).to(dst.device))

    # -- the step ------------------------------------------------------------------------------------
    def step(self, batch, labels, attention_mask=None):
        logits = self.forward(batch, attention_mask)
        nll, grad = self.loss.per_sample(labels, logits, with_grad=True)     # grad already / division_factor
        self.backward(grad)
        self.all_reduce_gradients()
        self.apply_gradients()
        return (nll / self.loss.division_factor).sum()
