"""CTCLoss -- host-side mirror of the reference's ``src/wav2vec2/losses.py:4-56``.

Same constructor ``CTCLoss(config, model_input_shape, division_factor=1)`` and
call ``loss(labels, logits) -> scalar``; the reference's conventions are kept:
  * every row's ``logit_length`` is the FULL frame count derived from the static
    ``model_input_shape`` (not from real audio length) -- losses.py:29-30,47-56;
  * ``label_length`` = number of labels != ``pad_id`` -- losses.py:32-33;
  * blank index = ``pad_id``; per-sample NLL / ``division_factor``, then
    Keras ``Reduction.SUM`` -- losses.py:6,45.
The alpha/beta recursions run in the HIP library (w2v2_ctc_loss).
"""

import numpy as np

from . import _native as N


class CTCLoss:
    def __init__(self, config, model_input_shape, division_factor=1):
        self.kernal_sizes = config.kernal_sizes
        self.strides = config.strides
        self.pad_id = config.pad_id
        self.division_factor = division_factor
        self.model_input_shape = model_input_shape

    def _get_logit_length(self, input_length):
        """Frames at the end of the conv stack (losses.py:47-56)."""
        for kernal_size, stride in zip(self.kernal_sizes, self.strides):
            input_length = 1 + (input_length - kernal_size) // stride
        return input_length

    def per_sample(self, labels, logits, with_grad=False, with_total=False):
        """Per-sample NLL (B,) [and d sum(nll) / d logits] [and the scalar the reference's `call` returns: sum_b nll_b / division_factor];
        torch CUDA tensors."""
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("CTCLoss (MI355X build) needs a HIP device; there is no CPU fallback")
        dev = torch.device("cuda", torch.cuda.current_device())
        if not isinstance(logits, torch.Tensor):
            logits = torch.as_tensor(np.asarray(logits, dtype=np.float32))
        logits = logits.detach().as_subclass(torch.Tensor).to(device=dev, dtype=torch.float32).contiguous()
        if not isinstance(labels, torch.Tensor) or not labels.is_cuda:
            # host labels: a label outside the vocabulary is a vocab / config mismatch -- say so before the launch.
            # (Device-resident labels are checked by the kernel instead: the sample's loss comes back NaN.)
            host = np.asarray(labels.cpu() if isinstance(labels, torch.Tensor) else labels)
            if host.size and (host.min() < 0 or host.max() >= logits.shape[-1]):
                raise ValueError(f"labels must lie in [0, {logits.shape[-1]}): got [{host.min()}, {host.max()}]")
            labels = torch.as_tensor(host)
        labels = labels.to(device=dev, dtype=torch.int32).contiguous()
        B, T, V = logits.shape
        if labels.dim() != 2 or labels.shape[0] != B:
            raise ValueError("labels must be (batch, max_label_len)")
        U = labels.shape[1]
        logit_len = int(self._get_logit_length(int(self.model_input_shape[1])))
        if logit_len > T:
            raise ValueError(f"model_input_shape implies {logit_len} frames but logits have {T}")
        nll = torch.empty((B,), device=dev, dtype=torch.float32)
        grad = torch.empty_like(logits) if with_grad else None
        total = torch.empty((1,), device=dev, dtype=torch.float32)
        lib = N.load()
        # label lengths (count of labels != pad_id), the uniform logit length, the division of loss and gradient by division_factor and
        # the SUM reduction are all evaluated by the native call: no framework kernel runs between the forward and the backward
        N.check(lib.w2v2_ctc_loss_fused(N.ptr(logits), B, T, V, N.ptr(labels), U, logit_len, self.pad_id, float(self.division_factor),
                                        N.ptr(nll), N.ptr(grad), N.ptr(total), N.current_stream()), "w2v2_ctc_loss_fused")
        self.last_total = total[0]            # sum_b nll_b / division_factor, a 0-d device tensor view (kept for callers that read the
                                              # attribute; `total()` and `__call__` do not depend on it: two interleaved calls stay correct)
        if with_total:
            return (nll, grad, total[0]) if with_grad else (nll, total[0])
        if with_grad:
            return nll, grad                  # grad already / division_factor
        return nll

    def __call__(self, labels, hidden_states):
        return self.per_sample(labels, hidden_states, with_total=True)[1]

    call = __call__
