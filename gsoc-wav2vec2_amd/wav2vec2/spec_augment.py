"""Time-span masking for training (host side) -- mirrors ``src/wav2vec2/spec_augment.py:43-90``.

The reference draws its randomness with numpy at Python level (spec_augment.py:14,53) and applies the
mask with ``tf.where(mask, masked_spec_embed, features)`` (:119-127).  The sampling stays on the host
here as well; the device only receives the resulting (batch, frames) byte mask.
"""

import numpy as np


def compute_mask_indices(shape, mask_prob, mask_length, min_masks=2, rng=None):
    """(batch, frames) uint8 mask, 1 = frame replaced by ``masked_spec_embed``.

    Same recipe as the reference: ONE span count for the whole batch,
    ``max(int(p * T / len + U[0,1)), min_masks)`` capped at ``T // len``; per row, span starts are a
    uniform random subset (without replacement) of ``[0, T - len]`` -- the reference takes the top-k of
    ``1 - log(u)``, i.e. of ``-log(u)`` -- and spans may overlap."""
    rng = rng if rng is not None else np.random
    batch_size, seqlen = shape
    if mask_length > seqlen:
        raise ValueError(f"`mask_length` ({mask_length}) must be smaller than `seq_length` ({seqlen}).")
    num_mask_spans = int(mask_prob * (seqlen / mask_length) + float(np.asarray(rng.uniform(0, 1, 1))[0]))
    num_mask_spans = max(num_mask_spans, min_masks)
    if num_mask_spans * mask_length > seqlen:
        num_mask_spans = seqlen // mask_length
    u = np.asarray(rng.uniform(0, 1, (batch_size, seqlen - (mask_length - 1))))
    z = -np.log(u)
    starts = np.argsort(-z, axis=-1, kind="stable")[:, :num_mask_spans]          # top-k of -log(u)
    mask = np.zeros((batch_size, seqlen), dtype=np.uint8)
    rows = np.arange(batch_size)[:, None, None]
    cols = starts[:, :, None] + np.arange(mask_length)[None, None, :]
    mask[rows, cols] = 1
    return mask
