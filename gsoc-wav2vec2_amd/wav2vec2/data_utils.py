"""Input-pipeline contract of the reference (``src/data_utils.py:52-78``), host side, without tf.data:
normalise -> truncate to ``audio_maxlen`` -> right-pad audio with 0.0 and labels with ``pad_id``.

The reference builds this out of ``tf.data`` + TFRecords; only the *contract* matters to the forward /
CTC path (zero padding AFTER normalisation is what feeds layer-0 GroupNorm), so it is restated here as
plain numpy for callers that bring their own decoded audio.
"""

import numpy as np

from .processor import Wav2Vec2Processor


def batchify(speech_list, text_list=None, audio_maxlen=246000, labels_maxlen=256, audio_pad_id=0.0,
             labels_pad_id=0, tokenizer=None, do_normalize=True):
    """List of 1-D waveforms (+ optional transcripts) -> ``(speech (B, audio_maxlen) float32,
    labels (B, labels_maxlen) int32 or None)``.

    Per sample, as ``CommonDataLoader.batchify`` / ``_pad`` do: normalise the UNPADDED waveform
    (data_utils.py:62-64 via the processor), keep the first ``audio_maxlen`` samples (:65), right-pad
    with ``audio_pad_id`` (:66-71); labels are tokenised, truncated and right-padded the same way."""
    norm = Wav2Vec2Processor(is_tokenizer=False, do_normalize=do_normalize)
    speech = np.full((len(speech_list), audio_maxlen), audio_pad_id, dtype=np.float32)
    for i, s in enumerate(speech_list):
        s = np.asarray(s, dtype=np.float32).reshape(-1)
        s = np.asarray(norm(s), dtype=np.float32).reshape(-1)[:audio_maxlen]
        speech[i, : len(s)] = s
    labels = None
    if text_list is not None:
        if tokenizer is None:
            raise ValueError("`tokenizer` (Wav2Vec2Processor(is_tokenizer=True, ...)) is needed for transcripts")
        labels = np.full((len(text_list), labels_maxlen), labels_pad_id, dtype=np.int32)
        for i, t in enumerate(text_list):
            ids = np.asarray(tokenizer(t), dtype=np.int32)[:labels_maxlen]
            labels[i, : len(ids)] = ids
    return speech, labels


def attention_mask_for(speech_lengths, audio_maxlen):
    """(B, audio_maxlen) int32 mask of real samples -- what the robust / xlsr models take
    (tests/test_wav2vec2.py:58-62 builds it by hand)."""
    n = np.minimum(np.asarray(speech_lengths, dtype=np.int64), audio_maxlen)
    return (np.arange(audio_maxlen)[None, :] < n[:, None]).astype(np.int32)
