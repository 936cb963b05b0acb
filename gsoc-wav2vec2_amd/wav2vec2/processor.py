"""Wav2Vec2Processor -- host pre/post-processing (reference ``src/wav2vec2/processor.py``).

Two roles behind one constructor, as in the reference (processor.py:9-35):

``is_tokenizer=False``  waveform normaliser: zero mean / unit variance along the last axis with the
                        population variance and eps 1e-5, applied BEFORE padding (processor.py:101-106).
``is_tokenizer=True``   character tokenizer for CTC labels and the greedy CTC collapse decode
                        (processor.py:52-94): text is upper-cased, ``-`` counts as a space, only
                        ``A-Z``, ``'`` and space survive, a space is written as the word delimiter ``|``;
                        decoding merges repeats, drops ``<pad>`` (the CTC blank) and turns ``|`` back into
                        a space.

Kept from the reference because callers depend on it: the constructor signature, ``__call__``,
``decode(ids, skip_special_tokens=True, group_tokens=True)`` and the ``vocab.json`` format
(token -> id).  Everything else is this build's own: the tokenizer is a single pass over a 256-entry
character table, the decoder a single pass over the ids.  Pure host code; nothing here touches the HIP
library.
"""

import json
import os

import numpy as np

PAD_TOKEN, UNK_TOKEN, WORD_DELIMITER = "<pad>", "<unk>", "|"


def _character_table():
    """chr -> token (or None = dropped) for the tokenizer's alphabet (processor.py:91-94)."""
    table = {}
    for code in range(ord("A"), ord("Z") + 1):
        table[chr(code)] = chr(code)
        table[chr(code).lower()] = chr(code)
    table["'"] = "'"
    table[" "] = WORD_DELIMITER
    table["-"] = WORD_DELIMITER           # a hyphen separates words
    return table


class Wav2Vec2Processor:
    def __init__(self, is_tokenizer, do_normalize=True, vocab_path="./vocab.json"):
        self.is_tokenizer = is_tokenizer
        self.do_normalize = do_normalize
        self.vocab_path = vocab_path
        if is_tokenizer:
            self._load_vocab()

    # ---- vocabulary ------------------------------------------------------------------------------
    def _load_vocab(self):
        # the reference fetches vocab.json over HTTP when the file is absent (processor.py:37-50); no network here
        if not os.path.isfile(self.vocab_path):
            raise ValueError(f"Couldn't find `vocab.json` at {self.vocab_path} (and there is no network to fetch it)")
        vocab = self.get_vocab()
        for needed in (PAD_TOKEN, UNK_TOKEN, WORD_DELIMITER):
            if needed not in vocab:
                raise ValueError(f"`{self.vocab_path}` has no `{needed}` entry")
        self._vocab = vocab
        self._unk = vocab[UNK_TOKEN]
        self._blank = vocab[PAD_TOKEN]
        # text side: one lookup per character straight to an id
        chars = _character_table()
        self._char_to_id = {ch: vocab.get(tok, self._unk) for ch, tok in chars.items()}
        # id side: id -> output text, with the delimiter already mapped to a space
        self._id_to_text = {i: (" " if tok == WORD_DELIMITER else tok) for tok, i in vocab.items()}

    def get_vocab(self):
        """token -> id, as stored in ``vocab.json``."""
        with open(self.vocab_path, "r") as f:
            return json.load(f)

    # ---- call ------------------------------------------------------------------------------------
    def __call__(self, input_values):
        """tokenizer: text -> list of label ids;  otherwise: waveform -> normalised waveform."""
        if self.is_tokenizer:
            return self._encode(input_values)
        return self._normalize(input_values) if self.do_normalize else input_values

    def _encode(self, text):
        lookup = self._char_to_id
        out = []
        for ch in text:
            if ch in lookup:
                out.append(lookup[ch])
            elif not ch.isascii():
                # str.upper() of a few non-ASCII letters lands in A-Z (e.g. the sharp s becomes "SS", the dotless i becomes "I")
                # and the reference upper-cases before filtering: follow it
                out.extend(lookup[u] for u in ch.upper() if "A" <= u <= "Z")
        return out

    def _tokenize(self, text):
        """The token strings behind ``__call__`` (for callers that want characters rather than ids)."""
        by_id = {i: tok for tok, i in self._vocab.items()}
        return [by_id[i] for i in self._encode(text)]

    # ---- decode ----------------------------------------------------------------------------------
    def decode(self, input_ids, skip_special_tokens=True, group_tokens=True):
        """Greedy CTC decode of a frame-level id sequence: merge runs of equal ids (``group_tokens``), drop the blank
        ``<pad>`` (``skip_special_tokens``), map ids to characters (unknown ids print as ``<unk>``), ``|`` -> space,
        strip the ends."""
        pieces = []
        previous = None
        for raw in input_ids:
            i = int(raw)
            repeated = group_tokens and i == previous
            previous = i
            if repeated or (skip_special_tokens and i == self._blank):
                continue
            pieces.append(self._id_to_text.get(i, UNK_TOKEN))
        return "".join(pieces).strip()

    # ---- normaliser ------------------------------------------------------------------------------
    def _normalize(self, x):
        """(x - mean) / sqrt(var + 1e-5) along the last axis, population variance, then squeeze.  Call before padding."""
        try:
            import torch
            if isinstance(x, torch.Tensor):
                mean = x.mean(dim=-1, keepdim=True)
                var = x.var(dim=-1, keepdim=True, unbiased=False)
                return ((x - mean) / torch.sqrt(var + 1e-5)).squeeze()
        except ImportError:  # pragma: no cover
            pass
        x = np.asarray(x, dtype=np.float32)
        mean = x.mean(axis=-1, keepdims=True, dtype=np.float64)
        var = x.var(axis=-1, keepdims=True, dtype=np.float64)
        return np.squeeze(((x - mean) / np.sqrt(var + 1e-5)).astype(np.float32))
