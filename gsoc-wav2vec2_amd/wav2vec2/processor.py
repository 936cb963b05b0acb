"""Wav2Vec2Processor -- host pre/post-processing (reference ``processor.py``).

``is_tokenizer=False``: zero-mean / unit-variance normalisation of the
waveform, to be applied BEFORE padding (processor.py:101-106).
``is_tokenizer=True``: the character tokenizer and the greedy CTC collapse
decode (processor.py:52-94).  Pure host code: numpy for arrays, torch ops for
torch tensors; nothing here touches the HIP library.
"""

import json
import os
import re
from itertools import groupby

import numpy as np


class Wav2Vec2Processor:
    def __init__(self, is_tokenizer, do_normalize=True, vocab_path="./vocab.json"):
        self.is_tokenizer = is_tokenizer
        self.do_normalize = do_normalize
        self.vocab_path = vocab_path

        if self.is_tokenizer:
            self._setup_vocab()
            self.token_to_id_mapping = self.get_vocab()
            self.id_to_token_mapping = {v: k for k, v in self.token_to_id_mapping.items()}
            self.unk_token = "<unk>"
            self.unk_id = self.token_to_id_mapping[self.unk_token]
            self.dimiliter_token = "|"
            self.dimiliter_id = self.token_to_id_mapping[self.dimiliter_token]
            special_tokens = ["<pad>"]
            self.special_ids = [self.token_to_id_mapping[k] for k in special_tokens]

    def _setup_vocab(self):
        # the reference downloads vocab.json when absent (processor.py:37-50); no network here
        if not os.path.isfile(self.vocab_path):
            raise ValueError(f"Couldn't find `vocab.json` at {self.vocab_path} (and there is no network to fetch it)")

    def __call__(self, input_values):
        if self.is_tokenizer:
            tokens = self._tokenize(input_values)
            return [self.token_to_id_mapping.get(k, self.unk_id) for k in tokens]
        if self.do_normalize:
            input_values = self._normalize(input_values)
        return input_values

    def decode(self, input_ids, skip_special_tokens=True, group_tokens=True):
        input_ids = [int(i) for i in input_ids]
        if group_tokens:
            input_ids = [t[0] for t in groupby(input_ids)]
        if skip_special_tokens:
            input_ids = [k for k in input_ids if k not in self.special_ids]
        tokens = [self.id_to_token_mapping.get(k, self.unk_token) for k in input_ids]
        tokens = [k if k != self.dimiliter_token else " " for k in tokens]
        return "".join(tokens).strip()

    def _tokenize(self, string: str):
        string = re.sub("-", " ", string)
        string = re.sub("[^A-Z' ]", "", string.upper())
        return list(string.replace(" ", self.dimiliter_token))

    def get_vocab(self):
        with open(self.vocab_path, "r") as f:
            return json.load(f)

    def _normalize(self, x):
        """(x - mean) / sqrt(var + 1e-5) along the last axis, population
        variance, then squeeze.  Call before padding."""
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
