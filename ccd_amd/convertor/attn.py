"""Label codec of the recognition head: words <-> class indices <-> padded target tensors.

Behavioural contract (what `DINO_Finetune`, `train_finetune.py` and checkpoints rely on; reference:
Dino/convertor/base.py:3-110, Dino/convertor/attn.py:81-154): the DICT90 alphabet occupies classes 0..89, then
`<UKN>` (90, when enabled), `<BOS/EOS>` (91; one shared class unless start_end_same=False) and `<PAD>` (92);
a target row is `<BOS> chars <EOS> <PAD>...` cut to `max_seq_len`; decoding takes the arg-max class per position,
drops `<PAD>` and stops at the first `<EOS>`.

Implementation: the alphabet is a code-point -> class lookup table (numpy), so encoding a batch is three array
operations instead of a Python loop per character, and decoding is one arg-max + one first-EOS scan on the device.
"""
from __future__ import annotations

import numpy as np
import torch

_DIGITS_LOWER = "0123456789abcdefghijklmnopqrstuvwxyz"
_UPPER_PUNCT = "ABCDEFGHIJKLMNOPQRSTUVWXYZ!\"#$%&'()*+,-./:;<=>?@[\\]_`~"
ALPHABETS = {
    "DICT36": _DIGITS_LOWER,
    "DICT37": _DIGITS_LOWER + " ",
    "DICT90": _DIGITS_LOWER + _UPPER_PUNCT,
    "DICT91": _DIGITS_LOWER + _UPPER_PUNCT + " ",
}
_NO_CLASS = -1


def _read_alphabet_file(path):
    chars = []
    with open(path, encoding="utf-8") as f:
        for number, raw in enumerate(f, start=1):
            entry = raw.rstrip("\r\n")
            if len(entry) > 1:
                raise ValueError(f"{path}:{number}: a dictionary line holds at most one character, found {len(entry)}")
            if entry:
                chars.append(entry)
    return chars


class AttnConvertor:
    """AttnConvertor(dict_type='DICT90', max_seq_len=40, with_unknown=True) - see the module docstring."""

    dicts = {name: tuple(chars) for name, chars in ALPHABETS.items()}      # same table name as the reference exposes

    def __init__(self, dict_type="DICT90", dict_file=None, dict_list=None, with_unknown=True, max_seq_len=40, lower=False,
                 start_end_same=True, **_ignored):
        if dict_file is not None:
            alphabet = _read_alphabet_file(dict_file)
        elif dict_list is not None:
            alphabet = list(dict_list)
        elif dict_type in ALPHABETS:
            alphabet = list(ALPHABETS[dict_type])
        else:
            raise NotImplementedError(f"unknown dictionary type {dict_type!r} (have {sorted(ALPHABETS)})")
        if len(set(alphabet)) != len(alphabet):
            raise AssertionError("dictionary holds a character twice")
        self.with_unknown, self.max_seq_len = bool(with_unknown), int(max_seq_len)
        self.lower, self.start_end_same = bool(lower), bool(start_end_same)
        # special classes behind the alphabet, in the reference's order
        self.idx2char = alphabet
        self.unknown_idx = self._append("<UKN>") if self.with_unknown else None
        self.start_idx = self._append("<BOS/EOS>")
        self.end_idx = self.start_idx if self.start_end_same else self._append("<BOS/EOS>")
        self.padding_idx = self._append("<PAD>")
        self.char2idx = {c: i for i, c in enumerate(self.idx2char)}
        # code point -> class; characters outside the table resolve to <UKN> (or to _NO_CLASS -> error)
        points = [ord(c) for c in alphabet if len(c) == 1]
        self._lut = np.full(max(points) + 1 if points else 1, _NO_CLASS, dtype=np.int64)
        for cls, c in enumerate(self.idx2char[:len(alphabet)]):
            if len(c) == 1:
                self._lut[ord(c)] = cls

    def _append(self, token):
        self.idx2char.append(token)
        return len(self.idx2char) - 1

    def num_classes(self):
        return len(self.idx2char)

    # ------------------------------------------------------------------ encode
    def _classes_of(self, word):
        if self.lower:
            word = word.lower()
        points = np.frombuffer(word.encode("utf-32-le"), dtype="<u4").astype(np.int64)
        inside = points < self._lut.size
        cls = np.where(inside, self._lut[np.minimum(points, self._lut.size - 1)], _NO_CLASS)
        missing = cls == _NO_CLASS
        if missing.any():
            if self.unknown_idx is None:
                bad = word[int(np.argmax(missing))]
                raise KeyError(f"character {bad!r} is not in the dictionary (pass with_unknown=True or a custom dict_file)")
            cls = np.where(missing, self.unknown_idx, cls)
        return cls

    def str2idx(self, strings):
        if not isinstance(strings, list):
            raise TypeError("str2idx expects a list of strings")
        return [self._classes_of(w).tolist() for w in strings]

    def idx2str(self, indexes):
        if not isinstance(indexes, list):
            raise TypeError("idx2str expects a list of index lists")
        table = self.idx2char
        return ["".join(table[i] for i in row) for row in indexes]

    def str2tensor(self, strings):
        """['hello', ...] -> int64 [N, max_seq_len]: <BOS> chars <EOS> <PAD>... (cut to max_seq_len)."""
        if not isinstance(strings, list) or not all(isinstance(w, str) for w in strings):
            raise TypeError("str2tensor expects a list of strings")
        T = self.max_seq_len
        target = np.full((len(strings), T), self.padding_idx, dtype=np.int64)
        for row, word in zip(target, strings):
            cls = self._classes_of(word)[:max(T - 1, 0)]
            row[0] = self.start_idx
            row[1:1 + cls.size] = cls
            if 1 + cls.size < T:
                row[1 + cls.size] = self.end_idx
        return torch.from_numpy(target)

    # ------------------------------------------------------------------ decode
    @torch.no_grad()
    def tensor2idx(self, outputs, img_metas=None):
        """[N, T, C] class scores -> (class indices, softmax confidences) per sample: positions up to the first <EOS>,
        <PAD> positions skipped."""
        probs = outputs.float().softmax(dim=-1)
        conf, cls = probs.max(dim=-1)                                        # [N, T]
        is_end = cls == self.end_idx
        before_end = torch.cumsum(is_end.int(), dim=1) == 0                  # strictly before the first <EOS>
        keep = (before_end & (cls != self.padding_idx)).cpu().numpy()
        cls_np, conf_np = cls.cpu().numpy(), conf.cpu().numpy()
        indexes = [cls_np[i][keep[i]].tolist() for i in range(cls_np.shape[0])]
        scores = [conf_np[i][keep[i]].tolist() for i in range(cls_np.shape[0])]
        return indexes, scores
