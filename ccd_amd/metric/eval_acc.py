"""Word / character accuracy of a recogniser over a labelled dataset.

Same surface and the same arithmetic as `Dino.metric.eval_acc.TextAccuracy` (reference :10-64; driven by test.py:184-218):
`TextAccuracy(charset_path, case_sensitive, model_eval).compute(model, dataloader)` ->
    {'ccr', 'cwr', 'ted', 'ned', 'ted/w', 'words', 'time'}
* a word counts as correct (cwr) when ground truth and prediction agree after lower-casing and dropping every character
  outside [A-Za-z0-9 + CJK] (:39-46);
* ted / ned: edit distance of those normalised strings, ned divided by the RAW ground-truth length (:48-50);
* ccr: position-wise equal characters of the RAW strings over the raw ground-truth length (:53-56).
The scoring is `update(gt_text, pt_text)`, separated from the model loop so that it can be checked against the reference's
numbers without a model.  (`editdistance` is not in this image: the Levenshtein distance is computed here.)
"""
from __future__ import annotations

import re
import time

import torch

_KEEP = re.compile("[^A-Z^a-z^0-9^一-龥]")      # the reference's pattern, verbatim semantics: '^' itself is kept too


def levenshtein(a: str, b: str) -> int:
    """Unit-cost edit distance (what `editdistance.eval` returns)."""
    if a == b:
        return 0
    if len(a) < len(b):
        a, b = b, a
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a, start=1):
        cur = [i]
        for j, cb in enumerate(b, start=1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb)))
        prev = cur
    return prev[-1]


class TextAccuracy:
    def __init__(self, charset_path=None, case_sensitive=False, model_eval="vision"):
        assert model_eval in ("vision", "language", "alignment")
        self.charset_path, self.case_sensitive, self.model_eval = charset_path, case_sensitive, model_eval
        self._names = ["ccr", "cwr", "ted", "ned", "ted/w", "words", "time"]
        self.total_num_char = self.total_num_word = self.correct_num_char = self.correct_num_word = 0.0
        self.total_ed = self.total_ned = self.inference_time = 0.0

    def update(self, gt_text, pt_text):
        """Score one batch of (ground truth, prediction) strings."""
        if self.case_sensitive:
            # the reference only defines its normalised strings under `not case_sensitive` (:40-44) and fails otherwise
            raise NotImplementedError("TextAccuracy is defined for case_sensitive=False (eval_acc.py:40-46)")
        for gt, pt in zip(gt_text, pt_text):
            gt_n, pt_n = _KEEP.sub("", gt.lower()), _KEEP.sub("", pt.lower())
            if gt_n == pt_n:
                self.correct_num_word += 1
            distance = levenshtein(gt_n, pt_n)
            self.total_ed += distance
            self.total_ned += float(distance) / max(len(gt), 1)
            self.total_num_word += 1
            self.correct_num_char += sum(1 for j in range(min(len(gt), len(pt))) if gt[j] == pt[j])
            self.total_num_char += len(gt)

    def result(self):
        mets = [self.correct_num_char / self.total_num_char, self.correct_num_word / self.total_num_word, self.total_ed,
                self.total_ned, self.total_ed / self.total_num_word, self.total_num_word, self.inference_time]
        return dict(zip(self._names, mets))

    @torch.no_grad()
    def compute(self, model, dataloader):
        net = model.module if hasattr(model, "module") else model
        device = next(net.parameters()).device
        for image_tensors, label_tensors in dataloader:
            image_tensors = image_tensors.to(device)
            start = time.time()
            out_dec = model(image_tensors, text=None, return_loss=False, test_speed=False)
            label_indexes, _scores = net.label_convertor.tensor2idx(out_dec)
            pt_text = net.label_convertor.idx2str(label_indexes)
            self.inference_time += time.time() - start
            self.update(list(label_tensors[0]), pt_text)
        return self.result()
