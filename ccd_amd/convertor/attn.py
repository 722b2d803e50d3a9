"""Text <-> index <-> tensor conversion of the recognition head (host logic; reference: Dino/convertor/base.py:3-110,
Dino/convertor/attn.py:6-154).  DICT90 + <UKN> = 90, <BOS/EOS> = 91, <PAD> = 92."""
from __future__ import annotations

import torch


class BaseConvertor:
    start_idx = end_idx = padding_idx = 0
    unknown_idx = None
    lower = False

    dicts = dict(
        DICT36=tuple('0123456789abcdefghijklmnopqrstuvwxyz'),
        DICT90=tuple('0123456789abcdefghijklmnopqrstuvwxyz' 'ABCDEFGHIJKLMNOPQRSTUVWXYZ!"#$%&\'()' '*+,-./:;<=>?@[\\]_`~'),
        DICT37=tuple('0123456789abcdefghijklmnopqrstuvwxyz '),
        DICT91=tuple('0123456789abcdefghijklmnopqrstuvwxyz' 'ABCDEFGHIJKLMNOPQRSTUVWXYZ!"#$%&\'()' '*+,-./:;<=>?@[\\]_`~ '))

    def __init__(self, dict_type='DICT90', dict_file=None, dict_list=None):
        assert dict_file is None or isinstance(dict_file, str)
        assert dict_list is None or isinstance(dict_list, list)
        self.idx2char = []
        if dict_file is not None:
            with open(dict_file, encoding="utf-8") as f:
                for line_num, line in enumerate(f):
                    line = line.strip('\r\n')
                    if len(line) > 1:
                        raise ValueError(f'Expect each line has 0 or 1 character, got {len(line)} characters at line '
                                         f'{line_num + 1}')
                    if line != '':
                        self.idx2char.append(line)
        elif dict_list is not None:
            self.idx2char = list(dict_list)
        elif dict_type in self.dicts:
            self.idx2char = list(self.dicts[dict_type])
        else:
            raise NotImplementedError(f'Dict type {dict_type} is not supported')
        assert len(set(self.idx2char)) == len(self.idx2char), 'Invalid dictionary: Has duplicated characters.'
        self.char2idx = {char: idx for idx, char in enumerate(self.idx2char)}

    def num_classes(self):
        return len(self.idx2char)

    def str2idx(self, strings):
        assert isinstance(strings, list)
        indexes = []
        for string in strings:
            if self.lower:
                string = string.lower()
            index = []
            for char in string:
                char_idx = self.char2idx.get(char, self.unknown_idx)
                if char_idx is None:
                    raise Exception(f'Chararcter: {char} not in dict, please check gt_label and use custom dict file, '
                                    'or set "with_unknown=True"')
                index.append(char_idx)
            indexes.append(index)
        return indexes

    def idx2str(self, indexes):
        assert isinstance(indexes, list)
        return [''.join(self.idx2char[i] for i in index) for index in indexes]


class AttnConvertor(BaseConvertor):
    def __init__(self, dict_type='DICT90', dict_file=None, dict_list=None, with_unknown=True, max_seq_len=40, lower=False,
                 start_end_same=True, **kwargs):
        super().__init__(dict_type, dict_file, dict_list)
        assert isinstance(with_unknown, bool) and isinstance(max_seq_len, int) and isinstance(lower, bool)
        self.with_unknown, self.max_seq_len, self.lower, self.start_end_same = with_unknown, max_seq_len, lower, start_end_same
        self.update_dict()

    def update_dict(self):
        self.unknown_idx = None
        if self.with_unknown:
            self.idx2char.append('<UKN>')
            self.unknown_idx = len(self.idx2char) - 1
        self.idx2char.append('<BOS/EOS>')
        self.start_idx = len(self.idx2char) - 1
        if not self.start_end_same:
            self.idx2char.append('<BOS/EOS>')
        self.end_idx = len(self.idx2char) - 1
        self.idx2char.append('<PAD>')
        self.padding_idx = len(self.idx2char) - 1
        self.char2idx = {char: idx for idx, char in enumerate(self.idx2char)}

    def str2tensor(self, strings):
        """['hello', ...] -> int64 [N, max_seq_len]: <BOS> chars <EOS> <PAD>... (truncated to max_seq_len)."""
        assert isinstance(strings, list) and all(isinstance(s, str) for s in strings)
        rows = []
        for index in self.str2idx(strings):
            src = [self.start_idx] + list(index) + [self.end_idx]
            row = [self.padding_idx] * self.max_seq_len
            if len(src) > self.max_seq_len:
                row = src[:self.max_seq_len]
            else:
                row[:len(src)] = src
            rows.append(row)
        return torch.tensor(rows, dtype=torch.long)

    def tensor2idx(self, outputs, img_metas=None):
        """[N, T, C] scores -> (indexes, scores) up to the first <EOS>, <PAD> skipped (attn.py:107-154)."""
        batch_size = outputs.size(0)
        ignore_indexes = [self.padding_idx]
        indexes, scores = [], []
        for idx in range(batch_size):
            seq = outputs[idx].softmax(dim=-1)
            max_value, max_idx = torch.max(seq, -1)
            str_index, str_score = [], []
            for char_index, char_score in zip(max_idx.cpu().tolist(), max_value.cpu().tolist()):
                if char_index in ignore_indexes:
                    continue
                if char_index == self.end_idx:
                    break
                str_index.append(char_index)
                str_score.append(char_score)
            indexes.append(str_index)
            scores.append(str_score)
        return indexes, scores
