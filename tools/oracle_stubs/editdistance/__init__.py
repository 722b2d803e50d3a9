"""Stand-in for the `editdistance` C extension (absent from this image) so that the reference's Dino/metric/eval_acc.py can be
imported by tools/gen_golden.py: textbook unit-cost Levenshtein distance, the quantity `editdistance.eval` returns."""


def eval(a, b):                                   # noqa: A001 - the package's public name
    a, b = list(a), list(b)
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a, start=1):
        cur = [i]
        for j, cb in enumerate(b, start=1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb)))
        prev = cur
    return prev[-1]
