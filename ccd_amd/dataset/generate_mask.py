"""Build the mask LMDB of an image LMDB: `python -m ccd_amd.dataset.generate_mask --root <image lmdb> --out <mask lmdb>`.

What mask_create/generate_mask.py does with 36 CPU processes (:36-88: for every `image-%09d`, PIL "L" conversion ->
`clusterpixels(image, 2)` -> PNG -> `mask-%09d`, plus `num-samples`), with the clustering done on the GPU in ragged batches
(ops.kmeans2_mask, csrc/kernels/datapipe.h) and the records written through ccd_amd.dataset.lmdb_file."""
from __future__ import annotations

import argparse
import io
import os

import numpy as np

from . import lmdb_file


def _gray(buf):
    from PIL import Image
    return np.asarray(Image.open(io.BytesIO(buf)).convert("L"))


def _png(mask):
    from PIL import Image
    out = io.BytesIO()
    Image.fromarray(mask.astype(np.uint8), mode="L").save(out, format="PNG")     # values 0 / 1, as cv2.imencode('.png', mask)
    return out.getvalue()


def generate(root, out, batch=512, device=None, start=1, end=None):
    """Masks of samples [start, end] (1-based, inclusive; default: all) of the image LMDB `root` -> mask LMDB `out`."""
    from .. import ops
    env = lmdb_file.LmdbReader(root)
    n = int(env.get(b"num-samples"))
    end = n if end is None else min(end, n)
    cnt = 0

    def records():
        """(key, PNG) in key order - `mask-%09d` by increasing index, then `num-samples` - one device batch in memory at a time:
        write_lmdb(presorted=True) streams them to disk (the reference's cached writer, generate_mask.py:48-58)."""
        nonlocal cnt
        pending, keys = [], []

        def flush():
            nonlocal pending, keys, cnt
            out_recs = [(k, _png(m)) for k, m in zip(keys, ops.kmeans2_mask(pending, device=device))] if pending else []
            cnt += len(out_recs)
            pending, keys = [], []
            return out_recs

        for index in range(start, end + 1):
            buf = env.get(b"image-%09d" % index)
            try:
                g = _gray(buf)
                if g.shape[0] < 2 and g.shape[1] < 2:                  # generate_mask.py:70-72
                    continue
            except Exception:
                print(f"Corrupted image for {index}")
                continue
            pending.append(g)
            keys.append(b"mask-%09d" % index)
            if len(pending) >= batch:
                yield from flush()
        yield from flush()
        yield b"num-samples", str(cnt).encode()

    os.makedirs(out, exist_ok=True)
    stat = lmdb_file.write_lmdb(out, records(), presorted=True)
    print(f"Created dataset with {cnt} samples")
    return stat


def main():
    ap = argparse.ArgumentParser(description=__doc__)
    ap.add_argument("--root", required=True)
    ap.add_argument("--out", required=True)
    ap.add_argument("--batch", type=int, default=512)
    a = ap.parse_args()
    generate(a.root, a.out, a.batch)


if __name__ == "__main__":
    main()
