#!/usr/bin/env python3
"""Hand-assemble a small LMDB `data.mdb` page by page from liblmdb's on-disk format description (mdb.c, data version 1) - an
INDEPENDENT code path from ccd_amd/dataset/lmdb_file.write_lmdb (which bulk-loads sorted records bottom-up): here the pages are
laid out the way a handful of mdb_put calls leave them - nodes allocated from the END of a page downward in insertion order
(not key order), node sizes rounded up to even, a value above the node limit moved to a two-page overflow run, a branch root
over two leaves whose first node has an empty key, two meta pages of which the SECOND is the newer transaction, and a stale
older tree left behind on pages the newer meta no longer references.  tests/test_datapipe_cpu.py reads the committed file with
LmdbReader.  liblmdb itself (py-lmdb) is not part of this image: this is still not a file liblmdb wrote.

    python tools/make_lmdb_fixture.py        -> tests/golden/lmdb_handmade/{data.mdb, expected.json}
"""
import hashlib
import io
import json
import os
import struct

import numpy as np
from PIL import Image

PSIZE = 4096
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "lmdb_handmade")


def png(seed, h, w, noise):
    rs = np.random.RandomState(seed)
    img = (np.full((h, w, 3), 200) + (rs.rand(h, w, 3) * noise)).clip(0, 255).astype(np.uint8)
    img[h // 4: 3 * h // 4, w // 8: w // 2] = 30
    buf = io.BytesIO()
    Image.fromarray(img).save(buf, format="PNG")
    return buf.getvalue()


def page_header(pgno, flags, lower, upper):
    return struct.pack("<QHHHH", pgno, 0, flags, lower, upper)


def leaf_page(pgno, records, insertion_order, overflow_of):
    """records: sorted [(key, value)]; nodes are written in `insertion_order` from the end of the page downward."""
    page = bytearray(PSIZE)
    upper = PSIZE
    where = {}
    for idx in insertion_order:
        key, val = records[idx]
        if key in overflow_of:
            data, flags, size = struct.pack("<Q", overflow_of[key]), 0x01, len(val)          # F_BIGDATA: the run's page number
        else:
            data, flags, size = val, 0, len(val)
        node = struct.pack("<HHHH", size & 0xFFFF, size >> 16, flags, len(key)) + key + data
        node += b"\0" * (len(node) & 1)                                                        # EVEN(size)
        upper -= len(node)
        page[upper:upper + len(node)] = node
        where[idx] = upper
    lower = 16 + 2 * len(records)
    assert lower <= upper
    page[:16] = page_header(pgno, 0x02, lower, upper)
    for i in range(len(records)):
        struct.pack_into("<H", page, 16 + 2 * i, where[i])
    return bytes(page)


def branch_page(pgno, children):
    """children: [(separator key, child pgno)], the first separator is empty."""
    page = bytearray(PSIZE)
    upper = PSIZE
    ptrs = []
    for key, child in children:
        node = struct.pack("<HHHH", child & 0xFFFF, (child >> 16) & 0xFFFF, child >> 32, len(key)) + key
        node += b"\0" * (len(node) & 1)
        upper -= len(node)
        page[upper:upper + len(node)] = node
        ptrs.append(upper)
    page[:16] = page_header(pgno, 0x01, 16 + 2 * len(children), upper)
    for i, ptr in enumerate(ptrs):
        struct.pack_into("<H", page, 16 + 2 * i, ptr)
    return bytes(page)


def overflow_run(pgno, value):
    npages = (16 + len(value) + PSIZE - 1) // PSIZE
    run = bytearray(npages * PSIZE)
    run[:16] = struct.pack("<QHHI", pgno, 0, 0x04, npages)
    run[16:16 + len(value)] = value
    return bytes(run), npages


def meta_page(pgno, txnid, main, last_pg):
    depth, branch, leaf, ovf, entries, root = main
    page = bytearray(PSIZE)
    page[:16] = page_header(pgno, 0x08, 0, 0)
    body = struct.pack("<IIQQ", 0xBEEFC0DE, 1, 0, 1 << 20)
    body += struct.pack("<IHHQQQQQ", PSIZE, 0, 0, 0, 0, 0, 0, 0xFFFFFFFFFFFFFFFF)              # FREE_DBI: empty, md_pad = page size
    body += struct.pack("<IHHQQQQQ", 0, 0, depth, branch, leaf, ovf, entries, root)           # MAIN_DBI
    body += struct.pack("<QQ", last_pg, txnid)
    page[16:16 + len(body)] = body
    return bytes(page)


def main():
    big = png(2, 48, 200, 255.0)                       # incompressible noise: several KiB -> overflow pages
    assert len(big) > 2 * PSIZE // 3
    recs = sorted({
        b"num-samples": b"3",
        b"image-000000001": png(1, 24, 64, 0.0), b"label-000000001": b"Hello",
        b"image-000000002": big, b"label-000000002": "naïve".encode("utf-8"),
        b"image-000000003": png(3, 31, 90, 8.0), b"label-000000003": b"",
    }.items())
    left = [r for r in recs if r[0] < b"label"]
    right = [r for r in recs if r[0] >= b"label"]
    # transaction 1 (older, pages 2-3): only the first sample; transaction 2 (pages 4-8): the full set.  Meta 0 -> txn 2's parent
    # would be how liblmdb alternates; here meta 0 holds txn 1 and meta 1 holds txn 2, so a reader must pick the larger txnid.
    old = sorted([(b"num-samples", b"1"), (b"image-000000001", recs[0][1]), (b"label-000000001", b"Hello")])
    pages = {2: leaf_page(2, old, [2, 0, 1], {})}
    run, npages = overflow_run(7, big)
    pages[4] = leaf_page(4, left, [0, 2, 1], {b"image-000000002": 7})           # inserted 1, 3, 2
    pages[5] = leaf_page(5, right, [3, 0, 2, 1], {})                             # num-samples first, then labels out of order
    pages[6] = branch_page(6, [(b"", 4), (right[0][0], 5)])
    last = 7 + npages - 1
    blob = bytearray((last + 1) * PSIZE)
    blob[0:PSIZE] = meta_page(0, 1, (1, 0, 1, 0, len(old), 2), 2)
    blob[PSIZE:2 * PSIZE] = meta_page(1, 2, (2, 1, 2, npages, len(recs), 6), last)
    for pg, data in pages.items():
        blob[pg * PSIZE:(pg + 1) * PSIZE] = data
    blob[3 * PSIZE:4 * PSIZE] = b"\xAA" * PSIZE                                 # a freed page full of garbage
    blob[7 * PSIZE:7 * PSIZE + len(run)] = run
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, "data.mdb"), "wb") as f:
        f.write(blob)
    expected = {k.decode(): {"len": len(v), "sha256": hashlib.sha256(v).hexdigest()} for k, v in recs}
    with open(os.path.join(OUT, "expected.json"), "w") as f:
        json.dump({"entries": len(recs), "depth": 2, "txnid": 2, "psize": PSIZE, "records": expected}, f, indent=1, sort_keys=True)
    print(f"wrote {OUT}/data.mdb ({len(blob)} bytes, {len(recs)} records, overflow run of {npages} pages)")


if __name__ == "__main__":
    main()
