"""Read (and bulk-write) LMDB environments without liblmdb: the reference keeps every dataset as an LMDB directory
(`data.mdb` + `lock.mdb`) with keys `num-samples`, `image-%09d`, `label-%09d` (Dino/dataset/dataset.py:66,135) and its
masks as `mask-%09d` (mask_create/generate_mask.py:60-85).  `py-lmdb` is a C extension that is not part of this image,
and the only operations the data path needs are "open read-only, no lock" + "get(key)" (dataset.py:55,133-145) and one
bulk load (generate_mask.py:30-33), so the on-disk format is implemented here directly.

File format (liblmdb 0.9, mdb.c - little endian, 64-bit): pages of `psize` bytes (4096 unless the environment was made
with another page size; stored in meta.mm_dbs[FREE].md_pad).  Every page starts with a 16-byte header
    u64 pgno | u16 pad | u16 flags | u16 lower | u16 upper         (overflow pages: u32 pages instead of lower/upper)
flags: 0x01 branch, 0x02 leaf, 0x04 overflow, 0x08 meta.  Pages 0 and 1 are meta pages: header, then
    u32 magic 0xBEEFC0DE | u32 version 1 | u64 address | u64 mapsize | MDB_db free | MDB_db main | u64 last_pg | u64 txnid
    MDB_db = u32 pad | u16 flags | u16 depth | u64 branch_pages | u64 leaf_pages | u64 overflow_pages | u64 entries | u64 root
and the one with the larger txnid is current.  A branch / leaf page holds `(lower - 16) / 2` u16 node offsets after the
header; a node is  u16 lo | u16 hi | u16 flags | u16 ksize | key | data.  Leaf: data size = lo | hi << 16, and with node
flag 0x01 (F_BIGDATA) the data is a u64 page number of an overflow run that holds the value.  Branch: child page =
lo | hi << 16 | flags << 32; the first node of a branch page has an empty key (smaller than everything).  Keys compare as
byte strings (memcmp, then length).
"""
from __future__ import annotations

import builtins
import mmap
import os
import struct

MAGIC, VERSION = 0xBEEFC0DE, 1
P_BRANCH, P_LEAF, P_OVERFLOW, P_META = 0x01, 0x02, 0x04, 0x08
F_BIGDATA = 0x01
PAGEHDR = 16
NODEHDR = 8
P_INVALID = 0xFFFFFFFFFFFFFFFF
_DB = struct.Struct("<IHHQQQQQ")          # MDB_db, 48 bytes
_META = struct.Struct("<IIQQ")            # magic, version, address, mapsize


class LmdbError(IOError):
    pass


class LmdbReader:
    """`LmdbReader(dir_or_file)`: read-only view of the main database of an LMDB environment (a snapshot of the newest
    committed transaction at open time).  Mirrors the calls the reference makes: `get(key) -> bytes | None`, `stat()`;
    plus `items()` in key order.  Thread/process safe (a private read-only mmap per instance)."""

    def __init__(self, path):
        path = os.fspath(path)
        if os.path.isdir(path):
            path = os.path.join(path, "data.mdb")
        if not os.path.isfile(path):
            raise LmdbError(f"{path} is not an LMDB environment (no data.mdb)")
        self.path = path
        self._fh = builtins.open(path, "rb")
        size = os.fstat(self._fh.fileno()).st_size
        if size < 2 * 512:
            raise LmdbError(f"{path}: too short for an LMDB file")
        self._mm = mmap.mmap(self._fh.fileno(), 0, access=mmap.ACCESS_READ)
        metas = []
        psize = None
        for probe in (0, 1):
            # the page size is only known from meta 0 (its md_pad); meta 1 starts one page later
            off = 0 if probe == 0 else (psize or 4096)
            hdr_flags = struct.unpack_from("<H", self._mm, off + 10)[0]
            magic, version, _addr, _mapsize = _META.unpack_from(self._mm, off + PAGEHDR)
            if magic != MAGIC or not (hdr_flags & P_META):
                if probe == 0:
                    raise LmdbError(f"{path}: bad magic - not an LMDB data file")
                continue
            if version != VERSION:
                raise LmdbError(f"{path}: LMDB data format version {version} is not supported")
            free = _DB.unpack_from(self._mm, off + PAGEHDR + _META.size)
            main = _DB.unpack_from(self._mm, off + PAGEHDR + _META.size + _DB.size)
            last_pg, txnid = struct.unpack_from("<QQ", self._mm, off + PAGEHDR + _META.size + 2 * _DB.size)
            if probe == 0:
                psize = free[0] or 4096
            metas.append((txnid, main, last_pg))
        self.psize = psize
        txnid, main, last_pg = max(metas, key=lambda m: m[0])
        self.txnid, self.last_pg = txnid, last_pg
        (_pad, self.db_flags, self.depth, self.branch_pages, self.leaf_pages, self.overflow_pages, self.entries,
         self.root) = main
        if self.db_flags & ~0x08:   # MDB_REVERSEKEY 0x02, DUPSORT 0x04, INTEGERKEY 0x08 ...: none used by the datasets
            if self.db_flags & 0x06:
                raise LmdbError(f"{path}: database flags {self.db_flags:#x} (reverse / duplicate keys) are not supported")

    # ------------------------------------------------------------------ pages
    def _page(self, pgno):
        off = pgno * self.psize
        if off + PAGEHDR > len(self._mm):
            raise LmdbError(f"{self.path}: page {pgno} beyond the end of the file (truncated copy?)")
        flags, lower, upper = struct.unpack_from("<HHH", self._mm, off + 10)
        return off, flags, lower, upper

    def _node(self, page_off, index):
        ptr = struct.unpack_from("<H", self._mm, page_off + PAGEHDR + 2 * index)[0]
        lo, hi, nflags, ksize = struct.unpack_from("<HHHH", self._mm, page_off + ptr)
        key_off = page_off + ptr + NODEHDR
        return lo, hi, nflags, ksize, key_off

    def _leaf_value(self, lo, hi, nflags, ksize, key_off):
        size = lo | (hi << 16)
        data_off = key_off + ksize
        if nflags & F_BIGDATA:
            ovf = struct.unpack_from("<Q", self._mm, data_off)[0]
            off, flags, _l, _u = self._page(ovf)
            if not flags & P_OVERFLOW:
                raise LmdbError(f"{self.path}: page {ovf} should be an overflow page")
            return bytes(self._mm[off + PAGEHDR: off + PAGEHDR + size])
        return bytes(self._mm[data_off: data_off + size])

    # ------------------------------------------------------------------ lookups
    def get(self, key, default=None):
        if isinstance(key, str):
            key = key.encode()
        if self.root == P_INVALID or self.entries == 0:
            return default
        pgno = self.root
        for _ in range(64):
            off, flags, lower, _upper = self._page(pgno)
            n = (lower - PAGEHDR) // 2
            if flags & P_BRANCH:
                # last node whose key <= search key (node 0 has the implicit lowest key)
                lo_i, hi_i = 1, n - 1
                child = 0
                while lo_i <= hi_i:
                    mid = (lo_i + hi_i) // 2
                    _lo, _hi, _nf, ks, ko = self._node(off, mid)
                    if bytes(self._mm[ko: ko + ks]) <= key:
                        child, lo_i = mid, mid + 1
                    else:
                        hi_i = mid - 1
                lo, hi, nflags, _ks, _ko = self._node(off, child)
                pgno = lo | (hi << 16) | (nflags << 32)
            elif flags & P_LEAF:
                lo_i, hi_i = 0, n - 1
                while lo_i <= hi_i:
                    mid = (lo_i + hi_i) // 2
                    lo, hi, nflags, ks, ko = self._node(off, mid)
                    k = bytes(self._mm[ko: ko + ks])
                    if k == key:
                        return self._leaf_value(lo, hi, nflags, ks, ko)
                    if k < key:
                        lo_i = mid + 1
                    else:
                        hi_i = mid - 1
                return default
            else:
                raise LmdbError(f"{self.path}: page {pgno} is neither branch nor leaf (flags {flags:#x})")
        raise LmdbError(f"{self.path}: tree deeper than 64 levels - corrupt file")

    def items(self):
        """(key, value) pairs in key order."""
        if self.root == P_INVALID or self.entries == 0:
            return
        stack = [self.root]
        while stack:
            pgno = stack.pop()
            off, flags, lower, _upper = self._page(pgno)
            n = (lower - PAGEHDR) // 2
            if flags & P_BRANCH:
                children = []
                for i in range(n):
                    lo, hi, nflags, _ks, _ko = self._node(off, i)
                    children.append(lo | (hi << 16) | (nflags << 32))
                stack.extend(reversed(children))
            else:
                for i in range(n):
                    lo, hi, nflags, ks, ko = self._node(off, i)
                    yield bytes(self._mm[ko: ko + ks]), self._leaf_value(lo, hi, nflags, ks, ko)

    def keys(self):
        for k, _ in self.items():
            yield k

    def stat(self):
        return {"psize": self.psize, "depth": self.depth, "branch_pages": self.branch_pages, "leaf_pages": self.leaf_pages,
                "overflow_pages": self.overflow_pages, "entries": self.entries}

    def __len__(self):
        return self.entries

    def close(self):
        if self._mm is not None:
            self._mm.close()
            self._fh.close()
            self._mm = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # the two context-manager calls of the reference (`with env.begin(write=False) as txn: txn.get(...)`) work unchanged
    def begin(self, write=False, **_kw):
        if write:
            raise LmdbError("LmdbReader is read-only; build environments with write_lmdb()")
        return _Txn(self)


class _Txn:
    def __init__(self, env):
        self.env = env

    def get(self, key, default=None):
        return self.env.get(key, default)

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def open(path, readonly=True, **_ignored):          # noqa: A001 - `lmdb.open(path, readonly=True, lock=False, ...)`
    if not readonly:
        raise LmdbError("only read-only environments can be opened; build them with write_lmdb()")
    return LmdbReader(path)


# ---------------------------------------------------------------------------------------------------- bulk writer
def write_lmdb(path, items, psize=4096, presorted=False):
    """Create the environment `path` (a directory) holding `items` (mapping or iterable of (key, value) byte strings) in
    ONE committed transaction - what generate_mask.py / the dataset converters do with put() in a loop.  Produces the same
    structures liblmdb's append-mode load does: sorted leaves filled front to back, values larger than a quarter page
    (liblmdb: node > (psize - 16) / 2 with its 2 minimum keys) on overflow runs, branch levels on top.
    `presorted=True`: `items` is an ITERATOR that yields strictly increasing keys (MDB_APPEND's contract); pages are written
    to the file as they fill and only the (first key, page) index of each level stays in memory - the reference builds its
    mask databases for multi-million-image datasets the same way, with a write cache flushed every 1000 records
    (generate_mask.py:48-58)."""
    def norm(kv):
        k, v = kv
        return (k.encode() if isinstance(k, str) else bytes(k)), bytes(v)
    if presorted:
        pairs = (norm(kv) for kv in items)
    else:
        pairs = sorted((norm(kv) for kv in (items.items() if hasattr(items, "items") else items)), key=lambda kv: kv[0])
    maxkey = 511
    os.makedirs(path, exist_ok=True)
    nodemax = (((psize - PAGEHDR) // 2) & ~1) - 2    # me_nodemax: larger leaf nodes put their data on overflow pages
    counts = {"leaf": 0, "branch": 0, "ovf": 0, "entries": 0}
    state = {"next_pg": 2}
    # (written next to the target and moved over it at the end: a bad key or an exception inside a streaming `items` generator must
    # not leave a truncated database with zeroed meta pages where a good one was)
    final_path, tmp_path = os.path.join(path, "data.mdb"), os.path.join(path, "data.mdb.tmp")
    f = builtins.open(tmp_path, "wb")
    f.write(b"\0" * (2 * psize))                      # the two meta pages are written last

    def emit(image):
        """Append page images (whole pages) at page number next_pg."""
        assert f.tell() == state["next_pg"] * psize and len(image) % psize == 0
        f.write(image)
        first = state["next_pg"]
        state["next_pg"] += len(image) // psize
        return first

    def page_image(pgno, flags, nodes):
        """nodes: list of encoded node byte strings (2-byte aligned); laid out from the page end downwards."""
        buf = bytearray(psize)
        upper = psize
        ptrs = []
        for nd in nodes:
            upper -= len(nd)
            buf[upper: upper + len(nd)] = nd
            ptrs.append(upper)
        lower = PAGEHDR + 2 * len(nodes)
        assert lower <= upper, "page overfull"
        struct.pack_into("<QHHHH", buf, 0, pgno, 0, flags, lower, upper)
        for i, p in enumerate(ptrs):
            struct.pack_into("<H", buf, PAGEHDR + 2 * i, p)
        return bytes(buf)

    def even(b):
        return b + b"\0" if len(b) & 1 else b

    # ---- leaves (and the overflow runs of their big values), streamed
    level = []                      # (first key, pgno) of the pages of the level being built
    cur, used, first_key, prev_key = [], PAGEHDR, None, None

    def flush(flags, kind):
        nonlocal cur, used, first_key
        if not cur:
            return
        pg = emit(page_image(state["next_pg"], flags, cur))
        level.append((first_key, pg))
        counts[kind] += 1
        cur, used, first_key = [], PAGEHDR, None

    try:
        for k, v in pairs:
            if not 0 < len(k) <= maxkey:
                raise ValueError(f"key length {len(k)} outside 1..{maxkey}")
            if prev_key is not None and k <= prev_key:
                raise ValueError(f"duplicate key {k!r}" if k == prev_key else f"keys out of order: {prev_key!r} before {k!r}")
            prev_key = k
            counts["entries"] += 1
            if NODEHDR + len(k) + len(v) > nodemax:
                npages = (PAGEHDR + len(v) + psize - 1) // psize
                img = bytearray(npages * psize)
                struct.pack_into("<QHHI", img, 0, state["next_pg"], 0, P_OVERFLOW, npages)
                img[PAGEHDR: PAGEHDR + len(v)] = v
                ovf = emit(bytes(img))
                counts["ovf"] += npages
                node = struct.pack("<HHHH", len(v) & 0xFFFF, len(v) >> 16, F_BIGDATA, len(k)) + k + struct.pack("<Q", ovf)
            else:
                node = struct.pack("<HHHH", len(v) & 0xFFFF, len(v) >> 16, 0, len(k)) + k + v
            node = even(node)
            if used + 2 + len(node) > psize:
                flush(P_LEAF, "leaf")
            if first_key is None:
                first_key = k
            cur.append(node)
            used += 2 + len(node)
        flush(P_LEAF, "leaf")

        # ---- branch levels
        depth = 1 if level else 0
        while len(level) > 1:
            below, level = level, []
            for k, child in below:
                def enc(key):
                    return even(struct.pack("<HHHH", child & 0xFFFF, (child >> 16) & 0xFFFF, (child >> 32) & 0xFFFF, len(key)) + key)
                node = enc(b"" if not cur else k)
                if used + 2 + len(node) > psize:
                    flush(P_BRANCH, "branch")
                    node = enc(b"")
                if first_key is None:
                    first_key = k
                cur.append(node)
                used += 2 + len(node)
            flush(P_BRANCH, "branch")
            depth += 1
        root = level[0][1] if level else P_INVALID
        last_pg = state["next_pg"] - 1

        def meta(pgno, txnid, with_tree):
            buf = bytearray(psize)
            struct.pack_into("<QHHHH", buf, 0, pgno, 0, P_META, 0, 0)
            mapsize = max(10 << 20, state["next_pg"] * psize)
            _META.pack_into(buf, PAGEHDR, MAGIC, VERSION, 0, mapsize)
            _DB.pack_into(buf, PAGEHDR + _META.size, psize, 0, 0, 0, 0, 0, 0, P_INVALID)
            if with_tree:
                _DB.pack_into(buf, PAGEHDR + _META.size + _DB.size, 0, 0, depth, counts["branch"], counts["leaf"], counts["ovf"],
                              counts["entries"], root)
                lp = last_pg
            else:
                _DB.pack_into(buf, PAGEHDR + _META.size + _DB.size, 0, 0, 0, 0, 0, 0, 0, P_INVALID)
                lp = 1
            struct.pack_into("<QQ", buf, PAGEHDR + _META.size + 2 * _DB.size, lp, txnid)
            return bytes(buf)

        f.seek(0)
        f.write(meta(0, 0, False))          # the state before the load (txn 0), as mdb_env_init_meta leaves it
        f.write(meta(1, 1, True))           # the committed load (txn 1 -> meta page 1)
    except BaseException:
        f.close()
        try:
            os.unlink(tmp_path)
        except OSError:
            pass
        raise
    f.close()
    os.replace(tmp_path, final_path)
    with builtins.open(os.path.join(path, "lock.mdb"), "wb") as fl:
        fl.write(b"\0" * 8192)
    return {"entries": counts["entries"], "depth": depth, "leaf_pages": counts["leaf"], "branch_pages": counts["branch"],
            "overflow_pages": counts["ovf"]}
