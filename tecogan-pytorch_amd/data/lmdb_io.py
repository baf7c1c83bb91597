"""Dependency-free access to the LMDB files the reference trains from
(scripts/create_lmdb.py:57 writes `{seq}_{n}x{h}x{w}_{i:04d}` -> raw RGB uint8 HWC bytes;
codes/data/base_dataset.py:43-57 reads them back through the `lmdb` package).

The `lmdb` Python package is a C extension that this path does not want as a dependency:
training reads every frame once (into HBM, see device_clip_store.py) and never writes.  This
module therefore parses LMDB's on-disk format directly -- `data.mdb` is a copy-on-write B+tree
of fixed-size pages, documented by LMDB's mdb.c (0.9.x, 64-bit little-endian layout):

  page header (16 B)   pgno u64 | pad u16 | flags u16 | lower u16 | upper u16
                       (overflow pages: the last 4 bytes are the page COUNT, data follows)
  flags                P_BRANCH 1, P_LEAF 2, P_OVERFLOW 4, P_META 8
  node ptrs            u16[] right after the header, (lower - 16) / 2 of them, each the
                       offset of a node inside the page
  node (8 B + key)     lo u16 | hi u16 | flags u16 | ksize u16 | key | data
                       leaf:   data size = lo | hi << 16; F_BIGDATA (1): data = u64 pgno of an
                               overflow run holding the value
                       branch: child pgno = lo | hi << 16 | flags << 32; node 0's key is empty
  meta (pages 0, 1)    magic 0xBEEFC0DE u32 | version u32 | address u64 | mapsize u64 |
                       MDB_db[2] (48 B each: pad u32 | flags u16 | depth u16 | branch_pages u64 |
                       leaf_pages u64 | overflow_pages u64 | entries u64 | root u64) |
                       last_pg u64 | txnid u64;  dbs[0].pad is the page size, dbs[1] the main
                       tree; the meta with the larger txnid is current.

`LMDBWriter` produces the same format by bulk-loading sorted items (fixtures, converters).
The `lmdb` package is not installed in the authoring container, so reader and writer are
checked against each other and against this layout, not against liblmdb itself."""
import mmap
import os
import struct

MAGIC = 0xBEEFC0DE
P_BRANCH, P_LEAF, P_OVERFLOW, P_META = 1, 2, 4, 8
F_BIGDATA = 1
HDR = 16
P_INVALID = 0xFFFFFFFFFFFFFFFF


def _data_file(path):
    return os.path.join(path, 'data.mdb') if os.path.isdir(path) else path


class LMDBReader:
    """Read-only view of the main database of an LMDB environment (directory holding
    data.mdb, or the file itself).  get() returns a zero-copy memoryview into the mapping."""

    def __init__(self, path):
        self.path = _data_file(path)
        self._f = open(self.path, 'rb')
        self._mm = mmap.mmap(self._f.fileno(), 0, access=mmap.ACCESS_READ)
        mv = memoryview(self._mm)
        best = None
        psize = struct.unpack_from('<I', mv, HDR + 24)[0]          # dbs[0].md_pad of meta page 0
        if psize < 512 or psize & (psize - 1):
            raise ValueError(f'{self.path}: not an LMDB data file (page size field {psize})')
        for pg in (0, 1):
            base = pg * psize
            if base + HDR + 136 > len(mv):
                continue
            flags = struct.unpack_from('<H', mv, base + 10)[0]
            magic, version = struct.unpack_from('<II', mv, base + HDR)
            if not (flags & P_META) or magic != MAGIC:
                continue
            depth = struct.unpack_from('<H', mv, base + HDR + 24 + 48 + 6)[0]
            entries, root = struct.unpack_from('<QQ', mv, base + HDR + 24 + 48 + 32)
            last_pg, txnid = struct.unpack_from('<QQ', mv, base + HDR + 24 + 96)
            if best is None or txnid > best['txnid']:
                best = dict(txnid=txnid, root=root, entries=entries, depth=depth, last_pg=last_pg,
                            version=version)
        if best is None:
            raise ValueError(f'{self.path}: no valid LMDB meta page')
        self.psize, self.meta, self._mv = psize, best, mv

    # -- page helpers ---------------------------------------------------------
    def _page(self, pgno):
        off = pgno * self.psize
        flags, lower = struct.unpack_from('<HH', self._mv, off + 10)
        return off, flags, (lower - HDR) >> 1

    def _node(self, off, i):
        ptr = struct.unpack_from('<H', self._mv, off + HDR + 2 * i)[0]
        lo, hi, flags, ksize = struct.unpack_from('<HHHH', self._mv, off + ptr)
        return off + ptr, lo, hi, flags, ksize

    def _value(self, npos, lo, hi, flags, ksize):
        size = lo | (hi << 16)
        dpos = npos + 8 + ksize
        if flags & F_BIGDATA:
            pgno = struct.unpack_from('<Q', self._mv, dpos)[0]
            dpos = pgno * self.psize + HDR
        return self._mv[dpos:dpos + size]

    def get(self, key):
        """Value of `key` (bytes or ascii str) as a memoryview, or None."""
        if isinstance(key, str):
            key = key.encode('ascii')
        pgno = self.meta['root']
        if pgno == P_INVALID:
            return None
        while True:
            off, flags, n = self._page(pgno)
            if flags & P_LEAF:
                lo_i, hi_i = 0, n - 1
                while lo_i <= hi_i:
                    mid = (lo_i + hi_i) >> 1
                    npos, lo, hi, nflags, ksize = self._node(off, mid)
                    k = bytes(self._mv[npos + 8:npos + 8 + ksize])
                    if k == key:
                        return self._value(npos, lo, hi, nflags, ksize)
                    if k < key:
                        lo_i = mid + 1
                    else:
                        hi_i = mid - 1
                return None
            if not (flags & P_BRANCH):
                raise ValueError(f'{self.path}: page {pgno} is neither branch nor leaf (flags {flags})')
            # last node whose key <= search key (node 0 has the implicit -infinity key)
            lo_i, hi_i, pick = 1, n - 1, 0
            while lo_i <= hi_i:
                mid = (lo_i + hi_i) >> 1
                npos, lo, hi, nflags, ksize = self._node(off, mid)
                if bytes(self._mv[npos + 8:npos + 8 + ksize]) <= key:
                    pick, lo_i = mid, mid + 1
                else:
                    hi_i = mid - 1
            npos, lo, hi, nflags, ksize = self._node(off, pick)
            pgno = lo | (hi << 16) | (nflags << 32)

    def items(self):
        """(key bytes, value memoryview) in key order."""
        root = self.meta['root']
        if root == P_INVALID:
            return
        stack = [root]
        while stack:
            pgno = stack.pop()
            off, flags, n = self._page(pgno)
            if flags & P_LEAF:
                for i in range(n):
                    npos, lo, hi, nflags, ksize = self._node(off, i)
                    yield bytes(self._mv[npos + 8:npos + 8 + ksize]), self._value(npos, lo, hi, nflags, ksize)
            else:
                kids = []
                for i in range(n):
                    npos, lo, hi, nflags, ksize = self._node(off, i)
                    kids.append(lo | (hi << 16) | (nflags << 32))
                stack.extend(reversed(kids))

    def keys(self):
        return [k for k, _ in self.items()]

    def __len__(self):
        return self.meta['entries']

    def close(self):
        self._mv.release()
        self._mm.close()
        self._f.close()


class LMDBWriter:
    """Bulk-load writer: LMDBWriter(dir).write(sorted-or-not dict/iterable of (key, bytes))."""

    def __init__(self, path, psize=4096):
        os.makedirs(path, exist_ok=True)
        self.path, self.psize = os.path.join(path, 'data.mdb'), psize
        self.nodemax = (((psize - HDR) // 2) & ~1) - 2

    def write(self, items):
        items = sorted(((k.encode('ascii') if isinstance(k, str) else bytes(k)), bytes(v))
                       for k, v in (items.items() if hasattr(items, 'items') else items))
        ps = self.psize
        pages = {}                      # pgno -> bytes (single pages) ; overflow runs stored whole
        next_pg = [2]

        def alloc(n=1):
            p = next_pg[0]
            next_pg[0] += n
            return p
        counts = dict(branch=0, leaf=0, overflow=0)

        def build_page(flags, nodes):
            """nodes: list of packed node bytes (already even-sized)."""
            buf = bytearray(ps)
            upper = ps
            ptrs = []
            for nd in nodes:
                upper -= len(nd)
                buf[upper:upper + len(nd)] = nd
                ptrs.append(upper)
            lower = HDR + 2 * len(nodes)
            assert lower <= upper
            for i, p in enumerate(ptrs):
                struct.pack_into('<H', buf, HDR + 2 * i, p)
            return buf, lower, upper

        def finish(pgno, flags, buf, lower, upper):
            struct.pack_into('<QHHHH', buf, 0, pgno, 0, flags, lower, upper)
            pages[pgno] = bytes(buf)

        # leaf level
        level = []                      # (first key, pgno)
        cur, cur_bytes, first = [], 0, None

        def flush_leaf():
            nonlocal cur, cur_bytes, first
            if not cur:
                return
            pgno = alloc()
            buf, lower, upper = build_page(P_LEAF, cur)
            finish(pgno, P_LEAF, buf, lower, upper)
            counts['leaf'] += 1
            level.append((first, pgno))
            cur, cur_bytes, first = [], 0, None
        for k, v in items:
            if 8 + len(k) + len(v) > self.nodemax:
                npg = (HDR + len(v) + ps - 1) // ps
                opg = alloc(npg)
                run = bytearray(npg * ps)
                struct.pack_into('<QHHI', run, 0, opg, 0, P_OVERFLOW, npg)
                run[HDR:HDR + len(v)] = v
                pages[opg] = bytes(run)
                counts['overflow'] += npg
                node = struct.pack('<HHHH', len(v) & 0xFFFF, len(v) >> 16, F_BIGDATA, len(k)) + k + \
                    struct.pack('<Q', opg)
            else:
                node = struct.pack('<HHHH', len(v) & 0xFFFF, len(v) >> 16, 0, len(k)) + k + v
            if len(node) & 1:
                node += b'\0'
            if cur and HDR + 2 * (len(cur) + 1) + cur_bytes + len(node) > ps:
                flush_leaf()
            if first is None:
                first = k
            cur.append(node)
            cur_bytes += len(node)
        flush_leaf()
        depth = 1 if level else 0
        # branch levels
        while len(level) > 1:
            up, cur, cur_bytes, first = [], [], 0, None
            for i, (k, pg) in enumerate(level):
                key = b'' if not cur else k                 # node 0 of a branch page: empty key
                node = struct.pack('<HHHH', pg & 0xFFFF, (pg >> 16) & 0xFFFF, (pg >> 32) & 0xFFFF,
                                   len(key)) + key
                if len(node) & 1:
                    node += b'\0'
                if cur and HDR + 2 * (len(cur) + 1) + cur_bytes + len(node) > ps:
                    pgno = alloc()
                    buf, lower, upper = build_page(P_BRANCH, cur)
                    finish(pgno, P_BRANCH, buf, lower, upper)
                    counts['branch'] += 1
                    up.append((first, pgno))
                    cur, cur_bytes, first = [], 0, None
                    node = struct.pack('<HHHH', pg & 0xFFFF, (pg >> 16) & 0xFFFF, (pg >> 32) & 0xFFFF, 0)
                if first is None:
                    first = k
                cur.append(node)
                cur_bytes += len(node)
            pgno = alloc()
            buf, lower, upper = build_page(P_BRANCH, cur)
            finish(pgno, P_BRANCH, buf, lower, upper)
            counts['branch'] += 1
            up.append((first, pgno))
            level = up
            depth += 1
        root = level[0][1] if level else P_INVALID
        last_pg = next_pg[0] - 1

        def meta(pgno, txnid, root_, depth_, entries):
            buf = bytearray(ps)
            struct.pack_into('<QHHHH', buf, 0, pgno, 0, P_META, 0, 0)
            struct.pack_into('<IIQQ', buf, HDR, MAGIC, 1, 0, max(next_pg[0] * ps, 1 << 20))
            # FREE_DBI: pad = page size, empty tree
            struct.pack_into('<IHHQQQQQ', buf, HDR + 24, ps, 0, 0, 0, 0, 0, 0, P_INVALID)
            struct.pack_into('<IHHQQQQQ', buf, HDR + 24 + 48, 0, 0, depth_, counts['branch'] if entries else 0,
                             counts['leaf'] if entries else 0, counts['overflow'] if entries else 0,
                             entries, root_)
            struct.pack_into('<QQ', buf, HDR + 24 + 96, last_pg if entries else 1, txnid)
            return bytes(buf)
        with open(self.path, 'wb') as f:
            f.write(meta(0, 0, P_INVALID, 0, 0))
            f.write(meta(1, 1, root, depth, len(items)))
            pg = 2
            while pg < next_pg[0]:
                blob = pages[pg]
                f.write(blob)
                pg += len(blob) // ps
        return len(items)


_FRAME_KEY = None


def is_frame_key(key):
    """`{seq}_{n}x{h}x{w}_{i:04d}` (scripts/create_lmdb.py:57) -- anything else in an LMDB (bookkeeping entries such
    as `__len__`) is not a frame."""
    global _FRAME_KEY
    if _FRAME_KEY is None:
        import re
        _FRAME_KEY = re.compile(r'^.+_\d+x\d+x\d+_\d+$')
    return _FRAME_KEY.match(key) is not None


def parse_lmdb_key(key):
    """`{seq}_{n}x{h}x{w}_{i:04d}` -> (seq, (n_frames, h, w), frame)  (base_dataset.py:35-41)."""
    parts = key.split('_')
    seq, size, frm = '_'.join(parts[:-2]), parts[-2], int(parts[-1])
    return seq, tuple(int(v) for v in size.split('x')), frm


def make_key(seq, n_frm, h, w, i):
    return f'{seq}_{n_frm}x{h}x{w}_{i:04d}'            # scripts/create_lmdb.py:57


if __name__ == '__main__':      # python -m tecogan_pytorch_amd.data.lmdb_io <env dir>: list the database
    import sys
    r = LMDBReader(sys.argv[1])
    print(f'{r.path}: page size {r.psize}, {len(r)} entries, depth {r.meta["depth"]}, txn {r.meta["txnid"]}')
    for i, (k, v) in enumerate(r.items()):
        if i < 10:
            print(' ', k.decode('ascii', 'replace'), len(v), 'bytes')
