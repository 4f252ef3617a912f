"""Reader for R `save()` files (`.rda` / `.RData`) holding numeric matrices -- the on-disk format next to the hot path.

The reference ships its example data as `data/nsclc.rda` (used by `vignettes/Fast-And-Versatile-NMF.Rmd:281-296` and
by `R/nnmf.R`'s examples): a bzip2-compressed `RDX2` stream in XDR (big-endian) serialisation, version 2, holding a
pairlist of named objects.  This module decodes the subset of R's serialisation needed for such data sets --
pairlists, symbols, character/real/integer/logical/string/generic vectors, attributes (`dim`, `dimnames`, `names`),
reference objects -- and returns `{name: RObject}`.  It is host-side glue, pure Python.

    objs = load_rda("nsclc.rda");  A = objs["nsclc"].matrix()      # (200, 100) float64, column-major data reshaped
"""
import bz2
import gzip
import lzma
import struct

import numpy as np

NILVALUE_SXP, REFSXP = 254, 255
SYMSXP, LISTSXP, CHARSXP, LGLSXP, INTSXP, REALSXP, STRSXP, VECSXP = 1, 2, 9, 10, 13, 14, 16, 19
NA_INT = -2147483648


class RObject:
    """A decoded R value: `.value` (numpy array, list, str or None) plus `.attributes` (dict)."""

    def __init__(self, value, attributes=None):
        self.value = value
        self.attributes = attributes or {}

    def matrix(self):
        """The value as a 2-D numpy array using the `dim` attribute (R stores column-major)."""
        dim = self.attributes.get("dim")
        if dim is None:
            raise ValueError("object has no dim attribute")
        d = [int(v) for v in dim.value]
        return np.asarray(self.value).reshape(d, order="F")

    def dimnames(self):
        dn = self.attributes.get("dimnames")
        if dn is None:
            return None
        return [None if e is None or e.value is None else list(e.value) for e in dn.value]


class _Reader:
    def __init__(self, data):
        self.b = data
        self.p = 0
        self.refs = []

    def take(self, n):
        if self.p + n > len(self.b):
            raise ValueError("truncated R data stream")
        out = self.b[self.p:self.p + n]
        self.p += n
        return out

    def int(self):
        return struct.unpack(">i", self.take(4))[0]

    def length(self):
        n = self.int()
        if n == -1:  # long vector: two more ints
            hi, lo = struct.unpack(">II", self.take(8))
            n = (hi << 32) | lo
        return n

    def item(self):
        flags = self.int()
        t = flags & 0xFF
        has_attr, has_tag = bool(flags & 0x200), bool(flags & 0x400)
        if t == NILVALUE_SXP:
            return None
        if t == REFSXP:
            idx = flags >> 8
            if idx == 0:
                idx = self.int()
            return self.refs[idx - 1]
        if t == SYMSXP:
            name = self.item()
            self.refs.append(name)
            return name
        if t == LISTSXP:
            # a pairlist node: (attributes)? (tag)? car cdr -- returned as a list of (tag, value)
            out = []
            while True:
                if has_attr:
                    self.item()
                tag = self.item() if has_tag else None
                car = self.item()
                out.append((tag.value if isinstance(tag, RObject) else tag, car))
                flags = self.int()
                t = flags & 0xFF
                has_attr, has_tag = bool(flags & 0x200), bool(flags & 0x400)
                if t == NILVALUE_SXP:
                    return out
                if t != LISTSXP:
                    raise ValueError("unsupported pairlist tail type %d" % t)
        if t == CHARSXP:
            n = self.int()
            return RObject(None if n == -1 else self.take(n).decode("utf-8", "replace"))
        if t in (LGLSXP, INTSXP):
            n = self.length()
            v = np.frombuffer(self.take(4 * n), dtype=">i4").astype(np.int32)
            val = v
        elif t == REALSXP:
            n = self.length()
            val = np.frombuffer(self.take(8 * n), dtype=">f8").astype(np.float64)
        elif t == STRSXP:
            n = self.length()
            val = [self.item().value for _ in range(n)]
        elif t == VECSXP:
            n = self.length()
            val = [self.item() for _ in range(n)]
        else:
            raise ValueError("unsupported R type %d in serialised data" % t)
        attrs = {}
        if has_attr:
            pl = self.item()
            for tag, v in (pl or []):
                attrs[tag] = v
        return RObject(val, attrs)


def _decompress(raw):
    if raw[:3] == b"BZh":
        return bz2.decompress(raw)
    if raw[:2] == b"\x1f\x8b":
        return gzip.decompress(raw)
    if raw[:6] == b"\xfd7zXZ\x00":
        return lzma.decompress(raw)
    return raw


def loads_rda(raw):
    """Decode the bytes of an R `save()` file; returns {object name: RObject}."""
    data = _decompress(raw)
    if data[:5] != b"RDX2\n":
        raise ValueError("not an RDX2 file (magic %r)" % data[:5])
    if data[5:7] != b"X\n":
        raise ValueError("only the XDR binary serialisation is supported (format %r)" % data[5:7])
    r = _Reader(data)
    r.p = 7
    version, _writer, _min_reader = r.int(), r.int(), r.int()
    if version != 2:
        raise ValueError("unsupported serialisation version %d" % version)
    top = r.item()
    if not isinstance(top, list):
        raise ValueError("top-level object of an .rda file must be a pairlist")
    return {name: value for name, value in top}


def load_rda(path):
    with open(path, "rb") as fh:
        return loads_rda(fh.read())
