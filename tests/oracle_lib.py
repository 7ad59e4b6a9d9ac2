"""ctypes view of oracle/liboracle.so — the CHECKER.  Only tests/, smoke() and bench.py's cpu_baseline use it."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORA_MAX_MATCH = 64


class OraParam(C.Structure):
    _fields_ = [("maxResult", C.c_int), ("minHitLen", C.c_int), ("maxResultPerHitFactor", C.c_int),
                ("considerSecondaryHitLen", C.c_uint64), ("considerSecondaryScoreFactor", C.c_double),
                ("outputExpandedResult", C.c_int)]


class OraResult(C.Structure):
    _fields_ = [("score", C.c_uint64), ("secondaryScore", C.c_uint64), ("hitLength", C.c_int32), ("queryLength", C.c_int32),
                ("nmatch", C.c_int32), ("kind", C.c_int32 * ORA_MAX_MATCH), ("id", C.c_uint64 * ORA_MAX_MATCH),
                ("taxid", C.c_uint64 * ORA_MAX_MATCH),
                ("expanded", C.POINTER(C.c_uint64)), ("expOff", C.c_int32 * (ORA_MAX_MATCH + 1))]

    def expanded_ids(self, q):
        """--expand-taxid: the ORIGINAL tax ids promoted into match q (Classifier.hpp:821-839)"""
        if not self.expanded:
            return []
        return [int(self.expanded[j]) for j in range(self.expOff[q], self.expOff[q + 1])]


class OraCounters(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in ("bitrank", "bitaccess", "ftab", "sampled", "filter", "hits", "bs_calls",
                                          "extends", "lf_steps", "locates", "read_bases", "bitrank_locate",
                                          "bitaccess_locate", "ext_single_row", "ext_two_records")]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}

    def algorithmic_bytes(self):
        """24 B per bit-rank, 8 B per bit-access, 16 B per ftab lookup, 8 B per sampled-SA read,
        8 B per selectedSA filter probe, read bytes in, 32 B per hit out.  Whole Query."""
        return (24 * self.bitrank + 8 * self.bitaccess + 16 * self.ftab + 8 * self.sampled + 8 * self.filter
                + self.read_bases + 32 * self.hits)

    def search_bytes(self):
        """the share of algorithmic_bytes() spent in SearchForwardAndReverse (kernels k_search_chains + k_adjust_select)"""
        return (24 * (self.bitrank - self.bitrank_locate) + 8 * (self.bitaccess - self.bitaccess_locate) + 16 * self.ftab
                + self.read_bases + 32 * self.hits)

    def locate_bytes(self):
        """the share spent in BackwardToSampledSA (kernel k_locate)"""
        return 24 * self.bitrank_locate + 8 * self.bitaccess_locate + 8 * self.sampled + 8 * self.filter


ORA_HIT_DTYPE = np.dtype([("sp", "<u8"), ("ep", "<u8"), ("l", "<i4"), ("strand", "<i4"), ("offset", "<i4")], align=True)

_lib = None


def lib():
    global _lib
    if _lib is None:
        so = os.path.join(ORACLE_DIR, "liboracle.so")
        subprocess.run(["make", "-C", ORACLE_DIR, "oracle"], check=True, stdout=subprocess.DEVNULL)   # make does the dependency check
        L = C.CDLL(so)
        L.ora_index_load.restype = C.c_void_p
        L.ora_index_load.argtypes = [C.c_char_p, C.c_void_p]
        L.ora_index_free.argtypes = [C.c_void_p]
        L.ora_fm_rank.restype = C.c_uint64
        L.ora_fm_rank.argtypes = [C.c_void_p, C.c_char, C.c_uint64, C.c_int, C.c_void_p]
        L.ora_rb_access.restype = C.c_char
        L.ora_rb_access.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p]
        L.ora_fm_backward_search.restype = C.c_uint64
        L.ora_fm_backward_search.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
        L.ora_fm_backward_to_sampled_sa.restype = C.c_uint64
        L.ora_fm_backward_to_sampled_sa.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
        L.ora_query_hits.restype = C.c_size_t
        L.ora_query_hits.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_void_p, C.c_size_t]
        L.ora_classify_batch.argtypes = [C.c_void_p] + [C.c_void_p] * 4 + [C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.ora_format_result.restype = C.c_size_t
        L.ora_format_result.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_char_p, C.c_size_t]
        L.ora_dust_mask_inplace.argtypes = [C.c_char_p, C.c_size_t]
        L.ora_tsv_header.restype = C.c_char_p
        L.ora_tsv_header_for.restype = C.c_char_p
        L.ora_tsv_header_for.argtypes = [C.c_void_p]
        _lib = L
    return _lib


class OracleIndex:
    def __init__(self, prefix, max_result=1, min_hit_len=0, hitk_factor=40, expand=False):
        p = OraParam(max_result, min_hit_len, hitk_factor, 2000, 0.995, int(expand))
        self.param = p
        self._h = lib().ora_index_load(prefix.encode(), C.addressof(p))
        if not self._h:
            raise RuntimeError("oracle: cannot load " + prefix)
        # ora_index = {ora_fm fm; ...}: fm is the first member, bwt sits inside fm after 3 u64 + char (padded)
        self._fm = self._h

    def close(self):
        if self._h:
            lib().ora_index_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # --- primitives
    def fm_rank(self, ch, p, incl):
        return lib().ora_fm_rank(self._fm, ch if isinstance(ch, bytes) else ch.encode(), int(p), int(incl), None)

    def backward_search(self, s: bytes, m: int):
        sp, ep = C.c_uint64(0), C.c_uint64(0)
        l = lib().ora_fm_backward_search(self._fm, s, m, C.byref(sp), C.byref(ep), None)
        return int(l), int(sp.value), int(ep.value)

    def locate(self, row):
        l = C.c_uint64(0)
        v = lib().ora_fm_backward_to_sampled_sa(self._fm, int(row), C.byref(l), None)
        return int(v), int(l.value)

    def query_hits(self, r1: bytes, r2: bytes | None = None, cap=4096):
        buf = np.zeros(cap, dtype=ORA_HIT_DTYPE)
        n = lib().ora_query_hits(self._h, r1, r2, buf.ctypes.data, cap)
        assert n <= cap
        return buf[:n]

    # --- batch
    def classify(self, bases1, offs1, bases2=None, offs2=None, dust=False, threads=1, counters=False):
        n = len(offs1) - 1
        res = (OraResult * max(n, 1))()
        cnt = OraCounters()
        b1 = np.ascontiguousarray(bases1, dtype=np.uint8)
        o1 = np.ascontiguousarray(offs1, dtype=np.uint64)
        b2 = None if bases2 is None else np.ascontiguousarray(bases2, dtype=np.uint8)
        o2 = None if offs2 is None else np.ascontiguousarray(offs2, dtype=np.uint64)
        lib().ora_classify_batch(self._h, b1.ctypes.data, o1.ctypes.data, b2.ctypes.data if b2 is not None else None,
                                 o2.ctypes.data if o2 is not None else None, n, int(dust), threads, C.addressof(res), C.addressof(cnt))
        return (res, cnt) if counters else res

    def format(self, read_id: str, r) -> bytes:
        buf = C.create_string_buffer(1 << 16)
        n = lib().ora_format_result(self._h, read_id.encode(), C.addressof(r), buf, len(buf))
        return buf.raw[:n]

    def tsv(self, ids, results) -> bytes:
        return lib().ora_tsv_header_for(self._h) + b"".join(self.format(i, results[k]) for k, i in enumerate(ids))


def dust_mask(s: bytes) -> bytes:
    buf = C.create_string_buffer(s, len(s) + 1)
    lib().ora_dust_mask_inplace(buf, len(s))
    return buf.raw[:len(s)]


def read_fastx(path):
    """(ids, bases uint8, offsets uint64) — id = first word, trailing /1 /2 stripped (ReadFiles.hpp:82-90)."""
    ids, seqs = [], []
    with open(path, "rb") as f:
        lines = f.read().split(b"\n")
    i = 0
    while i < len(lines):
        ln = lines[i]
        if ln.startswith(b"@"):
            ids.append(ln[1:].split()[0]); seqs.append(lines[i + 1].rstrip(b"\r")); i += 4
        elif ln.startswith(b">"):
            ids.append(ln[1:].split()[0]); i += 1
            parts = []
            while i < len(lines) and not lines[i].startswith(b">"):
                parts.append(lines[i].rstrip(b"\r")); i += 1
            seqs.append(b"".join(parts))
        else:
            i += 1
    ids = [x[:-2] if x.endswith((b"/1", b"/2")) else x for x in ids]
    offs = np.zeros(len(seqs) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([len(s) for s in seqs])
    bases = np.frombuffer(b"".join(seqs), dtype=np.uint8).copy()
    return [x.decode() for x in ids], bases, offs
