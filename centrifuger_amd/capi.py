"""ctypes binding of libcfr_hip.so (include/cfr_hip.h).  Thin: numpy arrays in, numpy arrays out.

The library is the product; there is no Python/CPU fallback here - if the shared object is
missing or no HIP device is usable, calls raise CfrError.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcfr_hip.so")

CFR_OK, CFR_ERR_IO, CFR_ERR_FORMAT, CFR_ERR_NO_DEVICE, CFR_ERR_HIP, CFR_ERR_ARG, CFR_ERR_CAPACITY, CFR_ERR_BUSY = range(8)


class CfrError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__(f"cfr status {status}: {msg}")
        self.status = status


class Params(C.Structure):
    _fields_ = [("max_result", C.c_int32), ("min_hit_len", C.c_int32), ("max_result_per_hit_factor", C.c_int32),
                ("output_expanded", C.c_int32), ("consider_secondary_hit_len", C.c_uint64),
                ("consider_secondary_score_factor", C.c_double)]


class IndexInfo(C.Structure):
    _fields_ = [("n", C.c_uint64), ("first_isa", C.c_uint64), ("block_size", C.c_uint64), ("precompute_width", C.c_uint64),
                ("sample_rate", C.c_uint64), ("selected_cnt", C.c_uint64), ("seq_cnt", C.c_uint64), ("node_cnt", C.c_uint64),
                ("min_hit_len", C.c_int32), ("last_chr", C.c_char), ("is_protein", C.c_uint8), ("pad", C.c_char * 2), ("device_bytes", C.c_uint64)]


class BatchStats(C.Structure):
    _fields_ = [("pack_ms", C.c_float), ("search_ms", C.c_float), ("adjust_ms", C.c_float), ("rows_ms", C.c_float),
                ("locate_ms", C.c_float), ("tail_ms", C.c_float), ("total_ms", C.c_float),
                ("n_chains", C.c_uint64), ("n_hits", C.c_uint64), ("n_rows", C.c_uint64)]


HIT_DTYPE = np.dtype([("sp", "<u8"), ("ep", "<u8"), ("l", "<i4"), ("strand", "<i4"), ("offset", "<i4"), ("pad", "<i4")])
RESULT_DTYPE = np.dtype([("score", "<u8"), ("secondary_score", "<u8"), ("hit_length", "<i4"), ("query_length", "<i4"),
                         ("n_match", "<i4"), ("pad", "<i4"), ("match_begin", "<u8")])
MATCH_DTYPE = np.dtype([("id", "<u8"), ("taxid", "<u8"), ("kind", "<i4"), ("pad", "<i4")])
SPAN_DTYPE = np.dtype([("begin", "<u8"), ("count", "<u8")])      # cfr_span: the --expand-taxid list of a match slot
# the narrow layout of cfr_classify_batch_resident_compact (20 + 12 bytes)
RESULT_COMPACT_DTYPE = np.dtype([("score", "<u4"), ("secondary_score", "<u4"), ("hit_length", "<u4"), ("query_length", "<u4"),
                                 ("n_match", "u1"), ("flags", "u1"), ("pad", "<u2")])
MATCH_COMPACT_DTYPE = np.dtype([("id_kind", "<u4"), ("taxid_lo", "<u4"), ("taxid_hi", "<u4")])
COMPACT_WIDE = 1


def expand_compact(cres, cmatch, max_result, wide=None):
    """cfr_result_compact / cfr_match_compact arrays -> the wide arrays (RESULT_DTYPE, MATCH_DTYPE) with match_begin = i * max_result.
    wide: DeviceIndex.compact_wide() of the same call - the reads flagged CFR_COMPACT_WIDE (values that do not fit the narrow
    layout) are patched in from it; without it a flagged read raises."""
    flagged = (cres["flags"] & COMPACT_WIDE) != 0
    if flagged.any() and wide is None:
        raise ValueError("a read of the batch does not fit the compact result layout (CFR_COMPACT_WIDE): pass DeviceIndex.compact_wide()")
    n = len(cres)
    res = np.zeros(n, dtype=RESULT_DTYPE)
    for f in ("score", "secondary_score", "hit_length", "query_length", "n_match"):
        res[f] = cres[f]
    res["match_begin"] = np.arange(n, dtype=np.uint64) * np.uint64(max_result)
    mat = np.zeros(len(cmatch), dtype=MATCH_DTYPE)
    mat["id"] = cmatch["id_kind"] & np.uint32(0x7fffffff)
    mat["kind"] = (cmatch["id_kind"] >> np.uint32(31)).astype(np.int32)
    mat["taxid"] = cmatch["taxid_lo"].astype(np.uint64) | (cmatch["taxid_hi"].astype(np.uint64) << np.uint64(32))
    if flagged.any():
        widx, wres, wmat = wide
        assert set(np.nonzero(flagged)[0].tolist()) <= set(int(x) for x in widx), "flagged reads missing from the wide side list"
        for j, i in enumerate(widx):
            i = int(i)
            if i >= n:
                continue
            mb = res["match_begin"][i]
            res[i] = wres[j]
            res["match_begin"][i] = mb
            src = int(wres[j]["match_begin"])
            mat[int(mb):int(mb) + max_result] = wmat[src:src + max_result]
    return res, mat

EXPORTS = [
    "cfr_classify_batch_resident_compact", "cfr_params_default", "cfr_last_error", "cfr_version", "cfr_index_open", "cfr_index_destroy", "cfr_index_get_info",
    "cfr_device_count", "cfr_device_index_create", "cfr_device_index_create_ex", "cfr_device_options_default",
    "cfr_device_index_destroy", "cfr_device_index_get_info", "cfr_device_index_set_dust", "cfr_dust_mask_device",
    "cfr_dust_mask_batch_literal",
    "cfr_selfcheck_tables", "cfr_build_index", "cfr_build_options_default",
    "cfr_rank_batch", "cfr_backward_search_batch", "cfr_locate_rows", "cfr_search_batch", "cfr_classify_batch",
    "cfr_classify_batch_resident", "cfr_last_batch_stats", "cfr_classify_from_hits", "cfr_dust_mask_batch",
    "cfr_format_tsv", "cfr_tsv_header", "cfr_host_alloc", "cfr_host_free",
    "cfr_classify_batch_submit", "cfr_classify_batch_wait", "cfr_compact_wide_reads", "cfr_index_digest", "cfr_index_mapped_bytes", "cfr_pack_reads", "cfr_classify_batch_packed",
    "cfr_classify_batch_expanded", "cfr_classify_from_hits_expanded", "cfr_format_tsv_expanded", "cfr_tsv_header_expanded",
]

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise CfrError(-1, f"{LIB_PATH} not built (run `python -c 'import __graft_entry__ as g; g.build()'` or make -C centrifuger_amd/csrc)")
        L = C.CDLL(LIB_PATH)
        L.cfr_last_error.restype = C.c_char_p
        L.cfr_version.restype = C.c_char_p
        L.cfr_tsv_header.restype = C.c_char_p
        L.cfr_format_tsv.restype = C.c_size_t
        L.cfr_format_tsv_expanded.restype = C.c_size_t
        L.cfr_tsv_header_expanded.restype = C.c_char_p
        L.cfr_host_alloc.restype = C.c_void_p
        L.cfr_host_alloc.argtypes = [C.c_size_t]
        L.cfr_host_free.argtypes = [C.c_void_p]
        for name in EXPORTS:
            if name not in ("cfr_last_error", "cfr_version", "cfr_tsv_header", "cfr_format_tsv", "cfr_index_destroy", "cfr_format_tsv_expanded", "cfr_tsv_header_expanded",
                            "cfr_device_index_destroy", "cfr_params_default", "cfr_host_alloc", "cfr_host_free", "cfr_build_options_default"):
                getattr(L, name).restype = C.c_int
        _lib = L
    return _lib


def _check(status):
    if status != CFR_OK:
        raise CfrError(status, lib().cfr_last_error().decode())


def _p(a, t=C.c_void_p):
    return None if a is None else a.ctypes.data_as(t)


def default_params(**kw) -> Params:
    p = Params()
    lib().cfr_params_default(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def _u8(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.uint8)


def _u64(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.uint64)


class BuildInput(C.Structure):
    _fields_ = [("n_seqs", C.c_uint64), ("seq_names", C.POINTER(C.c_char_p)), ("seq_taxids", C.c_void_p), ("seq_lens", C.c_void_p),
                ("text", C.c_void_p), ("n_nodes", C.c_uint64), ("node_taxid", C.c_void_p), ("node_parent", C.c_void_p),
                ("node_rank", C.POINTER(C.c_char_p)), ("n_names", C.c_uint64), ("name_taxid", C.c_void_p), ("name_text", C.POINTER(C.c_char_p)),
                ("n_genomes", C.c_uint64), ("genome_seq", C.c_void_p), ("genome_lens", C.c_void_p), ("n_extra", C.c_uint64),
                ("n_present_taxids", C.c_uint64), ("present_taxids", C.c_void_p)]


class BuildOptions(C.Structure):
    _fields_ = [("ftab_chars", C.c_int32), ("offrate", C.c_int32), ("device", C.c_int32), ("threads", C.c_int32), ("rbbwt_b", C.c_uint64),
                ("verbose", C.c_int32), ("protein", C.c_int32)]


class BuildReport(C.Structure):
    _fields_ = [("n", C.c_uint64), ("block_size", C.c_uint64), ("first_isa", C.c_uint64), ("seconds_sa", C.c_double),
                ("seconds_total", C.c_double), ("rounds", C.c_int32), ("pad", C.c_int32)]


def build_index(names, taxids, seqs, nodes, tax_names, out_prefix, ftab_chars=10, offrate=4, rbbwt_b=0, device=0, threads=0, verbose=False,
                genome_seq=None, n_extra=0, protein=False):
    """cfr_build_index: the native writer (suffix array on the MI355X).  seqs: list of np.uint8 ASCII arrays, or one
    concatenated array together with the lengths given as (text, lens).  genome_seq: sequence id (index into names) of every
    genome of the text in text order when that is not 0, 1, 2, ... over all names; the last n_extra names have no tax id.
    protein: seqs are amino-acid strings (letters of ARNDCEQGHILKMFPSTWYV), the index is the one centrifuger-build --protein writes."""
    if isinstance(seqs, tuple):
        text, lens = seqs
        text = np.ascontiguousarray(text, dtype=np.uint8)
        lens = np.ascontiguousarray(lens, dtype=np.uint64)
    else:
        lens = np.array([len(x) for x in seqs], dtype=np.uint64)
        text = np.ascontiguousarray(np.concatenate(seqs), dtype=np.uint8)

    def strs(items):
        arr = (C.c_char_p * len(items))()
        arr[:] = [x.encode() if isinstance(x, str) else x for x in items]
        return arr
    inp = BuildInput()
    keep = [strs(list(names)), _u64(np.array(taxids, dtype=np.uint64)), lens, text,
            _u64(np.array([t for t, _, _ in nodes], dtype=np.uint64)), _u64(np.array([p for _, p, _ in nodes], dtype=np.uint64)),
            strs([r for _, _, r in nodes]), _u64(np.array([t for t, _ in tax_names], dtype=np.uint64)), strs([x for _, x in tax_names])]
    inp.n_seqs = len(names)
    inp.seq_names = keep[0]
    inp.seq_taxids = _p(keep[1]); inp.seq_lens = _p(keep[2]); inp.text = _p(keep[3])
    inp.n_nodes = len(nodes); inp.node_taxid = _p(keep[4]); inp.node_parent = _p(keep[5]); inp.node_rank = keep[6]
    inp.n_names = len(tax_names); inp.name_taxid = _p(keep[7]); inp.name_text = keep[8]
    if genome_seq is not None:
        keep.append(_u64(np.array(genome_seq, dtype=np.uint64)))
        inp.n_genomes = len(keep[-1]); inp.genome_seq = _p(keep[-1]); inp.genome_lens = _p(keep[2]); inp.seq_lens = None
        inp.n_extra = n_extra
    opt = BuildOptions()
    lib().cfr_build_options_default(C.byref(opt))
    opt.ftab_chars, opt.offrate, opt.device, opt.threads, opt.rbbwt_b, opt.verbose = ftab_chars, offrate, device, threads, rbbwt_b, int(verbose)
    opt.protein = int(protein)
    rep = BuildReport()
    _check(lib().cfr_build_index(C.byref(inp), C.byref(opt), out_prefix.encode(), C.byref(rep)))
    return {"n": rep.n, "b": rep.block_size, "first_isa": rep.first_isa, "seconds_sa": rep.seconds_sa, "seconds_total": rep.seconds_total,
            "rounds": rep.rounds}


class Index:
    """cfr_index: host copy of <prefix>.{1,2,4}.cfr (Classifier::Init, Classifier.hpp:902-947)."""

    def __init__(self, prefix: str, params: Params | None = None):
        self._h = C.c_void_p()
        self.params = params if params is not None else default_params()
        _check(lib().cfr_index_open(prefix.encode(), C.byref(self.params), C.byref(self._h)))

    def info(self) -> IndexInfo:
        info = IndexInfo()
        _check(lib().cfr_index_get_info(self._h, C.byref(info)))
        return info

    def digest(self) -> int:
        """cfr_index_digest: 64-bit digest of everything the parser produced"""
        d = C.c_uint64(0)
        _check(lib().cfr_index_digest(self._h, C.byref(d)))
        return d.value

    def mapped_bytes(self):
        """cfr_index_mapped_bytes: (bytes of bit strings read from the file's mapping, bytes held as private copies)"""
        m, c = C.c_uint64(0), C.c_uint64(0)
        _check(lib().cfr_index_mapped_bytes(self._h, C.byref(m), C.byref(c)))
        return m.value, c.value

    def close(self):
        if self._h:
            lib().cfr_index_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def format_tsv(self, read_id: str, result, matches) -> bytes:
        r = np.ascontiguousarray(result).reshape(1)
        buf = C.create_string_buffer(1 << 16)
        n = lib().cfr_format_tsv(self._h, read_id.encode(), _p(r), _p(matches), buf, C.c_size_t(len(buf)))
        if n >= len(buf):
            buf = C.create_string_buffer(n + 1)
            lib().cfr_format_tsv(self._h, read_id.encode(), _p(r), _p(matches), buf, C.c_size_t(len(buf)))
        return buf.raw[:n]

    def format_tsv_expanded(self, read_id: str, result, matches, spans, ids) -> bytes:
        """cfr_format_tsv_expanded: the row(s) of one read with the expandedTaxIDs column"""
        r = np.ascontiguousarray(result).reshape(1)
        buf = C.create_string_buffer(1 << 16)
        n = lib().cfr_format_tsv_expanded(self._h, read_id.encode(), _p(r), _p(matches), _p(spans), _p(ids), buf, C.c_size_t(len(buf)))
        if n >= len(buf):
            buf = C.create_string_buffer(n + 1)
            lib().cfr_format_tsv_expanded(self._h, read_id.encode(), _p(r), _p(matches), _p(spans), _p(ids), buf, C.c_size_t(len(buf)))
        return buf.raw[:n]

    def classify_from_hits_expanded(self, hits, hit_begin, row_begin, row_vals, query_len, threads=1):
        """cfr_classify_from_hits_expanded -> (results, matches, spans, ids)"""
        n = len(hit_begin) - 1
        hits = np.ascontiguousarray(hits, dtype=HIT_DTYPE)
        hit_begin, row_begin, row_vals = _u64(hit_begin), _u64(row_begin), _u64(row_vals)
        query_len = np.ascontiguousarray(query_len, dtype=np.int32)
        results = np.zeros(n, dtype=RESULT_DTYPE)
        cap, icap = max(16, 4 * n), max(16, 4 * n)
        while True:
            matches = np.zeros(cap, dtype=MATCH_DTYPE)
            spans = np.zeros(cap, dtype=SPAN_DTYPE)
            ids = np.zeros(icap, dtype=np.uint64)
            nm, ni = C.c_size_t(0), C.c_size_t(0)
            st = lib().cfr_classify_from_hits_expanded(self._h, _p(hits), _p(hit_begin), _p(row_begin), _p(row_vals), _p(query_len),
                                                       C.c_size_t(n), C.c_int(threads), _p(results), _p(matches), _p(spans), C.c_size_t(cap),
                                                       C.byref(nm), _p(ids), C.c_size_t(icap), C.byref(ni))
            if st == CFR_ERR_CAPACITY:
                cap, icap = max(cap, int(nm.value) + 16), max(icap, int(ni.value) + 16)
                continue
            _check(st)
            return results, matches[:nm.value], spans[:nm.value], ids[:ni.value]

    def classify_from_hits(self, hits, hit_begin, row_begin, row_vals, query_len, threads=1):
        n = len(hit_begin) - 1
        hits = np.ascontiguousarray(hits, dtype=HIT_DTYPE)
        hit_begin, row_begin, row_vals = _u64(hit_begin), _u64(row_begin), _u64(row_vals)
        query_len = np.ascontiguousarray(query_len, dtype=np.int32)
        results = np.zeros(n, dtype=RESULT_DTYPE)
        cap = max(16, 4 * n)
        while True:
            matches = np.zeros(cap, dtype=MATCH_DTYPE)
            nm = C.c_size_t(0)
            st = lib().cfr_classify_from_hits(self._h, _p(hits), _p(hit_begin), _p(row_begin), _p(row_vals), _p(query_len),
                                              C.c_size_t(n), C.c_int(threads), _p(results), _p(matches), C.c_size_t(cap), C.byref(nm))
            if st == CFR_ERR_CAPACITY:
                cap = int(nm.value) + 16
                continue
            _check(st)
            return results, matches[:nm.value]


class DeviceOptions(C.Structure):
    """cfr_device_options (include/cfr_hip.h)."""
    _fields_ = [("profile", C.c_int32), ("ftabx_width", C.c_int32), ("text_mode", C.c_int32), ("run_block_layout", C.c_int32),
                ("loc_memo_gb", C.c_double), ("sub_batch", C.c_uint64)]


PROFILE_THROUGHPUT, PROFILE_FAST_LOAD = 0, 1


def default_device_options(**kw) -> DeviceOptions:
    o = DeviceOptions()
    lib().cfr_device_options_default(C.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    return o


class DeviceIndex:
    """cfr_dev_index: the flat index image in one GPU's HBM + the batch entry points."""

    def __init__(self, index: Index, device: int = 0, options: DeviceOptions | None = None):
        self.index = index
        self._d = C.c_void_p()
        if options is None:
            _check(lib().cfr_device_index_create(index._h, C.c_int(device), C.byref(self._d)))
        else:
            _check(lib().cfr_device_index_create_ex(index._h, C.c_int(device), C.byref(options), C.byref(self._d)))

    def info(self) -> IndexInfo:
        info = IndexInfo()
        _check(lib().cfr_device_index_get_info(self._d, C.byref(info)))
        return info

    def set_dust(self, on: bool):
        """SDUST on the device before every following classify call (the caller hands over unmasked reads)."""
        _check(lib().cfr_device_index_set_dust(self._d, C.c_int(1 if on else 0)))

    def dust_mask(self, bases, offsets):
        """The device SDUST scan alone: masks `bases` (uint8 numpy, host) in place."""
        assert bases.dtype == np.uint8 and bases.flags["C_CONTIGUOUS"] and bases.flags["WRITEABLE"]
        offsets = _u64(offsets)
        _check(lib().cfr_dust_mask_device(self._d, _p(bases), _p(offsets), C.c_size_t(len(offsets) - 1)))

    def close(self):
        if self._d:
            lib().cfr_device_index_destroy(self._d)
            self._d = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- probes
    def rank(self, chars, pos, inclusive):
        chars = np.ascontiguousarray(np.frombuffer(chars, dtype=np.uint8) if isinstance(chars, (bytes, bytearray)) else chars, dtype=np.uint8)
        pos, inclusive = _u64(pos), _u8(inclusive)
        n = len(pos)
        out_r = np.zeros(n, dtype=np.uint64)
        out_a = np.zeros(n, dtype=np.uint8)
        _check(lib().cfr_rank_batch(self._d, _p(chars), _p(pos), _p(inclusive), C.c_size_t(n), _p(out_r), _p(out_a)))
        return out_r, out_a

    def backward_search(self, bases, offsets, m):
        bases, offsets = _u8(bases), _u64(offsets)
        m = np.ascontiguousarray(m, dtype=np.uint32)
        n = len(m)
        l, sp, ep = (np.zeros(n, dtype=np.uint64) for _ in range(3))
        _check(lib().cfr_backward_search_batch(self._d, _p(bases), _p(offsets), _p(m), C.c_size_t(n), _p(l), _p(sp), _p(ep)))
        return l, sp, ep

    def locate(self, rows):
        rows = _u64(rows)
        n = len(rows)
        val = np.zeros(n, dtype=np.uint64)
        steps = np.zeros(n, dtype=np.uint32)
        _check(lib().cfr_locate_rows(self._d, _p(rows), C.c_size_t(n), _p(val), _p(steps)))
        return val, steps

    def selfcheck(self):
        """dict of the derived-table self-check (all bad_* must be 0)"""
        out = np.zeros(6, dtype=np.uint64)
        _check(lib().cfr_selfcheck_tables(self._d, _p(out)))
        return {"bad_sa_isa": int(out[0]), "bad_text": int(out[1]), "bad_lf": int(out[2]), "bad_memo": int(out[3]),
                "text_tables": bool(out[4]), "memo": int(out[5])}

    # ---- the path
    def search(self, bases1, offsets1, bases2=None, offsets2=None):
        bases1, offsets1, bases2, offsets2 = _u8(bases1), _u64(offsets1), _u8(bases2), _u64(offsets2)
        n = len(offsets1) - 1
        hit_begin = np.zeros(n + 1, dtype=np.uint64)
        cap = max(64, 16 * n)
        while True:
            hits = np.zeros(cap, dtype=HIT_DTYPE)
            st = lib().cfr_search_batch(self._d, _p(bases1), _p(offsets1), _p(bases2), _p(offsets2), C.c_size_t(n),
                                        _p(hits), C.c_size_t(cap), _p(hit_begin))
            if st == CFR_ERR_CAPACITY:
                cap = int(hit_begin[n]) + 16
                continue
            _check(st)
            return hits[:int(hit_begin[n])], hit_begin

    def classify(self, bases1, offsets1, bases2=None, offsets2=None, results=None, matches=None):
        """Host buffers in, host buffers out (cfr_classify_batch).  results / matches: caller-provided arrays (e.g. PinnedArray
        views; matches must hold max_result slots per read) instead of fresh numpy arrays."""
        bases1, offsets1, bases2, offsets2 = _u8(bases1), _u64(offsets1), _u8(bases2), _u64(offsets2)
        n = len(offsets1) - 1
        if results is not None and matches is not None:
            nm = C.c_size_t(0)
            _check(lib().cfr_classify_batch(self._d, _p(bases1), _p(offsets1), _p(bases2), _p(offsets2), C.c_size_t(n),
                                            _p(results), _p(matches), C.c_size_t(len(matches)), C.byref(nm)))
            return results, matches[:nm.value]
        results = np.zeros(n, dtype=RESULT_DTYPE)
        cap = max(16, max(1, self.index.params.max_result) * n)
        while True:
            matches = np.zeros(cap, dtype=MATCH_DTYPE)
            nm = C.c_size_t(0)
            st = lib().cfr_classify_batch(self._d, _p(bases1), _p(offsets1), _p(bases2), _p(offsets2), C.c_size_t(n),
                                          _p(results), _p(matches), C.c_size_t(cap), C.byref(nm))
            if st == CFR_ERR_CAPACITY:
                cap = int(nm.value) + 16
                continue
            _check(st)
            return results, matches[:nm.value]

    def classify_expanded(self, bases1, offsets1, bases2=None, offsets2=None, ids_cap=None):
        """cfr_classify_batch_expanded (the index must have been opened with output_expanded=1) -> (results, matches, spans, ids)"""
        bases1, offsets1, bases2, offsets2 = _u8(bases1), _u64(offsets1), _u8(bases2), _u64(offsets2)
        n = len(offsets1) - 1
        results = np.zeros(n, dtype=RESULT_DTYPE)
        cap = max(16, max(1, self.index.params.max_result) * n)
        icap = ids_cap if ids_cap is not None else max(16, 4 * n)
        while True:
            matches = np.zeros(cap, dtype=MATCH_DTYPE)
            spans = np.zeros(cap, dtype=SPAN_DTYPE)
            ids = np.zeros(max(icap, 1), dtype=np.uint64)
            nm, ni = C.c_size_t(0), C.c_size_t(0)
            st = lib().cfr_classify_batch_expanded(self._d, _p(bases1), _p(offsets1), _p(bases2), _p(offsets2), C.c_size_t(n), _p(results), _p(matches),
                                                   _p(spans), C.c_size_t(cap), C.byref(nm), _p(ids), C.c_size_t(icap), C.byref(ni))
            if st == CFR_ERR_CAPACITY:
                cap, icap = max(cap, int(nm.value) + 16), max(icap, int(ni.value))
                continue
            _check(st)
            return results, matches[:nm.value], spans[:nm.value], ids[:ni.value]

    def classify_packed(self, packed1, offsets1, packed2=None, offsets2=None, results=None, matches=None):
        """cfr_classify_batch_packed: the bases as packed blocks (pack_reads) instead of ASCII; same results as classify()"""
        packed1, offsets1, packed2, offsets2 = _u64(packed1), _u64(offsets1), _u64(packed2), _u64(offsets2)
        n = len(offsets1) - 1
        if results is None:
            results = np.zeros(n, dtype=RESULT_DTYPE)
        if matches is None:
            matches = np.zeros(max(16, max(1, self.index.params.max_result) * n), dtype=MATCH_DTYPE)
        while True:
            nm = C.c_size_t(0)
            st = lib().cfr_classify_batch_packed(self._d, _p(packed1), _p(offsets1), _p(packed2), _p(offsets2), C.c_size_t(n),
                                                 _p(results), _p(matches), C.c_size_t(len(matches)), C.byref(nm))
            if st == CFR_ERR_CAPACITY:
                matches = np.zeros(int(nm.value) + 16, dtype=MATCH_DTYPE)
                continue
            _check(st)
            return results, matches[:nm.value]

    def compact_wide(self):
        """cfr_compact_wide_reads: (read_index, results, matches) of the reads the last compact call flagged CFR_COMPACT_WIDE"""
        n = C.c_size_t(0)
        pi, pr, pm = C.c_void_p(), C.c_void_p(), C.c_void_p()
        _check(lib().cfr_compact_wide_reads(self._d, C.byref(n), C.byref(pi), C.byref(pr), C.byref(pm)))
        k = max(1, self.index.params.max_result)
        if n.value == 0:
            return np.zeros(0, dtype=np.uint32), np.zeros(0, dtype=RESULT_DTYPE), np.zeros(0, dtype=MATCH_DTYPE)
        idx = np.ctypeslib.as_array(C.cast(pi, C.POINTER(C.c_uint32)), shape=(n.value,)).copy()
        res = np.frombuffer(C.string_at(pr, n.value * RESULT_DTYPE.itemsize), dtype=RESULT_DTYPE).copy()
        mat = np.frombuffer(C.string_at(pm, n.value * k * MATCH_DTYPE.itemsize), dtype=MATCH_DTYPE).copy()
        return idx, res, mat

    def submit(self, bases1, offsets1, bases2=None, offsets2=None, results=None, matches=None):
        """cfr_classify_batch_submit: queue a batch, return a ticket object for wait().  The arrays are kept alive by the ticket."""
        bases1, offsets1, bases2, offsets2 = _u8(bases1), _u64(offsets1), _u8(bases2), _u64(offsets2)
        n = len(offsets1) - 1
        if results is None:
            results = np.zeros(n, dtype=RESULT_DTYPE)
        if matches is None:
            matches = np.zeros(max(16, max(1, self.index.params.max_result) * n), dtype=MATCH_DTYPE)
        t = C.c_uint64(0)
        _check(lib().cfr_classify_batch_submit(self._d, _p(bases1), _p(offsets1), _p(bases2), _p(offsets2), C.c_size_t(n),
                                               _p(results), _p(matches), C.c_size_t(len(matches)), C.byref(t)))
        return {"ticket": t.value, "keep": (bases1, offsets1, bases2, offsets2), "results": results, "matches": matches}

    def wait(self, job):
        """cfr_classify_batch_wait: the (results, matches) of a submitted batch"""
        nm = C.c_size_t(0)
        _check(lib().cfr_classify_batch_wait(self._d, C.c_uint64(job["ticket"]), C.byref(nm)))
        return job["results"], job["matches"][:nm.value]

    def classify_resident(self, d_bases1: int, d_offsets1: int, n: int, total1: int, d_bases2: int = 0, d_offsets2: int = 0,
                          total2: int = 0, results=None, matches=None):
        """Device pointers in (ints), host numpy out.  Caller must have synchronised the producer stream."""
        if results is None:
            results = np.zeros(n, dtype=RESULT_DTYPE)
        if matches is None:
            matches = np.zeros(max(16, max(1, self.index.params.max_result) * n), dtype=MATCH_DTYPE)
        nm = C.c_size_t(0)
        st = lib().cfr_classify_batch_resident(self._d, C.c_void_p(d_bases1), C.c_void_p(d_offsets1),
                                               C.c_void_p(d_bases2 or None), C.c_void_p(d_offsets2 or None), C.c_size_t(n),
                                               C.c_uint64(total1), C.c_uint64(total2), _p(results), _p(matches),
                                               C.c_size_t(len(matches)), C.byref(nm))
        _check(st)
        return results, matches[:nm.value]

    def classify_resident_compact(self, d_bases1: int, d_offsets1: int, n: int, total1: int, d_bases2: int = 0, d_offsets2: int = 0,
                                  total2: int = 0, results=None, matches=None):
        """classify_resident with the narrow result layout (RESULT_COMPACT_DTYPE / MATCH_COMPACT_DTYPE arrays; max_result > 0);
        expand_compact() gives the wide arrays."""
        k = max(1, self.index.params.max_result)
        if results is None:
            results = np.zeros(n, dtype=RESULT_COMPACT_DTYPE)
        if matches is None:
            matches = np.zeros(max(16, k * n), dtype=MATCH_COMPACT_DTYPE)
        nm = C.c_size_t(0)
        st = lib().cfr_classify_batch_resident_compact(self._d, C.c_void_p(d_bases1), C.c_void_p(d_offsets1),
                                                       C.c_void_p(d_bases2 or None), C.c_void_p(d_offsets2 or None), C.c_size_t(n),
                                                       C.c_uint64(total1), C.c_uint64(total2), _p(results), _p(matches),
                                                       C.c_size_t(len(matches)), C.byref(nm))
        _check(st)
        return results, matches[:nm.value]

    def last_stats(self) -> BatchStats:
        st = BatchStats()
        _check(lib().cfr_last_batch_stats(self._d, C.byref(st)))
        return st


def device_count() -> int:
    c = C.c_int(0)
    st = lib().cfr_device_count(C.byref(c))
    return c.value if st == CFR_OK else 0


def dust_mask(bases, offsets, threads=1, literal=False):
    """In-place SDUST masking of a flat read buffer (cfr_dust_mask_batch; literal=True: the reference's own data structure)."""
    assert bases.dtype == np.uint8 and bases.flags["C_CONTIGUOUS"] and bases.flags["WRITEABLE"]
    offsets = _u64(offsets)
    fn = lib().cfr_dust_mask_batch_literal if literal else lib().cfr_dust_mask_batch
    _check(fn(_p(bases), _p(offsets), C.c_size_t(len(offsets) - 1), C.c_int(threads)))
    return bases


def pack_reads(bases, threads=1, out=None):
    """cfr_pack_reads: uint64 blocks of 16 characters (2-bit codes + validity bits) of a flat ASCII read buffer"""
    bases = _u8(bases)
    nblk = (len(bases) + 15) // 16
    if out is None:
        out = np.empty(max(nblk, 1), dtype=np.uint64)
    _check(lib().cfr_pack_reads(_p(bases), C.c_uint64(len(bases)), C.c_int(threads), _p(out)))
    return out[:nblk]


def tsv_header() -> bytes:
    return lib().cfr_tsv_header()


class PinnedArray:
    """numpy view over cfr_host_alloc memory (pinned: D2H lands at PCIe rate)."""

    def __init__(self, count: int, dtype):
        self.dtype = np.dtype(dtype)
        self.nbytes = max(16, count * self.dtype.itemsize)
        self._p = lib().cfr_host_alloc(C.c_size_t(self.nbytes))
        if not self._p:
            raise CfrError(-1, "cfr_host_alloc failed")
        buf = (C.c_uint8 * self.nbytes).from_address(self._p)
        self.array = np.frombuffer(buf, dtype=self.dtype, count=count)

    def free(self):
        if self._p:
            self.array = None
            lib().cfr_host_free(C.c_void_p(self._p))
            self._p = None
