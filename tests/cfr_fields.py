"""Field-level reader of a .1.cfr file following SURVEY.md Appendix A (test helper).  `_space` bookkeeping
fields are reported as 'SPACE' so that two writers can be compared on everything that loaders consume."""
import struct


def parse_1cfr(path, protein=False):
    """protein: FMIndex<Sequence_RunBlockOneTree> with end markers (centrifuger-build --protein)."""
    d = open(path, "rb").read()
    pos = [0]
    out = []

    def u64(name, space=False):
        v = struct.unpack_from("<Q", d, pos[0])[0]
        out.append((name, "SPACE" if space else v))
        pos[0] += 8
        return v

    def i32(name):
        v = struct.unpack_from("<i", d, pos[0])[0]
        out.append((name, v))
        pos[0] += 4
        return v

    def raw(name, n):
        out.append((name, d[pos[0]:pos[0] + n]))
        pos[0] += n

    def alpha(tag):
        u64(tag + ".space", True); i32(tag + ".method"); n = u64(tag + ".n")
        if n:
            raw(tag + ".list", n); raw(tag + ".code", 1024); raw(tag + ".len", 512)
        return n

    def bitvec(tag):
        u64(tag + ".space", True); n = u64(tag + ".n"); i32(tag + ".rb"); i32(tag + ".sb"); i32(tag + ".ss"); i32(tag + ".st")
        if n:
            raw(tag + ".B", (n + 63) // 64 * 8); u64(tag + ".rspace", True); wc = u64(tag + ".wc")
            raw(tag + ".R", 2 * ((wc + 7) // 8) * 8); u64(tag + ".sspace", True); u64(tag + ".sn"); i32(tag + ".sspeed")

    def wavelet(tag):
        u64(tag + ".space", True); u64(tag + ".n"); a = alpha(tag + ".alpha"); nc = i32(tag + ".nodecnt"); i32(tag + ".selspeed")
        if a:
            for k in range(nc):
                u64(f"{tag}.node{k}.prefix"); i32(f"{tag}.node{k}.plen"); i32(f"{tag}.node{k}.c0"); i32(f"{tag}.node{k}.c1")
                bitvec(f"{tag}.node{k}.v")

    u64("n"); u64("abits"); u64("firstISA"); raw("lastChr", 1)
    u64("rb.space", True); u64("rb.n"); sigma = alpha("rb.alpha"); u64("b"); u64("blockCnt")
    if protein:
        bitvec("useRunBlock")
        for k in range(sigma):
            bitvec(f"alphabetRB{k}")
        wavelet("compressed")
        alpha("alphabets"); alpha("plain"); raw("C", 8 * (sigma + 1))
    else:
        bitvec("useRunBlock"); wavelet("wave"); wavelet("runs")
        alpha("alphabets"); alpha("plain"); raw("C", 40)
    u64("aux.n"); i32("strategy"); i32("rate"); u64("sampleSize"); u64("width"); ps = u64("psize"); u64("adjSA0")
    u64("fsea.size"); l = i32("fsea.l"); fn = u64("fsea.n"); raw("fsea.W", (fn * l + 63) // 64 * 8)
    raw("ftab", ps * 16); u64("maxLcp"); sc = u64("selCnt"); i32("filterRate"); raw("sel", sc * 16); raw("hasEnd", 1)
    if protein:
        u64("end.size"); l = i32("end.l"); fn = u64("end.n"); raw("end.W", (fn * l + 63) // 64 * 8)
    assert pos[0] == len(d), (pos[0], len(d))
    return out
