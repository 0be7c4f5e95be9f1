"""Python restatement of the reference's HOST-side semantics, for tests only.

Follows /root/reference/src/fastaIO.h:12-148 (ingest), src/main.cu:110-206 (chunk layout),
src/main.cu:270-305 (which chunks reach the output file).  Counting itself is the C oracle.
"""
import numpy as np

from . import oracle_lib as orc

_CODE = np.full(256, -1, np.int8)
for _c, _v in (("a", 0), ("A", 0), ("c", 1), ("C", 1), ("g", 2), ("G", 2), ("t", 3), ("T", 3)):
    _CODE[ord(_c)] = _v


def read_fasta_compat(raw: bytes):
    """ReadFasta + ReadFASTASequences (fastaIO.h:24-71,105-148) -> list of int8 code arrays.

    A record starts at a line whose first char is '>' (fastaIO.h:40); every other line,
    INCLUDING its newline, is appended to the current record (fastaIO.h:49-66);
    len = strlen(read) - 1 (fastaIO.h:53,65) drops the last char; chars are mapped by the
    switch at fastaIO.h:121-140, so embedded newlines become -1.
    """
    recs, cur = [], None
    pos = 0
    while pos < len(raw):
        nl = raw.find(b"\n", pos)
        line = raw[pos:] if nl < 0 else raw[pos:nl + 1]
        pos = len(raw) if nl < 0 else nl + 1
        if line[:1] == b">":
            cur = bytearray()
            recs.append(cur)
        elif cur is not None:
            cur.extend(line)
        else:
            raise ValueError("sequence line before the first header (UB in the reference)")
    out = []
    for r in recs:
        if len(r) == 0:
            raise ValueError("header without sequence line (UB in the reference, fastaIO.h:51-53)")
        codes = _CODE[np.frombuffer(bytes(r[:-1]), np.uint8)]
        out.append(codes)
    return out


def flatten(reads):
    """ProcessData (fastaIO.h:74-102): codes + -1 terminator per read; start/length tables."""
    nS = len(reads)
    length = np.array([len(r) for r in reads], np.int32)
    start = np.zeros(nS, np.int64)
    if nS > 1:
        start[1:] = np.cumsum(length[:-1].astype(np.int64) + 1)
    nN = int(length.astype(np.int64).sum()) + nS
    data = np.full(nN, -1, np.int8)
    for i, r in enumerate(reads):
        data[start[i]:start[i] + len(r)] = r
    return data, start, length


def remainder_chunk(reads, chunk_size=8192):
    """main.cu:270-305: only the last partial chunk (gnS % chunkSize reads) reaches the file.
    SelectChunkRemain takes `ushort chunkSize, ushort it` (main.cu:110), so the chunk starts at read
    (chunkSize mod 2^16) * (nChunk mod 2^16) and holds gnS - nChunk*chunkSize reads."""
    n_full = len(reads) // chunk_size
    first = (chunk_size & 0xFFFF) * (n_full & 0xFFFF)
    return reads[first:first + len(reads) - n_full * chunk_size]


def reference_cfrk_bytes(raw: bytes, k: int, chunk_size=8192):
    """What `cfrk in.fasta out.cfrk k 12 chunk_size` writes, per the oracle."""
    reads = remainder_chunk(read_fasta_compat(raw), chunk_size)
    if not reads:
        return b""
    data, start, length = flatten(reads)
    freq = orc.per_read_dense(data, start, length, k, orc.ORC_COMPAT)
    return orc.format_cfrk(freq, k)
