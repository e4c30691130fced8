"""A small FLAC ENCODER for the decoder tests (RFC 9639) -- test infrastructure only.  Every frame can be forced to a
subframe type / stereo mode / residual coding so that each branch of csrc/host_audio.cpp meets a stream it did not write."""
from typing import List, Optional, Sequence


class BitWriter:
    def __init__(self):
        self.acc, self.n, self.out = 0, 0, bytearray()

    def put(self, value: int, bits: int):
        if bits == 0:
            return
        value &= (1 << bits) - 1
        self.acc = (self.acc << bits) | value
        self.n += bits
        while self.n >= 8:
            self.n -= 8
            self.out.append((self.acc >> self.n) & 0xFF)
        self.acc &= (1 << self.n) - 1

    def unary(self, q: int):
        while q >= 32:
            self.put(0, 32)
            q -= 32
        self.put(1, q + 1)

    def align(self):
        if self.n:
            self.put(0, 8 - self.n)

    def bytes(self) -> bytes:
        assert self.n == 0
        return bytes(self.out)


def crc8(data: bytes) -> int:
    c = 0
    for b in data:
        c ^= b
        for _ in range(8):
            c = ((c << 1) ^ 0x07) & 0xFF if c & 0x80 else (c << 1) & 0xFF
    return c


def crc16(data: bytes) -> int:
    c = 0
    for b in data:
        c ^= b << 8
        for _ in range(8):
            c = ((c << 1) ^ 0x8005) & 0xFFFF if c & 0x8000 else (c << 1) & 0xFFFF
    return c


def _utf8_number(v: int) -> bytes:
    if v < 0x80:
        return bytes([v])
    n = 2
    while v >= 1 << (5 * n + 1):          # payload bits of an n-byte sequence: (7 - n) + 6 (n - 1) = 5 n + 1
        n += 1
    out = []
    for _ in range(n - 1):
        out.append(0x80 | (v & 0x3F))
        v >>= 6
    lead = ((0xFF << (8 - n)) & 0xFF) | v
    return bytes([lead] + out[::-1])


def _residual(bw: BitWriter, res: Sequence[int], block: int, order: int, porder: int, rice2: bool, escape: bool):
    bw.put(1 if rice2 else 0, 2)
    bw.put(porder, 4)
    pbits, esc = (5, 31) if rice2 else (4, 15)
    i = 0
    for pt in range(1 << porder):
        count = (block >> porder if porder else block) - (order if pt == 0 else 0)
        part = res[i:i + count]
        i += count
        if escape and pt % 2 == 1:
            nb = max([(x if x >= 0 else ~x).bit_length() + 1 for x in part] + [1])
            bw.put(esc, pbits)
            bw.put(nb, 5)
            for x in part:
                bw.put(x, nb)
            continue
        mean = sum(abs(x) for x in part) / max(len(part), 1)
        k = max(0, min(esc - 1, int(mean).bit_length()))
        bw.put(k, pbits)
        for x in part:
            u = 2 * x if x >= 0 else -2 * x - 1
            bw.unary(u >> k)
            bw.put(u, k)
    assert i == len(res)


def _subframe(bw: BitWriter, s: Sequence[int], bps: int, kind: str, order: int = 0, porder: int = 0, rice2: bool = False,
              escape: bool = False, coefs: Optional[Sequence[int]] = None, prec: int = 12, shift: int = 9):
    block = len(s)
    wasted = 0
    if any(s) and kind != "nowaste":
        while all(x % (1 << (wasted + 1)) == 0 for x in s) and wasted + 1 < bps:
            wasted += 1
    if kind == "nowaste":
        kind = "verbatim"
    s = [x >> wasted for x in s]
    bps -= wasted
    code = {"constant": 0, "verbatim": 1}.get(kind)
    if kind == "fixed":
        code = 8 + order
    elif kind == "lpc":
        order = len(coefs)
        code = 31 + order
    bw.put(0, 1)
    bw.put(code, 6)
    if wasted:
        bw.put(1, 1)
        bw.unary(wasted - 1)
    else:
        bw.put(0, 1)
    if kind == "constant":
        assert all(x == s[0] for x in s)
        bw.put(s[0], bps)
    elif kind == "verbatim":
        for x in s:
            bw.put(x, bps)
    elif kind == "fixed":
        for x in s[:order]:
            bw.put(x, bps)
        c = {0: [], 1: [1], 2: [2, -1], 3: [3, -3, 1], 4: [4, -6, 4, -1]}[order]
        res = [s[i] - sum(cj * s[i - 1 - j] for j, cj in enumerate(c)) for i in range(order, block)]
        _residual(bw, res, block, order, porder, rice2, escape)
    else:
        for x in s[:order]:
            bw.put(x, bps)
        bw.put(prec - 1, 4)
        bw.put(shift, 5)
        for cj in coefs:
            assert -(1 << (prec - 1)) <= cj < (1 << (prec - 1)), "coefficient does not fit the precision"
            bw.put(cj, prec)
        res = [s[i] - (sum(cj * s[i - 1 - j] for j, cj in enumerate(coefs)) >> shift) for i in range(order, block)]
        _residual(bw, res, block, order, porder, rice2, escape)


_BS_CODES = {192: 1, 576: 2, 1152: 3, 2304: 4, 4608: 5, 256: 8, 512: 9, 1024: 10, 2048: 11, 4096: 12, 8192: 13,
             16384: 14, 32768: 15}
_SS_CODES = {8: 1, 12: 2, 16: 4, 20: 5, 24: 6, 32: 7}


def encode(channels: List[List[int]], bps: int, rate: int, frames: List[dict], total_known: bool = True,
           variable: bool = False, id3: bool = False, explicit_codes: bool = True) -> bytes:
    """channels: per-channel integer samples.  frames: one dict per frame: {"n": block size, "stereo": None | "ls" | "sr" |
    "ms", "sub": [per-channel kwargs of _subframe]}."""
    nch, total = len(channels), len(channels[0])
    out = bytearray()
    if id3:
        out += b"ID3\x04\x00\x00" + bytes([0, 0, 0, 11]) + b"TIT2\x00\x00\x00\x01\x00\x00\x00"
    out += b"fLaC"
    blocks = [f["n"] for f in frames]
    si = BitWriter()
    si.put(min(blocks[:-1] or blocks), 16)
    si.put(max(blocks), 16)
    si.put(0, 24)
    si.put(0, 24)
    si.put(rate, 20)
    si.put(nch - 1, 3)
    si.put(bps - 1, 5)
    si.put(total if total_known else 0, 36)
    si.put(0, 128)
    out += bytes([0x00, 0, 0, 34]) + si.bytes()                      # STREAMINFO, not last
    out += bytes([0x84, 0, 0, 8]) + b"\x00" * 8                      # a VORBIS_COMMENT-typed block to skip, last
    pos = 0
    for fi, f in enumerate(frames):
        n = f["n"]
        chs = [c[pos:pos + n] for c in channels]
        assert len(chs[0]) == n, "frames must cover the samples exactly"
        bw = BitWriter()
        bw.put(0b11111111111110, 14)
        bw.put(0, 1)
        bw.put(1 if variable else 0, 1)
        bs_code = _BS_CODES.get(n) if not f.get("explicit_bs") else None
        if bs_code is None:
            bs_code = 6 if n <= 256 else 7
        bw.put(bs_code, 4)
        sr_code = {8000: 4, 16000: 5, 22050: 6, 24000: 7, 32000: 8, 44100: 9, 48000: 10, 96000: 11}.get(rate, 0) if explicit_codes else 0
        if f.get("sr_hz16"):
            sr_code = 13
        bw.put(sr_code, 4)
        stereo = f.get("stereo")
        bw.put({None: nch - 1, "ls": 8, "sr": 9, "ms": 10}[stereo], 4)
        bw.put(_SS_CODES[bps] if explicit_codes and bps in _SS_CODES else 0, 3)
        bw.put(0, 1)
        for b in _utf8_number(pos if variable else fi + f.get("number_offset", 0)):
            bw.put(b, 8)
        if bs_code == 6:
            bw.put(n - 1, 8)
        elif bs_code == 7:
            bw.put(n - 1, 16)
        if sr_code == 13:
            bw.put(rate, 16)
        hdr = bw.bytes()
        bw.put(crc8(hdr), 8)
        if stereo == "ls":
            chs, widths = [chs[0], [a - b for a, b in zip(chs[0], chs[1])]], [bps, bps + 1]
        elif stereo == "sr":
            chs, widths = [[a - b for a, b in zip(chs[0], chs[1])], chs[1]], [bps + 1, bps]
        elif stereo == "ms":
            chs, widths = [[(a + b) >> 1 for a, b in zip(chs[0], chs[1])], [a - b for a, b in zip(chs[0], chs[1])]], [bps, bps + 1]
        else:
            widths = [bps] * nch
        for c in range(nch):
            _subframe(bw, chs[c], widths[c], **f["sub"][c])
        bw.align()
        body = bw.bytes()
        out += body + crc16(body).to_bytes(2, "big")
        pos += n
    assert pos == total
    return bytes(out)
