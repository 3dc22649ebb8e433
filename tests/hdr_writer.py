"""Test-side writer of Radiance .hdr (RGBE) files — TEST INFRASTRUCTURE ONLY (the reference ships no .hdr file).

float -> RGBE follows the published encoder (G. Ward, "Real Pixels", Graphics Gems II): with v = max(r,g,b),
v < 1e-32 -> (0,0,0,0); else (m, e) = frexp(v), scale = m * 256 / v, bytes = trunc(rgb * scale), e + 128.
Scanlines are written flat or with the new-style run-length coding (per-component runs, marker 2 2 hi lo).
"""
import numpy as np


def float_to_rgbe(rgb):
    rgb = np.asarray(rgb, dtype=np.float32)
    v = rgb.max(axis=-1)
    m, e = np.frexp(v)
    with np.errstate(divide="ignore", invalid="ignore"):
        scale = np.where(v < 1e-32, 0.0, m * 256.0 / v)
    out = np.zeros(rgb.shape[:-1] + (4,), np.uint8)
    out[..., :3] = np.clip(np.floor(rgb * scale[..., None]), 0, 255).astype(np.uint8)
    out[..., 3] = np.where(v < 1e-32, 0, e + 128).astype(np.uint8)
    return out


def _rle_component(row):
    """New-style RLE of one component row (bytes): runs of >= 3 equal bytes as (128 + n, value), else literals <= 128."""
    out = bytearray()
    n = len(row)
    i = 0
    while i < n:
        run = 1
        while i + run < n and run < 127 and row[i + run] == row[i]:
            run += 1
        if run >= 3:
            out += bytes([128 + run, row[i]])
            i += run
            continue
        j = i
        while j < n and j - i < 128:
            r = 1
            while j + r < n and r < 3 and row[j + r] == row[j]:
                r += 1
            if r >= 3:
                break
            j += 1
        out += bytes([j - i]) + bytes(row[i:j])
        i = j
    return bytes(out)


def encode_hdr(rgbe, rle=True, extra_header=(), crlf=False):
    h, w = rgbe.shape[:2]
    nl = "\r\n" if crlf else "\n"
    head = "#?RADIANCE" + nl + "".join(x + nl for x in extra_header) + "FORMAT=32-bit_rle_rgbe" + nl + nl + f"-Y {h} +X {w}" + nl
    body = bytearray()
    for y in range(h):
        if rle and 8 <= w < 32768:
            body += bytes([2, 2, w >> 8, w & 255])
            for c in range(4):
                body += _rle_component(rgbe[y, :, c].tobytes())
        else:
            body += rgbe[y].tobytes()
    return head.encode("ascii") + bytes(body)
