"""Test-only baseline JPEG writer (T.81 Annex K Huffman tables, 4:2:0 or 4:2:2, JFIF) that turns quantised zig-zag
coefficients -- from the oracle or from the HIP kernels -- into a file an independent decoder (Pillow/libjpeg) can
read.  Purpose: pin the FDCT / quantiser / zig-zag / level-shift conventions of oracle/jpeg_oracle.c to real JPEG,
since the reference's own FDCT (external libgpujpeg) is not available.  Not part of the product."""
import io
import struct

import numpy as np

# T.81 Annex K.3 typical Huffman tables: (bits[1..16], huffval)
DC_L = ([0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0], list(range(12)))
DC_C = ([0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0], list(range(12)))
AC_L = ([0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d],
        [0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14, 0x32, 0x81, 0x91, 0xa1, 0x08,
         0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26, 0x27, 0x28,
         0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59,
         0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89,
         0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6,
         0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2,
         0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa])
AC_C = ([0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77],
        [0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22, 0x32, 0x81, 0x08, 0x14, 0x42, 0x91,
         0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26,
         0x27, 0x28, 0x29, 0x2a, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58,
         0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x82, 0x83, 0x84, 0x85, 0x86, 0x87,
         0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4,
         0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda,
         0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa])

ZIGZAG = [0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
          35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63]


def _codes(bits, vals):
    out, code, k = {}, 0, 0
    for length in range(1, 17):
        for _ in range(bits[length - 1]):
            out[vals[k]] = (code, length)
            code += 1
            k += 1
        code <<= 1
    return out


class _Bits:
    def __init__(self):
        self.buf = bytearray()
        self.acc = 0
        self.n = 0

    def put(self, code, length):
        self.acc = (self.acc << length) | (code & ((1 << length) - 1))
        self.n += length
        while self.n >= 8:
            b = (self.acc >> (self.n - 8)) & 0xFF
            self.buf.append(b)
            if b == 0xFF:
                self.buf.append(0)
            self.n -= 8
        self.acc &= (1 << self.n) - 1

    def flush(self):
        if self.n:
            self.put((1 << (8 - self.n)) - 1, 8 - self.n)


def _mag(v):
    a = abs(int(v))
    size = a.bit_length()
    return size, (v if v >= 0 else v + (1 << size) - 1)


def _block(bw, zz, pred, dc, ac):
    diff = int(zz[0]) - pred
    s, bits = _mag(diff)
    bw.put(*dc[s])
    if s:
        bw.put(bits, s)
    run = 0
    for k in range(1, 64):
        v = int(zz[k])
        if v == 0:
            run += 1
            continue
        while run > 15:
            bw.put(*ac[0xF0])
            run -= 16
        s, bits = _mag(v)
        bw.put(*ac[(run << 4) | s])
        bw.put(bits, s)
        run = 0
    if run:
        bw.put(*ac[0x00])
    return int(zz[0])


def header_bytes(width, height, qt_luma, qt_chroma, restart=0, sub=420):
    """SOI .. SOS of the stream write_jpeg() produces (the product builds the same header on the host)."""
    return write_jpeg(width, height, qt_luma, qt_chroma, None, None, None, restart, header_only=True, sub=sub)


def write_jpeg420(width, height, qt_luma, qt_chroma, coef_y, coef_cb, coef_cr, restart=0, header_only=False):
    return write_jpeg(width, height, qt_luma, qt_chroma, coef_y, coef_cb, coef_cr, restart, header_only, sub=420)


def write_jpeg(width, height, qt_luma, qt_chroma, coef_y, coef_cb, coef_cr, restart=0, header_only=False, sub=420, ycc=False):
    """coef_*: (n_blocks, 64) int16 zig-zag; component-0 blocks in raster order over a (hs*mcu_w) x (vs*mcu_h) block grid,
    the others over mcu_w x mcu_h.  sub=420: MCU 16x16 (hs=vs=2); 422: MCU 16x8 (hs=2, vs=1), both JFIF YCbCr;
    444: MCU 8x8 (hs=vs=1), components R, G, B without colour transform, libjpeg conventions for JCS_RGB (Adobe APP14
    transform 0, ids 'R','G','B', table 0 for every component; pass the table-0 quantiser as qt_chroma too).
    sub=444, ycc=True: 4:4:4 Y'CbCr instead (JFIF, components 1, 2, 3, the chroma tables for Cb and Cr).
    qt_*: 64 quantiser steps in natural order.  restart = MCUs per restart interval (0 = none).  Returns the stream."""
    hs, vs = (1, 1) if sub == 444 else ((2, 2) if sub == 420 else (2, 1))
    rgb = sub == 444 and not ycc
    mw, mh = (width + 8 * hs - 1) // (8 * hs), (height + 8 * vs - 1) // (8 * vs)
    out = io.BytesIO()
    out.write(b"\xff\xd8")
    if rgb:
        out.write(b"\xff\xee" + struct.pack(">H5sHHHB", 14, b"Adobe", 100, 0, 0, 0))
    else:
        out.write(b"\xff\xe0" + struct.pack(">H5sBBBHHBB", 16, b"JFIF\0", 1, 1, 0, 1, 1, 0, 0))
    for tid, qt in ((0, qt_luma), (1, qt_chroma)):
        out.write(b"\xff\xdb" + struct.pack(">HB", 67, tid) + bytes(int(qt[i]) for i in ZIGZAG))
    ids = (0x52, 0x47, 0x42) if rgb else (1, 2, 3)
    t12 = 0 if rgb else 1
    out.write(b"\xff\xc0" + struct.pack(">HBHHB", 17, 8, height, width, 3) + bytes([ids[0], (hs << 4) | vs, 0, ids[1], 0x11, t12, ids[2], 0x11, t12]))
    for (tc, th, (bits, vals)) in ((0, 0, DC_L), (1, 0, AC_L), (0, 1, DC_C), (1, 1, AC_C)):
        out.write(b"\xff\xc4" + struct.pack(">HB", 19 + len(vals), (tc << 4) | th) + bytes(bits) + bytes(vals))
    if restart:
        out.write(b"\xff\xdd" + struct.pack(">HH", 4, restart))
    out.write(b"\xff\xda" + struct.pack(">HB", 12, 3) + bytes([ids[0], 0x00, ids[1], t12 * 0x11, ids[2], t12 * 0x11, 0, 63, 0]))
    if header_only:
        return out.getvalue()
    dcl, acl = _codes(*DC_L), _codes(*AC_L)
    dcc, acc = (dcl, acl) if rgb else (_codes(*DC_C), _codes(*AC_C))
    bw = _Bits()
    py = pcb = pcr = 0
    bwid = hs * mw
    n_mcu = mw * mh
    for m in range(n_mcu):
        my, mx = divmod(m, mw)
        if restart and m and m % restart == 0:
            bw.flush()
            out.write(bytes(bw.buf))
            out.write(bytes([0xFF, 0xD0 + ((m // restart - 1) & 7)]))
            bw = _Bits()
            py = pcb = pcr = 0
        for dy in range(vs):
            for dx in range(hs):
                py = _block(bw, coef_y[(vs * my + dy) * bwid + hs * mx + dx], py, dcl, acl)
        pcb = _block(bw, coef_cb[my * mw + mx], pcb, dcc, acc)
        pcr = _block(bw, coef_cr[my * mw + mx], pcr, dcc, acc)
    bw.flush()
    out.write(bytes(bw.buf))
    out.write(b"\xff\xd9")
    return out.getvalue()


def write_jpeg_noninterleaved(width, height, qt, coefs, restart=0, qt_chroma=None):
    """Baseline R,G,B 4:4:4 stream with ONE SCAN PER COMPONENT (T.81 A.2.2: a non-interleaved scan walks the component's own
    ceil(size / 8) block grid; the restart interval counts its data units) -- the layout GPUJPEG writes for RGB input by default
    (gpujpeg.cpp:302: interleaved = 0).  coefs: three (n_blocks, 64) zig-zag arrays over the 8x8-block grid; table 0 for everything.
    qt_chroma given: the same layout for Y'CbCr components (RGB input with color_space_internal = a Y'CbCr space): JFIF, components 1, 2, 3,
    both quantiser tables and all four Huffman tables, the chroma ones for the Cb and Cr scans."""
    bw_, bh_ = (width + 7) // 8, (height + 7) // 8
    ycc = qt_chroma is not None
    out = io.BytesIO()
    out.write(b"\xff\xd8")
    if ycc:
        out.write(b"\xff\xe0" + struct.pack(">H5sBBBHHBB", 16, b"JFIF\0", 1, 1, 0, 1, 1, 0, 0))
    else:
        out.write(b"\xff\xee" + struct.pack(">H5sHHHB", 14, b"Adobe", 100, 0, 0, 0))
    out.write(b"\xff\xdb" + struct.pack(">HB", 67, 0) + bytes(int(qt[i]) for i in ZIGZAG))
    if ycc:
        out.write(b"\xff\xdb" + struct.pack(">HB", 67, 1) + bytes(int(qt_chroma[i]) for i in ZIGZAG))
    ids = (1, 2, 3) if ycc else (0x52, 0x47, 0x42)
    t12 = 1 if ycc else 0
    out.write(b"\xff\xc0" + struct.pack(">HBHHB", 17, 8, height, width, 3) + bytes([ids[0], 0x11, 0, ids[1], 0x11, t12, ids[2], 0x11, t12]))
    for (tc, th, (bits, vals)) in ((0, 0, DC_L), (1, 0, AC_L)) + (((0, 1, DC_C), (1, 1, AC_C)) if ycc else ()):
        out.write(b"\xff\xc4" + struct.pack(">HB", 19 + len(vals), (tc << 4) | th) + bytes(bits) + bytes(vals))
    if restart:
        out.write(b"\xff\xdd" + struct.pack(">HH", 4, restart))
    for c in range(3):
        dcl, acl = (_codes(*DC_C), _codes(*AC_C)) if ycc and c else (_codes(*DC_L), _codes(*AC_L))
        out.write(b"\xff\xda" + struct.pack(">HB", 8, 1) + bytes([ids[c], 0x11 if ycc and c else 0x00, 0, 63, 0]))
        bw = _Bits()
        pred = 0
        for u in range(bw_ * bh_):
            if restart and u and u % restart == 0:
                bw.flush()
                out.write(bytes(bw.buf))
                out.write(bytes([0xFF, 0xD0 + ((u // restart - 1) & 7)]))
                bw = _Bits()
                pred = 0
            pred = _block(bw, coefs[c][u], pred, dcl, acl)
        bw.flush()
        out.write(bytes(bw.buf))
    out.write(b"\xff\xd9")
    return out.getvalue()
