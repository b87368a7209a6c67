"""Decoder for Blosc 1 frames (c-blosc 1.x, the payload of HDF5 filter 32001) -- ev-licious compresses its event files with
it (ev-licious/src/evlicious/io/utils/h5_writer.py:8-26: zstd, level 1, bit shuffle), and neither the ``blosc`` module nor an
HDF5 plugin is part of this image.  Written from the published frame format; pinned against frames produced by the real
libblosc 1.21.0 (tests/golden/h5/make_blosc_fixtures.py, tests/test_h5lite_cpu.py).

Frame: 16-byte header {version, versionlz, flags, typesize, nbytes:i32, blocksize:i32, cbytes:i32}; flags bit 0 byte
shuffle, bit 1 "memcpyed" (the payload is the data), bit 2 bit shuffle, bit 4 "do not split", bits 5-7 the codec's format
(0 blosclz, 1 lz4 / lz4hc, 2 snappy, 3 zlib, 4 zstd).  Then ``bstarts`` (one int32 offset per block) and the blocks; a
block is 1 stream, or ``typesize`` streams (one per byte of the element) when it was split; a stream = int32 size + payload,
stored raw when its size equals the stream's uncompressed size.  The inner codecs come from the system's shared libraries
(libzstd.so.1, liblz4.so.1) through ctypes, zlib from the standard library; blosclz and snappy frames are not decoded.
"""
import ctypes
import ctypes.util
import struct
import zlib

import numpy as np

_MAX_SPLITS = 16
_MIN_BUFFERSIZE = 128
_libs = {}


def _lib(name):
    if name not in _libs:
        lib = None
        for cand in ("lib%s.so.1" % name, ctypes.util.find_library(name)):
            if not cand:
                continue
            try:
                lib = ctypes.CDLL(cand)
                break
            except OSError:
                continue
        _libs[name] = lib
    if _libs[name] is None:
        raise NotImplementedError("Blosc frame compressed with %s: lib%s is not on this machine" % (name, name))
    return _libs[name]


def _zstd(src, n):
    lib = _lib("zstd")
    lib.ZSTD_decompress.restype = ctypes.c_size_t
    lib.ZSTD_decompress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t]
    lib.ZSTD_isError.restype = ctypes.c_uint
    lib.ZSTD_isError.argtypes = [ctypes.c_size_t]
    dst = ctypes.create_string_buffer(n)
    got = lib.ZSTD_decompress(dst, n, src, len(src))
    if lib.ZSTD_isError(got) or got != n:
        raise ValueError("corrupt Blosc frame: zstd stream does not decode to %d bytes" % n)
    return dst.raw


def _lz4(src, n):
    lib = _lib("lz4")
    lib.LZ4_decompress_safe.restype = ctypes.c_int
    lib.LZ4_decompress_safe.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    dst = ctypes.create_string_buffer(n)
    if lib.LZ4_decompress_safe(src, dst, len(src), n) != n:
        raise ValueError("corrupt Blosc frame: lz4 stream does not decode to %d bytes" % n)
    return dst.raw


def _zlib(src, n):
    d = zlib.decompressobj()
    out = d.decompress(src, n + 1)          # bounded: a stream that inflates past its declared size is corrupt, not a bomb
    if len(out) != n or d.unconsumed_tail:
        raise ValueError("corrupt Blosc frame: zlib stream does not decode to %d bytes" % n)
    return out


_CODECS = {1: _lz4, 3: _zlib, 4: _zstd}
_CODEC_NAMES = {0: "blosclz", 1: "lz4", 2: "snappy", 3: "zlib", 4: "zstd"}


def _unshuffle(buf, typesize):
    """byte shuffle undone: byte j of every element was stored together (the tail that is no whole element unchanged)"""
    n = len(buf) // typesize
    body = np.frombuffer(buf, np.uint8, n * typesize).reshape(typesize, n).T
    return body.tobytes() + buf[n * typesize:]


def _bitunshuffle(buf, typesize):
    """bit shuffle undone: bit b of byte k of the elements was stored as row 8 k + b, eight elements per byte, first element in
    the least significant bit.  c-blosc shuffles a block only when its element count is a multiple of eight (shuffle.c,
    ``bitshuffle``: otherwise the block is copied as it is); bytes behind the last whole element are unchanged"""
    n = len(buf) // typesize
    if n == 0 or n % 8:
        return buf
    rows = np.frombuffer(buf, np.uint8, n * typesize).reshape(typesize * 8, n // 8)
    bits = np.unpackbits(rows, axis=1, bitorder="little")            # [8 typesize][n]
    body = np.packbits(bits.T.reshape(n, typesize, 8), axis=2, bitorder="little").reshape(n, typesize)
    return body.tobytes() + buf[n * typesize:]


MAX_FRAME_BYTES = 1 << 31   # Blosc 1 frames hold at most 2 GiB - 16


def decompress(frame, expected_nbytes=None):
    """One Blosc 1 frame (bytes-like) -> the bytes that were compressed.  expected_nbytes: the decoded size the caller knows
    (an HDF5 chunk's): a frame that declares another size is rejected before anything is allocated.  Malformed frames raise
    ValueError, never struct.error."""
    try:
        return _decompress(bytes(frame), expected_nbytes)
    except struct.error as e:
        raise ValueError("corrupt Blosc frame: %s" % e)


def _decompress(frame, expected_nbytes):
    if len(frame) < 16:
        raise ValueError("not a Blosc frame: %d bytes" % len(frame))
    version, _versionlz, flags, typesize = frame[0], frame[1], frame[2], frame[3]
    nbytes, blocksize, cbytes = struct.unpack_from("<iii", frame, 4)
    if version != 2 or nbytes < 0 or blocksize <= 0 and nbytes > 0 or cbytes > len(frame):
        raise ValueError("not a Blosc 1 frame (version %d, nbytes %d, blocksize %d, cbytes %d of %d)"
                         % (version, nbytes, blocksize, cbytes, len(frame)))
    if expected_nbytes is not None and nbytes != int(expected_nbytes):
        raise ValueError("Blosc frame declares %d bytes, the container expects %d" % (nbytes, int(expected_nbytes)))
    if nbytes == 0:
        return b""
    if flags & 0x2:                                                   # memcpyed
        if len(frame) < 16 + nbytes:
            raise ValueError("corrupt Blosc frame: memcpyed frame of %d bytes declares %d" % (len(frame), nbytes))
        return frame[16:16 + nbytes]
    fmt = flags >> 5
    if fmt not in _CODECS:
        raise NotImplementedError("Blosc frame compressed with %s: not decoded here" % _CODEC_NAMES.get(fmt, "codec %d" % fmt))
    codec = _CODECS[fmt]
    typesize = max(int(typesize), 1)
    nblocks = -(-nbytes // blocksize)
    if 16 + 4 * nblocks > len(frame):
        raise ValueError("corrupt Blosc frame: block table of %d entries in %d bytes" % (nblocks, len(frame)))
    bstarts = struct.unpack_from("<%di" % nblocks, frame, 16)
    out = []
    for j in range(nblocks):
        bsize = blocksize if (j + 1) * blocksize <= nbytes else nbytes - j * blocksize
        leftover = bsize != blocksize
        split = not (flags & 0x10) and typesize <= _MAX_SPLITS and bsize // typesize >= _MIN_BUFFERSIZE and not leftover
        nsplits = typesize if split else 1
        ne = bsize // nsplits
        pos = bstarts[j]
        parts = []
        for _ in range(nsplits):
            if pos < 16 or pos + 4 > len(frame):
                raise ValueError("corrupt Blosc frame: block start %d" % pos)
            (cb,) = struct.unpack_from("<i", frame, pos)
            pos += 4
            if cb < 0 or pos + cb > len(frame):
                raise ValueError("corrupt Blosc frame: stream of %d bytes at %d" % (cb, pos))
            parts.append(frame[pos:pos + cb] if cb == ne else codec(frame[pos:pos + cb], ne))
            pos += cb
        block = b"".join(parts)
        if (flags & 0x1) and typesize > 1:
            block = _unshuffle(block, typesize)
        elif (flags & 0x4) and bsize >= typesize:
            block = _bitunshuffle(block, typesize)
        out.append(block)
    res = b"".join(out)
    if len(res) != nbytes:
        raise ValueError("corrupt Blosc frame: decoded %d bytes, header says %d" % (len(res), nbytes))
    return res
