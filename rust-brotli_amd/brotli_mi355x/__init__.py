"""brotli_mi355x -- Python host-side binding of the MI355X brotli encoder library (C ABI via ctypes).

Mirrors the encoder half of the reference's own ctypes binding (c/py/brotli.py:67-195): same function
names, argument meaning and error behaviour (BrotliCompress, BrotliEncoderCompressWorkPool,
BrotliEncoderCreateWorkPool, BrotliEncoderMaxCompressedSizeMulti, BrotliEncoderVersion), plus a
streaming class over BrotliEncoderCompressStream and helpers for device-resident buffers / multi-GPU
chunking that the reference does not have.

The library is HIP only: there is no CPU fallback.  Importing fails loudly if the shared object has
not been built, and every call fails loudly (BrotliCompressorException) if no gfx950 device is usable.
"""
import ctypes
import os
from ctypes import POINTER, byref, c_char_p, c_double, c_int, c_int32, c_size_t, c_uint32, c_void_p

# The library asks the HIP runtime for 16 hardware queues (eight shard workers side by side) when it is loaded; the runtime
# reads the variable when it starts, so a process in which something else starts HIP first (torch) should import this
# module -- or set the variable -- before that.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BROTLI_MI355X_LIB") or os.path.join(os.path.dirname(_HERE), "libbrotli_mi355x.so")  # (the override is for profiling builds)

# BrotliEncoderParameter ids (c/brotli/encode.h:138-232)
BROTLI_PARAM_MODE = 0
BROTLI_PARAM_QUALITY = 1
BROTLI_PARAM_LGWIN = 2
BROTLI_PARAM_LGBLOCK = 3
BROTLI_PARAM_DISABLE_LITERAL_CONTEXT_MODELING = 4
BROTLI_PARAM_SIZE_HINT = 5
BROTLI_PARAM_LARGE_WINDOW = 6
BROTLI_PARAM_CATABLE = 167
BROTLI_PARAM_APPENDABLE = 168
BROTLI_PARAM_MAGIC_NUMBER = 169
BROTLI_PARAM_BYTE_ALIGN = 172
BROTLI_PARAM_BARE_STREAM = 173
BROTLI_OPERATION_PROCESS = 0
BROTLI_OPERATION_FLUSH = 1
BROTLI_OPERATION_FINISH = 2
BROTLI_OPERATION_EMIT_METADATA = 3


class BrotliCompressorException(Exception):
    pass


def _bind(lib):
    lib.BrotliEncoderVersion.restype = c_uint32
    lib.BrotliEncoderMaxCompressedSize.restype = c_size_t
    lib.BrotliEncoderMaxCompressedSize.argtypes = [c_size_t]
    lib.BrotliEncoderMaxCompressedSizeMulti.restype = c_size_t
    lib.BrotliEncoderMaxCompressedSizeMulti.argtypes = [c_size_t, c_size_t]
    lib.BrotliEncoderCreateInstance.restype = c_void_p
    lib.BrotliEncoderCreateInstance.argtypes = [c_void_p, c_void_p, c_void_p]
    lib.BrotliEncoderDestroyInstance.restype = None
    lib.BrotliEncoderDestroyInstance.argtypes = [c_void_p]
    lib.BrotliEncoderSetParameter.restype = c_int
    lib.BrotliEncoderSetParameter.argtypes = [c_void_p, c_int, c_uint32]
    lib.BrotliEncoderSetCustomDictionary.restype = None
    lib.BrotliEncoderSetCustomDictionary.argtypes = [c_void_p, c_size_t, c_char_p]
    lib.BrotliEncoderCompressStream.restype = c_int
    lib.BrotliEncoderCompressStream.argtypes = [c_void_p, c_int, POINTER(c_size_t), POINTER(c_void_p), POINTER(c_size_t),
                                                POINTER(c_void_p), POINTER(c_size_t)]
    lib.BrotliEncoderIsFinished.restype = c_int
    lib.BrotliEncoderIsFinished.argtypes = [c_void_p]
    lib.BrotliEncoderHasMoreOutput.restype = c_int
    lib.BrotliEncoderHasMoreOutput.argtypes = [c_void_p]
    lib.BrotliEncoderTakeOutput.restype = c_void_p
    lib.BrotliEncoderTakeOutput.argtypes = [c_void_p, POINTER(c_size_t)]
    lib.BrotliEncoderCompress.restype = c_int
    lib.BrotliEncoderCompress.argtypes = [c_int, c_int, c_int, c_size_t, c_char_p, POINTER(c_size_t), c_char_p]
    lib.BrotliEncoderCompressMulti.restype = c_int32
    lib.BrotliEncoderCompressMulti.argtypes = [c_size_t, POINTER(c_int), POINTER(c_uint32), c_size_t, c_char_p,
                                               POINTER(c_size_t), c_char_p, c_size_t, c_void_p, c_void_p, c_void_p]
    lib.BrotliEncoderCreateWorkPool.restype = c_void_p
    lib.BrotliEncoderCreateWorkPool.argtypes = [c_size_t, c_void_p, c_void_p, c_void_p]
    lib.BrotliEncoderDestroyWorkPool.restype = None
    lib.BrotliEncoderDestroyWorkPool.argtypes = [c_void_p]
    lib.BrotliEncoderCompressWorkPool.restype = c_int32
    lib.BrotliEncoderCompressWorkPool.argtypes = [c_void_p, c_size_t, POINTER(c_int), POINTER(c_uint32), c_size_t, c_char_p,
                                                  POINTER(c_size_t), c_char_p, c_size_t, c_void_p, c_void_p, c_void_p]
    lib.BrotliMi355xCompressChunk.restype = c_int32
    lib.BrotliMi355xCompressChunk.argtypes = [c_size_t, POINTER(c_int), POINTER(c_uint32), c_size_t, c_void_p, c_int,
                                              c_size_t, c_size_t, POINTER(c_size_t), c_char_p]
    lib.BrotliMi355xConcatChunks.restype = c_int32
    lib.BrotliMi355xConcatChunks.argtypes = [c_size_t, POINTER(c_char_p), POINTER(c_size_t), POINTER(c_size_t), c_char_p]
    lib.BrotliMi355xConcatChunkEnds.restype = c_int32
    lib.BrotliMi355xConcatChunkEnds.argtypes = [c_size_t, c_void_p, c_void_p, POINTER(c_size_t), POINTER(c_size_t), c_void_p, POINTER(c_size_t)]
    lib.BrotliMi355xCompressDevice.restype = c_int
    lib.BrotliMi355xCompressDevice.argtypes = [c_int, c_int, c_int, c_size_t, c_void_p, POINTER(c_size_t), c_char_p,
                                               POINTER(c_double)]
    lib.BrotliMi355xDeviceName.restype = c_char_p
    lib.BrotliMi355xLastError.restype = c_char_p
    return lib


class Library(object):
    """All entry points bound to one shared object (the product library by default)."""

    def __init__(self, path=LIB_PATH):
        if not os.path.exists(path):
            raise ImportError("%s is missing; build it with `make -C rust-brotli_amd` (or __graft_entry__.build()). "
                              "There is no CPU fallback." % path)
        self.path = path
        self.lib = _bind(ctypes.CDLL(path))

    # ---- c/py/brotli.py:54 ----
    def BrotliEncoderVersion(self):
        return self.lib.BrotliEncoderVersion()

    def BrotliEncoderMaxCompressedSizeMulti(self, input_size, num_threads):
        return self.lib.BrotliEncoderMaxCompressedSizeMulti(input_size, num_threads)

    def device_name(self):
        return self.lib.BrotliMi355xDeviceName().decode()

    def last_error(self):
        return self.lib.BrotliMi355xLastError().decode()

    @staticmethod
    def _options(compression_options_map):
        items = list(compression_options_map.items()) if hasattr(compression_options_map, "items") else list(compression_options_map)
        keys = (c_int * max(1, len(items)))(*[int(k) for k, _ in items])
        vals = (c_uint32 * max(1, len(items)))(*[int(v) for _, v in items])
        return len(items), keys, vals

    # ---- c/py/brotli.py:155-195 ----
    def BrotliCompress(self, any_input, compression_options_map={}, num_threads=4):
        data = bytes(any_input)
        n, keys, vals = self._options(compression_options_map)
        max_size = self.lib.BrotliEncoderMaxCompressedSizeMulti(len(data), num_threads)
        encoded = ctypes.create_string_buffer(max_size)
        encoded_size = c_size_t(max_size)
        ret = self.lib.BrotliEncoderCompressMulti(n, keys, vals, len(data), data, byref(encoded_size), encoded, num_threads,
                                                  None, None, None)
        if ret == 0:
            raise BrotliCompressorException("Insufficient space %d to compress %d bytes with %d threads (%s)" %
                                            (max_size, len(data), num_threads, self.last_error()))
        return bytearray(encoded.raw[:encoded_size.value])

    # ---- c/py/brotli.py:67-121 ----
    def BrotliEncoderCreateWorkPool(self, num_workers):
        return self.lib.BrotliEncoderCreateWorkPool(num_workers, None, None, None)

    def BrotliEncoderDestroyWorkPool(self, pool):
        self.lib.BrotliEncoderDestroyWorkPool(pool)

    def BrotliEncoderCompressWorkPool(self, work_pool, any_input, compression_options_map={}, num_threads=4):
        data = bytes(any_input)
        n, keys, vals = self._options(compression_options_map)
        max_size = self.lib.BrotliEncoderMaxCompressedSizeMulti(len(data), num_threads)
        encoded = ctypes.create_string_buffer(max_size)
        encoded_size = c_size_t(max_size)
        ret = self.lib.BrotliEncoderCompressWorkPool(work_pool, n, keys, vals, len(data), data, byref(encoded_size), encoded,
                                                     num_threads, None, None, None)
        if ret == 0:
            raise BrotliCompressorException("Insufficient space %d to compress %d bytes with %d threads (%s)" %
                                            (max_size, len(data), num_threads, self.last_error()))
        return bytearray(encoded.raw[:encoded_size.value])

    # ---- one-shot BrotliEncoderCompress (c/brotli/encode.h:318) ----
    def compress(self, data, quality=5, lgwin=22, mode=0):
        data = bytes(data)
        cap = self.lib.BrotliEncoderMaxCompressedSize(len(data)) + 16
        out = ctypes.create_string_buffer(cap)
        n = c_size_t(cap)
        if not self.lib.BrotliEncoderCompress(quality, lgwin, mode, len(data), data, byref(n), out):
            raise BrotliCompressorException("BrotliEncoderCompress failed: " + self.last_error())
        return ctypes.string_at(out, n.value)

    def compress_device(self, device_ptr, nbytes, quality=5, lgwin=22, mode=0, out_buffer=None):
        """One-shot compression of `nbytes` at device address `device_ptr` (e.g. tensor.data_ptr()).
        Returns (bytes, stats list of 32 doubles)."""
        cap = self.lib.BrotliEncoderMaxCompressedSize(nbytes) + 16
        out = out_buffer if out_buffer is not None else ctypes.create_string_buffer(cap)
        n = c_size_t(cap)
        stats = (c_double * 32)()
        if not self.lib.BrotliMi355xCompressDevice(quality, lgwin, mode, nbytes, c_void_p(device_ptr), byref(n), out, stats):
            raise BrotliCompressorException("BrotliMi355xCompressDevice failed: " + self.last_error())
        return ctypes.string_at(out, n.value), list(stats)

    # ---- multi-GPU helpers: one chunk per process, stitched on rank 0 ----
    def compress_chunk(self, data_or_ptr, nbytes, thread_index, num_threads, compression_options_map={}, on_device=False):
        n, keys, vals = self._options(compression_options_map)
        chunk_bytes = ((thread_index + 1) * nbytes) // num_threads - (thread_index * nbytes) // num_threads
        cap = self.lib.BrotliEncoderMaxCompressedSize(chunk_bytes) + 16
        out = ctypes.create_string_buffer(cap)
        size = c_size_t(cap)
        if on_device:
            src = c_void_p(data_or_ptr)
            keep = None
        else:
            keep = ctypes.create_string_buffer(bytes(data_or_ptr), max(1, nbytes))
            src = ctypes.cast(keep, c_void_p)
        ret = self.lib.BrotliMi355xCompressChunk(n, keys, vals, nbytes, src, 1 if on_device else 0, thread_index, num_threads,
                                                 byref(size), out)
        if ret == 0:
            raise BrotliCompressorException("BrotliMi355xCompressChunk failed: " + self.last_error())
        return ctypes.string_at(out, size.value)

    def concat_chunks(self, chunks):
        arr = (c_char_p * len(chunks))(*[bytes(c) for c in chunks])
        sizes = (c_size_t * len(chunks))(*[len(c) for c in chunks])
        cap = sum(len(c) for c in chunks) + 64
        out = ctypes.create_string_buffer(cap)
        n = c_size_t(cap)
        if not self.lib.BrotliMi355xConcatChunks(len(chunks), arr, sizes, byref(n), out):
            raise BrotliCompressorException("BrotliMi355xConcatChunks failed: " + self.last_error())
        return ctypes.string_at(out, n.value)

    def concat_chunk_views(self, views):
        """like concat_chunks, but on (address, size) pairs of host memory and without copying the result: returns a
        memoryview of a buffer owned by the Library (valid until the next call)"""
        n_chunks = len(views)
        arr = (c_void_p * n_chunks)(*[c_void_p(a) for a, _ in views])
        sizes = (c_size_t * n_chunks)(*[s for _, s in views])
        cap = sum(s for _, s in views) + 64
        if getattr(self, "_concat_buf", None) is None or len(self._concat_buf) < cap:
            self._concat_buf = ctypes.create_string_buffer(cap)
        n = c_size_t(len(self._concat_buf))
        if not self.lib.BrotliMi355xConcatChunks(n_chunks, ctypes.cast(arr, ctypes.POINTER(c_char_p)), sizes, byref(n), self._concat_buf):
            raise BrotliCompressorException("BrotliMi355xConcatChunks failed: " + self.last_error())
        return memoryview(self._concat_buf)[:n.value]

    def concat_chunk_ends(self, heads, tails, sizes, out_address, out_capacity):
        """BrotliMi355xConcatChunkEnds: heads / tails are (n, 8) uint8 host tensors (or anything with data_ptr()) holding the
        first / last bytes of every chunk; junction bytes go to out_address.  Returns (total size, [(dst, src, count)])"""
        n_chunks = len(sizes)
        csizes = (c_size_t * n_chunks)(*sizes)
        bodies = (c_size_t * (3 * n_chunks))()
        n = c_size_t(out_capacity)
        if not self.lib.BrotliMi355xConcatChunkEnds(n_chunks, c_void_p(heads.data_ptr()), c_void_p(tails.data_ptr()), csizes, byref(n),
                                                    c_void_p(out_address), bodies):
            raise BrotliCompressorException("BrotliMi355xConcatChunkEnds failed: " + self.last_error())
        return n.value, [(bodies[3 * i], bodies[3 * i + 1], bodies[3 * i + 2]) for i in range(n_chunks)]

    def encoder(self, **params):
        return Encoder(self, **params)


class Encoder(object):
    """Streaming encoder over BrotliEncoderCompressStream (what CompressorWriter does in the reference,
    src/enc/writer.rs:183-313): write() hands input over, finish() returns the stream."""

    def __init__(self, library, params=(), dictionary=None):
        self._l = library
        self._s = library.lib.BrotliEncoderCreateInstance(None, None, None)
        if not self._s:
            raise BrotliCompressorException("BrotliEncoderCreateInstance failed")
        items = params.items() if hasattr(params, "items") else params
        for k, v in items:
            if not library.lib.BrotliEncoderSetParameter(self._s, int(k), int(v)):
                raise BrotliCompressorException("invalid parameter %r=%r" % (k, v))
        if dictionary is not None:
            library.lib.BrotliEncoderSetCustomDictionary(self._s, len(dictionary), bytes(dictionary))
        self._out = bytearray()

    def _stream(self, op, data):
        buf = ctypes.create_string_buffer(bytes(data), max(1, len(data)))
        avail_in = c_size_t(len(data))
        next_in = c_void_p(ctypes.addressof(buf))
        chunk = ctypes.create_string_buffer(1 << 16)
        while True:
            avail_out = c_size_t(len(chunk))
            next_out = c_void_p(ctypes.addressof(chunk))
            total = c_size_t(0)
            ok = self._l.lib.BrotliEncoderCompressStream(self._s, op, byref(avail_in), byref(next_in), byref(avail_out),
                                                         byref(next_out), byref(total))
            if not ok:
                raise BrotliCompressorException("BrotliEncoderCompressStream failed: " + self._l.last_error())
            self._out += chunk.raw[:len(chunk) - avail_out.value]
            if avail_in.value == 0 and not self._l.lib.BrotliEncoderHasMoreOutput(self._s):
                break

    def write(self, data):
        self._stream(BROTLI_OPERATION_PROCESS, data)

    def flush(self, data=b""):
        """BROTLI_OPERATION_FLUSH (CompressorWriter::flush, src/enc/writer.rs): everything written so far becomes
        decodable; returns the bytes produced since the last flush / start"""
        self._stream(BROTLI_OPERATION_FLUSH, data)
        piece = bytes(self._out)
        self._flushed = getattr(self, "_flushed", b"") + piece
        self._out = bytearray()
        return piece

    def emit_metadata(self, payload):
        """BROTLI_OPERATION_EMIT_METADATA: pending input is flushed, then `payload` (at most 16 MiB) goes out as a
        metadata block, which decoders skip; returns the bytes produced"""
        self._stream(BROTLI_OPERATION_EMIT_METADATA, payload)
        piece = bytes(self._out)
        self._out = bytearray()
        return piece

    def set_parameter(self, key, value):
        return bool(self._l.lib.BrotliEncoderSetParameter(self._s, int(key), int(value)))

    def finish(self):
        self._stream(BROTLI_OPERATION_FINISH, b"")
        assert self._l.lib.BrotliEncoderIsFinished(self._s)
        return bytes(self._out)  # (after flush() calls: the remainder of the stream)

    def close(self):
        if self._s:
            self._l.lib.BrotliEncoderDestroyInstance(self._s)
            self._s = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default = None


def default_library():
    global _default
    if _default is None:
        _default = Library()
    return _default


def BrotliEncoderVersion():
    return default_library().BrotliEncoderVersion()


def BrotliEncoderMaxCompressedSizeMulti(input_size, num_threads):
    return default_library().BrotliEncoderMaxCompressedSizeMulti(input_size, num_threads)


def BrotliCompress(any_input, compression_options_map={}, num_threads=4):
    return default_library().BrotliCompress(any_input, compression_options_map, num_threads)


def BrotliEncoderCreateWorkPool(num_workers):
    return default_library().BrotliEncoderCreateWorkPool(num_workers)


def BrotliEncoderDestroyWorkPool(pool):
    return default_library().BrotliEncoderDestroyWorkPool(pool)


def BrotliEncoderCompressWorkPool(work_pool, any_input, compression_options_map={}, num_threads=4):
    return default_library().BrotliEncoderCompressWorkPool(work_pool, any_input, compression_options_map, num_threads)


# fail loudly at import time when the product library is absent
if not os.path.exists(LIB_PATH):
    raise ImportError("libbrotli_mi355x.so is missing; build it with `make -C rust-brotli_amd` "
                      "(or __graft_entry__.build()). There is no CPU fallback.")
