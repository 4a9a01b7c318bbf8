"""One shard per process (one process per GPU): the reference's compress_multi split carried over ranks.

Mirrors src/enc/threading/mod.rs:333-411: shard i of T is bytes [i*N/T, (i+1)*N/T) of the input; shard 0 is encoded
`appendable`, every later shard `catable` + `appendable` with the preceding (1 << lgwin) - 16 bytes of the stream as
LZ77 prefix; rank 0 stitches the shards with the BroCatli rules (src/concat/mod.rs:274-608, csrc/concat.cpp).  The only
collective is the gather of the compressed shards -- RCCL (`nccl` backend) between GPUs, `gloo` in the CPU tests.
"""
import ctypes

BROTLI_PARAM_LGWIN = 2
BROTLI_PARAM_CATABLE = 167
BROTLI_PARAM_APPENDABLE = 168
BROTLI_PARAM_MAGIC_NUMBER = 169


def shard_range(total, rank, world):
    """get_range, src/enc/threading/mod.rs:333-335"""
    return (rank * total) // world, ((rank + 1) * total) // world


def shard_window(total, rank, world, lgwin):
    """(first byte of the LZ77 prefix, shard start, shard end) for `rank`"""
    start, end = shard_range(total, rank, world)
    win = (1 << lgwin) - 16  # encode.rs:1231, 1243-1246
    return (max(0, start - win) if rank else start), start, end


def shard_params(params, rank):
    """parameter list of one shard (threading/mod.rs:354-358)"""
    out = list(params) + [(BROTLI_PARAM_APPENDABLE, 1)]
    if rank:
        # compress_part: catable = true, magic_number = false for every shard but the first (threading/mod.rs:354-357)
        out.append((BROTLI_PARAM_CATABLE, 1))
        out.append((BROTLI_PARAM_MAGIC_NUMBER, 0))
    return out


class ShardEncoder(object):
    """Binds the flat entry point brotli_mi355x_encode_stream of a loaded library (product or emulation)."""

    def __init__(self, cdll, segment_bytes=0):
        self.L = cdll
        self.segment_bytes = segment_bytes
        f = self.L.brotli_mi355x_encode_stream
        f.restype = ctypes.c_long
        f.argtypes = [ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_uint32), ctypes.c_size_t, ctypes.c_char_p,
                      ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_uint32,
                      ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_double), ctypes.c_char_p, ctypes.c_size_t]
        self.stats = (ctypes.c_double * 32)()
        self._err = ctypes.create_string_buffer(512)
        self._out = None
        self._out_owner = None

    def use_output_buffer(self, address, nbytes, owner):
        """Have encode() deliver into caller memory (e.g. a page-locked torch tensor, which the library fills by DMA instead of
        bouncing the stream through its own staging buffers); `owner` is kept alive with the encoder."""
        self._out = (ctypes.c_char * nbytes).from_address(address)
        self._out_owner = owner

    def encode(self, params, prefix, chunk, nbytes, on_device, copy=True):
        """chunk: device address (on_device) or bytes.  Returns the compressed shard as bytes -- or, with copy=False, as a
        view of this encoder's output buffer that is valid until its next call (a 20 MB bytes object costs a millisecond)."""
        keys = (ctypes.c_int * len(params))(*[k for k, _ in params])
        vals = (ctypes.c_uint32 * len(params))(*[v for _, v in params])
        cap = nbytes + nbytes // 4 + 4096
        if self._out is None or len(self._out) < cap:
            self._out = ctypes.create_string_buffer(cap)
        if on_device:
            src = ctypes.c_void_p(chunk)
            keep = None
        else:
            keep = ctypes.create_string_buffer(bytes(chunk), max(1, nbytes))
            src = ctypes.cast(keep, ctypes.c_void_p)
        n = self.L.brotli_mi355x_encode_stream(keys, vals, len(params), prefix, len(prefix), 1, src, nbytes, 1 if on_device else 0,
                                               self.segment_bytes, self._out, cap, self.stats, self._err, 512)
        if n < 0:
            raise RuntimeError(self._err.value.decode())
        return ctypes.string_at(self._out, n) if copy else memoryview(self._out)[:n]

    def encode_to_device(self, params, prefix, chunk_ptr, nbytes, out_ptr, out_capacity):
        """input and output both in device memory; returns the compressed size"""
        keys = (ctypes.c_int * len(params))(*[k for k, _ in params])
        vals = (ctypes.c_uint32 * len(params))(*[v for _, v in params])
        n = self.L.brotli_mi355x_encode_stream(keys, vals, len(params), prefix, len(prefix), 1, ctypes.c_void_p(chunk_ptr), nbytes, 3,
                                               self.segment_bytes, ctypes.cast(ctypes.c_void_p(out_ptr), ctypes.c_char_p), out_capacity,
                                               self.stats, self._err, 512)
        if n < 0:
            raise RuntimeError(self._err.value.decode())
        return n


def gather_shards(dist, comp, rank, world, device):
    """Variable-length gather of the compressed shards to rank 0 (sizes first, then padded payloads).
    Returns the list of shards on rank 0, None elsewhere."""
    import torch
    size_t = torch.tensor([len(comp)], dtype=torch.int64, device=device)
    sizes = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(sizes, size_t)
    sizes = [int(s.item()) for s in sizes]
    mx = max(1, max(sizes))
    payload = torch.zeros(mx, dtype=torch.uint8, device=device)
    if len(comp):  # (a rank without a shard in this round contributes an empty payload)
        payload[:len(comp)] = torch.frombuffer(bytearray(comp), dtype=torch.uint8).to(device)
    bufs = [torch.zeros(mx, dtype=torch.uint8, device=device) for _ in range(world)] if rank == 0 else None
    dist.gather(payload, bufs, dst=0)
    if rank != 0:
        return None
    return [bytes(bufs[r][:sizes[r]].cpu().numpy()) for r in range(world)]


def compress_sharded(dist, library, encoder, params, lgwin, prefix, chunk, nbytes, on_device, rank, world, device):
    """One step of the multi-process job: encode this rank's shard, gather, stitch on rank 0 (returns the stream there)."""
    comp = encoder.encode(shard_params(params, rank), prefix, chunk, nbytes, on_device)
    if world == 1:
        return comp
    shards = gather_shards(dist, comp, rank, world, device)
    if rank != 0:
        return None
    return library.concat_chunks(shards)


def compress_multi_over_ranks(dist, library, encoder, params, total, nshards, rank, world, device, shard_input):
    """BrotliEncoderCompressMulti(total bytes, nshards threads) with the shards dealt round-robin to the ranks (shard s ->
    rank s % world): every rank encodes its shards one after the other, each round of `world` shards is gathered to rank 0,
    which stitches all of them at the end.  The stream does not depend on `world`.
    shard_input(s) -> (prefix bytes, chunk (bytes or device address), nbytes, on_device).  Returns the stream on rank 0."""
    collected = {}
    for base in range(0, nshards, world):
        s = base + rank
        comp = b""
        if s < nshards:
            prefix, chunk, nbytes, on_device = shard_input(s)
            comp = encoder.encode(shard_params(params, s), prefix, chunk, nbytes, on_device)
        if world == 1:
            collected[s] = comp
        else:
            got = gather_shards(dist, comp, rank, world, device)
            if rank == 0:
                for r in range(world):
                    if base + r < nshards:
                        collected[base + r] = got[r]
    if rank != 0:
        return None
    return library.concat_chunks([collected[s] for s in range(nshards)])


class DeviceShardJob(object):
    """The same job with the compressed shards kept in HBM until rank 0 has all of them: each rank encodes straight into
    a device buffer, the gather runs GPU to GPU (RCCL over xGMI), rank 0 copies the shard bodies to their place in a
    pinned result and stitches the junctions there.  Buffers are allocated once and reused.

    The job is a two-stage pipeline: gather + copy-out + stitch of step i run on a side stream while the ranks already
    encode step i + 1 (double-buffered shard outputs), so step() hands back the stream of the PREVIOUS step (None the
    first time) and finish() the last one."""

    def __init__(self, dist, library, encoder, rank, world, shard_bytes):
        import torch
        self.dist, self.library, self.encoder, self.rank, self.world = dist, library, encoder, rank, world
        self.cap = shard_bytes + shard_bytes // 4 + 4096
        self.side = torch.cuda.Stream()
        self.out = [torch.zeros(self.cap, dtype=torch.uint8, device="cuda") for _ in range(2)]
        self.size = [torch.zeros(1, dtype=torch.int64, device="cuda") for _ in range(2)]
        self.sizes = [torch.zeros(world, dtype=torch.int64, device="cuda") for _ in range(2)]
        root = rank == 0
        self.rows = [torch.zeros((world, self.cap), dtype=torch.uint8, device="cuda") for _ in range(2)] if root else None
        self.stitched = [torch.zeros(world * self.cap + 64, dtype=torch.uint8).pin_memory() for _ in range(2)] if root else None
        self.ends_host = torch.zeros((world, 16), dtype=torch.uint8).pin_memory() if root else None
        self.lane = torch.arange(8, dtype=torch.int64, device="cuda") if root else None
        self.parity = 0
        self.pending = None  # (event, buffer index, total size) of the step whose result has not been handed out yet

    def _collect(self):
        """waits for the copy-out of the pending step and returns its stream (rank 0) / None"""
        if self.pending is None:
            return None
        event, b, total = self.pending
        self.pending = None
        event.synchronize()
        if self.rank != 0:
            return None
        return memoryview(self.stitched[b].numpy())[:total]

    def step(self, params, prefix, chunk_ptr, nbytes):
        import torch
        b = self.parity
        self.parity ^= 1
        # (the library call returns when the shard is complete in out[b]; the previous step's gather reads out[b ^ 1])
        n = self.encoder.encode_to_device(shard_params(params, self.rank), prefix, chunk_ptr, nbytes, self.out[b].data_ptr(), self.cap)
        previous = self._collect()
        with torch.cuda.stream(self.side):
            self.size[b][0] = n
            self.dist.all_gather_into_tensor(self.sizes[b], self.size[b])
            sizes = [int(v) for v in self.sizes[b].tolist()]
            mx = max(sizes)
            gather_list = [self.rows[b][r, :mx] for r in range(self.world)] if self.rank == 0 else None
            self.dist.gather(self.out[b][:mx], gather_list, dst=0)
            total = 0
            if self.rank == 0:
                # the stitcher only looks at the first and last bytes of every shard: fetch those, let it write the
                # junction bytes into the pinned result, then copy every shard body from HBM to its place in the result
                rows = self.rows[b]
                tail_at = torch.clamp(self.sizes[b] - 8, min=0)[:, None] + self.lane[None, :]
                ends = torch.cat([rows[:, :8], torch.gather(rows, 1, tail_at)], dim=1)
                self.ends_host.copy_(ends)
                total, bodies = self.library.concat_chunk_ends(self.ends_host[:, :8].contiguous(), self.ends_host[:, 8:].contiguous(), sizes,
                                                               self.stitched[b].data_ptr(), self.stitched[b].numel())
                for r, (dst, src, count) in enumerate(bodies):
                    if count:
                        self.stitched[b][dst:dst + count].copy_(rows[r, src:src + count], non_blocking=True)
            event = torch.cuda.Event()
            event.record(self.side)
        self.pending = (event, b, total)
        return previous

    def finish(self):
        return self._collect()
