"""ctypes binding of the C ABI in include/lz4amd.h and include/lz4.h.

Mirrors the reference's block-table driving code (programs/bench.c:347-355 blockParam_t,
466-480 compress loop, 522-542 decompress loop): a BlockTable is the list of
(src pointer, src size, dst pointer, dst capacity) rows, a Plan binds it to the device.
"""
import ctypes
import os

OP_COMPRESS, OP_DECOMPRESS, OP_COMPRESS_HC, OP_XXH32, OP_GATHER = 0, 1, 2, 3, 4

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class Lz4AmdError(RuntimeError):
    pass


def lib_path():
    # LZ4AMD_LIB: developer override (kernel experiments built next to the product library)
    return os.environ.get("LZ4AMD_LIB") or os.path.join(_HERE, "liblz4_amd.so")


def lib():
    """Load liblz4_amd.so (built in-tree by `python -m lz4_amd.build`).  Fails loudly."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise Lz4AmdError(f"{path} is missing: run `python -m lz4_amd.build` (there is no fallback codec)")
    L = ctypes.CDLL(path)
    vp, ip, i, fp = ctypes.c_void_p, ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.POINTER(ctypes.c_float)
    L.lz4amd_ctx_create.argtypes = [ctypes.POINTER(vp), i]
    L.lz4amd_ctx_destroy.argtypes = [vp]
    L.lz4amd_last_error.restype = ctypes.c_char_p
    L.lz4amd_device_cus.argtypes = [vp]
    L.lz4amd_compress_bound.argtypes = [i]
    L.lz4amd_plan_create.argtypes = [vp, ctypes.POINTER(vp), i, i, ctypes.POINTER(vp), ip, ctypes.POINTER(vp), ip, i]
    L.lz4amd_plan_destroy.argtypes = [vp]
    L.lz4amd_plan_launch.argtypes = [vp, vp]
    L.lz4amd_plan_launch_timed.argtypes = [vp, vp, fp, fp]
    L.lz4amd_plan_results.argtypes = [vp, ip, vp]
    L.lz4amd_plan_device_results.argtypes = [vp]
    L.lz4amd_plan_device_results.restype = vp
    for name in ("LZ4_compress_default", "LZ4_decompress_safe"):
        getattr(L, name).argtypes = [ctypes.c_char_p, ctypes.c_char_p, i, i]
    L.LZ4_compress_fast.argtypes = [ctypes.c_char_p, ctypes.c_char_p, i, i, i]
    L.LZ4_compress_HC.argtypes = [ctypes.c_char_p, ctypes.c_char_p, i, i, i]
    L.LZ4_compress_HC_extStateHC.argtypes = [vp, ctypes.c_char_p, ctypes.c_char_p, i, i, i]
    L.LZ4_compressBound.argtypes = [i]
    L.LZ4_versionString.restype = ctypes.c_char_p
    L.lz4amd_plan_create_decompress_chained.argtypes = [vp, ctypes.POINTER(vp), i, ctypes.POINTER(vp), ip, vp, ip, ctypes.c_char_p, i]
    L.lz4amd_plan_create_prefix.argtypes = [vp, ctypes.POINTER(vp), i, ctypes.POINTER(vp), ip, ctypes.POINTER(vp), ip, ip]
    L.lz4amd_hint_bytes.argtypes = [i]
    L.lz4amd_hint_bytes.restype = ctypes.c_size_t
    L.lz4amd_plan_attach_hints.argtypes = [vp, vp, ctypes.c_size_t]
    L.lz4amd_plan_hint_stats.argtypes = [vp, ctypes.POINTER(ctypes.c_uint), ctypes.POINTER(ctypes.c_uint)]
    L.lz4amd_plan_chain_stats.argtypes = [vp, ctypes.POINTER(ctypes.c_ulonglong)]
    L.lz4amd_plan_set_acceleration.argtypes = [vp, i]
    L.lz4amd_plan_make_hints.argtypes = [vp, i]
    L.lz4amd_plan_hints_made.argtypes = [vp, ctypes.POINTER(ctypes.c_uint)]
    _LIB = L
    return L


def compress_bound(n):
    return lib().lz4amd_compress_bound(int(n))


def hint_bytes(n):
    """Bytes of the entry-point table of a block of n source bytes (include/lz4amd.h)."""
    return int(lib().lz4amd_hint_bytes(int(n)))


def _check(rc, what):
    if rc != 0:
        msg = lib().lz4amd_last_error()
        raise Lz4AmdError(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")


class Context:
    """Per-device context (lz4amd_ctx)."""

    def __init__(self, device=0):
        self._h = ctypes.c_void_p()
        _check(lib().lz4amd_ctx_create(ctypes.byref(self._h), int(device)), "lz4amd_ctx_create")
        self.device = device

    @property
    def cus(self):
        return lib().lz4amd_device_cus(self._h)

    def close(self):
        if self._h:
            lib().lz4amd_ctx_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class BlockTable:
    """Rows of (device src pointer, src size, device dst pointer, dst capacity)."""

    def __init__(self, src_ptrs, src_sizes, dst_ptrs, dst_caps):
        n = len(src_ptrs)
        assert len(src_sizes) == n and len(dst_ptrs) == n and len(dst_caps) == n
        self.n = n
        self.src_ptrs = (ctypes.c_void_p * n)(*[int(p) for p in src_ptrs])
        self.dst_ptrs = (ctypes.c_void_p * n)(*[int(p) for p in dst_ptrs])
        self.src_sizes = (ctypes.c_int * n)(*[int(s) for s in src_sizes])
        self.dst_caps = (ctypes.c_int * n)(*[int(s) for s in dst_caps])


class Plan:
    """A block table bound to the device (lz4amd_plan)."""

    def __init__(self, ctx, op, table, level=0):
        self._h = ctypes.c_void_p()
        self.ctx, self.op, self.table = ctx, op, table
        _check(lib().lz4amd_plan_create(ctx._h, ctypes.byref(self._h), int(op), table.n,
                                        table.src_ptrs, table.src_sizes, table.dst_ptrs, table.dst_caps,
                                        int(level)), "lz4amd_plan_create")

    @classmethod
    def chained(cls, ctx, src_ptrs, src_sizes, dst0, dst_caps, stored=None, initial_prefix=0):
        """Dependent blocks in one launch (lz4amd_plan_create_decompress_chained): packed output at dst0."""
        self = cls.__new__(cls)
        self._h = ctypes.c_void_p()
        self.ctx, self.op = ctx, OP_DECOMPRESS
        self.table = BlockTable(src_ptrs, src_sizes, [dst0] * len(src_ptrs), dst_caps)
        flags = bytes(bytearray(1 if s else 0 for s in stored)) if stored is not None else None
        _check(lib().lz4amd_plan_create_decompress_chained(ctx._h, ctypes.byref(self._h), self.table.n, self.table.src_ptrs,
                                                           self.table.src_sizes, ctypes.c_void_p(int(dst0)), self.table.dst_caps,
                                                           flags, int(initial_prefix)), "lz4amd_plan_create_decompress_chained")
        return self

    @classmethod
    def compress_with_history(cls, ctx, table, prefix_sizes):
        """LZ4AMD_OP_COMPRESS where block i may reference the prefix_sizes[i] bytes of source right before it (linked blocks:
        lz4amd_plan_create_compress_prefix)."""
        self = cls.__new__(cls)
        self._h = ctypes.c_void_p()
        self.ctx, self.op, self.table = ctx, OP_COMPRESS, table
        pre = (ctypes.c_int * table.n)(*[int(x) for x in prefix_sizes])
        L = lib()
        L.lz4amd_plan_create_compress_prefix.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.POINTER(ctypes.c_void_p),
                                                         ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
        _check(L.lz4amd_plan_create_compress_prefix(ctx._h, ctypes.byref(self._h), table.n, table.src_ptrs, table.src_sizes,
                                                    table.dst_ptrs, table.dst_caps, pre), "lz4amd_plan_create_compress_prefix")
        return self

    def attach_hints(self, d_hints, stride):
        """Entry-point tables: block i's at d_hints + i * stride (written by a compress plan, read by a decompress plan)."""
        _check(lib().lz4amd_plan_attach_hints(self._h, ctypes.c_void_p(int(d_hints) if d_hints else None), int(stride)), "lz4amd_plan_attach_hints")

    def hint_stats(self):
        """(blocks decoded from their table, tables rejected) since the tables were attached."""
        u, r = ctypes.c_uint(), ctypes.c_uint()
        _check(lib().lz4amd_plan_hint_stats(self._h, ctypes.byref(u), ctypes.byref(r)), "lz4amd_plan_hint_stats")
        return u.value, r.value

    def chain_stats(self):
        """Side-by-side plan of dependent blocks, after a launch: dict(units, units_decoded_three_times, bytes_walked_by_patch, bytes_of_units_1_on)."""
        v = (ctypes.c_ulonglong * 4)()
        _check(lib().lz4amd_plan_chain_stats(self._h, v), "lz4amd_plan_chain_stats")
        return {"units": v[0], "units_decoded_three_times": v[1], "bytes_walked_by_patch": v[2], "bytes_of_units_1_on": v[3]}

    def make_hints(self, on=True):
        """Decompress plan with tables attached: blocks without a usable table get theirs written while they are decoded."""
        _check(lib().lz4amd_plan_make_hints(self._h, 1 if on else 0), "lz4amd_plan_make_hints")

    def hints_made(self):
        m = ctypes.c_uint()
        _check(lib().lz4amd_plan_hints_made(self._h, ctypes.byref(m)), "lz4amd_plan_hints_made")
        return m.value

    def set_acceleration(self, acceleration):
        _check(lib().lz4amd_plan_set_acceleration(self._h, int(acceleration)), "lz4amd_plan_set_acceleration")

    def launch(self, stream=0):
        _check(lib().lz4amd_plan_launch(self._h, ctypes.c_void_p(stream)), "lz4amd_plan_launch")

    def launch_timed(self, stream=0):
        """Run once with HIP events around every kernel; returns (kernel_ms[4], total_ms)."""
        k = (ctypes.c_float * 4)()
        t = ctypes.c_float()
        _check(lib().lz4amd_plan_launch_timed(self._h, ctypes.c_void_p(stream), k, ctypes.byref(t)),
               "lz4amd_plan_launch_timed")
        return list(k), t.value

    def results(self, stream=0):
        out = (ctypes.c_int * self.table.n)()
        _check(lib().lz4amd_plan_results(self._h, out, ctypes.c_void_p(stream)), "lz4amd_plan_results")
        return list(out)

    def close(self):
        if self._h:
            lib().lz4amd_plan_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _stream_handle(stream):
    if stream is None:
        import torch
        return torch.cuda.current_stream().cuda_stream
    return getattr(stream, "cuda_stream", stream)


def compress_blocks(ctx, data, block_size, stream=None, hc_level=None, hints=None, acceleration=1):
    """Compress a CUDA uint8 tensor as independent blocks of `block_size` bytes (LZ4_compress_default
    per block, or LZ4_compress_HC at `hc_level` when given).  hints: a CUDA uint8 tensor [n, hint_bytes(block_size)]
    that receives the blocks' entry-point tables.
    Returns (comp tensor [n, stride], sizes list, plan)."""
    import torch
    assert data.is_cuda and data.dtype == torch.uint8 and data.dim() == 1
    total = data.numel()
    n = (total + block_size - 1) // block_size
    stride = (compress_bound(block_size) + 255) & ~255
    comp = torch.empty((max(n, 1), stride), dtype=torch.uint8, device=data.device)
    base, cbase = data.data_ptr(), comp.data_ptr()
    sizes = [min(block_size, total - i * block_size) for i in range(n)]
    table = BlockTable([base + i * block_size for i in range(n)], sizes,
                       [cbase + i * stride for i in range(n)], [stride] * n)
    plan = Plan(ctx, OP_COMPRESS, table) if hc_level is None else Plan(ctx, OP_COMPRESS_HC, table, level=hc_level)
    if hints is not None:
        plan.attach_hints(hints.data_ptr(), hints.stride(0))
    if acceleration != 1:
        plan.set_acceleration(acceleration)
    s = _stream_handle(stream)
    plan.launch(s)
    return comp, plan.results(s), plan


def decompress_blocks(ctx, comp, csizes, block_size, total, stream=None, hints=None):
    """Inverse of compress_blocks (hints: the tables compress_blocks filled).  Returns (out tensor [total], results list, plan)."""
    import torch
    n = len(csizes)
    out = torch.empty(max(total, 1), dtype=torch.uint8, device=comp.device)
    stride = comp.stride(0)
    caps = [min(block_size, total - i * block_size) for i in range(n)]
    table = BlockTable([comp.data_ptr() + i * stride for i in range(n)], csizes,
                       [out.data_ptr() + i * block_size for i in range(n)], caps)
    plan = Plan(ctx, OP_DECOMPRESS, table)
    if hints is not None:
        plan.attach_hints(hints.data_ptr(), hints.stride(0))
    s = _stream_handle(stream)
    plan.launch(s)
    return out[:total], plan.results(s), plan
