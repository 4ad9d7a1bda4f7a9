"""Python host side of the MI355X FlashAttention (forward and backward): the reference's operator entry
point re-implemented over the C-ABI HIP library (``include/fa_mi355.h``).

Mirrors:
  * ``flash_attention(q, k, v, causal=False)``   code/triton_fa2/FA2-triton.py:240-244
    (positional (B,H,N,D) tensors; fp32 inputs are computed in fp16 and cast back :241-244)
  * ``_FlashAttnFn.forward``                      code/triton_fa2/FA2-triton.py:175-205
    (asserts ``is_cuda`` and ``D % 16 == 0 and D <= 128`` :176-178; allocates O and the
    softmax statistics; launches on the current stream; scale 1/sqrt(D) :183)

The north-star spells the entry ``flash_attn``; the reference spells it ``flash_attention``.
Both names are exported and are the same function.

``FlashAttnFn`` mirrors ``_FlashAttnFn`` including ``backward`` (FA2-triton.py:207-237).

There is NO CPU fallback: CPU tensors raise (as the reference's ``assert q.is_cuda`` does) and a
missing HIP library raises ``RuntimeError``.
"""
from __future__ import annotations

import ctypes
import math
import os
import threading

import torch

from . import _build

FA_DTYPE_BF16 = 0
FA_DTYPE_FP16 = 1
FA_DTYPE_FP8_E4M3 = 2
REFERENCE_MAX_HEAD_DIM = 128      # FA2-triton.py:178; the fused backward kernels stop here too (144 .. 256: _bwd_wide)
MAX_HEAD_DIM = 256                # forward, bf16 / fp16 (csrc/fa_fwd_kernel_wide.hpp)



class FlashAttnArgumentError(ValueError, AssertionError):
    """Bad argument to flash_attn.  Subclasses AssertionError because the reference signals
    the same conditions with ``assert`` (FA2-triton.py:176-178)."""


_lib_lock = threading.Lock()
_lib_handle = None


class _trace_range:
    """Marker range around a launch, like the reference's NVTX ranges "FA2_FWD" / "FA2_BWD"
    (FA2-triton.py:186,201,218,236).  On ROCm ``torch.cuda.nvtx`` maps to roctx, which rocprofv3 records with
    ``--marker-trace``.  Off unless FA_MI355_TRACE=1: the push/pop pair costs a few microseconds per call."""
    enabled = os.environ.get("FA_MI355_TRACE", "0") == "1"

    def __init__(self, name: str):
        self.name = name
        self.pushed = False

    def __enter__(self):
        if _trace_range.enabled:
            try:
                torch.cuda.nvtx.range_push(self.name)
                self.pushed = True
            except Exception:  # noqa: BLE001  (markers are best effort: a torch build without roctx just skips them)
                self.pushed = False
        return self

    def __exit__(self, *exc):
        if self.pushed:
            torch.cuda.nvtx.range_pop()
        return False


def _declare(lib):
    c = ctypes
    lib.fa_version.restype = c.c_int
    lib.fa_version.argtypes = []
    lib.fa_supported.restype = c.c_int
    lib.fa_supported.argtypes = [c.c_int, c.c_int]
    lib.fa_last_error.restype = c.c_char_p
    lib.fa_last_error.argtypes = []
    lib.fa_fwd.restype = c.c_int
    lib.fa_fwd.argtypes = [c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p,
                           c.c_int, c.c_int, c.c_int, c.c_int,
                           c.POINTER(c.c_int64), c.POINTER(c.c_int64),
                           c.POINTER(c.c_int64), c.POINTER(c.c_int64),
                           c.c_int, c.c_int, c.c_float,
                           c.POINTER(c.c_float), c.c_void_p]
    lib.fa_fp8_workspace_bytes.restype = c.c_size_t
    lib.fa_fp8_workspace_bytes.argtypes = [c.c_int, c.c_int, c.c_int, c.c_int]
    lib.fa_fp8_pv_native.restype = c.c_int
    lib.fa_fp8_pv_native.argtypes = []
    lib.fa_fwd_fp8.restype = c.c_int
    lib.fa_fwd_fp8.argtypes = [c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p,
                               c.c_int, c.c_int, c.c_int, c.c_int,
                               c.POINTER(c.c_int64), c.POINTER(c.c_int64),
                               c.POINTER(c.c_int64), c.POINTER(c.c_int64),
                               c.c_int, c.c_float, c.POINTER(c.c_float),
                               c.c_void_p, c.c_size_t, c.c_void_p]
    lib.fa_bwd_workspace_bytes.restype = c.c_size_t
    lib.fa_bwd_workspace_bytes.argtypes = [c.c_int, c.c_int, c.c_int]
    lib.fa_bwd.restype = c.c_int
    lib.fa_bwd.argtypes = [c.c_void_p] * 9 + [c.c_int] * 4 + [c.POINTER(c.c_int64)] * 8 + \
        [c.c_int, c.c_int, c.c_float, c.c_void_p, c.c_size_t, c.c_void_p]
    # extended variants: (B, H, H_kv, S_q, S_k, D) instead of (B, H, S, D)
    lib.fa_fwd_ex.restype = c.c_int
    lib.fa_fwd_ex.argtypes = lib.fa_fwd.argtypes[:5] + [c.c_int] * 6 + lib.fa_fwd.argtypes[9:]
    lib.fa_fwd_fp8_ex.restype = c.c_int
    lib.fa_fwd_fp8_ex.argtypes = lib.fa_fwd_fp8.argtypes[:5] + [c.c_int] * 6 + lib.fa_fwd_fp8.argtypes[9:]
    lib.fa_bwd_ex_workspace_bytes.restype = c.c_size_t
    lib.fa_bwd_ex_workspace_bytes.argtypes = [c.c_int] * 6
    if hasattr(lib, "fa_bwd_ds_workspace_bytes"):      # (FA_VERSION >= 133; A/B arms built from older sources lack it: recompute path)
        lib.fa_bwd_ds_workspace_bytes.restype = c.c_size_t
        lib.fa_bwd_ds_workspace_bytes.argtypes = [c.c_int] * 6
    lib.fa_bwd_ex.restype = c.c_int
    lib.fa_bwd_ex.argtypes = lib.fa_bwd.argtypes[:9] + [c.c_int] * 6 + lib.fa_bwd.argtypes[13:]
    lib.fa_fwd_dispatch.restype = c.c_int
    lib.fa_fwd_dispatch.argtypes = [c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p,
                                    c.c_int, c.c_int, c.c_int, c.c_int, c.c_int, c.c_void_p]
    lib.fa_fwd_launch_info.restype = c.c_int
    lib.fa_fwd_launch_info.argtypes = [c.c_int, c.c_int, c.c_int, c.c_int, c.c_int, c.c_int,
                                       c.POINTER(c.c_int), c.POINTER(c.c_int), c.POINTER(c.c_int)]
    if hasattr(lib, "fa_bwd_wide_ds"):         # (FA_VERSION >= 136: backward for head_dim 144 .. 256)
        lib.fa_bwd_wide_workspace_bytes.restype = c.c_size_t
        lib.fa_bwd_wide_workspace_bytes.argtypes = [c.c_int] * 3
        lib.fa_bwd_wide_ds.restype = c.c_int
        lib.fa_bwd_wide_ds.argtypes = [c.c_void_p] * 6 + [c.c_void_p, c.c_void_p, c.c_longlong] + [c.c_int] * 6 + [c.c_void_p] * 5 + \
            [c.c_int, c.c_int, c.c_float, c.c_void_p, c.c_size_t, c.c_void_p]
    if hasattr(lib, "fa_build_is_default"):    # (FA_VERSION >= 134)
        lib.fa_build_is_default.restype = c.c_int
        lib.fa_build_is_default.argtypes = []
    if hasattr(lib, "fa_diag_mfma_loop"):      # (diagnostics, FA_VERSION >= 132; A/B arms built from older sources lack them)
        lib.fa_device_cus.restype = c.c_int
        lib.fa_device_cus.argtypes = []
        lib.fa_diag_mfma_loop.restype = c.c_int
        lib.fa_diag_mfma_loop.argtypes = [c.c_int, c.c_int, c.c_void_p, c.c_void_p, c.POINTER(c.c_double), c.c_void_p]
    return lib


def load_library(path: str | None = None):
    """Load (building first if the in-tree .so is missing or stale and hipcc is available)
    the C-ABI library.  Raises RuntimeError if it cannot be produced -- never falls back."""
    global _lib_handle
    with _lib_lock:
        if _lib_handle is not None and path is None:
            return _lib_handle
        env_path = os.environ.get("FA_MI355_LIB") if path is None else None   # developer aid: profile another build of the library
        p = path or env_path or _build.LIB_PATH
        if path is None and env_path is None and _build.is_stale():
            try:
                _build.build()
            except Exception as e:  # noqa: BLE001
                if not os.path.exists(p):
                    raise RuntimeError(
                        f"HIP library {p} is missing and could not be built ({e}); "
                        "run `python -c 'import __graft_entry__ as g; g.build()'`") from e
        try:
            lib = _declare(ctypes.CDLL(p))
        except OSError as e:
            raise RuntimeError(f"cannot load HIP library {p}: {e}") from e
        if path is None:
            _lib_handle = lib
        return lib


def _dtype_code(dt: torch.dtype) -> int:
    if dt == torch.bfloat16:
        return FA_DTYPE_BF16
    if dt == torch.float16:
        return FA_DTYPE_FP16
    if dt == torch.float8_e4m3fn:
        return FA_DTYPE_FP8_E4M3
    raise FlashAttnArgumentError(f"unsupported dtype {dt}; expected bfloat16, float16, float32 "
                                 "(computed in float16 like the reference) or float8_e4m3fn "
                                 "(with descale=(q,k,v) per-tensor scales)")


def _strides3(t: torch.Tensor):
    """Element strides (batch, head, seq) for the C ABI; None (= NULL = contiguous) skips building the array, which is
    most of the host-side cost of a small launch."""
    if t.is_contiguous():
        return None
    sb, sh, ss, sd = t.stride()
    return (ctypes.c_int64 * 3)(sb, sh, ss)


def _kernel_ready(t: torch.Tensor) -> torch.Tensor:
    """Return t itself if the kernel can address it (unit head_dim stride, 16-byte aligned
    rows), otherwise a contiguous copy (the reference passes arbitrary strides to Triton,
    FA2-triton.py:190-193; the copy keeps that contract)."""
    if t.is_contiguous() and t.data_ptr() % 16 == 0:     # (head_dim % 16 == 0: every row of a contiguous tensor is 16-byte aligned)
        return t
    es = t.element_size()
    ok = (t.stride(3) == 1 and t.data_ptr() % 16 == 0
          and all((s * es) % 16 == 0 and s >= 0 for s in t.stride()[:3])
          and t.stride(2) >= t.shape[3])
    return t if ok else t.contiguous()


def check_args(q, k, v) -> None:
    """Argument validation shared with tests (pure host logic, no GPU needed)."""
    for name, t in (("q", q), ("k", k), ("v", v)):
        if not isinstance(t, torch.Tensor):
            raise TypeError(f"{name} must be a torch.Tensor, got {type(t).__name__}")
    if q.dim() != 4:
        raise FlashAttnArgumentError(f"q must be (B, H, N, D); got shape {tuple(q.shape)}")
    # the reference takes (B, H, N, D) from q for all three tensors (FA2-triton.py:176).  This build's extensions:
    # grouped-query attention (k, v with H_kv heads, H % H_kv == 0: query head h uses key/value head h // (H // H_kv))
    # and a key/value length of its own (k, v with N_k >= 1 rows; the causal mask is then bottom-right aligned)
    if k.dim() != 4 or v.shape != k.shape or k.shape[0] != q.shape[0] or k.shape[3] != q.shape[3] \
            or (q.shape[1] != 0 if k.shape[1] == 0 else q.shape[1] % k.shape[1] != 0) \
            or (k.shape[2] == 0 and q.shape[2] != 0):
        raise FlashAttnArgumentError(
            "q, k, v must have identical shapes (or k, v one shape (B, H_kv, N_k, D) with H_kv dividing q's head count); "
            f"got {tuple(q.shape)}, {tuple(k.shape)}, {tuple(v.shape)}")
    if k.dtype != q.dtype or v.dtype != q.dtype:
        raise FlashAttnArgumentError(f"q, k, v must share a dtype; got {q.dtype}, {k.dtype}, {v.dtype}")
    if k.device != q.device or v.device != q.device:
        raise FlashAttnArgumentError("q, k, v must be on the same device")
    if not q.is_cuda:                                   # FA2-triton.py:176
        raise FlashAttnArgumentError(
            "flash_attn needs GPU (ROCm 'cuda') tensors; there is no CPU path "
            "(reference: assert q.is_cuda, FA2-triton.py:176)")
    D = q.shape[3]
    if D % 16 != 0 or D > MAX_HEAD_DIM:                 # FA2-triton.py:178 (its bound is 128; 144 .. 256 is this library's extension)
        raise FlashAttnArgumentError(f"head_dim must satisfy D % 16 == 0 and D <= {MAX_HEAD_DIM}; got {D}")


class _on_device:
    """The launch context of a tensor's device: makes it current only if it is not already (``torch.cuda.device`` costs a few
    microseconds per call even then -- a third of the host time of a small launch) and hands out the raw current stream."""
    __slots__ = ("idx", "prev")

    def __init__(self, device: torch.device):
        self.idx = device.index if device.index is not None else torch.cuda.current_device()
        self.prev = -1

    def __enter__(self):
        cur = torch.cuda.current_device()
        if cur != self.idx:
            self.prev = cur
            torch.cuda.set_device(self.idx)
        return torch._C._cuda_getCurrentRawStream(self.idx)

    def __exit__(self, *exc):
        if self.prev >= 0:
            torch.cuda.set_device(self.prev)
        return False


def _fwd_raw(lib, q, k, v, causal: bool, scale: float, descale, want_lse: bool):
    """Launch the forward on kernel-ready tensors (head_dim 64 or 128).  Returns (o, lse or None)."""
    code = _dtype_code(q.dtype)
    B, H, N, D = q.shape
    Hkv, Nk = k.shape[1], k.shape[2]
    out_dtype = torch.bfloat16 if code == FA_DTYPE_FP8_E4M3 else q.dtype
    o = torch.empty((B, H, N, D), dtype=out_dtype, device=q.device)        # FA2-triton.py:179
    lse = torch.empty((B, H, N), dtype=torch.float32, device=q.device) if want_lse else None
    if B * H * N == 0:
        return o, lse
    dsc = (ctypes.c_float * 3)(*descale) if descale is not None else None
    with _on_device(q.device) as stream, _trace_range("FA2_FWD"):
        lse_ptr = lse.data_ptr() if lse is not None else None
        if code == FA_DTYPE_FP8_E4M3:
            nbytes = lib.fa_fp8_workspace_bytes(B, H, max(N, Nk), D)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=q.device)
            rc = lib.fa_fwd_fp8_ex(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), lse_ptr,
                                   B, H, Hkv, N, Nk, D, _strides3(q), _strides3(k), _strides3(v), _strides3(o),
                                   1 if causal else 0, scale, dsc, ws.data_ptr(), nbytes, stream)
        else:
            rc = lib.fa_fwd_ex(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), lse_ptr,
                               B, H, Hkv, N, Nk, D, _strides3(q), _strides3(k), _strides3(v), _strides3(o),
                               code, 1 if causal else 0, scale, dsc, stream)
    if rc != 0:
        raise RuntimeError(f"fa_fwd failed ({rc}): {lib.fa_last_error().decode()}")
    return o, lse


_ds_refused: dict = {}          # (device index, bytes) of hand-off workspaces that did not fit -> [calls left before the next look, current back-off]
_plan_cache: dict = {}          # (shape, device, CU count, environment switches) -> (batches per launch, heads per launch, hand-off bytes or 0, recompute bytes)
_ds_granted: set = set()       # (device index, bytes) of hand-off workspaces the allocator has provided before
_ws_poison = False              # test hook (tests/test_bwd_ds_gpu.py): fill every backward workspace with 0xFF before it is used


def _equal_parts(n: int, fit: int) -> int:
    """Largest part size <= fit that splits n into the fewest, equal-as-possible parts (40 with 16 fitting -> 14: 14, 14, 12)."""
    return -(-n // -(-n // fit))


def _ds_fits(device, nbytes: int) -> bool:
    """Can the hand-off workspace be had WITHOUT pushing the caching allocator into an out-of-memory retry (which frees every
    cached block: device synchronisations and hipFree calls, a stall of a training step near its memory limit)?  Yes if the
    allocator already holds that much unused (reserved - allocated: the block of the previous call, typically), else if the
    driver reports that much free plus a margin.  The driver query costs about a launch: it only runs when the cache has no
    room, i.e. on the first call of a shape and after the block was given up."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if torch.cuda.memory_reserved(idx) - torch.cuda.memory_allocated(idx) >= nbytes:
        return True
    free, _total = torch.cuda.mem_get_info(idx)
    return free >= nbytes + (256 << 20)


def _bwd_plan(lib, dims, device):
    """How one backward runs: ``(batches per launch, query heads per launch, workspace, its size)``.

    The dS hand-off (include/fa_mi355.h: fa_bwd_ds_workspace_bytes -- 2 S_q S_k bytes per query head on top of the row
    statistics) is taken where the library says it qualifies and pays.  It is a TRANSIENT allocation at the point of the
    backward where activations are largest: 8 GiB at (8,32,4096,128), kept under FA_MI355_BWD_DS_MAX_GIB (default 16 of the
    288 GB): a launch that needs more is split into equal chunks that run one after the other through the same workspace --
    whole batches first; where one batch alone does not fit (long sequences: 16 heads at S = 32768 are 32 GiB), groups of
    query heads that share a key/value head.  (batch, head) slices are independent, and a chunk of several GiB of dS is a
    launch of milliseconds.  If not even one group fits, FA_MI355_BWD_DS=0, or the memory is not there (`_ds_fits`: decided
    from the allocator's own figures, never by provoking an out-of-memory retry; looked at again after 1, 2, 4 ... 1024
    calls): the recompute path's small workspace and one launch.  The size rules are pure functions of the shape, the device
    and its CU count: computed once per (shape, device)."""
    B, H, Hkv = dims[0], dims[1], dims[2]
    dev_idx = device.index if device.index is not None else torch.cuda.current_device()
    cus = lib.fa_device_cus() if hasattr(lib, "fa_device_cus") else 256          # (of the current device: the caller runs under _on_device)
    key = (dims, dev_idx, cus, os.environ.get("FA_MI355_BWD_DS", "1"), os.environ.get("FA_MI355_BWD_DS_MAX_GIB", "16"),
           os.environ.get("FA_MI355_BWD_DS_MIN_BLOCKS_PER_CU", "2"))
    sizes = _plan_cache.get(key)
    if sizes is None:
        bc, hc, big = B, H, 0
        if key[3] != "0" and hasattr(lib, "fa_bwd_ds_workspace_bytes"):
            cap = float(key[4]) * 2 ** 30
            big = lib.fa_bwd_ds_workspace_bytes(*dims)
            if big > cap:
                per = lib.fa_bwd_ds_workspace_bytes(1, *dims[1:])
                G = H // Hkv
                if 0 < per <= cap:
                    bc = _equal_parts(B, min(B, int(cap // per)))
                    big = lib.fa_bwd_ds_workspace_bytes(bc, *dims[1:])
                elif per > cap and 0 < lib.fa_bwd_ds_workspace_bytes(1, G, 1, *dims[3:]) <= cap:
                    groups = _equal_parts(Hkv, max(1, min(Hkv, int(cap // lib.fa_bwd_ds_workspace_bytes(1, G, 1, *dims[3:])))))
                    bc, hc = 1, groups * G
                    big = lib.fa_bwd_ds_workspace_bytes(1, hc, groups, *dims[3:])
                    # a chunk must still fill the chip: with fewer than two 256-row query blocks per CU the launches of a chunk run
                    # half empty and the split costs more than the hand-off gains ((1,8,65536,128) head by head: -1.4 %)
                    if hc * ((dims[3] + 255) // 256) < float(key[5]) * cus:
                        big = 0
                else:
                    big = 0
            if not 0 < big <= cap:
                bc, hc, big = B, H, 0
        if len(_plan_cache) >= 4096:        # (a bound, not a policy: shapes seen by one process are few)
            _plan_cache.clear()
        sizes = _plan_cache[key] = (bc, hc, big, lib.fa_bwd_ex_workspace_bytes(*dims))
    bc, hc, big, small = sizes
    ws = None
    if big:
        state = _ds_refused.get((dev_idx, big))
        if state is not None and state[0] > 0:      # did not fit a moment ago: the recompute path for the next calls, then one more look
            state[0] -= 1
        else:
            # the allocator's figures are consulted on the FIRST call of a (device, size) and after a refusal only (they cost
            # up to a millisecond: torch builds its whole statistics table); once a workspace of this size has been had, the
            # caching allocator hands the same block back call after call and the hot path is one torch.empty
            ok = (dev_idx, big) in _ds_granted or _ds_fits(device, big)
            if ok:
                try:
                    ws = torch.empty(big, dtype=torch.uint8, device=device)
                except torch.cuda.OutOfMemoryError:          # (memory got tight since, or fragmentation)
                    ok = False
            if ok:
                _ds_granted.add((dev_idx, big))
                _ds_refused.pop((dev_idx, big), None)
            else:
                _ds_granted.discard((dev_idx, big))
                back = min(1024, 2 * state[1]) if state is not None else 1
                _ds_refused[(dev_idx, big)] = [back, back]
    if ws is None:
        bc, hc, big = B, H, 0
        ws = torch.empty(small, dtype=torch.uint8, device=device)
    if _ws_poison:
        ws.fill_(0xFF)
    return bc, hc, ws, big or small


def bwd_plan_info(q_shape, kv_shape, device=None) -> dict:
    """What `_bwd_plan` decides for a backward of these (B, H, N, D) / (B, H_kv, N_k, D) shapes on `device` right now (bench.py's
    `roofline.handoff`, tools/report_all.py): {"handoff": bool, "chunks": launches, "workspace_bytes": transient allocation,
    "ds_bytes_full": the hand-off image of the whole launch or 0 where the library declines it}."""
    lib = load_library()
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    B, H, N, D = q_shape
    Hkv, Nk = kv_shape[1], kv_shape[2]
    dims = (B, H, Hkv, N, Nk, D)
    with _on_device(device):
        bc, hc, ws, nbytes = _bwd_plan(lib, dims, device)
        full = lib.fa_bwd_ds_workspace_bytes(*dims) if hasattr(lib, "fa_bwd_ds_workspace_bytes") else 0
    small = lib.fa_bwd_ex_workspace_bytes(*dims)
    del ws
    return {"handoff": nbytes > small, "chunks": -(-B // bc) * -(-H // hc), "workspace_bytes": int(nbytes), "ds_bytes_full": int(full)}


def _bwd_wide(lib, q, k, v, o, lse, do, causal: bool, scale: float):
    """Backward for head_dim 144 .. 256 (include/fa_mi355.h, fa_bwd_wide_ds): the HIP kernel writes the P and the scaled dS image
    of a chunk of heads, three library GEMMs (torch.bmm -> hipBLASLt) finish it -- dV = P^T dO, dK = dS^T Q, dQ = dS K, the
    query heads of a key/value group stacked along the rows so that dK / dV come out summed over the group.  Chunks of whole
    key/value groups keep the two images under FA_MI355_BWD_WIDE_MAX_GIB (default 4) of transient memory."""
    if not hasattr(lib, "fa_bwd_wide_ds"):
        raise RuntimeError("this build of the library has no backward for head_dim > 128 (fa_bwd_wide_ds, FA_VERSION >= 136)")
    code = _dtype_code(q.dtype)
    B, H, N, D = q.shape
    Hkv, Nk = k.shape[1], k.shape[2]
    G = H // Hkv
    dq = torch.empty_like(o)
    dk = torch.empty((B, Hkv, Nk, D), dtype=o.dtype, device=o.device)
    dv = torch.empty_like(dk)
    if B * H * N == 0:
        return dq, dk.zero_(), dv.zero_()
    do = _kernel_ready(do.to(q.dtype))
    ld = -(-Nk // 8) * 8
    cap = float(os.environ.get("FA_MI355_BWD_WIDE_MAX_GIB", "4")) * 2 ** 30
    per_group = 2 * G * N * ld * 2                        # both images of one key/value group
    gc = max(1, min(Hkv, int(cap // per_group)))           # key/value groups per chunk (one batch at a time)
    with _on_device(q.device) as stream, _trace_range("FA2_BWD"):
        images = torch.empty((2, gc * G, N, ld), dtype=q.dtype, device=q.device)
        nbytes = lib.fa_bwd_wide_workspace_bytes(1, gc * G, N)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=q.device)
        for b in range(B):
            for g0 in range(0, Hkv, gc):
                g1 = min(Hkv, g0 + gc)
                hq = slice(g0 * G, g1 * G)
                qs, os_, dos, ls = q[b:b + 1, hq], o[b:b + 1, hq], do[b:b + 1, hq], lse[b:b + 1, hq]
                ks, vs = k[b:b + 1, g0:g1], v[b:b + 1, g0:g1]
                nh = (g1 - g0) * G
                p_img, ds_img = images[0, :nh], images[1, :nh]
                rc = lib.fa_bwd_wide_ds(qs.data_ptr(), ks.data_ptr(), vs.data_ptr(), os_.data_ptr(), dos.data_ptr(),
                                        ls.contiguous().data_ptr(), p_img.data_ptr(), ds_img.data_ptr(), ld,
                                        1, nh, g1 - g0, N, Nk, D, _strides3(qs), _strides3(ks), _strides3(vs), _strides3(os_),
                                        _strides3(dos), code, 1 if causal else 0, scale, ws.data_ptr(), nbytes, stream)
                if rc != 0:
                    raise RuntimeError(f"fa_bwd_wide_ds failed ({rc}): {lib.fa_last_error().decode()}")
                # [group][G * N rows][keys]: the rows of a group's query heads stacked
                p2 = p_img.view(g1 - g0, G * N, ld)[:, :, :Nk]
                ds2 = ds_img.view(g1 - g0, G * N, ld)[:, :, :Nk]
                q2 = qs[0].reshape(g1 - g0, G * N, D)
                do2 = dos[0].reshape(g1 - g0, G * N, D)
                dv[b, g0:g1] = torch.bmm(p2.transpose(1, 2), do2)
                dk[b, g0:g1] = torch.bmm(ds2.transpose(1, 2), q2)
                dq[b, hq] = torch.bmm(ds2, ks[0]).view(nh, N, D)
    return dq, dk, dv


def _bwd_raw(lib, q, k, v, o, lse, do, causal: bool, scale: float):
    """Launch the backward: pre-pass, then either dK/dV kernel (writing dS) + dQ GEMM, or dQ kernel + dK/dV kernel
    (recompute), by the workspace `_bwd_plan` could get.  Returns (dq, dk, dv), each written once."""
    code = _dtype_code(q.dtype)
    B, H, N, D = q.shape
    Hkv, Nk = k.shape[1], k.shape[2]
    dq = torch.empty_like(o)                                                        # contiguous, like o
    if B * H * N == 0:          # no query rows: nothing flows into the keys and values (the reference zero-fills, FA2-triton.py:211-213)
        return dq, torch.zeros((B, Hkv, Nk, D), dtype=o.dtype, device=o.device), torch.zeros((B, Hkv, Nk, D), dtype=o.dtype, device=o.device)
    dk = torch.empty((B, Hkv, Nk, D), dtype=o.dtype, device=o.device)
    dv = torch.empty_like(dk)
    do = _kernel_ready(do.to(q.dtype))
    with _on_device(q.device) as stream, _trace_range("FA2_BWD"):
        bc, hc, ws, nbytes = _bwd_plan(lib, (B, H, Hkv, N, Nk, D), q.device)
        G = H // Hkv
        for b0 in range(0, B, bc):
            b1 = min(B, b0 + bc)
            for h0 in range(0, H, hc):
                h1 = min(H, h0 + hc)
                if bc >= B and hc >= H:
                    qs, os_, dos, ls, dqs, ks, vs, dks, dvs = q, o, do, lse, dq, k, v, dk, dv
                else:       # views: a whole-batch chunk is a contiguous block; a head chunk (single batch) goes through the strides
                    qs, os_, dos, ls, dqs = (t[b0:b1, h0:h1] for t in (q, o, do, lse, dq))
                    ks, vs, dks, dvs = (t[b0:b1, h0 // G:h1 // G] for t in (k, v, dk, dv))
                rc = lib.fa_bwd_ex(qs.data_ptr(), ks.data_ptr(), vs.data_ptr(), os_.data_ptr(), dos.data_ptr(), ls.data_ptr(),
                                   dqs.data_ptr(), dks.data_ptr(), dvs.data_ptr(), b1 - b0, h1 - h0, (h1 - h0) // G, N, Nk, D,
                                   _strides3(qs), _strides3(ks), _strides3(vs), _strides3(os_), _strides3(dos),
                                   _strides3(dqs), _strides3(dks), _strides3(dvs),
                                   code, 1 if causal else 0, scale, ws.data_ptr(), nbytes, stream)
                if rc != 0:
                    raise RuntimeError(f"fa_bwd failed ({rc}): {lib.fa_last_error().decode()}")
    return dq, dk, dv


class FlashAttnFn(torch.autograd.Function):
    """Differentiable forward, mirroring ``_FlashAttnFn`` (FA2-triton.py:173-237): forward saves q, k, v and
    the softmax statistics (here O and LSE = m + ln l instead of m, l), backward recomputes P."""

    @staticmethod
    def forward(ctx, q, k, v, causal: bool, scale: float):
        o, lse = _fwd_raw(load_library(), q, k, v, causal, scale, None, True)
        ctx.save_for_backward(q, k, v, o, lse)                # FA2-triton.py:203
        ctx.causal = causal                                   # :204
        ctx.scale = scale
        ctx.mark_non_differentiable(lse)
        return o, lse

    @staticmethod
    def backward(ctx, do, _dlse):
        q, k, v, o, lse = ctx.saved_tensors                   # :209
        run = _bwd_wide if q.shape[3] > REFERENCE_MAX_HEAD_DIM else _bwd_raw
        dq, dk, dv = run(load_library(), q, k, v, o, lse, do, ctx.causal, ctx.scale)
        return dq, dk, dv, None, None                         # :237


def flash_attn(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, causal: bool = False, *,
               softmax_scale: float | None = None, return_lse: bool = False,
               descale: tuple[float, float, float] | None = None, fp8_checked: bool = False):
    """Fused attention forward on MI355X.  ``q, k, v``: (B, H, N, D) GPU tensors; ``k, v`` may have fewer heads
    (B, H_kv, N_k, D) with H % H_kv == 0 (grouped-query / multi-query attention; dk, dv then have H_kv heads too) and a
    length N_k of their own; with ``causal`` the mask is then bottom-right aligned (key j visible to query i iff
    j <= i + N_k - N); for N_k < N the first N - N_k queries see no key (their output rows are 0, their LSE -inf).

    Returns O with q's dtype (fp32 inputs are computed in fp16 and cast back, like the
    reference).  ``return_lse=True`` additionally returns the (B, H, N) fp32 natural
    log-sum-exp of the scaled scores (the reference keeps m and l instead: lse = m + ln l).
    ``descale`` = (q, k, v) per-tensor dequantisation scales of float8_e4m3fn inputs (rejected for other dtypes).
    float8 inputs: every kernel variant notices softmax weight lost below e4m3's range and redoes those rows exactly
    (include/fa_mi355.h, fa_fwd_fp8) -- the default one through a sampled bound (one score in 32), the one that returns the
    LSE through its exact row sums.  ``fp8_checked=True`` runs the exact-sum variant even when no LSE is asked for (~4 % slower).
    Differentiable like the reference's entry point: when autograd is recording and an input requires
    grad, the HIP backward kernels produce dq, dk, dv (bf16 / fp16 / fp32-via-fp16 inputs).
    """
    check_args(q, k, v)
    orig_dtype = q.dtype
    needs_grad = torch.is_grad_enabled() and (q.requires_grad or k.requires_grad or v.requires_grad)
    if q.dtype == torch.float32:                         # FA2-triton.py:241-243
        q, k, v = q.half(), k.half(), v.half()
    code = _dtype_code(q.dtype)
    if needs_grad and code == FA_DTYPE_FP8_E4M3:
        raise FlashAttnArgumentError("float8 inputs are forward-only (no backward kernel)")
    if descale is not None:
        # per-tensor dequantisation scales belong to float8 inputs; accepting them for 16-bit inputs on one path only
        # (the autograd Function takes none) would make the result depend on the grad mode
        if code != FA_DTYPE_FP8_E4M3:
            raise FlashAttnArgumentError("descale=(q, k, v scales) is only meaningful with float8_e4m3fn inputs")
        if len(descale) != 3:
            raise FlashAttnArgumentError("descale must hold three per-tensor scales (q, k, v)")
    lib = load_library()
    B, H, N, D_in = q.shape
    if softmax_scale is None:
        softmax_scale = 1.0 / math.sqrt(D_in)             # FA2-triton.py:183 (of the caller's head_dim)
    # Every head_dim the reference accepts (D % 16 == 0, D <= 128, :178) runs natively: the library picks the head_dim-64
    # or head_dim-128 kernel and the hardware's buffer bounds check supplies zeros for the columns past D (no padded
    # copies on the host, nothing extra stored).  144 .. 256 run on the wide-head kernels (backward: _bwd_wide).
    D = D_in
    if not lib.fa_supported(code, D):
        raise FlashAttnArgumentError(f"no gfx950 kernel for dtype={q.dtype}, head_dim={D}")
    q, k, v = _kernel_ready(q), _kernel_ready(k), _kernel_ready(v)
    if needs_grad:
        o, lse = FlashAttnFn.apply(q, k, v, bool(causal), float(softmax_scale))
    else:
        want_lse = return_lse or (fp8_checked and code == FA_DTYPE_FP8_E4M3)
        o, lse = _fwd_raw(lib, q, k, v, bool(causal), float(softmax_scale), descale, want_lse)
    if orig_dtype == torch.float32:
        o = o.to(orig_dtype)                              # FA2-triton.py:244
    return (o, lse) if return_lse else o


# the reference's spelling (FA2-triton.py:240)
flash_attention = flash_attn
