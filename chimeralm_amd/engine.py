"""`Engine`: thin Python owner of one `clm_handle` (one per GPU).  Torch is used only for device memory
and the current HIP stream; all arithmetic happens behind the C ABI (include/chimeralm_hip.h)."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _native as N

_TORCH_DT = {torch.float32: N.DT_F32, torch.float64: N.DT_F64, torch.bfloat16: N.DT_BF16, torch.float16: N.DT_F16}
_IDS_DT = {torch.int64: N.DT_I64, torch.int32: N.DT_I32, torch.uint8: N.DT_U8}


class EngineError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"chimeralm_hip error {code}: {msg}")
        self.code = code


class Engine:
    """One MI355X inference engine instance bound to `device` (e.g. "cuda:0")."""

    def __init__(self, device: torch.device | str | int = "cuda:0", precision: str = "fp32", chunk_reads: int = 256):
        self._lib = N.load()
        self._h = N._H()
        device = torch.device(device if not isinstance(device, int) else f"cuda:{device}")
        if device.type != "cuda":
            raise EngineError(N.E_UNSUPPORTED, "the engine runs on an MI355X (torch device type 'cuda' on ROCm) only; "
                                               "there is no CPU path")
        if precision not in N.PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(N.PRECISIONS)}")
        self.device = torch.device("cuda", device.index if device.index is not None else torch.cuda.current_device())
        self.precision = precision
        cfg = N.ClmConfig()
        self._lib.clm_default_config(C.byref(cfg))
        cfg.precision = N.PRECISIONS[precision]
        cfg.chunk_reads = int(chunk_reads)
        self.cfg = cfg
        rc = self._lib.clm_create(C.byref(cfg), self.device.index, C.byref(self._h))
        if rc:
            raise EngineError(rc, (self._lib.clm_last_error(None) or b"").decode())
        self.finalized = False

    # ------------------------------------------------------------------ helpers
    def _check(self, rc: int):
        if rc:
            raise EngineError(rc, (self._lib.clm_last_error(self._h) or b"").decode())

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.clm_destroy(self._h)
            self._h = N._H()

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass

    # ------------------------------------------------------------------ weights
    def load_weight(self, key: str, tensor: torch.Tensor):
        t = tensor.detach()
        if t.dtype not in _TORCH_DT:
            t = t.float()
        t = t.contiguous()
        shape = (C.c_int64 * t.dim())(*t.shape)
        self._check(self._lib.clm_load_weight(self._h, key.encode(), C.c_void_p(t.data_ptr()), _TORCH_DT[t.dtype],
                                              shape, t.dim()))
        self.finalized = False

    def load_state_dict(self, state_dict: dict[str, torch.Tensor]):
        """Load every `net.backbone.*` / `net.head.*` entry of a reference checkpoint and finalize."""
        for k, v in state_dict.items():
            kk = k[4:] if k.startswith("net.") else k
            if kk.startswith("backbone.") or kk.startswith("head."):
                self.load_weight(k, v)
        self.finalize()

    def finalize(self):
        self._check(self._lib.clm_finalize(self._h))
        self.finalized = True

    def reserve(self, batch: int, length: int):
        self._check(self._lib.clm_reserve(self._h, int(batch), int(length)))

    # ------------------------------------------------------------------ forward
    def forward(self, input_ids: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
        """input_ids [B, L] (int64 / int32 / uint8) on this engine's device -> logits fp32 [B, 2].
        Asynchronous on torch's current stream."""
        if input_ids.dim() != 2:
            raise ValueError("input_ids must be [batch, length]")
        if input_ids.device != self.device:
            raise EngineError(N.E_INVALID, f"input_ids is on {input_ids.device}, engine on {self.device}; "
                                           "no CPU execution path exists")
        if input_ids.dtype not in _IDS_DT:
            raise ValueError("input_ids dtype must be int64, int32 or uint8")
        if input_ids.stride(1) != 1:
            input_ids = input_ids.contiguous()
        B, L = input_ids.shape
        if out is None:
            out = torch.empty((B, self.cfg.n_classes), dtype=torch.float32, device=self.device)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        self._check(self._lib.clm_forward(self._h, C.c_void_p(input_ids.data_ptr()), _IDS_DT[input_ids.dtype],
                                          input_ids.stride(0), B, L, C.c_void_p(out.data_ptr()), C.c_void_p(stream)))
        return out

    __call__ = forward

    # ------------------------------------------------------------------ host batches (pinned slots of the BAM feeder)
    def stage_host_ids(self, host_ptr: int, ids_dtype: int, row_stride: int, batch: int, length: int) -> int:
        """Enqueue the H2D copy of a host batch on the engine's own copy stream (overlaps the running forward);
        returns the staging-buffer index to pass to `forward_staged` / `stage_wait`."""
        k = C.c_int(-1)
        self._check(self._lib.clm_stage_ids(self._h, C.c_void_p(host_ptr), int(ids_dtype), int(row_stride), int(batch),
                                            int(length), C.byref(k)))
        return k.value

    def forward_staged(self, staged: int, batch: int, out: torch.Tensor | None = None) -> torch.Tensor:
        if out is None:
            out = torch.empty((batch, self.cfg.n_classes), dtype=torch.float32, device=self.device)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        self._check(self._lib.clm_forward_staged(self._h, int(staged), C.c_void_p(out.data_ptr()), C.c_void_p(stream)))
        return out

    def check(self):
        """Wait for the current stream and raise for errors only the device can see (token ids outside the embedding table:
        the reference raises IndexError there)."""
        stream = torch.cuda.current_stream(self.device).cuda_stream
        self._check(self._lib.clm_check(self._h, C.c_void_p(stream)))

    # ------------------------------------------------------------------ 16-bit mode vs the reference's arithmetic
    def selfcheck(self, input_ids: torch.Tensor) -> tuple[float, int]:
        """Run `input_ids` through this engine's mode AND through the exact-fp32 kernels of the same handle; returns
        (max |logit difference|, number of reads whose label differs).  Synchronises the current stream (`clm_selfcheck`)."""
        if input_ids.dim() != 2 or input_ids.device != self.device or input_ids.dtype not in _IDS_DT:
            raise ValueError("selfcheck wants input_ids [batch, length] (int64 / int32 / uint8) on the engine's device")
        if input_ids.stride(1) != 1:
            input_ids = input_ids.contiguous()
        B, L = input_ids.shape
        diff, differ = C.c_float(0.0), C.c_int(0)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        self._check(self._lib.clm_selfcheck(self._h, C.c_void_p(input_ids.data_ptr()), _IDS_DT[input_ids.dtype],
                                            input_ids.stride(0), B, L, C.c_void_p(stream), C.byref(diff), C.byref(differ)))
        return float(diff.value), int(differ.value)

    def set_fallback(self, level: int | bool = 1):
        """`clm_set_fallback`: 0 = the engine's own mode; 1 = every later forward runs in the next arithmetic inside the 1e-3 gate
        (a 16-bit engine: fp16x3, fp32-class logits at about twice the exact rate; an fp16x3 engine: exact fp32); 2 = exact fp32."""
        self._check(self._lib.clm_set_fallback(self._h, int(level)))

    def set_f16c_min_len(self, min_len: int):
        """fp16c: reads shorter than `min_len` tokens run in the fp16x3 kernels (`clm_set_short_read_len`)."""
        self._check(self._lib.clm_set_short_read_len(self._h, int(min_len)))

    def set_mlp_compensation(self, on: bool = True):
        """fp16c: fc1 / fc2 on hi + lo weights too (`clm_set_mlp_compensation`; ~10 % slower, for weights whose MLP rounding shows)."""
        self._check(self._lib.clm_set_mlp_compensation(self._h, int(on)))

    def effective_precision(self, length: int) -> str:
        code = self._lib.clm_effective_precision(self._h, int(length))
        if code < 0:
            raise EngineError(code, "clm_effective_precision")
        return {N.PREC_F32: "fp32", N.PREC_BF16: "bf16", N.PREC_F16: "fp16", N.PREC_F16C: "fp16c", N.PREC_F16X3: "fp16x3"}[code]

    def stage_wait(self, staged: int):
        self._check(self._lib.clm_stage_wait(self._h, int(staged)))

    # ------------------------------------------------------------------ taps
    def debug_stop_after(self, layer: int = -1, stage: int = -1):
        self._check(self._lib.clm_debug_stop_after(self._h, layer, stage))

    def debug_fetch(self, name: str, shape, dtype=np.float32) -> np.ndarray:
        arr = np.empty(shape, dtype=dtype)
        self._check(self._lib.clm_debug_fetch(self._h, name.encode(), arr.ctypes.data_as(C.c_void_p), arr.nbytes))
        return arr

    def profile_enable(self, on: bool = True):
        self._check(self._lib.clm_profile_enable(self._h, int(on)))

    def profile_read(self, reset: bool = True) -> dict[str, tuple[float, int]]:
        ms = (C.c_double * N.N_STAGES)()
        n = (C.c_int64 * N.N_STAGES)()
        self._check(self._lib.clm_profile_read(self._h, ms, n, int(reset)))
        return {N.STAGES[i]: (ms[i], n[i]) for i in range(N.N_STAGES)}
