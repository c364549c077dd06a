"""ctypes binding of the Qwen2.5-VL row of libvqs_hip.so (include/vqs_qwen.h).  PyTorch is used for device memory and
the current HIP stream only.  No CPU fallback: a missing library or a failed launch raises."""
from __future__ import annotations

import ctypes
from typing import Dict, Sequence, Tuple

import torch

from ..engine import VqsError, VqsWeightDesc, load_library, _stream_ptr
from .config import Qwen25VLConfig
from .layout import text_layout, vision_layout

_i32, _f32, _vp, _sz = ctypes.c_int32, ctypes.c_float, ctypes.c_void_p, ctypes.c_size_t


class VqsQwenConfig(ctypes.Structure):      # field order = struct vqs_qwen_config
    _fields_ = [(n, _i32) for n in ("v_depth", "v_hidden", "v_heads", "v_mlp", "v_patch_dim", "v_merge_unit", "v_out_hidden",
                                    "v_fullatt_mask")] + [("v_eps", _f32)] + \
               [(n, _i32) for n in ("t_vocab", "t_hidden", "t_layers", "t_heads", "t_kv_heads", "t_mlp")] + [("t_eps", _f32)]


_SIGS = {
    "vqs_qwen_create": (_i32, [ctypes.POINTER(VqsQwenConfig), ctypes.POINTER(_vp)]),
    "vqs_qwen_destroy": (None, [_vp]),
    "vqs_qwen_last_error": (ctypes.c_char_p, [_vp]),
    "vqs_qwen_packed_bytes": (_sz, [_vp]),
    "vqs_qwen_bind_weights": (_i32, [_vp, ctypes.POINTER(VqsWeightDesc), _i32, _vp, _sz, _vp]),
    "vqs_qwen_vision_workspace_bytes": (_sz, [_vp, _i32, _i32]),
    "vqs_qwen_encode_vision": (_i32, [_vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _sz, _vp]),
    "vqs_qwen_score_workspace_bytes": (_sz, [_vp, _i32, _i32]),
    "vqs_qwen_profile_enable": (_i32, [_vp, _i32]),
    "vqs_qwen_profile_read": (_i32, [_vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double), _i32]),
    "vqs_qwen_kv_bytes": (_sz, [_vp, _i32, _i32]),
    "vqs_qwen_prefill": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _sz, _vp, _sz, _i32, _vp]),
    "vqs_qwen_decode_workspace_bytes": (_sz, [_vp, _i32]),
    "vqs_qwen_decode": (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _sz, _vp, _vp, _sz, _vp]),
    "vqs_qwen_debug_tap": (_i32, [_vp, ctypes.c_char_p, _vp, _sz]),
    "vqs_qwen_debug_option": (_i32, [_vp, ctypes.c_char_p, ctypes.c_int64]),
    "vqs_qwen_get_option": (_i32, [_vp, ctypes.c_char_p, ctypes.POINTER(ctypes.c_int64)]),
    "vqs_qwen_range_report": (_i32, [_vp, ctypes.POINTER(_f32), ctypes.POINTER(_f32), _i32]),
    "vqs_qwen_score": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _sz, _vp]),
}


class QwenEngine:
    def __init__(self, cfg: Qwen25VLConfig, weights: Dict[str, torch.Tensor], device="cuda:0", x_pitch: int = None, fp16: bool = None):
        """fp16: None = the library's default (the range-safe fp16 forms wherever the model is eligible: include/vqs_qwen.h), False = bind
        without fp16 weight copies (the reference's bf16 everywhere), True = insist (raises on an ineligible configuration)."""
        if not torch.cuda.is_available():
            raise VqsError("the Qwen2.5-VL HIP path needs an MI355X (no CPU fallback)")
        self.lib = load_library()
        for name, (res, args) in _SIGS.items():
            fn = getattr(self.lib, name)
            fn.restype, fn.argtypes = res, args
        self.cfg, self.device = cfg, torch.device(device)
        v, t = cfg.vision, cfg.text
        mask = 0
        for i in v.fullatt_blocks:
            mask |= 1 << i
        c = VqsQwenConfig(v.depth, v.hidden, v.heads, v.mlp, v.patch_dim, v.merge_unit, v.out_hidden, mask, v.rms_eps,
                          t.vocab, t.hidden, t.layers, t.heads, t.kv_heads, t.mlp, t.rms_eps)
        h = _vp()
        rc = self.lib.vqs_qwen_create(ctypes.byref(c), ctypes.byref(h))
        if rc != 0:
            raise VqsError(f"vqs_qwen_create failed ({rc}): unsupported configuration")
        self._h = h
        if x_pitch is not None:          # test hook (include/vqs_qwen.h): row pitch of the normalised activations / gate|up weight rows
            self._check(self.lib.vqs_qwen_debug_option(self._h, b"x_pitch", int(x_pitch)), "vqs_qwen_debug_option")
        if fp16 is not None:
            self._check(self.lib.vqs_qwen_debug_option(self._h, b"fp16", 1 if fp16 else 0), "vqs_qwen_debug_option")
        with torch.cuda.device(self.device):
            self._weights = {k: w.to(self.device, torch.bfloat16).contiguous() for k, w in weights.items()}
            descs = (VqsWeightDesc * len(self._weights))(*[VqsWeightDesc(k.encode(), w.data_ptr(), w.numel())
                                                            for k, w in self._weights.items()])
            self._packed = torch.empty(self.lib.vqs_qwen_packed_bytes(self._h), dtype=torch.uint8, device=self.device)
            self._check(self.lib.vqs_qwen_bind_weights(self._h, descs, len(self._weights), self._packed.data_ptr(),
                                                       self._packed.numel(), _stream_ptr()), "vqs_qwen_bind_weights")
            torch.cuda.synchronize()
        self._ws = None

    def set_option(self, name: str, value: int):
        """Execution-form switches of include/vqs_qwen.h (vqs_qwen_debug_option): "tail_precise" 1 (default) = the logits come from the
        precise re-evaluation of every sample's last prompt position, 0 = from the bf16 prefill's last row (rounds 2-4); "fp16" 0 / 1 = the
        bf16 forms / the range-safe fp16 forms (1 only if the weights were bound with it)."""
        self._check(self.lib.vqs_qwen_debug_option(self._h, name.encode(), int(value)), "vqs_qwen_debug_option")

    def get_option(self, name: str) -> int:
        v = ctypes.c_int64(0)
        self._check(self.lib.vqs_qwen_get_option(self._h, name.encode(), ctypes.byref(v)), "vqs_qwen_get_option")
        return int(v.value)

    @property
    def fp16_active(self) -> bool:
        return self.get_option("fp16") == 1

    def range_report(self):
        """-> (bounds, sigmas) float32 tensors of the bind-time range proof (vqs_qwen_range_report; empty when none was made)."""
        n = self.lib.vqs_qwen_range_report(self._h, None, None, 0)
        if n < 0:
            self._check(n, "vqs_qwen_range_report")
        b, s = (ctypes.c_float * max(n, 1))(), (ctypes.c_float * max(n, 1))()
        self.lib.vqs_qwen_range_report(self._h, b, s, n)
        return torch.tensor(list(b)[:n]), torch.tensor(list(s)[:n])

    def merged_values(self, merged: torch.Tensor) -> torch.Tensor:
        """fp32 values of an encode_vision result (undoes the fp16 forms' scale)."""
        if merged.dtype == torch.float16:
            _, sig = self.range_report()
            return merged.float() / float(sig[6 * self.cfg.vision.depth + 2])
        return merged.float()

    def _merged_matches_mode(self, merged: torch.Tensor):
        want = torch.float16 if self.fp16_active else torch.bfloat16
        if merged.dtype != want:
            raise VqsError(f"merged vision tokens are {merged.dtype}, the engine's operand format is {want}: option fp16 changed between encode_vision and score?")

    def _check(self, rc, what):
        if rc != 0:
            raise VqsError(f"{what} failed ({rc}): {self.lib.vqs_qwen_last_error(self._h).decode()}")

    def _workspace(self, need):
        if need == 0:
            raise VqsError("unsupported shape")
        if self._ws is None or self._ws.numel() < need:
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws

    def encode_vision(self, patches: torch.Tensor, grids: Sequence[Tuple[int, int, int]]) -> torch.Tensor:
        """patches [N, patch_dim] (HF processor order) -> merged vision tokens [N/4, out_hidden]: bf16, or with the fp16 forms a float16 tensor
        holding value * sigma (opaque: feed it to score_logits / prefill of THIS engine; merged_values() gives the numbers)."""
        lay = vision_layout(self.cfg, grids)
        with torch.cuda.device(self.device):
            dev = self.device
            px = patches.to(dev, torch.bfloat16).contiguous()
            N, Np = lay["N"], lay["Np"]
            if px.shape[0] != N:
                raise VqsError(f"{px.shape[0]} patch rows for grids that hold {N}")
            d = {k: lay[k].to(dev) for k in ("row_map", "inv_row", "win_valid", "cell_inv", "cos_w", "sin_w", "cos_f", "sin_f")}
            # the handle's operand format (include/vqs_qwen.h): bf16, or -- fp16 forms -- IEEE fp16 behind the merger's power-of-two scale
            f16 = self.fp16_active
            out = torch.empty(N // self.cfg.vision.merge_unit, self.cfg.vision.out_hidden, dtype=torch.float16 if f16 else torch.bfloat16, device=dev)
            ws = self._workspace(self.lib.vqs_qwen_vision_workspace_bytes(self._h, N, Np))
            self._check(self.lib.vqs_qwen_encode_vision(self._h, px.data_ptr(), N, d["row_map"].data_ptr(), d["inv_row"].data_ptr(),
                                                        d["win_valid"].data_ptr(), d["cell_inv"].data_ptr(), d["cos_w"].data_ptr(),
                                                        d["sin_w"].data_ptr(), d["cos_f"].data_ptr(), d["sin_f"].data_ptr(), Np,
                                                        lay["win_len"], lay["frame_len"], out.data_ptr(), ws.data_ptr(), ws.numel(),
                                                        _stream_ptr()), "vqs_qwen_encode_vision")
            return out

    def score_logits(self, merged: torch.Tensor, input_ids: torch.Tensor, attention_mask: torch.Tensor,
                     grids: Sequence[Tuple[int, int, int]]) -> torch.Tensor:
        """-> fp32 [B, vocab]: logits of the last valid position of every sample (= scores[0] of HF generate)."""
        lay = text_layout(self.cfg, input_ids.cpu(), attention_mask.cpu(), grids)
        with torch.cuda.device(self.device):
            dev = self.device
            B, L = input_ids.shape
            ids = input_ids.to(dev, torch.int32).contiguous()
            t = {k: lay[k].to(dev) for k in ("vis_slot", "seq_len", "last_row", "cos", "sin")}
            logits = torch.empty(B, self.cfg.text.vocab, dtype=torch.float32, device=dev)
            self._merged_matches_mode(merged)
            ws = self._workspace(self.lib.vqs_qwen_score_workspace_bytes(self._h, B, L))
            self._check(self.lib.vqs_qwen_score(self._h, merged.contiguous().data_ptr(), ids.data_ptr(), t["vis_slot"].data_ptr(),
                                                t["seq_len"].data_ptr(), t["last_row"].data_ptr(), t["cos"].data_ptr(),
                                                t["sin"].data_ptr(), B, L, logits.data_ptr(), ws.data_ptr(), ws.numel(),
                                                _stream_ptr()), "vqs_qwen_score")
            return logits

    def prefill(self, merged: torch.Tensor, input_ids: torch.Tensor, attention_mask: torch.Tensor,
                grids: Sequence[Tuple[int, int, int]], max_new_tokens: int):
        """score_logits that also keeps the KV cache for `max_new_tokens - 1` further positions.
        -> (fp32 [B, vocab] logits of the last prompt position, state for decode())."""
        lay = text_layout(self.cfg, input_ids.cpu(), attention_mask.cpu(), grids)
        with torch.cuda.device(self.device):
            dev = self.device
            B, L = input_ids.shape
            Lmax = L + max(int(max_new_tokens) - 1, 0)
            ids = input_ids.to(dev, torch.int32).contiguous()
            t = {k: lay[k].to(dev) for k in ("vis_slot", "seq_len", "last_row", "cos", "sin")}
            logits = torch.empty(B, self.cfg.text.vocab, dtype=torch.float32, device=dev)
            self._merged_matches_mode(merged)
            kv = torch.empty(self.lib.vqs_qwen_kv_bytes(self._h, B, Lmax), dtype=torch.uint8, device=dev)
            ws = self._workspace(self.lib.vqs_qwen_score_workspace_bytes(self._h, B, L))
            self._check(self.lib.vqs_qwen_prefill(self._h, merged.contiguous().data_ptr(), ids.data_ptr(), t["vis_slot"].data_ptr(),
                                                  t["seq_len"].data_ptr(), t["last_row"].data_ptr(), t["cos"].data_ptr(),
                                                  t["sin"].data_ptr(), B, L, logits.data_ptr(), ws.data_ptr(), ws.numel(), kv.data_ptr(),
                                                  kv.numel(), Lmax, _stream_ptr()), "vqs_qwen_prefill")
            # len_host_max: the longest sample's length, tracked on the HOST -- decode() checks the cache bound without reading the device
            # tensor (round 3 did `int(state["len"].max())` per step: a device sync that serialised the host's ~400 launches per step
            # with the GPU's work; the decode kernels themselves ignore / clamp a length outside the cache, qwen_decode.hip)
            # pos: the next token's M-RoPE position per axis, kept ON THE DEVICE -- decode() derives its cos / sin tables there (a few
            # tiny launches) instead of on the host: the host tables cost two blocking H2D copies per step, each a stream synchronise
            # (PyTorch syncs after a pageable copy), i.e. the host's launches never overlapped the GPU's work, and torch's CPU thread
            # pool stalled a step by 70 ms now and then (profiles/r4_call15_*)
            state = {"kv": kv, "len": t["seq_len"].clone(), "Lmax": Lmax, "B": B, "pos": lay["next_pos"].clone().to(dev), "steps": 0,
                     "len_host_max": int(lay["seq_len"].max())}
            return logits, state

    def decode(self, state, token_ids: torch.Tensor) -> torch.Tensor:
        """One further position per sample: token_ids long [B] (the tokens just generated; anything for stopped samples)
        -> fp32 [B, vocab] logits of that position.  Advances `state`."""
        from .layout import decode_tables
        B = state["B"]
        if state["len_host_max"] >= state["Lmax"]:
            raise VqsError("KV cache is full: prefill(..., max_new_tokens) sized it")
        cos, sin = decode_tables(self.cfg, state["pos"])
        with torch.cuda.device(self.device):
            dev = self.device
            ids = token_ids.to(dev, torch.int32).contiguous()
            cos, sin = cos.to(dev), sin.to(dev)          # no-ops: the tables were computed on the device
            logits = torch.empty(B, self.cfg.text.vocab, dtype=torch.float32, device=dev)
            need = self.lib.vqs_qwen_decode_workspace_bytes(self._h, B)
            if state.get("ws") is None or state["ws"].numel() < need:
                state["ws"] = torch.empty(need, dtype=torch.uint8, device=dev)
            self._check(self.lib.vqs_qwen_decode(self._h, ids.data_ptr(), state["len"].data_ptr(), cos.data_ptr(), sin.data_ptr(), B,
                                                 state["Lmax"], state["kv"].data_ptr(), state["kv"].numel(), logits.data_ptr(),
                                                 state["ws"].data_ptr(), state["ws"].numel(), _stream_ptr()), "vqs_qwen_decode")
            state["len"] += 1
            state["len_host_max"] += 1
            state["pos"] = state["pos"] + 1
            state["steps"] += 1
            return logits

    def tap(self, name, dst: torch.Tensor = None):
        """Register `dst` (device tensor, kept alive by the caller) for the named intermediate of the next passes
        (vqs_qwen_debug_tap); name None clears all taps."""
        if name is None:
            self._check(self.lib.vqs_qwen_debug_tap(self._h, None, None, 0), "vqs_qwen_debug_tap")
            return
        self._check(self.lib.vqs_qwen_debug_tap(self._h, name.encode(), dst.data_ptr(), dst.numel() * dst.element_size()), "vqs_qwen_debug_tap")

    def profile(self, on: bool):
        self._check(self.lib.vqs_qwen_profile_enable(self._h, 1 if on else 0), "vqs_qwen_profile_enable")

    def profile_read(self, reset: bool = True):
        """-> (launches, ms, flops, bytes) of the GEMM launches since the last reset (synchronises)."""
        ms, fl, by = ctypes.c_double(0), ctypes.c_double(0), ctypes.c_double(0)
        n = self.lib.vqs_qwen_profile_read(self._h, ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(by), 1 if reset else 0)
        if n < 0:
            self._check(n, "vqs_qwen_profile_read")
        return n, ms.value, fl.value, by.value

    def close(self):
        if getattr(self, "_h", None):
            self.lib.vqs_qwen_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
