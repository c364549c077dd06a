"""ctypes binding of ``libvqs_hip.so`` (C ABI in ``include/vqs.h``).

PyTorch is used here for device memory and the current HIP stream only; every FLOP of the scoring
pass runs in the hand-written gfx950 kernels behind the C ABI.  There is deliberately NO fallback:
if the shared library is missing or a call fails, this module raises.
"""
from __future__ import annotations

import ctypes
import os
from typing import Dict, List, Optional, Tuple

import torch

from .config import ClipT5Config

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvqs_hip.so")

_c_i32, _c_i64, _c_f32, _c_vp, _c_sz = ctypes.c_int32, ctypes.c_int64, ctypes.c_float, ctypes.c_void_p, ctypes.c_size_t


class VqsConfig(ctypes.Structure):
    _fields_ = [
        ("vis_hidden", _c_i32), ("vis_layers_run", _c_i32), ("vis_heads", _c_i32), ("vis_mlp", _c_i32),
        ("vis_patch", _c_i32), ("vis_image", _c_i32), ("vis_ln_eps", _c_f32),
        ("d_model", _c_i32), ("n_heads", _c_i32), ("d_kv", _c_i32), ("d_ff", _c_i32), ("enc_layers", _c_i32),
        ("dec_layers", _c_i32), ("vocab", _c_i32), ("rel_buckets", _c_i32), ("rel_max_distance", _c_i32),
        ("t5_ln_eps", _c_f32),
    ]


class VqsWeightDesc(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char_p), ("d_data", _c_vp), ("numel", _c_i64)]


class VqsError(RuntimeError):
    pass


# every symbol declared in include/vqs.h: (restype, argtypes)
_SIGNATURES = {
    "vqs_create": (_c_i32, [ctypes.POINTER(VqsConfig), ctypes.POINTER(_c_vp)]),
    "vqs_destroy": (None, [_c_vp]),
    "vqs_last_error": (ctypes.c_char_p, [_c_vp]),
    "vqs_packed_bytes": (_c_sz, [_c_vp]),
    "vqs_bind_weights": (_c_i32, [_c_vp, ctypes.POINTER(VqsWeightDesc), _c_i32, _c_vp, _c_sz, _c_vp]),
    "vqs_encode_workspace_bytes": (_c_sz, [_c_vp, _c_i32]),
    "vqs_encode_images": (_c_i32, [_c_vp, _c_vp, _c_i32, _c_vp, _c_vp, _c_sz, _c_vp]),
    "vqs_score_workspace_bytes": (_c_sz, [_c_vp, _c_i32, _c_i32, _c_i32]),
    "vqs_score": (_c_i32, [_c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_i32, _c_i32, _c_i32, _c_vp, _c_vp, _c_vp, _c_sz, _c_vp]),
    "vqs_workspace_offset": (_c_i64, [_c_vp, ctypes.c_char_p, _c_i32, _c_i32, _c_i32, ctypes.POINTER(_c_i64)]),
    "vqs_generate_workspace_bytes": (_c_sz, [_c_vp, _c_i32, _c_i32, _c_i32]),
    "vqs_generate": (_c_i32, [_c_vp, _c_vp, _c_vp, _c_vp, _c_i32, _c_i32, _c_i32, _c_vp, _c_vp, _c_sz, _c_vp]),
    "vqs_profile_enable": (_c_i32, [_c_vp, _c_i32]),
    "vqs_profile_read": (_c_i32, [_c_vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double), _c_i32]),
    "vqs_profile_bytes": (_c_i32, [_c_vp, ctypes.POINTER(ctypes.c_double)]),
    "vqs_profile_report": (ctypes.c_char_p, [_c_vp]),
    "vqs_gemm": (_c_i32, [_c_vp, _c_vp, _c_vp, _c_vp, _c_vp] + [_c_i32] * 10 + [_c_vp]),
    "vqs_normalize_u8": (_c_i32, [_c_vp, _c_vp, _c_i32, _c_i32, _c_i32, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float), _c_vp]),
    "vqs_rope": (_c_i32, [_c_vp, _c_vp, _c_vp, _c_i32, _c_i32, _c_i32, _c_i32, _c_i32, _c_vp]),
    "vqs_attention_hd": (_c_i32, [_c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_i32, _c_i32, _c_i32, _c_i32, _c_i32, _c_f32, _c_i32, _c_vp]),
    "vqs_gemm_rms": (_c_i32, [_c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_i32, _c_f32, _c_f32, _c_i32, _c_i32, _c_i32,
                              _c_i32, _c_i32, _c_i32, _c_i32, _c_i32, _c_i32, _c_i32, _c_vp]),
    "vqs_attention": (_c_i32, [_c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_i32, _c_i32, _c_i32, _c_f32, _c_vp]),
    "vqs_decoder_attention": (_c_i32, [_c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp] + [_c_i32] * 7 + [_c_vp]),
    "vqs_rmsnorm": (_c_i32, [_c_vp, _c_vp, _c_vp, _c_vp, _c_i32, _c_i32, _c_f32, _c_vp]),
    "vqs_layernorm": (_c_i32, [_c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_i32, _c_i32, _c_i32, _c_f32, _c_vp]),
    "vqs_norm_deferred": (_c_i32, [_c_i32, _c_vp, _c_vp, _c_vp, _c_i32, _c_vp, _c_vp, _c_vp, _c_i32, _c_i32, _c_f32, _c_vp]),
    "vqs_score_head": (_c_i32, [_c_vp, _c_i32, _c_i32, _c_vp, _c_vp, _c_vp, _c_i32, _c_i32, _c_vp]),
    "vqs_attention_lds_bytes": (ctypes.c_int64, [_c_i32, _c_i32, _c_i32]),
    "vqs_set_option": (_c_i32, [_c_vp, ctypes.c_char_p, _c_i32]),
    "vqs_get_option": (_c_i32, [_c_vp, ctypes.c_char_p, ctypes.POINTER(_c_i32)]),
    "vqs_debug_tap": (_c_i32, [_c_vp, ctypes.c_char_p, _c_vp, ctypes.c_size_t]),
    "vqs_debug_tap_window": (_c_i32, [_c_vp, _c_i32, _c_i32]),
    "vqs_debug_attention_f16": (_c_i32, [_c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_i32, _c_i32, _c_i32, ctypes.c_float, _c_vp]),
    "vqs_debug_norm16": (_c_i32, [_c_i32, _c_vp, _c_vp, _c_vp, _c_i32, _c_vp, _c_vp, _c_vp, _c_i32, _c_i32, ctypes.c_float, _c_i32, _c_vp]),
    "vqs_debug_gemm_form": (_c_i32, [_c_i32] * 11),
    "vqs_debug_gemm_batched": (_c_i32, [_c_vp, _c_vp, _c_vp] + [_c_i32] * 8 + [_c_i64] * 4 + [_c_i32, _c_i32, _c_vp]),
    "vqs_debug_heads_rows": (_c_i32, [_c_i32, _c_i32, _c_i32, _c_i32, _c_i32, _c_vp]),
    "vqs_debug_tile_order": (_c_i32, [_c_i32, _c_i32, _c_i32, _c_i32, _c_i32, _c_i32, _c_i32, _c_vp]),
    "vqs_relpos_bucket": (_c_i32, [_c_i32, _c_i32, _c_i32, _c_i32]),
}

_lib = None


def load_library(path: Optional[str] = None) -> ctypes.CDLL:
    """Load libvqs_hip.so and type every exported symbol.  Raises VqsError if it is absent."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("VQS_LIB_PATH", LIB_PATH)    # VQS_LIB_PATH: lab builds (tools/gemm_lab.sh) only
    if not os.path.exists(p):
        raise VqsError(
            f"{p} not found: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()' "
            f"or make -C t2v_metrics_amd/csrc).  There is no CPU fallback.")
    lib = ctypes.CDLL(p)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _lib = lib
    return lib


def exported_symbols():
    return sorted(_SIGNATURES)


def _stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def make_vqs_config(cfg: ClipT5Config) -> VqsConfig:
    v, t = cfg.vision, cfg.t5
    return VqsConfig(v.hidden, v.layers_run, v.heads, v.mlp, v.patch, v.image, v.ln_eps, t.d_model, t.heads, t.d_kv,
                     t.d_ff, t.layers, t.dec_layers, t.vocab, t.rel_buckets, t.rel_max_distance, t.ln_eps)


FP16_MAX_FINITE_ROUNDED = 65520.0      # values at or above this round to +-inf in IEEE fp16


_ENC_FP16_LINEARS = ("layer.0.SelfAttention.q.weight", "layer.0.SelfAttention.k.weight", "layer.0.SelfAttention.v.weight",
                     "layer.0.SelfAttention.o.weight", "layer.1.DenseReluDense.wi_0.weight", "layer.1.DenseReluDense.wi_1.weight")


def fp16_unsafe_weights(weights: Dict[str, torch.Tensor], stack: str = "vit") -> List[str]:
    """Names of the linear weights that an fp16 copy cannot hold (|w| >= 65 520, or not finite) among the tensors vqs_bind_weights
    converts: stack "vit" = the vision tower's and the projector's (option vit_fp16), "enc" = the T5 encoder's q / k / v / o / wi_0 / wi_1
    (option enc_fp16; wo keeps bf16 operands).  Any device; one reduction per tensor."""
    bad = []
    for k, w in weights.items():
        if w.dim() != 2 or not k.endswith(".weight"):
            continue
        if stack == "vit":
            mine = k.startswith("vision.encoder.layers.") or k.startswith("mm_projector.")
        elif stack == "dec":          # option dec_fp16 reads an fp16 copy of the cross-attention's Wk^T
            mine = k.startswith("decoder.block.") and k.endswith("layer.1.EncDecAttention.k.weight")
        else:
            mine = k.startswith("encoder.block.") and k.endswith(_ENC_FP16_LINEARS)
        if mine:
            m = float(w.detach().abs().max().float()) if w.numel() else 0.0
            if not (m < FP16_MAX_FINITE_ROUNDED):          # also catches NaN / inf
                bad.append(k)
    return bad


FP16_HEAD = 32768.0        # what a proven bound must stay under for an fp16 site to run: half of the fp16 maximum (room for the fp32 accumulation order)
FP16_OPTIONS = ("vit_fp16", "proj_fp16", "enc_fp16", "dec_fp16")


@torch.no_grad()
def fp16_range_proof(cfg: ClipT5Config, weights: Dict[str, torch.Tensor]) -> Dict[str, dict]:
    """Bind-time range proof of the fp16 execution options (include/vqs.h "Range safety"): for every tensor an option holds
    in IEEE fp16 a bound of its magnitude that follows from the WEIGHTS ALONE, so that "the option is on" means "no input can overflow it".
      * LayerNorm / RMSNorm output: |x^_k| <= ||x^||_2 <= sqrt(D), so |out_k| <= sqrt(D) |g_k| (+ |b_k|);
      * a linear fed by a norm output: |sum_k W_jk (g_k x^_k + b_k) + c_j| <= sqrt(D) ||W_j o g||_2 + |W_j . b + c_j|   (Cauchy-Schwarz);
      * softmax probabilities <= 1; an attention output is a convex combination of value rows: <= the value bound;
      * a linear fed by a tensor with element bounds u: <= sum_k |W_jk| u_k + |c_j|;  |quick_gelu(t)|, |gelu(t)| <= |t|;
      * the tower's residual stream (what option proj_fp16 casts to 16 bits): pre-LN output bound + the sum of every block's two output bounds.
    -> {option: {"holds": bound <= FP16_HEAD at every site, "worst_site": name, "worst_bound": float}}.  Any device; a second or two at XXL."""
    W = lambda n: weights[n].detach().float()       # noqa: E731
    worst: Dict[str, Tuple[str, float]] = {k: ("", 0.0) for k in FP16_OPTIONS}

    def note(opt: str, site: str, b) -> float:
        b = float(b.max()) if torch.is_tensor(b) else float(b)
        if not (b <= worst[opt][1]):                 # NaN-poisoning: a NaN bound replaces everything and stays
            worst[opt] = (site, b)
        return b

    v, t = cfg.vision, cfg.t5
    # ---- vision tower blocks (vit_fp16) and the stream-fed projector (proj_fp16)
    D, R = v.hidden, float(v.hidden) ** 0.5
    u_stream = R * W("vision.pre_layrnorm.weight").abs() + W("vision.pre_layrnorm.bias").abs()
    for i in range(v.layers_run):
        p = f"vision.encoder.layers.{i}."
        g1, b1 = W(p + "layer_norm1.weight"), W(p + "layer_norm1.bias")
        note("vit_fp16", p + "layer_norm1", R * g1.abs() + b1.abs())
        vmax = 0.0
        for nm in ("q_proj", "k_proj", "v_proj"):
            w_ = W(p + f"self_attn.{nm}.weight")
            rows = R * (w_ * g1[None, :]).norm(dim=1) + (w_ @ b1 + W(p + f"self_attn.{nm}.bias")).abs()
            note("vit_fp16", p + nm, rows)
            if nm == "v_proj":
                vmax = float(rows.max())
        d_attn = W(p + "self_attn.out_proj.weight").abs().sum(dim=1) * vmax + W(p + "self_attn.out_proj.bias").abs()
        note("vit_fp16", p + "out_proj", d_attn)
        g2, b2 = W(p + "layer_norm2.weight"), W(p + "layer_norm2.bias")
        note("vit_fp16", p + "layer_norm2", R * g2.abs() + b2.abs())
        w1 = W(p + "mlp.fc1.weight")
        u1 = R * (w1 * g2[None, :]).norm(dim=1) + (w1 @ b2 + W(p + "mlp.fc1.bias")).abs()
        note("vit_fp16", p + "fc1", u1)
        d_mlp = W(p + "mlp.fc2.weight").abs() @ u1 + W(p + "mlp.fc2.bias").abs()
        note("vit_fp16", p + "fc2", d_mlp)
        u_stream = u_stream + d_attn + d_mlp
    # proj_fp16's two sites sit behind power-of-two scales (options proj_fs_shift / proj_mid_shift, set from these bounds at bind time): the stream
    # is a sum over every block, its worst-case bound is out of fp16's reach on any real tower
    b_fs = note("proj_fp16", "hidden_states[-2] (the residual stream)", u_stream)
    b_mid = note("proj_fp16", "mm_projector.0", W("mm_projector.0.weight").abs() @ u_stream + W("mm_projector.0.bias").abs())
    # ---- T5 encoder attention side (enc_fp16); its final norm output is what dec_fp16 reads
    D, R = t.d_model, float(t.d_model) ** 0.5
    for i in range(t.layers):
        p = f"encoder.block.{i}."
        g = W(p + "layer.0.layer_norm.weight")
        note("enc_fp16", p + "layer.0.layer_norm", R * g.abs())
        for nm in ("q", "k", "v"):
            note("enc_fp16", p + "SelfAttention." + nm, R * (W(p + f"layer.0.SelfAttention.{nm}.weight") * g[None, :]).norm(dim=1))
        note("enc_fp16", p + "layer.1.layer_norm", R * W(p + "layer.1.layer_norm.weight").abs())
    note("dec_fp16", "encoder.final_layer_norm", R * W("encoder.final_layer_norm.weight").abs())
    # ---- precise decoder's cross-attention score path (dec_fp16): cross q (norm-fed), q . Wk_h (64 terms per head)
    for i in range(t.dec_layers):
        p = f"decoder.block.{i}.layer.1."
        g = W(p + "layer_norm.weight")
        q_rows = R * (W(p + "EncDecAttention.q.weight") * g[None, :]).norm(dim=1)
        qmax = note("dec_fp16", p + "EncDecAttention.q", q_rows)
        wk = W(p + "EncDecAttention.k.weight").abs().reshape(t.heads, t.d_kv, D)
        note("dec_fp16", p + "q.Wk", wk.sum(dim=1) * qmax)
    out = {k: {"holds": bool(b <= FP16_HEAD), "worst_site": s_, "worst_bound": b} for k, (s_, b) in worst.items()}
    shifts = [fp16_shift(b_fs), fp16_shift(b_mid)]
    out["proj_fp16"].update({"fs_shift": shifts[0], "mid_shift": shifts[1], "holds": all(x is not None for x in shifts)})
    return out


def fp16_shift(bound: float) -> Optional[int]:
    """Smallest s >= 0 with bound * 2^-s <= FP16_HEAD (None: the bound is not finite, or beyond 2^60 x the head room)."""
    if not (bound >= 0.0) or bound == float("inf"):
        return None
    s_ = 0
    while bound * 2.0 ** -s_ > FP16_HEAD:
        s_ += 1
        if s_ > 60:
            return None
    return s_


class VqsEngine:
    """One CLIP-FlanT5 replica on one GPU."""

    def __init__(self, cfg: ClipT5Config, weights: Dict[str, torch.Tensor], device="cuda:0",
                 options: Optional[Dict[str, int]] = None):
        """options: execution-form switches of include/vqs.h vqs_set_option (parity tests A/B the alternative forms)."""
        self.lib = load_library()
        self.cfg = cfg
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise VqsError("VqsEngine needs a HIP device (device='cuda[:i]'); there is no CPU path")
        self._h = _c_vp()
        c = make_vqs_config(cfg)
        rc = self.lib.vqs_create(ctypes.byref(c), ctypes.byref(self._h))
        if rc != 0:
            msg = self.lib.vqs_last_error(self._h).decode() if self._h else "vqs_create failed"
            raise VqsError(f"vqs_create: {msg}")
        self._ws: Optional[torch.Tensor] = None
        self._ws_shape = None
        self._ews: Optional[torch.Tensor] = None
        self._options: Dict[str, int] = {}
        self._explicit = set(options or {})          # options the caller chose: honoured even where the range proof fails (the backstop stays)
        self.fp16_auto_off: Dict[str, str] = {}      # option -> why the bind-time checks switched a DEFAULT off
        for k, v in (options or {}).items():
            self.set_option(k, v)
        self.bind(weights)

    # ------------------------------------------------------------------ helpers
    def _check(self, rc: int, what: str):
        if rc != 0:
            raise VqsError(f"{what} failed ({rc}): {self.lib.vqs_last_error(self._h).decode()}")

    def close(self):
        if getattr(self, "_h", None):
            self.lib.vqs_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def bind(self, weights: Dict[str, torch.Tensor]):
        with torch.cuda.device(self.device):
            self.weights = {}
            for k, w in weights.items():
                if w.dtype != torch.bfloat16 or w.device != self.device or not w.is_contiguous():
                    w = w.to(device=self.device, dtype=torch.bfloat16).contiguous()
                self.weights[k] = w
            n = len(self.weights)
            descs = (VqsWeightDesc * n)()
            self._names = [k.encode() for k in self.weights]   # keep the char* alive
            for i, (k, w) in enumerate(self.weights.items()):
                descs[i] = VqsWeightDesc(self._names[i], w.data_ptr(), w.numel())
            nbytes = self.lib.vqs_packed_bytes(self._h)
            self._packed = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            rc = self.lib.vqs_bind_weights(self._h, descs, n, self._packed.data_ptr(), nbytes, _stream_ptr())
            self._check(rc, "vqs_bind_weights")
            torch.cuda.current_stream().synchronize()   # the bucket LUT upload reads handle-owned host memory
            self._check_fp16_weights()

    def _auto_off(self, opt: str, why: str):
        import warnings
        self._check(self.lib.vqs_set_option(self._h, opt.encode(), 0), "vqs_set_option")
        self._options[opt] = 0
        self.fp16_auto_off[opt] = why
        warnings.warn(f"t2v_metrics_amd: execution option {opt} switched off at bind time -- {why}; that stage runs on bf16 operands "
                      "(the reference's dtype)", RuntimeWarning, stacklevel=3)

    def _check_fp16_weights(self):
        """Bind-time range safety of the fp16 execution options (round 6).  (1) The fp16 copies vqs_bind_weights made must hold the weights:
        an option whose weights do not fit is refused if the caller asked for it, switched off with a warning if it was on by default.
        (2) fp16_range_proof: an option that is on BY DEFAULT stays on only where every one of its fp16 tensors has a proven bound under
        half of the fp16 maximum; otherwise it is switched off (one warning) -- a drop-in user never meets an overflow.  An explicitly
        requested option is honoured (the status word's bit 1 and the wrapper's re-score remain as the backstop)."""
        self._fp16_unsafe = fp16_unsafe_weights(self.weights)
        self._enc_fp16_unsafe = fp16_unsafe_weights(self.weights, "enc")
        self._dec_fp16_unsafe = fp16_unsafe_weights(self.weights, "dec")
        for opt, bad in (("vit_fp16", self._fp16_unsafe), ("enc_fp16", self._enc_fp16_unsafe), ("dec_fp16", self._dec_fp16_unsafe)):
            if bad and self.get_option(opt):
                if opt in self._explicit:
                    raise VqsError("%s = 1 refused: these weights exceed the fp16 range (|w| >= 65520): %s" % (opt, ", ".join(bad[:4])))
                self._auto_off(opt, "weights outside the fp16 range: " + ", ".join(bad[:4]))
        self.range_proof = fp16_range_proof(self.cfg, self.weights)
        pr = self.range_proof["proj_fp16"]
        if pr["holds"]:           # the projector's two stream-fed tensors go behind the scales their bounds ask for (0 = none: round 5's kernels)
            for opt, key in (("proj_fs_shift", "fs_shift"), ("proj_mid_shift", "mid_shift")):
                if opt not in self._explicit:
                    self._check(self.lib.vqs_set_option(self._h, opt.encode(), int(pr[key])), "vqs_set_option")
                    self._options[opt] = int(pr[key])
        for opt in FP16_OPTIONS:
            r = self.range_proof[opt]
            if not r["holds"] and self.get_option(opt) and opt not in self._explicit:
                self._auto_off(opt, "no proof that %s stays inside the fp16 range (bound %.3g from the weights alone)" % (r["worst_site"], r["worst_bound"]))

    # ------------------------------------------------------------------ the two stages
    def encode_images(self, pixels: torch.Tensor) -> torch.Tensor:
        """pixels bf16 [N,3,image,image] (CLIP-normalised) -> projected features bf16 [N, n_patches, d_model]."""
        v = self.cfg.vision
        if pixels.dim() != 4 or tuple(pixels.shape[1:]) != (3, v.image, v.image):
            raise VqsError(f"pixels must be [N,3,{v.image},{v.image}], got {tuple(pixels.shape)}")
        with torch.cuda.device(self.device):
            px = pixels.to(device=self.device, dtype=torch.bfloat16).contiguous()
            N = px.shape[0]
            feats = torch.empty(N, v.n_patches, self.cfg.t5.d_model, dtype=torch.bfloat16, device=self.device)
            need = self.lib.vqs_encode_workspace_bytes(self._h, N)
            if self._ews is None or self._ews.numel() < need:
                self._ews = None
                self._ews = torch.empty(need, dtype=torch.uint8, device=self.device)
            self._ews_n = N
            rc = self.lib.vqs_encode_images(self._h, px.data_ptr(), N, feats.data_ptr(), self._ews.data_ptr(),
                                            self._ews.numel(), _stream_ptr())
            self._check(rc, "vqs_encode_images")
            return feats

    def score(self, feats: torch.Tensor, img_index: torch.Tensor, input_ids: torch.Tensor,
              labels: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """-> (label_logprobs fp32 [B,T], scores fp32 [B]).  input_ids int [B,L] with one -200 per row and
        trailing 0 padding; labels int [B,T] with -100 padding."""
        with torch.cuda.device(self.device):
            B, L = input_ids.shape
            T = labels.shape[1]
            if labels.shape[0] != B or img_index.shape[0] != B:
                raise VqsError("img_index, input_ids and labels must agree on the batch size")
            ids = input_ids.to(device=self.device, dtype=torch.int32).contiguous()
            lab = labels.to(device=self.device, dtype=torch.int32).contiguous()
            idx = img_index.to(device=self.device, dtype=torch.int32).contiguous()
            feats = feats.contiguous()
            lp = torch.empty(B, T, dtype=torch.float32, device=self.device)
            sc = torch.empty(B, dtype=torch.float32, device=self.device)
            need = self.lib.vqs_score_workspace_bytes(self._h, B, L, T)
            if need == 0:
                raise VqsError(f"unsupported score shape B={B} L={L} T={T}")
            if self._ws is None or self._ws.numel() < need:
                self._ws = None
                self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
            self._ws_shape = (B, L, T)
            rc = self.lib.vqs_score(self._h, feats.data_ptr(), idx.data_ptr(), ids.data_ptr(), lab.data_ptr(), B, L, T,
                                    lp.data_ptr(), sc.data_ptr(), self._ws.data_ptr(), self._ws.numel(), _stream_ptr())
            self._check(rc, "vqs_score")
            self._enc_out_f16 = bool(self.get_option("dec_precise") and self.get_option("cross_mode") and self.get_option("dec_fp16"))
            return lp, sc

    def normalize_u8(self, x_u8: torch.Tensor, mean, std) -> torch.Tensor:
        with torch.cuda.device(self.device):
            return normalize_u8(x_u8.to(self.device), mean, std)

    def generate(self, feats: torch.Tensor, img_index: torch.Tensor, input_ids: torch.Tensor,
                 max_new_tokens: int = 16) -> torch.Tensor:
        """Greedy decoding -> int32 [B, max_new_tokens] on the device (every step executed; cut at the first EOS = 1 on
        the host).  Incremental decoding over a K/V cache in the workspace; max_new_tokens <= 512."""
        with torch.cuda.device(self.device):
            B, L = input_ids.shape
            if img_index.shape[0] != B:
                raise VqsError("img_index and input_ids must agree on the batch size")
            ids = input_ids.to(device=self.device, dtype=torch.int32).contiguous()
            idx = img_index.to(device=self.device, dtype=torch.int32).contiguous()
            feats = feats.contiguous()
            tokens = torch.empty(B, max_new_tokens, dtype=torch.int32, device=self.device)
            need = self.lib.vqs_generate_workspace_bytes(self._h, B, L, max_new_tokens)
            if need == 0:
                raise VqsError(f"unsupported generate shape B={B} L={L} max_new_tokens={max_new_tokens}")
            if self._ws is None or self._ws.numel() < need:
                self._ws = None
                self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
            self._ws_shape = (B, L, 1)
            rc = self.lib.vqs_generate(self._h, feats.data_ptr(), idx.data_ptr(), ids.data_ptr(), B, L, max_new_tokens,
                                       tokens.data_ptr(), self._ws.data_ptr(), self._ws.numel(), _stream_ptr())
            self._check(rc, "vqs_generate")
            self._enc_out_f16 = False          # generation runs the bf16 decoder over a bf16 encoder output
            return tokens

    # ------------------------------------------------------------------ introspection (tests / bench)
    def stage(self, name: str) -> torch.Tensor:
        """View of a named intermediate of the most recent call (see vqs_workspace_offset)."""
        ld = _c_i64(0)
        t5, v = self.cfg.t5, self.cfg.vision
        if name == "vit_hidden":
            N = self._ews_n
            off = self.lib.vqs_workspace_offset(self._h, name.encode(), N, 0, 0, ctypes.byref(ld))
            if off < 0:
                raise VqsError(f"unknown stage {name}")
            n = N * v.seq * v.hidden
            return self._ews[off: off + 4 * n].view(torch.float32).view(N, v.seq, v.hidden)
        B, L, T = self._ws_shape
        off = self.lib.vqs_workspace_offset(self._h, name.encode(), B, L, T, ctypes.byref(ld))
        if off < 0:
            raise VqsError(f"unknown stage {name}")
        S = L - 1 + v.n_patches
        if name == "enc_in":
            return self._ws[off: off + 4 * B * S * t5.d_model].view(torch.float32).view(B, S, t5.d_model)
        if name == "enc_out":
            # IEEE fp16 when the last call was a scoring pass of the precise decoder with option dec_fp16 (default), bf16 otherwise
            dt = torch.float16 if getattr(self, "_enc_out_f16", False) else torch.bfloat16
            return self._ws[off: off + 2 * B * S * t5.d_model].view(dt).view(B, S, t5.d_model)
        if name == "dec_out":
            # the final norm's output is a split-bf16 tensor (planes hi | lo) under the precise decoder: its value is hi + lo;
            # the bf16 decoder (option dec_precise=0, vqs_generate) writes plane 0 only
            planes = self._ws[off: off + 2 * 2 * B * T * t5.d_model].view(torch.bfloat16).view(2, B, T, t5.d_model)
            if self._options.get("dec_precise", 1) and self._options.get("cross_mode", 1):
                return planes[0].float() + planes[1].float()
            return planes[0]
        if name == "logits":
            return self._ws[off: off + 4 * B * T * ld.value].view(torch.float32).view(B, T, ld.value)[..., : t5.vocab]
        if name == "enc_len":
            return self._ws[off: off + 4 * B].view(torch.int32)
        if name == "flags":
            return self._ws[off: off + 4].view(torch.int32)
        raise VqsError(f"unknown stage {name}")

    def get_option(self, name: str) -> int:
        """Current value of a scalar execution-form option (include/vqs.h), defaults included."""
        v = _c_i32(0)
        self._check(self.lib.vqs_get_option(self._h, name.encode(), ctypes.byref(v)), "vqs_get_option")
        return int(v.value)

    def set_option(self, name: str, value: int):
        if name == "vit_fp16" and int(value) == 1 and getattr(self, "_fp16_unsafe", None):
            raise VqsError("vit_fp16 = 1 refused: weights outside the fp16 range: " + ", ".join(self._fp16_unsafe[:4]))
        if name == "enc_fp16" and int(value) == 1 and getattr(self, "_enc_fp16_unsafe", None):
            raise VqsError("enc_fp16 = 1 refused: weights outside the fp16 range: " + ", ".join(self._enc_fp16_unsafe[:4]))
        if name == "dec_fp16" and int(value) == 1 and getattr(self, "_dec_fp16_unsafe", None):
            raise VqsError("dec_fp16 = 1 refused: weights outside the fp16 range: " + ", ".join(self._dec_fp16_unsafe[:4]))
        self._check(self.lib.vqs_set_option(self._h, name.encode(), int(value)), "vqs_set_option")
        self._options[name] = int(value)

    def tap(self, name: Optional[str], dst: Optional[torch.Tensor] = None):
        """Register `dst` (device tensor, kept alive by the caller) to receive the named intermediate of the next passes
        (vqs_debug_tap); name None clears all taps."""
        if name is None:
            self._check(self.lib.vqs_debug_tap(self._h, None, None, 0), "vqs_debug_tap")
            return
        nbytes = 0 if dst is None else dst.numel() * dst.element_size()
        self._check(self.lib.vqs_debug_tap(self._h, name.encode(), _ptr(dst), nbytes), "vqs_debug_tap")

    def tap_window(self, first: int = 0, count: int = 0):
        """Restrict the taps to `count` consecutive pairs (T5 stacks) / images (vision tower) starting at `first`
        (vqs_debug_tap_window); count 0 = whole tensors."""
        self._check(self.lib.vqs_debug_tap_window(self._h, int(first), int(count)), "vqs_debug_tap_window")

    def profile(self, on: bool):
        self._check(self.lib.vqs_profile_enable(self._h, 1 if on else 0), "vqs_profile_enable")

    def profile_read(self, reset: bool = True):
        ms, fl = ctypes.c_double(0), ctypes.c_double(0)
        n = self.lib.vqs_profile_read(self._h, ctypes.byref(ms), ctypes.byref(fl), 1 if reset else 0)
        if n < 0:
            self._check(n, "vqs_profile_read")
        return n, ms.value, fl.value

    def profile_report(self) -> str:
        """Per-call-site GEMM table of the launches covered by the last profile_read()."""
        return self.lib.vqs_profile_report(self._h).decode()

    def profile_bytes(self) -> float:
        """Algorithmic bytes of the GEMM launches since the last reset (read it BEFORE a resetting profile_read)."""
        b = ctypes.c_double(0)
        self._check(self.lib.vqs_profile_bytes(self._h, ctypes.byref(b)), "vqs_profile_bytes")
        return b.value


# ---------------------------------------------------------------------- single-kernel wrappers (tests, microbench)
def gemm(A, W, epilogue: int, bias=None, resid=None, out=None, S: int = 0, H: int = 0, variant: int = 0, tile_order=None,
         nt_store: bool = False, l2_touch: int = 0, ftype: int = 0):
    """C = epilogue(A @ W.T).  A [M,K] bf16, W [N,K] bf16.  See include/vqs.h for epilogue codes.
    tile_order = (gm, ns): workgroup -> tile order of the launch (a permutation of the tile list); nt_store: non-temporal result
    stores; l2_touch: 1 / 2 = A-panel L2 prefetch on / off (0: by shape).  All three are bitwise-neutral.
    ftype (bits 27-28 of the variant word): 0 bf16 tensors; 1 A, W and the 16-bit result IEEE fp16; 2 A, W fp16, result bf16."""
    if tile_order is not None:
        variant = (variant & 0xff) | (int(tile_order[0]) << 8) | (int(tile_order[1]) << 16)
    variant |= (1 << 24 if nt_store else 0) | ((int(l2_touch) & 3) << 25) | ((int(ftype) & 3) << 27)
    t16 = torch.float16 if ftype == 1 else torch.bfloat16
    if A.dtype != (torch.float16 if ftype else torch.bfloat16) or W.dtype != A.dtype:
        raise VqsError("gemm: operand dtype does not match ftype")
    lib = load_library()
    M, K = A.shape
    N = W.shape[0]
    dev = A.device
    if epilogue in (0, 1, 2):
        out = torch.empty(M, N, dtype=t16, device=dev) if out is None else out
        ldc = N
    elif epilogue in (3, 4):
        out = torch.empty(M, N, dtype=torch.float32, device=dev) if out is None else out
        ldc = N
    elif epilogue == 5:
        out = torch.empty(M, N // 2, dtype=t16, device=dev) if out is None else out
        ldc = N // 2
    elif epilogue == 6:
        nsel = N // (H * 64)
        out = torch.empty(nsel, M // S, H, S, 64, dtype=t16, device=dev) if out is None else out
        ldc = 0
    else:
        raise ValueError(epilogue)
    rc = lib.vqs_gemm(A.data_ptr(), W.data_ptr(), out.data_ptr(), _ptr(bias), _ptr(resid), M, N, K, A.stride(0),
                      W.stride(0), ldc, epilogue, S, H, variant, _stream_ptr())
    if rc != 0:
        raise VqsError(f"vqs_gemm failed ({rc})")
    return out


def gemm_batched(A, W, epilogue: int, split: bool = False, no_stream: bool = False, variant: int = 3, ldc: int = None, ftype: int = 0):
    """Test hook (vqs_debug_gemm_batched): C[z] = A[z] @ W[z].T for A [Z, M, K] bf16 (any row stride), W [Z, N, K] bf16, epilogue 3
    (fp32) or 0 (bf16; split=True also returns the lo plane of the split-bf16 result).  -> C [Z, M, ldc] (and lo).
    ftype (bits 27-28 of the variant word): 1 = fp16 operands (and an fp16 result for epilogue 0), 2 = fp16 operands, bf16 / split-bf16 result."""
    lib = load_library()
    Z, M, K = A.shape
    N = W.shape[1]
    ldc = N if ldc is None else ldc
    if A.dtype != (torch.float16 if ftype else torch.bfloat16) or W.dtype != A.dtype:
        raise VqsError("gemm_batched: operand dtype does not match ftype")
    variant |= (int(ftype) & 3) << 27
    dt = torch.float32 if epilogue == 3 else (torch.float16 if ftype == 1 else torch.bfloat16)
    out = torch.zeros((2 if split else 1), Z, M, ldc, dtype=dt, device=A.device)
    rc = lib.vqs_debug_gemm_batched(A.data_ptr(), W.data_ptr(), out.data_ptr(), M, N, K, A.stride(1), W.stride(1), ldc, epilogue, Z,
                                    A.stride(0), W.stride(0), M * ldc, (Z * M * ldc) if split else 0, 1 if no_stream else 0, variant,
                                    _stream_ptr())
    if rc != 0:
        raise VqsError(f"vqs_debug_gemm_batched failed ({rc})")
    return (out[0], out[1]) if split else out[0]


def gemm_resid_rms(A, W, hres, lnw, variant: int = 3):
    """Producer of the fused residual + RMSNorm: hres (fp32 [M,N], in place) += A @ W.T; returns
    (xhat bf16 [M,N] = hres * lnw, rowss fp32 [ceil(N/256), M] partial row sums of squares)."""
    lib = load_library()
    M, K = A.shape
    N = W.shape[0]
    xhat = torch.empty(M, N, dtype=torch.bfloat16, device=A.device)
    rowss = torch.zeros((N + 255) // 256, M, dtype=torch.float32, device=A.device)
    rc = lib.vqs_gemm_rms(A.data_ptr(), W.data_ptr(), xhat.data_ptr(), hres.data_ptr(), lnw.data_ptr(), rowss.data_ptr(), None,
                          0, 0.0, 0.0, M, N, K, A.stride(0), W.stride(0), N, 7, 0, 0, variant, _stream_ptr())
    if rc != 0:
        raise VqsError(f"vqs_gemm_rms (producer) failed ({rc})")
    return xhat, rowss


def gemm_rowscaled(A, W, epilogue: int, rowss, d_norm: int, eps: float, S: int = 0, H: int = 0, variant: int = 3):
    """Consumer: epilogue(rsqrt(mean_sq(row) + eps) * (A @ W.T)) with mean_sq from the producer's partial sums."""
    lib = load_library()
    M, K = A.shape
    N = W.shape[0]
    dev = A.device
    if epilogue == 0:
        out, ldc = torch.empty(M, N, dtype=torch.bfloat16, device=dev), N
    elif epilogue == 3:
        out, ldc = torch.empty(M, N, dtype=torch.float32, device=dev), N
    elif epilogue == 5:
        out, ldc = torch.empty(M, N // 2, dtype=torch.bfloat16, device=dev), N // 2
    elif epilogue == 6:
        out, ldc = torch.empty(N // (H * 64), M // S, H, S, 64, dtype=torch.bfloat16, device=dev), 0
    else:
        raise ValueError(epilogue)
    rc = lib.vqs_gemm_rms(A.data_ptr(), W.data_ptr(), out.data_ptr(), None, None, None, rowss.data_ptr(), rowss.shape[0],
                          1.0 / d_norm, eps, M, N, K, A.stride(0), W.stride(0), ldc, epilogue, S, H, variant, _stream_ptr())
    if rc != 0:
        raise VqsError(f"vqs_gemm_rms (consumer) failed ({rc})")
    return out


def attention(q, k, v, scale: float, bias_table=None, key_len=None):
    """q, k, v [B,H,S,64] bf16 -> bf16 [B*S, H*64]; fp16 tensors run the fp16 instantiation (test hook vqs_debug_attention_f16)."""
    lib = load_library()
    B, H, S, d = q.shape
    assert d == 64 and k.dtype == q.dtype and v.dtype == q.dtype
    out = torch.empty(B * S, H * 64, dtype=q.dtype, device=q.device)
    fn = lib.vqs_debug_attention_f16 if q.dtype == torch.float16 else lib.vqs_attention
    rc = fn(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), _ptr(bias_table), _ptr(key_len), B, H, S, scale, _stream_ptr())
    if rc != 0:
        raise VqsError(f"vqs_attention failed ({rc})")
    return out


def norm16(kind: int, x, delta, w, b=None, delta2=None, store_x: bool = True, types: int = 1, eps: float = 1e-6):
    """Test hook (vqs_debug_norm16): the norm kernels' 16-bit forms.  x fp32 [M,D] (updated in place as the engine's stream is),
    types 1 = RMSNorm with bf16 deltas and an fp16 operand out, 3 = LayerNorm with fp16 deltas and an fp16 operand out."""
    lib = load_library()
    M, D = x.shape
    out = torch.empty(M, D, dtype=torch.float16, device=x.device)
    rc = lib.vqs_debug_norm16(kind, x.data_ptr(), _ptr(delta), _ptr(delta2), 1 if store_x else 0, w.data_ptr(), _ptr(b), out.data_ptr(), M, D,
                              eps, types, _stream_ptr())
    if rc != 0:
        raise VqsError(f"vqs_debug_norm16 failed ({rc})")
    return out


def normalize_u8(x_u8: torch.Tensor, mean, std) -> torch.Tensor:
    """uint8 [N,H,W,3] on the device -> CLIP-normalised bf16 [N,3,H,W] (the fp32 arithmetic of the HF processor)."""
    lib = load_library()
    N, H, W, C = x_u8.shape
    if C != 3 or x_u8.dtype != torch.uint8 or not x_u8.is_cuda:
        raise VqsError("normalize_u8 expects a uint8 [N,H,W,3] tensor on the GPU")
    out = torch.empty(N, 3, H, W, dtype=torch.bfloat16, device=x_u8.device)
    m3 = (ctypes.c_float * 3)(*[float(v) for v in mean])
    s3 = (ctypes.c_float * 3)(*[float(v) for v in std])
    rc = lib.vqs_normalize_u8(x_u8.contiguous().data_ptr(), out.data_ptr(), N, H, W, m3, s3, _stream_ptr())
    if rc != 0:
        raise VqsError(f"vqs_normalize_u8 failed ({rc})")
    return out


def rope_(x, cos, sin):
    """In-place rotate-half RoPE on x bf16 [B,H,S,hd] with cos/sin fp32 [B*S, half]."""
    lib = load_library()
    B, H, S, hd = x.shape
    half = cos.shape[-1]
    rc = lib.vqs_rope(x.data_ptr(), cos.data_ptr(), sin.data_ptr(), B, H, S, hd, half, _stream_ptr())
    if rc != 0:
        raise VqsError(f"vqs_rope failed ({rc})")
    return x


def attention_hd(q, k, v, scale: float, causal: bool = False, key_len=None):
    """q [B,H,S,128], k/v [B,Hkv,S,128] bf16 -> out bf16 [B*S, H*128] (GQA, optional causal mask)."""
    lib = load_library()
    B, H, S, d = q.shape
    Hkv = k.shape[1]
    out = torch.empty(B * S, H * d, dtype=torch.bfloat16, device=q.device)
    rc = lib.vqs_attention_hd(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), _ptr(key_len), B, H, Hkv, S, d, scale,
                              1 if causal else 0, _stream_ptr())
    if rc != 0:
        raise VqsError(f"vqs_attention_hd failed ({rc})")
    return out


def decoder_attention(q, k, v, B, H, T, S, ldq, ldk, cross: bool, bias_table=None, key_len=None):
    lib = load_library()
    out = torch.empty(B * T, H * 64, dtype=torch.bfloat16, device=q.device)
    rc = lib.vqs_decoder_attention(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), _ptr(bias_table),
                                   _ptr(key_len), B, H, T, S, ldq, ldk, 1 if cross else 0, _stream_ptr())
    if rc != 0:
        raise VqsError(f"vqs_decoder_attention failed ({rc})")
    return out


def rmsnorm(x, w, eps, delta=None):
    """out = rmsnorm(x [+ delta]) * w; with delta (bf16 [M,D]), the fp32 x is updated in place (x += delta)."""
    lib = load_library()
    M, D = x.shape
    if delta is not None and (delta.dtype != torch.bfloat16 or tuple(delta.shape) != (M, D) or not delta.is_contiguous()):
        raise VqsError("rmsnorm: delta must be a contiguous bf16 [M, D] tensor")
    out = torch.empty(M, D, dtype=torch.bfloat16, device=x.device)
    rc = lib.vqs_rmsnorm(x.data_ptr(), _ptr(delta), w.data_ptr(), out.data_ptr(), M, D, eps, _stream_ptr())
    if rc != 0:
        raise VqsError(f"vqs_rmsnorm failed ({rc})")
    return out


def layernorm(x, w, b, eps, out_f32=False, delta=None):
    lib = load_library()
    M, D = x.shape
    if delta is not None and (delta.dtype != torch.bfloat16 or tuple(delta.shape) != (M, D) or not delta.is_contiguous()):
        raise VqsError("layernorm: delta must be a contiguous bf16 [M, D] tensor")
    out = torch.empty(M, D, dtype=torch.float32 if out_f32 else torch.bfloat16, device=x.device)
    rc = lib.vqs_layernorm(x.data_ptr(), _ptr(delta), w.data_ptr(), b.data_ptr(), out.data_ptr(), 1 if out_f32 else 0, M, D,
                           eps, _stream_ptr())
    if rc != 0:
        raise VqsError(f"vqs_layernorm failed ({rc})")
    return out


def norm_deferred(x, w, eps, delta, delta2=None, store_x=True, b=None):
    """The deferred-store norm forms (include/vqs.h): (delta, store_x=False) leaves x untouched; (delta, delta2) stores
    x = (x + delta) + delta2.  RMSNorm when b is None, LayerNorm otherwise; bf16 output."""
    lib = load_library()
    M, D = x.shape
    for d in (delta, delta2):
        if d is not None and (d.dtype != torch.bfloat16 or tuple(d.shape) != (M, D) or not d.is_contiguous()):
            raise VqsError("norm_deferred: deltas must be contiguous bf16 [M, D] tensors")
    out = torch.empty(M, D, dtype=torch.bfloat16, device=x.device)
    rc = lib.vqs_norm_deferred(0 if b is None else 1, x.data_ptr(), _ptr(delta), _ptr(delta2), 1 if store_x else 0, w.data_ptr(),
                               _ptr(b), out.data_ptr(), M, D, eps, _stream_ptr())
    if rc != 0:
        raise VqsError(f"vqs_norm_deferred failed ({rc})")
    return out


def score_head(logits, labels):
    lib = load_library()
    B, T, V = logits.shape
    lg = logits.contiguous()
    lab = labels.to(torch.int32).contiguous()
    lp = torch.empty(B, T, dtype=torch.float32, device=logits.device)
    sc = torch.empty(B, dtype=torch.float32, device=logits.device)
    rc = lib.vqs_score_head(lg.data_ptr(), V, V, lab.data_ptr(), lp.data_ptr(), sc.data_ptr(), B, T, _stream_ptr())
    if rc != 0:
        raise VqsError(f"vqs_score_head failed ({rc})")
    return lp, sc


def relpos_bucket(rel: int, bidirectional: bool, num_buckets: int = 32, max_distance: int = 128) -> int:
    return load_library().vqs_relpos_bucket(rel, 1 if bidirectional else 0, num_buckets, max_distance)
