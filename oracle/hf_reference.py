"""THE REFERENCE'S OWN ARITHMETIC, assembled from the HuggingFace modules it calls.  TEST INFRASTRUCTURE ONLY
(tests/, bench.py's ``cpu_baseline`` leg, tools/config1_cpu.py, tools/rounding_chaos.py).

The CLIP-FlanT5 wrapper is not in /root/reference (v3.1 dropped it); what it executed is
``CLIPVisionModel`` -> ``hidden_states[-2][:, 1:]`` -> ``mlp2x_gelu`` -> splice at the -200 sentinel ->
``T5ForConditionalGeneration(inputs_embeds, attention_mask, labels)`` -> ``exp(-CE)``, with every module cast to bf16
(/root/reference/t2v_metrics/models/vqascore_models/mm_utils.py:228), eval mode, no cache (:236-240), inference
mode.  This module builds exactly that from ``transformers`` (pin >= 4.52, pyproject.toml:26; 5.x installed) and the
same named weight tensors the HIP engine binds, so "reference as shipped" (dtype=bf16) and "reference arithmetic in
fp32" can be timed and compared on any host -- including the GPU box, where /root/reference does not exist.

Modules are created on the meta device and the caller's tensors are assigned (no 46 GB fp32 random init for XXL).
HF-5.x quirks: lm_head re-created untied, ``scale_decoder_outputs`` False, dropout 0.
"""
from __future__ import annotations

import time
from typing import Dict

import torch


def _vision(cfg, weights, dtype):
    from transformers import CLIPVisionConfig, CLIPVisionModel
    v = cfg.vision
    hc = CLIPVisionConfig(hidden_size=v.hidden, intermediate_size=v.mlp, num_hidden_layers=v.layers,
                          num_attention_heads=v.heads, image_size=v.image, patch_size=v.patch,
                          hidden_act="quick_gelu", layer_norm_eps=v.ln_eps, attention_dropout=0.0)
    with torch.device("meta"):
        m = CLIPVisionModel(hc)
    sd = {k[len("vision."):]: w.detach().to("cpu", dtype) for k, w in weights.items() if k.startswith("vision.")}
    own = dict(m.state_dict())
    if not any(k in own for k in sd):                       # transformers 4.x nests the tower under .vision_model
        sd = {"vision_model." + k: w for k, w in sd.items()}
    for k, t in own.items():                                # post_layernorm is unused (feature = hidden_states[-2])
        if k not in sd:
            sd[k] = torch.zeros(t.shape, dtype=dtype) if "post_layernorm" in k else None
    sd = {k: w for k, w in sd.items() if w is not None}
    missing, unexpected = m.load_state_dict(sd, strict=False, assign=True)
    assert not unexpected and all("position_ids" in k for k in missing), (missing, unexpected)
    emb = (m.vision_model if hasattr(m, "vision_model") else m).embeddings
    emb.position_ids = torch.arange(v.seq).unsqueeze(0)     # non-persistent buffer, still on meta
    return m.eval()


def _t5(cfg, weights, dtype):
    from transformers import T5Config, T5ForConditionalGeneration
    t = cfg.t5
    hc = T5Config(vocab_size=t.vocab, d_model=t.d_model, d_kv=t.d_kv, d_ff=t.d_ff, num_layers=t.layers,
                  num_decoder_layers=t.dec_layers, num_heads=t.heads,
                  relative_attention_num_buckets=t.rel_buckets, relative_attention_max_distance=t.rel_max_distance,
                  dropout_rate=0.0, layer_norm_epsilon=t.ln_eps, feed_forward_proj="gated-gelu",
                  tie_word_embeddings=False, pad_token_id=t.pad_id, eos_token_id=t.eos_id,
                  decoder_start_token_id=t.decoder_start_id)
    with torch.device("meta"):
        m = T5ForConditionalGeneration(hc)
        m.lm_head = torch.nn.Linear(t.d_model, t.vocab, bias=False)      # untied (flan-t5); HF 5.x ties on construction
    m.config.tie_word_embeddings = False
    m.config.scale_decoder_outputs = False
    sd = {k: w.detach().to("cpu", dtype) for k, w in weights.items() if not (k.startswith("vision.") or k.startswith("mm_projector."))}
    sd["encoder.embed_tokens.weight"] = sd["shared.weight"]
    sd["decoder.embed_tokens.weight"] = sd["shared.weight"]
    missing, unexpected = m.load_state_dict(sd, strict=False, assign=True)
    assert not missing and not unexpected, (missing, unexpected)
    return m.eval()


class HFReference:
    """``dtype=torch.bfloat16`` = the reference as shipped; ``torch.float32`` = its arithmetic on the same bf16-rounded
    weights in fp32.  ``attn`` selects HF's attention implementation ("sdpa" is what 5.x resolves to, "eager" the
    4.36-era path of the v3.0 release)."""

    def __init__(self, cfg, weights: Dict[str, torch.Tensor], dtype=torch.bfloat16, attn: str | None = None):
        self.cfg, self.dtype = cfg, dtype
        self.vm = _vision(cfg, weights, dtype)
        self.tm = _t5(cfg, weights, dtype)
        if attn is not None:
            for m in (self.vm, self.tm):
                m.config._attn_implementation = attn
        v, t = cfg.vision, cfg.t5
        self.proj = torch.nn.Sequential(torch.nn.Linear(v.hidden, t.d_model), torch.nn.GELU(), torch.nn.Linear(t.d_model, t.d_model))
        self.proj = self.proj.to(dtype).eval()
        with torch.no_grad():
            for i in (0, 2):
                self.proj[i].weight.copy_(weights[f"mm_projector.{i}.weight"].to("cpu", dtype))
                self.proj[i].bias.copy_(weights[f"mm_projector.{i}.bias"].to("cpu", dtype))
        self.stage_s = {}

    @torch.inference_mode()
    def encode_images(self, pixel_values: torch.Tensor) -> torch.Tensor:
        v = self.cfg.vision
        hs = self.vm(pixel_values=pixel_values.to(self.dtype), output_hidden_states=True).hidden_states
        return self.proj(hs[v.select_layer][:, 1:])

    @torch.inference_mode()
    def score(self, feats: torch.Tensor, img_index: torch.Tensor, input_ids: torch.Tensor, labels: torch.Tensor):
        t = self.cfg.t5
        B, L = input_ids.shape
        S_e = L - 1 + self.cfg.vision.n_patches
        emb = torch.zeros(B, S_e, t.d_model, dtype=self.dtype)
        mask = torch.zeros(B, S_e, dtype=torch.long)
        for b in range(B):
            row = input_ids[b][input_ids[b] != t.pad_id].long()
            sp = int((row == -200).nonzero()[0, 0])
            r = torch.cat([self.tm.shared(row[:sp]), feats[int(img_index[b])], self.tm.shared(row[sp + 1:])], 0)
            emb[b, : r.shape[0]] = r
            mask[b, : r.shape[0]] = 1
        keep = int(mask.sum(1).max())                        # the reference pads to the batch maximum only
        labels = labels.long()
        out = self.tm(inputs_embeds=emb[:, :keep], attention_mask=mask[:, :keep], labels=labels)
        logits = out.logits.float()
        self.last_logits = logits                            # tools/pin_oracle_fullsize.py compares them with the fp32 oracle's
        safe = labels.clamp(min=0)
        lp = torch.log_softmax(logits, -1).gather(-1, safe[..., None])[..., 0]
        lp = torch.where(labels == -100, torch.zeros_like(lp), lp)
        valid = (labels != -100).float()
        scores = torch.exp((lp * valid).sum(-1) / valid.sum(-1).clamp(min=1.0))
        return lp, scores

    def forward(self, pixel_values, img_index, input_ids, labels, timed: bool = False):
        t0 = time.perf_counter()
        feats = self.encode_images(pixel_values)
        t1 = time.perf_counter()
        lp, sc = self.score(feats, img_index, input_ids, labels)
        t2 = time.perf_counter()
        if timed:
            self.stage_s = {"vision+projector_s": t1 - t0, "t5_encoder+decoder+head_s": t2 - t1}
        return {"label_logprobs": lp, "scores": sc}


class HFEngine:
    """HFReference behind the engine interface of CLIPT5Model (encode_images / score), so the drop-in API can be driven
    on the CPU by the reference's own arithmetic: VQAScore(device='cpu', engine=HFEngine(cfg, weights)).  Checker /
    CPU-baseline only -- the product path constructs VqsEngine and has no CPU route."""

    def __init__(self, cfg, weights, dtype=torch.bfloat16, attn=None):
        self.ref = HFReference(cfg, weights, dtype, attn)
        self.cfg = cfg
        self.seconds = {"vision+projector": 0.0, "t5": 0.0}

    def encode_images(self, pixels):
        t0 = time.perf_counter()
        out = self.ref.encode_images(pixels)
        self.seconds["vision+projector"] += time.perf_counter() - t0
        return out

    def score(self, feats, img_index, input_ids, labels):
        t0 = time.perf_counter()
        out = self.ref.score(feats, img_index, input_ids, labels)
        self.seconds["t5"] += time.perf_counter() - t0
        return out
