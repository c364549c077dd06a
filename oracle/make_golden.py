"""Generate tests/golden/*.npz from the HuggingFace modules the reference executes.

Run in the build container (transformers 5.15.0 importable, no GPU needed):

    python oracle/make_golden.py

What is pinned (SURVEY.md §8c -- the reference itself has no golden vectors for this path):
  * relpos_buckets.npz   -- HF ``T5Attention._relative_position_bucket`` for every
                            relative position in [-700, 700], bidirectional and causal.
  * e2e_<cfg>_g<gain>.npz -- the whole pipeline assembled from HF modules, run in fp32 ("truth") and with every
                            module cast to bf16 as the reference does (mm_utils.py:228) on the CPU: label log-probs
                            and scores of both, plus the inputs.  The bf16 run is the reference's own numerical
                            noise floor against which the HIP path's error is judged.
  * clip_preprocess.npz  -- HF ``CLIPImageProcessor`` (resize shortest edge, centre crop, rescale, normalise) on seeded
                            uint8 images of several aspect ratios.
  * hf_tiny.npz / hf_small.npz
      - ``CLIPVisionModel(..., output_hidden_states=True).hidden_states[-2]`` on seeded pixels,
      - ``T5ForConditionalGeneration(inputs_embeds=..., attention_mask=..., labels=...)``
        logits and per-sample ``exp(-CrossEntropyLoss(mean))`` on seeded embeddings with a
        ragged key mask and ragged (-100 padded) labels,
    all from HF modules in fp32 loaded with the bf16-rounded seeded weights of
    ``t2v_metrics_amd.weights``.  The glue between the two modules (feature select, projector,
    splice) is not HF code and is pinned separately by hand-built cases in tests/test_oracle_*.py.

The HF-5.x quirks handled here (SURVEY.md §8c "Oracle gotchas"): lm_head is re-created untied,
``scale_decoder_outputs`` is forced False, dropout 0 / eval mode.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from t2v_metrics_amd.config import get_config  # noqa: E402
from t2v_metrics_amd.weights import make_seeded_weights  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def build_hf_vision(cfg, weights):
    from transformers import CLIPVisionConfig, CLIPVisionModel
    v = cfg.vision
    hc = CLIPVisionConfig(hidden_size=v.hidden, intermediate_size=v.mlp, num_hidden_layers=v.layers,
                          num_attention_heads=v.heads, image_size=v.image, patch_size=v.patch,
                          hidden_act="quick_gelu", layer_norm_eps=v.ln_eps, attention_dropout=0.0)
    m = CLIPVisionModel(hc).eval().float()
    sd = {k[len("vision."):]: w.float() for k, w in weights.items() if k.startswith("vision.")}
    missing, unexpected = m.load_state_dict(sd, strict=False)
    # post_layernorm is unused by the path (feature comes from hidden_states[-2]); position_ids is a buffer
    assert all(("post_layernorm" in k) or ("position_ids" in k) for k in missing), missing
    assert not unexpected, unexpected
    return m


def build_hf_t5(cfg, weights):
    from transformers import T5Config, T5ForConditionalGeneration
    t = cfg.t5
    hc = T5Config(vocab_size=t.vocab, d_model=t.d_model, d_kv=t.d_kv, d_ff=t.d_ff, num_layers=t.layers,
                  num_decoder_layers=t.dec_layers, num_heads=t.heads,
                  relative_attention_num_buckets=t.rel_buckets, relative_attention_max_distance=t.rel_max_distance,
                  dropout_rate=0.0, layer_norm_epsilon=t.ln_eps, feed_forward_proj="gated-gelu",
                  tie_word_embeddings=False, pad_token_id=t.pad_id, eos_token_id=t.eos_id,
                  decoder_start_token_id=t.decoder_start_id)
    m = T5ForConditionalGeneration(hc).eval().float()
    # HF 5.x ties lm_head to shared on construction; flan-t5 is untied.
    m.lm_head.weight = torch.nn.Parameter(torch.empty_like(m.shared.weight))
    m.config.tie_word_embeddings = False
    m.config.scale_decoder_outputs = False
    sd = {k: w.float() for k, w in weights.items() if not (k.startswith("vision.") or k.startswith("mm_projector."))}
    sd["encoder.embed_tokens.weight"] = sd["shared.weight"]
    sd["decoder.embed_tokens.weight"] = sd["shared.weight"]
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    assert m.lm_head.weight.data_ptr() != m.shared.weight.data_ptr()
    return m


def golden_relpos():
    from transformers.models.t5.modeling_t5 import T5Attention
    rp = torch.arange(-700, 701, dtype=torch.long)
    bi = T5Attention._relative_position_bucket(rp, bidirectional=True, num_buckets=32, max_distance=128)
    uni = T5Attention._relative_position_bucket(rp, bidirectional=False, num_buckets=32, max_distance=128)
    np.savez_compressed(os.path.join(GOLDEN, "relpos_buckets.npz"), relative_position=rp.numpy(),
                        bidirectional=bi.numpy().astype(np.int32), causal=uni.numpy().astype(np.int32))
    print("relpos_buckets.npz", bi[:5].tolist(), uni[695:705].tolist())


def golden_model(name: str, seed: int, n_img: int, B: int, S_e: int, T: int):
    cfg = get_config(name)
    w = make_seeded_weights(cfg, seed=seed, device="cpu", dtype=torch.bfloat16)
    g = torch.Generator().manual_seed(seed + 17)
    v, t = cfg.vision, cfg.t5
    pixels = torch.randn(n_img, 3, v.image, v.image, generator=g).to(torch.bfloat16).float()
    with torch.no_grad():
        vm = build_hf_vision(cfg, w)
        hs = vm(pixel_values=pixels, output_hidden_states=True).hidden_states
        assert len(hs) == v.layers + 1
        vit_hidden_m2 = hs[v.select_layer]                      # [n_img, 1+P, hidden]

        tm = build_hf_t5(cfg, w)
        emb = torch.randn(B, S_e, t.d_model, generator=g)
        lens = torch.randint(S_e // 2, S_e + 1, (B,), generator=g)
        lens[0] = S_e
        mask = torch.arange(S_e)[None, :] < lens[:, None]
        emb = emb * mask[..., None]
        labels = torch.randint(2, t.vocab, (B, T), generator=g)
        labels[:, -1] = t.eos_id
        if T > 2:                                              # ragged answers: -100 padded
            labels[1, -1] = -100
            labels[1, -2] = t.eos_id
        out = tm(inputs_embeds=emb, attention_mask=mask.long(), labels=labels)
        logits = out.logits
        ce = torch.nn.CrossEntropyLoss(reduction="mean")       # v3.0 scoring tail
        scores = torch.stack([(-ce(logits[k], labels[k])).exp() for k in range(B)])
        enc_out = out.encoder_last_hidden_state
    path = os.path.join(GOLDEN, f"hf_{name}.npz")
    np.savez_compressed(
        path, seed=seed, pixels=pixels.numpy(), vit_hidden_m2=vit_hidden_m2.numpy(), emb=emb.numpy(),
        mask=mask.numpy(), labels=labels.numpy(), enc_out=enc_out.numpy(), logits=logits.numpy(),
        scores=scores.numpy())
    print(path, os.path.getsize(path) // 1024, "KiB", "scores", scores.tolist())


def golden_e2e(name: str, seed: int, n_img: int, B: int, L: int, T: int, gain: float):
    """Full pipeline from HF modules: CLIPVisionModel -> hidden_states[-2][:,1:] -> mlp2x_gelu (torch Linear/GELU)
    -> splice -> T5ForConditionalGeneration(inputs_embeds, attention_mask, labels) -> per-sample exp(-CE).
    Stored twice: modules in fp32 ("truth": bf16-rounded weights, fp32 math) and modules cast to bf16 exactly as
    the reference does (mm_utils.py:228) and run on the CPU ("reference as shipped")."""
    cfg = get_config(name)
    w = make_seeded_weights(cfg, seed=seed, device="cpu", dtype=torch.bfloat16, lm_head_gain=gain)
    g = torch.Generator().manual_seed(seed + 31)
    v, t = cfg.vision, cfg.t5
    P = v.n_patches
    pixels = torch.randn(n_img, 3, v.image, v.image, generator=g).to(torch.bfloat16)
    ids = torch.randint(3, t.vocab, (B, L), generator=g)
    for b in range(B):
        n = L if b == 0 else int(torch.randint(max(3, L // 2), L + 1, (1,), generator=g))
        sp = int(torch.randint(0, n - 1, (1,), generator=g))
        ids[b, sp] = -200
        ids[b, n - 1] = t.eos_id
        ids[b, n:] = 0
    labels = torch.randint(3, t.vocab, (B, T), generator=g)
    labels[:, -1] = t.eos_id
    img_index = torch.randint(0, n_img, (B,), generator=g)
    S_e = L - 1 + P

    def run(dtype):
        with torch.no_grad():
            vm = build_hf_vision(cfg, w).to(dtype)
            tm = build_hf_t5(cfg, w).to(dtype)
            proj = torch.nn.Sequential(torch.nn.Linear(v.hidden, t.d_model), torch.nn.GELU(),
                                       torch.nn.Linear(t.d_model, t.d_model)).to(dtype)
            proj[0].weight.copy_(w["mm_projector.0.weight"]); proj[0].bias.copy_(w["mm_projector.0.bias"])
            proj[2].weight.copy_(w["mm_projector.2.weight"]); proj[2].bias.copy_(w["mm_projector.2.bias"])
            hs = vm(pixel_values=pixels.to(dtype), output_hidden_states=True).hidden_states
            feats = proj(hs[v.select_layer][:, 1:])
            emb = torch.zeros(B, S_e, t.d_model, dtype=dtype)
            mask = torch.zeros(B, S_e, dtype=torch.long)
            for b in range(B):
                row = ids[b][ids[b] != t.pad_id]
                sp = int((row == -200).nonzero()[0, 0])
                r = torch.cat([tm.shared(row[:sp]), feats[int(img_index[b])], tm.shared(row[sp + 1:])], 0)
                emb[b, : r.shape[0]] = r
                mask[b, : r.shape[0]] = 1
            out = tm(inputs_embeds=emb, attention_mask=mask, labels=labels)
            lp = torch.log_softmax(out.logits.float(), -1).gather(-1, labels[..., None])[..., 0]
            ce = torch.nn.CrossEntropyLoss(reduction="mean")
            sc = torch.stack([(-ce(out.logits[k].float(), labels[k])).exp() for k in range(B)])
        return lp, sc

    lp32, sc32 = run(torch.float32)
    lp16, sc16 = run(torch.bfloat16)
    path = os.path.join(GOLDEN, f"e2e_{name}_g{int(gain)}.npz")
    np.savez_compressed(path, seed=seed, gain=gain, pixels=pixels.float().numpy(), ids=ids.numpy(), labels=labels.numpy(),
                        img_index=img_index.numpy(), logprobs_fp32=lp32.numpy(), scores_fp32=sc32.numpy(),
                        logprobs_hf_bf16=lp16.numpy(), scores_hf_bf16=sc16.numpy())
    d = (lp16 - lp32).abs()
    print(path, os.path.getsize(path) // 1024, "KiB  HF-bf16 vs HF-fp32 |dlogP| max %.4g mean %.4g" % (d.max(), d.mean()),
          "logP range", float(lp32.min()), float(lp32.max()))


def golden_generate(name: str, seed: int, n_img: int, B: int, L: int, max_new: int, gain: float):
    """HF greedy search: T5ForConditionalGeneration.generate(inputs_embeds, attention_mask, max_new_tokens, do_sample=False)
    on the spliced embeddings (fp32 modules, bf16-rounded weights).  Stores the generated ids (HF pads finished rows with 0
    after EOS) and, from a teacher-forced re-run over those ids, the top-1/top-2 logit gap of every step."""
    cfg = get_config(name)
    w = make_seeded_weights(cfg, seed=seed, device="cpu", dtype=torch.bfloat16, lm_head_gain=gain)
    g = torch.Generator().manual_seed(seed + 77)
    v, t = cfg.vision, cfg.t5
    P = v.n_patches
    pixels = torch.randn(n_img, 3, v.image, v.image, generator=g).to(torch.bfloat16)
    ids = torch.randint(3, t.vocab, (B, L), generator=g)
    for b in range(B):
        n = L if b == 0 else int(torch.randint(max(3, L // 2), L + 1, (1,), generator=g))
        sp = int(torch.randint(0, n - 1, (1,), generator=g))
        ids[b, sp] = -200
        ids[b, n - 1] = t.eos_id
        ids[b, n:] = 0
    img_index = torch.randint(0, n_img, (B,), generator=g)
    S_e = L - 1 + P
    with torch.no_grad():
        vm = build_hf_vision(cfg, w).float()
        tm = build_hf_t5(cfg, w).float()
        proj = torch.nn.Sequential(torch.nn.Linear(v.hidden, t.d_model), torch.nn.GELU(), torch.nn.Linear(t.d_model, t.d_model))
        proj[0].weight.copy_(w["mm_projector.0.weight"]); proj[0].bias.copy_(w["mm_projector.0.bias"])
        proj[2].weight.copy_(w["mm_projector.2.weight"]); proj[2].bias.copy_(w["mm_projector.2.bias"])
        hs = vm(pixel_values=pixels.float(), output_hidden_states=True).hidden_states
        feats = proj(hs[v.select_layer][:, 1:])
        emb = torch.zeros(B, S_e, t.d_model)
        mask = torch.zeros(B, S_e, dtype=torch.long)
        for b in range(B):
            row = ids[b][ids[b] != t.pad_id]
            sp = int((row == -200).nonzero()[0, 0])
            r = torch.cat([tm.shared(row[:sp]), feats[int(img_index[b])], tm.shared(row[sp + 1:])], 0)
            emb[b, : r.shape[0]] = r
            mask[b, : r.shape[0]] = 1
        out = tm.generate(inputs_embeds=emb, attention_mask=mask, max_new_tokens=max_new, do_sample=False, num_beams=1)
        toks = out[:, 1:]                                    # drop the decoder start token
        if toks.shape[1] < max_new:                          # every row hit EOS early
            toks = torch.cat([toks, torch.zeros(B, max_new - toks.shape[1], dtype=toks.dtype)], 1)
        logits = tm(inputs_embeds=emb, attention_mask=mask, decoder_input_ids=out[:, :-1]).logits   # steps 0 .. n-1
        top2 = logits.topk(2, dim=-1).values
        margins = (top2[..., 0] - top2[..., 1])[:, :max_new]
        if margins.shape[1] < max_new:
            margins = torch.cat([margins, torch.zeros(B, max_new - margins.shape[1])], 1)
    path = os.path.join(GOLDEN, f"generate_{name}_g{int(gain)}.npz")
    np.savez_compressed(path, seed=seed, gain=gain, max_new=max_new, pixels=pixels.float().numpy(), ids=ids.numpy(),
                        img_index=img_index.numpy(), tokens=toks.numpy(), margins=margins.numpy())
    print(path, os.path.getsize(path) // 1024, "KiB tokens", toks.tolist(), "min margin %.3f" % float(margins.min()))


def build_hf_qwen(cfg, weights):
    """HF Qwen2_5_VLForConditionalGeneration at the dims of a Qwen25VLConfig, eager attention, fp32, seeded weights."""
    from transformers import Qwen2_5_VLConfig, Qwen2_5_VLForConditionalGeneration
    v, t = cfg.vision, cfg.text
    vis = dict(depth=v.depth, hidden_size=v.hidden, hidden_act="silu", intermediate_size=v.mlp, num_heads=v.heads,
               in_channels=v.in_channels, patch_size=v.patch, spatial_merge_size=v.spatial_merge,
               temporal_patch_size=v.temporal_patch, tokens_per_second=v.tokens_per_second, window_size=v.window,
               out_hidden_size=v.out_hidden, fullatt_block_indexes=list(v.fullatt_blocks))
    txt = dict(vocab_size=t.vocab, hidden_size=t.hidden, intermediate_size=t.mlp, num_hidden_layers=t.layers,
               num_attention_heads=t.heads, num_key_value_heads=t.kv_heads, hidden_act="silu", max_position_embeddings=32768,
               rms_norm_eps=t.rms_eps, tie_word_embeddings=False, use_sliding_window=False,
               rope_parameters={"rope_type": "default", "rope_theta": t.rope_theta, "mrope_section": list(t.mrope_section)})
    hc = Qwen2_5_VLConfig(text_config=txt, vision_config=vis, image_token_id=cfg.image_token_id, video_token_id=cfg.video_token_id,
                          vision_start_token_id=cfg.vision_start_token_id, vision_end_token_id=cfg.vision_end_token_id,
                          bos_token_id=None, eos_token_id=None, pad_token_id=None)
    hc._attn_implementation = "eager"
    m = Qwen2_5_VLForConditionalGeneration(hc).eval().float()
    missing, unexpected = m.load_state_dict({k: v.float() for k, v in weights.items()}, strict=True)
    return m


def golden_qwen(name: str, seed: int, grids, text_lens, gain: float, tag: str = ""):
    """One batch of video samples through HF Qwen2.5-VL (fp32 modules, bf16-rounded seeded weights): per sample a
    prompt [text | <vision_start> <video_pad>*n <vision_end> | text], run one by one exactly as the reference does
    (qwen2vl_model.py:190-230: batch 1, generate(max_new_tokens=1, output_scores=True)); stores the inputs, the merged
    vision tokens, the 3-D rope positions and the last-position logits."""
    from t2v_metrics_amd.qwen import get_qwen_config
    from t2v_metrics_amd.qwen.weights import make_seeded_qwen_weights
    cfg = get_qwen_config(name)
    w = make_seeded_qwen_weights(cfg, seed=seed, dtype=torch.bfloat16, lm_head_gain=gain)
    m = build_hf_qwen(cfg, w)
    g = torch.Generator().manual_seed(seed + 5)
    v = cfg.vision
    B = len(grids)
    pix, ids_rows, merged_all, pos_all, logits_all = [], [], [], [], []
    for b, ((t, h, wd), (n_pre, n_post)) in enumerate(zip(grids, text_lens)):
        n_patches = t * h * wd
        pv = torch.randn(n_patches, v.patch_dim, generator=g).to(torch.bfloat16).float()
        n_merged = n_patches // v.merge_unit
        pre = torch.randint(10, cfg.text.vocab, (n_pre,), generator=g)
        post = torch.randint(10, cfg.text.vocab, (n_post,), generator=g)
        ids = torch.cat([pre, torch.tensor([cfg.vision_start_token_id]), torch.full((n_merged,), cfg.video_token_id),
                         torch.tensor([cfg.vision_end_token_id]), post])[None]
        # mm_token_type_ids is what the HF processor hands over (0 text, 1 image, 2 video); without it the model falls
        # back to 1-D positions (compute_3d_position_ids :1135-1183) -- the reference passes the processor's dict
        mm_type = torch.where(ids == cfg.video_token_id, 2, 0)
        with torch.no_grad():
            out = m(input_ids=ids, attention_mask=torch.ones_like(ids), pixel_values_videos=pv, mm_token_type_ids=mm_type,
                    video_grid_thw=torch.tensor([[t, h, wd]]), output_hidden_states=False)
            m.model.rope_deltas = None
            gen = m.generate(input_ids=ids, attention_mask=torch.ones_like(ids), pixel_values_videos=pv, mm_token_type_ids=mm_type,
                             video_grid_thw=torch.tensor([[t, h, wd]]), max_new_tokens=1, do_sample=False,
                             output_scores=True, return_dict_in_generate=True)
            m.model.rope_deltas = None
            assert torch.allclose(gen.scores[0][0], out.logits[0, -1], atol=1e-4), "first generated scores != last prefill logits"
            merged = m.model.get_video_features(pv, torch.tensor([[t, h, wd]])).pooler_output
            merged = merged[0] if isinstance(merged, (list, tuple)) else merged
            pos, _ = m.model.get_rope_index(ids, mm_token_type_ids=mm_type, video_grid_thw=torch.tensor([[t, h, wd]]),
                                            attention_mask=torch.ones_like(ids))
        pix.append(pv); ids_rows.append(ids[0]); merged_all.append(merged.float()); pos_all.append(pos[:, 0]); logits_all.append(out.logits[0, -1].float())
    L = max(len(r) for r in ids_rows)
    ids_pad = torch.zeros(B, L, dtype=torch.long)
    mask = torch.zeros(B, L, dtype=torch.long)
    pos_pad = torch.zeros(3, B, L, dtype=torch.long)
    for b, r in enumerate(ids_rows):
        ids_pad[b, : len(r)] = r
        mask[b, : len(r)] = 1
        pos_pad[:, b, : len(r)] = pos_all[b]
    path = os.path.join(GOLDEN, f"qwen_{name.split('-')[-1]}{tag}.npz")
    np.savez_compressed(path, seed=seed, gain=gain, grids=np.asarray(grids), input_ids=ids_pad.numpy(), attention_mask=mask.numpy(),
                        pixel_values=torch.cat(pix).numpy(), merged=torch.cat(merged_all).numpy(), position_ids=pos_pad.numpy(),
                        logits=torch.stack(logits_all).numpy())
    print(path, os.path.getsize(path) // 1024, "KiB", "logits range", float(torch.stack(logits_all).min()), float(torch.stack(logits_all).max()))


def golden_qwen_generate(name: str, seed: int, grids, text_lens, gain: float, steps: int, tag: str):
    """HF generate WITH its KV cache, greedy, `steps` tokens per sample (batch 1 each, as the reference runs it,
    qwen2vl_model.py:222-230): stores the prompt, the pixels, the generated ids and the scores of every step.  Pins what
    vqs_qwen_prefill / vqs_qwen_decode must reproduce: the cached forward and the position of a generated token.  The rule the
    fixture pins (t2v_metrics_amd/qwen/layout.py, test_cached_generation_matches_hf_generate_fixture) is "last prompt position
    + 1 + step on every M-RoPE axis" -- NOT get_rope_index's "cache position + rope_delta" (= max prompt position + 1): the
    fixture includes a video whose temporal positions run past its spatial ones, where the two differ and the rope_delta rule
    misses HF generate's scores by 1.4."""
    from t2v_metrics_amd.qwen import get_qwen_config
    from t2v_metrics_amd.qwen.weights import make_seeded_qwen_weights
    cfg = get_qwen_config(name)
    w = make_seeded_qwen_weights(cfg, seed=seed, dtype=torch.bfloat16, lm_head_gain=gain)
    m = build_hf_qwen(cfg, w)
    g = torch.Generator().manual_seed(seed + 5)
    v = cfg.vision
    pix, ids_rows, gen_ids, gen_scores, pos_max = [], [], [], [], []
    for (t, h, wd), (n_pre, n_post) in zip(grids, text_lens):
        n_patches = t * h * wd
        pv = torch.randn(n_patches, v.patch_dim, generator=g).to(torch.bfloat16).float()
        n_merged = n_patches // v.merge_unit
        pre = torch.randint(10, cfg.text.vocab, (n_pre,), generator=g)
        post = torch.randint(10, cfg.text.vocab, (n_post,), generator=g)
        ids = torch.cat([pre, torch.tensor([cfg.vision_start_token_id]), torch.full((n_merged,), cfg.video_token_id),
                         torch.tensor([cfg.vision_end_token_id]), post])[None]
        mm_type = torch.where(ids == cfg.video_token_id, 2, 0)
        with torch.no_grad():
            m.model.rope_deltas = None
            gen = m.generate(input_ids=ids, attention_mask=torch.ones_like(ids), pixel_values_videos=pv, mm_token_type_ids=mm_type,
                             video_grid_thw=torch.tensor([[t, h, wd]]), max_new_tokens=steps, min_new_tokens=steps, do_sample=False,
                             output_scores=True, return_dict_in_generate=True, use_cache=True)
            m.model.rope_deltas = None
            pos, _ = m.model.get_rope_index(ids, mm_token_type_ids=mm_type, video_grid_thw=torch.tensor([[t, h, wd]]),
                                            attention_mask=torch.ones_like(ids))
        assert len(gen.scores) == steps
        pix.append(pv); ids_rows.append(ids[0]); gen_ids.append(gen.sequences[0, ids.shape[1]:])
        gen_scores.append(torch.stack([sc[0].float() for sc in gen.scores])); pos_max.append(int(pos.max()))
    B, L = len(grids), max(len(r) for r in ids_rows)
    ids_pad = torch.zeros(B, L, dtype=torch.long)
    mask = torch.zeros(B, L, dtype=torch.long)
    for b, r in enumerate(ids_rows):
        ids_pad[b, : len(r)] = r
        mask[b, : len(r)] = 1
    path = os.path.join(GOLDEN, f"qwen_{name.split('-')[-1]}{tag}.npz")
    np.savez_compressed(path, seed=seed, gain=gain, grids=np.asarray(grids), input_ids=ids_pad.numpy(), attention_mask=mask.numpy(),
                        pixel_values=torch.cat(pix).numpy(), gen_ids=torch.stack(gen_ids).numpy(), gen_scores=torch.stack(gen_scores).numpy(),
                        prompt_position_max=np.asarray(pos_max))
    print(path, os.path.getsize(path) // 1024, "KiB", "generated", torch.stack(gen_ids).tolist(), "max prompt positions", pos_max,
          "prompt lengths", [len(r) for r in ids_rows])


def golden_preprocess():
    """HF CLIPImageProcessor (PIL backend here: no torchvision) on seeded images at a small target size."""
    from PIL import Image
    from transformers import CLIPImageProcessor
    size = 56
    p = CLIPImageProcessor(size={"shortest_edge": size}, crop_size={"height": size, "width": size})
    rng = np.random.RandomState(7)
    shapes = [(70, 70), (90, 150), (200, 120), (56, 56), (40, 33)]
    out = {"size": size}
    for i, (h, w) in enumerate(shapes):
        arr = rng.randint(0, 256, (h, w, 3), dtype=np.uint8)
        out[f"img{i}"] = arr
        out[f"px{i}"] = p(images=Image.fromarray(arr), return_tensors="np")["pixel_values"][0].astype(np.float32)
    np.savez_compressed(os.path.join(GOLDEN, "clip_preprocess.npz"), **out)
    print("clip_preprocess.npz", os.path.getsize(os.path.join(GOLDEN, "clip_preprocess.npz")) // 1024, "KiB")


if __name__ == "__main__":
    os.makedirs(GOLDEN, exist_ok=True)
    torch.manual_seed(0)
    golden_relpos()
    golden_preprocess()
    golden_model("tiny", seed=3, n_img=2, B=3, S_e=24, T=3)
    golden_model("small", seed=5, n_img=1, B=2, S_e=40, T=2)
    golden_e2e("tiny", seed=21, n_img=3, B=8, L=12, T=2, gain=1.0)
    golden_e2e("tiny", seed=22, n_img=3, B=8, L=12, T=2, gain=4.0)
    golden_e2e("small", seed=23, n_img=2, B=6, L=20, T=2, gain=1.0)
    golden_e2e("small", seed=24, n_img=2, B=6, L=20, T=2, gain=4.0)
    golden_generate("tiny", seed=41, n_img=2, B=6, L=12, max_new=6, gain=8.0)
    golden_generate("small", seed=42, n_img=2, B=4, L=20, max_new=5, gain=8.0)
    golden_qwen("qwen-tiny", seed=51, grids=[(2, 8, 8), (1, 4, 12), (3, 12, 8)], text_lens=[(3, 4), (5, 2), (2, 6)], gain=4.0)
    golden_qwen("qwen-small", seed=52, grids=[(2, 8, 16), (2, 16, 8)], text_lens=[(4, 5), (3, 7)], gain=4.0)
    golden_qwen("qwen-tiny", seed=53, grids=[(2, 6, 10), (1, 10, 6), (2, 6, 10)], text_lens=[(3, 4), (5, 2), (2, 3)], gain=4.0, tag="_ragged")
    # generation with HF's KV cache; the second video is long and narrow: 8 temporal patches x 2 tokens/s run to position 14 while the
    # spatial axes advance the cursor by 2
    golden_qwen_generate("qwen-tiny", seed=54, grids=[(2, 8, 8), (8, 4, 4), (1, 4, 12)], text_lens=[(3, 4), (2, 3), (5, 2)], gain=4.0,
                         steps=4, tag="_gen4")
