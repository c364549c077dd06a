"""CPU oracle for the Qwen2.5-VL VQAScore path (BASELINE.json configs[4]).  TEST INFRASTRUCTURE
ONLY: imported by tests/ (and later smoke()/bench's cpu_baseline leg), never by the product path.

Plain fp32 torch restatement of what the reference executes for one sample
(/root/reference/t2v_metrics/models/vqascore_models/qwen2vl_model.py:222-301: ``model.generate(max_new_tokens=1,
do_sample=False, output_scores=True)`` -> fp32 logits of the first generated position -> softmax(logits / T)[answer id]),
i.e. ONE prefill through HF ``Qwen2_5_VLForConditionalGeneration`` (transformers 5.15.0 installed here; file:line below
refer to models/qwen2_5_vl/modeling_qwen2_5_vl.py unless another file is named).

Pinning: the reference holds no golden vector for this path and no checkpoint is reachable offline, so the oracle is
pinned against the HF modules themselves run in the build container on seeded weights
(oracle/make_golden.py::golden_qwen -> tests/golden/qwen_*.npz, tests/test_qwen_oracle.py).  Parity against a real
checkpoint remains unpinned.
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import torch
import torch.nn.functional as F


def rms_norm(x: torch.Tensor, w: torch.Tensor, eps: float) -> torch.Tensor:
    """Qwen2_5_VLRMSNorm :65-79 (fp32 statistics, weight applied after the normalisation)."""
    x = x.to(torch.float32)
    return w * (x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps))


def rotate_half(x: torch.Tensor) -> torch.Tensor:
    """:153-157"""
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


def vision_position_ids(grid_thw: Sequence[Sequence[int]], merge: int) -> torch.Tensor:
    """(h, w) patch coordinates, laid out block-major over merge x merge blocks, repeated per temporal patch
    (HF vision_utils.py:81-127) -> long [N, 2]."""
    out = []
    for t, h, w in grid_thw:
        hp, wp = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
        shape = (h // merge, merge, w // merge, merge)
        hp = hp.reshape(shape).transpose(1, 2).flatten()
        wp = wp.reshape(shape).transpose(1, 2).flatten()
        out.append(torch.stack([hp, wp], dim=-1).repeat(t, 1))
    return torch.cat(out, dim=0)


def vision_window_index(grid_thw: Sequence[Sequence[int]], merge: int, window: int, patch: int) -> Tuple[torch.Tensor, List[int]]:
    """Permutation of the merged cells that makes every attention window contiguous + cumulative window boundaries in
    PATCH units (HF vision_utils.py:130-188)."""
    index_all: List[torch.Tensor] = []
    cu = [0]
    base = 0
    ws = window // merge // patch
    unit = merge * merge
    for t, h, w in grid_thw:
        gh, gw = h // merge, w // merge
        index = torch.arange(t * gh * gw).reshape(t, gh, gw)
        pad_h, pad_w = ws - gh % ws, ws - gw % ws
        nh, nw = (gh + pad_h) // ws, (gw + pad_w) // ws
        padded = F.pad(index, (0, pad_w, 0, pad_h), "constant", -100)
        padded = padded.reshape(t, nh, ws, nw, ws).permute(0, 1, 3, 2, 4).reshape(t, nh * nw, ws, ws)
        seqlens = (padded != -100).sum([2, 3]).reshape(-1)
        flat = padded.reshape(-1)
        index_all.append(flat[flat != -100] + base)
        cu.extend((seqlens.cumsum(0) * unit + cu[-1]).tolist())
        base += t * gh * gw
    dedup = [cu[0]]
    for x in cu[1:]:
        if x != dedup[-1]:
            dedup.append(x)              # torch.unique_consecutive: empty (all-padding) windows vanish
    return torch.cat(index_all), dedup


def frame_seqlens(grid_thw: Sequence[Sequence[int]]) -> List[int]:
    """Full-attention segments: one per temporal patch, h*w patches each (HF vision_utils.py:42-65) -> boundaries."""
    cu = [0]
    for t, h, w in grid_thw:
        for _ in range(t):
            cu.append(cu[-1] + h * w)
    return cu


def mrope_position_ids(input_ids: torch.Tensor, attention_mask: torch.Tensor, image_id: int, video_id: int,
                       image_grids: Sequence[Sequence[int]], video_grids: Sequence[Sequence[int]], merge: int,
                       tokens_per_second: int, second_per_grid: float = 1.0) -> torch.Tensor:
    """3-D (t, h, w) rotary positions of every token (Qwen2_5_VLModel.get_rope_index :944-1061 with
    get_vision_position_ids :892-942): text runs advance all three axes together; a vision run of a (t, h, w) grid gets
    t*time_interval / row / column offsets from the run's start and advances the cursor by max(h, w) // merge."""
    B, L = input_ids.shape
    pos = torch.zeros(3, B, L, dtype=torch.long)
    it = {1: iter(image_grids), 2: iter(video_grids)}
    for b in range(B):
        valid = attention_mask[b].bool()
        ids = input_ids[b][valid]
        kind = torch.where(ids == image_id, 1, torch.where(ids == video_id, 2, 0)).tolist()
        cur, chunks, i = 0, [], 0
        while i < len(kind):
            j = i
            while j < len(kind) and kind[j] == kind[i]:
                j += 1
            if kind[i] == 0:
                chunks.append(torch.arange(j - i).view(1, -1).expand(3, -1) + cur)
                cur += j - i
            else:
                t, h, w = next(it[kind[i]])
                interval = tokens_per_second * int(second_per_grid) if kind[i] == 2 else 1
                gh, gw = h // merge, w // merge
                tt, hh, ww = torch.meshgrid(torch.arange(t) * interval, torch.arange(gh) + cur, torch.arange(gw) + cur, indexing="ij")
                vp = torch.stack([tt, hh, ww], dim=0).reshape(3, -1)
                vp[0] += cur
                chunks.append(vp)
                cur += max(h, w) // merge
            i = j
        pos[:, b, valid] = torch.cat(chunks, dim=1)
    return pos


class QwenOracle:
    def __init__(self, cfg, weights: Dict[str, torch.Tensor], device="cpu", hook=None):
        """device "cpu" = the oracle proper (fp32 copies of every weight held on the host).  Any other device: the SAME code evaluated in
        torch fp32 there (call it under ``with torch.device(dev)``), weights converted on use -- how tools/bench_qwen.py and
        tools/qwen_error_attribution.py get fp32 truth for several 7B-size samples in seconds.
        hook(cls, x, pos_dim): identity in the oracle (None).  The error-attribution tool passes a function that rounds the tensors of
        class `cls` the way the engine stores them (`pos_dim` = the dimension that indexes token positions, or None for vision rows);
        the classes name the engine's 16-bit storage points, they do not change what is computed."""
        self.cfg = cfg
        self.device = torch.device(device)
        self.hook = hook
        if self.device.type == "cpu":
            self.w = {k: v.detach().to(torch.float32).cpu() for k, v in weights.items()}
        else:
            self.w = weights

    def _w(self, name: str) -> torch.Tensor:
        w = self.w[name]
        return w if self.device.type == "cpu" else w.detach().to(self.device, torch.float32)

    def _r(self, cls: str, x: torch.Tensor, pos_dim=None) -> torch.Tensor:
        return x if self.hook is None else self.hook(cls, x, pos_dim)

    # ------------------------------------------------------------------------------------------ vision tower
    def _segment_attention(self, q, k, v, cu: Sequence[int], scale: float) -> torch.Tensor:
        """q,k,v [N, H, hd]; block-diagonal attention over segments cu[i]:cu[i+1] (VisionAttention.forward :268-289,
        eager_attention_forward :187-208 with fp32 softmax) -> [N, H*hd]."""
        out = torch.empty_like(q)
        for a, b in zip(cu[:-1], cu[1:]):
            s = torch.einsum("qhd,khd->hqk", q[a:b], k[a:b]) * scale
            p = self._r("vis.p", torch.softmax(s, dim=-1, dtype=torch.float32))
            out[a:b] = torch.einsum("hqk,khd->qhd", p, v[a:b])
        return out.reshape(q.shape[0], -1)

    def vision_tower(self, pixel_values: torch.Tensor, grid_thw: Sequence[Sequence[int]], return_stages: bool = False):
        """Qwen2_5_VisionTransformerPretrainedModel.forward :408-471 -> merged tokens [N/4, out_hidden] in the ORIGINAL
        (un-windowed) cell order."""
        v = self.cfg.vision
        # patch embed: Conv3d with stride = kernel == a matmul over the flattened receptive field (:116-122)
        h = self._r("vis.in", pixel_values.to(torch.float32)) @ self._w("model.visual.patch_embed.proj.weight").reshape(v.hidden, -1).t()
        N = h.shape[0]
        widx, cu_win = vision_window_index(grid_thw, v.spatial_merge, v.window, v.patch)
        cu_full = frame_seqlens(grid_thw)
        h = h.reshape(N // v.merge_unit, v.merge_unit, -1)[widx].reshape(N, -1)
        # 2-D rotary table: head_dim/2 frequencies, first half of them rotated by the row, second half by the column
        # (:125-134, :437-442)
        pid = vision_position_ids(grid_thw, v.spatial_merge)
        dim = v.head_dim // 2
        inv_freq = 1.0 / (10000.0 ** (torch.arange(0, dim, 2, dtype=torch.float32) / dim))
        rot = (pid.unsqueeze(-1).to(torch.float32) * inv_freq).flatten(1)                 # [N, head_dim/2]
        rot = rot.reshape(N // v.merge_unit, v.merge_unit, -1)[widx].reshape(N, -1)
        emb = torch.cat((rot, rot), dim=-1)
        cos, sin = emb.cos().unsqueeze(1), emb.sin().unsqueeze(1)                        # [N, 1, head_dim]
        stages = {}
        for i in range(v.depth):
            p = f"model.visual.blocks.{i}."
            r = self._r
            x = r(f"vis.norm.{i}", rms_norm(h, self._w(p + "norm1.weight"), v.rms_eps))
            qkv = r("vis.qkv", x @ self._w(p + "attn.qkv.weight").t() + self._w(p + "attn.qkv.bias"))
            q, k, val = qkv.reshape(N, 3, v.heads, v.head_dim).permute(1, 0, 2, 3).unbind(0)
            q = r("vis.rope", q * cos + rotate_half(q) * sin)                            # :160-172
            k = r("vis.rope", k * cos + rotate_half(k) * sin)
            a = r("vis.attn", self._segment_attention(q, k, val, cu_full if i in v.fullatt_blocks else cu_win, v.head_dim ** -0.5))
            h = h + r("vis.delta", a @ self._w(p + "attn.proj.weight").t() + self._w(p + "attn.proj.bias"))
            x = r(f"vis.norm.{i}", rms_norm(h, self._w(p + "norm2.weight"), v.rms_eps))
            g = F.silu(x @ self._w(p + "mlp.gate_proj.weight").t() + self._w(p + "mlp.gate_proj.bias"))
            u = x @ self._w(p + "mlp.up_proj.weight").t() + self._w(p + "mlp.up_proj.bias")
            h = h + r("vis.delta", r("vis.act", g * u) @ self._w(p + "mlp.down_proj.weight").t() + self._w(p + "mlp.down_proj.bias"))
            if return_stages:
                stages[f"vis_block{i}"] = h.clone()
        # merger (:137-150): RMSNorm, 4 neighbouring patches concatenated, Linear - GELU(erf) - Linear; then undo the
        # window permutation (:463-465)
        x = self._r("vis.norm.merger", rms_norm(h, self._w("model.visual.merger.ln_q.weight"), 1e-6)).reshape(N // v.merge_unit, -1)
        x = self._r("vis.mid", F.gelu(x @ self._w("model.visual.merger.mlp.0.weight").t() + self._w("model.visual.merger.mlp.0.bias")))
        x = self._r("vis.merged", x @ self._w("model.visual.merger.mlp.2.weight").t() + self._w("model.visual.merger.mlp.2.bias"))
        merged = x[torch.argsort(widx)]
        if return_stages:
            return merged, stages
        return merged

    # ------------------------------------------------------------------------------------------ language model
    def text_model(self, embeds: torch.Tensor, position_ids: torch.Tensor, attention_mask: torch.Tensor) -> torch.Tensor:
        """Qwen2_5_VLTextModel.forward :790-873 (prefill, causal + key padding mask, no cache) -> final-norm hidden."""
        t = self.cfg.text
        B, L, _ = embeds.shape
        hd = t.head_dim
        inv_freq = 1.0 / (t.rope_theta ** (torch.arange(0, hd, 2, dtype=torch.float32) / hd))
        freqs = position_ids[..., None].to(torch.float32) * inv_freq                     # [3, B, L, hd/2]   (:525-538)
        emb = torch.cat((freqs, freqs), dim=-1)
        cos3, sin3 = emb.cos(), emb.sin()
        sec = list(t.mrope_section) * 2                                                    # (:557-599)
        cos = torch.cat([m[i % 3] for i, m in enumerate(cos3.split(sec, dim=-1))], dim=-1).unsqueeze(1)
        sin = torch.cat([m[i % 3] for i, m in enumerate(sin3.split(sec, dim=-1))], dim=-1).unsqueeze(1)
        neg = torch.finfo(torch.float32).min
        causal = torch.tril(torch.ones(L, L, dtype=torch.bool))
        allow = causal[None, None] & attention_mask.bool()[:, None, None, :]
        add = torch.where(allow, 0.0, neg)
        rep = t.heads // t.kv_heads
        h = embeds.to(torch.float32)
        for i in range(t.layers):
            p = f"model.language_model.layers.{i}."
            r = self._r
            x = r("txt.norm", rms_norm(h, self._w(p + "input_layernorm.weight"), t.rms_eps), 1)
            q = r("txt.qkv", x @ self._w(p + "self_attn.q_proj.weight").t() + self._w(p + "self_attn.q_proj.bias"), 1).view(B, L, t.heads, hd).transpose(1, 2)
            k = r("txt.qkv", x @ self._w(p + "self_attn.k_proj.weight").t() + self._w(p + "self_attn.k_proj.bias"), 1).view(B, L, t.kv_heads, hd).transpose(1, 2)
            v = r("txt.qkv", x @ self._w(p + "self_attn.v_proj.weight").t() + self._w(p + "self_attn.v_proj.bias"), 1).view(B, L, t.kv_heads, hd).transpose(1, 2)
            q = r("txt.rope", q * cos + rotate_half(q) * sin, 2)
            k = r("txt.rope", k * cos + rotate_half(k) * sin, 2)
            k = k.repeat_interleave(rep, dim=1)                                           # repeat_kv :175-184
            v = v.repeat_interleave(rep, dim=1)
            s = q @ k.transpose(2, 3) * hd ** -0.5 + add
            a = r("txt.attn", r("txt.p", torch.softmax(s, dim=-1, dtype=torch.float32), 2) @ v, 2)
            del s
            h = h + r("txt.delta", a.transpose(1, 2).reshape(B, L, -1) @ self._w(p + "self_attn.o_proj.weight").t(), 1)
            x = r("txt.norm", rms_norm(h, self._w(p + "post_attention_layernorm.weight"), t.rms_eps), 1)
            g = F.silu(x @ self._w(p + "mlp.gate_proj.weight").t())
            h = h + r("txt.delta", r("txt.act", g * (x @ self._w(p + "mlp.up_proj.weight").t()), 1) @ self._w(p + "mlp.down_proj.weight").t(), 1)
        return self._r("txt.out", rms_norm(h, self._w("model.language_model.norm.weight"), t.rms_eps), 1)

    # ------------------------------------------------------------------------------------------ whole pass
    def forward(self, input_ids: torch.Tensor, attention_mask: torch.Tensor, pixel_values_videos: torch.Tensor,
                video_grid_thw: Sequence[Sequence[int]], return_stages: bool = False):
        """input_ids [B, L] (right-padded, attention_mask 0 on padding) containing video placeholder runs;
        pixel_values_videos [sum N_i, patch_dim] in sample order -> fp32 logits [B, vocab] of the LAST VALID position
        (= the scores of the first generated token, HF generation/utils.py greedy branch)."""
        c = self.cfg
        with torch.no_grad():
            merged = self.vision_tower(pixel_values_videos, video_grid_thw)
            emb = self._w("model.language_model.embed_tokens.weight")[input_ids.clamp(min=0)]
            mask = input_ids == c.video_token_id
            assert int(mask.sum()) == merged.shape[0], "video placeholder count != merged vision tokens (:1094-1133)"
            emb = emb.masked_scatter(mask[..., None].expand_as(emb), merged.to(emb.dtype))  # :1226-1232
            pos = mrope_position_ids(input_ids, attention_mask, c.image_token_id, c.video_token_id, [], video_grid_thw,
                                     c.vision.spatial_merge, c.vision.tokens_per_second)
            hid = self.text_model(emb, pos, attention_mask)
            last = attention_mask.long().sum(-1) - 1
            h_last = hid[torch.arange(hid.shape[0]), last]
            logits = h_last @ self._w("lm_head.weight").t()
        if return_stages:
            return {"logits": logits, "merged": merged, "position_ids": pos, "hidden": hid}
        return logits

    @staticmethod
    def answer_prob(logits: torch.Tensor, answer_id: int, temperature: float = 1.0) -> torch.Tensor:
        """qwen2vl_model.py:160-167,268-274: softmax(logits / T)[answer id] in fp32."""
        return torch.softmax(logits.to(torch.float32) / temperature, dim=-1)[..., answer_id]
