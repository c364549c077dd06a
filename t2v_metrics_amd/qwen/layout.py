"""Integer layout work of the Qwen2.5-VL path that stays on the host (as in the reference, where the HF processor and
``get_rope_index`` run on the CPU): window permutation of the vision tower, rotary tables, placeholder slots.
Implements what HF computes in vision_utils.py:81-188 and modeling_qwen2_5_vl.py:892-1061 for the input class the HIP
path supports: every video of a call has the same (t, h, w) grid and its merged grid is a multiple of the window
(every attention window full -- true for BASELINE.json configs[4]: 336 x 448 frames)."""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import torch

from .config import Qwen25VLConfig


def window_cells(t: int, h: int, w: int, merge: int, window: int, patch: int) -> torch.Tensor:
    """Merged-cell order that makes windows contiguous: long [t * (h/merge) * (w/merge)]."""
    gh, gw, ws = h // merge, w // merge, window // merge // patch
    if gh % ws or gw % ws:
        raise ValueError(f"grid {gh}x{gw} merged cells is not a multiple of the {ws}x{ws}-cell attention window")
    idx = torch.arange(t * gh * gw).reshape(t, gh // ws, ws, gw // ws, ws)
    return idx.permute(0, 1, 3, 2, 4).reshape(-1)


def vision_layout(cfg: Qwen25VLConfig, grids: Sequence[Tuple[int, int, int]]) -> Dict[str, torch.Tensor]:
    """For the videos of one call (same grid each): row_map int32 [N] (windowed position -> source patch row),
    cell_inv int32 [N/4] (original cell -> windowed cell), cos/sin fp32 [N, head_dim/2] in windowed order,
    win_len, frame_len."""
    v = cfg.vision
    if len(set(tuple(g) for g in grids)) != 1:
        raise ValueError("all videos of a call must share one (t, h, w) grid")
    t, h, w = grids[0]
    unit = v.merge_unit
    cells_per = t * (h // v.spatial_merge) * (w // v.spatial_merge)
    wc = window_cells(t, h, w, v.spatial_merge, v.window, v.patch)
    cell_order = torch.cat([wc + i * cells_per for i in range(len(grids))])          # windowed cell -> original cell
    row_map = (cell_order[:, None] * unit + torch.arange(unit)[None, :]).reshape(-1)
    cell_inv = torch.argsort(cell_order)
    # 2-D rotary angles per ORIGINAL patch row (block-major over merge x merge, repeated per temporal patch)
    hp, wp = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
    shape = (h // v.spatial_merge, v.spatial_merge, w // v.spatial_merge, v.spatial_merge)
    hp = hp.reshape(shape).transpose(1, 2).flatten().repeat(t)
    wp = wp.reshape(shape).transpose(1, 2).flatten().repeat(t)
    dim = v.head_dim // 2
    inv_freq = 1.0 / (10000.0 ** (torch.arange(0, dim, 2, dtype=torch.float32) / dim))
    ang = torch.cat([hp[:, None].float() * inv_freq, wp[:, None].float() * inv_freq], dim=1)   # [t*h*w, head_dim/2]
    ang = ang.repeat(len(grids), 1)[row_map]
    ws = v.window // v.spatial_merge // v.patch
    return {"row_map": row_map.to(torch.int32), "cell_inv": cell_inv.to(torch.int32), "cos": ang.cos().contiguous(),
            "sin": ang.sin().contiguous(), "win_len": ws * ws * unit, "frame_len": h * w}


def text_layout(cfg: Qwen25VLConfig, input_ids: torch.Tensor, attention_mask: torch.Tensor,
                grids: Sequence[Tuple[int, int, int]]) -> Dict[str, torch.Tensor]:
    """input_ids [B, L] right-padded with one video placeholder run per sample (sample b uses grids[b]) ->
    vis_slot int32 [B, L], seq_len int32 [B], last_row int32 [B], cos/sin fp32 [B*L, head_dim/2] (M-RoPE, sections applied)."""
    t_, v = cfg.text, cfg.vision
    B, L = input_ids.shape
    is_vid = input_ids == cfg.video_token_id
    slot = torch.cumsum(is_vid.reshape(-1).long(), 0) - 1
    vis_slot = torch.where(is_vid.reshape(-1), slot, torch.full_like(slot, -1)).reshape(B, L)
    seq_len = attention_mask.long().sum(-1)
    pos = torch.zeros(3, B, L, dtype=torch.long)
    for b in range(B):
        n = int(seq_len[b])
        kinds = is_vid[b, :n].tolist()
        cur, i, chunks = 0, 0, []
        while i < n:
            j = i
            while j < n and kinds[j] == kinds[i]:
                j += 1
            if not kinds[i]:
                chunks.append(torch.arange(j - i).view(1, -1).expand(3, -1) + cur)
                cur += j - i
            else:
                t, h, w = grids[b]
                gh, gw = h // v.spatial_merge, w // v.spatial_merge
                if j - i != t * gh * gw:
                    raise ValueError("placeholder run length does not match the video grid")
                tt, hh, ww = torch.meshgrid(torch.arange(t) * v.tokens_per_second, torch.arange(gh) + cur, torch.arange(gw) + cur,
                                            indexing="ij")
                vp = torch.stack([tt, hh, ww], 0).reshape(3, -1)
                vp[0] += cur
                chunks.append(vp)
                cur += max(h, w) // v.spatial_merge
            i = j
        pos[:, b, :n] = torch.cat(chunks, 1)
    hd = t_.head_dim
    inv_freq = 1.0 / (t_.rope_theta ** (torch.arange(0, hd, 2, dtype=torch.float32) / hd))
    freqs = pos[..., None].float() * inv_freq                                      # [3, B, L, hd/2]
    sec = list(t_.mrope_section)
    ang = torch.cat([m[i % 3] for i, m in enumerate(freqs.split(sec, dim=-1))], dim=-1).reshape(B * L, hd // 2)
    last_row = torch.arange(B) * L + seq_len - 1
    return {"vis_slot": vis_slot.to(torch.int32), "seq_len": seq_len.to(torch.int32), "last_row": last_row.to(torch.int32),
            "cos": ang.cos().contiguous(), "sin": ang.sin().contiguous(), "position_ids": pos}
