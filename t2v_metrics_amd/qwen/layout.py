"""Integer layout work of the Qwen2.5-VL path that stays on the host (as in the reference, where the HF processor and
``get_rope_index`` run on the CPU): window permutation of the vision tower, rotary tables, placeholder slots.
Implements what HF computes in vision_utils.py:81-188 and modeling_qwen2_5_vl.py:892-1061 for the input class the HIP
path supports: every video of one vision call has the same (t, h, w) grid (the wrapper groups samples by grid); partial
attention windows at the right / bottom edge of a frame are handled by a padded windowed layout."""
from __future__ import annotations

from typing import Dict, Sequence, Tuple

import torch

from .config import Qwen25VLConfig


def window_slots(t: int, h: int, w: int, merge: int, window: int, patch: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """Windowed, padded cell layout of ONE video: (slots long [n_windows * ws*ws] = original cell index or -1,
    valid_cells long [n_windows]).  Windows ordered (t, window row, window column), the cells of a window row-major with
    the present ones first -- the order HF's get_vision_window_index produces (vision_utils.py:130-188) once its padding
    entries are dropped; partial windows at the right / bottom edge keep their empty slots here."""
    gh, gw, ws = h // merge, w // merge, window // merge // patch
    nh, nw = -(-gh // ws), -(-gw // ws)
    index = torch.arange(t * gh * gw).reshape(t, gh, gw)
    padded = torch.nn.functional.pad(index, (0, nw * ws - gw, 0, nh * ws - gh), value=-1)
    win = padded.reshape(t, nh, ws, nw, ws).permute(0, 1, 3, 2, 4).reshape(t * nh * nw, ws * ws)
    valid = (win >= 0).sum(-1)
    # present cells first inside every window (stable: keeps row-major order)
    order = torch.argsort((win < 0).to(torch.int8), dim=-1, stable=True)
    win = torch.gather(win, 1, order)
    keep = valid > 0
    return win[keep].reshape(-1), valid[keep]


def vision_layout(cfg: Qwen25VLConfig, grids: Sequence[Tuple[int, int, int]]) -> Dict[str, torch.Tensor]:
    """Arrays vqs_qwen_encode_vision needs for the videos of one call (same grid each); see include/vqs_qwen.h."""
    v = cfg.vision
    if len(set(tuple(g) for g in grids)) != 1:
        raise ValueError("all videos of a call must share one (t, h, w) grid")
    t, h, w = grids[0]
    unit = v.merge_unit
    cells_per = t * (h // v.spatial_merge) * (w // v.spatial_merge)
    slots1, valid1 = window_slots(t, h, w, v.spatial_merge, v.window, v.patch)
    ws = v.window // v.spatial_merge // v.patch
    slots = torch.cat([torch.where(slots1 >= 0, slots1 + i * cells_per, slots1) for i in range(len(grids))])   # [n_win * ws*ws]
    valid = valid1.repeat(len(grids))
    row_map = torch.where(slots[:, None] >= 0, slots[:, None] * unit + torch.arange(unit)[None, :], torch.full((1, unit), -1)).reshape(-1)
    N = cells_per * len(grids) * unit
    Np = row_map.numel()
    real = row_map >= 0
    inv_row = torch.empty(N, dtype=torch.long)
    inv_row[row_map[real]] = torch.nonzero(real)[:, 0]
    cell_inv = torch.empty(N // unit, dtype=torch.long)
    cell_real = slots >= 0
    cell_inv[slots[cell_real]] = torch.nonzero(cell_real)[:, 0]
    # 2-D rotary angles per ORIGINAL patch row (block-major over merge x merge, repeated per temporal patch)
    hp, wp = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
    shape = (h // v.spatial_merge, v.spatial_merge, w // v.spatial_merge, v.spatial_merge)
    hp = hp.reshape(shape).transpose(1, 2).flatten().repeat(t)
    wp = wp.reshape(shape).transpose(1, 2).flatten().repeat(t)
    dim = v.head_dim // 2
    inv_freq = 1.0 / (10000.0 ** (torch.arange(0, dim, 2, dtype=torch.float32) / dim))
    ang_f = torch.cat([hp[:, None].float() * inv_freq, wp[:, None].float() * inv_freq], dim=1).repeat(len(grids), 1)   # [N, hd/2]
    ang_w = torch.where(real[:, None], ang_f[row_map.clamp(min=0)], torch.zeros(1, ang_f.shape[1]))
    return {"row_map": row_map.to(torch.int32), "inv_row": inv_row.to(torch.int32), "win_valid": (valid * unit).to(torch.int32),
            "cell_inv": cell_inv.to(torch.int32), "cos_w": ang_w.cos().contiguous(), "sin_w": ang_w.sin().contiguous(),
            "cos_f": ang_f.cos().contiguous(), "sin_f": ang_f.sin().contiguous(), "N": N, "Np": Np,
            "win_len": ws * ws * unit, "frame_len": h * w}


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
    # position of the first generated token: HF generate advances the prompt's LAST position by one on each axis per new token
    # (GenerationMixin updates position_ids incrementally; pinned by tests/golden/qwen_tiny_gen4.npz, whose long narrow video
    # separates this rule from "max prompt position + 1", the rope_deltas formula the cached forward would use on its own)
    next_pos = torch.stack([pos[:, b, int(seq_len[b]) - 1] + 1 for b in range(B)], dim=1)          # [3, B]
    return {"vis_slot": vis_slot.to(torch.int32), "seq_len": seq_len.to(torch.int32), "last_row": last_row.to(torch.int32),
            "cos": ang.cos().contiguous(), "sin": ang.sin().contiguous(), "position_ids": pos, "next_pos": next_pos}


def decode_tables(cfg: Qwen25VLConfig, positions: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """positions long [3, B] (t, h, w axes of the new token; equal when the prompt ends in text) -> cos, sin fp32 [B, head_dim/2],
    M-RoPE sections applied as in text_layout."""
    hd = cfg.text.head_dim
    inv_freq = 1.0 / (cfg.text.rope_theta ** (torch.arange(0, hd, 2, dtype=torch.float32, device=positions.device) / hd))
    freqs = positions[..., None].float() * inv_freq                                  # [3, B, hd/2]; on the positions' device
    ang = torch.cat([m[i % 3] for i, m in enumerate(freqs.split(list(cfg.text.mrope_section), dim=-1))], dim=-1)
    return ang.cos().contiguous(), ang.sin().contiguous()
