"""Integer half of the image input pipeline -- decode, expand2square, PIL bicubic resize, centre crop -> uint8 [S, S, 3] --
in a module that imports NOTHING but numpy and PIL and no sibling module, so that it can also run as a stand-alone worker
process (``python _imgprep.py --worker``; t2v_metrics_amd/imgpool.py).

Why worker PROCESSES: PIL decodes under the GIL, so the thread pool of rounds 1-3 gave no parallel speed-up on the decode
(2.5 s of host work per 256 PNG files however many threads, profiles/r3_call32_*) -- fine while one GPU had all 256 host threads
of the box, not with the 32 threads a rank gets at 8 GPUs per node.

Restates:
  * image_loader      /root/reference/t2v_metrics/models/model.py:10-14 (.npy = OpenCV-style BGR arrays, else PIL -> RGB)
  * expand2square     /root/reference/t2v_metrics/models/vqascore_models/mm_utils.py:128-139
  * the resize + crop of HF CLIPImageProcessor (models/clip/image_processing_clip.py:22-33, PIL backend): bicubic resize of the
    shortest edge to `size`, centre crop; pinned to the HF processor by tests/golden/clip_preprocess.npz.
The float half (x 1/255, normalise, bf16) runs on the GPU (vqs_normalize_u8).
"""
from __future__ import annotations

import os

import numpy as np
from PIL import Image

OPENAI_CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
OPENAI_CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def image_loader(image_path) -> Image.Image:
    """Decode one image path to an RGB PIL image.  ``.npy`` files hold OpenCV-style BGR arrays [H, W, 3] and are
    channel-flipped (reference model.py:10-14); every other suffix is handed to PIL."""
    if os.path.splitext(str(image_path))[1].lower() == '.npy':
        bgr = np.load(image_path)
        return Image.fromarray(np.ascontiguousarray(bgr[..., ::-1]), 'RGB')
    with Image.open(image_path) as im:
        return im.convert("RGB")


def expand2square(pil_img: Image.Image, background_color) -> Image.Image:
    """Pad the shorter side symmetrically (floor on the leading side) so the image becomes square."""
    w, h = pil_img.size
    if w == h:
        return pil_img
    side = max(w, h)
    canvas = Image.new(pil_img.mode, (side, side), background_color)
    canvas.paste(pil_img, ((side - w) // 2, (side - h) // 2))
    return canvas


def _resize_shortest_edge(img: Image.Image, size: int) -> Image.Image:
    w, h = img.size
    short, long = (w, h) if w <= h else (h, w)
    if short == size:
        return img
    new_short, new_long = size, int(size * long / short)
    nw, nh = (new_short, new_long) if w <= h else (new_long, new_short)
    return img.resize((nw, nh), resample=Image.BICUBIC)


def _center_crop(arr: np.ndarray, size: int) -> np.ndarray:
    """arr [H,W,C]; crops (or zero-pads, as HF does) to size x size around the centre."""
    h, w = arr.shape[:2]
    top = (h - size) // 2
    left = (w - size) // 2
    if top >= 0 and left >= 0:
        return arr[top: top + size, left: left + size]
    out = np.zeros((size, size, arr.shape[2]), dtype=arr.dtype)
    nh, nw = max(size, h), max(size, w)
    padded = np.zeros((nh, nw, arr.shape[2]), dtype=arr.dtype)
    pt, pl = int(np.ceil((nh - h) / 2)), int(np.ceil((nw - w) / 2))
    padded[pt: pt + h, pl: pl + w] = arr
    top, left = (nh - size) // 2, (nw - size) // 2
    out[:] = padded[top: top + size, left: left + size]
    return out


def clip_preprocess_u8(img: Image.Image, image_size: int = 336, pad_to_square: bool = True) -> np.ndarray:
    """The integer part of the preprocessing (pad, PIL bicubic resize, centre crop) -> uint8 [image_size, image_size, 3];
    the float part (rescale + normalise + bf16) runs on the GPU (vqs_normalize_u8) with the same fp32 arithmetic."""
    img = img.convert("RGB")
    if pad_to_square:
        img = expand2square(img, tuple(int(x * 255) for x in OPENAI_CLIP_MEAN))
    img = _resize_shortest_edge(img, image_size)
    return np.ascontiguousarray(_center_crop(np.asarray(img), image_size))


# ----------------------------------------------------------------------------------------------------------------------
# Worker process.  Protocol (one JSON object per line on stdin / stdout; imgpool.ImageProcessPool is the other end):
#   request  {"slot": i, "path": p, "size": S, "pad": bool, "shm": file under /dev/shm (or any tmpfs), "n": slots in it}
#            the result goes to bytes [i * S*S*3, (i+1) * S*S*3) of that file (a uint8 [n, S, S, 3] array, memory-mapped)
#   reply    {"slot": i, "ok": true}  or  {"slot": i, "ok": false, "etype": "FileNotFoundError", "errno": 2, "err": "..."}
# The worker keeps the last mapping open; it exits when stdin closes.
# ----------------------------------------------------------------------------------------------------------------------
def _worker_main():
    import json
    import sys
    mapped_key, mapped = None, None
    out = sys.stdout
    for line in sys.stdin:
        line = line.strip()
        if not line:
            continue
        slot = -1
        try:
            req = json.loads(line)
            slot, S = int(req["slot"]), int(req["size"])
            key = (req["shm"], int(req["n"]), S)
            if key != mapped_key:
                mapped = np.memmap(req["shm"], dtype=np.uint8, mode="r+", shape=(key[1], S, S, 3))
                mapped_key = key
            mapped[slot] = clip_preprocess_u8(image_loader(req["path"]), S, bool(req["pad"]))
            out.write(json.dumps({"slot": slot, "ok": True}) + "\n")
        except Exception as e:                                   # noqa: BLE001 -- reported to the parent, which raises
            out.write(json.dumps({"slot": slot, "ok": False, "etype": type(e).__name__, "errno": getattr(e, "errno", None),
                                  "err": f"{type(e).__name__}: {e}"[:500]}) + "\n")
        out.flush()


if __name__ == "__main__":
    _worker_main()
