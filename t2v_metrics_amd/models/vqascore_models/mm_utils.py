"""Prompt/image helpers of the CLIP-FlanT5 wrapper, restated from the helpers that survive in the reference
(/root/reference/t2v_metrics/models/vqascore_models/mm_utils.py:128-139 expand2square, :164-179
t5_tokenizer_image_token)."""
from typing import List, Sequence

import torch

from ...constants import IMAGE_TOKEN_INDEX, DEFAULT_IMAGE_TOKEN


from ..._imgprep import expand2square  # noqa: E402,F401  (mm_utils.py:128-139; defined in the torch-free module the image workers run)


def t5_tokenizer_image_token(prompt: str, tokenizer, image_token_index: int = IMAGE_TOKEN_INDEX,
                             return_tensors=None):
    """Tokenise every text chunk around ``<image>`` on its own (so each chunk carries the tokenizer's
    trailing </s>) and join the chunks with ONE sentinel id; T5 has no BOS to skip."""
    chunks: List[Sequence[int]] = [tokenizer(c).input_ids for c in prompt.split(DEFAULT_IMAGE_TOKEN)]
    input_ids: List[int] = []
    for i, ids in enumerate(chunks):
        if i > 0:
            input_ids.append(image_token_index)
        input_ids.extend(ids)
    if return_tensors is not None:
        if return_tensors == 'pt':
            return torch.tensor(input_ids, dtype=torch.long)
        raise ValueError(f'Unsupported tensor type: {return_tensors}')
    return input_ids
