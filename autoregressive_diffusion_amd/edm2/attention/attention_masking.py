"""make_train_mask / make_infer_mask with the reference's signatures (edm2/attention/attention_masking.py:27-90).
The tables come from the C-ABI host builder (bit-exact target); the returned object exposes `kv_num_blocks` and
`kv_indices` (int32, (B, H, rows[, cols])) like torch's BlockMask, plus `mask_mod` on token indices."""
from functools import lru_cache
import warnings
import torch

from ... import ops

SPARSE_BLOCK = 128


class TrainingMask:
    def __init__(self, n_frames, image_size):
        self.n_frames, self.image_size = int(n_frames), int(image_size)

    def __call__(self, b, h, q_idx, kv_idx):
        T = self.n_frames
        q, k = q_idx // self.image_size, kv_idx // self.image_size
        clean = (q < T) & (k <= q)
        noisy = (q >= T) & (((k < T) & (k < q - T)) | (k == q))
        return clean | noisy


class InferenceMask:
    def __init__(self, image_size):
        self.image_size = image_size

    def __call__(self, b, h, q_idx, kv_idx):
        return q_idx // self.image_size >= kv_idx // self.image_size


class BlockTable:
    """Minimal stand-in for torch.nn.attention.flex_attention.BlockMask (fields the reference's tests read)."""

    def __init__(self, num, idx, block, mask_mod, batch_size, num_heads, device):
        self.kv_num_blocks = torch.from_numpy(num).to(device)[None, None].expand(batch_size, num_heads, -1).contiguous()
        self.kv_indices = torch.from_numpy(idx).to(device)[None, None].expand(batch_size, num_heads, -1, -1).contiguous()
        self.BLOCK_SIZE = (block, block)
        self.mask_mod = mask_mod

    def to_dense(self):
        B, Hh, R = self.kv_num_blocks.shape
        Cc = self.kv_indices.shape[-1]
        dense = torch.zeros(B, Hh, R, Cc, dtype=torch.int32, device=self.kv_indices.device)
        for r in range(R):
            n = int(self.kv_num_blocks[0, 0, r])
            dense[:, :, r, self.kv_indices[0, 0, r, :n].long()] = 1
        return dense


def _device():
    return "cuda" if torch.cuda.is_available() else "cpu"


@lru_cache(maxsize=16)
def make_train_mask(batch_size, num_heads, n_frames, image_size):
    tab = ops.train_mask_table(n_frames, image_size)
    if tab is None:
        warnings.warn(f"The image size must be a divisor of the default block size ({SPARSE_BLOCK}), got "
                      f"image_size:{image_size} and n_frames:{n_frames}\n returning None")
        return None
    num, idx, blk = tab
    return BlockTable(num, idx, blk, TrainingMask(n_frames, image_size), batch_size, num_heads, _device())


@lru_cache(maxsize=16)
def make_infer_mask(batch_size, num_heads, n_frames, image_size):
    mask = InferenceMask(image_size)
    tab = ops.infer_mask_table(n_frames, image_size)
    if tab is None:
        if n_frames * image_size < SPARSE_BLOCK:
            def score_mod(score, b, h, q_idx, kv_idx):
                return torch.where(mask(b, h, q_idx, kv_idx), score, torch.full_like(score, -float("inf")))
            return score_mod, None
        return None, None      # dense fall-back of the reference: pure mask_mod, handled inside the kernel
    num, idx, blk = tab
    return None, BlockTable(num, idx, blk, mask, batch_size, num_heads, _device())
