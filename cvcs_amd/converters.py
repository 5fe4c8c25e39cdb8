"""GID-15 label <-> colour table (source/scripts/converters.py:5-36; pinned by tests/golden/converter_ref.npz)."""
import torch

PALETTE = (
    (0, 0, 0), (200, 0, 0), (250, 0, 150), (200, 150, 150), (250, 150, 150), (0, 200, 0), (150, 250, 0), (150, 200, 150),
    (200, 0, 200), (150, 0, 250), (150, 150, 250), (250, 200, 0), (200, 200, 0), (0, 0, 200), (0, 150, 200), (0, 200, 250),
)


class GID15Converter:
    def __init__(self):
        self.color_to_label = {c: i for i, c in enumerate(PALETTE)}

    def palette_u8(self, device="cpu"):
        return torch.tensor(PALETTE, dtype=torch.uint8, device=device)

    def iconvert(self, mask):
        """class-label mask [H,W] -> float RGB [H,W,3] in 0..1; labels outside the table stay white (converters.py:32)"""
        out = torch.ones(*mask.shape, 3, dtype=torch.float32)
        pal = torch.tensor(PALETTE, dtype=torch.float32) / 255
        m = mask.long().cpu()
        ok = (m >= 0) & (m < len(PALETTE))
        out[ok] = pal[m[ok]]
        return out
