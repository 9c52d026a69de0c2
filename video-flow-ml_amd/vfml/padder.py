"""InputPadder — stands in for `utils.utils.InputPadder` of the VideoFlow submodule
(reference processing/videoflow_core.py:29; used at :182-183 `.pad(x)` on the 5-D batch and
:191 `.unpad(flow)`).  RAFT semantics: pad H and W up to the next multiple of 8, 'sintel' mode
splits the padding evenly (extra pixel bottom/right), replicate border.  A no-op at every
resolution whose sides are already multiples of 8 (1080p, 4K tiles, 720p, 256^2).
"""
import torch.nn.functional as F


class InputPadder:
    def __init__(self, dims, mode="sintel"):
        self.ht, self.wd = int(dims[-2]), int(dims[-1])
        ph = -self.ht % 8
        pw = -self.wd % 8
        if mode == "sintel":
            self._pad = (pw // 2, pw - pw // 2, ph // 2, ph - ph // 2)
        else:
            self._pad = (pw // 2, pw - pw // 2, 0, ph)

    def pad(self, x):
        if self._pad == (0, 0, 0, 0):
            return x
        # torch's replicate pad takes <= 3 leading dims: fold [B,T] before, unfold after.
        lead = x.shape[:-3]
        y = F.pad(x.reshape((-1,) + tuple(x.shape[-3:])), self._pad, mode="replicate")
        return y.reshape(tuple(lead) + tuple(y.shape[-3:]))

    def unpad(self, x):
        l, r, t, b = self._pad
        H, W = x.shape[-2:]
        return x[..., t:H - b, l:W - r]
