"""Systematic resampling on the GPU (mirror of ``sample_cat_sys``,
pita/src/models/components/utils.py:111-120)."""
import torch

from . import _lib


def sample_cat_sys(bs, logits, u=None):
    """ids[bs] (int64 device tensor) for systematic resampling of ``softmax(logits)`` clipped to
    [1e-6, 1] (not renormalised).  ``u`` defaults to one float64 uniform from torch's CPU
    generator, exactly like the reference.  Returns ``(ids, None)`` like the reference."""
    logits = _lib.dev_tensor(logits, "logits").reshape(-1)
    assert logits.shape[0] == bs
    if u is None:
        u = torch.rand(size=(1,), dtype=torch.float64)
    u0 = float(u.reshape(-1)[0]) if isinstance(u, torch.Tensor) else float(u)
    ids = torch.empty(bs, device=logits.device, dtype=torch.int64)
    nbytes = int(_lib.lib().pita_resample_workspace_bytes(bs))
    ws = torch.empty((nbytes + 7) // 8, device=logits.device, dtype=torch.float64)  # 8-byte aligned scratch
    _lib.check(_lib.lib().pita_systematic_resample(logits.data_ptr(), bs, u0, ids.data_ptr(), ws.data_ptr(),
                                                   _lib.stream_ptr(logits.device)), "pita_systematic_resample")
    return ids, None


def gather_rows(x, ids):
    """x[ids] through the HIP row-gather kernel."""
    x = _lib.dev_tensor(x, "x")
    out = torch.empty_like(x)
    _lib.check(_lib.lib().pita_gather_rows(x.data_ptr(), ids.data_ptr(), out.data_ptr(), x.shape[0], x.shape[1],
                                           _lib.stream_ptr(x.device)), "pita_gather_rows")
    return out
