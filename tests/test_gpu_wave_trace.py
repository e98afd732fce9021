"""gsr_debug_wave_trace (measurement API): while a buffer is registered every wave of the default blend kernels leaves its start / end
time and its list length; nothing is written once it is switched off."""
import numpy as np
import pytest
import torch

from tests import util

pytestmark = pytest.mark.gpu


def test_wave_trace_records_every_wave_and_stops_when_switched_off(oracle):
    from mygauhuman_amd import _lib
    P, W, H = 4000, 160, 128
    tiles = ((W + 15) // 16) * ((H + 15) // 16)
    cam, g = util.make_scene(P, W, H, 3, 2, 0.03, 0.0)
    bg = np.zeros(3, np.float32)
    dc = np.ones((3, H, W), np.float32)
    dz = np.zeros((1, H, W), np.float32)
    words = 32 * (4 * tiles + 64)           # (include/gsr.h: always enough)
    buf = torch.zeros(words, dtype=torch.int64, device="cuda")
    _lib.check(_lib.lib.gsr_debug_wave_trace(buf.data_ptr(), words), "gsr_debug_wave_trace")
    try:
        f = util.hip_forward(cam, g, bg, "sh", debug=True)
        util.hip_backward(f, dc, dz, dz, debug=True)
        torch.cuda.synchronize()
    finally:
        _lib.check(_lib.lib.gsr_debug_wave_trace(None, 0), "gsr_debug_wave_trace")
    rec = buf.cpu().numpy().view(np.uint64).reshape(2, -1, 4)
    ranges = util.hip_query(f, "RANGES").astype(np.int64)
    lengths = ranges[:, 1] - ranges[:, 0]
    for half, name in ((0, "forward"), (1, "backward")):
        r = rec[half][rec[half][:, 0] > 0]
        assert len(r) == 4 * tiles, (name, len(r))                       # one record per quadrant wave of every tile
        assert np.all(r[:, 1] >= r[:, 0]) and (r[:, 1] - r[:, 0]).max() < 100_000_000   # 100 MHz ticks: < 1 s
        tile = (r[:, 3] & np.uint64(0xFFFFFFFF)).astype(np.int64)
        n = (r[:, 2] if half == 0 else (r[:, 2] >> np.uint64(32))).astype(np.int64)
        assert sorted(np.unique(tile).tolist()) == list(range(tiles)) and np.array_equal(n, lengths[tile])
    # a buffer that is too small for the image makes the call fail instead of tracing nothing
    small = torch.zeros(64, dtype=torch.int64, device="cuda")
    _lib.check(_lib.lib.gsr_debug_wave_trace(small.data_ptr(), 64), "gsr_debug_wave_trace")
    try:
        with pytest.raises(RuntimeError, match="gsr_debug_wave_trace"):
            util.hip_forward(cam, g, bg, "sh", debug=True)
    finally:
        _lib.check(_lib.lib.gsr_debug_wave_trace(None, 0), "gsr_debug_wave_trace")
    # switched off: a further frame leaves the buffer alone
    buf.zero_()
    f = util.hip_forward(cam, g, bg, "sh", debug=True)
    util.hip_backward(f, dc, dz, dz, debug=True)
    torch.cuda.synchronize()
    assert int(buf.abs().max()) == 0
