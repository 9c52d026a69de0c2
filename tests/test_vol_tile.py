"""hip.VolTile - the tile order of the correlation volumes (include/vfml.h vfml_corr_lookup, vol_tile) - is host index
arithmetic: checked here without a GPU against the definition in the header."""
import pytest
import torch


@pytest.mark.parametrize("tws,ths", [(2, 3), (3, 2), (0, 0), (0, 1), (4, 1)])
@pytest.mark.parametrize("h,w", [(135, 240), (67, 120), (16, 30), (21, 27), (1, 1)])
def test_tile_positions_follow_the_header_definition(tws, ths, h, w):
    from vfml import hip
    vt = hip.VolTile(tws, ths)
    assert vt.code == tws + 16 * ths
    TW, TH = 1 << tws, 1 << ths
    tiles_x, tiles_y = -(-w // TW), -(-h // TH)
    assert vt.count(h, w) == tiles_x * tiles_y * TW * TH
    pos = vt.position(h, w, "cpu").view(h, w)
    # tile after tile, left to right then down; a tile row-major
    for y, x in [(0, 0), (h - 1, w - 1), (h // 2, w // 3), (min(h - 1, TH), min(w - 1, TW))]:
        want = ((y // TH) * tiles_x + x // TW) * TW * TH + (y % TH) * TW + x % TW
        assert int(pos[y, x]) == want
    flat = pos.reshape(-1)
    assert flat.unique().numel() == h * w and int(flat.max()) < vt.count(h, w)      # injective, inside the whole tiles
    if (tws, ths) == (0, 0):
        assert torch.equal(flat, torch.arange(h * w))                               # no tiles: row-major


def test_rows_permutes_pixels_and_zero_fills_the_edge_tiles():
    from vfml import hip
    vt = hip.VolTile(2, 3)
    h, w, c = 5, 6, 3
    x = torch.arange(h * w * c, dtype=torch.float32)
    t = vt.rows(x, h, w, c).view(-1, c)
    assert t.shape[0] == vt.count(h, w) == 2 * 1 * 32
    pos = vt.position(h, w, "cpu")
    assert torch.equal(t[pos], x.view(h * w, c))
    mask = torch.ones(t.shape[0], dtype=torch.bool)
    mask[pos] = False
    assert torch.count_nonzero(t[mask]) == 0
