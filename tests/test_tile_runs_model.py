"""CPU: host model of the persistent LDS-DMA kernels' work distribution (csrc/conv_h8.hip / conv_x8.hip: block (XCD x, j) of `blocks` walks the
tiles start_x + j, + step, ... of its XCD's contiguous run; a two-stream launch holds stream 0's tiles, then stream 1's) and of the hazard that
profiles/r20_h8_affine_race.md describes: with TWO scale / shift images, the request for tile i + 1's vectors lands in the image of tile i - 1,
so tile i - 1 of a block is at risk exactly when tile i + 1 belongs to the other stream.  The model must reproduce what was measured on the GPU
before the fix - which batches of 640x480 frames differed from run to run, and in which frame - from the launch geometry alone; it documents why
every batch the earlier rounds used (1, 2, 3, 8, 16; 1024x1024 x 8) was safe and pins the analysis behind the three-image fix."""


def xcd_runs(tiles, blocks=256):
    """-> per block: its list of tiles (XCD-aware distribution of conv_h8_kernel: blocks b and b + 8 share an XCD)."""
    blocks = min(blocks, tiles)
    q, r = divmod(tiles, 8)
    out = []
    for bid in range(blocks):
        xcd = bid & 7
        start = xcd * (q + 1) if xcd < r else r * (q + 1) + (xcd - r) * q
        end = start + q + (1 if xcd < r else 0)
        step = (blocks >> 3) + (1 if xcd < (blocks & 7) else 0)
        out.append(list(range(start + (bid >> 3), end, step)))
    return out


def tiles_at_risk(mtiles, ntiles, groups=2):
    """Tiles whose epilogue could read the other stream's vectors with two images: tile i - 1 of a block whose tile i + 1 is in another
    (group, channel tile) - tile index = (group * mtiles + mt) * ntiles + nt."""
    tpg = mtiles * ntiles
    key = lambda t: (t // tpg, t % ntiles)
    risk = []
    for run in xcd_runs(groups * tpg):
        for i in range(1, len(run) - 1):
            if key(run[i + 1]) != key(run[i - 1]):
                risk.append(run[i - 1])
    return sorted(risk)


def frames_at_risk(batch, pixels_per_frame, ntiles, tile_rows=256):
    m = batch * pixels_per_frame
    mtiles = -(-m // tile_rows)
    tpg = mtiles * ntiles
    return sorted({((t % tpg) // ntiles) * tile_rows // pixels_per_frame for t in tiles_at_risk(mtiles, ntiles)})


RES3 = 60 * 80          # res3 maps of a 640x480 frame


def test_fp16_launches_of_the_res3_projection_block():
    # conv3 + shortcut as one GEMM: 512 channels = 2 channel tiles of 256; res3.0.conv1: 128 channels = one tile (conv_h8n_kernel);
    # launches below 224 tiles stay on conv_igemm.hip (key 32)
    observed = {8: [], 9: [7], 10: [], 11: [9], 12: [10], 13: [], 14: [10, 12], 15: [11], 16: []}      # profiles/r20_h8_affine_race.md
    for b, frames in observed.items():
        got = set(frames_at_risk(b, RES3, 2)) | set(frames_at_risk(b, RES3, 1))
        assert sorted(got) == frames, (b, sorted(got))


def test_bf16x3_launch_of_res3_conv1():
    # conv_x8.hip: 256 x 128 tiles; the 128-channel launch is taken from two rounds of tiles on (key 36): 14 frames and more;
    # the 512-channel launches have 4 channel tiles - 8 x (pixel tiles) tiles, every XCD run inside one stream
    observed = {9: [], 11: [], 13: [], 14: [10], 15: [11], 16: []}                                      # profiles/r20_x8_repeat_probe.txt
    for b, frames in observed.items():
        mtiles = -(-b * RES3 // 256)
        got = frames_at_risk(b, RES3, 1) if 2 * mtiles >= 2 * 256 else []
        assert got == frames, (b, got)
        assert frames_at_risk(b, RES3, 4) == []


def test_batches_and_sizes_of_the_earlier_rounds_were_safe():
    # what the fp16 / bf16x3 tests, benches and profiles of rounds 3-6 ran: 640x480 x 1, 2, 3, 8, 16 and 1024x1024 x 1, 2, 4, 8
    for pixels, batches in ((RES3, (1, 2, 3, 8, 16)), (128 * 128, (1, 2, 4, 8))):
        for b in batches:
            for ntiles in (1, 2):
                mtiles = -(-b * pixels // 256)
                if 2 * mtiles * ntiles >= 224:
                    assert frames_at_risk(b, pixels, ntiles) == [], (pixels, b, ntiles)
    # ... while four frames of 1280x720 (res3 maps of 90 x 160: 57 600 rows, the geometry of twelve 640x480 frames) were not
    assert frames_at_risk(4, 90 * 160, 2) == [3]
