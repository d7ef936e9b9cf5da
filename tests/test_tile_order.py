"""The blend kernels' workgroup -> tile map (csrc/cugs_common.h: cugs_xcd_remap + cugs_blend_tile) must be a bijection
on the tiles of ANY image size - a tile visited twice or never is a wrong image - and must spread every 16 consecutive
tile rows over all eight XCDs.  The C functions are restated here in Python from the header's arithmetic (CPU only; the
GPU parity tests cover the kernels themselves on many image sizes)."""
import re
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _row_group():
    text = open(os.path.join(ROOT, "cuda-gaussian-splatting_amd", "csrc", "cugs_common.h")).read()
    return int(re.search(r"#define CUGS_ROW_GROUP (\d+)", text).group(1))


def xcd_remap(bid, nwg):
    xcd, q, r = bid & 7, nwg >> 3, nwg & 7
    base = xcd * (q + 1) if xcd < r else r * (q + 1) + (xcd - r) * q
    return base + (bid >> 3)


def blend_tile(bid, ntx, nty, G):
    lin = xcd_remap(bid, ntx * nty)
    if G == 0:
        return lin
    row, col = divmod(lin, ntx)
    nunits = (nty + G - 1) // G
    c, urow = 0, row
    for _ in range(7):
        units = (nunits + 7 - c) >> 3
        last_in = 1 if (units > 0 and ((nunits - 1) & 7) == c) else 0
        rows_c = units * G - last_in * (nunits * G - nty)
        if urow >= rows_c:
            urow -= rows_c
            c += 1
    unit = c + 8 * (urow // G)
    return (unit * G + urow % G) * ntx + col


def test_tile_map_is_a_bijection_for_every_image_size():
    G = _row_group()
    for ntx in list(range(1, 24)) + [100, 120, 257]:
        for nty in list(range(1, 40)) + [67, 68, 256]:
            n = ntx * nty
            seen = sorted(blend_tile(b, ntx, nty, G) for b in range(n))
            assert seen == list(range(n)), (ntx, nty)


def test_sixteen_consecutive_tile_rows_reach_every_xcd():
    G = _row_group()
    if G == 0:
        return
    ntx, nty = 120, 68                                   # 1920 x 1080
    owner = {}
    for b in range(ntx * nty):
        owner.setdefault(blend_tile(b, ntx, nty, G) // ntx, set()).add(b & 7)      # workgroup b runs on XCD b % 8
    for r0 in range(0, nty - 8 * G):
        xcds = set().union(*(owner[r] for r in range(r0, r0 + 8 * G)))
        assert len(xcds) == 8, r0
